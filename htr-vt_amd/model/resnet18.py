"""Parameter containers of the CNN stem, same module tree / state_dict keys as the
reference `model/resnet18.py` (/root/reference/model_v1/model/resnet18.py:10-84).

These modules only OWN the parameters (so `state_dict()`, `load_state_dict(strict=True)`,
`deepcopy` for EMA and the default PyTorch initialisation behave exactly as in the
reference); the arithmetic of the stem runs in the HIP kernels driven by
`htrvt_amd.engine.Engine` -- calling `.forward` on them is an error on purpose."""
import torch.nn as nn


def _no_eager(*_a, **_k):
    raise RuntimeError("the stem executes in libhtrvt_hip.so through MaskedAutoencoderViT.forward; "
                       "there is no eager PyTorch path")


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes, eps=1e-05)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=1, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes, eps=1e-05)
        self.downsample = downsample
        self.stride = stride

    forward = _no_eager


class ResNet18(nn.Module):
    def __init__(self, nb_feat=384):
        super().__init__()
        c1 = nb_feat // 4
        self.conv1 = nn.Conv2d(1, c1, kernel_size=3, stride=(2, 1), padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(c1, eps=1e-05)
        inpl = c1
        for li, (planes, stride) in enumerate(((nb_feat // 4, (2, 1)), (nb_feat // 2, 2), (nb_feat, 2)), start=1):
            ds = nn.Sequential(nn.Conv2d(inpl, planes, kernel_size=1, stride=stride, bias=False),
                               nn.BatchNorm2d(planes, eps=1e-05))
            setattr(self, f"layer{li}", nn.Sequential(BasicBlock(inpl, planes, stride, ds), BasicBlock(planes, planes, 1, None)))
            inpl = planes

    forward = _no_eager
