"""MI355X-native HTR-VT model behind the reference's Python API.

Drop-in for /root/reference/model_v1/model/HTR_VT.py:
    create_model(nb_cls, img_size, **kwargs) -> nn.Module            (HTR_VT.py:244-254)
    module(x, mask_ratio=0.0, max_span_length=1, use_masking=False)  (HTR_VT.py:222-241)
        x: [B,1,H,W] float32 (or uint8 grey levels, read as value / 255) -> logits [B,N,nb_cls] float32
Same module tree, parameter names/shapes (150-tensor state_dict at d768) and the
same construction order, so `torch.manual_seed(s); create_model(...)` yields the
reference's initial weights and `load_state_dict(strict=True)` of a reference
checkpoint works.  The forward/backward arithmetic is one autograd node that
enqueues the gfx950 kernels of libhtrvt_hip.so (htrvt_amd.engine.Engine); there
is no eager PyTorch fallback.
"""
from functools import partial

import numpy as np
import torch
import torch.nn as nn

try:                                    # `from model import HTR_VT` (reference layout, htr-vt_amd on sys.path)
    from model import resnet18
except ImportError:                     # `from htrvt_amd.model import HTR_VT`
    from . import resnet18

import htrvt_amd                        # noqa: F401  (loads libhtrvt_hip.so or raises)
from htrvt_amd.engine import Engine, ModelShape


def _no_eager(*_a, **_k):
    raise RuntimeError("this sub-module only owns parameters; run the whole model (MaskedAutoencoderViT.forward)")


class Mlp(nn.Module):
    """parameter container with timm Mlp's names (fc1, fc2); timm==1.0.9 is what the reference imports"""

    def __init__(self, in_features, hidden_features):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.fc2 = nn.Linear(hidden_features, in_features)

    forward = _no_eager


class Attention(nn.Module):
    def __init__(self, dim, num_patches, num_heads=8, qkv_bias=False):
        super().__init__()
        assert dim % num_heads == 0, 'dim should be divisible by num_heads'
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.num_patches = num_patches
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)

    forward = _no_eager


class Block(nn.Module):
    def __init__(self, dim, num_heads, num_patches, mlp_ratio=4., qkv_bias=False, norm_layer=nn.LayerNorm):
        super().__init__()
        self.norm1 = norm_layer(dim, elementwise_affine=True)
        self.attn = Attention(dim, num_patches, num_heads=num_heads, qkv_bias=qkv_bias)
        self.norm2 = norm_layer(dim, elementwise_affine=True)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))

    forward = _no_eager


def get_2d_sincos_pos_embed(embed_dim, grid_size):
    """[gh*gw, D] float64 table; token t=(h,w) row-major, first D/2 dims encode w, last D/2 encode h
    (restates HTR_VT.py:86-131)."""
    gh, gw = int(grid_size[0]), int(grid_size[1])
    omega = 1.0 / 10000 ** (np.arange(embed_dim // 4, dtype=np.float64) / (embed_dim / 4.0))
    t = np.arange(gh * gw)
    w = (t % gw).astype(np.float32).astype(np.float64)[:, None] * omega[None, :]
    h = (t // gw).astype(np.float32).astype(np.float64)[:, None] * omega[None, :]
    return np.concatenate([np.sin(w), np.cos(w), np.sin(h), np.cos(h)], axis=1)


class LayerNorm(nn.Module):
    """the reference's parameter-free whitening over everything but the batch dimension (HTR_VT.py:134-136, eps 1e-5):
    `model.layer_norm`.  Inside the model both of its uses are fused into kernels (input: htrvt_img_stats + the first
    convolution; logits: htrvt_seq_whiten_fwd); called on its own it runs the same per-sample whitening kernel."""

    def forward(self, x):
        from htrvt_amd._lib import check, lib
        from htrvt_amd.ops import ptr, stream
        if not x.is_cuda:
            raise RuntimeError("htrvt_amd runs on an MI355X only (no CPU / eager fallback exists)")
        xf = x.detach().contiguous().float()
        B = xf.shape[0]
        y = torch.empty_like(xf)
        stats = torch.empty(B, 2, dtype=torch.float32, device=xf.device)
        check(lib.htrvt_seq_whiten_fwd(ptr(xf), ptr(y), ptr(stats), B, xf.numel() // B, 1e-5, 0, stream()), "seq_whiten_fwd")
        return y


class _HTRVTFunction(torch.autograd.Function):
    """One autograd node for the whole network: forward and backward are HIP kernel sequences."""

    @staticmethod
    def forward(ctx, module, img, keep, train, need, names, *tensors):
        eng = module._engine(img.device)
        P = dict(zip(names, tensors))
        # need: grad mode was on at the call and some parameter requires grad (decided by the caller: inside forward() grad
        # mode is always off, and ctx.needs_input_grad stays True under torch.no_grad() -- a no_grad validation pass of
        # the training model would otherwise keep 13 GB of activations and miss the fused eval path)
        y = eng.forward(P, img, keep_mask=keep, train=train, save=need)
        if need:
            ctx.saved_acts, eng.saved = eng.saved, None
            ctx.eng, ctx.names, ctx.P = eng, names, P
        else:
            ctx.mark_non_differentiable(y)     # nothing was kept for a backward: autograd must not ask for one
        return y

    @staticmethod
    def backward(ctx, dy):
        eng, names, P = ctx.eng, ctx.names, ctx.P
        G = {n: torch.zeros_like(t) for n, t in P.items() if t.requires_grad}
        eng.saved = ctx.saved_acts
        eng.backward(P, G, dy.contiguous().float())
        ctx.saved_acts = None
        return (None, None, None, None, None, None) + tuple(G.get(n) for n in names)


class MaskedAutoencoderViT(nn.Module):
    """HTR-VT encoder (the reference keeps the MAE class name)."""

    def __init__(self, nb_cls=80, img_size=[512, 32], patch_size=[8, 32], embed_dim=1024, depth=24, num_heads=16,
                 mlp_ratio=4., norm_layer=nn.LayerNorm, compute_dtype=torch.float32):
        super().__init__()
        self.layer_norm = LayerNorm()           # parameter-free (HTR_VT.py:134-136,157): no state_dict entry
        self.patch_embed = resnet18.ResNet18(embed_dim)
        self.grid_size = [img_size[0] // patch_size[0], img_size[1] // patch_size[1]]
        self.embed_dim = embed_dim
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.mask_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, self.num_patches, embed_dim), requires_grad=False)
        self.blocks = nn.ModuleList([Block(embed_dim, num_heads, self.num_patches, mlp_ratio, qkv_bias=True,
                                           norm_layer=norm_layer) for _ in range(depth)])
        self.norm = norm_layer(embed_dim, elementwise_affine=True)
        self.head = torch.nn.Linear(embed_dim, nb_cls)
        self.initialize_weights()
        eps = {m.eps for m in self.modules() if isinstance(m, nn.LayerNorm)}
        assert len(eps) == 1, f"one LayerNorm eps per model expected, got {eps}"
        self._shape = ModelShape(nb_cls, img_size, embed_dim, depth, num_heads, mlp_ratio, patch_size, ln_eps=eps.pop())
        self.compute_dtype = compute_dtype          # torch.float32 (parity), "split_bf16" (parity on the bf16 matrix cores) or torch.bfloat16 (throughput)
        self._engines = {}

    def initialize_weights(self):
        pe = get_2d_sincos_pos_embed(self.embed_dim, self.grid_size)
        self.pos_embed.data.copy_(torch.from_numpy(pe).float().unsqueeze(0))
        torch.nn.init.normal_(self.mask_token, std=.02)
        self.apply(self._init_weights)

    def _init_weights(self, m):
        if isinstance(m, nn.Linear):            # xavier-uniform weights, zero bias (HTR_VT.py:192-197)
            torch.nn.init.xavier_uniform_(m.weight)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)

    def __deepcopy__(self, memo):               # utils.ModelEma deep-copies the model: engines hold device scratch only
        import copy
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            new.__dict__[k] = {} if k == "_engines" else copy.deepcopy(v, memo)
        return new

    def _engine(self, device):
        key = (str(device), self.compute_dtype)
        if key not in self._engines:
            if self.compute_dtype == "split_bf16":      # float32 activations, hi + lo bf16 operands on the matrix cores (csrc/split.hip)
                self._engines[key] = Engine(self._shape, torch.float32, device, split_bf16=True)
            else:
                self._engines[key] = Engine(self._shape, self.compute_dtype, device)
        return self._engines[key]

    def generate_span_mask(self, x, mask_ratio, max_span_length):
        """HTR_VT.py:202-210, same CPU-RNG draws.  x: the reference's [N, L, D] token tensor -> keep-mask [N, L, 1] on
        x.device (1 keep / 0 mask, the same spans for every sample); or the sequence length L as an int -> the [L] mask
        this model's forward hands to the token kernel."""
        L = int(x) if isinstance(x, int) else x.shape[1]
        mask = torch.ones(L)
        for _ in range(int(L * mask_ratio) // max_span_length):
            idx = int(torch.randint(L - max_span_length, (1,)))
            mask[idx:idx + max_span_length] = 0
        if isinstance(x, int):
            return mask
        return mask.view(1, L, 1).expand(x.shape[0], L, 1).contiguous().to(x.device)

    def random_masking(self, x, mask_ratio, max_span_length):
        """HTR_VT.py:212-220 for callers that hold a token tensor [N, L, D] themselves (the forks' heads do): the model's
        own forward applies the mask inside htrvt_pool_tokens instead"""
        mask = self.generate_span_mask(x, mask_ratio, max_span_length)
        return x * mask + (1 - mask) * self.mask_token

    def forward(self, x, mask_ratio=0.0, max_span_length=1, use_masking=False, keep_mask=None):
        if not x.is_cuda:
            raise RuntimeError("htrvt_amd runs on an MI355X only: move the model and the input to cuda "
                               "(no CPU / eager fallback exists)")
        if keep_mask is None and use_masking:
            keep_mask = self.generate_span_mask(self.num_patches, mask_ratio, max_span_length)
        names, tensors = [], []
        for n, t in self.state_dict(keep_vars=True).items():
            names.append(n)
            tensors.append(t)
        # uint8 images (the data pipeline's raw grey levels) stay uint8: the first kernels read them as value / 255
        x = x.contiguous() if x.dtype == torch.uint8 else x.contiguous().float()
        if x.requires_grad and torch.is_grad_enabled():
            raise RuntimeError("htrvt_amd computes no gradient with respect to the input image (d loss / d image): "
                               "detach the image, or call under torch.no_grad()")
        need = torch.is_grad_enabled() and any(t.requires_grad for t in tensors)
        return _HTRVTFunction.apply(self, x, keep_mask, self.training, need, tuple(names), *tensors)


def create_model(nb_cls, img_size, **kwargs):
    return MaskedAutoencoderViT(nb_cls, img_size=img_size, patch_size=(4, 64), embed_dim=768, depth=4, num_heads=6,
                                mlp_ratio=4, norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)
