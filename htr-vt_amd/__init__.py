"""htrvt_amd -- MI355X-native (gfx950) kernels and host glue for the HTR-VT
forward / training hot path.  The directory is named `htr-vt_amd`; import it as
`htrvt_amd` (see /htrvt_amd.py at the repo root)."""
import os as _os

# The training step runs on several streams (main, weight gradients, weight packs, gradient collectives + RCCL's own):
# with HIP's default of 4 hardware queues per process two of them can share a queue and serialise (measured: the
# weight-gradient stream behind the main stream in a data-parallel run, +2.8 ms per step).  Takes effect when the HIP
# runtime has not started yet, i.e. when this package is imported before the first CUDA call; an explicit setting wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from . import _lib  # noqa: F401,E402  (raises loudly when libhtrvt_hip.so is missing)
from .ctc import ctc_forward_backward, ctc_loss, greedy_decode  # noqa: F401,E402
from .data import prepare_lines  # noqa: F401,E402
from .ema import ModelEma  # noqa: F401,E402
from .engine import Engine, ModelShape  # noqa: F401,E402


def mark_weights_dirty(model):
    """Tell a model's engines that its parameters / buffers were rewritten through raw device pointers (htrvt_adamw,
    htrvt_sam_first_step / _restore, htrvt_ema_update ...): such writes bump neither `_version` nor `data_ptr()`, which
    is what the packed-weight cache keys on.  Every raw-pointer writer in this package calls this."""
    for eng in getattr(model, "_engines", {}).values():
        eng.weights_epoch += 1


def create_model(nb_cls, img_size, **kwargs):
    from .model import HTR_VT
    return HTR_VT.create_model(nb_cls, img_size, **kwargs)
