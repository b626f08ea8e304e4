"""roctx ranges around the phases of a step (tracing): forward / CTC / backward / optimizer show up as
named ranges in `rocprofv3 --marker-trace` output next to the kernel rows.  libroctx64 ships with ROCm; when it cannot be
loaded the ranges are no-ops (tracing is an aid, not part of the path)."""
import contextlib
import ctypes

try:
    _roctx = ctypes.CDLL("libroctx64.so")
    _roctx.roctxRangePushA.argtypes = [ctypes.c_char_p]
    _roctx.roctxRangePushA.restype = ctypes.c_int
    _roctx.roctxRangePop.restype = ctypes.c_int
except OSError:      # pragma: no cover - ROCm images always carry it
    _roctx = None


def push(name: str) -> None:
    if _roctx is not None:
        _roctx.roctxRangePushA(name.encode())


def pop() -> None:
    if _roctx is not None:
        _roctx.roctxRangePop()


@contextlib.contextmanager
def range_(name: str):
    push(name)
    try:
        yield
    finally:
        pop()
