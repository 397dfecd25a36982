"""ctypes binding of libzs3hip.so (include/zs3hip.h).  torch is imported first on purpose: the library
then resolves libamdhip64.so.7 to the runtime torch already loaded, so torch's device pointers and
streams are native to it."""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ZS3_LIB") or os.path.join(_HERE, "lib", "libzs3hip.so")   # ZS3_LIB: A/B builds
_lib = None


class Zs3HipError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise Zs3HipError(
                f"{LIB_PATH} not found: build it with `python -m zs3_amd.build` (needs hipcc). "
                "zs3_amd has no CPU/eager fallback."
            )
        _lib = ctypes.CDLL(LIB_PATH)
    return _lib


def P(t):
    """Device pointer of a tensor (or NULL for None)."""
    if t is None:
        return ctypes.c_void_p(0)
    return ctypes.c_void_p(t.data_ptr())


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def check(rc, what):
    if rc != 0:
        raise Zs3HipError(f"{what} failed with code {rc}")


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise Zs3HipError("zs3_amd ops run only on MI355X tensors (got a CPU tensor); there is no CPU fallback")


F = ctypes.c_float
I = ctypes.c_int
