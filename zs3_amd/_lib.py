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


def _declare(handle):
    """argtypes / restype of every entry point, parsed from include/zs3hip.h: ctypes then converts plain Python ints,
    floats and None itself (about half the host time of a call that wraps 30 arguments in c_int / c_void_p objects --
    ~1000 launches per training step), and a call with the wrong number of arguments raises instead of corrupting the stack."""
    import re
    header = os.path.join(os.path.dirname(_HERE), "include", "zs3hip.h")
    if not os.path.exists(header):
        # without declared argtypes ctypes would pass 64-bit device pointers and stream handles as 32-bit C ints
        raise Zs3HipError(f"{header} not found: the ctypes signatures of libzs3hip.so are derived from it")
    text = re.sub(r"/\*.*?\*/", " ", open(header).read(), flags=re.S)
    scalar = {"int": ctypes.c_int, "long": ctypes.c_long, "float": ctypes.c_float, "double": ctypes.c_double,
              "unsigned long long": ctypes.c_ulonglong, "unsigned": ctypes.c_uint}
    declared, missing = set(), []
    for ret, name, args in re.findall(r"\b(int|long)\s+(zs3_\w+)\s*\(([^)]*)\)\s*;", text):
        fn = getattr(handle, name, None)
        if fn is None:
            missing.append(name)
            continue
        types = []
        for a in [x.strip() for x in args.split(",")]:
            if a in ("void", ""):
                continue
            if "*" in a:
                types.append(ctypes.c_void_p)
                continue
            base = re.sub(r"\bconst\b", "", a).split()
            base = " ".join(base[:-1]) if len(base) > 1 else base[0]     # drop the parameter name
            types.append(scalar[base])
        fn.argtypes = types
        fn.restype = scalar[ret]
        declared.add(name)
    handle._zs3_declared = declared
    if missing:   # a library built from another header (ZS3_LIB A/B builds): its calls would be marshalled wrongly
        raise Zs3HipError(f"{LIB_PATH} does not export {', '.join(missing[:5])} declared in {header}: header and library differ")


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise Zs3HipError(
                f"{LIB_PATH} not found: build it with `python -m zs3_amd.build` (needs hipcc). "
                "zs3_amd has no CPU/eager fallback."
            )
        _lib = ctypes.CDLL(LIB_PATH)
        _declare(_lib)
    return _lib


def P(t):
    """Device pointer of a tensor (or NULL for None) as a plain integer: the declared argtypes do the conversion."""
    return None if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def stream():
    """hipStream_t of torch's current stream on the current device, as an integer.  Asked ~1000 times per training step: the raw
    accessors cost 0.3 us, `torch.cuda.current_stream().cuda_stream` 10 us (3.4 + 3 ms of host time per step, forward + backward)."""
    if _raw_stream is not None and _raw_device is not None:
        return _raw_stream(_raw_device())
    return torch.cuda.current_stream().cuda_stream


def check(rc, what):
    if rc != 0:
        raise Zs3HipError(f"{what} failed with code {rc}")


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise Zs3HipError("zs3_amd ops run only on MI355X tensors (got a CPU tensor); there is no CPU fallback")


F = float    # scalar arguments travel as plain Python numbers (argtypes declared from the header convert them)
I = int
