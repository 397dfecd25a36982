"""zs3_amd -- MI355X-native (gfx950) implementation of the ZS3 data-parallel training hot path.

Host side: Python on PyTorch-ROCm mirroring the reference's module interface (zs3.modeling.deeplab.DeepLab,
zs3.modeling.gmmn.GMMNnetwork, zs3.utils.loss, zs3.base_trainer).  Compute: hand-written HIP kernels in
libzs3hip.so behind a C ABI (include/zs3hip.h), bound with ctypes in zs3_amd._lib.  There is no CPU or
eager fallback: every op raises if the library is missing or a tensor is not on the GPU.
"""
import torch  # noqa: F401  (must be imported before libzs3hip.so so both share torch's HIP runtime)

__version__ = "0.1.0"
