from .batchnorm import SynchronizedBatchNorm2d  # noqa: F401
from .replicate import patch_replication_callback  # noqa: F401
