"""SynchronizedBatchNorm2d with the reference's name and state-dict layout
(zs3/modeling/sync_batchnorm/batchnorm.py:145).

One process drives one GPU here, so there are no DataParallel replicas to synchronise: like the
vendored module when it is not replicated (batchnorm.py:48-58) this is plain batch-norm arithmetic,
invstd = 1/sqrt(var + eps).  Cross-rank statistics (torch.distributed / RCCL) are switched on by
zs3_amd.parallel.enable_sync_bn(model, process_group)."""
from ..layers import BatchNorm2d


class SynchronizedBatchNorm2d(BatchNorm2d):
    pass
