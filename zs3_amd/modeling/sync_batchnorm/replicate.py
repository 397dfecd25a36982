"""patch_replication_callback (zs3/modeling/sync_batchnorm/replicate.py:45-68).  The reference patches
DataParallel.replicate so that replicas find their SyncBN master; with one process per GPU a
DataParallel over a single device never replicates, so the call only validates its argument."""
from torch.nn.parallel import DataParallel


def patch_replication_callback(data_parallel):
    assert isinstance(data_parallel, DataParallel)
    return None
