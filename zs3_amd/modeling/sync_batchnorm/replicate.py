"""patch_replication_callback (zs3/modeling/sync_batchnorm/replicate.py:45-68).  The reference patches DataParallel.replicate so
that the replicas of one process find their SyncBN master.  Here one process drives one GPU and a DataParallel over a single
device never replicates, so what the call MEANS -- "this model is about to be trained on several GPUs" -- is what it does: every
reference script makes it right after wrapping the model (train_pascal.py:90-93, train_pascal_GMMN.py, train_context*.py), on
every rank, before the first iteration, which makes it the natural collective point to arm the one-process-per-GPU data
parallelism (zs3_amd.parallel.ensure_data_parallel: rank 0's parameters everywhere, the SyncBN communicator, and -- for the
supervised scripts, which run the model's full forward in training mode -- the bucketed gradient all-reduce, installed by
DeepLab.forward itself).  Without torch.distributed (or with one rank) it only validates its argument, like before."""
from torch.nn.parallel import DataParallel


def patch_replication_callback(data_parallel):
    assert isinstance(data_parallel, DataParallel)
    ids = list(getattr(data_parallel, "device_ids", None) or [])
    if len(ids) > 1:
        # the reference's own invocation `--gpu-ids 0,1` (train_pascal.py:88-93) builds DataParallel(model, device_ids=[0, 1]):
        # single-process replication would push modules that hold per-device weight planes, side streams and raw device pointers
        # through DataParallel.replicate / scatter -- undefined behaviour.  Say what to run instead.
        raise RuntimeError(
            f"zs3_amd drives one MI355X per process: DataParallel over device_ids={ids} is not supported.  Launch one process per "
            "GPU instead -- `torchrun --nproc-per-node N --master-addr 127.0.0.1 train_*.py ...` with "
            "torch.distributed.init_process_group('nccl') and torch.cuda.set_device(LOCAL_RANK) at the top of the script, and wrap "
            "the model as DataParallel(model, device_ids=[LOCAL_RANK]) (or leave the wrapper out): gradients, SyncBN statistics "
            "and the loss normalisation then go over RCCL by construction (INTEGRATION.md).")
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        from ... import parallel
        parallel.broadcast_parameters(data_parallel.module)   # also creates the SyncBN communicator, collectively
        object.__setattr__(data_parallel.module, "_zs3_broadcast_done", True)
    return None
