"""SynchronizedBatchNorm2d with the reference's name and state-dict layout
(zs3/modeling/sync_batchnorm/batchnorm.py:145).

The reference synchronises the replicas of one nn.DataParallel process (batchnorm.py:46-89: per-replica sum / sum of squares
gathered by the master, mean / inv-std broadcast back; train_pascal.py:279 switches it on for more than one GPU).  Here one
process drives one GPU, and the module synchronises BY CONSTRUCTION whenever there is something to synchronise with: as soon
as torch.distributed is initialised with more than one rank, every forward in training mode all-reduces its per-channel
partial sums (one fp64 collective of 2C+1 values per layer, zs3_amd.parallel.combine_bn_partials) and every backward the two
gradient sums -- no extra call in the training script.  With one rank (or no process group) it is plain batch-norm
arithmetic, like the vendored module when it is not replicated (batchnorm.py:48-58).

Deviation kept from round 1 and documented in DESIGN.md: invstd = 1/sqrt(var + eps) as in F.batch_norm, not the vendored
clamp(var, eps)^-1/2 (batchnorm.py:142), so that one rank and N ranks compute the same function of the global batch.

`sync_enabled = False` (zs3_amd.parallel.enable_sync_bn(model, enabled=False), bench.py --sync-bn 0) turns the exchange off;
`sync_group` selects a process group other than the default one."""
import torch.distributed as dist

from ..layers import BatchNorm2d


class SynchronizedBatchNorm2d(BatchNorm2d):
    sync_enabled = True
    sync_group = None
    _sync_cache = None   # (state the answer was computed under, answer); enable_sync_bn resets it

    def __getstate__(self):
        """copy.deepcopy / torch.save of a module that has trained under torch.distributed: the cached answer holds a
        ProcessGroup (not picklable, not copyable) and is recomputed on first use anyway"""
        state = self.__dict__.copy()
        state.pop("_sync_cache", None)
        if not isinstance(state.get("sync_group"), (type(None), bool)):
            state.pop("sync_group", None)      # an explicit process group does not survive a copy: the class default (None) returns
        return state

    @property
    def _zs3_sync_group(self):
        """What zs3_amd.functional reads -- in training mode only, so an eval forward never touches torch.distributed:
        None = local statistics, True = the default process group, or a group.  The answer is cached per module and recomputed
        when the process-group state changes (113 layers x every forward would otherwise each ask torch.distributed)."""
        from ... import parallel
        state = (self.sync_enabled, parallel.FORCE_COLLECTIVES, dist.is_available() and dist.is_initialized())
        hit = self._sync_cache
        if hit is not None and hit[0] == state:
            return hit[1]
        if not self.sync_enabled:
            ans = None
        elif parallel.FORCE_COLLECTIVES:          # single-rank plumbing tests run the whole reduction path
            ans = self.sync_group if self.sync_group is not None else True
        elif not state[2] or dist.get_world_size(self.sync_group) < 2:
            ans = None
        else:
            ans = self.sync_group if self.sync_group is not None else parallel.bn_group()
        self._sync_cache = (state, ans)
        return ans
