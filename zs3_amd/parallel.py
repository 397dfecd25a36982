"""Data parallelism for one-process-per-GPU runs: RCCL (torch.distributed backend "nccl" on ROCm) over xGMI.

The reference's multi-GPU path is single-process nn.DataParallel + a thread-based SyncBN
(train_pascal.py:90-93, sync_batchnorm/*: collectives C1-C4 of SURVEY.md section 2.2).  Here every rank owns
one MI355X and a shard of the batch; the only data-path exchange of the supervised step is the gradient
all-reduce (237 MB fp32), issued per bucket from grad hooks while backward is still running -- RCCL
executes on its own HIP stream, so the exchange overlaps with the remaining backward kernels -- and
joined by an end-of-backward callback.  Reduction is SUM: the loss is already normalised by the *global*
batch and valid-pixel weight (utils/loss.py, `group=`), which reproduces the reference's single-process
loss exactly.  Works unchanged on CPU tensors with the gloo backend (tests/test_distributed_cpu.py).
"""
import torch
import torch.distributed as dist


# single-GPU plumbing tests set this to run the SyncBN reduction path with one rank
FORCE_COLLECTIVES = False

# ---------------------------------------------------------------------------------- RCCL under the C ABI (round 6)
# With the "nccl" (= RCCL) backend the data-path collectives of the step -- SyncBN sums, gradient buckets, the CE's weight sums, the
# range flag -- are issued by libzs3hip.so itself (csrc/comm.hip: zs3_allreduce / zs3_bn_sync_exchange) ON THE STREAM THAT HOLDS THE
# DATA: the SyncBN exchange is pack -> all-reduce -> finalize in the compute stream's own order (torch.distributed hands every
# collective to RCCL's stream and back through two events: ~10 us of bubble on the dependent chain, 208 times per step), and every
# collective is an entry point a launch plan records.  One communicator per issuing stream (a communicator's operations must not
# run concurrently with each other: the main stream, the two ASPP lanes and the weight-gradient stream each get their own), created
# collectively at first use -- every rank reaches the same call sites in the same order.  torch.distributed keeps rendezvous, the
# one-time parameter broadcast and the gloo / CPU paths.  ZS3_NATIVE_RCCL=0: everything through torch.distributed as in round 5.
import os as _os

NATIVE_RCCL = _os.environ.get("ZS3_NATIVE_RCCL", "1") == "1"
_native_comms = {}
_native_loaded = [None]


_native_backend = {}      # id(process group) -> is its backend RCCL?  (asked ~430 times per step: dist.get_backend costs 20 us)


def native_available(group=None):
    """can the library issue this group's collectives itself?  (RCCL backend, library loadable, and -- over several ranks -- the
    first communicator's all-reduce came back with the right sum on every rank: native_probe).  COLLECTIVE at the first call per
    process group when it has more than one rank."""
    if not (NATIVE_RCCL and dist.is_initialized()):
        return False
    pg = None if group is True else group
    key = (id(pg), id(dist.group.WORLD))      # (a destroyed and re-created default group is another object)
    hit = _native_backend.get(key)
    if hit is None:
        try:
            hit = torch.cuda.is_available() and "nccl" in str(dist.get_backend(pg))
        except Exception:
            hit = False
        if hit and _native_loaded[0] is None:
            from ._lib import lib
            path = _os.path.join(_os.path.dirname(torch.__file__), "lib", "librccl.so")
            _native_loaded[0] = lib().zs3_comm_load(path.encode() if _os.path.exists(path) else b"") == 0
        hit = bool(hit and _native_loaded[0])
        if hit and dist.get_world_size(pg) > 1:
            hit = native_probe(pg)
        _native_backend.clear()
        _native_backend[key] = hit
    return hit


PROBE_TIMEOUT_S = float(_os.environ.get("ZS3_NATIVE_PROBE_TIMEOUT", "60"))


def native_probe(group=None):
    """One all-reduce through the library's own communicator of the current stream, checked on every rank: rank r contributes r + 1,
    every rank must read world * (world + 1) / 2.  A communicator that cannot be created, a wrong sum, or a collective that does not
    come back within PROBE_TIMEOUT_S (the communicator is aborted) on ANY rank (the verdicts are MIN-reduced through
    torch.distributed) sends the whole job to the torch.distributed collectives, with one warning -- the binding in csrc/comm.hip is
    the one part of the step that a one-GPU box cannot exercise with real peers, so the first multi-rank run checks it itself."""
    import time
    import warnings
    from ._lib import lib, stream
    pg = None if group is True else group
    rank, world = dist.get_rank(pg), dist.get_world_size(pg)
    st, ok, why, comm = stream(), True, "", 0
    try:
        comm = native_comm(st, pg)
        t = torch.full((4,), float(rank + 1), dtype=torch.float32, device="cuda")
        rc = lib().zs3_allreduce(comm, t.data_ptr(), 4, 0, 0, st)
        if rc != 0:
            ok, why = False, f"zs3_allreduce returned {rc}"
        else:
            ev = torch.cuda.Event()
            ev.record()
            t0 = time.time()
            while not ev.query():
                if time.time() - t0 > PROBE_TIMEOUT_S:
                    ok, why = False, f"the all-reduce did not complete within {PROBE_TIMEOUT_S:.0f} s"
                    lib().zs3_comm_abort(comm)
                    _native_comms.pop((int(st or 0), id(pg) if pg is not None else 0), None)
                    comm = 0
                    break
                time.sleep(0.001)
            if ok and t.tolist() != [world * (world + 1) / 2.0] * 4:
                ok, why = False, f"wrong sum {t.tolist()} over {world} ranks"
    except Exception as e:      # noqa: BLE001 (anything the binding raises: fall back, loudly)
        ok, why = False, repr(e)
    verdict = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cuda")
    dist.all_reduce(verdict, op=dist.ReduceOp.MIN, group=pg)
    if int(verdict.item()) == 1:
        return True
    warnings.warn("zs3_amd: the library's own RCCL communicator failed its first all-reduce"
                  + (f" on this rank ({why})" if why else " on another rank")
                  + "; the step's collectives go through torch.distributed instead (ZS3_NATIVE_RCCL=0 selects that from the start)")
    for key in [k for k in _native_comms if k[1] == (id(pg) if pg is not None else 0)]:
        lib().zs3_comm_destroy(_native_comms.pop(key))
    return False


def native_comm(stream_handle, group=None):
    """The library's communicator for collectives issued on the HIP stream `stream_handle` over the ranks of `group` (default:
    all), created on first use: rank 0 of the group draws the RCCL unique id, torch.distributed carries its 128 bytes to the
    others, every rank calls zs3_comm_create.  COLLECTIVE at first use per (stream, group)."""
    import ctypes
    pg = None if group is True else group
    key = (int(stream_handle or 0), id(pg) if pg is not None else 0)
    comm = _native_comms.get(key)
    if comm is None:
        from ._lib import lib
        n = lib().zs3_comm_unique_id_bytes()
        buf = ctypes.create_string_buffer(n)
        rank, world = dist.get_rank(pg), dist.get_world_size(pg)
        if rank == 0 and lib().zs3_comm_unique_id(buf) != 0:
            raise RuntimeError("zs3_comm_unique_id failed")
        box = [bytes(buf.raw)]
        if world > 1:
            dist.broadcast_object_list(box, src=dist.get_global_rank(pg, 0) if pg is not None else 0, group=pg)
        comm = int(lib().zs3_comm_create(box[0], world, rank))
        if not comm:
            raise RuntimeError("zs3_comm_create failed (RCCL's message is on stderr)")
        _native_comms[key] = comm
    return comm


def native_shutdown():
    """destroy the library's communicators (before torch.distributed.destroy_process_group, or when a process group is replaced)"""
    from ._lib import lib
    for comm in _native_comms.values():
        lib().zs3_comm_destroy(comm)
    _native_comms.clear()
    _native_backend.clear()


def native_allreduce(t, op="sum", group=None):
    """in-place all-reduce of a dense device tensor on the CURRENT stream through the library; False when not applicable"""
    if not (t.is_cuda and native_available(group)):
        return False
    from ._lib import check, lib, stream
    code = {torch.float32: 0, torch.float64: 1, torch.int32: 2, torch.int64: 3}.get(t.dtype)
    if code is None or not t.is_contiguous():
        return False
    st = stream()
    check(lib().zs3_allreduce(native_comm(st, group), t.data_ptr(), t.numel(), code, 0 if op == "sum" else 1, st), "zs3_allreduce")
    return True

AUTO = "auto"


def resolve_group(group):
    """`group` arguments across the package: "auto" (the default everywhere) = the default process group as soon as
    torch.distributed runs with more than one rank, nothing otherwise -- data parallelism by construction, the training script
    names no group; None / False = never exchange; True = the default group; anything else = that process group."""
    if group is None or group is False:
        return None
    if isinstance(group, str):
        if group != AUTO:
            raise ValueError(f"group must be 'auto', None, True or a process group, got {group!r}")
        if FORCE_COLLECTIVES and dist.is_available() and dist.is_initialized():
            return True
        return True if (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1) else None
    return group


def ensure_data_parallel(module, broadcast=True):
    """Arm `module` (a DeepLab) for one-process-per-GPU data parallelism; a no-op unless torch.distributed runs with more than
    one rank.  COLLECTIVE: every rank must call it at the same point -- which is the case for its two callers, the first
    training-mode forward of the model (zs3_amd.modeling.deeplab.DeepLab.forward) and `patch_replication_callback` (the call every
    reference script makes right after wrapping the model, train_pascal.py:92).  Once per module: parameters and buffers are
    made rank 0's (nn.DataParallel.replicate's guarantee, C1 of SURVEY 2.2), the SyncBN communicator is created, and GradSync
    hooks the parameters (bucketed SUM all-reduce from the weight-gradient stream while backward runs).  Returns the GradSync
    or None."""
    sync = getattr(module, "_zs3_grad_sync", None)
    if sync is not None:
        return sync
    if not (dist.is_available() and dist.is_initialized()):
        return None
    if dist.get_world_size() < 2 and not FORCE_COLLECTIVES:
        return None
    from . import functional as Fz
    if any(id(p) in Fz._grad_buffers for p in module.parameters()):
        return None      # the script manages a GradSync over these parameters itself
    if broadcast:
        broadcast_parameters(module)
    bn_group()
    sync = GradSync(list(module.parameters()), force=FORCE_COLLECTIVES and dist.get_world_size() < 2)
    object.__setattr__(module, "_zs3_grad_sync", sync)
    return sync


def disarm_data_parallel(module):
    sync = getattr(module, "_zs3_grad_sync", None)
    if sync is not None:
        sync.remove()
        object.__setattr__(module, "_zs3_grad_sync", None)


class GradSync:
    def __init__(self, params, process_group=None, bucket_mb=64.0, average=False, force=False):
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.force = force   # run the full hook / bucket / all-reduce path even with one rank (single-GPU plumbing test)
        self.average = average
        self.params = [p for p in params if p.requires_grad]
        cap = int(bucket_mb * 1024 * 1024 / 4)
        # buckets follow REVERSE registration order: that is roughly the order in which backward produces grads
        self.buckets, cur, cur_n = [], [], 0
        for p in reversed(self.params):
            if cur and cur_n + p.numel() > cap:
                self.buckets.append(cur)
                cur, cur_n = [], 0
            cur.append(p)
            cur_n += p.numel()
        if cur:
            self.buckets.append(cur)
        self.flat, self.where = [], {}
        for bi, bucket in enumerate(self.buckets):
            n = sum(p.numel() for p in bucket)
            self.flat.append(torch.zeros(n, dtype=bucket[0].dtype, device=bucket[0].device))
            off = 0
            for p in bucket:
                self.where[p] = (bi, off)
                off += p.numel()
        # zero-copy hand-off: the fused layers write a weight gradient straight into its slice of the flat bucket
        # (functional.grad_buffer) and autograd adopts that slice as .grad, so the all-reduce works on the gradients in
        # place -- no pack before and no unpack after the collective for the conv / linear weights (99.9 % of the bytes)
        from . import functional as Fz
        for p in self.params:
            bi, off = self.where[p]
            Fz.register_grad_buffer(p, self.flat[bi][off:off + p.numel()])
        self._pending = [len(b) for b in self.buckets]
        self._ready = [set() for _ in self.buckets]
        self._works = [None] * len(self.buckets)
        self._armed = False
        self.bytes_reduced = 0
        self._handles = [p.register_post_accumulate_grad_hook(self._hook) for p in self.params]

    # ---- hooks
    def _comm_stream(self, p):
        """Stream on which bucket copies and all-reduce launches are issued: the wgrad side stream when it is in use
        (weight gradients are produced there, so no extra wait is needed and the main stream is never blocked),
        otherwise the current stream."""
        if not p.is_cuda:
            return None
        from . import functional as Fz
        if Fz.WGRAD_SIDE_STREAM:
            pool = Fz.wgrad_streams(p.device)
            side = pool[0]
            Fz._wait_for(side, torch.cuda.current_stream(p.device))   # gradients produced on the main stream (BN, bias)
            for other in pool[1:]:                                    # ... and on the other streams of the wgrad pool
                Fz._wait_for(side, other)
            return side
        return None

    def _hook(self, p):
        if self.world == 1 and not self.force:
            return
        if not self._armed:
            self._armed = True
            torch.autograd.Variable._execution_engine.queue_callback(self.finish)
        bi, off = self.where[p]
        if p in self._ready[bi]:
            return
        self._ready[bi].add(p)
        if len(self._ready[bi]) == len(self.buckets[bi]):
            st = self._comm_stream(p)
            ctx = torch.cuda.stream(st) if st is not None else _null()
            with ctx:
                self._launch(bi)

    def _in_place(self, p):
        """p.grad already IS its slice of the flat bucket (written there by the layer's wgrad kernel)"""
        bi, off = self.where[p]
        return p.grad is not None and p.grad.data_ptr() == self.flat[bi].data_ptr() + 4 * off and \
            p.grad.untyped_storage().data_ptr() == self.flat[bi].untyped_storage().data_ptr()

    def _views(self, bi, params):
        return [self.flat[bi][self.where[p][1]:self.where[p][1] + p.numel()] for p in params]

    def _launch(self, bi):
        """pack the bucket's gradients (one multi-tensor copy instead of one kernel per parameter) and start its all-reduce"""
        have = [p for p in self.buckets[bi] if p in self._ready[bi] and p.grad is not None and not self._in_place(p)]
        if have:
            torch._foreach_copy_(self._views(bi, have), [_as_flat(p.grad, p) for p in have])
        if native_allreduce(self.flat[bi], "sum", self.group):      # issued by the library on this (weight-gradient) stream
            self._works[bi] = _NativeWork(torch.cuda.current_stream(self.flat[bi].device))
        else:
            self._works[bi] = dist.all_reduce(self.flat[bi], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self.bytes_reduced += self.flat[bi].numel() * 4

    def finish(self):
        """End of backward: flush incomplete buckets (parameters that received no gradient contribute zeros, so
        every rank issues the same collectives), wait, and scatter the reduced values back into .grad."""
        if self.world == 1 and not self.force:
            return
        anyp = self.params[0]
        st = self._comm_stream(anyp)
        ctx = torch.cuda.stream(st) if st is not None else _null()
        with ctx:
            for bi, bucket in enumerate(self.buckets):
                if self._works[bi] is None and self._ready[bi]:
                    for p in bucket:
                        if p not in self._ready[bi]:
                            _, off = self.where[p]
                            self.flat[bi][off:off + p.numel()].zero_()
                    self._launch(bi)
        for bi, bucket in enumerate(self.buckets):
            if self._works[bi] is None:
                continue
            self._works[bi].wait()   # the *current* (main) stream waits for the collective
            if self.average:
                self.flat[bi].div_(self.world)
            for p in bucket:
                if p.grad is None:
                    p.grad = torch.empty_like(p)
            back = [p for p in bucket if not self._in_place(p)]
            if back:
                torch._foreach_copy_([_as_flat(p.grad, p) for p in back], self._views(bi, back))
            self._works[bi] = None
            self._ready[bi] = set()
        # the f16x3 range flag (ops.range_flag) is per device, the gradients are not: one rank's overflow reaches
        # every rank as NaNs inside the all-reduced buckets, and only the rank whose own flag is up would skip the update.  MAX over
        # the group: every rank skips the step and falls back to bf16x3 forward products together.
        # (Only while the f16x3 forward can be active at all -- fp32 storage, x3 arithmetic, switch on: a condition every rank
        # evaluates alike, and after a fallback, which the ranks make together, the exchange is gone.)
        exchange_range_flag(self.params[0].device, self.group)
        self._armed = False

    def remove(self):
        from . import functional as Fz
        for h in self._handles:
            h.remove()
        for p in self.params:
            Fz.register_grad_buffer(p, None)


def exchange_range_flag(device, group=None):
    """MAX all-reduce of the device's f16x3 range flag over the data-parallel group, when that arithmetic can be active (see
    GradSync.finish; GMMNStep makes the same exchange before its classifier step).  Returns whether it exchanged."""
    from . import ops
    if not ops.fwd_f16() or not dist.is_initialized():
        return False
    flag = ops.range_flag(device)
    if not native_allreduce(flag, "max", group):
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
    return True


class _NativeWork:
    """a collective the library issued on `stream`: wait() makes the current stream wait for it (stream order, no host block)"""

    def __init__(self, stream):
        self.stream = stream

    def wait(self):
        from . import functional as Fz
        Fz._wait_for(torch.cuda.current_stream(self.stream.device), self.stream)


class _null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def _as_flat(t, like):
    """1-D view of a dense tensor in its own memory order (channels_last conv grads included)."""
    if t.is_contiguous():
        return t.view(-1)
    if t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last):
        return t.permute(0, 2, 3, 1).reshape(-1)
    return t.contiguous().view(-1)


def broadcast_parameters(module, src=0, group=None):
    """Make every rank start from rank `src`'s parameters and buffers (DataParallel.replicate, C1)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        if t.is_contiguous() or t.dim() != 4:
            dist.broadcast(t.data, src, group=group)
        else:
            flat = _as_flat(t.data, t).clone()
            dist.broadcast(flat, src, group=group)
            _as_flat(t.data, t).copy_(flat)
    # writes through .data do not bump the autograd version counter the bf16 weight-plane cache is keyed on
    from . import functional as Fz
    Fz.invalidate_planes(*module.parameters())
    if group is None or group is dist.group.WORLD:
        # a point every rank of the WORLD reaches together: create the SyncBN communicator here rather than in some forward.
        # (dist.new_group() is a collective over all ranks: under an explicit sub-group -- GMMNStep(group=pg) -- only that
        # group's ranks are here, and creating it would hang the others.)
        bn_group()


def all_reduce_tensors(tensors, group=None, average=False, force=False):
    """In-place SUM (or mean) all-reduce of a list of dense tensors as ONE collective: packed into a flat buffer with one
    multi-tensor copy, reduced, unpacked with another.  The GMMN step's exchange (generator parameters,
    219,648 fp32 = 0.88 MB, averaged once per outer iteration; `pred_conv` gradients, <= 62 KB, summed) uses this: the
    tensors are far too small for bucketing or overlap to matter, what matters is one launch instead of one per tensor."""
    tensors = [t for t in tensors if t is not None]
    if not tensors or not dist.is_initialized():
        return 0
    world = dist.get_world_size(group)
    if world == 1 and not (force or FORCE_COLLECTIVES):
        return 0
    views = [_as_flat(t, t) for t in tensors]
    flat = torch.empty(sum(v.numel() for v in views), dtype=views[0].dtype, device=views[0].device)
    parts = list(torch.split(flat, [v.numel() for v in views]))
    torch._foreach_copy_(parts, views)
    dist.all_reduce(flat, group=group)
    if average:
        flat.mul_(1.0 / world)
    torch._foreach_copy_(views, parts)
    for t, v in zip(tensors, views):        # a layout _as_flat had to copy: write the result back
        if v.untyped_storage().data_ptr() != t.untyped_storage().data_ptr():
            t.copy_(v.view(t.shape))
    return flat.numel() * flat.element_size()


_bn_group = None


def bn_group():
    """The process group of the SyncBN statistics: a communicator of its own over all ranks (created once, collectively, by the
    first synchronised BatchNorm forward).  The 208 tiny latency-critical all-reduces of a step sit on the main stream's
    dependent chain; on the default group they would queue on the same RCCL stream behind the 60 MB gradient buckets that
    GradSync launches from the weight-gradient stream while backward is still running.  `ZS3_BN_GROUP=0`: the default group."""
    global _bn_group
    import os
    if not dist.is_initialized() or dist.get_world_size() < 2:
        return True
    if _bn_group is None:
        # dist.new_group() is itself a collective: it happens here exactly once, and the callers that reach this first are
        # collective points by contract -- ensure_data_parallel (first training forward / patch_replication_callback) and
        # broadcast_parameters; a synchronised BatchNorm forward only gets here in training mode (eval forwards, e.g. a
        # rank-0-only validation, never touch torch.distributed).  The choice is remembered so later calls cost one test.
        _bn_group = dist.new_group() if os.environ.get("ZS3_BN_GROUP", "1") == "1" else True
    return _bn_group


def combine_bn_partials(partial, count, group=None):
    """SyncBN statistics (batchnorm.py:60-67,101-122 of the vendored module: sum / sum-of-squares reduced over replicas).
    One launch collapses the per-chunk partial sums [chunks,2,C] to fp64 totals and appends the local sample count
    (zs3_bn_sync_pack), ONE fp64 all-reduce carries both, and zs3_bn_*_finalize read the reduced buffer directly -- two kernels
    and a collective per layer and direction, nothing visits the host.  Returns (buffer, None): the fp64 exchange buffer
    [2C + 1] that ops.bn_fwd_finalize / bn_bwd_finalize accept in place of the partial sums (the count travels inside it)."""
    if not FORCE_COLLECTIVES and (not dist.is_initialized() or dist.get_world_size(group) == 1):
        return partial, count
    if not partial.is_cuda:   # gloo tests on CPU tensors: same arithmetic, spelled in torch
        c = partial.shape[2]
        buf = torch.empty(2 * c + 1, dtype=torch.float64, device=partial.device)
        torch.sum(partial.double(), dim=0, out=buf[:2 * c].view(2, c))
        buf[2 * c] = float(count)
    else:
        from . import ops
        if dist.is_initialized() and native_available(group):
            # pack + all-reduce issued by the library on the compute stream itself (zs3_bn_sync_exchange): no hand-over to another stream
            from ._lib import check, lib, stream
            c = partial.shape[2]
            buf = torch.empty(2 * c + 1, dtype=torch.float64, device=partial.device)
            st = stream()
            check(lib().zs3_bn_sync_exchange(native_comm(st, group), partial.data_ptr(), partial.shape[0], c, float(count),
                                             buf.data_ptr(), st), "zs3_bn_sync_exchange")
            return buf, None
        buf = ops.bn_sync_pack(partial, count)
    if dist.is_initialized():
        dist.all_reduce(buf, group=group)
    return buf, None


def combine_bn_partials_bwd(partial, count, group=None, out=(None, None)):
    """The backward exchange of SyncBN: -> (exchange buffer, None, dgamma, dbeta).  With the library's own collectives this rank's
    dgamma / dbeta (its local sums as they stand) are written by the pack launch itself into `out` (gradient-bucket slices) or fresh
    vectors; otherwise dgamma = dbeta = None and the caller runs the per-rank finalize as before."""
    if partial.is_cuda and dist.is_initialized() and native_available(group) and partial.dtype == torch.float32:
        from ._lib import check, lib, stream
        c = partial.shape[2]
        buf = torch.empty(2 * c + 1, dtype=torch.float64, device=partial.device)
        og, ob = out
        dgamma = og.view(c) if og is not None and og.numel() == c and og.dtype == torch.float32 else torch.empty(c, dtype=torch.float32, device=partial.device)
        dbeta = ob.view(c) if ob is not None and ob.numel() == c and ob.dtype == torch.float32 else torch.empty(c, dtype=torch.float32, device=partial.device)
        st = stream()
        check(lib().zs3_bn_sync_exchange_bwd(native_comm(st, group), partial.data_ptr(), partial.shape[0], c, float(count),
                                             buf.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), st), "zs3_bn_sync_exchange_bwd")
        return buf, None, dgamma, dbeta
    buf, cnt = combine_bn_partials(partial, count, group)
    return buf, cnt, None, None


def enable_sync_bn(module, group=None, enabled=True):
    """SynchronizedBatchNorm2d synchronises across ranks by itself whenever torch.distributed runs with more than one rank
    (modeling/sync_batchnorm/batchnorm.py); this call only selects a process group other than the default one, or switches
    the exchange off (`enabled=False`: per-rank statistics, what bench.py --sync-bn 0 measures).  Returns the number of
    SyncBN layers.  Uses (var + eps)^-1/2 like F.batch_norm, not the vendored clamp(var, eps)^-1/2."""
    from .modeling.sync_batchnorm.batchnorm import SynchronizedBatchNorm2d
    n = 0
    for m in module.modules():
        if isinstance(m, SynchronizedBatchNorm2d):
            m.sync_group = group
            m.sync_enabled = bool(enabled)
            m._sync_cache = None
            n += 1
    return n
