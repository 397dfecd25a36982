"""The training step as a recorded launch plan: the loop body of zs3/base_trainer.py:16-20 issued from C.

    step = StepPlan(model, criterion, optimizer)
    for image, target in loader:
        scheduler(optimizer, i, epoch)              # learning rates may change every iteration
        prediction, loss = step(image, target)      # == zero_grad(); model(image); criterion(...); backward(); optimizer.step()

The first calls run the ordinary eager step (every launch a Python -> ctypes call: ~30 ms of host time per step); once the step has
settled the next call is RECORDED -- it still runs eagerly, and every entry point of libzs3hip.so it calls appends its arguments
to a plan (include/zs3hip.h "recorded launch plans", csrc/plan.hip) -- and every call after that REPLAYS the plan: one C call
issues the ~900 launches of forward, loss, backward and optimizer on the streams they were recorded on, after the few scalars
that differ between iterations have been patched (the groups' learning rate / weight decay, the dropout seeds).  Results are
bit-identical to the eager step (tests/test_gpu_plan.py); `verify()` checks it on the live model.

What makes a replay valid, and how it is kept:
* every device buffer of the recorded step stays where it was: the recording runs under a private allocator pool
  (torch.cuda.MemPool) that nothing else allocates from, so activations / gradients / workspaces keep their addresses from
  replay to replay without being held as tensors; parameters, optimizer state and BatchNorm buffers are the model's own;
* the input batch is REBOUND, not copied: a new `image` / `target` storage is patched into the plan (zs3_plan_replace_ptr);
* cross-stream order is part of the plan (zs3_stream_wait); while recording, the ASPP branches run on one stream and
  side-stream operands are kept alive instead of handed to the allocator's record_stream bookkeeping (functional.PLAN_RECORDING);
* nothing of the step may run outside the library: the tensor-library stragglers of earlier rounds (loss clone, bias-gradient
  reduction, zero fills, channel pads, the optimizer's per-step table upload for the schedule) are library calls now;
* anything that changes what the step launches drops the plan and the next calls run eagerly / re-record: another input shape
  or dtype, train / eval flips, requires_grad or parameter storage changes, another set of optimizer hyper-parameters than lr and
  weight decay, a precision / storage mode switch or range-guard fallback (functional.PLAN_EPOCH).
Several ranks: with the RCCL backend the step's collectives are entry points of the library (csrc/comm.hip) on the streams that hold
the data, so the N > 1 step records and replays like the one-GPU step.  Not planned (the call runs eagerly, every time):
collectives through torch.distributed (gloo, ZS3_NATIVE_RCCL=0), an optimizer other than zs3_amd.optim.SGD, CPU tensors, gradient
mode off.
"""
import ctypes
import os

import torch

from . import functional as Fz
from . import ops
from ._lib import Zs3HipError, check, lib

ENABLED = os.environ.get("ZS3_PLAN", "1") == "1"


def collectives_recordable():
    """With more than one rank a step contains collectives: they are part of a plan only when the library issues them itself (RCCL
    backend: csrc/comm.hip); through torch.distributed (gloo, ZS3_NATIVE_RCCL=0) a recording would silently miss them."""
    import torch.distributed as dist
    from . import parallel
    if parallel.FORCE_COLLECTIVES or (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return parallel.native_available()
    return True


def _module_scalars(root):
    """per module of `root`: train / eval and the scalars its launches carry as arguments (dropout probability, BatchNorm momentum /
    eps), read from the instance dictionaries (a getattr() that misses costs nn.Module an exception: 3 x 500 of them per call was
    1.5 ms of host time in front of every step)"""
    out = []
    for m in root.modules():
        d = m.__dict__
        out.append((d.get("training"), d.get("p"), d.get("momentum"), d.get("eps")))
    return tuple(out)


class LaunchPlan:
    """Thin owner of one zs3_plan handle."""

    def __init__(self):
        self.handle = int(lib().zs3_plan_create())
        self.nops = 0

    def begin(self):
        check(lib().zs3_plan_record_begin(self.handle), "zs3_plan_record_begin")

    def end(self):
        n = lib().zs3_plan_record_end(self.handle)
        if n < 0:
            check(n, "zs3_plan_record_end")
        self.nops = n
        return n

    def replay(self, first=0, count=-1):
        rc = lib().zs3_plan_replay(self.handle, first, count)
        if rc != 0:
            op = lib().zs3_plan_failed_op(self.handle)
            raise Zs3HipError(f"zs3_plan_replay: op {op} ({self.op_name(op)}) failed with code {rc}")

    def op_name(self, op):
        buf = ctypes.create_string_buffer(64)
        lib().zs3_plan_op_name(self.handle, op, buf, 64)
        return buf.value.decode()

    def names(self):
        return [self.op_name(i) for i in range(self.nops)]

    def find(self, name, nth=0):
        return lib().zs3_plan_find_op(self.handle, name.encode(), nth)

    def replace_u64(self, old, new):
        return lib().zs3_plan_replace_u64(self.handle, old, new)

    def replace_ptr(self, old, new):
        return lib().zs3_plan_replace_ptr(self.handle, old, new)

    def find_ptr(self, ptr, cap=64):
        """[(op, arg), ...] where the device pointer occurs among the recorded arguments"""
        where = (ctypes.c_int * (2 * cap))()
        n = lib().zs3_plan_find_ptr(self.handle, ptr, where, cap)
        if n > cap:
            return self.find_ptr(ptr, n)
        return [(where[2 * k], where[2 * k + 1]) for k in range(n)]

    def set_ptr(self, op, arg, ptr):
        check(lib().zs3_plan_set_ptr(self.handle, op, arg, ptr), "zs3_plan_set_ptr")

    def patch(self, op, arg, carray):
        check(lib().zs3_plan_patch(self.handle, op, arg, carray, ctypes.sizeof(carray)), "zs3_plan_patch")

    def time_ops(self, indices):
        """arm HIP-event pairs around these ops (ascending indices) for the NEXT replay -> set id for timed_ms()"""
        arr = (ctypes.c_int * len(indices))(*indices)
        rc = lib().zs3_plan_time_ops(self.handle, arr, len(indices))
        if rc < 0:
            check(rc, "zs3_plan_time_ops")
        return rc

    def timed_ms(self, set_id, n):
        out = (ctypes.c_float * n)()
        rc = lib().zs3_plan_timed_ms(self.handle, set_id, out, n)
        if rc < 0:
            check(rc, "zs3_plan_timed_ms")
        return list(out)[:rc]

    def close(self):
        if self.handle:
            lib().zs3_plan_destroy(self.handle)
            self.handle = 0

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class StepPlan:
    """See the module docstring.  `warmup`: eager calls before the recording (weight planes, momentum buffers, kernel choices and
    the optimizer's block maps exist and are final after two)."""

    def __init__(self, model, criterion, optimizer, warmup=2, enabled=None):
        self.model, self.criterion, self.optimizer = model, criterion, optimizer
        self.warmup = max(2, int(warmup))
        self.enabled = ENABLED if enabled is None else bool(enabled)
        self.replays = self.recordings = self.eager_calls = 0
        self._one = None
        self._giveup = None          # fingerprint of a configuration whose step turned out not to be replayable
        self._drop()
        if torch.cuda.is_available():
            Fz.warm_streams()        # (the step's streams take their hardware queues in a fixed order: functional.warm_streams)

    # ------------------------------------------------------------------------------------------------ state
    def _drop(self):
        plan = getattr(self, "_plan", None)
        if plan is not None:
            plan.close()
        self._plan = self._pool = self._key = self._hyper = None
        self._held = None            # tensors of the recorded step that live past it: prediction, loss, optimizer tables, ...
        self._seeds, self._inputs, self._input_at, self._sgd_ops = [], None, None, []
        self._grads, self._grads_moved = [], False
        self._settled = 0

    def _plannable(self, image, target):
        if not (self.enabled and image.is_cuda and target.is_cuda and torch.is_grad_enabled()):
            return False
        from .optim import SGD
        if not isinstance(self.optimizer, SGD) or len(self.optimizer.param_groups) > lib().zs3_sgd_max_groups():
            return False
        # more than one rank: plannable when the collectives are the library's own (SyncBN sums, gradient buckets, CE weight sums and
        # the range flag as zs3_allreduce / zs3_bn_sync_exchange calls, recorded like launches)
        return collectives_recordable()

    def _fingerprint(self, image, target):
        """everything that decides WHAT the step launches and WHERE its persistent operands live"""
        # per module: train / eval, and the scalars its launches carry as arguments (dropout probability, BatchNorm momentum / eps)
        mods = _module_scalars(self.model)
        params = tuple((p.data_ptr(), p.requires_grad) for g in self.optimizer.param_groups for p in g["params"])
        hyper = tuple((g["momentum"], g["nesterov"], g.get("dampening", 0)) for g in self.optimizer.param_groups)
        owner = getattr(self.criterion, "__self__", None)      # SegmentationLosses: class weights, ignore index, batch averaging
        w = getattr(owner, "weight", None)
        crit = (id(owner), (w.data_ptr(), w._version) if torch.is_tensor(w) else None, getattr(owner, "ignore_index", None),
                getattr(owner, "batch_average", None), getattr(self.criterion, "__name__", None))
        return (tuple(image.shape), image.dtype, tuple(image.stride()), tuple(target.shape), target.dtype, tuple(target.stride()),
                image.device, mods, params, hyper, crit, Fz.PLAN_EPOCH[0], ops.PREC_DEFAULT, ops.ACT_DTYPE, ops.FWD_F16,
                Fz.WGRAD_SIDE_STREAM, Fz.WGRAD_STREAMS, torch.cuda.current_stream(image.device).cuda_stream)

    # ------------------------------------------------------------------------------------------------ the three ways to run a step
    def _eager(self, image, target):
        self.optimizer.zero_grad()
        prediction = self.model(image)
        loss = self.criterion(prediction, target)
        if self._one is None or self._one.device != loss.device:
            # the seed of backward as a tensor that lives outside every step (autograd's own `ones_like(loss)` is a fresh fill per
            # call, and a fill the tensor library launches is not part of a plan)
            self._one = torch.ones((), dtype=torch.float32, device=loss.device)
        loss.backward(self._one if loss.dtype == torch.float32 and loss.dim() == 0 else None)
        self.optimizer.step()
        return prediction, loss

    def _record(self, image, target):
        dev = image.device
        plan, pool = LaunchPlan(), torch.cuda.MemPool()
        drawn, next_seed = [], Fz.next_seed

        def logged_seed():
            v = next_seed()
            drawn.append(v)
            return v

        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        torch.cuda.synchronize(dev)
        Fz.next_seed, Fz.PLAN_RECORDING = logged_seed, True
        torch._C._cuda_beginAllocateToPool(idx, pool.id)      # every allocation of the step, from either thread, into the pool
        try:
            plan.begin()
            try:
                prediction, loss = self._eager(image, target)
            finally:
                plan.end()
        except BaseException:
            # the step itself failed (out of memory, a shape error): nothing was recorded that anyone will replay
            plan.close()
            torch.cuda.synchronize(dev)
            Fz._plan_keep.clear()
            self._settled = 0
            raise
        finally:
            torch._C._cuda_endAllocateToPool(idx, pool.id)
            Fz.next_seed, Fz.PLAN_RECORDING = next_seed, False
        keep, Fz._plan_keep = Fz._plan_keep, []
        torch.cuda.synchronize(dev)      # the side-stream readers of `keep` are done: from here on stream order on main protects them
        del keep
        if len(set(drawn)) != len(drawn):
            plan.close()
            self._settled = 0            # (two equal 63-bit seeds: cannot tell the launches apart; eager now, another recording later)
            return prediction, loss
        # The optimizer's record table {param, grad, momentum buffer, ...} was uploaded by the tensor library during the step, into pool
        # memory that an EARLIER intermediate of the same step had used: a replay re-runs that intermediate's producer and would
        # overwrite the table, whose upload is not part of the plan.  Its content is final (the gradients' addresses are the
        # pool's): move it to memory outside the pool and point the recorded launch there.
        # (ONLY the optimizer launch's argument: the table's pool address belonged to other tensors earlier in the step.)
        moved, sgd_ops, k = [], [], 0
        while plan.find("zs3_sgd_multi_g", k) >= 0:
            sgd_ops.append(plan.find("zs3_sgd_multi_g", k))
            k += 1
        for t in getattr(self.optimizer, "_zs3_tables", ()):
            fixed = t.clone()
            at = [(op, a) for op, a in plan.find_ptr(t.data_ptr()) if op in sgd_ops and a == 0]
            if len(at) != 1:
                raise RuntimeError("StepPlan: the optimizer's table is not an argument of exactly one recorded optimizer launch")
            plan.set_ptr(at[0][0], 0, fixed.data_ptr())
            moved.append(fixed)
        torch.cuda.synchronize(dev)
        self._plan, self._pool = plan, pool
        self._seeds = list(drawn)
        # where the batch enters the plan (the two tensors are alive and outside the pool: their addresses mean nothing else)
        self._inputs = [image.data_ptr(), target.data_ptr()]
        self._input_at = [plan.find_ptr(image.data_ptr()), plan.find_ptr(target.data_ptr())]
        if not self._input_at[0] or not self._input_at[1] or image.data_ptr() == target.data_ptr():
            # the step did not read its batch where the caller's tensors live (a cast or a .contiguous() copy the tensor library made
            # in front of the first launch: uint8 / float64 labels, a non-contiguous image): this configuration stays eager
            plan.close()
            self._plan, self._pool, self._giveup = None, None, self._key
            return prediction, loss
        self._held = (prediction, loss, moved, image, target)
        self._grads = [(p, p.grad) for g in self.optimizer.param_groups for p in g["params"] if p.grad is not None]
        self._grads_moved = False
        self._sgd_ops = sgd_ops
        self._hyper = bytes(self.optimizer.group_hyper())
        self.recordings += 1
        return prediction, loss

    def _replay(self, image, target):
        plan = self._plan
        # the parameters' .grad are the tensors the replayed launches write -- also after an eager step in between (it left its own
        # gradient tensors there) or a zero_grad() of the caller's
        for p, g in self._grads:
            if p.grad is not g:
                p.grad = g
        self._grads_moved = False
        for slot, t in enumerate((image, target)):
            ptr = t.data_ptr()
            if ptr != self._inputs[slot]:
                for op, a in self._input_at[slot]:
                    plan.set_ptr(op, a, ptr)
                self._inputs[slot] = ptr
        hyper = self.optimizer.group_hyper()
        if bytes(hyper) != self._hyper:
            for op in self._sgd_ops:
                plan.patch(op, 6, hyper)
            self._hyper = bytes(hyper)
        for k, old in enumerate(self._seeds):           # the eager step draws its dropout seeds in this order
            new = Fz.next_seed()
            if plan.replace_u64(old, new) < 1:
                raise RuntimeError("StepPlan: a recorded dropout seed is gone from the plan")
            self._seeds[k] = new
        plan.replay()
        self._held = self._held[:3] + (image, target)   # the batch the queued launches read stays referenced until the next call
        self.replays += 1
        return self._held[0], self._held[1]

    # ------------------------------------------------------------------------------------------------ the call
    def __call__(self, image, target, eager=False):
        """-> (prediction, loss).  eager=True: this one call runs the ordinary eager step whatever the state of the plan (an
        instrumented step: per-launch timing events live in the Python wrappers); the plan stays valid -- it owns its buffers, and
        parameters / optimizer state / BatchNorm buffers are shared, so eager and replayed steps interleave freely."""
        if eager or not self._plannable(image, target):
            self.eager_calls += 1
            self._grads_moved = True
            return self._eager(image, target)
        key = self._fingerprint(image, target)
        if key != self._key:
            self._drop()
            self._key = key
        if self._plan is not None:
            return self._replay(image, target)
        if key == self._giveup:
            self.eager_calls += 1
            return self._eager(image, target)
        self._settled += 1
        if self._settled <= self.warmup:
            self.eager_calls += 1
            return self._eager(image, target)
        return self._record(image, target)

    # ------------------------------------------------------------------------------------------------ checking a live plan
    def state_tensors(self):
        """every tensor a step changes persistently: parameters, momentum buffers, BatchNorm buffers"""
        out = [p.data for p in self.model.parameters()] + [b for b in self.model.buffers()]
        for st in self.optimizer.state.values():
            out.extend(v for v in st.values() if torch.is_tensor(v) and v.is_cuda)
        return out

    def verify(self, image, target, poison=True):
        """Replay the recorded plan and run the same step eagerly from the same state with the same dropout seeds and compare the
        results bit for bit: loss, prediction, every gradient and every persistent tensor.  `poison`: the pool's free memory -- where
        the plan's activations and workspaces live between replays -- is filled with NaN patterns first, so that a launch missing
        from the plan (a fill or copy the tensor library made while recording) shows up instead of finding last step's bytes.
        Leaves the model one step further (the eager step's result) and DROPS the plan (the next calls settle and record again).
        Returns a list of mismatching names (empty = identical)."""
        if self._plan is None:
            raise RuntimeError("StepPlan.verify: no recorded plan (call the step warmup + 1 times first)")
        state = self.state_tensors()
        saved = [t.clone() for t in state]
        rng = Fz._rng.getstate() if Fz._rng is not None else None
        if poison:
            self._poison()
        pred_p, loss_p = self._replay(image, target)
        got = {"loss": loss_p.clone(), "prediction": pred_p.clone()}
        got.update({f"grad[{i}]": p.grad.clone() for i, p in enumerate(self.model.parameters()) if p.grad is not None})
        got.update({f"state[{i}]": t.clone() for i, t in enumerate(state)})
        for t, s in zip(state, saved):
            t.copy_(s)
        if rng is not None:
            Fz._rng.setstate(rng)
        Fz.refresh_planes(*[p for g in self.optimizer.param_groups for p in g["params"]])
        pred_e, loss_e = self._eager(image, target)
        want = {"loss": loss_e, "prediction": pred_e}
        want.update({f"grad[{i}]": p.grad for i, p in enumerate(self.model.parameters()) if p.grad is not None})
        want.update({f"state[{i}]": t for i, t in enumerate(state)})
        torch.cuda.synchronize()
        bad = [k for k in want if k not in got or not torch.equal(got[k], want[k])]
        # the eager step left fresh gradient tensors on the parameters; the plan keeps writing its own: hand them back
        self._drop()
        return bad

    def _poison(self):
        """fill every free block of the plan's pool with 0xFF bytes (NaNs as fp32 and as bf16)"""
        dev = self._held[1].device
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        torch.cuda.synchronize(dev)
        free = []
        for seg in torch.cuda.memory_snapshot():
            if seg.get("device") != idx or tuple(seg.get("segment_pool_id", (0, 0))) != tuple(self._pool.id):
                continue
            free.extend(b["size"] for b in seg["blocks"] if b["state"] == "inactive")
        torch._C._cuda_beginAllocateToPool(idx, self._pool.id)
        try:
            hold = []
            for size in sorted(free, reverse=True):
                t = torch.empty(size, dtype=torch.uint8, device=dev)
                t.fill_(255)
                hold.append(t)
            torch.cuda.synchronize(dev)
            del hold
        finally:
            torch._C._cuda_endAllocateToPool(idx, self._pool.id)

    def conv_ops(self):
        """indices of the recorded convolution launches (forward and data gradient), in launch order -- the order ops.PROFILE lists
        them in an eager step"""
        if self._plan is None:
            return []
        return [i for i, n in enumerate(self._plan.names()) if n in ("zs3_conv_igemm", "zs3_conv_igemm_in", "zs3_conv_igemm_bnstats")]

    def time_next_replay(self, indices):
        return self._plan.time_ops(sorted(indices))

    def timed_ms(self, set_id, n):
        return self._plan.timed_ms(set_id, n)

    @property
    def recorded_ops(self):
        return self._plan.nops if self._plan is not None else 0

    def close(self):
        self._drop()


class ForwardPlan:
    """A gradient-free forward of one tensor -> one tensor as a recorded plan: `fn(image)` (the frozen-backbone feature pass of the
    GMMN step, train_pascal_GMMN.py:154-158: `model.forward_before_class_prediction(image)` under no_grad, BatchNorm in training
    mode, dropout live) runs eagerly twice, is recorded on the third call and replayed afterwards.  Same rules as StepPlan (private
    pool, the input rebound by pointer, dropout seeds patched in drawing order, anything that changes the launches drops the plan);
    simpler in one respect -- no autograd thread, no optimizer.  The recorded output buffer is the plan's own: the caller gets a
    COPY (the GMMN step reads batch t's features while batch t + 1's pass is already running)."""

    def __init__(self, fn, modules, warmup=2, enabled=None):
        self.fn, self.modules = fn, list(modules)
        self.warmup = max(2, int(warmup))
        self.enabled = ENABLED if enabled is None else bool(enabled)
        self.replays = self.recordings = self.eager_calls = 0
        self._plans = {}      # fingerprint (stream included) -> state dict

    def _fingerprint(self, image):
        mods = tuple(_module_scalars(root) for root in self.modules)
        params = tuple(p.data_ptr() for root in self.modules for p in root.parameters())
        return (tuple(image.shape), image.dtype, tuple(image.stride()), image.device, mods, params, Fz.PLAN_EPOCH[0], ops.PREC_DEFAULT,
                ops.ACT_DTYPE, ops.FWD_F16, torch.cuda.current_stream(image.device).cuda_stream)

    def close(self):
        for st in self._plans.values():
            if st.get("plan") is not None:
                st["plan"].close()
        self._plans = {}

    def __call__(self, image):
        if not (self.enabled and image.is_cuda) or torch.is_grad_enabled() or not collectives_recordable():
            self.eager_calls += 1
            return self.fn(image)
        key = self._fingerprint(image)
        st = self._plans.get(key)
        if st is None:
            if len(self._plans) > 4:       # (shapes / modes come and go: keep the table small, the pools with it)
                self.close()
            st = self._plans[key] = {"seen": 0, "plan": None}
        if st["plan"] is not None:
            plan = st["plan"]
            if image.data_ptr() != st["input"]:
                for op, a in st["input_at"]:
                    plan.set_ptr(op, a, image.data_ptr())
                st["input"] = image.data_ptr()
            for k, old in enumerate(st["seeds"]):
                new = Fz.next_seed()
                if plan.replace_u64(old, new) < 1:
                    raise RuntimeError("ForwardPlan: a recorded dropout seed is gone from the plan")
                st["seeds"][k] = new
            plan.replay()
            st["image"] = image
            self.replays += 1
            return st["out"].clone()
        st["seen"] += 1
        if st["seen"] <= self.warmup:
            self.eager_calls += 1
            return self.fn(image)
        dev = image.device
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        plan, pool = LaunchPlan(), torch.cuda.MemPool()
        drawn, next_seed = [], Fz.next_seed

        def logged_seed():
            v = next_seed()
            drawn.append(v)
            return v

        torch.cuda.synchronize(dev)
        Fz.next_seed, Fz.PLAN_RECORDING = logged_seed, True
        torch._C._cuda_beginAllocateToPool(idx, pool.id)
        try:
            plan.begin()
            try:
                out = self.fn(image)
            finally:
                plan.end()
        finally:
            torch._C._cuda_endAllocateToPool(idx, pool.id)
            Fz.next_seed, Fz.PLAN_RECORDING = next_seed, False
        Fz._plan_keep.clear()
        at = plan.find_ptr(image.data_ptr())
        if not at or len(set(drawn)) != len(drawn) or not plan.find_ptr(out.data_ptr()):
            plan.close()                   # (the pass copied its input, or its result is not what a recorded launch wrote: stay eager)
            st["seen"] = 0
            return out
        st.update(plan=plan, pool=pool, seeds=list(drawn), input=image.data_ptr(), input_at=at, out=out, image=image)
        self.recordings += 1
        return out.clone()
