"""Fused optimizer steps on the HIP kernels with the torch.optim constructors and state-dict layout
(train_pascal.py:55-60 builds torch.optim.SGD; train_pascal_GMMN.py:65-67 builds torch.optim.Adam)."""
import ctypes

import torch

from . import functional as Fz
from ._lib import F, I, P, check, lib, stream


def _same_layout(a, b):
    """Same element order in memory: strides agree on every dimension of extent > 1 (the strides of extent-1 dimensions,
    e.g. of a [Cout, Cin, 1, 1] weight, are arbitrary and differ between channels_last and contiguous tensors)."""
    return a.shape == b.shape and all(sa == sb for sa, sb, n in zip(a.stride(), b.stride(), a.shape) if n > 1)


class SGD(torch.optim.SGD):
    """torch.optim.SGD semantics (momentum, dampening=0, weight_decay, nesterov), state-dict compatible
    (`momentum_buffer`).  All parameters of all groups that share momentum/nesterov are updated by ONE multi-tensor
    launch: a small table of {param, grad, buffer, n, lr, wd} records goes to the device each step (gradient tensors
    are new objects every step), 16 K elements per workgroup."""

    def zero_grad(self, set_to_none=True):
        """torch.optim.Optimizer.zero_grad(set_to_none=True) without its per-call profiler scope and foreach bookkeeping (1.2 ms of
        host time per step for the 320 parameters of the network; the step is host-bound in the 2-byte mode)"""
        if not set_to_none:
            return super().zero_grad(set_to_none=False)
        for group in self.param_groups:
            for p in group["params"]:
                p.grad = None

    def group_hyper(self):
        """{lr, weight_decay} of every parameter group as the host float array zs3_sgd_multi_g takes (and a plan patches)"""
        vals = []
        for group in self.param_groups:
            vals.extend((float(group["lr"]), float(group["weight_decay"])))
        return (ctypes.c_float * len(vals))(*vals)

    @torch.no_grad()
    def step(self, closure=None):
        """One multi-tensor launch per (device, momentum, nesterov).  The host side runs at every step boundary, where the
        GPU has nothing else queued, so it is kept short: the workgroup -> (tensor, chunk) map depends only on the tensor
        sizes and lives on the device across steps; the record table is assembled with numpy."""
        import struct

        import numpy as np
        loss = closure() if closure is not None else None
        chunk = lib().zs3_sgd_chunk()
        by_cfg = {}
        touched, keep, tables = [], [], []
        # learning rate / weight decay of the groups as launch arguments (zs3_sgd_multi_g: the table holds the group INDEX, so a
        # schedule changes no device memory and a recorded plan follows it by patching one argument); more groups than the launch
        # carries: the packed-floats table of zs3_sgd_multi
        by_group = len(self.param_groups) <= lib().zs3_sgd_max_groups()
        for gi, group in enumerate(self.param_groups):
            if group.get("dampening", 0) != 0 or group.get("maximize", False):
                raise NotImplementedError("zs3_amd.optim.SGD supports dampening=0, maximize=False")
            lr, mom, wd, nest = group["lr"], group["momentum"], group["weight_decay"], group["nesterov"]
            packed = gi if by_group else struct.unpack("<q", struct.pack("<ff", lr, wd))[0]
            for p in group["params"]:
                g = p.grad
                if g is None:
                    continue
                # parameters are dense in *some* permutation (channels_last conv weights): the update is elementwise,
                # so p, grad and the buffer only need to share strides
                if g.stride() != p.stride() and not _same_layout(g, p):
                    g = torch.empty_strided(p.shape, p.stride(), dtype=p.dtype, device=p.device).copy_(g)
                    keep.append(g)
                state = self.state[p]
                first = 0
                buf = None
                if mom != 0:
                    buf = state.get("momentum_buffer")
                    if buf is None:
                        buf = state["momentum_buffer"] = torch.empty_strided(p.shape, p.stride(), dtype=p.dtype, device=p.device)
                        first = 1
                    elif buf.stride() != p.stride() and not _same_layout(buf, p):
                        # a buffer restored by load_state_dict keeps the strides it was saved with (torch.optim.SGD on the
                        # reference: NCHW-contiguous); the kernel walks p, grad and buffer by raw pointer, so re-lay it once
                        buf = torch.empty_strided(p.shape, p.stride(), dtype=p.dtype, device=p.device).copy_(buf)
                        state["momentum_buffer"] = buf
                rec = by_cfg.setdefault((p.device, float(mom), bool(nest)), [])
                rec.append((p.data_ptr(), g.data_ptr(), buf.data_ptr() if buf is not None else 0, p.numel(), packed, first))
                touched.append(p)
        cache = self.__dict__.setdefault("_zs3_blockmaps", {})
        for (dev, mom, nest), recs in by_cfg.items():
            arr = np.asarray(recs, dtype=np.int64)
            sizes = tuple(arr[:, 3].tolist())
            hit = cache.get((dev, mom, nest))
            if hit is None or hit[0] != sizes:
                nchunks = (arr[:, 3] + chunk - 1) // chunk
                ent = np.repeat(np.arange(len(recs), dtype=np.int32), nchunks)
                within = np.concatenate([np.arange(n, dtype=np.int32) for n in nchunks]) if len(recs) else ent
                hit = (sizes, torch.from_numpy(np.stack((ent, within), 1).copy()).to(dev), int(nchunks.sum()))
                cache[(dev, mom, nest)] = hit
            table = torch.from_numpy(arr).pin_memory()
            table_d = table.to(dev, non_blocking=True)
            # (skip flag: while the f16x3 forward's range flag is up the gradients are not finite and the step is skipped)
            from . import ops
            flag = ops.range_flag(dev) if dev.type == "cuda" else None
            if by_group:
                check(lib().zs3_sgd_multi_g(P(table_d), P(hit[1]), I(hit[2]), F(mom), I(int(nest)), P(flag), self.group_hyper(),
                                            I(len(self.param_groups)), stream()), "zs3_sgd_multi_g")
            else:
                check(lib().zs3_sgd_multi(P(table_d), P(hit[1]), I(hit[2]), F(mom), I(int(nest)), P(flag), stream()), "zs3_sgd_multi")
            keep.extend((table, table_d, hit[1]))
            tables.append(table_d)
        self._keepalive = keep   # pinned staging buffers must outlive the asynchronous copies
        self._zs3_tables = tables   # (zs3_amd/plan.py moves a recorded step's tables out of the step's allocator pool)
        Fz.refresh_planes(*touched)   # one launch re-splits every updated conv weight into its bf16 hi/lo planes
        return loss


class Adam(torch.optim.Adam):
    """torch.optim.Adam (amsgrad=False, maximize=False) update rule; state keys 'step', 'exp_avg', 'exp_avg_sq'."""

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        touched = []
        for group in self.param_groups:
            if group.get("amsgrad", False) or group.get("maximize", False):
                raise NotImplementedError("zs3_amd.optim.Adam supports amsgrad=False, maximize=False")
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                state = self.state[p]
                if len(state) == 0:
                    state["step"] = torch.tensor(0.0)
                    state["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    state["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                state["step"] += 1
                g = p.grad if _same_layout(p.grad, p) else p.grad.contiguous()
                check(lib().zs3_adam_step(P(p), P(g), P(state["exp_avg"]), P(state["exp_avg_sq"]), ctypes.c_long(p.numel()),
                                          F(group["lr"]), F(b1), F(b2), F(group["eps"]), F(group["weight_decay"]),
                                          I(int(state["step"])), P(getattr(self, "_step_dev", None)), stream()),
                      "zs3_adam_step")
                touched.append(p)
        Fz.invalidate_planes(*touched)
        return loss
