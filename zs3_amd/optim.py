"""Fused optimizer steps on the HIP kernels with the torch.optim constructors and state-dict layout
(train_pascal.py:55-60 builds torch.optim.SGD; train_pascal_GMMN.py:65-67 builds torch.optim.Adam)."""
import ctypes

import torch

from . import functional as Fz
from ._lib import F, I, P, check, lib, stream


class SGD(torch.optim.SGD):
    """torch.optim.SGD semantics (momentum, dampening=0, weight_decay, nesterov); one fused kernel per parameter."""

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        touched = []
        for group in self.param_groups:
            if group.get("dampening", 0) != 0 or group.get("maximize", False):
                raise NotImplementedError("zs3_amd.optim.SGD supports dampening=0, maximize=False")
            lr, mom, wd, nest = group["lr"], group["momentum"], group["weight_decay"], group["nesterov"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                g = p.grad
                dense_p = p.data if p.data.is_contiguous() else None
                # parameters are dense in *some* permutation (channels_last conv weights): the update is elementwise,
                # so it only needs p, grad and the buffer to share strides
                if g.stride() != p.stride():
                    g = torch.empty_strided(p.shape, p.stride(), dtype=p.dtype, device=p.device).copy_(g)
                state = self.state[p]
                first = 0
                if mom != 0 and "momentum_buffer" not in state or (mom != 0 and state["momentum_buffer"] is None):
                    state["momentum_buffer"] = torch.empty_strided(p.shape, p.stride(), dtype=p.dtype, device=p.device)
                    first = 1
                buf = state.get("momentum_buffer") if mom != 0 else None
                check(lib().zs3_sgd_step(P(p), P(g), P(buf), ctypes.c_long(p.numel()), F(lr), F(mom), F(wd), I(int(nest)),
                                         I(first), stream()), "zs3_sgd_step")
                touched.append(p)
        Fz.invalidate_planes(*touched)
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
                g = p.grad if p.grad.stride() == p.stride() else p.grad.contiguous()
                check(lib().zs3_adam_step(P(p), P(g), P(state["exp_avg"]), P(state["exp_avg_sq"]), ctypes.c_long(p.numel()),
                                          F(group["lr"]), F(b1), F(b2), F(group["eps"]), F(group["weight_decay"]),
                                          I(int(state["step"])), P(getattr(self, "_step_dev", None)), stream()),
                      "zs3_adam_step")
                touched.append(p)
        Fz.invalidate_planes(*touched)
        return loss
