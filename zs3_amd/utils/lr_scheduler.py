"""Per-iteration learning-rate schedule with the call signature of zs3/utils/lr_scheduler.py:14-76:
`LR_Scheduler(mode, base_lr, num_epochs, iters_per_epoch, lr_step, warmup_epochs)(optimizer, i, epoch, best_pred)`.
poly: base*(1 - T/N)^0.9, cos: base/2*(1 + cos(pi*T/N)), step: base*0.1^(epoch // lr_step); T = epoch*iters + i,
N = num_epochs*iters; linear warm-up over warmup_epochs; group 0 gets lr, every other group 10*lr."""
import math


class LR_Scheduler:
    def __init__(self, mode, base_lr, num_epochs, iters_per_epoch=0, lr_step=0, warmup_epochs=0, verbose=True):
        if mode not in ("poly", "cos", "step"):
            raise NotImplementedError(mode)
        if mode == "step" and not lr_step:
            raise AssertionError("step mode needs lr_step")
        self.mode, self.lr, self.lr_step = mode, base_lr, lr_step
        self.iters_per_epoch = iters_per_epoch
        self.N = num_epochs * iters_per_epoch
        self.warmup_iters = warmup_epochs * iters_per_epoch
        self.epoch, self.verbose = -1, verbose

    def lr_at(self, i, epoch):
        t = epoch * self.iters_per_epoch + i
        progress = 1.0 * t / self.N if self.N else 0.0
        value = {"poly": lambda: self.lr * pow(1 - progress, 0.9),
                 "cos": lambda: 0.5 * self.lr * (1 + math.cos(progress * math.pi)),
                 "step": lambda: self.lr * (0.1 ** (epoch // self.lr_step))}[self.mode]()
        if 0 < self.warmup_iters and t < self.warmup_iters:
            value *= 1.0 * t / self.warmup_iters
        return value

    def __call__(self, optimizer, i, epoch, best_pred):
        value = self.lr_at(i, epoch)
        assert value >= 0
        if epoch > self.epoch:
            self.epoch = epoch
            if self.verbose:
                print("\n=>Epoches %i, learning rate = %.4f, previous best = %.4f" % (epoch, value, best_pred))
        for index, group in enumerate(optimizer.param_groups):
            group["lr"] = value if index == 0 else value * 10
