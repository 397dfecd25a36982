"""Per-iteration LR schedule -- zs3/utils/lr_scheduler.py:14-76 (poly / cos / step, 10x on groups >= 1)."""
import math


class LR_Scheduler:
    def __init__(self, mode, base_lr, num_epochs, iters_per_epoch=0, lr_step=0, warmup_epochs=0, verbose=True):
        if mode == "step":
            assert lr_step
        self.mode, self.lr, self.lr_step = mode, base_lr, lr_step
        self.iters_per_epoch = iters_per_epoch
        self.N = num_epochs * iters_per_epoch
        self.epoch = -1
        self.warmup_iters = warmup_epochs * iters_per_epoch
        self.verbose = verbose

    def lr_at(self, i, epoch):
        t = epoch * self.iters_per_epoch + i
        if self.mode == "cos":
            lr = 0.5 * self.lr * (1 + math.cos(1.0 * t / self.N * math.pi))
        elif self.mode == "poly":
            lr = self.lr * pow((1 - 1.0 * t / self.N), 0.9)
        elif self.mode == "step":
            lr = self.lr * (0.1 ** (epoch // self.lr_step))
        else:
            raise NotImplementedError
        if self.warmup_iters > 0 and t < self.warmup_iters:
            lr = lr * 1.0 * t / self.warmup_iters
        return lr

    def __call__(self, optimizer, i, epoch, best_pred):
        lr = self.lr_at(i, epoch)
        if epoch > self.epoch:
            if self.verbose:
                print("\n=>Epoches %i, learning rate = %.4f, previous best = %.4f" % (epoch, lr, best_pred))
            self.epoch = epoch
        assert lr >= 0
        groups = optimizer.param_groups
        groups[0]["lr"] = lr
        for g in groups[1:]:
            g["lr"] = lr * 10
