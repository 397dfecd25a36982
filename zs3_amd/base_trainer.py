"""Supervised epoch loop with the surface of the reference's BaseTrainer (zs3/base_trainer.py:4-57): subclasses
inject model / optimizer / criterion / scheduler / loaders / writer / summary / saver and call `training(epoch)`.

Behaviour kept: single-sample batches are skipped (:11), the LR schedule is applied before every step (:15), the
step is zero_grad -> forward -> loss -> backward -> step (:16-20), the scalar loss is logged per iteration (:23-25),
an image dump happens ten times per epoch (:28-38), and with `no_val` a checkpoint is written every epoch (:46-57).
Every iteration's loss value is read on the host (the reference reads it twice per iteration, each a device synchronisation,
base_trainer.py:21,24) -- one iteration LATE: `LossLog` copies the 4 bytes to pinned memory behind the step and looks at them
when the next step has been queued, so no step waits for its own loss and the host keeps running ahead of the GPU.  Every
value still reaches the running sum, the progress bar and `train/total_loss_iter` under its own iteration number."""
import torch


class LossLog:
    """Loss values of consecutive iterations, read without stalling the iteration that produced them: push(loss) queues an
    asynchronous copy of the 0-dim device tensor into a pinned slot and returns the (index, value) pairs that have ARRIVED --
    normally the previous iteration's; flush() waits for the rest."""

    def __init__(self, device=None, depth=2):
        self.depth = depth
        self.buf = torch.zeros(depth, dtype=torch.float32).pin_memory() if torch.cuda.is_available() else torch.zeros(depth)
        # the range flag of the f16x3 forward travels with the loss: same moment, same asynchronous copy, one iteration late
        self.flags = torch.zeros(depth, dtype=torch.int32).pin_memory() if torch.cuda.is_available() else None
        self.pending = []      # (index, slot, event)
        self.count = 0

    def push(self, loss):
        out = []
        if len(self.pending) >= self.depth - 1:      # the slot about to be reused must have been read
            out.extend(self._take(len(self.pending) - (self.depth - 2)))
        slot = self.count % self.depth
        if loss.is_cuda:
            self.buf[slot:slot + 1].copy_(loss.detach().reshape(1), non_blocking=True)
            from . import ops
            if self.flags is not None and ops._range_flags:
                self.flags[slot:slot + 1].copy_(ops.range_flag(loss.device), non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
        else:
            self.buf[slot] = float(loss)
            ev = None
        self.pending.append((self.count, slot, ev))
        self.count += 1
        return out

    def _take(self, n):
        got = []
        for _ in range(n):
            idx, slot, ev = self.pending.pop(0)
            if ev is not None:
                ev.synchronize()
                if self.flags is not None and int(self.flags[slot]):
                    from . import functional as Fz
                    self.flags[slot] = 0
                    Fz.check_forward_range(flag_value=1)     # fall back to bf16x3 forward products (the flagged steps were skipped)
            got.append((idx, float(self.buf[slot])))
        return got

    def flush(self):
        return self._take(len(self.pending))


class BaseTrainer:
    # ------------------------------------------------------------------ pieces of one epoch
    def _to_device(self, sample):
        image, target = sample["image"], sample["label"]
        if self.args.cuda:
            return image.cuda(), target.cuda()
        return image, target

    def _train_iteration(self, image, target):
        """zero_grad -> forward -> loss -> backward -> step (base_trainer.py:16-20), through zs3_amd.plan.StepPlan: with the fused
        optimizer (zs3_amd.optim.SGD) on one GPU the third iteration of a batch shape is recorded and the later ones are replayed from
        C, bit-identical to the eager lines; any other setting (torch.optim, several ranks, ZS3_PLAN=0) runs exactly those lines."""
        plan = self.__dict__.get("_zs3_step_plan")
        if plan is None or plan.model is not self.model or plan.optimizer is not self.optimizer or plan.criterion is not self.criterion:
            from .plan import StepPlan
            plan = self.__dict__["_zs3_step_plan"] = StepPlan(self.model, self.criterion, self.optimizer)
        return plan(image, target)

    def _epoch_end(self, epoch, running, seen_images):
        self.writer.add_scalar("train/total_loss_epoch", running, epoch)
        print("[Epoch: %d, numImages: %5d]" % (epoch, seen_images))
        print(f"Loss: {running:.3f}")
        if self.args.no_val:  # no validation pass: keep a checkpoint per epoch
            state = {"epoch": epoch + 1, "state_dict": self.model.module.state_dict(),
                     "optimizer": self.optimizer.state_dict(), "best_pred": self.best_pred}
            self.saver.save_checkpoint(state, False)

    # ------------------------------------------------------------------ the reference entry point
    def training(self, epoch):
        self.model.train()
        iterator = self.train_loader
        try:
            from tqdm import tqdm
            iterator = tqdm(iterator)
        except Exception:  # pragma: no cover
            pass
        per_epoch = len(self.train_loader)
        dump_every = per_epoch // 10
        running, last_index, last_batch = 0.0, -1, 0
        log, steps_of = LossLog(), {}

        def account(pairs):
            nonlocal running
            for k, value in pairs:
                running += value
                self.writer.add_scalar("train/total_loss_iter", value, steps_of.pop(k))

        for index, sample in enumerate(iterator):
            last_index = index
            if len(sample["image"]) <= 1:
                continue
            image, target = self._to_device(sample)
            last_batch = image.shape[0]
            self.scheduler(self.optimizer, index, epoch, self.best_pred)
            prediction, loss = self._train_iteration(image, target)
            step = index + per_epoch * epoch
            steps_of[log.count] = step
            account(log.push(loss))          # (the previous iteration's value: this one's is still being computed)
            if hasattr(iterator, "set_description"):
                iterator.set_description("Train loss: %.3f" % (running / (index + 1)))
            if index % dump_every == 0:   # ZeroDivisionError for loaders shorter than 10 batches, like the reference
                self.summary.visualize_image(self.writer, self.args.dataset, image, target, prediction, step)
        account(log.flush())
        self._epoch_end(epoch, running, last_index * self.args.batch_size + last_batch)
