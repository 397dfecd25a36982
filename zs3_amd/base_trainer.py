"""Supervised epoch loop with the surface of the reference's BaseTrainer (zs3/base_trainer.py:4-57): subclasses
inject model / optimizer / criterion / scheduler / loaders / writer / summary / saver and call `training(epoch)`.

Behaviour kept: single-sample batches are skipped (:11), the LR schedule is applied before every step (:15), the
step is zero_grad -> forward -> loss -> backward -> step (:16-20), the scalar loss is logged per iteration (:23-25),
an image dump happens ten times per epoch (:28-38), and with `no_val` a checkpoint is written every epoch (:46-57).
The loss value is read back once per iteration (the reference syncs twice)."""


class BaseTrainer:
    # ------------------------------------------------------------------ pieces of one epoch
    def _to_device(self, sample):
        image, target = sample["image"], sample["label"]
        if self.args.cuda:
            return image.cuda(), target.cuda()
        return image, target

    def _train_iteration(self, image, target):
        self.optimizer.zero_grad()
        prediction = self.model(image)
        loss = self.criterion(prediction, target)
        loss.backward()
        self.optimizer.step()
        return prediction, loss.item()

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
        for index, sample in enumerate(iterator):
            last_index = index
            if len(sample["image"]) <= 1:
                continue
            image, target = self._to_device(sample)
            last_batch = image.shape[0]
            self.scheduler(self.optimizer, index, epoch, self.best_pred)
            prediction, value = self._train_iteration(image, target)
            running += value
            step = index + per_epoch * epoch
            if hasattr(iterator, "set_description"):
                iterator.set_description("Train loss: %.3f" % (running / (index + 1)))
            self.writer.add_scalar("train/total_loss_iter", value, step)
            if index % dump_every == 0:   # ZeroDivisionError for loaders shorter than 10 batches, like the reference
                self.summary.visualize_image(self.writer, self.args.dataset, image, target, prediction, step)
        self._epoch_end(epoch, running, last_index * self.args.batch_size + last_batch)
