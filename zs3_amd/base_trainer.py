"""BaseTrainer -- the supervised loop of zs3/base_trainer.py:4-57 (same injected attributes and call order)."""


def _progress(it):
    try:
        from tqdm import tqdm
        return tqdm(it)
    except Exception:  # pragma: no cover
        return it


class BaseTrainer:
    def training(self, epoch):
        train_loss = 0.0
        self.model.train()
        tbar = _progress(self.train_loader)
        num_img_tr = len(self.train_loader)
        i, image = -1, None
        for i, sample in enumerate(tbar):
            if len(sample["image"]) <= 1:  # single-sample batches are skipped (base_trainer.py:11)
                continue
            image, target = sample["image"], sample["label"]
            if self.args.cuda:
                image, target = image.cuda(), target.cuda()
            self.scheduler(self.optimizer, i, epoch, self.best_pred)
            self.optimizer.zero_grad()
            output = self.model(image)
            loss = self.criterion(output, target)
            loss.backward()
            self.optimizer.step()
            loss_value = loss.item()
            train_loss += loss_value
            if hasattr(tbar, "set_description"):
                tbar.set_description("Train loss: %.3f" % (train_loss / (i + 1)))
            self.writer.add_scalar("train/total_loss_iter", loss_value, i + num_img_tr * epoch)
            if i % (num_img_tr // 10) == 0:
                self.summary.visualize_image(self.writer, self.args.dataset, image, target, output, i + num_img_tr * epoch)
        self.writer.add_scalar("train/total_loss_epoch", train_loss, epoch)
        print("[Epoch: %d, numImages: %5d]" % (epoch, i * self.args.batch_size + image.data.shape[0]))
        print(f"Loss: {train_loss:.3f}")
        if self.args.no_val:
            self.saver.save_checkpoint({"epoch": epoch + 1, "state_dict": self.model.module.state_dict(),
                                        "optimizer": self.optimizer.state_dict(), "best_pred": self.best_pred}, False)
