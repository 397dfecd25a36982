"""GMMN training step of ZS3 on MI355X -- the loop body of zs3/train_pascal_GMMN.py:139-268
(train_context_GMMN.py is identical).

Semantics kept from the reference: frozen-backbone feature pass in train() mode under no_grad; per image,
per class (ascending label order) one generator forward over *all* pixels of the class; an MMD + Adam step
for seen classes of images without unseen pixels, on `batch_size_generator` rows sampled with replacement
(the same indices for fake and real rows); generated features written back for every class; real features
kept for images without unseen pixels (`real_seen_features`); one CE/SGD step of `pred_conv` on the stitched
feature batch; generator_loss_batch divides by len(unique labels) including 255.

What is MI355X-native here: the class masks are resolved with one device sort per image instead of
boolean indexing (no per-class host sync), cat(embd, noise) is fused with the class gather, the MLP runs
on the MFMA conv kernel, the generator backward touches only the sampled rows (rows of an MLP are
independent, so this equals scatter-add + full backward), MMD fwd/bwd are two launches, and the scalar
losses are read back once per step instead of `.item()` per (image, class).

noise="cpu" draws z and the sample indices from the CPU default generator exactly like the reference
(:216,:229) -- bit-parity mode for tests; noise="device" draws z with the counter-based device RNG.
"""
import torch

from . import functional as Fz
from . import ops
from ._lib import I, P, check, lib, require_gpu, stream
from .utils.loss import _MMD  # noqa: F401  (kept importable for users)
import ctypes


def _linear_rows(x, w, b, act=Fz.ACT_NONE, leak=0.2):
    wp = Fz.weight_planes(w, need_t=True)
    n, c = x.shape
    y, _ = ops.conv2d_fwd(x.view(1, 1, n, c), wp, shift=b, act=act, leak=leak)
    return y.view(n, wp.cout), wp


class GMMNStep:
    def __init__(self, model, generator, optimizer, optimizer_generator, criterion, *, seen, unseen, noise_dim=300,
                 embed_dim=300, feature_dim=256, batch_size_generator=128, real_seen_features=True,
                 sigma=(2, 5, 10, 20, 40, 80), noise="device"):
        self.model = model.module if hasattr(model, "module") else model
        self.generator = generator
        self.optimizer, self.optimizer_generator = optimizer, optimizer_generator
        self.criterion = criterion
        self.seen, self.unseen = set(int(s) for s in seen), set(int(u) for u in unseen)
        self.noise_dim, self.embed_dim, self.feature_dim = noise_dim, embed_dim, feature_dim
        self.bsg, self.real_seen_features = batch_size_generator, real_seen_features
        self.sigma = tuple(float(s) for s in sigma)
        self.noise = noise
        if not isinstance(generator.model, torch.nn.Sequential):
            raise NotImplementedError("GMMNStep needs the hidden-layer generator (hidden_size > 0)")

    # ------------------------------------------------------------------ generator pieces
    def _generator_forward(self, x, training):
        lin1, lrelu, drop, lin2 = self.generator.model[0], self.generator.model[1], self.generator.model[2], self.generator.model[3]
        h, _ = _linear_rows(x, lin1.weight, lin1.bias, Fz.ACT_LEAKY, lrelu.negative_slope)
        seed = None
        hd = h
        if training and drop.p > 0:
            seed = Fz.next_seed()
            hd = ops.dropout(h, drop.p, seed)
        out, _ = _linear_rows(hd, lin2.weight, lin2.bias)
        return out, h, hd, seed

    def _generator_backward_rows(self, x, h, hd, seed, ridx, d_out):
        """Gradients of the two Linear layers from the sampled rows only (d_out: [S, feature_dim] for rows ridx)."""
        lin1, lrelu, drop, lin2 = self.generator.model[0], self.generator.model[1], self.generator.model[2], self.generator.model[3]
        s = ridx.shape[0]
        wp2 = Fz.weight_planes(lin2.weight, need_t=True)
        hd_s = ops.gather_rows(hd, ridx)
        dw2 = ops.conv2d_wgrad(d_out.view(1, 1, s, -1), hd_s.view(1, 1, s, -1), wp2.cout, wp2.cin, 1, 1)
        db2 = ops.colstats(d_out)[:, 0].sum(0)
        dhd = ops.conv2d_dgrad(d_out.view(1, 1, s, -1), wp2, (1, s)).view(s, -1)
        if seed is not None:
            dhd = ops.dropout(dhd, drop.p, seed, row_idx=ridx)
        h_s = ops.gather_rows(h, ridx)
        dpre = torch.empty_like(dhd)
        ops.bn_act_bwd(dhd, h_s, None, None, None, None, None, None, dres=dpre, act=Fz.ACT_LEAKY,
                       leak=lrelu.negative_slope, want_dy=False)
        x_s = ops.gather_rows(x, ridx)
        wp1 = Fz.weight_planes(lin1.weight, need_t=True)
        dw1 = ops.conv2d_wgrad(dpre.view(1, 1, s, -1), x_s.view(1, 1, s, -1), wp1.cout, wp1.cin, 1, 1)
        db1 = ops.colstats(dpre)[:, 0].sum(0)
        lin1.weight.grad = dw1.view(lin1.weight.shape)
        lin1.bias.grad = db1
        lin2.weight.grad = dw2.view(lin2.weight.shape)
        lin2.bias.grad = db2

    # ------------------------------------------------------------------ one iteration
    def __call__(self, image, target, embedding):
        require_gpu(image, target, embedding)
        model, dev = self.model, image.device
        b = image.shape[0]
        with torch.no_grad():
            real = ops.nhwc(model.forward_before_class_prediction(image))          # [B, fh, fw, D]
        fh, fw, d = real.shape[1], real.shape[2], real.shape[3]
        npix = fh * fw
        real_rows = real.reshape(b, npix, d)
        fake = torch.empty((b, fh, fw, d), dtype=torch.float32, device=dev)
        fake_rows = fake.view(b, npix, d)
        # labels at feature resolution (nearest), per-image class histogram: one host sync per step
        tgt = ops.nearest_rows(target.contiguous().float(), (fh, fw)).t().contiguous()       # [B, npix]
        tgt_l = tgt.long()
        hist = torch.zeros((b, 256), dtype=torch.int64, device=dev).scatter_add_(1, tgt_l, torch.ones_like(tgt_l))
        order = torch.argsort(tgt_l, dim=1, stable=True)                                      # pixels grouped by class
        hist_h = hist.cpu().tolist()
        training = self.generator.training
        n_mmd = int(sum(1 for i in range(b) for c in range(255) if hist_h[i][c] > 0))
        mmd_losses = torch.zeros(max(n_mmd, 1), dtype=torch.float32, device=dev)
        mmd_slots = []  # (slot, image, n_unique)
        slot = 0
        one = torch.ones(1, dtype=torch.float32, device=dev)
        sig = (ctypes.c_float * len(self.sigma))(*self.sigma)
        for i in range(b):
            classes = [c for c in range(256) if hist_h[i][c] > 0]
            has_unseen = any(c in self.unseen for c in classes)
            emb_rows = ops.nearest_rows(embedding[i].contiguous(), (fh, fw))                 # [npix, embed_dim]
            use_real = self.real_seen_features and not has_unseen
            if use_real:
                fake_rows[i].copy_(real_rows[i])
            else:
                fake_rows[i].zero_()   # ignore-label pixels keep zero features (:198, :242)
            off = 0
            for c in classes:
                n_c = int(hist_h[i][c])
                idx_c = order[i, off:off + n_c]
                off += n_c
                if c == 255:
                    continue
                if self.noise == "cpu":
                    z = torch.rand((n_c, self.noise_dim)).to(dev, non_blocking=True)
                else:
                    z = ops.uniform((n_c, self.noise_dim), Fz.next_seed(), dev)
                x = ops.gather_cat(emb_rows, idx_c, self.embed_dim, z, self.noise_dim, self.embed_dim + self.noise_dim)
                fake_c, h, hd, seed = self._generator_forward(x, training)
                if c in self.seen and not has_unseen:
                    ridx = torch.randint(low=0, high=n_c, size=(self.bsg,)).to(dev, non_blocking=True)
                    s = self.bsg
                    gen_s = ops.gather_rows(fake_c, ridx)
                    real_s = ops.gather_rows(real_rows[i], idx_c[ridx])
                    t = (2 * s + 31) // 32
                    gmat = torch.empty((2 * s, 2 * s), dtype=torch.float32, device=dev)
                    tile = torch.empty(2 * t * t, dtype=torch.float64, device=dev)
                    loss = mmd_losses[slot:slot + 1]
                    check(lib().zs3_mmd_fwd(P(gen_s), I(d), P(real_s), I(d), I(s), I(d), sig, I(len(self.sigma)), P(gmat),
                                            P(tile), P(loss), stream()), "zs3_mmd_fwd")
                    dgen = torch.empty_like(gen_s)
                    check(lib().zs3_mmd_bwd(P(gen_s), I(d), P(real_s), I(d), I(s), I(d), P(gmat), P(loss), P(one), P(dgen),
                                            I(d), stream()), "zs3_mmd_bwd")
                    self._generator_backward_rows(x, h, hd, seed, ridx, dgen)
                    self.optimizer_generator.step()
                    mmd_slots.append((slot, i, len(classes)))
                    slot += 1
                if not use_real:
                    ops.scatter_rows(fake_c, idx_c, fake_rows[i])
        # ---- classifier update on the stitched features (only pred_conv receives gradients)
        self.optimizer.zero_grad()
        out = model.forward_class_prediction(ops.nchw(fake), image.shape[2:])
        closs = self.criterion(out, target)
        closs.backward()
        self.optimizer.step()
        vals = torch.cat((mmd_losses, closs.detach().reshape(1))).cpu()   # the single read-back of the step
        g_batch = 0.0
        for sl, i, nuniq in mmd_slots:
            g_batch += float(vals[sl]) / nuniq
        return g_batch, float(vals[-1]), out


class GMMNTrainer:
    """Trainer with the attribute/driver surface of zs3/train_pascal_GMMN.py:21-311 (`training(epoch, args)`),
    built from injected pieces (model, generator, optimizers, criteria, loaders, scheduler, writer...)."""

    def __init__(self, args, model, generator, optimizer, optimizer_generator, criterion, train_loader, scheduler,
                 writer=None, noise="device"):
        self.args, self.model, self.generator = args, model, generator
        self.optimizer, self.optimizer_generator = optimizer, optimizer_generator
        self.criterion, self.train_loader, self.scheduler, self.writer = criterion, train_loader, scheduler, writer
        self.best_pred = 0.0
        self.step_fn = GMMNStep(model, generator, optimizer, optimizer_generator, criterion,
                                seen=args.seen_classes_idx_metric, unseen=args.unseen_classes_idx_metric,
                                noise_dim=args.noise_dim, embed_dim=args.embed_dim, feature_dim=args.feature_dim,
                                batch_size_generator=args.batch_size_generator,
                                real_seen_features=args.real_seen_features, noise=noise)

    def training(self, epoch, args=None):
        train_loss = 0.0
        self.model.train()
        num_img_tr = len(self.train_loader)
        for i, sample in enumerate(self.train_loader):
            if len(sample["image"]) <= 1:
                continue
            image, target, embedding = sample["image"].cuda(), sample["label"].cuda(), sample["label_emb"].cuda()
            self.scheduler(self.optimizer, i, epoch, self.best_pred)
            g_loss, c_loss, _ = self.step_fn(image, target, embedding)
            train_loss += c_loss
            if self.writer is not None:
                self.writer.add_scalar("train/total_loss_iter", c_loss, i + num_img_tr * epoch)
                self.writer.add_scalar("train/generator_loss", g_loss, i + num_img_tr * epoch)
        if self.writer is not None:
            self.writer.add_scalar("train/total_loss_epoch", train_loss, epoch)
        return train_loss
