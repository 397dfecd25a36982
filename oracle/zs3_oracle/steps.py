"""Oracle training steps, LR schedule, metrics and the synthetic batch generator.

Follows zs3/base_trainer.py:5-25 (supervised), zs3/train_pascal_GMMN.py:139-268 (GMMN step; the
train_context_GMMN.py body is identical), zs3/train_context_GMMN_GCNcontext.py:239-457 (GCN-context step),
zs3/utils/lr_scheduler.py:46-76, zs3/utils/metrics.py:35-82.
Test infrastructure only.
"""
import numpy as np
import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------- LR schedule
def poly_lr(base_lr, it, epoch, iters_per_epoch, num_epochs):
    """lr_scheduler.py:46-51 (poly mode): lr = base * (1 - T/N)^0.9 with T = epoch*iters + it."""
    t = epoch * iters_per_epoch + it
    return base_lr * pow(1 - 1.0 * t / (num_epochs * iters_per_epoch), 0.9)


def apply_lr(optimizer, lr):
    """lr_scheduler.py:68-76: group 0 <- lr, every further group <- 10*lr."""
    for gi, group in enumerate(optimizer.param_groups):
        group["lr"] = lr if gi == 0 else lr * 10


# ----------------------------------------------------------------------------- metrics
def confusion_matrix(gt, pred, num_class):
    """metrics.py:73-79: bincount of num_class*gt + pred over 0 <= gt < num_class."""
    gt = np.asarray(gt)
    pred = np.asarray(pred)
    keep = (gt >= 0) & (gt < num_class)
    idx = num_class * gt[keep].astype("int") + pred[keep]
    return np.bincount(idx, minlength=num_class * num_class).reshape(num_class, num_class)


def miou_from_confusion(cm):
    """metrics.py:35-41: nanmean(nan_to_num(diag / (rowsum + colsum - diag)))."""
    cm = np.asarray(cm, dtype=np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        iou = np.diag(cm) / (cm.sum(1) + cm.sum(0) - np.diag(cm))
    return float(np.nanmean(np.nan_to_num(iou))), iou


def pixel_accuracy(cm):
    """metrics.py:11-12: trace / total."""
    cm = np.asarray(cm, dtype=np.float64)
    return float(np.diag(cm).sum() / cm.sum())


def fw_iou(cm):
    """metrics.py:52-59: sum over classes with freq > 0 of freq * iou."""
    cm = np.asarray(cm, dtype=np.float64)
    freq = cm.sum(1) / cm.sum()
    with np.errstate(divide="ignore", invalid="ignore"):
        iou = np.diag(cm) / (cm.sum(1) + cm.sum(0) - np.diag(cm))
    return float((freq[freq > 0] * iou[freq > 0]).sum())


# ----------------------------------------------------------------------------- synthetic data
def nearest_index(out_size, in_size):
    """Source index of F.interpolate(mode='nearest'): floor(i * in/out) (used at train_pascal_GMMN.py:175-186)."""
    scale = np.float32(in_size) / np.float32(out_size)
    return np.minimum(np.floor(np.arange(out_size, dtype=np.float32) * scale).astype(np.int64), in_size - 1)


def make_synthetic_batch(batch, size, num_classes=21, unseen=(10, 14), seed=1, embed_dim=300, with_label_emb=True,
                         grid=9, border=8):
    """image ~ N(0,1); label = nearest-upsampled grid x grid map of random seen
    classes with a `border`-px frame of 255; every 4th image additionally holds one unseen class; the
    embedding table is row-normalised randn(C, embed_dim); label_emb = table[label, 255 -> 0] as
    [B, embed_dim, H, W] (zs3/dataloaders/datasets/base.py:45-51)."""
    g = torch.Generator().manual_seed(seed)
    image = torch.randn(batch, 3, size, size, generator=g)
    seen = torch.tensor([c for c in range(num_classes) if c not in set(unseen)])
    cells = seen[torch.randint(0, len(seen), (batch, grid, grid), generator=g)]
    for b in range(3, batch, 4):
        u = unseen[(b // 4) % len(unseen)]
        cells[b, grid // 2, grid // 2] = u
        cells[b, 0, 1] = u
    src = torch.from_numpy(nearest_index(size, grid))
    label = cells[:, src][:, :, src].float()
    border = min(border, size // 8)
    if border > 0:
        label[:, :border] = 255
        label[:, -border:] = 255
        label[:, :, :border] = 255
        label[:, :, -border:] = 255
    table = torch.randn(num_classes, embed_dim, generator=g)
    table = table / table.norm(dim=1, keepdim=True)
    out = {"image": image, "label": label, "table": table}
    if with_label_emb:
        lab = label.long()
        lab = torch.where(lab == 255, torch.zeros_like(lab), lab)
        out["label_emb"] = table[lab].permute(0, 3, 1, 2).contiguous()
    return out


# ----------------------------------------------------------------------------- steps
def supervised_step(model, optimizer, criterion, image, target):
    """base_trainer.py:16-21: zero_grad, forward, loss, backward, step.  (LR is applied by the caller.)"""
    optimizer.zero_grad()
    out = model(image)
    loss = criterion(out, target)
    loss.backward()
    optimizer.step()
    return float(loss.item()), out.detach()


def gmmn_step(model, generator, optimizer, optimizer_generator, criterion, criterion_generator, image, target,
              embedding, *, seen, unseen, noise_dim=300, embed_dim=300, feature_dim=256, batch_size_generator=128,
              real_seen_features=True):
    """One iteration of train_pascal_GMMN.py:139-268 (LR scheduling is done by the caller).

    Draws noise z and the MMD sample indices from the *CPU default generator*, like the reference
    (:216,:229).  Returns (generator_loss_batch, classifier_loss)."""
    with torch.no_grad():
        real = model.forward_before_class_prediction(image)  # :154-157
    b, _, fh, fw = real.shape
    fake = torch.zeros_like(real)
    g_batch = 0.0
    for i in range(b):
        real_i = real[i].permute(1, 2, 0).reshape(-1, feature_dim)
        tgt_i = F.interpolate(target[i][None, None], size=(fh, fw), mode="nearest").reshape(-1)  # :175-179
        emb_i = F.interpolate(embedding[i][None], size=(fh, fw), mode="nearest")[0]  # :180-189
        emb_i = emb_i.permute(1, 2, 0).reshape(-1, embed_dim)
        fake_i = torch.zeros_like(real_i)
        classes = torch.unique(tgt_i)  # ascending, may contain 255 (:201)
        has_unseen = any(float(c) in [float(u) for u in unseen] for c in classes)  # :204-207
        g_sample = 0.0
        for c in classes:
            if float(c) == 255:
                continue
            optimizer_generator.zero_grad()
            mask = tgt_i == c
            n_c = int(mask.sum())
            z = torch.rand((n_c, noise_dim)).to(real)  # CPU RNG (:216)
            fake_c = generator(emb_i[mask], z)
            if float(c) in [float(s) for s in seen] and not has_unseen:  # :224
                idx = torch.randint(low=0, high=n_c, size=(batch_size_generator,)).to(real.device)  # :229-233
                g_loss = criterion_generator(fake_c[idx], real_i[mask][idx])
                g_sample += float(g_loss.item())
                g_loss.backward()
                optimizer_generator.step()
            fake_i[mask] = fake_c.detach()  # written back even when no MMD step was taken (:242)
        g_batch += g_sample / len(classes)  # divides by the count *including* 255 (:243)
        chosen = real_i if (real_seen_features and not has_unseen) else fake_i  # :244-259
        fake[i] = chosen.reshape(fh, fw, feature_dim).permute(2, 0, 1)
    optimizer.zero_grad()
    out = model.forward_class_prediction(fake.detach(), image.shape[2:])  # :262
    loss = criterion(out, target)
    loss.backward()
    optimizer.step()
    return g_batch, float(loss.item())


def gcn_context_step(model, generator, generator_gcn, optimizer, optimizer_generator, optimizer_generator_gcn, criterion,
                     criterion_generator, image, target, embedding, *, seen, unseen, noise_dim=300, embed_dim=300,
                     feature_dim=256, batch_size_generator=128, real_seen_features=True, context_aware=False,
                     gcn_weight=0.1, gcn_avg_feat=False):
    """One iteration of train_context_GMMN_GCNcontext.py:239-457 (LR scheduling by the caller): the GMMN step plus,
    per image, a cluster-graph generator update (MMD between the GCN generator's cluster features and the clusters' real
    seed features, :399-415) and, per batch, a CE term on the clusters' features through `pred_conv` (:431-454).
    Noise, sample indices and dropout masks come from the CPU default generator in the reference's call order.
    Returns (generator_loss_batch, generator_GCN_loss_batch, classifier_loss)."""
    from .gcn import cluster_graph
    with torch.no_grad():
        real = model.forward_before_class_prediction(image)  # :259-262
    b, _, fh, fw = real.shape
    fake = torch.zeros_like(real)
    g_batch, g_gcn_batch = 0.0, 0.0
    feats_gcn, target_gcn = [], []
    unseen_f, seen_f = [float(u) for u in unseen], [float(s) for s in seen]
    for i in range(b):
        real_map = real[i].permute(1, 2, 0).contiguous()                       # [fh, fw, D] (:280)
        real_i = real_map.view(-1, feature_dim)
        tgt_map = F.interpolate(target[i][None, None], size=(fh, fw), mode="nearest")  # :282-286
        tgt_i = tgt_map.reshape(-1)
        emb_map = F.interpolate(embedding[i][None], size=(fh, fw), mode="nearest")    # :288-297
        classes = torch.unique(tgt_i)
        has_unseen = any(float(c) in unseen_f for c in classes)               # :301-304
        adj, _, labels, emb_gcn, feat_gcn = cluster_graph(tgt_map.numpy().squeeze(), emb_map.numpy().squeeze(),
                                                          real_map.numpy().transpose(2, 0, 1), avg_feat=gcn_avg_feat)  # :307-322
        if adj is not None:
            target_gcn = target_gcn + list(labels)                            # :323-324
        emb_i = emb_map.permute(0, 2, 3, 1).reshape(-1, embed_dim)             # :327-331
        fake_i = torch.zeros_like(real_i)
        g_sample = 0.0
        for c in classes:                                                     # "normal generator" (:337-376)
            if float(c) == 255:
                continue
            optimizer_generator.zero_grad()
            mask = tgt_i == c
            n_c = int(mask.sum())
            if context_aware:
                z = emb_i[tgt_i != 255].mean(0).repeat(n_c, 1)                # :345-348
            else:
                z = torch.rand((n_c, noise_dim))                              # :350
            fake_c = generator(emb_i[mask], z.float())
            if float(c) in seen_f and not has_unseen:                         # :358-373
                idx = torch.randint(low=0, high=n_c, size=(batch_size_generator,))
                g_loss = criterion_generator(fake_c[idx], real_i[mask][idx])
                g_sample += float(g_loss.item())
                g_loss.backward()
                optimizer_generator.step()
            fake_i[mask] = fake_c.detach()                                    # :375
        g_batch += g_sample / len(classes)                                    # :376
        chosen = real_i if (real_seen_features and not has_unseen) else fake_i  # :380-396
        fake[i] = chosen.reshape(fh, fw, feature_dim).permute(2, 0, 1)
        if adj is not None:                                                   # "GCN generator" (:399-425)
            optimizer_generator_gcn.zero_grad()
            emb_gcn_t = torch.from_numpy(np.asarray(emb_gcn)).float()
            z_gcn = torch.rand((emb_gcn_t.shape[0], noise_dim))               # :402
            fake_gcn = generator_gcn(emb_gcn_t, z_gcn.float(), torch.from_numpy(adj))
            real_gcn = torch.from_numpy(np.asarray(feat_gcn)).float()
            if not has_unseen:                                                # :412-418
                g_gcn_loss = criterion_generator(fake_gcn, real_gcn)
                g_gcn_loss.backward()
                optimizer_generator_gcn.step()
                g_gcn_batch += float(g_gcn_loss.item())
            feats_gcn.append((real_gcn if (real_seen_features and not has_unseen) else fake_gcn).detach().numpy())  # :420-427
    optimizer.zero_grad()                                                      # classification (:431-457)
    out = model.forward_class_prediction(fake.detach(), image.shape[2:])
    loss = criterion(out, target)
    loss.backward()
    if feats_gcn:
        f_gcn = torch.from_numpy(np.vstack(feats_gcn).transpose(1, 0).copy())[None, :, :, None]     # [1, D, K, 1] (:438-441)
        out_gcn = model.decoder.forward_class_prediction(f_gcn)
        t_gcn = torch.tensor(np.array(target_gcn), dtype=torch.float32)[None, :, None]              # [1, K, 1] (:445-448)
        (gcn_weight * criterion(out_gcn, t_gcn)).backward()
    optimizer.step()
    return g_batch, g_gcn_batch, float(loss.item())
