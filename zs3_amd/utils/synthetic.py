"""Synthetic ZS3 batches generated directly on the device: image ~ N(0,1); label = a
nearest-upsampled grid of random seen classes with a border of 255, every 4th image holding one unseen class;
row-normalised random embedding table; label_emb = table[label] as [B, embed_dim, H, W]
(zs3/dataloaders/datasets/base.py:45-51 builds the same tensor on the CPU per sample)."""
import torch


def make_batch(batch, size, num_classes=21, unseen=(10, 14), seed=1, embed_dim=300, with_label_emb=False, grid=9,
               border=8, device="cuda"):
    g = torch.Generator(device="cpu").manual_seed(seed)
    image = torch.randn(batch, 3, size, size, generator=g)
    seen = torch.tensor([c for c in range(num_classes) if c not in set(unseen)])
    cells = seen[torch.randint(0, len(seen), (batch, grid, grid), generator=g)]
    for b in range(3, batch, 4):
        u = unseen[(b // 4) % len(unseen)]
        cells[b, grid // 2, grid // 2] = u
        cells[b, 0, 1] = u
    src = torch.clamp((torch.arange(size, dtype=torch.float32) * (grid / size)).floor().long(), max=grid - 1)
    label = cells[:, src][:, :, src].float()
    border = min(border, size // 8)
    if border > 0:
        label[:, :border] = 255
        label[:, -border:] = 255
        label[:, :, :border] = 255
        label[:, :, -border:] = 255
    table = torch.randn(num_classes, embed_dim, generator=g)
    table = table / table.norm(dim=1, keepdim=True)
    out = {"image": image.to(device), "label": label.to(device), "table": table.to(device)}
    if with_label_emb:
        lab = out["label"].long()
        lab = torch.where(lab == 255, torch.zeros_like(lab), lab)
        out["label_emb"] = out["table"][lab].permute(0, 3, 1, 2).contiguous()
    return out
