"""Synthetic stand-ins for the datasets the reference scripts download (no network / torchvision here):
a list of (images, labels) batches shaped like a DataLoader's output (PRE_PROCESS.IMG = 1)."""
import torch


def batches(n_batches, batch, shape, seed=1234, device="cpu"):
    out = []
    for i in range(n_batches):
        g = torch.Generator(device=device).manual_seed(seed + i)
        out.append((torch.randn(batch, *shape, generator=g, device=device),
                    torch.zeros(batch, dtype=torch.long, device=device)))
    return out
