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


def randomize_bn(model, seed=0):
    """A freshly constructed net has running_mean = 0 and beta = 0, so every folded conv bias is exactly
    0 and weight_quantize() stops at log2(0) -- as the reference does.  Stand-in statistics for demos."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)
                m.weight.copy_(torch.rand(m.weight.shape, generator=g) + 0.5)
                m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
    return model
