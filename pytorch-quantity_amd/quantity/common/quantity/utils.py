"""Model-preparation helpers: BatchNorm folding, directory walk, tensor fingerprint.

Drop-in for reference quantity/common/quantity/utils.py (merge_bn :7-65, walk_dirs :67-77,
tid :80-97).  merge_bn changes the weights every table is computed from, so its arithmetic is the
reference's, operation for operation (golden G8 pins it bit for bit).
"""
import os

import torch
from torch import nn

from .fabu_layer import Identity

__all__ = ["merge_bn", "walk_dirs", "tid"]


def _replace_submodule(root, dotted_name, new_module):
    parent = root
    parts = dotted_name.split(".")
    for p in parts[:-1]:
        parent = getattr(parent, p)
    parent.add_module(parts[-1], new_module)


def merge_bn(model, device="cpu"):
    """Fold every BatchNorm2d into the Conv2d registered just before it and replace the BN by
    `Identity`.  Pairing rule (reference utils.py:12-16): walk named_modules() in registration
    order, remember the last Conv2d seen, fold the next BatchNorm2d into it.

        scale = gamma / sqrt(running_var + 1e-5)
        W' = scale[:, None, None, None] * W
        b' = scale * (b - running_mean) + beta            (b = 0 if the conv has no bias)

    `device` is accepted for signature compatibility ('cpu' | 'cuda'); tensors are folded on
    whatever device they already live on.
    """
    pending_conv = None
    folded = []
    for name, layer in list(model.named_modules()):
        kind = type(layer).__name__
        if kind == "Conv2d":
            pending_conv = layer
        elif kind == "BatchNorm2d":
            assert pending_conv is not None, "Please put bn right after the conv in your __init__()."
            conv = pending_conv
            w = conv.weight.data
            assert w is not None, "The conv weight can`t be None"
            gamma, beta = layer.weight.data, layer.bias.data
            mean, var = layer.running_mean, layer.running_var
            b = conv.bias.data if conv.bias is not None else torch.zeros(conv.out_channels, device=w.device,
                                                                        dtype=w.dtype)
            scale = gamma / torch.sqrt(var + 1e-5)
            scale = scale.to(w.device)
            b = b.to(w.device)
            conv.weight = nn.Parameter(scale.view(scale.size()[0], 1, 1, 1) * w)
            conv.bias = nn.Parameter(scale * (b - mean.to(w.device)) + beta.to(w.device))
            folded.append(name)
            pending_conv = None
    for name in folded:
        _replace_submodule(model, name, Identity())
        print("The layer change: {} ==>Identity".format(name))
    return model


def walk_dirs(dir_name, file_type=None):
    """All file paths under dir_name (optionally only those ending in file_type)."""
    found = []
    for root, _dirs, files in os.walk(dir_name):
        for fname in files:
            path = root + "/" + fname
            if file_type is None or not file_type or path.endswith(file_type):
                found.append(path)
    return found


def tid(tensor):
    """Value fingerprint of a tensor (four 4-digit fields: max+min and mean of the last-axis
    slice 0 and of the whole tensor).  The reference uses it to recover graph edges; this build
    tracks tensor identity instead and keeps `tid` as the fallback matcher and for API parity."""
    x = tensor.detach()
    first = x[..., 0]

    def field(v):
        return str(int(v * 1e4 % 1e4))

    return "".join([
        field((first.max() + first.min()).item()),
        field((x.max() + x.min()).item()),
        field(first.mean().item()),
        field(x.mean().item()),
    ])
