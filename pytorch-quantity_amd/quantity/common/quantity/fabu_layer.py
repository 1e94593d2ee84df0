"""Marker layers: arithmetic that a model would normally write inline (``a + b``, ``torch.cat``,
``flatten``) exposed as ``nn.Module`` objects, so that forward hooks can see them by name.

Drop-in for reference quantity/common/quantity/fabu_layer.py (Eltwise :5-11, Concat :14-20,
Identity :23-29, View :31-36): same class names, constructor and forward signatures, so pickled
models and model definitions written against the reference import unchanged.  These layers do no
quantisation work themselves; `tools.Reconstruction` swaps Eltwise for `NewAdd`.
"""
import torch
from torch import nn

__all__ = ["Eltwise", "Concat", "Identity", "View"]


class _Marker(nn.Module):
    """Parameter-free module; exists only so that hooks and named_modules() can find the op."""

    def extra_repr(self):
        return "marker"


class Eltwise(_Marker):
    """Element-wise sum of two feature maps (the residual add of a ResNet block)."""

    def forward(self, x, y):
        return torch.add(x, y)


class Concat(_Marker):
    """Concatenation of two feature maps, channel axis by default."""

    def forward(self, x, y, dim=1):
        return torch.cat([x, y], dim)


class Identity(_Marker):
    """Pass-through; what a folded BatchNorm is replaced with (see utils.merge_bn)."""

    def forward(self, x):
        return x


class View(_Marker):
    """Flatten everything after the batch axis into a fresh tensor (the hook needs a new tensor,
    not an alias of the pooling output)."""

    def forward(self, x):
        return x.reshape(x.shape[0], -1).clone()
