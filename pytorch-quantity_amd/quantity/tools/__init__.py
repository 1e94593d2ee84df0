"""tools -- orchestration layer of the drop-in (calibration run, table rewriting, model rebuilding).

Same three public names as the reference package quantity/tools (reference tools/__init__.py:1-3);
``from tools import reconstruction`` (the submodule, as resnet_reconstruction.py:14 does) works too.
"""
from . import pytorch_quantizer, reconstruction, rewriter

Quantity = pytorch_quantizer.Quantity
Reconstruction = reconstruction.Reconstruction
BiasReWriter = rewriter.BiasReWriter

__all__ = ["Quantity", "Reconstruction", "BiasReWriter"]
