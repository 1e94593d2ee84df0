"""tools -- drop-in for the reference package quantity/tools (tools/__init__.py:1-3)."""
from .pytorch_quantizer import Quantity
from .reconstruction import Reconstruction
from .rewriter import BiasReWriter
