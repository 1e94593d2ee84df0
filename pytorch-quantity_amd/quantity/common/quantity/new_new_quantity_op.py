"""Alias for the module name the reference's documentation uses: README.md:52 and BASELINE.json call the
quantize-op file `new_new_quantity_op.py`, while the package actually imports `new_quantity_op`
(reference quantity/common/quantity/__init__.py:6).  Both names resolve to the SAME module object, so
classes pickled under either path are the same classes."""
import sys

from . import new_quantity_op as _real

sys.modules[__name__] = _real
