"""common.quantity -- the calibrator, the quantize-op modules and their helpers, backed by hand-written
HIP kernels for MI355X (libfq_hip.so, C ABI in include/fq.h).

Drop-in for the reference package of the same import path: reference
quantity/common/quantity/__init__.py:1-6 exports the same 21 public names, so
``from common.quantity import Eltwise, View`` / ``merge_bn`` / ``NewConv2d`` ... keep working and
whole-model pickles (common.quantity.new_quantity_op.NewConv2d, ...) stay loadable.
"""
from . import bit_reader as _bit_reader
from . import distribution_collector as _collector
from . import fabu_layer as _markers
from . import new_quantity_op as _ops
from . import quantizer as _quantizer
from . import utils as _utils

# statistics engine
DistributionCollector = _collector.DistributionCollector
Quantizer = _quantizer.Quantizer

# table files
BitReader = _bit_reader.BitReader

# model preparation
merge_bn = _utils.merge_bn
walk_dirs = _utils.walk_dirs
tid = _utils.tid

# marker layers
Eltwise = _markers.Eltwise
Concat = _markers.Concat
Identity = _markers.Identity
View = _markers.View

# quantize-op modules
RightShift = _ops.RightShift
Sp = _ops.Sp
BiasAdd = _ops.BiasAdd
NewConv2d = _ops.NewConv2d
NewAdd = _ops.NewAdd
NewLinear = _ops.NewLinear
QuanDequan = _ops.QuanDequan
TestConv = _ops.TestConv
TestLinear = _ops.TestLinear
Quantity = _ops.Quantity
DeQuantity = _ops.DeQuantity

__all__ = [
    "DistributionCollector", "Quantizer", "BitReader", "merge_bn", "walk_dirs", "tid",
    "Eltwise", "Concat", "Identity", "View",
    "RightShift", "Sp", "BiasAdd", "NewConv2d", "NewAdd", "NewLinear", "QuanDequan",
    "TestConv", "TestLinear", "Quantity", "DeQuantity",
]
