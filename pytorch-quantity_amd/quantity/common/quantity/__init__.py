"""common.quantity -- drop-in for the reference package of the same import path
(reference quantity/common/quantity/__init__.py:1-6 exports the same 21 names), backed by
hand-written HIP kernels for MI355X (libfq_hip.so, include/fq.h)."""
from .distribution_collector import DistributionCollector
from .quantizer import Quantizer
from .bit_reader import BitReader
from .utils import merge_bn, walk_dirs, tid
from .fabu_layer import Eltwise, Concat, Identity, View
from .new_quantity_op import (RightShift, Sp, BiasAdd, NewConv2d, NewAdd, NewLinear, QuanDequan, TestConv,
                              TestLinear, Quantity, DeQuantity)
