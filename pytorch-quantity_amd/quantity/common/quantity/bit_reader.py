"""Parsers for the two text artefacts of a calibration run.

Drop-in for reference quantity/common/quantity/bit_reader.py (BitReader :7-9, get_feat_info
:20-35, get_weight_info :37-64).  File grammar (written by tools.Quantity):

  feat.table    one line per cared tensor:   "<module name> <out bit> [<in bit> ...]"
                first line                   "image <bit>"
  weight.table  one line per parameter:      "<param name ending in .weight|.bias> <bit>"

Deliberate difference: the reference runs ``eval`` on the bit field; here the
field is parsed as a number only.
"""
from collections import OrderedDict

__all__ = ["BitReader"]


def _to_int(text):
    try:
        return int(text)
    except ValueError:
        return int(float(text))


def _rows(path):
    with open(path, "r") as fh:
        for raw in fh:
            fields = raw.strip().split(" ")
            if fields and fields[0]:
                yield fields


class BitReader(object):

    def __init__(self, feat_table=None, weight_table=None):
        self._feat_table = feat_table
        self._weight_table = weight_table

    def get_feat_info(self):
        """-> (feat_bits {name: int}, infeat_bits {name: [str, ...]}) in file order."""
        assert self._feat_table, "BitReader was built without a feat table"
        out_bits, in_bits = {}, {}
        for fields in _rows(self._feat_table):
            name = fields[0]
            out_bits[name] = _to_int(fields[1])
            in_bits[name] = fields[2:]
        print("feat count:", len(out_bits))
        return out_bits, in_bits

    def get_weight_info(self):
        """-> (weight_bits, bias_bits): OrderedDicts keyed by layer name (suffix stripped)."""
        assert self._weight_table, "BitReader was built without a weight table"
        weight_bits, bias_bits = OrderedDict(), OrderedDict()
        for fields in _rows(self._weight_table):
            param, bit = fields
            bit = _to_int(bit)
            if param.endswith(".weight"):
                weight_bits[param[:-len(".weight")]] = bit
            elif param.endswith(".bias"):
                bias_bits[param[:-len(".bias")]] = bit
            else:
                print("Unknow layer name {}".format(param))
        print("weight count:", len(weight_bits))
        print("bias count:", len(bias_bits))
        return weight_bits, bias_bits
