"""Post-pass over the tables and quantised-parameter JSON files: align every bias bit with its
layer's output bit, and cap weight bits so that weight + input - output <= MAX_SHIFT.

Drop-in for reference quantity/tools/rewriter.py (BiasReWriter :11, rewrite_bias_dir :38-59,
rewrite_bias_table :61-73, max_shift_limit_weight :75-103, rewrite_weight_dir :105-126,
rewrite_weight_table :128-140): same class, constructor and method signatures, same files.

The rescale is  around(q / 2^old * 2^new)  in fp32 followed by an int8 cast that WRAPS (the
reference casts with astype(np.int8): 128 -> -128, 200 -> -56); that wrap is reproduced, not fixed,
because the JSON bytes are the product's output contract.
"""
import json
import os.path as osp

import numpy as np

from common.quantity import BitReader, walk_dirs
from ._jsonio import dump_int_array

__all__ = ["BiasReWriter"]


def _wrap_int8(values_f32):
    """float (integer valued) -> int8 with two's-complement wrap-around."""
    v = values_f32.astype(np.int64)
    return ((v + 128) % 256 - 128).astype(np.int8)


def _rescale_file(src, dst, old_bit, new_bit):
    with open(src, "r") as fh:
        q = np.array(json.load(fh), dtype=np.float32)
    q = q / 2 ** old_bit * 2 ** new_bit
    dump_int_array(_wrap_int8(np.around(q)), dst)


def _retable(path, suffix_len, known, new_bits):
    with open(path, "r") as fh:
        rows = [ln.strip().split(" ")[:2] for ln in fh.readlines()]
    out = []
    for name, bit in rows:
        layer = name[:-suffix_len]
        if layer in known:
            bit = str(new_bits[layer])
        out.append("{} {}\n".format(name, bit))
    with open(path, "w") as fh:
        fh.writelines(out)


class BiasReWriter(object):

    def __init__(self, weight_dir, bias_dir, output_weight_dir, output_bias_dir, weight_file, feat_file,
                 max_shift_limit=None):
        self._weight_dir = weight_dir
        self._bias_dir = bias_dir
        self._output_weight_dir = output_weight_dir
        self._output_bias_dir = output_bias_dir
        self._weight_file = weight_file
        self._max_shift_limit = max_shift_limit
        self._bit_reader = BitReader(feat_table=feat_file, weight_table=weight_file)

    def get_weight_info(self):
        return self._bit_reader.get_weight_info()

    def get_feat_info(self):
        return self._bit_reader.get_feat_info()

    # ---- bias: bit := output bit of the layer ---------------------------------------------------
    def rewrite_bias_dir(self, old_bias_bits, new_bias_bits):
        for path in walk_dirs(self._bias_dir, file_type=".json"):
            assert path.endswith(".bias.json"), path
            layer = osp.basename(path)[:-len(".bias.json")]
            if layer not in new_bias_bits:
                print("Can't find {} in weight table, but json file exists.".format(layer))
                continue
            _rescale_file(path, osp.join(self._output_bias_dir, osp.basename(path)),
                          old_bias_bits[layer], new_bias_bits[layer])

    def rewrite_bias_table(self, old_bias_bits, new_bias_bits):
        _retable(self._weight_file, len(".bias"), old_bias_bits.keys(), new_bias_bits)

    # ---- weights: weight + input - output <= MAX_SHIFT ------------------------------------------
    def max_shift_limit_weight(self, feat_bits, infeat_bits, weight_bits):
        if self._max_shift_limit is None:
            return True, {}
        changed = False
        capped = {}
        for layer, wbit in weight_bits.items():
            assert layer in feat_bits, "{} not in {}".format(layer, feat_bits)
            assert layer in infeat_bits, "{} not in {}".format(layer, infeat_bits)
            assert len(set(infeat_bits[layer])) == 1, infeat_bits[layer]
            in_bit = int(infeat_bits[layer][0])
            out_bit = feat_bits[layer]
            new_bit = wbit
            if wbit + in_bit - out_bit > self._max_shift_limit:
                new_bit = self._max_shift_limit - in_bit + out_bit
                print("weight bit: {} => {}".format(wbit, new_bit))
                changed = True
            capped[layer] = new_bit
        if not changed:
            print("Nothing needs to change.")
        return changed, capped

    def rewrite_weight_dir(self, old_weight_bits, new_weight_bits):
        paths = walk_dirs(self._weight_dir, file_type=".json")
        print("files path:", paths)
        for path in paths:
            assert path.endswith(".weight.json"), path
            layer = osp.basename(path)[:-len(".weight.json")]
            if layer not in new_weight_bits:
                print("Can't find {} in weight table, but json file exists.".format(layer))
                continue
            _rescale_file(path, osp.join(self._output_weight_dir, osp.basename(path)),
                          old_weight_bits[layer], new_weight_bits[layer])

    def rewrite_weight_table(self, old_weight_bits, new_weight_bits):
        _retable(self._weight_file, len(".weight"), old_weight_bits.keys(), new_weight_bits)
