"""Byte-identical writer for the quantised-parameter JSON files.

The reference writes nested Python lists with ``json.dump(content, fh, indent=4)``
(quantity/tools/pytorch_quantizer.py:663-669, rewriter.py:57-59), which is the dominant cost of
weight_quantize (42-47 s on ResNet-18).  The text format is fully determined by the array's shape
and integer values, so it is generated directly from the ndarray: one "%d" per line, 4 spaces per
nesting level, "[" / "]" on their own lines, "," after every item but the last, no trailing newline.
dump_int_array() streams it from native code (fq_json_dump_i32 in libfq_hip.so); dumps_int_array()
is the same grammar in NumPy, kept as the readable specification and cross-checked in
tests/test_jsonio.py against both json.dump and the native writer, byte for byte.
"""
import numpy as np

from common.quantity import _native

__all__ = ["dump_int_array", "dumps_int_array"]


def dumps_int_array(a, indent=4):
    a = np.asarray(a)
    if a.ndim == 0:
        return "%d" % int(a)
    if a.size == 0:
        # json.dump of nested empty lists: "[]" at the first empty level
        def empty(shape, level):
            if shape[0] == 0:
                return "[]"
            pad = " " * (indent * (level + 1))
            inner = empty(shape[1:], level + 1)
            return "[\n" + ",\n".join(pad + inner for _ in range(shape[0])) + "\n" + " " * (indent * level) + "]"
        return empty(a.shape, 0)
    nd = a.ndim
    flat = a.reshape(-1).astype(np.int64)
    # innermost rows: every scalar on its own line at depth nd
    pad = " " * (indent * nd)
    last = a.shape[-1]
    txt = np.char.mod("%d", flat)
    rows = txt.reshape(-1, last)
    sep = ",\n" + pad
    row_strs = [pad + sep.join(r) for r in rows]          # body of each innermost list

    def wrap(items, level):
        # items: bodies already indented at depth level+1; returns the list text WITHOUT leading pad
        return "[\n" + items + "\n" + " " * (indent * level) + "]"

    cur = [wrap(r, nd - 1) for r in row_strs]
    for level in range(nd - 2, -1, -1):
        group = a.shape[level]
        p = " " * (indent * (level + 1))
        joiner = ",\n" + p
        cur = [wrap(p + joiner.join(cur[i:i + group]), level) for i in range(0, len(cur), group)]
    assert len(cur) == 1
    return cur[0]


def dump_int_array(a, path, indent=4):
    _native.json_dump_i32(np.asarray(a), path, indent)
