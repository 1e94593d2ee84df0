"""ctypes view of oracle/libfq_oracle.so -- TEST INFRASTRUCTURE ONLY (see fq_oracle.c header).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libfq_oracle.so")

BINS = 2048
KL_CANDIDATES = 1920


def build(force=False):
    src = os.path.join(_HERE, "fq_oracle.c")
    hdr = os.path.join(_HERE, "..", "include", "fq_log.h")
    if (not force and os.path.exists(_LIB) and os.path.getmtime(_LIB) >= os.path.getmtime(src)
            and os.path.getmtime(_LIB) >= os.path.getmtime(hdr)):
        return _LIB
    subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB)
        f32p = ctypes.POINTER(ctypes.c_float)
        f64p = ctypes.POINTER(ctypes.c_double)
        i64p = ctypes.POINTER(ctypes.c_int64)
        i32p = ctypes.POINTER(ctypes.c_int32)
        u64 = ctypes.c_uint64
        ci = ctypes.c_int
        L.orc_absmax.restype = ctypes.c_float
        L.orc_absmax.argtypes = [f32p, u64, ctypes.c_float]
        L.orc_interval.restype = ctypes.c_float
        L.orc_interval.argtypes = [ctypes.c_float, ci]
        L.orc_hist2048.restype = None
        L.orc_hist2048.argtypes = [f32p, u64, ctypes.c_float, i64p]
        L.orc_interval_n.restype = ctypes.c_float
        L.orc_interval_n.argtypes = [ctypes.c_float, ci, ci]
        L.orc_hist_n.restype = None
        L.orc_hist_n.argtypes = [f32p, u64, ctypes.c_float, i64p, ci]
        L.orc_normalize_i64_n.restype = None
        L.orc_normalize_i64_n.argtypes = [i64p, f64p, ci]
        L.orc_normalize_f64_n.restype = None
        L.orc_normalize_f64_n.argtypes = [f64p, f64p, ci]
        L.orc_kl_threshold_n.restype = ci
        L.orc_kl_threshold_n.argtypes = [f64p, ci, f64p, ci]
        L.orc_np_sum.restype = ctypes.c_double
        L.orc_np_sum.argtypes = [f64p, ctypes.c_int64]
        L.orc_normalize_i64.restype = None
        L.orc_normalize_i64.argtypes = [i64p, f64p]
        L.orc_normalize_f64.restype = None
        L.orc_normalize_f64.argtypes = [f64p, f64p]
        L.orc_kl_threshold.restype = ci
        L.orc_kl_threshold.argtypes = [f64p, f64p, ci]
        L.orc_bits_from_threshold.restype = ci
        L.orc_bits_from_threshold.argtypes = [ci, ctypes.c_float, f32p]
        L.orc_bits_from_absmax.restype = ci
        L.orc_bits_from_absmax.argtypes = [ctypes.c_float]
        for name in ("orc_quantity", "orc_quandequan", "orc_rightshift"):
            getattr(L, name).restype = None
            getattr(L, name).argtypes = [f32p, f32p, u64, ci, ci]
        L.orc_dequantity.restype = None
        L.orc_dequantity.argtypes = [f32p, f32p, u64, ci]
        L.orc_sp.restype = None
        L.orc_sp.argtypes = [f32p, f32p, u64, ci]
        L.orc_add_sat.restype = None
        L.orc_add_sat.argtypes = [f32p, f32p, f32p, u64, ci]
        L.orc_recon_epilogue.restype = None
        L.orc_recon_epilogue.argtypes = [f32p, f32p, f32p, u64, u64, u64, ci, ci, ci]
        L.orc_quantize_param_i32.restype = None
        L.orc_quantize_param_i32.argtypes = [f32p, i32p, u64, ci]
        L.orc_conv2d_int.restype = None
        L.orc_conv2d_int.argtypes = [i32p, i32p, i64p] + [ci] * 14
        _lib = L
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


def absmax(x, running=0.0):
    x = _f32(x).ravel()
    return np.float32(lib().orc_absmax(_p(x, ctypes.c_float), x.size, np.float32(running)))


def interval(max_val, statistic=1, bins=BINS):
    return np.float32(lib().orc_interval_n(np.float32(max_val), int(statistic), int(bins)))


def hist2048(x, iv, hist=None, bins=None):
    """The |x| histogram with INTERVAL_NUM = bins (default 2048; with `hist` given: its length)."""
    x = _f32(x).ravel()
    if hist is None:
        hist = np.zeros(BINS if bins is None else int(bins), dtype=np.int64)
    assert hist.dtype == np.int64 and hist.flags.c_contiguous and (bins is None or hist.size == bins)
    lib().orc_hist_n(_p(x, ctypes.c_float), x.size, np.float32(iv), _p(hist, ctypes.c_int64), hist.size)
    return hist


def np_sum(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return lib().orc_np_sum(_p(a, ctypes.c_double), a.size)


def normalize(hist):
    """quantizer.py:95-96 for a histogram of any length (INTERVAL_NUM bins)."""
    n = int(np.asarray(hist).size)
    p = np.empty(n, dtype=np.float64)
    if np.asarray(hist).dtype == np.float64:
        h = np.ascontiguousarray(hist, dtype=np.float64)
        lib().orc_normalize_f64_n(_p(h, ctypes.c_double), _p(p, ctypes.c_double), n)
    else:
        h = np.ascontiguousarray(hist, dtype=np.int64)
        lib().orc_normalize_i64_n(_p(h, ctypes.c_int64), _p(p, ctypes.c_double), n)
    return p


def kl_threshold(p, want_curve=False, use_fq_log=False):
    """quantizer.py:98-167 on a normalised distribution of any length in (128, 4096]: the sweep runs t = 128 .. len(p) - 1."""
    p = np.ascontiguousarray(p, dtype=np.float64)
    curve = np.empty(p.size - 128, dtype=np.float64) if want_curve else None
    t = lib().orc_kl_threshold_n(_p(p, ctypes.c_double), p.size,
                                 _p(curve, ctypes.c_double) if want_curve else None,
                                 1 if use_fq_log else 0)
    assert t >= 0, "orc_kl_threshold_n: unsupported length %d" % p.size
    return (t, curve) if want_curve else t


def bits_from_threshold(thr, iv):
    tv = ctypes.c_float()
    b = lib().orc_bits_from_threshold(int(thr), np.float32(iv), ctypes.byref(tv))
    return b, np.float32(tv.value)


def bits_from_absmax(m):
    return lib().orc_bits_from_absmax(np.float32(m))


def _unary(fn, x, *ints):
    x = _f32(x)
    y = np.empty_like(x)
    fn(_p(x, ctypes.c_float), _p(y, ctypes.c_float), x.size, *ints)
    return y


def quantity(x, ib, bitwidth=8):
    return _unary(lib().orc_quantity, x, ib, bitwidth)


def dequantity(x, ob):
    return _unary(lib().orc_dequantity, x, ob)


def sp(x, bitwidth=8):
    return _unary(lib().orc_sp, x, bitwidth)


def rightshift(x, rs, bitwidth=8):
    return _unary(lib().orc_rightshift, x, rs, bitwidth)


def quandequan(x, bit, bitwidth=8):
    return _unary(lib().orc_quandequan, x, bit, bitwidth)


def add_sat(a, b, bitwidth=8):
    a = _f32(a)
    b = _f32(b)
    y = np.empty_like(a)
    lib().orc_add_sat(_p(a, ctypes.c_float), _p(b, ctypes.c_float), _p(y, ctypes.c_float), a.size, bitwidth)
    return y


def recon_epilogue(acc, qbias, rs, ob, bitwidth=8):
    acc = _f32(acc)
    qbias = _f32(qbias)
    outer, C = acc.shape[0], acc.shape[1]
    inner = int(np.prod(acc.shape[2:])) if acc.ndim > 2 else 1
    y = np.empty_like(acc)
    lib().orc_recon_epilogue(_p(acc, ctypes.c_float), _p(qbias, ctypes.c_float), _p(y, ctypes.c_float),
                             outer, C, inner, rs, ob, bitwidth)
    return y


def quantize_param_i32(w, bit):
    w = _f32(w)
    q = np.empty(w.shape, dtype=np.int32)
    lib().orc_quantize_param_i32(_p(w, ctypes.c_float), _p(q, ctypes.c_int32), w.size, bit)
    return q


def conv2d_int(x, w, stride=(1, 1), pad=(0, 0), dil=(1, 1), groups=1):
    x = np.ascontiguousarray(x, dtype=np.int32)
    w = np.ascontiguousarray(w, dtype=np.int32)
    N, C, H, W = x.shape
    K, Cg, R, S = w.shape
    P = (H + 2 * pad[0] - dil[0] * (R - 1) - 1) // stride[0] + 1
    Q = (W + 2 * pad[1] - dil[1] * (S - 1) - 1) // stride[1] + 1
    acc = np.empty((N, K, P, Q), dtype=np.int64)
    lib().orc_conv2d_int(_p(x, ctypes.c_int32), _p(w, ctypes.c_int32), _p(acc, ctypes.c_int64),
                         N, C, H, W, K, R, S, stride[0], stride[1], pad[0], pad[1], dil[0], dil[1], groups)
    return acc
