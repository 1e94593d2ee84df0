"""Oracle-backed stand-ins for the common.quantity._native entry points the integer-simulation modules call,
used ONLY by the CPU test-suite to exercise the Python side of the resident-activation planner
(common/quantity/resident.py: tracing, plan decisions, handles, deferred convolutions) on a box without a GPU.

TEST INFRASTRUCTURE: the product never imports this; on its forward paths there is the HIP library and nothing
else.  Every double computes on host tensors with oracle/fq_oracle.c / NumPy, following the reference's fp32
chain literally (Quantity -> integer conv -> RightShift -> BiasAdd -> Sp -> DeQuantity, NewAdd, nn.ReLU), so a
planned model that matches the un-planned one here is checked against the reference arithmetic, not against
the kernels' own shortcuts.
"""
import contextlib

import numpy as np
import torch

from oracle import fq_oracle as orc


def _np(t):
    return t.detach().cpu().numpy()


def pad16(c):
    return (int(c) + 15) // 16 * 16


def quantize_i8_nhwc(x, ib, cpad=None):
    a = _np(x).astype(np.float32)
    q = orc.quantity(a, ib).astype(np.int8)
    C = a.shape[1]
    cpad = pad16(C) if cpad is None else int(cpad)
    if a.ndim == 2:
        out = np.zeros((a.shape[0], cpad), dtype=np.int8)
        out[:, :C] = q
        return torch.from_numpy(out)
    out = np.zeros((a.shape[0],) + a.shape[2:] + (cpad,), dtype=np.int8)
    out[..., :C] = np.moveaxis(q, 1, -1)
    return torch.from_numpy(out)


def quantize_i8_unfold_w(x, ib, S, stride_w, pad_w, dil_w, cpad2):
    a = orc.quantity(_np(x).astype(np.float32), ib).astype(np.int8)
    N, C, H, W = a.shape
    Q = (W + 2 * pad_w - dil_w * (S - 1) - 1) // stride_w + 1
    out = np.zeros((N, H, Q, cpad2), dtype=np.int8)
    for q in range(Q):
        for s in range(S):
            iw = q * stride_w - pad_w + s * dil_w
            if 0 <= iw < W:
                out[:, :, q, s * C:(s + 1) * C] = np.moveaxis(a[:, :, :, iw], 1, -1)
    return torch.from_numpy(out)


def _conv_fp32(xq, wq, qbias, stride, padding, dilation, rs, ob):
    """int8 NHWC x int8 KRSC -> the reference's fp32 NCHW output (before any ReLU)."""
    x = np.moveaxis(_np(xq).astype(np.int32), -1, 1)                  # NCHW, padded channels are zeros
    w = np.moveaxis(_np(wq).astype(np.int32), -1, 1)                  # KCRS
    acc = orc.conv2d_int(np.ascontiguousarray(x), np.ascontiguousarray(w), tuple(stride), tuple(padding), tuple(dilation))
    return orc.recon_epilogue(acc.astype(np.float32), _np(qbias).astype(np.float32), rs, ob)


def conv2d_i8(xq, wq, qbias, stride, padding, dilation, rs, ob, bitwidth=8):
    linear = xq.dim() == 2
    if linear:
        xq = xq[:, None, None, :]
    y = torch.from_numpy(_conv_fp32(xq, wq, qbias, stride, padding, dilation, rs, ob))
    return y.view(y.shape[0], y.shape[1]) if linear else y


def _to_i8_nhwc(y, bit, kpad):
    q = orc.quantity(y, bit).astype(np.int8)
    out = np.zeros((y.shape[0], y.shape[2], y.shape[3], kpad), dtype=np.int8)
    out[..., :y.shape[1]] = np.moveaxis(q, 1, -1)
    return torch.from_numpy(out)


def conv2d_i8_resident(xq, wq, qbias, stride, padding, dilation, rs, ob, want_f32, want_i8, relu):
    y = _conv_fp32(xq, wq, qbias, stride, padding, dilation, rs, ob)
    if relu:
        y = np.maximum(y, np.float32(0))
    q = _to_i8_nhwc(y, ob, pad16(y.shape[1])) if want_i8 else None    # = the next layer's Quantity(ib = ob)
    return (torch.from_numpy(y) if want_f32 else None), q


def conv2d_i8_stem(x, w_stem, qbias, K, S, stride, padding, ib, rs, ob, relu):
    """The stem layer from the fp32 image: Quantity -> integer conv -> tail -> ReLU -> the next layer's Quantity(ob)."""
    C, R = x.shape[1], w_stem.shape[0]
    w = _np(w_stem).reshape(R, 64, 8, 4)[:, :K, :S, :C].astype(np.int32)                  # [R, K, S, C]
    w = np.ascontiguousarray(np.transpose(w, (1, 3, 0, 2)))                                # [K, C, R, S]
    xq = orc.quantity(_np(x).astype(np.float32), ib).astype(np.int32)
    acc = orc.conv2d_int(np.ascontiguousarray(xq), w, tuple(stride), tuple(padding), (1, 1))
    y = orc.recon_epilogue(acc.astype(np.float32), _np(qbias).astype(np.float32), rs, ob)
    if relu:
        y = np.maximum(y, np.float32(0))
    return _to_i8_nhwc(y, ob, pad16(K))


def _deq(t, g, channels=None):
    a = _np(t).astype(np.float32)
    if channels is not None:
        a = a[..., :channels]
    return orc.dequantity(np.ascontiguousarray(a), g)


def _add_chain(xf, yf, g_wide, want_wide, ib, want_narrow, relu):
    s = orc.add_sat(xf, yf)
    if relu:
        s = np.maximum(s, np.float32(0))
    wide = narrow = None
    if want_wide:
        e = s.astype(np.float64) * 2.0 ** g_wide
        assert np.all(e == np.rint(e)) and np.abs(e).max(initial=0) <= 32768, "exact sum does not fit int16"
        wide = torch.from_numpy(e.astype(np.int16))
    if want_narrow:
        narrow = torch.from_numpy(orc.quantity(s, ib).astype(np.int8))
    return wide, narrow


def add_resident(x, gx, y, gy, want_wide, g_wide, want_narrow, ib, relu):
    assert g_wide == max(0, gx, gy) or not want_wide
    return _add_chain(_deq(x, gx), _deq(y, gy), g_wide, want_wide, ib, want_narrow, relu)


def conv2d_i8_add_resident(xq, wq, qbias, stride, padding, dilation, rs, ob, res, g_res, want_wide, g_wide, want_narrow, ib, relu):
    y = _conv_fp32(xq, wq, qbias, stride, padding, dilation, rs, ob)                          # NCHW fp32
    kpad = pad16(y.shape[1])
    y_nhwc = np.zeros((y.shape[0], y.shape[2], y.shape[3], kpad), dtype=np.float32)
    y_nhwc[..., :y.shape[1]] = np.moveaxis(y, 1, -1)
    return _add_chain(y_nhwc, _deq(res, g_res), g_wide, want_wide, ib, want_narrow, relu)


def block_tail_i8(xq, w3q, qbias3, rs3, ob3, res, g_res, want_wide, g_wide, want_narrow, ib, relu, w1q=None, qbias1=None, rs1=0,
                  relu1=False):
    """conv3 + NewAdd (+ ReLU) + the next conv1 (+ ReLU): the reference's chain, one op after the other."""
    one = ((1, 1), (0, 0), (1, 1))
    wide, narrow = conv2d_i8_add_resident(xq, w3q, qbias3, one[0], one[1], one[2], rs3, ob3, res, g_res, want_wide, g_wide, True, ib, relu)
    q1 = None
    if w1q is not None:
        _, q1 = conv2d_i8_resident(narrow, w1q, qbias1, one[0], one[1], one[2], rs1, 0, False, True, relu1)
    return wide, (narrow if want_narrow else None), q1


def block_tail_proj_i8(xq, w3q, qbias3, rs3, ob3, xpq, wpq, qbiasp, rsp, obp, stride_p, want_wide, g_wide, want_narrow, ib, relu,
                       w1q=None, qbias1=None, rs1=0, relu1=False):
    """The tail of a stage's first block: the projection shortcut as its own convolution, then block_tail_i8 on its output."""
    _, res = conv2d_i8_resident(xpq, wpq, qbiasp, (stride_p, stride_p), (0, 0), (1, 1), rsp, obp, False, True, False)
    return block_tail_i8(xq, w3q, qbias3, rs3, ob3, res, obp, want_wide, g_wide, want_narrow, ib, relu, w1q, qbias1, rs1, relu1)


def dequant_nhwc_to_nchw(q, g, channels):
    return torch.from_numpy(np.ascontiguousarray(np.moveaxis(_deq(q, g, channels), -1, 1)))


def maxpool_i8_nhwc(x, kernel, stride, padding):
    t = x.permute(0, 3, 1, 2).float()
    y = torch.nn.functional.max_pool2d(t, kernel, stride, padding)
    return y.permute(0, 2, 3, 1).contiguous().to(torch.int8)


def avgpool_global_nhwc(q, g, channels):
    x = dequant_nhwc_to_nchw(q, g, channels)
    return torch.nn.functional.avg_pool2d(x, (x.shape[2], x.shape[3]))


def add_sat(a, b, bitwidth=8, out=None):
    return torch.from_numpy(orc.add_sat(_np(a).astype(np.float32), _np(b).astype(np.float32)))


def quantity(x, ib, bitwidth=8, out=None):
    return torch.from_numpy(orc.quantity(_np(x).astype(np.float32), ib))


_DOUBLES = dict(quantize_i8_nhwc=quantize_i8_nhwc, quantize_i8_unfold_w=quantize_i8_unfold_w, conv2d_i8=conv2d_i8,
                conv2d_i8_resident=conv2d_i8_resident, conv2d_i8_stem=conv2d_i8_stem,
                conv2d_i8_add_resident=conv2d_i8_add_resident, add_resident=add_resident, block_tail_i8=block_tail_i8,
                block_tail_proj_i8=block_tail_proj_i8,
                dequant_nhwc_to_nchw=dequant_nhwc_to_nchw, maxpool_i8_nhwc=maxpool_i8_nhwc, avgpool_global_nhwc=avgpool_global_nhwc,
                add_sat=add_sat, quantity=quantity)


@contextlib.contextmanager
def installed():
    """Patch common.quantity._native with the doubles for the duration of a test."""
    from common.quantity import _native
    orc.build()
    saved = {k: getattr(_native, k) for k in _DOUBLES}
    for k, v in _DOUBLES.items():
        setattr(_native, k, v)
    try:
        yield _native
    finally:
        for k, v in saved.items():
            setattr(_native, k, v)
