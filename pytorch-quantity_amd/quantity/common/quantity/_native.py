"""ctypes binding of libfq_hip.so (include/fq.h) for torch.cuda tensors.

This is the only place the Python drop-in modules touch native code.  There is deliberately no CPU
fallback: if the HIP library is missing or a tensor is not on the GPU, the call raises.
PyTorch is used here for device memory and the current HIP stream only.
"""
import ctypes
import os

import numpy as np
import torch

BINS = 2048
SUPPORTED_BINS = (512, 1024, 2048, 4096)      # INTERVAL_NUM the histogram / KL kernels are instantiated for (fq.h: fq_hist_seg_n)
KL_CANDIDATES = 1920

_PKG_ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", ".."))
LIB_PATH = os.path.join(_PKG_ROOT, "lib", "libfq_hip.so")


class FqError(RuntimeError):
    pass


class _Seg(ctypes.Structure):
    _fields_ = [("ptr", ctypes.c_void_p), ("n", ctypes.c_uint64), ("row", ctypes.c_int32),
                ("reserved", ctypes.c_int32)]


class _PairSeg(ctypes.Structure):
    _fields_ = [("a", ctypes.c_void_p), ("b", ctypes.c_void_p), ("relu_out", ctypes.c_void_p), ("n", ctypes.c_size_t),
                ("row_a", ctypes.c_int32), ("row_sum", ctypes.c_int32)]


CHAIN_MAX = 6                # FQ_CHAIN_MAX


class _ChainSeg(ctypes.Structure):
    _fields_ = [("head", ctypes.c_void_p), ("y", ctypes.c_void_p * CHAIN_MAX), ("n", ctypes.c_size_t), ("len", ctypes.c_int32),
                ("row_y", ctypes.c_int32 * CHAIN_MAX), ("row_sum", ctypes.c_int32 * CHAIN_MAX)]


class _ChanSeg(ctypes.Structure):
    _fields_ = [("ptr", ctypes.c_void_p), ("N", ctypes.c_int32), ("C", ctypes.c_int32), ("HW", ctypes.c_int64),
                ("row0", ctypes.c_int32), ("reserved", ctypes.c_int32)]


_lib = None


def lib():
    """Load libfq_hip.so or fail loudly."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise FqError(
            "libfq_hip.so not found at %s. Build it with `make -C pytorch-quantity_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback." % LIB_PATH)
    L = ctypes.CDLL(LIB_PATH)
    vp, ci, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
    L.fq_version.restype = ci
    L.fq_status_string.restype = ctypes.c_char_p
    L.fq_status_string.argtypes = [ci]
    L.fq_last_hip_error.restype = ci
    L.fq_conv2d_i8_last_variant.restype = ci
    L.fq_conv2d_i8_last_variant.argtypes = []
    L.fq_absmax_seg.restype = ci
    L.fq_absmax_seg.argtypes = [ctypes.POINTER(_Seg), ci, vp, vp]
    L.fq_hist2048_seg.restype = ci
    L.fq_hist2048_seg.argtypes = [ctypes.POINTER(_Seg), ci, vp, vp, vp]
    L.fq_hist_seg_n.restype = ci
    L.fq_hist_seg_n.argtypes = [ctypes.POINTER(_Seg), ci, vp, vp, ci, vp]
    L.fq_kl_workspace_bytes_n.restype = sz
    L.fq_kl_workspace_bytes_n.argtypes = [ci, ci]
    L.fq_kl_threshold_n.restype = ci
    L.fq_kl_threshold_n.argtypes = [vp, ci, ci, vp, vp, vp, sz, vp]
    L.fq_hist2048_chain_seg.restype = ci
    L.fq_hist2048_chain_seg.argtypes = [ctypes.POINTER(_ChainSeg), ci, vp, vp, vp]
    L.fq_hist2048_pair_seg.restype = ci
    L.fq_hist2048_pair_seg.argtypes = [ctypes.POINTER(_PairSeg), ci, vp, vp, vp]
    L.fq_absmax_chan.restype = ci
    L.fq_absmax_chan.argtypes = [ctypes.POINTER(_ChanSeg), ci, vp, vp]
    L.fq_hist2048_chan.restype = ci
    L.fq_hist2048_chan.argtypes = [ctypes.POINTER(_ChanSeg), ci, vp, vp, vp]
    L.fq_kl_workspace_bytes.restype = sz
    L.fq_kl_workspace_bytes.argtypes = [ci]
    L.fq_kl_threshold.restype = ci
    L.fq_kl_threshold.argtypes = [vp, ci, vp, vp, vp, sz, vp]
    L.fq_kl_threshold_ex.restype = ci
    L.fq_kl_threshold_ex.argtypes = [vp, ci, vp, vp, vp, vp, ci, vp, sz, vp]
    L.fq_bits_from_threshold.restype = ci
    L.fq_bits_from_threshold.argtypes = [vp, vp, ci, vp, vp]
    L.fq_bits_from_absmax.restype = ci
    L.fq_bits_from_absmax.argtypes = [vp, ci, vp]
    for name in ("fq_quandequan_f32", "fq_quantity_f32", "fq_rightshift_f32"):
        getattr(L, name).restype = ci
        getattr(L, name).argtypes = [vp, vp, sz, ci, ci, vp]
    L.fq_dequantity_f32.restype = ci
    L.fq_dequantity_f32.argtypes = [vp, vp, sz, ci, vp]
    L.fq_sp_f32.restype = ci
    L.fq_sp_f32.argtypes = [vp, vp, sz, ci, vp]
    L.fq_add_sat_f32.restype = ci
    L.fq_add_sat_f32.argtypes = [vp, vp, vp, sz, ci, vp]
    L.fq_recon_epilogue_f32.restype = ci
    L.fq_recon_epilogue_f32.argtypes = [vp, vp, vp, sz, sz, sz, ci, ci, ci, vp]
    L.fq_quantize_param_i32.restype = ci
    L.fq_quantize_param_i32.argtypes = [vp, vp, sz, ci, vp]
    L.fq_quantize_i8_nhwc.restype = ci
    L.fq_quantize_i8_nhwc.argtypes = [vp, vp, ci, ci, ci, ci, ci, vp]
    L.fq_quantize_i8_unfold_w.restype = ci
    L.fq_quantize_i8_unfold_w.argtypes = [vp, vp] + [ci] * 10 + [vp]
    L.fq_conv2d_i8.restype = ci
    L.fq_conv2d_i8.argtypes = [vp, vp, vp, vp] + [ci] * 16 + [vp]
    L.fq_conv2d_i8_resident.restype = ci
    L.fq_conv2d_i8_resident.argtypes = [vp, vp, vp, vp, vp] + [ci] * 17 + [vp]
    L.fq_bias_add_absmax_f32.restype = ci
    L.fq_bias_add_absmax_f32.argtypes = [vp, vp, ci, ci, ci, vp, vp, vp]
    L.fq_add_absmax_f32.restype = ci
    L.fq_add_absmax_f32.argtypes = [vp, vp, vp, sz, vp, vp, vp]
    L.fq_bias_add_hist_f32.restype = ci
    L.fq_bias_add_hist_f32.argtypes = [vp, vp, ci, ci, ci, vp, vp, vp, vp]
    L.fq_add_hist_f32.restype = ci
    L.fq_add_hist_f32.argtypes = [vp, vp, vp, sz, vp, vp, vp, vp]
    L.fq_conv1x1_f32.restype = ci
    L.fq_conv_f32_workspace_bytes.restype = sz
    L.fq_conv_f32_workspace_bytes.argtypes = []
    L.fq_conv1x1_f32.argtypes = [vp, vp, vp, vp, vp] + [ci] * 6 + [vp, vp, vp, vp, sz, vp]
    L.fq_conv_kxk_f32.restype = ci
    L.fq_conv_kxk_f32.argtypes = [vp, vp, vp, vp, vp] + [ci] * 9 + [vp, vp, vp, vp, sz, vp]
    L.fq_conv3x3_wino_f32_supported.restype = ci
    L.fq_conv3x3_wino_f32_supported.argtypes = [ci] * 5
    L.fq_conv3x3_wino_f32_packed_floats.restype = sz
    L.fq_conv3x3_wino_f32_packed_floats.argtypes = [ci, ci]
    L.fq_conv3x3_wino_f32_pack.restype = ci
    L.fq_conv3x3_wino_f32_pack.argtypes = [vp, vp, ci, ci, vp]
    L.fq_conv3x3_wino_f32.restype = ci
    L.fq_conv3x3_wino_f32.argtypes = [vp, vp, vp, vp, vp] + [ci] * 5 + [vp, vp, vp, vp]
    L.fq_conv3x3_wino_qd_f32.restype = ci
    L.fq_conv3x3_wino_qd_f32.argtypes = [vp, vp, vp, vp] + [ci] * 7 + [vp]
    L.fq_conv_stem_f32_packed_rows.restype = ci
    L.fq_conv_stem_f32_packed_rows.argtypes = [ci, ci, ci]
    L.fq_conv_stem_f32.restype = ci
    L.fq_conv_stem_f32.argtypes = [vp, vp, vp, vp, vp] + [ci] * 9 + [vp, vp, vp, vp]
    L.fq_read_npy_batch_f32.restype = ci
    L.fq_read_npy_batch_f32.argtypes = [vp, ci, ctypes.c_char_p, ctypes.c_size_t, vp, ctypes.c_size_t, ci, vp]
    L.fq_conv1x1_add_f32.restype = ci
    L.fq_conv1x1_add_f32.argtypes = [vp] * 7 + [ci] * 6 + [vp, vp, vp, sz, vp]
    L.fq_conv1x1_add_hist_f32.restype = ci
    L.fq_conv1x1_add_hist_f32.argtypes = [vp] * 5 + [ci] * 6 + [vp, vp, vp, vp, vp, sz, vp]
    L.fq_conv1x1_qd_f32.restype = ci
    L.fq_conv1x1_qd_f32.argtypes = [vp, vp, vp, vp] + [ci] * 8 + [vp, sz, vp]
    L.fq_conv1x1_sb_packed_bytes.restype = sz
    L.fq_conv1x1_sb_packed_bytes.argtypes = [ci, ci]
    L.fq_conv1x1_sb_supported.restype = ci
    L.fq_conv1x1_sb_supported.argtypes = [ci, ci]
    L.fq_conv1x1_sb_pack.restype = ci
    L.fq_conv1x1_sb_pack.argtypes = [vp, vp, ci, ci, vp]
    L.fq_conv1x1_sb_f32.restype = ci
    L.fq_conv1x1_sb_f32.argtypes = L.fq_conv1x1_f32.argtypes
    L.fq_conv1x1_sb_qd_f32.restype = ci
    L.fq_conv1x1_sb_qd_f32.argtypes = L.fq_conv1x1_qd_f32.argtypes
    L.fq_conv1x1_sb_add_f32.restype = ci
    L.fq_conv1x1_sb_add_f32.argtypes = L.fq_conv1x1_add_f32.argtypes
    L.fq_conv1x1_sb_add_hist_f32.restype = ci
    L.fq_conv1x1_sb_add_hist_f32.argtypes = L.fq_conv1x1_add_hist_f32.argtypes
    L.fq_conv_kxk_qd_f32.restype = ci
    L.fq_conv_kxk_qd_f32.argtypes = [vp, vp, vp, vp] + [ci] * 11 + [vp, sz, vp]
    L.fq_conv_stem_qd_f32.restype = ci
    L.fq_conv_stem_qd_f32.argtypes = [vp, vp, vp, vp] + [ci] * 11 + [vp]
    L.fq_maxpool2d_f32.restype = ci
    L.fq_maxpool2d_f32.argtypes = [vp, vp] + [ci] * 9 + [vp]
    L.fq_avgpool_global_f32.restype = ci
    L.fq_avgpool_global_f32.argtypes = [vp, vp, ci, ci, vp]
    L.fq_conv2d_i8_stem.restype = ci
    L.fq_conv2d_i8_stem.argtypes = [vp, vp, vp, vp] + [ci] * 16 + [vp]
    L.fq_conv2d_i8_add_resident.restype = ci
    L.fq_conv2d_i8_add_resident.argtypes = [vp, vp, vp, vp, ci, ci, vp, ci, vp, ci, ci, ci] + [ci] * 15 + [vp]
    L.fq_block_tail_i8_supported.restype = ci
    L.fq_block_tail_i8_supported.argtypes = [ci] * 9
    L.fq_block_tail_i8.restype = ci
    L.fq_block_tail_i8.argtypes = [vp, vp, vp, ci, ci, vp, ci, ci, vp, ci, vp, ci, ci, vp, vp, ci, ci, vp, ctypes.c_long, ci, ci, ci, vp]
    L.fq_block_tail_proj_i8_supported.restype = ci
    L.fq_block_tail_proj_i8_supported.argtypes = [ci] * 8
    L.fq_block_tail_proj_i8.restype = ci
    L.fq_block_tail_proj_i8.argtypes = [vp, vp, vp, ci, ci, vp, vp, vp, ci, ci, ci, ci, ci, vp, ci, vp, ci, ci, vp, vp, ci, ci, vp] + [ci] * 7 + [vp]
    L.fq_add_resident.restype = ci
    L.fq_add_resident.argtypes = [vp, ci, ci, vp, ci, ci, vp, ci, vp, ci, ci, sz, vp]
    L.fq_dequant_nhwc_to_nchw.restype = ci
    L.fq_dequant_nhwc_to_nchw.argtypes = [vp, ci, ci, vp, ci, ci, ci, ci, vp]
    L.fq_maxpool_i8_nhwc.restype = ci
    L.fq_maxpool_i8_nhwc.argtypes = [vp, vp] + [ci] * 10 + [vp]
    L.fq_avgpool_global_nhwc.restype = ci
    L.fq_avgpool_global_nhwc.argtypes = [vp, ci, ci, vp, ci, ci, ci, ci, vp]
    L.fq_json_dump_i32.restype = ci
    L.fq_json_dump_i32.argtypes = [ctypes.c_char_p, vp, ci, vp, ci]
    _lib = L
    return L


def _check(rc, what):
    if rc != 0:
        L = lib()
        raise FqError("%s failed: %s (hip error %d)" % (what, L.fq_status_string(rc).decode(), L.fq_last_hip_error()))


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream(t):
    """The HIP stream torch currently launches on for t's device (so the kernels order with torch's own
    work and are recorded by a torch.cuda.graph capture)."""
    if _raw_stream is not None:                          # ~10x cheaper than building a torch.cuda.Stream object
        idx = t.device.index
        return ctypes.c_void_p(_raw_stream(torch.cuda.current_device() if idx is None else idx))
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _need_cuda(t, dtype, what):
    if not isinstance(t, torch.Tensor) or t.device.type != "cuda":
        raise FqError("%s: expected a torch.cuda tensor (the MI355X path has no CPU fallback)" % what)
    if t.dtype != dtype:
        raise FqError("%s: expected dtype %s, got %s" % (what, dtype, t.dtype))


def dense_view(t):
    """A tensor whose storage is one dense run (any permutation of a contiguous layout, e.g.
    channels_last) can be histogrammed in storage order; anything else is made contiguous."""
    if t.is_contiguous():
        return t
    if t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last):
        return t
    return t.contiguous()


def _seg_array(tensors, rows):
    n = len(tensors)
    arr = (_Seg * max(n, 1))()
    keep = []
    for i, (t, r) in enumerate(zip(tensors, rows)):
        _need_cuda(t, torch.float32, "segment %d" % i)
        d = dense_view(t)
        keep.append(d)
        arr[i].ptr = d.data_ptr()
        arr[i].n = d.numel()
        arr[i].row = int(r)
        arr[i].reserved = 0
    return arr, keep


def _chan_array(tensors, row0s):
    """fq_chan_seg table for dense [N, C, ...] tensors (one row per channel, starting at row0)."""
    arr = (_ChanSeg * max(len(tensors), 1))()
    keep = []
    for i, (t, r0) in enumerate(zip(tensors, row0s)):
        _need_cuda(t, torch.float32, "channel segment %d" % i)
        d = t if t.is_contiguous() else t.contiguous()
        keep.append(d)
        hw = 1
        for v in d.shape[2:]:
            hw *= int(v)
        arr[i].ptr = d.data_ptr()
        arr[i].N, arr[i].C, arr[i].HW = int(d.shape[0]), int(d.shape[1]), hw
        arr[i].row0 = int(r0)
        arr[i].reserved = 0
    return arr, keep


def absmax_chan(tensors, row0s, max_inout):
    """max_inout[row0 + c] = max(., max|x[:, c]|) for every tensor.  fq_absmax_chan."""
    if not tensors:
        return
    _need_cuda(max_inout, torch.float32, "max_inout")
    arr, keep = _chan_array(tensors, row0s)
    _check(lib().fq_absmax_chan(arr, len(tensors), max_inout.data_ptr(), _stream(max_inout)), "fq_absmax_chan")
    del keep


def hist2048_chan(tensors, row0s, interval, hist):
    """hist[row0 + c] += 2048-bin histogram of |x[:, c]| (x != 0) with bin width interval[row].  fq_hist2048_chan."""
    if not tensors:
        return
    _need_cuda(interval, torch.float32, "interval")
    _need_cuda(hist, torch.int64, "hist")
    arr, keep = _chan_array(tensors, row0s)
    _check(lib().fq_hist2048_chan(arr, len(tensors), interval.data_ptr(), hist.data_ptr(), _stream(hist)), "fq_hist2048_chan")
    del keep


def absmax_seg(tensors, rows, max_inout):
    """max_inout[row] = max(max_inout[row], max|x|) for every (tensor, row).  fq_absmax_seg."""
    if not tensors:
        return
    _need_cuda(max_inout, torch.float32, "max_inout")
    arr, keep = _seg_array(tensors, rows)
    assert max(int(r) for r in rows) < max_inout.numel()
    _check(lib().fq_absmax_seg(arr, len(tensors), max_inout.data_ptr(), _stream(max_inout)), "fq_absmax_seg")
    return keep


def hist2048_seg(tensors, rows, interval, hist):
    """hist[row] += 2048-bin histogram of |x|/interval[row], x != 0.  fq_hist2048_seg."""
    if not tensors:
        return
    _need_cuda(interval, torch.float32, "interval")
    _need_cuda(hist, torch.int64, "hist")
    bins = int(hist.shape[-1])                    # INTERVAL_NUM: 2048 as shipped; 512 / 1024 / 4096 through fq_hist_seg_n
    assert hist.is_contiguous() and bins in SUPPORTED_BINS
    assert max(int(r) for r in rows) < hist.numel() // bins and interval.numel() >= hist.numel() // bins
    arr, keep = _seg_array(tensors, rows)
    if bins == BINS:
        _check(lib().fq_hist2048_seg(arr, len(tensors), interval.data_ptr(), hist.data_ptr(), _stream(hist)), "fq_hist2048_seg")
    else:
        _check(lib().fq_hist_seg_n(arr, len(tensors), interval.data_ptr(), hist.data_ptr(), bins, _stream(hist)), "fq_hist_seg_n")
    return keep


def hist2048_pair_seg(a_tensors, b_tensors, rows_a, rows_sum, interval, hist, relu_outs=None):
    """fq_hist2048_pair_seg: for every pair, a counted into hist[row_a] (row_a None / -1: not counted) and a + b (the fp32 addition
    of an Eltwise) into hist[row_sum], one pass over both; relu_outs[i] (None or a dense tensor of a's size) receives max(a + b, 0)
    as nn.ReLU computes it.  Dense fp32 CUDA tensors of equal size, 16-byte aligned."""
    if not a_tensors:
        return
    _need_cuda(interval, torch.float32, "interval")
    _need_cuda(hist, torch.int64, "hist")
    assert hist.is_contiguous() and hist.shape[-1] == BINS and len(a_tensors) == len(b_tensors) == len(rows_a) == len(rows_sum)
    relu_outs = [None] * len(a_tensors) if relu_outs is None else list(relu_outs)
    arr = (_PairSeg * len(a_tensors))()
    keep = []
    for i, (a, b, ra, rs, r) in enumerate(zip(a_tensors, b_tensors, rows_a, rows_sum, relu_outs)):
        _need_cuda(a, torch.float32, "pair %d a" % i)
        _need_cuda(b, torch.float32, "pair %d b" % i)
        assert a.shape == b.shape, "the operands of a pair have different shapes"
        da, db = dense_view(a), dense_view(b)
        if da.stride() != db.stride():                         # element i of one must be element i of the other in storage order
            da, db = a.contiguous(), b.contiguous()
        keep += [da, db]
        arr[i].a, arr[i].b, arr[i].n = da.data_ptr(), db.data_ptr(), da.numel()
        arr[i].relu_out = None
        if r is not None:
            _need_cuda(r, torch.float32, "pair %d relu_out" % i)
            assert r.shape == a.shape and r.stride() == da.stride() and r.data_ptr() not in (da.data_ptr(), db.data_ptr())
            arr[i].relu_out = r.data_ptr()
        arr[i].row_a, arr[i].row_sum = (-1 if ra is None else int(ra)), int(rs)
        assert arr[i].row_sum < hist.numel() // BINS and arr[i].row_a < hist.numel() // BINS
    _check(lib().fq_hist2048_pair_seg(arr, len(a_tensors), interval.data_ptr(), hist.data_ptr(), _stream(hist)), "fq_hist2048_pair_seg")
    return keep


def hist2048_chain_seg(chains, interval, hist):
    """fq_hist2048_chain_seg.  chains: [(head, [y_1 .. y_L], [row of y_k or None], [row of S_k])] with S_1 = y_1 + head and
    S_k = y_k + relu(S_(k-1)): every y_k counted into its row and every S_k into its row, one pass over the L + 1 tensors, nothing
    written.  Dense fp32 CUDA tensors of one size and layout, 16-byte aligned, L <= CHAIN_MAX."""
    if not chains:
        return
    _need_cuda(interval, torch.float32, "interval")
    _need_cuda(hist, torch.int64, "hist")
    assert hist.is_contiguous() and hist.shape[-1] == BINS
    rows = hist.numel() // BINS
    arr = (_ChainSeg * len(chains))()
    keep = []
    for i, (head, ys, rows_y, rows_s) in enumerate(chains):
        assert 1 <= len(ys) <= CHAIN_MAX and len(ys) == len(rows_y) == len(rows_s)
        ts = [head] + list(ys)
        for t in ts:
            _need_cuda(t, torch.float32, "chain %d" % i)
            assert t.shape == head.shape, "the tensors of a chain have different shapes"
        dense = [dense_view(t) for t in ts]
        if any(d.stride() != dense[0].stride() for d in dense):          # element i of one must be element i of the others
            dense = [t.contiguous() for t in ts]
        keep += dense
        arr[i].head, arr[i].n, arr[i].len = dense[0].data_ptr(), dense[0].numel(), len(ys)
        for k in range(len(ys)):
            arr[i].y[k] = dense[1 + k].data_ptr()
            arr[i].row_y[k] = -1 if rows_y[k] is None else int(rows_y[k])
            arr[i].row_sum[k] = int(rows_s[k])
            assert arr[i].row_sum[k] < rows and arr[i].row_y[k] < rows
    _check(lib().fq_hist2048_chain_seg(arr, len(chains), interval.data_ptr(), hist.data_ptr(), _stream(hist)), "fq_hist2048_chain_seg")
    return keep


KL_AUTO, KL_EXHAUSTIVE, KL_SCREENED = 0, 1, 2


def kl_threshold(hist, want_curve=False, mode=KL_AUTO, want_evidence=False):
    """Threshold bin per row (int32 cuda tensor); optionally the [rows,1920] float64 KL curves and/or the evidence
    (best KL, runner-up KL: float64 cuda tensors).  Returns thr, or (thr, curve), or (thr, best, runner_up), or
    (thr, curve, best, runner_up).  mode: fq.h FQ_KL_AUTO / FQ_KL_EXHAUSTIVE / FQ_KL_SCREENED."""
    _need_cuda(hist, torch.int64, "hist")
    assert hist.is_contiguous() and hist.dim() == 2 and hist.shape[1] in SUPPORTED_BINS
    rows = hist.shape[0]
    dev = hist.device
    thr = torch.empty(rows, dtype=torch.int32, device=dev)
    if hist.shape[1] != BINS:
        # INTERVAL_NUM 512 / 1024 / 4096: the exhaustive sweep only (fq_kl_threshold_n); the evidence comes from its curve
        bins = int(hist.shape[1])
        curve = torch.empty(rows, bins - 128, dtype=torch.float64, device=dev)
        if rows:
            wsb = lib().fq_kl_workspace_bytes_n(rows, bins)
            ws = torch.empty((wsb + 7) // 8, dtype=torch.float64, device=dev)
            _check(lib().fq_kl_threshold_n(hist.data_ptr(), rows, bins, thr.data_ptr(), curve.data_ptr(), ws.data_ptr(), wsb, _stream(hist)),
                   "fq_kl_threshold_n")
        out = (thr,) + ((curve,) if want_curve else ())
        if want_evidence:
            c = torch.where(torch.isnan(curve) | (curve >= 66666.0), torch.full_like(curve, float("inf")), curve)
            two = torch.topk(c, 2, dim=1, largest=False).values if rows else c[:, :2]
            out = out + (two[:, 0].contiguous(), two[:, 1].contiguous())
        return out if len(out) > 1 else thr
    curve = torch.empty(rows, KL_CANDIDATES, dtype=torch.float64, device=dev) if want_curve else None
    best = torch.empty(rows, dtype=torch.float64, device=dev) if want_evidence else None
    runner = torch.empty(rows, dtype=torch.float64, device=dev) if want_evidence else None
    if rows:
        wsb = lib().fq_kl_workspace_bytes(rows)
        ws = torch.empty((wsb + 7) // 8, dtype=torch.float64, device=dev)
        _check(lib().fq_kl_threshold_ex(hist.data_ptr(), rows, thr.data_ptr(), best.data_ptr() if want_evidence else None,
                                        runner.data_ptr() if want_evidence else None,
                                        curve.data_ptr() if want_curve else None, int(mode), ws.data_ptr(), wsb,
                                        _stream(hist)), "fq_kl_threshold_ex")
    out = (thr,) + ((curve,) if want_curve else ()) + ((best, runner) if want_evidence else ())
    return out if len(out) > 1 else thr


def _relu_ptr(relu_out, like):
    if relu_out is None:
        return None
    assert relu_out.shape == like.shape and relu_out.is_contiguous() and relu_out.dtype == torch.float32 and relu_out.is_cuda
    return relu_out.data_ptr()


def bias_add_absmax(y, bias, max_dev, row, relu_out=None):
    """fq_bias_add_absmax_f32: y[n][c][...] += bias[c] in place and max_dev[row] = max(max_dev[row], max |y|);
    relu_out (optional, same shape): also receives max(y, 0)."""
    _need_cuda(y, torch.float32, "fq_bias_add_absmax_f32")
    _need_cuda(bias, torch.float32, "fq_bias_add_absmax_f32")
    _need_cuda(max_dev, torch.float32, "fq_bias_add_absmax_f32")
    assert y.is_contiguous() and y.dim() >= 2 and bias.is_contiguous() and bias.numel() == y.shape[1]
    assert max_dev.is_contiguous() and 0 <= row < max_dev.numel()
    N, C = int(y.shape[0]), int(y.shape[1])
    hw = y.numel() // max(N * C, 1)
    _check(lib().fq_bias_add_absmax_f32(y.data_ptr(), bias.data_ptr(), N, C, hw, max_dev.data_ptr() + 4 * int(row),
                                        _relu_ptr(relu_out, y), _stream(y)), "fq_bias_add_absmax_f32")
    return y


def add_absmax(x, y, max_dev, row, out=None, relu_out=None):
    """fq_add_absmax_f32: returns x + y (in `out` if given) and folds max |x + y| into max_dev[row]; relu_out (optional):
    also receives max(x + y, 0)."""
    _need_cuda(x, torch.float32, "fq_add_absmax_f32")
    _need_cuda(y, torch.float32, "fq_add_absmax_f32")
    _need_cuda(max_dev, torch.float32, "fq_add_absmax_f32")
    assert x.shape == y.shape and x.is_contiguous() and y.is_contiguous() and 0 <= row < max_dev.numel()
    z = torch.empty_like(x) if out is None else out
    assert z.shape == x.shape and z.is_contiguous() and z.dtype == torch.float32 and z.is_cuda
    _check(lib().fq_add_absmax_f32(x.data_ptr(), y.data_ptr(), z.data_ptr(), x.numel(), max_dev.data_ptr() + 4 * int(row),
                                   _relu_ptr(relu_out, x), _stream(x)), "fq_add_absmax_f32")
    return z


def _hist_row_ptrs(interval_dev, hist_dev, row):
    _need_cuda(interval_dev, torch.float32, "interval")
    _need_cuda(hist_dev, torch.int64, "hist")
    assert interval_dev.is_contiguous() and hist_dev.is_contiguous() and hist_dev.dim() == 2 and hist_dev.shape[1] == BINS
    assert 0 <= row < hist_dev.shape[0] and interval_dev.numel() == hist_dev.shape[0]
    return interval_dev.data_ptr() + 4 * int(row), hist_dev.data_ptr() + 8 * BINS * int(row)


def bias_add_hist(y, bias, interval_dev, hist_dev, row, relu_out=None):
    """fq_bias_add_hist_f32: y[n][c][...] += bias[c] in place and every output value counted into hist_dev[row] with the
    bin width interval_dev[row]; relu_out (optional, same shape): also receives max(y, 0)."""
    _need_cuda(y, torch.float32, "fq_bias_add_hist_f32")
    _need_cuda(bias, torch.float32, "fq_bias_add_hist_f32")
    assert y.is_contiguous() and y.dim() >= 2 and bias.is_contiguous() and bias.numel() == y.shape[1]
    ivp, hp = _hist_row_ptrs(interval_dev, hist_dev, row)
    N, C = int(y.shape[0]), int(y.shape[1])
    hw = y.numel() // max(N * C, 1)
    _check(lib().fq_bias_add_hist_f32(y.data_ptr(), bias.data_ptr(), N, C, hw, ivp, hp, _relu_ptr(relu_out, y), _stream(y)),
           "fq_bias_add_hist_f32")
    return y


def add_hist(x, y, interval_dev, hist_dev, row, out=None, relu_out=None):
    """fq_add_hist_f32: returns x + y (in `out` if given) with every sum counted into hist_dev[row]; relu_out (optional):
    also receives max(x + y, 0)."""
    _need_cuda(x, torch.float32, "fq_add_hist_f32")
    _need_cuda(y, torch.float32, "fq_add_hist_f32")
    assert x.shape == y.shape and x.is_contiguous() and y.is_contiguous()
    ivp, hp = _hist_row_ptrs(interval_dev, hist_dev, row)
    z = torch.empty_like(x) if out is None else out
    assert z.shape == x.shape and z.is_contiguous() and z.dtype == torch.float32 and z.is_cuda
    _check(lib().fq_add_hist_f32(x.data_ptr(), y.data_ptr(), z.data_ptr(), x.numel(), ivp, hp, _relu_ptr(relu_out, x),
                                 _stream(x)), "fq_add_hist_f32")
    return z


def _qd_args(qd, max_dev, hist_dev, relu_out, name):
    """(bit, bitwidth) of a fused QuanDequan epilogue; it excludes the statistic epilogues and the ReLU copy."""
    bit, bitwidth = (int(qd[0]), int(qd[1])) if isinstance(qd, (tuple, list)) else (int(qd), 8)
    if max_dev is not None or hist_dev is not None or relu_out is not None:
        raise FqError(name + ": the QuanDequan epilogue takes no statistic and no ReLU copy")
    return bit, bitwidth


# The tail split of the float convolutions (include/fq.h, fq_conv_f32_workspace_bytes): the CALLER owns the 16 MB scratch.  Here
# that is one zero-filled tensor from torch's allocator per (device, stream) -- launches of one stream run one after the
# other, launches of different streams may overlap and must not share counters.  conv_tail_split = False (or
# FQ_CONV_TAIL_SPLIT=0) hands the kernels no workspace at all: every output is then ONE fma chain whatever the batch size is
# (with the split, which tiles are cut depends on the launch's tile count, i.e. on N).
conv_tail_split = os.environ.get("FQ_CONV_TAIL_SPLIT", "1") != "0"
_conv_ws = {}


def conv_workspace(x):
    """(pointer, bytes) of this stream's tail-split workspace, or (None, 0): split switched off, or the first use of a stream
    falls inside a graph capture (an allocation made there would belong to the graph's private pool)."""
    if not conv_tail_split:
        return None, 0
    key = (x.device.index, torch.cuda.current_stream(x.device).cuda_stream)
    ws = _conv_ws.get(key)
    if ws is None:
        if torch.cuda.is_current_stream_capturing():
            return None, 0
        ws = torch.zeros(int(lib().fq_conv_f32_workspace_bytes()), dtype=torch.uint8, device=x.device)
        _conv_ws[key] = ws
    return ws.data_ptr(), ws.numel()


def conv_sb_enabled():
    """FQ_CONV_SPLIT_BF16=1 runs the float 1x1 convolutions on the split-bf16 kernels (fq_conv1x1_sb_f32 and its add / QuanDequan
    forms) instead of the fp32 MFMA ones.  Off by default: as accurate, not faster yet."""
    return os.environ.get("FQ_CONV_SPLIT_BF16", "0") == "1"


def conv_sb_supported(cin, cout):
    return bool(lib().fq_conv1x1_sb_supported(int(cin), int(cout)))


def pack_sb_weight(weight):
    """[Cout, Cin(, 1, 1)] fp32 -> the pack the split-bf16 1x1 kernels read (fq_conv1x1_sb_pack): 3 * Cout * Cin bf16 bit patterns,
    the hi / mid / lo piece of every weight in the kernel's fragment order; handed around as an int16 tensor of SHAPE [3, Cout, Cin]
    (the shape tells conv1x1_f32 / conv1x1_add_f32 / conv1x1_add_hist_f32 what it is; the memory order is the pack's)."""
    _need_cuda(weight, torch.float32, "fq_conv1x1_sb_pack")
    cout, cin = int(weight.shape[0]), int(weight.shape[1])
    w = weight.detach().reshape(cout, cin).contiguous()
    out = torch.empty((3, cout, cin), dtype=torch.int16, device=w.device)
    assert out.numel() * 2 == lib().fq_conv1x1_sb_packed_bytes(cin, cout)
    _check(lib().fq_conv1x1_sb_pack(w.data_ptr(), out.data_ptr(), cin, cout, _stream(w)), "fq_conv1x1_sb_pack")
    return out


def _sb(wt):
    """(is the weight operand a split-bf16 pack?, Cin, Cout)"""
    if wt.dtype == torch.int16:
        assert wt.is_cuda and wt.dim() == 3 and wt.shape[0] == 3 and wt.is_contiguous()
        return True, int(wt.shape[2]), int(wt.shape[1])
    _need_cuda(wt, torch.float32, "fq_conv1x1_f32")
    assert wt.dim() == 2 and wt.is_contiguous()
    return False, int(wt.shape[0]), int(wt.shape[1])


def conv1x1_f32(x, wt, bias, stride=1, max_dev=None, interval_dev=None, hist_dev=None, row=None, relu_out=None, out=None, qd=None):
    """fq_conv1x1_f32: the float 1x1 convolution (padding 0, groups 1) of x [N, Cin, H, W] with the TRANSPOSED weights
    wt [Cin, Cout] on the fp32 matrix cores; max_dev/row: abs-max of the output folded into max_dev[row]; interval_dev/
    hist_dev/row: the output histogrammed into hist_dev[row]; relu_out: also receives max(y, 0).  qd = bit or (bit, bitwidth):
    fq_conv1x1_qd_f32 instead -- QuanDequan(bit) applied in the epilogue (TestConv.forward in one kernel).  Returns y."""
    _need_cuda(x, torch.float32, "fq_conv1x1_f32")
    sb, wcin, Cout = _sb(wt)                                    # (an int16 [3, Cout, Cin] pack: the split-bf16 kernels)
    assert x.dim() == 4 and x.is_contiguous() and wcin == x.shape[1]
    N, Cin, H, W = (int(v) for v in x.shape)
    s = int(stride)
    fn, fn_qd = (lib().fq_conv1x1_sb_f32, lib().fq_conv1x1_sb_qd_f32) if sb else (lib().fq_conv1x1_f32, lib().fq_conv1x1_qd_f32)
    shape = (N, Cout, (H - 1) // s + 1, (W - 1) // s + 1)
    if out is False:                                            # only the ReLU's output is wanted: y is not written
        assert relu_out is not None and qd is None and tuple(relu_out.shape) == shape
        y = None
    else:
        y = torch.empty(shape, dtype=torch.float32, device=x.device) if out is None else out
        assert tuple(y.shape) == shape and y.is_contiguous() and y.dtype == torch.float32 and y.is_cuda
    if bias is not None:
        _need_cuda(bias, torch.float32, "fq_conv1x1_f32")
        assert bias.is_contiguous() and bias.numel() == Cout
    if qd is not None:
        bit, bw = _qd_args(qd, max_dev, hist_dev, relu_out, "fq_conv1x1_qd_f32")
        _check(fn_qd(x.data_ptr(), wt.data_ptr(), None if bias is None else bias.data_ptr(), y.data_ptr(),
                     N, Cin, H, W, Cout, s, bit, bw, *conv_workspace(x), _stream(x)), "fq_conv1x1_qd_f32")
        return y
    mp = ivp = hp = None
    if hist_dev is not None:
        ivp, hp = _hist_row_ptrs(interval_dev, hist_dev, row)
    elif max_dev is not None:
        _need_cuda(max_dev, torch.float32, "fq_conv1x1_f32")
        assert max_dev.is_contiguous() and 0 <= row < max_dev.numel()
        mp = max_dev.data_ptr() + 4 * int(row)
    _check(fn(x.data_ptr(), wt.data_ptr(), None if bias is None else bias.data_ptr(),
              None if y is None else y.data_ptr(), _relu_ptr(relu_out, relu_out if y is None else y), N, Cin, H, W,
              Cout, s, mp, ivp, hp, *conv_workspace(x), _stream(x)), "fq_conv1x1_f32")
    return y


def conv1x1_add_f32(x, wt, bias, stride, res, max_dev, row_y, row_sum, relu_out, out=None, sum_out=None):
    """fq_conv1x1_add_f32: the 1x1 convolution of conv1x1_f32, the Eltwise that consumes it and the ReLU behind that in one
    kernel (pass 1).  v = conv(x) + bias: abs-max folded into max_dev[row_y], written to `out` when given; s = v + res: abs-max
    into max_dev[row_sum], written to `sum_out` when given; relu_out receives max(s, 0).  Returns relu_out."""
    for t in (x, bias, res, max_dev, relu_out):
        _need_cuda(t, torch.float32, "fq_conv1x1_add_f32")
    sb, wcin, Cout = _sb(wt)
    assert x.dim() == 4 and x.is_contiguous() and wcin == x.shape[1]
    N, Cin, H, W = (int(v) for v in x.shape)
    s = int(stride)
    shape = (N, Cout, (H - 1) // s + 1, (W - 1) // s + 1)
    assert bias.is_contiguous() and bias.numel() == Cout and max_dev.is_contiguous()
    assert 0 <= row_y < max_dev.numel() and 0 <= row_sum < max_dev.numel() and row_y != row_sum
    for t in (res, relu_out, out, sum_out):
        assert t is None or (tuple(t.shape) == shape and t.is_contiguous() and t.dtype == torch.float32 and t.is_cuda)
    _check((lib().fq_conv1x1_sb_add_f32 if sb else lib().fq_conv1x1_add_f32)(x.data_ptr(), wt.data_ptr(), bias.data_ptr(), res.data_ptr(),
                                    None if out is None else out.data_ptr(), None if sum_out is None else sum_out.data_ptr(),
                                    relu_out.data_ptr(), N, Cin, H, W, Cout, s, max_dev.data_ptr() + 4 * int(row_y),
                                    max_dev.data_ptr() + 4 * int(row_sum), *conv_workspace(x), _stream(x)), "fq_conv1x1_add_f32")
    return relu_out


def conv1x1_add_hist_f32(x, wt, bias, stride, res, interval_dev, hist_dev, row_y, row_sum, relu_out):
    """fq_conv1x1_add_hist_f32: the chain of conv1x1_add_f32 in pass 2 -- v = conv(x) + bias counted into hist_dev[row_y], s = v + res
    into hist_dev[row_sum] (bin widths interval_dev[row]); neither is written; relu_out receives max(s, 0).  Returns relu_out."""
    for t in (x, bias, res, relu_out):
        _need_cuda(t, torch.float32, "fq_conv1x1_add_hist_f32")
    sb, wcin, Cout = _sb(wt)
    assert x.dim() == 4 and x.is_contiguous() and wcin == x.shape[1]
    N, Cin, H, W = (int(v) for v in x.shape)
    s = int(stride)
    shape = (N, Cout, (H - 1) // s + 1, (W - 1) // s + 1)
    assert bias.is_contiguous() and bias.numel() == Cout and row_y != row_sum
    for t in (res, relu_out):
        assert tuple(t.shape) == shape and t.is_contiguous()
    ivy, hy = _hist_row_ptrs(interval_dev, hist_dev, row_y)
    ivs, hs = _hist_row_ptrs(interval_dev, hist_dev, row_sum)
    _check((lib().fq_conv1x1_sb_add_hist_f32 if sb else lib().fq_conv1x1_add_hist_f32)(x.data_ptr(), wt.data_ptr(), bias.data_ptr(), res.data_ptr(), relu_out.data_ptr(),
                                         N, Cin, H, W, Cout, s, ivy, hy, ivs, hs, *conv_workspace(x), _stream(x)),
           "fq_conv1x1_add_hist_f32")
    return relu_out


def conv1x1_add_f32_supported(cin, cout):
    return cin % 16 == 0 and cout % 128 == 0


def pack_kxk_weight(weight):
    """[Cout, Cin, R, S] -> Wt [(r*S + s)*Cin + ci][Cout], the layout fq_conv_kxk_f32 reads."""
    cout, cin, r, s = (int(v) for v in weight.shape)
    return weight.detach().permute(2, 3, 1, 0).reshape(r * s * cin, cout).contiguous()


def conv_kxk_f32(x, wt, bias, kernel, stride, pad, max_dev=None, interval_dev=None, hist_dev=None, row=None, relu_out=None,
                 out=None, qd=None):
    """fq_conv_kxk_f32: the float R x S convolution (zero padding `pad`, dilation 1, groups 1) of x [N, Cin, H, W] with the
    weights packed by pack_kxk_weight; statistics / relu_out / out as in conv1x1_f32.  Returns y."""
    _need_cuda(x, torch.float32, "fq_conv_kxk_f32")
    _need_cuda(wt, torch.float32, "fq_conv_kxk_f32")
    R, S = int(kernel[0]), int(kernel[1])
    assert x.dim() == 4 and x.is_contiguous() and wt.dim() == 2 and wt.is_contiguous() and wt.shape[0] == R * S * x.shape[1]
    N, Cin, H, W = (int(v) for v in x.shape)
    Cout, st, pd = int(wt.shape[1]), int(stride), int(pad)
    shape = (N, Cout, (H + 2 * pd - R) // st + 1, (W + 2 * pd - S) // st + 1)
    if out is False:                                            # only the ReLU's output is wanted: y is not written
        assert relu_out is not None and qd is None and tuple(relu_out.shape) == shape
        y = None
    else:
        y = torch.empty(shape, dtype=torch.float32, device=x.device) if out is None else out
        assert tuple(y.shape) == shape and y.is_contiguous() and y.dtype == torch.float32 and y.is_cuda
    if bias is not None:
        _need_cuda(bias, torch.float32, "fq_conv_kxk_f32")
        assert bias.is_contiguous() and bias.numel() == Cout
    if qd is not None:
        bit, bw = _qd_args(qd, max_dev, hist_dev, relu_out, "fq_conv_kxk_qd_f32")
        _check(lib().fq_conv_kxk_qd_f32(x.data_ptr(), wt.data_ptr(), None if bias is None else bias.data_ptr(), y.data_ptr(),
                                        N, Cin, H, W, Cout, R, S, st, pd, bit, bw, *conv_workspace(x), _stream(x)), "fq_conv_kxk_qd_f32")
        return y
    mp = ivp = hp = None
    if hist_dev is not None:
        ivp, hp = _hist_row_ptrs(interval_dev, hist_dev, row)
    elif max_dev is not None:
        _need_cuda(max_dev, torch.float32, "fq_conv_kxk_f32")
        assert max_dev.is_contiguous() and 0 <= row < max_dev.numel()
        mp = max_dev.data_ptr() + 4 * int(row)
    _check(lib().fq_conv_kxk_f32(x.data_ptr(), wt.data_ptr(), None if bias is None else bias.data_ptr(),
                                 None if y is None else y.data_ptr(), _relu_ptr(relu_out, relu_out if y is None else y), N, Cin, H, W,
                                 Cout, R, S, st, pd, mp, ivp, hp, *conv_workspace(x), _stream(x)), "fq_conv_kxk_f32")
    return y


def conv_wino_enabled():
    """FQ_CONV_WINO=0 keeps the stride-1 3x3 layers of the float forward on the direct kernel (fq_conv_kxk_f32)."""
    return os.environ.get("FQ_CONV_WINO", "1") != "0"


def conv_wino_supported(n, cin, h, w, cout):
    """True when fq_conv3x3_wino_f32 takes a stride-1, pad-1 3x3 convolution of x [n, cin, h, w] to cout channels."""
    return bool(lib().fq_conv3x3_wino_f32_supported(int(n), int(cin), int(h), int(w), int(cout)))


def pack_wino_weight(weight):
    """[Cout, Cin, 3, 3] (device, fp32) -> the transformed, packed weights fq_conv3x3_wino_f32 reads."""
    _need_cuda(weight, torch.float32, "fq_conv3x3_wino_f32_pack")
    cout, cin, r, s = (int(v) for v in weight.shape)
    assert r == 3 and s == 3
    w = weight.detach().contiguous()
    u = torch.empty(int(lib().fq_conv3x3_wino_f32_packed_floats(cin, cout)), dtype=torch.float32, device=w.device)
    _check(lib().fq_conv3x3_wino_f32_pack(w.data_ptr(), u.data_ptr(), cin, cout, _stream(w)), "fq_conv3x3_wino_f32_pack")
    return u


def conv_wino_f32(x, u, bias, cout, max_dev=None, interval_dev=None, hist_dev=None, row=None, relu_out=None, out=None, qd=None):
    """fq_conv3x3_wino_f32: the stride-1, pad-1 3x3 float convolution of x [N, Cin, H, W] with the weights packed by
    pack_wino_weight; statistics / relu_out / out as in conv1x1_f32.  Returns y."""
    _need_cuda(x, torch.float32, "fq_conv3x3_wino_f32")
    _need_cuda(u, torch.float32, "fq_conv3x3_wino_f32")
    assert x.dim() == 4 and x.is_contiguous() and u.is_contiguous()
    N, Cin, H, W = (int(v) for v in x.shape)
    Cout = int(cout)
    assert u.numel() == 16 * Cin * Cout
    shape = (N, Cout, H, W)
    if out is False:                                            # only the ReLU's output is wanted: y is not written
        assert relu_out is not None and tuple(relu_out.shape) == shape
        y = None
    else:
        y = torch.empty(shape, dtype=torch.float32, device=x.device) if out is None else out
        assert tuple(y.shape) == shape and y.is_contiguous() and y.dtype == torch.float32 and y.is_cuda
    if bias is not None:
        _need_cuda(bias, torch.float32, "fq_conv3x3_wino_f32")
        assert bias.is_contiguous() and bias.numel() == Cout
    if qd is not None:
        bit, bw = _qd_args(qd, max_dev, hist_dev, relu_out, "fq_conv3x3_wino_qd_f32")
        _check(lib().fq_conv3x3_wino_qd_f32(x.data_ptr(), u.data_ptr(), None if bias is None else bias.data_ptr(), y.data_ptr(),
                                            N, Cin, H, W, Cout, bit, bw, _stream(x)), "fq_conv3x3_wino_qd_f32")
        return y
    mp = ivp = hp = None
    if hist_dev is not None:
        ivp, hp = _hist_row_ptrs(interval_dev, hist_dev, row)
    elif max_dev is not None:
        _need_cuda(max_dev, torch.float32, "fq_conv3x3_wino_f32")
        assert max_dev.is_contiguous() and 0 <= row < max_dev.numel()
        mp = max_dev.data_ptr() + 4 * int(row)
    _check(lib().fq_conv3x3_wino_f32(x.data_ptr(), u.data_ptr(), None if bias is None else bias.data_ptr(),
                                     None if y is None else y.data_ptr(), _relu_ptr(relu_out, relu_out if y is None else y), N, Cin,
                                     H, W, Cout, mp, ivp, hp, _stream(x)), "fq_conv3x3_wino_f32")
    return y


def conv_stem_f32_supported(weight, stride):
    """True when fq_conv_stem_f32 takes a convolution with this weight [Cout, Cin, R, S] and stride."""
    cout, cin, r, s = (int(v) for v in weight.shape)
    return stride == 2 and cout <= 64 and lib().fq_conv_stem_f32_packed_rows(cin, r, s) > 0


def pack_stem_weight(weight):
    """[Cout, Cin, R, S] -> the [Cin*R*8, 64] layout fq_conv_stem_f32 reads (taps padded to 8, channels to 64, zeros)."""
    cout, cin, r, s = (int(v) for v in weight.shape)
    rows = lib().fq_conv_stem_f32_packed_rows(cin, r, s)
    assert rows > 0 and cout <= 64, "fq_conv_stem_f32: unsupported stem shape"
    wp = torch.zeros((cin, r, rows // (cin * r), 64), dtype=torch.float32, device=weight.device)
    wp[:, :, :s, :cout] = weight.detach().permute(1, 2, 3, 0)
    return wp.view(rows, 64)


def conv_stem_f32(x, wp, bias, cout, kernel, stride, pad, max_dev=None, interval_dev=None, hist_dev=None, row=None,
                  relu_out=None, out=None, qd=None):
    """fq_conv_stem_f32: the float stem convolution (kernel = (R, S), e.g. 7x7 stride 2) of x [N, Cin, H, W] with the packed
    weights wp (pack_stem_weight); statistics / relu_out / out as in conv1x1_f32.  Returns y."""
    _need_cuda(x, torch.float32, "fq_conv_stem_f32")
    _need_cuda(wp, torch.float32, "fq_conv_stem_f32")
    assert x.dim() == 4 and x.is_contiguous() and wp.is_contiguous()
    N, Cin, H, W = (int(v) for v in x.shape)
    R, S = int(kernel[0]), int(kernel[1])
    shape = (N, int(cout), (H + 2 * pad - R) // stride + 1, (W + 2 * pad - S) // stride + 1)
    if out is False:                                            # only the ReLU's output is wanted: y is not written
        assert relu_out is not None and qd is None and tuple(relu_out.shape) == shape
        y = None
    else:
        y = torch.empty(shape, dtype=torch.float32, device=x.device) if out is None else out
        assert tuple(y.shape) == shape and y.is_contiguous() and y.dtype == torch.float32 and y.is_cuda
    if bias is not None:
        _need_cuda(bias, torch.float32, "fq_conv_stem_f32")
        assert bias.is_contiguous() and bias.numel() == cout
    if qd is not None:
        bit, bw = _qd_args(qd, max_dev, hist_dev, relu_out, "fq_conv_stem_qd_f32")
        _check(lib().fq_conv_stem_qd_f32(x.data_ptr(), wp.data_ptr(), None if bias is None else bias.data_ptr(), y.data_ptr(),
                                         N, Cin, H, W, int(cout), R, S, int(stride), int(pad), bit, bw, _stream(x)),
               "fq_conv_stem_qd_f32")
        return y
    mp = ivp = hp = None
    if hist_dev is not None:
        ivp, hp = _hist_row_ptrs(interval_dev, hist_dev, row)
    elif max_dev is not None:
        _need_cuda(max_dev, torch.float32, "fq_conv_stem_f32")
        assert max_dev.is_contiguous() and 0 <= row < max_dev.numel()
        mp = max_dev.data_ptr() + 4 * int(row)
    _check(lib().fq_conv_stem_f32(x.data_ptr(), wp.data_ptr(), None if bias is None else bias.data_ptr(),
                                  None if y is None else y.data_ptr(), _relu_ptr(relu_out, relu_out if y is None else y), N, Cin, H, W,
                                  int(cout), R, S, int(stride), int(pad), mp, ivp, hp, _stream(x)), "fq_conv_stem_f32")
    return y


def maxpool2d_f32(x, kernel, stride, pad):
    """fq_maxpool2d_f32: torch.nn.functional.max_pool2d(x, kernel, stride, pad) (floor mode, no dilation), bit for bit."""
    _need_cuda(x, torch.float32, "fq_maxpool2d_f32")
    assert x.dim() == 4 and x.is_contiguous()
    N, C, H, W = (int(v) for v in x.shape)
    (kh, kw), (sh, sw), (ph, pw) = kernel, stride, pad
    y = torch.empty((N, C, (H + 2 * ph - kh) // sh + 1, (W + 2 * pw - kw) // sw + 1), dtype=torch.float32, device=x.device)
    _check(lib().fq_maxpool2d_f32(x.data_ptr(), y.data_ptr(), N * C, H, W, kh, kw, sh, sw, ph, pw, _stream(x)), "fq_maxpool2d_f32")
    return y


def avgpool_global_f32(x):
    """fq_avgpool_global_f32: the average of every [H, W] plane of x [N, C, H, W] -> [N, C, 1, 1], torch's summation order."""
    _need_cuda(x, torch.float32, "fq_avgpool_global_f32")
    assert x.dim() == 4 and x.is_contiguous()
    N, C, H, W = (int(v) for v in x.shape)
    y = torch.empty((N, C, 1, 1), dtype=torch.float32, device=x.device)
    _check(lib().fq_avgpool_global_f32(x.data_ptr(), y.data_ptr(), N * C, H * W, _stream(x)), "fq_avgpool_global_f32")
    return y


def bits_from_threshold(thr, interval):
    """Host helper: (bits int32[], threshold_value float32[]) from threshold bins and fp32 intervals."""
    thr = np.ascontiguousarray(thr, dtype=np.int32)
    interval = np.ascontiguousarray(interval, dtype=np.float32)
    bits = np.empty(thr.shape, dtype=np.int32)
    tv = np.empty(thr.shape, dtype=np.float32)
    _check(lib().fq_bits_from_threshold(thr.ctypes.data, interval.ctypes.data, thr.size, bits.ctypes.data,
                                        tv.ctypes.data), "fq_bits_from_threshold")
    return bits, tv


def bits_from_absmax(absmax):
    absmax = np.ascontiguousarray(absmax, dtype=np.float32)
    bits = np.empty(absmax.shape, dtype=np.int32)
    _check(lib().fq_bits_from_absmax(absmax.ctypes.data, absmax.size, bits.ctypes.data), "fq_bits_from_absmax")
    return bits


def _out_like(x, out):
    if out is None:
        return torch.empty_like(x, memory_format=torch.contiguous_format)
    _need_cuda(out, torch.float32, "out")
    assert out.is_contiguous() and out.numel() == x.numel()
    return out


def _unary(fn_name, x, ints, out):
    _need_cuda(x, torch.float32, fn_name)
    xc = x if x.is_contiguous() else x.contiguous()
    y = _out_like(xc, out)
    _check(getattr(lib(), fn_name)(xc.data_ptr(), y.data_ptr(), xc.numel(), *ints, _stream(xc)), fn_name)
    return y.view(x.shape) if y.shape != x.shape else y


def quandequan(x, bit, bitwidth=8, out=None):
    return _unary("fq_quandequan_f32", x, (int(bit), int(bitwidth)), out)


def quantity(x, ib, bitwidth=8, out=None):
    return _unary("fq_quantity_f32", x, (int(ib), int(bitwidth)), out)


def dequantity(x, ob, out=None):
    return _unary("fq_dequantity_f32", x, (int(ob),), out)


def sp(x, bitwidth=8, out=None):
    return _unary("fq_sp_f32", x, (int(bitwidth),), out)


def rightshift(x, rs, bitwidth=8, out=None):
    return _unary("fq_rightshift_f32", x, (int(rs), int(bitwidth)), out)


def add_sat(a, b, bitwidth=8, out=None):
    _need_cuda(a, torch.float32, "fq_add_sat_f32")
    _need_cuda(b, torch.float32, "fq_add_sat_f32")
    if a.shape != b.shape:
        a, b = torch.broadcast_tensors(a, b)
    ac = a.contiguous()
    bc = b.contiguous()
    y = _out_like(ac, out)
    _check(lib().fq_add_sat_f32(ac.data_ptr(), bc.data_ptr(), y.data_ptr(), ac.numel(), int(bitwidth), _stream(ac)),
           "fq_add_sat_f32")
    return y


def recon_epilogue(acc, qbias, rs, ob, bitwidth=8, out=None):
    """acc: [N, C, ...] fp32 (conv/linear accumulator), qbias: fp32[C] integer valued."""
    _need_cuda(acc, torch.float32, "fq_recon_epilogue_f32")
    _need_cuda(qbias, torch.float32, "fq_recon_epilogue_f32")
    ac = acc.contiguous()
    C = ac.shape[1]
    assert qbias.numel() == C
    outer = ac.shape[0]
    inner = ac.numel() // (outer * C) if ac.numel() else 0
    y = _out_like(ac, out)
    _check(lib().fq_recon_epilogue_f32(ac.data_ptr(), qbias.contiguous().data_ptr(), y.data_ptr(), outer, C, inner,
                                       int(rs), int(ob), int(bitwidth), _stream(ac)), "fq_recon_epilogue_f32")
    return y


def quantize_param_i32(w, bit):
    _need_cuda(w, torch.float32, "fq_quantize_param_i32")
    wc = w.contiguous()
    q = torch.empty(wc.shape, dtype=torch.int32, device=w.device)
    _check(lib().fq_quantize_param_i32(wc.data_ptr(), q.data_ptr(), wc.numel(), int(bit), _stream(wc)),
           "fq_quantize_param_i32")
    return q


def json_dump_i32(array, path, indent=4):
    """Write a host int array as nested JSON lists, byte-identical to json.dump(a.tolist(), indent=4)."""
    a = np.asarray(array)
    nd = a.ndim
    a = np.ascontiguousarray(a, dtype=np.int32).reshape(a.shape)     # ascontiguousarray promotes 0-d to 1-d
    assert a.ndim == nd
    shape = np.array(a.shape, dtype=np.int64)
    _check(lib().fq_json_dump_i32(os.fsencode(path), a.ctypes.data, a.ndim, shape.ctypes.data if a.ndim else None,
                                  int(indent)), "fq_json_dump_i32(%s)" % path)


def pad16(c):
    return (int(c) + 15) // 16 * 16


def quantize_i8_nhwc(x, ib, cpad=None):
    """fp32 [N, C, *spatial] -> int8 [N, *spatial, Cpad] holding clamp(rint(x * 2^ib)); Cpad = C rounded
    up to 16 (zero filled) unless given."""
    _need_cuda(x, torch.float32, "fq_quantize_i8_nhwc")
    xc = x.contiguous()
    N, C = xc.shape[0], xc.shape[1]
    spatial = tuple(xc.shape[2:])
    HW = 1
    for s in spatial:
        HW *= s
    cpad = pad16(C) if cpad is None else int(cpad)
    y = torch.empty((N,) + spatial + (cpad,), dtype=torch.int8, device=x.device)
    _check(lib().fq_quantize_i8_nhwc(xc.data_ptr(), y.data_ptr(), N, C, HW, cpad, int(ib), _stream(xc)),
           "fq_quantize_i8_nhwc")
    return y


def pack_weight_krsc(w, cpad=None):
    """Integer-valued fp32 weights [K, C, R, S] (or [K, F]) -> int8 [K, R, S, Cpad] for fq_conv2d_i8."""
    if w.dim() == 2:
        w = w[:, :, None, None]
    K, C, R, S = w.shape
    cpad = pad16(C) if cpad is None else int(cpad)
    out = torch.zeros(K, R, S, cpad, dtype=torch.int8, device=w.device)
    out[..., :C] = w.permute(0, 2, 3, 1).to(torch.int8)
    return out.contiguous()


# Which integer-convolution kernels ran (fq_conv2d_i8_last_variant): set conv_variant_log = {} and every call below counts its
# kernel there by name -- tests and bench.py assert with it that the dispatch they checked is the dispatch they time.
CONV_VARIANTS = {0: "none", 1: "c64_halo", 2: "stream", 3: "halo8", 4: "halo", 5: "dma2", 6: "dma3", 7: "tile_c128", 8: "tile_c64",
                 9: "tile_general", 10: "stem", 11: "block_tail", 12: "block_tail_proj", 13: "linear_wave"}
conv_variant_log = None


def _note_variant():
    log = conv_variant_log
    if log is not None:
        v = int(lib().fq_conv2d_i8_last_variant())
        name = CONV_VARIANTS.get(v & 0xff, "?") + ("" if (v >> 8) == 0 else "/%d" % (v >> 8))
        log[name] = log.get(name, 0) + 1


def conv2d_i8(xq, wq, qbias, stride, padding, dilation, rs, ob, bitwidth=8):
    """xq int8 [N,H,W,C] (or [N,C] for Linear), wq int8 [K,R,S,C]; returns fp32 [N,K,P,Q] (or [N,K])."""
    _need_cuda(xq, torch.int8, "fq_conv2d_i8")
    _need_cuda(wq, torch.int8, "fq_conv2d_i8")
    _need_cuda(qbias, torch.float32, "fq_conv2d_i8")
    linear = xq.dim() == 2
    if linear:
        xq = xq[:, None, None, :]
    N, H, W, C = xq.shape
    K, R, S, Cw = wq.shape
    assert C == Cw and xq.is_contiguous() and wq.is_contiguous() and qbias.numel() == K
    P = (H + 2 * padding[0] - dilation[0] * (R - 1) - 1) // stride[0] + 1
    Q = (W + 2 * padding[1] - dilation[1] * (S - 1) - 1) // stride[1] + 1
    y = torch.empty(N, K, P, Q, dtype=torch.float32, device=xq.device)
    _check(lib().fq_conv2d_i8(xq.data_ptr(), wq.data_ptr(), qbias.contiguous().data_ptr(), y.data_ptr(), N, H, W, C, K, R, S,
                              stride[0], stride[1], padding[0], padding[1], dilation[0], dilation[1], int(rs), int(ob),
                              int(bitwidth), _stream(xq)), "fq_conv2d_i8")
    _note_variant()
    return y.view(N, K) if linear else y


def conv2d_i8_resident(xq, wq, qbias, stride, padding, dilation, rs, ob, want_f32, want_i8, relu):
    """fq_conv2d_i8_resident: returns (y fp32 [N,K,P,Q] or None, q int8 [N,P,Q,Kpad] or None).  q holds the
    integers before DeQuantity (value = q * 2^-ob), channels zero-padded to a multiple of 16."""
    _need_cuda(xq, torch.int8, "fq_conv2d_i8_resident")
    _need_cuda(wq, torch.int8, "fq_conv2d_i8_resident")
    _need_cuda(qbias, torch.float32, "fq_conv2d_i8_resident")
    N, H, W, C = xq.shape
    K, R, S, Cw = wq.shape
    assert C == Cw and xq.is_contiguous() and wq.is_contiguous() and qbias.numel() == K and (want_f32 or want_i8)
    P = (H + 2 * padding[0] - dilation[0] * (R - 1) - 1) // stride[0] + 1
    Q = (W + 2 * padding[1] - dilation[1] * (S - 1) - 1) // stride[1] + 1
    kpad = pad16(K)
    y = torch.empty(N, K, P, Q, dtype=torch.float32, device=xq.device) if want_f32 else None
    q = torch.empty(N, P, Q, kpad, dtype=torch.int8, device=xq.device) if want_i8 else None
    _check(lib().fq_conv2d_i8_resident(xq.data_ptr(), wq.data_ptr(), qbias.contiguous().data_ptr(),
                                       y.data_ptr() if want_f32 else None, q.data_ptr() if want_i8 else None, kpad,
                                       1 if relu else 0, N, H, W, C, K, R, S, stride[0], stride[1], padding[0], padding[1],
                                       dilation[0], dilation[1], int(rs), int(ob), _stream(xq)), "fq_conv2d_i8_resident")
    _note_variant()
    return y, q


STEM_MAX_K, STEM_MAX_R, STEM_MAX_S, STEM_MAX_C = 64, 8, 8, 4


def stem_supported(C, K, R, S, stride, dilation, rs):
    """True when fq_conv2d_i8_stem takes this layer (include/fq.h); otherwise the unfold path computes the same integers."""
    if C > STEM_MAX_C or K > STEM_MAX_K or R > STEM_MAX_R or S > STEM_MAX_S or tuple(dilation) != (1, 1) or not 1 <= rs <= 16:
        return False
    pr, pc = 7 * stride[0] + R, 15 * stride[1] + S
    pcs = max(15 * stride[1] + 8, pc) | 1
    return pr * pc <= 1024 and (pr - 1) * pcs + 15 * stride[1] + 8 <= 1536


def pack_weight_stem(w):
    """Integer-valued fp32 weights [K, C, R, S] (K <= 64, C <= 4, S <= 8) -> int8 [R, 64, 32]: byte 4*s + c."""
    K, C, R, S = w.shape
    out = torch.zeros(R, STEM_MAX_K, 8, 4, dtype=torch.int8, device=w.device)
    out[:, :K, :S, :C] = w.permute(2, 0, 3, 1).to(torch.int8)            # [R, K, S, C]
    return out.view(R, STEM_MAX_K, 32).contiguous()


def conv2d_i8_stem(x, w_stem, qbias, K, S, stride, padding, ib, rs, ob, relu):
    """fq_conv2d_i8_stem: fp32 NCHW image -> int8 [N,P,Q,Kpad] (the integers before DeQuantity(ob), ReLU folded in)."""
    _need_cuda(x, torch.float32, "fq_conv2d_i8_stem")
    _need_cuda(w_stem, torch.int8, "fq_conv2d_i8_stem")
    _need_cuda(qbias, torch.float32, "fq_conv2d_i8_stem")
    N, C, H, W = x.shape
    R = w_stem.shape[0]
    assert x.is_contiguous() and w_stem.is_contiguous() and tuple(w_stem.shape[1:]) == (STEM_MAX_K, 32) and qbias.numel() == K
    P = (H + 2 * padding[0] - R) // stride[0] + 1
    Q = (W + 2 * padding[1] - S) // stride[1] + 1
    kpad = pad16(K)
    q = torch.empty(N, P, Q, kpad, dtype=torch.int8, device=x.device)
    _check(lib().fq_conv2d_i8_stem(x.data_ptr(), w_stem.data_ptr(), qbias.contiguous().data_ptr(), q.data_ptr(), kpad,
                                   1 if relu else 0, N, C, H, W, K, R, S, stride[0], stride[1], padding[0], padding[1],
                                   int(ib), int(rs), int(ob), _stream(x)), "fq_conv2d_i8_stem")
    _note_variant()
    return q


_INT_BYTES = {torch.int8: 1, torch.int16: 2}


def conv2d_i8_add_resident(xq, wq, qbias, stride, padding, dilation, rs, ob, res, g_res, want_wide, g_wide, want_narrow, ib, relu):
    """fq_conv2d_i8_add_resident: NewAdd(conv(xq), res) without writing the conv result.  res: int8 / int16
    [N,P,Q,Kpad] standing for res * 2^-g_res.  Returns (wide int16 or None, narrow int8 or None)."""
    _need_cuda(xq, torch.int8, "fq_conv2d_i8_add_resident")
    _need_cuda(wq, torch.int8, "fq_conv2d_i8_add_resident")
    _need_cuda(qbias, torch.float32, "fq_conv2d_i8_add_resident")
    if not isinstance(res, torch.Tensor) or res.device.type != "cuda" or res.dtype not in _INT_BYTES:
        raise FqError("fq_conv2d_i8_add_resident: the residual must be an int8 / int16 torch.cuda tensor")
    N, H, W, C = xq.shape
    K, R, S, Cw = wq.shape
    assert C == Cw and xq.is_contiguous() and wq.is_contiguous() and res.is_contiguous() and (want_wide or want_narrow)
    P = (H + 2 * padding[0] - dilation[0] * (R - 1) - 1) // stride[0] + 1
    Q = (W + 2 * padding[1] - dilation[1] * (S - 1) - 1) // stride[1] + 1
    kpad = pad16(K)
    assert tuple(res.shape) == (N, P, Q, kpad), (tuple(res.shape), (N, P, Q, kpad))
    wide = torch.empty(N, P, Q, kpad, dtype=torch.int16, device=xq.device) if want_wide else None
    narrow = torch.empty(N, P, Q, kpad, dtype=torch.int8, device=xq.device) if want_narrow else None
    _check(lib().fq_conv2d_i8_add_resident(xq.data_ptr(), wq.data_ptr(), qbias.contiguous().data_ptr(), res.data_ptr(),
                                           _INT_BYTES[res.dtype], int(g_res), wide.data_ptr() if want_wide else None, int(g_wide),
                                           narrow.data_ptr() if want_narrow else None, int(ib), 1 if relu else 0, kpad, N, H, W, C,
                                           K, R, S, stride[0], stride[1], padding[0], padding[1], dilation[0], dilation[1],
                                           int(rs), int(ob), _stream(xq)), "fq_conv2d_i8_add_resident")
    _note_variant()
    return wide, narrow


def block_tail_supported(C, K3, C2, rs3, rs1, ob3, g_res, res_bytes, ib):
    """Does fq_block_tail_i8 take this chain?  (shapes, integer tails, and grids for which NewAdd's packed-int16 form holds)"""
    return bool(lib().fq_block_tail_i8_supported(int(C), int(K3), int(C2), int(rs3), int(rs1), int(ob3), int(g_res), int(res_bytes),
                                                 int(ib)))


def block_tail_i8(xq, w3q, qbias3, rs3, ob3, res, g_res, want_wide, g_wide, want_narrow, ib, relu, w1q=None, qbias1=None, rs1=0,
                  relu1=False):
    """fq_block_tail_i8: conv3 (1x1) + NewAdd with the shortcut `res` (+ ReLU) + the next block's conv1 (1x1, + ReLU) in one
    kernel.  xq int8 [N,H,W,C], w3q int8 [K3,1,1,C], res int8 / int16 [N,H,W,K3], w1q int8 [C2,1,1,K3] or None.
    Returns (wide int16 or None, narrow int8 or None, q1 int8 [N,H,W,C2] or None)."""
    for t, n in ((xq, "x"), (w3q, "w3")):
        _need_cuda(t, torch.int8, "fq_block_tail_i8")
    _need_cuda(qbias3, torch.float32, "fq_block_tail_i8")
    if not isinstance(res, torch.Tensor) or res.device.type != "cuda" or res.dtype not in _INT_BYTES:
        raise FqError("fq_block_tail_i8: the shortcut must be an int8 / int16 torch.cuda tensor")
    N, H, W, C = xq.shape
    K3 = int(w3q.shape[0])
    assert tuple(w3q.shape[1:]) == (1, 1, C) and xq.is_contiguous() and w3q.is_contiguous() and res.is_contiguous()
    assert tuple(res.shape) == (N, H, W, K3) and qbias3.numel() == K3
    C2 = 0
    if w1q is not None:
        _need_cuda(w1q, torch.int8, "fq_block_tail_i8")
        _need_cuda(qbias1, torch.float32, "fq_block_tail_i8")
        C2 = int(w1q.shape[0])
        assert tuple(w1q.shape[1:]) == (1, 1, K3) and w1q.is_contiguous() and qbias1.numel() == C2
    assert C2 or want_wide or want_narrow
    wide = torch.empty(N, H, W, K3, dtype=torch.int16, device=xq.device) if want_wide else None
    narrow = torch.empty(N, H, W, K3, dtype=torch.int8, device=xq.device) if want_narrow else None
    q1 = torch.empty(N, H, W, C2, dtype=torch.int8, device=xq.device) if C2 else None
    _check(lib().fq_block_tail_i8(xq.data_ptr(), w3q.data_ptr(), qbias3.contiguous().data_ptr(), int(rs3), int(ob3), res.data_ptr(),
                                  _INT_BYTES[res.dtype], int(g_res), wide.data_ptr() if want_wide else None, int(g_wide),
                                  narrow.data_ptr() if want_narrow else None, int(ib), 1 if relu else 0,
                                  w1q.data_ptr() if C2 else None, qbias1.contiguous().data_ptr() if C2 else None, int(rs1),
                                  1 if relu1 else 0, q1.data_ptr() if C2 else None, N * H * W, C, K3, C2, _stream(xq)),
           "fq_block_tail_i8")
    _note_variant()
    return wide, narrow, q1


def block_tail_proj_supported(C, K3, C2, CP, rs3, rs1, rsp, stride_p):
    """Does fq_block_tail_proj_i8 take this chain (the tail of a stage's first block with its projection shortcut computed in the
    kernel)?"""
    return bool(lib().fq_block_tail_proj_i8_supported(int(C), int(K3), int(C2), int(CP), int(rs3), int(rs1), int(rsp), int(stride_p)))


def block_tail_proj_i8(xq, w3q, qbias3, rs3, ob3, xpq, wpq, qbiasp, rsp, obp, stride_p, want_wide, g_wide, want_narrow, ib, relu,
                       w1q=None, qbias1=None, rs1=0, relu1=False):
    """fq_block_tail_proj_i8: fq_block_tail_i8 whose shortcut is the 1x1 projection convolution of xpq (int8 [N,Hp,Wp,CP], weights
    wpq int8 [K3,1,1,CP], stride stride_p), computed inside the kernel.  Returns (wide, narrow, q1) as block_tail_i8."""
    for t in (xq, w3q, xpq, wpq):
        _need_cuda(t, torch.int8, "fq_block_tail_proj_i8")
    for t in (qbias3, qbiasp):
        _need_cuda(t, torch.float32, "fq_block_tail_proj_i8")
    N, H, W, C = xq.shape
    Np, Hp, Wp, CP = xpq.shape
    K3 = int(w3q.shape[0])
    assert tuple(w3q.shape[1:]) == (1, 1, C) and tuple(wpq.shape) == (K3, 1, 1, CP) and Np == N
    assert xq.is_contiguous() and w3q.is_contiguous() and xpq.is_contiguous() and wpq.is_contiguous()
    assert qbias3.numel() == K3 and qbiasp.numel() == K3
    C2 = 0
    if w1q is not None:
        _need_cuda(w1q, torch.int8, "fq_block_tail_proj_i8")
        _need_cuda(qbias1, torch.float32, "fq_block_tail_proj_i8")
        C2 = int(w1q.shape[0])
        assert tuple(w1q.shape[1:]) == (1, 1, K3) and w1q.is_contiguous() and qbias1.numel() == C2
    assert C2 or want_wide or want_narrow
    wide = torch.empty(N, H, W, K3, dtype=torch.int16, device=xq.device) if want_wide else None
    narrow = torch.empty(N, H, W, K3, dtype=torch.int8, device=xq.device) if want_narrow else None
    q1 = torch.empty(N, H, W, C2, dtype=torch.int8, device=xq.device) if C2 else None
    _check(lib().fq_block_tail_proj_i8(xq.data_ptr(), w3q.data_ptr(), qbias3.contiguous().data_ptr(), int(rs3), int(ob3),
                                       xpq.data_ptr(), wpq.data_ptr(), qbiasp.contiguous().data_ptr(), int(rsp), int(obp),
                                       int(stride_p), Hp, Wp, wide.data_ptr() if want_wide else None, int(g_wide),
                                       narrow.data_ptr() if want_narrow else None, int(ib), 1 if relu else 0,
                                       w1q.data_ptr() if C2 else None, qbias1.contiguous().data_ptr() if C2 else None, int(rs1),
                                       1 if relu1 else 0, q1.data_ptr() if C2 else None, N, H, W, C, K3, C2, CP, _stream(xq)),
           "fq_block_tail_proj_i8")
    _note_variant()
    return wide, narrow, q1


def add_resident(x, gx, y, gy, want_wide, g_wide, want_narrow, ib, relu):
    """fq_add_resident on two integer NHWC tensors of one shape: returns (wide int16 or None, narrow int8 or None)."""
    for t in (x, y):
        if not isinstance(t, torch.Tensor) or t.device.type != "cuda" or t.dtype not in _INT_BYTES:
            raise FqError("fq_add_resident: expected int8 / int16 torch.cuda tensors")
    assert x.shape == y.shape and x.is_contiguous() and y.is_contiguous() and (want_wide or want_narrow)
    wide = torch.empty(x.shape, dtype=torch.int16, device=x.device) if want_wide else None
    narrow = torch.empty(x.shape, dtype=torch.int8, device=x.device) if want_narrow else None
    _check(lib().fq_add_resident(x.data_ptr(), _INT_BYTES[x.dtype], int(gx), y.data_ptr(), _INT_BYTES[y.dtype], int(gy),
                                 wide.data_ptr() if want_wide else None, int(g_wide),
                                 narrow.data_ptr() if want_narrow else None, int(ib), 1 if relu else 0, x.numel(),
                                 _stream(x)), "fq_add_resident")
    return wide, narrow


def dequant_nhwc_to_nchw(q, g, channels):
    """int8 / int16 [N, *spatial, Cpad] standing for q * 2^-g  ->  fp32 [N, channels, *spatial]."""
    if not isinstance(q, torch.Tensor) or q.device.type != "cuda" or q.dtype not in _INT_BYTES:
        raise FqError("fq_dequant_nhwc_to_nchw: expected an int8 / int16 torch.cuda tensor")
    assert q.is_contiguous()
    N, cpad = q.shape[0], q.shape[-1]
    spatial = tuple(q.shape[1:-1])
    HW = 1
    for v in spatial:
        HW *= v
    y = torch.empty((N, int(channels)) + spatial, dtype=torch.float32, device=q.device)
    _check(lib().fq_dequant_nhwc_to_nchw(q.data_ptr(), _INT_BYTES[q.dtype], int(g), y.data_ptr(), N, int(channels), HW, cpad,
                                         _stream(q)), "fq_dequant_nhwc_to_nchw")
    return y


def maxpool_i8_nhwc(x, kernel, stride, padding):
    """nn.MaxPool2d on int8 [N, H, W, Cpad] -> int8 [N, P, Q, Cpad] (fq_maxpool_i8_nhwc)."""
    _need_cuda(x, torch.int8, "fq_maxpool_i8_nhwc")
    assert x.dim() == 4 and x.is_contiguous()
    N, H, W, cpad = x.shape
    P = (H + 2 * padding[0] - kernel[0]) // stride[0] + 1
    Q = (W + 2 * padding[1] - kernel[1]) // stride[1] + 1
    y = torch.empty(N, P, Q, cpad, dtype=torch.int8, device=x.device)
    _check(lib().fq_maxpool_i8_nhwc(x.data_ptr(), y.data_ptr(), N, H, W, cpad, kernel[0], kernel[1], stride[0], stride[1],
                                    padding[0], padding[1], _stream(x)), "fq_maxpool_i8_nhwc")
    return y


def avgpool_global_nhwc(q, g, channels):
    """int8 / int16 [N, H, W, Cpad] standing for q * 2^-g -> fp32 [N, channels, 1, 1], the mean over the plane."""
    if not isinstance(q, torch.Tensor) or q.device.type != "cuda" or q.dtype not in _INT_BYTES:
        raise FqError("fq_avgpool_global_nhwc: expected an int8 / int16 torch.cuda tensor")
    assert q.dim() == 4 and q.is_contiguous()
    N, H, W, cpad = q.shape
    y = torch.empty(N, int(channels), 1, 1, dtype=torch.float32, device=q.device)
    _check(lib().fq_avgpool_global_nhwc(q.data_ptr(), _INT_BYTES[q.dtype], int(g), y.data_ptr(), N, int(channels), H * W, cpad,
                                        _stream(q)), "fq_avgpool_global_nhwc")
    return y


def quantize_i8_unfold_w(x, ib, S, stride_w, pad_w, dil_w, cpad2):
    """Stem input: fp32 [N, C<=4, H, W] -> int8 [N, H, Q, cpad2] with the kernel width folded into the
    channel axis (folded channel = s*C + c).  See fq_quantize_i8_unfold_w in include/fq.h."""
    _need_cuda(x, torch.float32, "fq_quantize_i8_unfold_w")
    xc = x.contiguous()
    N, C, H, W = xc.shape
    Q = (W + 2 * pad_w - dil_w * (S - 1) - 1) // stride_w + 1
    y = torch.empty(N, H, Q, cpad2, dtype=torch.int8, device=x.device)
    _check(lib().fq_quantize_i8_unfold_w(xc.data_ptr(), y.data_ptr(), N, C, H, W, int(S), int(stride_w), int(pad_w),
                                         int(dil_w), int(cpad2), int(ib), _stream(xc)), "fq_quantize_i8_unfold_w")
    return y


def pack_weight_unfold_w(w, cpad2):
    """Integer-valued fp32 weights [K, C, R, S] -> int8 [K, R, 1, cpad2] with folded channel s*C + c."""
    K, C, R, S = w.shape
    out = torch.zeros(K, R, 1, cpad2, dtype=torch.int8, device=w.device)
    out[:, :, 0, :S * C] = w.permute(0, 2, 3, 1).reshape(K, R, S * C).to(torch.int8)
    return out.contiguous()


def read_npy_batch(paths, header, dst, threads=4):
    """fq_read_npy_batch_f32: the payloads of the .npy files `paths` (all with the header bytes `header`) into dst[i]
    (a float32 HOST tensor [len(paths), ...], contiguous -- pinned if an asynchronous upload follows).  Returns the list of
    per-file success flags.  One foreign call: the interpreter lock is released for all the reads."""
    n = len(paths)
    if n == 0:
        return []
    assert dst.device.type == "cpu" and dst.dtype == torch.float32 and dst.is_contiguous() and dst.shape[0] == n
    arr = (ctypes.c_char_p * n)(*[os.fsencode(p) for p in paths])
    ok = (ctypes.c_int * n)()
    elems = dst[0].numel()
    _check(lib().fq_read_npy_batch_f32(ctypes.cast(arr, ctypes.c_void_p), n, header, len(header), dst.data_ptr(), elems,
                                       int(threads), ctypes.cast(ok, ctypes.c_void_p)), "fq_read_npy_batch_f32")
    return [bool(v) for v in ok]
