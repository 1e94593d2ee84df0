"""The quantize-op ``nn.Module`` API: integer-simulation conv/linear (NewConv2d / NewLinear),
fake-quant (QuanDequan, TestConv / TestLinear) and the element-wise pieces they are made of.

Drop-in for reference quantity/common/quantity/new_quantity_op.py: same class names, constructor
signatures, attribute names and `quantize_infor` keys (weight_bit, bias_bit, input_bit, output_bit),
so models pickled by `tools.Reconstruction` load against this module path.

What runs underneath is different: every forward is a hand-written HIP kernel reached through the
C ABI (include/fq.h).  The reference composes each op from 3-7 torch element-wise calls; here each
op is one pass, and NewConv2d / NewLinear fuse RightShift -> BiasAdd -> Sp -> DeQuantity into the
epilogue of the contraction.  Forward passes require CUDA (ROCm) tensors: there is no CPU path.

Reference lines: RightShift :11-44, Quantity :48-58, DeQuantity :61-68, Sp :71-91, BiasAdd :95-101,
NewConv2d :104-163, NewAdd :166-174, NewLinear :177-236, QuanDequan :239-257, TestConv :259-355,
TestLinear :358-452.
"""
import os
import weakref

import torch
from torch import nn

from . import _native, _float_conv
from .resident import QHandle, DeferredConv, resident_of, as_f32, carry

QUANTIZE_BIT = 8

# TestConv / TestLinear can dump weights as text and 2048-bin PNG histograms the way the reference
# constructor always does (174 s for ResNet-18).  Off unless asked for.
DUMP_VISUALIZATION = os.environ.get("FQ_DUMP_VISUALIZATION", "0") not in ("", "0", "false", "False")

__all__ = ["RightShift", "Sp", "BiasAdd", "NewConv2d", "NewAdd", "NewLinear", "QuanDequan", "TestConv",
           "TestLinear", "Quantity", "DeQuantity", "QUANTIZE_BIT"]


def _check_width(bits):
    assert bits == 8 or bits == 16, "Not support bit width."


class RightShift(nn.Module):
    """x / 2^rs, rounded half away from zero, saturated to `bits` wide integers (kept as fp32)."""

    def __init__(self, bits, rs):
        super(RightShift, self).__init__()
        self.rs = rs
        self.Bit_width = bits

    def forward(self, x):
        _check_width(self.Bit_width)
        return _native.rightshift(x, self.rs, self.Bit_width)


class Quantity(nn.Module):
    """clamp(round_half_even(x * 2^ib)) to the QUANTIZE_BIT range."""

    def __init__(self, ib):
        super(Quantity, self).__init__()
        self.ib = ib

    def forward(self, x):
        return _native.quantity(x, self.ib, 8 if QUANTIZE_BIT == 8 else 16)


class DeQuantity(nn.Module):
    """x / 2^ob."""

    def __init__(self, ob):
        super(DeQuantity, self).__init__()
        self.ob = ob

    def forward(self, x):
        return _native.dequantity(x, self.ob)


class Sp(nn.Module):
    """Saturating truncation to the int8 / int16 range."""

    def __init__(self, bits):
        super(Sp, self).__init__()
        self.bitwidth = bits

    def forward(self, x):
        _check_width(self.bitwidth)
        return _native.sp(x, self.bitwidth)


class BiasAdd(nn.Module):

    def __init__(self):
        super(BiasAdd, self).__init__()

    def forward(self, x, y):
        return torch.add(x, y)


def _quantize_params(layer, weight_bit, bias_bit, out_count, weight_16bit_range=False):
    """Integer-valued fp32 weights / bias of a conv or linear layer (reference :135-163, :208-236).
    One-time parameter preparation, done with torch ops on whatever device the layer lives on."""
    assert layer.weight is not None, "The layer weight can`t be None"
    w = layer.weight.data
    b = layer.bias.data if layer.bias is not None else torch.zeros(out_count, device=w.device, dtype=w.dtype)
    w_width = 8 if (QUANTIZE_BIT == 8 or not weight_16bit_range) else 16
    b_width = 8 if QUANTIZE_BIT == 8 else 16
    return _round_clamp(w, weight_bit, w_width), _round_clamp(b, bias_bit, b_width)


def _round_clamp(t, bit, width):
    """clamp(round_half_even(t * 2^bit)) -- the Quantity kernel when the parameter already lives on
    the GPU, the same torch expression the reference uses when the model is still on the host."""
    if t.device.type == "cuda" and t.dtype == torch.float32:
        return _native.quantity(t, bit, width)
    r = torch.round(torch.mul(t, pow(2, bit)))
    return r.clamp(-128, 127) if width == 8 else r.clamp(-32768.0, 32767.0)


class _QuantizedInputCache(object):
    """One-entry memo of fq_quantize_i8_nhwc results.  A residual block hands the SAME tensor to its
    first conv and to its projection shortcut, both with the same input bit; the second caller reuses
    the int8 NHWC copy instead of re-reading the fp32 tensor.  Keyed by tensor identity + version.

    HIP-graph capture: an entry made outside a capture is never served inside one (the quantise launch
    would be missing from the graph and every replay would convolve the capture-time input), and an
    entry made inside a capture lives in the graph's private pool, so it is only served to the same
    capture (resident.GraphedForward clears the memo before and after capturing).  The input tensor is
    held weakly: the memo does not keep the last batch alive."""

    def __init__(self):
        self._key, self._ref, self._val = None, None, None

    def get(self, x, ib, cpad):
        capturing = bool(x.is_cuda and torch.cuda.is_current_stream_capturing())
        key = (id(x), x._version, x.data_ptr(), tuple(x.shape), int(ib), int(cpad), capturing)
        if key == self._key and self._ref is not None and self._ref() is x:
            return self._val
        val = _native.quantize_i8_nhwc(x, ib, cpad)
        self._key, self._val = key, val
        self._ref = weakref.ref(x, self._drop)
        return val

    def _drop(self, ref):
        if self._ref is ref:
            self.clear()

    def clear(self):
        self._key, self._ref, self._val = None, None, None


_xq_cache = _QuantizedInputCache()


class _IntegerSimLayer(nn.Module):
    """Shared body of NewConv2d / NewLinear: Quantity -> integer contraction -> fused tail.

    Default forward = two HIP kernels: fq_quantize_i8_nhwc (Quantity fused with the NCHW->NHWC int8
    repack) and fq_conv2d_i8 (int8 MFMA implicit GEMM, int32 accumulation, whole tail fused).
    `use_int8_mfma = False` keeps the reference's structure instead (Quantity kernel -> fp32 conv on
    integer-valued tensors -> fused tail kernel); both are exact below 2^24 per partial sum."""

    use_int8_mfma = True
    use_stem_kernel = True        # False: the stem goes through fq_quantize_i8_unfold_w + the general kernel (same integers)

    def _int8_ok(self, layer):
        if not self.use_int8_mfma or QUANTIZE_BIT != 8:
            return False
        if isinstance(layer, nn.Conv2d):
            return layer.groups == 1 and layer.padding_mode == "zeros" and not isinstance(layer.padding, str)
        return isinstance(layer, nn.Linear)

    @staticmethod
    def _stem_fold(layer):
        """Cpad2 if the conv is a stem-like layer (<= 4 input channels, kernel wider than 1) whose width
        is better folded into the channel axis, else 0."""
        if isinstance(layer, nn.Conv2d) and layer.in_channels <= 4 and layer.kernel_size[1] > 1:
            return _native.pad16(layer.kernel_size[1] * layer.in_channels)
        return 0

    def _packed_weight(self, layer):
        w = layer.weight
        cached = getattr(self, "_w_i8", None)
        if cached is None or cached[0] != (w.data_ptr(), w._version, str(w.device)):
            fold = self._stem_fold(layer)
            packed = _native.pack_weight_unfold_w(w.detach(), fold) if fold else _native.pack_weight_krsc(w.detach())
            object.__setattr__(self, "_w_i8", ((w.data_ptr(), w._version, str(w.device)), packed))
            cached = self._w_i8
        return cached[1]

    def _stem_weight(self, layer):
        """Weights packed for fq_conv2d_i8_stem (one kernel for the whole stem layer), cached like _packed_weight."""
        w = layer.weight
        cached = getattr(self, "_w_stem", None)
        if cached is None or cached[0] != (w.data_ptr(), w._version, str(w.device)):
            object.__setattr__(self, "_w_stem", ((w.data_ptr(), w._version, str(w.device)), _native.pack_weight_stem(w.detach())))
            cached = self._w_stem
        return cached[1]

    def __getstate__(self):
        state = self.__dict__.copy()
        state.pop("_w_i8", None)                  # derived data: rebuilt on first forward after loading
        state.pop("_w_stem", None)
        return state

    def _setup(self, layer, quantize_infor, out_count, wide_weights):
        self.weight_bit = quantize_infor["weight_bit"]
        self.bias_bit = quantize_infor["bias_bit"]
        self.input_bit = quantize_infor["input_bit"]
        self.output_bit = quantize_infor["output_bit"]
        self.rs_bit = self.weight_bit + self.input_bit - self.output_bit
        self.Quan = Quantity(self.input_bit)
        self.RightShift = RightShift(QUANTIZE_BIT, self.rs_bit)
        self.BiasAdd = BiasAdd()
        self.Sp = Sp(QUANTIZE_BIT)
        self.DeQuan = DeQuantity(self.output_bit)
        # as in the reference, .weight / .bias keep the layer's ORIGINAL float parameters
        self.weight = layer.weight
        self.bias = layer.bias
        qw, qb = _quantize_params(layer, self.weight_bit, self.bias_bit, out_count, wide_weights)
        # the contraction carries no bias; the quantised bias is added after the shift
        layer.weight = nn.Parameter(qw)
        layer.bias = nn.Parameter(torch.zeros(out_count, device=qw.device, dtype=qw.dtype))
        # the reference keeps this as a plain attribute, so .cuda() leaves it behind (its ReconModel
        # is CPU-only in practice); a buffer moves with the module and is pickled the same way
        # (non-persistent: state_dict keys stay the reference's)
        self.register_buffer("quantized_bias", qb, persistent=False)

    def _tail(self, acc):
        return _native.recon_epilogue(acc, self.quantized_bias, self.rs_bit, self.output_bit,
                                      8 if QUANTIZE_BIT == 8 else 16, out=acc)


class NewConv2d(_IntegerSimLayer):
    """Integer simulation of a convolution: int8 activations x int8 weights accumulated exactly,
    shifted right by weight_bit + input_bit - output_bit, bias added, saturated, de-quantised."""

    def __init__(self, conv_module, quantize_infor):
        super(NewConv2d, self).__init__()
        self.Conv = conv_module
        self._setup(conv_module, quantize_infor, conv_module.out_channels, False)

    def forward(self, input):
        conv = self.Conv
        ready = getattr(input, "next_out", None) if type(input) is QHandle else None
        if ready is not None and ready[0] is self:        # the NewAdd that produced `input` ran this convolution in its kernel
            return ready[1]
        if self._int8_ok(conv):
            wq = self._packed_weight(conv)
            plan = self.__dict__.get("_resident")         # set by common.quantity.resident.enable()
            fold = self._stem_fold(conv)
            if (fold and plan is not None and plan.emit_int and not plan.emit_f32 and not plan.defer and self.use_stem_kernel
                    and _native.stem_supported(conv.in_channels, conv.out_channels, conv.kernel_size[0], conv.kernel_size[1],
                                               conv.stride, conv.dilation, self.rs_bit)):
                # the whole layer in one kernel: fp32 image in, int8 NHWC out (no unfolded copy of the image)
                x = as_f32(input)
                q = _native.conv2d_i8_stem(x if x.is_contiguous() else x.contiguous(), self._stem_weight(conv),
                                           self.quantized_bias, conv.out_channels, conv.kernel_size[1], conv.stride,
                                           conv.padding, self.input_bit, self.rs_bit, self.output_bit, plan.relu)
                return QHandle((q.shape[0], conv.out_channels, q.shape[1], q.shape[2]), q, self.output_bit, q,
                               self.output_bit, plan.relu)
            if fold:
                input = as_f32(input)
                xq = _native.quantize_i8_unfold_w(input, self.input_bit, conv.kernel_size[1], conv.stride[1],
                                                  conv.padding[1], conv.dilation[1], wq.shape[-1])
                geom = ((conv.stride[0], 1), (conv.padding[0], 0), (conv.dilation[0], 1))
            else:
                xq = self._resident_input(input, wq.shape[-1])
                if xq is None:
                    xq = _xq_cache.get(as_f32(input), self.input_bit, wq.shape[-1])
                geom = (conv.stride, conv.padding, conv.dilation)
            if plan is None:
                return _native.conv2d_i8(xq, wq, self.quantized_bias, geom[0], geom[1], geom[2], self.rs_bit,
                                         self.output_bit, 8)
            if plan.defer:
                return DeferredConv(self, xq, wq, geom)   # the resident NewAdd that consumes it runs it
            y, q = _native.conv2d_i8_resident(xq, wq, self.quantized_bias, geom[0], geom[1], geom[2], self.rs_bit,
                                              self.output_bit, plan.emit_f32, plan.emit_int, plan.relu)
            handle = None
            if q is not None:
                handle = QHandle((q.shape[0], conv.out_channels, q.shape[1], q.shape[2]), q, self.output_bit, q,
                                 self.output_bit, plan.relu)
            if y is None:
                return handle
            if handle is not None:
                carry(y, handle)
            if plan.relu:
                y._fq_relu_done = True
            return y
        q = self.Quan(as_f32(input))
        acc = conv(q)               # integer-valued fp32 in, exact below 2^24 per partial sum
        return self._tail(acc)

    def _resident_input(self, input, cpad):
        """int8 NHWC operand already in HBM (left by the producer), or None."""
        h = resident_of(input)
        if h is None:
            return None
        if h.narrow is not None and h.bit == self.input_bit and h.narrow.shape[-1] == cpad:
            return h.narrow
        if h.exact is not None and h.exact.dtype == torch.int8 and h.grid == self.input_bit and h.exact.shape[-1] == cpad:
            return h.exact
        return None


class NewLinear(_IntegerSimLayer):

    def __init__(self, linear_module, quantize_infor):
        super(NewLinear, self).__init__()
        self.Linear = linear_module
        self._setup(linear_module, quantize_infor, linear_module.out_features, True)

    def forward(self, input):
        lin = self.Linear
        input = as_f32(input)
        if self._int8_ok(lin) and input.dim() == 2:
            wq = self._packed_weight(lin)
            xq = _native.quantize_i8_nhwc(input, self.input_bit, wq.shape[-1])
            return _native.conv2d_i8(xq, wq, self.quantized_bias, (1, 1), (0, 0), (1, 1), self.rs_bit,
                                     self.output_bit, 8)
        q = self.Quan(input)
        acc = lin(q)
        return self._tail(acc)


class NewAdd(nn.Module):
    """Residual add followed by the int8 saturation (applied to de-quantised values, as in the
    reference, where it is in effect a clamp to [-128, 127])."""

    def __init__(self):
        super(NewAdd, self).__init__()
        self.Sp = Sp(QUANTIZE_BIT)

    def forward(self, x, y):
        plan = self.__dict__.get("_resident")             # set by common.quantity.resident.enable()
        if plan is not None and plan.resident_add and self.Sp.bitwidth == 8:
            fused = self._fused_conv_add(plan, x, y)
            if fused is not None:
                return fused
            hx, hy = resident_of(x), resident_of(y)
            if (hx is not None and hy is not None and hx.exact is not None and hy.exact is not None
                    and hx.exact.shape == hy.exact.shape and max(0, hx.grid, hy.grid) == plan.grid):
                want_narrow = plan.emit_int and plan.narrow_bit is not None
                wide, narrow = _native.add_resident(hx.exact, hx.grid, hy.exact, hy.grid, plan.want_wide or not want_narrow,
                                                    plan.grid, want_narrow, plan.narrow_bit if want_narrow else 0, plan.relu)
                handle = QHandle(hx.shape, wide, plan.grid, narrow, plan.narrow_bit, plan.relu)
                if not plan.emit_f32:
                    return handle
                out = handle.to_f32()
                if plan.emit_int:
                    carry(out, handle)
                if plan.relu:
                    out._fq_relu_done = True
                return out
        out = _native.add_sat(as_f32(x), as_f32(y), self.Sp.bitwidth)
        if plan is not None and plan.relu:
            out = torch.relu_(out)                        # the ReLU module after this add passes through
            out._fq_relu_done = True
        return out


_BT_ALONE_MAX_C = int(os.environ.get("FQ_BT_ALONE_MAX_C", "64"))     # conv3 + NewAdd WITHOUT a next conv1 on fq_block_tail_i8 up to this width


def _block_tail_on():
    return os.environ.get("FQ_BLOCK_TAIL", "1") != "0"


def _newadd_fused_conv_add(self, plan, x, y):
    """conv -> add in one kernel when one operand is a DeferredConv and the other a resident activation."""
    if type(x) is DeferredConv and type(y) is DeferredConv:
        fused = _newadd_fused_conv_proj_add(self, plan, x, y)
        if fused is not None:
            return fused
        # (not taken after all: the projection runs as its own launch and the add sees its handle)
        if plan.fuse_arg == 0:
            y = y.materialise()
        else:
            x = x.materialise()
    d, other = (x, y) if type(x) is DeferredConv else ((y, x) if type(y) is DeferredConv else (None, None))
    if d is None or type(other) is DeferredConv or d._handle is not None:
        return None
    h = resident_of(other)
    L = d.layer
    if h is None or h.exact is None or max(0, L.output_bit, h.grid) != plan.grid or plan.emit_f32:
        return None
    want_narrow = plan.emit_int and plan.narrow_bit is not None
    want_wide = plan.want_wide or not want_narrow
    if h.exact.shape[0] != d.xq.shape[0] or h.exact.shape[-1] != _native.pad16(L.Conv.out_channels):
        return None
    nxt = plan.fuse_next
    one = tuple(tuple(int(v) for v in g) for g in d.geom) == ((1, 1), (0, 0), (1, 1)) and tuple(d.wq.shape[1:3]) == (1, 1)
    if (nxt is None and one and tuple(h.exact.shape[:3]) == tuple(d.xq.shape[:3]) and _block_tail_on()
            and d.xq.shape[-1] <= _BT_ALONE_MAX_C and L.Conv.out_channels == d.wq.shape[0]
            and _native.block_tail_supported(d.xq.shape[-1], L.Conv.out_channels, 0, L.rs_bit, 0, L.output_bit, h.grid, h.exact.element_size(),
                                             plan.narrow_bit if want_narrow else plan.grid - 1)):
        # conv3 + NewAdd alone on the same kernel (no next convolution to fuse) for the 64-channel stage, whose tensors come from
        # HBM: its barrier-free, wave-local epilogue streams them 1.5-1.6 x faster than the general kernel's when nothing is
        # cache resident (scripts/block_tail_probe.py), 6 % faster inside the network.  Deeper stages live in the Infinity
        # Cache at these sizes and the general kernel's 3 workgroups per CU win there.
        wide, narrow, _ = _native.block_tail_i8(d.xq, d.wq, L.quantized_bias, L.rs_bit, L.output_bit, h.exact, h.grid, want_wide,
                                                plan.grid, want_narrow, plan.narrow_bit if want_narrow else 0, plan.relu)
        ref = wide if wide is not None else narrow
        return QHandle((ref.shape[0], L.Conv.out_channels, ref.shape[1], ref.shape[2]), wide, plan.grid, narrow, plan.narrow_bit,
                       plan.relu)
    if nxt is not None and want_narrow and want_wide and tuple(h.exact.shape[:3]) == tuple(d.xq.shape[:3]):
        # conv3 + NewAdd + the next block's conv1 in one kernel (fq_block_tail_i8): the re-quantised sum is that convolution's
        # operand, staged in LDS, and reaches HBM only if somebody else reads it too
        np_ = nxt.__dict__.get("_resident")
        w1 = nxt._packed_weight(nxt.Conv)
        if (np_ is not None and w1.shape[-1] == L.Conv.out_channels
                and _native.block_tail_supported(d.xq.shape[-1], L.Conv.out_channels, nxt.Conv.out_channels, L.rs_bit, nxt.rs_bit,
                                                 L.output_bit, h.grid, h.exact.element_size(), plan.narrow_bit)):
            wide, narrow, q1 = _native.block_tail_i8(d.xq, d.wq, L.quantized_bias, L.rs_bit, L.output_bit, h.exact, h.grid, True,
                                                     plan.grid, plan.narrow_to_hbm, plan.narrow_bit, plan.relu, w1,
                                                     nxt.quantized_bias, nxt.rs_bit, np_.relu)
            out = QHandle((wide.shape[0], L.Conv.out_channels, wide.shape[1], wide.shape[2]), wide, plan.grid, narrow,
                          plan.narrow_bit, plan.relu)
            out.next_out = (nxt, QHandle((q1.shape[0], nxt.Conv.out_channels, q1.shape[1], q1.shape[2]), q1, nxt.output_bit, q1,
                                         nxt.output_bit, np_.relu))
            return out
    wide, narrow = _native.conv2d_i8_add_resident(d.xq, d.wq, L.quantized_bias, d.geom[0], d.geom[1], d.geom[2], L.rs_bit,
                                                  L.output_bit, h.exact, h.grid, want_wide, plan.grid, want_narrow,
                                                  plan.narrow_bit if want_narrow else 0, plan.relu)
    ref = wide if wide is not None else narrow
    return QHandle((ref.shape[0], L.Conv.out_channels, ref.shape[1], ref.shape[2]), wide, plan.grid, narrow, plan.narrow_bit,
                   plan.relu)


def _newadd_fused_conv_proj_add(self, plan, x, y):
    """The first block of a stage: conv3 -> add <- projection, (-> the next block's conv1), in ONE kernel (fq_block_tail_proj_i8)
    when both operands arrive as DeferredConvs and the plan says so; None otherwise."""
    if not plan.fuse_proj or plan.fuse_arg is None or plan.emit_f32 or not _block_tail_on():
        return None
    d, dp = (x, y) if plan.fuse_arg == 0 else (y, x)
    if d._handle is not None or dp._handle is not None:
        return None
    L, P = d.layer, dp.layer
    one = lambda dd: tuple(tuple(int(v) for v in g) for g in dd.geom[1:]) == ((0, 0), (1, 1)) and tuple(dd.wq.shape[1:3]) == (1, 1)
    sp = tuple(int(v) for v in dp.geom[0])
    if (not one(d) or not one(dp) or tuple(int(v) for v in d.geom[0]) != (1, 1) or sp[0] != sp[1]
            or max(0, L.output_bit, P.output_bit) != plan.grid or d.xq.shape[0] != dp.xq.shape[0]
            or (dp.xq.shape[1] - 1) // sp[0] + 1 != d.xq.shape[1] or (dp.xq.shape[2] - 1) // sp[0] + 1 != d.xq.shape[2]
            or L.Conv.out_channels != d.wq.shape[0] or P.Conv.out_channels != dp.wq.shape[0] or d.wq.shape[0] != dp.wq.shape[0]):
        return None
    want_narrow = plan.emit_int and plan.narrow_bit is not None
    want_wide = plan.want_wide or not want_narrow
    nxt = plan.fuse_next
    np_ = nxt.__dict__.get("_resident") if nxt is not None else None
    if nxt is not None and (np_ is None or not want_narrow or not want_wide):
        return None
    w1 = nxt._packed_weight(nxt.Conv) if nxt is not None else None
    if w1 is not None and w1.shape[-1] != L.Conv.out_channels:
        return None
    if not _native.block_tail_proj_supported(d.xq.shape[-1], L.Conv.out_channels, nxt.Conv.out_channels if nxt is not None else 0,
                                             dp.xq.shape[-1], L.rs_bit, nxt.rs_bit if nxt is not None else 0, P.rs_bit, sp[0]):
        return None
    if nxt is not None:
        wide, narrow, q1 = _native.block_tail_proj_i8(d.xq, d.wq, L.quantized_bias, L.rs_bit, L.output_bit, dp.xq, dp.wq,
                                                      P.quantized_bias, P.rs_bit, P.output_bit, sp[0], True, plan.grid,
                                                      plan.narrow_to_hbm, plan.narrow_bit, plan.relu, w1, nxt.quantized_bias,
                                                      nxt.rs_bit, np_.relu)
        out = QHandle((wide.shape[0], L.Conv.out_channels, wide.shape[1], wide.shape[2]), wide, plan.grid, narrow, plan.narrow_bit,
                      plan.relu)
        out.next_out = (nxt, QHandle((q1.shape[0], nxt.Conv.out_channels, q1.shape[1], q1.shape[2]), q1, nxt.output_bit, q1,
                                     nxt.output_bit, np_.relu))
        return out
    wide, narrow, _ = _native.block_tail_proj_i8(d.xq, d.wq, L.quantized_bias, L.rs_bit, L.output_bit, dp.xq, dp.wq, P.quantized_bias,
                                                 P.rs_bit, P.output_bit, sp[0], want_wide, plan.grid, want_narrow,
                                                 plan.narrow_bit if want_narrow else 0, plan.relu)
    ref = wide if wide is not None else narrow
    return QHandle((ref.shape[0], L.Conv.out_channels, ref.shape[1], ref.shape[2]), wide, plan.grid, narrow, plan.narrow_bit,
                   plan.relu)


NewAdd._fused_conv_add = _newadd_fused_conv_add


class QuanDequan(nn.Module):
    """Fake quantisation: clamp(round_half_even(x * 2^bit)) / 2^bit in one fused pass."""

    def __init__(self, Bitwidth, bit):
        super(QuanDequan, self).__init__()
        self.bitwidth = Bitwidth
        self.bit = bit

    def forward(self, quantized_x, out=None):
        return _native.quandequan(quantized_x, self.bit, 8 if self.bitwidth == 8 else 16, out=out)


def _fake_quant_param(t, bit, bitwidth):
    """One-time fake quantisation of a parameter tensor at construction (reference :305-309): the
    QuanDequan kernel for parameters on the GPU, the reference's torch expression for a model that
    is still on the host (Reconstruction is usually run before .cuda()).  Not a forward path."""
    if t.device.type == "cuda" and t.dtype == torch.float32:
        return _native.quandequan(t, bit, 8 if bitwidth == 8 else 16)
    s = pow(2, bit)
    r = torch.round(torch.mul(t, s))
    r = r.clamp(-128, 127) if bitwidth == 8 else r.clamp(-32768.0, 32767.0)
    return torch.div(r, s)


class _FakeQuantLayer(nn.Module):
    """Shared body of TestConv / TestLinear: weights and bias fake-quantised once, output
    fake-quantised on every forward."""

    def _setup(self, name, layer, quantize_infor, new_model_path, out_count):
        self.name = name
        self.path = os.path.join(os.path.dirname(new_model_path), "quantity_results")
        if not os.path.exists(self.path):
            os.makedirs(self.path)
        self.weight_bit = quantize_infor["weight_bit"]
        self.bias_bit = quantize_infor["bias_bit"]
        self.input_bit = quantize_infor["input_bit"]
        self.output_bit = quantize_infor["output_bit"]
        self.weight_qdp = QuanDequan(QUANTIZE_BIT, self.weight_bit)
        self.bias_qdp = QuanDequan(QUANTIZE_BIT, self.bias_bit)
        self.output_qdp = QuanDequan(QUANTIZE_BIT, self.output_bit)
        self.feature_extract(layer, out_count)

    def feature_extract(self, layer=None, out_count=None):
        if layer is None:
            layer = self.Conv if hasattr(self, "Conv") else self.linear
        assert layer.weight is not None, "The layer weight can`t be None"
        w = layer.weight.data
        if layer.bias is None:
            # (the reference dereferences a missing attribute here; a zero bias is what it meant)
            b = torch.zeros(out_count, device=w.device, dtype=w.dtype)
        else:
            b = layer.bias.data
        self.weight = layer.weight          # originals, as in the reference
        self.bias = layer.bias
        w_q = _fake_quant_param(w, self.weight_qdp.bit, self.weight_qdp.bitwidth)
        b_q = _fake_quant_param(b, self.bias_qdp.bit, self.bias_qdp.bitwidth)
        layer.weight = nn.Parameter(w_q)
        layer.bias = nn.Parameter(b_q)
        if DUMP_VISUALIZATION:
            self._dump(w, b, w_q, b_q)

    def _dump(self, w, b, w_q, b_q):
        stem = os.path.join(self.path, self.name.replace(".", "_"))

        def as_text(t):
            return "  ".join(str(v) for v in t.detach().cpu().numpy().flatten())

        with open(stem + "_weight.txt", "a") as fh:
            fh.write(as_text(w) + "\n\n\n\n")
            fh.write(as_text(w_q) + "\n\n\n\n")
        with open(stem + "_bias.txt", "a") as fh:
            fh.write(as_text(b) + "\n\n\n\n")
            fh.write(as_text(b_q) + "\n\n\n\n")
        self.plot_hist(w.cpu().numpy(), 2048, stem + "_weight_o.png", title="weight")
        self.plot_hist(b.cpu().numpy(), 2048, stem + "_bias_o.png", title="bias")
        self.plot_hist(w_q.cpu().numpy(), 2048, stem + "_weight_q.png", title="weight")
        self.plot_hist(b_q.cpu().numpy(), 2048, stem + "_bias_q.png", title="bias")

    def plot_hist(self, ndarray, bins, save_path, title):
        if save_path is None:
            raise NotImplementedError("the path is not exists")
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
        fig = plt.figure()
        plt.grid()
        plt.title(title)
        plt.xlabel("bins")
        plt.ylabel("counter/frequency")
        plt.hist(ndarray.flatten(), bins, density=True, histtype="bar", facecolor="blue")
        fig.savefig(save_path, bbox_inches="tight")
        plt.close(fig)


class TestConv(_FakeQuantLayer):
    __test__ = False        # not a pytest class

    def __init__(self, name, module, quantize_infor, new_model_path):
        super(TestConv, self).__init__()
        self.Conv = module
        self._setup(name, module, quantize_infor, new_model_path, module.out_channels)

    def forward(self, x):
        # One kernel where the convolution is this library's (1x1, R x S with zero padding, the 7x7 stem) and nobody hooks
        # the inner nn.Conv2d: QuanDequan rides in the epilogue, the reference's second 8 B/element pass disappears.
        # (the reference calls output_qdp as a module, new_quantity_op.py:284: a hook on it must fire, and its own `bit` /
        # `bitwidth` -- not a copy taken at construction -- decide the map; either keeps the two-pass form or feeds the kernel)
        qdp = self.output_qdp
        fused = None if _float_conv._hooked(qdp) else _float_conv.call_qd(self.Conv, x, qdp.bit, qdp.bitwidth)
        if fused is not None:
            return fused
        out = _float_conv.call(self.Conv, x)
        return self.output_qdp(out, out=out if out.is_contiguous() else None)


class TestLinear(_FakeQuantLayer):
    __test__ = False

    def __init__(self, name, module, quantize_infor, new_model_path):
        super(TestLinear, self).__init__()
        self.linear = module
        self._setup(name, module, quantize_infor, new_model_path, module.out_features)

    def forward(self, x):
        # one kernel, as TestConv: the linear layer is a 1x1 convolution of a 1 x 1 plane (fq_conv1x1_qd_f32); hooks on the
        # nn.Linear or on output_qdp keep the reference's two calls
        qdp = self.output_qdp
        fused = None if _float_conv._hooked(qdp) else _float_conv.call_linear_qd(self.linear, x, qdp.bit, qdp.bitwidth)
        if fused is not None:
            return fused
        out = self.linear(x)
        return self.output_qdp(out, out=out if out.is_contiguous() else None)
