"""Resident integer activations for the integer-simulation model (ReconModel).

The reference's NewConv2d / NewAdd hand fp32 NCHW tensors from module to module
(new_quantity_op.py:124-133, :171-174): DeQuantity writes 4 bytes per activation, nn.ReLU reads and
writes them, the next layer's Quantity reads them again and recovers -- exactly -- the integer the
previous layer's tail already held.  `enable(model, example)` keeps that integer in HBM instead:

  * one traced forward (module hooks + a TorchFunctionMode) records, for the output of every
    NewConv2d / NewAdd, who consumes it: another integer layer, a ReLU, or anything else;
  * producers whose consumers are integer layers emit int8 NHWC straight from the MFMA epilogue
    (conv) or int16 + int8 NHWC from the fused add kernel, with the following nn.ReLU folded in
    (max(., 0) commutes with the power-of-two scale); consumers read those bytes directly;
  * anything else still receives the ordinary fp32 NCHW tensor (a value used by both kinds gets both).

Every resident value is `integer * 2^-grid` and stands for exactly the fp32 number the reference
would have produced, so model outputs are bit-identical to the fp32-boundary path
(tests/test_gpu_resident.py); the saving is HBM traffic: 18-29 bytes per activation become 2-6.

The plan assumes a static dataflow (as a captured graph would).  Consumers stay adaptive -- a layer
that unexpectedly receives fp32 quantises it as usual -- and a resident handle reaching code that
is not part of the plan raises instead of computing garbage.
"""
import torch
from torch import nn
from torch.overrides import TorchFunctionMode

from . import _native

__all__ = ["QHandle", "enable", "disable", "is_enabled", "describe", "capture", "GraphedForward", "MultiStreamGraphs"]

MAX_WIDE_GRID = 8          # int16 holds |s| <= 128 on a grid of 2^-8


class QHandle(object):
    """An activation kept as integers in HBM.  Not a tensor on purpose: code outside the plan that
    touches it fails loudly.

    exact : int8 / int16 NHWC tensor, value = exact * 2^-grid (the reference's fp32 value, exactly)
    narrow: int8 NHWC tensor = Quantity(value, bit) for the next conv (conv outputs: the same tensor)
    """
    __slots__ = ("shape", "exact", "grid", "narrow", "bit", "relu_done", "next_out")

    def __init__(self, shape, exact, grid, narrow, bit, relu_done):
        self.shape, self.exact, self.grid, self.narrow, self.bit, self.relu_done = shape, exact, grid, narrow, bit, relu_done
        # (consumer NewConv2d, its finished output handle): set by a NewAdd that ran that consumer inside its own kernel
        # (fq_block_tail_i8, Plan.fuse_next) -- the consumer hands it out instead of launching
        self.next_out = None

    def to_f32(self):
        """The fp32 NCHW tensor this handle stands for (DeQuantity + layout change, one kernel)."""
        if self.exact is None:
            raise _native.FqError("resident activation without an exact payload cannot leave the integer domain")
        return _native.dequant_nhwc_to_nchw(self.exact, self.grid, self.shape[1])

    def __repr__(self):
        return "QHandle(shape=%s, exact=%s@%s, narrow_bit=%s, relu=%s)" % (
            tuple(self.shape), None if self.exact is None else str(self.exact.dtype).replace("torch.", ""), self.grid,
            self.bit if self.narrow is not None else None, self.relu_done)


class DeferredConv(object):
    """A convolution whose only consumer is a resident NewAdd: nothing has been launched yet, the add runs
    it with the residual fused into its store phase (fq_conv2d_i8_add_resident), so the convolution's own
    int8 result never reaches HBM.  Anything else that touches it materialises the plain result."""
    __slots__ = ("layer", "xq", "wq", "geom", "_handle")

    def __init__(self, layer, xq, wq, geom):
        self.layer, self.xq, self.wq, self.geom, self._handle = layer, xq, wq, geom, None

    def materialise(self):
        if self._handle is None:
            L = self.layer
            _, q = _native.conv2d_i8_resident(self.xq, self.wq, L.quantized_bias, self.geom[0], self.geom[1], self.geom[2],
                                              L.rs_bit, L.output_bit, False, True, False)
            self._handle = QHandle((q.shape[0], L.Conv.out_channels, q.shape[1], q.shape[2]), q, L.output_bit, q, L.output_bit,
                                   False)
        return self._handle

    def to_f32(self):
        return self.materialise().to_f32()


def block_tail_enabled():
    """FQ_BLOCK_TAIL=0: keep conv3 + add and the next conv1 as two launches (A/B timing)."""
    import os
    return os.environ.get("FQ_BLOCK_TAIL", "1") != "0"


def block_tail_proj_enabled():
    """FQ_BLOCK_TAIL_PROJ=0: a stage's first block keeps its projection shortcut as a launch of its own (A/B timing)."""
    import os
    return os.environ.get("FQ_BLOCK_TAIL_PROJ", "1") != "0"


def carry(t, handle):
    """Attach the integer form `handle` to the fp32 tensor t that stands for the same values (a producer that serves both kinds of
    consumers).  The attachment holds for the tensor AS WRITTEN BY ITS PRODUCER: its version counter is remembered, and a tensor
    that was written to since -- an in-place nn.ReLU between two consumers is legal PyTorch -- no longer has an integer form
    (resident_of).  Found by scripts/recon_fuzz.py: the consumer behind such a ReLU read the integers from before it."""
    t._fq_resident = handle
    t._fq_resident_version = t._version
    return t


def resident_of(x):
    """The integer form of an activation, if it has one (a handle, or an fp32 tensor carrying one that is still current)."""
    if type(x) is QHandle:
        return x
    if type(x) is DeferredConv:
        return x.materialise()
    h = getattr(x, "_fq_resident", None)
    if h is not None and getattr(x, "_fq_resident_version", None) != x._version:
        return None                                          # written to since its producer attached the integers
    return h


def as_f32(x):
    return x.to_f32() if type(x) in (QHandle, DeferredConv) else x


class Plan(object):
    """What one producer emits.  Plain data (pickles with the module)."""
    __slots__ = ("relu", "emit_f32", "emit_int", "narrow_bit", "want_wide", "grid", "resident_add", "defer", "fuse_arg",
                 "fuse_next", "narrow_to_hbm", "fuse_proj")

    def __init__(self):
        self.relu = False            # the nn.ReLU that consumes this output is fused
        self.emit_f32 = True         # some consumer needs the fp32 NCHW tensor
        self.emit_int = False        # some consumer reads the integer form
        self.narrow_bit = None       # NewAdd: Quantity bit of the conv consumers (None: no narrow output)
        self.want_wide = False       # NewAdd: exact int16 sum needed (next add, or fp32 via dequant)
        self.grid = None             # NewAdd: grid of the exact sum
        self.resident_add = False    # NewAdd: operands arrive as integers
        self.defer = False           # NewConv2d: only consumer is a resident NewAdd, which runs this conv itself
        self.fuse_arg = None         # NewAdd: operand position (0 / 1) that arrives as a DeferredConv
        self.fuse_next = None        # NewAdd: the 1x1 NewConv2d consuming this sum that runs inside the add's kernel too
        self.narrow_to_hbm = True    # NewAdd with fuse_next: somebody besides that convolution reads the int8 re-quantisation
        self.fuse_proj = False       # NewAdd with fuse_arg: the OTHER operand is a deferred 1x1 projection that the kernel computes too

    def __getstate__(self):
        return {k: getattr(self, k) for k in self.__slots__}

    def __setstate__(self, state):
        self.__init__()
        for k, v in state.items():
            setattr(self, k, v)

    def __repr__(self):
        return "Plan(%s)" % ", ".join("%s=%r" % (k, getattr(self, k)) for k in self.__slots__)


class _ReluPassThrough(object):
    """Instance-level forward of an nn.ReLU whose producer already applied it."""

    def __init__(self, module):
        self.module = module

    def __call__(self, x):
        if type(x) is QHandle:
            if not x.relu_done:
                raise _native.FqError("resident activation reached a ReLU that its producer did not fuse "
                                      "(dataflow changed since resident.enable(); call it again)")
            return x
        if getattr(x, "_fq_relu_done", False):
            return x
        return type(self.module).forward(self.module, x)


def _pair(v):
    return (int(v[0]), int(v[1])) if isinstance(v, (tuple, list)) else (int(v), int(v))


def _maxpool_supported(m):
    k, st, pd, dl = _pair(m.kernel_size), _pair(m.stride if m.stride is not None else m.kernel_size), _pair(m.padding), \
        _pair(m.dilation)
    return (dl == (1, 1) and not m.ceil_mode and not m.return_indices and 2 * pd[0] <= k[0] and 2 * pd[1] <= k[1]
            and min(st) >= 1)


def _avgpool_is_global(m, h, w):
    k = _pair(m.kernel_size)
    return (k == (int(h), int(w)) and _pair(m.padding) == (0, 0) and not m.ceil_mode
            and getattr(m, "divisor_override", None) is None)


class _MaxPoolResident(object):
    """Instance-level forward of an nn.MaxPool2d between integer layers: pooling the int8 NHWC integers is
    pooling the values (max commutes with the monotone scale q -> q * 2^-g)."""

    def __init__(self, module):
        self.module = module

    def __call__(self, x):
        m = self.module
        plan = m.__dict__.get("_resident")
        h = resident_of(x)
        if plan is None or h is None or h.exact is None or h.exact.dtype != torch.int8 or not _maxpool_supported(m):
            return type(m).forward(m, as_f32(x))
        k, st, pd = _pair(m.kernel_size), _pair(m.stride if m.stride is not None else m.kernel_size), _pair(m.padding)
        y = _native.maxpool_i8_nhwc(h.exact, k, st, pd)
        narrow = y if (h.narrow is h.exact or h.bit == h.grid) else None
        out = QHandle((y.shape[0], h.shape[1], y.shape[1], y.shape[2]), y, h.grid, narrow, h.grid, h.relu_done)
        if not plan.emit_f32:
            return out
        t = out.to_f32()
        if plan.emit_int:
            carry(t, out)
        return t


class _AvgPoolResident(object):
    """Instance-level forward of an nn.AvgPool2d that covers the whole plane of a resident activation."""

    def __init__(self, module):
        self.module = module

    def __call__(self, x):
        m = self.module
        h = resident_of(x)
        if (h is not None and h.exact is not None and _avgpool_is_global(m, h.exact.shape[1], h.exact.shape[2])
                and h.exact.shape[1] * h.exact.shape[2] * 32768 < (1 << 24)):
            return _native.avgpool_global_nhwc(h.exact, h.grid, h.shape[1])
        return type(m).forward(m, as_f32(x))


# ---- tracing -------------------------------------------------------------------------------------

class _Value(object):
    __slots__ = ("producer", "kind", "src", "consumers", "foreign", "order")

    def __init__(self, producer, kind, src, order):
        self.producer, self.kind, self.src, self.order = producer, kind, src, order
        self.consumers = []          # (module, argument position)
        self.foreign = False         # touched by code outside NewConv2d / NewLinear / NewAdd / nn.ReLU


def _iter_tensors(obj):
    if isinstance(obj, torch.Tensor):
        yield obj
    elif isinstance(obj, (list, tuple)):
        for o in obj:
            for t in _iter_tensors(o):
                yield t
    elif isinstance(obj, dict):
        for o in obj.values():
            for t in _iter_tensors(o):
                yield t


class _Tracer(TorchFunctionMode):

    def __init__(self, planned_types):
        super(_Tracer, self).__init__()
        self.planned_types = planned_types
        self.values = {}             # id(tensor) -> _Value
        self.keep = []               # keeps traced tensors alive so ids are not reused
        self.depth = 0
        self.calls = {}              # module -> number of forward calls
        self.order = 0
        self.produced = []           # _Value of every NewConv2d / NewAdd output, in execution order
        self.relu_values = []        # _Value of every nn.ReLU output whose input is traced
        self.avgpool_shapes = {}     # nn.AvgPool2d module -> shape of its (traced) input

    def __torch_function__(self, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if self.depth == 0:
            for t in _iter_tensors(args):
                v = self.values.get(id(t))
                if v is not None:
                    v.foreign = True
            for t in _iter_tensors(kwargs):
                v = self.values.get(id(t))
                if v is not None:
                    v.foreign = True
        return func(*args, **kwargs)

    # module hooks
    def pre(self, module, args):
        self.depth += 1
        self.calls[module] = self.calls.get(module, 0) + 1
        for pos, a in enumerate(args):
            v = self.values.get(id(a)) if isinstance(a, torch.Tensor) else None
            if v is not None:
                v.consumers.append((module, pos))
                if isinstance(module, nn.AvgPool2d) and pos == 0:
                    self.avgpool_shapes[module] = tuple(a.shape)       # (depth > 0 here: not a foreign touch)

    def post(self, module, args, output):
        self.depth -= 1
        if not isinstance(output, torch.Tensor):
            return
        self.order += 1
        if isinstance(module, nn.ReLU):
            src = self.values.get(id(args[0])) if args and isinstance(args[0], torch.Tensor) else None
            if src is None:
                return
            v = _Value(module, "relu", src, self.order)
            self.relu_values.append(v)
        elif isinstance(module, nn.MaxPool2d):
            src = self.values.get(id(args[0])) if args and isinstance(args[0], torch.Tensor) else None
            if src is None:
                return
            v = _Value(module, "maxpool", src, self.order)
            self.produced.append(v)
        elif isinstance(module, nn.AvgPool2d):
            return                                          # its output is an ordinary fp32 tensor
        else:
            v = _Value(module, "add" if type(module).__name__ == "NewAdd" else "contraction", None, self.order)
            self.produced.append(v)
        self.values[id(output)] = v
        self.keep.append(output)

    def mark_foreign(self, obj):
        for t in _iter_tensors(obj):
            v = self.values.get(id(t))
            if v is not None:
                v.foreign = True


def _clear(model):
    for m in model.modules():
        m.__dict__.pop("_resident", None)
        fwd = m.__dict__.get("forward")
        if isinstance(fwd, (_ReluPassThrough, _MaxPoolResident, _AvgPoolResident)):
            del m.__dict__["forward"]
    model.__dict__.pop("_fq_resident_enabled", None)


def disable(model):
    """Back to the reference's fp32 module boundaries."""
    _clear(model)
    return model


def is_enabled(model):
    return bool(model.__dict__.get("_fq_resident_enabled"))


def enable(model, example_input, verify=True):
    """Trace one forward of `model` (an integer-simulation model built by Reconstruction.ReconModel,
    on the GPU) and switch every eligible NewConv2d / NewAdd (and the nn.ReLU / nn.MaxPool2d / global
    nn.AvgPool2d between them) to resident integer activations.  Returns a summary dict.
    `example_input` is any valid input batch; the plan does not depend on its size.  With `verify`
    (default) the planned model is run once on `example_input` and must reproduce the traced forward
    bit for bit, otherwise the plan is removed and FqError raised."""
    from .new_quantity_op import NewConv2d, NewLinear, NewAdd, QUANTIZE_BIT
    _clear(model)
    if QUANTIZE_BIT != 8:
        raise _native.FqError("resident activations are defined for QUANTIZE_BIT = 8")
    planned_types = (NewConv2d, NewLinear, NewAdd, nn.ReLU, nn.MaxPool2d, nn.AvgPool2d)
    tracer = _Tracer(planned_types)
    hooks = []
    for m in model.modules():
        if isinstance(m, planned_types):
            hooks.append(m.register_forward_pre_hook(tracer.pre))
            hooks.append(m.register_forward_hook(tracer.post))
    was_training = model.training
    model.eval()
    try:
        with torch.no_grad():
            with tracer:
                traced_out = model(example_input)
            tracer.mark_foreign(traced_out)
    finally:
        for h in hooks:
            h.remove()
        model.train(was_training)

    relu_value = dict((id(c.src), c) for c in tracer.relu_values)

    def effective(v):
        """(value the consumers see, fused ReLU module or None)"""
        if v.kind != "maxpool" and not v.foreign and len(v.consumers) == 1 and isinstance(v.consumers[0][0], nn.ReLU):
            after = relu_value.get(id(v))
            if after is not None:
                return after, v.consumers[0][0]
        return v, None

    def conv_can_emit(m):
        return isinstance(m, NewConv2d) and tracer.calls.get(m, 0) == 1 and m._int8_ok(m.Conv)

    def conv_can_read(m):
        return isinstance(m, NewConv2d) and m._int8_ok(m.Conv) and not m._stem_fold(m.Conv)

    # pass 1 (execution order): integer format of every produced value
    fmt = {}                         # id(effective _Value) -> (bytes, grid)
    eff_of = {}                      # id(produced _Value) -> (effective _Value, relu module)
    operands = {}                    # NewAdd module -> [value at arg 0, value at arg 1]
    for v in tracer.values.values():
        for (m, pos) in v.consumers:
            if isinstance(m, NewAdd) and pos < 2:
                operands.setdefault(m, [None, None])[pos] = v
    add_resident, pool_resident = set(), set()

    def avg_can_read(m):
        shape = tracer.avgpool_shapes.get(m)
        return (isinstance(m, nn.AvgPool2d) and shape is not None and len(shape) == 4
                and _avgpool_is_global(m, shape[2], shape[3]) and shape[2] * shape[3] * 32768 < (1 << 24))

    for v in tracer.produced:
        e, relu_mod = effective(v)
        eff_of[id(v)] = (e, relu_mod)
        m = v.producer
        if v.kind == "contraction":
            if conv_can_emit(m):
                fmt[id(e)] = (1, m.output_bit)
        elif v.kind == "maxpool":
            f = fmt.get(id(v.src))
            if f is not None and f[0] == 1 and tracer.calls.get(m, 0) == 1 and _maxpool_supported(m):
                pool_resident.add(m)
                fmt[id(e)] = f
        else:
            ops = operands.get(m)
            if tracer.calls.get(m, 0) != 1 or ops is None or ops[0] is None or ops[1] is None:
                continue
            fx, fy = fmt.get(id(ops[0])), fmt.get(id(ops[1]))
            if fx is None or fy is None:
                continue
            g = max(0, fx[1], fy[1])
            if g > MAX_WIDE_GRID or min(fx[1], fy[1]) < -16:
                continue
            add_resident.add(m)
            fmt[id(e)] = (2, g)

    # pass 2: what every producer has to emit
    summary = {"resident_convs": 0, "resident_adds": 0, "resident_pools": 0, "fused_relus": 0, "fp32_outputs": 0,
               "int_only_outputs": 0}
    for v in tracer.produced:
        m = v.producer
        e, relu_mod = eff_of[id(v)]
        if id(e) not in fmt:
            continue                                        # plain fp32 producer
        plan = Plan()
        plan.relu = relu_mod is not None
        need_f32 = e.foreign
        int_consumers = 0
        narrow_bits = []
        for (c, pos) in e.consumers:
            if conv_can_read(c):
                narrow_bits.append(c.input_bit)
            elif (isinstance(c, NewAdd) and c in add_resident) or avg_can_read(c):
                int_consumers += 1
                plan.want_wide = True                       # these read the exact value
            elif c in pool_resident and v.kind != "add":
                int_consumers += 1                          # int8 max-pool of an int8 activation
            else:
                need_f32 = True
        if v.kind in ("contraction", "maxpool"):
            grid = m.output_bit if v.kind == "contraction" else fmt[id(e)][1]
            ok = [b for b in narrow_bits if b == grid]
            if len(ok) != len(narrow_bits):
                need_f32 = True                             # a consumer quantises at another bit: from fp32
            int_consumers += len(ok)
            plan.narrow_bit = grid
        else:
            plan.resident_add = True
            plan.grid = fmt[id(e)][1]
            if narrow_bits:
                plan.narrow_bit = narrow_bits[0]
                ok = [b for b in narrow_bits if b == plan.narrow_bit]
                if len(ok) != len(narrow_bits):
                    need_f32 = True
                int_consumers += len(ok)
            if need_f32:
                plan.want_wide = True                       # fp32 leaves through the exact int16 sum
        plan.emit_int = int_consumers > 0
        plan.emit_f32 = need_f32 or not plan.emit_int
        m.__dict__["_resident"] = plan
        if plan.relu:
            relu_mod.__dict__["forward"] = _ReluPassThrough(relu_mod)
            summary["fused_relus"] += 1
        if v.kind == "maxpool":
            m.__dict__["forward"] = _MaxPoolResident(m)
        summary[{"contraction": "resident_convs", "add": "resident_adds", "maxpool": "resident_pools"}[v.kind]] += 1
        summary["fp32_outputs" if plan.emit_f32 else "int_only_outputs"] += 1
    # a convolution whose value goes to one resident add and nowhere else is run BY that add
    summary["fused_conv_adds"] = 0
    for add_mod in add_resident:
        ops = operands[add_mod]
        for pos in (0, 1):
            v = ops[pos]
            conv = v.producer
            plan = conv.__dict__.get("_resident") if isinstance(conv, NewConv2d) else None
            if (v.kind == "contraction" and plan is not None and not plan.relu and not plan.emit_f32 and not v.foreign
                    and v.consumers == [(add_mod, pos)] and ops[1 - pos] is not v
                    and not add_mod.__dict__["_resident"].emit_f32):
                plan.defer = True
                add_mod.__dict__["_resident"].fuse_arg = pos
                summary["fused_conv_adds"] += 1
                break
    # ... and when the re-quantised sum of such an add feeds a 1x1 convolution (the next bottleneck's conv1), that convolution
    # runs inside the same kernel (fq_block_tail_i8): its operand is staged in LDS and, if nobody else reads it, never written
    summary["fused_block_tails"] = 0
    for v in tracer.produced:
        add_mod = v.producer
        plan = add_mod.__dict__.get("_resident")
        if v.kind != "add" or plan is None or plan.fuse_arg is None or plan.emit_f32 or not plan.want_wide or not block_tail_enabled():
            continue
        e, _relu = eff_of[id(v)]
        conv3 = operands[add_mod][plan.fuse_arg].producer
        other = fmt.get(id(operands[add_mod][1 - plan.fuse_arg]))            # (bytes, grid) of the shortcut
        if other is None:
            continue
        readers = [c for (c, _pos) in e.consumers if conv_can_read(c)]
        nxt = None
        for c in readers:
            cp = c.__dict__.get("_resident")
            k = c.Conv
            if (cp is not None and not cp.defer and cp.emit_int and not cp.emit_f32 and tracer.calls.get(c, 0) == 1
                    and tuple(k.kernel_size) == (1, 1) and tuple(k.stride) == (1, 1) and tuple(k.padding) == (0, 0)
                    and tuple(conv3.Conv.kernel_size) == (1, 1) and tuple(conv3.Conv.stride) == (1, 1)
                    and tuple(conv3.Conv.padding) == (0, 0) and c.input_bit == plan.narrow_bit
                    and conv3.Conv.out_channels == k.in_channels
                    and _native.block_tail_supported(conv3.Conv.in_channels, conv3.Conv.out_channels, k.out_channels,
                                                     conv3.rs_bit, c.rs_bit, conv3.output_bit, other[1], other[0],
                                                     plan.narrow_bit)):
                nxt = c
                break
        if nxt is not None:
            plan.fuse_next = nxt
            plan.narrow_to_hbm = len(readers) > 1
            summary["fused_block_tails"] += 1
    # ... and when the OTHER operand of such an add is a 1x1 projection of the block's input that nobody else reads (the first
    # block of a stage), the kernel computes that convolution as well (fq_block_tail_proj_i8): its K3 bytes per pixel are neither
    # written nor read back
    summary["fused_projections"] = 0
    for add_mod in add_resident:
        plan = add_mod.__dict__.get("_resident")
        if plan is None or plan.fuse_arg is None or plan.emit_f32 or not block_tail_enabled() or not block_tail_proj_enabled():
            continue
        ops = operands[add_mod]
        v3, vp = ops[plan.fuse_arg], ops[1 - plan.fuse_arg]
        conv3, proj = v3.producer, vp.producer
        pp = proj.__dict__.get("_resident") if isinstance(proj, NewConv2d) else None
        if (vp.kind != "contraction" or pp is None or pp.relu or pp.emit_f32 or pp.defer or vp.foreign
                or vp.consumers != [(add_mod, 1 - plan.fuse_arg)] or not conv_can_read(proj) or tracer.calls.get(proj, 0) != 1):
            continue
        k3, kp = conv3.Conv, proj.Conv
        nxt = plan.fuse_next
        if (tuple(kp.kernel_size) != (1, 1) or tuple(kp.padding) != (0, 0) or kp.stride[0] != kp.stride[1]
                or tuple(k3.kernel_size) != (1, 1) or tuple(k3.stride) != (1, 1) or tuple(k3.padding) != (0, 0)
                or kp.out_channels != k3.out_channels or (kp.out_channels % 16) or (kp.in_channels % 16)
                or not _native.block_tail_proj_supported(k3.in_channels, k3.out_channels, nxt.Conv.out_channels if nxt is not None else 0,
                                                         kp.in_channels, conv3.rs_bit, nxt.rs_bit if nxt is not None else 0,
                                                         proj.rs_bit, kp.stride[0])):
            continue
        pp.defer = True
        plan.fuse_proj = True
        summary["fused_projections"] += 1
    for m in tracer.avgpool_shapes:
        if avg_can_read(m):
            m.__dict__["forward"] = _AvgPoolResident(m)
            summary["resident_pools"] += 1
    model.__dict__["_fq_resident_enabled"] = True
    if verify:
        with torch.no_grad():
            planned_out = model(example_input)
        same = all(torch.equal(a, b) for a, b in zip(_iter_tensors(planned_out), _iter_tensors(traced_out)))
        if not same:
            _clear(model)
            raise _native.FqError("resident plan does not reproduce the fp32-boundary forward on the example input; "
                                  "plan removed (please report the model)")
    return summary


def describe(model):
    """{module name: Plan} of the current plan (for logs and tests)."""
    return dict((name, m.__dict__["_resident"]) for name, m in model.named_modules() if "_resident" in m.__dict__)


class GraphedForward(object):
    """One forward of a model captured as a HIP graph (torch.cuda.CUDAGraph) and replayed per call.

    The integer-simulation forward is ~100 short kernels; below batch ~64 the Python / launch path, not
    the GPU, sets its rate.  Every kernel of this library is enqueued on the caller's stream without
    synchronising, so the whole forward captures as is.  The input is copied into a static buffer and the
    returned tensor is the graph's static output: it is overwritten by the next call (clone it to keep it).
    Batch shape is fixed at capture time."""

    def __init__(self, model, example_input, warmup=2):
        from . import new_quantity_op
        self.model = model
        self.static_in = example_input.detach().clone()
        new_quantity_op._xq_cache.clear()       # nothing memoised before the capture may be served inside it
        side = torch.cuda.Stream(device=example_input.device)
        side.wait_stream(torch.cuda.current_stream(example_input.device))
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(warmup):
                model(self.static_in)
        torch.cuda.current_stream(example_input.device).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.static_out = model(self.static_in)
        new_quantity_op._xq_cache.clear()       # ... and nothing from the graph's private pool outside it

    def __call__(self, x):
        if x.shape != self.static_in.shape or x.dtype != self.static_in.dtype:
            raise _native.FqError("GraphedForward was captured for input %s %s" % (tuple(self.static_in.shape), self.static_in.dtype))
        self.static_in.copy_(x, non_blocking=True)
        self.graph.replay()
        return self.static_out


class MultiStreamGraphs(object):
    """One batch as S HIP graphs of batch / S images replayed on S streams.

    At these layer sizes a kernel of the integer-simulation forward runs for 20-60 us, of which a third does not scale
    with the reduction depth (prologue, epilogue, the half-empty last round of workgroups, drain:
    profiles/r02d_conv_loop_ablation.txt); kernels of ONE stream run strictly one after the other, so that part is exposed
    54 times per forward.  Two independent halves of the batch on two streams let one half's kernels fill the CUs the other
    half's tail leaves idle: ResNet-50, 256 images, 3.21 ms as one graph -> 2.99 ms as two graphs of 128 (79 800 -> 85 700
    images/s; 512 images: 84 200 -> 90 500).  Three or four
    streams are slower again (the kernels get too small).  Same logits: every image goes through the same kernels."""

    def __init__(self, model, example_input, streams=2, warmup=2):
        n = int(example_input.shape[0])
        if streams < 2 or n % streams:
            raise _native.FqError("MultiStreamGraphs: the batch (%d) must split evenly over %d streams" % (n, streams))
        dev = example_input.device
        self.sizes = [n // streams] * streams
        self.graphs = [GraphedForward(model, p.contiguous(), warmup) for p in example_input.chunk(streams)]
        self.streams = [torch.cuda.Stream(device=dev) for _ in range(streams)]

    def __call__(self, x):
        if int(x.shape[0]) != sum(self.sizes):
            raise _native.FqError("MultiStreamGraphs was captured for %d images" % sum(self.sizes))
        main = torch.cuda.current_stream(x.device)
        outs, off = [], 0
        for st, g, n in zip(self.streams, self.graphs, self.sizes):
            st.wait_stream(main)                              # x is ready
            with torch.cuda.stream(st):
                outs.append(g(x[off:off + n]))                # copy into the graph's static input + replay, on st
            off += n
        for st in self.streams:
            main.wait_stream(st)
        return torch.cat(outs)                                # (static outputs: overwritten by the next call)


def capture(model, example_input, warmup=2, streams=1):
    """HIP-graph capture of `model`'s forward at the shape of `example_input` (see GraphedForward); streams > 1: the batch
    split into that many graphs replayed concurrently (see MultiStreamGraphs)."""
    if streams > 1:
        return MultiStreamGraphs(model, example_input, streams, warmup)
    return GraphedForward(model, example_input, warmup)
