"""The float convolutions this package runs on its own fp32 MFMA kernels instead of torch's (no reference counterpart: the
reference calls torch's Conv2d.forward -- pytorch_quantizer.py:288-296 inside the calibration forward, new_quantity_op.py:283-292
inside TestConv): 1x1 layers on fq_conv1x1_f32, R x S layers with zero padding (the 3x3 ones) on fq_conv_kxk_f32, the 7x7
stride-2 stem on fq_conv_stem_f32 (csrc/) -- every convolution of a ResNet, so that the calibration forward is deterministic
and never enters the convolution library (whose first-use solver search costs seconds in a fresh process).  Which call qualifies, the
weights in the kernels' layout (cached on the module), the once-per-process check of every module against an independent
implementation of the same fp32 mathematics, and the plain (no statistic) forward.  Shared by tools.Quantity (which adds the
statistic epilogues) and TestConv.

Nothing of this is stored ON the module: the reference pickles whole models (reconstruction.py:107-140, torch.save(self.model)),
and a packed CUDA copy of the weights or a flag in m.__dict__ would travel into that file (and survive model.cpu()).  The
per-module state lives in a WeakKeyDictionary of this file instead and dies with the module."""
import os
import weakref

import torch

from . import _native

__all__ = ["enabled", "kind", "weight", "runner", "verified", "plain", "call", "call_qd", "call_linear_qd", "state", "is_verified",
           "is_off", "forget", "kernel_key", "own_convs", "TOL"]

TOL = 1e-5                                      # |own - torch| <= TOL * (|W| * |x| + |b|): summation order only
# module -> {"verified": the set of kernels (kernel_key) that agreed with torch on this module, "off": one disagreed (the module
#            keeps torch's convolution), "wt": {kind: (tag of the parameter, weights in the kernel's layout)}}
_STATE = weakref.WeakKeyDictionary()


def state(m):
    st = _STATE.get(m)
    if st is None:
        st = _STATE[m] = {}
    return st


def _uses_sb(m, k):
    """Does kind k of this module run on the split-bf16 form of the 1x1 kernels?  (opt-in; from 128 input channels on: with
    64 the four K steps of a tile do not pay for the split's prologue, 0.95 x)"""
    w = m.weight
    return k == "c1" and _native.conv_sb_enabled() and w.shape[1] >= 128 and _native.conv_sb_supported(w.shape[1], w.shape[0])


def kernel_key(m, k):
    """What the once-per-process check is keyed by: a module can run on several kernels in one process (the direct or the
    Winograd kernel depending on the input's shape and FQ_CONV_WINO, the fp32 or the split-bf16 1x1 kernel depending on
    FQ_CONV_SPLIT_BF16), and each of them is checked the first time the module runs on it."""
    return (k, _uses_sb(m, k))


def is_verified(m, k=None):
    """Has the module passed its check on the kernel kind k runs on now (None: on any kernel)?"""
    done = _STATE.get(m, {}).get("verified")
    return bool(done) if k is None else bool(done) and kernel_key(m, k) in done


def is_off(m):
    return bool(_STATE.get(m, {}).get("off"))


def forget(m=None):
    """Drop the cached weights and the check result of one module (None: of every module)."""
    if m is None:
        _STATE.clear()
    else:
        _STATE.pop(m, None)


def enabled():
    return os.environ.get("FQ_OWN_CONV1X1", "1") != "0"


def kind(m, x, wino=True):
    """"c1" (fq_conv1x1_f32), "kxk" (fq_conv_kxk_f32), "wino" (fq_conv3x3_wino_f32: the Winograd form of the stride-1 3x3
    layers; wino=False or FQ_CONV_WINO=0 keeps them on "kxk"), "stem" (fq_conv_stem_f32) or None: which own kernel takes this call
    of the nn.Conv2d m.  Every kind has the plain, the statistic and the QuanDequan form, so TestConv's two ways through a layer
    (fused with QuanDequan, or not when somebody watches the module) see the same sums."""
    if (not torch.is_tensor(x) or not x.is_cuda or x.dtype != torch.float32 or m.weight.dtype != torch.float32 or m.bias is None
            or is_off(m) or m.groups != 1 or m.dilation != (1, 1) or m.stride[0] != m.stride[1] or x.dim() != 4
            or not x.is_contiguous() or isinstance(m.padding, str) or m.padding[0] != m.padding[1] or m.padding_mode != "zeros"
            or x.numel() >= 2 ** 30 or x.shape[0] * m.out_channels * x.shape[2] * x.shape[3] >= 2 ** 30):
        return None
    if m.kernel_size == (1, 1) and m.padding == (0, 0) and m.out_channels % 4 == 0:
        return "c1"
    if (x.shape[2] + 2 * m.padding[0] >= m.kernel_size[0] and x.shape[3] + 2 * m.padding[1] >= m.kernel_size[1]
            and _native.conv_stem_f32_supported(m.weight, m.stride[0])):
        return "stem"
    if (m.kernel_size != (1, 1) and m.in_channels % 16 == 0 and m.out_channels % 4 == 0
            and x.shape[2] + 2 * m.padding[0] >= m.kernel_size[0] and x.shape[3] + 2 * m.padding[1] >= m.kernel_size[1]
            and m.kernel_size[0] * m.kernel_size[1] * m.in_channels * m.out_channels < 2 ** 30):
        # (the Winograd epilogue reads the bias 16 bytes at a time: a bias that is a view into a flat parameter buffer at an
        #  odd offset keeps the layer on the direct kernel, whose bias reads are scalar)
        if (wino and m.kernel_size == (3, 3) and m.stride == (1, 1) and m.padding == (1, 1) and _native.conv_wino_enabled()
                and m.bias.data_ptr() % 16 == 0
                and _native.conv_wino_supported(x.shape[0], m.in_channels, x.shape[2], x.shape[3], m.out_channels)):
            return "wino"
        return "kxk"
    return None


def weight(m, k):
    """The weights in the layout the kernel reads (Wt [Cin][Cout] / the packed stem matrix / the transformed Winograd
    weights), one entry per kind, rebuilt when the parameter was written to or replaced."""
    w = m.weight
    tag = (w._version, w.data_ptr(), w.device, _native.conv_sb_enabled())
    by_kind = state(m).setdefault("wt", {})
    cached = by_kind.get(k)
    if cached is None or cached[0] != tag:
        sb = _uses_sb(m, k)
        packed = (_native.pack_sb_weight(w) if sb                # (opt-in: the split-bf16 form of the 1x1 kernels takes this pack)
                  else w.detach().view(w.shape[0], w.shape[1]).t().contiguous() if k == "c1"
                  else _native.pack_kxk_weight(w) if k == "kxk"
                  else _native.pack_wino_weight(w) if k == "wino" else _native.pack_stem_weight(w))
        cached = (tag, packed)
        by_kind[k] = cached
    return cached[1]


def runner(m, k, x):
    """run(**epilogue) -> y: the kernel for this call (epilogue: max_dev/row, interval_dev/hist_dev/row, relu_out, out)."""
    wq, s = weight(m, k), m.stride[0]
    if k == "c1":
        return lambda **kw: _native.conv1x1_f32(x, wq, m.bias, s, **kw)
    if k == "kxk":
        return lambda **kw: _native.conv_kxk_f32(x, wq, m.bias, m.kernel_size, s, m.padding[0], **kw)
    if k == "wino":
        return lambda **kw: _native.conv_wino_f32(x, wq, m.bias, m.out_channels, **kw)
    return lambda **kw: _native.conv_stem_f32(x, wq, m.bias, m.out_channels, m.kernel_size, s, m.padding[0], **kw)


def verified(m, run, x, k=None):
    """Once per process, module and kernel (kernel_key(m, k); k = None: once per module, whatever kernel `run` is): the own
    kernel against torch on this very input.  Returns torch's result when the module fails (and marks it: it keeps the library
    convolution from now on), None when it passes."""
    if is_verified(m, k):
        return None
    # The reference result comes from an independent implementation of the same fp32 mathematics that is NOT the convolution
    # library: asking that library for a layer it will never run again would put its first-use solver search (tens of
    # milliseconds to seconds per configuration; 0.4 s for the 7x7 stem alone, scripts/_dbg/one_shot_probe.py) into a one-shot
    # calibration for nothing.
    # (compared on the first and the last images -- every tile shape and image boundary occurs there, and the tiles of the last,
    #  K-sliced round of a launch are the last ones; the abs-max is checked on the whole output.  The two groups are VIEWS of x and
    #  of the kernel's output, compared one after the other: the gathered copies this used to make -- up to 100 MB each -- and the
    #  full-size temporary of `own.abs().max()` were allocator growth in a fresh process, most of the 0.16 s these checks cost a
    #  one-shot calibration of ResNet-50; round 5)
    n_all = int(x.shape[0])
    if m.kernel_size == (1, 1):                                 # 1x1: a GEMM (rocBLAS through torch.matmul), 16 + 16 images
        per = 16
    else:                                                       # R x S (the stem included): im2col (unfold) + GEMM, 8 + 8 images
        per = max(1, min(8, (1 << 27) // max(x[0].numel() * m.kernel_size[0] * m.kernel_size[1], 1)))
    groups = [(0, n_all)] if n_all <= 2 * per else [(0, per), (n_all - per, n_all)]
    w2 = m.weight.view(m.out_channels, -1)
    scratch = torch.zeros(1, dtype=torch.float32, device=x.device)
    own = run(max_dev=scratch, row=0)
    # (vector_norm(inf) = max |.| in one pass, no temporary; NaN propagates as in abs().max())
    ok = scratch[0] == torch.linalg.vector_norm(own.reshape(-1), float("inf"))
    for lo, hi in groups:
        xh = x[lo:hi]
        if m.kernel_size == (1, 1):
            s = m.stride[0]
            cols = (xh if s == 1 else xh[:, :, ::s, ::s]).reshape(hi - lo, x.shape[1], -1)
        else:
            cols = torch.nn.functional.unfold(xh, m.kernel_size, padding=m.padding, stride=m.stride)
        ref = torch.matmul(w2, cols) + m.bias.view(1, -1, 1)
        bound = torch.matmul(w2.abs(), cols.abs()) + m.bias.abs().view(1, -1, 1)
        del cols
        cmp = own[lo:hi].reshape(hi - lo, m.out_channels, -1)
        ok = ok & ((cmp - ref).abs() <= TOL * bound).all()
        del ref, bound
    # (one device-side verdict, one host synchronisation: the first forward of a process checks 53 modules)
    if not bool(ok):
        state(m)["off"] = True
        return torch.nn.Conv2d.forward(m, x)
    state(m).setdefault("verified", set()).add(kernel_key(m, k))
    return None


def plain(m, k, x, check=True):
    """The convolution alone on the own kernel (bias in its epilogue).  check=False: without the once-per-module check (for
    a forward whose values nobody uses)."""
    run = runner(m, k, x)
    if not check:
        return run()
    ref = verified(m, run, x, k)
    return ref if ref is not None else run()


def call(m, x):
    """m(x) for an nn.Conv2d with the own kernel as its forward when the call qualifies -- through Module.__call__, so
    forward hooks on m still fire; nothing stays attached to the module (whole models are pickled)."""
    k = kind(m, x) if (enabled() and not torch.is_grad_enabled() and "forward" not in m.__dict__) else None
    if k is None:
        return m(x)
    m.forward = lambda inp: plain(m, k, inp)
    try:
        return m(x)
    finally:
        del m.forward


class own_convs(object):
    """Context manager: inside it every nn.Conv2d of `model` that the own kernels take runs on them WITHOUT the once-per-module
    check -- for forwards whose values nobody uses (Quantity's build_net_structure trace: only which tensor OBJECT reaches which
    module matters there).  A fresh process then does not enter the convolution library for the trace either: its first-use solver
    search was 0.27 s of a one-shot script's 0.29 s in Quantity(model) (scripts/_dbg/ctor_probe.py).  Forward hooks still fire
    (the patched forward is an instance attribute, Module.__call__ runs it); nothing stays on the modules."""

    def __init__(self, model):
        self.model, self.patched = model, []

    def __enter__(self):
        if not (enabled() and torch.cuda.is_available()):
            return self
        for m in self.model.modules():
            if type(m) is not torch.nn.Conv2d or m.bias is None or "forward" in m.__dict__:
                continue

            def forward(x, m=m):
                k = kind(m, x) if not torch.is_grad_enabled() else None
                return plain(m, k, x, check=False) if k is not None else torch.nn.Conv2d.forward(m, x)
            m.forward = forward
            self.patched.append(m)
        return self

    def __exit__(self, *exc):
        for m in self.patched:
            m.__dict__.pop("forward", None)
        self.patched = []
        return False


def _hooked(m):
    """Does anybody watch this module's input or output?  (Forward hooks of the module, or torch's global module hooks.)"""
    import torch.nn.modules.module as _mod
    return bool(m._forward_hooks or m._forward_pre_hooks or getattr(m, "_forward_hooks_with_kwargs", None)
                or _mod._global_forward_hooks or _mod._global_forward_pre_hooks)


def call_qd(m, x, bit, bitwidth):
    """QuanDequan(m(x), bit) for an nn.Conv2d in ONE kernel (fq_conv1x1_qd_f32 / fq_conv_kxk_qd_f32 / fq_conv_stem_qd_f32: the
    fake-quantisation applied where the value leaves the accumulator, reference new_quantity_op.py:283-292), or None when
    the call does not qualify and the caller runs the two passes.  It qualifies when the own kernel takes the layer
    (kind()), the module has passed its once-per-process check against torch, and nobody observes the un-quantised
    convolution output -- a forward hook on m must see what the reference's hook would see, so hooked modules keep the
    two-pass form.  The result equals fq_quandequan_f32 of the own kernel's plain output bit for bit (same sum, same map)."""
    if not (enabled() and not torch.is_grad_enabled() and "forward" not in m.__dict__) or _hooked(m):
        return None
    k = kind(m, x)
    if k is None:
        return None
    run = runner(m, k, x)
    if not is_verified(m, k) and verified(m, run, x, k) is not None:
        return None                                             # the module just failed its check: two passes, torch's convolution
    return run(qd=(int(bit), 8 if bitwidth == 8 else 16))


def call_linear_qd(m, x, bit, bitwidth):
    """QuanDequan(m(x), bit) for an nn.Linear in ONE kernel (reference new_quantity_op.py:248-256 = TestLinear.forward: the
    linear layer, then output_qdp): a linear layer over a batch is the 1x1 convolution of a 1 x 1 plane, so it runs on
    fq_conv1x1_qd_f32 with x seen as [N, in_features, 1, 1] and the weight matrix transposed once per module (cached like the
    convolutions' weights).  None when the call does not qualify (the caller then runs the two passes): not a 2-D contiguous
    fp32 CUDA batch, no bias, out_features % 4, somebody hooks the nn.Linear (a hook must see the un-quantised output), or
    the module failed its once-per-process check against torch.addmm (same bound as the convolutions: summation order only)."""
    if (not enabled() or torch.is_grad_enabled() or "forward" in m.__dict__ or _hooked(m) or is_off(m) or type(m) is not torch.nn.Linear
            or not torch.is_tensor(x) or not x.is_cuda or x.dtype != torch.float32 or x.dim() != 2 or not x.is_contiguous()
            or m.bias is None or m.weight.dtype != torch.float32 or m.out_features % 4 or x.shape[0] == 0
            or x.numel() >= 2 ** 30 or x.shape[0] * m.out_features >= 2 ** 30 or m.in_features * m.out_features >= 2 ** 30
            or x.shape[1] != m.in_features):
        return None
    w = m.weight
    tag = (w._version, w.data_ptr(), w.device)
    cached = state(m).get("wt_linear")
    if cached is None or cached[0] != tag:
        cached = state(m)["wt_linear"] = (tag, w.detach().t().contiguous())
    x4 = x.view(x.shape[0], x.shape[1], 1, 1)
    if not is_verified(m, "linear"):
        try:
            own = _native.conv1x1_f32(x4, cached[1], m.bias, 1).view(x.shape[0], -1)
        except _native.FqError:                                 # a limit of the kernel this guard does not know: the two passes
            state(m)["off"] = True
            return None
        ref = torch.addmm(m.bias, x, w.t())
        bound = torch.addmm(m.bias.abs(), x.abs(), w.abs().t())
        if not bool(((own - ref).abs() <= TOL * bound).all()):
            state(m)["off"] = True
            return None
        state(m).setdefault("verified", set()).add(kernel_key(m, "linear"))
    return _native.conv1x1_f32(x4, cached[1], m.bias, 1, qd=(int(bit), 8 if bitwidth == 8 else 16)).view(x.shape[0], -1)
