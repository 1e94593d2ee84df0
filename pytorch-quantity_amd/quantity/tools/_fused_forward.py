"""The fused calibration forward of tools.Quantity: what the patched module forwards and the forward hooks do so that a hooked
tensor's statistic is taken by the kernel that PRODUCES it (no reference counterpart: the reference's hook copies every output to
the host, pytorch_quantizer.py:509-513, and its model runs torch's own kernels, :288-296).  Split out of pytorch_quantizer.py in
round 4; `Quantity` inherits this mixin, the shared per-calibration state lives in `_hook_state._HookState` (ctl below).

One calibration forward is a sequence of EVENTS -- a patched module forward is entered, the calibration's forward hook of the
same call fires -- and each fusion is a small state machine over them.  States live in ctl; the table is the contract:

| fusion (switch) | state (ctl field) | event | action (method) |
|---|---|---|---|
| own convolution (own_conv1x1) | fuse_bias = (m, (kind, x)) set by the patched Conv2d.forward, which returned an EMPTY tensor | hook of that call | run the own kernel into the tensor with the pass's statistic in its epilogue (`_finish_own_conv`); first use: checked against torch (`_float_conv.verified`) |
| library convolution + bias producer (fuse_bias_absmax) | fuse_bias = (m, x) with the bias-free convolution's result | hook of that call | `fq_bias_add_absmax_f32` / `_hist_f32` in place (`_finish_fused_conv`) |
| conv -> ReLU (fuse_relu) | relu_after[m] = the nn.ReLU that consumed m's output last forward; relu_ready = (output, r, relu, version) | the patched ReLU.forward receives exactly that tensor, unmodified | hand out r, launch nothing (`_run_with_relu` prepared it) |
| conv whose output only its ReLU reads (skip_unread_outputs) | relu_only_ok (proven by the poison probe) | as above, and pass 2 does not want the tensor | the kernel writes the ReLU's result only (out=False) |
| conv3 + Eltwise + ReLU (fuse_conv_add) | defer_ok[conv] = its Eltwise (proven); deferred[id(output)] = (output, conv, x, key, row, version, x_version) | hook of the Eltwise whose operand is that very tensor | ONE launch `fq_conv1x1_add_f32` / `_add_hist_f32` (`_finish_deferred`); anything unexpected -- other operand shape, a version counter that moved, small planes with both tensors kept -- runs the convolution alone first (`_run_deferred`) |
| | deferred not empty | the forward ends | RuntimeError: the model left the path the probe saw (`_forward_with_stats`) |
| the proofs | poison = _DeferralProbe, mode learn / poison | two probe forwards before the first batch | `_probe_forward`, `_prove_deferral` (NaN poisoning + the keeper scan, `_hook_state._DeferralProbe`) |

Every fusion falls back to the unfused form of the same arithmetic; none changes a table (tests/test_gpu_conv_add_fusion.py,
tests/test_gpu_float_forward_kernels.py, tests/test_gpu_r50_tables.py)."""
import weakref

import torch

from common.quantity import _native, _float_conv
from ._hook_state import _AFTER_FORWARD, _DeferralProbe, _EagerStats  # noqa: F401

__all__ = ["_FusedForward"]


class _StopForward(Exception):
    """Raised by the feature hook to end a forward pass early (pass 2 only needs a prefix of the net
    when the deeper activations were kept from pass 1)."""


# Once-per-process check results, per module.  Kept in _float_conv's WeakKeyDictionary, never on the module: the reference
# pickles whole models (reconstruction.py:107-140) and nothing of this package may travel into that file.
_RELU_VERIFIED = "relu_fusion_verified"
_POOL_VERIFIED = "pool_verified"                  # the own pooling kernel gave torch's bits here
_POOL_OFF = "pool_off"
_FUSION_VERIFIED = "bias_fusion_verified"         # conv-without-bias + fq_bias_add_absmax_f32 == its forward


def _flag(m, name):
    return bool(_float_conv.state(m).get(name))


def _set_flag(m, name):
    _float_conv.state(m)[name] = True


class _FusedForward(object):
    """Mixin of tools.Quantity (which provides model, net_info, _hook_ctl, the switches fuse_* / own_* / skip_unread_outputs /
    materialize_all, _model_device, input_size, _stat_stream)."""

    def _patch_fused_convs(self, model):
        """Instance-level forwards for the duration of a GPU calibration (undo: del module.forward; returns the patched modules):
        every hooked nn.Conv2d with a bias leaves its work to its forward hook -- the whole convolution with the statistic in
        the epilogue where common.quantity._float_conv takes the layer (1x1, R x S with zero padding, the 7x7/2 stem), else
        convolution-without-bias here and fq_bias_add_absmax_f32 / fq_bias_add_hist_f32 in the hook; `Eltwise` likewise
        (fq_add_absmax_f32 / fq_add_hist_f32); nn.MaxPool2d / a global nn.AvgPool2d run on fq_maxpool2d_f32 /
        fq_avgpool_global_f32; an out-of-place nn.ReLU fed by one of the producers hands out the copy that producer wrote."""
        patched = []
        if not self.fuse_bias_absmax or "Conv2d" not in self._all_op_type or "Conv2d" not in self._cared_op_type:
            return patched
        ctl = self._hook_ctl
        for m in model.modules():
            if type(m) is not torch.nn.Conv2d or m.bias is None or m.padding_mode != "zeros" or "forward" in m.__dict__:
                continue

            def forward(x, m=m):
                if (not torch.is_tensor(x) or not x.is_cuda or x.dtype != torch.float32 or m.weight.dtype != torch.float32
                        or torch.is_grad_enabled()):
                    return torch.nn.Conv2d.forward(m, x)
                if ctl.own_plain:                    # per-channel calibration: the convolution only, statistics by its hooks
                    own = _float_conv.kind(m, x) if self.own_conv1x1 else None
                    if own is None:
                        return torch.nn.Conv2d.forward(m, x)
                    y = _float_conv.plain(m, own, x, check=ctl.own_plain != "unchecked")
                    if ctl.poison is not None:
                        y = ctl.poison.conv_done(m, y)
                    return y
                if ctl.fuse_collector is None or ctl.fuse_off:
                    return torch.nn.Conv2d.forward(m, x)
                if ctl.fuse_stat == "hist" and m not in ctl.fuse_verified:
                    return torch.nn.Conv2d.forward(m, x)    # pass 2 fuses verified modules only
                own = _float_conv.kind(m, x) if self.own_conv1x1 else None
                if own is None and m not in ctl.fuse_warm and not _flag(m, _FUSION_VERIFIED):
                    ctl.fuse_warm.add(m)                 # the first call of a shape may run a one-off MIOpen kernel:
                    return torch.nn.Conv2d.forward(m, x)    # plain forward now, verification on the next batch
                if own is not None:
                    ctl.fuse_bias = (m, (own, x))        # the hook of this very call runs the whole convolution
                    s, p, k = m.stride[0], m.padding[0], m.kernel_size
                    return torch.empty((x.shape[0], m.out_channels, (x.shape[2] + 2 * p - k[0]) // s + 1,
                                        (x.shape[3] + 2 * p - k[1]) // s + 1), dtype=torch.float32, device=x.device)
                y = m._conv_forward(x, m.weight, None)
                ctl.fuse_bias = (m, x)                   # the hook of this very call adds the bias
                return y
            m.forward = forward
            patched.append(m)
        if "Eltwise" in self._all_op_type and "Eltwise" in self._cared_op_type:
            from common.quantity.fabu_layer import Eltwise
            for m in model.modules():
                if type(m) is not Eltwise or "forward" in m.__dict__:
                    continue

                def forward(x, y, m=m):
                    if ctl.poison is not None:
                        z = ctl.poison.eltwise(m, x, y)
                        if z is not None:
                            return z
                    # (per-channel calibration, own_plain: the sum and the ReLU behind it in one launch, no statistic)
                    if ((ctl.fuse_collector is None and not ctl.own_plain) or ctl.fuse_off or torch.is_grad_enabled()
                            or not torch.is_tensor(x)
                            or not torch.is_tensor(y) or not x.is_cuda or x.dtype != torch.float32 or y.dtype != torch.float32
                            or x.shape != y.shape or not x.is_contiguous() or not y.is_contiguous() or y.device != x.device
                            or (not ctl.own_plain and ctl.fuse_stat == "hist" and m not in ctl.fuse_verified)):
                        for t in (x, y):                        # (an operand whose convolution was left for this call)
                            d = ctl.deferred.get(id(t)) if ctl.deferred else None
                            if d is not None and d[0] is t:
                                del ctl.deferred[id(t)]
                                self._run_deferred(d)
                        return Eltwise.forward(m, x, y)
                    ctl.fuse_bias = (m, (x, y))          # the hook of this very call computes the sum (+ its abs-max)
                    return torch.empty_like(x)
                m.forward = forward
                patched.append(m)
        if self.own_pools:
            def pair(v):
                return (int(v), int(v)) if isinstance(v, int) else (int(v[0]), int(v[1]))

            def pool_active(x):
                return ((ctl.own_plain or (ctl.fuse_collector is not None and not ctl.fuse_off)) and torch.is_tensor(x)
                        and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous()
                        and not torch.is_grad_enabled() and x.numel() < 2 ** 32 - 1)

            def checked(m, cls, x, y):
                if not _flag(m, _POOL_VERIFIED):           # once per process: the same bits as torch's kernel?
                    ref = cls.forward(m, x)
                    if not torch.equal(y, ref):
                        _set_flag(m, _POOL_OFF)
                        return ref
                    _set_flag(m, _POOL_VERIFIED)
                return y
            for m in model.modules():
                if "forward" in m.__dict__:
                    continue
                if type(m) is torch.nn.MaxPool2d:
                    def forward(x, m=m):
                        if (_flag(m, _POOL_OFF) or not pool_active(x) or pair(m.dilation) != (1, 1) or m.ceil_mode
                                or m.return_indices):
                            return torch.nn.MaxPool2d.forward(m, x)
                        k, p = pair(m.kernel_size), pair(m.padding)
                        st = pair(m.stride if m.stride is not None else m.kernel_size)
                        return checked(m, torch.nn.MaxPool2d, x, _native.maxpool2d_f32(x, k, st, p))
                    m.forward = forward
                    patched.append(m)
                elif type(m) is torch.nn.AvgPool2d:
                    def forward(x, m=m):
                        if (_flag(m, _POOL_OFF) or not pool_active(x) or pair(m.kernel_size) != tuple(x.shape[2:])
                                or pair(m.padding) != (0, 0) or m.ceil_mode or m.divisor_override is not None
                                or x.shape[2] * x.shape[3] > 144):
                            return torch.nn.AvgPool2d.forward(m, x)
                        return checked(m, torch.nn.AvgPool2d, x, _native.avgpool_global_f32(x))
                    m.forward = forward
                    patched.append(m)
        # an out-of-place nn.ReLU fed directly by one of the modules above is served by that module's kernel
        for m in model.modules():
            if type(m) is not torch.nn.ReLU or m.inplace or "forward" in m.__dict__:
                continue

            def forward(x, m=m):
                last, ready = ctl.last_out, ctl.relu_ready
                if last is not None and last[1] is x:
                    ctl.relu_after[last[0]] = m           # (re)learned on every call: who feeds this ReLU
                if ctl.poison is not None:
                    z = ctl.poison.relu(m, x)
                    if z is not None:
                        return z
                if ready is not None and ready[0] is x and ready[2] is m and ready[3] == x._version:
                    ctl.relu_ready = None                 # (same tensor object, not written to since)
                    return ready[1]
                return torch.nn.functional.relu(x)
            m.forward = forward
            patched.append(m)
        return patched

    def _finish_own_conv(self, module, m, kind, x, key, output):
        """Forward-hook half of a convolution that runs on fq_conv1x1_f32 / fq_conv_stem_f32: `output` is the empty tensor
        the patched forward returned.  Returns True when the statistic of `output` is done."""
        ctl = self._hook_ctl
        coll = ctl.fuse_collector
        run = _float_conv.runner(m, kind, x)
        if module is not m or coll is None or key is None:     # not a cared tensor: the convolution only
            run(out=output)
            return False
        row = coll.row_of(key)
        if ctl.fuse_stat == "hist":                         # pass 2 (verified in pass 1)
            if (kind == "c1" and self.fuse_conv_add and not self.materialize_all and ctl.defer_ok.get(m) is not None
                    and ctl.eager is not None):
                ctl.deferred[id(output)] = (output, m, x, key, row, output._version, x._version)       # (see below)
                return True
            self._run_with_relu(m, output, lambda r, o: run(interval_dev=coll.interval_device, hist_dev=coll.hist_device, row=row,
                                                            relu_out=r, out=o), key)
            ctl.hist_fused += 1
            return True
        ref = _float_conv.verified(m, run, x, kind)            # first use of this kernel: against torch, once per process
        if ref is not None:
            output.copy_(ref)                                  # this module keeps the library convolution from now on
            return False
        ctl.fuse_verified.add(m)
        if kind == "c1" and self.fuse_conv_add and ctl.defer_ok.get(m) is not None and ctl.eager is not None:
            # its kernel runs inside the launch of the Eltwise that adds this tensor (_finish_deferred); until then `output`
            # is an allocation nobody reads -- which the poison probe has shown for this model
            ctl.deferred[id(output)] = (output, m, x, key, row, output._version, x._version)
            return True
        self._run_with_relu(m, output, lambda r, o: run(max_dev=coll.max_device, row=row, relu_out=r, out=o), key)
        coll.note_max_refreshed()
        ctl.own_conv1x1 = ctl.own_conv1x1 + 1
        return True

    def _run_deferred(self, d):
        """A deferred convolution on its own after all (what its hook would have launched)."""
        output, m, x, _key, row, _v, x_version = d
        self._deferred_input_intact(x, x_version)
        ctl = self._hook_ctl
        coll = ctl.fuse_collector
        if coll is None:
            _float_conv.runner(m, "c1", x)(out=output)
            return
        if ctl.fuse_stat == "hist":
            _float_conv.runner(m, "c1", x)(interval_dev=coll.interval_device, hist_dev=coll.hist_device, row=row, out=output)
            ctl.hist_fused += 1
        else:
            _float_conv.runner(m, "c1", x)(max_dev=coll.max_device, row=row, out=output)
            coll.note_max_refreshed()
            ctl.own_conv1x1 = ctl.own_conv1x1 + 1
        if ctl.eager is not None:
            ctl.eager.note(_key, output)

    @staticmethod
    def _deferred_input_intact(x, x_version):
        """A deferred convolution runs LATER than the model called it, on the input tensor it was called with.  The poison probe
        runs the convolution at its own position, so a model that writes that input in place between the convolution and its
        Eltwise is invisible to it -- the version counter is not: such a forward cannot be calibrated with the deferral."""
        if x._version != x_version:
            raise _native.FqError("the input of a 1x1 convolution was written in place between the convolution and the Eltwise "
                                  "that consumes its output; the one-kernel residual tail cannot run on it: set "
                                  "Quantity.fuse_conv_add = False (FQ_FUSE_CONV_ADD=0)")

    def _finish_deferred(self, module, m, a, b, key, output):
        """Hook half of an Eltwise one of whose operands is a deferred convolution: convolution + bias, that tensor's abs-max,
        the sum, its abs-max and the ReLU behind it in one launch.  Returns True when done; False after running the
        convolution alone (the Eltwise then takes its usual path)."""
        ctl = self._hook_ctl
        d = None
        for t in (a, b):
            e = ctl.deferred.get(id(t))
            if e is not None and e[0] is t:
                d = e
                break
        if d is None:
            return False
        del ctl.deferred[id(d[0])]
        t3, conv, x, conv_key, conv_row, version, x_version = d
        self._deferred_input_intact(x, x_version)
        coll = ctl.fuse_collector
        other = b if t3 is a else a
        relu = ctl.relu_after.get(m) if self.fuse_relu else None
        keep = lambda name: self.materialize_all or (ctl.keep_feats and (ctl.keep_names is None or name in ctl.keep_names))
        keep_y, keep_s = keep(conv_key), keep(key) if key is not None else True
        # Both wanted by pass 2's cache: the sum is conv3's output + the shortcut, and the shortcut exists anyway -- so the sum is
        # NOT written (4 B per element less, and small planes keep the one-kernel form); the cache keeps the shortcut in its place
        # and pass 2 histograms the pair (fq_hist2048_pair_seg).  Only for an Eltwise whose shortcut the probe forward has seen
        # unmodified until the end of the forward (ctl.pair_ok), and never when somebody wants every tensor materialised.
        pair = (keep_y and keep_s and key is not None and self.pair_hist and not self.materialize_all and m in ctl.pair_ok
                and ctl.fuse_stat != "hist" and ctl.eager is not None and getattr(coll, "supports_pairs", False)
                and other.dtype == torch.float32 and other.data_ptr() % 16 == 0 and t3.data_ptr() % 16 == 0)
        if pair:
            keep_s = False
        # (small planes with BOTH tensors kept: three store streams of partial lines make the one kernel slower than the two,
        #  scripts/conv_add_bench.py: 356 vs 340 us at 14 x 14, 283 vs 276 at 7 x 7)
        small = t3.shape[2] * t3.shape[3] < 28 * 28
        hist = ctl.fuse_stat == "hist"
        if (module is not m or coll is None or key is None or ctl.defer_ok.get(conv) is not m
                or relu is None or not _flag(m, _FUSION_VERIFIED) or not _flag(m, _RELU_VERIFIED) or t3._version != version
                or other is t3 or other.shape != t3.shape or not other.is_contiguous()
                or (m not in ctl.fuse_verified if hist else (keep_y and keep_s and small))):
            self._run_deferred(d)
            return False
        r = torch.empty_like(t3)
        if hist:                                                # pass 2: both histograms, neither tensor written
            _native.conv1x1_add_hist_f32(x, _float_conv.weight(conv, "c1"), conv.bias, conv.stride[0], other, coll.interval_device,
                                         coll.hist_device, conv_row, coll.row_of(key), r)
            ctl.hist_fused += 2
            ctl.relu_ready = (output, r, relu, output._version)
            ctl.fused_relus.add(relu)
            ctl.deferred_hists += 1
            return True
        _native.conv1x1_add_f32(x, _float_conv.weight(conv, "c1"), conv.bias, conv.stride[0], other, coll.max_device, conv_row,
                                coll.row_of(key), r, out=t3 if keep_y else None, sum_out=output if keep_s else None)
        coll.note_max_refreshed()
        if keep_y and ctl.eager is not None:
            ctl.eager.note(conv_key, t3)                        # (what its own hook left out: the tensor exists only now)
        if pair:
            # (the shortcut is the ReLU output another tail of this forward wrote?  Then it is max(that sum, 0) and pass 2 can
            #  re-make it from that sum's own pair instead of the cache keeping it: src = that sum's key)
            src = ctl.sum_relu.get(id(other))
            ctl.pairs[key] = (conv_key, other, other._version, src[1] if src is not None and src[0]() is other else None)
            ctl.pair_sums += 1
        if key is not None:
            ctl.sum_relu[id(r)] = (weakref.ref(r), key)
        ctl.fuse_verified.add(m)
        ctl.relu_ready = (output, r, relu, output._version)
        ctl.fused_relus.add(relu)
        ctl.deferred_adds += 1
        return True

    def _wanted(self, key):
        """Does anything of this calibration read the hooked tensor `key` of the running forward from HBM again?  (Pass 1: what
        pass 2's cache keeps; pass 2: nothing.)"""
        ctl = self._hook_ctl
        if self.materialize_all:
            return True
        if ctl.fuse_stat == "hist":
            return False
        return ctl.keep_feats and (ctl.keep_names is None or key in ctl.keep_names)

    def _run_with_relu(self, m, output, run, key=None):
        """run(relu_out, out) launches m's fused kernel (out: where the module's own output goes).  When an out-of-place
        nn.ReLU is known to consume `output` directly, the kernel writes that ReLU's result as well and the patched ReLU.forward
        hands it out instead of launching -- and when that ReLU is PROVEN to be the only reader of `output` (relu_only_ok) and
        pass 2 does not want the tensor, `output` itself is not written (out = False)."""
        ctl = self._hook_ctl
        relu = ctl.relu_after.get(m) if self.fuse_relu else None
        if relu is None:
            run(None, output)
            return
        r = torch.empty_like(output)
        skip = (key is not None and self.skip_unread_outputs and m in ctl.relu_only_ok and _flag(m, _RELU_VERIFIED)
                and ctl.eager is not None and not self._wanted(key))
        run(r, False if skip else output)
        if skip:
            ctl.skipped_outputs += 1
        if not _flag(m, _RELU_VERIFIED):              # once per process: the same bits as torch's ReLU?
            if not torch.equal(r, torch.nn.functional.relu(output)):
                self.fuse_relu = False
                return
            _set_flag(m, _RELU_VERIFIED)
        ctl.relu_ready = (output, r, relu, output._version)      # holds the tensor itself: identity, not a reusable id
        ctl.fused_relus.add(relu)

    def _finish_fused_conv(self, module, pending, key, output):
        """Forward-hook half of the fused conv: add the bias (and take the abs-max when the tensor is a cared one).
        Returns True when the statistics of `output` are done."""
        m, x = pending
        ctl = self._hook_ctl
        coll = ctl.fuse_collector
        if isinstance(x, tuple) and isinstance(x[0], str):  # a convolution on fq_conv1x1_f32 / fq_conv_stem_f32: output is still empty
            return self._finish_own_conv(module, m, x[0], x[1], key, output)
        if isinstance(x, tuple):                            # Eltwise: output is an empty tensor waiting for x + y
            a, b = x
            if ctl.deferred and self._finish_deferred(module, m, a, b, key, output):
                return True
            if ctl.own_plain and module is m:
                return self._plain_eltwise(m, a, b, output)
            if module is not m or coll is None or key is None:
                torch.add(a, b, out=output)
                return False
            row = coll.row_of(key)
            if ctl.fuse_stat == "hist":                  # pass 2 (verified in pass 1): the sum, histogrammed on the way out
                self._run_with_relu(m, output, lambda r, _o: _native.add_hist(a, b, coll.interval_device, coll.hist_device, row,
                                                                              out=output, relu_out=r))
                ctl.hist_fused += 1
                return True
            if not _flag(m, _FUSION_VERIFIED):        # first use: the kernel against torch.add, once per process
                scratch = torch.zeros(1, dtype=torch.float32, device=output.device)
                z = _native.add_absmax(a, b, scratch, 0)
                want = torch.add(a, b)
                if not bool((z == want).all() & (scratch[0] == want.abs().max())):        # (one synchronisation)
                    ctl.fuse_off = True
                    output.copy_(want)
                    return False
                _set_flag(m, _FUSION_VERIFIED)
            ctl.fuse_verified.add(m)
            self._run_with_relu(m, output, lambda r, _o: _native.add_absmax(a, b, coll.max_device, row, out=output, relu_out=r))
            coll.note_max_refreshed()
            return True
        if module is not m or coll is None or key is None or not output.is_contiguous() or output.dim() < 2:
            output.add_(m.bias.view(1, -1, *([1] * (output.dim() - 2))))     # what torch does
            return False
        row = coll.row_of(key)
        if ctl.fuse_stat == "hist":                      # pass 2 (verified in pass 1)
            self._run_with_relu(m, output, lambda r, _o: _native.bias_add_hist(output, m.bias, coll.interval_device,
                                                                               coll.hist_device, row, relu_out=r))
            ctl.hist_fused += 1
            return True
        if m in ctl.fuse_verified or _flag(m, _FUSION_VERIFIED):
            ctl.fuse_verified.add(m)
            self._run_with_relu(m, output, lambda r, _o: _native.bias_add_absmax(output, m.bias, coll.max_device, row, relu_out=r))
            coll.note_max_refreshed()
            return True
        # First fused use of this module (its second batch).  Two things are checked once per module:
        #   * the kernel itself: on the same convolution result it must leave exactly torch's `raw + bias` and that
        #     tensor's maximum;
        #   * the decomposition: torch's own forward must equal convolution-without-bias + bias bit for bit -- on layers
        #     where torch's forward is reproducible at all (MIOpen's Winograd kernels for some 3x3 shapes are not: two
        #     identical calls differ in the last bit, so there is nothing bitwise to compare against).
        ref = torch.nn.Conv2d.forward(m, x)
        raw = m._conv_forward(x, m.weight, None)
        want = raw + m.bias.view(1, -1, *([1] * (raw.dim() - 2)))
        scratch = torch.zeros(1, dtype=torch.float32, device=output.device)
        _native.bias_add_absmax(raw, m.bias, scratch, 0)
        kernel_ok = torch.equal(raw, want) and float(scratch[0]) == float(want.abs().max())
        reproducible = torch.equal(ref, torch.nn.Conv2d.forward(m, x))
        if kernel_ok and (torch.equal(raw, ref) or not reproducible):
            ctl.fuse_verified.add(m)
            _set_flag(m, _FUSION_VERIFIED)             # a property of (module, MIOpen, this library): checked once per process
            _native.bias_add_absmax(output, m.bias, coll.max_device, row)
            coll.note_max_refreshed()
            return True
        ctl.fuse_off = True                              # never silently different: torch's add, statistics as usual
        output.add_(m.bias.view(1, -1, *([1] * (output.dim() - 2))))
        return False

    def _plain_eltwise(self, m, a, b, output):
        """Per-channel calibration (own_plain): an Eltwise and the out-of-place ReLU behind it as ONE launch of the add producer
        (four passes over the tensor at HBM speed where torch's add + ReLU make five at a third of it: 5.8 -> 1.3 ms per
        256-image ResNet-50 forward).  The per-channel statistics are taken from the finished tensors, so the kernel's own
        abs-max goes to a scratch word nobody reads.  Returns False: the statistics of `output` are not done."""
        ctl = self._hook_ctl
        if not _flag(m, _FUSION_VERIFIED):            # first use: the kernel against torch.add, once per process
            probe = torch.zeros(1, dtype=torch.float32, device=output.device)
            z = _native.add_absmax(a, b, probe, 0)
            want = torch.add(a, b)
            if not bool((z == want).all() & (probe[0] == want.abs().max())):
                ctl.fuse_off = True
                output.copy_(want)
                return False
            _set_flag(m, _FUSION_VERIFIED)
        scratch = torch.empty(1, dtype=torch.float32, device=output.device)       # (never read)
        self._run_with_relu(m, output, lambda r, _o: _native.add_absmax(a, b, scratch, 0, out=output, relu_out=r))
        return False

    def _training_state_modules(self):
        """Modules whose forward in training mode changes state or draws random numbers: anything in training mode that owns
        buffers (BatchNorm's running statistics) or is a dropout layer.  (A parameter-free module left in training mode --
        e.g. the Identity that merge_bn puts in a BatchNorm's place -- does not count.)"""
        from torch.nn.modules.dropout import _DropoutNd
        return [m for m in self.model.modules()
                if m.training and (isinstance(m, _DropoutNd) or next(m.buffers(recurse=False), None) is not None)]

    def _probe_forward(self, own_plain, deferral=None):
        """One forward of the model on a random input of INPUT_SHAPE with the hooks watching for in-place consumers; returns
        whether a hooked tensor was written to after its hook ran.  The reference feeds its models random input exactly
        once, in build_net_structure (pytorch_quantizer.py:21-62); this additional draw comes from a generator of its own,
        so the global RNG stream a user script sees afterwards is the reference's."""
        ctl = self._hook_ctl
        probe = _EagerStats(lambda tensors: None, _AFTER_FORWARD)
        saved = ctl.own_plain
        ctl.eager, ctl.own_plain, ctl.poison = probe, own_plain, deferral
        self._probe_out = None
        try:
            dev = self._model_device(self.model)
            gen = torch.Generator(device=dev)
            gen.manual_seed(0x5eed)
            shapes = [self.input_size] if isinstance(self.input_size, tuple) else list(self.input_size)
            with torch.no_grad():
                self._probe_out = self.model(*[torch.rand(*s_, device=dev, generator=gen) for s_ in shapes])
        finally:
            ctl.eager, ctl.own_plain, ctl.poison = None, saved, None
        return bool(probe.modified())

    @staticmethod
    def _only_our_hook(m):
        """Nobody but this calibration's own forward hook watches module m."""
        import torch.nn.modules.module as _mod
        return (len(m._forward_hooks) <= 1 and not m._forward_pre_hooks and not _mod._global_forward_hooks
                and not _mod._global_forward_pre_hooks)

    def _prove_deferral(self, probe, first_feats, first_out):
        """After the learning probe forward: pick the (1x1 convolution, Eltwise, ReLU) chains fq_conv1x1_add_f32 can take and the
        (convolution, ReLU) chains whose convolution output nobody else reads, run the poison forward (_DeferralProbe) and
        return ({conv: Eltwise}, {conv}) when it changes nothing, else ({}, set())."""
        ctl = self._hook_ctl
        cared = set(self.net_info.keys())
        for elt, conv in (probe.pairs.items() if self.fuse_conv_add else ()):
            relu = ctl.relu_after.get(elt)
            if (relu is None or probe.keys.get(conv) not in cared or probe.keys.get(elt) not in cared
                    or conv.kernel_size != (1, 1) or conv.padding != (0, 0)
                    or not _native.conv1x1_add_f32_supported(conv.in_channels, conv.out_channels)
                    or not (self._only_our_hook(conv) and self._only_our_hook(elt) and self._only_our_hook(relu))
                    or conv in probe.candidates):
                continue
            probe.candidates[conv] = (elt, relu)
        for (_y, conv) in (probe.conv_out.values() if self.skip_unread_outputs else ()):
            relu = ctl.relu_after.get(conv)
            if (relu is None or conv in probe.candidates or probe.keys.get(conv) not in cared
                    or not (self._only_our_hook(conv) and self._only_our_hook(relu))):
                continue
            probe.relu_only[conv] = relu
        # (the loop variable outlives the loop: it held the LAST convolution's output, whose reference count then exceeded the
        #  control's by one -- a suspect in every calibration, i.e. a heap pass of 11-16 ms that found nothing; round 5)
        _y = None
        # keepers (code that stores one of these tensors and reads it after the forward: invisible to the poison): whoever still
        # refers to a convolution's output or to a sum now that the learning forward has returned, and is not this calibration
        # (cheap first: a tensor nobody else holds has exactly the reference count of a CONTROL tensor put into the same containers
        #  of this calibration; only tensors that exceed it -- normally none -- are looked up in the heap, which costs tens of ms)
        import sys
        ctl.last_out = ctl.relu_ready = None                   # stale after the forward; they would count as holders
        out_of = dict((conv, y) for (y, conv) in probe.conv_out.values())
        c_conv, c_sum = torch.empty(1), torch.empty(1)
        probe.conv_out[id(c_conv)] = (c_conv, None)             # a convolution's output sits in these three containers ...
        probe.outputs["control conv"], out_of["control conv"] = c_conv, c_conv
        probe.outputs["control sum"] = c_sum                    # ... an Eltwise's in this one
        base_conv, base_sum = sys.getrefcount(c_conv), sys.getrefcount(c_sum)
        del probe.conv_out[id(c_conv)], probe.outputs["control conv"], out_of["control conv"], probe.outputs["control sum"]
        self.deferral_refused = {}
        watch, suspects = {}, []                               # conv -> the tensors of its chain that would stay un-written
        for conv in list(probe.candidates) + list(probe.relu_only):
            y = out_of.get(conv)
            s_ = probe.outputs.get(probe.candidates[conv][0]) if conv in probe.candidates else None     # the sum, as the model holds it
            watch[conv] = (y, s_)
            if torch.is_tensor(y) and (sys.getrefcount(y) - 1 > base_conv or y._use_count() > 1):       # (- 1: `watch` holds it too)
                suspects.append(y)
            if torch.is_tensor(s_) and (sys.getrefcount(s_) - 1 > base_sum or s_._use_count() > 1):
                suspects.append(s_)
        held = {}
        if suspects:
            ours = [probe.conv_out, probe.outputs, first_feats, self._probe_feats, out_of, watch, suspects]
            ours += list(probe.conv_out.values()) + list(watch.values())
            held = probe.holders(suspects, ours)
        for conv, ts in watch.items():
            kept = [h for t in ts if torch.is_tensor(t) for h in held.get(id(t), ())]
            if kept:
                self.deferral_refused[probe.keys.get(conv)] = kept
                probe.candidates.pop(conv, None)
                probe.relu_only.pop(conv, None)
        del out_of, watch, suspects
        if not probe.candidates and not probe.relu_only:
            return {}, set()
        probe.mode = "poison"
        named = self._probe_feats
        named.clear()
        self._probe_forward("unchecked", probe)
        skip = probe.poisoned_keys()

        def same(u, v):
            if not (torch.is_tensor(u) and torch.is_tensor(v)) or u.shape != v.shape or u.dtype != v.dtype:
                return False
            if u.dtype == torch.float32:
                return torch.equal(u.contiguous().view(torch.int32), v.contiguous().view(torch.int32))
            return torch.equal(u, v)
        ok = set(first_feats) == set(named)
        for k, t in first_feats.items():
            if not ok:
                break
            if k not in skip:
                ok = same(t, named[k])
        if ok and torch.is_tensor(first_out):
            ok = same(first_out, self._probe_out)
        self._probe_out = None
        if not ok:
            return {}, set()
        # ... and the Eltwises whose shortcut operand was still as the add saw it when the learning forward ended: pass 2 may read it
        # in place of a stored sum (Quantity.pair_hist).  (Both probe forwards ran the same code on the same input.)
        ctl.pair_ok = set(elt for elt, (o, v) in probe.others.items() if torch.is_tensor(o) and o._version == v)
        probe.others = {}
        return {conv: pair[0] for conv, pair in probe.candidates.items()}, set(probe.relu_only)
