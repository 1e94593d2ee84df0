"""KL-divergence threshold search: histogram -> threshold bin -> fractional bit count.

Drop-in for reference quantity/common/quantity/quantizer.py (Quantizer :7, quantize :43-75,
quantize_worker :77-93, normalize_distribution :95-96, threshold_distribution :98-167,
compute_kl_divergence :169-174): same constructor, quantize(), .bits and .threshold_value.

All rows go through one fq_kl_threshold call (1920 candidate thresholds x rows as independent
workgroups, float64 in the reference's operation order).  Only the last scalar step -- threshold bin
to bits, which the reference does with math.log -- runs on the host, with the same CPython/NumPy
scalar arithmetic, so exact powers of two round the way they do in the reference.
`worker_num` is accepted and ignored.
"""
import math

import numpy as np
import torch

from . import _native

__all__ = ["Quantizer"]


def _rows_to_device(distributions, names, device):
    """Stack per-tensor histograms (NumPy int32 / merged float64, or device int64) into int64[T,2048]."""
    if isinstance(distributions, torch.Tensor):
        assert (distributions.dtype == torch.int64 and distributions.dim() == 2 and distributions.shape[0] == len(names)
                and distributions.shape[1] in _native.SUPPORTED_BINS)
        return distributions.contiguous()
    rows = []
    for n in names:
        h = distributions[n]
        if isinstance(h, torch.Tensor):
            rows.append(h.to(device=device, dtype=torch.int64))
            continue
        h = np.asarray(h)
        if h.dtype.kind == "f":
            hi = h.astype(np.int64)
            if not np.array_equal(hi.astype(h.dtype), h):
                raise ValueError("histogram %r holds non-integer counts" % n)
            h = hi
        rows.append(torch.from_numpy(np.ascontiguousarray(h, dtype=np.int64)).to(device))
    return torch.stack(rows).contiguous() if rows else torch.zeros(0, _native.BINS, dtype=torch.int64, device=device)


class Quantizer(object):

    def __init__(self, tensor_list, worker_num=1, debug=False, device=None):
        self._tensor_list = list(tensor_list)
        self._worker_num = worker_num
        self._debug = debug
        self._device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self._bits = {}
        self._threshold_value = {}
        self._threshold_bin = {}
        self._kl_best = {}
        self._kl_runner_up = {}
        self._quantized_flag = False

    @property
    def bits(self):
        assert self._quantized_flag, "Please use quantize() first."
        return self._bits

    @property
    def threshold_value(self):
        assert self._quantized_flag, "Please use quantize() first."
        return self._threshold_value

    @property
    def threshold_bins(self):
        """The raw threshold bin t* in [128, 2047] per tensor (not exposed by the reference)."""
        assert self._quantized_flag, "Please use quantize() first."
        return self._threshold_bin

    @property
    def kl_margin(self):
        """{name: (KL(t*), smallest KL of any other candidate)} (not exposed by the reference).  The argmin is decided by
        fq_log, a correctly rounded logarithm; the reference's np.log is faithful but not correctly rounded, so where the
        two values are within a few ulps of each other the reference could have picked the other candidate.  near_ties()
        lists those rows.  (Exact when the search ran exhaustively -- always below 256 rows, i.e. for every per-tensor
        calibration; in the screened mode of fq_kl_threshold_ex the runner-up is the closed-form value, within
        FQ_KL_SCREEN_BOUND = 2e-13 of the exact one, unless it was itself a survivor.)"""
        assert self._quantized_flag, "Please use quantize() first."
        return {n: (self._kl_best[n], self._kl_runner_up[n]) for n in self._kl_best}

    def near_ties(self, rel=1e-12):
        """Names whose best and second-best KL differ by less than rel * |best| (an argmin a last-bit difference in the
        logarithm could flip).  Empty on every golden, fuzz and ResNet histogram seen so far."""
        out = []
        for n, (b, r) in self.kl_margin.items():
            if math.isfinite(b) and math.isfinite(r) and (r - b) <= rel * abs(b):
                out.append(n)
        return out

    def _result_order(self):
        """The reference fills .bits worker by worker (worker 0 takes the first chunk plus the
        remainder, quantizer.py:48-75); weight.table lines follow that dict order when the KL branch
        of weight_quantize is used, so the order is part of the output contract."""
        names, w = self._tensor_list, max(int(self._worker_num), 1)
        per = len(names) // w
        order = []
        for i in range(w):
            order += names[i * per:(i + 1) * per]
            if i == 0:
                order += names[w * per:]
        return order

    def quantize(self, distributions, distribution_intervals):
        """distributions: {name: 2048 counts} (or an int64[T,2048] device tensor in tensor_list
        order); distribution_intervals: {name: bin width}."""
        if self._debug and self._quantized_flag:
            return
        self._quantized_flag = True
        hist = _rows_to_device(distributions, self._tensor_list, self._device)
        thr_dev, best_dev, runner_dev = _native.kl_threshold(hist, want_evidence=True)
        thr = dict(zip(self._tensor_list, thr_dev.cpu().numpy()))
        self._kl_best = dict(zip(self._tensor_list, (float(v) for v in best_dev.cpu().numpy())))
        self._kl_runner_up = dict(zip(self._tensor_list, (float(v) for v in runner_dev.cpu().numpy())))
        for name in self._result_order():
            t = int(thr[name])
            # reference quantizer.py:86-90; NumPy scalar typing decides fp32 vs float64 here exactly
            # as it does there (np.float32 interval -> fp32 product, Python float -> float64)
            threshold_bias = (t + 0.5) * distribution_intervals[name]
            bit = int(8 - 1 - math.ceil(math.log(threshold_bias, 2)))
            self._threshold_bin[name] = t
            self._threshold_value[name] = threshold_bias
            self._bits[name] = bit
            print("{} ".format(name), "bit:", bit)
        ties = self.near_ties()
        if ties:
            print("[WARNING] KL near-ties (best and second-best threshold within 1e-12 relative):", ties)
