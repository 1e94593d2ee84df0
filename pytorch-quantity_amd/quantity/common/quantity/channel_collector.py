"""Per-channel calibration: one abs-max / 2048-bin histogram / KL threshold per (tensor, channel).

EXTENSION, not part of the reference API: the reference's calibrator is per tensor
(distribution_collector.py:40-42; feat.table holds one bit per layer), so there is nothing to be
bit-compatible with.  A histogram row here is (tensor, channel): fq_absmax_chan / fq_hist2048_chan read
the NCHW activations IN PLACE (a workgroup owns one channel and a group of images, see csrc/fq_calib.hip) and
write row = first_row(tensor) + c; one launch covers all tensors of a forward.  Arithmetic per row is exactly
the per-tensor arithmetic (same interval formula, same binning, same KL sweep), which
tests/test_gpu_per_channel.py checks against the CPU oracle channel by channel.
"""
import numpy as np
import torch

from . import _native
from ._collectives import StatCollectives

__all__ = ["ChannelCollector"]

_MAX_SEGS = 1024            # FQ_MAX_SEGS per C-ABI call


class ChannelCollector(StatCollectives):

    def __init__(self, channels, statistic=1, device=None):
        """channels: ordered {tensor name: channel count C} (dimension 1 of the activation)."""
        self._names = list(channels.keys())
        self._channels = dict(channels)
        self._first = {}
        row = 0
        for n in self._names:
            self._first[n] = row
            row += int(channels[n])
        self._rows = row
        self._statistic = statistic
        self._device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self._max = torch.zeros(row, dtype=torch.float32, device=self._device)
        # (W equal row blocks when distributed -- what the reduce-scatter after pass 2 takes; the kernels only see the real rows)
        self._hist_padded = torch.zeros(self.padded_rows(row), _native.BINS, dtype=torch.int64, device=self._device)
        self._hist = self._hist_padded[:row]
        self._interval = None
        self._own_block = None           # (first row, [S][2048]) after reduce_scatter_hist(): this rank's rows, summed over ranks

    @property
    def rows(self):
        return self._rows

    def row_range(self, name):
        return self._first[name], self._first[name] + self._channels[name]

    def _dense(self, tensors):
        ts, row0s = [], []
        for n in self._names:
            if n not in tensors:             # a partial dict: the calibration loop may hand tensors over one by one
                continue
            t = tensors[n]
            assert t.shape[1] == self._channels[n], (n, tuple(t.shape), self._channels[n])
            ts.append(t.detach())
            row0s.append(self._first[n])
        return ts, row0s

    def refresh_max_val(self, tensors):
        ts, row0s = self._dense(tensors)
        for i in range(0, len(ts), _MAX_SEGS):
            _native.absmax_chan(ts[i:i + _MAX_SEGS], row0s[i:i + _MAX_SEGS], self._max)

    def intervals(self):
        """fp32 bin width per row: statistic * max / 2048 + 1e-12 with NumPy fp32 scalars (the
        reference's expression, distribution_collector.py:61); rows that never left zero get 1e-12."""
        m = self._max.cpu().numpy()
        # element-wise the reference's scalar expression (NumPy-2 weak promotion keeps everything fp32; a row that never
        # left zero gives 0 + 1e-12); one vector expression instead of 42 667 scalar ones (61 ms of Python for ResNet-50)
        iv = (self._statistic * m / _native.BINS + 1e-12).astype(np.float32, copy=False)
        assert iv.dtype == np.float32
        self._interval = torch.from_numpy(iv).to(self._device)
        return iv

    def add_to_distributions(self, tensors):
        if self._interval is None:
            self.intervals()
        ts, row0s = self._dense(tensors)
        for i in range(0, len(ts), _MAX_SEGS):
            _native.hist2048_chan(ts[i:i + _MAX_SEGS], row0s[i:i + _MAX_SEGS], self._interval, self._hist)

    def _stat_tensors(self):             # all_reduce_max() / reduce_scatter_hist(): StatCollectives
        return self._max, self._hist_padded

    @property
    def max_device(self):
        return self._max

    @property
    def hist_device(self):
        return self._hist

    def quantize(self):
        """-> {tensor name: [bits per channel]} via the KL sweep (quantizer.py:86-90 per row).  After reduce_scatter_hist() the
        sweep runs on this rank's row block only and the ranks exchange (threshold bin, bits) with one all-gather."""
        import time
        iv = self._interval.cpu().numpy()
        torch.cuda.synchronize(self._device)
        t0 = time.perf_counter()
        if self._own_block is None:
            thr = _native.kl_threshold(self._hist).cpu().numpy()
            self.kl_seconds = round(time.perf_counter() - t0, 4)      # the sweep of all rows (device time: .cpu() waits)
            # quantizer.py:86-90 per row through the C helper (the host libm call CPython's math.log(x, 2) makes), all rows at once
            all_bits, _thr_val = _native.bits_from_threshold(thr, iv)
        else:
            lo, block = self._own_block
            per = int(block.shape[0])
            thr_own = _native.kl_threshold(block).cpu().numpy()
            self.kl_seconds = round(time.perf_counter() - t0, 4)      # the sweep of this rank's rows
            iv_own = np.ones(per, dtype=np.float32)                   # (padding rows: any positive width; their result is dropped)
            n_real = max(0, min(lo + per, self._rows) - lo)
            iv_own[:n_real] = iv[lo:lo + n_real]
            bits_own, _thr_val = _native.bits_from_threshold(thr_own, iv_own)
            local = torch.from_numpy(np.stack([np.asarray(thr_own, dtype=np.int32), np.asarray(bits_own, dtype=np.int32)])).to(self._device)
            both = self.all_gather_rows(local).cpu().numpy()
            thr, all_bits = both[0, :self._rows], both[1, :self._rows]
        bits = {}
        for n in self._names:
            lo, hi = self.row_range(n)
            bits[n] = all_bits[lo:hi].tolist()                    # (Python ints; a comprehension over 42 667 NumPy scalars took 2 ms)
        self.threshold_bins = thr
        return bits
