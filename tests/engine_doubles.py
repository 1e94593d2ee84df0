"""Oracle-backed stand-ins for the statistics engine (DistributionCollector / Quantizer), used ONLY by
the CPU test-suite to exercise the host logic of tools.Quantity (graph discovery, merge groups, table
and JSON writers, sharding + all-reduce plumbing) on a box without a GPU.

TEST INFRASTRUCTURE: the product never imports this; its engine is the HIP library and nothing else.
The doubles compute with oracle/fq_oracle.c (the CPU restatement) on host NumPy arrays and use the
same method names the product classes expose.
"""
import math

import numpy as np
import torch

from oracle import fq_oracle as orc
from common.quantity._collectives import StatCollectives

BINS = 2048


def _np(t):
    if isinstance(t, torch.Tensor):
        return t.detach().cpu().numpy().astype(np.float32, copy=False).ravel()
    return np.asarray(t, dtype=np.float32).ravel()


class OracleCollector(StatCollectives):

    def __init__(self, tensor_list, interval_num=2048, statistic=1, worker_num=1, debug=False):
        assert interval_num == BINS
        self._tensor_list = list(tensor_list)
        self._row = {n: i for i, n in enumerate(self._tensor_list)}
        self._statistic = statistic
        self._interval_num = interval_num
        T = len(self._tensor_list)
        self._max = np.zeros(T, dtype=np.float32)
        self._hist = np.zeros((T, BINS), dtype=np.int64)
        self._refreshed = False
        self._added = False

    @property
    def max_vals(self):
        assert self._refreshed
        return {n: (self._max[i] if self._max[i] > 0 else 0) for i, n in enumerate(self._tensor_list)}

    @property
    def distribution_intervals(self):
        mv = self.max_vals
        iv = {n: self._statistic * mv[n] / self._interval_num + 1e-12 for n in self._tensor_list}
        self._distribution_intervals = iv
        return iv

    @property
    def distributions(self):
        assert self._added
        return {n: self._hist[i].astype(np.int32) for i, n in enumerate(self._tensor_list)}

    def refresh_max_val(self, tensors):
        self._refreshed = True
        for i, n in enumerate(self._tensor_list):
            self._max[i] = orc.absmax(_np(tensors[n]), self._max[i])

    def add_to_distributions(self, tensors):
        self._added = True
        if not hasattr(self, "_distribution_intervals"):
            self.distribution_intervals
        for i, n in enumerate(self._tensor_list):
            orc.hist2048(_np(tensors[n]), np.float32(self._distribution_intervals[n]), self._hist[i])

    # all_reduce_max / all_reduce_hist are the PRODUCT's (common.quantity._collectives.StatCollectives): the double only
    # says which host arrays hold its state, so the world_size-2 gloo tests run the lines the GPU collectors run
    def _stat_tensors(self):
        return torch.from_numpy(self._max), torch.from_numpy(self._hist)

    def _note_max_reduced(self):
        self._refreshed = True

    def _note_hist_reduced(self):
        self._added = True

    def merged_distributions(self, groups):
        merged = self._hist.copy()
        for g in groups:
            idx = [self._row[n] for n in g]
            merged[idx] = merged[idx].sum(axis=0, keepdims=True)
        return {n: merged[i] for i, n in enumerate(self._tensor_list)}

    def quantize_param(self, tensor, bit):
        a = tensor.detach().cpu().numpy() if isinstance(tensor, torch.Tensor) else np.asarray(tensor)
        return orc.quantize_param_i32(a, bit)


class OracleQuantizer(object):

    def __init__(self, tensor_list, worker_num=1, debug=False):
        self._tensor_list = list(tensor_list)
        self._worker_num = worker_num
        self._bits, self._threshold_value, self._threshold_bin = {}, {}, {}

    @property
    def bits(self):
        return self._bits

    @property
    def threshold_value(self):
        return self._threshold_value

    @property
    def threshold_bins(self):
        return self._threshold_bin

    def quantize(self, distributions, distribution_intervals):
        names, w = self._tensor_list, max(int(self._worker_num), 1)
        per = len(names) // w
        order = []
        for i in range(w):                       # the reference's per-worker result order
            order += names[i * per:(i + 1) * per] + (names[w * per:] if i == 0 else [])
        for n in order:
            t = orc.kl_threshold(orc.normalize(np.asarray(distributions[n])))
            tb = (t + 0.5) * distribution_intervals[n]
            self._threshold_bin[n] = t
            self._threshold_value[n] = tb
            self._bits[n] = int(8 - 1 - math.ceil(math.log(tb, 2)))


class OracleChannelCollector(StatCollectives):
    """Stand-in for common.quantity.channel_collector.ChannelCollector (one row per (tensor, channel)): same methods, the
    oracle's abs-max / histogram / KL on the channel slices.  Like the per-tensor double it only SAYS where its state is;
    the all-reduces are the product's."""

    def __init__(self, channels, statistic=1, device=None):
        self._names = list(channels.keys())
        self._channels = dict(channels)
        self._first, row = {}, 0
        for n in self._names:
            self._first[n] = row
            row += int(channels[n])
        self._rows = row
        self._statistic = statistic
        self._max = np.zeros(row, dtype=np.float32)
        # (W equal row blocks when distributed: what the product's reduce-scatter takes, StatCollectives.padded_rows)
        self._hist_padded = np.zeros((self.padded_rows(row), BINS), dtype=np.int64)
        self._hist = self._hist_padded[:row]
        self._interval = None
        self._own_block = None

    @property
    def rows(self):
        return self._rows

    def row_range(self, name):
        return self._first[name], self._first[name] + self._channels[name]

    def _slices(self, tensors):
        for n in self._names:
            if n not in tensors:
                continue
            t = tensors[n].detach().cpu().numpy().astype(np.float32, copy=False)
            assert t.shape[1] == self._channels[n]
            for c in range(t.shape[1]):
                yield self._first[n] + c, np.ascontiguousarray(t[:, c]).ravel()

    def refresh_max_val(self, tensors):
        for row, x in self._slices(tensors):
            self._max[row] = orc.absmax(x, self._max[row])

    def intervals(self):
        self._interval = (self._statistic * self._max / BINS + 1e-12).astype(np.float32, copy=False)
        return self._interval

    def add_to_distributions(self, tensors):
        if self._interval is None:
            self.intervals()
        for row, x in self._slices(tensors):
            orc.hist2048(x, np.float32(self._interval[row]), self._hist[row])

    def _stat_tensors(self):
        return torch.from_numpy(self._max), torch.from_numpy(self._hist_padded)

    def _row_result(self, hist_row, interval):
        t = orc.kl_threshold(orc.normalize(hist_row))
        tb = (t + 0.5) * interval
        return t, int(8 - 1 - math.ceil(math.log(tb, 2)))

    def quantize(self):
        """Same protocol as the product's ChannelCollector.quantize: all rows here when not distributed; after
        reduce_scatter_hist() (the product's line) this rank's block only, then the product's all-gather of (threshold, bits)."""
        if self._own_block is None:
            res = [self._row_result(self._hist[r], self._interval[r]) for r in range(self._rows)]
            thr, all_bits = [a for a, _b in res], [b for _a, b in res]
        else:
            lo, block = self._own_block
            block = block.numpy()
            local = np.zeros((2, block.shape[0]), dtype=np.int32)
            for j in range(block.shape[0]):
                if lo + j < self._rows:
                    local[0, j], local[1, j] = self._row_result(block[j], self._interval[lo + j])
            both = self.all_gather_rows(torch.from_numpy(local)).numpy()
            thr, all_bits = both[0, :self._rows].tolist(), both[1, :self._rows].tolist()
        self.threshold_bins = np.asarray(thr, dtype=np.int32)
        bits = {}
        for n in self._names:
            lo, hi = self.row_range(n)
            bits[n] = [int(b) for b in all_bits[lo:hi]]
        return bits
