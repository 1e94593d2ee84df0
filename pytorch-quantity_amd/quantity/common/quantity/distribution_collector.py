"""Running abs-max and 2048-bin |x| histograms of named tensors, accumulated over calibration
batches -- on the GPU, state resident in HBM.

Drop-in for reference quantity/common/quantity/distribution_collector.py
(DistributionCollector :7, refresh_max_val :70-78, distribution_intervals :52-63,
add_to_distributions :80-119, _add_to_distribution :127-135): same constructor, methods and
properties.  What changed underneath:

  * tensors stay on the device: one segmented HIP launch (fq_absmax_seg / fq_hist2048_seg) covers
    every tensor of a batch; the reference copies each tensor to the host, forks a
    multiprocessing.Pool per call and walks the elements in a Python loop.
  * state is `fp32[T]` maxima and `int64[T][2048]` histograms in HBM (the reference keeps int32
    NumPy arrays and wraps past 2^31-1 per bin); they are only downloaded when a property is read.
  * `worker_num` is accepted and ignored (there is no process pool).

NumPy arrays are accepted too (they are uploaded); there is no CPU compute path.
"""
import numpy as np
import torch

from . import _native
from ._collectives import StatCollectives

__all__ = ["DistributionCollector"]


def _as_device_f32(t, device):
    if isinstance(t, torch.Tensor):
        if t.device.type != "cuda":
            t = t.to(device)
        return t.detach() if t.dtype == torch.float32 else t.detach().float()
    a = np.ascontiguousarray(np.asarray(t), dtype=np.float32)
    return torch.from_numpy(a).to(device)


class DistributionCollector(StatCollectives):

    def __init__(self, tensor_list, interval_num=2048, statistic=1, worker_num=1, debug=False, device=None):
        if interval_num not in _native.SUPPORTED_BINS:
            raise ValueError("the MI355X histogram / KL kernels are built for INTERVAL_NUM in %r, got %r"
                             % (_native.SUPPORTED_BINS, interval_num))
        self._tensor_list = list(tensor_list)
        self._row = {name: i for i, name in enumerate(self._tensor_list)}
        self._interval_num = interval_num
        self._statistic = statistic
        self._worker_num = worker_num
        self._debug = debug
        self._device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        T = len(self._tensor_list)
        self._max_dev = torch.zeros(T, dtype=torch.float32, device=self._device)
        self._hist_dev = torch.zeros(T, interval_num, dtype=torch.int64, device=self._device)
        self._interval_dev = None
        self._interval_key = None
        self._max_vals_refreshed_flag = False
        self._added_to_distributions_flag = False
        self._keepalive = None

    # ------------------------------------------------------------------ device-side accessors
    @property
    def device(self):
        return self._device

    @property
    def max_device(self):
        """fp32[T] running abs-max on the device (row order = tensor_list order)."""
        return self._max_dev

    @property
    def hist_device(self):
        """int64[T, 2048] histograms on the device."""
        return self._hist_dev

    def row_of(self, name):
        return self._row[name]

    # ------------------------------------------------------------------ reference API
    @property
    def max_vals(self):
        assert self._max_vals_refreshed_flag, "Please use refresh_max_val() first."
        host = self._max_dev.cpu().numpy()
        # the reference's running max starts as the Python int 0 and only becomes np.float32 once a
        # positive maximum is seen (distribution_collector.py:42,:78)
        return {n: (host[i] if host[i] > 0 else 0) for i, n in enumerate(self._tensor_list)}

    @property
    def distribution_intervals(self):
        """Bin width per tensor: statistic * max / 2048 + 1e-12, evaluated with NumPy scalars so the
        types and roundings are the reference's (np.float32 throughout; the Python float 1e-12 for
        a tensor that never left zero).  As in the reference the returned dict is kept (and may be
        edited by the caller, e.g. merge groups) and is what add_to_distributions() bins with."""
        assert self._max_vals_refreshed_flag, "Please use refresh_max_val() first."
        mv = self.max_vals
        intervals = {}
        for name in self._tensor_list:
            intervals[name] = self._statistic * mv[name] / self._interval_num + 1e-12
        self._distribution_intervals = intervals
        return intervals

    @property
    def distributions(self):
        assert self._added_to_distributions_flag, "Please use add_to_distributions() first."
        host = self._hist_dev.cpu().numpy()
        narrow = host.max(initial=0) <= np.iinfo(np.int32).max
        return {n: (host[i].astype(np.int32) if narrow else host[i].copy()) for i, n in enumerate(self._tensor_list)}

    def refresh_max_val(self, tensors):
        """Fold one batch into the running abs-max of every tensor in tensor_list."""
        self._max_vals_refreshed_flag = True
        rows = self._rows_of(tensors)
        segs = [_as_device_f32(tensors[self._tensor_list[r]], self._device) for r in rows]
        self._keepalive = _native.absmax_seg(segs, rows, self._max_dev)

    def add_to_distributions(self, tensors):
        """Add one batch to every tensor's histogram, binning with the current intervals."""
        if self._debug and self._added_to_distributions_flag:
            return
        self._added_to_distributions_flag = True
        if not hasattr(self, "_distribution_intervals"):
            print("interval:", self.distribution_intervals)
        self._sync_intervals()
        rows = self._rows_of(tensors)
        segs = [_as_device_f32(tensors[self._tensor_list[r]], self._device) for r in rows]
        self._keepalive = _native.hist2048_seg(segs, rows, self._interval_dev, self._hist_dev)

    # ------------------------------------------------------------------ beyond the reference API
    @property
    def supports_pairs(self):
        """add_pairs_to_distributions(): a tensor and a sum of two tensors in one pass (the 2048-bin kernel only)."""
        return self._interval_num == _native.BINS

    @property
    def fused_hist_ok(self):
        """May the producers histogram their own output (fq_*_hist_f32)?  Those epilogues exist for INTERVAL_NUM = 2048 only;
        with another bin count pass 2 takes every histogram through fq_hist_seg_n."""
        return self._interval_num == _native.BINS

    def add_pairs_to_distributions(self, pairs):
        """pairs: [(a, b, name of a's row or None, name of the sum's row[, relu_out])].  a is counted into its row and a + b -- the
        fp32 addition an Eltwise performs (fabu_layer.py:5-11) -- into the sum's row, exactly as add_to_distributions() would count
        the two stored tensors; the sum itself is never written; relu_out (optional, a tensor like a) receives max(a + b, 0), the
        next block's shortcut (fq_hist2048_pair_seg)."""
        if not pairs:
            return
        self._added_to_distributions_flag = True
        if not hasattr(self, "_distribution_intervals"):
            self.distribution_intervals
        self._sync_intervals()
        a = [_as_device_f32(p[0], self._device) for p in pairs]
        b = [_as_device_f32(p[1], self._device) for p in pairs]
        rows_a = [None if p[2] is None else self.row_of(p[2]) for p in pairs]
        rows_s = [self.row_of(p[3]) for p in pairs]
        relus = [p[4] if len(p) > 4 else None for p in pairs]
        self._keepalive_pairs = _native.hist2048_pair_seg(a, b, rows_a, rows_s, self._interval_dev, self._hist_dev, relus)

    def add_chains_to_distributions(self, chains):
        """chains: [(head, [(y_k, name of y_k's row or None, name of S_k's row), ...])]: a stage of residual blocks whose sums pass
        1 did not write -- S_1 = y_1 + head, S_k = y_k + relu(S_(k-1)) -- counted in one pass over the conv3 outputs and the one
        shortcut (fq_hist2048_chain_seg); nothing is written."""
        if not chains:
            return
        self._added_to_distributions_flag = True
        if not hasattr(self, "_distribution_intervals"):
            self.distribution_intervals
        self._sync_intervals()
        jobs = []
        for head, blocks in chains:
            jobs.append((_as_device_f32(head, self._device), [_as_device_f32(b[0], self._device) for b in blocks],
                         [None if b[1] is None else self.row_of(b[1]) for b in blocks], [self.row_of(b[2]) for b in blocks]))
        self._keepalive_chains = _native.hist2048_chain_seg(jobs, self._interval_dev, self._hist_dev)

    supports_partial = True     # refresh_max_val / add_to_distributions accept a dict holding only SOME of the tensors

    def row_of(self, name):
        """Row of one tensor in max_device / hist_device (for kernels that fold a statistic into their own pass)."""
        if not hasattr(self, "_row_index"):
            self._row_index = {n: r for r, n in enumerate(self._tensor_list)}
        return self._row_index[name]

    def note_max_refreshed(self):
        """fq_bias_add_absmax_f32 wrote a maximum straight into max_device."""
        self._max_vals_refreshed_flag = True

    def prepare_distributions(self):
        """Upload the current bin widths (distribution_intervals, as edited by the caller) so that producer kernels can
        histogram their own output (fq_bias_add_hist_f32 / fq_add_hist_f32) straight into hist_device."""
        if not hasattr(self, "_distribution_intervals"):
            self.distribution_intervals
        self._sync_intervals()
        self._added_to_distributions_flag = True

    @property
    def interval_device(self):
        """fp32[T] bin widths on the device (valid after prepare_distributions() / add_to_distributions())."""
        return self._interval_dev

    def _rows_of(self, tensors):
        """Rows of the tensors present in `tensors` (the reference passes all of them; the calibration loop also feeds
        them in groups, from inside the forward hooks, while they are still in the Infinity Cache)."""
        if len(tensors) == len(self._tensor_list):
            return list(range(len(self._tensor_list)))
        return [r for r, n in enumerate(self._tensor_list) if n in tensors]

    # all_reduce_max() / all_reduce_hist(): StatCollectives (one MAX all-reduce of fp32[T], one SUM all-reduce of the
    # flat int64[T*2048] buffer; RCCL over xGMI when the process group is 'nccl')
    def _stat_tensors(self):
        return self._max_dev, self._hist_dev

    def _note_max_reduced(self):
        self._max_vals_refreshed_flag = True        # a rank that owned no batch still holds the global maxima

    def _note_hist_reduced(self):
        self._added_to_distributions_flag = True

    def merged_distributions(self, groups):
        """int64[T, 2048] device tensor in tensor_list order in which every group (a list of tensor
        names) holds the sum of its members' histograms (reference pytorch_quantizer.py:432-445,
        applied group after group)."""
        assert self._added_to_distributions_flag, "Please use add_to_distributions() first."
        merged = self._hist_dev.clone()
        for group in groups:
            idx = torch.tensor([self._row[n] for n in group], dtype=torch.long, device=self._device)
            merged[idx] = merged[idx].sum(dim=0, keepdim=True)
        return merged

    def quantize_param(self, tensor, bit):
        """clip(rint(w * 2^bit), -128, 127) as an int32 ndarray of the tensor's shape
        (reference pytorch_quantizer.py:656-657,:663)."""
        t = _as_device_f32(tensor, self._device)
        return _native.quantize_param_i32(t, bit).cpu().numpy()

    # ------------------------------------------------------------------ internals
    def _sync_intervals(self):
        vals = [np.float32(self._distribution_intervals[n]) for n in self._tensor_list]
        key = tuple(v.tobytes() for v in vals)
        if key != self._interval_key:
            host = np.array(vals, dtype=np.float32)
            self._interval_dev = torch.from_numpy(host).to(self._device)
            self._interval_key = key

    def set_intervals(self, intervals):
        """Install an interval dict (name -> scalar) without going through the property."""
        self._distribution_intervals = intervals
