"""The two collectives of data-parallel calibration (no reference counterpart: the
reference is single process): after pass 1 one MAX all-reduce of the fp32[rows] running maxima, after
pass 2 one SUM all-reduce of the flat int64[rows * 2048] histograms.  With the process group 'nccl' that
is RCCL over xGMI on the device buffers themselves (284 B and 1.16 MB for ResNet-50: latency bound, one
call each); with 'gloo' the same lines run on host tensors.  Max and integer sum are order independent,
so the tables are bit-identical for any number of ranks.

One mixin, used by DistributionCollector, ChannelCollector and by the CPU test double
(tests/engine_doubles.py), so the world_size-2 gloo tests execute the product's own lines.
"""

__all__ = ["StatCollectives"]


class StatCollectives(object):

    def _stat_tensors(self):
        """(maxima, histograms): torch tensors that ALIAS the collector's state (reduced in place)."""
        raise NotImplementedError

    def _note_max_reduced(self):
        """A rank that owned no batch holds the global maxima from here on."""

    def _note_hist_reduced(self):
        pass

    def all_reduce_max(self):
        import torch.distributed as dist
        mx, _hist = self._stat_tensors()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        self._note_max_reduced()

    def all_reduce_hist(self):
        import torch.distributed as dist
        _mx, hist = self._stat_tensors()
        dist.all_reduce(hist.view(-1), op=dist.ReduceOp.SUM)          # one call on the flat buffer
        self._note_hist_reduced()
