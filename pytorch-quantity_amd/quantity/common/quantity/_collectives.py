"""The two collectives of data-parallel calibration (no reference counterpart: the
reference is single process): after pass 1 one MAX all-reduce of the fp32[rows] running maxima, after
pass 2 one SUM all-reduce of the flat int64[rows * 2048] histograms.  With the process group 'nccl' that
is RCCL over xGMI on the device buffers themselves (284 B and 1.16 MB for ResNet-50: latency bound, one
call each); with 'gloo' the same lines run on host tensors.  Max and integer sum are order independent,
so the tables are bit-identical for any number of ranks.

The per-(tensor, channel) rows (ChannelCollector: 42 667 rows x 2048 bins x 8 B = 699 MB for ResNet-50) are exchanged SHARDED
instead ("tensors sharded by row + all_gather of bits"): rank r owns the contiguous row block [r*S, (r+1)*S),
S = ceil(rows / W); after pass 2 ONE reduce-scatter (SUM) hands every rank the global histograms of its own block only --
a ring moves (W-1)/W x 699 MB per rank, half of what the all-reduce of the whole buffer moves, and no rank ever holds or
reduces the full buffer -- each rank runs the KL sweep on its S rows (47 ms -> 6 ms at W = 8), and ONE all-gather of
int32[2][S] (threshold bin, bits: 2 x 21 KB per rank at W = 8) gives every rank every row's result.  The block is
contiguous rather than `row mod W` because that is the layout a reduce-scatter delivers without a permuting copy of the
699 MB; the sweep's cost per row does not depend on the row, so the blocks are balanced either way.

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

    # ---- the row-sharded exchange of the per-channel rows ---------------------------------------------------------
    @staticmethod
    def row_shard(rows, rank=None, world=None):
        """(S, lo, hi): block size and this rank's rows [lo, hi) of `rows` (clipped; a rank beyond the rows owns none).
        Not distributed: the one rank owns everything."""
        import torch.distributed as dist
        if world is None:
            on = dist.is_available() and dist.is_initialized()
            rank, world = (dist.get_rank(), dist.get_world_size()) if on else (0, 1)
        per = -(-int(rows) // world)
        return per, min(rank * per, rows), min((rank + 1) * per, rows)

    @staticmethod
    def padded_rows(rows):
        """Rows to allocate so that the histogram buffer is W equal blocks (what a reduce-scatter takes)."""
        import torch.distributed as dist
        world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        return -(-int(rows) // world) * world

    def reduce_scatter_hist(self):
        """After pass 2: this rank's row block summed over all ranks, one call.  `_stat_tensors()[1]` must be the PADDED buffer
        ([W * S][bins]).  Returns (lo, block): block is [S][bins], rows lo .. lo + S - 1 (rows beyond the last real one are zero).
        The collector's own buffer keeps this rank's LOCAL counts; nothing reads it afterwards."""
        import torch
        import torch.distributed as dist
        _mx, hist = self._stat_tensors()
        world, rank = dist.get_world_size(), dist.get_rank()
        assert hist.shape[0] % world == 0, "the histogram buffer must be padded to W equal row blocks (padded_rows)"
        per = hist.shape[0] // world
        block = torch.empty((per, hist.shape[1]), dtype=hist.dtype, device=hist.device)
        dist.reduce_scatter_tensor(block, hist, op=dist.ReduceOp.SUM)
        self._own_block = (rank * per, block)
        return self._own_block

    @staticmethod
    def all_gather_rows(local):
        """local: int32 [k][S] (this rank's values for its row block) -> int32 [k][W * S] on every rank, one call."""
        import torch
        import torch.distributed as dist
        world = dist.get_world_size()
        local = local.contiguous()
        out = torch.empty(world * local.numel(), dtype=local.dtype, device=local.device)        # (flat: the form every backend takes)
        dist.all_gather_into_tensor(out, local.view(-1))
        return out.view(world, local.shape[0], local.shape[1]).permute(1, 0, 2).reshape(local.shape[0], -1)
