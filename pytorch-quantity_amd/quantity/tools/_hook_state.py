"""What the forward hooks, the patched module forwards and the calibration loop of tools.pytorch_quantizer tell each other
during one calibration: the statistics gatherer of a forward (_EagerStats), the proof that a tensor may stay un-materialised
(_DeferralProbe) and the shared state itself (_HookState).  No reference counterpart: the reference's hook copies every output
to the host (pytorch_quantizer.py:513) and needs none of it."""
from collections import OrderedDict

import torch

__all__ = ["_AFTER_FORWARD", "_EagerStats", "_DeferralProbe", "_HookState"]

_AFTER_FORWARD = 1 << 62      # _EagerStats limit that is never reached: one flush, after the forward


class _EagerStats(object):
    """Statistics launches from INSIDE the forward hooks.

    The reference copies every hooked output to the host inside the hook (pytorch_quantizer.py:513), i.e. it sees the
    value the module returned -- also when a later in-place op (nn.ReLU(inplace=True)) overwrites that tensor.  Taking
    the statistics after the whole forward would see the overwritten values, so hooked tensors are handed to `fn`
    (collector.refresh_max_val / add_to_distributions on a partial dict) from the hook itself:
      limit = 0      one launch per tensor, before any later module can touch it (always correct);
      limit = L > 0  tensors are gathered until L bytes are pending (_AFTER_FORWARD: all of them, one launch per
                     forward -- the fast path: one balanced launch streams at 6+ TB/s, 36 us launches do not).
                     Only valid for models without in-place consumers, which the first forward establishes
                     (`modified` below); every flush re-checks the version counters and refuses silently wrong data."""

    def __init__(self, fn, limit):
        self.fn, self.limit = fn, int(limit)
        self.pending, self.bytes = OrderedDict(), 0
        self.seen = []                     # (key, tensor, version at capture) of the whole forward

    def add(self, key, t):
        self.pending[key] = t
        self.seen.append((key, t, t._version))
        self.bytes += t.numel() * t.element_size()
        if self.bytes >= self.limit:
            self.flush()

    retain = True      # False: tensors served by their producers are not kept alive by this object (no cache wanted)

    def note(self, key, t):
        """A tensor whose statistics were already taken by its producer: only watched for in-place consumers (the first
        forward of a calibration decides that; later forwards with `retain` off do not even hold a reference)."""
        if self.retain:
            self.seen.append((key, t, t._version))

    def flush(self, extra=None):
        """Hand the pending tensors (plus `extra`: tensors kept from an earlier forward) to fn in one call."""
        if self.limit > 0:
            for (k, t, v) in self.seen:
                if k in self.pending and t._version != v:
                    raise RuntimeError("a hooked activation was modified in place before its statistics were taken; "
                                       "set Quantity.stats_group_bytes = 0")
        if extra:
            self.pending.update(extra)
        if not self.pending:
            return
        self.fn(self.pending)
        self.pending, self.bytes = OrderedDict(), 0

    def modified(self):
        """True if any tensor captured during this forward has been written to since (an in-place consumer)."""
        return any(t._version != v for (_k, t, v) in self.seen)


class _DeferralProbe(object):
    """The proof that a convolution's output may stay un-materialised until the Eltwise that adds it (Quantity.fuse_conv_add).

    The reference's model code is arbitrary Python: `y = self.conv3(x)` may be read by anything before `self.Eltwise(y, r)`.
    Two probe forwards on the same random input settle it for the model at hand.  The first (mode "learn", the in-place
    probe that runs anyway) records which own 1x1 convolution's output OBJECT arrives at which Eltwise, and which nn.ReLU
    consumes that Eltwise's output.  The second (mode "poison") hands the model a NaN-filled tensor in place of each such
    convolution output and of each such sum, while the real values travel privately from the convolution to its Eltwise to
    its ReLU.  If every hooked tensor outside those pairs and the model's output come out bit for bit as in the first forward,
    nothing but the designated Eltwise read the convolution's output and nothing but the designated ReLU read the sum: NaN
    poisons everything it touches.  (Control flow that depends on the data is not covered by a probe; the production path
    therefore also refuses to end a forward with a convolution still waiting.)
    What poisoning cannot see is a KEEPER -- code that stores the tensor and reads it after the forward; `holders()` finds those
    after the learning forward (whoever still refers to the tensor then is not a frame of the model), and a chain with one is
    not taken.
    The same forward proves the simpler chain convolution -> out-of-place nn.ReLU (`relu_only`): when nothing but that ReLU reads
    a convolution's output, and pass 2 does not want the tensor kept, the kernel writes the ReLU's result only."""

    def __init__(self):
        self.mode = "learn"
        self.conv_out = {}          # learn: id(output) -> (output, conv module)
        self.pairs = {}             # learn: Eltwise module -> the conv module whose output it received
        self.keys = {}              # module -> its hook key of this forward
        self.outputs = {}           # learn: module -> the output OBJECT its forward returned (what model code may have kept)
        self.candidates = {}        # conv module -> (Eltwise module, nn.ReLU module)
        self.relu_only = {}         # conv module -> the nn.ReLU that is the only reader of its output (skip_unread_outputs)
        self.private = {}           # poison: id(poisoned tensor) -> (poisoned tensor, real tensor, the one module that may read it)
        self.others = {}            # learn: Eltwise module -> (its OTHER operand -- the shortcut --, that tensor's version at the add)

    def conv_done(self, m, y):
        if self.mode == "learn":
            self.conv_out[id(y)] = (y, m)
            return y
        pair = self.candidates.get(m)
        reader = pair[0] if pair is not None else self.relu_only.get(m)
        if reader is None:
            return y
        bad = torch.full_like(y, float("nan"))
        self.private[id(bad)] = (bad, y, reader)
        return bad

    def real(self, t, reader):
        e = self.private.get(id(t)) if torch.is_tensor(t) else None
        return e[1] if e is not None and e[0] is t and e[2] is reader else None

    def eltwise(self, m, x, y):
        """The Eltwise's result for this probe forward, or None (its own forward runs)."""
        if self.mode == "learn":
            for t in (x, y):
                e = self.conv_out.get(id(t)) if torch.is_tensor(t) else None
                if e is not None and e[0] is t:
                    self.pairs.setdefault(m, e[1])
                    other = y if t is x else x
                    if torch.is_tensor(other) and other is not t:
                        self.others.setdefault(m, (other, other._version))
                    break
            return None
        rx, ry = self.real(x, m), self.real(y, m)
        if rx is None and ry is None:
            return None
        conv = self.pairs[m]
        s = torch.add(x if rx is None else rx, y if ry is None else ry)
        bad = torch.full_like(s, float("nan"))
        self.private[id(bad)] = (bad, s, self.candidates[conv][1])
        return bad

    def relu(self, m, x):
        r = self.real(x, m) if self.mode == "poison" else None
        return None if r is None else torch.nn.functional.relu(r)

    @staticmethod
    def holders(tensors, ours):
        """Who, besides this calibration, still holds one of `tensors` once the learning forward has RETURNED: {id(t): [what]}.
        NaN poisoning finds every reader whose result reaches a hooked tensor or the model's output; it cannot find one that merely
        KEEPS the tensor -- `self.feat = y`, a list a user hook appends to, a view parked somewhere -- and reads it after the
        forward: in production that reader would hold memory no kernel ever wrote.  After the forward the model's frames are
        gone, so whatever still refers to the tensor object is either one of `ours` (containers of this calibration, by
        identity) or such a keeper.
          * Python references: ONE gc.get_referrers pass over the heap for all tensors together (a pass per tensor cost 0.4 s of
            a 0.5 s calibration: the call walks every tracked object), then gc.get_referents of the few objects it found;
          * views: the TensorImpl's use count (a view keeps its base alive).
        NOT found: a detach() / .data alias (another tensor object on the same storage; torch exposes no dependable count of a
        storage's users -- torch._C._storage_Use_Count moved by itself between two looks at an untouched tensor on this stack)."""
        import gc
        import types
        tensors = [t for t in tensors if torch.is_tensor(t)]
        found = {}
        if not tensors:
            return found
        watched = dict((id(t), t) for t in tensors)
        mine = set(id(o) for o in ours)
        mine.add(id(tensors))
        mine.add(id(watched))
        refs = gc.get_referrers(*tensors)
        mine.add(id(refs))
        for r in refs:
            if id(r) in mine or isinstance(r, types.FrameType):
                continue
            if isinstance(r, tuple) and any(id(rr) in mine for rr in gc.get_referrers(r)):
                continue                                    # a tuple inside one of our containers
            for o in gc.get_referents(r):
                if id(o) in watched:
                    found.setdefault(id(o), []).append("a %s" % type(r).__name__)
        for t in tensors:
            if t._use_count() > 1:
                found.setdefault(id(t), []).append("a view of it")
        return found

    def poisoned_keys(self):
        mods = set(self.candidates) | set(e for e, _r in self.candidates.values()) | set(self.relu_only)
        return set(k for mod, k in self.keys.items() if mod in mods)


class _HookState(object):
    """What the forward hooks, the patched forwards (_patch_fused_convs) and the calibration loop tell each other during
    ONE calibration (created by regist_hook_outfeature, dropped with the hooks).  Three groups:

    the pass          which statistic the producers fold in (`fuse_stat`), into which engine (`fuse_collector`; None: no
                      producer fusion), whether it was switched off by a failed check (`fuse_off`), which modules passed
                      theirs (`fuse_verified`; `fuse_warm`: library convolutions that have had their first, unfused call),
                      and what the cache wants kept (`keep_feats`, `keep_names`);
    the forward       where it ends early (`stop_after`), its time stamps for the cache plan (`events`), the statistics
                      gatherer of this forward (`eager`), the plain-kernel mode of the per-channel path (`own_plain`);
    module to module  work a patched forward left for the hook of the very same call (`fuse_bias`), the last hooked output
                      (`last_out`) and which ReLU consumes which producer (`relu_after`), a ReLU result a producer has already
                      written (`relu_ready`), a convolution waiting for the Eltwise that consumes it (`deferred`, `defer_ok`).
    The rest are counters for Quantity.timings."""
    __slots__ = ("stop_after", "events", "eager", "fuse_bias", "fuse_collector", "fuse_off", "fuse_verified", "fuse_warm",
                 "relu_after", "relu_ready", "last_out", "fused_relus", "fuse_stat", "hist_fused", "keep_feats", "keep_names",
                 "own_plain", "own_conv1x1", "deferred", "defer_ok", "deferred_adds", "deferred_hists", "poison", "relu_only_ok", "skipped_outputs",
                 "pair_ok", "pairs", "pair_sums", "sum_relu")

    def __init__(self):
        self.stop_after = None          # ordinal of the last module pass 2 needs (the hook raises _StopForward there)
        self.events = None              # [(ordinal, event)] while the first forward of pass 1 is being timed
        self.eager = None               # _EagerStats of the running forward
        self.fuse_bias = None           # (module, operands): the hook of this very call finishes the module's work
        self.fuse_collector = None
        self.fuse_off = False
        self.fuse_verified = set()
        self.fuse_warm = set()
        self.relu_after = {}            # producer module -> the out-of-place nn.ReLU that consumed its output last time
        self.relu_ready = None          # (producer's output, its ReLU, the nn.ReLU module, version): served by the patched ReLU
        self.last_out = None
        self.fused_relus = set()
        self.fuse_stat = "max"          # "max": pass 1, "hist": pass 2
        self.hist_fused = 0
        self.keep_feats = True          # False: nothing will be cached, producers' tensors need not outlive their hook
        self.keep_names = None          # names of the tensors pass 2 wants kept from THIS forward (None: all of them)
        self.own_plain = False          # per-channel calibration: convolutions on the own kernels without statistics
        self.own_conv1x1 = 0
        self.deferred = {}              # id(output) -> (output, conv, x, key, row, version): convolutions whose kernel has not
        #                                 run yet -- each runs inside the launch of the Eltwise that consumes it
        self.defer_ok = {}              # conv module -> its Eltwise, proven by the poison probe (_prove_deferral)
        self.deferred_adds = 0          # launches of fq_conv1x1_add_f32 (pass 1)
        self.deferred_hists = 0         # launches of fq_conv1x1_add_hist_f32 (pass 2)
        self.poison = None              # the _DeferralProbe of a running probe forward
        self.relu_only_ok = set()       # convolutions whose output only their nn.ReLU reads (same proof)
        self.skipped_outputs = 0        # launches that did not write the convolution's own output
        self.pair_ok = set()            # Eltwise modules whose shortcut operand nobody writes to after the add (probe forward)
        self.pairs = {}                 # THIS forward: key of a sum that was not written -> (key of conv3's output, the shortcut
        #                                 tensor, its version): pass 2 histograms the pair (fq_hist2048_pair_seg)
        self.pair_sums = 0              # sums pass 1 did not write because pass 2 histograms (conv3 output, shortcut) instead
        self.sum_relu = {}              # THIS forward: id(r) -> (weak reference to r, key of the sum): the ReLU outputs the one-kernel
        #                                 tails wrote -- a pair whose shortcut is one of them can re-make it in pass 2
