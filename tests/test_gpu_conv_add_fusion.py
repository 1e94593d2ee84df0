"""Calibration pass 1 with the last 1x1 convolution of a residual block, the Eltwise that adds it to the shortcut and the ReLU
behind it as ONE kernel (Quantity.fuse_conv_add, fq_conv1x1_add_f32): which chains are taken is PROVEN per model by the poison
probe (tools.pytorch_quantizer._DeferralProbe), the tables do not depend on it, and whether the two intermediate tensors reach
HBM follows what pass 2's cache wants.  Reference: the forward at pytorch_quantizer.py:288-296 through fabu_layer.py:5-11."""
import os

import pytest
import torch

import cases
from workdir_util import product_workdir

pytestmark = pytest.mark.gpu


def _block_net(variant="plain", hw=64):
    """stem -> 2 x [1x1 -> ReLU -> 3x3 -> ReLU -> 1x1 (128 channels)] + shortcut -> Eltwise -> ReLU -> pool -> fc.
    variant "peek": block 1's convolution output is ALSO read by something else before its Eltwise;
    variant "inplace_relu": the ReLU behind the Eltwise works in place."""
    from torch import nn
    from common.quantity import Eltwise, View

    class Block(nn.Module):
        def __init__(self, cin, mid, cout, stride, peek):
            super().__init__()
            self.c1, self.r1 = nn.Conv2d(cin, mid, 1), nn.ReLU()
            self.c2, self.r2 = nn.Conv2d(mid, mid, 3, stride=stride, padding=1), nn.ReLU()
            self.c3 = nn.Conv2d(mid, cout, 1)
            self.down = nn.Conv2d(cin, cout, 1, stride=stride) if (stride != 1 or cin != cout) else None
            self.add = Eltwise()
            self.r3 = nn.ReLU(variant == "inplace_relu")
            self.peek = (nn.ReLU(), Eltwise()) if peek else None
            if peek:
                self.peek_relu, self.peek_add = self.peek

        def forward(self, x):
            y = self.c3(self.r2(self.c2(self.r1(self.c1(x)))))
            side = self.peek_relu(y) if self.peek else None                    # a second reader of the convolution's output
            z = self.r3(self.add(y, x if self.down is None else self.down(x)))
            return z if side is None else self.peek_add(z, side)

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.stem, self.relu0 = nn.Conv2d(3, 32, 3, padding=1), nn.ReLU()
            self.b1 = Block(32, 16, 128, 2, peek=False)
            self.b2 = Block(128, 32, 128, 1, peek=variant == "peek")
            self.pool, self.view, self.fc = nn.AvgPool2d(hw // 2), View(), nn.Linear(128, 10)

        def forward(self, x):
            return self.fc(self.view(self.pool(self.b2(self.b1(self.relu0(self.stem(x)))))))
    return Net()


def _calibrate(model, fuse, cache_gb=None, plan="", batches=5, monkeypatch=None, hw=64, skip=None, pair=None, chain=None):
    from tools import Quantity
    if monkeypatch is not None and cache_gb is not None:
        monkeypatch.setenv("FQ_ACT_CACHE_GB", cache_gb)
        monkeypatch.setenv("FQ_CACHE_PLAN", plan)
    with product_workdir(input_shape="1,3,%d,%d" % (hw, hw), device="gpu", max_cali_img_num=batches - 1) as tmp:
        q = Quantity(model)
        q.fuse_conv_add = fuse
        q.skip_unread_outputs = fuse if skip is None else skip
        if pair is not None:
            q.pair_hist = pair
        if chain is not None:
            q.pair_chain = chain
        bits = q.activation_quantize(cases.calib_batches(batches, (8, 3, hw, hw), seed=91))
        table = open(os.path.join(tmp, "test", "workdir", "feat.table")).read()
        return dict(bits), table, dict(q._collector.max_vals), q._collector.hist_device.clone(), dict(q.timings)


@pytest.mark.parametrize("cache_gb,plan", [("0", ""), ("1", "A"), ("0.07", "A"), ("1", "B"), ("0.05", "B")])
def test_tables_do_not_depend_on_the_fusion_nor_on_what_the_cache_keeps(monkeypatch, cache_gb, plan):
    """32 x 32 planes behind the blocks: the one-kernel form is taken whatever the cache keeps -- nothing (no cache), both
    tensors of every batch (1 GB), whole batches until the budget is spent (plan A), the deepest tensors of every batch (B)."""
    model = cases.seed_model(_block_net(), base_seed=8).eval().cuda()
    want = _calibrate(model, False, cache_gb, plan, monkeypatch=monkeypatch)
    got = _calibrate(model, True, cache_gb, plan, monkeypatch=monkeypatch)
    assert got[1] == want[1] and got[0] == want[0]
    assert got[2] == want[2]                                   # every abs-max, bit for bit
    assert torch.equal(got[3], want[3])                        # every histogram
    assert want[4]["conv_add_chains_proven"] == 0 and want[4]["conv_add_launches"] == 0
    assert got[4]["conv_add_chains_proven"] == 2               # both blocks' (c3, add, r3)
    # every forward of pass 1 takes both chains, except the one in which the Eltwise modules have their first, checked use
    assert got[4]["conv_add_launches"] in (2 * 4, 2 * 5)
    # pass 2: both chains of every forward it re-runs in full (no cache: all five; plan A with 70 MB: the batches that did not
    # fit; everything kept: no forward at all); plan B re-runs a prefix, which may end inside a block
    redo = {("0", ""): (10,), ("1", "A"): (0,), ("0.07", "A"): (6,), ("1", "B"): (0,), ("0.05", "B"): (0, 5, 10)}[(cache_gb, plan)]
    assert got[4]["conv_add_hist_launches"] in redo, got[4]
    assert want[4]["conv_add_hist_launches"] == 0
    if plan:
        assert got[4]["cache_plan"]["kind"] == plan and got[4]["cache_bytes"] > 0
        assert (got[4]["cache_bytes"] < 100e6) == (cache_gb != "1")      # (a batch's hooked tensors are 30 MB: partial caches are partial)


@pytest.mark.parametrize("cache_gb,plan", [("0", ""), ("1", "A"), ("0.05", "B")])
def test_convolution_outputs_only_their_relu_reads_are_not_written(monkeypatch, cache_gb, plan):
    """skip_unread_outputs alone (the conv + Eltwise fusion off): the two inner convolutions of each block run on the own kernels
    and are followed by an out-of-place nn.ReLU that the poison probe proves to be the only reader -- 4 chains (the 3-channel
    3x3 stem stays with the library); their own output reaches HBM only when pass 2's cache keeps it.  Tables, maxima and
    histograms are those of the plain run."""
    model = cases.seed_model(_block_net(), base_seed=8).eval().cuda()
    want = _calibrate(model, False, cache_gb, plan, monkeypatch=monkeypatch)
    got = _calibrate(model, False, cache_gb, plan, monkeypatch=monkeypatch, skip=True)
    assert got[1] == want[1] and got[2] == want[2] and torch.equal(got[3], want[3])
    assert want[4]["relu_only_chains_proven"] == 0 and want[4]["launches_without_own_output"] == 0
    assert got[4]["relu_only_chains_proven"] == 4 and got[4]["conv_add_chains_proven"] == 0
    n = got[4]["launches_without_own_output"]
    if cache_gb == "0":
        assert n >= 4 * 4 + 4 * 5              # pass 1 (all but the modules' first, checked use) + every forward of pass 2
    elif plan == "A":
        assert n == 0                          # every batch is kept whole, pass 2 runs no forward
    else:
        assert 0 < n < 4 * 5 + 4 * 5           # the early tensors are not kept, the deep ones are


def test_small_planes_with_both_tensors_kept(monkeypatch):
    """8 x 8 planes and a cache that keeps everything.  With the sum written (pair_hist off) three store streams of partial lines
    make the one kernel slower than the two (scripts/conv_add_bench.py), so the convolution is launched on its own after all.
    With pair_hist (the default) the sum is not written -- the cache keeps the shortcut in its place and pass 2 histograms the
    pair -- so the same planes take the one kernel with its two store streams.  Same tables either way; without a cache the
    planes take the one kernel too."""
    model = cases.seed_model(_block_net(hw=16), base_seed=8).eval().cuda()
    want = _calibrate(model, False, "1", "A", monkeypatch=monkeypatch, hw=16)
    old = _calibrate(model, True, "1", "A", monkeypatch=monkeypatch, hw=16, pair=False)
    assert old[4]["conv_add_chains_proven"] == 2 and old[4]["conv_add_launches"] == 0 and old[4]["sums_left_to_pass2_pairs"] == 0
    assert old[1] == want[1] and old[2] == want[2] and torch.equal(old[3], want[3])
    got = _calibrate(model, True, "1", "A", monkeypatch=monkeypatch, hw=16, pair=True)
    assert got[4]["conv_add_launches"] in (2 * 4, 2 * 5) and got[4]["sums_left_to_pass2_pairs"] == got[4]["conv_add_launches"]
    assert got[1] == want[1] and got[2] == want[2] and torch.equal(got[3], want[3])
    cold = _calibrate(model, True, "0", "", monkeypatch=monkeypatch, hw=16)
    assert cold[4]["conv_add_launches"] == 2 * 5 and cold[4]["conv_add_hist_launches"] == 2 * 5 and cold[4]["sums_left_to_pass2_pairs"] == 0
    assert cold[1] == want[1] and cold[2] == want[2] and torch.equal(cold[3], want[3])


@pytest.mark.parametrize("cache_gb,plan", [("1", "A"), ("0.07", "A"), ("1", "B"), ("0.05", "B")])
def test_a_sum_whose_operands_the_cache_keeps_is_left_to_pass_2(monkeypatch, cache_gb, plan):
    """Quantity.pair_hist: a residual sum = conv3's output + the shortcut.  Where pass 2's cache keeps both tensors of a chain,
    pass 1 does not write the sum; the cache holds the shortcut (block 1: the projection's output, a kept tensor anyway; block 2:
    the previous block's ReLU output) and pass 2 counts conv3's output and the sum in one pass over the pair
    (fq_hist2048_pair_seg).  Every maximum, every histogram and the table are those of the run that writes and re-reads the sums,
    and of the unfused run; the cache holds no more bytes than before."""
    model = cases.seed_model(_block_net(), base_seed=8).eval().cuda()
    want = _calibrate(model, False, cache_gb, plan, monkeypatch=monkeypatch)
    off = _calibrate(model, True, cache_gb, plan, monkeypatch=monkeypatch, pair=False)
    on = _calibrate(model, True, cache_gb, plan, monkeypatch=monkeypatch, pair=True)
    # pair_chain (the default): block 2's shortcut is block 1's ReLU output = max(block 1's sum, 0), so the cache does not keep it
    # either -- pass 2 re-makes it while it counts block 1's pair (fq_hist2048_pair_seg's relu_out) and counts block 2's pair in a
    # second launch.  Without the chain the shortcut is kept as a tensor.
    flat = _calibrate(model, True, cache_gb, plan, monkeypatch=monkeypatch, pair=True, chain=False)
    for got in (off, on, flat):
        assert got[1] == want[1] and got[0] == want[0] and got[2] == want[2] and torch.equal(got[3], want[3])
    assert off[4]["sums_left_to_pass2_pairs"] == 0
    n = on[4]["sums_left_to_pass2_pairs"]
    if cache_gb == "1":
        assert n == on[4]["conv_add_launches"] and n in (2 * 4, 2 * 5)            # every chain of every fused forward
        assert flat[4]["cache_bytes"] <= off[4]["cache_bytes"]
        # one 8 x 128 x 32 x 32 fp32 tensor (4 MB) less per batch than the flat form: block 2's shortcut
        assert flat[4]["cache_bytes"] - on[4]["cache_bytes"] >= 4 * 8 * 128 * 32 * 32 * 4
    else:
        assert 0 < n <= on[4]["conv_add_launches"]                                # the chains / batches the partial cache keeps
        assert on[4]["cache_bytes"] <= flat[4]["cache_bytes"] + 1


def test_a_shortcut_written_in_place_after_the_add_keeps_its_sum_materialised(monkeypatch):
    """The shortcut must still hold, in pass 2, what the add saw.  A model that writes to it after the add (here: the block's
    input gets a constant added in place once the sum exists) is found by the probe forward's version counters: that Eltwise's
    sums are written as before, the other block's are left to pass 2.  Same tables as the unfused run."""
    from torch import nn
    net = _block_net()

    class Meddling(type(net.b2)):
        def forward(self, x):
            y = self.c3(self.r2(self.c2(self.r1(self.c1(x)))))
            z = self.r3(self.add(y, x))
            x.add_(1.0)                                            # after the add: the sum is right, the shortcut no longer is
            return z
    net.b2.__class__ = Meddling
    model = cases.seed_model(net, base_seed=8).eval().cuda()
    want = _calibrate(model, False, "1", "A", monkeypatch=monkeypatch)
    got = _calibrate(model, True, "1", "A", monkeypatch=monkeypatch, pair=True)
    assert got[1] == want[1] and got[2] == want[2] and torch.equal(got[3], want[3])
    # block 2's shortcut is block 1's ReLU output -- not a hooked tensor, so the calibration itself does not mind the write;
    # only block 1's chain (shortcut = the projection, untouched) may leave its sum to pass 2
    assert 0 < got[4]["sums_left_to_pass2_pairs"] <= got[4]["conv_add_launches"] // 2 + 1


def test_a_convolution_output_with_a_second_reader_is_not_deferred():
    """Block 2's convolution output is also read by an expression outside any module: the poison forward differs from the
    clean one, so no chain of this model is deferred -- and the tables are those of the unfused run."""
    model = cases.seed_model(_block_net("peek"), base_seed=8).eval().cuda()
    want = _calibrate(model, False)
    got = _calibrate(model, True)
    assert got[4]["conv_add_chains_proven"] == 0 and got[4]["conv_add_launches"] == 0
    assert got[1] == want[1] and got[2] == want[2] and torch.equal(got[3], want[3])


def test_an_inplace_relu_behind_the_eltwise_is_left_alone():
    model = cases.seed_model(_block_net("inplace_relu"), base_seed=8).eval().cuda()
    want = _calibrate(model, False)
    got = _calibrate(model, True)
    assert got[4]["conv_add_chains_proven"] == 0 and got[4]["inplace_consumers"] is True
    assert got[1] == want[1] and got[2] == want[2] and torch.equal(got[3], want[3])


def test_a_foreign_hook_on_the_convolution_keeps_its_output_materialised():
    """Somebody else's forward hook on the block's last convolution must see the convolution's output, as the reference's
    hook would: that chain is not deferred (the other one still is)."""
    model = cases.seed_model(_block_net(), base_seed=8).eval().cuda()
    want = _calibrate(model, False)
    seen = []
    h = model.b2.c3.register_forward_hook(lambda m, i, o: seen.append(float(o.abs().max())))
    try:
        got = _calibrate(model, True)
    finally:
        h.remove()
    assert got[4]["conv_add_chains_proven"] == 1
    assert got[1] == want[1] and got[2] == want[2] and torch.equal(got[3], want[3])
    assert seen and all(v == v and 0.0 < v < 1e6 for v in seen)      # the foreign hook saw real tensors in every forward


def test_a_forward_that_leaves_the_proven_path_is_refused():
    """Data-dependent control flow the probe cannot see: a calibration batch takes a branch in which the convolution's output
    never reaches its Eltwise.  The forward is refused instead of calibrating on memory nobody wrote."""
    from tools import Quantity
    model = cases.seed_model(_block_net(), base_seed=8).eval().cuda()
    block = model.b2
    plain = type(block).forward

    def moody(self, x):
        if float(x.flatten()[0]) == 12345.0:                  # never on the probe's input
            y = self.c3(self.r2(self.c2(self.r1(self.c1(x)))))
            return self.r3(y)                                 # (no Eltwise)
        return plain(self, x)
    batches = cases.calib_batches(3, (8, 3, 64, 64), seed=91)
    with product_workdir(input_shape="1,3,64,64", device="gpu", max_cali_img_num=2):
        q = Quantity(model)
        block.forward = moody.__get__(block)
        try:
            calls = {"n": 0}
            b1 = model.b1
            real_b1 = b1.forward

            def tagged(x):
                out = real_b1(x)
                calls["n"] += 1
                if calls["n"] == 4:                           # probe, poison probe, batch 0, then batch 1: the other branch
                    out = out.clone()
                    out.flatten()[0] = 12345.0
                return out
            b1.forward = tagged
            with pytest.raises(RuntimeError, match="fuse_conv_add"):
                q.activation_quantize(batches)
        finally:
            del block.forward, b1.forward


# ---- what NaN poisoning cannot see: code that KEEPS a tensor -------------------------------------------
def _keeper_net(kind):
    """_block_net() whose second block does something else with its last convolution's output y (or with the sum s):
      "compare"  a comparison reader: mask = (y > 0) gates the block's result -- NaN > 0 is False everywhere, so the poisoned
                 forward differs and the poison probe itself refuses the chain;
      "stash"    self.feat = y: nothing in the forward depends on it, the owner reads it AFTER the forward;
      "view"     self.feat = y[:, :4]: the same, through a view;
      "sum"      self.feat = s, the Eltwise's result before the ReLU."""
    from torch import nn
    from common.quantity import Eltwise
    net = _block_net()
    block = net.b2
    if kind == "compare":
        # (the graph discovery only follows tensors made by modules of the known op types, by class NAME: the comparison lives in
        #  a module called ReLU -- a subclass, which the fused forward does not patch -- and its result joins through an Eltwise)
        ReLU = type("ReLU", (nn.ReLU,), {"forward": lambda self, t: (t > 0).float()})
        block.gate, block.gate_add = ReLU(), Eltwise()

    def forward(self, x):
        y = self.c3(self.r2(self.c2(self.r1(self.c1(x)))))
        if kind == "stash":
            self.feat = y
        elif kind == "view":
            self.feat = y[:, :4]
        s = self.add(y, x if self.down is None else self.down(x))
        if kind == "sum":
            self.feat = s
        z = self.r3(s)
        if kind == "compare":
            z = self.gate_add(z, self.gate(y))
        return z
    block.forward = forward.__get__(block)
    return net


@pytest.mark.parametrize("kind", ["compare", "stash", "view", "sum"])
def test_readers_the_poison_cannot_see_are_found_and_their_chain_is_not_deferred(kind):
    """A comparison reader is caught by the poison forward itself; a KEEPER (attribute stash, view) influences
    nothing the probe compares, so `_DeferralProbe.holders` looks for whoever still refers to the tensor after the learning
    forward.  Either way block 2's chain keeps its own kernels -- the kept tensor holds what the convolution computed, in every
    forward -- block 1's chain is still fused, and the tables are the unfused run's."""
    plain = cases.seed_model(_block_net(), base_seed=8).eval().cuda()
    want = _calibrate(plain, False)
    model = cases.seed_model(_keeper_net(kind), base_seed=8).eval().cuda()
    got = _calibrate(model, True)
    if kind == "compare":
        ref = _calibrate(cases.seed_model(_keeper_net(kind), base_seed=8).eval().cuda(), False)
        assert got[4]["conv_add_chains_proven"] == 0                     # the poison forward differed: nothing is deferred
        assert got[1] == ref[1] and got[2] == ref[2] and torch.equal(got[3], ref[3])
        return
    assert got[4]["conv_add_chains_proven"] == 1, got[4]                  # block 1 only
    assert got[1] == want[1] and got[2] == want[2] and torch.equal(got[3], want[3])
    # the keeper's tensor after the LAST forward of the calibration: real values, the ones an unfused run leaves
    kept = model.b2.feat
    assert bool(torch.isfinite(kept).all()) and float(kept.abs().max()) > 0
    ref_model = cases.seed_model(_keeper_net(kind), base_seed=8).eval().cuda()
    _calibrate(ref_model, False)
    assert torch.equal(kept, ref_model.b2.feat)


def test_a_clean_model_needs_no_heap_pass(monkeypatch):
    """The keeper scan's heap pass (gc.get_referrers over every tracked object: 11-16 ms on a ResNet-50 process) is only for
    tensors whose reference count exceeds a control tensor's.  In a model nobody keeps anything of, no tensor does -- a loop
    variable of the proof itself used to hold the last convolution's output and made it a suspect in EVERY calibration."""
    from tools import _hook_state
    calls = []
    real = _hook_state._DeferralProbe.holders
    monkeypatch.setattr(_hook_state._DeferralProbe, "holders", staticmethod(lambda tensors, ours: (calls.append(len(tensors)), real(tensors, ours))[1]))
    model = cases.seed_model(_block_net(), base_seed=8).eval().cuda()
    got = _calibrate(model, True)
    assert got[4]["conv_add_chains_proven"] == 2 and calls == [], calls
    # ... and a keeper still is one
    model = cases.seed_model(_keeper_net("stash"), base_seed=8).eval().cuda()
    got = _calibrate(model, True)
    assert got[4]["conv_add_chains_proven"] == 1 and len(calls) >= 1 and all(n >= 1 for n in calls), calls


def test_a_list_collector_hooked_on_the_parent_block_sees_what_the_unfused_run_shows_it():
    """register_forward_hook on the BLOCK (not on a module of the chain): it receives the block's input and its result -- the
    ReLU's output, which the one-kernel tail does write -- and keeps them in a list.  The chain stays fused (nothing it keeps is
    a skipped tensor), and every kept tensor equals the unfused run's."""
    def run(fuse):
        model = cases.seed_model(_block_net(), base_seed=8).eval().cuda()
        seen = []
        h = model.b2.register_forward_hook(lambda m, i, o: seen.append((i[0], o)))
        try:
            out = _calibrate(model, fuse)
        finally:
            h.remove()
        return out, seen
    want, seen_w = run(False)
    got, seen_g = run(True)
    assert got[4]["conv_add_chains_proven"] == 2
    assert got[1] == want[1] and got[2] == want[2] and torch.equal(got[3], want[3])
    assert len(seen_g) >= 10 and all(torch.equal(a[1], b[1]) for a, b in zip(seen_g[-10:], seen_w[-10:]))


def test_an_input_written_in_place_before_the_eltwise_is_refused():
    """a deferred convolution runs later, on the input tensor object it was called with.  A model that writes that
    input in place between the convolution and the Eltwise is invisible to the probes (both run the convolution at its own
    position); the version counter is not, and such a forward raises instead of calibrating on the overwritten input."""
    from tools import Quantity
    net = _block_net()
    block = net.b2

    def forward(self, x):
        t = self.r2(self.c2(self.r1(self.c1(x))))
        y = self.c3(t)
        if getattr(self, "scribble", False):
            t.mul_(2.0)                                       # the convolution's INPUT, after the convolution
        return self.r3(self.add(y, x if self.down is None else self.down(x)))
    block.forward = forward.__get__(block)
    model = cases.seed_model(net, base_seed=8).eval().cuda()
    with product_workdir(input_shape="1,3,64,64", device="gpu", max_cali_img_num=2):
        q = Quantity(model)
        calls = {"n": 0}
        real = model.b1.forward

        def counted(x):
            calls["n"] += 1
            block.scribble = calls["n"] >= 4                  # probe, poison probe, batch 0 clean; from batch 1 on it scribbles
            return real(x)
        model.b1.forward = counted
        try:
            with pytest.raises(Exception, match="written in place"):
                q.activation_quantize(cases.calib_batches(3, (8, 3, 64, 64), seed=91))
        finally:
            del model.b1.forward


def test_a_relu_output_that_is_the_shortcut_of_two_later_blocks(monkeypatch):
    """A chain is linear.  Here block 1's ReLU output is the shortcut of block 2 AND of block 3 (a skip over two blocks): block 2
    continues block 1's chain, block 3 keeps its shortcut as a tensor (a single pair) -- and every statistic equals the unfused
    run's."""
    from torch import nn
    from common.quantity import Eltwise, View

    class Body(nn.Module):
        def __init__(self, cin, mid, cout):
            super().__init__()
            self.c1, self.r1 = nn.Conv2d(cin, mid, 1), nn.ReLU()
            self.c2, self.r2 = nn.Conv2d(mid, mid, 3, padding=1), nn.ReLU()
            self.c3 = nn.Conv2d(mid, cout, 1)
            self.add, self.r3 = Eltwise(), nn.ReLU()

        def forward(self, x, shortcut):
            return self.r3(self.add(self.c3(self.r2(self.c2(self.r1(self.c1(x))))), shortcut))

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.stem, self.relu0 = nn.Conv2d(3, 32, 3, padding=1), nn.ReLU()
            self.down = nn.Conv2d(32, 128, 1)
            self.b1, self.b2, self.b3 = Body(32, 16, 128), Body(128, 32, 128), Body(128, 32, 128)
            self.pool, self.view, self.fc = nn.AvgPool2d(32), View(), nn.Linear(128, 10)

        def forward(self, x):
            x0 = self.relu0(self.stem(x))
            x1 = self.b1(x0, self.down(x0))
            x2 = self.b2(x1, x1)
            x3 = self.b3(x2, x1)                                   # the shortcut from two blocks back
            return self.fc(self.view(self.pool(x3)))

    model = cases.seed_model(Net(), base_seed=8).eval().cuda()
    want = _calibrate(model, False, "1", "A", monkeypatch=monkeypatch, hw=32)
    got = _calibrate(model, True, "1", "A", monkeypatch=monkeypatch, hw=32, pair=True, chain=True)
    assert got[4]["conv_add_chains_proven"] == 3 and got[4]["sums_left_to_pass2_pairs"] >= 3 * 4
    assert got[1] == want[1] and got[2] == want[2] and torch.equal(got[3], want[3])
    part = _calibrate(model, True, "0.02", "B", monkeypatch=monkeypatch, hw=32, pair=True, chain=True)
    assert part[1] == want[1] and part[2] == want[2] and torch.equal(part[3], want[3])
