"""Host logic of common.quantity.resident on a box without a GPU: the tracer, the plan, the handles and the
module glue run for real; only the kernel entry points are replaced by oracle-backed doubles
(tests/native_doubles.py), which follow the reference's fp32 chain literally."""
import io
import pickle

import pytest
import torch
from torch import nn

import native_doubles


def _info(i, o, w=5):
    return {"weight_bit": w, "bias_bit": o, "input_bit": i, "output_bit": o}


class _Block(nn.Module):
    def __init__(self, cin, mid, cout, bits, project):
        from common.quantity import NewConv2d, NewAdd
        super(_Block, self).__init__()
        b_in, b1, b2, b3, b_sc, b_out = bits
        self.conv1 = NewConv2d(nn.Conv2d(cin, mid, 1, bias=False), _info(b_in, b1))
        self.relu1 = nn.ReLU(False)
        self.conv2 = NewConv2d(nn.Conv2d(mid, mid, 3, padding=1, bias=False), _info(b1, b2))
        self.relu2 = nn.ReLU(False)
        self.conv3 = NewConv2d(nn.Conv2d(mid, cout, 1, bias=False), _info(b2, b3))
        self.downsample = (nn.Sequential(NewConv2d(nn.Conv2d(cin, cout, 1, bias=False), _info(b_in, b_sc)))
                           if project else nn.Sequential())
        self.Eltwise = NewAdd()
        self.relu3 = nn.ReLU(False)

    def forward(self, x):
        y = self.relu1(self.conv1(x))
        y = self.relu2(self.conv2(y))
        y = self.conv3(y)
        return self.relu3(self.Eltwise(y, self.downsample(x)))


class _MiniResNet(nn.Module):
    """stem(3x3, folded) -> ReLU -> MaxPool -> bottleneck with projection -> bottleneck with identity ->
    AvgPool(global) -> flatten -> fc: every construct the ResNet-50 plan relies on, at toy size."""

    def __init__(self):
        from common.quantity import NewConv2d, NewLinear
        super(_MiniResNet, self).__init__()
        torch.manual_seed(11)
        self.conv1 = NewConv2d(nn.Conv2d(3, 16, 3, padding=1, bias=True), _info(5, 4))
        self.relu = nn.ReLU(False)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.block1 = _Block(16, 16, 32, (4, 4, 3, 3, 4, 3), True)
        self.block2 = _Block(32, 16, 32, (3, 4, 4, 2, None, 2), False)
        self.avgpool = nn.AvgPool2d(4)
        self.fc = NewLinear(nn.Linear(32, 10), _info(2, 1))

    def forward(self, x):
        x = self.maxpool(self.relu(self.conv1(x)))
        x = self.block2(self.block1(x))
        x = self.avgpool(x)
        return self.fc(x.view(x.size(0), -1))


def _mini_resnet():
    return _MiniResNet().eval()


def test_plan_of_a_mini_resnet_and_bit_identical_outputs():
    from common.quantity import resident
    with native_doubles.installed():
        net = _mini_resnet()
        x = torch.randn(2, 3, 8, 8)
        with torch.no_grad():
            plain = net(x)
        summary = resident.enable(net, x)                      # verify=True: compares with the traced forward itself
        assert summary == {"resident_convs": 8, "resident_adds": 2, "resident_pools": 2, "fused_relus": 7, "fp32_outputs": 0,
                           "int_only_outputs": 11, "fused_conv_adds": 2, "fused_block_tails": 0, "fused_projections": 0}, summary   # (16 / 32 channels: too thin)
        plans = resident.describe(net)
        assert plans["conv1"].relu and plans["conv1"].emit_int and not plans["conv1"].emit_f32          # stem -> int8 max-pool
        assert plans["maxpool"].emit_int and not plans["maxpool"].emit_f32 and plans["maxpool"].narrow_bit == 4
        assert plans["block1.conv3"].defer and not plans["block1.downsample.0"].defer                    # the add runs conv3
        assert plans["block1.Eltwise"].fuse_arg == 0 and plans["block1.Eltwise"].grid == 4               # max(0, 3, 4)
        assert plans["block1.Eltwise"].want_wide and plans["block1.Eltwise"].narrow_bit == 3            # -> block2 conv1 + add
        assert plans["block2.Eltwise"].want_wide and plans["block2.Eltwise"].narrow_bit is None         # -> global average pool
        assert plans["block2.Eltwise"].grid == 4                                                        # max(0, 2, 4)
        with torch.no_grad():
            assert torch.equal(net(x), plain)
            assert torch.equal(net(torch.cat([x, x]))[2:], plain)            # the plan does not depend on the batch size
            mid = net.block1(net.maxpool(net.relu(net.conv1(x))))
        assert type(mid).__name__ == "QHandle" and mid.exact.dtype == torch.int16 and mid.narrow.dtype == torch.int8
        with pytest.raises(Exception):
            torch.relu(mid)                                               # code outside the plan fails loudly
        # the plan (plain data on the modules) survives pickling of the whole model
        buf = io.BytesIO()
        pickle.dump(net, buf)
        again = pickle.loads(buf.getvalue())
        with torch.no_grad():
            assert torch.equal(again(x), plain)
        resident.disable(net)
        assert not resident.describe(net) and "forward" not in net.relu.__dict__ and "forward" not in net.maxpool.__dict__
        with torch.no_grad():
            assert torch.equal(net(x), plain)


class _WideNet(nn.Module):
    """Three bottlenecks at ResNet-50's first-stage widths (64 -> 256) behind a 3x3 stem."""

    def __init__(self):
        from common.quantity import NewConv2d
        super(_WideNet, self).__init__()
        torch.manual_seed(5)
        self.conv1 = NewConv2d(nn.Conv2d(3, 64, 3, padding=1, bias=True), _info(5, 4))
        self.relu = nn.ReLU(False)
        self.block1 = _Block(64, 64, 256, (4, 4, 3, 3, 4, 3), True)
        self.block2 = _Block(256, 64, 256, (3, 4, 4, 2, None, 2), False)
        self.block3 = _Block(256, 64, 256, (2, 4, 4, 2, None, 2), False)
        self.avgpool = nn.AvgPool2d(4)

    def forward(self, x):
        x = self.relu(self.conv1(x))
        x = self.block3(self.block2(self.block1(x)))
        return self.avgpool(x)


def test_block_tail_plan_runs_the_next_conv1_inside_the_add():
    """Two bottlenecks at ResNet-50's first-stage widths (64 -> 256): block 1's conv3 is deferred into its NewAdd, and that add's
    re-quantised sum feeds block 2's conv1 and nothing else -- so the add also runs THAT convolution (fq_block_tail_i8,
    Plan.fuse_next) and the int8 sum is not written; block 1's own conv1 / projection read the stem's int8 output (two readers:
    no fusion there).  Block 1's shortcut is a 1x1 projection of the block's input that only the add reads: it is deferred as well
    and the same kernel computes it (fq_block_tail_proj_i8, Plan.fuse_proj) -- no launch, no tensor.  Same outputs, bit for bit;
    the plan survives pickling; FQ_BLOCK_TAIL=0 keeps the two launches, FQ_BLOCK_TAIL_PROJ=0 the projection's own."""
    from common.quantity import resident
    with native_doubles.installed() as nat:
        net = _WideNet().eval()
        x = torch.randn(1, 3, 4, 4)
        with torch.no_grad():
            plain = net(x)
        calls = {"bt": 0, "add": 0, "proj": 0, "conv": 0}
        real_bt, real_add, real_proj, real_conv = nat.block_tail_i8, nat.conv2d_i8_add_resident, nat.block_tail_proj_i8, nat.conv2d_i8_resident

        def bt(*a, **k):
            calls["bt"] += 1
            return real_bt(*a, **k)

        def add(*a, **k):
            calls["add"] += 1
            return real_add(*a, **k)

        def proj(*a, **k):
            calls["proj"] += 1
            return real_proj(*a, **k)

        def conv(*a, **k):
            calls["conv"] += 1
            return real_conv(*a, **k)
        nat.block_tail_i8, nat.conv2d_i8_add_resident, nat.block_tail_proj_i8, nat.conv2d_i8_resident = bt, add, proj, conv
        try:
            summary = resident.enable(net, x)
            assert summary["fused_conv_adds"] == 3 and summary["fused_block_tails"] == 2 and summary["fused_projections"] == 1, summary
            plans = resident.describe(net)
            assert plans["block1.Eltwise"].fuse_next is net.block2.conv1 and not plans["block1.Eltwise"].narrow_to_hbm
            assert plans["block1.Eltwise"].fuse_proj and plans["block1.downsample.0"].defer and plans["block1.conv3"].defer
            assert plans["block2.Eltwise"].fuse_next is net.block3.conv1 and not plans["block2.Eltwise"].fuse_proj
            assert plans["block3.Eltwise"].fuse_next is None                 # its sum goes to the average pool
            for k in calls:
                calls[k] = 0
            with torch.no_grad():
                assert torch.equal(net(x), plain)
            # block 1: tail + projection + the next conv1 in one launch; block 2: tail + the next conv1; block 3: its tail alone.
            # Launches of their own: the stem's ... no: conv1 (folded stem) is not conv2d_i8_resident; block 1's conv1 and conv2,
            # conv2 of blocks 2 and 3 -- four; neither projection nor conv3 nor a fused conv1 among them
            assert calls == {"bt": 2, "add": 0, "proj": 1, "conv": 4}, calls
            with torch.no_grad():
                mid = net.block1(net.relu(net.conv1(x)))
            assert type(mid).__name__ == "QHandle" and mid.narrow is None and mid.next_out[0] is net.block2.conv1
            again = pickle.loads(pickle.dumps(net))
            with torch.no_grad():
                assert torch.equal(again(x), plain)
        finally:
            nat.block_tail_i8, nat.conv2d_i8_add_resident, nat.block_tail_proj_i8, nat.conv2d_i8_resident = real_bt, real_add, real_proj, real_conv


def test_block_tail_can_be_switched_off(monkeypatch):
    from common.quantity import resident
    monkeypatch.setenv("FQ_BLOCK_TAIL", "0")
    with native_doubles.installed():
        net = _mini_resnet()
        assert resident.enable(net, torch.randn(1, 3, 8, 8))["fused_block_tails"] == 0
        wide = _WideNet().eval()
        summary = resident.enable(wide, torch.randn(1, 3, 4, 4))
        assert summary["fused_block_tails"] == 0 and summary["fused_projections"] == 0


def test_projection_fusion_can_be_switched_off(monkeypatch):
    from common.quantity import resident
    monkeypatch.setenv("FQ_BLOCK_TAIL_PROJ", "0")
    with native_doubles.installed():
        net = _WideNet().eval()
        x = torch.randn(1, 3, 4, 4)
        with torch.no_grad():
            plain = net(x)
        summary = resident.enable(net, x)
        assert summary["fused_block_tails"] == 2 and summary["fused_projections"] == 0
        assert not resident.describe(net)["block1.downsample.0"].defer
        with torch.no_grad():
            assert torch.equal(net(x), plain)


def test_values_with_foreign_consumers_keep_their_fp32_form():
    from common.quantity import NewConv2d, NewAdd, resident

    class Net(nn.Module):
        def __init__(self):
            super(Net, self).__init__()
            torch.manual_seed(3)
            self.c1 = NewConv2d(nn.Conv2d(3, 16, 3, padding=1), _info(5, 4))
            self.c2 = NewConv2d(nn.Conv2d(16, 16, 3, padding=1), _info(4, 3))
            self.c3 = NewConv2d(nn.Conv2d(16, 16, 1), _info(3, 3))
            self.c4 = NewConv2d(nn.Conv2d(16, 16, 1), _info(3, 2))
            self.c5 = NewConv2d(nn.Conv2d(16, 8, 1), _info(4, 2))        # quantises at ANOTHER bit than c4 produces
            self.add = NewAdd()
            self.relu = nn.ReLU()                                           # one instance, every call site

        def forward(self, x):
            a = self.relu(self.c1(x))                  # fused, int8 only
            b = self.relu(self.c2(a))                  # fused; read by c3, by the add AND by a user op
            c = self.c3(b)                             # deferred into the add
            d = self.relu(self.add(c, b))              # resident add, fused ReLU
            e = self.c4(d)                             # read by c5 at a different bit and by a user op: fp32
            return self.c5(e) + e[:, :8] * 0.5 + b.mean()

    with native_doubles.installed():
        net = Net().eval()
        x = torch.randn(2, 3, 6, 6)
        with torch.no_grad():
            plain = net(x)
        summary = resident.enable(net, x)
        plans = resident.describe(net)
        assert plans["c1"].relu and not plans["c1"].emit_f32
        assert plans["c2"].relu and plans["c2"].emit_f32 and plans["c2"].emit_int
        assert plans["c3"].defer and plans["add"].fuse_arg == 0 and plans["add"].relu
        assert plans["c4"].emit_f32 and not plans["c4"].emit_int            # bit mismatch + user op
        assert summary["fused_relus"] == 3 and summary["fused_conv_adds"] == 1
        with torch.no_grad():
            assert torch.equal(net(x), plain)


def test_verify_refuses_a_plan_that_changes_the_output(monkeypatch):
    """enable(verify=True) must notice a wrong integer path (here: a sabotaged add double) and remove the plan."""
    from common.quantity import resident
    with native_doubles.installed() as nat:
        net = _mini_resnet()
        x = torch.randn(1, 3, 8, 8)
        good = nat.conv2d_i8_add_resident

        def bad(*a, **k):
            wide, narrow = good(*a, **k)
            return (wide + 1 if wide is not None else None), narrow

        monkeypatch.setattr(nat, "conv2d_i8_add_resident", bad)
        with pytest.raises(nat.FqError):
            resident.enable(net, x)
        assert not resident.is_enabled(net) and not resident.describe(net)
