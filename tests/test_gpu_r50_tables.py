"""BASELINE configs 2 and 5 at their own model sizes (north_star: "bit-identical feat.table / weight.table vs CPU on
ResNet-50"): the bottleneck ResNet-50 / ResNet-101 calibrated with the HIP engine and with the CPU oracle engine on the
SAME activations must give identical maxima, 2048-bin histograms, tables and JSON files.

The activations come from the GPU forward (its fp32 convolutions -- this library's own MFMA kernels -- differ from a CPU's
in the last bits, so running the CPU engine on its own forward would compare convolution implementations, not calibrators): the HIP run tapes every tensor
its statistics were taken from, the oracle run replays the tape.

The tape sees BOTH ways a tensor reaches the statistics: handed to the collector (refresh_max_val /
add_to_distributions), or -- the default pass-1 path that bench.py times -- served by its producer
(the 53 convolutions' own epilogues -- fq_conv1x1_f32 / fq_conv_kxk_f32 / fq_conv_stem_f32; fq_bias_add_absmax_f32 with
FQ_OWN_CONV1X1=0 --, fq_add_absmax_f32 for the 16 Eltwise adds, the 49 ReLUs fed from those kernels); those are taped from the forward hook (`_EagerStats.note`), after the fused kernel wrote them.

pytest -m gpu"""
import hashlib
import os

import numpy as np
import pytest
import torch

import cases
from workdir_util import product_workdir

pytestmark = pytest.mark.gpu


def _state(wd):
    out = {"feat": open(os.path.join(wd, "feat.table")).read(), "weight_table": open(os.path.join(wd, "weight.table")).read()}
    for d in ("weight", "bias", "new_weight", "new_bias"):
        out[d] = {f: hashlib.sha256(open(os.path.join(wd, d, f), "rb").read()).hexdigest()
                  for f in sorted(os.listdir(os.path.join(wd, d)))}
    return out


def _hip_vs_oracle(oracle, model, batches, input_shape, rows, fused, monkeypatch, weights=True):
    """Runs both engines; returns (hip timings, tape sizes).  Asserts every equality itself."""
    from common.quantity import DistributionCollector
    from engine_doubles import OracleCollector, OracleQuantizer
    from tools import Quantity, pytorch_quantizer as pq

    tape = {"max": [], "hist": []}
    n_batches = len(batches)

    class RecordingCollector(DistributionCollector):
        """Records what the statistics were taken from, one merged dict per forward."""

        def _record(self, kind, tensors):
            cur = self.__dict__.setdefault("_open_" + kind, {})
            assert not set(cur) & set(tensors), "a tensor reached the statistics twice in one forward"
            cur.update({k: v.detach().cpu().numpy().copy() for k, v in tensors.items()})
            if len(cur) == len(self._tensor_list):
                tape[kind].append(dict(cur))
                cur.clear()

        def refresh_max_val(self, tensors):
            self._record("max", tensors)
            super().refresh_max_val(tensors)

        def add_to_distributions(self, tensors):
            self._record("hist", tensors)
            super().add_to_distributions(tensors)

    # tensors whose statistic was folded into their producer's kernel never reach the collector: tape them where the
    # calibration loop registers them (after the fused kernel ran, so the values are the final ones)
    orig_note = pq._EagerStats.note
    noted = {"n": 0}

    def note(self, key, t):
        orig_note(self, key, t)
        kind = "max" if getattr(self.fn, "__name__", "") == "refresh_max_val" else "hist"
        self.fn.__self__._record(kind, {key: t})
        noted["n"] += 1

    monkeypatch.setattr(pq._EagerStats, "note", note)

    class ReplayCollector(OracleCollector):
        _replay = True

        def refresh_max_val(self, tensors):
            super().refresh_max_val(tape["max"].pop(0) if self._replay else tensors)

        def add_to_distributions(self, tensors):
            super().add_to_distributions(tape["hist"].pop(0) if self._replay else tensors)

    class HipQuantity(Quantity):
        collector_cls = RecordingCollector
        fuse_bias_absmax = fused
        fuse_relu = fused
        materialize_all = True        # (conv3 + Eltwise in one kernel: both tensors must exist in HBM to be taped)

    class CpuQuantity(Quantity):
        collector_cls = ReplayCollector
        quantizer_cls = OracleQuantizer

    with product_workdir(input_shape=input_shape, device="gpu", max_cali_img_num=n_batches - 1) as tmp:
        q = HipQuantity(model)
        q.activation_quantize(batches)
        timings = dict(q.timings)
        hip_hist = q._collector.hist_device.cpu().numpy()
        hip_max = q._collector.max_device.cpu().numpy()
        if weights:
            q.collector_cls = DistributionCollector              # weights: plain HIP collector
            q.weight_quantize()
            hip = _state(os.path.join(tmp, "test", "workdir"))
        else:
            hip = {"feat": open(os.path.join(tmp, "test", "workdir", "feat.table")).read()}
    assert len(tape["max"]) == n_batches and len(tape["hist"]) == n_batches, (len(tape["max"]), len(tape["hist"]))
    monkeypatch.setattr(pq._EagerStats, "note", orig_note)

    with product_workdir(input_shape=input_shape, device="gpu", max_cali_img_num=n_batches - 1) as tmp:
        q2 = CpuQuantity(model)
        q2.activation_quantize(batches)                          # forwards run, their outputs are ignored
        cpu_hist = q2._collector._hist
        cpu_max = q2._collector._max
        if weights:
            ReplayCollector._replay = False                      # weights come straight from the parameters
            q2.weight_quantize()
            cpu = _state(os.path.join(tmp, "test", "workdir"))
        else:
            cpu = {"feat": open(os.path.join(tmp, "test", "workdir", "feat.table")).read()}

    np.testing.assert_array_equal(hip_max, cpu_max)
    np.testing.assert_array_equal(hip_hist, cpu_hist)            # rows x 2048 bins, exact
    assert hip["feat"] == cpu["feat"]
    assert len(hip["feat"].strip().split("\n")) == rows
    if weights:
        assert hip["weight_table"] == cpu["weight_table"]
        for d in ("weight", "bias", "new_weight", "new_bias"):
            assert hip[d] == cpu[d], d
    return timings, noted["n"]


@pytest.mark.parametrize("arch,rows,batch,fused", [("r50", 71, 4, False), ("r101", 139, 2, False),
                                                  ("r50", 71, 2, True), ("r101", 139, 1, True)])
def test_bottleneck_tables_hip_equals_cpu_oracle(oracle, monkeypatch, arch, rows, batch, fused):
    """fused = False: every tensor goes through fq_absmax_seg / fq_hist2048_seg.
    fused = True (the DEFAULT switches, what bench.py times): four batches, so that the last two run pass 1 entirely on
    the fused kernels (a module's first batch is a plain forward, its second verifies the decomposition, from the third
    on the ReLUs are served too)."""
    from common.quantity import merge_bn
    from model.resnet.ResNet_fabu import ResNet50, ResNet101
    ctor = ResNet50 if arch == "r50" else ResNet101         # r101: 139 rows = two chunked kernel launches
    model = merge_bn(cases.seed_model(ctor(), gamma_scale=0.7 if arch == "r50" else 0.5).eval()).cuda()
    batches = cases.calib_batches(4 if fused else 2, (batch, 3, 224, 224), seed=77)
    timings, noted = _hip_vs_oracle(oracle, model, batches, "1,3,224,224", rows, fused, monkeypatch)
    if fused:
        n_conv = sum(1 for m in model.modules() if isinstance(m, torch.nn.Conv2d))
        n_add = rows - 2 - n_conv                               # rows = image + convs + fc + Eltwise
        assert timings["fused_bias_absmax_convs"] == n_conv, timings
        assert timings["fused_add_absmax_eltwise"] == n_add, timings
        assert timings["fused_relus"] == 1 + 3 * n_add, timings   # stem ReLU + three per bottleneck block: all of them
        assert noted >= 2 * (n_conv + n_add)                    # at least the last two batches were fully fused
    else:
        assert timings["fused_bias_absmax_convs"] == 0 and noted == 0


def test_config5_r101_at_512_tables_hip_equals_cpu_oracle(oracle, monkeypatch):
    """BASELINE config 5's shape: ResNet-101 @3x512x512 (139 rows, 132 M cared elements per image), default switches,
    three batches of one image (the third runs pass 1 fully fused); activation tables only (the weights are the
    ResNet-101 weights the 224^2 test above already covers)."""
    from common.quantity import merge_bn
    from model.resnet.ResNet_fabu import ResNet101
    model = merge_bn(cases.seed_model(ResNet101(input_size=512), gamma_scale=0.5).eval()).cuda()
    batches = cases.calib_batches(3, (1, 3, 512, 512), seed=512)
    timings, noted = _hip_vs_oracle(oracle, model, batches, "1,3,512,512", 139, True, monkeypatch, weights=False)
    assert timings["fused_bias_absmax_convs"] == 104 and timings["fused_add_absmax_eltwise"] == 33, timings
    assert noted >= 104 + 33


def test_config5_fused_fakequant_on_the_largest_r101_512_activation(oracle):
    """The fused QuanDequan kernel (ReconTest's per-output pass) on a config-5 sized tensor, 16 x 256 x 256 x 256 fp32
    = 1 GiB -- four times ResNet-101 @512^2's largest activation at batch 16, far beyond the Infinity Cache, so the
    streaming (non-temporal) form of the kernel runs -- against the CPU oracle, exact, out of place and in place."""
    from common.quantity import _native
    n = 16 * 256 * 256 * 256
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(n, generator=g, device="cuda") * 9.0
    x[::100003] = 1e6                                           # saturation on both sides, ties, zeros
    x[1::100003] = -1e6
    x[2::100003] = 0.5 / 8
    x[3::100003] = 0.0
    host = x.cpu().numpy()
    for bit in (3, -1):
        y = _native.quandequan(x, bit, 8)
        want = oracle.quandequan(host, bit)
        assert np.array_equal(y.cpu().numpy(), want)
        del y, want
    y = _native.quandequan(x, 3, 8, out=x)                      # in place, as TestConv.forward calls it
    assert y.data_ptr() == x.data_ptr()
    assert np.array_equal(x.cpu().numpy(), oracle.quandequan(host, 3))


def test_config5_recontest_r101_at_512_every_layer_output_equals_the_oracle(oracle):
    """BASELINE config 5's second half: the fake-quant evaluation model (ReconTest) of ResNet-101 @3x512x512.  There is no
    reference capture at this size (the reference's Python histogram loop needs minutes per image), so the check is per
    layer and exact: for each of the 105 TestConv / TestLinear modules the output the model hands on must be the CPU
    oracle's QuanDequan of the float convolution's own result (the same kernel run without the epilogue on the same
    input; the classifier as the 1x1 convolution of a 1 x 1 plane) -- i.e. the QuanDequan epilogue of fq_conv1x1_qd_f32 /
    fq_conv_kxk_qd_f32 / fq_conv_stem_qd_f32 on every real
    activation of the model, from 16 M-element planes down to 16 x 16 -- and the fake-quantised parameters must be the
    oracle's QuanDequan of the folded ones."""
    from common.quantity import merge_bn
    from model.resnet.ResNet_fabu import ResNet101
    from tools import Quantity, Reconstruction
    with product_workdir(input_shape="1,3,512,512", device="gpu", max_cali_img_num=1):
        model = merge_bn(cases.seed_model(ResNet101(input_size=512), gamma_scale=0.5).eval()).cuda()
        q = Quantity(model)
        q.activation_quantize(cases.calib_batches(2, (1, 3, 512, 512), seed=512))
        q.weight_quantize()
        float_model = merge_bn(cases.seed_model(ResNet101(input_size=512), gamma_scale=0.5).eval())
        folded = {k: v.clone() for k, v in float_model.state_dict().items()}
        rec = Reconstruction(float_model.cuda())
        info = rec.get_quantity_information()
        net = rec.ReconTest(info, "./workdir/recontest.pth")
        from common.quantity import _float_conv, _native
        checked = {"n": 0, "elems": 0, "fused": 0}
        calls = {"qd": 0}
        real_qd = _native.quandequan

        def counted_qd(*a, **k):
            calls["qd"] += 1
            return real_qd(*a, **k)

        def check(name, inner, bit):
            # (hooks on the OUTER TestConv / TestLinear only: a hook on the inner nn.Conv2d would -- rightly -- switch the
            #  layer back to the two-pass form, because such a hook must see the un-quantised convolution output)
            def hook(mod, inputs, out):
                x = inputs[0]
                if isinstance(inner, torch.nn.Conv2d):
                    k = _float_conv.kind(inner, x)
                    assert k is not None, name                 # every convolution of the model is this library's
                    raw = _float_conv.plain(inner, k, x, check=False)
                    checked["fused"] += 1
                else:                                          # the classifier: the 1x1 convolution of a 1 x 1 plane
                    wt = inner.weight.detach().t().contiguous()
                    raw = _native.conv1x1_f32(x.view(x.shape[0], -1, 1, 1), wt, inner.bias, 1).view(x.shape[0], -1)
                    checked["fused"] += 1
                want = oracle.quandequan(raw.cpu().numpy(), bit)
                assert np.array_equal(out.detach().cpu().numpy(), want), name
                checked["n"] += 1
                checked["elems"] += want.size
            return hook

        for name, m in net.named_modules():
            if type(m).__name__ in ("TestConv", "TestLinear"):
                inner = m.Conv if hasattr(m, "Conv") else m.linear
                m.register_forward_hook(check(name, inner, m.output_bit))
                w = inner.weight.detach().cpu().numpy()
                assert np.array_equal(w, oracle.quandequan(folded[name + ".weight"].numpy(), m.weight_bit)), name
                assert np.array_equal(inner.bias.detach().cpu().numpy(),
                                      oracle.quandequan(folded[name + ".bias"].numpy(), m.bias_bit)), name
        _native.quandequan = counted_qd
        try:
            with torch.no_grad():
                logits = net(cases.fixed_input((1, 3, 512, 512), seed=5).cuda())
        finally:
            _native.quandequan = real_qd
        assert checked["n"] == 105 and checked["elems"] > 60_000_000, checked
        # the 104 convolutions and the classifier ran as ONE kernel each (QuanDequan in the epilogue): no standalone pass
        assert checked["fused"] == 105 and calls["qd"] == 0, (checked, calls)
        assert torch.isfinite(logits).all()


# ------------------------------------------------------------------------------------------------------------------
# G4-R50: the reference's own ReconModel / ReconTest on the fabu ResNet-50 (tests/golden/make_golden_r50.py)
# ------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def g4r50(golden_dir):
    import json
    with open(os.path.join(golden_dir, "g4_r50_tables.json")) as fh:
        tables = json.load(fh)
    return tables, np.load(os.path.join(golden_dir, "g4_r50_recon.npz"))


def _r50_folded(tables, golden_dir=None):
    """The seeded ResNet-50 with BatchNorm folded EXACTLY as the reference folded it in the build container: the fold
    factors gamma / sqrt(var + 1e-5) come from the fixture (torch.sqrt on CPU tensors runs through MKL and differs in the
    last bit between the build container's Intel host and the GPU box's AMD host -- one ulp in a weight is enough to flip
    its 8-bit rounding; cases.fold_bn_with_scales).  The product's merge_bn is the reference's expression and, like it,
    follows the machine it runs on; tests/test_host_logic.py pins it to golden G8."""
    from model.resnet.ResNet_fabu import ResNet50
    scales = np.load(os.path.join(os.path.dirname(os.path.abspath(cases.__file__)), "g4_r50_bn_scales.npz"))
    return cases.fold_bn_with_scales(cases.seed_model(ResNet50(), gamma_scale=tables["gamma_scale"]).eval(), scales)


def _r50_rec(tables, tmp):
    from tools import Reconstruction
    wd = os.path.join(tmp, "test", "workdir")
    os.makedirs(wd, exist_ok=True)
    with open(os.path.join(wd, "feat.table"), "w") as fh:
        fh.write(tables["feat_table"])
    with open(os.path.join(wd, "weight.table"), "w") as fh:
        fh.write(tables["weight_table"])
    rec = Reconstruction(_r50_folded(tables))
    rec.merge_bn()                                            # nothing left to fold
    return rec, wd


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_r50_reconmodel_logits_equal_the_reference(g4r50):
    """BASELINE config 3 at its own size: the integer-simulation ResNet-50 built from the REFERENCE's tables must give
    the reference's CPU logits exactly -- with fp32 module boundaries (the drop-in default), with resident integer
    activations, and as a captured HIP graph.  Exactness holds because every accumulator stays below 2^24 (golden)."""
    from common.quantity import resident
    tables, g4 = g4r50
    assert tables["recon_max_abs_accumulator"] < 2 ** 24
    x = cases.fixed_input(tuple(tables["input"]["shape"]), seed=tables["input"]["seed"]).cuda()
    with product_workdir(input_shape="1,3,224,224", device="gpu") as tmp:
        rec, wd = _r50_rec(tables, tmp)
        info = rec.get_quantity_information()
        assert {k: {kk: vv for kk, vv in v.items() if kk != "layer"} for k, v in info.items()} == tables["quantity_information"]
        net = rec.ReconModel(info, os.path.join(wd, "recon.pth")).cuda()
        assert sorted(net.state_dict().keys()) == tables["recon_state_dict_keys"]
        with torch.no_grad():
            c1 = net.conv1(x).cpu().numpy()
            logits = net(x).cpu().numpy()
            # stage by stage (sha256 of the whole tensor + a sub-sample in the fixture), so that a mismatch names its layer
            s = net.maxpool(net.relu(net.conv1(x)))
            for stage in ("layer1", "layer2", "layer3", "layer4"):
                s = getattr(net, stage)(s)
                got = s.cpu().numpy()
                np.testing.assert_array_equal(got[:, :16, ::4, ::4], g4["recon_%s_sample" % stage], err_msg=stage)
                assert _sha(got) == tables["recon_stage_sha256"][stage], stage
            pooled = net.view(net.avgpool(s))
            np.testing.assert_array_equal(pooled.cpu().numpy(), g4["recon_fc_input"])          # AvgPool2d(7): sum / 49
            # the classifier alone, on the reference's own input to it
            np.testing.assert_array_equal(net.fc(torch.from_numpy(g4["recon_fc_input"]).cuda()).cpu().numpy(), g4["logits_recon"])
        np.testing.assert_array_equal(c1[:, :8, ::8, ::8], g4["recon_conv1_out_sample"])
        assert _sha(c1) == tables["recon_conv1_out_sha256"]
        np.testing.assert_array_equal(logits, g4["logits_recon"])
        summary = resident.enable(net, x)
        assert summary["resident_convs"] == 53 and summary["fused_conv_adds"] == 16, summary
        with torch.no_grad():
            np.testing.assert_array_equal(net(x).cpu().numpy(), g4["logits_recon"])
            big = net(torch.cat([x, torch.flip(x, dims=[0]), x])).cpu().numpy()       # another batch size, same plan
        np.testing.assert_array_equal(big[:2], g4["logits_recon"])
        np.testing.assert_array_equal(big[2:4], g4["logits_recon"][::-1])
        graphed = resident.capture(net, x)
        np.testing.assert_array_equal(graphed(x).cpu().numpy(), g4["logits_recon"])
        np.testing.assert_array_equal(graphed(torch.flip(x, dims=[0])).cpu().numpy(), g4["logits_recon"][::-1])


@pytest.mark.timeout(1200)
def test_r50_reconmodel_at_the_batch_the_bench_times_equals_the_reference(g4r50):
    """bench.py times the integer-simulation forward at 256 images, where the dispatch differs from the
    2- and 6-image forwards of the test above -- launches of >= 256 workgroups take the eight-wave conv3x3_i8_halo8 kernel, the
    64-channel 3x3 layers the persistent c64 kernel, the 7x7 layers other tile shapes.  The reference's golden input (2 images)
    tiled to 256: every pair of rows of the logits must be the reference's CPU logits, with fp32 module boundaries, resident,
    as one HIP graph and as two graphs on two streams -- and the kernels that ran are the ones the bench's plan records
    (fq_conv2d_i8_last_variant through _native.conv_variant_log: an assertion, not a belief)."""
    from common.quantity import _native, resident
    tables, g4 = g4r50
    x = cases.fixed_input(tuple(tables["input"]["shape"]), seed=tables["input"]["seed"]).cuda()
    big = x.repeat(128, 1, 1, 1).contiguous()
    want = np.tile(g4["logits_recon"], (128, 1))
    with product_workdir(input_shape="1,3,224,224", device="gpu") as tmp:
        rec, wd = _r50_rec(tables, tmp)
        net = rec.ReconModel(rec.get_quantity_information(), os.path.join(wd, "recon.pth")).cuda()
        with torch.no_grad():
            np.testing.assert_array_equal(net(big).cpu().numpy(), want)                  # the reference's module boundaries
        resident.enable(net, big)
        _native.conv_variant_log = log = {}
        try:
            with torch.no_grad():
                got = net(big).cpu().numpy()
        finally:
            _native.conv_variant_log = None
        np.testing.assert_array_equal(got, want)
        # 48 launches for 54 layers: the stem, 16 3x3 layers (three 64-channel ones on the stationary-weight kernel, the chip-filling
        # ones on the eight-wave halo kernel), the 1x1 layers, the classifier -- and six block tails on fq_block_tail_i8, five of which
        # also run the next block's 1x1 reduction (stages 1 and 2), so those five layers have no launch of their own; the first of
        # them (stage 1, block 0) computes its projection shortcut as well (fq_block_tail_proj_i8): that layer has no launch either
        assert sum(log.values()) == 48 and log.get("block_tail/128") == 5 and log.get("block_tail_proj/128") == 1, log
        assert log.get("stem/64") == 1 and log.get("c64_halo/64") == 3, log
        assert sum(v for k, v in log.items() if k.startswith("halo8")) >= 7, log
        # all 13 stride-1 3x3 layers on the resident-halo kernels; the three stride-2 ones and the deep 1x1 reductions on LDS-DMA
        assert sum(v for k, v in log.items() if k.startswith(("halo", "c64_halo"))) == 13, log
        assert sum(v for k, v in log.items() if k.startswith("dma")) >= 3, log
        small = {}
        _native.conv_variant_log = small
        try:
            with torch.no_grad():
                np.testing.assert_array_equal(net(x).cpu().numpy(), g4["logits_recon"])
        finally:
            _native.conv_variant_log = None
        assert not any(k.startswith("halo8") for k in small), small                      # (what the 2-image test never reached)
        graphed = resident.capture(net, big)
        np.testing.assert_array_equal(graphed(big).cpu().numpy(), want)
        np.testing.assert_array_equal(graphed(torch.flip(big, dims=[0])).cpu().numpy(), want[::-1])
        dual = resident.capture(net, big, streams=2)
        np.testing.assert_array_equal(dual(big).cpu().numpy(), want)


def test_r50_recontest_logits_match_the_reference(g4r50):
    """Fake-quant ResNet-50: float convolutions followed by QuanDequan.  The convolutions are this library's fp32 MFMA
    kernels (fq_conv1x1_f32 / fq_conv_kxk_f32 / fq_conv_stem_f32: a fixed k-ordered fma chain), the reference's are oneDNN's
    on the CPU: same mathematics, different summation order, so a value can land on the other side of a rounding tie and
    an output may differ by one quantisation step (2^-output_bit), which then propagates.
    Observed on MI355X (scripts/_dbg/recontest_diff.py, round 3): the first layer's sample and ALL logits identical to the
    reference's.  Stated tolerance: first layer >= 99.99 % identical and never more than one step apart; logits >= 99 %
    identical, none more than one step of the classifier's grid apart, the same arg-max per image."""
    tables, g4 = g4r50
    x = cases.fixed_input(tuple(tables["input"]["shape"]), seed=tables["input"]["seed"]).cuda()
    with product_workdir(input_shape="1,3,224,224", device="gpu") as tmp:
        rec, wd = _r50_rec(tables, tmp)
        info = rec.get_quantity_information()
        net = rec.ReconTest(info, os.path.join(wd, "recontest.pth")).cuda()
        with torch.no_grad():
            c1 = net.conv1(x).cpu().numpy()
            logits = net(x).cpu().numpy()
    step1 = 2.0 ** -info["conv1"]["output_bit"]
    sample = c1[:, :8, ::8, ::8]
    assert np.mean(sample == g4["recontest_conv1_out_sample"]) >= 0.9999
    assert np.max(np.abs(sample - g4["recontest_conv1_out_sample"])) <= step1 * (1 + 1e-6)
    step = 2.0 ** -info["fc"]["output_bit"]
    diff = np.abs(logits - g4["logits_recontest"])
    assert np.max(diff) <= step * (1 + 1e-6), (float(np.max(diff)), step)
    assert np.mean(diff == 0) >= 0.99, float(np.mean(diff == 0))
    assert np.array_equal(logits.argmax(1), g4["logits_recontest"].argmax(1))


def test_r50_end_to_end_feat_table_equals_the_reference(g4r50, golden_dir):
    """north_star's target itself: the fabu ResNet-50, calibrated END TO END on the GPU -- this library's float forward
    (fp32 MFMA convolutions, pooling, adds), its abs-max / histogram kernels and its KL search -- on the recipe the
    reference was run on in the build container (tests/golden/make_golden_r50.py: seed 77, 2 batches of 2x3x224x224,
    gamma 0.5, MAX_CALI_IMG_NUM 1) writes the reference's feat.table, all 71 rows.

    The float forward is not bit-identical to the reference's CPU forward (summation order of the convolutions), so the
    statistics differ in their last digits; the tolerance is stated on what the reference's KL search was handed
    (g4_r50_calib_stats.npz, `r50stats` capture): every merged interval within 2e-6 relative (observed on MI355X:
    8.5e-7, 60 of 71 rows differ at all), every 2048-bin histogram within 2e-3 of its element count in L1 distance
    (observed: 1.0e-3 on the classifier row, 70 of 71 rows differ) -- and none of that moves a threshold across a bit."""
    from common.quantity import Quantizer
    from tools import Quantity
    tables, _ = g4r50
    stats = np.load(os.path.join(golden_dir, "g4_r50_calib_stats.npz"))
    seen = {}

    class SpyQuantizer(Quantizer):
        def quantize(self, distributions, distribution_intervals):
            seen["names"] = list(distribution_intervals.keys())
            seen["interval"] = np.array([float(distribution_intervals[k]) for k in seen["names"]])
            seen["hist"] = distributions.cpu().numpy().copy()
            return super().quantize(distributions, distribution_intervals)

    class SpyQuantity(Quantity):
        quantizer_cls = SpyQuantizer

    recipe = tables["calib"]
    with product_workdir(input_shape="1,3,224,224", device="gpu", max_cali_img_num=recipe["n_batches"] - 1) as tmp:
        q = SpyQuantity(_r50_folded(tables).cuda())
        q.activation_quantize(cases.calib_batches(recipe["n_batches"], tuple(recipe["shape"]), seed=recipe["seed"]))
        feat = open(os.path.join(tmp, "test", "workdir", "feat.table")).read()
    assert seen["names"] == list(stats["names"])
    ref_i, ref_h = stats["interval"], stats["hist"]
    assert np.max(np.abs(seen["interval"] - ref_i) / ref_i) <= 2e-6
    l1 = np.abs(seen["hist"] - ref_h).sum(axis=1) / ref_h.sum(axis=1)
    assert l1.max() <= 2e-3, (float(l1.max()), seen["names"][int(l1.argmax())])
    got, want = feat.strip().split("\n"), tables["feat_table"].strip().split("\n")
    assert len(got) == len(want) == 71
    assert [a for a, b in zip(got, want) if a != b] == []         # (names the differing rows on failure)
    assert feat == tables["feat_table"]


def test_r50_weight_tables_equal_the_reference(g4r50):
    """weight.table and all 214 JSON files of ResNet-50 (25.5 M parameters) written by the HIP engine are byte-identical
    to the reference's (weights never pass through a convolution, so this is exact on any device)."""
    from tools import Quantity
    tables, _ = g4r50
    with product_workdir(input_shape="1,3,224,224", device="gpu") as tmp:
        model = _r50_folded(tables).cuda()
        q = Quantity(model)
        wd = os.path.join(tmp, "test", "workdir")
        with open(os.path.join(wd, "feat.table"), "w") as fh:
            fh.write(tables["feat_table"])
        q.weight_quantize()
        got = _state(wd)
    assert got["weight_table"] == tables["weight_table"]
    for d in ("weight", "bias", "new_weight", "new_bias"):
        assert got[d] == tables["files"][d], d


def test_r50_at_the_batch_the_bench_times_every_fusion_and_the_cache_change_no_histogram(monkeypatch):
    """The configuration bench.py times -- ResNet-50, 256 images per batch, a cache that keeps the deep tensors of every batch, every
    producer fusion, the residual sums left to pass 2's chain kernel -- against the plainest path the engine has on the same
    batches: no cache (every image through the network twice), no proof-based fusion, no pair / chain histograms.  All 71 maxima
    and all 71 x 2048 bins must be equal (the own convolutions are deterministic, so 'equal' means bit for bit), and so must
    the table.  Three batches: the third runs pass 1 with every module checked and every fusion active."""
    from tools import Quantity
    from common.quantity import merge_bn
    from model.resnet.ResNet_fabu import ResNet50
    model = merge_bn(cases.seed_model(ResNet50(), gamma_scale=0.5).eval()).cuda()
    g = torch.Generator(device="cuda").manual_seed(4321)
    batches = [(torch.randn(256, 3, 224, 224, generator=g, device="cuda"), None) for _ in range(3)]

    def run(cache_gb, plan, **switches):
        monkeypatch.setenv("FQ_ACT_CACHE_GB", cache_gb)
        monkeypatch.setenv("FQ_CACHE_PLAN", plan)
        with product_workdir(input_shape="1,3,224,224", device="gpu", max_cali_img_num=2) as tmp:
            q = Quantity(model)
            for k, v in switches.items():
                setattr(q, k, v)
            q.activation_quantize(batches)
            return (open(os.path.join(tmp, "test", "workdir", "feat.table")).read(), q._collector.max_device.clone(),
                    q._collector.hist_device.clone(), dict(q.timings))
    plain = run("0", "", fuse_conv_add=False, skip_unread_outputs=False, pair_hist=False, fuse_hist=False)
    fused = run("24", "B")                               # 8 GB per batch: the deep half of the network is kept, as in the bench
    assert fused[3]["cache_plan"]["kind"] == "B" and fused[3]["cache_bytes"] > 10e9
    assert fused[3]["sums_left_to_pass2_pairs"] >= 16 and fused[3]["conv_add_launches"] >= 16 * 2
    assert plain[3]["cache_bytes"] == 0 and plain[3]["sums_left_to_pass2_pairs"] == 0 and plain[3]["conv_add_launches"] == 0
    assert torch.equal(fused[1], plain[1])
    assert torch.equal(fused[2], plain[2])
    assert fused[0] == plain[0] and len(fused[0].strip().split("\n")) == 71
    whole = run("60", "A")                               # whole batches kept: every sum of every batch goes through the chains
    assert whole[3]["cache_plan"]["kind"] == "A" and torch.equal(whole[2], plain[2]) and torch.equal(whole[1], plain[1])


def test_r101_chains_longer_than_the_kernel_takes_are_cut_and_change_no_histogram(monkeypatch):
    """ResNet-101's third stage is 23 bottlenecks: 22 identity blocks chained on each other's ReLU output.  fq_hist2048_chain_seg
    takes 6 blocks per launch, so the cache keeps a shortcut at every sixth block and pass 2 walks chains of 6, 6, 6, 5 (and the
    cache plan counts exactly those shortcuts).  Every maximum, histogram and table line equals the plain path's."""
    from tools import Quantity
    from common.quantity import merge_bn, _native
    from model.resnet.ResNet_fabu import ResNet101
    model = merge_bn(cases.seed_model(ResNet101(), gamma_scale=0.5).eval()).cuda()
    g = torch.Generator(device="cuda").manual_seed(101)
    batches = [(torch.randn(32, 3, 224, 224, generator=g, device="cuda"), None) for _ in range(3)]
    chains = []
    real = _native.hist2048_chain_seg

    def spy(chains_, *a, **k):
        chains.extend(len(c[1]) for c in chains_)
        return real(chains_, *a, **k)
    monkeypatch.setattr(_native, "hist2048_chain_seg", spy)

    def run(cache_gb, plan, **switches):
        monkeypatch.setenv("FQ_ACT_CACHE_GB", cache_gb)
        monkeypatch.setenv("FQ_CACHE_PLAN", plan)
        with product_workdir(input_shape="1,3,224,224", device="gpu", max_cali_img_num=2) as tmp:
            q = Quantity(model)
            for k, v in switches.items():
                setattr(q, k, v)
            q.activation_quantize(batches)
            return (open(os.path.join(tmp, "test", "workdir", "feat.table")).read(), q._collector.max_device.clone(),
                    q._collector.hist_device.clone(), dict(q.timings))
    plain = run("0", "", fuse_conv_add=False, skip_unread_outputs=False, pair_hist=False, fuse_hist=False)
    assert not chains
    whole = run("16", "A")                               # 3.2 GB per batch: every batch kept whole
    assert whole[3]["cache_plan"]["kind"] == "A" and whole[3]["sums_left_to_pass2_pairs"] >= 33 * 2
    assert torch.equal(whole[1], plain[1]) and torch.equal(whole[2], plain[2]) and whole[0] == plain[0]
    assert len(whole[0].strip().split("\n")) == 139
    # per batch: stage 1 (3 blocks), stage 2 (4), stage 3 (23 = 6 + 6 + 6 + 5), stage 4 (3)
    per_batch = sorted(chains[:7])
    assert per_batch == [3, 3, 4, 5, 6, 6, 6], chains[:14]
    chains.clear()
    part = run("4", "B")                                 # the deep suffix of every batch: the first chain starts mid-stage
    assert part[3]["cache_plan"]["kind"] == "B" and part[3]["sums_left_to_pass2_pairs"] > 0 and chains
    assert torch.equal(part[1], plain[1]) and torch.equal(part[2], plain[2]) and part[0] == plain[0]
    # the plan's own account of the shortcuts it keeps holds (slack: the one forward that ran before the plan existed may hold a
    # boundary block's conv3 output privately, and a chain cut by the suffix boundary keeps one shortcut the refund assumed away)
    assert part[3]["cache_bytes"] <= 1.1 * 4 * 2 ** 30
