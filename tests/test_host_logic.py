"""Host-side logic of the drop-in (no GPU): graph discovery, merge groups, file writers, the BN fold,
the C-ABI symbol table.  Goldens come from the imported reference (tests/golden/make_golden_*.py)."""
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch

import cases
from workdir_util import product_workdir

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _golden(golden_dir, name):
    with open(os.path.join(golden_dir, name)) as fh:
        return json.load(fh)


# ---------------------------------------------------------------- C ABI
def test_c_abi_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "fq.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = re.findall(r"\b(fq_[a-z0-9_]+)\s*\(", hdr)
    assert len(names) >= 18
    lib_path = os.path.join(ROOT, "pytorch-quantity_amd", "lib", "libfq_hip.so")
    assert os.path.isfile(lib_path), "build the library first: make -C pytorch-quantity_amd/csrc"
    lib = ctypes.CDLL(lib_path)
    for n in names:
        assert hasattr(lib, n), "libfq_hip.so does not export %s" % n
    lib.fq_version.restype = ctypes.c_int
    assert lib.fq_version() == 103


def test_host_bits_helpers_match_python(oracle):
    from common.quantity import _native
    rng = np.random.default_rng(3)
    thr = rng.integers(128, 2048, 500).astype(np.int32)
    iv = np.exp(rng.uniform(-12, 3, 500)).astype(np.float32)
    iv[:8] = np.float32(2.0) ** np.arange(-8, 0)
    bits, tv = _native.bits_from_threshold(thr, iv)
    import math
    for t, i, b, v in zip(thr, iv, bits, tv):
        tb = (int(t) + 0.5) * i
        assert isinstance(tb, np.float32) and tb == v
        assert b == int(8 - 1 - math.ceil(math.log(tb, 2)))
        assert (b, v) == oracle.bits_from_threshold(int(t), i)
    m = np.concatenate([np.exp(rng.uniform(-10, 8, 300)), 2.0 ** np.arange(-10, 10)]).astype(np.float32)
    got = _native.bits_from_absmax(m)
    for x, b in zip(m, got):
        assert b == int(8 - 1 - math.ceil(math.log(x, 2))) == oracle.bits_from_absmax(x)


def test_no_cpu_fallback_for_device_ops():
    from common.quantity import _native, QuanDequan, NewAdd
    x = torch.randn(16)
    with pytest.raises(_native.FqError):
        _native.quandequan(x, 3)
    with pytest.raises(_native.FqError):
        QuanDequan(8, 3)(x)
    with pytest.raises(_native.FqError):
        NewAdd()(x, x)
    with pytest.raises(_native.FqError):
        _native.absmax_seg([x], [0], torch.zeros(1))
    q = torch.zeros(1, 2, 2, 16, dtype=torch.int8)
    for call in (lambda: _native.add_resident(q, 3, q, 3, True, 3, True, 3, False),
                 lambda: _native.dequant_nhwc_to_nchw(q, 3, 16),
                 lambda: _native.maxpool_i8_nhwc(q, (2, 2), (2, 2), (0, 0)),
                 lambda: _native.avgpool_global_nhwc(q, 3, 16),
                 lambda: _native.conv2d_i8_resident(q, q, torch.zeros(1), (1, 1), (0, 0), (1, 1), 8, 3, True, True, False)):
        with pytest.raises(_native.FqError):
            call()


def test_missing_library_fails_loudly(monkeypatch):
    """No lazy build, no fallback: if libfq_hip.so is absent every entry point raises FqError."""
    from common.quantity import _native
    monkeypatch.setattr(_native, "_lib", None)
    monkeypatch.setattr(_native, "LIB_PATH", "/nonexistent/libfq_hip.so")
    with pytest.raises(_native.FqError, match="no CPU fallback"):
        _native.lib()
    with pytest.raises(_native.FqError):
        _native.json_dump_i32([1, 2], "/tmp/should_not_exist.json")


# ---------------------------------------------------------------- JSON writer
@pytest.mark.parametrize("shape", [(), (5,), (1,), (3, 4), (2, 3, 4, 5), (4, 1, 1, 1), (2, 0), (0,), (0, 3), (3, 0, 2),
                                   (64, 3, 7, 7), (10, 512)])
def test_json_writer_is_byte_identical_to_json_dump(tmp_path, shape):
    from tools import _jsonio
    rng = np.random.default_rng(len(shape))
    a = rng.integers(-40000, 40000, size=shape).astype(np.int32)
    ref = json.dumps(a.tolist(), indent=4)
    p = str(tmp_path / "x.json")
    _jsonio.dump_int_array(a, p)
    assert open(p).read() == ref
    assert _jsonio.dumps_int_array(a) == ref


# ---------------------------------------------------------------- merge_bn (G8)
@pytest.mark.parametrize("tag", ["nobias", "bias"])
def test_merge_bn_matches_reference(golden_dir, tag):
    import torch.nn as nn
    from common.quantity import merge_bn
    g = np.load(os.path.join(golden_dir, "g8_merge_bn.npz"))
    seq = nn.Sequential(nn.Conv2d(3, 8, 3, padding=1, bias=(tag == "bias")), nn.BatchNorm2d(8), nn.ReLU(False))
    sd = {k[len(tag) + 5:]: torch.from_numpy(g[k]) for k in g.files if k.startswith(tag + "/pre/")}
    seq.load_state_dict(sd)
    seq.eval()
    merged = merge_bn(seq)
    assert type(merged[1]).__name__ == str(g[tag + "/bn_type_after"]) == "Identity"
    np.testing.assert_array_equal(merged[0].weight.detach().numpy(), g[tag + "/w"])
    np.testing.assert_array_equal(merged[0].bias.detach().numpy(), g[tag + "/b"])
    np.testing.assert_array_equal(merged(torch.from_numpy(g[tag + "/x"])).detach().numpy(), g[tag + "/y_merged"])


# ---------------------------------------------------------------- graph discovery (G7)
def _discover(ctor, shape, **seed_kw):
    from common.quantity import merge_bn
    from tools import Quantity
    with product_workdir(input_shape=shape, device="cpu"):
        model = merge_bn(cases.seed_model(ctor(), **seed_kw).eval())
        q = Quantity(model)
        return {"net_info": dict(q.net_info), "net_info_order": list(q.net_info.keys()),
                "cared_op_layer_names": q.cared_op_layer_names, "merge_groups": q.get_merge_groups(q.net_info),
                "layers_num": q.layers_num}


@pytest.mark.parametrize("tag", ["r50", "r101"])
def test_graph_discovery_matches_reference_bottleneck(golden_dir, tag):
    from model.resnet.ResNet_fabu import ResNet50, ResNet101
    ref = _golden(golden_dir, "g7_netinfo.json")[tag]
    got = _discover(ResNet50 if tag == "r50" else ResNet101, "1,3,224,224", gamma_scale=0.5)
    for key in ("net_info_order", "cared_op_layer_names", "merge_groups", "layers_num", "net_info"):
        assert got[key] == ref[key], key


def test_graph_discovery_survives_exploding_activations():
    """The reference's value fingerprints collide on a deep net once activations grow (it raises
    'Same input and output id' on ResNet-101 with unit-scale BN); identity tracking does not."""
    from model.resnet.ResNet_fabu import ResNet101
    got = _discover(lambda: ResNet101(input_size=64), "1,3,64,64")
    assert len(got["net_info_order"]) == 138 and len(got["merge_groups"]) == 33


# ---------------------------------------------------------------- rewriter (G6)
def test_rewriter_matches_reference(golden_dir, tmp_path):
    from tools import BiasReWriter
    g = _golden(golden_dir, "g6_rewriter.json")
    d = str(tmp_path)
    for sub in ("weight", "bias", "new_weight", "new_bias"):
        os.makedirs(os.path.join(d, sub))
    for fname in ("feat.table", "weight.table"):
        with open(os.path.join(d, fname), "w") as fh:
            fh.write(g["inputs"][fname])
    for rel, content in g["inputs"]["params"].items():
        with open(os.path.join(d, rel), "w") as fh:
            json.dump(content, fh, indent=4)
    rw = BiasReWriter(os.path.join(d, "weight"), os.path.join(d, "bias"), os.path.join(d, "new_weight"),
                      os.path.join(d, "new_bias"), os.path.join(d, "weight.table"), os.path.join(d, "feat.table"),
                      max_shift_limit=12)
    weight_bits, bias_bits = rw.get_weight_info()
    feat_bits, infeat_bits = rw.get_feat_info()
    rw.rewrite_bias_table(bias_bits, feat_bits)
    rw.rewrite_bias_dir(bias_bits, feat_bits)
    need, new_w = rw.max_shift_limit_weight(feat_bits, infeat_bits, weight_bits)
    assert need == g["need_rewrite"] and new_w == g["new_weight_bits"]
    rw.rewrite_weight_table(weight_bits, new_w)
    rw.rewrite_weight_dir(weight_bits, new_w)
    assert open(os.path.join(d, "weight.table")).read() == g["weight.table"]
    for rel, text in g["files"].items():
        assert open(os.path.join(d, rel)).read() == text, rel


# ---------------------------------------------------------------- input side: PRE_PROCESS.IMG = 2 (.npy files)
def test_npy_input_mode_equals_loader_mode(tmp_path, oracle):
    """pytorch_quantizer.py:276-280: each calibration item is the path of a .npy holding one CHW image."""
    import yaml
    from common.quantity import merge_bn
    from engine_doubles import OracleCollector, OracleQuantizer
    from model.resnet.ResNet_18_fabu import ResNet18
    from tools import Quantity

    class CpuQuantity(Quantity):
        collector_cls = OracleCollector
        quantizer_cls = OracleQuantizer

    imgs = [cases.fixed_input((3, 32, 32), seed=500 + i) for i in range(3)]
    tables = []
    for mode in (1, 2):
        with product_workdir(device="cpu", max_cali_img_num=2) as tmp:
            ucfg_path = os.path.join(tmp, "test", "user_configs.yml")
            ucfg = yaml.safe_load(open(ucfg_path))
            ucfg["PRE_PROCESS"]["IMG"] = mode
            yaml.safe_dump(ucfg, open(ucfg_path, "w"))
            if mode == 1:
                items = [(im[None], torch.zeros(1, dtype=torch.long)) for im in imgs]
            else:
                items = []
                for i, im in enumerate(imgs):
                    path = str(tmp_path / ("img%d.npy" % i))
                    np.save(path, im.numpy())
                    items.append(path)
            q = CpuQuantity(merge_bn(cases.seed_model(ResNet18()).eval()))
            q.activation_quantize(items)
            tables.append(open(os.path.join(tmp, "test", "workdir", "feat.table")).read())
    assert tables[0] == tables[1] and tables[0].startswith("image ")


# ---------------------------------------------------------------- H6: dilation -> zero padded kernel
def test_dilation_to_zero_padding(golden_dir):
    """pytorch_quantizer.py:679-693: a k x k kernel with dilation 2 as a dense (2k-1) x (2k-1) kernel -- against the
    reference function's own outputs (golden G10, tests/golden/make_golden_r50.py h6) and against what it means."""
    from tools import Quantity
    g10 = np.load(os.path.join(golden_dir, "g10_dilation.npz"))
    for tag in ("k3", "k1", "k5", "k2"):
        got = Quantity.dilation_to_zero_padding(None, torch.from_numpy(g10[tag + "_in"]), (2, 2))
        assert got.dtype == torch.float32
        np.testing.assert_array_equal(got.numpy(), g10[tag + "_out"])
        got_np = Quantity.dilation_to_zero_padding(None, g10[tag + "_in"], (2, 2))          # the reference passes ndarrays
        np.testing.assert_array_equal(np.asarray(got_np), g10[tag + "_out"])
    w = torch.arange(2 * 3 * 3 * 3, dtype=torch.float32).reshape(2, 3, 3, 3) + 1
    dense = Quantity.dilation_to_zero_padding(None, w, (2, 2))
    assert dense.shape == (2, 3, 5, 5)
    assert torch.equal(dense[..., ::2, ::2], w)
    x = torch.randn(1, 3, 9, 9)
    ref = torch.nn.functional.conv2d(x, w, dilation=2)
    got = torch.nn.functional.conv2d(x, dense)
    assert torch.allclose(ref, got, atol=1e-4)
    with pytest.raises(AssertionError):
        Quantity.dilation_to_zero_padding(None, w, (3, 3))


def test_testconv_visualization_dump_is_optional(tmp_path, monkeypatch):
    """TestConv's constructor side effects (reference new_quantity_op.py:312-337): text dumps and PNG
    histograms are written only when asked for; the results directory is always created."""
    import torch.nn as nn
    from common.quantity import new_quantity_op as nq
    info = dict(weight_bit=7, bias_bit=4, input_bit=4, output_bit=4)
    path = str(tmp_path / "m.pth")
    nq.TestConv("a.b", nn.Conv2d(3, 4, 3), info, path)
    res = tmp_path / "quantity_results"
    assert res.is_dir() and not list(res.iterdir())
    monkeypatch.setattr(nq, "DUMP_VISUALIZATION", True)
    layer = nq.TestLinear("fc", nn.Linear(8, 3), info, path)
    names = sorted(p.name for p in res.iterdir())
    assert names == ["fc_bias.txt", "fc_bias_o.png", "fc_bias_q.png", "fc_weight.txt", "fc_weight_o.png", "fc_weight_q.png"]
    # weights were fake-quantised at construction: multiples of 2^-7 within the int8 range
    w = layer.linear.weight.detach() * 128
    assert torch.equal(w, torch.round(w)) and w.abs().max() <= 128


def test_image_file_input_mode(tmp_path):
    """PRE_PROCESS.IMG = 0: decode -> BGR -> resize -> minus MEAN -> [1,3,H,W] float32."""
    import yaml
    from PIL import Image
    from tools import Quantity
    rng = np.random.default_rng(1)
    rgb = rng.integers(0, 256, (32, 32, 3), dtype=np.uint8)
    p_same = str(tmp_path / "a.png")
    Image.fromarray(rgb).save(p_same)
    big = rng.integers(0, 256, (64, 48, 3), dtype=np.uint8)
    p_big = str(tmp_path / "b.png")
    Image.fromarray(big).save(p_big)
    with product_workdir(device="cpu") as tmp:
        ucfg_path = os.path.join(tmp, "test", "user_configs.yml")
        ucfg = yaml.safe_load(open(ucfg_path))
        ucfg["PRE_PROCESS"]["IMG"] = 0
        yaml.safe_dump(ucfg, open(ucfg_path, "w"))
        q = Quantity.__new__(Quantity)
        q.user_config = ucfg
        x = q.preprocess(p_same)
        assert x.shape == (1, 3, 32, 32) and x.dtype == torch.float32
        np.testing.assert_array_equal(x[0].numpy(), rgb[:, :, ::-1].transpose(2, 0, 1).astype(np.float32) - 128.0)
        y = q.preprocess(p_big)
        assert y.shape == (1, 3, 32, 32) and y.min() >= -128 and y.max() <= 127
        assert torch.equal(y, torch.round(y))
        assert q.preprocess(str(tmp_path / "missing.png")) is False


def test_eager_stats_groups_launches_and_notices_inplace_consumers():
    """tools.pytorch_quantizer._EagerStats (host logic of the in-hook statistics launches) on CPU tensors."""
    import torch
    from tools._hook_state import _AFTER_FORWARD, _EagerStats
    calls = []
    fn = lambda tensors: calls.append(list(tensors))
    a, b, c = torch.ones(100), torch.ones(100), torch.ones(100)

    e = _EagerStats(fn, 0)                                   # one launch per tensor, from the hook
    e.add("a", a); e.add("b", b)
    assert calls == [["a"], ["b"]] and not e.modified()
    a.mul_(2.0)                                              # an in-place consumer AFTER the statistics were taken
    assert e.modified()
    e.flush()                                                # nothing pending: no call
    assert len(calls) == 2

    calls.clear()
    e = _EagerStats(fn, 800)                                 # 400-byte tensors: two per launch
    e.add("a", a); assert calls == []
    e.add("b", b); assert calls == [["a", "b"]]
    e.add("c", c)
    e.flush(extra={"kept": torch.zeros(3)})                  # end of forward: the rest plus tensors kept from pass 1
    assert calls == [["a", "b"], ["c", "kept"]]

    calls.clear()
    e = _EagerStats(fn, _AFTER_FORWARD)                      # the default for models without in-place consumers
    e.add("a", a); e.add("b", b)
    assert calls == []
    b.add_(1.0)                                              # ... which such a model must not do
    with pytest.raises(RuntimeError):
        e.flush()


def test_float_conv_dispatch_declines_what_the_kernels_do_not_take():
    """common/quantity/_float_conv.py on CPU tensors and unsupported layers: kind() says None, call() is plain m(x) (hooks
    fire once, nothing stays attached to the module), TestConv therefore behaves as in the reference off the GPU."""
    import torch
    from torch import nn
    from common.quantity import _float_conv
    x = torch.randn(2, 8, 5, 5)
    for conv in (nn.Conv2d(8, 16, 1), nn.Conv2d(8, 16, 3, padding=1), nn.Conv2d(8, 16, 1, bias=False), nn.Conv2d(8, 16, 1, groups=2)):
        conv.eval()
        assert _float_conv.kind(conv, x) is None
        seen = []
        h = conv.register_forward_hook(lambda m, i, o: seen.append(1))
        with torch.no_grad():
            assert torch.equal(_float_conv.call(conv, x), conv(x))
        h.remove()
        assert len(seen) == 2 and "forward" not in conv.__dict__ and not _float_conv.is_verified(conv)


def test_deferral_probe_finds_keepers_of_a_tensor_in_one_heap_pass():
    """tools._hook_state._DeferralProbe.holders (host logic, CPU tensors): whoever still refers to a watched tensor and is not one
    of the calibration's own containers -- a module attribute, a user's list, a view -- is reported against that tensor; clean
    tensors are not; and the scan is ONE gc pass for all tensors (a pass per tensor cost 0.4 s of a 0.5 s calibration)."""
    import time
    import torch
    from tools._hook_state import _DeferralProbe

    class Holder(object):
        pass
    clean, stashed, listed, viewed = (torch.zeros(4, 4) for _ in range(4))
    ours = {"a": clean, "b": stashed, "c": listed, "d": viewed}         # what the calibration itself keeps
    user, user_list = Holder(), []
    user.feat = stashed
    user_list.append(listed)
    view = viewed[:2]
    every = [clean, stashed, listed, viewed]
    got = _DeferralProbe.holders(every, [ours, every])
    assert id(clean) not in got
    assert got[id(stashed)] == ["a dict"] and got[id(listed)] == ["a list"] and got[id(viewed)] == ["a view of it"]
    del view
    many = [torch.zeros(2) for _ in range(64)]
    t0 = time.perf_counter()
    assert _DeferralProbe.holders(many, [many]) == {}
    one_pass = time.perf_counter() - t0
    t0 = time.perf_counter()
    for t in many[:8]:
        _DeferralProbe.holders([t], [many])
    assert one_pass < (time.perf_counter() - t0) * 2                     # 64 tensors in one pass: cheaper than 16 single scans


def test_graph_discovery_reads_values_only_for_an_edge_identity_does_not_show(monkeypatch):
    """The value fingerprint (tid: four reductions and four host reads per tensor) is the fallback matcher: a net whose
    edges all show as tensor identity never computes one -- on the GPU that was 123 x 4 synchronisations per Quantity(model) --
    and a tensor that was re-wrapped between two modules (here: multiplied by one) is still found through it."""
    from torch import nn
    from tools import Quantity, pytorch_quantizer
    calls = []
    real = pytorch_quantizer.tid
    monkeypatch.setattr(pytorch_quantizer, "tid", lambda t: (calls.append(1), real(t))[1])
    got = _discover(cases.tiny_concat_net, "1,3,8,8")
    assert calls == [] and len(got["net_info_order"]) > 5

    class Scaled(nn.Module):
        def __init__(self):
            super(Scaled, self).__init__()
            self.conv = nn.Conv2d(3, 4, 1)
            self.relu = nn.ReLU()
            self.conv2 = nn.Conv2d(4, 4, 1)

        def forward(self, x):
            return self.conv2(self.relu(self.conv(x) * torch.ones(4, 1, 1)))
    with product_workdir(input_shape="1,3,8,8", device="cpu"):
        q = Quantity(Scaled().eval())
    assert q.layers_num == 3 and q.net_info["Conv2d_3"]["inputs"] == ["Conv2d_1"]        # (through ReLU_2, which is pruned)
    assert len(calls) == 2                                   # the one output before ReLU_2, and its input
