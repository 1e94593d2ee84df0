"""Seeded input generators shared by the golden-capture script (make_golden_kernels.py) and by the
tests that replay the same inputs through the oracle and through the HIP path.

Every generator is pure numpy with a fixed PCG64 seed, so the inputs regenerate bit-identically
wherever numpy is the same major version; the fixtures also carry a sha256 of each input so a
silent generator drift is caught instead of producing a bogus parity failure.
"""
import hashlib

import numpy as np

BINS = 2048


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _rng(seed):
    return np.random.default_rng(seed)


# --------------------------------------------------------------------------------------------
# G1: activation-like tensors for abs-max / interval / 2048-bin histogram
#   each case = list of batches for pass 1 (max) and list of batches for pass 2 (histogram);
#   normally the same list (the reference runs the same images twice).
# --------------------------------------------------------------------------------------------

def g1_cases():
    c = {}

    def normal(seed, n, scale=1.0):
        return (_rng(seed).standard_normal(n, dtype=np.float32) * np.float32(scale)).astype(np.float32)

    c["normal_1000"] = dict(p1=[normal(1, 1000)])
    c["normal_100352"] = dict(p1=[normal(2, 100352)])
    c["normal_802816"] = dict(p1=[normal(3, 802816, 3.7)])
    c["normal_3batch"] = dict(p1=[normal(4, 100352), normal(5, 100352, 1.3), normal(6, 100352, 0.7)])
    lap = _rng(7).laplace(0.0, 1.0, 100352).astype(np.float32)
    c["laplace_100352"] = dict(p1=[lap])
    c["relu_sparse"] = dict(p1=[np.maximum(normal(8, 100352), 0).astype(np.float32)])
    c["all_zero"] = dict(p1=[np.zeros(1000, dtype=np.float32)])
    o = normal(9, 50000, 0.01)
    o[12345] = np.float32(1000.0)
    c["single_outlier"] = dict(p1=[o])
    c["n1"] = dict(p1=[np.array([-2.5], dtype=np.float32)])
    c["n1_zero"] = dict(p1=[np.array([0.0], dtype=np.float32)])
    c["tiny_values"] = dict(p1=[normal(10, 4096, 1e-10)])           # 1e-12 term of the interval matters
    c["negative_only"] = dict(p1=[-np.abs(normal(11, 30000))])
    c["ragged_1023"] = dict(p1=[normal(12, 1023, 5.0)])
    c["ragged_4099"] = dict(p1=[normal(13, 4099, 0.3)])
    # pass 2 sees larger values than pass 1 did (shuffle=True loader quirk): clamp to the last bin
    c["pass2_exceeds"] = dict(p1=[normal(14, 20000)], p2=[normal(15, 20000, 2.0)])
    # integers on exact bin edges: max = 2048 -> interval = 1 + 1e-12 -> fl32 = 1.0
    e = np.arange(0, 2049, dtype=np.float32)
    c["exact_edges"] = dict(p1=[e])
    # uniform in (0, 1): every bin populated
    c["uniform"] = dict(p1=[_rng(16).random(300000, dtype=np.float32)])
    for k, v in c.items():
        v.setdefault("p2", v["p1"])
    return c


# --------------------------------------------------------------------------------------------
# G2: histograms for the KL threshold sweep.  dtype int32 = as the collector returns them;
#     float64 = the merged-group form (np.zeros(2048) += int32 hists).
# --------------------------------------------------------------------------------------------

def g2_cases():
    h = {}
    j = np.arange(BINS, dtype=np.float64)
    r = _rng(100)

    def poisson(lam, seed):
        return _rng(seed).poisson(lam).astype(np.int32)

    h["gauss_s300"] = poisson(2e4 * np.exp(-0.5 * (j / 300.0) ** 2), 101)
    h["gauss_s80"] = poisson(5e4 * np.exp(-0.5 * (j / 80.0) ** 2), 102)
    h["gauss_s700"] = poisson(3e3 * np.exp(-0.5 * (j / 700.0) ** 2), 103)
    h["laplace_b60"] = poisson(1e5 * np.exp(-j / 60.0), 104)
    h["laplace_b250"] = poisson(2e4 * np.exp(-j / 250.0), 105)
    h["heavy_tail"] = poisson(1e5 / (1.0 + (j / 20.0) ** 2), 106)
    spike = np.zeros(BINS, dtype=np.int32)
    spike[37] = 1000
    h["spike_low"] = spike
    spike2 = np.zeros(BINS, dtype=np.int32)
    spike2[1500] = 77
    h["spike_high"] = spike2
    two = np.zeros(BINS, dtype=np.int32)
    two[5] = 900
    two[2047] = 3
    h["spike_plus_lastbin"] = two
    h["uniform_const"] = np.full(BINS, 123, dtype=np.int32)
    h["uniform_noise"] = poisson(np.full(BINS, 150.0), 107)
    h["empty"] = np.zeros(BINS, dtype=np.int32)
    big = poisson(3e3 * np.exp(-0.5 * (j / 200.0) ** 2), 108).astype(np.int64) * 40000  # counts > 2^24
    h["big_counts"] = np.minimum(big, 2**31 - 1).astype(np.int32)
    sp = np.zeros(BINS, dtype=np.int32)
    idx = r.choice(BINS, 90, replace=False)
    sp[idx] = r.integers(1, 50, 90)
    h["sparse_random"] = sp
    first = np.zeros(BINS, dtype=np.int32)
    first[:128] = poisson(np.full(128, 400.0), 109)
    h["first128_only"] = first
    lastheavy = poisson(2e4 * np.exp(-0.5 * (j / 150.0) ** 2), 110)
    lastheavy[2047] = 50000
    h["lastbin_heavy"] = lastheavy
    h["relu_like"] = poisson(8e4 * np.exp(-j / 35.0) + 30.0 * np.exp(-0.5 * ((j - 900) / 200.0) ** 2), 111)
    h["bimodal"] = poisson(4e3 * np.exp(-0.5 * ((j - 200) / 60.0) ** 2)
                           + 2e3 * np.exp(-0.5 * ((j - 1200) / 150.0) ** 2), 112)
    ones = np.ones(BINS, dtype=np.int32)
    h["all_ones"] = ones
    few = np.zeros(BINS, dtype=np.int32)
    few[[0, 1, 2, 130, 131, 700]] = [5, 1, 1, 2, 1, 1]
    h["few_samples"] = few
    # merged-group (float64) forms
    h["merged_f64_a"] = (h["gauss_s300"].astype(np.float64) + h["laplace_b60"].astype(np.float64))
    h["merged_f64_b"] = (h["gauss_s80"].astype(np.float64) + h["heavy_tail"].astype(np.float64)
                         + h["uniform_noise"].astype(np.float64))
    h["merged_f64_big"] = h["big_counts"].astype(np.float64) * 3.0   # > 2^31 per bin, still integer valued
    return h


def g2_intervals():
    """One fp32 interval per G2 case (only used for bits / threshold_value)."""
    r = _rng(200)
    names = list(g2_cases().keys())
    iv = {}
    for i, n in enumerate(names):
        m = np.float32(np.exp(r.uniform(-4.0, 6.0)))
        iv[n] = np.float32(1) * m / 2048 + 1e-12    # same expression shape as the reference
        assert isinstance(iv[n], np.float32)
    # exact powers of two for (t+0.5)*interval are unreachable (t+0.5 is odd/2), keep as is
    return iv


# --------------------------------------------------------------------------------------------
# G5: element-wise op vectors (ties, negatives, saturation, negative shifts)
# --------------------------------------------------------------------------------------------

def g5_inputs():
    r = _rng(300)
    base = np.array([0.0, -0.0, 0.5, -0.5, 1.5, -1.5, 2.5, -2.5, 0.49999997, -0.49999997,
                     126.5, 127.5, 128.5, -127.5, -128.5, -129.5, 1e-8, -1e-8, 300.0, -300.0,
                     3.0, -3.0, 1.0, -1.0, 0.25, -0.25, 0.75, -0.75, 63.5, -63.5,
                     32767.5, -32768.5, 40000.0, -40000.0], dtype=np.float32)
    rnd = (r.standard_normal(2000, dtype=np.float32) * np.float32(40.0)).astype(np.float32)
    ints = r.integers(-70000, 70000, 1000).astype(np.float32)      # integer-valued accumulators
    halves = (r.integers(-600, 600, 500).astype(np.float32) + np.float32(0.5))
    return np.concatenate([base, rnd, ints, halves]).astype(np.float32)


# --------------------------------------------------------------------------------------------
# End-to-end fixtures: deterministic weights / data that do not depend on module construction order
# --------------------------------------------------------------------------------------------

def seed_model(model, base_seed=0, gamma_scale=1.0):
    """Fill every parameter and BatchNorm statistic from a generator keyed by the tensor's
    state_dict name, so the reference's model file and ours get identical values.  NumPy's PCG64
    streams are used (not torch.randn, whose CPU kernels are not guaranteed bit-identical across
    CPU micro-architectures): the GPU box must regenerate exactly the weights the goldens saw."""
    import zlib

    import torch
    sd = model.state_dict()
    with torch.no_grad():
        for key in sorted(sd.keys()):
            t = sd[key]
            if not t.is_floating_point():
                continue
            g = np.random.default_rng(base_seed * 1000003 + zlib.crc32(key.encode()))
            shape = tuple(t.shape)
            if key.endswith("running_var"):
                v = g.random(shape, dtype=np.float32) + np.float32(0.5)
            elif key.endswith("running_mean"):
                v = g.standard_normal(shape, dtype=np.float32) * np.float32(0.1)
            elif t.dim() == 1 and key.endswith("weight"):
                v = (g.random(shape, dtype=np.float32) + np.float32(0.5)) * np.float32(gamma_scale)
            elif t.dim() == 1:
                v = g.standard_normal(shape, dtype=np.float32) * np.float32(0.1)
            else:
                fan_in = int(np.prod(shape[1:]))
                v = g.standard_normal(shape, dtype=np.float32) * np.float32((2.0 / fan_in) ** 0.5)
            t.copy_(torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).to(t.dtype))
    return model


def fold_bn_with_scales(model, scales):
    """merge_bn (reference quantity/common/quantity/utils.py:7-65) with the per-channel factor
    gamma / sqrt(running_var + 1e-5) TAKEN FROM A FIXTURE instead of recomputed.  torch.sqrt on CPU tensors goes through
    MKL's vector math library, whose last bit differs between Intel and AMD hosts (measured: build container vs GPU box,
    scripts/_dbg notes in DESIGN.md), so a model folded on the GPU box is not bit-identical to the one the reference
    folded in the build container -- and a one-ulp difference in a weight can flip its 8-bit rounding.  Given the same
    factors, the remaining operations (one multiply per weight, multiply-add per bias) are single IEEE operations and
    identical everywhere; they are done in NumPy here.  `scales`: {BatchNorm module name: float32[C]}.
    Returns the model with every BatchNorm2d replaced by the drop-in's Identity (as merge_bn does)."""
    import torch
    import torch.nn as nn
    from common.quantity import Identity
    pending, folded = None, []
    for name, layer in list(model.named_modules()):
        kind = type(layer).__name__
        if kind == "Conv2d":
            pending = layer
        elif kind == "BatchNorm2d":
            conv, pending = pending, None
            sc = np.asarray(scales[name], dtype=np.float32)
            w = conv.weight.data.numpy()
            b = conv.bias.data.numpy() if conv.bias is not None else np.zeros(conv.out_channels, dtype=np.float32)
            mean, beta = layer.running_mean.numpy(), layer.bias.data.numpy()
            conv.weight = nn.Parameter(torch.from_numpy(sc.reshape(-1, 1, 1, 1) * w))
            conv.bias = nn.Parameter(torch.from_numpy(sc * (b - mean) + beta))
            folded.append(name)
    for name in folded:
        parent = model
        parts = name.split(".")
        for part in parts[:-1]:
            parent = getattr(parent, part)
        parent.add_module(parts[-1], Identity())
    return model


def calib_batches(n_batches, shape, seed=1234):
    """List of (images, labels) pairs as a DataLoader would yield them (PRE_PROCESS.IMG = 1)."""
    import torch
    out = []
    for i in range(n_batches):
        g = np.random.default_rng(seed + i)
        out.append((torch.from_numpy(g.standard_normal(tuple(shape), dtype=np.float32)),
                    torch.zeros(shape[0], dtype=torch.long)))
    return out


def fixed_input(shape, seed=99):
    import torch
    return torch.from_numpy(np.random.default_rng(seed).standard_normal(tuple(shape), dtype=np.float32))


def tiny_vgg_net():
    """A plain stack -- 3x3 convolutions, max-pools, a two-layer classifier, no residual -- at widths the own float kernels take
    (16 / 64 channels: the direct R x S kernel and, for 64 -> 64 and 64 -> 128, the Winograd one)."""
    import torch.nn as nn
    from common.quantity import View

    class TinyVgg(nn.Module):
        def __init__(self):
            super(TinyVgg, self).__init__()
            self.c1 = nn.Conv2d(3, 16, 3, padding=1)
            self.r1 = nn.ReLU(False)
            self.c2 = nn.Conv2d(16, 64, 3, padding=1)
            self.r2 = nn.ReLU(False)
            self.p1 = nn.MaxPool2d(2, 2)
            self.c3 = nn.Conv2d(64, 64, 3, padding=1)
            self.r3 = nn.ReLU(False)
            self.c4 = nn.Conv2d(64, 128, 3, padding=1)
            self.r4 = nn.ReLU(False)
            self.p2 = nn.MaxPool2d(2, 2)
            self.c5 = nn.Conv2d(128, 32, 1)
            self.r5 = nn.ReLU(False)
            self.view = View()
            self.f1 = nn.Linear(32 * 4 * 4, 24)
            self.r6 = nn.ReLU(False)
            self.f2 = nn.Linear(24, 10)

        def forward(self, x):
            x = self.p1(self.r2(self.c2(self.r1(self.c1(x)))))
            x = self.p2(self.r4(self.c4(self.r3(self.c3(x)))))
            x = self.view(self.r5(self.c5(x)))
            return self.f2(self.r6(self.f1(x)))

    return TinyVgg()


def tiny_separable_net():
    """Depthwise-separable blocks (a grouped 3x3 convolution the own kernels do not take, then a 1x1 one they do), a stride-2
    depthwise layer, an identity shortcut, a global average pool: what a MobileNet is made of."""
    import torch.nn as nn
    from common.quantity import Eltwise, View

    class TinySeparable(nn.Module):
        def __init__(self):
            super(TinySeparable, self).__init__()
            self.stem = nn.Conv2d(3, 16, 3, stride=2, padding=1)
            self.r0 = nn.ReLU(False)
            self.dw1 = nn.Conv2d(16, 16, 3, padding=1, groups=16)
            self.r1 = nn.ReLU(False)
            self.pw1 = nn.Conv2d(16, 32, 1)
            self.r2 = nn.ReLU(False)
            self.dw2 = nn.Conv2d(32, 32, 3, stride=2, padding=1, groups=32)
            self.r3 = nn.ReLU(False)
            self.pw2 = nn.Conv2d(32, 64, 1)
            self.r4 = nn.ReLU(False)
            self.dw3 = nn.Conv2d(64, 64, 3, padding=1, groups=64)
            self.r5 = nn.ReLU(False)
            self.pw3 = nn.Conv2d(64, 64, 1)
            self.Eltwise = Eltwise()
            self.r6 = nn.ReLU(False)
            self.pool = nn.AvgPool2d(4)
            self.view = View()
            self.fc = nn.Linear(64, 7)

        def forward(self, x):
            x = self.r0(self.stem(x))
            x = self.r2(self.pw1(self.r1(self.dw1(x))))
            x = self.r4(self.pw2(self.r3(self.dw2(x))))
            y = self.pw3(self.r5(self.dw3(x)))
            x = self.r6(self.Eltwise(y, x))
            return self.fc(self.view(self.pool(x)))

    return TinySeparable()


def tiny_concat_net():
    """Small net with a Concat fed by two convolutions and an Eltwise fed by a Concat consumer and a
    convolution: covers the Concat merge group (shared interval, pooled histogram)."""
    import torch.nn as nn
    from common.quantity import Concat, Eltwise, View

    class TinyConcatNet(nn.Module):
        def __init__(self):
            super(TinyConcatNet, self).__init__()
            self.stem = nn.Conv2d(3, 8, 3, padding=1)
            self.relu0 = nn.ReLU(False)
            self.branch_a = nn.Conv2d(8, 8, 3, padding=1)
            self.branch_b = nn.Conv2d(8, 8, 1)
            self.Concat = Concat()
            self.relu1 = nn.ReLU(False)
            self.mix = nn.Conv2d(16, 8, 3, padding=1)
            self.skip = nn.Conv2d(8, 8, 1)
            self.Eltwise = Eltwise()
            self.relu2 = nn.ReLU(False)
            self.pool = nn.AvgPool2d(8)
            self.view = View()
            self.fc = nn.Linear(8, 5)

        def forward(self, x):
            s = self.relu0(self.stem(x))
            c = self.relu1(self.Concat(self.branch_a(s), self.branch_b(s)))
            y = self.relu2(self.Eltwise(self.mix(c), self.skip(s)))
            return self.fc(self.view(self.pool(y)))

    return TinyConcatNet()


# ---------------------------------------------------------------- random topologies (scripts/model_fuzz.py, golden G11)
import torch.nn as nn  # noqa: E402


class RandomNet(nn.Module):
    """A random graph: `plan` is a list of steps over a dictionary of live tensors; modules are attributes m0, m1, ..."""

    def __init__(self, rng, size, odd=False, share=False, bn=False):
        super(RandomNet, self).__init__()
        from common.quantity import Eltwise, Concat, View      # (the product's, or the reference's when a golden is captured)
        self.plan, self.n = [], 0
        self.cin = rng.choice([1, 3, 3, 4]) if odd else 3       # (odd: grey-scale and four-channel images too)
        ch = {"x": self.cin}
        hw = {"x": size}
        cur = "x"

        import random as _random
        rng_bn = _random.Random(rng.random() + 1.0) if bn else None
        self.has_bn = bool(bn)

        def add(module):
            name = "m%d" % self.n
            self.n += 1
            setattr(self, name, module)
            return name

        def conv(src, cout, k=None, s=1):
            k = k if k is not None else rng.choice([1, 1, 3, 3, 5])
            if hw[src] // s < 2:
                s = 1
            # (odd: now and then a layer the own kernels do not take -- depthwise, dilated -- which stays on the library in the
            #  middle of a fused forward)
            kw, pad = {}, k // 2
            if odd and k == 3 and rng.random() < 0.25:
                if rng.random() < 0.5 and cout == ch[src]:
                    kw["groups"] = cout
                elif hw[src] >= 8:
                    kw["dilation"], pad = 2, 2
            # bn: a BatchNorm2d behind about half of the convolutions (registered right after its convolution, as merge_bn pairs
            # them), some of those convolutions without a bias; its own generator, so that the other graphs stay what they were
            with_bn = rng_bn is not None and rng_bn.random() < 0.5
            if with_bn and rng_bn.random() < 0.5:
                kw["bias"] = False
            m = add(nn.Conv2d(ch[src], cout, k, stride=s, padding=pad, **kw))
            out = "t%d" % self.n
            self.plan.append(("call", m, [src], out))
            ch[out], hw[out] = cout, (hw[src] + 2 * pad - (kw.get("dilation", 1) * (k - 1) + 1)) // s + 1
            if with_bn:
                b = add(nn.BatchNorm2d(cout))
                o2 = "t%d" % self.n
                self.plan.append(("call", b, [out], o2))
                ch[o2], hw[o2] = ch[out], hw[out]
                out = o2
            return out

        inplace_p = 0.3 if rng.random() < 0.2 else 0.0      # one model in five has in-place ReLUs (everything then runs per tensor)

        # share: ONE nn.ReLU module serves several places of the graph, as torchvision's blocks write it (`self.relu` three times
        # in a Bottleneck): a second generator decides, so that the graphs drawn without it stay what they were
        import random as _random
        rng_share = _random.Random(rng.random()) if share else None
        shared = []

        def relu(src):
            if rng_share is not None and shared and rng_share.random() < 0.6:
                m = rng_share.choice(shared[-2:])
                self.n += 1                                   # (keeps the tensor names unique)
            else:
                m = add(nn.ReLU(rng.random() < inplace_p))
                shared.append(m)
            out = "t%d" % self.n
            self.plan.append(("call", m, [src], out))
            ch[out], hw[out] = ch[src], hw[src]
            return out

        widths = [8, 16, 64, 128, 128, 256]                 # (fq_conv1x1_add_f32 takes Cin % 16 == 0, Cout % 128 == 0)
        if odd:                                             # ... and widths that are no multiple of 8 or 16: partial tiles, padded channels
            widths = [8, 12, 20, 36, 64, 100, 128, 256]
        cur = relu(conv("x", rng.choice([8, 16, 64] if not odd else [8, 12, 20, 64]), k=rng.choice([3, 5, 7]), s=rng.choice([1, 2])))
        for _ in range(rng.randint(2, 5)):
            kind = rng.choice(["plain", "res", "res", "resproj", "concat", "pool", "twice", "shared", "bneck2", "projhead"])
            c = ch[cur]

            def bottleneck(src, mid, cout, project):
                y = relu(conv(src, mid, k=1))
                y = relu(conv(y, mid, k=3))
                y = conv(y, cout, k=1)
                short = conv(src, cout, k=1) if project else src
                m = add(Eltwise())
                out = "t%d" % self.n
                self.plan.append(("call", m, [y, short], out))
                ch[out], hw[out] = cout, hw[y]
                return relu(out)
            if kind == "bneck2":
                # ResNet's shape: two or three bottlenecks of 4 C channels in a row (the integer model fuses conv3 + NewAdd + ReLU + the
                # next conv1 into one kernel there; the float forward conv3 + Eltwise + ReLU)
                C = rng.choice([64, 64, 128])
                if c != 4 * C:
                    cur = relu(conv(cur, 4 * C, k=1))
                for _ in range(rng.randint(2, 3)):
                    cur = bottleneck(cur, C, 4 * C, False)
                continue
            if kind == "projhead":
                # a stage's first block: 64 -> 256 with a projection shortcut, then an identity block
                if c != 64:
                    cur = relu(conv(cur, 64, k=1))
                cur = bottleneck(cur, 64, 256, True)
                cur = bottleneck(cur, 64, 256, False)
                continue
            if kind == "plain":
                cur = relu(conv(cur, rng.choice(widths), s=rng.choice([1, 1, 2])))
            elif kind in ("res", "resproj"):
                mid = rng.choice([8, 16, 16, 64] if not odd else [8, 12, 20, 64])
                cout = c if kind == "res" else rng.choice(widths)
                s = rng.choice([1, 2]) if kind == "resproj" and hw[cur] >= 8 else 1
                y = relu(conv(cur, mid, k=1))
                y = relu(conv(y, mid, k=3, s=s))
                y = conv(y, cout, k=1)
                short = cur if kind == "res" else conv(cur, cout, k=1, s=s)
                m = add(Eltwise())
                out = "t%d" % self.n
                self.plan.append(("call", m, [y, short] if rng.random() < 0.7 else [short, y], out))
                ch[out], hw[out] = cout, hw[y]
                cur = relu(out) if rng.random() < 0.85 else out
            elif kind == "concat":
                a, b = conv(cur, rng.choice([8, 16] if not odd else [8, 12, 20]), k=3), conv(cur, rng.choice([8, 16] if not odd else [4, 12, 16]), k=1)
                m = add(Concat())
                out = "t%d" % self.n
                self.plan.append(("call", m, [a, b], out))
                ch[out], hw[out] = ch[a] + ch[b], hw[a]
                cur = relu(out)
            elif kind == "pool" and odd and hw[cur] <= 8 and rng.random() < 0.5:
                m = add(nn.UpsamplingNearest2d(scale_factor=2))
                out = "t%d" % self.n
                self.plan.append(("call", m, [cur], out))
                ch[out], hw[out] = c, 2 * hw[cur]
                cur = relu(conv(out, c, k=3))
            elif kind == "pool" and hw[cur] >= 6:
                k, s, p = rng.choice([(3, 2, 1), (2, 2, 0), (3, 1, 1)])
                m = add(nn.MaxPool2d(k, s, p))
                out = "t%d" % self.n
                self.plan.append(("call", m, [cur], out))
                ch[out], hw[out] = c, (hw[cur] + 2 * p - k) // s + 1
                cur = out
            elif kind == "twice":
                # one convolution output read by TWO consumers: its ReLU and, raw, an Eltwise further down
                y = conv(cur, c, k=1)
                r = relu(y)
                z = conv(r, c, k=3)
                m = add(Eltwise())
                out = "t%d" % self.n
                self.plan.append(("call", m, [z, y], out))
                ch[out], hw[out] = c, hw[z]
                cur = relu(out)
            elif kind == "shared":
                # the sum of a residual block read by two branches
                y = conv(cur, c, k=1)
                m = add(Eltwise())
                s_ = "t%d" % self.n
                self.plan.append(("call", m, [y, cur], s_))
                ch[s_], hw[s_] = c, hw[y]
                a, b = conv(s_, 8, k=1), relu(s_)
                b = conv(b, 8, k=3)
                m = add(Concat())
                out = "t%d" % self.n
                self.plan.append(("call", m, [a, b], out))
                ch[out], hw[out] = 16, hw[a]
                cur = relu(out)
        if rng.random() < 0.5 and hw[cur] > 1:
            m = add(nn.AvgPool2d(hw[cur]))
            out = "t%d" % self.n
            self.plan.append(("call", m, [cur], out))
            ch[out], hw[out] = ch[cur], 1
            cur = out
        m = add(View())
        out = "t%d" % self.n
        self.plan.append(("call", m, [cur], out))
        feat = ch[cur] * hw[cur] * hw[cur]
        m2 = add(nn.Linear(feat, 10))
        self.plan.append(("call", m2, [out], "y"))

    def forward(self, x):
        t = {"x": x}
        for _op, m, ins, out in self.plan:
            t[out] = getattr(self, m)(*[t[i] for i in ins])
        return t["y"]


def random_net(index, seed, odd=False, device="cpu", share=False, bn=False):
    """Model `index` of the seeded family: (model in eval mode, image size, batch size, the generator's rng after the draw)."""
    import random
    import torch
    rng = random.Random(seed * 100003 + index)
    size = rng.choice([16, 24, 32])
    torch.manual_seed(seed * 7919 + index)
    model = RandomNet(rng, size, odd, share, bn).eval()
    with torch.no_grad():
        for p in model.parameters():
            p.mul_(1.5)
        for m in model.modules():                             # (bn: statistics and affine terms that are not the identity)
            if isinstance(m, nn.BatchNorm2d):
                m.running_mean.normal_(0.0, 0.2)
                m.running_var.uniform_(0.5, 1.5)
                m.weight.uniform_(0.6, 1.4)
                m.bias.normal_(0.0, 0.2)
    bs = rng.choice([4, 8])
    return model.to(device), size, bs, rng


# --------------------------------------------------------------------------------------------
# G14: INTERVAL_NUM other than 2048 (configs.yml:24; distribution_collector.py:9-14 and quantizer.py:98-167 are generic in the
# histogram's length).  Tensors for the collector and histograms of `bins` bins for the KL sweep.
# --------------------------------------------------------------------------------------------
G14_BINS = (512, 1024, 4096)


def g14_tensor_cases():
    c = g1_cases()
    return {k: c[k] for k in ("normal_100352", "normal_3batch", "relu_sparse", "all_zero", "single_outlier", "tiny_values",
                              "ragged_4099", "pass2_exceeds", "uniform")}


def g14_hists(bins):
    j = np.arange(bins, dtype=np.float64)
    s = bins / 2048.0

    def poisson(lam, seed):
        return _rng(seed).poisson(lam).astype(np.int32)

    h = {}
    h["gauss"] = poisson(2e4 * np.exp(-0.5 * (j / (300.0 * s)) ** 2), 1401)
    h["laplace"] = poisson(1e5 * np.exp(-j / (60.0 * s)), 1402)
    h["heavy_tail"] = poisson(1e5 / (1.0 + (j / (20.0 * s)) ** 2), 1403)
    h["relu_like"] = poisson(8e4 * np.exp(-j / (35.0 * s)) + 30.0 * np.exp(-0.5 * ((j - 900 * s) / (200.0 * s)) ** 2), 1404)
    sp = np.zeros(bins, dtype=np.int32)
    idx = _rng(1405).choice(bins, 90, replace=False)
    sp[idx] = _rng(1406).integers(1, 50, 90)
    h["sparse_random"] = sp
    h["empty"] = np.zeros(bins, dtype=np.int32)
    h["merged_f64"] = h["gauss"].astype(np.float64) + h["laplace"].astype(np.float64)
    if bins > 2048:                      # the reference's interpreted sweep costs ~10 s per 4096-bin histogram
        h = {k: h[k] for k in ("gauss", "relu_like", "sparse_random", "merged_f64")}
    return h
