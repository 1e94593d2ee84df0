"""Data-parallel calibration plumbing on CPU: two gloo ranks (one process each) run the drop-in
orchestrator with the oracle-backed engine; batches are dealt round-robin, maxima are MAX-reduced
and histograms SUM-reduced, and the resulting feat.table must be byte-identical to a single-process
run over the same batches (integer sums and maxima are order independent) -- and to the reference's
golden table."""
import json
import os
import subprocess
import sys
import tempfile
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import os, sys, json
    sys.path[:0] = [r"{root}", r"{root}/pytorch-quantity_amd/quantity", r"{root}/tests", r"{root}/tests/golden"]
    import torch, torch.distributed as dist
    import cases
    from engine_doubles import OracleCollector, OracleQuantizer
    from workdir_util import product_workdir
    from common.quantity import merge_bn
    from model.resnet.ResNet_18_fabu import ResNet18
    from tools import Quantity

    class CpuQuantity(Quantity):
        collector_cls = OracleCollector
        quantizer_cls = OracleQuantizer

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        dist.init_process_group("gloo")
    rank = dist.get_rank() if world > 1 else 0
    torch.set_num_threads(2)
    with product_workdir(device="cpu", max_cali_img_num=int(os.environ.get("FQ_TEST_MAX_CALI", "3"))) as tmp:
        model = merge_bn(cases.seed_model(ResNet18()).eval())
        q = CpuQuantity(model)
        bits = q.activation_quantize(cases.calib_batches(int(os.environ.get("FQ_TEST_BATCHES", "5")), (2, 3, 32, 32)))
        if os.environ.get("FQ_TEST_WEIGHTS") == "1":
            q.weight_quantize()
        wd = os.path.join(tmp, "test", "workdir")
        listing = sorted(os.path.relpath(os.path.join(d, f), wd) for d, _s, fs in os.walk(wd) for f in fs)
        json.dump(listing, open(r"{out}" + ".files.rank%d" % rank, "w"))
        if rank == 0:
            table = open(os.path.join(tmp, "test", "workdir", "feat.table")).read()
            hs = {{k: int(v.sum()) for k, v in q._collector.distributions.items()}}
            json.dump({{"table": table, "hist_sums": hs, "max": {{k: float(v) for k, v in q._collector.max_vals.items()}}}},
                      open(r"{out}", "w"))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
''')


def _run(world, out, max_cali=3, weights=False, batches=5, port=29617, threads=2):
    script = os.path.join(tempfile.mkdtemp(prefix="fq_dist_"), "worker.py")
    with open(script, "w") as fh:
        fh.write(WORKER.format(root=ROOT, out=out))
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), FQ_TEST_MAX_CALI=str(max_cali), FQ_TEST_WEIGHTS="1" if weights else "0",
               FQ_TEST_BATCHES=str(batches))
    if world == 1:
        cmd = [sys.executable, script]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
               "--master-addr", "127.0.0.1", "--master-port", str(port), script]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.load(open(out))


@pytest.mark.timeout(1800)
def test_two_rank_gloo_calibration_is_shard_count_invariant(tmp_path):
    one = _run(1, str(tmp_path / "w1.json"), weights=True)
    two = _run(2, str(tmp_path / "w2.json"), weights=True)
    # only rank 0 writes files: rank 1's scratch tree holds no table and no JSON, rank 0's holds what a single process writes
    files0 = json.load(open(str(tmp_path / "w2.json") + ".files.rank0"))
    files1 = json.load(open(str(tmp_path / "w2.json") + ".files.rank1"))
    assert files1 == [], files1
    assert files0 == json.load(open(str(tmp_path / "w1.json") + ".files.rank0"))
    assert "feat.table" in files0 and "weight.table" in files0 and any(f.startswith("new_bias/") for f in files0)
    assert one["table"] == two["table"]
    assert one["hist_sums"] == two["hist_sums"]
    assert one["max"] == two["max"]
    assert one["table"].startswith("image ")
    # 4 batches of 2 images were used (MAX_CALI_IMG_NUM = 3 -> batches 0..3): every histogram saw them all
    assert one["hist_sums"]["image"] == 4 * 2 * 3 * 32 * 32


@pytest.mark.timeout(1800)
def test_rank_without_batches_still_gets_global_statistics(tmp_path):
    """One calibration batch, two ranks: rank 1 owns nothing, yet after the all-reduces every rank holds
    the global maxima / histograms and the table equals the single-process one."""
    one = _run(1, str(tmp_path / "w1.json"), max_cali=0)
    two = _run(2, str(tmp_path / "w2.json"), max_cali=0)
    assert one["table"] == two["table"]
    assert one["hist_sums"] == two["hist_sums"] and one["hist_sums"]["image"] == 2 * 3 * 32 * 32


@pytest.mark.timeout(1800)
@pytest.mark.parametrize("world", [4, 8])
def test_four_and_eight_rank_gloo_calibration_equals_the_single_process_table(tmp_path, world):
    """BASELINE config 4's rank count (and half of it) on CPU: 10 calibration batches dealt round robin over 4 / 8 gloo ranks
    (uneven shares: ranks 0 and 1 own one batch more than the others), one MAX and one SUM all-reduce -- the product's own
    lines (common/quantity/_collectives.py) -- and rank 0's feat.table equals the single-process one byte for byte; no
    other rank writes a file."""
    one = _run(1, str(tmp_path / "w1.json"), max_cali=9, batches=10)
    many = _run(world, str(tmp_path / "wn.json"), max_cali=9, batches=10, port=29640 + world, threads=1)
    assert one["table"] == many["table"]
    assert one["hist_sums"] == many["hist_sums"] and one["max"] == many["max"]
    assert one["hist_sums"]["image"] == 10 * 2 * 3 * 32 * 32
    for r in range(1, world):
        assert json.load(open(str(tmp_path / "wn.json") + ".files.rank%d" % r)) == []


# ---------------------------------------------------------------- the per-channel rows (BASELINE config 4's exchange)
CHANNEL_WORKER = textwrap.dedent('''
    import os, sys, json
    sys.path[:0] = [r"{root}", r"{root}/pytorch-quantity_amd/quantity", r"{root}/tests", r"{root}/tests/golden"]
    import numpy as np, torch, torch.distributed as dist
    import cases
    from engine_doubles import OracleChannelCollector
    from workdir_util import product_workdir
    from common.quantity import merge_bn
    from tools import Quantity

    class CpuQuantity(Quantity):
        channel_collector_cls = OracleChannelCollector

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        dist.init_process_group("gloo")
    rank = dist.get_rank() if world > 1 else 0
    torch.set_num_threads(2)
    # every collective this process issues: (name, dtype, elements)
    calls = []
    for name in ("all_reduce", "reduce_scatter_tensor", "all_gather_into_tensor"):
        def spy(*a, _real=getattr(dist, name), _name=name, **k):
            t = a[1] if _name == "reduce_scatter_tensor" else a[0]
            calls.append([_name, str(t.dtype), int(t.numel())])
            return _real(*a, **k)
        setattr(dist, name, spy)
    with product_workdir(input_shape="1,3,16,16", device="cpu", max_cali_img_num=int(os.environ.get("FQ_TEST_MAX_CALI", "4"))) as tmp:
        model = merge_bn(cases.seed_model(cases.tiny_vgg_net()).eval())
        q = CpuQuantity(model)
        bits = q.activation_quantize_per_channel(cases.calib_batches(int(os.environ.get("FQ_TEST_BATCHES", "5")), (2, 3, 16, 16)))
        wd = os.path.join(tmp, "test", "workdir")
        listing = sorted(os.path.relpath(os.path.join(d, f), wd) for d, _s, fs in os.walk(wd) for f in fs)
        json.dump(listing, open(r"{out}" + ".files.rank%d" % rank, "w"))
        c = q._channel_collector
        mx, hist = c._stat_tensors()
        weights = 1 + torch.arange(hist.shape[1])
        if c._own_block is None:                      # one process: every row is this rank's
            lo, block = 0, hist[:c.rows]
        else:
            lo, block = c._own_block
        json.dump({{"bits": bits, "rows": c.rows, "max": mx.tolist(), "calls": calls,
                   "own_lo": int(lo), "own_row_sums": block.sum(1).tolist(), "own_row_digests": (block * weights).sum(1).tolist(),
                   "thr": [int(t) for t in c.threshold_bins],
                   "table": open(os.path.join(wd, "feat_channel.table")).read() if rank == 0 else None}},
                  open(r"{out}" + ".rank%d" % rank, "w"))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
''')


def _run_channels(world, out, max_cali=4, batches=5, port=29671):
    script = os.path.join(tempfile.mkdtemp(prefix="fq_dist_ch_"), "worker.py")
    with open(script, "w") as fh:
        fh.write(CHANNEL_WORKER.format(root=ROOT, out=out))
    env = dict(os.environ, OMP_NUM_THREADS="2", FQ_TEST_MAX_CALI=str(max_cali), FQ_TEST_BATCHES=str(batches))
    if world == 1:
        cmd = [sys.executable, script]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
               "--master-addr", "127.0.0.1", "--master-port", str(port), script]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    return [json.load(open(out + ".rank%d" % k)) for k in range(world)]


@pytest.mark.timeout(1800)
@pytest.mark.parametrize("world,max_cali", [(2, 4), (3, 0), (3, 4)])
def test_per_channel_rows_are_shard_count_invariant(tmp_path, world, max_cali):
    """BASELINE config 4's exchange on gloo ranks -- the product's lines (`_collectives.py`, under
    `activation_quantize_per_channel`; the rows themselves come from the oracle): one MAX all-reduce of the per-(tensor, channel)
    maxima, ONE reduce-scatter of the per-channel histograms (rank r receives the global rows of its block [r*S, (r+1)*S) and
    nothing else), the KL sweep of that block on its owner, ONE all-gather of (threshold bin, bits).  No rank all-reduces the
    histogram buffer.  Every rank ends with the single-process maxima, thresholds and bits -- also a rank that owned no batch
    (three ranks, one batch) -- each rank's block equals the single-process histograms of those rows, and only rank 0 writes
    feat_channel.table, byte-identical for W = 1 / 2 / 3."""
    one = _run_channels(1, str(tmp_path / "c1.json"), max_cali=max_cali)[0]
    many = _run_channels(world, str(tmp_path / "cn.json"), max_cali=max_cali, port=29671 + world + 3 * max_cali)
    rows = one["rows"]
    assert rows > 3 and sum(one["own_row_sums"][:3]) == (max_cali + 1) * 2 * 3 * 16 * 16        # the image's three channels
    assert one["calls"] == []                                                                  # one process: no collective at all
    per = -(-rows // world)
    for k, r in enumerate(many):
        for key in ("bits", "rows", "max", "thr"):
            assert r[key] == one[key], key
        # this rank's block: the single-process histograms of rows [k * per, (k + 1) * per), zero rows behind the last real one
        lo = k * per
        assert r["own_lo"] == lo and len(r["own_row_sums"]) == per
        want_sums = (one["own_row_sums"][lo:lo + per] + [0] * per)[:per]
        want_dig = (one["own_row_digests"][lo:lo + per] + [0] * per)[:per]
        assert r["own_row_sums"] == want_sums and r["own_row_digests"] == want_dig
        # the exchange itself: MAX over fp32[rows]; one reduce-scatter whose INPUT is the padded buffer; one all-gather of
        # int32[2][per]; and no all-reduce of anything histogram sized
        assert ["all_reduce", "torch.float32", rows] in r["calls"]
        assert r["calls"].count(["reduce_scatter_tensor", "torch.int64", world * per * 2048]) == 1
        assert r["calls"].count(["all_gather_into_tensor", "torch.int32", world * 2 * per]) == 1
        assert not [c for c in r["calls"] if c[0] == "all_reduce" and c[2] >= 2048]
    assert many[0]["table"] == one["table"] and one["table"].startswith("image ")
    files = [json.load(open(str(tmp_path / "cn.json") + ".files.rank%d" % k)) for k in range(world)]
    assert "feat_channel.table" in files[0] and all(f == [] for f in files[1:])
