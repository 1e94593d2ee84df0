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
