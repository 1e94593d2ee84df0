"""bench.py's own synchronisation helpers at world size 2 (gloo, CPU): the int8-sim section of the multi-GPU bench
must not hang when only rank 0 holds the tables (Quantity writes files on rank 0 only) or when one rank fails locally.
share_tables() broadcasts the two tables into every rank's scratch tree; run_section() wraps rank-local work and ends in
an agreement (one MIN all-reduce) that every rank takes part in -- no collective is ever skipped by an exception."""
import json
import os
import subprocess
import sys
import tempfile
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import os, sys, json, tempfile
    sys.path[:0] = [r"{root}", r"{root}/pytorch-quantity_amd/quantity", r"{root}/tests", r"{root}/tests/golden"]
    import torch, torch.distributed as dist
    import bench
    from common.quantity import BitReader
    dist.init_process_group("gloo")
    rank = dist.get_rank()
    g3 = json.load(open(r"{root}/tests/golden/g3_r18_e2e.json"))
    wd = os.path.join(tempfile.mkdtemp(prefix="fq_sync_%d_" % rank), "workdir")
    if rank == 0:                                   # what Quantity leaves behind: files on rank 0 only
        os.makedirs(wd)
        open(os.path.join(wd, "feat.table"), "w").write(g3["feat_table"])
        open(os.path.join(wd, "weight.table"), "w").write(g3["weight_table_after_quantize"])
    assert os.path.isfile(os.path.join(wd, "feat.table")) == (rank == 0)
    bench.share_tables(wd)
    reader = BitReader(os.path.join(wd, "feat.table"), os.path.join(wd, "weight.table"))
    feat = reader.get_feat_info()[0]                 # every rank can rebuild the model now

    def local(fail):
        if fail:
            raise RuntimeError("boom on rank %d" % rank)
        return rank + 10

    out = {{"rank": rank, "n_feat": len(feat)}}
    ok, res, err = bench.run_section(lambda: local(rank == 1))          # one rank fails: everybody learns it
    out["one_fails"] = [ok, res, err]
    ok, res, err = bench.run_section(lambda: local(False))              # and the next collective still lines up
    out["none_fails"] = [ok, res, err]
    out["all_ok"] = [bench.all_ok(True), bench.all_ok(rank == 0)]
    dist.barrier()
    json.dump(out, open(r"{out}" + ".rank%d" % rank, "w"))
    dist.destroy_process_group()
''')


@pytest.mark.timeout(600)
def test_bench_recon_section_helpers_keep_two_ranks_in_step(tmp_path):
    script = os.path.join(tempfile.mkdtemp(prefix="fq_sync_"), "worker.py")
    out = str(tmp_path / "sync.json")
    with open(script, "w") as fh:
        fh.write(WORKER.format(root=ROOT, out=out))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29631", script]
    r = subprocess.run(cmd, env=dict(os.environ, OMP_NUM_THREADS="2"), capture_output=True, text=True, timeout=500)
    assert r.returncode == 0, r.stderr[-3000:]
    r0, r1 = (json.load(open(out + ".rank%d" % k)) for k in (0, 1))
    assert r0["n_feat"] == r1["n_feat"] > 0
    assert r0["one_fails"][0] is False and r1["one_fails"][0] is False
    assert r0["one_fails"][1] == 10 and r0["one_fails"][2] is None            # rank 0's own work succeeded ...
    assert r1["one_fails"][1] is None and "boom on rank 1" in r1["one_fails"][2]
    assert r0["none_fails"] == [True, 10, None] and r1["none_fails"] == [True, 11, None]
    assert r0["all_ok"] == [True, False] and r1["all_ok"] == [True, False]


@pytest.mark.parametrize("argv,per_gpu,batch,scaling", [
    ([], 5120, 256, "weak"),                                        # the default: BASELINE configs[1], 20 steps
    (["--steps", "20", "--warmup", "5"], 5120, 256, "weak"),        # the driver's command: same 5 120 images
    (["--steps", "40"], 5120, 128, "weak"),                         # --steps only cuts the same images differently
    (["--steps", "10", "--images", "1000"], 1000, 100, "weak"),
    # config 4 (strong scaling): the batch stays 256 at every N; ceil(50 000 / 256) = 196 batches, the busiest of 8 ranks owns 25
    (["--gpus", "8", "--total-images", "50000"], 6400, 256, "strong"),
    (["--gpus", "1", "--total-images", "50000", "--steps", "20"], 50176, 256, "strong"),      # the N = 1 leg: --steps is derived
    (["--gpus", "2", "--total-images", "1024", "--steps", "2"], 512, 256, "strong"),
    (["--gpus", "8", "--total-images", "1024", "--batch", "64"], 128, 64, "strong"),
    (["--batch", "64", "--steps", "4"], 256, 64, "weak"),
])
def test_bench_workload_arithmetic(monkeypatch, argv, per_gpu, batch, scaling):
    """`--steps` must not change the workload (round-1 review): images per GPU come from --images / --total-images, the
    batch follows from them."""
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.setattr(sys, "argv", ["bench.py"] + argv)
    args = bench.parse_args()
    assert (args.images_per_gpu, args.batch, args.scaling) == (per_gpu, batch, scaling)
    assert args.images_per_gpu == args.batch * args.steps
