"""The RCCL path on hardware, as far as one GPU allows: a single-rank `nccl` process group (torch.distributed backend nccl IS
RCCL on ROCm) around the product's calibration -- init with device_id, the MAX all-reduce of the fp32 maxima and the SUM
all-reduce of the int64 histograms on the DEVICE buffers (common/quantity/_collectives.py), rank-0 file writing -- must give
the reference's ResNet-18 tables; and bench.py's own multi-rank plumbing (barriers, table broadcast, agreement all-reduces)
must run to its JSON line under the same launcher.  World sizes > 1 are covered on CPU with gloo (tests/test_distributed_cpu.py,
tests/test_bench_sync_cpu.py); no multi-GPU box was available.   pytest -m gpu"""
import json
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import os, sys, json
    sys.path[:0] = [r"{root}", r"{root}/pytorch-quantity_amd/quantity", r"{root}/tests", r"{root}/tests/golden"]
    import torch, torch.distributed as dist
    import cases
    from workdir_util import product_workdir
    from common.quantity import merge_bn
    from model.resnet.ResNet_18_fabu import ResNet18
    from tools import Quantity
    torch.cuda.set_device(0)
    with product_workdir(device="gpu", max_cali_img_num=1) as tmp:          # no process group yet: the single-process per-channel rows
        q0 = Quantity(merge_bn(cases.seed_model(ResNet18()).eval()).cuda())
        single = q0.activation_quantize_per_channel(cases.calib_batches(3, (4, 3, 32, 32)))
        single_rows = int(q0._channel_collector.rows)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    calls, real = [], dist.all_reduce
    def counted(t, op=dist.ReduceOp.SUM, **kw):
        calls.append([str(op).split(".")[-1], str(t.dtype), t.device.type, t.numel()])
        return real(t, op=op, **kw)
    dist.all_reduce = counted
    real_rs, real_ag = dist.reduce_scatter_tensor, dist.all_gather_into_tensor
    def counted_rs(out, inp, op=dist.ReduceOp.SUM, **kw):
        calls.append(["REDUCE_SCATTER", str(inp.dtype), inp.device.type, inp.numel()])
        return real_rs(out, inp, op=op, **kw)
    def counted_ag(out, inp, **kw):
        calls.append(["ALL_GATHER", str(inp.dtype), inp.device.type, inp.numel()])
        return real_ag(out, inp, **kw)
    dist.reduce_scatter_tensor, dist.all_gather_into_tensor = counted_rs, counted_ag
    with product_workdir(device="gpu", max_cali_img_num=1) as tmp:
        q = Quantity(merge_bn(cases.seed_model(ResNet18()).eval()).cuda())
        q.activation_quantize(cases.calib_batches(3, (4, 3, 32, 32)))
        out = {{"table": open(os.path.join(tmp, "test", "workdir", "feat.table")).read(),
               "calls": list(calls)}}
        by_module = q.activation_quantize_per_channel(cases.calib_batches(3, (4, 3, 32, 32)))
        out["per_channel"] = {{k: [int(b) for b in v] for k, v in by_module.items()}}
        out["per_channel_single"] = {{k: [int(b) for b in v] for k, v in single.items()}}
        out["rows"], out["calls_all"] = single_rows, list(calls)
    dist.barrier()
    json.dump(out, open(r"{out}", "w"))
    dist.destroy_process_group()
''')


def _torchrun(args, port, timeout=1100, nproc=1, extra_env=None):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(port)] + args
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **(extra_env or {}))
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)


@pytest.mark.timeout(1200)
def test_single_rank_rccl_calibration_gives_the_reference_tables(tmp_path, golden_dir):
    with open(os.path.join(golden_dir, "g3_r18_e2e.json")) as fh:
        g3 = json.load(fh)
    script, out = str(tmp_path / "worker.py"), str(tmp_path / "rccl.json")
    with open(script, "w") as fh:
        fh.write(WORKER.format(root=ROOT, out=out))
    r = _torchrun([script], 29671)
    assert r.returncode == 0, r.stderr[-3000:]
    with open(out) as fh:
        got = json.load(fh)
    assert got["table"] == g3["feat_table"]
    # the two collectives of the per-tensor calibration really ran through RCCL, on the device buffers, one call each
    assert ["MAX", "torch.float32", "cuda", 30] in got["calls"]
    assert ["SUM", "torch.int64", "cuda", 30 * 2048] in got["calls"]
    assert len(got["per_channel"]) == 30 and len(got["per_channel"]["image"]) == 3
    # the per-channel exchange through RCCL on the device buffers: MAX of fp32[rows], ONE reduce-scatter of the histogram rows, ONE
    # all-gather of int32 (threshold bin, bits) -- and the same bits as the process computed before it joined a group
    rows = got["rows"]
    assert got["per_channel"] == got["per_channel_single"]
    assert ["MAX", "torch.float32", "cuda", rows] in got["calls_all"]
    assert got["calls_all"].count(["REDUCE_SCATTER", "torch.int64", "cuda", rows * 2048]) == 1
    assert got["calls_all"].count(["ALL_GATHER", "torch.int32", "cuda", 2 * rows]) == 1
    assert not [c for c in got["calls_all"] if c[0] == "SUM" and c[3] == rows * 2048]


@pytest.mark.timeout(1200)
def test_bench_runs_under_the_distributed_launcher_with_rccl():
    """bench.py exactly as the driver launches it for N > 1, with N = 1: RCCL init, barriers, the MAX of the elapsed times,
    the table broadcast and the agreement all-reduces of the int8 section all execute; small workload."""
    r = _torchrun([os.path.join(ROOT, "bench.py"), "--gpus", "1", "--images", "256", "--steps", "2", "--warmup", "1",
                   "--int8-batch", "64", "--no-cold", "--no-cpu-baseline"], 29672)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.strip().split("\n") if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["scaling"] == "weak" and d["value"] > 0 and d["config"]["images_total"] == 256
    assert d["own_conv_launches"] == d["expected_own_conv_launches"] == 53 * 2 and "error" not in d
    assert d["ranks_seen"] == 1 and d["backend"].startswith("nccl") and d["devices"][0][:2] == [0, 0]
    assert d["int8_sim_resident"]["bit_identical_logits"] is True and d["int8_sim_images_per_s"] > 0
    assert "recon_errors" not in d and "recon_error" not in d


@pytest.mark.timeout(1500)
def test_bench_eight_rank_flow_runs_to_its_json_line_on_one_gpu():
    """The command the driver will issue on an 8-GPU node (BASELINE config 4: `python bench.py --gpus 8 --total-images ...`,
    strong scaling) in its BARE form -- no launcher around it: bench.py starts its own ranks (bench.spawn_ranks) -- as 8 ranks
    that share this box's one GPU (FQ_BENCH_BACKEND=gloo: RCCL refuses duplicate devices; every collective, barrier, table
    broadcast and agreement all-reduce of the flow still executes, over gloo) on a small image count: exactly one JSON line on
    stdout with the whole job's throughput, the per-rank sharding, what the collectives saw (ranks_seen, devices, backend)
    and no error key -- so that the first real 8-GPU run cannot die on plumbing."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", FQ_BENCH_BACKEND="gloo", FQ_BENCH_POOL_FRAC="0.05", OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--total-images", "1024", "--batch", "64",
                        "--steps", "2", "--warmup", "1", "--int8-batch", "32", "--no-cold", "--no-cpu-baseline", "--no-per-channel"],
                       env=env, capture_output=True, text=True, timeout=1400)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.strip().split("\n") if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["scaling"] == "strong" and d["value"] > 0
    assert d["ranks_seen"] == 8 and sorted(x[0] for x in d["devices"]) == list(range(8)) and d["backend"].startswith("gloo")
    assert d["config"]["images_total"] == 1024 and d["config"]["images_per_gpu"] == 128 and d["config"]["parallelism"] == "dp8"
    assert d["config"]["batch"] == 64 and d["config"]["batches_total"] == 16 and d["steps"] == 2
    # every rank's timed region stayed on the own convolution kernels (the bench would have exited 3 otherwise)
    assert d["own_conv_launches"] == d["expected_own_conv_launches"] == 53 * 2 and "error" not in d
    assert "recon_errors" not in d and "recon_error" not in d
