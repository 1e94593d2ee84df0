"""tools.Quantity.reserve_pool (EXTENSION; no reference counterpart -- the reference is a one-shot CPU script): the service-mode
switch behind bench.py's `value`.  A fresh process never grows its allocator pool for the activation cache (value_cold /
one_shot); a plain script that calls reserve_pool() once gets pass 1's activations kept for pass 2 from its next calibration on --
same feat.table, a cheaper pass 2.  Run in a child process so that the pool is really empty at the start.   pytest -m gpu"""
import json
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = textwrap.dedent('''
    import os, sys, json, time
    sys.path[:0] = [r"{root}", r"{root}/pytorch-quantity_amd/quantity", r"{root}/tests", r"{root}/tests/golden"]
    import torch
    import cases
    from workdir_util import product_workdir
    from common.quantity import merge_bn
    from model.resnet.ResNet_fabu import ResNet50
    from tools import Quantity
    out = {{}}
    model = merge_bn(cases.seed_model(ResNet50(), gamma_scale=0.5).eval()).cuda()
    data = [(torch.randn(32, 3, 224, 224, generator=torch.Generator(device="cuda").manual_seed(7 + i), device="cuda"), None)
            for i in range(4)]
    def run(tag):
        with product_workdir(input_shape="1,3,224,224", device="gpu", max_cali_img_num=3) as tmp:
            q = Quantity(model)
            q.profile_phases = True                    # .timings are device times (a synchronisation at the phase boundaries)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            q.activation_quantize(data)
            torch.cuda.synchronize()
            out[tag] = {{"seconds": time.perf_counter() - t0, "cache_bytes": q.timings["cache_bytes"], "pass2_s": q.timings["pass2_s"],
                        "plan": q.timings["cache_plan"], "table": open(os.path.join(tmp, "test", "workdir", "feat.table")).read()}}
    run("fresh")                       # the process's first calibration: no pool, no cache
    run("fresh_again")                 # code warm, pool still small: still no cache
    out["reserved_before"] = torch.cuda.memory_reserved()
    out["reserved_after"] = Quantity.reserve_pool(0.5)
    out["total"] = torch.cuda.mem_get_info()[1]
    run("pooled")
    run("pooled_again")
    json.dump(out, open(r"{out}", "w"))
''')


def test_reserve_pool_without_a_gpu_is_a_noop():
    import torch
    if torch.cuda.is_available():
        pytest.skip("this host has a GPU")
    sys.path.insert(0, os.path.join(ROOT, "pytorch-quantity_amd", "quantity"))
    from tools import Quantity
    assert Quantity.reserve_pool(0.8) == 0


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_a_plain_script_gets_the_activation_cache_after_reserve_pool(tmp_path):
    script, out = str(tmp_path / "svc.py"), str(tmp_path / "svc.json")
    with open(script, "w") as fh:
        fh.write(SCRIPT.format(root=ROOT, out=out))
    env = {k: v for k, v in os.environ.items() if k not in ("FQ_ACT_CACHE_GB", "FQ_CACHE_PLAN")}
    r = subprocess.run([sys.executable, script], env=env, capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.load(open(out))
    # a fresh process: nothing kept, every image through the network twice
    assert d["fresh"]["cache_bytes"] == 0 and d["fresh_again"]["cache_bytes"] == 0
    # reserve_pool(0.5): the pool now holds half the device ...
    assert d["reserved_after"] >= 0.45 * d["total"] > d["reserved_before"]
    # ... and the very next calibration keeps pass 1's activations for pass 2 (4 batches x 2.1 GB fit whole), at no cost to the table
    for tag in ("pooled", "pooled_again"):
        assert d[tag]["cache_bytes"] > 0 and d[tag]["plan"] is not None
        assert d[tag]["table"] == d["fresh"]["table"]
    # (pass 2 of the cached run histograms kept tensors and re-runs at most a prefix; the fresh one runs every image through the
    #  whole network again -- a factor, not a margin: the comparison does not depend on the box's mood)
    assert d["pooled_again"]["pass2_s"] < d["fresh_again"]["pass2_s"]
