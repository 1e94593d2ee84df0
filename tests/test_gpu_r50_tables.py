"""north_star target: feat.table / weight.table / quantised-weight JSON of the fabu ResNet-50 produced
by the HIP path are byte-identical to the CPU path (the oracle-backed engine) on the same activations.

The model forward runs once, on the GPU; a recording collector hands every batch's activations to the
HIP engine and keeps host copies; a second orchestrator run replays those host copies through the
oracle engine.  Both runs therefore see the very same tensors (MIOpen convolutions are not bitwise
reproducible from call to call, so re-running the forward would not do).   pytest -m gpu"""
import hashlib
import os

import numpy as np
import pytest

import cases
from workdir_util import product_workdir

pytestmark = pytest.mark.gpu


def _state(wd):
    out = {"feat": open(os.path.join(wd, "feat.table")).read(), "weight": open(os.path.join(wd, "weight.table")).read()}
    for d in ("weight", "bias", "new_weight", "new_bias"):
        out[d] = {f: hashlib.sha256(open(os.path.join(wd, d, f), "rb").read()).hexdigest()
                  for f in sorted(os.listdir(os.path.join(wd, d)))}
    return out


@pytest.mark.parametrize("arch,rows,batch", [("r50", 71, 4), ("r101", 139, 2)])
def test_bottleneck_tables_hip_equals_cpu_oracle(oracle, arch, rows, batch):
    from common.quantity import DistributionCollector, merge_bn
    from engine_doubles import OracleCollector, OracleQuantizer
    from model.resnet.ResNet_fabu import ResNet50, ResNet101
    from tools import Quantity

    tape = {"max": [], "hist": []}

    class RecordingCollector(DistributionCollector):
        """Records what the engine was fed.  The calibration loop hands the tensors of one forward over in several
        partial dicts (from inside the hooks); the tape holds one merged dict per forward."""

        def _record(self, kind, tensors):
            cur = self.__dict__.setdefault("_open_" + kind, {})
            assert not set(cur) & set(tensors), "a tensor was handed over twice in one forward"
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

    class ReplayCollector(OracleCollector):
        def refresh_max_val(self, tensors):
            super().refresh_max_val(tape["max"].pop(0) if self._replay else tensors)

        def add_to_distributions(self, tensors):
            super().add_to_distributions(tape["hist"].pop(0) if self._replay else tensors)

        _replay = True

    class HipQuantity(Quantity):
        collector_cls = RecordingCollector
        fuse_bias_absmax = False          # the tape wants to see every tensor pass through refresh_max_val

    class CpuQuantity(Quantity):
        collector_cls = ReplayCollector
        quantizer_cls = OracleQuantizer

    batches = cases.calib_batches(2, (batch, 3, 224, 224), seed=77)
    ctor = ResNet50 if arch == "r50" else ResNet101         # r101: 139 rows = two chunked kernel launches
    model = merge_bn(cases.seed_model(ctor(), gamma_scale=0.7 if arch == "r50" else 0.5).eval()).cuda()

    with product_workdir(input_shape="1,3,224,224", device="gpu", max_cali_img_num=1) as tmp:
        q = HipQuantity(model)
        q.activation_quantize(batches)
        hip_hist = q._collector.hist_device.cpu().numpy()
        hip_max = q._collector.max_device.cpu().numpy()
        q.collector_cls = DistributionCollector                  # weights: plain HIP collector
        q.weight_quantize()
        hip = _state(os.path.join(tmp, "test", "workdir"))
    assert len(tape["max"]) == 2 and len(tape["hist"]) == 2

    with product_workdir(input_shape="1,3,224,224", device="gpu", max_cali_img_num=1) as tmp:
        q2 = CpuQuantity(model)
        q2.overlap_streams = False
        q2.activation_quantize(batches)                          # forwards run, their outputs are ignored
        cpu_hist = q2._collector._hist
        cpu_max = q2._collector._max
        ReplayCollector._replay = False                          # weights come straight from the parameters
        q2.weight_quantize()
        ReplayCollector._replay = True
        cpu = _state(os.path.join(tmp, "test", "workdir"))

    np.testing.assert_array_equal(hip_max, cpu_max)
    np.testing.assert_array_equal(hip_hist, cpu_hist)            # 71 rows x 2048 bins, exact
    assert hip["feat"] == cpu["feat"]
    assert hip["weight"] == cpu["weight"]
    for d in ("weight", "bias", "new_weight", "new_bias"):
        assert hip[d] == cpu[d], d
    assert len(hip["feat"].strip().split("\n")) == rows
