"""Calibrate the fabu ResNet-50 @224 on synthetic ImageNet-shaped batches (BASELINE config 2); under
`python -m torch.distributed.run --nproc-per-node N resnet50_quantity.py` the batches are sharded over
N GPUs and combined with one MAX and one SUM all-reduce (config 4)."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, '../')
from tools import Quantity  # noqa: E402
from common.quantity import merge_bn  # noqa: E402
from model.resnet.ResNet_fabu import ResNet50  # noqa: E402


class Batches(object):
    """Indexable calibration set: batch i is generated on the device on demand (seed 1234 + i)."""

    def __init__(self, n, batch):
        self.n, self.batch = n, batch

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        g = torch.Generator(device="cuda").manual_seed(1234 + i)
        return torch.randn(self.batch, 3, 224, 224, generator=g, device="cuda"), None


def main():
    if "RANK" in os.environ:
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group("nccl")
    model = merge_bn(ResNet50().eval()).cuda()
    q = Quantity(model)                      # user_configs.yml must say INPUT_SHAPE: 1,3,224,224
    q.activation_quantize(Batches(q._max_img_num + 1, 64))
    q.weight_quantize()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
