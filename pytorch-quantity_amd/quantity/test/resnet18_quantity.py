"""Calibrate the CIFAR ResNet-18 fixture: activation tables, weight tables, quantised-parameter JSON.

Counterpart of the reference's quantity/test/resnet18_quantity.py (same API calls, :50-53): run from
this directory (cwd-relative configs).  Data is synthetic unless torchvision + CIFAR-10 are available;
weights are loaded from user_configs.yml PATH.MODEL_PATH when that file exists.
"""
import os
import sys

import torch
import yaml

sys.path.insert(0, '../')
from tools import Quantity  # noqa: E402
from common.quantity import merge_bn  # noqa: E402
from model.resnet.ResNet_18_fabu import ResNet18  # noqa: E402
import _synthetic  # noqa: E402


def main():
    with open("./user_configs.yml") as fh:
        user_config = yaml.safe_load(fh)
    model = ResNet18()
    weights = user_config["PATH"]["MODEL_PATH"]
    if os.path.isfile(weights):
        model.load_state_dict(torch.load(weights, map_location="cpu"))
    else:
        _synthetic.randomize_bn(model)      # no checkpoint: give BatchNorm non-trivial statistics
    model.eval()
    model = merge_bn(model)
    if user_config["SETTINGS"]["DEVICE"] == "gpu":
        model.cuda()
    loader = _synthetic.batches(2, 100, (3, 32, 32))
    calibrator = Quantity(model)
    calibrator.activation_quantize(loader)
    calibrator.weight_quantize()
    calibrator.rewrite_weight()


if __name__ == "__main__":
    main()
