"""Rebuild the calibrated ResNet-18 as its integer-simulation model and compare predictions with the
float model.  Counterpart of the reference's quantity/test/resnet_reconstruction.py (:71-72,:103-104);
run resnet18_quantity.py first (it writes ./workdir/feat.table and weight.table)."""
import os
import sys

import torch
import yaml

sys.path.insert(0, '../')
from tools import reconstruction  # noqa: E402
import model.resnet.ResNet_18_fabu as resnet  # noqa: E402
import _synthetic  # noqa: E402


def main():
    assert os.path.isfile("../tools/configs.yml"), "configs.yml is necessary"
    assert os.path.isfile("user_configs.yml"), "user_configs.yml is necessary"
    with open("./user_configs.yml") as fh:
        user_config = yaml.safe_load(fh)
    model = resnet.ResNet18()
    weights = user_config["PATH"]["MODEL_PATH"]
    if os.path.isfile(weights):
        model.load_state_dict(torch.load(weights, map_location="cpu"))
    else:
        _synthetic.randomize_bn(model)      # no checkpoint: give BatchNorm non-trivial statistics
    model.eval()
    data = _synthetic.batches(4, 100, (3, 32, 32), device="cuda")

    rebuild = reconstruction.Reconstruction(model)
    float_model = rebuild.merge_bn().eval().cuda()
    with torch.no_grad():
        float_pred = [float_model(x).argmax(1) for x, _ in data]
    info = rebuild.get_quantity_information()
    int8_model = rebuild.ReconModel(info, user_config["PATH"]["QUANTITY_MODEL_PATH"]).cuda()
    agree = total = 0
    with torch.no_grad():
        for (x, _), ref in zip(data, float_pred):
            agree += int((int8_model(x).argmax(1) == ref).sum())
            total += x.shape[0]
    print("int8-simulation model agrees with the float model on %.1f %% of %d images" % (100.0 * agree / total, total))


if __name__ == "__main__":
    main()
