"""Scratch tree laid out the way the drop-in's cwd-relative config paths expect:
<tmp>/tools/configs.yml, <tmp>/test/user_configs.yml, cwd = <tmp>/test."""
import contextlib
import os
import shutil
import tempfile

import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
PKG_QUANTITY = os.path.join(os.path.dirname(HERE), "pytorch-quantity_amd", "quantity")


@contextlib.contextmanager
def product_workdir(input_shape="1,3,32,32", device="cpu", max_cali_img_num=1, gpu=0, keep=None, interval_num=None):
    tmp = tempfile.mkdtemp(prefix="fq_wd_")
    os.makedirs(os.path.join(tmp, "tools"))
    os.makedirs(os.path.join(tmp, "test"))
    with open(os.path.join(PKG_QUANTITY, "tools", "configs.yml")) as fh:
        cfg = yaml.safe_load(fh)
    cfg["SETTINGS"]["MAX_CALI_IMG_NUM"] = max_cali_img_num
    if interval_num is not None:
        cfg["SETTINGS"]["INTERVAL_NUM"] = int(interval_num)
    with open(os.path.join(tmp, "tools", "configs.yml"), "w") as fh:
        yaml.safe_dump(cfg, fh)
    with open(os.path.join(PKG_QUANTITY, "test", "user_configs.yml")) as fh:
        ucfg = yaml.safe_load(fh)
    ucfg["MODEL"]["INPUT_SHAPE"] = input_shape
    ucfg["SETTINGS"]["DEVICE"] = device
    ucfg["SETTINGS"]["GPU"] = gpu
    ucfg["PATH"]["QUANTITY_MODEL_PATH"] = os.path.join(tmp, "test", "workdir", "quantity_model.pth")
    with open(os.path.join(tmp, "test", "user_configs.yml"), "w") as fh:
        yaml.safe_dump(ucfg, fh)
    old = os.getcwd()
    os.chdir(os.path.join(tmp, "test"))
    try:
        yield tmp
    finally:
        os.chdir(old)
        if keep is None:
            shutil.rmtree(tmp, ignore_errors=True)
