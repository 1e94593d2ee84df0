"""Import shim for running the *reference* (lswzjuer/pytorch-quantity, mounted read-only at
/root/reference) inside the build container, so that golden vectors can be captured from it.

TEST INFRASTRUCTURE ONLY.  Nothing here is imported by the product package and nothing here can run
on the GPU box (/root/reference does not exist there).  No reference source is copied: the reference
is imported from where it lies.

What the reference needs on a current stack:
  * stub modules for termcolor / cv2 / torchvision (imported, unused on the live paths)
  * yaml.load without an explicit Loader (PyYAML 6 requires one)
  * time.clock (removed in Python 3.8)
  * cwd == <scratch>/test with <scratch>/tools/configs.yml and <scratch>/test/user_configs.yml
"""
import contextlib
import os
import shutil
import sys
import tempfile
import time
import types

REFERENCE_ROOT = "/root/reference/quantity"


def reference_available():
    return os.path.isdir(REFERENCE_ROOT)


def _install_stubs():
    if "termcolor" not in sys.modules:
        m = types.ModuleType("termcolor")
        m.colored = lambda s, *a, **k: s
        sys.modules["termcolor"] = m
    if "cv2" not in sys.modules:
        sys.modules["cv2"] = types.ModuleType("cv2")
    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        for sub in ("models", "transforms", "datasets"):
            sm = types.ModuleType("torchvision." + sub)
            setattr(tv, sub, sm)
            sys.modules["torchvision." + sub] = sm
        sys.modules["torchvision"] = tv
    import yaml
    if not getattr(yaml.load, "_fq_patched", False):
        _orig = yaml.load

        def _load(stream, Loader=None):
            return _orig(stream, Loader=Loader or yaml.SafeLoader)

        _load._fq_patched = True
        yaml.load = _load
    if not hasattr(time, "clock"):
        time.clock = time.perf_counter
    import matplotlib
    matplotlib.use("Agg")


def import_reference():
    """Return (common.quantity, tools) modules of the reference."""
    assert reference_available(), "reference tree is only present in the build container"
    _install_stubs()
    # The product package uses the same top-level names (common, tools, model); make sure we get
    # the reference's.
    for k in list(sys.modules):
        if k == "common" or k.startswith("common.") or k == "tools" or k.startswith("tools.") \
                or k == "model" or k.startswith("model."):
            del sys.modules[k]
    sys.path.insert(0, REFERENCE_ROOT)
    import common.quantity as cq  # noqa
    import tools as tl  # noqa
    return cq, tl


@contextlib.contextmanager
def reference_workdir(input_shape="1,3,32,32", device="cpu", max_cali_img_num=1, extra_tool_cfg=None):
    """Scratch tree laid out the way the reference's cwd-relative paths expect; cwd = <tmp>/test."""
    import yaml
    tmp = tempfile.mkdtemp(prefix="fq_ref_")
    os.makedirs(os.path.join(tmp, "tools"))
    os.makedirs(os.path.join(tmp, "test"))
    with open(os.path.join(REFERENCE_ROOT, "tools", "configs.yml")) as f:
        cfg = yaml.safe_load(f)
    cfg["SETTINGS"]["MAX_CALI_IMG_NUM"] = max_cali_img_num
    if extra_tool_cfg:
        cfg["SETTINGS"].update(extra_tool_cfg)
    with open(os.path.join(tmp, "tools", "configs.yml"), "w") as f:
        yaml.safe_dump(cfg, f)
    with open(os.path.join(REFERENCE_ROOT, "test", "user_configs.yml")) as f:
        ucfg = yaml.safe_load(f)
    ucfg["MODEL"]["INPUT_SHAPE"] = input_shape
    ucfg["SETTINGS"]["DEVICE"] = device
    ucfg["PATH"]["QUANTITY_MODEL_PATH"] = os.path.join(tmp, "test", "workdir", "quantity_model.pth")
    with open(os.path.join(tmp, "test", "user_configs.yml"), "w") as f:
        yaml.safe_dump(ucfg, f)
    old = os.getcwd()
    os.chdir(os.path.join(tmp, "test"))
    try:
        yield tmp
    finally:
        os.chdir(old)
        shutil.rmtree(tmp, ignore_errors=True)
