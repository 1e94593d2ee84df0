"""SURVEY 8f-3, file inputs at speed (reference quantity/tools/pytorch_quantizer.py:252-284: PRE_PROCESS.IMG = 2 / 0, one
file per calibration item, one forward per file).  On the GPU the drop-in groups `Quantity.file_batch` consecutive files of a
rank into one forward, decoded by a thread pool straight into a pinned staging tensor: the tables must not depend on the
grouping -- maxima and integer histograms are order independent, and the float kernels compute every output element in a
fixed summation order whatever the batch size.   pytest -m gpu"""
import os

import numpy as np
import pytest
import torch
import yaml

import cases
from workdir_util import product_workdir

pytestmark = pytest.mark.gpu


def _set_mode(tmp, mode):
    path = os.path.join(tmp, "test", "user_configs.yml")
    ucfg = yaml.safe_load(open(path))
    ucfg["PRE_PROCESS"]["IMG"] = mode
    yaml.safe_dump(ucfg, open(path, "w"))


def _r18():
    from common.quantity import merge_bn
    from model.resnet.ResNet_18_fabu import ResNet18
    return merge_bn(cases.seed_model(ResNet18()).eval()).cuda()


def _calibrate(items, mode, file_batch, max_cali):
    from tools import Quantity

    class Q(Quantity):
        pass
    Q.file_batch = file_batch
    with product_workdir(device="gpu", max_cali_img_num=max_cali) as tmp:
        _set_mode(tmp, mode)
        q = Q(_r18())
        q.activation_quantize(items)
        table = open(os.path.join(tmp, "test", "workdir", "feat.table")).read()
        return table, q._collector.hist_device.cpu().numpy(), q._collector.max_device.cpu().numpy()


def test_npy_files_grouped_into_batches_give_the_tables_of_one_forward_per_file(tmp_path):
    imgs = [cases.fixed_input((3, 32, 32), seed=900 + i) for i in range(23)]
    paths = []
    for i, im in enumerate(imgs):
        paths.append(str(tmp_path / ("img%02d.npy" % i)))
        np.save(paths[-1], im.numpy())
    loader = [(im[None], torch.zeros(1, dtype=torch.long)) for im in imgs]
    t_loader, h_loader, m_loader = _calibrate(loader, 1, 1, 20)               # items 0..20 used: 21 images
    t_one, h_one, m_one = _calibrate(paths, 2, 1, 20)                         # the reference's form: one forward per file
    t_grp, h_grp, m_grp = _calibrate(paths, 2, 8, 20)                         # groups of 8, 8, 5
    t_all, h_all, m_all = _calibrate(paths, 2, 64, 20)                        # one group
    for t, h, m in ((t_one, h_one, m_one), (t_grp, h_grp, m_grp), (t_all, h_all, m_all)):
        assert t == t_loader
        assert np.array_equal(h, h_loader) and np.array_equal(m, m_loader)
    assert int(h_loader[0].sum()) <= 21 * 3 * 32 * 32 and t_loader.startswith("image ")


def test_image_files_grouped_into_batches_give_the_tables_of_one_forward_per_file(tmp_path):
    from PIL import Image
    rng = np.random.default_rng(3)
    paths = []
    for i in range(11):
        size = (32, 32) if i % 3 else (40, 52)                               # some need the resize
        paths.append(str(tmp_path / ("im%02d.png" % i)))
        Image.fromarray(rng.integers(0, 256, size + (3,), dtype=np.uint8)).save(paths[-1])
    paths.insert(4, str(tmp_path / "missing.png"))                           # unreadable files are dropped from a group
    readable = [p for p in paths if os.path.exists(p)]
    t_one, h_one, m_one = _calibrate(readable, 0, 1, 100)
    t_grp, h_grp, m_grp = _calibrate(paths, 0, 4, 100)
    assert t_grp == t_one and np.array_equal(h_grp, h_one) and np.array_equal(m_grp, m_one)


def test_npy_reader_fills_the_pinned_batch_without_an_intermediate_array(tmp_path):
    from tools import Quantity
    a = np.arange(3 * 5 * 7, dtype=np.float32).reshape(3, 5, 7)
    np.save(str(tmp_path / "a.npy"), a)
    np.save(str(tmp_path / "f.npy"), np.asfortranarray(a))
    np.save(str(tmp_path / "d.npy"), a.astype(np.float64))
    dst = np.zeros((3, 5, 7), dtype=np.float32)
    assert Quantity._read_npy_into(str(tmp_path / "a.npy"), dst) and np.array_equal(dst, a)
    assert not Quantity._read_npy_into(str(tmp_path / "f.npy"), dst)          # Fortran order / another dtype / another shape:
    assert not Quantity._read_npy_into(str(tmp_path / "d.npy"), dst)          # the caller falls back to np.load
    assert not Quantity._read_npy_into(str(tmp_path / "a.npy"), np.zeros((3, 5, 8), dtype=np.float32))
