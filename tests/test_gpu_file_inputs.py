"""SURVEY 8f-3, file inputs at speed (reference quantity/tools/pytorch_quantizer.py:252-284: PRE_PROCESS.IMG = 2 / 0, one
file per calibration item, one forward per file).  On the GPU the drop-in groups `Quantity.file_batch` consecutive files of a
rank into one forward, decoded by a thread pool straight into a pinned staging tensor: the tables must not depend on the
grouping -- maxima and integer histograms are order independent, and the float kernels compute every output element as one
fma chain whatever the batch size, EXCEPT in the tiles the tail split cuts (launches of more than 256 tiles whose last round
over the CUs is partly filled: which tiles those are depends on the batch size).  So: bit-identical histograms for any
grouping with the split off, the same table and histograms within a stated distance with it on (last test).   pytest -m gpu"""
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


def test_unreadable_file_in_the_middle_of_a_group_does_not_leak_its_staging_slot(tmp_path):
    """a group with an unreadable file that is NOT at its head used to come back as a compacted COPY of the
    staging slot, the slot stayed marked as handed out without ever getting an event, and the ring's next lap (it has six
    slots) died on 'staging ring too small for the look-ahead'.  Ten groups of four with holes in groups 1, 5 and 8 (one of
    them two holes wide) must calibrate, and give the tables of the readable files one per forward."""
    from PIL import Image
    rng = np.random.default_rng(11)
    paths = []
    for i in range(40):
        paths.append(str(tmp_path / ("im%02d.png" % i)))
        Image.fromarray(rng.integers(0, 256, (32, 32, 3), dtype=np.uint8)).save(paths[-1])
    for at in (6, 21, 22, 34):                                               # inside groups 1, 5, 5 and 8, never at a group head
        os.remove(paths[at])
    readable = [p for p in paths if os.path.exists(p)]
    t_one, h_one, m_one = _calibrate(readable, 0, 1, 100)
    t_grp, h_grp, m_grp = _calibrate(paths, 0, 4, 100)
    assert t_grp == t_one and np.array_equal(h_grp, h_one) and np.array_equal(m_grp, m_one)
    npy = []
    for i in range(40):                                                      # the same through the .npy reader's fallback path
        npy.append(str(tmp_path / ("a%02d.npy" % i)))
        np.save(npy[-1], cases.fixed_input((3, 32, 32), seed=700 + i).numpy())
    t_a, h_a, m_a = _calibrate(npy, 2, 1, 100)
    t_b, h_b, m_b = _calibrate(npy, 2, 4, 100)
    assert t_a == t_b and np.array_equal(h_a, h_b) and np.array_equal(m_a, m_b)


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


def _calibrate_net(net, items, mode, file_batch, max_cali, shape):
    from tools import Quantity

    class Q(Quantity):
        pass
    Q.file_batch = file_batch
    with product_workdir(device="gpu", max_cali_img_num=max_cali, input_shape=shape) as tmp:
        _set_mode(tmp, mode)
        q = Q(net)
        q.activation_quantize(items)
        table = open(os.path.join(tmp, "test", "workdir", "feat.table")).read()
        return table, q._collector.hist_device.cpu().numpy(), q._collector.max_device.cpu().numpy()


@pytest.mark.timeout(900)
def test_grouping_on_launches_of_more_than_256_tiles_with_the_tail_split_on_and_off(tmp_path):
    """one file per forward against 64 files per forward where a launch has MORE than 256 tiles (the earlier tests
    stay below that and never reach the tail split).  A 128 -> 256 1x1 convolution on 40 x 40 planes: 64 images are 102 400
    columns = 1 600 tiles of 128 x 128 (6 rounds of 256 + 64: the last 64 tiles are cut into 4 K slices each), one image is
    26 tiles (never split).  Split OFF: every output is one fma chain whatever N -- maxima and histograms bit-identical.
    Split ON (the default): the 64-file launch computes 4 % of its outputs as sums of four partial chains, so a few values
    may land in a neighbouring bin -- same feat.table, maxima within 4 ulp, histograms within 1e-4 of their mass in L1."""
    from torch import nn
    from common.quantity import _native, View

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.c0 = nn.Conv2d(3, 128, 3, padding=1)
            self.r0 = nn.ReLU()
            self.c1 = nn.Conv2d(128, 256, 1)
            self.r1 = nn.ReLU()
            self.pool = nn.AvgPool2d(40)
            self.view = View()
            self.fc = nn.Linear(256, 10)

        def forward(self, x):
            return self.fc(self.view(self.pool(self.r1(self.c1(self.r0(self.c0(x)))))))

    torch.manual_seed(5)
    net = Net().eval().cuda()
    paths = []
    for i in range(65):                                                       # items 0 .. 64 are used: 64 + 1 files
        paths.append(str(tmp_path / ("im%02d.npy" % i)))
        np.save(paths[-1], cases.fixed_input((3, 40, 40), seed=300 + i).numpy())
    old = _native.conv_tail_split
    got = {}
    try:
        for split in (False, True):
            _native.conv_tail_split = split
            got[split] = (_calibrate_net(net, paths, 2, 1, 64, "1,3,40,40"), _calibrate_net(net, paths, 2, 64, 64, "1,3,40,40"))
    finally:
        _native.conv_tail_split = old
    (t1, h1, m1), (t64, h64, m64) = got[False]
    assert t1 == t64 and np.array_equal(h1, h64) and np.array_equal(m1, m64)
    (s1, g1, n1), (s64, g64, n64) = got[True]
    assert s1 == s64 == t1                                                    # the table does not move
    assert np.array_equal(g1, h1) and np.array_equal(n1, m1)                  # one image per forward never splits
    assert np.allclose(n64, m1, rtol=4 * 2.0 ** -23, atol=0)
    mass = h1.sum(axis=1).astype(np.float64)
    dist = np.abs(g64.astype(np.int64) - h1.astype(np.int64)).sum(axis=1)
    assert np.array_equal(g64.sum(axis=1), h1.sum(axis=1)) and float((dist / np.maximum(mass, 1)).max()) <= 1e-4


def test_the_upload_of_the_next_host_batch_is_issued_before_this_one_is_handed_out():
    """_device_items (round 4): a host batch's copy rides a side stream, and the copy of batch i + 1 is ISSUED before batch i is
    handed to the caller (whose kernels it then runs beside) -- one item of look-ahead, never more.  A loader that refills its own
    pinned buffer is not pulled ahead: its copy must be done before the loader is asked again, and waiting for that in front of
    the current batch's launches would stall the host.  The batches arrive unchanged and in order either way."""
    from tools import Quantity
    with product_workdir(device="gpu", max_cali_img_num=100):
        q = Quantity(_r18())
        for pinned, want in ((False, ["pull0", "pull1", "got0", "pull2", "got1", "pull3", "got2", "got3"]),
                             (True, ["pull0", "got0", "pull1", "got1", "pull2", "got2", "pull3", "got3"])):
            log = []
            host = [torch.full((2, 3, 32, 32), float(k)) for k in range(4)]
            if pinned:
                host = [h.pin_memory() for h in host]

            def loader():
                for k, h in enumerate(host):
                    log.append("pull%d" % k)
                    yield (h, 0)
            got = []
            for i, x in q._device_items(loader()):
                log.append("got%d" % i)
                assert x.is_cuda
                got.append(x)
            torch.cuda.synchronize()
            assert log == want, (pinned, log)
            assert all(torch.equal(g.cpu(), h) for g, h in zip(got, host))
