#!/usr/bin/env python3
"""GPU box: where the time of the file-input path goes (no model): np.save N files, then group decode alone, pinned
allocation alone, H2D alone."""
import os, sys, tempfile, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "pytorch-quantity_amd", "quantity"))
from tools.pytorch_quantizer import Quantity, _FileGroup
N, B = 2048, 256
d = tempfile.mkdtemp(prefix="fq_probe_")
x = np.random.randn(3, 224, 224).astype(np.float32)
paths = []
for i in range(N):
    paths.append(os.path.join(d, "i%05d.npy" % i)); np.save(paths[-1], x)
q = Quantity.__new__(Quantity)
q.user_config = {"PRE_PROCESS": {"IMG": 2}}
q.device = "gpu"
print("cpus", os.cpu_count(), "decode workers", q.decode_workers)
for rep in range(3):
    t0 = time.perf_counter()
    for g in range(0, N, B):
        b = q._preprocess_group(_FileGroup(paths[g:g + B]), 2)
    dt = time.perf_counter() - t0
    print("group decode: %.3f s for %d files = %.0f files/s (%.1f GB/s)" % (dt, N, N / dt, N * x.nbytes / dt / 1e9))
t0 = time.perf_counter()
for _ in range(8):
    t = torch.empty((B, 3, 224, 224), dtype=torch.float32, pin_memory=True); del t
print("pinned alloc+free x8: %.3f s" % (time.perf_counter() - t0))
keep = [torch.empty((B, 3, 224, 224), dtype=torch.float32, pin_memory=True) for _ in range(3)]
t0 = time.perf_counter()
print("3 live pinned buffers: %.3f s" % (time.perf_counter() - t0))
torch.cuda.synchronize(); t0 = time.perf_counter()
for k in range(8):
    dev = keep[k % 3].cuda(non_blocking=True)
torch.cuda.synchronize()
print("H2D 8 x %d MB: %.3f s = %.1f GB/s" % (keep[0].numel() * 4 >> 20, time.perf_counter() - t0, 8 * keep[0].numel() * 4 / (time.perf_counter() - t0) / 1e9))
# raw read into one pinned buffer, single thread
v = keep[0].numpy(); h = Quantity._npy_header(paths[0])[0]
t0 = time.perf_counter()
for j in range(B):
    Quantity._read_npy_into(paths[j], v[j], h)
dt = time.perf_counter() - t0
print("single-thread readinto: %.0f files/s (%.2f GB/s)" % (B / dt, B * x.nbytes / dt / 1e9))
# sequential group decode as the pipeline calls it (header compare, plain loop), main thread and helper thread
for rep in range(2):
    t0 = time.perf_counter()
    for g in range(0, N, B):
        b = q._preprocess_group(_FileGroup(paths[g:g + B]), 2)
    dt = time.perf_counter() - t0
    print("sequential group decode (main thread): %.3f s = %.0f files/s" % (dt, N / dt))
from concurrent.futures import ThreadPoolExecutor
ex = ThreadPoolExecutor(1)
t0 = time.perf_counter()
for g in range(0, N, B):
    b = ex.submit(q._preprocess_group, _FileGroup(paths[g:g + B]), 2).result()
dt = time.perf_counter() - t0
print("sequential group decode (helper thread): %.3f s = %.0f files/s" % (dt, N / dt))
# the whole input pipeline without a model
import yaml
q.prefetch_inputs = True
q._max_img_num = N - 1
q.file_batch = B
q._file_kept, q._file_kept_bytes = {}, 0
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 0
for i, img in q._device_items(paths):
    n += img.shape[0]
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("_device_items alone: %d images in %.3f s = %.0f img/s; waits %s" % (n, dt, n / dt, {k: round(v, 3) for k, v in q.input_wait_s.items()}))
