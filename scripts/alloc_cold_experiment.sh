#!/bin/bash
# GPU box: what does a GB of device memory cost a COLD process, by API, on VRAM that an earlier process has
# used (what a calibration script normally finds) and on VRAM nothing has touched since the box came up?
# One API per process (scripts/alloc_probe.hip); between them a process that writes 200 GB and exits ("dirty").
# -> profiles/r04_alloc_probe.txt
cd "$(dirname "$0")/.."
GB=${1:-32}
dirty() { python3 -c "
import torch
xs = [torch.empty(20 << 30, dtype=torch.uint8, device='cuda').fill_(7) for _ in range(10)]
torch.cuda.synchronize(); print('  (a process wrote 200 GB and exited)')"; }
torch_alloc() { python3 -c "
import time, torch
torch.empty(1, device='cuda'); torch.cuda.synchronize()
t = time.perf_counter()
xs = [torch.empty(1 << 30, dtype=torch.uint8, device='cuda') for _ in range($GB)]
torch.cuda.synchronize(); t1 = time.perf_counter()
for x in xs: x[::4096].fill_(1)
torch.cuda.synchronize(); t2 = time.perf_counter()
print('torch %-28s: alloc %7.2f ms/GB   alloc + first touch %7.2f ms/GB' % (torch.cuda.get_allocator_backend(), (t1 - t) * 1e3 / $GB, (t2 - t) * 1e3 / $GB))"; }
echo "== first thing on this box"
scripts/_bin/alloc_probe $GB malloc
for mode in malloc async vmm; do
  dirty; echo "== $mode after the box's memory was used"
  scripts/_bin/alloc_probe $GB $mode
done
dirty; echo "== torch, native caching allocator"; torch_alloc
dirty; echo "== torch, PYTORCH_HIP_ALLOC_CONF=backend:cudaMallocAsync"; PYTORCH_HIP_ALLOC_CONF=backend:cudaMallocAsync torch_alloc
