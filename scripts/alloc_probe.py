#!/usr/bin/env python3
"""GPU probe: what does keeping activations alive cost? (allocator growth vs one arena)"""
import os, sys, time
import torch
def sync(): torch.cuda.synchronize()
free, total = torch.cuda.mem_get_info(); print("free %.1f GB total %.1f GB" % (free/2**30, total/2**30))
print("alloc conf:", os.environ.get("PYTORCH_HIP_ALLOC_CONF"), os.environ.get("PYTORCH_CUDA_ALLOC_CONF"))
mode = sys.argv[1] if len(sys.argv) > 1 else "grow"
if mode == "grow":
    keep = []
    sizes = [802816*64*4]*30 + [200704*64*4]*41          # ~ one R50 batch of cared tensors (bytes)
    for step in range(44):
        sync(); t0 = time.perf_counter()
        batch = [torch.empty(s, dtype=torch.uint8, device="cuda") for s in sizes]
        for b in batch[:3]: b.fill_(1)
        sync(); dt = time.perf_counter() - t0
        keep.append(batch)
        if step % 4 == 0 or dt > 0.02:
            print("step %2d: alloc+touch %.1f ms, held %.1f GB" % (step, dt*1e3, torch.cuda.memory_allocated()/2**30))
elif mode == "arena":
    for gb in (8, 64, 160):
        sync(); t0 = time.perf_counter()
        a = torch.empty(gb * 2**30, dtype=torch.uint8, device="cuda")
        sync(); t1 = time.perf_counter()
        a[::4096].fill_(1)
        sync(); t2 = time.perf_counter()
        a.fill_(0); sync(); t3 = time.perf_counter()
        src = torch.empty(4 * 2**30, dtype=torch.uint8, device="cuda"); sync(); t4 = time.perf_counter()
        a[:4 * 2**30].copy_(src); sync(); t5 = time.perf_counter()
        print("arena %3d GB: malloc %.1f ms, sparse touch %.1f ms, full fill %.1f ms, 4GB d2d copy %.2f ms" %
              (gb, (t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, (t5-t4)*1e3))
        del a, src; torch.cuda.empty_cache()
