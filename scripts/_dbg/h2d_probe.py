"""GPU box: host -> device copy rate of a pinned 154 MB buffer (one 256-image batch), alone, in 1 / 2 / 4 chunks on as many streams."""
import time, torch
n = 256 * 3 * 224 * 224
h = torch.empty(n, dtype=torch.float32, pin_memory=True).normal_()
d = torch.empty(n, dtype=torch.float32, device="cuda")
for chunks in (1, 2, 4, 8):
    streams = [torch.cuda.Stream() for _ in range(chunks)]
    step = n // chunks
    def go():
        for c, s in enumerate(streams):
            with torch.cuda.stream(s):
                d[c * step:(c + 1) * step].copy_(h[c * step:(c + 1) * step], non_blocking=True)
    go(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5): go()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 5
    print("%d chunk(s): %.2f ms = %.1f GB/s" % (chunks, dt * 1e3, n * 4 / dt / 1e9))
