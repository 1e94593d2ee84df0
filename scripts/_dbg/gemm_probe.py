import os, sys, torch
sys.path.insert(0, "pytorch-quantity_amd/quantity")
from common.quantity import _native as nat
def t(cin, cout, h, B, mode):
    x = torch.randn(B, cin, h, h, device="cuda"); wt = torch.randn(cin, cout, device="cuda").contiguous(); bias = torch.randn(cout, device="cuda")
    y = torch.empty(B, cout, h, h, device="cuda"); mx = torch.zeros(1, device="cuda")
    run = (lambda: nat.conv1x1_f32(x, wt, bias, 1, max_dev=mx, row=0, out=y)) if mode == "max" else (lambda: nat.conv1x1_f32(x, wt, bias, 1, out=y))
    run(); run()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(10): run()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    tiles = ((B*h*h + 127)//128) * ((cout+127)//128)
    print("%5d->%-5d %2dx%-2d b%-4d %s: %.3f ms %6.1f TFLOP/s  tiles %d (%.2f per 768 slots)" % (cin, cout, h, h, B, mode, ms, 2.0*B*cout*h*h*cin/ms/1e9, tiles, tiles/768.0))
for args in [(1024,1024,14,256), (2048,2048,14,256), (4096,4096,16,96), (512,256,28,256), (512,256,28,245), (512,256,28,1024), (1024,256,14,256), (1024,256,14,251), (1024,256,14,752)]:
    t(*args, "none")
t(512,256,28,245,"max"); t(1024,256,14,251,"max")
