"""GPU box: is a float convolution's output the same bits every time it runs?  Small shapes (few tiles: the tail split cuts them
into K slices) through fq_conv1x1_f32 / fq_conv_kxk_f32 / fq_conv3x3_wino_f32 / fq_conv_stem_f32, 30 runs each.
FQ_CONV_TAIL_SPLIT=0 for the unsplit form."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "pytorch-quantity_amd", "quantity"))
from common.quantity import _native as nat
SHAPES = [  # N, Cin, H, W, Cout, k, stride, pad
    (8, 256, 8, 8, 256, 3, 2, 1), (8, 256, 8, 8, 256, 3, 1, 1), (8, 64, 8, 8, 256, 1, 1, 0), (8, 256, 4, 4, 64, 1, 1, 0), (4, 128, 16, 16, 128, 3, 2, 1),
    (8, 8, 16, 16, 256, 1, 2, 0), (4, 1024, 4, 4, 256, 1, 1, 0), (8, 64, 12, 12, 64, 3, 1, 1), (8, 16, 16, 16, 128, 5, 1, 2), (256, 256, 14, 14, 256, 3, 2, 1)]
for (N, Cin, H, W, Cout, k, st, pd) in SHAPES:
    g = torch.Generator(device="cuda").manual_seed(N + Cin + Cout)
    x = torch.randn(N, Cin, H, W, device="cuda", generator=g)
    w = torch.randn(Cout, Cin, k, k, device="cuda", generator=g) * (Cin * k * k) ** -0.5
    b = torch.randn(Cout, device="cuda", generator=g)
    kinds = []
    if k == 1:
        wt = w.view(Cout, Cin).t().contiguous()
        kinds.append(("c1", lambda: nat.conv1x1_f32(x, wt, b, st)))
    else:
        wk = nat.pack_kxk_weight(w)
        kinds.append(("kxk", lambda: nat.conv_kxk_f32(x, wk, b, (k, k), st, pd)))
        if k == 3 and st == 1 and pd == 1 and nat.conv_wino_supported(N, Cin, H, W, Cout):
            ww = nat.pack_wino_weight(w)
            kinds.append(("wino", lambda: nat.conv_wino_f32(x, ww, b, Cout)))
    for name, fn in kinds:
        ref = fn().clone()
        diff = 0
        for _ in range(30):
            y = fn()
            diff += int(not torch.equal(y.view(torch.int32), ref.view(torch.int32)))
        d = (fn() - ref).abs().max().item()
        print("%-5s N %3d Cin %4d %2dx%-2d -> %4d k%d s%d: %2d of 30 runs differ from the first  (max |diff| %.2e)" % (name, N, Cin, H, W, Cout, k, st, diff, d))
