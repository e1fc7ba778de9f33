import math, sys
sys.path.insert(0, "/root/repo")
import torch
from ddpm_ood_amd import ops
dev = torch.device("cuda:0")
for H, Cout, Cin in ((32,128,128),(32,128,384),(16,256,256)):
    for B in (64, 128, 192, 256, 320, 384, 512):
        if H == 16 and B < 128: continue
        g = torch.Generator(device=dev).manual_seed(1)
        x = torch.randn(B, Cin, H, H, device=dev, generator=g)
        w = torch.randn(Cout, Cin, 3, 3, device=dev, generator=g) / math.sqrt(Cin * 9)
        b = torch.randn(Cout, device=dev, generator=g)
        gs, gh = ops.gn_scale_shift(x, torch.ones(Cin, device=dev), torch.zeros(Cin, device=dev), 32, 1e-6)
        pk, w44 = ops.pack_conv_weight(w), ops.pack_wino44_weight(w)
        f = lambda: ops.conv(x, w, b, gscale=gs, gshift=gh, act=ops.ACT_SILU, packed=pk, wino44=w44)
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        print(f"H={H} Cin={Cin} B={B}: {e0.elapsed_time(e1)*100:8.1f} us", flush=True)
