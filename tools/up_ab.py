"""A/B of the Upsample convolution: conv_wino_up_kernel (F(2x2), 9 positions, f32 MFMA) against conv_wino44h_kernel's
upsample-on-load form (split-f16 F(4x4) over the virtual upsampled image).   python tools/up_ab.py [B]"""
import math, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from ddpm_ood_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda:0")
for C, H in ((256, 16), (256, 8)):
    x = torch.randn(B, C, H, H, device=dev)
    w = torch.randn(C, C, 3, 3, device=dev) / math.sqrt(C * 9)
    b = torch.randn(C, device=dev)
    wino, wh = ops.pack_wino_weight(w), ops.pack_wino44h_weight(w)
    for name, kw in (("wino_up", dict(wino=wino)), ("wino44h_up", dict(wino44h=wh))):
        for _ in range(3): y = ops.conv(x, w, b, mode=ops.CONV_UPSAMPLE2, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): y = ops.conv(x, w, b, mode=ops.CONV_UPSAMPLE2, **kw)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        fl = 2.0 * B * C * C * 9 * (2 * H) ** 2
        print(f"{C}->{C}@{H}->{2*H} B={B} {name:12s} {us:8.1f} us  {fl / us / 1e6:7.1f} alg TFLOP/s")
