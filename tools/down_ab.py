"""A/B of the Downsample convolution (3x3, stride 2): conv_mfma_kernel (fp32 MFMA) against conv_s2h_kernel (direct, split-f16
operands on the f16 MFMA).   python tools/down_ab.py [B]"""
import math, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from ddpm_ood_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda:0")
for C, H in ((128, 32), (256, 16)):
    x = torch.randn(B, C, H, H, device=dev)
    w = torch.randn(C, C, 3, 3, device=dev) / math.sqrt(C * 9)
    b = torch.randn(C, device=dev)
    ws = ops.pack_conv_s2h_weight(w)
    for name, kw in (("conv_mfma", dict()), ("conv_s2h", dict(wino44h=ws))):
        for _ in range(3): y = ops.conv(x, w, b, mode=ops.CONV_STRIDE2, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): y = ops.conv(x, w, b, mode=ops.CONV_STRIDE2, **kw)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        fl = 2.0 * B * C * C * 9 * (H // 2) ** 2
        print(f"{C}->{C}@{H}->{H//2} B={B} {name:10s} {us:8.1f} us  {fl / us / 1e6:7.1f} TFLOP/s")
