"""A/B timing of single conv layers (development tool): python tools/conv_ab.py"""
import sys, time, math
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import os
import torch
from ddpm_ood_amd import ops

dev = torch.device("cuda:0")
SHAPES = [(256, 128, 0, 128, 32, True), (256, 256, 128, 128, 32, True), (256, 256, 0, 256, 16, True),
          (256, 256, 0, 256, 8, True), (256, 256, 0, 256, 32, False)]
for B, C1, C2, Cout, H, gn in SHAPES:
    Cin = C1 + C2
    x = torch.randn(B, C1, H, H, device=dev)
    if os.environ.get('ZERO'):
        x.zero_()
    x2 = torch.randn(B, C2, H, H, device=dev) if C2 else None
    w = torch.randn(Cout, Cin, 3, 3, device=dev) / math.sqrt(Cin * 9)
    b = torch.randn(Cout, device=dev)
    if os.environ.get('ZERO'):
        w.zero_()
    pk = ops.pack_conv_weight(w)
    gs = gh = None
    if gn:
        gs, gh = ops.gn_scale_shift(x, torch.ones(Cin, device=dev), torch.zeros(Cin, device=dev), 32, 1e-6, x2=x2)
    f = lambda: ops.conv(x, w, b, x2=x2, gscale=gs, gshift=gh, act=int(gn), packed=pk)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 10
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    fl = 2.0 * B * H * H * Cout * Cin * 9
    print(f"{C1}+{C2}->{Cout}@{H} gn={gn}: {ms*1e3:8.1f} us  {fl/ms/1e9:7.2f} TFLOP/s", flush=True)
