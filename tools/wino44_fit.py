"""Per-chunk / per-item / per-launch cost of the F(4x4) Winograd kernel from a batch x channel sweep (development tool)."""
import math
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from ddpm_ood_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
rows = []
for H, Cout in ((32, 128), (16, 256)):
    for Cin in (128, 256, 512):
        for B in (256, 512, 768):
            g = torch.Generator(device=dev).manual_seed(1)
            x = torch.randn(B, Cin, H, H, device=dev, generator=g)
            w = torch.randn(Cout, Cin, 3, 3, device=dev, generator=g) / math.sqrt(Cin * 9)
            b = torch.randn(Cout, device=dev, generator=g)
            gs, gh = ops.gn_scale_shift(x, torch.ones(Cin, device=dev), torch.zeros(Cin, device=dev), 32, 1e-6)
            pk, w44 = ops.pack_conv_weight(w), ops.pack_wino44_weight(w)
            f = lambda: ops.conv(x, w, b, gscale=gs, gshift=gh, act=ops.ACT_SILU, packed=pk, wino44=w44)  # noqa: E731
            for _ in range(3):
                f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                f()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 100
            tiles = (H // 4) ** 2
            items_per_wg = (Cout // 64) * B * tiles / 32 / 256
            rows.append((H, Cin, B, us, items_per_wg, Cin // 4))
            print(f"H={H} Cin={Cin} B={B}: {us:8.1f} us  items/WG {items_per_wg:.1f}  chunks/item {Cin // 4}", flush=True)
for H in (32, 16):
    r = [x for x in rows if x[0] == H]
    A = np.array([[x[4] * x[5], x[4], 1.0] for x in r])  # us = c * chunks + E * items + F
    y = np.array([x[3] for x in r])
    sol, *_ = np.linalg.lstsq(A, y, rcond=None)
    print(f"H={H}: per chunk {sol[0]:.3f} us, per item {sol[1]:.2f} us, per launch {sol[2]:.1f} us; max resid {np.abs(A @ sol - y).max():.1f} us")
