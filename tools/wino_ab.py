"""A/B timing of the Winograd conv layers of the `small` UNet (development tool).
    python tools/wino_ab.py          # F(4x4, 3x3) where it applies (DDPM_CONV_WINO44=0: F(2x2, 3x3) everywhere;
                                     # DDPM_WINO44_F16X3=0: the fp32-MFMA F(4x4) kernel instead of the split-f16 one)
    DDPM_WINO_WAVES=4 python tools/wino_ab.py      # the four-wave F(2x2) kernel vs the default eight-wave one"""
import math
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402
from ddpm_ood_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
SHAPES = [(256, 128, 0, 128, 32), (256, 256, 128, 128, 32), (256, 256, 0, 256, 16), (256, 256, 256, 256, 16),
          (256, 256, 0, 256, 8), (256, 256, 256, 256, 8)]
if len(sys.argv) > 1:  # batch size of every shape, e.g. 16 for the small-batch regime
    SHAPES = [(int(sys.argv[1]),) + s[1:] for s in SHAPES]
tot = 0.0
for B, C1, C2, Cout, H in SHAPES:
    Cin = C1 + C2
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(B, C1, H, H, device=dev, generator=g)
    x2 = torch.randn(B, C2, H, H, device=dev, generator=g) if C2 else None
    w = torch.randn(Cout, Cin, 3, 3, device=dev, generator=g) / math.sqrt(Cin * 9)
    b = torch.randn(Cout, device=dev, generator=g)
    temb = torch.randn(B, Cout, device=dev, generator=g)
    pk, wn, w44, w44h = ops.pack_conv_weight(w), ops.pack_wino_weight(w), ops.pack_wino44_weight(w), ops.pack_wino44h_weight(w)
    gs, gh = ops.gn_scale_shift(x, torch.ones(Cin, device=dev), torch.zeros(Cin, device=dev), 32, 1e-6, x2=x2)
    f = lambda: ops.conv(x, w, b, x2=x2, gscale=gs, gshift=gh, act=ops.ACT_SILU, packed=pk, wino=wn, wino44=w44, wino44h=w44h, chan_add=temb)  # noqa: E731
    y = f()
    ref = ops.conv(x, w, b, x2=x2, gscale=gs, gshift=gh, act=ops.ACT_SILU, packed=pk, chan_add=temb)
    err = (y - ref).abs().max().item()
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 20
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    tot += ms
    fl = 2.0 * B * H * H * Cout * Cin * 9
    print(f"{C1}+{C2}->{Cout}@{H}: {ms * 1e3:8.1f} us  {fl / ms / 1e9:7.2f} alg TFLOP/s  ({fl / ms / 1e9 * 16 / 36 / 157.3:.3f} of MFMA peak)  "
          f"err vs direct {err:.1e}", flush=True)
print(f"REG={os.environ.get('DDPM_W44H_REG', '1')} F16X3={os.environ.get('DDPM_WINO44_F16X3', '1')} WAVES={os.environ.get('DDPM_WINO_WAVES', '8')} WINO44={os.environ.get('DDPM_CONV_WINO44', '1')} SPLIT44={os.environ.get('DDPM_WINO44_SPLIT', '4')} total {tot * 1e3:.1f} us", flush=True)
