"""Development: the small-launch kernels (conv_d3s.hip) forced onto larger launches (DDPM_CONV_D3S=2), one shape per process:
    python tools/d3s_debug.py B C1 C2 Cout H k [full]      (full: GroupNorm prologue, temb, residual, statistics -- as the engine)"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["DDPM_CONV_D3S"] = "2"
from ddpm_ood_amd import ops  # noqa: E402

B, C1, C2, Cout, H, k = (int(v) for v in sys.argv[1:7])
full = len(sys.argv) > 7
dev = "cuda"
g = torch.Generator().manual_seed(1)
x = torch.randn(B, C1, H, H, generator=g).to(dev)
x2 = torch.randn(B, C2, H, H, generator=g).to(dev) if C2 else None
w = (torch.randn(Cout, C1 + C2, k, k, generator=g) / math.sqrt((C1 + C2) * k * k)).to(dev)
b = torch.randn(Cout, generator=g).to(dev)
kw = {}
if full:
    Cin = C1 + C2
    gs, gh = ops.gn_scale_shift(x, torch.ones(Cin, device=dev), torch.zeros(Cin, device=dev), 32, 1e-6, x2=x2)
    kw = dict(gscale=gs, gshift=gh, act=ops.ACT_SILU if k == 3 else ops.ACT_NONE, chan_add=torch.randn(B, Cout + 64, generator=g).to(dev),
              chan_add_offset=32, residual=torch.randn(B, Cout, H, H, generator=g).to(dev), want_stats=True)
planes = ops.pack_conv_d3h_weight(w) if k == 3 else ops.pack_conv_d1s_weight(w)
y = ops.conv(x, w, b, x2=x2, d3h=planes, **kw)
torch.cuda.synchronize()
y0 = ops.conv(x, w, b, x2=x2, **kw)
torch.cuda.synchronize()
if full:
    print("stats", None if y[1] is None else tuple(y[1].shape), None if y0[1] is None else tuple(y0[1].shape))
    y, y0 = y[0], y0[0]
print(sys.argv[1:], "max diff", (y - y0).abs().max().item(), "equal" if torch.equal(y, y0) else "different kernels")
