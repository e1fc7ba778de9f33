"""Cycle stamps of one workgroup of the F(4x4) kernel (needs a library built with -DW44_PROBE; development tool)."""
import ctypes as C
import math
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402
from ddpm_ood_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.load()
for H, Cin, Cout, B in ((32, 128, 128, 256), (32, 384, 128, 256), (16, 256, 256, 256)):
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(B, Cin, H, H, device=dev, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, device=dev, generator=g) / math.sqrt(Cin * 9)
    b = torch.randn(Cout, device=dev, generator=g)
    gs, gh = ops.gn_scale_shift(x, torch.ones(Cin, device=dev), torch.zeros(Cin, device=dev), 32, 1e-6)
    w44 = ops.pack_wino44_weight(w)
    out = torch.empty(B, Cout, H, H, device=dev)
    stamps = torch.zeros(64, dtype=torch.int64, device=dev)
    d = _lib.ConvDesc()
    d.in1, d.C1, d.w_raw, d.w_wino44, d.bias, d.out = x.data_ptr(), Cin, w.data_ptr(), w44.data_ptr(), b.data_ptr(), out.data_ptr()
    d.gscale, d.gshift = gs.data_ptr(), gh.data_ptr()
    d.B, d.Cout, d.Hi, d.Wi, d.Ho, d.Wo, d.ksize, d.act = B, Cout, H, H, H, H, 3, ops.ACT_SILU
    d.scratch, d.scratch_floats = stamps.data_ptr(), 128
    for _ in range(3):
        assert lib.ddpm_conv_f32(C.byref(d), None) == 0
    torch.cuda.synchronize()
    t = stamps.cpu().view(-1, 4)
    t = t[t[:, 0] > 0]
    nch = Cin // 4
    print(f"H={H} Cin={Cin}: items {len(t)}")
    for i, r in enumerate(t.tolist()):
        gap = (r[0] - t[i - 1][3].item()) if i else 0
        print(f"  item {i}: chunks 0-1 {r[1] - r[0]:7d} cyc | steady {(r[2] - r[1]) / (nch - 2):7.0f} cyc/chunk | epilogue {r[3] - r[2]:7d} | gap {gap}")
