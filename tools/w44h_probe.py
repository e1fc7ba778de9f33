"""Cycle stamps of one steady-state block of six phases of conv_wino44h.hip (library built with -DW44H_PROBE):
per phase and wave (0: producer group 0, 2: producer group 1, 4: pixel) the cycles from phase entry to the end of each MFMA job, to the
end of the waitcnt and to the barrier release."""
import ctypes as C, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ddpm_ood_amd import ops, _lib
from ddpm_ood_amd._lib import ConvDesc
dev = torch.device("cuda:0")
lib = _lib.load()
B, C1, C2, Cout, H = (int(v) for v in (sys.argv[1:6] if len(sys.argv) > 5 else (256, 256, 256, 256, 16)))
Cin = C1 + C2
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn(B, C1, H, H, device=dev, generator=g)
x2 = torch.randn(B, C2, H, H, device=dev, generator=g) if C2 else None
w = torch.randn(Cout, Cin, 3, 3, device=dev, generator=g) / math.sqrt(Cin * 9)
b = torch.randn(Cout, device=dev, generator=g)
wh = ops.pack_wino44h_weight(w)
gs, gh = ops.gn_scale_shift(x, torch.ones(Cin, device=dev), torch.zeros(Cin, device=dev), 32, 1e-6, x2=x2)
out = torch.empty(B, Cout, H, H, device=dev)
dbg = torch.zeros(4096, dtype=torch.int64, device=dev)
d = ConvDesc()
d.in1, d.C1 = x.data_ptr(), C1
if x2 is not None:
    d.in2, d.C2 = x2.data_ptr(), C2
d.w_raw, d.bias, d.gscale, d.gshift, d.out = w.data_ptr(), b.data_ptr(), gs.data_ptr(), gh.data_ptr(), out.data_ptr()
d.B, d.Cout, d.Hi, d.Wi, d.Ho, d.Wo, d.ksize, d.mode, d.act = B, Cout, H, H, H, H, 3, 0, 1
d.w_wino44h = wh.data_ptr()
d.scratch, d.scratch_floats = dbg.data_ptr(), 16  # too small for a split: only the probe writes here
for _ in range(3):
    assert lib.ddpm_conv_f32(C.byref(d), None) == 0, lib.ddpm_last_error()
torch.cuda.synchronize()
t = dbg.cpu().reshape(-1, 8)[:18].tolist()
for wi, name in enumerate(("producer g0 (wave 0)", "producer g1 (wave 2)", "pixel (wave 4)")):
    print(name)
    for ph in range(6):
        r = t[wi * 6 + ph]
        print(f"  phase {12 + ph}: jobs {r[1] - r[0]:5d} {r[2] - r[1]:5d} {r[3] - r[2]:5d}  wait {r[4] - r[3]:5d}  barrier {r[5] - r[4]:5d}  total {r[5] - r[0]:5d}"
              + (f"  (next entry +{t[wi * 6 + ph + 1][0] - r[5]})" if ph < 5 else ""))
