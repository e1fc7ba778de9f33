"""Cycle stamps of four steady-state chunk intervals of conv_wino44r.hip (library built with -DW44R_PROBE: tools/w44r_abl.sh
"PROBE:-DW44R_PROBE"): per wave the cycles spent in the pixel-load issue, the MFMA segment, the activation, the V task and the
barrier(s).    DDPM_OOD_HIP_LIB=$PWD/abl_lib/lib_PROBE.so python tools/w44r_probe.py [B C1 C2 Cout H]"""
import ctypes as C, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ddpm_ood_amd import ops, _lib
from ddpm_ood_amd._lib import ConvDesc
dev = torch.device("cuda:0")
lib = _lib.load()
B, C1, C2, Cout, H = (int(v) for v in (sys.argv[1:6] if len(sys.argv) > 5 else (1024, 128, 0, 128, 32)))
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
t = dbg.cpu().reshape(-1, 16)[:32].tolist()
names = ("loads", "mfma", "act", "bar4-7", "vtask", "bar0-3")
for wv in range(8):
    print(f"wave {wv}")
    for iv in range(4):
        r = t[wv * 4 + iv]
        seg = "  ".join(f"{n} {r[i + 1] - r[i]:5d}" for i, n in enumerate(names))
        nxt = f"  (next +{t[wv * 4 + iv + 1][0] - r[6]})" if iv < 3 else ""
        vt = f"  | vtask: col0 {r[8] - r[4]:5d} col1 {r[9] - r[8]:5d} row+split {r[10] - r[9]:5d} store {r[11] - r[10]:5d} rest {r[5] - r[11]:5d}" if r[8] and r[11] else ""
        print(f"  interval {4 + iv}: {seg}  total {r[6] - r[0]:5d}{nxt}{vt}")
