"""Phase stamps of ONE item of conv_wino44r.hip (library built with -DW44R_PROBE): the per-launch fixed cost -- entry, border
zeroing, the two fill stages, first V task, chunk loop, the four output-transform passes.  Run with DDPM_CONV_WINO44=2 (any launch
size, no channel split) so that small launches stay on this kernel.
    DDPM_CONV_WINO44=2 DDPM_OOD_HIP_LIB=$PWD/abl_lib/lib_PROBE.so python tools/w44r_phase_probe.py [B C1 C2 Cout H]"""
import ctypes as C, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ddpm_ood_amd import ops, _lib
from ddpm_ood_amd._lib import ConvDesc
dev = torch.device("cuda:0")
lib = _lib.load()
B, C1, C2, Cout, H = (int(v) for v in (sys.argv[1:6] if len(sys.argv) > 5 else (128, 256, 0, 256, 8)))
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
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3):
    assert lib.ddpm_conv_f32(C.byref(d), None) == 0, lib.ddpm_last_error()
torch.cuda.synchronize()
e0.record()
for _ in range(20):
    lib.ddpm_conv_f32(C.byref(d), None)
e1.record()
torch.cuda.synchronize()
print(f"B={B} {C1}+{C2}->{Cout}@{H}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per launch (back to back)")
t = dbg.cpu()[512:512 + 8 * 32].reshape(8, 32).tolist()
names = ("zero+sync", "to item", "issue st0", "wait+act st0", "stage 1", "reqs+barrier", "A loads+V task 0+zero", "barrier",
         "chunk loop", "nops+prep", "pass 0", "pass 1", "pass 2", "pass 3")
for wv in range(8):
    r = [t[wv][31]] + t[wv][:14]
    print(f"wave {wv}: " + "  ".join(f"{n} {r[i + 1] - r[i]}" for i, n in enumerate(names)) + f"  | total {r[14] - r[0]}")
