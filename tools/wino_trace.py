"""Sampling timeline of the Winograd kernel's chunk loop (development tool, needs the WINO_TRACE builds in abl/)."""
import ctypes, glob, math, os, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
if len(sys.argv) > 1:
    os.environ["DDPM_OOD_HIP_LIB"] = sys.argv[1]
    import torch
    from ddpm_ood_amd import ops, _lib
    lib = _lib.load()
    dev = torch.device("cuda:0")
    B, C, H = 256, 128, 32
    x = torch.randn(B, C, H, H, device=dev)
    w = torch.randn(C, C, 3, 3, device=dev) / math.sqrt(C * 9)
    b = torch.randn(C, device=dev)
    gs, gh = ops.gn_scale_shift(x, torch.ones(C, device=dev), torch.zeros(C, device=dev), 32, 1e-6)
    ww = ops.pack_wino_weight(w)
    f = lambda: ops.conv(x, w, b, gscale=gs, gshift=gh, act=1, wino=ww)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    lib.ddpm_debug_wino_trace(None, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        f()
    e1.record()
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 4)()
    lib.ddpm_debug_wino_trace(out, 0)
    wgs = 5 * 2048
    chunks = out[2]
    print(f"{Path(sys.argv[1]).name}: {e0.elapsed_time(e1) / 5 * 1e3:7.1f} us/launch  period {out[0] / (chunks - wgs):8.1f}  "
          f"t(step) - t(0) = {out[1] / chunks:8.1f} cycles")
else:
    for so in sorted(glob.glob(str(ROOT / "abl" / "lib_trace*.so")), key=lambda s: int(s.split("trace")[-1][:-3])):
        subprocess.run([sys.executable, __file__, so])
