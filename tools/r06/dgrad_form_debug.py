"""Development: which kernel a training input-gradient convolution runs at a small batch, and its error, per form."""
import ctypes, json, math, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
import torch.nn.functional as F
from ddpm_ood_amd import _lib, ops
from ddpm_ood_amd import train_ops as T

dev = torch.device("cuda:0")
lib = _lib.load()
def report():
    buf = ctypes.create_string_buffer(1 << 18)
    n = lib.ddpm_prof_report(buf, len(buf))
    return json.loads(buf.value.decode()) if n > 0 else {}
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
for cin, cout, H in ((256, 256, 32), (384, 128, 32), (512, 256, 16), (128, 128, 16), (256, 256, 8)):
    g = torch.Generator().manual_seed(1)
    w = torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(9 * cin)
    dy = torch.randn(B, cout, H, H, generator=g)
    ref = F.conv_transpose2d(dy.double(), w.double(), padding=1)
    wt = T.conv_weight_rot180t(w.to(dev))
    for form in ("wino44h", "wino", "direct"):
        lib.ddpm_prof_enable(1)
        y = ops.conv(dy.to(dev), wt, wino44h=ops.pack_wino44h_weight(wt) if form == "wino44h" else None,
                     wino=ops.pack_wino_weight(wt) if form in ("wino44h", "wino") else None)
        torch.cuda.synchronize()
        lib.ddpm_prof_enable(0)
        names = [k for k in report() if "pack" not in k]
        err = float((y.cpu().double() - ref).abs().max() / ref.abs().max())
        print(f"B={B} {cout}->{cin}@{H} form {form:8s}: max err {err:.2e}  kernels {names}")
