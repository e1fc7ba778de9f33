import os, math, sys, json, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from ddpm_ood_amd import ops, _lib
dev = torch.device("cuda:0")
lib = _lib.load()
torch.manual_seed(0)
B, Cin, Cout, H = (int(v) for v in sys.argv[1:5])
x = torch.randn(B, Cin, H, H); w = torch.randn(Cout, Cin, 3, 3) / math.sqrt(9 * Cin)
b = torch.randn(Cout); temb = torch.randn(B, Cout); res = torch.randn(B, Cout, H, H)
wh = ops.pack_wino44h_weight(w.to(dev))
base = F.conv2d(x, w, padding=1)
for name, kw, ref in (("none", {}, base), ("bias", dict(bias=b), base + b[None, :, None, None]),
                      ("temb", dict(chan_add=temb), base + temb[:, :, None, None]), ("res", dict(residual=res), base + res)):
    lib.ddpm_prof_enable(1)
    y = ops.conv(x.to(dev), w.to(dev), kw.get("bias", torch.zeros(Cout)).to(dev), wino44h=wh,
                 chan_add=None if "chan_add" not in kw else kw["chan_add"].to(dev),
                 residual=None if "residual" not in kw else kw["residual"].to(dev)).cpu()
    lib.ddpm_prof_enable(0)
    buf = ctypes.create_string_buffer(1 << 16); lib.ddpm_prof_report(buf, len(buf))
    e = (y - ref).abs()
    print(name, "kernels", list(json.loads(buf.value.decode())), "max err", round(e.max().item(), 5),
          "err by n%8", [round(v, 3) for v in e.amax(dim=(1, 2, 3)).reshape(-1, 8).amax(0).tolist()],
          "by cout%32 blocks", [round(v, 2) for v in e.amax(dim=(0, 2, 3)).reshape(-1, 32).amax(1).tolist()][:8])
