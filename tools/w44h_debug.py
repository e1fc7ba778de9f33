"""Development probe for conv_wino44h.hip: where (input channel / cout / pixel) does the output disagree with F.conv2d?"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["DDPM_CONV_WINO44"] = "2"
import torch, torch.nn.functional as F
from ddpm_ood_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)

def run(x, w, b=None):
    wh = ops.pack_wino44h_weight(w.to(dev))
    bb = torch.zeros(w.shape[0]) if b is None else b
    return ops.conv(x.to(dev), w.to(dev), bb.to(dev), wino44h=wh).cpu()

B, Cin, Cout, H = (int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (2, 16, 64, 32)))
x = torch.randn(B, Cin, H, H); w = torch.randn(Cout, Cin, 3, 3) / 12
ref = F.conv2d(x, w, padding=1); y = run(x, w)
e = (y - ref)
print("full: max err", e.abs().max().item(), "ref max", ref.abs().max().item(), "y max", y.abs().max().item())
print("ratio y/ref median", (y / ref).median().item())
print("err by image", e.abs().amax(dim=(1, 2, 3)).tolist())
print("err by cout (first 16)", [round(v, 3) for v in e.abs().amax(dim=(0, 2, 3))[:16].tolist()])
print("err by cout block of 8", [round(v, 3) for v in e.abs().amax(dim=(0, 2, 3)).reshape(8, 8).amax(1).tolist()])
em = e.abs().amax(dim=(0, 1))
print("err by image x cout-block", [[round(v, 2) for v in r] for r in e.abs().amax(dim=(2, 3)).reshape(B, -1, 8).amax(2).tolist()])
print("err map rows (max over cols):", [round(v, 2) for v in em.amax(1).tolist()])
print("err map cols (max over rows):", [round(v, 2) for v in em.amax(0).tolist()])
# one input channel at a time
for c in range(Cin):
    xc = torch.zeros_like(x); xc[:, c] = x[:, c]
    yc = run(xc, w); rc = F.conv2d(xc, w, padding=1)
    print(f"only channel {c}: max err {(yc - rc).abs().max().item():.4f} (ref max {rc.abs().max().item():.3f}), ratio {((yc * rc).sum() / (rc * rc).sum()).item():.4f}")
# one tap at a time, one channel
for tap in range(9):
    wt = torch.zeros_like(w); wt[:, :, tap // 3, tap % 3] = w[:, :, tap // 3, tap % 3]
    yt = run(x, wt); rt = F.conv2d(x, wt, padding=1)
    print(f"only tap {tap}: max err {(yt - rt).abs().max().item():.4f}, projection {((yt * rt).sum() / (rt * rt).sum()).item():.4f}")
# delta image, delta weight: where does the energy land?
xd = torch.zeros(B, Cin, H, H); xd[0, 3, 10 % H, 13 % H] = 1.0; xd[B - 1, 9, 2, 5] = 2.0
wd = torch.zeros(Cout, Cin, 3, 3); wd[5, 3, 1, 1] = 1.0; wd[7, 9, 1, 1] = 1.0
yd = run(xd, wd)
nz = (yd.abs() > 1e-3).nonzero()
print("delta probe: nonzeros", nz[:20].tolist(), "values", [round(yd[tuple(i)].item(), 4) for i in nz[:20]])
