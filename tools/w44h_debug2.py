import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["DDPM_CONV_WINO44"] = "2"
import torch, torch.nn.functional as F
from ddpm_ood_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
B, Cin, Cout, H = (int(v) for v in sys.argv[1:5])
x = torch.randn(B, Cin, H, H); w = torch.randn(Cout, Cin, 3, 3) / math.sqrt(9 * Cin)
wh = ops.pack_wino44h_weight(w.to(dev))
ref = F.conv2d(x, w, padding=1)
ys = [ops.conv(x.to(dev), w.to(dev), torch.zeros(Cout, device=dev), wino44h=wh).cpu() for _ in range(3)]
print("run-to-run identical:", torch.equal(ys[0], ys[1]), torch.equal(ys[1], ys[2]))
for y in ys[:2]:
    e = (y - ref).abs()
    print("max err", e.max().item(), "count > 1e-4:", int((e > 1e-4).sum()), "of", e.numel())
    idx = (e > 1e-4).nonzero()
    from collections import Counter
    print(" by n:", sorted(Counter(idx[:, 0].tolist()).items()))
    print(" by cout:", sorted(Counter(idx[:, 1].tolist()).items())[:40])
    print(" by row:", sorted(Counter(idx[:, 2].tolist()).items()))
    print(" by col:", sorted(Counter(idx[:, 3].tolist()).items()))
    top = e.flatten().topk(8)
    for v, i in zip(top.values.tolist(), top.indices.tolist()):
        n, r = divmod(i, Cout * H * H); c, r = divmod(r, H * H); yy, xx = divmod(r, H)
        print(f"  ({n},{c},{yy},{xx}) err {v:.5f} y {y[n, c, yy, xx].item():.5f} ref {ref[n, c, yy, xx].item():.5f}")
