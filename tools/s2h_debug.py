import math, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from ddpm_ood_amd import ops
dev = torch.device("cuda:0")
B, C, Co, H = 1, 8, 64, 32
w = torch.zeros(Co, C, 3, 3); w[:, 0, :, :] = torch.arange(1, 10).float().view(3, 3)  # tap id
ws = ops.pack_conv_s2h_weight(w.to(dev))
for (r, c) in ((0, 0), (0, 1), (1, 0), (5, 6), (5, 7), (31, 31)):
    x = torch.zeros(B, C, H, H); x[0, 0, r, c] = 1.0
    y = ops.conv(x.to(dev), w.to(dev), None, mode=ops.CONV_STRIDE2, wino44h=ws).cpu()
    ref = torch.nn.functional.conv2d(x, w, None, stride=2, padding=1)
    nz = y[0, 0].nonzero().tolist(); nzr = ref[0, 0].nonzero().tolist()
    print((r, c), "hip", [(p, float(y[0, 0, p[0], p[1]])) for p in nz][:8], "ref", [(p, float(ref[0, 0, p[0], p[1]])) for p in nzr][:8])
