"""conv_s2h timing vs input channels (slope = time per 8-channel chunk, intercept = per-workgroup overhead)."""
import math, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from ddpm_ood_amd import ops
dev = torch.device("cuda:0")
B, Co, H = 1024, 128, 32
for C in (8, 16, 32, 64, 128, 256):
    x = torch.randn(B, C, H, H, device=dev)
    w = torch.randn(Co, C, 3, 3, device=dev) / math.sqrt(C * 9)
    ws = ops.pack_conv_s2h_weight(w)
    for _ in range(3): y = ops.conv(x, w, None, mode=ops.CONV_STRIDE2, wino44h=ws)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): y = ops.conv(x, w, None, mode=ops.CONV_STRIDE2, wino44h=ws)
    e1.record(); torch.cuda.synchronize()
    print(f"Cin={C:4d} chunks={C//8:3d}  {e0.elapsed_time(e1) * 100:8.1f} us")
