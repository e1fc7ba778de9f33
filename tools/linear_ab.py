import sys, time, os
sys.path.insert(0, "/root/repo")
import torch
from ddpm_ood_amd import ops
dev = torch.device("cuda:0")
for B, K, N, act in [(256,128,512,0),(256,512,512,1),(256,512,2432,1),(16,512,2432,1)]:
    x = torch.randn(B,K,device=dev); w = torch.randn(N,K,device=dev)/K**.5; b = torch.randn(N,device=dev)
    y = ops.conv(x,w,b,act=act)
    ref = torch.nn.functional.linear(torch.nn.functional.silu(x) if act else x, w, b)
    err = (y-ref).abs().max().item()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): ops.conv(x,w,b,act=act)
    e1.record(); torch.cuda.synchronize()
    print(os.environ.get("DDPM_LINEAR_SKINNY","1"), B,K,N,act, "err", err, "us", e0.elapsed_time(e1)*1000/200)
