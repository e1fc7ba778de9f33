"""Timing of the attention kernel at the `big` UNet's shapes (development tool): python tools/attn_ab.py"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402
from ddpm_ood_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
for B, heads, N in [(8, 1, 4096), (8, 2, 1024), (8, 3, 256), (256, 1, 64)]:
    C = 256 * heads
    g = torch.Generator(device=dev).manual_seed(1)
    qkv = torch.randn(B, 3 * C, N, device=dev, generator=g)
    res = torch.randn(B, C, N, device=dev, generator=g)
    y = ops.attention(qkv, res, heads, 1.0 / 16.0)
    q, k, v = (t.reshape(1, heads, 256, N) for t in qkv[:1].split(C, dim=1))
    ref = torch.einsum("bhij,bhdj->bhdi", (torch.einsum("bhdi,bhdj->bhij", q, k) / 16.0).softmax(-1), v).reshape(1, C, N) + res[:1]
    err = (y[:1] - ref).abs().max().item()
    for _ in range(3):
        ops.attention(qkv, res, heads, 1.0 / 16.0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 10
    for _ in range(n):
        ops.attention(qkv, res, heads, 1.0 / 16.0)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    fl = 4.0 * B * N * N * C
    print(f"B={B} heads={heads} N={N}: {ms * 1e3:8.1f} us  {fl / ms / 1e9:7.2f} TFLOP/s ({fl / ms / 1e9 / 157.3:.3f})  err {err:.1e}", flush=True)
