"""Timing of the attention kernels at the `big` UNet's shapes (development tool): python tools/attn_ab.py
new = register-resident kernel + its pre-pass (attention_fa.hip, caller scratch), old = LDS-exchange kernel (attention.hip)."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import os  # noqa: E402

os.environ.setdefault("DDPM_ATTN_FA", "2")
import torch  # noqa: E402
from ddpm_ood_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for B, heads, N in [(8, 1, 4096), (16, 1, 4096), (8, 2, 1024), (16, 2, 1024), (8, 3, 256), (16, 3, 256), (256, 1, 64), (1024, 1, 64)]:
    C = 256 * heads
    g = torch.Generator(device=dev).manual_seed(1)
    qkv = torch.randn(B, 3 * C, N, device=dev, generator=g)
    res = torch.randn(B, C, N, device=dev, generator=g)
    q, k, v = (t.reshape(1, heads, 256, N) for t in qkv[:1].split(C, dim=1))
    ref = torch.einsum("bhij,bhdj->bhdi", (torch.einsum("bhdi,bhdj->bhij", q, k) / 16.0).softmax(-1), v).reshape(1, C, N) + res[:1]
    fl = 4.0 * B * N * N * C
    for name, scr in (("old", False), ("new", True)):
        y = ops.attention(qkv, res, heads, 1.0 / 16.0, use_scratch=scr)
        err = (y[:1] - ref).abs().max().item()
        ms = timed(lambda: ops.attention(qkv, res, heads, 1.0 / 16.0, use_scratch=scr))
        print(f"B={B} heads={heads} N={N} {name}: {ms * 1e3:8.1f} us  {fl / ms / 1e9:7.2f} fp32-equivalent TFLOP/s  err {err:.1e}", flush=True)
