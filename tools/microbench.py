"""Per-kernel timing of UNet forwards through the library's hipEvent profiler (development tool).

    python tools/microbench.py [--batch 256] [--size 32] [--iters 3] [--model small]

Prints wall time per forward and, per kernel class, launches / ms / achieved TFLOP/s / GB/s.
"""

import argparse
import ctypes
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--size", type=int, default=32)
    ap.add_argument("--channels", type=int, default=1)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--model", default="small")
    ap.add_argument("--dims", type=int, default=2)
    ap.add_argument("--graph", default="0", help="DDPM_UNET_GRAPH: 0 | 1")
    a = ap.parse_args()
    import os
    os.environ["DDPM_UNET_GRAPH"] = a.graph

    from ddpm_ood_amd import DiffusionModelUNet, _lib
    from ddpm_ood_amd.synthetic import random_state_dict
    from ddpm_ood_amd.trainer import MODEL_CONFIGS

    lib = _lib.load()
    dev = torch.device("cuda:0")
    m = DiffusionModelUNet(a.dims, a.channels, a.channels, **MODEL_CONFIGS[a.model])
    m.load_state_dict(random_state_dict(a.model, a.channels, spatial_dims=a.dims, seed=1))
    m = m.to(dev).eval()
    x = torch.randn((a.batch, a.channels) + (a.size,) * a.dims, device=dev)
    t = torch.full((a.batch,), 500, dtype=torch.int64, device=dev)
    t0 = time.perf_counter()
    m(x, timesteps=t)
    torch.cuda.synchronize()
    print(f"first forward (incl. weight packing): {time.perf_counter() - t0:.3f} s", flush=True)
    for _ in range(2):
        m(x, timesteps=t)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        m(x, timesteps=t)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.iters
    print(f"forward wall: {dt * 1e3:.2f} ms  (B={a.batch}, {a.size}x{a.size}, {a.model})", flush=True)

    lib.ddpm_prof_enable(1)
    for _ in range(a.iters):
        m(x, timesteps=t)
    lib.ddpm_prof_enable(0)
    torch.cuda.synchronize()
    buf = ctypes.create_string_buffer(1 << 16)
    lib.ddpm_prof_report(buf, len(buf))
    prof = json.loads(buf.value.decode())
    tot = sum(v["ms"] for v in prof.values())
    print(f"{'kernel':28s} {'launch':>6s} {'ms/fwd':>9s} {'%':>6s} {'TFLOP/s':>8s} {'GB/s':>8s}")
    for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"]):
        ms = v["ms"] / a.iters
        print(f"{k:28s} {v['launches'] // a.iters:6d} {ms:9.3f} {100 * v['ms'] / tot:6.1f} "
              f"{v['flops'] / v['ms'] / 1e9:8.2f} {v['bytes'] / v['ms'] / 1e6:8.1f}")
    print(f"sum of kernel time: {tot / a.iters:.2f} ms/forward", flush=True)


if __name__ == "__main__":
    main()
