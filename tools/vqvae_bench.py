"""Time the README-configuration VQ-VAE (4 stride-2 levels, 256 ch, 3 res layers, 2048 x 128 codes) on the device:
encode 128^3 -> 8^3 x 128, re-quantise + decode (development tool for SURVEY 8(f) row f-1)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from ddpm_ood_amd.vqvae import VQVAE

cfg = dict(spatial_dims=3, in_channels=1, out_channels=1, num_channels=(256,) * 4, num_res_layers=3,
           num_res_channels=(256,) * 4, downsample_parameters=((2, 4, 1, 1),) * 4,
           upsample_parameters=((2, 4, 1, 1, 0),) * 4, num_embeddings=2048, embedding_dim=128)
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = VQVAE(**cfg).to(dev).eval()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
x = torch.rand(B, 1, 128, 128, 128, device=dev)
with torch.no_grad():
    for name, f, arg in (("encode_stage_2_inputs", m.encode_stage_2_inputs, x),):
        z = f(arg); torch.cuda.synchronize()
        t0 = time.perf_counter(); z = f(arg); torch.cuda.synchronize()
        print(f"{name}: {(time.perf_counter() - t0) * 1e3:.1f} ms for B={B}", flush=True)
    y = m.decode_stage_2_outputs(z); torch.cuda.synchronize()
    t0 = time.perf_counter(); y = m.decode_stage_2_outputs(z); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # decoder FLOPs: per level 3 res units x 2 convs (27 taps, 256 -> 256) at 8^3..64^3 + first conv + transposed convs
    fl = 0
    for lv, s in enumerate((8, 16, 32, 64)):
        fl += 6 * 2 * s ** 3 * 256 * 256 * 27
        fl += 2 * (2 * s) ** 3 * 256 * (256 if lv < 3 else 1) * 8
    fl += 2 * 8 ** 3 * 128 * 256 * 27
    print(f"decode_stage_2_outputs: {dt * 1e3:.1f} ms for B={B}  ({fl * B / dt / 1e12:.1f} TFLOP/s, {fl / 1e12:.2f} TFLOP / volume)", flush=True)

    # per-kernel-class breakdown of one decode through the library's hipEvent profiler
    import ctypes, json
    from ddpm_ood_amd import _lib
    lib = _lib.load()
    lib.ddpm_prof_enable(1)
    m.decode_stage_2_outputs(z)
    lib.ddpm_prof_enable(0)
    torch.cuda.synchronize()
    buf = ctypes.create_string_buffer(1 << 18)
    lib.ddpm_prof_report(buf, len(buf))
    prof = json.loads(buf.value.decode())
    tot = sum(v["ms"] for v in prof.values())
    print(f"{'kernel':60s} {'launch':>6s} {'ms':>9s} {'%':>6s} {'TFLOP/s':>8s} {'GB/s':>8s}")
    for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"]):
        print(f"{k:60s} {v['launches']:6d} {v['ms']:9.3f} {100 * v['ms'] / tot:6.1f} "
              f"{v['flops'] / v['ms'] / 1e9:8.2f} {v['bytes'] / v['ms'] / 1e6:8.1f}")
    print(f"sum of kernel time: {tot:.2f} ms / decode", flush=True)
