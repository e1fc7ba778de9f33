"""bench.py -- reconstructions/sec of the multi-t DDPM reconstruction hot path on MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): FashionMNIST-shaped 32x32x1 synthetic images, `small`
UNet with seeded random weights, 100 PLMS timesteps, inference_skip_factor=4 -> 25 t-starts,
1 250 UNet forwards per image, batch 256 per GPU.  One "step" = one batch of 256 images per
rank through the whole hot path (noise, add_noise, every PLMS trajectory, clamp + MSE, LPIPS,
score gather) = 6 400 reconstructions per rank, inputs resident in HBM when timing starts.
Weak scaling: every rank gets its own 256-image shard of a 256*N-image set; the only
collective is the per-step all_gather of the dense score tensor (RCCL).

One JSON line on rank 0.  `roofline` = the dominant kernel (the 3x3 conv with the GroupNorm+SiLU
prologue, Winograd F(2x2,3x3) on fp32 MFMA): algorithmic FLOPs / hipEvent-measured launch time, sampled in situ
(first UNet step of each of the 25 t-starts of the LAST timed step), against the 157.3 TFLOP/s
dense f32 MFMA peak.  `cpu_baseline` = the CPU oracle timed on this box's host cores on a
bounded sample of the same workload (rank 0, N = 1 only).
"""

import argparse
import ctypes
import json
import os
import shutil
import sys
import tempfile
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

F32_MFMA_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md, dense f32-input MFMA
BATCH = 256
SKIP = 4
SCHED = dict(beta_schedule="scaled_linear_beta", beta_start=0.0015, beta_end=0.0195)


def make_args(run_root, n_images, batch):
    ids = f"synthetic:blobs:n={n_images}:seed=0"
    return argparse.Namespace(
        seed=2, output_dir=str(run_root), model_name="fashionmnist_synthetic", validation_ids=ids, in_ids=ids,
        out_ids=ids, spatial_dimension=2, image_size=None, image_roi=None, latent_pad=None, vqvae_checkpoint=None,
        ddpm_checkpoint_epoch=None, prediction_type="epsilon", model_type="small", b_scale=1.0, snr_shift=1,
        simplex_noise=0, batch_size=batch, augmentation=0, cache_data=1, num_workers=0, first_n_val=None, first_n=None,
        eval_checkpoint=None, drop_last=False, is_grayscale=1, run_val=1, run_in=0, run_out=0,
        num_inference_steps=100, inference_skip_factor=SKIP, **SCHED)


def cpu_baseline_worker():
    """Runs in a fresh subprocess (own OpenMP pool): the oracle on the host cores.
    Sample: 8 images x t in {10, 490, 970} (inference_skip_factor=48) = 24 reconstructions,
    150 UNet forwards per image -- the same mean of 50 forwards per reconstruction as the timed
    workload.  kind "port": the reference's own dependencies cannot be installed."""
    import oracle
    from ddpm_ood_amd.data import get_data_loader
    from ddpm_ood_amd.synthetic import random_state_dict
    from ddpm_ood_amd.trainer import MODEL_CONFIGS, batch_noise

    model = oracle.DiffusionModelUNet(2, 1, 1, **MODEL_CONFIGS["small"]).eval()
    model.load_state_dict(random_state_dict("small", 1, seed=1))
    pl = oracle.PerceptualLoss(dimensions=2, include_pixel_loss=False, is_fake_3d=False, lpips_normalize=True)
    loader = get_data_loader("synthetic:blobs:n=8:seed=0", batch_size=8, is_grayscale=True)
    kw = dict(model=model, vqvae=oracle.PassthroughVQVAE(), perceptual=pl,
              noise_fn=lambda batch, t, shape: batch_noise(2, batch["index"], t, shape),
              beta_schedule=SCHED["beta_schedule"], beta_start=SCHED["beta_start"], beta_end=SCHED["beta_end"])
    oracle.get_scores(loader, "val", 1000, **kw)  # warm-up (oneDNN primitive creation): t = 10 only
    t0 = time.perf_counter()
    rows = oracle.get_scores(loader, "val", 48, **kw)
    dt = time.perf_counter() - t0
    assert sorted({r["t"] for r in rows}) == [10, 490, 970]
    print(json.dumps({"value": round(len(rows) / dt, 4), "unit": "reconstructions/s",
                      "cores": torch.get_num_threads(), "kind": "port", "seconds": round(dt, 2),
                      "sample": "8 images x t_start in {10, 490, 970}: 24 reconstructions, 1200 UNet "
                                "image-forwards (mean 50 per reconstruction as in the timed workload), CPU fp32 "
                                "oracle incl. LPIPS + MSE"}))


def cpu_baseline(_state_dict=None):
    """Oracle timing in a subprocess with a bounded thread count and a hard timeout: on the GPU
    box an OpenMP pool as wide as its 256 logical CPUs made these small convolutions crawl
    (> 8 min); min(cpu_count, 32) threads is what is used and reported as `cores`."""
    import subprocess

    threads = min(os.cpu_count() or 1, 32)
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads))
    try:
        out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--cpu-baseline-worker"], env=env,
                             capture_output=True, text=True, timeout=240)
        return json.loads(out.stdout.strip().splitlines()[-1])
    except Exception as e:  # timeout / parse error: report it, never fake a number
        return {"value": None, "unit": "reconstructions/s", "cores": threads, "kind": "port",
                "sample": f"failed: {type(e).__name__}: {e}"[:300]}


# kernels the roofline object may describe: profiler key -> (label, executed / algorithmic MFMA FLOPs)
ROOFLINE_KERNELS = {
    "conv3x3_wino_gn_silu": ("conv_wino_kernel<true, NR, ONEIMG> (3x3 conv as Winograd F(2x2,3x3), persistent: items of "
                             "64 couts x 64 tiles, GN+SiLU prologue, fp32 MFMA)", 16.0 / 36.0),
    "conv3x3_mfma_gn_silu": ("conv_mfma_kernel<9,1,true,128> (3x3 conv, 128x128 tile, GN+SiLU prologue, fp32 MFMA)", 1.0),
}


def roofline_of(prof):
    """The dominant kernel = the profiler class with the most time in the sampled UNet steps.
    `achieved` is ALGORITHMIC: the direct convolution's 2*9*Cin*Cout*pixels per launch
    over the hipEvent launch time, so with the Winograd kernel (which executes 16/36 of those
    multiplies) it may exceed what a direct convolution could reach; `mfma_executed_*` is the rate
    of MFMA work actually issued, the number to read against the 157.3 TFLOP/s pipe."""
    cands = [(v["ms"], k) for k, v in prof.items() if k in ROOFLINE_KERNELS and v["ms"] > 0]
    if not cands:
        return None
    _, key = max(cands)
    dom = prof[key]
    label, executed = ROOFLINE_KERNELS[key]
    achieved = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
    roofline = {"bound": "mfma", "kernel": label, "profile_key": key,
                "achieved": round(achieved, 2), "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / F32_MFMA_PEAK_TFLOPS, 4),
                "mfma_executed_tflops": round(achieved * executed, 2),
                "mfma_executed_frac": round(achieved * executed / F32_MFMA_PEAK_TFLOPS, 4),
                "launches_timed": dom["launches"], "avg_launch_ms": round(dom["ms"] / dom["launches"], 4),
                "flops_per_launch": dom["flops"] / dom["launches"],
                "algorithmic_GBps": round(dom["bytes"] / (dom["ms"] * 1e-3) / 1e9, 1),
                "traffic": None}
    pmc = ROOT / "profiles" / "pmc_traffic.json"  # written from a separate rocprofv3 --pmc pass, if any
    if pmc.exists():
        roofline["traffic"] = json.load(open(pmc)).get(key + "_bytes_per_launch")
    return roofline


_T0 = time.perf_counter()


def log(msg):
    """progress on stderr (stdout carries exactly one JSON line)"""
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench +{time.perf_counter() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=BATCH, help="images per GPU per step (256 = the reference default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-worker", action="store_true", help=argparse.SUPPRESS)
    a = ap.parse_args()
    if a.cpu_baseline_worker:
        return cpu_baseline_worker()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if a.gpus > 1 and world == 1:
            raise SystemExit("launch N > 1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", init_method="env://")

    from ddpm_ood_amd import _lib, synthetic
    from ddpm_ood_amd.data import get_data_loader
    from ddpm_ood_amd.trainer import Reconstruct

    lib = _lib.load()
    run_root = Path(tempfile.mkdtemp(prefix=f"ddpm_bench_r{rank}_"))
    args = make_args(run_root, a.batch * world, a.batch)
    sd = synthetic.write_checkpoint(run_root / args.model_name, "small", 1, seed=1)
    out_stream = sys.stdout
    sys.stdout = open(os.devnull, "w")  # the trainer prints like the reference; keep stdout to ONE JSON line
    try:
        rec = Reconstruct(args)
        rec.quiet = True
        loader = get_data_loader(args.validation_ids, batch_size=a.batch, is_grayscale=True, rank=rank, world=world)
        loader.images = loader.images.to(rec.device)  # inputs resident in HBM before timing starts

        def step(profile=False):
            rec.profile_first_steps = profile
            rows = rec.get_scores(loader, "val", SKIP)
            rec.profile_first_steps = False
            return rows

        log(f"setup done (model on device, {a.batch} images resident)")
        for i in range(a.warmup):
            step()
            log(f"warmup step {i} done")
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(a.steps):
            rows = step(profile=(i == a.steps - 1))
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        log(f"timed region done: {a.steps} step(s) in {dt:.2f} s")
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device=rec.device)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
    finally:
        sys.stdout = out_stream

    n_t = len({r["t"] for r in rows})
    recon_per_step = a.batch * world * n_t
    value = recon_per_step * a.steps / dt

    buf = ctypes.create_string_buffer(1 << 16)
    n = lib.ddpm_prof_report(buf, len(buf))
    prof = json.loads(buf.value.decode()) if n > 0 else {}
    roofline = roofline_of(prof)

    line = {
        "metric": "reconstructions/sec (whole node), FashionMNIST 32x32", "value": round(value, 3),
        "unit": "reconstructions/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(dt / a.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: FashionMNIST-shaped 32x32x1, small UNet (17.7M params, random "
                               "init), 100 PLMS timesteps, inference_skip_factor=4 (25 t-starts, 1250 UNet forwards "
                               "per image)",
                   "images_per_gpu_per_step": a.batch, "reconstructions_per_step": recon_per_step,
                   "unet_forwards_per_image": rec.last_stats["unet_forwards"] // a.batch, "sharding": f"images x{world}"},
        "roofline": roofline,
        "kernels": {k: {"launches": v["launches"], "ms": round(v["ms"], 3),
                        "TFLOPs": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2) if v["ms"] > 0 else None,
                        "GBps": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1) if v["ms"] > 0 else None}
                    for k, v in prof.items()},
    }
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(sd)
        log(f"cpu baseline done: {line['cpu_baseline']}")
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    shutil.rmtree(run_root, ignore_errors=True)
    if rank == 0:
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
