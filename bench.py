"""bench.py -- reconstructions/sec of the multi-t DDPM reconstruction hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config cfg1|cfg2|cfg3|cfg4|cfg5] [--scaling weak|strong]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment launches its N ranks itself (it re-executes
this file through torch.distributed.run on 127.0.0.1 and relays the one JSON line), so both command lines work.

Default workload (BASELINE.json configs[1], the configuration the metric is quoted on): FashionMNIST-shaped
32x32x1 synthetic images, `small` UNet with seeded random weights, 100 PLMS timesteps, inference_skip_factor=4
-> 25 t-starts, 1 250 UNet forwards per image, batch 1 024 per GPU (sized for the 288 GB of HBM: the persistent
convolution kernels amortise their per-launch and per-item costs over more items -- measured 617 / 645 / 663
reconstructions/s at batch 256 / 512 / 1 024; --batch 256 is the reference's default batch size).  One "step" = one
batch of 1 024 images per rank through the whole hot path (noise, add_noise, every PLMS trajectory, clamp + MSE,
LPIPS, score gather) = 25 600 reconstructions per rank, inputs resident in HBM when timing starts.

--scaling strong (default for N > 1: north_star's target is ">= 6x STRONG scaling 1 -> 8 GPUs") a fixed --images set
                 (default: ONE batch of the config, 1 024 images for cfg2 / cfg3 -- exactly the N = 1 workload) is split
                 round-robin over the N ranks (each rank's batch is its share); a step is one pass over the whole set,
                 so the work per step does not grow with N and the N = 1 line is the same under either mode.
--scaling weak   (default for N = 1) every rank gets its own 1 024-image shard of a 1 024*N-image set per step.
The only collective is the per-step all_gather of the dense score tensor (RCCL); the line carries `rccl_world_size`
and the per-rank device list gathered over that group.

--config selects the other BASELINE configurations (same metric, their own `roofline`):
    cfg1  32x32x1 `small`, first_n = 16, k = 64 (t in {10, 650}): the latency-bound small-batch regime of configs[0]
    cfg3  32x32x3 `small`                       (batch 1 024, k = 4)
    cfg4  64x64x3 `big` attention-heavy UNet    (batch 16,  k = 2: 50 t-starts, 2 550 forwards per image)
    cfg5  128^3 volumes, README VQ-VAE (4 x stride 2, 256 ch, 2 048 codes) -> [128, 8, 8, 8] latents -> 3-D
          `small` UNet -> re-quantise + decode -> 2.5-D LPIPS  (batch 64, k = 4)

One JSON line on rank 0.
`dtype`     "f32 (split-f16 MFMA products, fp32 accumulate)": storage, transforms, reductions and accumulation are fp32;
            the MFMA products of the 3x3 / 1x1 convolutions and of attention are rebuilt from exact f16 partial products
            (`arithmetic` spells it out).  `value_fp32_products` (N = 1, cfg2) is the same workload with every split-f16
            family switched to its fp32-MFMA form (ddpm_set_split_f16(0): bit-exact fp32 products), one timed step after
            the main timed region -- so the line carries both arithmetics.
`roofline`  the kernel class with the most time in the sampled launches (first UNet step -- and, for the LDM, the decode
            -- of each t-start of the LAST timed step, hipEvent-bracketed on the launch stream by the library).
            `achieved` = MFMA FLOPs actually EXECUTED per launch / launch time.  Pricing per class (MFMA_KERNELS below):
            split-f16 kernels against the 2 500 TFLOP/s dense f16 MFMA peak with 4 (F(4x4), Downsample) or 3 (attention)
            executed f16 products per fp32 product; fp32-MFMA kernels against the 157.3 TFLOP/s dense f32 MFMA peak; the
            DMA-fed 1x1 against 8 TB/s of HBM (algorithmic bytes / time).  Winograd kernels execute 36/144 (F(4x4)), 16/36
            (F(2x2)) or 9/36 (F(2x2) upsample form) of the direct convolution's multiplies: the direct-conv-equivalent
            rate is reported separately as `algorithmic_equiv_tflops` and is NOT a roofline fraction.  `traffic` = HBM
            bytes per launch from a separate builder-side `rocprofv3 --pmc` pass at the same batch (profiles/
            pmc_traffic.json; counters cannot be collected inside this process), null if there is none for that batch.
`rooflines` every MFMA kernel class the same way (cfg4: the attention kernel; cfg5: the 3-D convolutions).
`cpu_baseline`  the CPU oracle timed on this box's host cores on a bounded sample of the same workload (rank 0, N = 1,
            default config only).
`value_batch256` (N = 1, cfg2 only): the same workload at the reference's default batch of 256 images
            (/root/reference/reconstruct.py:91), two timed steps after the main timed region.
`per_rank`  (N > 1) every rank's images, start-up seconds, own seconds inside the timed region and gather milliseconds per step.
`value_dataset_scale`  (N > 1, strong scaling, cfg2 / cfg3) the same split over a 10 000-image set (the reference's test-split
            scale, get_train_and_val_dataloader.py:21-31), one timed pass: a rank then runs full 1 024-image batches instead of the
            1 024 / N images the default --images leaves it.
`numeric_guard`  batches the trainer had to run again on fp32 products / batches whose scores stayed non-finite (0 / 0 on
            the synthetic workload; include/ddpm_ood_hip.h, "Numeric guard").
"""

import argparse
import ctypes
import json
import os
import shutil
import socket
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
HBM_PEAK_GBPS = 8000.0
F16_MFMA_PEAK_TFLOPS = 2500.0  # dense f16 / bf16 MFMA (same guide)
ARITHMETIC = ("fp32 storage, fp32 accumulation and fp32 transforms / reductions everywhere; the MFMA products of the "
              "Winograd F(4x4) 3x3 convolutions at 32x32 / 16x16 (hi/lo operands, all four exact f16 partial products), of the 1x1 "
              "convolutions and of attention (three of the four partial products: 22 mantissa bits per product) are split-f16 on "
              "v_mfma_f32_32x32x16_f16 with fp32 accumulate; DDPM_WINO44_F16X3=0 / DDPM_CONV1X1_F16X3=0 / DDPM_ATTN_F16X3=0 restore "
              "bit-exact fp32 MFMA products; all other kernels plain fp32")
SCHED = dict(beta_schedule="scaled_linear_beta", beta_start=0.0015, beta_end=0.0195)

VQ_README = dict(spatial_dims=3, in_channels=1, out_channels=1, num_channels=(256, 256, 256, 256), num_res_layers=3,
                 num_res_channels=(256, 256, 256, 256), downsample_parameters=((2, 4, 1, 1),) * 4,
                 upsample_parameters=((2, 4, 1, 1, 0),) * 4, num_embeddings=2048, embedding_dim=128)

CONFIGS = {
    "cfg1": dict(model_type="small", channels=1, size=32, spatial=2, skip=64, batch=16, metric_tag="FashionMNIST 32x32, first_n=16",
                 workload="BASELINE configs[0]: FashionMNIST-shaped 32x32x1, small UNet, 100 PLMS timesteps, "
                          "inference_skip_factor=64 (2 t-starts: 10 and 650, 68 UNet forwards per image), first_n=16"),
    "cfg2": dict(model_type="small", channels=1, size=32, spatial=2, skip=4, batch=1024, metric_tag="FashionMNIST 32x32",
                 workload="BASELINE configs[1]: FashionMNIST-shaped 32x32x1, small UNet (17.7M params, random init), "
                          "100 PLMS timesteps, inference_skip_factor=4 (25 t-starts, 1250 UNet forwards per image)"),
    "cfg3": dict(model_type="small", channels=3, size=32, spatial=2, skip=4, batch=1024, metric_tag="CIFAR10 32x32x3",
                 workload="BASELINE configs[2]: CIFAR10-shaped 32x32x3, small UNet, 100 PLMS timesteps, "
                          "inference_skip_factor=4 (25 t-starts, 1250 UNet forwards per image)"),
    "cfg4": dict(model_type="big", channels=3, size=64, spatial=2, skip=2, batch=16, metric_tag="CelebA 64x64x3 big UNet",
                 workload="BASELINE configs[3]: CelebA-shaped 64x64x3, big attention-heavy UNet (172.6M params, "
                          "attention over 4096/1024/256 tokens), 100 PLMS timesteps, inference_skip_factor=2 "
                          "(50 t-starts, 2550 UNet forwards per image)"),
    "cfg5": dict(model_type="small", channels=1, size=128, spatial=3, skip=4, batch=64, vqvae=VQ_README,
                 metric_tag="Decathlon-shaped 128^3 LDM",
                 workload="BASELINE configs[4]: 128^3 volumes, README VQ-VAE (4 stride-2 levels, 256 ch, 2048 codes x "
                          "128) -> [128,8,8,8] latents, small 3-D UNet (47.5M params), 100 PLMS timesteps, "
                          "inference_skip_factor=4 (25 t-starts: 1250 UNet forwards + 25 re-quantise/decodes + 25 "
                          "2.5-D LPIPS over 128 slices per volume)"),
}


def make_args(run_root, cfg, n_images, batch):
    kind = "blobs3d" if cfg["spatial"] == 3 else "blobs"
    ids = f"synthetic:{kind}:n={n_images}:size={cfg['size']}:channels={cfg['channels']}:seed=0"
    return argparse.Namespace(
        seed=2, output_dir=str(run_root), model_name="bench_synthetic", validation_ids=ids, in_ids=ids,
        out_ids=ids, spatial_dimension=cfg["spatial"], image_size=None, image_roi=None, latent_pad=None,
        vqvae_checkpoint=None, ddpm_checkpoint_epoch=None, prediction_type="epsilon", model_type=cfg["model_type"],
        b_scale=1.0, snr_shift=1, simplex_noise=0, batch_size=batch, augmentation=0, cache_data=1, num_workers=0,
        first_n_val=None, first_n=None, eval_checkpoint=None, drop_last=False, is_grayscale=int(cfg["channels"] == 1),
        run_val=1, run_in=0, run_out=0, num_inference_steps=100, inference_skip_factor=cfg["skip"], **SCHED)


def shard_sizes(scaling: str, world: int, batch: int, images: int):
    """(total images per step, images of each rank) -- pure bookkeeping, tested on CPU (tests/test_dist_gloo.py)."""
    if scaling == "weak":
        return batch * world, [batch] * world
    from ddpm_ood_amd.data import partition

    return images, [len(partition(images, r, world)) for r in range(world)]


def cpu_baseline_worker():
    """Runs in a fresh subprocess (own OpenMP pool): the oracle on the host cores.
    Sample: 16 images x t in {10, 490, 970} (inference_skip_factor=48) = 48 reconstructions,
    150 UNet forwards per image -- the same mean of 50 forwards per reconstruction as the timed
    workload; 15-25 s of CPU work on the GPU boxes' hosts (round 5 timed half of it: the figure moved
    1.9-3.3 box to box on 8-13 s samples).  kind "port": the reference's own dependencies cannot be installed."""
    import oracle
    from ddpm_ood_amd.data import get_data_loader
    from ddpm_ood_amd.synthetic import random_state_dict
    from ddpm_ood_amd.trainer import MODEL_CONFIGS, batch_noise

    model = oracle.DiffusionModelUNet(2, 1, 1, **MODEL_CONFIGS["small"]).eval()
    model.load_state_dict(random_state_dict("small", 1, seed=1))
    pl = oracle.PerceptualLoss(dimensions=2, include_pixel_loss=False, is_fake_3d=False, lpips_normalize=True)
    loader = get_data_loader("synthetic:blobs:n=16:seed=0", batch_size=16, is_grayscale=True)
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
                      "sample": "16 images x t_start in {10, 490, 970}: 48 reconstructions, 2400 UNet "
                                "image-forwards (mean 50 per reconstruction as in the timed workload), CPU fp32 "
                                "oracle incl. LPIPS + MSE"}))


def cpu_baseline():
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


# MFMA kernel classes by profiler-key prefix -> (description, executed / algorithmic MFMA FLOPs).
# Longest prefix wins.  The library counts `flops` as the ALGORITHMIC (direct-form) work of the op.
MFMA_KERNELS = [
    # split-f16 F(4x4): 36 of 144 multiplies, each as FOUR exact f16 partial products (two K = 16 MFMAs per 8 channels) ->
    # executed f16 MFMA FLOPs = 4 x 36 / 144 = 1.0 x the direct convolution's, priced against the dense f16 peak
    ("conv3x3_wino44h", "conv_wino44r_kernel (conv_wino44r.hip; host side conv_wino44h.hip): 3x3 conv "
                        "as Winograd F(4x4,3x3), position GEMMs on v_mfma_f32_32x32x16_f16 with split-f16 operands (hi + lo, four exact "
                        "partial products per fp32 product, fp32 accumulate), fp32 transforms, GN+SiLU prologue, persistent",
     ("f16", 4.0 * 36.0 / 144.0)),
    ("conv3x3_wino44", "conv_wino44_kernel: 3x3 conv as Winograd F(4x4,3x3) (36 of 144 multiplies), persistent, GN+SiLU prologue, fp32 MFMA", 36.0 / 144.0),
    ("conv3x3_wino_up", "conv_wino_up_kernel: nearest-x2 + 3x3 conv in the Winograd domain (9 of 16 positions), fp32 MFMA", 9.0 / 36.0),
    ("conv3x3_wino", "conv_wino_kernel: 3x3 conv as Winograd F(2x2,3x3), persistent, GN+SiLU prologue, fp32 MFMA", 16.0 / 36.0),
    ("conv3x3_mfma_up_folded", "conv_mfma_kernel<4>: nearest-x2 + 3x3 conv folded into four 2x2-tap convs, fp32 MFMA", 16.0 / 36.0),
    ("conv3d_wino44h", "conv_wino44r_kernel, 3-D: Winograd F(4x4,3x3) per depth tap with split-f16 position GEMMs on "
                       "v_mfma_f32_32x32x16_f16 (four exact partial products per fp32 product, fp32 accumulate), taps accumulated in "
                       "the transform domain", ("f16", 4.0 * 36.0 / 144.0)),
    ("conv3d_wino44", "conv_wino44_kernel, 3-D: Winograd F(4x4,3x3) per depth tap (36 of 144 multiplies), taps accumulated in the transform domain, fp32 MFMA", 36.0 / 144.0),
    ("conv3d_wino", "conv_wino_kernel, 3-D: Winograd F(2x2,3x3) per depth tap, taps accumulated in the transform domain, fp32 MFMA", 16.0 / 36.0),
    ("conv3d_", "conv_mfma_kernel: 3-D convolution (depth taps merged into one chunk stream), fp32 MFMA", 1.0),
    # split-f16 direct convolution: four exact f16 partial products per fp32 product (two K = 16 MFMAs per 8 channels and tap)
    ("conv3x3_s2h", "conv_s2h_kernel: Downsample 3x3 stride-2 conv, direct form on v_mfma_f32_32x32x16_f16 with split-f16 operands "
                    "(hi + lo, four exact partial products per fp32 product, fp32 accumulate)", ("f16", 4.0)),
    # split-f16 direct convolution of small launches: three f16 partial products per fp32 product x 10 / 9 taps (one zero tap pads
    # the fifth two-tap K-step); launch time includes the reduce pass over the channel slices
    ("conv3x3_d3s", "conv_d3s_kernel: one-shot direct 3x3 conv of launches far smaller than the chip (8x8 / 16x16 levels of a few "
                    "images) on v_mfma_f32_32x32x16_f16 with split-f16 operands (three partial products per fp32 product, fp32 "
                    "accumulate), channel slices of 32 reduced in a fixed order", ("f16", 3.0 * 10.0 / 9.0)),
    ("conv3x3_mfma", "conv_mfma_kernel<9>: direct 3x3 conv (stride 2, ragged extents, 3-D depth-tap launches), fp32 MFMA", 1.0),
    # split-f16: three v_mfma_f32_32x32x16_f16 per fp32 product, 16x the f32 MFMA rate -> the layer is HBM-bound
    ("conv1x1_dma", "conv1x1_dma_kernel: LDS-DMA-fed 1x1 conv (skip connections; with GroupNorm prologue: q / k / v), "
                    "fp32 products from three f16 MFMAs (hi / lo split, fp32 accumulate)", "hbm"),
    ("conv1x1_mfma", "conv_mfma_kernel<1>: 1x1 conv / Linear (fused QKV, time MLP), fp32 MFMA", 1.0),
    ("lpips_conv_mfma", "lpips_conv_mfma_kernel: AlexNet 5x5 layer of the 2.5-D LPIPS as an implicit GEMM, fp32 MFMA", 1.0),
    ("attention_fa", "attention_fa_kernel: register-resident flash attention on v_mfma_f32_16x16x32_f16 (q tile, scores and output in "
                     "registers; K / V f16 planes from a pre-pass through LDS-DMA; fp32 products from three f16 MFMAs, fp32 "
                     "accumulate); launch time includes the pre-pass", ("f16", 3.0)),
    # split-f16: executed MFMA work = 3 f16 MFMAs per fp32 product, priced against the dense f16 peak
    ("attention", "attention_kernel: QK^T, online softmax, AV (+ residual); fp32 products from three f16 MFMAs "
                  "(hi / lo split, fp32 accumulate)", ("f16", 3.0)),
]


def mfma_class(key):
    for prefix, label, ratio in MFMA_KERNELS:
        if key.startswith(prefix):
            return prefix, label, ratio
    return None


def rooflines_of(prof, batch=None):
    """Aggregate the in-situ profile by MFMA kernel class.  Per class: executed-MFMA TFLOP/s against the f32 MFMA
    peak (`frac`), the direct-conv-equivalent rate where the kernel executes fewer multiplies than the op it
    replaces, and the algorithmic HBM rate."""
    agg = {}
    for key, v in prof.items():
        cls = mfma_class(key)
        if cls is None or v["ms"] <= 0:
            continue
        prefix, label, ratio = cls
        a = agg.setdefault(prefix, {"label": label, "ratio": ratio, "ms": 0.0, "flops": 0.0, "bytes": 0.0, "launches": 0})
        a["ms"] += v["ms"]; a["flops"] += v["flops"]; a["bytes"] += v["bytes"]; a["launches"] += v["launches"]
    out = {}
    pmc_path = ROOT / "profiles" / "pmc_traffic.json"  # builder-side rocprofv3 --pmc pass (tools/pmc_collect.sh)
    pmc = json.load(open(pmc_path)) if pmc_path.exists() else {}
    for prefix, a in agg.items():
        alg = a["flops"] / (a["ms"] * 1e-3) / 1e12
        if a["ratio"] == "hbm":  # priced against HBM: algorithmic bytes (in + residual + out + weights) per second
            gbps = a["bytes"] / (a["ms"] * 1e-3) / 1e9
            r = {"bound": "hbm", "kernel": a["label"], "profile_key": prefix, "achieved": round(gbps, 1),
                 "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(gbps / HBM_PEAK_GBPS, 4),
                 "algorithmic_equiv_tflops": round(alg, 2), "algorithmic_GBps": round(gbps, 1),
                 "f16_mfma_tflops_executed": round(3 * alg, 2), "f16_mfma_peak_tflops": F16_MFMA_PEAK_TFLOPS,
                 "launches_timed": a["launches"], "avg_launch_ms": round(a["ms"] / a["launches"], 4),
                 "ms_in_sample": round(a["ms"], 3), "bytes_per_launch": a["bytes"] / a["launches"],
                 "traffic": None}
        elif isinstance(a["ratio"], tuple):  # ("f16", executed f16 MFMA FLOPs / algorithmic FLOPs)
            ex = a["ratio"][1] * alg
            r = {"bound": "mfma", "kernel": a["label"], "profile_key": prefix, "achieved": round(ex, 2),
                 "peak": F16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ex / F16_MFMA_PEAK_TFLOPS, 4),
                 "executed_over_algorithmic_flops": a["ratio"][1], "algorithmic_equiv_tflops": round(alg, 2),
                 "algorithmic_over_f32_mfma_peak": round(alg / F32_MFMA_PEAK_TFLOPS, 4),
                 "launches_timed": a["launches"], "avg_launch_ms": round(a["ms"] / a["launches"], 4),
                 "ms_in_sample": round(a["ms"], 3), "flops_per_launch": a["flops"] / a["launches"],
                 "algorithmic_GBps": round(a["bytes"] / (a["ms"] * 1e-3) / 1e9, 1), "traffic": None}
        else:
            ex = alg * a["ratio"]
            r = {"bound": "mfma", "kernel": a["label"], "profile_key": prefix, "achieved": round(ex, 2),
                 "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ex / F32_MFMA_PEAK_TFLOPS, 4),
                 "executed_over_algorithmic_flops": round(a["ratio"], 4),
                 "algorithmic_equiv_tflops": round(alg, 2), "launches_timed": a["launches"],
                 "avg_launch_ms": round(a["ms"] / a["launches"], 4), "ms_in_sample": round(a["ms"], 3),
                 "flops_per_launch": a["flops"] / a["launches"],
                 "algorithmic_GBps": round(a["bytes"] / (a["ms"] * 1e-3) / 1e9, 1), "traffic": None}
        # HBM bytes per launch from the PMC counters: measured builder-side (separate rocprofv3 --pmc passes, the guide's
        # gfx950 correction), per batch size; a figure taken at another batch than the timed one is NOT `traffic`
        at_batch = pmc.get("by_batch", {}).get(str(batch), {})
        for k in (prefix + "_gn_silu", prefix):
            if k + "_bytes_per_launch" in at_batch:
                r["traffic"] = at_batch[k + "_bytes_per_launch"]
                r["traffic_source"] = (f"profiles/pmc_traffic.json by_batch[{batch}] (builder-side rocprofv3 --pmc pass at "
                                       "the timed batch, not this run)")
                break
            if k + "_bytes_per_launch" in pmc:
                r["traffic_at_other_batch"] = {"batch": pmc.get("batch", 256), "bytes_per_launch": pmc[k + "_bytes_per_launch"]}
                break
        out[prefix] = r
    return out


_T0 = time.perf_counter()


def log(msg):
    """progress on stderr (stdout carries exactly one JSON line)"""
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench +{time.perf_counter() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


def write_vqvae(run_root, cfg):
    """Random-init VQ-VAE of the README shape (product class, CPU init), codebook spread so codes are used."""
    from ddpm_ood_amd.vqvae import VQVAE

    torch.manual_seed(3)
    vq = VQVAE(**cfg).eval()
    with torch.no_grad():
        vq.quantizer.quantizer.embedding.weight.mul_(3.0)
    d = run_root / "vqvae"
    d.mkdir(parents=True, exist_ok=True)
    torch.save({"model_state_dict": vq.state_dict()}, d / "checkpoint.pth")
    json.dump(cfg, open(d / "vqvae_config.json", "w"))
    return str(d / "checkpoint.pth")


def self_launch(n: int):
    """`python bench.py --gpus N` without a launcher: start the N ranks here (one process per GPU, rendezvous on
    127.0.0.1 at a free port) and pass their output through; the exit code is the launcher's."""
    import socket
    import subprocess

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(ROOT / "bench.py")] + sys.argv[1:]
    log(f"no WORLD_SIZE in the environment: launching {n} ranks: {' '.join(cmd)}")
    raise SystemExit(subprocess.run(cmd).returncode)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="cfg2")
    ap.add_argument("--batch", type=int, default=None, help="images per GPU per batch (default: the config's)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default=None,
                    help="default: strong for --gpus > 1 (fixed image set split over the ranks), weak for 1")
    ap.add_argument("--images", type=int, default=None,
                    help="--scaling strong: size of the fixed image set (default: one batch of the config)")
    ap.add_argument("--no-batch256", action="store_true", help="skip the extra batch-256 measurement (cfg2, N = 1)")
    ap.add_argument("--no-fp32-products", action="store_true",
                    help="skip the extra step on the fp32-MFMA kernels (`value_fp32_products`; cfg2, N = 1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dataset-scale", action="store_true",
                    help="N > 1, strong scaling: skip the extra pass over a --dataset-images set (`value_dataset_scale`)")
    ap.add_argument("--dataset-images", type=int, default=10000,
                    help="N > 1, strong scaling: size of the dataset-scale image set (default: the FashionMNIST / CIFAR10 test split)")
    ap.add_argument("--cpu-baseline-worker", action="store_true", help=argparse.SUPPRESS)
    a = ap.parse_args()
    if a.cpu_baseline_worker:
        return cpu_baseline_worker()
    cfg = CONFIGS[a.config]
    batch = a.batch or cfg["batch"]
    if a.scaling is None:
        a.scaling = "strong" if a.gpus > 1 else "weak"

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(a.gpus)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # test hooks (the trainer's): DDPM_DIST_BACKEND=gloo + DDPM_DIST_SHARED_DEVICE=1 run the N-rank bench on ONE GPU
    # (RCCL refuses two ranks per device) -- everything but the transport of the collectives is the N-rank path
    backend = os.environ.get("DDPM_DIST_BACKEND", "nccl")
    if os.environ.get("DDPM_DIST_SHARED_DEVICE", "0") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend, init_method="env://")

    from ddpm_ood_amd import _lib, synthetic
    from ddpm_ood_amd.data import get_data_loader
    from ddpm_ood_amd.trainer import Reconstruct

    lib = _lib.load()
    n_images, per_rank = shard_sizes(a.scaling, world, batch, a.images or batch)
    run_root = Path(tempfile.mkdtemp(prefix=f"ddpm_bench_r{rank}_"))
    args = make_args(run_root, cfg, n_images, batch)
    ddpm_channels = cfg["vqvae"]["embedding_dim"] if cfg.get("vqvae") else cfg["channels"]
    sd = synthetic.random_state_dict(cfg["model_type"], ddpm_channels, spatial_dims=cfg["spatial"], seed=1)
    (run_root / args.model_name).mkdir(parents=True)
    torch.save({"epoch": 0, "global_step": 0, "model_state_dict": sd, "optimizer_state_dict": {}, "best_loss": 1000},
               run_root / args.model_name / "checkpoint.pth")
    del sd
    if cfg.get("vqvae"):
        args.vqvae_checkpoint = write_vqvae(run_root, cfg["vqvae"])
    out_stream = sys.stdout
    sys.stdout = open(os.devnull, "w")  # the trainer prints like the reference; keep stdout to ONE JSON line
    try:
        rec = Reconstruct(args)
        rec.quiet = True
        loader = get_data_loader(args.validation_ids, batch_size=batch, is_grayscale=bool(args.is_grayscale),
                                 spatial_dimension=cfg["spatial"], rank=rank, world=world)
        assert len(loader.names) == per_rank[rank]
        loader.images = loader.images.to(rec.device)  # inputs resident in HBM before timing starts

        def step(profile=False):
            rec.profile_first_steps = profile
            rows = rec.get_scores(loader, "val", cfg["skip"])
            rec.profile_first_steps = False
            return rows

        t_setup = time.perf_counter()
        log(f"setup done ({a.config}: model on device, {per_rank[rank]} images resident on rank 0)")
        # per-rank start-up (weight upload + packing into the engine blob, workspace allocation, code-object loading, LPIPS
        # weight packing) happens on first use: do it here, outside the timed region, whatever --warmup says -- the first
        # t-start only (2 UNet forwards per image)
        rec.max_t_start = 10
        step()
        rec.max_t_start = None
        torch.cuda.synchronize()
        startup_s = time.perf_counter() - _T0  # process start -> first (2-forward) pass done, this rank
        first_pass_s = time.perf_counter() - t_setup
        log("start-up done (weights packed, workspace allocated, kernels loaded: t_start = 10 only, untimed)")
        for i in range(a.warmup):
            step()
            log(f"warmup step {i} done")
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        gather_ms = []
        for i in range(a.steps):
            rows = step(profile=(i == a.steps - 1))
            gather_ms.append(rec.last_stats.get("gather_ms", 0.0))
        torch.cuda.synchronize()
        own_s = time.perf_counter() - t0  # this rank's own work (its last gather waited for the slowest rank)
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        log(f"timed region done: {a.steps} step(s) in {dt:.2f} s")
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device=rec.device if backend == "nccl" else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
    finally:
        sys.stdout = out_stream

    main_stats = dict(rec.last_stats)  # of the timed workload: the extra legs below overwrite rec.last_stats
    forwards_per_image = main_stats["unet_forwards"] // max(per_rank[rank], 1)
    if a.config in ("cfg2", "cfg3"):
        assert forwards_per_image == 1250, forwards_per_image  # 25 t-starts, sum over t of (t / 10 + 1): nothing skipped
    devices = [f"rank {rank}: {socket.gethostname()} cuda:{local_rank} {torch.cuda.get_device_name(local_rank)}"]
    if world > 1:  # what the RCCL group actually spans: every rank reports its device through the group itself
        gathered = [None] * world
        dist.all_gather_object(gathered, devices[0])
        devices = gathered

    # the same workload at the reference's default batch (reconstruct.py:91), two timed steps
    b256 = None
    if world == 1 and a.config == "cfg2" and batch != 256 and not a.no_batch256:
        sys.stdout = open(os.devnull, "w")
        try:
            l256 = get_data_loader(f"synthetic:blobs:n=256:size={cfg['size']}:channels={cfg['channels']}:seed=0",
                                   batch_size=256, is_grayscale=bool(args.is_grayscale), spatial_dimension=cfg["spatial"])
            l256.images = l256.images.to(rec.device)
            rec.get_scores(l256, "val", cfg["skip"])  # warm-up (workspace of the new shape)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(2):
                r256 = rec.get_scores(l256, "val", cfg["skip"])
            torch.cuda.synchronize()
            dt256 = time.perf_counter() - t0
        finally:
            sys.stdout = out_stream
        b256 = {"value": round(2 * len(r256) / dt256, 3), "steps": 2, "ms_per_step": round(dt256 / 2 * 1e3, 2)}
        log(f"batch-256 measurement done: {b256}")

    # the same workload with bit-exact fp32 MFMA products everywhere (the run-time switch of the numeric guard): one timed step
    fp32p = None
    if world == 1 and a.config == "cfg2" and not a.no_fp32_products:
        from ddpm_ood_amd import _lib as L

        sys.stdout = open(os.devnull, "w")
        prev_split = L.set_split_f16(False)
        try:
            rec.get_scores(loader, "val", 64)  # warm-up of the other kernels (2 t-starts)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r32 = rec.get_scores(loader, "val", cfg["skip"])
            torch.cuda.synchronize()
            dt32 = time.perf_counter() - t0
        finally:
            L.set_split_f16(prev_split)
            sys.stdout = out_stream
        fp32p = {"value": round(len(r32) / dt32, 3), "steps": 1, "ms_per_step": round(dt32 * 1e3, 2),
                 "arithmetic": "every MFMA product bit-exact fp32 (ddpm_set_split_f16(0) = DDPM_WINO44_F16X3=0 "
                               "DDPM_CONV1X1_F16X3=0 DDPM_ATTN_F16X3=0 DDPM_DOWN_S2H=0): conv_wino44_kernel / "
                               "conv_wino_up_kernel / conv_mfma_kernel / f32 attention loops"}
        log(f"fp32-products measurement done: {fp32p}")

    # N > 1: what each rank spent where (the driver computes the efficiency; these make its curve diagnosable)
    per_rank_timing = None
    if world > 1:
        mine = {"rank": rank, "images": per_rank[rank], "startup_s": round(startup_s, 2), "first_pass_s": round(first_pass_s, 2),
                "timed_s": round(own_s, 3), "gather_ms_per_step": [round(g, 2) for g in gather_ms]}
        per_rank_timing = [None] * world
        dist.all_gather_object(per_rank_timing, mine)

    # N > 1, strong scaling, cfg2 / cfg3: the same split over an image set of the reference's DATASET scale (FashionMNIST / CIFAR10
    # test split = 10 000 images, /root/reference/src/data/get_train_and_val_dataloader.py:21-31 shards it over the ranks): one
    # timed pass.  The default --images (one batch of 1 024) leaves a rank 1 024 / N images -- the sub-chip batches of the kernels'
    # worst case -- which no run over a real dataset would see; both numbers are on the line.
    ds_scale = None
    if world > 1 and a.scaling == "strong" and a.config in ("cfg2", "cfg3") and not a.no_dataset_scale:
        sys.stdout = open(os.devnull, "w")
        try:
            n_ds = a.dataset_images
            ids_ds = f"synthetic:blobs:n={n_ds}:size={cfg['size']}:channels={cfg['channels']}:seed=0"
            l_ds = get_data_loader(ids_ds, batch_size=batch, is_grayscale=bool(args.is_grayscale), spatial_dimension=cfg["spatial"],
                                   rank=rank, world=world)
            l_ds.images = l_ds.images.to(rec.device)
            dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r_ds = rec.get_scores(l_ds, "val", cfg["skip"])
            torch.cuda.synchronize()
            dist.barrier()
            dt_ds = time.perf_counter() - t0
            tt = torch.tensor([dt_ds], dtype=torch.float64, device=rec.device if backend == "nccl" else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt_ds = float(tt.item())
        finally:
            sys.stdout = out_stream
        ds_scale = {"images": n_ds, "images_per_rank": len(l_ds.names), "reconstructions": len(r_ds), "seconds": round(dt_ds, 2),
                    "value": round(len(r_ds) / dt_ds, 3), "unit": "reconstructions/s", "steps": 1,
                    "note": "same strong-scaling split over an image set of the reference's dataset scale; one timed pass, max over ranks"}
        log(f"dataset-scale pass done: {ds_scale}")

    n_t = len({r["t"] for r in rows})
    assert len(rows) == n_images * n_t, (len(rows), n_images, n_t)  # every rank's scores came back through the gather
    recon_per_step = n_images * n_t
    value = recon_per_step * a.steps / dt

    buf = ctypes.create_string_buffer(1 << 16)
    n = lib.ddpm_prof_report(buf, len(buf))
    prof = json.loads(buf.value.decode()) if n > 0 else {}
    rooflines = rooflines_of(prof, batch)
    dominant = max(rooflines.values(), key=lambda r: r["ms_in_sample"]) if rooflines else None

    line = {
        "metric": f"reconstructions/sec (whole node), {cfg['metric_tag']}", "value": round(value, 3),
        "unit": "reconstructions/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(dt / a.steps * 1e3, 2), "higher_is_better": True, "scaling": a.scaling,
        "vs_baseline": None, "dtype": "f32 (split-f16 MFMA products, fp32 accumulate)", "data": "synthetic",
        "arithmetic": ARITHMETIC, "rccl_world_size": dist.get_world_size() if world > 1 else 1,
        **({"dist_backend": backend} if backend != "nccl" else {}), "devices": devices,
        "config": {"workload": cfg["workload"], "name": a.config, "images_per_gpu_per_batch": batch,
                   "images_per_step": n_images, "reconstructions_per_step": recon_per_step,
                   "unet_forwards_per_image": forwards_per_image,
                   "sharding": f"images x{world} ({a.scaling})",
                   "lpips_weights": "pretrained" if main_stats.get("lpips_pretrained") else "seeded random",
                   **({"lpips_2p5d_views_computed": "all 3 (DDPM_LPIPS_ALL_VIEWS=1)"
                       if os.environ.get("DDPM_LPIPS_ALL_VIEWS", "0") == "1" else
                       "last of 3 (the reference's loop overwrites the other two: same scores)"}
                      if cfg.get("spatial") == 3 else {})},
        "roofline": dominant,
        "rooflines": {k: {kk: r[kk] for kk in ("bound", "achieved", "unit", "frac", "algorithmic_equiv_tflops",
                                                "avg_launch_ms", "launches_timed", "ms_in_sample", "algorithmic_GBps")}
                      for k, r in rooflines.items()},
        "kernels": {k: {"launches": v["launches"], "ms": round(v["ms"], 3),
                        "TFLOPs": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2) if v["ms"] > 0 else None,
                        "GBps": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1) if v["ms"] > 0 else None}
                    for k, v in prof.items()},
    }
    line["numeric_guard"] = {k: main_stats.get(k, 0) for k in ("batches_rerun_fp32", "batches_nonfinite")}
    if per_rank_timing:
        line["per_rank"] = per_rank_timing
    if ds_scale:
        line["value_dataset_scale"] = ds_scale["value"]
        line["dataset_scale"] = ds_scale
    if b256:
        line["value_batch256"] = b256["value"]
        line["batch256"] = b256
    if fp32p:
        line["value_fp32_products"] = fp32p["value"]
        line["fp32_products"] = fp32p
    if rank == 0 and world == 1 and not a.no_cpu_baseline and a.config == "cfg2":
        line["cpu_baseline"] = cpu_baseline()
        log(f"cpu baseline done: {line['cpu_baseline']}")
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    shutil.rmtree(run_root, ignore_errors=True)
    if rank == 0:
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
