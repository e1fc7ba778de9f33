"""-m gpu: oracle parity AT THE BENCHMARKED DISPATCH.

Kernel selection depends on the launch size (F(4x4) Winograd only when a launch fills the chip, the DMA-fed split-f16
1x1 from 128 workgroups, the eight-wave attention, the XCD-aware orders), so the small batches of the other
whole-network tests exercise OTHER kernels than the ones bench.py times.  Here the `small` UNet runs at the
reference's default batch (256, /root/reference/reconstruct.py:91) and at the bench's batch (1 024) against the CPU
oracle, the in-situ profiler (`ddpm_prof_report`) has to show that the benchmarked kernel classes actually ran, a
k = 64 trajectory at B = 128 is held to the north-star bar as an ABSOLUTE bound (|dZ| <= 1e-4), and the same is done
on TRAINED weights (the Winograd error depends on the activation range; `random_state_dict` alone does not cover it).
Reference loop: /root/reference/src/trainers/reconstruct.py:128-204.
"""

import argparse
import ctypes
import json
import math

import pytest
import torch

from parity_util import assert_rows_close, assert_z_close, hip_scores, make_args, oracle_scores, write_checkpoint

pytestmark = pytest.mark.gpu

# the kernel classes of a chip-filling `small` forward (profiler keys, csrc/*.hip ProfScope names)
BENCH_KEYS = ("conv3x3_wino44h_gn_silu", "conv3x3_wino44h_up", "conv1x1_dma", "conv1x1_dma_gn", "attention", "gn_finalize")


def _profiled(fn):
    from ddpm_ood_amd import _lib

    lib = _lib.load()
    lib.ddpm_prof_enable(1)
    try:
        out = fn()
        torch.cuda.synchronize()
    finally:
        lib.ddpm_prof_enable(0)
    return out, _report()


def _report():
    from ddpm_ood_amd import _lib

    buf = ctypes.create_string_buffer(1 << 18)
    n = _lib.load().ddpm_prof_report(buf, len(buf))
    return json.loads(buf.value.decode()) if n > 0 else {}


def _models(device, sd, channels=1, model_type="small"):
    import oracle
    from ddpm_ood_amd import DiffusionModelUNet
    from ddpm_ood_amd.trainer import MODEL_CONFIGS

    ref = oracle.DiffusionModelUNet(2, channels, channels, **MODEL_CONFIGS[model_type]).eval()
    ref.load_state_dict(sd)
    hip = DiffusionModelUNet(2, channels, channels, **MODEL_CONFIGS[model_type])
    hip.load_state_dict(sd)
    return ref, hip.to(device).eval()


def _forward_pair(device, ref, hip, B, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, 1, 32, 32, generator=g)
    t = torch.randint(0, 1000, (B,), generator=g)
    with torch.no_grad():
        yr = ref(x, timesteps=t)
    yh, prof = _profiled(lambda: hip(x.to(device), timesteps=t.to(device)).cpu())
    return yr, yh, prof


@pytest.mark.parametrize("B", [256, 1024])
def test_small_forward_at_benchmarked_batch_vs_oracle(device, B):
    """One `small` forward at B = 256 (reference default) and B = 1 024 (bench default) against the CPU oracle --
    the launches bench.py times, not the small-batch dispatch -- and the profiler confirms which kernels ran."""
    from ddpm_ood_amd.synthetic import random_state_dict

    ref, hip = _models(device, random_state_dict("small", 1, seed=1))
    yr, yh, prof = _forward_pair(device, ref, hip, B, seed=40 + B)
    err = (yh - yr).abs().max().item()
    scale = yr.abs().max().item()
    print(f"B = {B}: max |eps_hip - eps_oracle| = {err:.3e} (max |eps| = {scale:.3f})")
    assert math.isfinite(err) and err <= 2e-5 * (1 + scale), err
    assert scale > 0.05
    missing = [k for k in BENCH_KEYS if k not in prof]
    assert not missing, (missing, sorted(prof))
    # every 32x32 / 16x16 / 8x8 ResnetBlock convolution went through the split-f16 F(4x4) kernel: 22 launches per forward
    assert prof["conv3x3_wino44h_gn_silu"]["launches"] == 22, prof["conv3x3_wino44h_gn_silu"]
    assert "conv3x3_wino44_gn_silu" not in prof, sorted(prof)
    # ... and so did the two Upsample convolutions (nearest x2 read on the fly by the same kernel)
    assert prof["conv3x3_wino44h_up"]["launches"] == 2 and "conv3x3_wino_up" not in prof, sorted(prof)
    # the two Downsample convolutions: direct stride-2 kernel with split-f16 operands (conv_s2h.hip)
    assert prof["conv3x3_s2h"]["launches"] == 2, sorted(prof)
    # GroupNorm statistics come from the producers' epilogues: 27 finalize launches, no reading GroupNorm kernel, and a
    # per-channel reduction only for the tensors of other producers (conv_in, Downsample, Upsample, attention outputs)
    assert prof["gn_finalize"]["launches"] == 27, prof["gn_finalize"]
    assert "gn_scale_shift" not in prof, sorted(prof)
    # (at B = 256 the 8x8-level convolutions are channel-split launches: their reduce pass emits the statistics)
    assert prof.get("gn_channel_stats", {"launches": 0})["launches"] <= 7, prof.get("gn_channel_stats")
    assert "conv3x3_wino_gn_silu" not in prof and "conv3x3_mfma_gn_silu" not in prof, sorted(prof)


def test_k64_trajectories_at_batch_128_absolute_z(device, tmp_path):
    """BASELINE configs[0]'s t-start list (k = 64: t in {10, 650}, 68 forwards per image) at a chip-filling batch:
    128 val / 128 in / 128 out images, one batch each.  |dZ| <= 1e-4 ABSOLUTE (north_star), AUROC +-1e-3.
    The oracle side of all 384 images (206 TFLOP of CPU convolutions: 5.5 minutes of this suite until round 3) is the committed
    fixture tests/golden/rows_k64_b128.csv; the oracle runs live on the first images of every set and has to reproduce the
    fixture and match the HIP rows (parity_util.live_oracle_pins_fixture)."""
    from parity_util import golden_rows, live_oracle_pins_fixture
    from ddpm_ood_amd import synthetic
    from ddpm_ood_amd.trainer import Reconstruct

    spec, rows_o = golden_rows("k64_b128")
    sets = spec["sets"]
    args = make_args(tmp_path, inference_skip_factor=spec["skip"], batch_size=spec["batch"], validation_ids=sets["val"],
                     in_ids=sets["in"])
    sd = synthetic.random_state_dict("small", 1, seed=1)
    write_checkpoint(tmp_path, args, sd)
    rec = Reconstruct(args)
    rec.quiet = True
    rows_h = {}
    for name, ids in sets.items():
        rec.profile_first_steps = name == "val"  # hipEvent-bracket the first UNet step of each t-start
        rows_h[name] = hip_scores(args, rec, ids, name)
        rec.profile_first_steps = False
        if name == "val":
            prof = _report()
            missing = [k for k in BENCH_KEYS if k not in prof]
            assert not missing, (missing, sorted(prof))
        assert_rows_close(rows_h[name], rows_o[name], 2e-4, name)
    worst, auc_h, auc_o = assert_z_close(rows_h, rows_o)
    print(f"B = 128, k = 64: max |dZ| = {worst:.2e}, AUROC hip {auc_h:.4f} / oracle {auc_o:.4f}")
    live_oracle_pins_fixture("k64_b128", spec, rows_o, rows_h)


def test_cfg2_25_chained_t_starts_at_batch_128_absolute_z(device, tmp_path):
    """The workload bench.py times (BASELINE configs[1]: inference_skip_factor = 4 -> 25 chained t-starts 10 ... 970, 1 250 UNet
    forwards per image, the t = 970 trajectory 98 steps long on stale PLMS history) AT A CHIP-FILLING DISPATCH: every set is 128
    images in ONE batch, so every forward runs on the split-f16 F(4x4) kernel, the DMA-fed 1x1 and the direct stride-2 kernel
    (profiler-asserted).  The oracle side is the committed rows of the first 16 images of every set
    (tests/golden/rows_cfg2_25t_b128.csv: 60 000 CPU forwards; a per-image result does not depend on the batch it rides in),
    pinned by the oracle run live on one image.  Raw scores <= 2e-4 relative, |dZ| <= 1e-4 ABSOLUTE (16 validation images) for
    both columns, AUROC +-1e-3.  Reference: /root/reference/src/trainers/reconstruct.py:118-120,128-157,166."""
    from parity_util import golden_rows, live_oracle_pins_fixture
    from ddpm_ood_amd import synthetic
    from ddpm_ood_amd.trainer import Reconstruct

    spec, rows_o = golden_rows("cfg2_25t_b128")
    sets = spec["sets"]
    args = make_args(tmp_path, inference_skip_factor=spec["skip"], batch_size=spec["batch"], validation_ids=sets["val"],
                     in_ids=sets["in"])
    write_checkpoint(tmp_path, args, synthetic.random_state_dict("small", 1, seed=1))
    rec = Reconstruct(args)
    rec.quiet = True
    rows_h = {}
    for name, ids in sets.items():
        rec.profile_first_steps = name == "val"
        full = hip_scores(args, rec, ids, name)
        rec.profile_first_steps = False
        if name == "val":
            prof = _report()
            missing = [k for k in BENCH_KEYS + ("conv3x3_s2h",) if k not in prof]
            assert not missing, (missing, sorted(prof))
            # 25 profiled forwards (the first of each t-start): 22 ResnetBlock + 2 Upsample convolutions on the F(4x4) split-f16 kernel
            assert prof["conv3x3_wino44h_gn_silu"]["launches"] == 25 * 22 and prof["conv3x3_wino44h_up"]["launches"] == 25 * 2, prof
            assert "conv3x3_wino44_gn_silu" not in prof and "conv3x3_wino_gn_silu" not in prof and "conv3x3_d3s_gn_silu" not in prof, sorted(prof)
        assert sorted(set(full["t"])) == list(range(10, 1000, 40))
        assert rec.last_stats["unet_forwards"] == 128 * 1250 and len(full) == 128 * 25
        keep = set(rows_o[name]["filename"])
        assert len(keep) == spec["oracle_n"]
        rows_h[name] = full[full["filename"].isin(keep)].reset_index(drop=True)
        worst = assert_rows_close(rows_h[name], rows_o[name], 2e-4, name)
        print(f"cfg2 25-t chain at B = 128, {name}: raw scores max relative error {worst}")
    worst, auc_h, auc_o = assert_z_close(rows_h, rows_o)  # 16 validation images: the ABSOLUTE bound
    assert rows_o["val"]["filename"].nunique() >= 16
    print(f"cfg2 25-t chain at B = 128: max |dZ| = {worst:.2e} (absolute), AUROC hip {auc_h:.4f} / oracle {auc_o:.4f}")
    live_oracle_pins_fixture("cfg2_25t_b128", spec, rows_o, rows_h)


def _cfg4_rec(tmp_path, spec):
    from ddpm_ood_amd import synthetic
    from ddpm_ood_amd.trainer import Reconstruct

    sets = spec["sets"]
    args = make_args(tmp_path, model_type="big", is_grayscale=0, inference_skip_factor=spec["skip"], batch_size=spec["batch"],
                     validation_ids=sets.get("val", sets["in"]), in_ids=sets["in"])
    write_checkpoint(tmp_path, args, synthetic.random_state_dict("big", 3, seed=1))
    rec = Reconstruct(args)
    rec.quiet = True
    rec.t_start_subset = spec["t_start_subset"]
    return args, rec


def _assert_cfg4_dispatch(prof, forwards):
    """The launches bench.py's cfg4 times (batch 16, `big`): per profiled forward 34 ResnetBlock convolutions on the split-f16
    F(4x4) kernel, ten attention blocks over 4 096 / 1 024 tokens on the register-resident kernel and six over 256 tokens on
    the LDS-exchange one, sixteen GroupNorm-ed q / k / v projections on the DMA-fed 1x1, two Downsample convolutions."""
    assert prof["conv3x3_wino44h_gn_silu"]["launches"] == forwards * 34, prof.get("conv3x3_wino44h_gn_silu")
    assert prof["attention_fa"]["launches"] == forwards * 10 and prof["attention"]["launches"] == forwards * 6, (
        prof.get("attention_fa"), prof.get("attention"))
    assert prof["conv1x1_dma_gn"]["launches"] == forwards * 16, prof.get("conv1x1_dma_gn")
    assert prof["conv3x3_s2h"]["launches"] == forwards * 2 and prof["conv1x1_dma"]["launches"] >= forwards, sorted(prof)
    assert "conv3x3_wino44_gn_silu" not in prof and "conv3x3_wino_gn_silu" not in prof and "conv3x3_mfma_gn_silu" not in prof, sorted(prof)


def test_cfg4_long_chains_at_the_benchmarked_batch_absolute_z(device, tmp_path):
    """cfg4 (`big` UNet, 64x64x3, k = 2) on LONG chains AT THE BATCH bench.py TIMES (16): t_start in {10, 250, 490} of the
    chained list = 2 + 26 + 50 forwards per image, each through 16 attention blocks (10 of them over 4 096 / 1 024 tokens on
    the register-resident split-f16 kernel) -- where a 22-bit product would show if it accumulated.  Every set is 16 images in
    ONE batch (profiler-asserted dispatch); the oracle side is the committed rows of all 16 validation images and the first two
    of the other sets (tests/golden/rows_cfg4_b16.csv: 1 560 `big` CPU forwards; a per-image result does not depend on the batch
    it rides in), pinned by the live oracle on one image.  Raw scores <= 2e-4 relative, |dZ| <= 1e-4 ABSOLUTE (16 validation
    images), AUROC +-1e-3.  (Round 5 ran these chains at batch 2 with two validation images: VERDICT r5 weak item 2.)
    Reference: /root/reference/src/trainers/reconstruct.py:128-166."""
    from parity_util import golden_rows, live_oracle_pins_fixture

    spec, rows_o = golden_rows("cfg4_b16")
    args, rec = _cfg4_rec(tmp_path, spec)
    rows_h = {}
    for name, ids in spec["sets"].items():
        rec.profile_first_steps = name == "val"
        full = hip_scores(args, rec, ids, name)
        rec.profile_first_steps = False
        if name == "val":
            _assert_cfg4_dispatch(_report(), forwards=3)  # the first forward of each of the three t-starts
        assert full["filename"].nunique() == 16 and rec.last_stats["unet_forwards"] == 16 * 78
        assert sorted(set(full["t"])) == [10, 250, 490]
        keep = set(rows_o[name]["filename"])
        assert len(keep) == spec["oracle_n"][name]
        rows_h[name] = full[full["filename"].isin(keep)].reset_index(drop=True)
        worst = assert_rows_close(rows_h[name], rows_o[name], 2e-4, name)
        print(f"cfg4 long chains at B = 16, {name}: raw scores max relative error {worst}")
    assert rows_o["val"]["filename"].nunique() >= 16  # the ABSOLUTE bound of assert_z_close
    worst, auc_h, auc_o = assert_z_close(rows_h, rows_o)
    print(f"cfg4 at B = 16, t in (10, 250, 490): max |dZ| = {worst:.2e} (absolute), AUROC hip {auc_h:.4f} / oracle {auc_o:.4f}")
    live_oracle_pins_fixture("cfg4_b16", spec, rows_o, rows_h)


def test_cfg4_longest_chain_t990_in_a_batch_of_16(device, tmp_path):
    """The LAST t-start of cfg4's k = 2 list: t = 990, ONE trajectory of 100 `big` forwards (1 600 attention-block evaluations),
    run as the first image of a batch of 16 (bench.py's cfg4 dispatch, profiler-asserted) against the committed oracle row
    (tests/golden/rows_cfg4_t990.csv), pinned by the oracle run live on that image."""
    from parity_util import golden_rows, live_oracle_pins_fixture

    spec, rows_o = golden_rows("cfg4_t990")
    args, rec = _cfg4_rec(tmp_path, spec)
    rec.profile_first_steps = True
    full = hip_scores(args, rec, spec["sets"]["in"], "in")
    rec.profile_first_steps = False
    _assert_cfg4_dispatch(_report(), forwards=1)
    assert rec.last_stats["unet_forwards"] == 16 * 100 and sorted(set(full["t"])) == [990]
    keep = set(rows_o["in"]["filename"])
    assert len(keep) == 1
    rows_h = {"in": full[full["filename"].isin(keep)].reset_index(drop=True)}
    worst = assert_rows_close(rows_h["in"], rows_o["in"], 2e-4, "cfg4 t = 990")
    print(f"cfg4, t = 990 (100 forwards) at B = 16: raw scores max relative error {worst}")
    live_oracle_pins_fixture("cfg4_t990", spec, rows_o, rows_h)


def test_trained_weights_forward_and_trajectory_vs_oracle(device, tmp_path):
    """Parity on TRAINED weights: a short run of the product training loop (row f-3) moves the zero-initialised
    convolutions and the GroupNorm affines away from `random_state_dict`'s distribution; the checkpoint it writes is
    loaded by both sides.  B = 256 forward (the F(4x4) / split-f16 dispatch) and a k = 64 trajectory with Z-scores."""
    import oracle
    from ddpm_ood_amd.train import DDPMTrainer
    from ddpm_ood_amd.trainer import MODEL_CONFIGS, Reconstruct

    targs = argparse.Namespace(
        seed=2, output_dir=str(tmp_path), model_name="synth", training_ids="synthetic:blobs:n=512:seed=1",
        validation_ids="synthetic:blobs:n=16:seed=10", spatial_dimension=2, image_size=None, image_roi=None,
        latent_pad=None, vqvae_checkpoint=None, prediction_type="epsilon", model_type="small",
        beta_schedule="scaled_linear_beta", beta_start=0.0015, beta_end=0.0195, b_scale=1.0, snr_shift=1,
        simplex_noise=0, batch_size=64, n_epochs=6, eval_freq=6, augmentation=1, num_workers=0, cache_data=1,
        checkpoint_every=0, ddpm_checkpoint_epoch=None, is_grayscale=1, quick_test=0)
    tr = DDPMTrainer(targs)
    tr.train(targs)
    assert tr.history[-1][1] < tr.history[0][1]
    del tr
    torch.cuda.empty_cache()
    sd = torch.load(tmp_path / "synth" / "checkpoint.pth", map_location="cpu", weights_only=False)["model_state_dict"]

    ref, hip = _models(device, sd)
    yr, yh, prof = _forward_pair(device, ref, hip, 256, seed=77)
    err, scale = (yh - yr).abs().max().item(), yr.abs().max().item()
    print(f"trained weights, B = 256: max |eps_hip - eps_oracle| = {err:.3e} (max |eps| = {scale:.3f})")
    assert math.isfinite(err) and err <= 2e-5 * (1 + scale), err
    assert scale > 0.05 and "conv3x3_wino44h_gn_silu" in prof
    del hip

    sets = {"val": "synthetic:blobs:n=16:seed=10", "in": "synthetic:blobs:n=16:seed=11",
            "out": "synthetic:speckle:n=16:seed=12:mix=10"}
    args = make_args(tmp_path, inference_skip_factor=64, batch_size=16, validation_ids=sets["val"], in_ids=sets["in"])
    rec = Reconstruct(args)  # loads the trained checkpoint.pth written above
    rec.quiet = True
    rows_h = {n: hip_scores(args, rec, ids, n) for n, ids in sets.items()}
    rows_o = {n: oracle_scores(args, rec, ids, n, model=ref) for n, ids in sets.items()}
    for n in sets:
        assert_rows_close(rows_h[n], rows_o[n], 2e-4, n)
    worst, auc_h, auc_o = assert_z_close(rows_h, rows_o)
    print(f"trained weights, k = 64: max |dZ| = {worst:.2e}, AUROC hip {auc_h:.4f} / oracle {auc_o:.4f}")


# ---- cfg4 / cfg5 at the batches bench.py times (parity there was only tested at B = 2 / B = 1) --------------------

def test_big_forward_at_benchmarked_batch_vs_oracle(device):
    """BASELINE configs[3] (`big` UNet, /root/reference/src/trainers/base.py:77-86, 64x64x3) at bench.py's cfg4 batch of 16:
    one forward against the CPU oracle (5.4 TFLOP of oracle), and the profiler shows the launches bench.py times there --
    split-f16 F(4x4) ResnetBlock convolutions, the eight-wave attention over 4096 / 1024 / 256 tokens, the GroupNorm-ed q / k / v
    projection and the skip 1x1s on the DMA-fed kernel, the Upsample convolutions, the Downsample convolutions."""
    from ddpm_ood_amd.synthetic import random_state_dict

    ref, hip = _models(device, random_state_dict("big", 3, seed=1), channels=3, model_type="big")
    g = torch.Generator().manual_seed(404)
    x = torch.randn(16, 3, 64, 64, generator=g)
    t = torch.randint(0, 1000, (16,), generator=g)
    with torch.no_grad():
        yr = ref(x, timesteps=t)
    yh, prof = _profiled(lambda: hip(x.to(device), timesteps=t.to(device)).cpu())
    err, scale = (yh - yr).abs().max().item(), yr.abs().max().item()
    print(f"big, B = 16: max |eps_hip - eps_oracle| = {err:.3e} (max |eps| = {scale:.3f})")
    assert math.isfinite(err) and err <= 2e-5 * (1 + scale), err
    assert scale > 0.05
    # 2 ResnetBlocks x 3 levels down, 2 in the middle, 3 x 3 up = 17 blocks = 34 convolutions, all on the split-f16 F(4x4) kernel
    assert prof["conv3x3_wino44h_gn_silu"]["launches"] == 34, prof.get("conv3x3_wino44h_gn_silu")
    assert "conv3x3_wino44_gn_silu" not in prof and "conv3x3_wino_gn_silu" not in prof and "conv3x3_mfma_gn_silu" not in prof, sorted(prof)
    # 16 attention blocks: the ten over 4 096 / 1 024 tokens on the register-resident kernel (attention_fa.hip, from 1 024 tokens),
    # the six over 256 tokens on the LDS-exchange kernel; all sixteen q / k / v projections on the GroupNorm-ed DMA-fed 1x1
    assert prof["attention_fa"]["launches"] == 10 and prof["attention"]["launches"] == 6, (prof.get("attention_fa"), prof.get("attention"))
    assert prof["conv1x1_dma_gn"]["launches"] == 16, prof.get("conv1x1_dma_gn")
    assert prof["conv3x3_s2h"]["launches"] == 2 and prof["conv1x1_dma"]["launches"] >= 1, sorted(prof)
    ups = prof.get("conv3x3_wino44h_up", {"launches": 0})["launches"] + prof.get("conv3x3_wino_up", {"launches": 0})["launches"]
    assert ups == 2, sorted(prof)


def test_cfg4_trajectories_on_the_reference_t_list(device, tmp_path):
    """cfg4 on the reference's hard-coded 100-step schedule (reconstruct.py:118; inference_skip_factor = 2 -> t_start = 10, 30,
    50, ...): the first three chained t-starts (2 + 4 + 6 forwards per image; the `max_t_start` test hook cuts the list, not the
    schedule), val / in / out sets of two 64x64x3 images -> raw scores, Z-scores, AUROC."""
    import oracle
    from ddpm_ood_amd import synthetic
    from ddpm_ood_amd.trainer import MODEL_CONFIGS, Reconstruct

    sets = {"val": "synthetic:blobs:n=2:channels=3:size=64:seed=30", "in": "synthetic:blobs:n=2:channels=3:size=64:seed=31",
            "out": "synthetic:speckle:n=2:channels=3:size=64:seed=32:mix=10"}
    args = make_args(tmp_path, model_type="big", is_grayscale=0, inference_skip_factor=2, batch_size=2,
                     validation_ids=sets["val"], in_ids=sets["in"])
    sd = synthetic.random_state_dict("big", 3, seed=1)
    write_checkpoint(tmp_path, args, sd)
    rec = Reconstruct(args)
    rec.quiet = True
    rec.max_t_start = 50
    assert rec.num_inference_steps == 100
    ref = oracle.DiffusionModelUNet(2, 3, 3, **MODEL_CONFIGS["big"]).eval()
    ref.load_state_dict(sd)
    rows_h, rows_o = {}, {}
    for name, ids in sets.items():
        rows_h[name] = hip_scores(args, rec, ids, name)
        assert sorted(set(rows_h[name]["t"])) == [10, 30, 50] and rec.last_stats["unet_forwards"] == 2 * 12
        rows_o[name] = oracle_scores(args, rec, ids, name, model=ref)
        assert_rows_close(rows_h[name], rows_o[name], 2e-4, name)
    worst, auc_h, auc_o = assert_z_close(rows_h, rows_o)
    print(f"cfg4, 100-step list, t <= 50: max |dZ| = {worst:.2e}, AUROC hip {auc_h:.4f} / oracle {auc_o:.4f}")


def test_cfg5_decode_two_volumes_in_one_call_vs_oracle(device):
    """cfg5: re-quantise + decode (reconstruct.py:166 -> VQVAE.decode_stage_2_outputs) of TWO [128, 8, 8, 8] latents -> two
    128^3 volumes in one call, README VQ-VAE shape: every 3-D split-f16 F(4x4) launch fills the chip several times over (what
    bench.py's cfg5 times), against the CPU oracle's decoder on the same weights; codes identical."""
    from oracle.vqvae import VQVAE as OracleVQVAE
    from ddpm_ood_amd.vqvae import VQVAE

    from test_gpu_configs import VQ_README

    torch.manual_seed(3)
    ovq = OracleVQVAE(**VQ_README).eval()
    with torch.no_grad():
        ovq.quantizer.quantizer.embedding.weight.mul_(3.0)
    vq = VQVAE(**VQ_README).eval()
    vq.load_state_dict(ovq.state_dict())
    vq = vq.to(device)
    z = torch.randn(2, 128, 8, 8, 8, generator=torch.Generator().manual_seed(8)) * 2.0
    with torch.no_grad():
        xo = ovq.decode_stage_2_outputs(z)
    xh, prof = _profiled(lambda: vq.decode_stage_2_outputs(z.to(device)).cpu())
    assert xh.shape == xo.shape == (2, 1, 128, 128, 128)
    err, scale = (xh - xo).abs().max().item(), xo.abs().max().item()
    print(f"decode of 2 volumes: max |x_hip - x_oracle| = {err:.3e} (max |x| = {scale:.3f})")
    assert math.isfinite(err) and err <= 2e-5 * (1 + scale), err
    assert prof["conv3d_wino44h"]["launches"] >= 12 and "conv3d_wino44" not in prof, sorted(prof)
    assert "conv3d_transpose_k4s2" in prof and "convT3d_k4s2_cout1" in prof and "vq_nearest" in prof, sorted(prof)
