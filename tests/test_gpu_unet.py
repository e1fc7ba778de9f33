"""-m gpu: whole-network and whole-trajectory parity of the HIP path against the CPU oracle.

Tolerance (stated by BASELINE.json's north_star): per-image MSE / LPIPS Z-scores within 1e-4
fp32 on identical inputs / seeds.  A single UNet forward is held to 1e-4 * (1 + max|ref|).
"""

import math

import pandas as pd
import pytest
import torch

pytestmark = pytest.mark.gpu

SMALL = dict(num_channels=(128, 256, 256), attention_levels=(False, False, True), num_res_blocks=1,
             num_head_channels=256)


def _pair(device, channels=1, cfg=SMALL, seed=1, spatial_dims=2, **extra):
    import oracle
    from ddpm_ood_amd import DiffusionModelUNet
    from ddpm_ood_amd.synthetic import random_state_dict

    sd = random_state_dict(channels=channels, seed=seed, config=cfg, spatial_dims=spatial_dims)
    ref = oracle.DiffusionModelUNet(spatial_dims, channels, channels, **cfg, **extra).eval()
    ref.load_state_dict(sd)
    hip = DiffusionModelUNet(spatial_dims, channels, channels, **cfg, **extra)
    hip.load_state_dict(sd)
    return ref, hip.to(device).eval()


@pytest.mark.parametrize("channels,B,H", [(1, 2, 32), (3, 3, 32), (1, 5, 16), (1, 1, 64), (1, 3, 28)])
def test_unet_forward_small(device, channels, B, H):
    ref, hip = _pair(device, channels)
    g = torch.Generator().manual_seed(4)
    x = torch.randn(B, channels, H, H, generator=g)
    t = torch.tensor([10, 650, 990, 0, 330][:B])
    with torch.no_grad():
        yr = ref(x, timesteps=t)
    yh = hip(x.to(device), timesteps=t.to(device)).cpu()
    err = (yh - yr).abs().max().item()
    assert math.isfinite(err) and err <= 1e-4 * (1 + yr.abs().max().item()), err
    assert yr.abs().max() > 0.05  # not vacuous (finding 12)


@pytest.mark.parametrize("B,mode", [(16, "1"), (7, "1"), (20, "2"), (33, "2")])
def test_unet_forward_small_launch_kernels_vs_oracle(device, B, mode, monkeypatch):
    """The one-shot kernels of launches far smaller than the chip (conv_d3s.hip: 3x3 at the 8x8 / 16x16 levels, 1x1 skip
    connections and q / k / v; BASELINE configs[0] runs 16 images) inside a whole forward: the profiler must see them, the
    forward must agree with the same forward without them (DDPM_CONV_D3S=0) and with the CPU oracle.  Mode 2 forces them onto
    launches beyond their size gate -- larger scratch than any other kernel asks for: the engine's sizing pass has to take the
    decisions of the run (a NULL scale / shift in the dry run once sized them out: out-of-bounds scratch)."""
    import ctypes
    import json

    from ddpm_ood_amd import _lib

    monkeypatch.setenv("DDPM_CONV_D3S", mode)
    ref, hip = _pair(device, 1)
    g = torch.Generator().manual_seed(B)
    x = torch.randn(B, 1, 32, 32, generator=g)
    t = torch.randint(0, 1000, (B,), generator=g)
    lib = _lib.load()
    lib.ddpm_prof_enable(1)
    y = hip(x.to(device), timesteps=t.to(device)).cpu()
    torch.cuda.synchronize()
    lib.ddpm_prof_enable(0)
    buf = ctypes.create_string_buffer(1 << 18)
    prof = json.loads(buf.value.decode()) if lib.ddpm_prof_report(buf, len(buf)) > 0 else {}
    assert any(k.startswith("conv3x3_d3s") for k in prof) and any(k.startswith("conv1x1_d1s") for k in prof), sorted(prof)
    assert torch.equal(y, hip(x.to(device), timesteps=t.to(device)).cpu())  # fixed-order slices: bit-reproducible
    monkeypatch.setenv("DDPM_CONV_D3S", "0")
    hip0 = _pair(device, 1)[1]  # (a new engine: the switch decides which planes it packs)
    y0 = hip0(x.to(device), timesteps=t.to(device)).cpu()
    scale = y0.abs().max().item()
    assert scale > 0.05 and not torch.equal(y, y0)
    assert (y - y0).abs().max().item() <= 2e-5 * (1 + scale)
    n = min(B, 6)
    with torch.no_grad():
        yr = ref(x[:n], timesteps=t[:n])
    assert (y[:n] - yr).abs().max().item() <= 1e-4 * (1 + yr.abs().max().item())


@pytest.mark.parametrize("channels,B,H", [(1, 96, 32), (3, 70, 32), (1, 300, 16), (1, 20, 64), (1, 90, 28)])
def test_unet_forward_fused_groupnorm_statistics_vs_reading_groupnorm(device, channels, B, H, monkeypatch):
    """Launch sets of >= 64 K pixels take the GroupNorm statistics from the producers' epilogues: the forward
    must agree with the reading GroupNorm (DDPM_GN_FUSED=0) to fp32 rounding and with the CPU oracle -- across image sizes
    with 1, 2 and 8 slices per image, a size the F(4x4) kernel has no tiling for (28: statistics from channel_stats), and
    concatenations whose groups straddle the seam (384 = 256 + 128 channels)."""
    from ddpm_ood_amd import _lib

    ref, hip = _pair(device, channels)
    g = torch.Generator().manual_seed(B + H)
    x = torch.randn(B, channels, H, H, generator=g)
    t = torch.randint(0, 1000, (B,), generator=g)
    import ctypes
    import json

    lib = _lib.load()
    lib.ddpm_prof_enable(1)
    y_fused = hip(x.to(device), timesteps=t.to(device)).cpu()
    torch.cuda.synchronize()
    lib.ddpm_prof_enable(0)
    buf = ctypes.create_string_buffer(1 << 18)
    prof = json.loads(buf.value.decode()) if lib.ddpm_prof_report(buf, len(buf)) > 0 else {}
    assert "gn_finalize" in prof and "gn_scale_shift" not in prof, sorted(prof)
    monkeypatch.setenv("DDPM_GN_FUSED", "0")
    y_read = hip(x.to(device), timesteps=t.to(device)).cpu()
    scale = y_read.abs().max().item()
    assert scale > 0.05
    assert (y_fused - y_read).abs().max().item() <= 2e-5 * (1 + scale)
    n = min(B, 8)  # (the oracle runs on the host: a few images are enough)
    with torch.no_grad():
        yr = ref(x[:n], timesteps=t[:n])
    assert (y_fused[:n] - yr).abs().max().item() <= 1e-4 * (1 + yr.abs().max().item())


def test_unet_forward_generic_config_and_proj_attn(device):
    """Channel counts without an MFMA tiling (64/96) run entirely on the direct kernels; also covers
    use_proj_attn=True, 2 res blocks and attention at an upper level with n = 256 tokens."""
    cfg = dict(num_channels=(64, 256), attention_levels=(False, True), num_res_blocks=2, num_head_channels=256)
    ref, hip = _pair(device, 1, cfg, use_proj_attn=True)
    x = torch.randn(2, 1, 32, 32, generator=torch.Generator().manual_seed(8))
    t = torch.tensor([500, 20])
    with torch.no_grad():
        yr = ref(x, timesteps=t)
    yh = hip(x.to(device), timesteps=t.to(device)).cpu()
    assert (yh - yr).abs().max().item() <= 1e-4 * (1 + yr.abs().max().item())


def test_unet_forward_big_cfg4(device):
    """BASELINE configs[3]: the attention-heavy `big` UNet (172.6 M params) at 64x64x3: attention at every
    level (n = 4096 / 1024 / 256 tokens, 1 / 2 / 3 heads of 256), two res-blocks per level."""
    from ddpm_ood_amd.trainer import MODEL_CONFIGS

    ref, hip = _pair(device, 3, MODEL_CONFIGS["big"])
    x = torch.randn(1, 3, 64, 64, generator=torch.Generator().manual_seed(6))
    t = torch.tensor([870])
    with torch.no_grad():
        yr = ref(x, timesteps=t)
    yh = hip(x.to(device), timesteps=t.to(device)).cpu()
    err = (yh - yr).abs().max().item()
    assert math.isfinite(err) and err <= 1e-4 * (1 + yr.abs().max().item()), err
    assert yr.abs().max() > 0.05


@pytest.mark.parametrize("B,size", [(2, 8), (1, 16)])
def test_unet_forward_3d_cfg5(device, B, size):
    """BASELINE configs[4]: the `small` UNet over 3-D VQ-VAE latents [B, 128, 8, 8, 8] (47.5 M params).
    conv3d: one launch per conv, the depth taps part of the chunk stream; attention sees 2^3 = 8 tokens."""
    ref, hip = _pair(device, 128, SMALL, spatial_dims=3)
    x = torch.randn(B, 128, size, size, size, generator=torch.Generator().manual_seed(12))
    t = torch.tensor([650, 30][:B])
    with torch.no_grad():
        yr = ref(x, timesteps=t)
    yh = hip(x.to(device), timesteps=t.to(device)).cpu()
    err = (yh - yr).abs().max().item()
    assert math.isfinite(err) and err <= 1e-4 * (1 + yr.abs().max().item()), err
    assert yr.abs().max() > 0.05
    if size == 8:
        # the 8^3 level runs in the Winograd domain (2-D F(2x2, 3x3) per depth tap with the GroupNorm + SiLU prologue,
        # concat and temb: four 8x8 slices per work item); out-of-volume depth taps must stay zero through the prologue
        import ctypes
        import json

        from ddpm_ood_amd import _lib
        lib = _lib.load()
        lib.ddpm_prof_enable(1)
        hip(x.to(device), timesteps=t.to(device))
        lib.ddpm_prof_enable(0)
        torch.cuda.synchronize()
        buf = ctypes.create_string_buffer(1 << 18)
        lib.ddpm_prof_report(buf, len(buf))
        assert "conv3d_wino_gn_silu" in json.loads(buf.value.decode())


def test_unet_forward_batch_1024_matches_256_image_chunks(device):
    """The bench's batch (1 024 images per launch set, sized for the 288 GB of HBM) against the same images in four
    chunks of 256: per-image arithmetic does not depend on the batch except where a launch is split over K (the 8x8
    level at B = 256), so the two agree to fp32 summation-order noise -- and no 32-bit offset wraps at 1 024."""
    _, hip = _pair(device, 1)
    g = torch.Generator().manual_seed(21)
    x = torch.randn(1024, 1, 32, 32, generator=g).to(device)
    t = torch.randint(0, 1000, (1024,), generator=g).to(device)
    y = hip(x, timesteps=t)
    parts = torch.cat([hip(x[i:i + 256].contiguous(), timesteps=t[i:i + 256].contiguous()) for i in range(0, 1024, 256)])
    torch.cuda.synchronize()
    err = (y - parts).abs().max().item()
    assert math.isfinite(err) and err <= 2e-5 * (1 + parts.abs().max().item()), err
    assert parts.abs().max() > 0.05


def test_graph_replay_equals_eager(device, monkeypatch):
    """the small-batch forward replayed from a captured hipGraph
    (ddpm_unet_forward_graphed: eager, capture, then replay) returns bit-identical results to the eager launch
    sequence, for changing inputs and timesteps, and survives a parameter update (graphs are dropped)."""
    from ddpm_ood_amd import DiffusionModelUNet, _lib
    from ddpm_ood_amd.synthetic import random_state_dict

    sd = random_state_dict("small", 1, seed=1)
    m = DiffusionModelUNet(2, 1, 1, **SMALL)
    m.load_state_dict(sd)
    m = m.to(device).eval()
    g = torch.Generator().manual_seed(9)
    xs = [torch.randn(4, 1, 32, 32, generator=g).to(device) for _ in range(4)]
    ts = [torch.tensor(t, device=device) for t in ([10, 10, 10, 10], [650, 30, 0, 990], [5, 6, 7, 8], [640] * 4)]
    monkeypatch.setenv("DDPM_UNET_GRAPH", "0")
    eager = [m(x, timesteps=t).clone() for x, t in zip(xs, ts)]
    monkeypatch.setenv("DDPM_UNET_GRAPH", "1")
    graphed = [m(x, timesteps=t) for x, t in zip(xs, ts)]  # call 1 eager, call 2 captures, calls 3-4 replay
    torch.cuda.synchronize()
    assert _lib.load().ddpm_unet_num_graphs(m._engine) == 1
    for a, b in zip(eager, graphed):
        assert torch.equal(a, b)
    assert len({y.data_ptr() for y in graphed}) == 4  # results are not views of the static output buffer
    with torch.no_grad():  # parameter update -> blob re-sync -> graphs dropped, results follow the new weights
        m.out[2].conv.weight.mul_(2.0)
        m.out[2].conv.bias.zero_()
    y2 = m(xs[0], timesteps=ts[0])
    assert _lib.load().ddpm_unet_num_graphs(m._engine) == 0
    monkeypatch.setenv("DDPM_UNET_GRAPH", "0")
    assert torch.equal(y2, m(xs[0], timesteps=ts[0]))
    assert not torch.equal(y2, eager[0])
    monkeypatch.delenv("DDPM_UNET_GRAPH")
    assert not m._use_graph(16)  # opt-in: measured no gain (kernel-time-bound even at B = 4)


def test_graph_replay_follows_the_split_f16_switch(device, monkeypatch):
    """a captured forward bakes in WHICH kernels ran.  After ddpm_set_split_f16(0) -- the numeric guard's second pass,
    bench.py's fp32-products leg -- the replay must come from the fp32-MFMA kernels, not from the graph captured with the split-f16
    ones (the engine drops its graphs when the switch epoch changes), and flipping back must return the split-f16 result."""
    from ddpm_ood_amd import DiffusionModelUNet, _lib
    from ddpm_ood_amd.synthetic import random_state_dict

    m = DiffusionModelUNet(2, 1, 1, **SMALL)
    m.load_state_dict(random_state_dict("small", 1, seed=1))
    m = m.to(device).eval()
    g = torch.Generator().manual_seed(11)
    x = torch.randn(4, 1, 32, 32, generator=g).to(device)
    t = torch.tensor([10, 650, 30, 990], device=device)
    monkeypatch.setenv("DDPM_UNET_GRAPH", "0")
    y_split = m(x, timesteps=t).clone()
    prev = _lib.set_split_f16(False)
    try:
        y_fp32 = m(x, timesteps=t).clone()
    finally:
        _lib.set_split_f16(prev)
    assert not torch.equal(y_split, y_fp32)  # (different kernels: different last bits)
    monkeypatch.setenv("DDPM_UNET_GRAPH", "1")
    for _ in range(3):  # eager, capture, replay -- with the split-f16 kernels
        assert torch.equal(m(x, timesteps=t), y_split)
    assert _lib.load().ddpm_unet_num_graphs(m._engine) == 1
    prev = _lib.set_split_f16(False)
    try:
        for _ in range(3):  # same key (same buffers, same extents): the stale graph must not be replayed
            assert torch.equal(m(x, timesteps=t), y_fp32)
    finally:
        _lib.set_split_f16(prev)
    for _ in range(3):
        assert torch.equal(m(x, timesteps=t), y_split)


def test_unet_missing_key_and_bad_shape(device):
    from ddpm_ood_amd import DiffusionModelUNet

    m = DiffusionModelUNet(2, 1, 1, **SMALL).to(device)
    with pytest.raises(ValueError):
        m(torch.zeros(1, 1, 30, 30, device=device), timesteps=torch.zeros(1, dtype=torch.long, device=device))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 1, 32, 32), timesteps=torch.zeros(1, dtype=torch.long))


def _args(tmp_path, **kw):
    import argparse

    d = dict(seed=2, output_dir=str(tmp_path), model_name="fashionmnist_synth", validation_ids="synthetic:blobs:n=4:seed=10",
             in_ids="synthetic:blobs:n=4:seed=11", out_ids="synthetic:noise:n=4:seed=12:name=MNIST",
             spatial_dimension=2, image_size=None, image_roi=None, latent_pad=None, vqvae_checkpoint=None,
             ddpm_checkpoint_epoch=None, prediction_type="epsilon", model_type="small",
             beta_schedule="scaled_linear_beta", beta_start=0.0015, beta_end=0.0195, b_scale=1.0, snr_shift=1,
             simplex_noise=0, batch_size=4, augmentation=0, cache_data=1, num_workers=0, first_n_val=None,
             first_n=None, eval_checkpoint=None, drop_last=False, is_grayscale=1, run_val=1, run_in=1, run_out=1,
             num_inference_steps=100, inference_skip_factor=64)
    d.update(kw)
    return argparse.Namespace(**d)


def test_trajectory_scores_match_oracle(device, tmp_path):
    """cfg1-shaped: k = 64 -> t in {10, 650}, 68 UNet forwards per image, stale PLMS history
    carried from the t=10 trajectory into the t=650 one (Q3).  HIP path vs CPU oracle on the same
    images, weights and per-image noise: MSE / LPIPS per (image, t) and the Z-scores built from them."""
    import oracle
    from ddpm_ood_amd import synthetic
    from ddpm_ood_amd.data import get_data_loader
    from ddpm_ood_amd.trainer import Reconstruct, batch_noise

    args = _args(tmp_path)
    sd = synthetic.write_checkpoint(tmp_path / args.model_name, "small", 1, seed=1)
    rec = Reconstruct(args)
    rows_h = {}
    for name, ids in (("val", args.validation_ids), ("in", args.in_ids), ("out", args.out_ids)):
        loader = get_data_loader(ids, batch_size=4, is_grayscale=True)
        rows_h[name] = pd.DataFrame(rec.get_scores(loader, name, 64))
    assert rec.last_stats["unet_forwards"] == 4 * 68

    ref = oracle.DiffusionModelUNet(2, 1, 1, **SMALL).eval()
    ref.load_state_dict(sd)
    pl = oracle.PerceptualLoss(dimensions=2, include_pixel_loss=False, is_fake_3d=False, lpips_normalize=True)
    pl.perceptual_function.load_state_dict(rec._perceptual().perceptual_function.state_dict())

    class NoiseFromTrainer:  # the oracle draws noise through the same pure function of (seed, image, t)
        def __init__(self):
            self.calls = []

    rows_o = {}
    for name, ids in (("val", args.validation_ids), ("in", args.in_ids), ("out", args.out_ids)):
        loader = get_data_loader(ids, batch_size=4, is_grayscale=True)
        rows_o[name] = pd.DataFrame(oracle.get_scores(
            loader, name, 64, model=ref, vqvae=oracle.PassthroughVQVAE(), perceptual=pl,
            noise_fn=lambda batch, t, shape: batch_noise(2, batch["index"], t, shape),
            beta_schedule="scaled_linear_beta", beta_start=0.0015, beta_end=0.0195))
    for name in rows_h:
        h, o = rows_h[name], rows_o[name]
        assert list(h["filename"]) == list(o["filename"]) and list(h["t"]) == list(o["t"])
        assert sorted(set(h["t"])) == [10, 650]
        for col in ("mse", "perceptual_difference"):
            rel = ((h[col] - o[col]).abs() / (o[col].abs() + 1e-6)).max()
            assert rel < 2e-4, (name, col, rel)
    dh, _, auc_h = oracle.z_scores_and_auroc(rows_h["val"], rows_h["in"], rows_h["out"])
    do, _, auc_o = oracle.z_scores_and_auroc(rows_o["val"], rows_o["in"], rows_o["out"])
    for col in ("z_score_mse", "z_score_perceptual_difference"):
        # (a handful of validation images: per element relative to max(1, |Z|), see parity_util.assert_z_close)
        assert ((dh[col] - do[col]).abs() / do[col].abs().clip(lower=1.0)).max() < 1e-4, col
    assert abs(auc_h - auc_o) <= 1e-3


VQ_CFG = dict(spatial_dims=3, in_channels=1, out_channels=1, num_channels=(16, 32), num_res_layers=1,
              num_res_channels=(16, 32), downsample_parameters=((2, 4, 1, 1), (2, 4, 1, 1)),
              upsample_parameters=((2, 4, 1, 1, 0), (2, 4, 1, 1, 0)), num_embeddings=64, embedding_dim=128)


def test_ldm_trajectory_3d_matches_oracle(device, tmp_path):
    """cfg5-shaped (scaled down): 32^3 volumes -> VQ-VAE (2 stride-2 levels, 64 codes x 128) -> latents
    [B, 128, 8, 8, 8] -> `small` 3-D UNet PLMS trajectories (t = 10, 650) -> re-quantise + decode -> 2.5-D LPIPS
    + MSE.  Everything on the HIP library -- the 16 / 32-channel VQ-VAE layers have no MFMA tiling and take the generic
    kernel (ddpm_convnd_generic_f32); there is no PyTorch-ROCm / MIOpen route in the product -- vs the CPU oracle, 2e-4."""
    import json

    import oracle
    from oracle.vqvae import VQVAE as OracleVQVAE
    from ddpm_ood_amd import synthetic
    from ddpm_ood_amd.data import get_data_loader
    from ddpm_ood_amd.trainer import Reconstruct, batch_noise

    torch.manual_seed(3)
    vq = OracleVQVAE(**VQ_CFG).eval()
    with torch.no_grad():  # spread the codebook so that nearest-code decisions are not knife-edge
        vq.quantizer.quantizer.embedding.weight.mul_(3.0)
    vq_dir = tmp_path / "vqvae"
    vq_dir.mkdir()
    torch.save({"model_state_dict": vq.state_dict()}, vq_dir / "checkpoint.pth")
    json.dump(VQ_CFG, open(vq_dir / "vqvae_config.json", "w"))

    ids = "synthetic:blobs3d:n=2:size=32:seed=5"
    args = _args(tmp_path, model_name="decathlon_synth", validation_ids=ids, in_ids=ids, out_ids=ids,
                 spatial_dimension=3, vqvae_checkpoint=str(vq_dir / "checkpoint.pth"), batch_size=2)
    sd = synthetic.random_state_dict("small", 128, spatial_dims=3, seed=1)
    (tmp_path / args.model_name).mkdir()
    torch.save({"epoch": 0, "global_step": 0, "model_state_dict": sd, "best_loss": 1.0},
               tmp_path / args.model_name / "checkpoint.pth")
    rec = Reconstruct(args)
    loader = get_data_loader(ids, batch_size=2, is_grayscale=True, spatial_dimension=3)
    rows_h = pd.DataFrame(rec.get_scores(loader, "val", 64))

    ref = oracle.DiffusionModelUNet(3, 128, 128, **SMALL).eval()
    ref.load_state_dict(sd)
    pl = oracle.PerceptualLoss(dimensions=3, include_pixel_loss=False, is_fake_3d=True, lpips_normalize=True)
    pl.perceptual_function.load_state_dict(rec._perceptual().perceptual_function.state_dict())
    rows_o = pd.DataFrame(oracle.get_scores(
        loader, "val", 64, model=ref, vqvae=vq, perceptual=pl, spatial_dimension=3,
        noise_fn=lambda batch, t, shape: batch_noise(2, batch["index"], t, shape),
        beta_schedule="scaled_linear_beta", beta_start=0.0015, beta_end=0.0195))
    assert list(rows_h["t"]) == list(rows_o["t"]) == [10, 10, 650, 650]
    for col in ("mse", "perceptual_difference"):
        rel = ((rows_h[col] - rows_o[col]).abs() / (rows_o[col].abs() + 1e-9)).max()
        assert rel < 2e-4, (col, rel, rows_h[col].tolist(), rows_o[col].tolist())


def test_trajectory_native_28x28_lpips_pad_branch(device, tmp_path):
    """The reference's README runs FashionMNIST at its native 28x28: the UNet sees 28 -> 14 -> 7 (ragged MFMA
    tiles) and LPIPS gets both images zero-padded to 32 (reconstruct.py:171-178)."""
    import oracle
    from ddpm_ood_amd import synthetic
    from ddpm_ood_amd.data import get_data_loader
    from ddpm_ood_amd.trainer import Reconstruct, batch_noise

    ids = "synthetic:blobs:n=3:size=28:seed=21"
    args = _args(tmp_path, validation_ids=ids, in_ids=ids, out_ids=ids, batch_size=3)
    sd = synthetic.write_checkpoint(tmp_path / args.model_name, "small", 1, seed=1)
    rec = Reconstruct(args)
    loader = get_data_loader(ids, batch_size=3, is_grayscale=True)
    assert next(iter(loader))["image"].shape == (3, 1, 28, 28)
    rows_h = pd.DataFrame(rec.get_scores(loader, "val", 64))
    ref = oracle.DiffusionModelUNet(2, 1, 1, **SMALL).eval()
    ref.load_state_dict(sd)
    pl = oracle.PerceptualLoss(dimensions=2, include_pixel_loss=False, is_fake_3d=False, lpips_normalize=True)
    pl.perceptual_function.load_state_dict(rec._perceptual().perceptual_function.state_dict())
    rows_o = pd.DataFrame(oracle.get_scores(
        loader, "val", 64, model=ref, vqvae=oracle.PassthroughVQVAE(), perceptual=pl,
        noise_fn=lambda batch, t, shape: batch_noise(2, batch["index"], t, shape),
        beta_schedule="scaled_linear_beta", beta_start=0.0015, beta_end=0.0195))
    for col in ("mse", "perceptual_difference"):
        rel = ((rows_h[col] - rows_o[col]).abs() / (rows_o[col].abs() + 1e-9)).max()
        assert rel < 1e-4, (col, rel)
