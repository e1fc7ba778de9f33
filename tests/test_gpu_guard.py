"""-m gpu: operand range of the split-f16 kernel forms that read UN-NORMALISED tensors, and the numeric guard.

The ResnetBlock convolutions sit behind a GroupNorm, the Upsample / Downsample convolutions and the skip 1x1s read the raw
residual stream (generative's Upsample / Downsample / ResnetBlock.skip_connection inside DiffusionModelUNet.forward, call
site /root/reference/src/trainers/reconstruct.py:151-153).  Their split-f16 products are exact inside the f16 exponent
range and overflow to inf beyond it: here every such form is held against a float64 convolution over
input scales 1e-3 ... 100, one deliberately overflowing input shows what "beyond" looks like (non-finite output, never a
finite wrong one), and the guard of include/ddpm_ood_hip.h ("Numeric guard") is exercised end to end: status word bits,
the run-time switch to the fp32-MFMA kernels, the trainer's automatic second pass, NaN kept like torch.clamp_ keeps it.
"""

import math

import pandas as pd
import pytest
import torch
import torch.nn.functional as F

from parity_util import assert_rows_close, hip_scores, make_args, oracle_scores, write_checkpoint

pytestmark = pytest.mark.gpu

SCALES = (1e-3, 3e-2, 1.0, 30.0, 100.0)


def _rel_err(y, ref):
    err = y.cpu().double() - ref
    return err.abs().max().item() / ref.abs().max().item(), err.pow(2).mean().sqrt().item() / ref.pow(2).mean().sqrt().item()


@pytest.mark.parametrize("H", [16, 8])
def test_upsample_split_f16_operand_range(device, H, monkeypatch):
    """conv_wino44h_kernel<.., UP>: nearest x2 + conv3x3 on the raw residual stream (V pre-scale 2^0).  Full precision for
    patches up to ~650 / ~5 000 (worst case / typical data); two decades below O(1) the lo halves go subnormal and the error
    rises gracefully (absolute 2^-25 per operand), still far inside the 1e-4 bar."""
    monkeypatch.setenv("DDPM_CONV_WINO44", "2")
    from ddpm_ood_amd import ops

    g = torch.Generator().manual_seed(3 + H)
    x0 = torch.randn(3, 256, H, H, generator=g)
    w = torch.randn(256, 256, 3, 3, generator=g) / math.sqrt(256 * 9)
    b = torch.randn(256, generator=g)
    wh = ops.pack_wino44h_weight(w.to(device))
    for s in SCALES:
        x = x0 * s
        ref = F.conv2d(F.interpolate(x.double(), scale_factor=2, mode="nearest"), w.double(), b.double() * s, padding=1)
        y = ops.conv(x.to(device), w.to(device), (b * s).to(device), mode=ops.CONV_UPSAMPLE2, wino44h=wh)
        emax, erms = _rel_err(y, ref)
        print(f"upsample {H} -> {2 * H}, input scale {s:g}: max {emax:.2e} rms {erms:.2e}")
        assert math.isfinite(emax) and emax < (2e-4 if s >= 1e-2 else 1e-3) and erms < (1e-5 if s >= 1e-2 else 1e-4), (s, emax, erms)
    # beyond the range: the hi half of the transformed patch is inf -> the output is NOT finite (never finite and wrong)
    y = ops.conv((x0 * 3e4).to(device), w.to(device), b.to(device), mode=ops.CONV_UPSAMPLE2, wino44h=wh)
    assert not torch.isfinite(y).all()
    # ... and the same input on the fp32-MFMA kernels (run-time switch) is fine
    from ddpm_ood_amd import _lib

    assert _lib.set_split_f16(False) is True
    try:
        y32 = ops.conv((x0 * 3e4).to(device), w.to(device), b.to(device), mode=ops.CONV_UPSAMPLE2, wino44h=wh,
                       wino=ops.pack_wino_weight(w.to(device)))
    finally:
        _lib.set_split_f16(True)
    ref = F.conv2d(F.interpolate((x0 * 3e4).double(), scale_factor=2, mode="nearest"), w.double(), b.double(), padding=1)
    assert _rel_err(y32, ref)[0] < 2e-5


def test_downsample_split_f16_operand_range(device):
    """conv_s2h_kernel (both forms): the direct stride-2 convolution has no transform gain -- inputs up to 8 188."""
    from ddpm_ood_amd import ops

    g = torch.Generator().manual_seed(5)
    for B in (4, 260):  # 64 x 128 form / chip-filling four-tile form
        x0 = torch.randn(B, 128, 16, 16, generator=g)
        w = torch.randn(128, 128, 3, 3, generator=g) / math.sqrt(128 * 9)
        ws = ops.pack_conv_s2h_weight(w.to(device))
        n = min(B, 8)
        for s in SCALES:
            ref = F.conv2d((x0[:n] * s).double(), w.double(), None, stride=2, padding=1)
            y = ops.conv((x0 * s).to(device), w.to(device), None, mode=ops.CONV_STRIDE2, wino44h=ws)
            emax, erms = _rel_err(y[:n], ref)
            assert math.isfinite(emax) and emax < (4e-6 if s >= 1e-2 else 2e-5), (B, s, emax)
        y = ops.conv((x0 * 3e4).to(device), w.to(device), None, mode=ops.CONV_STRIDE2, wino44h=ws)
        assert not torch.isfinite(y).all()


def test_skip_1x1_split_f16_operand_range(device):
    """conv1x1_dma_kernel<true, false> (ResnetBlock.skip_connection over the virtual concat): inputs up to 6.5e4."""
    from ddpm_ood_amd import ops

    g = torch.Generator().manual_seed(9)
    x0 = torch.randn(96, 256, 16, 16, generator=g)
    x20 = torch.randn(96, 128, 16, 16, generator=g)
    w = torch.randn(256, 384, 1, 1, generator=g) / math.sqrt(384)
    b = torch.randn(256, generator=g)
    for s in SCALES:
        ref = F.conv2d(torch.cat([x0[:8], x20[:8]], 1).double() * s, w.double(), b.double() * s)
        y = ops.conv((x0 * s).to(device), w.to(device), (b * s).to(device), x2=(x20 * s).to(device))
        emax, _ = _rel_err(y[:8], ref)
        assert math.isfinite(emax) and emax < (3e-6 if s >= 1e-2 else 3e-5), (s, emax)
    y = ops.conv((x0 * 1e5).to(device), w.to(device), b.to(device), x2=(x20 * 1e5).to(device))
    assert not torch.isfinite(y).all()


def test_status_word_bits_and_nan_propagation(device):
    """The PLMS update flags a non-finite eps, clamp + MSE flags a non-finite reconstruction AND keeps the NaN (torch.clamp_
    semantics: fminf / fmaxf alone would have written 0 -- a finite wrong score), the quantiser flags a non-finite latent."""
    from ddpm_ood_amd import PNDMScheduler, _lib, ops

    _lib.status_read(clear=True)
    x = torch.rand(2, 1, 32, 32, device=device)
    good = x.clone()
    mse = ops.clamp_mse_(x.clone(), good, 1.0)
    assert _lib.status_read() == 0 and torch.isfinite(mse).all()
    bad = x.clone()
    bad[1, 0, 3, 4] = float("nan")
    bad[0, 0, 0, 0] = float("inf")
    mse = ops.clamp_mse_(x.clone(), bad, 1.0)
    assert _lib.status_read(clear=False) == 2 and _lib.status_read() == 2 and _lib.status_read() == 0  # sticky until cleared
    assert math.isnan(mse[1].item()) and torch.isnan(bad[1, 0, 3, 4]) and bad[0, 0, 0, 0].item() == 1.0  # inf clamps, NaN stays
    ref = torch.square(x[0].cpu() - bad[0].cpu()).mean().item()
    assert abs(mse[0].item() - ref) < 1e-6

    sched = PNDMScheduler(num_train_timesteps=1000, skip_prk_steps=True, schedule="scaled_linear_beta", beta_start=0.0015,
                          beta_end=0.0195)
    sched.set_timesteps(100)
    eps = torch.randn(2, 1, 32, 32, device=device)
    sched.step(eps, sched.timesteps[-2], x)
    assert _lib.status_read() == 0
    eps[1, 0, 5, 5] = float("inf")
    sched.step(eps, sched.timesteps[-2], x)
    assert _lib.status_read() == 1
    assert "eps" in _lib.status_text(1) and _lib.status_text(0) == "clean"

    z = torch.randn(1, 16, 4, 4, device=device)
    codebook = torch.randn(32, 16, device=device)
    ops.vq_nearest(z, codebook)
    assert _lib.status_read() == 0
    z[0, 3, 1, 1] = float("nan")
    ops.vq_nearest(z, codebook)
    assert _lib.status_read() == 4


def test_trainer_reruns_an_overflowing_batch_on_fp32_products(device, tmp_path, capfd, monkeypatch):
    """A checkpoint whose residual stream is far beyond the f16 range (conv_in scaled by 3e4: GroupNorm hides the scale from the
    ResnetBlock convolutions, the Upsample / Downsample / skip convolutions see it raw).  fp32 is fine with it -- the oracle
    scores are ordinary numbers -- while the split-f16 kernels overflow: the trainer must notice (status word), run the batch
    again on the fp32-MFMA kernels, say so, and return scores that match the oracle; afterwards the split-f16 kernels are on
    again."""
    import oracle
    from ddpm_ood_amd import _lib, synthetic
    from ddpm_ood_amd.trainer import MODEL_CONFIGS, Reconstruct

    # (the kernels a batch of three would take since round 4 -- conv_d3s.hip's one-shot forms -- read raw tensors with a 2^0
    # pre-scale and survive this stream; the throughput kernels of larger batches, 2^3 on the Downsample input, do not: test those)
    monkeypatch.setenv("DDPM_CONV_D3S", "0")
    ids = "synthetic:blobs:n=3:seed=21"
    args = make_args(tmp_path, inference_skip_factor=64, batch_size=3, validation_ids=ids, in_ids=ids)
    sd = synthetic.random_state_dict("small", 1, seed=1)
    sd["conv_in.conv.weight"] = sd["conv_in.conv.weight"] * 3e4
    sd["conv_in.conv.bias"] = sd["conv_in.conv.bias"] * 3e4
    write_checkpoint(tmp_path, args, sd)
    rec = Reconstruct(args)
    rec.quiet = True
    rec.max_t_start = 10  # 2 forwards per image
    ref = oracle.DiffusionModelUNet(2, 1, 1, **MODEL_CONFIGS["small"]).eval()
    ref.load_state_dict(sd)
    o = oracle_scores(args, rec, ids, "in", model=ref)
    assert o["mse"].notna().all() and (o["mse"] < 10).all()
    h = hip_scores(args, rec, ids, "in")
    err = capfd.readouterr().err
    assert rec.last_stats["batches_rerun_fp32"] == 1 and rec.last_stats["batches_nonfinite"] == 0, rec.last_stats
    assert "running 3 image(s) again with fp32 MFMA products" in err and rec.last_stats["images_rerun_fp32"] == 3
    assert _lib.split_f16() is True and _lib.status_read() == 0
    assert_rows_close(h, o, 1e-3, "guarded")  # (fp32 on a 3e4-scale stream: absolute rounding is 3e4 x the usual)
    # the small-launch kernels take the same stream without a second pass (raw inputs are split at 2^0: range 65 504)
    monkeypatch.setenv("DDPM_CONV_D3S", "1")
    rec1 = Reconstruct(args)
    rec1.quiet = True
    rec1.max_t_start = 10
    h1 = hip_scores(args, rec1, ids, "in")
    assert rec1.last_stats["batches_nonfinite"] == 0, rec1.last_stats
    assert_rows_close(h1, o, 1e-3, "small-launch kernels")
    # a stream inside the range does not trigger anything
    sd2 = synthetic.random_state_dict("small", 1, seed=1)
    write_checkpoint(tmp_path, args, sd2)
    rec2 = Reconstruct(args)
    rec2.quiet = True
    rec2.max_t_start = 10
    hip_scores(args, rec2, ids, "in")
    assert rec2.last_stats["batches_rerun_fp32"] == 0 and rec2.last_stats["batches_nonfinite"] == 0


def test_genuine_overflow_is_written_as_nan_like_the_reference(device, tmp_path, capfd):
    """Weights that overflow fp32 itself: the second pass is non-finite too, the scores are NaN (what the reference's
    `.item()` rows would hold, reconstruct.py:192-204) and the run says so instead of raising."""
    from ddpm_ood_amd import synthetic
    from ddpm_ood_amd.trainer import Reconstruct

    ids = "synthetic:blobs:n=2:seed=21"
    args = make_args(tmp_path, inference_skip_factor=64, batch_size=2, validation_ids=ids, in_ids=ids)
    sd = synthetic.random_state_dict("small", 1, seed=1)
    sd["out.2.conv.weight"] = sd["out.2.conv.weight"] * float("inf")
    write_checkpoint(tmp_path, args, sd)
    rec = Reconstruct(args)
    rec.quiet = True
    rec.max_t_start = 10
    h = hip_scores(args, rec, ids, "in")
    assert rec.last_stats["batches_rerun_fp32"] == 1 and rec.last_stats["batches_nonfinite"] == 1
    assert h["mse"].isna().all()
    assert "genuine overflow" in capfd.readouterr().err
