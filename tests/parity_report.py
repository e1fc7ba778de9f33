"""Print the actual HIP-vs-oracle errors (development tool; the pass/fail version is tests/test_gpu_unet.py).

    python tools/parity_report.py [--n 8] [--skip 64]
"""

import argparse
import sys
import tempfile
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))

import pandas as pd  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=8)
    ap.add_argument("--skip", type=int, default=64)
    a = ap.parse_args()

    import oracle
    from ddpm_ood_amd import DiffusionModelUNet, synthetic
    from ddpm_ood_amd.data import get_data_loader
    from ddpm_ood_amd.trainer import MODEL_CONFIGS, Reconstruct, batch_noise
    sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "tests"))
    from test_gpu_unet import _args

    import os
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    dev = torch.device("cuda:0")
    tmp = Path(tempfile.mkdtemp())
    args = _args(tmp, validation_ids=f"synthetic:blobs:n={a.n}:seed=10", in_ids=f"synthetic:blobs:n={a.n}:seed=11",
                 out_ids=f"synthetic:noise:n={a.n}:seed=12:name=MNIST", batch_size=a.n, inference_skip_factor=a.skip)
    sd = synthetic.write_checkpoint(tmp / args.model_name, "small", 1, seed=1)
    ref = oracle.DiffusionModelUNet(2, 1, 1, **MODEL_CONFIGS["small"]).eval()
    ref.load_state_dict(sd)
    hip = DiffusionModelUNet(2, 1, 1, **MODEL_CONFIGS["small"])
    hip.load_state_dict(sd)
    hip = hip.to(dev)
    x = torch.randn(4, 1, 32, 32, generator=torch.Generator().manual_seed(4))
    t = torch.tensor([10, 650, 990, 330])
    with torch.no_grad():
        yr = ref(x, timesteps=t)
    yh = hip(x.to(dev), timesteps=t.to(dev)).cpu()
    print(f"single forward: max|hip-ref| = {(yh - yr).abs().max():.3e}  (max|ref| = {yr.abs().max():.3f})")

    rec = Reconstruct(args)
    rec.quiet = True
    pl = oracle.PerceptualLoss(dimensions=2, include_pixel_loss=False, is_fake_3d=False, lpips_normalize=True)
    pl.perceptual_function.load_state_dict(rec._perceptual().perceptual_function.state_dict())
    rows_h, rows_o = {}, {}
    for name, ids in (("val", args.validation_ids), ("in", args.in_ids), ("out", args.out_ids)):
        loader = get_data_loader(ids, batch_size=a.n, is_grayscale=True)
        rows_h[name] = pd.DataFrame(rec.get_scores(loader, name, a.skip))
        rows_o[name] = pd.DataFrame(oracle.get_scores(
            loader, name, a.skip, model=ref, vqvae=oracle.PassthroughVQVAE(), perceptual=pl,
            noise_fn=lambda batch, tt, shape: batch_noise(2, batch["index"], tt, shape),
            beta_schedule="scaled_linear_beta", beta_start=0.0015, beta_end=0.0195))
        for col in ("mse", "perceptual_difference"):
            h, o = rows_h[name][col], rows_o[name][col]
            print(f"{name:4s} {col:22s} max rel err = {((h - o).abs() / (o.abs() + 1e-12)).max():.3e}")
    dh, _, auc_h = oracle.z_scores_and_auroc(rows_h["val"], rows_h["in"], rows_h["out"])
    do, _, auc_o = oracle.z_scores_and_auroc(rows_o["val"], rows_o["in"], rows_o["out"])
    for col in ("z_score_mse", "z_score_perceptual_difference"):
        print(f"{col:30s} max abs err = {(dh[col] - do[col]).abs().max():.3e}   (max |z| = {do[col].abs().max():.2f})")
    print(f"AUROC hip = {auc_h:.6f}  oracle = {auc_o:.6f}")


if __name__ == "__main__":
    main()
