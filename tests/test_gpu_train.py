"""-m gpu: row f-3 -- the DDPM training loop on the ROCm device, and the hand-over of its checkpoint to the HIP
reconstruction path (/root/reference/src/trainers/ddpm_trainer.py:66-124, base.py:156,166-187)."""

import argparse

import pandas as pd
import pytest
import torch

pytestmark = pytest.mark.gpu


def _train_args(tmp_path, **kw):
    d = dict(seed=2, output_dir=str(tmp_path), model_name="fashionmnist_trained",
             training_ids="synthetic:blobs:n=128:seed=1", validation_ids="synthetic:blobs:n=16:seed=10",
             spatial_dimension=2, image_size=None, image_roi=None, latent_pad=None, vqvae_checkpoint=None,
             prediction_type="epsilon", model_type="small", beta_schedule="scaled_linear_beta", beta_start=0.0015,
             beta_end=0.0195, b_scale=1.0, snr_shift=1, simplex_noise=0, batch_size=32, n_epochs=3, eval_freq=3,
             augmentation=1, num_workers=0, cache_data=1, checkpoint_every=2, ddpm_checkpoint_epoch=None,
             is_grayscale=1, quick_test=0)
    d.update(kw)
    return argparse.Namespace(**d)


def test_train_then_reconstruct_with_the_trained_checkpoint(device, tmp_path):
    from ddpm_ood_amd.train import DDPMTrainer, unet_forward_torch
    from ddpm_ood_amd.trainer import Reconstruct
    from parity_util import hip_scores, make_args

    args = _train_args(tmp_path)
    tr = DDPMTrainer(args)
    tr.train(args)
    losses = [l for _, l in tr.history]
    assert len(losses) == 3 and all(torch.isfinite(torch.tensor(losses)))
    assert losses[-1] < losses[0]  # eps-MSE goes down from the zero-output start (loss ~ 1 / batch element)
    run = tmp_path / args.model_name
    assert (run / "checkpoint.pth").exists() and (run / "checkpoint_2.pth").exists()
    ck = torch.load(run / "checkpoint.pth", map_location="cpu", weights_only=False)
    assert set(ck) == {"epoch", "global_step", "model_state_dict", "optimizer_state_dict", "best_loss"}
    assert ck["global_step"] == 128 * ck["epoch"]

    # the HIP engine and the training forward agree on the TRAINED weights (same parameter holders)
    x = torch.randn(4, 1, 32, 32, device=device, generator=torch.Generator(device=device).manual_seed(3))
    t = torch.tensor([10, 330, 650, 970], device=device)
    tr.model.eval()
    with torch.no_grad():
        y_t = unet_forward_torch(tr.model, x, t)
    y_h = tr.model(x, timesteps=t)
    assert (y_h - y_t).abs().max().item() <= 1e-4 * (1 + y_t.abs().max().item())
    assert y_t.abs().max() > 1e-3  # the zero-initialised output conv has moved

    # resume picks up epoch / optimizer state (base.py:133-158)
    tr2 = DDPMTrainer(_train_args(tmp_path, n_epochs=4))
    # (start_epoch = saved epoch + 1 although the saved value already is "next epoch": the reference's own
    # off-by-one, base.py:139 with :170, kept)
    assert tr2.start_epoch == ck["epoch"] + 1 and tr2.optimizer.state_dict()["state"]

    # and the reconstruction path loads it
    rargs = make_args(tmp_path, model_name=args.model_name, validation_ids="synthetic:blobs:n=2:seed=10",
                      in_ids="synthetic:blobs:n=2:seed=11", inference_skip_factor=64)
    rec = Reconstruct(rargs)
    rows = hip_scores(rargs, rec, "synthetic:blobs:n=2:seed=11", "in")
    assert len(rows) == 4 and rows["mse"].between(0, 1).all()
