"""Oracle restatement of ``Reconstruct.get_scores`` -- the hot loops.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows, line for line in *behaviour*
(not in text), /root/reference/src/trainers/reconstruct.py:72-250:
  :97-105   one PNDMScheduler per batch (PLMS state is NOT reset between t-starts, Q3)
  :106-117  optional SNR shift of the alpha-bar table
  :118-120  set_timesteps(100); start points = reversed(timesteps)[1::k]
  :123-126  encode + optional latent pad
  :130-147  noise + add_noise            (noise is an explicit host-generated input, Q2)
  :149-157  PLMS loop: eps = model(x, t); x = scheduler.step(eps, t, x)
  :159-168  inverse pad, decode, / b_scale, clamp_(0, 1)
  :170-187  LPIPS (28-px inputs zero-padded to 32; 3D per item)
  :188-191  per-image MSE over the non-batch dims
  :192-204  one row per (image, t_start)
autocast is a no-op on CPU (Q5) so everything is fp32.
"""

from __future__ import annotations

from pathlib import Path

import torch
import torch.nn.functional as F

from .scheduler import PNDMScheduler


def snr_shift_tables(scheduler, snr_shift: float) -> None:
    """/root/reference/src/trainers/reconstruct.py:106-117."""
    snr = scheduler.alphas_cumprod / (1 - scheduler.alphas_cumprod)
    target_snr = snr * snr_shift
    new_ac = 1 / (torch.pow(target_snr, -1) + 1)
    new_alphas = torch.zeros_like(new_ac)
    new_alphas[0] = new_ac[0]
    for i in range(1, len(new_alphas)):
        new_alphas[i] = new_ac[i] / new_ac[i - 1]
    scheduler.betas = 1 - new_alphas
    scheduler.alphas = new_alphas
    scheduler.alphas_cumprod = new_ac


@torch.no_grad()
def get_scores(loader, dataset_name: str, inference_skip_factor: int, *, model, vqvae, perceptual,
               noise_generator: torch.Generator = None, noise_fn=None, spatial_dimension: int = 2,
               prediction_type: str = "epsilon", beta_schedule: str = "linear_beta",
               beta_start: float = 1e-4, beta_end: float = 2e-2, b_scale: float = 1.0,
               snr_shift: float = 1.0, latent_pad=None, num_inference_steps: int = 100,
               reset_scheduler_per_t: bool = False, timestep_list: str = "monai",
               return_reconstructions: bool = False, max_t_start: int = None, t_start_subset=None):
    results = []
    recons = []
    model.eval()
    for batch in loader:
        sched = PNDMScheduler(num_train_timesteps=1000, skip_prk_steps=True,
                              prediction_type=prediction_type, schedule=beta_schedule,
                              beta_start=beta_start, beta_end=beta_end, timestep_list=timestep_list)
        if snr_shift != 1:
            snr_shift_tables(sched, snr_shift)
        sched.set_timesteps(num_inference_steps)
        timesteps = sched.timesteps
        start_points = reversed(timesteps)[1::inference_skip_factor]
        if max_t_start is not None:  # test hook: a prefix of the chained t-start list (same trajectories, fewer of them)
            start_points = start_points[start_points <= int(max_t_start)]
        if t_start_subset is not None:  # test hook: some members of the chained list, in the list's order (long chains at a
            # price the CPU oracle can pay; the PLMS history a trajectory inherits is then the previous KEPT one's)
            start_points = start_points[torch.isin(start_points, torch.as_tensor(list(t_start_subset), dtype=start_points.dtype))]

        images_original = batch["image"].float()
        images = vqvae.encode_stage_2_inputs(images_original)
        if latent_pad:
            images = F.pad(images, latent_pad, mode="constant", value=0)
        for t_start in start_points:
            if reset_scheduler_per_t:
                sched.set_timesteps(num_inference_steps)
            start_ts = torch.Tensor([t_start] * images.shape[0]).long()
            # noise is an explicit input: either a pure function of (batch, t_start)
            # shared with the HIP trainer, or a host generator drawn in (batch, t_start) order
            if noise_fn is not None:
                noise = noise_fn(batch, int(t_start), images.shape)
            else:
                noise = torch.randn(images.shape, generator=noise_generator, dtype=torch.float32)
            x = sched.add_noise(original_samples=images * b_scale, noise=noise, timesteps=start_ts)
            for step in timesteps[timesteps <= t_start]:
                ts = torch.Tensor([step] * images.shape[0]).long()
                eps = model(x, timesteps=ts)
                x, _ = sched.step(eps, step, x)
            if latent_pad:
                x = F.pad(x, [-p for p in latent_pad], mode="constant", value=0)
            x = vqvae.decode_stage_2_outputs(x)
            x = x / b_scale
            x.clamp_(0, 1)
            if spatial_dimension == 2:
                if images_original.shape[3] == 28:
                    pd = perceptual(F.pad(images_original, (2, 2, 2, 2)), F.pad(x, (2, 2, 2, 2)))
                else:
                    pd = perceptual(images_original, x)
            else:
                pd = torch.empty(images.shape[0])
                for b in range(images.shape[0]):
                    pd[b] = perceptual(images_original[b, None, ...], x[b, None, ...])
            non_batch = tuple(range(images_original.dim()))[1:]
            mse = torch.square(images_original - x).mean(axis=non_batch)
            for b in range(images.shape[0]):
                filename = batch["image_meta_dict"]["filename_or_obj"][b]
                stem = Path(filename).stem.replace(".nii", "").replace(".gz", "")
                results.append({"filename": stem, "type": dataset_name, "t": t_start.item(),
                                "perceptual_difference": pd[b].item(), "mse": mse[b].item()})
            if return_reconstructions:
                recons.append(x.clone())
    if return_reconstructions:
        return results, recons
    return results
