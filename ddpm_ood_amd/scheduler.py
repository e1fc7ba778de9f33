"""``PNDMScheduler`` / ``DDPMScheduler`` with the MONAI-Generative call surface.

Drop-in for ``generative.networks.schedulers.PNDMScheduler`` as exercised by the reference:
  ctor kwargs                          /root/reference/src/trainers/reconstruct.py:98-105
  readable + assignable tables         /root/reference/src/trainers/reconstruct.py:106-117
  set_timesteps / timesteps            /root/reference/src/trainers/reconstruct.py:118-120,149
  add_noise(original_samples=, noise=, timesteps=)   :143-147
  step(model_output, timestep, sample) -> (prev_sample, None)   :155-157
Host side: the schedule tables and the ~15 scalar operations of a PLMS step stay on the
host as fp32 torch CPU scalars (bit-identical to what the reference computes with 0-d
tensors); device side: ONE fused kernel per step (ddpm_plms_step_f32) instead of 2-6
full-tensor ATen launches, and one for add_noise.  PLMS state (ets, counter, cur_sample)
persists across calls until set_timesteps, exactly like the reference (quirk Q3).
"""

from __future__ import annotations

import numpy as np
import torch

from . import ops

_ALIASES = {
    "linear": "linear_beta", "linear_beta": "linear_beta",
    "scaled_linear": "scaled_linear_beta", "scaled_linear_beta": "scaled_linear_beta",
    "sigmoid": "sigmoid_beta", "sigmoid_beta": "sigmoid_beta",
    "cosine": "cosine",
}


def noise_schedule(schedule: str, num_train_timesteps: int, beta_start: float = 1e-4, beta_end: float = 2e-2,
                   sig_range: float = 6.0, s: float = 8e-3) -> torch.Tensor:
    name = _ALIASES.get(schedule)
    if name is None:
        raise ValueError(f"Unknown beta schedule {schedule}")
    T = num_train_timesteps
    if name == "linear_beta":
        return torch.linspace(beta_start, beta_end, T, dtype=torch.float32)
    if name == "scaled_linear_beta":
        return torch.linspace(beta_start ** 0.5, beta_end ** 0.5, T, dtype=torch.float32) ** 2
    if name == "sigmoid_beta":
        return torch.sigmoid(torch.linspace(-sig_range, sig_range, T)) * (beta_end - beta_start) + beta_start
    x = torch.linspace(0, T, T + 1)
    ac = torch.cos(((x / T) + s) / (1 + s) * torch.pi * 0.5) ** 2
    ac = ac / ac[0].item()
    return 1.0 - torch.clip(ac[1:] / ac[:-1], 0.0001, 0.9999)


class Scheduler:
    def __init__(self, num_train_timesteps: int = 1000, schedule: str = "linear_beta", **schedule_args):
        self.num_train_timesteps = num_train_timesteps
        self.betas = noise_schedule(schedule, num_train_timesteps, **schedule_args)
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.one = torch.tensor(1.0)
        self.timesteps = torch.arange(num_train_timesteps - 1, -1, -1)

    def add_noise(self, original_samples: torch.Tensor, noise: torch.Tensor, timesteps: torch.Tensor,
                  b_scale: float = 1.0) -> torch.Tensor:
        """sqrt(abar_t) * x0 + sqrt(1 - abar_t) * noise.  ``b_scale`` lets the trainer fold the
        reference's ``images * self.b_scale`` (reconstruct.py:144) into the same kernel."""
        ac = self.alphas_cumprod.to(dtype=torch.float32, device="cpu")
        t = timesteps.to("cpu").long()
        sa = (ac[t] ** 0.5).numpy()
        sb = ((1 - ac[t]) ** 0.5).numpy()
        return ops.add_noise(original_samples, noise, sa, sb, b_scale)


class DDPMScheduler(Scheduler):
    def __init__(self, num_train_timesteps: int = 1000, schedule: str = "linear_beta",
                 prediction_type: str = "epsilon", **schedule_args):
        super().__init__(num_train_timesteps, schedule, **schedule_args)
        self.prediction_type = prediction_type


class PNDMScheduler(Scheduler):
    def __init__(self, num_train_timesteps: int = 1000, schedule: str = "linear_beta", skip_prk_steps: bool = False,
                 set_alpha_to_one: bool = False, prediction_type: str = "epsilon", steps_offset: int = 0,
                 timestep_list: str = "monai", **schedule_args):
        super().__init__(num_train_timesteps, schedule, **schedule_args)
        if prediction_type not in ("epsilon", "v_prediction"):
            raise ValueError("Argument `prediction_type` must be a member of PNDMPredictionType")
        if not skip_prk_steps:
            raise NotImplementedError("the reconstruction path constructs PNDMScheduler(skip_prk_steps=True) only")
        if timestep_list not in ("monai", "diffusers"):
            raise ValueError("timestep_list must be 'monai' or 'diffusers'")
        self.prediction_type = prediction_type
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.init_noise_sigma = 1.0
        self.pndm_order = 4
        self.skip_prk_steps = skip_prk_steps
        self.steps_offset = steps_offset
        self.timestep_list = timestep_list  # 100-entry (default) vs 101-entry list
        self._coef_cache = {}
        self.cur_model_output = 0
        self.counter = 0
        self.cur_sample = None
        self.ets: list = []
        self.set_timesteps(num_train_timesteps)

    def set_timesteps(self, num_inference_steps: int, device=None) -> None:
        if num_inference_steps > self.num_train_timesteps:
            raise ValueError(
                f"`num_inference_steps`: {num_inference_steps} cannot be larger than `self.num_train_timesteps`:"
                f" {self.num_train_timesteps}")
        self.num_inference_steps = num_inference_steps
        step_ratio = self.num_train_timesteps // self.num_inference_steps
        ts = (np.arange(0, num_inference_steps) * step_ratio).round().astype(np.int64)
        ts += self.steps_offset
        if self.timestep_list == "diffusers":
            plms = np.concatenate([ts[:-1], ts[-2:-1], ts[-1:]])[::-1].copy()
        else:
            plms = ts[::-1].copy()
        self.timesteps = torch.from_numpy(plms.astype(np.int64))  # host tensor: supports reversed(), masks, iteration
        # the PLMS update steps by num_train_timesteps // (requested steps): the 101-entry diffusers list repeats
        # one timestep, it does not change the ratio (diffusers keeps the requested count for it)
        self._step_ratio = step_ratio
        self.num_inference_steps = len(self.timesteps)
        self.ets = []
        self.counter = 0
        self.cur_sample = None

    def step(self, model_output: torch.Tensor, timestep, sample: torch.Tensor):
        return self.step_plms(model_output, int(timestep), sample), None

    def plms_coefficients(self, timestep: int, prev_timestep: int):
        """fp32 scalars of _get_prev_sample, computed with the same 0-d torch CPU ops as the reference (memoised per
        (timestep, prev_timestep, table): ~70 us of 0-d tensor arithmetic otherwise, on every PLMS step)."""
        key = (timestep, prev_timestep, id(self.alphas_cumprod))
        hit = self._coef_cache.get(key)
        if hit is not None:
            return hit
        a_t = self.alphas_cumprod[timestep]
        a_p = self.alphas_cumprod[prev_timestep] if prev_timestep >= 0 else self.final_alpha_cumprod
        b_t = 1 - a_t
        b_p = 1 - a_p
        sample_coeff = (a_p / a_t) ** 0.5
        denom = a_t * b_p ** 0.5 + (a_t * b_t * a_p) ** 0.5
        out = (float(sample_coeff), float(a_p - a_t), float(denom), float(a_t ** 0.5), float(b_t ** 0.5))
        self._coef_cache[key] = out
        return out

    def step_plms(self, model_output: torch.Tensor, timestep: int, sample: torch.Tensor) -> torch.Tensor:
        ratio = self._step_ratio
        prev_timestep = timestep - ratio
        if self.counter != 1:
            self.ets = self.ets[-3:]
            self.ets.append(model_output)
        else:
            prev_timestep = timestep
            timestep = timestep + ratio

        if len(self.ets) == 1 and self.counter == 0:
            kind, es = 0, [model_output]
            self.cur_sample = sample
        elif len(self.ets) == 1 and self.counter == 1:
            kind, es = 1, [model_output, self.ets[-1]]
            sample = self.cur_sample
            self.cur_sample = None
        elif len(self.ets) == 2:
            kind, es = 2, [self.ets[-1], self.ets[-2]]
        elif len(self.ets) == 3:
            kind, es = 3, [self.ets[-1], self.ets[-2], self.ets[-3]]
        else:
            kind, es = 4, [self.ets[-1], self.ets[-2], self.ets[-3], self.ets[-4]]

        sc, ce, dn, va, vb = self.plms_coefficients(timestep, prev_timestep)
        prev = ops.plms_step(sample, es, kind, sc, ce, dn, v_prediction=self.prediction_type == "v_prediction",
                             v_a=va, v_b=vb)
        self.counter += 1
        return prev
