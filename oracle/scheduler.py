"""Oracle restatement of the MONAI-Generative 0.2.x schedulers used on the hot path.

TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED: ``generative`` is absent
from /root/reference; this follows SURVEY.md Appendix A.4 and the reference call sites:
  * ctor kwargs            /root/reference/src/trainers/reconstruct.py:98-105
  * betas/alphas rewrite   /root/reference/src/trainers/reconstruct.py:106-117
  * set_timesteps(100)     /root/reference/src/trainers/reconstruct.py:118-120
  * add_noise              /root/reference/src/trainers/reconstruct.py:143-147
  * step -> 2-tuple        /root/reference/src/trainers/reconstruct.py:155-157
Everything is torch CPU fp32 (the reference's autocast is a no-op on CPU).
"""

from __future__ import annotations

import math

import numpy as np
import torch

# schedule-name drift in the reference: accept both spellings
_ALIASES = {
    "linear": "linear_beta",
    "linear_beta": "linear_beta",
    "scaled_linear": "scaled_linear_beta",
    "scaled_linear_beta": "scaled_linear_beta",
    "sigmoid": "sigmoid_beta",
    "sigmoid_beta": "sigmoid_beta",
    "cosine": "cosine",
}


def make_betas(schedule: str, num_train_timesteps: int, beta_start: float = 1e-4,
               beta_end: float = 2e-2, sig_range: float = 6.0, s: float = 8e-3) -> torch.Tensor:
    """NoiseSchedules[...] of generative/networks/schedulers/scheduler.py (Appendix A.4)."""
    name = _ALIASES.get(schedule)
    if name is None:
        raise ValueError(f"Unknown beta schedule {schedule}")
    T = num_train_timesteps
    if name == "linear_beta":
        return torch.linspace(beta_start, beta_end, T, dtype=torch.float32)
    if name == "scaled_linear_beta":
        return torch.linspace(beta_start ** 0.5, beta_end ** 0.5, T, dtype=torch.float32) ** 2
    if name == "sigmoid_beta":
        b = torch.linspace(-sig_range, sig_range, T)
        return torch.sigmoid(b) * (beta_end - beta_start) + beta_start
    # cosine
    x = torch.linspace(0, T, T + 1)
    ac = torch.cos(((x / T) + s) / (1 + s) * torch.pi * 0.5) ** 2
    ac = ac / ac[0].item()
    alphas = torch.clip(ac[1:] / ac[:-1], 0.0001, 0.9999)
    return 1.0 - alphas


class _Scheduler:
    def __init__(self, num_train_timesteps: int = 1000, schedule: str = "linear_beta", **schedule_args):
        self.num_train_timesteps = num_train_timesteps
        self.betas = make_betas(schedule, num_train_timesteps, **schedule_args)
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.one = torch.tensor(1.0)
        self.timesteps = torch.arange(num_train_timesteps - 1, -1, -1)

    def add_noise(self, original_samples: torch.Tensor, noise: torch.Tensor,
                  timesteps: torch.Tensor) -> torch.Tensor:
        ac = self.alphas_cumprod.to(device=original_samples.device, dtype=original_samples.dtype)
        timesteps = timesteps.to(original_samples.device)
        shape = (-1,) + (1,) * (original_samples.ndim - 1)
        sqrt_ac = (ac[timesteps] ** 0.5).reshape(shape)
        sqrt_1m = ((1 - ac[timesteps]) ** 0.5).reshape(shape)
        return sqrt_ac * original_samples + sqrt_1m * noise


class DDPMScheduler(_Scheduler):
    """Only the table construction is on the path (/root/reference/src/trainers/base.py:97-103)."""

    def __init__(self, num_train_timesteps: int = 1000, schedule: str = "linear_beta",
                 prediction_type: str = "epsilon", **schedule_args):
        super().__init__(num_train_timesteps, schedule, **schedule_args)
        self.prediction_type = prediction_type


class PNDMScheduler(_Scheduler):
    """PLMS (skip_prk_steps=True) multistep scheduler, state persisting across calls.

    ``timestep_list`` selects the open question Q9 of SURVEY Appendix C:
    "monai" (default) -> n entries 990..0; "diffusers" -> n+1 entries with the
    second-to-last step duplicated.
    """

    def __init__(self, num_train_timesteps: int = 1000, schedule: str = "linear_beta",
                 skip_prk_steps: bool = False, set_alpha_to_one: bool = False,
                 prediction_type: str = "epsilon", steps_offset: int = 0,
                 timestep_list: str = "monai", **schedule_args):
        super().__init__(num_train_timesteps, schedule, **schedule_args)
        if prediction_type not in ("epsilon", "v_prediction"):
            raise ValueError("Argument `prediction_type` must be a member of PNDMPredictionType")
        if not skip_prk_steps:
            raise NotImplementedError("the hot path constructs PNDMScheduler(skip_prk_steps=True) only")
        self.prediction_type = prediction_type
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.init_noise_sigma = 1.0
        self.pndm_order = 4
        self.skip_prk_steps = skip_prk_steps
        self.steps_offset = steps_offset
        self.timestep_list = timestep_list
        self.cur_model_output = 0
        self.counter = 0
        self.cur_sample = None
        self.ets: list = []
        self.set_timesteps(num_train_timesteps)

    def set_timesteps(self, num_inference_steps: int, device=None) -> None:
        if num_inference_steps > self.num_train_timesteps:
            raise ValueError("num_inference_steps cannot exceed num_train_timesteps")
        self.num_inference_steps = num_inference_steps
        step_ratio = self.num_train_timesteps // self.num_inference_steps
        ts = (np.arange(0, num_inference_steps) * step_ratio).round().astype(np.int64)
        ts += self.steps_offset
        if self.timestep_list == "diffusers":
            plms = np.concatenate([ts[:-1], ts[-2:-1], ts[-1:]])[::-1].copy()
        else:
            plms = ts[::-1].copy()
        self.timesteps = torch.from_numpy(plms.astype(np.int64))
        # MONAI-Generative re-derives num_inference_steps from the list length (a no-op for its 100-entry
        # list); the diffusers variant keeps the REQUESTED count for the step ratio (its 101-entry list
        # repeats one timestep), so the ratio is stored separately.
        self._step_ratio = step_ratio
        self.num_inference_steps = len(self.timesteps)
        self.ets = []
        self.counter = 0
        self.cur_sample = None

    def step(self, model_output: torch.Tensor, timestep, sample: torch.Tensor):
        return self.step_plms(model_output, int(timestep), sample), None

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
            self.cur_sample = sample
        elif len(self.ets) == 1 and self.counter == 1:
            model_output = (model_output + self.ets[-1]) / 2
            sample = self.cur_sample
            self.cur_sample = None
        elif len(self.ets) == 2:
            model_output = (3 * self.ets[-1] - self.ets[-2]) / 2
        elif len(self.ets) == 3:
            model_output = (23 * self.ets[-1] - 16 * self.ets[-2] + 5 * self.ets[-3]) / 12
        else:
            model_output = (1 / 24) * (55 * self.ets[-1] - 59 * self.ets[-2] + 37 * self.ets[-3] - 9 * self.ets[-4])

        prev_sample = self._get_prev_sample(sample, timestep, prev_timestep, model_output)
        self.counter += 1
        return prev_sample

    def _get_prev_sample(self, sample, timestep: int, prev_timestep: int, model_output):
        a_t = self.alphas_cumprod[timestep]
        a_p = self.alphas_cumprod[prev_timestep] if prev_timestep >= 0 else self.final_alpha_cumprod
        b_t = 1 - a_t
        b_p = 1 - a_p
        if self.prediction_type == "v_prediction":
            model_output = (a_t ** 0.5) * model_output + (b_t ** 0.5) * sample
        sample_coeff = (a_p / a_t) ** 0.5
        denom = a_t * b_p ** 0.5 + (a_t * b_t * a_p) ** 0.5
        return sample_coeff * sample - (a_p - a_t) * model_output / denom


def ddim_step_closed_form(sample, eps, a_t: float, a_p: float):
    """Deterministic DDIM step (eta=0) in float64 -- the independent anchor for
    _get_prev_sample ('PNDM transfer == DDIM closed form')."""
    x0 = (sample - math.sqrt(1 - a_t) * eps) / math.sqrt(a_t)
    return math.sqrt(a_p) * x0 + math.sqrt(1 - a_p) * eps
