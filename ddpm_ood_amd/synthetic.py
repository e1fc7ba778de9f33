"""Synthetic checkpoints for benchmarks / tests (no trained weights can be fetched here).

A freshly constructed DiffusionModelUNet is identically zero (conv2 of every ResnetBlock and the
output conv are zero-initialised, SURVEY finding 12), so parity would pass vacuously: the
zero-initialised convolutions are overwritten with N(0, 0.02) and GroupNorm affines are
perturbed.  File layout = the reference's ``save_checkpoint`` dict
(/root/reference/src/trainers/base.py:166-187).
"""

from __future__ import annotations

from pathlib import Path

import torch

from .trainer import MODEL_CONFIGS
from .unet import DiffusionModelUNet


def random_state_dict(model_type: str = "small", channels: int = 1, spatial_dims: int = 2, seed: int = 1,
                      config: dict = None):
    torch.manual_seed(seed)
    cfg = config or MODEL_CONFIGS[model_type]
    m = DiffusionModelUNet(spatial_dims=spatial_dims, in_channels=channels, out_channels=channels,
                           with_conditioning=False, **cfg)
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, v in m.state_dict().items():
        v = v.clone()
        if float(v.abs().max()) == 0.0:  # zero-initialised conv
            v = torch.randn(v.shape, generator=g) * 0.02
        elif ("norm" in k or k.startswith("out.0")) and k.endswith("weight"):
            v = v + 0.1 * torch.randn(v.shape, generator=g)
        elif ("norm" in k or k.startswith("out.0")) and k.endswith("bias"):
            v = 0.1 * torch.randn(v.shape, generator=g)
        sd[k] = v.float()
    return sd


def write_checkpoint(run_dir, model_type: str = "small", channels: int = 1, seed: int = 1, config: dict = None):
    run_dir = Path(run_dir)
    run_dir.mkdir(parents=True, exist_ok=True)
    sd = random_state_dict(model_type, channels, 2, seed, config)
    torch.save({"epoch": 0, "global_step": 0, "model_state_dict": sd, "optimizer_state_dict": {},
                "best_loss": 1000}, run_dir / "checkpoint.pth")
    return sd
