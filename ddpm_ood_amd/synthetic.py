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


def condition_vqvae_state_dict(sd: dict, enc_gain: float = 256.0, dec_gain: float = 64.0) -> dict:
    """A freshly initialised VQ-VAE (generative's VQVAE, /root/reference/src/trainers/base.py:44-61) is degenerate as a test
    object: its encoder output is ~1e-2 against codebook rows of norm ~11 (every position of every volume lands on the same one
    or two codes) and its decoder output is a bias-driven constant below zero (clamp -> 0: the reconstruction, hence both scores,
    ignore the codes).  This turns such a state_dict into one that behaves like a trained model in the two respects the
    reconstruction path is sensitive to, by a fixed rule (no data, no RNG): every convolution bias is zeroed, the encoder's last
    convolution is scaled by `enc_gain` (latents of the codebook's scale: tens of distinct codes per volume), the decoder's
    last transposed convolution is scaled by `dec_gain` around a bias of 0.5 (reconstructions that span [0, 1] and move with the
    codes).  Key names: `encoder.blocks.<i>.conv.*` / `decoder.blocks.<i>.conv.*` (the oracle's and the product's VQVAE)."""
    out = {k: v.clone() for k, v in sd.items()}
    enc_last = max(int(k.split(".")[2]) for k in out if k.startswith("encoder.blocks."))
    dec_last = max(int(k.split(".")[2]) for k in out if k.startswith("decoder.blocks."))
    for k in out:
        if (k.startswith("encoder.") or k.startswith("decoder.")) and k.endswith(".bias"):
            out[k].zero_()
    out[f"encoder.blocks.{enc_last}.conv.weight"] *= enc_gain
    out[f"decoder.blocks.{dec_last}.conv.weight"] *= dec_gain
    out[f"decoder.blocks.{dec_last}.conv.bias"].fill_(0.5)
    return out


def write_checkpoint(run_dir, model_type: str = "small", channels: int = 1, seed: int = 1, config: dict = None):
    run_dir = Path(run_dir)
    run_dir.mkdir(parents=True, exist_ok=True)
    sd = random_state_dict(model_type, channels, 2, seed, config)
    torch.save({"epoch": 0, "global_step": 0, "model_state_dict": sd, "optimizer_state_dict": {},
                "best_loss": 1000}, run_dir / "checkpoint.pth")
    return sd
