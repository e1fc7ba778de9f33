"""Stage-1 model surface used by the reconstruction path.

``PassthroughVQVAE`` is the identity stand-in the reference uses for pixel-space DDPMs
(/root/reference/src/networks/passthrough_vqvae.py:4-26, selected at
/root/reference/src/trainers/base.py:62-64).  The real 3D VQ-VAE encode / decode is the LDM
row of SURVEY.md 8(f) ("next" rank f-1) and is not built yet: asking for it fails loudly.
"""

import torch


class PassthroughVQVAE(torch.nn.Module):
    """This fake VQ-VAE just returns inputs."""

    def __init__(self):
        super().__init__()
        self.latent_channels = 1

    def reconstruct(self, x: torch.Tensor) -> torch.Tensor:
        return x

    def decode(self, x: torch.Tensor) -> torch.Tensor:
        return x

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return x

    def encode_stage_2_inputs(self, x: torch.Tensor) -> torch.Tensor:
        return x

    def decode_stage_2_outputs(self, x: torch.Tensor) -> torch.Tensor:
        return x


class VQVAE(torch.nn.Module):
    def __init__(self, **config):
        super().__init__()
        raise NotImplementedError(
            "VQ-VAE (latent-diffusion) reconstruction is SURVEY.md 8(f) row f-1 and is not built in this round; "
            "run pixel-space models without --vqvae_checkpoint")
