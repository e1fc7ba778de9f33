"""Oracle restatement of the stage-1 model surface used on the hot path.

TEST INFRASTRUCTURE (see oracle/__init__.py).
``PassthroughVQVAE`` follows /root/reference/src/networks/passthrough_vqvae.py:4-26
(identity for pixel-space DDPMs; selected at /root/reference/src/trainers/base.py:62-64).
"""

import torch


class PassthroughVQVAE(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.latent_channels = 1

    def reconstruct(self, x):
        return x

    def decode(self, x):
        return x

    def forward(self, x):
        return x

    def encode_stage_2_inputs(self, x):
        return x

    def decode_stage_2_outputs(self, x):
        return x
