"""ddpm_ood_amd -- MI355X-native multi-t DDPM reconstruction path of marksgraham/ddpm-ood.

Only what the hot path needs: the HIP kernels and their C ABI
(``csrc/``, ``include/ddpm_ood_hip.h``) and the host-side mirrors of the reference's call
surface (UNet, PNDM scheduler, perceptual loss, stage-1 passthrough, trainer, scorer).
"""

from .unet import DiffusionModelUNet  # noqa: F401
from .scheduler import PNDMScheduler, DDPMScheduler  # noqa: F401
from .perceptual import PerceptualLoss  # noqa: F401
from .vqvae import VQVAE, PassthroughVQVAE  # noqa: F401

__all__ = ["DiffusionModelUNet", "PNDMScheduler", "DDPMScheduler", "PerceptualLoss", "VQVAE", "PassthroughVQVAE"]
