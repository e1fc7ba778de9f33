"""CPU fp32 oracle for the multi-t DDPM reconstruction hot path.

THIS PACKAGE IS TEST INFRASTRUCTURE, NOT PRODUCT.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it. Nothing under ``ddpm_ood_amd/`` imports it; the product path runs on the HIP
library and fails loudly when that library is missing.

PARITY UNPINNED.  The arithmetic of this path does not live in the reference tree: it is
in third-party packages that are neither vendored nor installed here
(``monai-generative`` [no version pin, 0.2.x API line by the call-site kwargs],
``monai==1.2.0``, ``lpips==0.1.4``, ``torch==1.13.1`` -- /root/reference/requirements.txt:1-5)
and the reference ships no tests, golden vectors or fixtures
(SURVEY.md section 4 / 8c).  The reference itself cannot be imported in this container
(src/trainers/base.py:8-10 dies on ``import generative``).  This oracle therefore
restates the *published* algorithms of those packages in plain
PyTorch CPU fp32 and anchors on
  * the reference's own call sites (every module cites the file:line it follows),
  * closed-form known-answer tests that need no third-party code
    (PNDM transfer == DDIM step, add_noise closed form, alpha-bar table values,
    timestep embedding at t=0, torch.nn.functional per-op ground truth),
  * pandas / scikit-learn for the Z-score / AUROC stage.
Fidelity to the real MONAI-Generative / lpips wheels remains an assumption until someone
diffs against an installation that has them.
"""

from .scheduler import PNDMScheduler, DDPMScheduler, make_betas  # noqa: F401
from .unet import DiffusionModelUNet  # noqa: F401
from .lpips import LPIPSAlex, PerceptualLoss  # noqa: F401
from .vqvae import PassthroughVQVAE  # noqa: F401
from .reconstruct import get_scores  # noqa: F401
from .ood import z_scores_and_auroc  # noqa: F401
