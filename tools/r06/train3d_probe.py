"""Probe: native vs ATen training step of the 3-D latent UNet (the LDM configuration's stage 2): python tools/r06/train3d_probe.py [B] [S]"""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch  # noqa: E402

from ddpm_ood_amd import DiffusionModelUNet  # noqa: E402
from ddpm_ood_amd.synthetic import random_state_dict  # noqa: E402
from ddpm_ood_amd.train import unet_forward_torch  # noqa: E402
from ddpm_ood_amd.train_native import NativeUNetStep  # noqa: E402
from ddpm_ood_amd.trainer import MODEL_CONFIGS  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
S = int(sys.argv[2]) if len(sys.argv) > 2 else 16
C = 128
dev = torch.device("cuda:0")
sd = random_state_dict("small", C, spatial_dims=3, seed=1)
x = torch.rand(B, C, S, S, S, device=dev)
t = torch.randint(0, 1000, (B,)).to(dev)
noise = torch.randn(B, C, S, S, S, device=dev)


def model():
    m = DiffusionModelUNet(3, C, C, **MODEL_CONFIGS["small"])
    m.load_state_dict(sd)
    return m.to(dev).train()


def timed(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


with torch.no_grad():
    st = NativeUNetStep(model())

    def native():
        st.loss_and_grads(x, t, noise)
        st.adam_step()

    dt = timed(native)
print(f"native 3-D: batch {B} x {S}^3: {dt * 1e3:.1f} ms per step = {B / dt:.1f} volumes/s (scale {st.loss_scale:g}, retries {st.overflow_retries})")
m = model()
for p in m.parameters():
    p.requires_grad_(True)
opt = torch.optim.Adam(m.parameters(), lr=2.5e-5)


def aten():
    opt.zero_grad(set_to_none=True)
    loss = torch.nn.functional.mse_loss(unet_forward_torch(m, x, t), noise)
    loss.backward()
    opt.step()


dt = timed(aten)
print(f"aten   3-D: batch {B} x {S}^3: {dt * 1e3:.1f} ms per step = {B / dt:.1f} volumes/s")
