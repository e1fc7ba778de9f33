"""Development (not collected by pytest; lives under tests/ because it runs the CPU oracle, which only tests may import):
per-parameter gradient error of the native training step against CPU float64 autograd over the oracle UNet."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

import oracle  # noqa: E402
from ddpm_ood_amd import DiffusionModelUNet  # noqa: E402
from ddpm_ood_amd.synthetic import random_state_dict  # noqa: E402
from ddpm_ood_amd.train_native import NativeUNetStep  # noqa: E402
from ddpm_ood_amd.trainer import MODEL_CONFIGS  # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
sd = random_state_dict("small", 1, seed=1)
g = torch.Generator().manual_seed(11)
x = torch.rand(B, 1, 32, 32, generator=g)
t = torch.randint(0, 1000, (B,), generator=g)
noise = torch.randn(B, 1, 32, 32, generator=g)
ref = oracle.DiffusionModelUNet(2, 1, 1, **MODEL_CONFIGS["small"]).double().train()
ref.load_state_dict({k: v.double() for k, v in sd.items()})
loss_r = torch.nn.functional.mse_loss(ref(x.double(), timesteps=t), noise.double())
loss_r.backward()
hip = DiffusionModelUNet(2, 1, 1, **MODEL_CONFIGS["small"])
hip.load_state_dict(sd)
hip = hip.to(dev).train()
with torch.no_grad():
    st = NativeUNetStep(hip)
    loss_h = st.loss_and_grads(x.to(dev), t.to(dev), noise.to(dev))
print("loss", float(loss_h.cpu()), float(loss_r))
pr, ph = dict(ref.named_parameters()), dict(hip.named_parameters())
gmax = max(float(p.grad.abs().max()) for p in pr.values() if p.grad is not None)
bad = 0
for k in pr:
    gr, gh = pr[k].grad, ph[k].grad
    if gr is None:
        continue
    rel = float((gh.cpu().double() - gr).abs().max() / max(float(gr.abs().max()), 1e-5 * gmax))
    flag = "" if rel < 1e-4 else "   <<<<<<"
    bad += rel >= 1e-4
    print(f"{k:60s} {rel:.2e}{flag}")
print("bad", bad)
