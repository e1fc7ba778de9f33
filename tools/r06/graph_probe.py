"""Probe: the native training step captured into one HIP graph (torch.cuda.CUDAGraph) and replayed -- how much of a step is launch
overhead?  python tools/r06/graph_probe.py [batch]"""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch  # noqa: E402

from ddpm_ood_amd import DiffusionModelUNet  # noqa: E402
from ddpm_ood_amd import train_ops as T  # noqa: E402
from ddpm_ood_amd.synthetic import random_state_dict  # noqa: E402
from ddpm_ood_amd.train_native import NativeUNetStep  # noqa: E402
from ddpm_ood_amd.trainer import MODEL_CONFIGS  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
sd = random_state_dict("small", 1, seed=1)
x = torch.rand(B, 1, 32, 32, device=dev)
t = torch.randint(0, 1000, (B,)).to(dev)
m = DiffusionModelUNet(2, 1, 1, **MODEL_CONFIGS["small"])
m.load_state_dict(sd)
m = m.to(dev).train()
with torch.no_grad():
    st = NativeUNetStep(m)
    noise = T.randn((B, 1, 32, 32), dev, 1, 1)

    def step():
        st.loss_and_grads(x, t, noise)

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / 10
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        step()
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        step()
    torch.cuda.synchronize()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t0) / 10
print(f"batch {B}: loss_and_grads eager {eager * 1e3:.2f} ms, graph replay {graph * 1e3:.2f} ms")
