"""Training-step throughput of the `small` UNet on one MI355X: the native step (train_native.NativeUNetStep -- hand-written HIP forward /
backward / Adam) beside the ATen route (PyTorch-ROCm autograd over MIOpen / rocBLAS, DDPM_TRAIN_NATIVE=0), same weights, same batch.
    python tools/train_step_bench.py [batch] [steps] [native|aten|amp|both|all]
(amp: the ATen route under fp16 autocast + GradScaler -- the reference's own training arithmetic, ddpm_trainer.py:96-109.)
Row f-3 (/root/reference/src/trainers/ddpm_trainer.py:78-109, base.py:156).  `rocprofv3 --kernel-trace --stats -- python
tools/train_step_bench.py 64 5 native` is the profile committed as profiles/r06_train_native_kernel_trace_stats.csv."""
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

from ddpm_ood_amd import DiffusionModelUNet  # noqa: E402
from ddpm_ood_amd import train_ops as T  # noqa: E402
from ddpm_ood_amd.synthetic import random_state_dict  # noqa: E402
from ddpm_ood_amd.train import unet_forward_torch  # noqa: E402
from ddpm_ood_amd.train_native import NativeUNetStep  # noqa: E402
from ddpm_ood_amd.trainer import MODEL_CONFIGS  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
which = sys.argv[3] if len(sys.argv) > 3 else "both"
dev = torch.device("cuda:0")
sd = random_state_dict("small", 1, seed=1)
x = torch.rand(B, 1, 32, 32, device=dev)
t = torch.randint(0, 1000, (B,)).to(dev)


def model():
    m = DiffusionModelUNet(2, 1, 1, **MODEL_CONFIGS["small"])
    m.load_state_dict(sd)
    return m.to(dev).train()


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


if which in ("native", "both", "all"):
    with torch.no_grad():
        st = NativeUNetStep(model())
        k = [0]

        def native_step():
            k[0] += 1
            noise = T.randn((B, 1, 32, 32), dev, 1, k[0])
            st.loss_and_grads(x, t, noise)
            st.adam_step()

        dt = timed(native_step)
        if os.environ.get("DDPM_PROF_SHAPES"):  # one step under the library's own per-launch timers, one row per layer shape
            import ctypes

            from ddpm_ood_amd import _lib

            lib = _lib.load()
            lib.ddpm_prof_enable(1)
            native_step()
            torch.cuda.synchronize()
            lib.ddpm_prof_enable(0)
            buf = ctypes.create_string_buffer(1 << 20)
            lib.ddpm_prof_report(buf, len(buf))
            import json

            prof = json.loads(buf.value.decode())
            tot = sum(v["ms"] for v in prof.values())
            print(f"{'kernel':64s} {'launch':>6s} {'ms':>9s} {'%':>6s} {'TFLOP/s':>8s} {'GB/s':>8s}")
            for kname, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"]):
                print(f"{kname:64s} {v['launches']:6d} {v['ms']:9.3f} {100 * v['ms'] / tot:6.1f} "
                      f"{v['flops'] / v['ms'] / 1e9 if v['ms'] else 0:8.1f} {v['bytes'] / v['ms'] / 1e6 if v['ms'] else 0:8.1f}")
            print(f"{'sum':64s} {'':6s} {tot:9.3f}")
    print(f"native: batch {B}: {dt * 1e3:.2f} ms per step = {B / dt:.0f} images/s")
if which in ("aten", "both", "all"):
    m = model()
    for p in m.parameters():
        p.requires_grad_(True)
    opt = torch.optim.Adam(m.parameters(), lr=2.5e-5)

    def aten_step():
        noise = torch.randn(B, 1, 32, 32, device=dev)
        opt.zero_grad(set_to_none=True)
        loss = torch.nn.functional.mse_loss(unet_forward_torch(m, x, t), noise)
        loss.backward()
        opt.step()

    dt = timed(aten_step)
    print(f"aten:   batch {B}: {dt * 1e3:.2f} ms per step = {B / dt:.0f} images/s")
if which in ("amp", "all"):
    m = model()
    for p in m.parameters():
        p.requires_grad_(True)
    opt = torch.optim.Adam(m.parameters(), lr=2.5e-5)
    scaler = torch.amp.GradScaler("cuda")

    def amp_step():
        noise = torch.randn(B, 1, 32, 32, device=dev)
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.float16):
            loss = torch.nn.functional.mse_loss(unet_forward_torch(m, x, t).float(), noise)
        scaler.scale(loss).backward()
        scaler.step(opt)
        scaler.update()

    dt = timed(amp_step)
    print(f"aten fp16 autocast: batch {B}: {dt * 1e3:.2f} ms per step = {B / dt:.0f} images/s")
