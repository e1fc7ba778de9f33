"""Times ddpm_conv_wgrad_f32 on the 3x3 shapes of the `small` UNet's training step (batch 256 by default), the split-f16 form and
(DDPM_WGRAD_F16X3=0) the fp32-MFMA form:  python tools/wgrad_ab.py [batch]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

from ddpm_ood_amd import train_ops as T  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")
tot = 0.0
for cin, cout, hw, n in ((128, 128, 32, 4), (256, 256, 16, 4), (256, 256, 32, 1), (384, 128, 32, 1), (256, 256, 8, 8), (512, 256, 16, 1)):
    a = torch.randn(B, cin, hw, hw, device=dev)
    dy = torch.randn(B, cout, hw, hw, device=dev) * 1e-3
    for _ in range(2):
        T.conv_wgrad(a, dy, 3, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        T.conv_wgrad(a, dy, 3, 1)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    tf = 2.0 * B * hw * hw * cin * cout * 9 / ms / 1e9
    tot += n * ms
    print(f"{cin:4d}->{cout:4d}@{hw:2d}x{hw:2d} B{B}: {ms * 1e3:8.1f} us  {tf:6.1f} TFLOP/s")
print(f"weighted by the step's counts: {tot:.3f} ms")
