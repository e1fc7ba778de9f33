"""A/B of the DMA-fed 1x1 convolution: split-f16 MFMA loop (default) vs the f32 MFMA loop (DDPM_CONV1X1_F16X3=0).

    python tools/conv1x1_ab.py [--batch 256]

Times the five skip-connection 1x1 convolutions of the `small` UNet's up path at the bench batch (hipEvents,
50 launches each) in two subprocesses -- the switch is read once per process."""
import argparse
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

LAYERS = [  # C1, C2, Cout, H  (concat skip of the up path; bias + identity-free residual add as in ResnetBlock)
    (256, 256, 256, 8), (256, 256, 256, 8), (256, 256, 256, 16), (256, 128, 256, 16), (256, 128, 128, 32),
    (128, 128, 128, 32),
]


def child(batch):
    import torch
    from ddpm_ood_amd import ops

    dev = torch.device("cuda:0")
    tot = 0.0
    for C1, C2, Cout, H in LAYERS:
        x = torch.randn(batch, C1, H, H, device=dev)
        x2 = torch.randn(batch, C2, H, H, device=dev)
        w = torch.randn(Cout, C1 + C2, 1, 1, device=dev) / (C1 + C2) ** 0.5
        b = torch.randn(Cout, device=dev)
        r = torch.randn(batch, Cout, H, H, device=dev)
        wh = ops.pack_conv1x1_h_weight(w) if os.environ.get("AB_PRESPLIT", "1") == "1" else None
        pk = ops.pack_conv_weight(w)
        for _ in range(5):
            y = ops.conv(x, w, b, x2=x2, residual=r, packed=pk, wino44h=wh)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            y = ops.conv(x, w, b, x2=x2, residual=r, packed=pk, wino44h=wh)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 50
        npix = batch * H * H
        flops = 2.0 * npix * Cout * (C1 + C2)
        byts = 4.0 * npix * (C1 + C2 + 2 * Cout)
        ref = torch.nn.functional.conv2d(torch.cat([x[:4], x2[:4]], 1).double(), w.double(), b.double()) + r[:4].double()
        err = ((y[:4].double() - ref).abs().max() / ref.abs().max()).item()
        print(f"  {C1}+{C2}->{Cout} @{H}x{H}: {ms * 1e3:8.1f} us  {flops / ms / 1e9:6.1f} TFLOP/s  "
              f"{byts / ms / 1e9:6.2f} TB/s  max rel err vs f64 {err:.2e}")
        tot += ms
    print(f"  sum {tot:.3f} ms")
    # fused q / k / v behind GroupNorm: small @8x8, big @64x64 / 32x32 / 16x16 (B = batch / 64: cfg4's 16 at --batch 1024)
    for C, Cout, H in [(256, 768, 8), (256, 768, 64), (512, 1536, 32), (768, 2304, 16)]:
        b_ = batch if H == 8 else max(batch // 64, 1)
        x = torch.randn(b_, C, H, H, device=dev)
        w = torch.randn(Cout, C, 1, 1, device=dev) / C ** 0.5
        b = torch.randn(Cout, device=dev)
        gs, gh = ops.gn_scale_shift(x, torch.ones(C, device=dev), torch.zeros(C, device=dev), 32, 1e-6)
        wh = ops.pack_conv1x1_h_weight(w) if os.environ.get("AB_PRESPLIT", "1") == "1" else None
        pk = ops.pack_conv_weight(w)
        for _ in range(5):
            y = ops.conv(x, w, b, gscale=gs, gshift=gh, packed=pk, wino44h=wh)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            y = ops.conv(x, w, b, gscale=gs, gshift=gh, packed=pk, wino44h=wh)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 50
        print(f"  GN + {C}->{Cout} @{H}x{H} B={b_}: {ms * 1e3:8.1f} us  {2.0 * b_ * H * H * Cout * C / ms / 1e9:6.1f} TFLOP/s")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--child", action="store_true")
    a = ap.parse_args()
    if a.child:
        child(a.batch)
    else:
        for v, pre in (("0", "0"), ("1", "0"), ("1", "1")):
            print(f"DDPM_CONV1X1_F16X3={v} pre-split weights={pre}", flush=True)
            env = dict(os.environ, DDPM_CONV1X1_F16X3=v, AB_PRESPLIT=pre)
            subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "--batch", str(a.batch)], env=env, check=True)
