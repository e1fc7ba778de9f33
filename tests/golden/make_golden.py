"""Generate the committed golden fixtures with the CPU oracle (run here, in the build container).

    python tests/golden/make_golden.py

The reference itself cannot produce vectors (its third-party dependencies are not installable
here and it ships no tests -- SURVEY.md 8c), so these fixtures pin the ORACLE, not the
reference: ``-m "not gpu"`` tests check that the oracle still reproduces them, ``-m gpu`` tests
check the HIP path against them.  Weights are not stored (70 MB): they are regenerated from
``ddpm_ood_amd.synthetic.random_state_dict(seed)``; a checksum of the regenerated state_dict is
stored so that RNG drift is detected instead of silently changing the expected outputs.

Files (all small):
  schedule.npz          alpha-bar tables, PLMS timesteps / start points, PLMS coefficients
  unet_forward.npz      x, t -> eps for the `small` UNet, B = 2, 32x32x1, seed-1 weights
  ops.npz               one I/O pair for each fused op (conv prologue/epilogue combos, GN, attention)
  trajectory_rows.csv   get_scores on 3 x 4 synthetic images, k = 64 (t = 10, 650; stale PLMS history)
  ood_scores.json       Z-score / AUROC of those rows through the pandas / sklearn scorer
  cli_flags.json        flag names / defaults of the reference CLI (data parsed from its argparse calls)
  train_cli_flags.json  the same for the reference's train_ddpm.py
"""

import ast
import hashlib
import json
import math
import sys
from pathlib import Path

import numpy as np
import pandas as pd
import torch
import torch.nn.functional as F

HERE = Path(__file__).resolve().parent
ROOT = HERE.parents[1]
sys.path.insert(0, str(ROOT))

import oracle  # noqa: E402
from ddpm_ood_amd.data import get_data_loader  # noqa: E402
from ddpm_ood_amd.synthetic import random_state_dict  # noqa: E402
from ddpm_ood_amd.trainer import MODEL_CONFIGS, batch_noise  # noqa: E402

SCHED = dict(schedule="scaled_linear_beta", beta_start=0.0015, beta_end=0.0195)


def state_dict_digest(sd) -> str:
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(k.encode())
        h.update(sd[k].contiguous().numpy().tobytes())
    return h.hexdigest()


def schedule():
    out = {}
    for name, kw in (("scaled", SCHED), ("linear", dict(schedule="linear_beta", beta_start=1e-4, beta_end=2e-2))):
        s = oracle.PNDMScheduler(num_train_timesteps=1000, skip_prk_steps=True, **kw)
        s.set_timesteps(100)
        out[f"{name}_alphas_cumprod"] = s.alphas_cumprod.numpy()
        out[f"{name}_timesteps"] = s.timesteps.numpy()
        coefs = []
        for t in range(0, 1000, 10):
            a_t, a_p = s.alphas_cumprod[t], s.alphas_cumprod[t - 10] if t >= 10 else s.final_alpha_cumprod
            coefs.append([float((a_p / a_t) ** 0.5), float(a_p - a_t),
                          float(a_t * (1 - a_p) ** 0.5 + (a_t * (1 - a_t) * a_p) ** 0.5)])
        out[f"{name}_plms_coef"] = np.asarray(coefs, dtype=np.float32)
    for k in (1, 2, 4, 16, 64):
        out[f"start_points_k{k}"] = reversed(s.timesteps)[1::k].numpy()
    np.savez_compressed(HERE / "schedule.npz", **out)


def unet_forward():
    sd = random_state_dict("small", 1, seed=1)
    m = oracle.DiffusionModelUNet(2, 1, 1, **MODEL_CONFIGS["small"]).eval()
    m.load_state_dict(sd)
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 1, 32, 32, generator=g)
    t = torch.tensor([10, 650])
    with torch.no_grad():
        y = m(x, timesteps=t)
    np.savez_compressed(HERE / "unet_forward.npz", x=x.numpy(), t=t.numpy(), eps=y.numpy(),
                        state_dict_sha256=np.array(state_dict_digest(sd)))
    return sd, m


def ops():
    g = torch.Generator().manual_seed(21)
    out = {}
    # resnet-style conv: GN(32) + SiLU prologue over a virtual concat (12 channels / group straddles
    # the seam), bias, per-image channel add
    x1, x2 = torch.randn(2, 64, 8, 8, generator=g), torch.randn(2, 32, 8, 8, generator=g) * 2 + 1
    gamma, beta = torch.randn(96, generator=g) * 0.2 + 1, torch.randn(96, generator=g) * 0.2
    w, b = torch.randn(128, 96, 3, 3, generator=g) / math.sqrt(96 * 9), torch.randn(128, generator=g)
    temb = torch.randn(2, 128, generator=g)
    y = F.conv2d(F.silu(F.group_norm(torch.cat([x1, x2], 1), 32, gamma, beta, 1e-6)), w, b, padding=1)
    out.update(res_x1=x1, res_x2=x2, res_gamma=gamma, res_beta=beta, res_w=w, res_b=b, res_temb=temb,
               res_y=y + temb[:, :, None, None])
    # upsample + conv and stride-2 conv
    xu = torch.randn(1, 8, 4, 4, generator=g)
    wu, bu = torch.randn(128, 8, 3, 3, generator=g) / math.sqrt(72), torch.randn(128, generator=g)
    out.update(up_x=xu, up_w=wu, up_b=bu,
               up_y=F.conv2d(F.interpolate(xu, scale_factor=2.0, mode="nearest"), wu, bu, padding=1),
               down_y=F.conv2d(xu, wu, bu, stride=2, padding=1))
    # attention, 1 head of 256 over 64 tokens
    qkv = torch.randn(1, 768, 64, generator=g)
    q, k, v = qkv.split(256, dim=1)
    s = torch.einsum("bdi,bdj->bij", q, k) / 16.0
    out.update(att_qkv=qkv, att_y=torch.einsum("bij,bdj->bdi", s.softmax(-1), v))
    np.savez_compressed(HERE / "ops.npz", **{k: v.numpy() for k, v in out.items()})


def trajectory(sd, model):
    pl = oracle.PerceptualLoss(dimensions=2, include_pixel_loss=False, is_fake_3d=False, lpips_normalize=True)
    frames = []
    specs = {"val": "synthetic:blobs:n=4:seed=10", "in": "synthetic:blobs:n=4:seed=11",
             "out": "synthetic:noise:n=4:seed=12"}
    for name, ids in specs.items():
        loader = get_data_loader(ids, batch_size=4, is_grayscale=True)
        rows = oracle.get_scores(loader, name, 64, model=model, vqvae=oracle.PassthroughVQVAE(), perceptual=pl,
                                 noise_fn=lambda batch, t, shape: batch_noise(2, batch["index"], t, shape),
                                 beta_schedule=SCHED["schedule"], beta_start=SCHED["beta_start"],
                                 beta_end=SCHED["beta_end"])
        frames.append(pd.DataFrame(rows))
    df = pd.concat(frames, ignore_index=True)
    df.to_csv(HERE / "trajectory_rows.csv", float_format="%.9e")
    val, inn, out = (df[df["type"] == t] for t in ("val", "in", "out"))
    zdf, zmean, auc = oracle.z_scores_and_auroc(val, inn, out)
    json.dump({"auroc_mse": auc,
               "z_score_mse": zdf["z_score_mse"].tolist(),
               "z_score_perceptual_difference": zdf["z_score_perceptual_difference"].tolist(),
               "specs": specs, "lpips_seed": 1234, "noise_seed": 2, "weight_seed": 1},
              open(HERE / "ood_scores.json", "w"), indent=1)


def cli_flags(ref_name="reconstruct.py", out_name="cli_flags.json"):
    """Flag names / defaults of a reference CLI, as data (no source text is kept)."""
    ref = Path("/root/reference") / ref_name
    if not ref.exists():
        return
    flags = {}
    for node in ast.walk(ast.parse(ref.read_text())):
        if isinstance(node, ast.Call) and getattr(node.func, "attr", "") == "add_argument":
            name = node.args[0].value.lstrip("-")
            kw = {k.arg: k.value for k in node.keywords}
            default = ast.literal_eval(kw["default"]) if "default" in kw else None
            typ = getattr(kw.get("type"), "id", None) or getattr(kw.get("type"), "attr", None)
            flags[name] = {"default": default, "type": typ}
    json.dump(flags, open(HERE / out_name, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    torch.set_num_threads(8)
    schedule()
    ops()
    cli_flags()
    cli_flags("train_ddpm.py", "train_cli_flags.json")
    sd, model = unet_forward()
    trajectory(sd, model)
    print("golden fixtures written to", HERE)
