"""CPU: pin the oracle.  The reference has no tests or golden vectors (parity
unpinned), so the anchors are (i) closed-form known answers that need no third-party code,
(ii) torch.nn.functional as per-op ground truth, (iii) the committed golden fixtures produced
by tests/golden/make_golden.py."""

import json
import math
from pathlib import Path

import numpy as np
import pandas as pd
import pytest
import torch

import oracle
from oracle.scheduler import ddim_step_closed_form

G = Path(__file__).resolve().parent / "golden"
SCHED = dict(schedule="scaled_linear_beta", beta_start=0.0015, beta_end=0.0195)


def _pndm(**kw):
    s = oracle.PNDMScheduler(num_train_timesteps=1000, skip_prk_steps=True, **{**SCHED, **kw})
    s.set_timesteps(100)
    return s


def test_alpha_bar_table_known_values():
    """SURVEY 8c [MEASURED fp64]: abar[0]=0.998500, [10]=0.983195, [650]=0.027863, [970]=0.000249, [990]=0.000170."""
    s = _pndm()
    b = np.linspace(0.0015 ** 0.5, 0.0195 ** 0.5, 1000, dtype=np.float64) ** 2
    ref = np.cumprod(1 - b)
    assert np.abs(s.alphas_cumprod.numpy() - ref).max() < 2e-6
    for t, v in ((0, 0.998500), (10, 0.983195), (650, 0.027863), (970, 0.000249), (990, 0.000170)):
        assert abs(float(s.alphas_cumprod[t]) - v) < 1.5e-6


def test_timesteps_start_points_and_forward_counts():
    """BASELINE.md workload arithmetic: k -> (reconstructions, UNet forwards) per image."""
    s = _pndm()
    assert s.timesteps.tolist() == list(range(990, -1, -10))
    want = {64: (2, 68), 16: (7, 350), 4: (25, 1250), 2: (50, 2550), 1: (99, 5049)}
    for k, (n_rec, n_fwd) in want.items():
        starts = reversed(s.timesteps)[1::k]
        assert len(starts) == n_rec
        assert sum(int((s.timesteps <= t).sum()) for t in starts) == n_fwd
    assert reversed(s.timesteps)[1::64].tolist() == [10, 650]
    assert reversed(s.timesteps)[1::4].tolist() == list(range(10, 990, 40))
    d = _pndm(timestep_list="diffusers")
    assert len(d.timesteps) == 101 and d.timesteps[:3].tolist() == [990, 980, 980]  # Q9 variant


def test_pndm_transfer_equals_ddim_closed_form():
    s = _pndm()
    g = torch.Generator().manual_seed(0)
    x = torch.randn(4, 1, 8, 8, generator=g, dtype=torch.float64)
    e = torch.randn(4, 1, 8, 8, generator=g, dtype=torch.float64)
    s.alphas_cumprod = s.alphas_cumprod.double()
    for t in (990, 650, 10):
        got = s._get_prev_sample(x, t, t - 10, e)
        ref = ddim_step_closed_form(x, e, float(s.alphas_cumprod[t]), float(s.alphas_cumprod[t - 10]))
        assert (got - ref).abs().max() < 1e-12


def test_add_noise_closed_form():
    s = _pndm()
    x0, n = torch.rand(3, 1, 4, 4), torch.randn(3, 1, 4, 4)
    t = torch.tensor([10, 650, 990])
    ac = s.alphas_cumprod[t].reshape(3, 1, 1, 1)
    assert torch.allclose(s.add_noise(x0, n, t), ac.sqrt() * x0 + (1 - ac).sqrt() * n, atol=1e-7)


def test_plms_state_machine_heun_start_and_stale_history():
    """Constant eps: every multistep formula collapses to eps, so the trajectory equals repeated DDIM
    steps; the second call of a fresh scheduler re-does the FIRST transfer (Heun-style start)."""
    s = _pndm()
    x0 = torch.randn(2, 1, 4, 4, dtype=torch.float64)
    s.alphas_cumprod = s.alphas_cumprod.double()
    e = torch.full_like(x0, 0.3)
    x1, _ = s.step(e, 650, x0)
    assert s.counter == 1 and len(s.ets) == 1
    x2, _ = s.step(e, 640, x1)  # counter == 1: goes back to cur_sample and repeats 650 -> 640
    assert torch.allclose(x2, x1) and s.counter == 2 and len(s.ets) == 1
    # Q3: a new trajectory on the SAME scheduler starts with counter >= 2 and stale ets
    y0 = torch.randn_like(x0)
    e_new = torch.full_like(x0, -0.5)
    y1, _ = s.step(e_new, 990, y0)
    mixed = (3 * e_new - e) / 2  # 2-term formula mixing the stale eps of the previous trajectory
    ref = ddim_step_closed_form(y0, mixed, float(s.alphas_cumprod[990]), float(s.alphas_cumprod[980]))
    assert torch.allclose(y1, ref, atol=1e-12)
    s.set_timesteps(100)
    assert s.counter == 0 and s.ets == []


def test_timestep_embedding_known_answer():
    from oracle.unet import get_timestep_embedding

    e = get_timestep_embedding(torch.tensor([0, 7]), 128)
    assert torch.equal(e[0], torch.cat([torch.ones(64), torch.zeros(64)]))
    assert abs(float(e[1, 0]) - math.cos(7.0)) < 1e-6 and abs(float(e[1, 64]) - math.sin(7.0)) < 1e-6


def test_unet_param_counts_and_zero_init():
    from ddpm_ood_amd.trainer import MODEL_CONFIGS

    m = oracle.DiffusionModelUNet(2, 1, 1, **MODEL_CONFIGS["small"])
    assert sum(p.numel() for p in m.parameters()) == 17_709_953
    y = m(torch.randn(1, 1, 32, 32), timesteps=torch.tensor([5]))
    assert float(y.abs().max()) == 0.0  # finding 12: a fresh UNet is identically zero
    m3 = oracle.DiffusionModelUNet(3, 128, 128, **MODEL_CONFIGS["small"])
    assert sum(p.numel() for p in m3.parameters()) == 47_493_888


def test_golden_schedule():
    z = np.load(G / "schedule.npz")
    s = _pndm()
    assert np.array_equal(z["scaled_alphas_cumprod"], s.alphas_cumprod.numpy())
    assert np.array_equal(z["scaled_timesteps"], s.timesteps.numpy())
    assert z["start_points_k64"].tolist() == [10, 650]


def test_golden_unet_forward_and_weight_digest():
    import sys
    sys.path.insert(0, str(G))
    from make_golden import state_dict_digest
    from ddpm_ood_amd.synthetic import random_state_dict
    from ddpm_ood_amd.trainer import MODEL_CONFIGS

    z = np.load(G / "unet_forward.npz")
    sd = random_state_dict("small", 1, seed=1)
    assert state_dict_digest(sd) == str(z["state_dict_sha256"]), "synthetic weights drifted (RNG change?)"
    m = oracle.DiffusionModelUNet(2, 1, 1, **MODEL_CONFIGS["small"]).eval()
    m.load_state_dict(sd)
    with torch.no_grad():
        y = m(torch.from_numpy(z["x"]), timesteps=torch.from_numpy(z["t"]))
    assert np.abs(y.numpy() - z["eps"]).max() < 2e-5  # oneDNN blocking may differ between hosts
    assert np.abs(z["eps"]).max() > 0.05


def test_golden_ood_scores_via_independent_numpy():
    """Z-score / AUROC of the golden rows recomputed with numpy only (no pandas groupby, no sklearn)."""
    df = pd.read_csv(G / "trajectory_rows.csv", index_col=0)
    gold = json.load(open(G / "ood_scores.json"))
    val, inn, out = (df[df["type"] == t] for t in ("val", "in", "out"))
    _, _, auc = oracle.z_scores_and_auroc(val, inn, out)
    assert abs(auc - gold["auroc_mse"]) < 1e-12
    ts = sorted(val["t"].unique())
    mu = {t: val[val.t == t]["mse"].to_numpy().mean() for t in ts}
    sd = {t: val[val.t == t]["mse"].to_numpy().std(ddof=1) for t in ts}
    score = lambda d: {f: np.mean([(r.mse - mu[r.t]) / sd[r.t] for r in d[d.filename == f].itertuples()])
                       for f in d.filename.unique()}
    si, so = score(inn), score(out)
    pairs = [(a > b) + 0.5 * (a == b) for b in si.values() for a in so.values()]
    assert abs(np.mean(pairs) - auc) < 1e-12  # AUROC = P(score_out > score_in)


@pytest.mark.parametrize("case", ["k64_b128", "cfg3"])
def test_committed_oracle_rows_are_this_oracles_output(case):
    """tests/golden/rows_<case>.csv (the oracle side of the expensive -m gpu parity tests, make_golden_rows.py) against the
    oracle run live on the first image of every set: a stale fixture -- or an oracle that changed without the fixture being
    regenerated -- fails here, on the CPU, before any GPU test trusts the file."""
    import sys

    sys.path.insert(0, str(Path(__file__).resolve().parent))
    from parity_util import golden_rows, live_oracle_pins_fixture

    spec, rows = golden_rows(case)
    live_oracle_pins_fixture(case, dict(spec, live=1), rows)
