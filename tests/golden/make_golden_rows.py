"""Generate the ORACLE side of the expensive workload-level parity tests as committed fixtures.

    python tests/golden/make_golden_rows.py [case ...]        (build container, CPU only; ~10 min for all cases)

The `-m gpu` suite used to spend two thirds of its time waiting for the CPU oracle to price hundreds of image
trajectories (one test: 384 images x 68 UNet forwards = 206 TFLOP of oneDNN on the GPU box's host).  The oracle side of
those tests is a pure function of seeds -- weights `random_state_dict(seed)`, images `synthetic:...:seed=`, per-image noise
`batch_noise(seed, index, t)`, LPIPS weights `LPIPS(seed)` -- so it is computed once here and committed as rows
(`rows_<case>.csv`, one line per (set, image, t): what /root/reference/src/trainers/reconstruct.py:192-204 appends), with the
specification and the digests of the regenerated weights in `rows_<case>.json`.  The tests

  * compare the HIP path with the committed rows (all images), and
  * run the oracle LIVE on the first few images of every set and hold it to the committed rows (1e-5) and to the HIP rows,
    so a stale or foreign fixture cannot pass: the fixture is pinned to the oracle code of the day.

The per-image result does not depend on the batch it rides in (noise is a function of the image index, the PLMS history
is per element), which is what makes "first few images" a valid sample of the same computation.
"""

import argparse
import json
import sys
import time
from pathlib import Path

import pandas as pd
import torch

HERE = Path(__file__).resolve().parent
ROOT = HERE.parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
sys.path.insert(0, str(HERE))

from make_golden import state_dict_digest  # noqa: E402

CASES = {
    # tests/test_gpu_dispatch.py::test_k64_trajectories_at_batch_128_absolute_z
    "k64_b128": dict(channels=1, model_type="small", skip=64, batch=128, live=4,
                     sets={"val": "synthetic:blobs:n=128:seed=10", "in": "synthetic:blobs:n=128:seed=11",
                           "out": "synthetic:speckle:n=128:seed=12:mix=10"}),
    # tests/test_gpu_configs.py::test_cfg2_all_25_chained_t_starts
    "cfg2_25t": dict(channels=1, model_type="small", skip=4, batch=3, live=1,
                     sets={"val": "synthetic:blobs:n=2:seed=10", "in": "synthetic:blobs:n=2:seed=11",
                           "out": "synthetic:speckle:n=1:seed=12:mix=10"}),
    # tests/test_gpu_dispatch.py::test_cfg2_25_chained_t_starts_at_batch_128_absolute_z -- the workload bench.py times (BASELINE
    # configs[1]: all 25 chained t-starts, 1 250 forwards per image) at a chip-filling dispatch: the HIP side runs each set's 128
    # images as ONE batch, the oracle prices the first `oracle_n` of them (a per-image result does not depend on its batch)
    "cfg2_25t_b128": dict(channels=1, model_type="small", skip=4, batch=128, live=1, live_sets=["val"], oracle_n=16,
                          sets={"val": "synthetic:blobs:n=128:seed=10", "in": "synthetic:blobs:n=128:seed=11",
                                "out": "synthetic:speckle:n=128:seed=12:mix=10"}),
    # tests/test_gpu_dispatch.py::test_cfg4_long_chains_at_the_benchmarked_batch_absolute_z -- BASELINE configs[3] (`big` UNet,
    # 64x64x3, k = 2) on LONG chains AT bench.py's cfg4 BATCH OF 16: t_start in {10, 250, 490} of the chained k = 2 list (2 + 26 + 50
    # forwards per image through 16 attention blocks of up to 4 096 tokens; `t_start_subset` keeps the list's order, the PLMS
    # history a trajectory inherits is the previous kept one's).  The HIP side runs every set's 16 images as ONE batch; the oracle
    # prices all 16 validation images (so the Z bound is the ABSOLUTE one) and the first two of the other sets (1 560 `big` CPU
    # forwards).  Round 5's `cfg4_t490` (batch 2, two validation images) is superseded by this case.
    "cfg4_b16": dict(channels=3, model_type="big", skip=2, batch=16, live=1, live_sets=["in"], t_start_subset=[10, 250, 490],
                     oracle_n={"val": 16, "in": 2, "out": 2},
                     sets={"val": "synthetic:blobs:n=16:channels=3:size=64:seed=30", "in": "synthetic:blobs:n=16:channels=3:size=64:seed=31",
                           "out": "synthetic:speckle:n=16:channels=3:size=64:seed=32:mix=10"}),
    # tests/test_gpu_dispatch.py::test_cfg4_longest_chain_t990_in_a_batch_of_16 -- the LAST t-start of cfg4's k = 2 list: t = 990,
    # 100 `big` forwards on one trajectory (the first image of a batch of 16)
    "cfg4_t990": dict(channels=3, model_type="big", skip=2, batch=16, live=1, live_sets=["in"], t_start_subset=[990], oracle_n=1,
                      sets={"in": "synthetic:blobs:n=16:channels=3:size=64:seed=31"}),
    # tests/test_gpu_configs.py::test_cfg5_z_scores_at_unet_batch_16_on_an_unspread_codebook -- BASELINE configs[4] (LDM: README VQ-VAE
    # + 3-D `small` UNet over 128-channel latents) as a val / in / out experiment: 16 volumes of 64^3 per set (latents [128, 4, 4, 4]),
    # every set ONE batch of 16 through the UNet, k = 64 (t in {10, 650}), on an UN-spread codebook (N(0, 1) rows as initialised:
    # nearest-code near-ties are possible).  The rows carry the oracle's re-quantised codes per (volume, t) -- column `codes` -- so
    # that the test can tell a volume whose decode saw a code flip from one that did not.
    "cfg5_z64": dict(channels=128, model_type="small", spatial_dims=3, skip=64, batch=16, live=1, live_sets=["in"], vq_seed=3,
                     sets={"val": "synthetic:blobs3d:n=16:size=64:seed=40", "in": "synthetic:blobs3d:n=16:size=64:seed=41",
                           "out": "synthetic:blobs3d:n=16:size=64:seed=42"}),
    # tests/test_gpu_configs.py::test_cfg3_three_channel_two_ood_sets_and_sensitive_auroc
    "cfg3": dict(channels=3, model_type="small", skip=64, batch=32, live=2,
                 sets={"val": "synthetic:blobs:n=16:channels=3:seed=10", "in": "synthetic:blobs:n=32:channels=3:seed=11",
                       "SVHN": "synthetic:speckle:n=32:channels=3:seed=13:mix=5:name=SVHN",
                       "CelebA": "synthetic:blobs:n=32:channels=3:seed=12:name=CelebA"}),
}


def oracle_n(case: dict, sname: str):
    """Images of a set the oracle prices (None: all): one number for every set, or a dict per set."""
    n = case.get("oracle_n")
    return n.get(sname) if isinstance(n, dict) else n


def set_type(name: str) -> str:
    return name if name in ("val", "in") else "out"


def oracle_rows(case: dict, name: str, ids: str, first_n=None) -> pd.DataFrame:
    """The oracle's rows for one set (optionally only its first images), inputs exactly as tests/parity_util.oracle_scores."""
    import oracle
    from ddpm_ood_amd.data import get_data_loader
    from ddpm_ood_amd.perceptual import LPIPS
    from ddpm_ood_amd.synthetic import random_state_dict
    from ddpm_ood_amd.trainer import MODEL_CONFIGS, batch_noise
    from parity_util import SCHED

    if case.get("spatial_dims", 2) == 3:
        return oracle_rows_ldm(case, name, ids, first_n)
    c = case["channels"]
    key = (c, case["model_type"])
    if key not in _MODELS:
        sd = random_state_dict(case["model_type"], c, seed=1)
        m = oracle.DiffusionModelUNet(2, c, c, **MODEL_CONFIGS[case["model_type"]]).eval()
        m.load_state_dict(sd)
        pl = oracle.PerceptualLoss(dimensions=2, include_pixel_loss=False, is_fake_3d=False, lpips_normalize=True)
        lp = LPIPS().state_dict()  # the product's default (seeded) LPIPS weights: a CPU-side constant
        pl.perceptual_function.load_state_dict(lp)
        _MODELS[key] = (m, pl, state_dict_digest(sd), state_dict_digest(lp))
    m, pl, _, _ = _MODELS[key]
    loader = get_data_loader(ids, batch_size=case["batch"], is_grayscale=c == 1, spatial_dimension=2, first_n=first_n)
    return pd.DataFrame(oracle.get_scores(
        loader, set_type(name), case["skip"], model=m, vqvae=oracle.PassthroughVQVAE(), perceptual=pl,
        noise_fn=lambda batch, t, shape: batch_noise(2, batch["index"], t, shape), t_start_subset=case.get("t_start_subset"),
        **SCHED))


def oracle_vqvae(case: dict):
    """The README VQ-VAE (/root/reference/README.md:153-158) with seeded default initialisation, conditioned by
    ddpm_ood_amd.synthetic.condition_vqvae_state_dict (latents of the codebook's scale, reconstructions that span [0, 1] and
    depend on the codes -- a freshly initialised VQ-VAE decodes every latent to the same clamped-away constant); the codebook
    stays as initialised (N(0, 1) rows): NOT spread, so nearest-code near-ties are as likely as the geometry makes them."""
    from oracle.vqvae import VQVAE as OracleVQVAE
    from ddpm_ood_amd.synthetic import condition_vqvae_state_dict
    from parity_util import VQ_README

    torch.manual_seed(case["vq_seed"])
    vq = OracleVQVAE(**VQ_README).eval()
    vq.load_state_dict(condition_vqvae_state_dict(vq.state_dict()))
    return vq


def oracle_rows_ldm(case: dict, name: str, ids: str, first_n=None) -> pd.DataFrame:
    """3-D latent-diffusion case (BASELINE configs[4]): encode -> PLMS over the latent -> re-quantise + decode -> MSE, 2.5-D LPIPS
    (/root/reference/src/trainers/reconstruct.py:124-187).  Adds the column `codes`: the code indices the oracle's
    decode_stage_2_outputs chose for that (volume, t), space-separated."""
    import oracle
    from ddpm_ood_amd.data import get_data_loader
    from ddpm_ood_amd.perceptual import LPIPS
    from ddpm_ood_amd.synthetic import random_state_dict
    from ddpm_ood_amd.trainer import MODEL_CONFIGS, batch_noise
    from parity_util import SCHED

    c = case["channels"]
    key = (c, case["model_type"], 3)
    if key not in _MODELS:
        sd = random_state_dict(case["model_type"], c, spatial_dims=3, seed=1)
        m = oracle.DiffusionModelUNet(3, c, c, **MODEL_CONFIGS[case["model_type"]]).eval()
        m.load_state_dict(sd)
        pl = oracle.PerceptualLoss(dimensions=3, include_pixel_loss=False, is_fake_3d=True, lpips_normalize=True)
        lp = LPIPS().state_dict()
        pl.perceptual_function.load_state_dict(lp)
        vq = oracle_vqvae(case)
        _MODELS[key] = (m, pl, state_dict_digest(sd), state_dict_digest(lp), vq, state_dict_digest(vq.state_dict()))
    m, pl, _, _, vq, _ = _MODELS[key]
    codes = []
    plain_decode = vq.decode_stage_2_outputs

    def recording_decode(z):
        idx = vq.quantizer.quantizer.quantize(z)  # what decode_stage_2_outputs is about to choose
        codes.extend(" ".join(str(int(v)) for v in row.reshape(-1)) for row in idx)
        return plain_decode(z)

    vq.decode_stage_2_outputs = recording_decode
    try:
        loader = get_data_loader(ids, batch_size=case["batch"], is_grayscale=True, spatial_dimension=3, first_n=first_n)
        df = pd.DataFrame(oracle.get_scores(
            loader, set_type(name), case["skip"], model=m, vqvae=vq, perceptual=pl, spatial_dimension=3,
            noise_fn=lambda batch, t, shape: batch_noise(2, batch["index"], t, shape), t_start_subset=case.get("t_start_subset"),
            **SCHED))
    finally:
        del vq.decode_stage_2_outputs  # (the instance attribute: the class method is back)
    assert len(codes) == len(df)  # rows are appended per (batch, t_start, item): the order of the decode calls
    df["codes"] = codes
    return df


_MODELS = {}


def digests(case: dict):
    if case.get("spatial_dims", 2) == 3:
        e = _MODELS[(case["channels"], case["model_type"], 3)]
        return {"state_dict_sha256": e[2], "lpips_sha256": e[3], "vqvae_sha256": e[5]}
    key = (case["channels"], case["model_type"])
    return {"state_dict_sha256": _MODELS[key][2], "lpips_sha256": _MODELS[key][3]}


def load(case_name: str):
    """(spec, {set name: rows}) of a committed case."""
    spec = json.load(open(HERE / f"rows_{case_name}.json"))
    df = pd.read_csv(HERE / f"rows_{case_name}.csv", index_col=0)
    return spec, {n: df[df["set"] == n].drop(columns="set").reset_index(drop=True) for n in spec["sets"]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("cases", nargs="*", default=sorted(CASES))
    a = ap.parse_args()
    torch.set_num_threads(min(32, torch.get_num_threads()))
    for name in a.cases:
        case = CASES[name]
        frames = []
        t0 = time.time()
        for sname, ids in case["sets"].items():
            with torch.no_grad():
                df = oracle_rows(case, sname, ids, first_n=oracle_n(case, sname))
            df.insert(0, "set", sname)
            frames.append(df)
            print(f"{name}/{sname}: {len(df)} rows, {time.time() - t0:.0f} s", flush=True)
        pd.concat(frames, ignore_index=True).to_csv(HERE / f"rows_{name}.csv", float_format="%.9e")
        json.dump({**{k: v for k, v in case.items()}, **digests(case), "noise_seed": 2, "weight_seed": 1,
                   "generator": "tests/golden/make_golden_rows.py (CPU fp32 oracle, build container)"},
                  open(HERE / f"rows_{name}.json", "w"), indent=1)


if __name__ == "__main__":
    main()
