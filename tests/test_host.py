"""CPU: host-side logic of the product package (no GPU, no HIP compute calls)."""

import json
import sys
from pathlib import Path

import numpy as np
import pandas as pd
import pytest
import torch

import oracle

G = Path(__file__).resolve().parent / "golden"
SCHED = dict(schedule="scaled_linear_beta", beta_start=0.0015, beta_end=0.0195)


def test_cli_flags_match_reference():
    import reconstruct as cli

    ref = json.load(open(G / "cli_flags.json"))
    assert len(ref) == 34
    args = cli.parse_args([])
    for name, spec in ref.items():
        assert hasattr(args, name), f"--{name} missing"
        assert getattr(args, name) == spec["default"], name
    a = cli.parse_args(["--image_roi", "(-1, 20)", "--latent_pad", "(1,1,1,1)", "--first_n", "16",
                        "--inference_skip_factor", "64", "--is_grayscale", "1"])
    assert a.image_roi == (-1, 20) and a.latent_pad == (1, 1, 1, 1) and a.first_n == "16"
    import ood_detection as ocli

    o = ocli.parse_args(["--model_name", "fashionmnist", "--output_dir", "x"])
    assert (o.max_t, o.min_t, o.t_skip, o.seed) == (1000, 0, 1, 2)


def test_product_scheduler_host_logic_equals_oracle():
    from ddpm_ood_amd.scheduler import PNDMScheduler, noise_schedule

    for name in ("linear", "linear_beta", "scaled_linear", "scaled_linear_beta", "sigmoid_beta", "cosine"):
        assert torch.equal(noise_schedule(name, 1000, 1e-4, 2e-2), oracle.make_betas(name, 1000, 1e-4, 2e-2))
    with pytest.raises(ValueError):
        noise_schedule("quadratic", 10)
    h = PNDMScheduler(num_train_timesteps=1000, skip_prk_steps=True, **SCHED)
    o = oracle.PNDMScheduler(num_train_timesteps=1000, skip_prk_steps=True, **SCHED)
    h.set_timesteps(100)
    o.set_timesteps(100)
    assert torch.equal(h.timesteps, o.timesteps) and torch.equal(h.alphas_cumprod, o.alphas_cumprod)
    z = np.load(G / "schedule.npz")
    got = np.asarray([h.plms_coefficients(t, t - 10)[:3] for t in range(0, 1000, 10)], dtype=np.float32)
    assert np.array_equal(got, z["scaled_plms_coef"])
    # the scheduler surface the reference exercises: reversed(), mask indexing, iteration of 0-d tensors
    starts = reversed(h.timesteps)[1::64]
    assert [int(t) for t in starts] == [10, 650]
    assert [int(s) for s in h.timesteps[h.timesteps <= starts[0]]] == [10, 0]
    h.betas = h.betas * 1.0  # assignable tables (SNR shift, reconstruct.py:106-117)
    # --timestep_list=diffusers: 101 entries (the second timestep repeated), but the PLMS update still steps by
    # 1000 // 100 = 10 (diffusers keeps the requested step count for the ratio), not 1000 // 101 = 9
    for cls in (PNDMScheduler, oracle.PNDMScheduler):
        d = cls(num_train_timesteps=1000, skip_prk_steps=True, timestep_list="diffusers", **SCHED)
        d.set_timesteps(100)
        assert [int(t) for t in d.timesteps[:4]] == [990, 980, 980, 970] and len(d.timesteps) == 101
        assert d._step_ratio == 10
    m = PNDMScheduler(num_train_timesteps=1000, skip_prk_steps=True, **SCHED)
    m.set_timesteps(100)
    assert m._step_ratio == 10 and len(m.timesteps) == 100


def test_snr_shift_matches_oracle():
    from ddpm_ood_amd.scheduler import PNDMScheduler
    from ddpm_ood_amd.trainer import snr_shift_tables
    from oracle.reconstruct import snr_shift_tables as o_shift

    h = PNDMScheduler(num_train_timesteps=1000, skip_prk_steps=True, **SCHED)
    o = oracle.PNDMScheduler(num_train_timesteps=1000, skip_prk_steps=True, **SCHED)
    snr_shift_tables(h, 0.25)
    o_shift(o, 0.25)
    assert torch.equal(h.alphas_cumprod, o.alphas_cumprod) and torch.equal(h.betas, o.betas)


def test_data_ingest_and_partition(tmp_path):
    from ddpm_ood_amd.data import get_data_loader, partition, scale_intensity

    paths = []
    rng = np.random.default_rng(0)
    for i in range(5):
        p = tmp_path / f"img_{i}.npy"
        np.save(p, rng.integers(0, 255, (28, 28), dtype=np.uint8))
        paths.append(str(p))
    (tmp_path / "FashionMNIST_test.csv").write_text(",".join(paths) + "\n")
    ld = get_data_loader(str(tmp_path / "FashionMNIST_test.csv"), batch_size=2, is_grayscale=True, image_size=32,
                         first_n=4)
    batches = list(ld)
    assert [b["image"].shape for b in batches] == [(2, 1, 32, 32), (2, 1, 32, 32)]
    assert batches[0]["image_meta_dict"]["filename_or_obj"][1] == paths[1]
    x = torch.cat([b["image"] for b in batches])
    assert float(x.min()) == 0.0 and float(x.max()) == 1.0  # per-image min-max (Q13)
    flip = get_data_loader(str(tmp_path / "FashionMNIST_test.csv"), 4, is_grayscale=True, image_size=32, first_n=4,
                           add_vflip=True)
    assert torch.equal(next(iter(flip))["image"], torch.flip(x, dims=(2,)))
    assert scale_intensity(torch.ones(1, 1, 2, 2)).abs().max() == 0  # constant image -> 0
    assert partition(10, 1, 4) == [1, 5, 9] and sorted(sum((partition(10, r, 4) for r in range(4)), [])) == list(range(10))
    with pytest.raises(FileNotFoundError):
        get_data_loader(str(tmp_path / "absent.csv"), 2)


def test_noise_is_a_pure_function_of_seed_image_t():
    from ddpm_ood_amd.trainer import batch_noise

    a = batch_noise(2, [3, 7], 650, (2, 1, 8, 8))
    b = batch_noise(2, [7], 650, (1, 1, 8, 8))
    assert torch.equal(a[1], b[0]) and not torch.equal(a[0], a[1])
    assert not torch.equal(batch_noise(3, [7], 650, (1, 1, 8, 8)), b)


def test_ood_scoring_matches_oracle_and_handles_duplicates(tmp_path):
    from ddpm_ood_amd import ood

    df = pd.read_csv(G / "trajectory_rows.csv", index_col=0)
    val, inn, out = (df[df["type"] == t] for t in ("val", "in", "out"))
    dup = pd.concat([inn, inn.iloc[:3].assign(mse=99.0)])  # DDP padding duplicates: keep-first (Q21)
    d1, m1, a1 = ood.score(val, dup, out)
    d2, m2, a2 = oracle.z_scores_and_auroc(val, inn, out)
    assert abs(a1 - a2) < 1e-12 and np.allclose(d1["z_score_mse"], d2["z_score_mse"], rtol=0, atol=1e-9)
    assert np.allclose(d1["z_score_perceptual_difference"], d2["z_score_perceptual_difference"], rtol=0, atol=1e-9)
    gold = json.load(open(G / "ood_scores.json"))
    assert np.allclose(d1["z_score_mse"], gold["z_score_mse"], atol=1e-9)
    # strict t window (ood_detection.py:59-61)
    d3, _, _ = ood.score(val, inn, out, max_t=650, min_t=0)
    assert set(d3["t"]) == {10}
    # end-to-end through the CLI-shaped main()
    run = tmp_path / "fashionmnist_x" / "ood"
    run.mkdir(parents=True)
    val.to_csv(run / "results_val.csv")
    inn.to_csv(run / "results_in.csv")
    for name in ood.out_datasets_for("fashionmnist_x"):
        out.to_csv(run / f"results_{name}.csv")
    import argparse
    res = ood.main(argparse.Namespace(output_dir=str(tmp_path), model_name="fashionmnist_x", max_t=1000, min_t=0))
    assert set(res) == {"MNIST", "FashionMNIST_vflip", "FashionMNIST_hflip"} and all(v == a2 for v in res.values())
    assert ood.count_model_evaluations([10, 650]) == 68
    with pytest.raises(ValueError):
        ood.out_datasets_for("imagenet")


def test_ood_scoring_nan_rows_product_and_oracle_agree():
    """The trainer writes NaN rows on a genuine fp32 overflow (as the reference would).  pandas' groupby mean / std skip them
    (ood_detection.py:152-174) and roc_auc_score refuses a NaN score (:206): product and oracle follow both."""
    from ddpm_ood_amd import ood

    df = pd.read_csv(G / "trajectory_rows.csv", index_col=0)
    val, inn, out = (df[df["type"] == t].copy() for t in ("val", "in", "out"))
    # one NaN row in val (skipped by the per-t mean / std) and one in `in` (skipped by that image's mean over t)
    val.iloc[0, val.columns.get_loc("mse")] = np.nan
    inn.iloc[1, inn.columns.get_loc("mse")] = np.nan
    d1, m1, a1 = ood.score(val, inn, out)
    d2, m2, a2 = oracle.z_scores_and_auroc(val, inn, out)
    assert abs(a1 - a2) < 1e-12
    z1, z2 = d1["z_score_mse"].to_numpy(), np.asarray(d2["z_score_mse"], dtype=np.float64)
    assert np.array_equal(np.isnan(z1), np.isnan(z2)) and np.isnan(z1).sum() == 1
    assert np.allclose(z1[~np.isnan(z1)], z2[~np.isnan(z2)], rtol=0, atol=1e-9)
    # an image whose every row is NaN has a NaN score: both sides raise, as sklearn does
    name = inn["filename"].iloc[0]
    inn.loc[inn["filename"] == name, "mse"] = np.nan
    with pytest.raises(ValueError):
        ood.score(val, inn, out)
    with pytest.raises(ValueError):
        oracle.z_scores_and_auroc(val, inn, out)


def test_reference_exceptions_are_kept(tmp_path):
    """FileNotFoundError / ValueError of the reference's setup (base.py:47-50,87-88; reconstruct.py:31-32)
    -- checked up to the point where a GPU becomes necessary."""
    from ddpm_ood_amd.trainer import dataset_stem
    from ddpm_ood_amd.vqvae import VQVAE, PassthroughVQVAE

    assert dataset_stem("/data/FashionMNIST_test.csv") == "FashionMNIST"
    assert dataset_stem("synthetic:noise:n=4:name=MNIST") == "MNIST"
    x = torch.ones(1, 1, 2, 2)
    p = PassthroughVQVAE()
    assert p.encode_stage_2_inputs(x) is x and p.decode_stage_2_outputs(x) is x
    with pytest.raises(ValueError):  # mismatched per-level tuples, as the MONAI-Generative ctor raises
        VQVAE(spatial_dims=3, in_channels=1, out_channels=1, num_channels=(8, 8), num_res_channels=(8,))
    if not torch.cuda.is_available():
        import reconstruct as cli
        from ddpm_ood_amd.trainer import Reconstruct

        with pytest.raises(RuntimeError, match="no CPU fallback"):
            Reconstruct(cli.parse_args(["--output_dir", str(tmp_path), "--model_name", "m"]))


def test_unet_holder_rejects_bad_configs():
    from ddpm_ood_amd import DiffusionModelUNet

    with pytest.raises(ValueError):
        DiffusionModelUNet(2, 1, 1, num_channels=(100, 128), attention_levels=(False, False), num_res_blocks=1)
    with pytest.raises(ValueError):
        DiffusionModelUNet(2, 1, 1, num_channels=(128, 128), attention_levels=(False,), num_res_blocks=1)
    with pytest.raises(NotImplementedError):
        DiffusionModelUNet(2, 1, 1, num_channels=(128,), attention_levels=(False,), with_conditioning=True)
    m = DiffusionModelUNet(2, 1, 1, num_channels=(128, 128), attention_levels=(False, True), num_res_blocks=1,
                           num_head_channels=128)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 1, 8, 8), timesteps=torch.zeros(1, dtype=torch.long))


def test_vqvae_restatement_and_3d_ingest(tmp_path):
    """VQ-VAE surface of the LDM configuration (base.py:44-61, reconstruct.py:124,166): same state_dict keys in
    oracle and product, nearest-code quantiser against a brute-force search, encode -> 8^3 x 128 latents."""
    from oracle.vqvae import VQVAE as OV
    from ddpm_ood_amd.vqvae import VQVAE as PV
    from ddpm_ood_amd.data import get_data_loader

    cfg = dict(spatial_dims=3, in_channels=1, out_channels=1, num_channels=(8, 16), num_res_layers=1,
               num_res_channels=(8, 16), downsample_parameters=((2, 4, 1, 1), (2, 4, 1, 1)),
               upsample_parameters=((2, 4, 1, 1, 0), (2, 4, 1, 1, 0)), num_embeddings=32, embedding_dim=12)
    torch.manual_seed(0)
    o, p = OV(**cfg).eval(), PV(**cfg).eval()
    assert list(o.state_dict()) == list(p.state_dict())
    assert "quantizer.quantizer.embedding.weight" in o.state_dict()
    p.load_state_dict(o.state_dict())
    ld = get_data_loader("synthetic:blobs3d:n=2:size=16:seed=1", batch_size=2, is_grayscale=True, spatial_dimension=3)
    x = next(iter(ld))["image"]
    assert x.shape == (2, 1, 16, 16, 16) and float(x.min()) == 0.0 and float(x.max()) == 1.0
    with torch.no_grad():
        z = o.encode(x)
        q = o.encode_stage_2_inputs(x)
        assert z.shape == (2, 12, 4, 4, 4)
        e = o.quantizer.quantizer.embedding.weight
        flat = z.movedim(1, -1).reshape(-1, 12)
        brute = ((flat[:, None, :] - e[None]) ** 2).sum(-1).argmin(1)
        assert torch.equal(o.index_quantize(x).reshape(-1), brute)
        assert torch.allclose(q.movedim(1, -1).reshape(-1, 12), e[brute], atol=1e-6)
        # the product VQ-VAE has one backend, the HIP library: CPU tensors are refused (tests/test_gpu_ops.py compares the two)
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            p.encode_stage_2_inputs(x)
        assert o.decode_stage_2_outputs(q).shape == x.shape
    np.save(tmp_path / "vol.npy", np.random.default_rng(0).random((20, 18, 16)).astype(np.float32))
    (tmp_path / "Task01_test.csv").write_text(str(tmp_path / "vol.npy") + "\n")
    ld = get_data_loader(str(tmp_path / "Task01_test.csv"), 1, is_grayscale=True, spatial_dimension=3,
                         image_roi=(16, 16, -1), image_size=8)
    assert next(iter(ld))["image"].shape == (1, 1, 8, 8, 8)


def _write_nifti(path, data, slope=0.0, inter=0.0, big_endian=False):
    """Minimal single-file NIfTI-1 writer for the ingest test (header fields as in the NIfTI-1 standard)."""
    import gzip
    import struct

    e = ">" if big_endian else "<"
    codes = {"uint8": 2, "int16": 4, "float32": 16, "float64": 64}
    hdr = bytearray(352)
    struct.pack_into(e + "i", hdr, 0, 348)
    dim = [data.ndim] + list(data.shape) + [1] * (7 - data.ndim)
    struct.pack_into(e + "8h", hdr, 40, *dim)
    struct.pack_into(e + "h", hdr, 70, codes[str(data.dtype)])
    struct.pack_into(e + "h", hdr, 72, data.dtype.itemsize * 8)
    struct.pack_into(e + "f", hdr, 108, 352.0)
    struct.pack_into(e + "ff", hdr, 112, slope, inter)
    hdr[344:348] = b"n+1\0"
    raw = bytes(hdr) + data.astype(data.dtype.newbyteorder(e)).tobytes(order="F")
    with (gzip.open(path, "wb") if str(path).endswith(".gz") else open(path, "wb")) as f:
        f.write(raw)


def test_nifti_ingest(tmp_path):
    """Row f-4: NIfTI-1 volumes listed in the reference's one-row CSV (get_train_and_val_dataloader.py:10-16,69-76)."""
    from ddpm_ood_amd.data import get_data_loader, read_nifti

    rng = np.random.default_rng(3)
    a = rng.integers(0, 2000, size=(12, 10, 8)).astype(np.int16)
    b = rng.random((12, 10, 8)).astype(np.float32)
    _write_nifti(tmp_path / "a.nii.gz", a, slope=0.5, inter=-3.0)
    _write_nifti(tmp_path / "b.nii", b, big_endian=True)
    np.testing.assert_allclose(read_nifti(tmp_path / "a.nii.gz"), a.astype(np.float32) * 0.5 - 3.0)
    np.testing.assert_array_equal(read_nifti(tmp_path / "b.nii"), b)
    (tmp_path / "ids.csv").write_text(f"{tmp_path / 'a.nii.gz'},{tmp_path / 'b.nii'}\n")
    loader = get_data_loader(str(tmp_path / "ids.csv"), batch_size=2, is_grayscale=True, spatial_dimension=3,
                             image_roi=[8, 8, 8])
    batch = next(iter(loader))
    assert batch["image"].shape == (2, 1, 8, 8, 8)
    assert float(batch["image"].min()) == 0.0 and float(batch["image"].max()) == 1.0   # ScaleIntensity per image
    want = torch.from_numpy(b)[2:10, 1:9, :]  # centre crop, then min-max
    want = (want - want.min()) / (want.max() - want.min())
    assert torch.allclose(batch["image"][1, 0], want, atol=1e-6)
    (tmp_path / "bad.nii").write_bytes(b"\0" * 400)
    with pytest.raises(ValueError):
        read_nifti(tmp_path / "bad.nii")
    (tmp_path / "ids2.csv").write_text(f"{tmp_path / 'x.jpg'}\n")
    with pytest.raises(NotImplementedError):
        get_data_loader(str(tmp_path / "ids2.csv"), batch_size=1)


def test_ingest_follows_the_reference_transform_order(tmp_path):
    """the Decathlon command line.  (1) BraTS NIfTI volumes are X x Y x Z x 4 -- channel LAST on
    disk; EnsureChannelFirst + x[0, None] must pick modality 0, not slice the first spatial axis.  (2) crop / resize
    run per image before batching, so out-sets whose volumes differ in native shape can be batched.  (3) ambiguous
    layouts raise instead of producing 'valid' garbage.  (4) CenterSpatialCrop starts at s // 2 - r // 2."""
    from ddpm_ood_amd.data import center_crop, get_data_loader

    rng = np.random.default_rng(5)
    brats = rng.random((12, 12, 10, 4)).astype(np.float32)
    other = rng.random((14, 10, 9)).astype(np.float32)          # a Task0x out-set volume of another shape
    _write_nifti(tmp_path / "brats.nii.gz", brats)
    _write_nifti(tmp_path / "other.nii", other)
    (tmp_path / "ids.csv").write_text(f"{tmp_path / 'brats.nii.gz'},{tmp_path / 'other.nii'}\n")
    kw = dict(batch_size=2, is_grayscale=True, spatial_dimension=3)
    ld = get_data_loader(str(tmp_path / "ids.csv"), image_roi=[8, 8, 8], image_size=4, **kw)
    x = next(iter(ld))["image"]
    assert x.shape == (2, 1, 4, 4, 4)
    want = torch.from_numpy(brats[..., 0])[2:10, 2:10, 1:9]      # modality 0, centre crop 8^3
    want = torch.nn.functional.interpolate(want[None, None], size=(4, 4, 4), mode="area")[0, 0]
    want = (want - want.min()) / (want.max() - want.min())
    assert torch.allclose(x[0, 0], want, atol=1e-6)
    # without crop / resize the two volumes cannot share a batch -- the error says so -- but batch_size = 1 works
    with pytest.raises(RuntimeError, match="different shapes"):
        next(iter(get_data_loader(str(tmp_path / "ids.csv"), **kw)))
    shapes = [tuple(b["image"].shape) for b in get_data_loader(str(tmp_path / "ids.csv"), **{**kw, "batch_size": 1})]
    assert shapes == [(1, 1, 12, 12, 10), (1, 1, 14, 10, 9)]
    # ambiguous layouts
    np.save(tmp_path / "chw.npy", rng.random((4, 12, 12, 10)).astype(np.float32))
    (tmp_path / "amb.csv").write_text(f"{tmp_path / 'chw.npy'}\n")
    with pytest.raises(ValueError, match="ambiguous"):
        get_data_loader(str(tmp_path / "amb.csv"), **kw)
    np.save(tmp_path / "hw.npy", rng.random((8, 8)).astype(np.float32))
    (tmp_path / "rgb.csv").write_text(f"{tmp_path / 'hw.npy'}\n")
    with pytest.raises(ValueError, match="channel-first"):
        get_data_loader(str(tmp_path / "rgb.csv"), batch_size=1, is_grayscale=False)
    np.save(tmp_path / "rgb.npy", rng.integers(0, 255, (3, 8, 8), dtype=np.uint8))   # the reference's CIFAR layout
    (tmp_path / "rgb2.csv").write_text(f"{tmp_path / 'rgb.npy'}\n")
    assert next(iter(get_data_loader(str(tmp_path / "rgb2.csv"), batch_size=1)))["image"].shape == (1, 3, 8, 8)
    # monai CenterSpatialCrop: even size, odd roi -> [1, 4), not [0, 3)
    a = torch.arange(4.0).reshape(1, 4, 1)
    assert center_crop(a, (3, -1)).flatten().tolist() == [1.0, 2.0, 3.0]
    assert center_crop(a, (9, 1)).shape == (1, 4, 1)


def test_lpips_loads_a_state_dict_shaped_like_the_real_package():
    """lpips 0.1.4 registers each lin layer twice (lin0 ... lin4 AND lins.0 ... lins.4), so
    ``lpips.LPIPS(net="alex").state_dict()`` carries duplicate keys; the reference's PerceptualLoss prefixes
    everything with ``perceptual_function.``.  Both must load; missing / unknown / disagreeing keys must raise."""
    from ddpm_ood_amd.perceptual import LPIPS

    src = LPIPS(seed=7)
    real = {}
    for k, v in src.state_dict().items():
        real[k] = v.clone()
        if k.startswith("lins."):
            n = k.split(".")[1]
            real[f"lin{n}." + k.split(".", 2)[2]] = v.clone()
    assert len(real) == len(src.state_dict()) + 5 and "lin3.model.1.weight" in real
    dst = LPIPS(seed=1)
    assert not dst.pretrained
    dst.load_pretrained_state_dict(real)
    assert dst.pretrained and all(torch.equal(a, b) for a, b in zip(dst.state_dict().values(), src.state_dict().values()))
    dst2 = LPIPS(seed=2)
    dst2.load_pretrained_state_dict({"perceptual_function." + k: v for k, v in real.items()})
    assert torch.equal(dst2.lins[4].model[1].weight, src.lins[4].model[1].weight)
    only_alias = {k: v for k, v in real.items() if not k.startswith("lins.")}  # lpips' own alex.pth naming
    LPIPS(seed=3).load_pretrained_state_dict(only_alias)
    bad = dict(real)
    bad["lin2.model.1.weight"] = bad["lin2.model.1.weight"] + 1
    with pytest.raises(ValueError, match="disagree"):
        LPIPS().load_pretrained_state_dict(bad)
    with pytest.raises(KeyError, match="missing"):
        LPIPS().load_pretrained_state_dict({k: v for k, v in real.items() if "slice3" not in k})
    with pytest.raises(KeyError, match="unexpected"):
        LPIPS().load_pretrained_state_dict({**real, "net.slice6.0.weight": torch.zeros(1)})


def test_training_forward_and_cli(tmp_path):
    """Row f-3: the differentiable ATen forward used for training evaluates the same parameter holders as the HIP
    engine and equals the oracle UNet; train_ddpm.py keeps the reference's flags (/root/reference/train_ddpm.py:7-83,
    names / defaults extracted as data into tests/golden/train_cli_flags.json)."""
    import json

    import train_ddpm
    from ddpm_ood_amd import DiffusionModelUNet
    from ddpm_ood_amd.synthetic import random_state_dict
    from ddpm_ood_amd.train import unet_forward_torch

    cfg = dict(num_channels=(32, 64), attention_levels=(False, True), num_res_blocks=2, num_head_channels=32)
    sd = random_state_dict(config=cfg, channels=3, seed=3)
    m = DiffusionModelUNet(2, 3, 3, use_proj_attn=True, **cfg)
    m.load_state_dict(sd)
    o = oracle.DiffusionModelUNet(2, 3, 3, use_proj_attn=True, **cfg).eval()
    o.load_state_dict(sd)
    x, t = torch.randn(2, 3, 16, 16, generator=torch.Generator().manual_seed(1)), torch.tensor([10, 650])
    for p in m.parameters():
        p.requires_grad_(True)
    y = unet_forward_torch(m, x, t)
    with torch.no_grad():
        assert (y - o(x, timesteps=t)).abs().max() < 1e-5
    y.square().mean().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for n, p in m.named_parameters()
               if "proj_attn" in n or "conv" in n)
    flags = json.load(open(G / "train_cli_flags.json"))
    a = train_ddpm.parse_args(["--model_name", "m", "--output_dir", "o"])
    assert set(flags) == set(vars(a)) - {"amp"}, set(flags) ^ set(vars(a))  # --amp: this repo's extension (default: fp32)
    assert a.amp == 0
    for k, v in flags.items():
        assert getattr(a, k) == v["default"] or k in ("model_name", "output_dir"), k


def test_bench_roofline_object_is_the_executed_mfma_fraction():
    """`roofline.frac` must be EXECUTED MFMA work over the 157.3 TFLOP/s peak (never > 1); the
    direct-conv-equivalent rate of a Winograd kernel lives in its own key; `traffic` is labelled as a builder-side pass."""
    import bench

    prof = {"conv3x3_wino_gn_silu": {"launches": 10, "ms": 3.0, "flops": 10 * 66.8e9, "bytes": 10 * 159e6},
            "conv3x3_wino44_gn_silu": {"launches": 12, "ms": 4.0, "flops": 12 * 100e9, "bytes": 12 * 200e6},
            "conv3x3_wino_up": {"launches": 2, "ms": 1.0, "flops": 2 * 154.6e9, "bytes": 1e9},
            "attention": {"launches": 4, "ms": 0.1, "flops": 4 * 0.1e9, "bytes": 1e8},
            "conv3d_wino": {"launches": 1, "ms": 2.0, "flops": 3.4e11, "bytes": 1e9},
            "conv3d_k3": {"launches": 1, "ms": 1.0, "flops": 1e11, "bytes": 1e9},
            "conv3d_wino44": {"launches": 1, "ms": 2.0, "flops": 3.4e11, "bytes": 1e9},
            "gn_scale_shift": {"launches": 27, "ms": 0.5, "flops": 1e9, "bytes": 2e9}}
    hbm = bench.rooflines_of({"conv1x1_dma": {"launches": 7, "ms": 1.0, "flops": 7 * 26e9, "bytes": 7 * 600e6},
                              "conv1x1_dma_gn": {"launches": 1, "ms": 0.5, "flops": 1e10, "bytes": 3e8}})["conv1x1_dma"]
    # the split-f16 1x1 is priced against HBM: algorithmic bytes per second over 8 TB/s
    assert hbm["bound"] == "hbm" and hbm["unit"] == "GB/s" and hbm["launches_timed"] == 8
    for kk in ("bound", "achieved", "unit", "frac", "algorithmic_equiv_tflops", "avg_launch_ms", "launches_timed",
               "ms_in_sample", "algorithmic_GBps"):  # the keys the bench line copies from every class
        assert kk in hbm and kk in bench.rooflines_of({"attention": {"launches": 1, "ms": 1.0, "flops": 1e9, "bytes": 1e6}})["attention"]
    assert abs(hbm["achieved"] - (7 * 600e6 + 3e8) / 1.5e-3 / 1e9) < 0.1 and abs(hbm["frac"] - hbm["achieved"] / 8000.0) < 1e-3
    att = bench.rooflines_of({"attention": {"launches": 2, "ms": 2.0, "flops": 2 * 140e9, "bytes": 1e8}})["attention"]
    # split-f16 attention: three f16 MFMAs per fp32 product against the dense f16 peak
    assert att["peak"] == 2500.0 and abs(att["achieved"] - 3 * 140.0) < 0.01 and abs(att["frac"] - 420.0 / 2500.0) < 1e-3
    assert abs(att["algorithmic_equiv_tflops"] - 140.0) < 0.01
    r = bench.rooflines_of(prof)
    h = bench.rooflines_of({"conv3x3_wino44h_gn_silu": {"launches": 2, "ms": 1.0, "flops": 2 * 200e9, "bytes": 1e9}})
    assert set(h) == {"conv3x3_wino44h"} and h["conv3x3_wino44h"]["peak"] == 2500.0
    assert abs(h["conv3x3_wino44h"]["achieved"] - 400.0) < 0.01  # 4 partial products x 36 / 144 of the multiplies = 1.0 x algorithmic
    h3 = bench.rooflines_of({"conv3d_wino44h": {"launches": 1, "ms": 1.0, "flops": 3e11, "bytes": 1e9}})
    assert set(h3) == {"conv3d_wino44h"} and h3["conv3d_wino44h"]["peak"] == 2500.0  # not priced as the fp32 3-D kernel
    assert set(r) == {"conv3x3_wino", "conv3x3_wino44", "conv3x3_wino_up", "attention", "conv3d_wino", "conv3d_wino44",
                      "conv3d_"}  # MFMA classes only
    assert r["conv3d_wino44"]["executed_over_algorithmic_flops"] == 0.25
    assert abs(r["conv3x3_wino44"]["achieved"] - 12 * 100e9 / 4e-3 / 1e12 * 36 / 144) < 0.01  # F(4x4): a quarter of the multiplies
    w = r["conv3x3_wino"]
    alg = 10 * 66.8e9 / 3.0e-3 / 1e12
    assert abs(w["algorithmic_equiv_tflops"] - alg) < 0.01 and abs(w["achieved"] - alg * 16 / 36) < 0.01
    assert abs(w["frac"] - alg * 16 / 36 / 157.3) < 1e-3 and w["frac"] < 1 and w["bound"] == "mfma"
    assert abs(r["conv3x3_wino_up"]["achieved"] - 2 * 154.6e9 / 1e-3 / 1e12 * 9 / 36) < 0.01
    assert r["conv3d_wino"]["executed_over_algorithmic_flops"] == round(16 / 36, 4) and r["conv3d_"]["executed_over_algorithmic_flops"] == 1.0
    assert w["traffic"] is None or "builder-side" in w["traffic_source"]
    assert all(v["frac"] <= 1.0 for v in r.values())


def test_product_code_never_imports_the_oracle():
    """The oracle is test infrastructure: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may use it."""
    import re
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    pat = re.compile(r"^\s*(import\s+oracle\b|from\s+oracle\b)", re.M)
    offenders = [str(p.relative_to(root)) for p in list((root / "ddpm_ood_amd").rglob("*.py")) +
                 [root / "reconstruct.py", root / "ood_detection.py"] if pat.search(p.read_text())]
    assert offenders == []
    bench = (root / "bench.py").read_text()
    assert len(pat.findall(bench)) == 1 and "def cpu_baseline_worker" in bench.split("import oracle")[0]


def test_bench_launches_its_own_ranks(monkeypatch):
    """`python bench.py --gpus N` without a launcher (no WORLD_SIZE) re-executes itself through torch.distributed.run
    with N ranks on 127.0.0.1 and passes the arguments through; for N > 1 the default is strong scaling."""
    import subprocess
    import types

    import bench

    seen = {}

    def fake_run(cmd, **kw):
        seen["cmd"] = cmd
        return types.SimpleNamespace(returncode=7)

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    assert bench.shard_sizes("strong", 8, 1024, 1024) == (1024, [128] * 8)  # the default strong set: one batch over the ranks


def test_png_and_pnm_ingest(tmp_path):
    """PIL-format ingest (the reference's LoadImaged reads PNG through PIL; get_train_and_val_dataloader.py:60-76):
    a minimal PNG / PGM / PPM reader, every PNG scanline filter, 8- and 16-bit, grey and colour; axes swapped as
    MONAI's PILReader does; a colour file under --is_grayscale=1 keeps its first channel."""
    import struct
    import zlib

    from ddpm_ood_amd.data import get_data_loader, read_image

    rng = np.random.default_rng(0)

    def write_png(path, arr, depth=8, filters=(0, 1, 2, 3, 4)):
        h, w = arr.shape[:2]
        c = 1 if arr.ndim == 2 else arr.shape[2]
        ctype = {1: 0, 2: 4, 3: 2, 4: 6}[c]
        bpp = c * depth // 8
        raw = arr.astype(">u2" if depth == 16 else np.uint8).reshape(h, -1).view(np.uint8).reshape(h, w * bpp).astype(np.int32)
        out, prev = bytearray(), np.zeros(w * bpp, dtype=np.int32)
        for y in range(h):
            ft, cur = filters[y % len(filters)], raw[y]
            a = np.concatenate([np.zeros(bpp, dtype=np.int32), cur[:-bpp]])
            cprev = np.concatenate([np.zeros(bpp, dtype=np.int32), prev[:-bpp]])
            if ft == 0:
                line = cur
            elif ft == 1:
                line = cur - a
            elif ft == 2:
                line = cur - prev
            elif ft == 3:
                line = cur - ((a + prev) >> 1)
            else:
                pa, pb, pc = np.abs(prev - cprev), np.abs(a - cprev), np.abs(a + prev - 2 * cprev)
                pred = np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, prev, cprev))
                line = cur - pred
            out += bytes([ft]) + (line & 255).astype(np.uint8).tobytes()
            prev = cur

        def chunk(kind, body):
            return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body))

        blob = zlib.compress(bytes(out))
        path.write_bytes(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 0))
                         + chunk(b"IDAT", blob[:50]) + chunk(b"IDAT", blob[50:]) + chunk(b"IEND", b""))

    grey = rng.integers(0, 256, (28, 20))
    write_png(tmp_path / "g.png", grey)
    assert torch.equal(read_image(str(tmp_path / "g.png")), torch.from_numpy(grey.T.astype(np.float32)))
    rgb = rng.integers(0, 256, (9, 11, 3))
    write_png(tmp_path / "c.png", rgb)
    assert torch.equal(read_image(str(tmp_path / "c.png")), torch.from_numpy(np.swapaxes(rgb, 0, 1).astype(np.float32)))
    g16 = rng.integers(0, 65536, (6, 7))
    write_png(tmp_path / "g16.png", g16, depth=16)
    assert torch.equal(read_image(str(tmp_path / "g16.png")), torch.from_numpy(g16.T.astype(np.float32)))
    (tmp_path / "p.pgm").write_bytes(b"P5\n# comment\n20 28\n255\n" + grey.astype(np.uint8).tobytes())
    assert torch.equal(read_image(str(tmp_path / "p.pgm")), torch.from_numpy(grey.T.astype(np.float32)))
    (tmp_path / "p.ppm").write_bytes(b"P6 11 9 255\n" + rgb.astype(np.uint8).tobytes())
    assert torch.equal(read_image(str(tmp_path / "p.ppm")), torch.from_numpy(np.swapaxes(rgb, 0, 1).astype(np.float32)))
    # through the loader: a greyscale run over a grey and a colour file (first channel), min-max scaled
    sq = rng.integers(0, 256, (16, 16))
    write_png(tmp_path / "a.png", sq)
    write_png(tmp_path / "b.png", np.stack([sq, sq // 2, sq // 3], axis=-1))
    (tmp_path / "ids.csv").write_text(f"{tmp_path / 'a.png'},{tmp_path / 'b.png'}\n")
    ld = get_data_loader(str(tmp_path / "ids.csv"), 2, is_grayscale=True)
    x = next(iter(ld))["image"]
    want = torch.from_numpy(sq.T.astype(np.float32))
    want = (want - want.min()) / (want.max() - want.min())
    assert x.shape == (2, 1, 16, 16) and torch.allclose(x[0, 0], want) and torch.allclose(x[1, 0], want)
    with pytest.raises(ValueError, match="not channel-first"):
        get_data_loader(str(tmp_path / "ids.csv"), 2, is_grayscale=False)
    (tmp_path / "x.jpg").write_bytes(b"\xff\xd8\xff")
    with pytest.raises(NotImplementedError, match="JPEG"):
        read_image(str(tmp_path / "x.jpg"))


def test_product_lpips_refuses_cpu_tensors():
    """The product's PerceptualLoss has no CPU arithmetic left (the CPU restatement is oracle/lpips.py): host tensors raise,
    like every other op of the path."""
    from ddpm_ood_amd.perceptual import PerceptualLoss

    pl = PerceptualLoss(dimensions=2, include_pixel_loss=False, is_fake_3d=False, lpips_normalize=True)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        pl(torch.rand(2, 1, 32, 32), torch.rand(2, 1, 32, 32))


def test_native_training_host_logic():
    """Row f-3, CPU-checkable parts: which UNets the native step covers, that it refuses a model on the CPU (no fallback), and
    that the GEMM descriptor the Python wrapper fills has the header's field order."""
    import ctypes as C

    from ddpm_ood_amd import DiffusionModelUNet
    from ddpm_ood_amd._lib import GemmDesc
    from ddpm_ood_amd.train_native import NativeUNetStep, native_supported
    from ddpm_ood_amd.trainer import MODEL_CONFIGS

    small2d = DiffusionModelUNet(2, 1, 1, **MODEL_CONFIGS["small"])
    assert native_supported(small2d)
    assert native_supported(DiffusionModelUNet(3, 128, 128, **MODEL_CONFIGS["small"]))       # the LDM's latent UNet
    assert not native_supported(DiffusionModelUNet(3, 1, 1, **MODEL_CONFIGS["small"]))        # 1-channel volumes: no conv3d tiling
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        NativeUNetStep(small2d)
    names = [f[0] for f in GemmDesc._fields_]
    assert names[:7] == ["A", "B", "C", "M", "N", "K", "k_inner"] and names[-4:] == ["scratch", "scratch_floats", "split_f16", "reserved0"]
    assert C.sizeof(GemmDesc) % 8 == 0
