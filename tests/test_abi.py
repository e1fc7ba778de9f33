"""CPU: the C-ABI library loads and exports every symbol include/ddpm_ood_hip.h declares
(no compute calls without a GPU); host-only entry points behave."""

import ctypes as C
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


def _declared():
    text = (ROOT / "include" / "ddpm_ood_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ddpm_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    from ddpm_ood_amd import _lib

    if not _lib.lib_path().exists():
        import __graft_entry__ as g

        g.build()
    return _lib.load()


def test_every_declared_symbol_is_exported_and_bound(lib):
    from ddpm_ood_amd import _lib

    names = _declared()
    assert len(names) >= 23
    for n in names:
        assert hasattr(lib, n), f"{n} declared in the header but not exported"
    assert sorted(_lib.SIGNATURES) == names, "ctypes table and header disagree"


def test_struct_layouts_match_header(lib):
    from ddpm_ood_amd._lib import ConvDesc, UNetConfig

    # field order of the C structs (pointers 8 B, ints 4 B): sizes computed by hand from the header
    assert C.sizeof(ConvDesc) == 8 + 8 + 4 + 4 + 8 * 5 + 8 + 8 + 8 + 8 + 4 * 10 + 4 * 4 + 8 + 8 + 8 + 8 + 8 + 8 + 8 + 8 + 8  # incl. padding after stride; + scratch, scratch_floats, w_wino44, w_wino44h, stats_out, w_d3h
    assert C.sizeof(UNetConfig) == 4 * 4 + 4 * 8 * 4 + 4 + 4 + 4
    from ddpm_ood_amd._lib import GemmDesc

    assert C.sizeof(GemmDesc) == 3 * 8 + 4 * 4 + 8 * 8 + 2 * 4 + 6 * 8 + 2 * 4 + 8 + 8 + 2 * 4  # ddpm_gemm_desc (ABI 10)
    assert GemmDesc.a_m.offset == 40 and GemmDesc.batch.offset == 104 and GemmDesc.alpha.offset == 160


def test_host_only_entry_points(lib):
    assert lib.ddpm_abi_version() == 10
    assert lib.ddpm_set_split_f16(0) == 1 and lib.ddpm_get_split_f16() == 0 and lib.ddpm_set_split_f16(1) == 0
    assert lib.ddpm_reload_env() == 0 and lib.ddpm_get_split_f16() == 1  # a reload keeps the run-time switch
    assert lib.ddpm_packed_conv_weight_floats(128, 128, 3) == 128 * 128 * 9
    assert lib.ddpm_packed_conv_weight_floats(96, 128, 3) == 0      # Cout % 128
    assert lib.ddpm_packed_conv_weight_floats(128, 3, 3) == 0       # Cin % 8
    assert lib.ddpm_conv_f32(None, None) == -1
    assert b"NULL" in lib.ddpm_last_error()


def test_engine_parameter_table_matches_monai_keys(lib):
    """The native engine's parameter names are exactly the MONAI-Generative state_dict keys
     of the Python holder module (+ the host-computed 'freqs' table), sizes included."""
    from ddpm_ood_amd import DiffusionModelUNet
    from ddpm_ood_amd.trainer import MODEL_CONFIGS

    for model_type, channels, nparams in (("small", 1, 17_709_953), ("big", 3, 172_573_187)):
        m = DiffusionModelUNet(2, channels, channels, **MODEL_CONFIGS[model_type])
        assert sum(p.numel() for p in m.parameters()) == nparams
        cfg = m._config()
        h = C.c_void_p(lib.ddpm_unet_create(C.byref(cfg)))
        assert h.value
        got = {lib.ddpm_unet_param_name(h, i).decode(): lib.ddpm_unet_param_numel(h, i)
               for i in range(lib.ddpm_unet_num_params(h))}
        want = {k: v.numel() for k, v in m.state_dict().items()}
        want["freqs"] = cfg.num_channels[0] // 2
        assert got == want
        assert lib.ddpm_unet_param_blob_floats(h) >= 1.9 * nparams  # raw + MFMA-packed copies of the big weights
        ws = lib.ddpm_unet_workspace_bytes(h, 256, 32, 32)
        assert 0 < ws < 8 << 30
        assert lib.ddpm_unet_workspace_bytes(h, 1, 30, 30) == 0  # 30 -> 15 -> 8 -> 16: skip extents mismatch
        lib.ddpm_unet_destroy(h)


def test_unsupported_configs_fail_loudly(lib):
    from ddpm_ood_amd._lib import UNetConfig

    cfg = UNetConfig()
    cfg.spatial_dims, cfg.num_levels = 4, 1
    assert not lib.ddpm_unet_create(C.byref(cfg))
    assert b"spatial_dims" in lib.ddpm_last_error()


def test_product_does_not_import_oracle():
    for p in (ROOT / "ddpm_ood_amd").rglob("*.py"):
        assert not re.search(r"^\s*(from|import)\s+oracle\b", p.read_text(), flags=re.M), p
    for p in (ROOT / "reconstruct.py", ROOT / "ood_detection.py", ROOT / "train_ddpm.py"):
        assert "oracle" not in p.read_text()
    # development tools are not the product either, but the rule is about the directory: only tests/, smoke() and bench.py's
    # cpu_baseline leg touch oracle/ (the two tools that compare against it live under tests/ since round 6)
    for p in (ROOT / "tools").rglob("*.py"):
        assert not re.search(r"^\s*(from|import)\s+oracle\b", p.read_text(), flags=re.M), p


def test_missing_library_is_an_error(monkeypatch, tmp_path):
    from ddpm_ood_amd import _lib

    monkeypatch.setattr(_lib, "_LIB", None)
    monkeypatch.setenv("DDPM_OOD_HIP_LIB", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.HipLibraryMissing, match="no CPU fallback"):
        _lib.load()
