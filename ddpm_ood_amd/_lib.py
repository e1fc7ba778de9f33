"""ctypes binding of libddpm_ood_hip.so (C ABI: include/ddpm_ood_hip.h).

The product path has no CPU / PyTorch fallback: if the HIP library is missing this module
raises at first use, and every op refuses non-ROCm tensors.
"""

from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import torch

_LIB = None
LIB_NAME = "libddpm_ood_hip.so"
ABI_VERSION = 10


class HipLibraryMissing(RuntimeError):
    pass


class ConvDesc(C.Structure):
    _fields_ = [
        ("in1", C.c_void_p), ("in2", C.c_void_p), ("C1", C.c_int), ("C2", C.c_int),
        ("w_packed", C.c_void_p), ("w_raw", C.c_void_p), ("bias", C.c_void_p),
        ("gscale", C.c_void_p), ("gshift", C.c_void_p),
        ("chan_add", C.c_void_p), ("chan_add_stride", C.c_int),
        ("residual", C.c_void_p), ("out", C.c_void_p),
        ("B", C.c_int), ("Cout", C.c_int),
        ("Hi", C.c_int), ("Wi", C.c_int), ("Ho", C.c_int), ("Wo", C.c_int),
        ("ksize", C.c_int), ("mode", C.c_int), ("act", C.c_int), ("force_direct", C.c_int),
        ("Di", C.c_int), ("Do", C.c_int), ("dims", C.c_int), ("depth_taps", C.c_int),
        ("w_folded", C.c_void_p), ("out_act", C.c_int), ("reserved", C.c_int), ("w_wino", C.c_void_p),
        ("scratch", C.c_void_p), ("scratch_floats", C.c_size_t), ("w_wino44", C.c_void_p),
        ("w_wino44h", C.c_void_p), ("stats_out", C.c_void_p), ("w_d3h", C.c_void_p),
    ]


MAX_LEVELS = 8


class UNetConfig(C.Structure):
    _fields_ = [
        ("spatial_dims", C.c_int), ("in_channels", C.c_int), ("out_channels", C.c_int),
        ("num_levels", C.c_int),
        ("num_channels", C.c_int * MAX_LEVELS), ("attention_levels", C.c_int * MAX_LEVELS),
        ("num_res_blocks", C.c_int * MAX_LEVELS), ("num_head_channels", C.c_int * MAX_LEVELS),
        ("norm_num_groups", C.c_int), ("norm_eps", C.c_float), ("use_proj_attn", C.c_int),
    ]


class GemmDesc(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p),
        ("M", C.c_int), ("N", C.c_int), ("K", C.c_int), ("k_inner", C.c_int),
        ("a_m", C.c_int64), ("a_k", C.c_int64), ("a_k_outer", C.c_int64),
        ("b_n", C.c_int64), ("b_k", C.c_int64), ("b_k_outer", C.c_int64),
        ("c_m", C.c_int64), ("c_n", C.c_int64),
        ("batch", C.c_int), ("batch_inner", C.c_int),
        ("a_batch", C.c_int64), ("a_batch_outer", C.c_int64), ("b_batch", C.c_int64), ("b_batch_outer", C.c_int64),
        ("c_batch", C.c_int64), ("c_batch_outer", C.c_int64),
        ("alpha", C.c_float), ("beta", C.c_float),
        ("scratch", C.c_void_p), ("scratch_floats", C.c_size_t),
        ("split_f16", C.c_int), ("reserved0", C.c_int),
    ]


# name -> (restype, argtypes); this table is also what tests/test_abi.py checks against the header
SIGNATURES = {
    "ddpm_abi_version": (C.c_int, []),
    "ddpm_last_error": (C.c_char_p, []),
    "ddpm_conv_f32": (C.c_int, [C.POINTER(ConvDesc), C.c_void_p]),
    "ddpm_conv_scratch_floats": (C.c_size_t, [C.POINTER(ConvDesc)]),
    "ddpm_conv_takes_wino44h": (C.c_int, [C.POINTER(ConvDesc)]),
    "ddpm_conv_stats_parts": (C.c_int, [C.POINTER(ConvDesc)]),
    "ddpm_conv_s2h_weight_halves": (C.c_size_t, [C.c_int, C.c_int]),
    "ddpm_pack_conv_s2h_weight": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "ddpm_conv_d3h_weight_halves": (C.c_size_t, [C.c_int, C.c_int]),
    "ddpm_pack_conv_d3h_weight": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "ddpm_conv_d1s_weight_halves": (C.c_size_t, [C.c_int, C.c_int]),
    "ddpm_pack_conv_d1s_weight": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "ddpm_conv1x1_h_weight_halves": (C.c_size_t, [C.c_int, C.c_int]),
    "ddpm_pack_conv1x1_h_weight": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "ddpm_packed_conv_weight_floats": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "ddpm_pack_conv_weight_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                            C.c_void_p]),
    "ddpm_pack_conv_weight_taps_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                                 C.c_void_p]),
    "ddpm_pack_conv3d_weight_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "ddpm_packed_convtr_weight_floats": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "ddpm_pack_convtr_weight_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "ddpm_conv3d_k4s2_cin1_f32": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 6 + [C.c_void_p]),
    "ddpm_convtr3d_parity_weights_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "ddpm_parity_interleave3_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "ddpm_convtr3d_k4s2_cout1_f32": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 5 + [C.c_void_p]),
    "ddpm_wino_weight_floats": (C.c_size_t, [C.c_int, C.c_int]),
    "ddpm_pack_wino_weight_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "ddpm_wino44_weight_floats": (C.c_size_t, [C.c_int, C.c_int]),
    "ddpm_pack_wino44_weight_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "ddpm_pack_wino3d_weight_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "ddpm_convnd_generic_f32": (C.c_int, [C.c_void_p] * 5 + [C.c_int] * 12 + [C.c_void_p]),
    "ddpm_wino44h_weight_halves": (C.c_size_t, [C.c_int, C.c_int]),
    "ddpm_pack_wino44h_weight": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "ddpm_pack_wino44h_weight3d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "ddpm_pack_wino44_weight3d_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "ddpm_folded_upsample_weight_floats": (C.c_size_t, [C.c_int, C.c_int]),
    "ddpm_fold_upsample_weight_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "ddpm_gn_scale_shift_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p]),
    "ddpm_gn_finalize_f32": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p]),
    "ddpm_channel_stats_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "ddpm_attention_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_float, C.c_void_p]),
    "ddpm_attention_scratch_floats": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "ddpm_attention_ws_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_float, C.c_void_p, C.c_size_t, C.c_void_p]),
    "ddpm_timestep_embedding_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "ddpm_add_noise_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float,
                                     C.c_void_p, C.c_int, C.c_int64, C.c_void_p]),
    "ddpm_plms_step_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                     C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_int64,
                                     C.c_void_p]),
    "ddpm_clamp_mse_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_int, C.c_int64, C.c_void_p]),
    "ddpm_vq_nearest_f32": (C.c_int, [C.c_void_p] * 5 + [C.c_int, C.c_int, C.c_int64, C.c_int, C.c_void_p]),
    "ddpm_lpips_conv_f32": (C.c_int, [C.c_void_p] * 6 + [C.c_int] * 10 + [C.c_void_p]),
    "ddpm_maxpool3s2_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p]),
    "ddpm_lpips_conv_biasmap_f32": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 9 + [C.c_void_p]),
    "ddpm_lpips_conv_mfma_supported": (C.c_int, [C.c_int] * 5),
    "ddpm_lpips_pack_conv_weight_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "ddpm_lpips_conv_mfma_f32": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 7 + [C.c_void_p]),
    "ddpm_lpips_layer_f32": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 4 + [C.c_void_p]),
    "ddpm_prof_enable": (C.c_int, [C.c_int]),
    "ddpm_prof_report": (C.c_int, [C.c_char_p, C.c_size_t]),
    "ddpm_status_read": (C.c_int, [C.POINTER(C.c_uint), C.c_int, C.c_void_p]),
    "ddpm_vq_near_ties_read": (C.c_int, [C.POINTER(C.c_uint), C.c_int, C.c_void_p]),
    "ddpm_set_split_f16": (C.c_int, [C.c_int]),
    "ddpm_get_split_f16": (C.c_int, []),
    "ddpm_split_f16_active": (C.c_int, []),
    "ddpm_reload_env": (C.c_int, []),
    "ddpm_unet_create": (C.c_void_p, [C.POINTER(UNetConfig)]),
    "ddpm_unet_destroy": (None, [C.c_void_p]),
    "ddpm_unet_param_blob_floats": (C.c_size_t, [C.c_void_p]),
    "ddpm_unet_bind_param_blob": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ddpm_unet_num_params": (C.c_int, [C.c_void_p]),
    "ddpm_unet_param_name": (C.c_char_p, [C.c_void_p, C.c_int]),
    "ddpm_unet_param_numel": (C.c_int64, [C.c_void_p, C.c_int]),
    "ddpm_unet_set_param": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "ddpm_unet_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "ddpm_unet_workspace_bytes3d": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "ddpm_unet_forward3d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                      C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "ddpm_unet_forward_graphed": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                            C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "ddpm_unet_num_graphs": (C.c_int, [C.c_void_p]),
    "ddpm_unet_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                    C.c_void_p, C.c_size_t, C.c_void_p]),
    # ---- training step (ABI 10)
    "ddpm_gemm_f32": (C.c_int, [C.POINTER(GemmDesc), C.c_void_p]),
    "ddpm_gemm_scratch_floats": (C.c_size_t, [C.POINTER(GemmDesc)]),
    "ddpm_conv_wgrad_scratch_floats": (C.c_size_t, [C.c_int] * 9),
    "ddpm_conv_wgrad_f32": (C.c_int, [C.c_void_p] * 3 + [C.c_int] * 9 + [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                       C.c_int, C.c_void_p]),
    "ddpm_conv3d_wgrad_scratch_floats": (C.c_size_t, [C.c_int] * 10),
    "ddpm_conv3d_wgrad_f32": (C.c_int, [C.c_void_p] * 3 + [C.c_int] * 10 + [C.c_void_p, C.c_size_t, C.c_void_p]),
    "ddpm_resample3_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "ddpm_conv_weight_rot180t_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "ddpm_gn_stats_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p]),
    "ddpm_gn_apply_f32": (C.c_int, [C.c_void_p] * 5 + [C.c_int] * 5 + [C.c_void_p]),
    "ddpm_gn_forward_f32": (C.c_int, [C.c_void_p] * 6 + [C.c_int] * 4 + [C.c_float, C.c_int, C.c_void_p]),
    "ddpm_gn_backward_f32": (C.c_int, [C.c_void_p] * 6 + [C.c_int] + [C.c_void_p] * 5 + [C.c_int] * 5 + [C.c_void_p]),
    "ddpm_row_sum_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    "ddpm_col_sum_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_float, C.c_int, C.c_void_p]),
    "ddpm_silu_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "ddpm_silu_backward_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "ddpm_axpby_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_int64, C.c_void_p]),
    "ddpm_scale_check_f32": (C.c_int, [C.c_void_p, C.c_float, C.c_int64, C.c_void_p]),
    "ddpm_chan_copy_f32": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int] * 8 + [C.c_void_p]),
    "ddpm_resample2_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "ddpm_softmax_rows_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    "ddpm_softmax_backward_rows_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    "ddpm_mse_loss_grad_f32": (C.c_int, [C.c_void_p] * 4 + [C.c_int64, C.c_float, C.c_void_p]),
    "ddpm_fill_f32": (C.c_int, [C.c_void_p, C.c_float, C.c_int64, C.c_void_p]),
    "ddpm_randn_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_uint64, C.c_uint64, C.c_void_p]),
    "ddpm_adam_step_f32": (C.c_int, [C.c_void_p] * 4 + [C.c_int64] + [C.c_float] * 4 + [C.c_int, C.c_float, C.c_void_p]),
}


def lib_path() -> Path:
    env = os.environ.get("DDPM_OOD_HIP_LIB")
    return Path(env) if env else Path(__file__).resolve().parent / LIB_NAME


def load():
    """Load the shared library (once).  Raises HipLibraryMissing -- never falls back."""
    global _LIB
    if _LIB is not None:
        return _LIB
    p = lib_path()
    if not p.exists():
        raise HipLibraryMissing(
            f"{p} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"or ddpm_ood_amd/csrc/build.sh.  There is no CPU fallback for the reconstruction path.")
    lib = C.CDLL(str(p))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.ddpm_abi_version() != ABI_VERSION:
        raise HipLibraryMissing(f"{p}: ABI version {lib.ddpm_abi_version()} != {ABI_VERSION}")
    _LIB = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().ddpm_last_error().decode() or f"error code {rc}"
        if rc == -2:
            raise KeyError(f"{what}: {msg}")
        if rc == -1:
            raise ValueError(f"{what}: {msg}")
        raise RuntimeError(f"{what}: {msg}")


def require_device_f32(t: torch.Tensor, name: str) -> torch.Tensor:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"{name} must be a ROCm device tensor: the HIP reconstruction path has no CPU fallback")
    if t.dtype != torch.float32:
        raise TypeError(f"{name} must be float32, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def ptr(t) -> int | None:
    return None if t is None else t.data_ptr()


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


# ---- numeric guard (include/ddpm_ood_hip.h, "Numeric guard of the split-f16 kernel families") -------------------
STATUS_BITS = {1: "non-finite UNet output (eps) at a PLMS step", 2: "non-finite reconstruction at clamp + MSE",
               4: "non-finite latent at the VQ-VAE quantiser", 8: "non-finite gradient after a scaled backward (training)"}
STATUS_NONFINITE_GRAD = 8


def status_read(clear: bool = True) -> int:
    """The device status word of the current device (synchronises the current stream)."""
    word = C.c_uint(0)
    check(load().ddpm_status_read(C.byref(word), int(clear), stream_ptr()), "status_read")
    return int(word.value)


def vq_near_ties_read(clear: bool = True) -> int:
    """Latent positions whose two nearest codes were within 1e-5 (relative) of each other since the last clear: how close
    the quantiser came to a code flip (the one discontinuous op of the path)."""
    n = C.c_uint(0)
    check(load().ddpm_vq_near_ties_read(C.byref(n), int(clear), stream_ptr()), "vq_near_ties_read")
    return int(n.value)


def status_text(word: int) -> str:
    return "; ".join(t for b, t in STATUS_BITS.items() if word & b) or "clean"


def set_split_f16(on: bool) -> bool:
    """False: every split-f16 kernel family runs its fp32-MFMA form.  Returns the previous setting."""
    return bool(load().ddpm_set_split_f16(int(bool(on))))


def split_f16() -> bool:
    """The master switch (what set_split_f16 sets)."""
    return bool(load().ddpm_get_split_f16())


def split_f16_active() -> bool:
    """Master switch on AND at least one split-f16 family enabled: would set_split_f16(False) change the dispatch?"""
    return bool(load().ddpm_split_f16_active())


def reload_env() -> None:
    """Parse the DDPM_* switches again (the library reads them once per process)."""
    if _LIB is not None:
        _LIB.ddpm_reload_env()
