"""Thin Python wrappers over the stand-alone C-ABI operators (include/ddpm_ood_hip.h).

Used by the scheduler / trainer mirrors and by the per-kernel parity tests.  Every function
takes ROCm device tensors and launches on torch's current HIP stream; none falls back to
PyTorch ops.
"""

from __future__ import annotations

import ctypes as C
from ctypes import byref as C_byref

import numpy as np
import torch

from . import _lib
from ._lib import ConvDesc, check, ptr, require_device_f32, stream_ptr

CONV_NORMAL, CONV_STRIDE2, CONV_UPSAMPLE2 = 0, 1, 2
ACT_NONE, ACT_SILU, ACT_RELU = 0, 1, 2


def pack_conv_weight(weight: torch.Tensor) -> torch.Tensor | None:
    """torch [Cout, Cin, k, k] (or [out, in]) -> MFMA-packed weight, or None if unpackable."""
    lib = _lib.load()
    w = require_device_f32(weight, "weight")
    cout, cin = w.shape[0], w.shape[1]
    k = w.shape[2] if w.ndim == 4 else 1
    n = lib.ddpm_packed_conv_weight_floats(cout, cin, k)
    if n == 0:
        return None
    out = torch.empty(n, dtype=torch.float32, device=w.device)
    check(lib.ddpm_pack_conv_weight_f32(ptr(w), ptr(out), cout, cin, k, 0, cout, stream_ptr()), "pack_conv_weight")
    return out


def pack_wino_weight(weight: torch.Tensor) -> torch.Tensor | None:
    """torch [Cout, Cin, 3, 3] -> Winograd-domain weights U = G g G^T in the MFMA layout (None if unsupported)."""
    lib = _lib.load()
    w = require_device_f32(weight, "weight")
    if w.ndim != 4 or tuple(w.shape[2:]) != (3, 3):
        return None
    n = lib.ddpm_wino_weight_floats(w.shape[0], w.shape[1])
    if n == 0:
        return None
    out = torch.empty(n, dtype=torch.float32, device=w.device)
    check(lib.ddpm_pack_wino_weight_f32(ptr(w), ptr(out), w.shape[0], w.shape[1], stream_ptr()), "pack_wino_weight")
    return out


def fold_upsample_weight(weight: torch.Tensor) -> torch.Tensor | None:
    """torch [Cout, Cin, 3, 3] of an Upsample conv -> the 4 folded 2x2-tap weights (None if not foldable)."""
    lib = _lib.load()
    w = require_device_f32(weight, "weight")
    n = lib.ddpm_folded_upsample_weight_floats(w.shape[0], w.shape[1])
    if n == 0 or w.ndim != 4 or w.shape[2] != 3:
        return None
    out = torch.empty(n, dtype=torch.float32, device=w.device)
    check(lib.ddpm_fold_upsample_weight_f32(ptr(w), ptr(out), w.shape[0], w.shape[1], stream_ptr()), "fold")
    return out


def conv(x, weight, bias=None, *, x2=None, gscale=None, gshift=None, act=ACT_NONE, mode=CONV_NORMAL,
         chan_add=None, chan_add_offset=0, residual=None, packed=None, force_direct=False, folded=None,
         wino=None, out_act=ACT_NONE):
    """Fused conv / linear.  x: [B, C1, H, W]; x2: optional second source of a virtual concat."""
    lib = _lib.load()
    x = require_device_f32(x, "x")
    w = require_device_f32(weight, "weight")
    if x.ndim == 2:  # Linear: [B, C] == [B, C, 1, 1]
        x = x[:, :, None, None]
        was_linear = True
    else:
        was_linear = False
    B, C1, Hi, Wi = x.shape
    cout = w.shape[0]
    k = w.shape[2] if w.ndim == 4 else 1
    if mode == CONV_STRIDE2:
        Ho, Wo = (Hi + 1) // 2, (Wi + 1) // 2
    elif mode == CONV_UPSAMPLE2:
        Ho, Wo = 2 * Hi, 2 * Wi
    else:
        Ho, Wo = Hi, Wi
    if packed is None and not force_direct:
        packed = pack_conv_weight(w)
    out = torch.empty((B, cout, Ho, Wo), dtype=torch.float32, device=x.device)
    keep = [x, w, out, packed]
    d = ConvDesc()
    d.in1 = ptr(x)
    d.C1 = C1
    if x2 is not None:
        x2 = require_device_f32(x2, "x2")
        d.in2, d.C2 = ptr(x2), x2.shape[1]
        keep.append(x2)
    d.w_packed = ptr(packed)
    d.w_raw = ptr(w)
    if bias is not None:
        bias = require_device_f32(bias, "bias")
        d.bias = ptr(bias)
    if gscale is not None:
        gscale = require_device_f32(gscale, "gscale")
        gshift = require_device_f32(gshift, "gshift")
        d.gscale, d.gshift = ptr(gscale), ptr(gshift)
    if chan_add is not None:
        chan_add = require_device_f32(chan_add, "chan_add")
        d.chan_add = chan_add.data_ptr() + 4 * chan_add_offset
        d.chan_add_stride = chan_add.shape[1]
    if residual is not None:
        residual = require_device_f32(residual, "residual")
        d.residual = ptr(residual)
    d.out = ptr(out)
    d.B, d.Cout, d.Hi, d.Wi, d.Ho, d.Wo = B, cout, Hi, Wi, Ho, Wo
    d.ksize, d.mode, d.act, d.force_direct = k, mode, act, int(force_direct)
    d.w_folded = ptr(folded)
    d.w_wino = ptr(wino)
    d.out_act = out_act
    check(lib.ddpm_conv_f32(C.byref(d), stream_ptr()), "conv")
    return out[:, :, 0, 0] if was_linear else out


def conv3d_supported(weight: torch.Tensor) -> bool:
    """3x3x3 stride-1 pad-1 weights with an MFMA tiling (Cin % 4 == 0, Cout % 128 == 0)."""
    return (weight.ndim == 5 and tuple(weight.shape[2:]) == (3, 3, 3) and weight.shape[0] % 128 == 0
            and weight.shape[1] % 4 == 0)


def pack_conv3d_weight(weight: torch.Tensor) -> torch.Tensor:
    """torch [Cout, Cin, 3, 3, 3] -> three MFMA-packed 2-D slabs, one per depth tap."""
    lib = _lib.load()
    w = require_device_f32(weight, "weight")
    cout, cin = w.shape[:2]
    slab = cout * cin * 9
    out = torch.empty(3 * slab, dtype=torch.float32, device=w.device)
    for kd in range(3):
        check(lib.ddpm_pack_conv_weight_taps_f32(ptr(w), out.data_ptr() + 4 * kd * slab, cout, cin, 3, 27, 9 * kd,
                                                 stream_ptr()), "pack_conv3d_weight")
    return out


def conv3d(x, weight, bias=None, *, act=ACT_NONE, out_act=ACT_NONE, residual=None, packed=None):
    """F.conv3d(act(x), weight, bias, stride 1, pad 1) (+ residual, + output activation) on NCDHW tensors as three
    depth-tap launches of the 2-D MFMA kernel (centre tap first; the last launch applies ``out_act``)."""
    lib = _lib.load()
    x = require_device_f32(x, "x")
    w = require_device_f32(weight, "weight")
    if not conv3d_supported(w):
        raise ValueError("conv3d: only 3x3x3 weights with Cin % 4 == 0 and Cout % 128 == 0 have an MFMA tiling")
    B, C, D, H, W = x.shape
    cout = w.shape[0]
    if packed is None:
        packed = pack_conv3d_weight(w)
    out = torch.empty((B, cout, D, H, W), dtype=torch.float32, device=x.device)
    if bias is not None:
        bias = require_device_f32(bias, "bias")
    if residual is not None:
        residual = require_device_f32(residual, "residual")
    slab = cout * C * 9
    taps = (1,) if D == 1 else (1, 0, 2)  # depth 1: the outer depth taps only ever see padding
    for i, kd in enumerate(taps):
        d = ConvDesc()
        d.in1, d.C1 = ptr(x), C
        d.w_packed = packed.data_ptr() + 4 * kd * slab
        d.w_raw = ptr(w)
        d.out = ptr(out)
        d.B, d.Cout, d.Hi, d.Wi, d.Ho, d.Wo = B, cout, H, W, H, W
        d.ksize, d.mode, d.act = 3, CONV_NORMAL, act
        d.Di, d.Do, d.kd = D, D, kd
        if D == 1:
            d.w_raw = None  # 27-tap tensor: must not reach a 9-tap fallback kernel
        if i == 0:
            d.bias, d.residual = ptr(bias), ptr(residual)
        else:
            d.accumulate = 1
        if i == len(taps) - 1:
            d.out_act = out_act
        check(lib.ddpm_conv_f32(C_byref(d), stream_ptr()), "conv3d")
    return out


def gn_scale_shift(x, gamma, beta, groups: int, eps: float, x2=None):
    lib = _lib.load()
    x = require_device_f32(x, "x")
    B, C1 = x.shape[:2]
    hw = x[0, 0].numel()
    C2 = 0
    if x2 is not None:
        x2 = require_device_f32(x2, "x2")
        C2 = x2.shape[1]
    gamma = require_device_f32(gamma, "gamma")
    beta = require_device_f32(beta, "beta")
    scale = torch.empty((B, C1 + C2), dtype=torch.float32, device=x.device)
    shift = torch.empty_like(scale)
    check(lib.ddpm_gn_scale_shift_f32(ptr(x), ptr(x2), C1, C2, ptr(gamma), ptr(beta), ptr(scale), ptr(shift), B, hw,
                                      groups, eps, stream_ptr()), "gn_scale_shift")
    return scale, shift


def attention(qkv, residual, num_heads: int, scale: float):
    """qkv: [B, 3C, N] -> [B, C, N] = softmax(scale q^T k) v (+ residual)."""
    lib = _lib.load()
    qkv = require_device_f32(qkv, "qkv")
    B, C3, N = qkv.shape
    Cc = C3 // 3
    if residual is not None:
        residual = require_device_f32(residual, "residual")
    out = torch.empty((B, Cc, N), dtype=torch.float32, device=qkv.device)
    check(lib.ddpm_attention_f32(ptr(qkv), ptr(residual), ptr(out), B, Cc, N, num_heads, scale, stream_ptr()),
          "attention")
    return out


def timestep_embedding(timesteps, freqs, dim: int):
    lib = _lib.load()
    if not timesteps.is_cuda or timesteps.dtype != torch.int64:
        raise RuntimeError("timesteps must be an int64 ROCm device tensor")
    freqs = require_device_f32(freqs, "freqs")
    out = torch.empty((timesteps.shape[0], dim), dtype=torch.float32, device=timesteps.device)
    check(lib.ddpm_timestep_embedding_f32(ptr(timesteps.contiguous()), ptr(freqs), ptr(out), timesteps.shape[0], dim,
                                          stream_ptr()), "timestep_embedding")
    return out


def add_noise(x0, noise, sqrt_ac: np.ndarray, sqrt_1m_ac: np.ndarray, b_scale: float = 1.0):
    lib = _lib.load()
    x0 = require_device_f32(x0, "original_samples")
    noise = require_device_f32(noise, "noise")
    B = x0.shape[0]
    a = np.ascontiguousarray(sqrt_ac, dtype=np.float32)
    b = np.ascontiguousarray(sqrt_1m_ac, dtype=np.float32)
    assert a.shape == (B,) and b.shape == (B,)
    out = torch.empty_like(x0)
    check(lib.ddpm_add_noise_f32(ptr(x0), ptr(noise), a.ctypes.data_as(C.POINTER(C.c_float)),
                                 b.ctypes.data_as(C.POINTER(C.c_float)), float(b_scale), ptr(out), B,
                                 x0[0].numel(), stream_ptr()), "add_noise")
    return out


def plms_step(sample, ets, kind: int, sample_coeff: float, coef_eps: float, denom: float, *, v_prediction=False,
              v_a: float = 0.0, v_b: float = 0.0, out=None):
    """ets: newest-first list of eps tensors (1..4 of them)."""
    lib = _lib.load()
    sample = require_device_f32(sample, "sample")
    es = [require_device_f32(e, "model_output") for e in ets] + [None] * (4 - len(ets))
    if out is None:
        out = torch.empty_like(sample)
    check(lib.ddpm_plms_step_f32(ptr(sample), ptr(es[0]), ptr(es[1]), ptr(es[2]), ptr(es[3]), kind,
                                 int(v_prediction), v_a, v_b, sample_coeff, coef_eps, denom, ptr(out),
                                 sample.numel(), stream_ptr()), "plms_step")
    return out


def clamp_mse_(orig, recon, b_scale: float = 1.0):
    """In place: recon <- clamp(recon / b_scale, 0, 1); returns per-image MSE [B]."""
    lib = _lib.load()
    orig = require_device_f32(orig, "images_original")
    if not recon.is_contiguous():
        raise ValueError("recon must be contiguous (it is updated in place)")
    recon = require_device_f32(recon, "reconstructions")
    B = orig.shape[0]
    mse = torch.empty(B, dtype=torch.float32, device=orig.device)
    check(lib.ddpm_clamp_mse_f32(ptr(orig), ptr(recon), float(b_scale), ptr(mse), B, orig[0].numel(), stream_ptr()),
          "clamp_mse")
    return mse


# ---- LPIPS-AlexNet pieces (src/losses/perceptual_loss.py:105-186) -------------------------------------

def lpips_conv(x, weight, bias, stride: int, pad: int, relu: bool = True, in_scale=None, in_shift=None):
    """relu(conv2d(x * in_scale[c] + in_shift[c], weight, bias, stride, pad)); a 1-channel x feeds every input
    channel of the layer (the ScalingLayer's broadcast)."""
    lib = _lib.load()
    x = require_device_f32(x, "x")
    w = require_device_f32(weight, "weight")
    N, Cx, H, W = x.shape
    cout, cin, k, _ = w.shape
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    out = torch.empty((N, cout, max(Ho, 0), max(Wo, 0)), dtype=torch.float32, device=x.device)
    opt = [None if t is None else require_device_f32(t, "lpips_conv operand") for t in (bias, in_scale, in_shift)]
    check(lib.ddpm_lpips_conv_f32(ptr(x), ptr(w), ptr(opt[0]), ptr(opt[1]), ptr(opt[2]), ptr(out), N, Cx, cin, H, W,
                                  cout, k, stride, pad, int(relu), stream_ptr()), "lpips_conv")
    return out


def maxpool3s2(x):
    """MaxPool2d(3, 2)."""
    lib = _lib.load()
    x = require_device_f32(x, "x")
    N, Cc, H, W = x.shape
    out = torch.empty((N, Cc, (H - 3) // 2 + 1, (W - 3) // 2 + 1), dtype=torch.float32, device=x.device)
    check(lib.ddpm_maxpool3s2_f32(ptr(x), ptr(out), N * Cc, H, W, stream_ptr()), "maxpool3s2")
    return out


def lpips_layer(f0, f1, lin, out=None):
    """out[n] (+)= spatial mean of the lin-weighted squared difference of the channel-normalised features."""
    lib = _lib.load()
    f0 = require_device_f32(f0, "f0")
    f1 = require_device_f32(f1, "f1")
    lin = require_device_f32(lin, "lin")
    if f0.shape != f1.shape:
        raise ValueError(f"feature maps differ in shape: {tuple(f0.shape)} vs {tuple(f1.shape)}")
    N, Cc, H, W = f0.shape
    acc = out is not None
    if out is None:
        out = torch.empty(N, dtype=torch.float32, device=f0.device)
    check(lib.ddpm_lpips_layer_f32(ptr(f0), ptr(f1), ptr(lin), ptr(out), N, Cc, H * W, int(acc), stream_ptr()),
          "lpips_layer")
    return out


def vq_nearest(x, codebook):
    """VQ-VAE quantiser, eval path: (indices int64 [B, *spatial], x + (codebook[indices] - x) [B, D, *spatial])."""
    lib = _lib.load()
    x = require_device_f32(x, "x")
    e = require_device_f32(codebook, "codebook")
    B, D = x.shape[:2]
    S = x[0, 0].numel()
    K = e.shape[0]
    if e.shape[1] != D:
        raise ValueError(f"codebook is [{K}, {e.shape[1]}] but the latent has {D} channels")
    idx = torch.empty((B,) + tuple(x.shape[2:]), dtype=torch.int32, device=x.device)
    out = torch.empty_like(x)
    norms = torch.empty(K, dtype=torch.float32, device=x.device)
    check(lib.ddpm_vq_nearest_f32(ptr(x), ptr(e), ptr(norms), idx.data_ptr(), ptr(out), B, D, S, K, stream_ptr()),
          "vq_nearest")
    return idx.long(), out
