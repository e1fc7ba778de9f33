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


def pack_wino44_weight(weight: torch.Tensor) -> torch.Tensor | None:
    """torch [Cout, Cin, 3, 3] -> F(4x4, 3x3) Winograd-domain weights (6 x 6 positions) in the MFMA layout."""
    lib = _lib.load()
    w = require_device_f32(weight, "weight")
    if w.ndim != 4 or tuple(w.shape[2:]) != (3, 3):
        return None
    n = lib.ddpm_wino44_weight_floats(w.shape[0], w.shape[1])
    if n == 0:
        return None
    out = torch.empty(n, dtype=torch.float32, device=w.device)
    check(lib.ddpm_pack_wino44_weight_f32(ptr(w), ptr(out), w.shape[0], w.shape[1], stream_ptr()), "pack_wino44_weight")
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


def pack_wino44h_weight(weight: torch.Tensor) -> torch.Tensor | None:
    """torch [Cout, Cin, 3, 3] -> F(4x4, 3x3) Winograd-domain weights as split-f16 planes (hi = f16(2^10 U), lo = the
    remainder) in the order conv_wino44h.hip's LDS-DMA lands them; None without a tiling (Cout % 64, Cin % 16)."""
    lib = _lib.load()
    w = require_device_f32(weight, "weight")
    if w.ndim != 4 or tuple(w.shape[2:]) != (3, 3):
        return None
    n = lib.ddpm_wino44h_weight_halves(w.shape[0], w.shape[1])
    if n == 0:
        return None
    out = torch.empty(n, dtype=torch.float16, device=w.device)
    check(lib.ddpm_pack_wino44h_weight(ptr(w), out.data_ptr(), w.shape[0], w.shape[1], stream_ptr()), "pack_wino44h_weight")
    return out


def pack_conv1x1_h_weight(w):
    """[Cout, Cin(, 1, 1)] -> the pre-split f16 planes of the DMA-fed 1x1 kernel (conv1x1_dma.hip), or None if Cout % 128 or
    Cin % 16.  Pass it as conv(..., wino44h=...) of a 1x1 convolution."""
    lib = _lib.load()
    w = require_device_f32(w, "weight")
    if w.ndim == 4 and (w.shape[2] != 1 or w.shape[3] != 1):
        return None
    n = lib.ddpm_conv1x1_h_weight_halves(w.shape[0], w.shape[1])
    if n == 0:
        return None
    out = torch.empty(n, dtype=torch.float16, device=w.device)
    check(lib.ddpm_pack_conv1x1_h_weight(ptr(w), out.data_ptr(), w.shape[0], w.shape[1], stream_ptr()), "pack_conv1x1_h_weight")
    return out


def pack_conv_d3h_weight(w):
    """[Cout, Cin, 3, 3] -> split-f16 planes of the direct 3x3 kernel (conv_d3h.hip), or None (Cout % 128, Cin % 8).  Pass it as
    conv(..., d3h=...)."""
    lib = _lib.load()
    w = require_device_f32(w, "weight")
    if w.ndim != 4 or w.shape[2] != 3 or w.shape[3] != 3:
        return None
    n = lib.ddpm_conv_d3h_weight_halves(w.shape[0], w.shape[1])
    if n == 0:
        return None
    out = torch.empty(n, dtype=torch.float16, device=w.device)
    check(lib.ddpm_pack_conv_d3h_weight(ptr(w), out.data_ptr(), w.shape[0], w.shape[1], stream_ptr()), "pack_conv_d3h_weight")
    return out


def pack_conv_d1s_weight(w):
    """[Cout, Cin, 1, 1] (or [Cout, Cin]) -> split-f16 planes of the small-launch 1x1 kernel (conv_d3s.hip), or None (Cout % 64,
    Cin % 128).  Pass it as conv(..., d3h=...) of a 1x1 convolution."""
    lib = _lib.load()
    w = require_device_f32(w, "weight")
    if w.ndim == 4 and (w.shape[2] != 1 or w.shape[3] != 1):
        return None
    n = lib.ddpm_conv_d1s_weight_halves(w.shape[0], w.shape[1])
    if n == 0:
        return None
    out = torch.zeros(n, dtype=torch.float16, device=w.device)
    check(lib.ddpm_pack_conv_d1s_weight(ptr(w), out.data_ptr(), w.shape[0], w.shape[1], 0, w.shape[0], stream_ptr()),
          "pack_conv_d1s_weight")
    return out


def pack_conv_s2h_weight(w):
    """[Cout, Cin, 3, 3] -> split-f16 planes of the direct stride-2 kernel (conv_s2h.hip), or None if the shape has no tiling.
    Pass it as conv(..., mode=CONV_STRIDE2, wino44h=...)."""
    lib = _lib.load()
    w = require_device_f32(w, "weight")
    if w.ndim != 4 or w.shape[2] != 3 or w.shape[3] != 3:
        return None
    n = lib.ddpm_conv_s2h_weight_halves(w.shape[0], w.shape[1])
    if n == 0:
        return None
    out = torch.empty(n, dtype=torch.float16, device=w.device)
    check(lib.ddpm_pack_conv_s2h_weight(ptr(w), out.data_ptr(), w.shape[0], w.shape[1], stream_ptr()), "pack_conv_s2h_weight")
    return out


def conv(x, weight, bias=None, *, x2=None, gscale=None, gshift=None, act=ACT_NONE, mode=CONV_NORMAL,
         chan_add=None, chan_add_offset=0, residual=None, packed=None, force_direct=False, folded=None,
         wino=None, out_act=ACT_NONE, wino44=None, wino44h=None, want_stats=False, d3h=None):
    """Fused conv / linear.  x: [B, C1, H, W]; x2: optional second source of a virtual concat.
    want_stats: return (out, stats) with the produced tensor's per-channel GroupNorm statistics (ddpm_conv_desc.stats_out)."""
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
    if packed is False:  # the caller supplies the Winograd weights of a launch that takes them: no direct-MFMA form (the
        packed = None    # dispatcher's last resort is then the plain direct kernel on the raw weights)
    elif packed is None and not force_direct:
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
    d.w_wino44 = ptr(wino44)
    d.w_wino44h = wino44h.data_ptr() if wino44h is not None else None
    d.w_d3h = d3h.data_ptr() if d3h is not None else None
    d.out_act = out_act
    need = lib.ddpm_conv_scratch_floats(C.byref(d))  # small batches: split-K partial slabs (0 otherwise)
    if need:
        scratch = torch.empty(need, dtype=torch.float32, device=x.device)
        keep.append(scratch)
        d.scratch, d.scratch_floats = ptr(scratch), need
    stats = None
    if want_stats:  # GroupNorm statistics of the produced tensor from the epilogue: [B, Cout, parts, 2] (None: not emitted)
        parts = lib.ddpm_conv_stats_parts(C.byref(d))
        if parts > 0:
            stats = torch.empty((B, cout, parts, 2), dtype=torch.float32, device=x.device)
            d.stats_out = ptr(stats)
    check(lib.ddpm_conv_f32(C.byref(d), stream_ptr()), "conv")
    if want_stats:
        return out, stats
    return out[:, :, 0, 0] if was_linear else out


def conv_takes_wino44h(x_shape, cout: int) -> bool:
    """Would conv() run a plain stride-1 3x3 convolution of an input of this shape ([B, Cin, H, W]) to `cout` channels on the
    split-f16 F(4x4) kernel, given its packed weights?  (ddpm_conv_takes_wino44h: the dispatcher's own rule.)"""
    B, cin, H, W = x_shape
    d = ConvDesc()
    d.C1, d.B, d.Cout, d.Hi, d.Wi, d.Ho, d.Wo = cin, B, cout, H, W, H, W
    d.ksize, d.mode, d.act, d.out_act = 3, CONV_NORMAL, ACT_NONE, ACT_NONE
    return bool(_lib.load().ddpm_conv_takes_wino44h(C.byref(d)))


CONV_TRANSPOSE2 = 3
WINO_MAX_TENSOR_BYTES = 2 ** 31 - 1  # 32-bit buffer offsets of the Winograd kernels


def conv3d_supported(weight: torch.Tensor, stride: int = 1, transposed: bool = False) -> bool:
    """Does this conv3d / conv_transpose3d weight have an MFMA tiling?  k3 s1 p1 and k4 s2 p1 need Cin % 4 == 0 and
    Cout % 128 == 0; the transposed k4 s2 p1 (weight [Cin, Cout, 4, 4, 4]) needs Cin % 8 == 0 and Cout % 128 == 0."""
    if weight.ndim != 5:
        return False
    k = tuple(weight.shape[2:])
    if transposed:
        return k == (4, 4, 4) and stride == 2 and weight.shape[0] % 8 == 0 and weight.shape[1] % 128 == 0
    if not ((k == (3, 3, 3) and stride in (1, 2)) or (k == (4, 4, 4) and stride == 2)):
        return False
    return weight.shape[0] % 128 == 0 and weight.shape[1] % 4 == 0


def pack_conv3d_weight(weight: torch.Tensor) -> torch.Tensor:
    """torch [Cout, Cin, k, k, k] (k = 3 or 4) -> k MFMA-packed slabs of k x k taps, one per depth tap."""
    lib = _lib.load()
    w = require_device_f32(weight, "weight")
    cout, cin, k = w.shape[0], w.shape[1], w.shape[2]
    out = torch.empty(cout * cin * k ** 3, dtype=torch.float32, device=w.device)
    check(lib.ddpm_pack_conv3d_weight_f32(ptr(w), ptr(out), cout, cin, k, stream_ptr()), "pack_conv3d_weight")
    return out


def pack_wino3d_weight(weight: torch.Tensor) -> torch.Tensor | None:
    """torch [Cout, Cin, 3, 3, 3] -> per-depth-tap Winograd-domain weights U_kd = G w[:, :, kd] G^T (None if the
    shape has no Winograd tiling: Cout % 64, Cin % 8)."""
    lib = _lib.load()
    w = require_device_f32(weight, "weight")
    if w.ndim != 5 or tuple(w.shape[2:]) != (3, 3, 3) or lib.ddpm_wino_weight_floats(w.shape[0], w.shape[1]) == 0:
        return None
    out = torch.empty(3 * 16 * w.shape[0] * w.shape[1], dtype=torch.float32, device=w.device)
    check(lib.ddpm_pack_wino3d_weight_f32(ptr(w), ptr(out), w.shape[0], w.shape[1], stream_ptr()), "pack_wino3d_weight")
    return out


def pack_wino44_3d_weight(weight: torch.Tensor) -> torch.Tensor | None:
    """torch [Cout, Cin, 3, 3, 3] -> per-depth-tap F(4x4, 3x3) weights U_kd = G w[:, :, kd] G^T, 6 x 6 (None if the shape
    has no tiling: Cout % 64, Cin % 8)."""
    lib = _lib.load()
    w = require_device_f32(weight, "weight")
    if w.ndim != 5 or tuple(w.shape[2:]) != (3, 3, 3) or lib.ddpm_wino44_weight_floats(w.shape[0], w.shape[1]) == 0:
        return None
    out = torch.empty(3 * 36 * w.shape[0] * w.shape[1], dtype=torch.float32, device=w.device)
    check(lib.ddpm_pack_wino44_weight3d_f32(ptr(w), ptr(out), w.shape[0], w.shape[1], stream_ptr()), "pack_wino44_3d_weight")
    return out


def pack_convT_weight(weight: torch.Tensor) -> torch.Tensor:
    """torch ConvTranspose weight [Cin, Cout, 4, 4(, 4)] -> per-output-parity 2 x 2 (x 2)-tap packed weights."""
    lib = _lib.load()
    w = require_device_f32(weight, "weight")
    cin, cout, dims = w.shape[0], w.shape[1], w.ndim - 2
    n = lib.ddpm_packed_convtr_weight_floats(cout, cin, dims)
    if n == 0 or tuple(w.shape[2:]) != (4,) * dims:
        raise ValueError("pack_convT_weight: needs a [Cin % 8 == 0, Cout % 128 == 0, 4, 4(, 4)] weight")
    out = torch.empty(n, dtype=torch.float32, device=w.device)
    check(lib.ddpm_pack_convtr_weight_f32(ptr(w), ptr(out), cin, cout, dims, stream_ptr()), "pack_convT_weight")
    return out


def pack_wino44h_3d_weight(weight: torch.Tensor) -> torch.Tensor | None:
    """torch [Cout, Cin, 3, 3, 3] -> split-f16 F(4x4, 3x3) planes per depth tap (conv_wino44h.hip, 3-D form)."""
    lib = _lib.load()
    w = require_device_f32(weight, "weight")
    if w.ndim != 5 or tuple(w.shape[2:]) != (3, 3, 3):
        return None
    n = lib.ddpm_wino44h_weight_halves(w.shape[0], w.shape[1])
    if n == 0:
        return None
    out = torch.empty(3 * (n - 64) + 64, dtype=torch.float16, device=w.device)
    check(lib.ddpm_pack_wino44h_weight3d(ptr(w), out.data_ptr(), w.shape[0], w.shape[1], stream_ptr()), "pack_wino44h_3d_weight")
    return out


def conv3d(x, weight, bias=None, *, act=ACT_NONE, out_act=ACT_NONE, residual=None, packed=None, stride: int = 1,
           wino=None, out=None, wino44=None, wino44h=None, chan_add=None, depth_taps: int = 0):
    """F.conv3d(act(x), weight, bias, stride, padding=1) (+ chan_add[n, co] + residual, + output activation) on NCDHW tensors:
    kernel 3 stride 1 or 2 (the 3-D UNet's ResnetBlock / Downsample convolutions), or kernel 4 stride 2 (the VQ-VAE's).  ONE launch of the MFMA kernel: the depth taps are part of its chunk
    stream (chunk = (depth tap, channel group)), so the output is written once.  ``wino`` (pack_wino3d_weight): a
    stride-1 conv without input activation whose slices hold >= 64 2x2 tiles takes the Winograd kernel instead (2-D
    F(2x2, 3x3) per depth tap, the taps accumulated in the transform domain: 2.25x fewer multiplies); ``wino44``
    (pack_wino44_3d_weight): F(4x4, 3x3) per depth tap where a slice holds >= 32 4x4 tiles and the launch fills the chip."""
    lib = _lib.load()
    x = require_device_f32(x, "x")
    w = require_device_f32(weight, "weight")
    if not conv3d_supported(w, stride):
        raise ValueError("conv3d: only k3 s1 / k4 s2 weights with Cin % 4 == 0 and Cout % 128 == 0 have an MFMA tiling")
    B, Cc, D, H, W = x.shape
    cout, k = w.shape[0], w.shape[2]
    if Cc != w.shape[1]:
        raise ValueError(f"conv3d: input has {Cc} channels, weight expects {w.shape[1]}")
    if stride == 2 and (D < 2 or H < 2 or W < 2):
        raise ValueError("conv3d k4 s2: every extent must be >= 2")
    if stride == 1:
        Do, Ho, Wo = D, H, W
    elif k == 3:
        Do, Ho, Wo = (D + 1) // 2, (H + 1) // 2, (W + 1) // 2
    else:
        Do, Ho, Wo = D // 2, H // 2, W // 2
    if packed is None:
        packed = pack_conv3d_weight(w)
    if out is None:
        out = torch.empty((B, cout, Do, Ho, Wo), dtype=torch.float32, device=x.device)
    if bias is not None:
        bias = require_device_f32(bias, "bias")
    if residual is not None:
        residual = require_device_f32(residual, "residual")
    if (wino is not None or wino44 is not None or wino44h is not None) and stride == 1 and B > 1:
        # the Winograd kernel addresses pixels with 32-bit buffer offsets (tensors < 2 GiB): a larger batch is walked
        # in sub-batches (views along dim 0, no copies) instead of dropping to the direct kernel
        per = max(Cc, cout) * D * H * W * 4
        nb = max(1, WINO_MAX_TENSOR_BYTES // per)
        if B > nb:
            for s0 in range(0, B, nb):
                conv3d(x[s0:s0 + nb], w, bias, act=act, out_act=out_act, packed=packed, stride=stride, wino=wino,
                       wino44=wino44, wino44h=wino44h, residual=None if residual is None else residual[s0:s0 + nb],
                       out=out[s0:s0 + nb], chan_add=None if chan_add is None else chan_add[s0:s0 + nb], depth_taps=depth_taps)
            return out
    d = ConvDesc()
    d.in1, d.C1 = ptr(x), Cc
    d.w_packed = ptr(packed)
    d.bias, d.residual, d.out = ptr(bias), ptr(residual), ptr(out)
    d.B, d.Cout, d.Hi, d.Wi, d.Ho, d.Wo = B, cout, H, W, Ho, Wo
    d.ksize, d.mode, d.act, d.out_act = k, (CONV_NORMAL if stride == 1 else CONV_STRIDE2), act, out_act
    d.Di, d.Do, d.dims = D, Do, 3
    d.depth_taps = depth_taps  # 3 / 6: the first / last depth tap of the weight is all zeros (conv_transpose_parity)
    if chan_add is not None:
        chan_add = require_device_f32(chan_add, "chan_add")
        d.chan_add, d.chan_add_stride = ptr(chan_add), chan_add.shape[1]
    if wino is not None and stride == 1:
        d.w_wino = ptr(wino)
    if wino44 is not None and stride == 1:
        d.w_wino44 = ptr(wino44)
    if wino44h is not None and stride == 1:
        d.w_wino44h = wino44h.data_ptr()
    need = lib.ddpm_conv_scratch_floats(C_byref(d))  # launches smaller than the chip: split-K partial slabs
    if need:
        scratch = torch.empty(need, dtype=torch.float32, device=x.device)
        d.scratch, d.scratch_floats = ptr(scratch), need
    check(lib.ddpm_conv_f32(C_byref(d), stream_ptr()), "conv3d")
    return out


def conv_transpose(x, weight, bias=None, *, out_act=ACT_NONE, packed=None):
    """F.conv_transpose{2,3}d(x, weight, bias, stride=2, padding=1) (+ output activation) for kernel 4: one launch,
    grid.z = output parity, each parity a 2 x 2 (x 2)-tap convolution over the input."""
    lib = _lib.load()
    x = require_device_f32(x, "x")
    w = require_device_f32(weight, "weight")
    dims = x.ndim - 2
    if w.ndim != x.ndim or x.shape[1] != w.shape[0]:
        raise ValueError(f"conv_transpose: input {tuple(x.shape)} does not match weight {tuple(w.shape)}")
    if packed is None:
        packed = pack_convT_weight(w)
    B, Cc = x.shape[:2]
    D, H, W = ((1,) + tuple(x.shape[2:])) if dims == 2 else tuple(x.shape[2:])
    cout = w.shape[1]
    out = torch.empty((B, cout) + tuple(2 * e for e in x.shape[2:]), dtype=torch.float32, device=x.device)
    if bias is not None:
        bias = require_device_f32(bias, "bias")
    d = ConvDesc()
    d.in1, d.C1 = ptr(x), Cc
    d.w_packed = ptr(packed)
    d.bias, d.out = ptr(bias), ptr(out)
    d.B, d.Cout, d.Hi, d.Wi, d.Ho, d.Wo = B, cout, H, W, 2 * H, 2 * W
    d.ksize, d.mode, d.out_act = 4, CONV_TRANSPOSE2, out_act
    if dims == 3:
        d.Di, d.Do, d.dims = D, 2 * D, 3
    check(lib.ddpm_conv_f32(C_byref(d), stream_ptr()), "conv_transpose")
    return out


def conv_transpose_parity_supported(x, weight) -> bool:
    """ConvTranspose3d k4 s2 p1 whose eight parity convolutions (3x3x3 over the INPUT grid) have the split-f16 F(4x4) tiling:
    slices of at least 32 4x4 tiles in whole tile rows, channel counts with an MFMA tiling."""
    if x.ndim != 5 or weight.ndim != 5 or tuple(weight.shape[2:]) != (4, 4, 4):
        return False
    cin, cout = weight.shape[0], weight.shape[1]
    D, H, W = x.shape[2:]
    if D < 2 or H % 4 or W % 4 or cin % 16 or cout % 128:
        return False
    twc, thr = W // 4, H // 4
    tr = 32 // twc if twc and 32 % twc == 0 else 0  # tile rows per item
    # (the kernel's item: 32 tiles in whole tile rows, its 4 tr + 2 staged rows inside the slice)
    return tr > 0 and thr % tr == 0 and H >= 4 * tr + 2 and W <= 64


def pack_convT_parity_weights(weight):
    """[Cin, Cout, 4, 4, 4] -> per output parity q = 4 qz + 2 qy + qx: (the 3x3x3 weight [Cout, Cin, 3, 3, 3], its MFMA-packed form,
    its split-f16 F(4x4) planes)."""
    lib = _lib.load()
    w = require_device_f32(weight, "weight")
    cin, cout = w.shape[0], w.shape[1]
    g = torch.empty((8, cout, cin, 3, 3, 3), dtype=torch.float32, device=w.device)
    check(lib.ddpm_convtr3d_parity_weights_f32(ptr(w), ptr(g), cin, cout, stream_ptr()), "convtr3d_parity_weights")
    return [(g[q], pack_conv3d_weight(g[q]), pack_wino44h_3d_weight(g[q])) for q in range(8)]


def conv_transpose_parity(x, parity_weights, bias=None, *, out_act=ACT_NONE, sub_batch: int = 8):
    """F.conv_transpose3d(x, w, bias, stride 2, padding 1) (+ ReLU) as eight stride-1 3x3x3 convolutions -- each on the split-f16
    F(4x4) kernel, walking only its two non-zero depth taps -- and one interleave pass; `sub_batch` volumes at a time so that
    the eight parity tensors stay a bounded scratch (8 x the input extent x Cout)."""
    lib = _lib.load()
    x = require_device_f32(x, "x")
    B, _, D, H, W = x.shape
    cout = parity_weights[0][0].shape[0]
    out = torch.empty((B, cout, 2 * D, 2 * H, 2 * W), dtype=torch.float32, device=x.device)
    for s0 in range(0, B, sub_batch):
        xb = x[s0:s0 + sub_batch]
        nb = xb.shape[0]
        tmp = torch.empty((8, nb, cout, D, H, W), dtype=torch.float32, device=x.device)
        for q, (g, packed, w44h) in enumerate(parity_weights):
            conv3d(xb, g, bias, out_act=out_act, packed=packed, wino44h=w44h, out=tmp[q], depth_taps=6 if q & 4 else 3)
        check(lib.ddpm_parity_interleave3_f32(ptr(tmp), ptr(out[s0:s0 + nb]), nb * cout, D, H, W, stream_ptr()), "parity_interleave3")
    return out


def convnd_generic(x, weight, bias=None, *, stride=1, padding=1, transposed=False, residual=None, relu=False):
    """Generic (transposed) conv2d / conv3d on the HIP library for shapes without an MFMA tiling (any channel counts):
    x [B, Cin, (D,) H, W], torch weight layout, symmetric kernel / stride / padding, output_padding 0."""
    lib = _lib.load()
    x = require_device_f32(x, "x")
    w = require_device_f32(weight, "weight")
    dims = x.ndim - 2
    if dims not in (2, 3) or w.ndim != x.ndim or len(set(w.shape[2:])) != 1:
        raise ValueError("convnd_generic: needs x [B, C, (D,) H, W] and a cubic / square kernel")
    k = w.shape[2]
    cin, cout = (w.shape[0], w.shape[1]) if transposed else (w.shape[1], w.shape[0])
    if x.shape[1] != cin:
        raise ValueError(f"convnd_generic: input has {x.shape[1]} channels, weight expects {cin}")
    ext = (lambda e: (e - 1) * stride - 2 * padding + k) if transposed else (lambda e: (e + 2 * padding - k) // stride + 1)
    out = torch.empty((x.shape[0], cout) + tuple(ext(e) for e in x.shape[2:]), dtype=torch.float32, device=x.device)
    if bias is not None:
        bias = require_device_f32(bias, "bias")
    if residual is not None:
        residual = require_device_f32(residual, "residual")
        if residual.shape != out.shape:
            raise ValueError("convnd_generic: residual shape")
    D, H, W = (1,) * (3 - dims) + tuple(x.shape[2:])
    check(lib.ddpm_convnd_generic_f32(ptr(x), ptr(w), ptr(bias), ptr(residual), ptr(out), x.shape[0], cin, cout, D, H, W, dims, k,
                                      stride, padding, int(transposed), int(relu), stream_ptr()), "convnd_generic")
    return out


def conv3d_k4s2_cin1(x, weight, bias=None, relu: bool = False):
    """First VQ-VAE encoder layer: relu?(F.conv3d(x[B, 1, D, H, W], weight[Cout, 1, 4, 4, 4], bias, stride 2, pad 1))."""
    lib = _lib.load()
    x = require_device_f32(x, "x")
    w = require_device_f32(weight, "weight")
    if x.ndim != 5 or x.shape[1] != 1 or tuple(w.shape[1:]) != (1, 4, 4, 4):
        raise ValueError("conv3d_k4s2_cin1: needs x [B, 1, D, H, W] and weight [Cout, 1, 4, 4, 4]")
    B, _, D, H, W = x.shape
    out = torch.empty((B, w.shape[0], D // 2, H // 2, W // 2), dtype=torch.float32, device=x.device)
    if bias is not None:
        bias = require_device_f32(bias, "bias")
    check(lib.ddpm_conv3d_k4s2_cin1_f32(ptr(x), ptr(w), ptr(bias), ptr(out), B, w.shape[0], D, H, W, int(relu),
                                        stream_ptr()), "conv3d_k4s2_cin1")
    return out


def convT3d_k4s2_cout1(x, weight, bias=None):
    """Last VQ-VAE decoder layer: F.conv_transpose3d(x[B, Cin, D, H, W], weight[Cin, 1, 4, 4, 4], bias, stride 2, pad 1)."""
    lib = _lib.load()
    x = require_device_f32(x, "x")
    w = require_device_f32(weight, "weight")
    if x.ndim != 5 or tuple(w.shape) != (x.shape[1], 1, 4, 4, 4):
        raise ValueError("convT3d_k4s2_cout1: needs x [B, Cin, D, H, W] and weight [Cin, 1, 4, 4, 4]")
    B, Cc, D, H, W = x.shape
    out = torch.empty((B, 1, 2 * D, 2 * H, 2 * W), dtype=torch.float32, device=x.device)
    if bias is not None:
        bias = require_device_f32(bias, "bias")
    check(lib.ddpm_convtr3d_k4s2_cout1_f32(ptr(x), ptr(w), ptr(bias), ptr(out), B, Cc, D, H, W, stream_ptr()),
          "convT3d_k4s2_cout1")
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


def gn_finalize(stats, gamma, beta, groups: int, eps: float, hw: int, stats2=None):
    """scale / shift of F.group_norm from per-channel statistics slabs [B, C, parts, 2] = {mean, M2} per slice (the
    `stats` of conv(..., want_stats=True) or channel_stats); stats2: second source of a virtual concat."""
    lib = _lib.load()
    stats = require_device_f32(stats, "stats")
    B, C1, parts1, _ = stats.shape
    C2 = parts2 = 0
    if stats2 is not None:
        stats2 = require_device_f32(stats2, "stats2")
        _, C2, parts2, _ = stats2.shape
    gamma = require_device_f32(gamma, "gamma")
    beta = require_device_f32(beta, "beta")
    scale = torch.empty((B, C1 + C2), dtype=torch.float32, device=stats.device)
    shift = torch.empty_like(scale)
    check(lib.ddpm_gn_finalize_f32(ptr(stats), parts1, C1, ptr(stats2), parts2, C2, ptr(gamma), ptr(beta), ptr(scale),
                                   ptr(shift), B, hw, groups, eps, stream_ptr()), "gn_finalize")
    return scale, shift


def channel_stats(x):
    """[B, C, 1, 2] = per-channel {mean, sum of squared deviations} of x [B, C, ...]."""
    lib = _lib.load()
    x = require_device_f32(x, "x")
    B, Cc = x.shape[:2]
    hw = x[0, 0].numel()
    stats = torch.empty((B, Cc, 1, 2), dtype=torch.float32, device=x.device)
    check(lib.ddpm_channel_stats_f32(ptr(x), ptr(stats), B, Cc, hw, stream_ptr()), "channel_stats")
    return stats


def attention(qkv, residual, num_heads: int, scale: float, use_scratch: bool = True):
    """qkv: [B, 3C, N] -> [B, C, N] = softmax(scale q^T k) v (+ residual).  use_scratch: give the library the scratch its
    register-resident kernel (csrc/attention_fa.hip) wants for the f16 planes of q / k / v, as the UNet engine does."""
    lib = _lib.load()
    qkv = require_device_f32(qkv, "qkv")
    B, C3, N = qkv.shape
    Cc = C3 // 3
    if residual is not None:
        residual = require_device_f32(residual, "residual")
    out = torch.empty((B, Cc, N), dtype=torch.float32, device=qkv.device)
    nscr = lib.ddpm_attention_scratch_floats(B, Cc, N, num_heads) if use_scratch else 0
    if nscr:
        scratch = torch.empty(nscr, dtype=torch.float32, device=qkv.device)
        check(lib.ddpm_attention_ws_f32(ptr(qkv), ptr(residual), ptr(out), B, Cc, N, num_heads, scale, ptr(scratch), nscr,
                                        stream_ptr()), "attention")
    else:
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


def lpips_conv_biasmap(x, weight, bias_map, stride: int, pad: int, relu: bool = True):
    """relu(conv2d(x, weight, None, stride, pad) + bias_map[None]); bias_map: [Cout, Ho, Wo]."""
    lib = _lib.load()
    x = require_device_f32(x, "x")
    w = require_device_f32(weight, "weight")
    bias_map = require_device_f32(bias_map, "bias_map")
    N, cin, H, W = x.shape
    cout, _, k, _ = w.shape
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    if tuple(bias_map.shape[-3:]) != (cout, Ho, Wo) or w.shape[1] != cin:
        raise ValueError(f"lpips_conv_biasmap: bias_map {tuple(bias_map.shape)} / weight {tuple(w.shape)} do not fit the input")
    out = torch.empty((N, cout, Ho, Wo), dtype=torch.float32, device=x.device)
    check(lib.ddpm_lpips_conv_biasmap_f32(ptr(x), ptr(w), ptr(bias_map), ptr(out), N, cin, H, W, cout, k, stride, pad,
                                          int(relu), stream_ptr()), "lpips_conv_biasmap")
    return out


def lpips_pack_conv_weight(weight):
    """torch [Cout, Cin, k, k] -> the MFMA operand layout of lpips_conv_mfma (Cout % 32 == 0, Cin % 2 == 0)."""
    lib = _lib.load()
    w = require_device_f32(weight, "weight")
    cout, cin, k, _ = w.shape
    out = torch.empty(w.numel(), dtype=torch.float32, device=w.device)
    check(lib.ddpm_lpips_pack_conv_weight_f32(ptr(w), ptr(out), cout, cin, k, stream_ptr()), "lpips_pack_conv_weight")
    return out


def lpips_conv_mfma_supported(cin: int, h: int, w: int, cout: int, k: int) -> bool:
    return bool(_lib.load().ddpm_lpips_conv_mfma_supported(cin, h, w, cout, k))


def lpips_conv_mfma(x, packed, bias, cout: int, k: int, relu: bool = True):
    """relu(conv2d(x, w, bias, stride 1, padding k // 2)) on the fp32 MFMA pipe; packed = lpips_pack_conv_weight(w)."""
    lib = _lib.load()
    x = require_device_f32(x, "x")
    N, cin, H, W = x.shape
    out = torch.empty((N, cout, H, W), dtype=torch.float32, device=x.device)
    bias = None if bias is None else require_device_f32(bias, "bias")
    check(lib.ddpm_lpips_conv_mfma_f32(ptr(x), ptr(packed), ptr(bias), ptr(out), N, cin, H, W, cout, k, int(relu),
                                       stream_ptr()), "lpips_conv_mfma")
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
