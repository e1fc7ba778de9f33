"""Python wrappers over the training-step entry points of the C ABI (include/ddpm_ood_hip.h, "Training step", ABI 10).

Same rules as ``ops.py``: ROCm device tensors only, launches on torch's current HIP stream, no PyTorch compute -- the only
torch calls are ``torch.empty`` (the caching allocator hands out memory; it launches nothing).  Used by
``train_native.py`` (row f-3: /root/reference/src/trainers/ddpm_trainer.py:78-109) and by tests/test_gpu_train_ops.py.
"""

from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import GemmDesc, check, ptr, require_device_f32, stream_ptr

ACT_NONE, ACT_SILU = 0, 1


def _empty(shape, like):
    return torch.empty(shape, dtype=torch.float32, device=like.device)


def gemm(A, B, Cout, M, N, K, *, a_m, a_k, b_k, b_n, c_m, c_n, batch=1, a_batch=0, b_batch=0, c_batch=0, k_inner=0,
         a_k_outer=0, b_k_outer=0, batch_inner=0, a_batch_outer=0, b_batch_outer=0, c_batch_outer=0, alpha=1.0, beta=0.0,
         a_off=0, b_off=0, c_off=0, split_f16=False):
    """C = alpha A B + beta C with element strides (ddpm_gemm_desc).  *_off: element offsets into the three tensors."""
    g = GemmDesc()
    g.A, g.B, g.C = A.data_ptr() + 4 * a_off, B.data_ptr() + 4 * b_off, Cout.data_ptr() + 4 * c_off
    g.M, g.N, g.K, g.k_inner = M, N, K, k_inner
    g.a_m, g.a_k, g.a_k_outer = a_m, a_k, a_k_outer
    g.b_n, g.b_k, g.b_k_outer = b_n, b_k, b_k_outer
    g.c_m, g.c_n = c_m, c_n
    g.batch, g.batch_inner = batch, batch_inner
    g.a_batch, g.a_batch_outer, g.b_batch, g.b_batch_outer = a_batch, a_batch_outer, b_batch, b_batch_outer
    g.c_batch, g.c_batch_outer = c_batch, c_batch_outer
    g.alpha, g.beta = alpha, beta
    g.split_f16 = int(split_f16)  # both operands inside the f16 exponent range (the caller's promise): K-major products on the f16 MFMA
    lib = _lib.load()
    need = lib.ddpm_gemm_scratch_floats(C.byref(g))  # K slices for products with a small (M, N, batch) grid and a long K
    if need:
        scratch = torch.empty(need, dtype=torch.float32, device=A.device)
        g.scratch, g.scratch_floats = scratch.data_ptr(), need
    check(lib.ddpm_gemm_f32(C.byref(g), stream_ptr()), "gemm")
    return Cout


def conv_wgrad(a, dy, ksize: int, stride: int = 1, out=None, force_generic: bool = False, a_absmax=None, dy_absmax=None):
    """dw[Cout, Cin, k, k] of F.conv2d(a, w, stride=stride, padding=k // 2) given dy.  a_absmax / dy_absmax: int32 tensors of float
    bit patterns whose largest is the largest |a| / |dy| (gn_forward / gn_backward emit them): the split-f16 form then does not
    read the tensor once more to measure it."""
    lib = _lib.load()
    a, dy = require_device_f32(a, "a"), require_device_f32(dy, "dy")
    B, Cin, Hi, Wi = a.shape
    Cout, Ho, Wo = dy.shape[1:]
    if out is None:
        out = _empty((Cout, Cin, ksize, ksize), a)
    need = 0 if force_generic else lib.ddpm_conv_wgrad_scratch_floats(B, Cin, Cout, Hi, Wi, Ho, Wo, ksize, stride)
    scratch = _empty((need,), a) if need else None
    check(lib.ddpm_conv_wgrad_f32(ptr(a), ptr(dy), ptr(out), B, Cin, Cout, Hi, Wi, Ho, Wo, ksize, stride, ptr(scratch), need,
                                  int(force_generic), ptr(a_absmax), 0 if a_absmax is None else a_absmax.numel(), ptr(dy_absmax),
                                  0 if dy_absmax is None else dy_absmax.numel(), stream_ptr()), "conv_wgrad")
    return out


def conv_weight_rot180t(w, out=None):
    """[Cout, Cin, k, k(, k)] (or [Cout, Cin]) -> [Cin, Cout, k, k(, k)]: the weights of the input-gradient convolution."""
    w = require_device_f32(w, "w")
    cout, cin = w.shape[:2]
    spatial = tuple(w.shape[2:])
    taps = 1
    for k in spatial:
        taps *= k
    if out is None:
        out = _empty((cin, cout) + (spatial if spatial else (1, 1)), w)
    check(_lib.load().ddpm_conv_weight_rot180t_f32(ptr(w), ptr(out), cout, cin, taps, stream_ptr()), "conv_weight_rot180t")
    return out


def conv3d_wgrad(a, dy, stride: int = 1, out=None):
    """dw[Cout, Cin, 3, 3, 3] of F.conv3d(a, w, stride=stride, padding=1) given dy (NCDHW)."""
    lib = _lib.load()
    a, dy = require_device_f32(a, "a"), require_device_f32(dy, "dy")
    B, Cin, Di, Hi, Wi = a.shape
    Cout, Do, Ho, Wo = dy.shape[1:]
    if out is None:
        out = _empty((Cout, Cin, 3, 3, 3), a)
    need = lib.ddpm_conv3d_wgrad_scratch_floats(B, Cin, Cout, Di, Hi, Wi, Do, Ho, Wo, stride)
    if need == 0:
        raise ValueError("conv3d_wgrad: needs Cin % 64 == 0, Cout % 64 == 0 and an even W <= 64")
    scratch = _empty((need,), a)
    check(lib.ddpm_conv3d_wgrad_f32(ptr(a), ptr(dy), ptr(out), B, Cin, Cout, Di, Hi, Wi, Do, Ho, Wo, stride, ptr(scratch), need,
                                    stream_ptr()), "conv3d_wgrad")
    return out


def gn_stats(x, groups: int, eps: float):
    x = require_device_f32(x, "x")
    B, Cc = x.shape[:2]
    mr = _empty((B, groups, 2), x)
    check(_lib.load().ddpm_gn_stats_f32(ptr(x), ptr(mr), B, Cc, x[0, 0].numel(), groups, eps, stream_ptr()), "gn_stats")
    return mr


def gn_apply(x, mean_rstd, gamma, beta, groups: int, act: int = ACT_NONE):
    y = torch.empty_like(x)
    B, Cc = x.shape[:2]
    check(_lib.load().ddpm_gn_apply_f32(ptr(x), ptr(mean_rstd), ptr(gamma), ptr(beta), ptr(y), B, Cc, x[0, 0].numel(), groups, act,
                                        stream_ptr()), "gn_apply")
    return y


def gn_forward(x, gamma, beta, groups: int, eps: float, act: int = ACT_NONE, want_absmax: bool = False):
    """(y, mean_rstd) = gn_apply(x, gn_stats(x)) in one call -- one kernel for the UNet's plane sizes.  want_absmax: also the
    [B * groups] int32 tensor of the float bit patterns of max |y| per (image, group) (conv_wgrad's a_absmax)."""
    x = require_device_f32(x, "x")
    B, Cc = x.shape[:2]
    y, mr = torch.empty_like(x), _empty((B, groups, 2), x)
    amax = torch.empty((B * groups,), dtype=torch.int32, device=x.device) if want_absmax else None
    check(_lib.load().ddpm_gn_forward_f32(ptr(x), ptr(gamma), ptr(beta), ptr(y), ptr(mr), ptr(amax), B, Cc, x[0, 0].numel(), groups, eps,
                                          act, stream_ptr()), "gn_forward")
    return (y, mr, amax) if want_absmax else (y, mr)


def gn_backward(x, dy, mean_rstd, gamma, beta, groups: int, act: int, dgamma, dbeta, dx=None, accumulate: bool = False,
                want_absmax: bool = False, want_rowsum: bool = False):
    """dx (+= if accumulate) and the parameter gradients.  want_absmax / want_rowsum: returns (dx, absmax, rowsum) -- the bit
    patterns of max |dx| per (image, group) (conv_wgrad's dy_absmax) and the [B, C] plane sums of the final dx (None if not asked)."""
    B, Cc = x.shape[:2]
    if dx is None:
        dx, accumulate = torch.empty_like(x), False
    ws = _empty((B, Cc, 2), x)
    amax = torch.empty((B * groups,), dtype=torch.int32, device=x.device) if want_absmax else None
    rows = _empty((B, Cc), x) if want_rowsum else None
    check(_lib.load().ddpm_gn_backward_f32(ptr(x), ptr(dy), ptr(mean_rstd), ptr(gamma), ptr(beta), ptr(dx), int(accumulate),
                                           ptr(dgamma), ptr(dbeta), ptr(ws), ptr(amax), ptr(rows), B, Cc, x[0, 0].numel(), groups, act,
                                           stream_ptr()), "gn_backward")
    return (dx, amax, rows) if (want_absmax or want_rowsum) else dx


def row_sum(x, rows: int, cols: int, out=None):
    if out is None:
        out = _empty((rows,), x)
    check(_lib.load().ddpm_row_sum_f32(ptr(x), ptr(out), rows, cols, stream_ptr()), "row_sum")
    return out


def col_sum(x, rows: int, cols: int, out=None, row_stride=None, alpha: float = 1.0, accumulate: bool = False):
    if out is None:
        out = _empty((cols,), x)
    check(_lib.load().ddpm_col_sum_f32(ptr(x), ptr(out), rows, cols, cols if row_stride is None else row_stride, alpha,
                                       int(accumulate), stream_ptr()), "col_sum")
    return out


def silu(x):
    y = torch.empty_like(x)
    check(_lib.load().ddpm_silu_f32(ptr(x), ptr(y), x.numel(), stream_ptr()), "silu")
    return y


def silu_backward(x, dy):
    dx = torch.empty_like(x)
    check(_lib.load().ddpm_silu_backward_f32(ptr(x), ptr(dy), ptr(dx), x.numel(), stream_ptr()), "silu_backward")
    return dx


def axpby(a, b=None, alpha: float = 1.0, beta: float = 1.0, out=None):
    if out is None:
        out = torch.empty_like(a)
    check(_lib.load().ddpm_axpby_f32(ptr(a), ptr(b), ptr(out), alpha, beta, a.numel(), stream_ptr()), "axpby")
    return out


def scale_check_(x, alpha: float):
    """x *= alpha in place; a non-finite value sets _lib.STATUS_NONFINITE_GRAD in the device status word."""
    check(_lib.load().ddpm_scale_check_f32(ptr(x), alpha, x.numel(), stream_ptr()), "scale_check")
    return x


def chan_copy(src, dst, C_, csrc0: int = 0, cdst0: int = 0, accumulate: bool = False):
    B = src.shape[0]
    check(_lib.load().ddpm_chan_copy_f32(ptr(src), ptr(dst), B, C_, src.shape[1], csrc0, dst.shape[1], cdst0, src[0, 0].numel(),
                                         int(accumulate), stream_ptr()), "chan_copy")
    return dst


def _resample(x, mode: int):
    """mode 0: nearest x2; 1: its adjoint (2x2(x2) block sums); 2: zero-stuffing x2 -- on [B, C, H, W] or [B, C, D, H, W]."""
    lib = _lib.load()
    B, Cc = x.shape[:2]
    sp = tuple(x.shape[2:])
    small = tuple(v // 2 for v in sp) if mode == 1 else sp
    out = _empty((B, Cc) + (small if mode == 1 else tuple(2 * v for v in sp)), x)
    if len(sp) == 2:
        check(lib.ddpm_resample2_f32(ptr(x), ptr(out), B * Cc, small[0], small[1], mode, stream_ptr()), "resample2")
    else:
        check(lib.ddpm_resample3_f32(ptr(x), ptr(out), B * Cc, small[0], small[1], small[2], mode, stream_ptr()), "resample3")
    return out


def upsample2(x):
    return _resample(x, 0)


def sumpool2(x):
    return _resample(x, 1)


def zero_stuff2(x):
    return _resample(x, 2)


def softmax_rows_(s, rows: int, cols: int):
    check(_lib.load().ddpm_softmax_rows_f32(ptr(s), rows, cols, stream_ptr()), "softmax_rows")
    return s


def softmax_backward_rows_(p, dp, rows: int, cols: int):
    check(_lib.load().ddpm_softmax_backward_rows_f32(ptr(p), ptr(dp), rows, cols, stream_ptr()), "softmax_backward_rows")
    return dp


def mse_loss_grad(pred, target):
    """(loss [1] on the device, dpred) of F.mse_loss(pred, target)."""
    n = pred.numel()
    nb = (n + 255) // 256
    dpred, partial = torch.empty_like(pred), _empty((nb,), pred)
    check(_lib.load().ddpm_mse_loss_grad_f32(ptr(pred), ptr(target), ptr(dpred), ptr(partial), n, 2.0 / n, stream_ptr()),
          "mse_loss_grad")
    # partial viewed as [nb rows, 1 column] -> one value, scaled by 1 / n
    loss = col_sum(partial, nb, 1, row_stride=1, alpha=1.0 / n)
    return loss, dpred


def fill_(t, value: float):
    check(_lib.load().ddpm_fill_f32(ptr(t), value, t.numel(), stream_ptr()), "fill")
    return t


def randn(shape, device, seed: int, stream_id: int):
    out = torch.empty(shape, dtype=torch.float32, device=device)
    check(_lib.load().ddpm_randn_f32(ptr(out), out.numel(), seed & (2 ** 64 - 1), stream_id & (2 ** 64 - 1), stream_ptr()), "randn")
    return out


def adam_step_(p, g, m, v, lr: float, beta1: float, beta2: float, eps: float, step: int, grad_scale: float = 1.0):
    check(_lib.load().ddpm_adam_step_f32(ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), lr, beta1, beta2, eps, step, grad_scale,
                                         stream_ptr()), "adam_step")
