"""Native training step of the 2-D DiffusionModelUNet: forward, backward and Adam on hand-written HIP kernels only
(the step of /root/reference/src/trainers/ddpm_trainer.py:78-109 with the optimiser of
/root/reference/src/trainers/base.py:156).

``NativeUNetStep`` evaluates the SAME parameter holders the inference engine reads (``unet.DiffusionModelUNet``, MONAI-Generative
key names) -- a checkpoint written after native steps loads into the reconstruction path unchanged -- and keeps every
parameter / gradient / Adam moment in ONE flat device buffer each (the holders' ``.data`` and ``.grad`` become views of them): one
Adam launch, one all-reduce payload.  No ATen / MIOpen / rocBLAS kernel runs between handing over (noisy, timesteps, noise) and
the updated parameters; ``torch.empty`` only asks the caching allocator for memory.

Forward (activations kept for the backward):
  GroupNorm statistics + a MATERIALISED GroupNorm/SiLU output (ddpm_gn_forward_f32: one kernel) -> the inference path's
  convolution kernels (ddpm_conv_f32: split-f16 F(4x4) Winograd where a launch fills the chip, its bias / temb / residual
  epilogue), attention as two batched GEMMs around a row softmax (the probabilities are kept).
Backward (on the loss gradient times a power of two -- 1e-6-sized gradients are below f16's normal range --, the factor taken out
of the flat gradient buffer by ddpm_scale_check_f32, which also flags an overflow; see loss_and_grads):
  3x3 / 1x1 input gradients = ddpm_conv_f32 with the weights rotated by 180 degrees and transposed; Downsample through a
  zero-stuffed dY; Upsample followed by a 2x2 sum; 3x3 weight gradients on ddpm_conv_wgrad_f32 (f16 MFMA at split precision,
  64 x 64 x 9 taps per workgroup; its operand maxima and the bias / temb plane sums come from the GroupNorm kernels that wrote the
  operands); every other contraction (Linear / 1x1 weight and input gradients, the five attention products) on ddpm_gemm_f32;
  GroupNorm + SiLU backward, bias sums, SiLU backward, softmax backward, MSE, Adam in train_ops.hip.
"""

from __future__ import annotations

import math
import os

import torch

from . import ops
from . import train_ops as T

SILU, NONE = T.ACT_SILU, T.ACT_NONE


def native_supported(model) -> bool:
    """Every 2-D UNet; 3-D UNets whose convolutions all have an MFMA tiling (the latent UNet of the LDM configuration, BASELINE
    configs[4]: 128 latent channels in and out, num_channels multiples of 128) -- the conv3d forward and weight-gradient kernels
    have no generic form; a 3-D UNet over 1-channel volumes keeps the ATen route."""
    sd = getattr(model, "spatial_dims", 2)
    if sd == 2:
        return True
    return (sd == 3 and model.in_channels % 64 == 0 and model.out_channels % 128 == 0
            and all(c % 128 == 0 for c in model.block_out_channels))


class NativeUNetStep:
    """Build it AFTER the model is on its device: construction re-homes every parameter into one flat buffer (``.data`` becomes a
    view of ``self.flat``, ``.grad`` a view of ``self.gflat``); a later ``model.to(...)`` / ``load_state_dict(assign=True)`` would
    replace those views and silently detach the model from the stepper (``load_state_dict`` with the default ``assign=False``
    copies INTO the views and is fine -- that is how a checkpoint is resumed)."""

    def __init__(self, model, lr: float = 2.5e-5, betas=(0.9, 0.999), eps: float = 1e-8):
        if not native_supported(model):
            raise NotImplementedError("native training covers 2-D UNets and 3-D UNets whose channel counts have an MFMA tiling")
        self.model = model
        self.lr, self.betas, self.eps = lr, betas, eps
        self.step_count = 0
        self.G, self.gn_eps = model.norm_num_groups, model.norm_eps
        # Which 3x3 kernel multiplies (see _conv).  Forward and input gradients both run on the inference path's split-f16 F(4x4)
        # kernel.  Its operands must sit in the f16 exponent range, and a gradient does not: |dY| is 1e-7 .. 1e-4 at batch 256, below
        # f16's smallest normal (6e-5) -- unscaled, the input gradients lost the low half of the split and a parameter gradient was
        # 1.5e-4 off at the first layers (which round 6 first read as F(4x4)'s transform rounding and answered with F(2x2) on the
        # fp32 pipe).  Every backward kernel is linear in the gradient, so the backward runs on dpred * 2^k and the flat gradient
        # buffer is multiplied by 2^-k afterwards (exact): 4e-7 .. 7e-6 (profiles/r06_train_loss_scale_grad_error.log).
        #   DDPM_TRAIN_DGRAD=wino: F(2x2) on the fp32 MFMA (needs no scale; 6e-6; ~15 % slower steps)
        #   DDPM_TRAIN_LOSS_SCALE=<power of two>: a fixed scale instead of the adaptive one (1: none)
        self.fwd_form = os.environ.get("DDPM_TRAIN_FWD", "wino44h")
        self.dgrad_form = os.environ.get("DDPM_TRAIN_DGRAD", "wino44h")
        ls = os.environ.get("DDPM_TRAIN_LOSS_SCALE", "")
        self.loss_scale = float(ls) if ls else None  # None: adaptive (set from the first batch's element count, lowered on overflow)
        self.scale_adaptive = self.loss_scale is None and self.dgrad_form == "wino44h"
        if self.loss_scale is None and not self.scale_adaptive:
            self.loss_scale = 1.0
        self._w44_cache = {}           # (input shape, couts) -> does the dispatcher take the F(4x4) kernel
        self._scaled = False           # the running backward's gradients are scaled into f16's range
        self.overflow_retries = 0      # backward passes repeated at a lower scale
        self.fp32_dgrad_steps = 0      # steps whose input gradients fell back to the fp32 pipe
        self._clean_steps = 0
        # operand maxima / plane sums taken from the GroupNorm kernels that wrote the tensors (0: every consumer reads them again; A/B)
        self.fused_stats = os.environ.get("DDPM_TRAIN_FUSED_STATS", "1") != "0"
        self.conv1_dgrad_conv = os.environ.get("DDPM_TRAIN_1X1_DGRAD", "conv") != "gemm"
        self._flatten()

    # ---- flat parameter / gradient / moment buffers ----------------------------------------------------------------------
    def _flatten(self):
        params = list(self.model.parameters())
        dev = params[0].device
        if dev.type != "cuda":
            raise RuntimeError("native training needs the model on a ROCm device (no CPU fallback)")
        total = sum(p.numel() for p in params)
        self.flat = torch.empty(total, dtype=torch.float32, device=dev)
        self.gflat = torch.empty(total, dtype=torch.float32, device=dev)
        self.m = torch.empty(total, dtype=torch.float32, device=dev)
        self.v = torch.empty(total, dtype=torch.float32, device=dev)
        for t in (self.gflat, self.m, self.v):
            T.fill_(t, 0.0)
        self.offsets = {}
        off = 0
        for p in params:
            n = p.numel()
            view = self.flat[off: off + n].view(p.shape)
            src = p.data.float().contiguous()
            T.axpby(src.view(-1), None, 1.0, 0.0, out=self.flat[off: off + n])  # copy (a HIP kernel of this library)
            p.data = view
            p.grad = self.gflat[off: off + n].view(p.shape)
            self.offsets[id(p)] = (off, n)
            off += n
        self.params = params
        self.model._plist = None  # the engine's change detector looks at data pointers / versions
        self.model._synced_key = None

    def g(self, p):
        return p.grad

    # ---- primitives --------------------------------------------------------------------------------------------------------
    def _conv(self, x, w, b=None, *, chan_add=None, residual=None, stride2=False, form=None):
        """ddpm_conv_f32.  form of a stride-1 3x3: "wino44h" = split-f16 F(4x4) Winograd where the launch fills the chip (the
        inference path's kernel: ~3e-6 rms relative rounding per convolution, from the 6x6 transforms), "wino" = F(2x2) on the
        fp32 MFMA (~1e-6), "direct" = the fp32 MFMA direct form (bit-exact fp32 products).  5-D inputs: F.conv3d on NCDHW (a
        1x1x1 convolution / per-voxel Linear is the 2-D op over a (D H) x W image)."""
        form = form or self.fwd_form
        if x.ndim == 5:
            if w.ndim == 5 and w.shape[2] == 3:
                s1 = not stride2
                return ops.conv3d(x, w, b, chan_add=chan_add, residual=residual, stride=1 if s1 else 2,
                                  wino44h=ops.pack_wino44h_3d_weight(w) if s1 and form == "wino44h" else None,
                                  wino=ops.pack_wino3d_weight(w) if s1 and form in ("wino44h", "wino") else None)
            B, Cc, D, H, W = x.shape
            w2 = w.reshape(w.shape[0], w.shape[1], 1, 1)
            r2 = None if residual is None else residual.reshape(B, -1, D * H, W)
            return ops.conv(x.reshape(B, Cc, D * H, W), w2, b, chan_add=chan_add, residual=r2).reshape(B, w.shape[0], D, H, W)
        if stride2:
            return ops.conv(x, w, b, mode=ops.CONV_STRIDE2, wino44h=ops.pack_conv_s2h_weight(w))
        if w.ndim == 4 and w.shape[2] == 3:
            # (every step re-packs: the weights changed.  ONE form is packed: F(4x4) where the dispatcher would take it for this
            # shape -- asked once per shape --, else F(2x2); the direct-MFMA packing would be an unused third)
            w44 = form == "wino44h" and self._takes_wino44h(tuple(x.shape), w.shape[0])
            return ops.conv(x, w, b, chan_add=chan_add, residual=residual,
                            wino44h=ops.pack_wino44h_weight(w) if w44 else None,
                            wino=ops.pack_wino_weight(w) if form in ("wino44h", "wino") and not w44 else None,
                            packed=False if form in ("wino44h", "wino") else None)
        return ops.conv(x, w, b, chan_add=chan_add, residual=residual)

    def _takes_wino44h(self, x_shape, cout):
        key = (x_shape, cout)
        if key not in self._w44_cache:
            self._w44_cache[key] = ops.conv_takes_wino44h(x_shape, cout)
        return self._w44_cache[key]

    def _wgrad3(self, a, dy, w, stride=1, a_absmax=None, dy_absmax=None):
        """The 3x3(x3) weight gradient into the weight's gradient view.  *_absmax: the operand maxima where the kernel that
        wrote the operand emitted them (GroupNorm forward / backward) -- otherwise the split-f16 form reads the operand to find it."""
        if a.ndim == 5:
            T.conv3d_wgrad(a, dy, stride, out=self.g(w))
        else:
            T.conv_wgrad(a, dy, 3, stride, out=self.g(w), a_absmax=a_absmax, dy_absmax=dy_absmax)

    def _bias_grad(self, dy, bias_param, keep_rows=False):
        B, Cc = dy.shape[:2]
        rows = T.row_sum(dy, B * Cc, dy[0, 0].numel())  # [B, C]: also the temb gradient of a ResnetBlock
        if bias_param is not None:
            T.col_sum(rows, B, Cc, out=self.g(bias_param))
        return rows if keep_rows else None

    def _linear_fwd(self, x, lin):
        return ops.conv(x, lin.weight, lin.bias)  # [B, Cin] -> [B, Cout]

    def _linear_bwd(self, x, lin, dy, dx=None, accumulate=False, need_dx=True):
        B, cin = x.shape
        cout = dy.shape[1]
        # dW[o, i] = sum_b dy[b, o] x[b, i]
        T.gemm(dy, x, self.g(lin.weight), cout, cin, B, a_m=1, a_k=cout, b_k=cin, b_n=1, c_m=cin, c_n=1)
        T.col_sum(dy, B, cout, out=self.g(lin.bias))
        if not need_dx:
            return None
        if dx is None:
            dx, accumulate = torch.empty_like(x), False
        # dx[b, i] = sum_o dy[b, o] W[o, i]
        T.gemm(dy, lin.weight, dx, B, cin, cout, a_m=cout, a_k=1, b_k=cin, b_n=1, c_m=cin, c_n=1, beta=1.0 if accumulate else 0.0)
        return dx

    def _gn_fwd(self, x, norm, act):
        """y = act(GroupNorm(x)) and the backward's context; ctx[4] = the maxima of |y| per (image, group) (2-D: _wgrad3's a_absmax)."""
        if x.ndim == 4:
            y, mr, amax = T.gn_forward(x, norm.weight, norm.bias, self.G, self.gn_eps, act, want_absmax=True)
        else:
            (y, mr), amax = T.gn_forward(x, norm.weight, norm.bias, self.G, self.gn_eps, act), None
        return y, (x, mr, norm, act, amax)

    def _gn_bwd(self, ctx, da, dx=None, accumulate=False, want=False):
        """want: returns (dx, maxima of |dx| per (image, group), [B, C] plane sums of dx) -- the weight gradient's dy_absmax and the
        bias / time-embedding gradient of the convolution whose output gradient dx is."""
        x, mr, norm, act = ctx[:4]
        return T.gn_backward(x, da, mr, norm.weight, norm.bias, self.G, act, self.g(norm.weight), self.g(norm.bias), dx=dx,
                             accumulate=accumulate, want_absmax=want and x.ndim == 4, want_rowsum=want)

    def _conv3_bwd(self, a, w, dy, need_dx=True, a_absmax=None, dy_absmax=None):
        """dW into the weight's gradient view; returns da = conv(dy, rot180(w)^T)."""
        self._wgrad3(a, dy, w, a_absmax=a_absmax, dy_absmax=dy_absmax)
        if not need_dx:
            return None
        return self._conv(dy, T.conv_weight_rot180t(w), form=self.dgrad_form)

    def _conv1_wgrad(self, x, w, dy):
        """1x1 convolution / per-pixel Linear: dW[o, i] = sum over (image, pixel) dy[b, o, p] x[b, i, p]."""
        B, cin = x.shape[:2]
        cout, hw = dy.shape[1], dy[0, 0].numel()
        # (under the gradient scale dy sits in the f16 exponent range like the activation x: the product multiplies on the f16 MFMA
        # at split precision; an unscaled backward keeps the fp32 MFMA)
        T.gemm(dy, x, self.g(w), cout, cin, B * hw, k_inner=hw, a_m=hw, a_k=1, a_k_outer=cout * hw, b_n=hw, b_k=1,
               b_k_outer=cin * hw, c_m=cin, c_n=1, split_f16=self._scaled)

    def _conv1_dgrad(self, w, dy, dx, accumulate):
        """dx[b, i, p] (+)= sum_o W[o, i] dy[b, o, p]; returns the result (a new tensor on the convolution route).  A 1x1 input
        gradient IS a 1x1 convolution with the transposed weight: it runs on the inference path's DMA-fed 1x1 kernel (split-f16;
        the gradient scale of loss_and_grads keeps dy in range), the accumulation as its residual epilogue.
        DDPM_TRAIN_1X1_DGRAD=gemm: the strided fp32 GEMM, in place (A/B)."""
        if self.conv1_dgrad_conv:
            return self._conv(dy, T.conv_weight_rot180t(w), residual=dx if accumulate else None)
        B, cout = dy.shape[:2]
        cin, hw = w.shape[1], dy[0, 0].numel()
        if dx is None:
            dx = torch.empty((B, cin) + tuple(dy.shape[2:]), dtype=torch.float32, device=dy.device)
        T.gemm(w, dy, dx, cin, hw, cout, a_m=1, a_k=cin, b_k=hw, b_n=1, c_m=hw, c_n=1, batch=B, a_batch=0, b_batch=cout * hw,
               c_batch=cin * hw, beta=1.0 if accumulate else 0.0)
        return dx

    # ---- ResnetBlock --------------------------------------------------------------------------------------------------------
    def _resnet_fwd(self, blk, x, es):
        a1, c1 = self._gn_fwd(x, blk.norm1, SILU)
        te = self._linear_fwd(es, blk.time_emb_proj)
        h1 = self._conv(a1, blk.conv1.conv.weight, blk.conv1.conv.bias, chan_add=te)
        a2, c2 = self._gn_fwd(h1, blk.norm2, SILU)
        ident = isinstance(blk.skip_connection, torch.nn.Identity)
        skip = x if ident else self._conv(x, blk.skip_connection.conv.weight, blk.skip_connection.conv.bias)
        out = self._conv(a2, blk.conv2.conv.weight, blk.conv2.conv.bias, residual=skip)
        return out, (blk, x, a1, c1, a2, c2, ident)

    def _resnet_bwd(self, ctx, dout):
        blk, x, a1, c1, a2, c2, ident = ctx
        es = self._es
        rows = self._bias_grad(dout, blk.conv2.conv.bias, keep_rows=not ident)
        da2 = self._conv3_bwd(a2, blk.conv2.conv.weight, dout, a_absmax=c2[4] if self.fused_stats else None)
        # the GroupNorm backward also leaves the maxima and the plane sums of dh1: conv1's weight gradient does not measure dh1, its
        # bias gradient and the time embedding's gradient do not read dh1 again
        if self.fused_stats:
            dh1, dh1_max, dte = self._gn_bwd(c2, da2, want=True)
            T.col_sum(dte, dh1.shape[0], dh1.shape[1], out=self.g(blk.conv1.conv.bias))
        else:  # (A/B: DDPM_TRAIN_FUSED_STATS=0 -- every consumer reads dh1 for itself)
            dh1, dh1_max = self._gn_bwd(c2, da2), None
            dte = self._bias_grad(dh1, blk.conv1.conv.bias, keep_rows=True).view(dh1.shape[0], dh1.shape[1])
        self._linear_bwd(es, blk.time_emb_proj, dte, dx=self._des, accumulate=True)
        da1 = self._conv3_bwd(a1, blk.conv1.conv.weight, dh1, a_absmax=c1[4] if self.fused_stats else None, dy_absmax=dh1_max)
        dx = self._gn_bwd(c1, da1)
        if ident:
            T.axpby(dx, dout, 1.0, 1.0, out=dx)
        else:
            sk = blk.skip_connection.conv
            self._conv1_wgrad(x, sk.weight, dout)
            T.col_sum(rows, dout.shape[0], dout.shape[1], out=self.g(sk.bias))
            dx = self._conv1_dgrad(sk.weight, dout, dx, accumulate=True)
        return dx

    # ---- AttentionBlock -------------------------------------------------------------------------------------------------------
    def _attn_fwd(self, blk, x, head_channels):
        B, Cc = x.shape[:2]
        n = x[0, 0].numel()
        heads = Cc // head_channels if head_channels else 1
        d = Cc // heads
        scale = 1.0 / math.sqrt(d)
        xn, c = self._gn_fwd(x, blk.norm, NONE)
        q = self._conv(xn, blk.to_q.weight, blk.to_q.bias)
        k = self._conv(xn, blk.to_k.weight, blk.to_k.bias)
        v = self._conv(xn, blk.to_v.weight, blk.to_v.bias)
        P = torch.empty((B * heads, n, n), dtype=torch.float32, device=x.device)
        zb = dict(batch=B * heads, batch_inner=heads, a_batch=d * n, a_batch_outer=Cc * n, b_batch=d * n, b_batch_outer=Cc * n)
        # S[z, i, j] = scale sum_c q[z, c, i] k[z, c, j]
        T.gemm(q, k, P, n, n, d, a_m=1, a_k=n, b_k=n, b_n=1, c_m=n, c_n=1, c_batch=n * n, c_batch_outer=heads * n * n, alpha=scale,
               **zb)
        T.softmax_rows_(P, B * heads * n, n)
        o = torch.empty_like(x)
        # o[z, c, i] = sum_j v[z, c, j] P[z, i, j]
        T.gemm(v, P, o, d, n, n, a_m=n, a_k=1, b_k=1, b_n=n, c_m=n, c_n=1, batch=B * heads, batch_inner=heads, a_batch=d * n,
               a_batch_outer=Cc * n, b_batch=n * n, b_batch_outer=heads * n * n, c_batch=d * n, c_batch_outer=Cc * n)
        if self.model.use_proj_attn:
            o_in = o
            o = self._conv(o_in, blk.proj_attn.weight, blk.proj_attn.bias)
        else:
            o_in = None
        out = T.axpby(o, x, 1.0, 1.0)
        return out, (blk, c, xn, q, k, v, P, o_in, heads, d, scale)

    def _attn_bwd(self, ctx, dout):
        blk, c, xn, q, k, v, P, o_in, heads, d, scale = ctx
        B, Cc = dout.shape[:2]
        n = dout[0, 0].numel()
        do = dout
        if o_in is not None:
            self._conv1_wgrad(o_in, blk.proj_attn.weight, dout)
            self._bias_grad(dout, blk.proj_attn.bias)
            do = self._conv1_dgrad(blk.proj_attn.weight, dout, None, accumulate=False)
        zq = dict(batch=B * heads, batch_inner=heads)
        cn = dict(c_m=n, c_n=1)
        dv, dq, dk = torch.empty_like(v), torch.empty_like(q), torch.empty_like(k)
        dP = torch.empty_like(P)
        # dv[z, c, j] = sum_i do[z, c, i] P[z, i, j]
        T.gemm(do, P, dv, d, n, n, a_m=n, a_k=1, b_k=n, b_n=1, a_batch=d * n, a_batch_outer=Cc * n, b_batch=n * n,
               b_batch_outer=heads * n * n, c_batch=d * n, c_batch_outer=Cc * n, **cn, **zq)
        # dP[z, i, j] = sum_c do[z, c, i] v[z, c, j]
        T.gemm(do, v, dP, n, n, d, a_m=1, a_k=n, b_k=n, b_n=1, a_batch=d * n, a_batch_outer=Cc * n, b_batch=d * n,
               b_batch_outer=Cc * n, c_batch=n * n, c_batch_outer=heads * n * n, **cn, **zq)
        T.softmax_backward_rows_(P, dP, B * heads * n, n)  # dP now holds dS
        # dq[z, c, i] = scale sum_j dS[z, i, j] k[z, c, j]
        T.gemm(k, dP, dq, d, n, n, a_m=n, a_k=1, b_k=1, b_n=n, a_batch=d * n, a_batch_outer=Cc * n, b_batch=n * n,
               b_batch_outer=heads * n * n, c_batch=d * n, c_batch_outer=Cc * n, alpha=scale, **cn, **zq)
        # dk[z, c, j] = scale sum_i dS[z, i, j] q[z, c, i]
        T.gemm(q, dP, dk, d, n, n, a_m=n, a_k=1, b_k=n, b_n=1, a_batch=d * n, a_batch_outer=Cc * n, b_batch=n * n,
               b_batch_outer=heads * n * n, c_batch=d * n, c_batch_outer=Cc * n, alpha=scale, **cn, **zq)
        dxn = torch.empty_like(xn)
        for i, (lin, dt) in enumerate(((blk.to_q, dq), (blk.to_k, dk), (blk.to_v, dv))):
            self._conv1_wgrad(xn, lin.weight, dt)
            self._bias_grad(dt, lin.bias)
            dxn = self._conv1_dgrad(lin.weight, dt, dxn, accumulate=i > 0)
        dx = self._gn_bwd(c, dxn)
        T.axpby(dx, dout, 1.0, 1.0, out=dx)
        return dx

    # ---- the network ----------------------------------------------------------------------------------------------------------
    def forward(self, x: torch.Tensor, timesteps: torch.Tensor) -> torch.Tensor:
        m = self.model
        tape = []
        ch0 = m.block_out_channels[0]
        if ch0 % 2:
            raise NotImplementedError("odd num_channels[0]")
        if getattr(self, "_freqs", None) is None or self._freqs.device != x.device:
            self._freqs = m._freqs().to(x.device)
        e0 = ops.timestep_embedding(timesteps, self._freqs, ch0)
        te0, te2 = m.time_embed[0], m.time_embed[2]
        e1 = self._linear_fwd(e0, te0)
        e2 = T.silu(e1)
        emb = self._linear_fwd(e2, te2)
        self._es = T.silu(emb)
        self._emb_ctx = (e0, e1, e2, emb)
        h = self._conv(x, m.conv_in.conv.weight, m.conv_in.conv.bias)
        tape.append(("conv_in", x))
        skips = [h]
        for blk, hc in zip(m.down_blocks, m.num_head_channels):
            for j, r in enumerate(blk.resnets):
                h, ctx = self._resnet_fwd(r, h, self._es)
                tape.append(("resnet", ctx))
                if hasattr(blk, "attentions"):
                    h, ctx = self._attn_fwd(blk.attentions[j], h, hc)
                    tape.append(("attn", ctx))
                skips.append(h)
                tape.append(("skip_push", None))
            if blk.downsampler is not None:
                op = blk.downsampler.op.conv
                tape.append(("down", (op, h)))
                h = self._conv(h, op.weight, op.bias, stride2=True)
                skips.append(h)
                tape.append(("skip_push", None))
        mid = m.middle_block
        h, ctx = self._resnet_fwd(mid.resnet_1, h, self._es)
        tape.append(("resnet", ctx))
        h, ctx = self._attn_fwd(mid.attention, h, m.num_head_channels[-1])
        tape.append(("attn", ctx))
        h, ctx = self._resnet_fwd(mid.resnet_2, h, self._es)
        tape.append(("resnet", ctx))
        for blk, hc in zip(m.up_blocks, reversed(m.num_head_channels)):
            for j, r in enumerate(blk.resnets):
                s = skips.pop()
                c1, c2 = h.shape[1], s.shape[1]
                cat = torch.empty((h.shape[0], c1 + c2) + tuple(h.shape[2:]), dtype=torch.float32, device=h.device)
                T.chan_copy(h, cat, c1, 0, 0)
                T.chan_copy(s, cat, c2, 0, c1)
                tape.append(("cat", (c1, c2)))
                h, ctx = self._resnet_fwd(r, cat, self._es)
                tape.append(("resnet", ctx))
                if hasattr(blk, "attentions"):
                    h, ctx = self._attn_fwd(blk.attentions[j], h, hc)
                    tape.append(("attn", ctx))
            if blk.upsampler is not None:
                op = blk.upsampler.conv.conv
                u = T.upsample2(h)
                tape.append(("up", (op, u)))
                h = self._conv(u, op.weight, op.bias)
        a, c = self._gn_fwd(h, m.out[0], SILU)
        oc = m.out[2].conv
        tape.append(("out", (oc, a, c)))
        pred = self._conv(a, oc.weight, oc.bias)
        self._tape = tape
        return pred

    def backward(self, dpred: torch.Tensor) -> None:
        """Fills every parameter's gradient view (overwrites: call after zeroing is NOT needed, each gradient has one writer)."""
        m = self.model
        self._des = T.fill_(torch.empty_like(self._es), 0.0)
        dskips = []  # gradients of the skip tensors, in the order the skips were created
        dh = None
        for kind, ctx in reversed(self._tape):
            if kind == "out":
                oc, a, c = ctx
                self._bias_grad(dpred, oc.bias)
                self._wgrad3(a, dpred, oc.weight)
                da = self._conv(dpred, T.conv_weight_rot180t(oc.weight), form=self.dgrad_form)
                dh = self._gn_bwd(c, da)
            elif kind == "up":
                op, u = ctx
                self._bias_grad(dh, op.bias)
                du = self._conv3_bwd(u, op.weight, dh)
                dh = T.sumpool2(du)
            elif kind == "resnet":
                dh = self._resnet_bwd(ctx, dh)
            elif kind == "attn":
                dh = self._attn_bwd(ctx, dh)
            elif kind == "cat":
                c1, c2 = ctx
                dcat = dh
                shp = tuple(dcat.shape)
                dh = torch.empty((shp[0], c1) + shp[2:], dtype=torch.float32, device=dcat.device)
                ds = torch.empty((shp[0], c2) + shp[2:], dtype=torch.float32, device=dcat.device)
                T.chan_copy(dcat, dh, c1, 0, 0)
                T.chan_copy(dcat, ds, c2, c1, 0)
                dskips.append(ds)
            elif kind == "skip_push":
                T.axpby(dh, dskips.pop(), 1.0, 1.0, out=dh)
            elif kind == "down":
                op, hin = ctx
                self._bias_grad(dh, op.bias)
                self._wgrad3(hin, dh, op.weight, stride=2)
                dh = self._conv(T.zero_stuff2(dh), T.conv_weight_rot180t(op.weight), form=self.dgrad_form)
            elif kind == "conv_in":
                T.axpby(dh, dskips.pop(), 1.0, 1.0, out=dh)  # skips[0] = conv_in's output
                ci = m.conv_in.conv
                self._bias_grad(dh, ci.bias)
                self._wgrad3(ctx, dh, ci.weight)
        assert not dskips
        # time embedding: es = silu(emb), emb = Linear2(silu(Linear0(e0)))
        e0, e1, e2, emb = self._emb_ctx
        demb = T.silu_backward(emb, self._des)
        de2 = self._linear_bwd(e2, m.time_embed[2], demb)
        de1 = T.silu_backward(e1, de2)
        self._linear_bwd(e0, m.time_embed[0], de1, need_dx=False)
        self._tape = None

    SCALE_GROWTH_INTERVAL = 2000  # clean steps after which the adaptive scale doubles (gradients shrink as training converges)

    def loss_and_grads(self, noisy, timesteps, target):
        """F.mse_loss(model(noisy, timesteps), target) and every parameter gradient; returns the loss as a 1-element device tensor.

        The backward runs on dpred * loss_scale (a power of two) and the flat gradient buffer is unscaled afterwards; the unscale
        pass flags non-finite values in the device status word, which is read here (one 4-byte read-back per step: the trainer
        reads the loss every step anyway).  Overflow (the scale pushed an input-gradient operand past f16): the scale drops by
        2^4 and the backward runs again on the same tape -- twice at most, then once with the input gradients on the fp32 pipe; a
        gradient that is still non-finite is a genuine one and is left in place, as the reference would."""
        from . import _lib

        pred = self.forward(noisy, timesteps)
        loss, dpred0 = T.mse_loss_grad(pred, target)
        if self.scale_adaptive and self.loss_scale is None:
            # dpred = 2 (pred - target) / n: n / 2 brings it to the error's own magnitude; the gradients further down are smaller
            # still (2^11 more was measured safe at batch 4 .. 256 -- and the overflow check below is what makes a guess harmless)
            self.loss_scale = 2.0 ** (math.floor(math.log2(max(pred.numel() / 2, 1))) + 8)
        tape, form = self._tape, self.dgrad_form
        for attempt in range(4):  # (a bit left in the word by something else costs one spurious repeat)
            scale = self.loss_scale if self.dgrad_form == "wino44h" or not self.scale_adaptive else 1.0
            dpred = dpred0 if scale == 1.0 else T.axpby(dpred0, None, scale, 0.0)
            self._tape, self._scaled = tape, scale >= 1024.0
            self.backward(dpred)
            T.scale_check_(self.gflat, 1.0 / scale)
            if not (_lib.status_read(clear=True) & _lib.STATUS_NONFINITE_GRAD) or not self.scale_adaptive or attempt == 3:
                break
            self.overflow_retries += 1
            self._clean_steps = 0
            if attempt < 2:
                self.loss_scale = max(self.loss_scale / 16.0, 1.0)
            else:
                self.dgrad_form = "wino"  # this step only
                self.fp32_dgrad_steps += 1
        self.dgrad_form = form
        if self.scale_adaptive:
            self._clean_steps += 1
            if self._clean_steps >= self.SCALE_GROWTH_INTERVAL:
                self.loss_scale, self._clean_steps = self.loss_scale * 2.0, 0
        return loss

    def adam_step(self, grad_scale: float = 1.0):
        self.step_count += 1
        T.adam_step_(self.flat, self.gflat, self.m, self.v, self.lr, self.betas[0], self.betas[1], self.eps, self.step_count,
                     grad_scale)
        self.model._synced_key = None  # updated in place by a kernel torch knows nothing about: the inference engine re-packs

    # ---- torch.optim.Adam-compatible state (the checkpoint dict of base.py:166-187 carries optimizer.state_dict()) -----------
    def state_dict(self):
        state = {}
        for i, p in enumerate(self.params):
            off, n = self.offsets[id(p)]
            if self.step_count:
                state[i] = {"step": torch.tensor(float(self.step_count)), "exp_avg": self.m[off: off + n].view(p.shape).clone(),
                            "exp_avg_sq": self.v[off: off + n].view(p.shape).clone()}
        group = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": 0, "amsgrad": False, "maximize": False,
                 "foreach": None, "capturable": False, "differentiable": False, "fused": None, "decoupled_weight_decay": False,
                 "params": list(range(len(self.params)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        if not sd or not sd.get("state"):
            return
        steps = set()
        for i, p in enumerate(self.params):
            st = sd["state"].get(i)
            if st is None:
                continue
            off, n = self.offsets[id(p)]
            self.m[off: off + n].copy_(st["exp_avg"].reshape(-1).to(self.m.device))
            self.v[off: off + n].copy_(st["exp_avg_sq"].reshape(-1).to(self.v.device))
            steps.add(int(float(st["step"])))
        if steps:
            self.step_count = max(steps)
