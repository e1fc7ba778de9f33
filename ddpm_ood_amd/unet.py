"""``DiffusionModelUNet`` with the MONAI-Generative call surface, executed by the HIP engine.

Drop-in for ``generative.networks.nets.DiffusionModelUNet`` as the reference uses it:
  ctor kwargs     /root/reference/src/trainers/base.py:66-86
  .to / .eval / .parameters / .load_state_dict(ckpt["model_state_dict"])   base.py:75,89,145
  __call__(x, timesteps=int64[B] on device) -> eps     src/trainers/reconstruct.py:151-153
The nn.Module tree below only HOLDS the parameters under the MONAI-Generative key names
(SURVEY.md A.5); there is no PyTorch forward.  ``forward`` hands raw device pointers to
``ddpm_unet_forward`` (include/ddpm_ood_hip.h), which runs the whole network as fused HIP
kernels on torch's current stream.
"""

from __future__ import annotations

import ctypes as C
import math
from typing import Sequence

import torch
import torch.nn as nn

from . import _lib
from ._lib import UNetConfig, check, ptr, require_device_f32, stream_ptr


def _conv_nd(spatial_dims):
    return {2: nn.Conv2d, 3: nn.Conv3d}[spatial_dims]


class _Holder(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover - parameter containers only
        raise RuntimeError("parameter holder: the forward pass lives in libddpm_ood_hip.so")


class _Convolution(_Holder):
    def __init__(self, sd, cin, cout, k=3, stride=1, pad=1):
        super().__init__()
        self.conv = _conv_nd(sd)(cin, cout, k, stride, pad)


def _zero(m):
    for p in m.parameters():
        p.detach().zero_()
    return m


class _ResnetBlock(_Holder):
    def __init__(self, sd, cin, temb, cout, g, eps):
        super().__init__()
        self.norm1 = nn.GroupNorm(g, cin, eps=eps, affine=True)
        self.conv1 = _Convolution(sd, cin, cout)
        self.time_emb_proj = nn.Linear(temb, cout)
        self.norm2 = nn.GroupNorm(g, cout, eps=eps, affine=True)
        self.conv2 = _zero(_Convolution(sd, cout, cout))
        self.skip_connection = nn.Identity() if cin == cout else _Convolution(sd, cin, cout, 1, 1, 0)


class _AttentionBlock(_Holder):
    def __init__(self, c, g, eps):
        super().__init__()
        self.norm = nn.GroupNorm(g, c, eps=eps, affine=True)
        self.to_q = nn.Linear(c, c)
        self.to_k = nn.Linear(c, c)
        self.to_v = nn.Linear(c, c)
        self.proj_attn = nn.Linear(c, c)


class _Downsample(_Holder):
    def __init__(self, sd, c):
        super().__init__()
        self.op = _Convolution(sd, c, c, 3, 2, 1)


class _Upsample(_Holder):
    def __init__(self, sd, c):
        super().__init__()
        self.conv = _Convolution(sd, c, c)


class _Block(_Holder):
    pass


class DiffusionModelUNet(nn.Module):
    def __init__(self, spatial_dims: int, in_channels: int, out_channels: int,
                 num_res_blocks: Sequence[int] | int = (2, 2, 2, 2),
                 num_channels: Sequence[int] = (32, 64, 64, 64),
                 attention_levels: Sequence[bool] = (False, False, True, True),
                 norm_num_groups: int = 32, norm_eps: float = 1e-6, resblock_updown: bool = False,
                 num_head_channels: int | Sequence[int] = 8, with_conditioning: bool = False,
                 transformer_num_layers: int = 1, cross_attention_dim=None, num_class_embeds=None,
                 upcast_attention: bool = False, use_flash_attention: bool = False,
                 use_proj_attn: bool = False):
        super().__init__()
        if with_conditioning or cross_attention_dim is not None or num_class_embeds is not None:
            raise NotImplementedError("conditioning is off the reconstruction path (base.py:74,85)")
        if resblock_updown:
            raise NotImplementedError("resblock_updown=True is not used by the reference configs")
        if any((c % norm_num_groups) != 0 for c in num_channels):
            raise ValueError("DiffusionModelUNet expects all num_channels being multiple of norm_num_groups")
        if len(num_channels) != len(attention_levels):
            raise ValueError("DiffusionModelUNet expects num_channels being same size of attention_levels")
        if isinstance(num_head_channels, int):
            num_head_channels = (num_head_channels,) * len(attention_levels)
        if len(num_head_channels) != len(attention_levels):
            raise ValueError("num_head_channels should have the same length as attention_levels.")
        if isinstance(num_res_blocks, int):
            num_res_blocks = (num_res_blocks,) * len(num_channels)
        if len(num_res_blocks) != len(num_channels):
            raise ValueError("`num_res_blocks` should be a single integer or a tuple of integers with the same "
                             "length as `num_channels`.")
        self.spatial_dims = spatial_dims
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.block_out_channels = tuple(num_channels)
        self.num_res_blocks = tuple(num_res_blocks)
        self.attention_levels = tuple(bool(a) for a in attention_levels)
        self.num_head_channels = tuple(num_head_channels)
        self.norm_num_groups = norm_num_groups
        self.norm_eps = norm_eps
        self.use_proj_attn = use_proj_attn
        self.with_conditioning = False

        sd, g, eps = spatial_dims, norm_num_groups, norm_eps
        ch0 = num_channels[0]
        ted = 4 * ch0
        self.conv_in = _Convolution(sd, in_channels, ch0)
        self.time_embed = nn.Sequential(nn.Linear(ch0, ted), nn.SiLU(), nn.Linear(ted, ted))
        self.down_blocks = nn.ModuleList()
        out_c = ch0
        L = len(num_channels)
        for i in range(L):
            in_c, out_c = out_c, num_channels[i]
            blk = _Block()
            blk.resnets = nn.ModuleList(
                [_ResnetBlock(sd, in_c if j == 0 else out_c, ted, out_c, g, eps) for j in range(num_res_blocks[i])])
            if attention_levels[i]:
                blk.attentions = nn.ModuleList([_AttentionBlock(out_c, g, eps) for _ in range(num_res_blocks[i])])
            blk.downsampler = _Downsample(sd, out_c) if i != L - 1 else None
            self.down_blocks.append(blk)
        mid = _Block()
        mid.resnet_1 = _ResnetBlock(sd, num_channels[-1], ted, num_channels[-1], g, eps)
        mid.attention = _AttentionBlock(num_channels[-1], g, eps)
        mid.resnet_2 = _ResnetBlock(sd, num_channels[-1], ted, num_channels[-1], g, eps)
        self.middle_block = mid
        self.up_blocks = nn.ModuleList()
        rev_c, rev_r, rev_a = list(reversed(num_channels)), list(reversed(num_res_blocks)), \
            list(reversed(attention_levels))
        out_c = rev_c[0]
        for i in range(L):
            prev_c, out_c = out_c, rev_c[i]
            in_c = rev_c[min(i + 1, L - 1)]
            n = rev_r[i] + 1
            blk = _Block()
            blk.resnets = nn.ModuleList([
                _ResnetBlock(sd, (prev_c if j == 0 else out_c) + (in_c if j == n - 1 else out_c), ted, out_c, g, eps)
                for j in range(n)])
            if rev_a[i]:
                blk.attentions = nn.ModuleList([_AttentionBlock(out_c, g, eps) for _ in range(n)])
            blk.upsampler = _Upsample(sd, out_c) if i != L - 1 else None
            self.up_blocks.append(blk)
        self.out = nn.Sequential(nn.GroupNorm(g, ch0, eps=eps, affine=True), nn.SiLU(),
                                 _zero(_Convolution(sd, ch0, out_channels)))

        self._engine = None      # ddpm_unet* (ctypes void pointer)
        self._blob = None        # packed parameters on the device
        self._synced_key = None  # (device, parameter versions) the blob was built from
        self._workspace = None
        self._plist = None       # cached parameter list for _param_key
        self._static = {}        # graph replay: (device, input shape) -> static (x, timesteps, out, workspace)
        for p in self.parameters():
            p.requires_grad_(False)

    # ---- engine management -------------------------------------------------------------------
    def _config(self) -> UNetConfig:
        cfg = UNetConfig()
        cfg.spatial_dims = self.spatial_dims
        cfg.in_channels, cfg.out_channels = self.in_channels, self.out_channels
        cfg.num_levels = len(self.block_out_channels)
        for i, (c, a, r, h) in enumerate(zip(self.block_out_channels, self.attention_levels, self.num_res_blocks,
                                             self.num_head_channels)):
            cfg.num_channels[i], cfg.attention_levels[i] = c, int(a)
            cfg.num_res_blocks[i], cfg.num_head_channels[i] = r, h
        cfg.norm_num_groups, cfg.norm_eps = self.norm_num_groups, self.norm_eps
        cfg.use_proj_attn = int(self.use_proj_attn)
        return cfg

    def _freqs(self) -> torch.Tensor:
        # generative.networks.nets.diffusion_model_unet.get_timestep_embedding,
        # evaluated on the host with the same torch ops so the table is bit-identical.
        half = self.block_out_channels[0] // 2
        exponent = -math.log(10000) * torch.arange(0, half, dtype=torch.float32)
        return torch.exp(exponent / half)

    def _param_key(self, device):
        """Cheap change detector for the packed blob (runs on EVERY forward: ~20 us; the full
        (data_ptr, version) tuple of 182 tensors cost 0.8 ms, more than a small-batch forward): in-place updates
        (optimizer steps, load_state_dict) bump a tensor's version counter, .to() / re-assignment replace storage."""
        self._key_calls = getattr(self, "_key_calls", 0) + 1
        if self._plist is None or self._key_calls % 64 == 0:
            # parameters replaced without _apply (setattr of a new nn.Parameter, `p.data = new` on an interior tensor)
            # leave a stale list behind: rebuild it every 64 calls as a backstop for the hooks below
            self._plist = list(self.parameters())
        ps = self._plist
        ptrs = 0
        for p in ps:
            ptrs ^= p.data_ptr()
        return (str(device), sum(p._version for p in ps), ptrs, len(ps))

    def load_state_dict(self, state_dict, strict=True, assign=False):
        out = super().load_state_dict(state_dict, strict=strict, assign=assign)
        self._plist = None  # assign=True replaces the parameter objects
        self._synced_key = None
        return out

    def __setattr__(self, name, value):
        if isinstance(value, (torch.nn.Parameter, torch.nn.Module)) and "_plist" in self.__dict__:
            self.__dict__["_plist"] = None
            self.__dict__["_synced_key"] = None
        super().__setattr__(name, value)

    def _apply(self, fn, *a, **k):  # .to() / .cuda() / .float(): parameters are replaced
        self._plist = None
        self._synced_key = None
        return super()._apply(fn, *a, **k)

    def _sync(self, device) -> None:
        lib = _lib.load()
        if self._engine is None:
            cfg = self._config()
            h = lib.ddpm_unet_create(C.byref(cfg))
            if not h:
                raise ValueError("DiffusionModelUNet: " + lib.ddpm_last_error().decode())
            self._engine = C.c_void_p(h)
        key = self._param_key(device)
        if key == self._synced_key:
            return
        n = lib.ddpm_unet_param_blob_floats(self._engine)
        self._blob = torch.zeros(n, dtype=torch.float32, device=device)
        check(lib.ddpm_unet_bind_param_blob(self._engine, ptr(self._blob)), "bind_param_blob")
        sd = dict(self.state_dict())
        sd["freqs"] = self._freqs()
        expected = {lib.ddpm_unet_param_name(self._engine, i).decode()
                    for i in range(lib.ddpm_unet_num_params(self._engine))}
        keep = []
        for name in sorted(expected):
            if name not in sd:
                raise KeyError(f"missing key '{name}' in state_dict")
            t = sd[name].detach().to(device=device, dtype=torch.float32).contiguous()
            keep.append(t)
            check(lib.ddpm_unet_set_param(self._engine, name.encode(), ptr(t), t.numel(), stream_ptr()),
                  f"set_param({name})")
        torch.cuda.current_stream().synchronize()  # `keep` may be freed after this point
        self._synced_key = key

    def __del__(self):
        try:
            if self._engine is not None:
                _lib.load().ddpm_unet_destroy(self._engine)
        except Exception:
            pass

    # ---- the call surface -------------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, x: torch.Tensor, timesteps: torch.Tensor, context=None, class_labels=None):
        if context is not None or class_labels is not None:
            raise NotImplementedError("conditioning is off the reconstruction path")
        x = require_device_f32(x, "x")
        if x.ndim != 2 + self.spatial_dims or x.shape[1] != self.in_channels:
            raise ValueError(f"expected [B, {self.in_channels}, *spatial{self.spatial_dims}], got {tuple(x.shape)}")
        if timesteps.dtype != torch.int64:
            timesteps = timesteps.long()
        if not timesteps.is_cuda:
            timesteps = timesteps.to(x.device)
        timesteps = timesteps.contiguous()
        if timesteps.shape != (x.shape[0],):
            raise ValueError("timesteps must have shape [B]")
        self._sync(x.device)
        lib = _lib.load()
        B = x.shape[0]
        D, H, W = ((1,) + tuple(x.shape[2:])) if self.spatial_dims == 2 else tuple(x.shape[2:])
        need = lib.ddpm_unet_workspace_bytes3d(self._engine, B, D, H, W)
        if need == 0:
            raise ValueError("DiffusionModelUNet: " + lib.ddpm_last_error().decode())
        if self._workspace is None or self._workspace.numel() < need or self._workspace.device != x.device:
            self._workspace = torch.empty(need, dtype=torch.uint8, device=x.device)
        if self._use_graph(B):
            return self._forward_graphed(x, timesteps, B, D, H, W)
        out = torch.empty((B, self.out_channels) + tuple(x.shape[2:]), dtype=torch.float32, device=x.device)
        check(lib.ddpm_unet_forward3d(self._engine, ptr(x), ptr(timesteps), ptr(out), B, D, H, W,
                                      ptr(self._workspace), self._workspace.numel(), stream_ptr()), "unet_forward")
        return out

    # ---- hipGraph replay for the launch-bound small-batch regime -------------------
    # Measured on MI355X (profiles/r02_small_batch.log): at B = 4 and B = 16 a `small` 32x32 forward is 3.3 ms of
    # KERNEL time (the persistent convolutions are latency-bound: few items, each a serial stream of 16-64 chunks), so
    # the host is not the limiter and replay gains nothing (3.41 vs 3.30 ms).  Replay therefore is opt-in.
    def _use_graph(self, B: int) -> bool:
        import os

        return os.environ.get("DDPM_UNET_GRAPH", "0") in ("1", "on")

    def _forward_graphed(self, x, timesteps, B, D, H, W):
        """Static input / timestep / output tensors per input shape keep the captured kernel arguments valid; the
        result is cloned because the PLMS scheduler keeps up to four earlier outputs alive."""
        lib = _lib.load()
        key = (str(x.device), tuple(x.shape))
        st = self._static.get(key)
        if st is None or st[3] is not self._workspace:
            st = (torch.empty_like(x), torch.empty_like(timesteps),
                  torch.empty((B, self.out_channels) + tuple(x.shape[2:]), dtype=torch.float32, device=x.device),
                  self._workspace)
            self._static[key] = st
        xs, ts, outs, _ = st
        xs.copy_(x)
        ts.copy_(timesteps)
        check(lib.ddpm_unet_forward_graphed(self._engine, ptr(xs), ptr(ts), ptr(outs), B, D, H, W,
                                            ptr(self._workspace), self._workspace.numel(), stream_ptr()),
              "unet_forward_graphed")
        return outs.clone()
