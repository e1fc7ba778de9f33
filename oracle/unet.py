"""Oracle restatement of MONAI-Generative 0.2.x ``DiffusionModelUNet`` (unconditional).

TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED: the real module lives in
the un-vendored ``generative`` package; this follows SURVEY.md Appendix A.1-A.3/A.5 and the
reference's constructor call /root/reference/src/trainers/base.py:65-86 and forward call
/root/reference/src/trainers/reconstruct.py:150-153.  Plain torch.nn.functional on CPU
fp32 -- the per-op ground truth named in SURVEY 8(c).

state_dict key names follow Appendix A.5 so the same checkpoint file loads into this
oracle and into the HIP-backed ``ddpm_ood_amd.DiffusionModelUNet``.
"""

from __future__ import annotations

import math
from typing import Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F


def _conv_nd(spatial_dims: int):
    return {1: nn.Conv1d, 2: nn.Conv2d, 3: nn.Conv3d}[spatial_dims]


class Convolution(nn.Module):
    """monai.networks.blocks.Convolution(conv_only=True): a ConvNd stored as ``.conv``."""

    def __init__(self, spatial_dims, in_channels, out_channels, kernel_size=3, strides=1, padding=1):
        super().__init__()
        self.conv = _conv_nd(spatial_dims)(in_channels, out_channels, kernel_size, strides, padding)

    def forward(self, x):
        return self.conv(x)


def zero_module(m: nn.Module) -> nn.Module:
    for p in m.parameters():
        p.detach().zero_()
    return m


def get_timestep_embedding(timesteps: torch.Tensor, embedding_dim: int, max_period: int = 10000):
    half = embedding_dim // 2
    exponent = -math.log(max_period) * torch.arange(0, half, dtype=torch.float32, device=timesteps.device)
    freqs = torch.exp(exponent / half)
    args = timesteps[:, None].float() * freqs[None, :]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if embedding_dim % 2 == 1:
        emb = F.pad(emb, (0, 1, 0, 0))
    return emb


class ResnetBlock(nn.Module):
    def __init__(self, spatial_dims, in_channels, temb_channels, out_channels=None,
                 norm_num_groups=32, norm_eps=1e-6):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels or in_channels
        self.norm1 = nn.GroupNorm(norm_num_groups, in_channels, eps=norm_eps, affine=True)
        self.conv1 = Convolution(spatial_dims, in_channels, self.out_channels)
        self.time_emb_proj = nn.Linear(temb_channels, self.out_channels)
        self.norm2 = nn.GroupNorm(norm_num_groups, self.out_channels, eps=norm_eps, affine=True)
        self.conv2 = zero_module(Convolution(spatial_dims, self.out_channels, self.out_channels))
        if self.out_channels == in_channels:
            self.skip_connection = nn.Identity()
        else:
            self.skip_connection = Convolution(spatial_dims, in_channels, self.out_channels,
                                               kernel_size=1, padding=0)

    def forward(self, x, emb):
        h = self.conv1(F.silu(self.norm1(x)))
        temb = self.time_emb_proj(F.silu(emb))
        h = h + temb.reshape(temb.shape + (1,) * (x.ndim - 2))
        h = self.conv2(F.silu(self.norm2(h)))
        return self.skip_connection(x) + h


class AttentionBlock(nn.Module):
    """A.3.  ``proj_attn`` exists in the state_dict; applied only when ``use_proj_attn``."""

    def __init__(self, spatial_dims, num_channels, num_head_channels=None, norm_num_groups=32,
                 norm_eps=1e-6, use_proj_attn=False):
        super().__init__()
        self.num_channels = num_channels
        self.num_heads = num_channels // num_head_channels if num_head_channels is not None else 1
        self.scale = 1 / math.sqrt(num_channels / self.num_heads)
        self.norm = nn.GroupNorm(norm_num_groups, num_channels, eps=norm_eps, affine=True)
        self.to_q = nn.Linear(num_channels, num_channels)
        self.to_k = nn.Linear(num_channels, num_channels)
        self.to_v = nn.Linear(num_channels, num_channels)
        self.proj_attn = nn.Linear(num_channels, num_channels)
        self.use_proj_attn = use_proj_attn

    def _heads_to_batch(self, x):
        b, n, c = x.shape
        h = self.num_heads
        return x.reshape(b, n, h, c // h).permute(0, 2, 1, 3).reshape(b * h, n, c // h)

    def _batch_to_heads(self, x):
        bh, n, d = x.shape
        h = self.num_heads
        return x.reshape(bh // h, h, n, d).permute(0, 2, 1, 3).reshape(bh // h, n, d * h)

    def forward(self, x):
        residual = x
        shape = x.shape
        b, c = shape[:2]
        x = self.norm(x)
        x = x.view(b, c, -1).transpose(1, 2)
        q = self._heads_to_batch(self.to_q(x))
        k = self._heads_to_batch(self.to_k(x))
        v = self._heads_to_batch(self.to_v(x))
        scores = torch.baddbmm(
            torch.empty(q.shape[0], q.shape[1], k.shape[1], dtype=q.dtype, device=q.device),
            q, k.transpose(-1, -2), beta=0, alpha=self.scale)
        probs = scores.softmax(dim=-1)
        x = self._batch_to_heads(torch.bmm(probs, v))
        if self.use_proj_attn:
            x = self.proj_attn(x)
        x = x.transpose(-1, -2).reshape(shape)
        return x + residual


class Downsample(nn.Module):
    def __init__(self, spatial_dims, num_channels):
        super().__init__()
        self.op = Convolution(spatial_dims, num_channels, num_channels, strides=2, kernel_size=3, padding=1)

    def forward(self, x, emb=None):
        return self.op(x)


class Upsample(nn.Module):
    def __init__(self, spatial_dims, num_channels):
        super().__init__()
        self.conv = Convolution(spatial_dims, num_channels, num_channels)

    def forward(self, x, emb=None):
        x = F.interpolate(x, scale_factor=2.0, mode="nearest")
        return self.conv(x)


class DownBlock(nn.Module):
    def __init__(self, spatial_dims, in_channels, out_channels, temb_channels, num_res_blocks,
                 add_downsample, with_attn, num_head_channels, g, eps, use_proj_attn):
        super().__init__()
        res, att = [], []
        for i in range(num_res_blocks):
            res.append(ResnetBlock(spatial_dims, in_channels if i == 0 else out_channels,
                                   temb_channels, out_channels, g, eps))
            if with_attn:
                att.append(AttentionBlock(spatial_dims, out_channels, num_head_channels, g, eps, use_proj_attn))
        self.resnets = nn.ModuleList(res)
        if with_attn:
            self.attentions = nn.ModuleList(att)
        self.with_attn = with_attn
        self.downsampler = Downsample(spatial_dims, out_channels) if add_downsample else None

    def forward(self, h, emb):
        outs = []
        for i, r in enumerate(self.resnets):
            h = r(h, emb)
            if self.with_attn:
                h = self.attentions[i](h)
            outs.append(h)
        if self.downsampler is not None:
            h = self.downsampler(h, emb)
            outs.append(h)
        return h, outs


class MidBlock(nn.Module):
    def __init__(self, spatial_dims, in_channels, temb_channels, num_head_channels, g, eps, use_proj_attn):
        super().__init__()
        self.resnet_1 = ResnetBlock(spatial_dims, in_channels, temb_channels, in_channels, g, eps)
        self.attention = AttentionBlock(spatial_dims, in_channels, num_head_channels, g, eps, use_proj_attn)
        self.resnet_2 = ResnetBlock(spatial_dims, in_channels, temb_channels, in_channels, g, eps)

    def forward(self, h, emb):
        return self.resnet_2(self.attention(self.resnet_1(h, emb)), emb)


class UpBlock(nn.Module):
    def __init__(self, spatial_dims, in_channels, prev_output_channel, out_channels, temb_channels,
                 num_res_blocks, add_upsample, with_attn, num_head_channels, g, eps, use_proj_attn):
        super().__init__()
        res, att = [], []
        for i in range(num_res_blocks):
            res_skip = in_channels if i == num_res_blocks - 1 else out_channels
            res_in = prev_output_channel if i == 0 else out_channels
            res.append(ResnetBlock(spatial_dims, res_in + res_skip, temb_channels, out_channels, g, eps))
            if with_attn:
                att.append(AttentionBlock(spatial_dims, out_channels, num_head_channels, g, eps, use_proj_attn))
        self.resnets = nn.ModuleList(res)
        if with_attn:
            self.attentions = nn.ModuleList(att)
        self.with_attn = with_attn
        self.upsampler = Upsample(spatial_dims, out_channels) if add_upsample else None

    def forward(self, h, skips, emb):
        for i, r in enumerate(self.resnets):
            s = skips[-1]
            skips = skips[:-1]
            h = r(torch.cat([h, s], dim=1), emb)
            if self.with_attn:
                h = self.attentions[i](h)
        if self.upsampler is not None:
            h = self.upsampler(h, emb)
        return h


class DiffusionModelUNet(nn.Module):
    def __init__(self, spatial_dims: int, in_channels: int, out_channels: int,
                 num_res_blocks: Sequence[int] | int = (2, 2, 2, 2),
                 num_channels: Sequence[int] = (32, 64, 64, 64),
                 attention_levels: Sequence[bool] = (False, False, True, True),
                 norm_num_groups: int = 32, norm_eps: float = 1e-6,
                 num_head_channels: int | Sequence[int] = 8, with_conditioning: bool = False,
                 use_proj_attn: bool = False):
        super().__init__()
        if with_conditioning:
            raise NotImplementedError("the hot path builds with_conditioning=False (base.py:74,85)")
        if any(c % norm_num_groups for c in num_channels):
            raise ValueError("DiffusionModelUNet expects all num_channels being multiple of norm_num_groups")
        if len(num_channels) != len(attention_levels):
            raise ValueError("DiffusionModelUNet expects num_channels being same size of attention_levels")
        if isinstance(num_head_channels, int):
            num_head_channels = (num_head_channels,) * len(attention_levels)
        if isinstance(num_res_blocks, int):
            num_res_blocks = (num_res_blocks,) * len(num_channels)
        self.spatial_dims = spatial_dims
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.block_out_channels = tuple(num_channels)
        g, eps = norm_num_groups, norm_eps
        ch0 = num_channels[0]
        ted = ch0 * 4
        self.conv_in = Convolution(spatial_dims, in_channels, ch0)
        self.time_embed = nn.Sequential(nn.Linear(ch0, ted), nn.SiLU(), nn.Linear(ted, ted))

        self.down_blocks = nn.ModuleList()
        out_c = ch0
        for i, c in enumerate(num_channels):
            in_c, out_c = out_c, c
            last = i == len(num_channels) - 1
            self.down_blocks.append(DownBlock(spatial_dims, in_c, out_c, ted, num_res_blocks[i], not last,
                                              attention_levels[i], num_head_channels[i], g, eps, use_proj_attn))
        self.middle_block = MidBlock(spatial_dims, num_channels[-1], ted, num_head_channels[-1], g, eps,
                                     use_proj_attn)
        self.up_blocks = nn.ModuleList()
        rev_c = list(reversed(num_channels))
        rev_r = list(reversed(num_res_blocks))
        rev_a = list(reversed(attention_levels))
        rev_h = list(reversed(num_head_channels))
        out_c = rev_c[0]
        for i in range(len(rev_c)):
            prev_c, out_c = out_c, rev_c[i]
            in_c = rev_c[min(i + 1, len(num_channels) - 1)]
            last = i == len(num_channels) - 1
            self.up_blocks.append(UpBlock(spatial_dims, in_c, prev_c, out_c, ted, rev_r[i] + 1, not last,
                                          rev_a[i], rev_h[i], g, eps, use_proj_attn))
        self.out = nn.Sequential(
            nn.GroupNorm(g, ch0, eps=eps, affine=True), nn.SiLU(),
            zero_module(Convolution(spatial_dims, ch0, out_channels)))

    def forward(self, x: torch.Tensor, timesteps: torch.Tensor, context=None, class_labels=None):
        t_emb = get_timestep_embedding(timesteps, self.block_out_channels[0]).to(dtype=x.dtype)
        emb = self.time_embed(t_emb)
        h = self.conv_in(x)
        skips = [h]
        for blk in self.down_blocks:
            h, outs = blk(h, emb)
            skips.extend(outs)
        h = self.middle_block(h, emb)
        for blk in self.up_blocks:
            n = len(blk.resnets)
            take, skips = skips[-n:], skips[:-n]
            h = blk(h, take, emb)
        return self.out(h)
