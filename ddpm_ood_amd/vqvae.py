"""Stage-1 model surface used by the reconstruction path.

``PassthroughVQVAE`` is the identity stand-in the reference uses for pixel-space DDPMs
(/root/reference/src/networks/passthrough_vqvae.py:4-26, selected at
/root/reference/src/trainers/base.py:62-64).

``VQVAE`` mirrors ``generative.networks.nets.VQVAE`` for the latent-diffusion
configuration: ctor from ``vqvae_config.json`` + ``load_state_dict`` (base.py:44-61),
``encode_stage_2_inputs`` / ``decode_stage_2_outputs`` (reconstruct.py:124,166).  SURVEY.md 8(f) row f-1: on a
ROCm device every layer of the reference configuration (README.md:153-158: 4 stride-2 levels, 256 channels,
embedding 128 x 2 048) runs on the library's HIP kernels -- the 3x3x3 convolutions, the k4 s2 down-convolutions and
the k4 s2 transposed convolutions on the fp32-MFMA kernel (one launch each, depth taps / output parities inside) -- the
largest one (32^3 -> 64^3) since round 6 as eight parity convolutions on the split-f16 F(4x4) kernel + one interleave
pass (``ops.conv_transpose_parity``) --, the single-channel first / last layers on ``conv3d_edge.hip``, the nearest-code
search on ``vq.hip``.  Shapes without an MFMA tiling (channel counts not multiples of 128 / 8) take the library's generic
HIP convolution and say so once (no PyTorch-ROCm route).
"""

import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib, ops


class PassthroughVQVAE(torch.nn.Module):
    """This fake VQ-VAE just returns inputs."""

    def __init__(self):
        super().__init__()
        self.latent_channels = 1

    def reconstruct(self, x: torch.Tensor) -> torch.Tensor:
        return x

    def decode(self, x: torch.Tensor) -> torch.Tensor:
        return x

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return x

    def encode_stage_2_inputs(self, x: torch.Tensor) -> torch.Tensor:
        return x

    def decode_stage_2_outputs(self, x: torch.Tensor) -> torch.Tensor:
        return x


_FALLBACK_WARNED = set()


def _note_generic(what: str) -> None:
    """One line per layer shape that has no MFMA tiling and takes the generic HIP kernel (slow, same library)."""
    if what not in _FALLBACK_WARNED:
        _FALLBACK_WARNED.add(what)
        import sys

        print(f"NOTE: VQ-VAE layer {what} has no MFMA tiling; it runs on the generic HIP convolution (ddpm_convnd_generic_f32)",
              file=sys.stderr, flush=True)


def _require_device(x):
    if not x.is_cuda:
        raise RuntimeError("VQVAE: the HIP path has no CPU fallback (the CPU oracle under oracle/ is test infrastructure only)")


class _Convolution(nn.Module):
    """monai.networks.blocks.Convolution restricted to what VQVAE uses: a (transposed) ConvNd stored as
    ``.conv`` optionally followed by ``.adn`` = ReLU (adn_ordering "DA" / "NDA" with norm=None, dropout=0)."""

    def __init__(self, spatial_dims, cin, cout, strides=1, kernel_size=3, dilation=1, padding=1, output_padding=0,
                 conv_only=False, is_transposed=False):
        super().__init__()
        if is_transposed:
            conv_t = {2: nn.ConvTranspose2d, 3: nn.ConvTranspose3d}[spatial_dims]
            self.conv = conv_t(cin, cout, kernel_size, strides, padding, output_padding, dilation=dilation)
        else:
            conv_t = {2: nn.Conv2d, 3: nn.Conv3d}[spatial_dims]
            self.conv = conv_t(cin, cout, kernel_size, strides, padding, dilation=dilation)
        self.conv_only = conv_only
        self.is_transposed = is_transposed
        self.geom = (spatial_dims, kernel_size, strides, dilation, padding, output_padding)
        self._packed = None

    def _hip_kind(self, x):
        """Which specialised HIP kernel computes this layer for input x (None: the generic HIP convolution)."""
        sd, k, s, dil, pad, opad = self.geom
        w = self.conv.weight
        if x.is_cuda and sd == 2 and not self.is_transposed and (k, s, dil, pad) == (3, 1, 1, 1) and \
                w.shape[0] % 128 == 0 and w.shape[1] % 4 == 0:
            return "conv2d"  # 2-D VQ-VAEs: the stride-1 3x3 layers with an MFMA tiling on the UNet's convolution
        if not x.is_cuda or sd != 3 or dil != 1 or pad != 1 or opad != 0:
            return None
        even = all(e % 2 == 0 and e >= 2 for e in x.shape[2:])
        if self.is_transposed:
            if (k, s) != (4, 2):
                return None
            if ops.conv3d_supported(w, 2, transposed=True):
                # the split-f16 form (eight parity convolutions on the F(4x4) kernel, round 6) where its tiling exists and the
                # split-f16 families are on; the fp32-MFMA transposed kernel otherwise (smaller levels, ddpm_set_split_f16(0))
                if ops.conv_transpose_parity_supported(x, w) and _lib.split_f16_active() and \
                        os.environ.get("DDPM_CONVT_PARITY", "1") != "0":
                    return "convT_parity"
                return "convT"
            return "convT_cout1" if w.shape[1] == 1 else None
        if (k, s) == (3, 1):
            return "conv" if ops.conv3d_supported(w, 1) else None
        if (k, s) == (4, 2) and even:
            if ops.conv3d_supported(w, 2):
                return "conv"
            return "conv_cin1" if w.shape[1] == 1 else None
        return None

    def forward(self, x):
        w, b = self.conv.weight, self.conv.bias
        kind = self._hip_kind(x)
        out_act = ops.ACT_NONE if self.conv_only else ops.ACT_RELU
        if kind == "conv2d":
            key = (w.data_ptr(), w._version)
            if self._packed is None or self._packed[0] != key:
                # the Winograd kernels take 2-D descriptors only without an output activation (the conv_only layers): a ReLU
                # layer can only reach conv_mfma, so its Winograd planes would be dead device memory
                wino = ops.pack_wino_weight(w.detach()) if out_act == ops.ACT_NONE else None
                self._packed = (key, ops.pack_conv_weight(w.detach()), wino)
            return ops.conv(x.float().contiguous(), w.detach(), b.detach() if b is not None else None, packed=self._packed[1],
                            wino=self._packed[2], out_act=out_act)
        if kind == "convT_parity":
            key = (w.data_ptr(), w._version, "parity")
            if self._packed is None or self._packed[0] != key:
                self._packed = (key, ops.pack_convT_parity_weights(w.detach()))
            return ops.conv_transpose_parity(x.float().contiguous(), self._packed[1], b.detach(), out_act=out_act)
        if kind in ("conv", "convT"):
            key = (w.data_ptr(), w._version)
            if self._packed is None or self._packed[0] != key:
                pack = ops.pack_conv3d_weight if kind == "conv" else ops.pack_convT_weight
                k3 = kind == "conv" and self.geom[1:3] == (3, 1)
                wino = ops.pack_wino3d_weight(w.detach()) if k3 else None
                wino44 = ops.pack_wino44_3d_weight(w.detach()) if k3 else None
                wino44h = ops.pack_wino44h_3d_weight(w.detach()) if k3 else None
                self._packed = (key, pack(w.detach()), wino, wino44, wino44h)
            x = x.float().contiguous()
            if kind == "conv":
                return ops.conv3d(x, w.detach(), b.detach(), packed=self._packed[1], out_act=out_act,
                                  stride=self.geom[2], wino=self._packed[2], wino44=self._packed[3], wino44h=self._packed[4])
            return ops.conv_transpose(x, w.detach(), b.detach(), packed=self._packed[1], out_act=out_act)
        if kind == "conv_cin1":
            return ops.conv3d_k4s2_cin1(x.float().contiguous(), w.detach(), b.detach(), relu=not self.conv_only)
        if kind == "convT_cout1" and self.conv_only:
            return ops.convT3d_k4s2_cout1(x.float().contiguous(), w.detach(), b.detach())
        # no MFMA tiling (channel counts, 2-D, other kernel / stride): the generic kernel of the same library -- never PyTorch
        _require_device(x)
        sd, k, s, dil, pad, opad = self.geom
        if dil != 1 or opad != 0 or s not in (1, 2):
            raise NotImplementedError(f"VQVAE: {type(self.conv).__name__} with dilation {dil} / output_padding {opad} / "
                                      f"stride {s} is not built (no BASELINE configuration uses it)")
        _note_generic(f"{type(self.conv).__name__}{tuple(w.shape)}")
        return ops.convnd_generic(x.float().contiguous(), w.detach(), b.detach() if b is not None else None, stride=s,
                                  padding=pad, transposed=self.is_transposed, relu=not self.conv_only)


class _ResidualUnit(nn.Module):
    def __init__(self, spatial_dims, num_channels, num_res_channels):
        super().__init__()
        self.conv1 = _Convolution(spatial_dims, num_channels, num_res_channels)
        self.conv2 = _Convolution(spatial_dims, num_res_channels, num_channels, conv_only=True)

        self._packed = None  # (key, packed conv1, packed conv2) for the HIP path

    def _hip_weights(self):
        w1, w2 = self.conv1.conv.weight, self.conv2.conv.weight
        key = (w1.data_ptr(), w1._version, w2.data_ptr(), w2._version)
        if self._packed is None or self._packed[0] != key:
            self._packed = (key, ops.pack_conv3d_weight(w1.detach()), ops.pack_conv3d_weight(w2.detach()),
                            ops.pack_wino3d_weight(w1.detach()), ops.pack_wino3d_weight(w2.detach()),
                            ops.pack_wino44_3d_weight(w1.detach()), ops.pack_wino44_3d_weight(w2.detach()),
                            ops.pack_wino44h_3d_weight(w1.detach()), ops.pack_wino44h_3d_weight(w2.detach()))
        return self._packed[1:]

    def forward(self, x):
        w1, w2 = self.conv1.conv.weight, self.conv2.conv.weight
        if x.is_cuda and x.ndim == 5 and ops.conv3d_supported(w1) and ops.conv3d_supported(w2):
            # 95 % of the decoder's FLOPs: both 3x3x3 convolutions on the fp32 MFMA pipe, one launch each (Winograd
            # per depth tap -- F(4x4) from 32^3 up, F(2x2) at 16^3; the direct kernel below that), ReLU /
            # residual fused into the epilogues
            p1, p2, u1, u2, v1, v2, h1, h2 = self._hip_weights()
            x = x.float().contiguous()
            h = ops.conv3d(x, w1.detach(), self.conv1.conv.bias.detach(), out_act=ops.ACT_RELU, packed=p1, wino=u1,
                           wino44=v1, wino44h=h1)
            return ops.conv3d(h, w2.detach(), self.conv2.conv.bias.detach(), residual=x, out_act=ops.ACT_RELU,
                              packed=p2, wino=u2, wino44=v2, wino44h=h2)
        _require_device(x)
        x = x.float().contiguous()
        h = self.conv1(x)  # generic kernel (2-D layers with an MFMA tiling: the UNet's convolution), ReLU fused
        c2 = self.conv2
        sd, k, s, dil, pad, opad = c2.geom
        if c2._hip_kind(h) == "conv2d":  # relu(x + conv2(h)): residual and ReLU in the MFMA kernel's epilogue
            w2 = c2.conv.weight
            key = (w2.data_ptr(), w2._version)
            if c2._packed is None or c2._packed[0] != key:
                c2._packed = (key, ops.pack_conv_weight(w2.detach()), None)  # (ReLU epilogue: conv_mfma only, no Winograd planes)
            return ops.conv(h, w2.detach(), c2.conv.bias.detach() if c2.conv.bias is not None else None, packed=c2._packed[1],
                            residual=x, out_act=ops.ACT_RELU)
        return ops.convnd_generic(h, c2.conv.weight.detach(), c2.conv.bias.detach() if c2.conv.bias is not None else None,
                                  stride=s, padding=pad, residual=x, relu=True)


class _Stack(nn.Module):
    def __init__(self, blocks):
        super().__init__()
        self.blocks = nn.ModuleList(blocks)

    def forward(self, x):
        for b in self.blocks:
            x = b(x)
        return x


class _EMAQuantizer(nn.Module):
    def __init__(self, num_embeddings, embedding_dim):
        super().__init__()
        self.embedding = nn.Embedding(num_embeddings, embedding_dim)
        self.register_buffer("ema_cluster_size", torch.zeros(num_embeddings))
        self.register_buffer("ema_w", self.embedding.weight.data.clone())

    def quantize(self, x):
        """nearest code by squared L2 over channel-last flattened inputs -> indices [B, *spatial] (HIP: vq.hip)"""
        _require_device(x)
        return ops.vq_nearest(x.float().contiguous(), self.embedding.weight.detach())[0]

    def forward(self, x):
        _require_device(x)  # search + lookup + straight-through form x + (q - x) in one HIP kernel
        return ops.vq_nearest(x.float().contiguous(), self.embedding.weight.detach())[1]


class _VectorQuantizer(nn.Module):
    def __init__(self, quantizer):
        super().__init__()
        self.quantizer = quantizer

    def forward(self, x):
        return self.quantizer(x)


class VQVAE(nn.Module):
    """MONAI-Generative's VQVAE as the reconstruction path uses it (reference: /root/reference/src/trainers/base.py:44-61,
    /root/reference/src/trainers/reconstruct.py:124,166): inference only, on the device only -- every layer runs on a HIP kernel
    of libddpm_ood_hip (CPU tensors raise; parameters are read detached, so nothing here is differentiable: VQ-VAE TRAINING is
    off the path).  3-D README shapes take the MFMA kernels; 2-D stride-1 3x3 layers with Cout % 128 == 0 and
    Cin % 4 == 0 the UNet's MFMA convolution (Winograd forms only for the conv_only layers: they have no output activation); everything else the generic kernel ddpm_convnd_generic_f32 (correct,
    slow: one thread per output), announced once per layer shape on stderr."""

    def __init__(self, spatial_dims: int, in_channels: int, out_channels: int, num_channels=(96, 96, 192),
                 num_res_layers: int = 3, num_res_channels=(96, 96, 192),
                 downsample_parameters=((2, 4, 1, 1), (2, 4, 1, 1), (2, 4, 1, 1)),
                 upsample_parameters=((2, 4, 1, 1, 0), (2, 4, 1, 1, 0), (2, 4, 1, 1, 0)), num_embeddings: int = 32,
                 embedding_dim: int = 64, embedding_init: str = "normal", commitment_cost: float = 0.25,
                 decay: float = 0.5, epsilon: float = 1e-5, dropout: float = 0.0, adn_ordering: str = "NDA",
                 act="RELU", output_act=None, ddp_sync: bool = True, use_checkpointing: bool = False):
        super().__init__()
        if isinstance(num_res_channels, int):
            num_res_channels = (num_res_channels,) * len(num_channels)
        if not (len(num_channels) == len(num_res_channels) == len(downsample_parameters) == len(upsample_parameters)):
            raise ValueError("`num_channels`, `num_res_channels`, `downsample_parameters` and `upsample_parameters` "
                             "should have the same length.")
        if output_act is not None or str(act).upper() != "RELU":
            raise NotImplementedError("only act='RELU' / output_act=None (the reference's configuration)")
        self.spatial_dims, self.in_channels, self.out_channels = spatial_dims, in_channels, out_channels
        self.num_embeddings, self.embedding_dim = num_embeddings, embedding_dim
        sd, n = spatial_dims, len(num_channels)
        enc = []
        for i in range(n):
            s, k, dil, pad = downsample_parameters[i]
            enc.append(_Convolution(sd, in_channels if i == 0 else num_channels[i - 1], num_channels[i], s, k, dil, pad))
            enc += [_ResidualUnit(sd, num_channels[i], num_res_channels[i]) for _ in range(num_res_layers)]
        enc.append(_Convolution(sd, num_channels[-1], embedding_dim, conv_only=True))
        self.encoder = _Stack(enc)
        rc, rr = list(reversed(num_channels)), list(reversed(num_res_channels))
        dec = [_Convolution(sd, embedding_dim, rc[0], conv_only=True)]
        for i in range(n):
            dec += [_ResidualUnit(sd, rc[i], rr[i]) for _ in range(num_res_layers)]
            s, k, dil, pad, opad = upsample_parameters[i]
            dec.append(_Convolution(sd, rc[i], out_channels if i == n - 1 else rc[i + 1], s, k, dil, pad, opad,
                                    conv_only=i == n - 1, is_transposed=True))
        self.decoder = _Stack(dec)
        self.quantizer = _VectorQuantizer(_EMAQuantizer(num_embeddings, embedding_dim))

    def encode(self, images):
        return self.encoder(images)

    def quantize(self, encodings):
        return self.quantizer(encodings), torch.zeros((), device=encodings.device)

    def decode(self, quantizations):
        return self.decoder(quantizations)

    def index_quantize(self, images):
        return self.quantizer.quantizer.quantize(self.encode(images))

    def forward(self, images):
        q, loss = self.quantize(self.encode(images))
        return self.decode(q), loss

    def encode_stage_2_inputs(self, x):
        e, _ = self.quantize(self.encode(x))
        return e

    def decode_stage_2_outputs(self, z):
        e, _ = self.quantize(z)  # the denoised latent is re-quantised before decoding
        return self.decode(e)
