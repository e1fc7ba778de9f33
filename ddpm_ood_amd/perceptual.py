"""``PerceptualLoss`` (LPIPS-AlexNet) with the reference's call surface.

Mirror of /root/reference/src/losses/perceptual_loss.py:47-186 over an in-tree restatement of
``lpips.LPIPS(net='alex', version='0.1', lpips=True, spatial=False)`` (the
``lpips`` wheel is not installed here).  On a GPU the whole score runs on the library's HIP
kernels (``csrc/lpips.hip`` for the 11x11 / 5x5 layers, max-pools and the
normalise-diff-lin-mean reduction, the MFMA convolution with a ReLU epilogue for the three 3x3
layers); CPU tensors take the plain PyTorch path below, which is what the oracle is checked
against.  24 MFLOP per image pair at 32x32 -- 6e-5 of a reconstruction's cost.

state_dict keys follow lpips (``net.slice1.0.weight`` ... ``lins.0.model.1.weight``,
``scaling_layer.shift/scale``) so real LPIPS weights can be loaded with ``--lpips_weights``
(``LPIPS.load_pretrained_state_dict`` folds the package's duplicate ``linN`` aliases); without them
seeded synthetic weights are used (the trained ones need the network) and the trainer says so loudly:
the ``perceptual_difference`` column is then NOT an LPIPS value, only ``mse`` is meaningful.
"""

from __future__ import annotations

import os
from typing import Dict, Tuple

import torch
import torch.nn as nn


class _ScalingLayer(nn.Module):
    def __init__(self):
        super().__init__()
        self.register_buffer("shift", torch.tensor([-0.030, -0.088, -0.188])[None, :, None, None])
        self.register_buffer("scale", torch.tensor([0.458, 0.448, 0.450])[None, :, None, None])


class _Alex(nn.Module):
    def __init__(self):
        super().__init__()
        spec = {
            "slice1": [("0", nn.Conv2d(3, 64, 11, 4, 2)), ("1", nn.ReLU())],
            "slice2": [("2", nn.MaxPool2d(3, 2)), ("3", nn.Conv2d(64, 192, 5, 1, 2)), ("4", nn.ReLU())],
            "slice3": [("5", nn.MaxPool2d(3, 2)), ("6", nn.Conv2d(192, 384, 3, 1, 1)), ("7", nn.ReLU())],
            "slice4": [("8", nn.Conv2d(384, 256, 3, 1, 1)), ("9", nn.ReLU())],
            "slice5": [("10", nn.Conv2d(256, 256, 3, 1, 1)), ("11", nn.ReLU())],
        }
        for name, mods in spec.items():
            seq = nn.Sequential()
            for k, m in mods:
                seq.add_module(k, m)
            setattr(self, name, seq)


class _NetLinLayer(nn.Module):
    def __init__(self, chn_in):
        super().__init__()
        self.model = nn.Sequential(nn.Dropout(), nn.Conv2d(chn_in, 1, 1, 1, 0, bias=False))


class LPIPS(nn.Module):
    CHNS = (64, 192, 384, 256, 256)

    def __init__(self, seed: int = 1234, **_ignored):
        super().__init__()
        self.scaling_layer = _ScalingLayer()
        self.net = _Alex()
        self.lins = nn.ModuleList([_NetLinLayer(c) for c in self.CHNS])
        g = torch.Generator().manual_seed(seed)
        with torch.no_grad():
            for p in self.net.parameters():
                bound = 1.0 / (p[0].numel() ** 0.5) if p.ndim > 1 else 0.05
                p.copy_((torch.rand(p.shape, generator=g) * 2 - 1) * bound)
            for lin in self.lins:
                w = lin.model[1].weight
                w.copy_(torch.rand(w.shape, generator=g) / w.shape[1])
        self.eval()
        for p in self.parameters():
            p.requires_grad_(False)
        self._packed = {}  # MFMA-packed 3x3 weights, built on first device use
        self.pretrained = False  # True once trained weights were loaded (load_pretrained_state_dict)

    def load_pretrained_state_dict(self, state_dict) -> None:
        """Load trained LPIPS-AlexNet weights saved from the real package: ``lpips.LPIPS(net="alex").state_dict()``
        (keys ``scaling_layer.*``, ``net.sliceN.K.*``, ``lins.N.model.1.weight`` AND the duplicate aliases
        ``linN.model.1.weight`` -- lpips 0.1.4 registers every lin layer twice), the reference's
        ``PerceptualLoss(...).state_dict()`` (same keys under ``perceptual_function.``), or lpips' own
        ``weights/v0.1/alex.pth`` merged with torchvision's AlexNet features.  Aliases are folded, every tensor this
        module owns must be present with the right shape, and unknown keys raise."""
        sd = {}
        for k, v in state_dict.items():
            if k.startswith("perceptual_function."):
                k = k[len("perceptual_function."):]
            for n in range(5):
                if k.startswith(f"lin{n}."):
                    k = f"lins.{n}." + k[len(f"lin{n}."):]
            if k in sd and not torch.equal(sd[k], v):
                raise ValueError(f"LPIPS state_dict: aliases of '{k}' disagree")
            sd[k] = v
        own = self.state_dict()
        missing = sorted(set(own) - set(sd))
        unexpected = sorted(set(sd) - set(own))
        if missing or unexpected:
            raise KeyError(f"LPIPS state_dict: missing {missing}, unexpected {unexpected}")
        for k, v in sd.items():
            if tuple(v.shape) != tuple(own[k].shape):
                raise ValueError(f"LPIPS state_dict: {k} is {tuple(v.shape)}, expected {tuple(own[k].shape)}")
        self.load_state_dict(sd, strict=True)
        self._packed = {}
        self.pretrained = True

    def _features_hip(self, x, normalize: bool):
        """AlexNet slices 1-5 on the HIP kernels.  The "2x - 1" of normalize=True and the ScalingLayer
        (x - shift) / scale are one per-channel affine applied while conv1 reads its input."""
        from . import ops

        shift, scale = self.scaling_layer.shift.reshape(-1), self.scaling_layer.scale.reshape(-1)
        a = (2.0 if normalize else 1.0) / scale
        b = ((-1.0 if normalize else 0.0) - shift) / scale
        convs = [self.net.slice1[0], self.net.slice2[1], self.net.slice3[1], self.net.slice4[0], self.net.slice5[0]]
        outs = []
        c0 = convs[0]
        if x.shape[1] == 1 and x.shape[0] >= 64:
            # grey slices in bulk (2.5-D LPIPS): the 1 -> 3 broadcast and the per-channel affine fold into one input
            # channel, conv(a_c x + b_c) = (sum_c a_c w_c) * x + [(sum_c b_c w_c) * inside + bias] -- a third of the
            # multiplies; the bracket does not depend on the image (a per-position bias map, computed once per shape
            # by the general kernel on a zero image)
            key = ("c0", c0.weight.data_ptr(), c0.weight._version, bool(normalize), tuple(x.shape[2:]), str(x.device))
            fold = self._packed.get("c0")
            if fold is None or fold[0] != key:
                wa = (c0.weight.detach() * a[None, :, None, None]).sum(1, keepdim=True).contiguous()
                zero = torch.zeros((1, 1) + tuple(x.shape[2:]), dtype=torch.float32, device=x.device)
                bmap = ops.lpips_conv(zero, c0.weight, c0.bias, 4, 2, False, a.contiguous(), b.contiguous())[0]
                fold = self._packed["c0"] = (key, wa, bmap.contiguous())
            h = ops.lpips_conv_biasmap(x, fold[1], fold[2], 4, 2, True)
        else:
            h = ops.lpips_conv(x, c0.weight, c0.bias, 4, 2, True, a.contiguous(), b.contiguous())
        outs.append(h)
        h = ops.maxpool3s2(h)
        c1 = convs[1]
        if ops.lpips_conv_mfma_supported(c1.weight.shape[1], h.shape[2], h.shape[3], c1.weight.shape[0], 5):
            # maps of >= 64 pixels (2.5-D LPIPS over volumes): the 5x5 layer on the fp32 MFMA pipe
            key = ("l5", id(c1), c1.weight.data_ptr(), c1.weight._version)
            packed = self._packed.get("l5")
            if packed is None or packed[0] != key or packed[1].device != h.device:
                packed = self._packed["l5"] = (key, ops.lpips_pack_conv_weight(c1.weight.detach()))
            h = ops.lpips_conv_mfma(h, packed[1], c1.bias, c1.weight.shape[0], 5, True)
        else:
            h = ops.lpips_conv(h, c1.weight, c1.bias, 1, 2, True)
        outs.append(h)
        h = ops.maxpool3s2(h)
        for c in convs[2:]:
            packed = self._packed.get(id(c))
            if packed is None or packed.device != h.device:
                packed = self._packed[id(c)] = ops.pack_conv_weight(c.weight)
            if packed is not None:   # 3x3, Cin % 4 == 0, Cout % 128 == 0: the MFMA kernel, ReLU in its epilogue
                h = ops.conv(h, c.weight, c.bias, packed=packed, out_act=ops.ACT_RELU)
            else:
                h = ops.lpips_conv(h, c.weight, c.bias, 1, 1, True)
            outs.append(h)
        return outs

    def _forward_hip(self, in0, in1, normalize: bool):
        from . import ops

        n = in0.shape[0]
        feats = self._features_hip(torch.cat([in0, in1], 0).contiguous(), normalize)
        val = None
        for k, f in enumerate(feats):
            lin = self.lins[k].model[1].weight.reshape(-1)
            val = ops.lpips_layer(f[:n], f[n:], lin, val)
        return val.reshape(n, 1, 1, 1)

    def forward(self, in0, in1, normalize: bool = False):
        if not (isinstance(in0, torch.Tensor) and in0.is_cuda and in1.is_cuda):
            raise RuntimeError("LPIPS inputs must be ROCm device tensors: the HIP reconstruction path has no CPU fallback "
                               "(the CPU restatement is oracle/lpips.py, test infrastructure only)")
        if in0.shape != in1.shape or in0.shape[1] not in (1, 3):
            raise ValueError(f"LPIPS wants two equal [N, 1|3, H, W] batches, got {tuple(in0.shape)} and {tuple(in1.shape)}")
        return self._forward_hip(in0.float(), in1.float(), normalize)


class PerceptualLoss(nn.Module):
    def __init__(self, dimensions: int, include_pixel_loss: bool = True, is_fake_3d: bool = True,
                 drop_ratio: float = 0.0, fake_3d_axis: Tuple[int, ...] = (2, 3, 4), lpips_kwargs: Dict = None,
                 lpips_normalize: bool = True, spatial: bool = False):
        super().__init__()
        if dimensions not in (2, 3):
            raise NotImplementedError("Perceptual loss is implemented only in 2D and 3D.")
        if dimensions == 3 and is_fake_3d is False:
            raise NotImplementedError("True 3D perceptual loss is not implemented yet.")
        self.dimensions = dimensions
        self.include_pixel_loss = include_pixel_loss
        self.fake_3D_views = (
            ([((0, 2, 1, 3, 4), (1, 3, 4))] if 2 in fake_3d_axis else [])
            + ([((0, 3, 1, 2, 4), (1, 2, 4))] if 3 in fake_3d_axis else [])
            + ([((0, 4, 1, 2, 3), (1, 2, 3))] if 4 in fake_3d_axis else [])
        ) if is_fake_3d else None
        self.keep_ratio = 1 - drop_ratio
        self.lpips_normalize = lpips_normalize
        self.perceptual_function = LPIPS(**(lpips_kwargs or {}))
        self.perceptual_factor = 1

    @torch.no_grad()
    def forward(self, y: torch.Tensor, y_pred: torch.Tensor) -> torch.Tensor:
        y = y.float()
        y_pred = y_pred.float()
        if self.dimensions == 3 and self.fake_3D_views:
            # Reference quirk Q7 (perceptual_loss.py:112-122): the loop ASSIGNS `loss` per view, so only the last view
            # reaches the caller.  The earlier views are dead stores -- same returned value with or without them -- and
            # are not computed here; DDPM_LPIPS_ALL_VIEWS=1 computes them anyway (the reference's literal work).
            views = self.fake_3D_views if os.environ.get("DDPM_LPIPS_ALL_VIEWS", "0") == "1" else self.fake_3D_views[-1:]
            loss = torch.zeros(())
            for permute_dims, view_dims in views:
                loss = self._calculate_fake_3d_loss(y, y_pred, permute_dims, view_dims) * self.perceptual_factor
            return loss
        return self.perceptual_function.forward(y, y_pred, normalize=self.lpips_normalize) * self.perceptual_factor

    def _calculate_fake_3d_loss(self, y, y_pred, permute_dims, view_dims):
        ys = y.permute(*permute_dims).contiguous().view(-1, *(y.shape[d] for d in view_dims))
        ps = y_pred.permute(*permute_dims).contiguous().view(-1, *(y_pred.shape[d] for d in view_dims))
        n = int(ps.shape[0] * self.keep_ratio)  # keep_ratio == 1 on the path; identity order (Q7)
        return torch.mean(self.perceptual_function.forward(ys[:n], ps[:n], normalize=self.lpips_normalize))

    def get_perceptual_factor(self) -> float:
        return self.perceptual_factor

    def set_perceptual_factor(self, perceptual_factor: float) -> float:
        self.perceptual_factor = perceptual_factor
        return self.get_perceptual_factor()
