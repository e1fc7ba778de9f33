"""Oracle restatement of lpips.LPIPS(net='alex', version='0.1') and the reference's wrapper.

TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED (``lpips==0.1.4`` and its
torchvision AlexNet weights are not obtainable here).  Follows SURVEY.md Appendix A.7 and
  * wrapper ctor / kwargs   /root/reference/src/losses/perceptual_loss.py:47-103
  * 2D forward              /root/reference/src/losses/perceptual_loss.py:105-129
  * 2.5D forward            /root/reference/src/losses/perceptual_loss.py:131-186
    (quirk Q7: the loop over views overwrites ``loss`` so only the last view counts).
Weights are seeded random (conv: default torch init; lin: non-negative) since the trained
ones cannot be fetched; the arithmetic is what is restated, not the metric's meaning.
"""

from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F


class _ScalingLayer(nn.Module):
    def __init__(self):
        super().__init__()
        self.register_buffer("shift", torch.tensor([-0.030, -0.088, -0.188])[None, :, None, None])
        self.register_buffer("scale", torch.tensor([0.458, 0.448, 0.450])[None, :, None, None])

    def forward(self, x):
        return (x - self.shift) / self.scale


class _AlexSlices(nn.Module):
    """torchvision alexnet.features split after each ReLU; indices kept as in lpips."""

    def __init__(self):
        super().__init__()
        self.slice1 = nn.Sequential()
        self.slice2 = nn.Sequential()
        self.slice3 = nn.Sequential()
        self.slice4 = nn.Sequential()
        self.slice5 = nn.Sequential()
        self.slice1.add_module("0", nn.Conv2d(3, 64, 11, 4, 2))
        self.slice1.add_module("1", nn.ReLU())
        self.slice2.add_module("2", nn.MaxPool2d(3, 2))
        self.slice2.add_module("3", nn.Conv2d(64, 192, 5, 1, 2))
        self.slice2.add_module("4", nn.ReLU())
        self.slice3.add_module("5", nn.MaxPool2d(3, 2))
        self.slice3.add_module("6", nn.Conv2d(192, 384, 3, 1, 1))
        self.slice3.add_module("7", nn.ReLU())
        self.slice4.add_module("8", nn.Conv2d(384, 256, 3, 1, 1))
        self.slice4.add_module("9", nn.ReLU())
        self.slice5.add_module("10", nn.Conv2d(256, 256, 3, 1, 1))
        self.slice5.add_module("11", nn.ReLU())

    def forward(self, x):
        h1 = self.slice1(x)
        h2 = self.slice2(h1)
        h3 = self.slice3(h2)
        h4 = self.slice4(h3)
        h5 = self.slice5(h4)
        return [h1, h2, h3, h4, h5]


class _NetLinLayer(nn.Module):
    def __init__(self, chn_in):
        super().__init__()
        self.model = nn.Sequential(nn.Dropout(), nn.Conv2d(chn_in, 1, 1, 1, 0, bias=False))

    def forward(self, x):
        return self.model(x)


def _normalize_tensor(f, eps=1e-10):
    return f / (torch.sqrt(torch.sum(f ** 2, dim=1, keepdim=True)) + eps)


class LPIPSAlex(nn.Module):
    CHNS = (64, 192, 384, 256, 256)

    def __init__(self, seed: int = 1234):
        super().__init__()
        self.scaling_layer = _ScalingLayer()
        self.net = _AlexSlices()
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

    def forward(self, in0, in1, normalize: bool = False):
        if normalize:
            in0 = 2 * in0 - 1
            in1 = 2 * in1 - 1
        f0 = self.net(self.scaling_layer(in0))
        f1 = self.net(self.scaling_layer(in1))
        val = 0
        for k in range(5):
            d = (_normalize_tensor(f0[k]) - _normalize_tensor(f1[k])) ** 2
            val = val + self.lins[k](d).mean([2, 3], keepdim=True)
        return val


class PerceptualLoss(nn.Module):
    def __init__(self, dimensions: int, include_pixel_loss: bool = True, is_fake_3d: bool = True,
                 drop_ratio: float = 0.0, fake_3d_axis=(2, 3, 4), lpips_kwargs=None,
                 lpips_normalize: bool = True, spatial: bool = False, seed: int = 1234):
        super().__init__()
        if dimensions not in (2, 3):
            raise NotImplementedError("Perceptual loss is implemented only in 2D and 3D.")
        if dimensions == 3 and is_fake_3d is False:
            raise NotImplementedError("True 3D perceptual loss is not implemented yet.")
        self.dimensions = dimensions
        self.fake_3D_views = (
            ([((0, 2, 1, 3, 4), (1, 3, 4))] if 2 in fake_3d_axis else [])
            + ([((0, 3, 1, 2, 4), (1, 2, 4))] if 3 in fake_3d_axis else [])
            + ([((0, 4, 1, 2, 3), (1, 2, 3))] if 4 in fake_3d_axis else [])
        ) if is_fake_3d else None
        self.keep_ratio = 1 - drop_ratio
        self.lpips_normalize = lpips_normalize
        self.perceptual_function = LPIPSAlex(seed)
        self.perceptual_factor = 1

    def forward(self, y, y_pred):
        y = y.float()
        y_pred = y_pred.float()
        if self.dimensions == 3 and self.fake_3D_views:
            loss = torch.zeros(())
            for permute_dims, view_dims in self.fake_3D_views:  # Q7: last view wins
                loss = self._fake_3d(y, y_pred, permute_dims, view_dims) * self.perceptual_factor
            return loss
        return self.perceptual_function(y, y_pred, normalize=self.lpips_normalize) * self.perceptual_factor

    def _fake_3d(self, y, y_pred, permute_dims, view_dims):
        ys = y.permute(*permute_dims).contiguous().view(-1, *(y.shape[d] for d in view_dims))
        ps = y_pred.permute(*permute_dims).contiguous().view(-1, *(y_pred.shape[d] for d in view_dims))
        # keep_ratio == 1 on the path: the reference's randperm only reorders the slices
        #  -- identity order is used so that the result is deterministic.
        n = int(ps.shape[0] * self.keep_ratio)
        return torch.mean(self.perceptual_function(ys[:n], ps[:n], normalize=self.lpips_normalize))
