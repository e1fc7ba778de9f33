"""-m gpu: conv_wino44h.hip -- Winograd F(4x4, 3x3) with the position GEMMs on the f16 MFMA (split-f16 products, fp32
accumulate) against ``F.conv2d`` (the op it replaces inside DiffusionModelUNet's ResnetBlocks,
/root/reference/src/trainers/reconstruct.py:151-153), against the fp32-MFMA F(4x4) kernel it supersedes, and against a
float64 convolution over several decades of operand scale (the error budget of the split: DESIGN.md 3.7).

Tolerances are those of tests/test_gpu_ops.py::test_conv_winograd_f4x4 (the fp32 F(4x4) kernel): 2e-4 * (1 + max|ref|) on
single values, 1e-5 relative on the rms -- the split must not cost accuracy.
"""

import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ref_conv(x, x2, w, b, gn, chan_add, residual, dtype=torch.float32):
    xin = (x if x2 is None else torch.cat([x, x2], 1)).to(dtype)
    if gn is not None:
        gamma, beta = gn
        xin = F.silu(F.group_norm(xin, 32, gamma.to(dtype), beta.to(dtype), 1e-6))
    y = F.conv2d(xin, w.to(dtype), b.to(dtype), padding=1)
    if chan_add is not None:
        y = y + chan_add.to(dtype)[:, :, None, None]
    if residual is not None:
        y = y + residual.to(dtype)
    return y


CASES = [
    # B, C1, C2, Cout, H, gn, chan_add, residual
    (2, 64, 0, 64, 32, False, False, False),      # plain: 2 parts per image, one cout tile, 4 chunks
    (3, 128, 0, 128, 32, True, True, True),
    (2, 256, 128, 128, 32, True, True, False),    # virtual concat: 48 chunks
    (5, 256, 0, 256, 16, True, False, True),      # two images per item, ragged last item
    (19, 128, 128, 64, 8, True, True, True),      # eight images per item, ragged
    (1, 64, 0, 64, 64, True, False, False),       # W = 64: two tile rows per item, 8 parts
    (300, 128, 0, 128, 16, True, True, True),     # 2 x 150 items: several items per persistent workgroup
    (70, 64, 64, 128, 32, True, True, True),      # 2 x 2 x 70 = 280 items, concat boundary inside a chunk's halves
    (3, 16, 0, 64, 32, False, False, False),      # the smallest stream: two chunks
    (2, 128, 0, 128, 32, False, True, True),      # no prologue (act = none)
    (9, 64, 0, 64, 8, False, False, True),        # eight images per item without prologue
]


def _inputs(case, scale_x=1.0, scale_w=1.0):
    B, C1, C2, Cout, H, gn, chan, res = case
    g = torch.Generator().manual_seed(hash(case) & 0xFFFF)
    Cin = C1 + C2
    x = torch.randn(B, C1, H, H, generator=g) * scale_x
    x2 = (torch.randn(B, C2, H, H, generator=g) * 1.5 + 0.3) * scale_x if C2 else None
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9) * scale_w
    b = torch.randn(Cout, generator=g)
    gamma = torch.randn(Cin, generator=g) * 0.2 + 1
    beta = torch.randn(Cin, generator=g) * 0.2
    chan_add = torch.randn(B, Cout + 64, generator=g) if chan else None
    residual = torch.randn(B, Cout, H, H, generator=g) if res else None
    return x, x2, w, b, gamma, beta, chan_add, residual


def _run(device, case, tensors, **which):
    from ddpm_ood_amd import ops

    B, C1, C2, Cout, H, gn, chan, res = case
    x, x2, w, b, gamma, beta, chan_add, residual = tensors
    d = lambda t: None if t is None else t.to(device)
    gs = gh = None
    if gn:
        gs, gh = ops.gn_scale_shift(d(x), d(gamma), d(beta), 32, 1e-6, x2=d(x2))
    kw = dict(x2=d(x2), gscale=gs, gshift=gh, act=int(gn), chan_add=d(chan_add), chan_add_offset=32, residual=d(residual))
    return ops.conv(d(x), d(w), d(b), **which, **kw)


@pytest.mark.parametrize("case", CASES)
def test_conv_wino44h_vs_conv2d(device, case, monkeypatch):
    monkeypatch.setenv("DDPM_CONV_WINO44", "2")  # also for launches smaller than the chip
    from ddpm_ood_amd import ops

    B, C1, C2, Cout, H, gn, chan, res = case
    t = _inputs(case)
    x, x2, w, b, gamma, beta, chan_add, residual = t
    ref = _ref_conv(x, x2, w, b, (gamma, beta) if gn else None, chan_add[:, 32:32 + Cout] if chan else None, residual)
    wh = ops.pack_wino44h_weight(w.to(device))
    assert wh is not None and wh.numel() == 2 * 36 * Cout * (C1 + C2) + 64 and wh.dtype == torch.float16
    y = _run(device, case, t, wino44h=wh)
    y44 = _run(device, case, t, wino44=ops.pack_wino44_weight(w.to(device)))
    torch.cuda.synchronize()
    assert not torch.equal(y, y44)  # the split-f16 kernel really ran (and is not the fp32 F(4x4) kernel)
    err = y.cpu() - ref
    assert math.isfinite(err.abs().max().item())
    assert err.abs().max().item() < 2e-4 * (1 + ref.abs().max().item()), err.abs().max().item()
    assert err.pow(2).mean().sqrt().item() < 1e-5 * (1 + ref.pow(2).mean().sqrt().item())
    assert torch.equal(y, _run(device, case, t, wino44h=wh))  # no dependence on leftover LDS state


def test_conv_wino44h_is_selected_by_default_and_switchable(device, monkeypatch):
    """With both packed forms in the descriptor the split-f16 kernel wins; DDPM_WINO44_F16X3=0 restores the fp32 F(4x4)
    kernel bit for bit (the A/B switch of DESIGN.md 4.1)."""
    from ddpm_ood_amd import ops

    case = (70, 128, 0, 128, 32, True, True, True)  # 2 x 2 x 70 = 280 items: fills the chip
    t = _inputs(case)
    w = t[2].to(device)
    wh, w44 = ops.pack_wino44h_weight(w), ops.pack_wino44_weight(w)
    monkeypatch.delenv("DDPM_CONV_WINO44", raising=False)
    both = _run(device, case, t, wino44h=wh, wino44=w44)
    only_h = _run(device, case, t, wino44h=wh)
    only_44 = _run(device, case, t, wino44=w44)
    monkeypatch.setenv("DDPM_WINO44_F16X3", "0")
    off = _run(device, case, t, wino44h=wh, wino44=w44)
    torch.cuda.synchronize()
    assert torch.equal(both, only_h) and torch.equal(off, only_44) and not torch.equal(both, off)


@pytest.mark.parametrize("scale_x,scale_w", [(1.0, 1.0), (1e-3, 1.0), (30.0, 1.0), (1.0, 1e-2), (1.0, 20.0), (1e-2, 10.0)])
def test_conv_wino44h_error_budget_vs_float64(device, scale_x, scale_w, monkeypatch):
    """The split's error budget over operand scales (no GroupNorm in front, so that the activation scale reaches the
    transform): relative rms error against a float64 convolution at most 1.5x the fp32 F(4x4) kernel's + 2e-7, max error
    at most 2x + 1e-6 of the output scale -- i.e. the 22-bit operand representation is not what limits the accuracy."""
    monkeypatch.setenv("DDPM_CONV_WINO44", "2")
    from ddpm_ood_amd import ops

    case = (4, 256, 0, 128, 32, False, False, False)
    t = _inputs(case, scale_x, scale_w)
    x, x2, w, b, *_ = t
    ref = _ref_conv(x, None, w, b, None, None, None, dtype=torch.float64)
    yh = _run(device, case, t, wino44h=ops.pack_wino44h_weight(w.to(device))).cpu().double()
    y4 = _run(device, case, t, wino44=ops.pack_wino44_weight(w.to(device))).cpu().double()
    scale = (ref - b.double()[None, :, None, None]).pow(2).mean().sqrt().item()
    eh, e4 = (yh - ref), (y4 - ref)
    rms_h, rms_4 = eh.pow(2).mean().sqrt().item() / scale, e4.pow(2).mean().sqrt().item() / scale
    max_h, max_4 = eh.abs().max().item() / scale, e4.abs().max().item() / scale
    print(f"scale x {scale_x:g} w {scale_w:g}: split-f16 rms {rms_h:.2e} max {max_h:.2e} | fp32 F(4x4) rms {rms_4:.2e} max {max_4:.2e}")
    assert math.isfinite(max_h)
    assert rms_h <= 1.5 * rms_4 + 2e-7, (rms_h, rms_4)
    assert max_h <= 2.0 * max_4 + 1e-6, (max_h, max_4)


SPLIT_CASES = [
    # B, C1, C2, Cout, H, gn, chan, res: fewer items than CUs -> S workgroups share an item's channel stream
    (256, 256, 0, 256, 8, True, True, True),    # 4 x 32 = 128 items, S = 2
    (16, 128, 256, 128, 32, True, True, False),  # 2 x 4 x 16 = 128 items, 48 chunks, S = 2
]


@pytest.mark.parametrize("case", SPLIT_CASES)
def test_conv_wino44h_channel_split(device, case, monkeypatch):
    monkeypatch.delenv("DDPM_CONV_WINO44", raising=False)
    from ddpm_ood_amd import ops

    B, C1, C2, Cout, H, gn, chan, res = case
    t = _inputs(case)
    x, x2, w, b, gamma, beta, chan_add, residual = t
    ref = _ref_conv(x, x2, w, b, (gamma, beta) if gn else None, chan_add[:, 32:32 + Cout], residual)
    wh = ops.pack_wino44h_weight(w.to(device))
    y = _run(device, case, t, wino44h=wh, wino=ops.pack_wino_weight(w.to(device)))
    err = y.cpu() - ref
    assert err.abs().max().item() < 2e-4 * (1 + ref.abs().max().item()), err.abs().max().item()
    assert err.pow(2).mean().sqrt().item() < 1e-5 * (1 + ref.pow(2).mean().sqrt().item())
    assert torch.equal(y, _run(device, case, t, wino44h=wh, wino=ops.pack_wino_weight(w.to(device))))  # fixed slab order


def test_pack_wino44h_layout(device):
    """The packed planes against a float64 restatement of U = 2^su G g G^T and of the kernel's slot order
    [cout tile][chunk][phase: rows (0,5), (1,2), (3,4)][position 12][plane][cout 64][channel 8]; behind them the layer's
    max |U| and the epilogue scale 1 / (2^3 2^su) with max |2^su U| in [2^14, 2^15); hi + lo reproduces 2^su U to 2^-21 of
    its magnitude wherever the low half is a normal f16 (|2^su U| >= 2^-3: 17 binades below the maximum)."""
    from ddpm_ood_amd import ops

    g = torch.Generator().manual_seed(3)
    Cout, Cin = 128, 32
    for wscale in (0.05, 5e-4, 40.0):
        w = torch.randn(Cout, Cin, 3, 3, generator=g) * wscale
        G = torch.tensor([[0.25, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6],
                          [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=torch.float64)
        U = torch.einsum("ra,ocab,sb->ocrs", G, w.double(), G)  # [Cout, Cin, 6, 6]
        raw = ops.pack_wino44h_weight(w.to(device)).cpu()
        n = 2 * 36 * Cout * Cin
        umax, oscale = raw[n:n + 4].view(torch.float32).tolist()
        assert abs(umax - U.abs().max().item()) <= 1e-6 * umax
        su = round(-math.log2(oscale * 8))
        assert oscale == 2.0 ** -(su + 3) and 2.0 ** 14 <= umax * 2.0 ** su < 2.0 ** 15
        U = U * 2.0 ** su
        packed = raw[:n].double().reshape(Cout // 64, Cin // 8, 3, 12, 2, 64, 8)
        for t, (ra, rb) in enumerate([(0, 5), (1, 2), (3, 4)]):
            for s_ in range(12):
                r, c = (ra if s_ < 6 else rb), s_ % 6
                want = U[:, :, r, c].reshape(Cout // 64, 64, Cin // 8, 8).permute(0, 2, 1, 3)  # [tile, chunk, cout, ch]
                hi, lo = packed[:, :, t, s_, 0], packed[:, :, t, s_, 1]
                assert ((hi - want).abs() <= 2.0 ** -10 * want.abs() + 2.0 ** -24).all()  # hi plane = f16(2^su U)
                assert ((hi + lo - want).abs() <= 2.0 ** -21 * want.abs() + 2.0 ** -24).all()


CASES_3D = [
    # B, C, Cout, D, H, residual + ReLU
    (1, 64, 128, 4, 32, False),      # slices of 64 tiles: two items per slice
    (2, 128, 128, 3, 32, True),      # the VQ-VAE residual unit's second conv: + x, ReLU
    (1, 64, 128, 6, 64, True),       # 64 x 64 slices: two tile rows per item
    (1, 64, 128, 1, 32, False),      # a depth-1 volume only has its centre tap
]


@pytest.mark.parametrize("case", CASES_3D)
def test_conv3d_wino44h_vs_conv3d(device, case, monkeypatch):
    """The 3-D form (2-D F(4x4) per depth tap, the taps accumulated in the transform domain; out-of-volume taps read zeros):
    the stride-1 3x3x3 convolutions of the VQ-VAE residual units (nn.Conv3d inside generative's VQVAE,
    /root/reference/src/trainers/reconstruct.py:124,166) against F.conv3d, and against the fp32-MFMA 3-D F(4x4) kernel."""
    monkeypatch.setenv("DDPM_CONV_WINO44", "2")
    from ddpm_ood_amd import ops

    B, C, Cout, D, H, res = case
    g = torch.Generator().manual_seed(hash(case) & 0xFFFF)
    x = torch.randn(B, C, D, H, H, generator=g)
    w = torch.randn(Cout, C, 3, 3, 3, generator=g) / math.sqrt(27 * C)
    b = torch.randn(Cout, generator=g)
    r = torch.randn(B, Cout, D, H, H, generator=g) if res else None
    ref = F.conv3d(x, w, b, padding=1)
    if res:
        ref = F.relu(ref + r)
    d = lambda t: None if t is None else t.to(device)
    kw = dict(residual=d(r), out_act=ops.ACT_RELU if res else ops.ACT_NONE)
    wh = ops.pack_wino44h_3d_weight(d(w))
    assert wh is not None and wh.numel() == 3 * 2 * 36 * Cout * C + 64
    y = ops.conv3d(d(x), d(w), d(b), wino44h=wh, **kw)
    y4 = ops.conv3d(d(x), d(w), d(b), wino44=ops.pack_wino44_3d_weight(d(w)), **kw)
    torch.cuda.synchronize()
    assert not torch.equal(y, y4)  # the split-f16 kernel ran
    for got in (y, y4):
        err = got.cpu() - ref
        assert err.abs().max().item() < 2e-4 * (1 + ref.abs().max().item()), err.abs().max().item()
        assert err.pow(2).mean().sqrt().item() < 1e-5 * (1 + ref.pow(2).mean().sqrt().item())
    assert torch.equal(y, ops.conv3d(d(x), d(w), d(b), wino44h=wh, **kw))

XITEM_CASES = [
    (300, 128, 0, 128, 16, True, True, True),     # two images per item, 2 x 150 items
    (70, 64, 64, 128, 32, True, True, True),      # one image per two items, 280 items, concat
    (520, 64, 64, 256, 8, True, True, False),     # eight images per item: 4 x 65 items
    (66, 128, 0, 128, 32, False, True, True),     # no prologue
    (1, 64, 0, 64, 64, True, False, False),       # 8 items
]


@pytest.mark.parametrize("case", XITEM_CASES)
def test_conv_wino44h_stream_across_items_is_bit_identical(device, case, monkeypatch):
    """Round 4: a workgroup's pixel waves keep staging across item boundaries (the next item's chunks 0 and 1 enter the pixel
    ring during the current item's last chunks; DDPM_W44H_XITEM).  Same arithmetic in the same order: the output must not
    change by a bit against the per-item refill, on launches where workgroups own SEVERAL items (more items than CUs)."""
    from ddpm_ood_amd import ops

    B, C1, C2, Cout, H, gn, chan, res = case
    t = _inputs(case)
    wh = ops.pack_wino44h_weight(t[2].to(device))
    monkeypatch.setenv("DDPM_CONV_WINO44", "2")
    y1 = _run(device, case, t, wino44h=wh)
    assert torch.equal(y1, _run(device, case, t, wino44h=wh))
    monkeypatch.setenv("DDPM_W44H_XITEM", "0")
    y0 = _run(device, case, t, wino44h=wh)
    torch.cuda.synchronize()
    assert torch.equal(y0, y1)
    x, x2, w, b, gamma, beta, chan_add, residual = t
    ref = _ref_conv(x, x2, w, b, (gamma, beta) if gn else None, chan_add[:, 32:32 + Cout] if chan else None, residual)
    assert (y1.cpu() - ref).abs().max().item() < 2e-4 * (1 + ref.abs().max().item())


def test_conv3d_wino44h_stream_across_items_is_bit_identical(device, monkeypatch):
    """The 3-D form of the same: 2 x 40 slices of 32 x 32 = 320 items (depth taps in the chunk stream, out-of-volume taps zero)."""
    from ddpm_ood_amd import ops

    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 64, 40, 32, 32, generator=g).to(device)
    w = (torch.randn(128, 64, 3, 3, 3, generator=g) / math.sqrt(27 * 64)).to(device)
    b = torch.randn(128, generator=g).to(device)
    r = torch.randn(2, 128, 40, 32, 32, generator=g).to(device)
    wh = ops.pack_wino44h_3d_weight(w)
    kw = dict(residual=r, out_act=ops.ACT_RELU)
    y1 = ops.conv3d(x, w, b, wino44h=wh, **kw)
    monkeypatch.setenv("DDPM_W44H_XITEM", "0")
    y0 = ops.conv3d(x, w, b, wino44h=wh, **kw)
    torch.cuda.synchronize()
    assert torch.equal(y0, y1)
    ref = F.relu(F.conv3d(x, w, b, padding=1) + r)
    assert (y1 - ref).abs().max().item() < 2e-4 * (1 + ref.abs().max().item())


# ---- every item shape of the kernel, pinned bit for bit ------------------------------------------------------------------------
# Until round 6 conv_wino44r.hip was held, output bit for output bit, to its LDS-fed predecessor (conv_wino44h_kernel, retired in
# round 6) over these 72 + 6 cases.  The predecessor's place is taken by digests of those very outputs, written from the
# unchanged kernel source that round 5's suite (and profiles/r06_w44r_ir_route_sets3_bit_identity.log, this round) had held
# to it (tests/golden/wino44h_digests.json, tools/r06/make_w44_digests.py): a refactor of the kernel
# that changes one rounding anywhere shows up here, on every item shape (one image per item, two, eight; concat; residual; no
# prologue; several items per workgroup; channel-split launches; the 3-D and the Upsample form).

def _digest(*tensors):
    import hashlib

    h = hashlib.sha256()
    for t in tensors:
        if t is not None:
            h.update(t.detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()[:24]


def _digests():
    import json
    from pathlib import Path

    return json.load(open(Path(__file__).resolve().parent / "golden" / "wino44h_digests.json"))


def _run_pinned_2d(device, case, monkeypatch):
    from ddpm_ood_amd import ops

    t = _inputs(case)
    w = t[2].to(device)
    wh = ops.pack_wino44h_weight(w)
    split = case in SPLIT_CASES
    if split:
        monkeypatch.delenv("DDPM_CONV_WINO44", raising=False)
    else:
        monkeypatch.setenv("DDPM_CONV_WINO44", "2")
    kw = dict(wino=ops.pack_wino_weight(w)) if split else {}
    y1, st1 = _run(device, case, t, wino44h=wh, want_stats=True, **kw)
    y1b = _run(device, case, t, wino44h=wh, **kw)
    torch.cuda.synchronize()
    return t, y1, st1, y1b


@pytest.mark.parametrize("case", CASES + XITEM_CASES + SPLIT_CASES)
def test_every_item_shape_is_pinned_bit_for_bit(device, case, monkeypatch):
    t, y1, st1, y1b = _run_pinned_2d(device, case, monkeypatch)
    assert torch.equal(y1, y1b)  # with and without statistics: the same output
    B, C1, C2, Cout, H, gn, chan, res = case
    x, x2, wt, b, gamma, beta, chan_add, residual = t
    ref = _ref_conv(x, x2, wt, b, (gamma, beta) if gn else None, chan_add[:, 32:32 + Cout] if chan else None, residual)
    assert (y1.cpu() - ref).abs().max().item() < 2e-4 * (1 + ref.abs().max().item())
    assert _digest(y1, st1) == _digests()["2d"][repr(tuple(case))], "the kernel's rounding changed (see the comment above)"


def _run_pinned_3d(device, case, monkeypatch):
    monkeypatch.setenv("DDPM_CONV_WINO44", "2")
    from ddpm_ood_amd import ops

    B, C, Cout, D, H, res = case
    g = torch.Generator().manual_seed(sum(case) * 7 + 1)
    x = torch.randn(B, C, D, H, H, generator=g).to(device)
    w = (torch.randn(Cout, C, 3, 3, 3, generator=g) / math.sqrt(27 * C)).to(device)
    b = torch.randn(Cout, generator=g).to(device)
    r = torch.randn(B, Cout, D, H, H, generator=g).to(device) if res else None
    kw = dict(residual=r, out_act=ops.ACT_RELU if res else ops.ACT_NONE)
    y1 = ops.conv3d(x, w, b, wino44h=ops.pack_wino44h_3d_weight(w), **kw)
    u1 = ref = None
    if H <= 32:  # Upsample form: 2-D, the same channel counts, low-res H / 2 -> H
        x2d = torch.randn(max(B, 2) * 9, C, H // 2, H // 2, generator=g).to(device)
        w2 = (torch.randn(Cout, C, 3, 3, generator=g) / math.sqrt(9 * C)).to(device)
        u1 = ops.conv(x2d, w2, b, mode=ops.CONV_UPSAMPLE2, wino44h=ops.pack_wino44h_weight(w2))
        ref = F.conv2d(F.interpolate(x2d, scale_factor=2.0, mode="nearest"), w2, b, padding=1)
    torch.cuda.synchronize()
    return y1, u1, ref


@pytest.mark.parametrize("case", CASES_3D)
def test_3d_and_upsample_forms_are_pinned_bit_for_bit(device, case, monkeypatch):
    """The 3-D form (depth taps in the chunk stream) and the Upsample form (nearest x2 read on the fly)."""
    y1, u1, ref = _run_pinned_3d(device, case, monkeypatch)
    if u1 is not None:
        assert (u1 - ref).abs().max().item() < 2e-4 * (1 + ref.abs().max().item())
    assert _digest(y1, u1) == _digests()["3d"][repr(tuple(case))]


# ---- GroupNorm statistics from the epilogue (ddpm_conv_desc.stats_out, ABI 7) ------------------------------------------

STATS_PARTS = {32: 2, 16: 1, 8: 1, 64: 8}  # image extent -> slices per (image, channel): items per image of the kernel


@pytest.mark.parametrize("case", CASES)
def test_conv_wino44h_emits_groupnorm_statistics(device, case, monkeypatch):
    """The epilogue's per-(image, cout, slice) {mean, M2} equal those of the tensor it wrote (float64 reference over the
    kernel's own output), the output is bit-identical with and without them, and they are bit-reproducible."""
    monkeypatch.setenv("DDPM_CONV_WINO44", "2")
    from ddpm_ood_amd import ops

    B, C1, C2, Cout, H, gn, chan, res = case
    t = _inputs(case)
    wh = ops.pack_wino44h_weight(t[2].to(device))
    y_plain = _run(device, case, t, wino44h=wh)
    y, st = _run(device, case, t, wino44h=wh, want_stats=True)
    assert torch.equal(y, y_plain)
    parts = STATS_PARTS[H]
    assert st is not None and tuple(st.shape) == (B, Cout, parts, 2), None if st is None else st.shape
    # slice p = tile rows [p TR, (p + 1) TR) = a contiguous slab of H / parts pixel rows
    yd = y.double().cpu().view(B, Cout, parts, (H // parts) * H)
    mean = yd.mean(-1)
    m2 = (yd - mean[..., None]).pow(2).sum(-1)
    st = st.cpu().double()
    sd = (m2 / yd.shape[-1]).sqrt()
    assert (st[..., 0] - mean).abs().max().item() <= 2e-6 * (1 + mean.abs().max().item() + sd.max().item())
    assert ((st[..., 1] - m2).abs() / (m2 + 1e-3 * m2.mean())).max().item() <= 2e-5
    y2, st2 = _run(device, case, t, wino44h=wh, want_stats=True)
    assert torch.equal(st2.cpu().double(), st)


def _check_stats(y, st, parts):
    B, Cout, H = y.shape[0], y.shape[1], y.shape[2]
    assert st is not None and tuple(st.shape) == (B, Cout, parts, 2), None if st is None else st.shape
    yd = y.double().cpu().view(B, Cout, parts, (H // parts) * H)
    mean = yd.mean(-1)
    m2 = (yd - mean[..., None]).pow(2).sum(-1)
    st = st.cpu().double()
    sd = (m2 / yd.shape[-1]).sqrt()
    assert (st[..., 0] - mean).abs().max().item() <= 2e-6 * (1 + mean.abs().max().item() + sd.max().item())
    assert ((st[..., 1] - m2).abs() / (m2 + 1e-3 * m2.mean())).max().item() <= 2e-5


@pytest.mark.parametrize("case,parts", [((16, 128, 0, 128, 32, True, True, True), 4), ((32, 256, 0, 256, 16, True, True, True), 1),
                                        ((128, 256, 256, 256, 8, True, False, False), 1)])  # 64 items each: four-way split
def test_conv_wino44h_split_launch_statistics_from_reduce_pass(device, case, parts, monkeypatch):
    """A launch smaller than the chip splits the channel stream over 2 / 4 workgroups; the reduce pass that adds the partial
    slabs (+ bias / temb / residual) then writes the statistics: 256-float slices of the plane (4 at 32x32)."""
    from ddpm_ood_amd import ops

    monkeypatch.delenv("DDPM_CONV_WINO44", raising=False)
    t = _inputs(case)
    wh = ops.pack_wino44h_weight(t[2].to(device))
    y_plain = _run(device, case, t, wino44h=wh)
    y, st = _run(device, case, t, wino44h=wh, want_stats=True)
    assert torch.equal(y, y_plain)
    monkeypatch.setenv("DDPM_CONV_WINO44", "2")
    y_unsplit = _run(device, case, t, wino44h=wh)
    assert not torch.equal(y, y_unsplit)  # (the split launch really ran: partial sums are added in another order)
    _check_stats(y, st, parts)


def test_conv_stats_parts_zero_where_not_emitted(device, monkeypatch):
    """Dispatches without an emitting epilogue (kernels other than conv_wino44h / the Upsample kernel) report 0 parts: the
    caller falls back to a reading GroupNorm."""
    from ddpm_ood_amd import ops

    monkeypatch.delenv("DDPM_CONV_WINO44", raising=False)
    case = (4, 128, 0, 128, 32, True, True, True)
    t = _inputs(case)
    w = t[2].to(device)
    y, st = _run(device, case, t, wino44=ops.pack_wino44_weight(w), want_stats=True)
    assert st is None
    y, st = _run(device, case, t, wino=ops.pack_wino_weight(w), want_stats=True)
    assert st is None
    y, st = _run(device, case, t, want_stats=True)  # direct MFMA kernel
    assert st is None


# ---- Upsample convolutions on the same kernel (nearest x2 read on the fly by the pixel waves) ------------------------------

UP_CASES = [
    # B, Cin, Cout, low-res H
    (2, 256, 256, 16),    # the 16 -> 32 Upsample of the small UNet: one image per item, 2 parts
    (3, 256, 256, 8),     # 8 -> 16: two images per item, ragged
    (11, 64, 128, 4),     # 4 -> 8: eight images per item, ragged
    (70, 128, 64, 16),    # 2 x 70 items: several per persistent workgroup
]


@pytest.mark.parametrize("case", UP_CASES)
def test_conv_wino44h_upsample_vs_interpolate_conv2d(device, case, monkeypatch):
    """F.interpolate(scale_factor=2, mode="nearest") + conv3x3 (generative's Upsample, between the levels of the up path,
    /root/reference/src/trainers/reconstruct.py:151-153) as split-f16 F(4x4) over the virtual upsampled image, with the
    GroupNorm statistics of the output from the epilogue."""
    monkeypatch.setenv("DDPM_CONV_WINO44", "2")
    from ddpm_ood_amd import ops

    B, Cin, Cout, H = case
    g = torch.Generator().manual_seed(B * 7 + H)
    x = torch.randn(B, Cin, H, H, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
    b = torch.randn(Cout, generator=g)
    ref = F.conv2d(F.interpolate(x.double(), scale_factor=2, mode="nearest"), w.double(), b.double(), padding=1)
    d = lambda t: t.to(device)
    wh = ops.pack_wino44h_weight(d(w))
    y, st = ops.conv(d(x), d(w), d(b), mode=ops.CONV_UPSAMPLE2, wino44h=wh, want_stats=True)
    y_up = ops.conv(d(x), d(w), d(b), mode=ops.CONV_UPSAMPLE2, wino=ops.pack_wino_weight(d(w)))  # the F(2x2) Upsample kernel
    torch.cuda.synchronize()
    assert not torch.equal(y, y_up)
    err = y.cpu().double() - ref
    assert err.abs().max().item() < 2e-4 * (1 + ref.abs().max().item()), err.abs().max().item()
    assert err.pow(2).mean().sqrt().item() < 1e-5 * (1 + ref.pow(2).mean().sqrt().item())
    _check_stats(y, st, STATS_PARTS[2 * H])
    monkeypatch.setenv("DDPM_UP_WINO44H", "1")
    y2, _ = ops.conv(d(x), d(w), d(b), mode=ops.CONV_UPSAMPLE2, wino44h=wh, want_stats=True)
    assert torch.equal(y, y2)
