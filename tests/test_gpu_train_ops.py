"""-m gpu: the kernels of the native training step (row f-3; include/ddpm_ood_hip.h "Training step", ABI 10) one by one against
CPU torch in float64 -- what loss.backward() / optimizer.step() of /root/reference/src/trainers/ddpm_trainer.py:78-109 dispatch to."""

import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a.double().cpu() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


# ---- ddpm_gemm_f32 -----------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K", [(64, 64, 32), (100, 70, 45), (256, 512, 32), (5, 300, 257), (128, 192, 4096), (68, 132, 72)])
def test_gemm_plain_and_transposed_operands(device, M, N, K):
    from ddpm_ood_amd import train_ops as T

    g = torch.Generator().manual_seed(M + N + K)
    A, B = torch.randn(M, K, generator=g), torch.randn(K, N, generator=g)
    ref = A.double() @ B.double()
    Ad, Bd = A.to(device), B.to(device)
    C = torch.empty(M, N, device=device)
    T.gemm(Ad, Bd, C, M, N, K, a_m=K, a_k=1, b_k=N, b_n=1, c_m=N, c_n=1)
    assert _rel(C, ref) < 2e-6
    # A^T and B^T stored, C written transposed, alpha / beta
    At, Bt = A.t().contiguous().to(device), B.t().contiguous().to(device)
    C0 = torch.randn(N, M, generator=g)
    Ct = C0.clone().to(device)
    T.gemm(At, Bt, Ct, M, N, K, a_m=1, a_k=M, b_k=1, b_n=K, c_m=1, c_n=M, alpha=0.5, beta=2.0)
    assert _rel(Ct, 0.5 * ref.t() + 2.0 * C0.double()) < 2e-6


def test_gemm_two_level_k_and_batch_like_a_1x1_weight_gradient_and_attention(device):
    from ddpm_ood_amd import train_ops as T

    g = torch.Generator().manual_seed(5)
    B, cout, cin, hw = 3, 70, 40, 48
    dy, x = torch.randn(B, cout, hw, generator=g), torch.randn(B, cin, hw, generator=g)
    ref = torch.einsum("bop,bip->oi", dy.double(), x.double())
    dw = torch.empty(cout, cin, device=device)
    T.gemm(dy.to(device), x.to(device), dw, cout, cin, B * hw, k_inner=hw, a_m=hw, a_k=1, a_k_outer=cout * hw, b_n=hw, b_k=1,
           b_k_outer=cin * hw, c_m=cin, c_n=1)
    assert _rel(dw, ref) < 2e-6
    # ... and on the f16 MFMA at split precision when the caller vouches for the operands' range (both K-major, aligned extents)
    Bq, co2, ci2, hw2 = 5, 192, 128, 64
    dy2, x2 = 0.3 * torch.randn(Bq, co2, hw2, generator=g), 2.0 * torch.randn(Bq, ci2, hw2, generator=g)
    ref2 = torch.einsum("bop,bip->oi", dy2.double(), x2.double())
    outs = []
    for flag in (True, False):
        dw2 = torch.empty(co2, ci2, device=device)
        T.gemm(dy2.to(device), x2.to(device), dw2, co2, ci2, Bq * hw2, k_inner=hw2, a_m=hw2, a_k=1, a_k_outer=co2 * hw2, b_n=hw2,
               b_k=1, b_k_outer=ci2 * hw2, c_m=ci2, c_n=1, split_f16=flag)
        assert _rel(dw2, ref2) < 2e-6, (flag, _rel(dw2, ref2))
        outs.append(dw2)
    assert not torch.equal(outs[0], outs[1])  # (two different kernels ran)
    # one batch level (a 1x1 convolution's input gradient: dx[b] += W^T dy[b], the weight shared by every image)
    w = torch.randn(cout, cin, generator=g)
    dx0 = torch.randn(B, cin, hw, generator=g)
    dx = dx0.clone().to(device)
    T.gemm(w.to(device), dy.to(device), dx, cin, hw, cout, a_m=1, a_k=cin, b_k=hw, b_n=1, c_m=hw, c_n=1, batch=B, a_batch=0,
           b_batch=cout * hw, c_batch=cin * hw, beta=1.0)
    assert _rel(dx, dx0.double() + torch.einsum("oi,bop->bip", w.double(), dy.double())) < 2e-6
    # batched with (image, head) batch levels: S[b, h, i, j] = scale sum_c q[b, h d + c, i] k[b, h d + c, j]
    Bn, heads, d, n = 2, 3, 16, 40
    q, k = torch.randn(Bn, heads * d, n, generator=g), torch.randn(Bn, heads * d, n, generator=g)
    ref = 0.25 * torch.einsum("bhci,bhcj->bhij", q.view(Bn, heads, d, n).double(), k.view(Bn, heads, d, n).double())
    S = torch.empty(Bn * heads, n, n, device=device)
    T.gemm(q.to(device), k.to(device), S, n, n, d, a_m=1, a_k=n, b_k=n, b_n=1, c_m=n, c_n=1, batch=Bn * heads, batch_inner=heads,
           a_batch=d * n, a_batch_outer=heads * d * n, b_batch=d * n, b_batch_outer=heads * d * n, c_batch=n * n,
           c_batch_outer=heads * n * n, alpha=0.25)
    assert _rel(S.view(Bn, heads, n, n), ref) < 2e-6


# ---- ddpm_conv_wgrad_f32 -----------------------------------------------------------------------------------------------------
WGRAD = [  # B, Cin, Cout, H, W, ksize, stride
    (4, 128, 128, 32, 32, 3, 1), (3, 192, 64, 16, 16, 3, 1), (5, 64, 128, 8, 8, 3, 1), (2, 64, 64, 64, 64, 3, 1),
    (3, 64, 64, 12, 12, 3, 1),   # ragged pixel tiles (5 + 5 + 2 rows)
    (4, 128, 128, 32, 32, 3, 2), (3, 64, 128, 16, 16, 3, 2), (2, 64, 64, 8, 8, 3, 2),
    (4, 1, 128, 32, 32, 3, 1), (4, 128, 3, 32, 32, 3, 1), (2, 96, 40, 8, 8, 1, 1),  # no MFMA tiling: one workgroup per (cout, cin)
    (37, 1, 128, 32, 32, 3, 1), (40, 128, 1, 32, 32, 3, 1),  # ... and per image slice when the pairs alone do not fill the chip
]


@pytest.mark.parametrize("case", WGRAD)
def test_conv_wgrad_vs_autograd(device, case):
    from ddpm_ood_amd import _lib
    from ddpm_ood_amd import train_ops as T

    B, cin, cout, H, W, k, s = case
    g = torch.Generator().manual_seed(sum(case))
    a = torch.randn(B, cin, H, W, generator=g)
    w = torch.zeros(cout, cin, k, k, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(a.double(), w, stride=s, padding=k // 2)
    dy = torch.randn(y.shape, generator=g)
    (ref,) = torch.autograd.grad(y, w, dy.double())
    mfma = k == 3 and cin % 64 == 0 and cout % 64 == 0
    assert not mfma or _lib.load().ddpm_conv_wgrad_scratch_floats(B, cin, cout, H, W, y.shape[2], y.shape[3], k, s) > 0
    dw = T.conv_wgrad(a.to(device), dy.to(device), k, s)
    assert dw.shape == ref.shape and _rel(dw, ref) < 3e-6, _rel(dw, ref)
    if mfma:  # the two forms agree, and the MFMA form is bit-reproducible (fixed-order reduce, no atomics)
        gen = T.conv_wgrad(a.to(device), dy.to(device), k, s, force_generic=True)
        assert _rel(gen, ref) < 3e-6
        assert torch.equal(dw, T.conv_wgrad(a.to(device), dy.to(device), k, s))


@pytest.mark.parametrize("case", [(4, 128, 128, 32, 32), (3, 192, 64, 16, 16), (5, 64, 128, 8, 8), (2, 64, 64, 64, 64)])
@pytest.mark.parametrize("dy_scale,a_scale", [(1.0, 1.0), (3e-7, 40.0), (5e3, 1e-3)])
def test_conv_wgrad_split_f16_form(device, monkeypatch, case, dy_scale, a_scale):
    """The split-f16 form of the 3x3 weight gradient (stride 1, W in 8 .. 64): operands of any magnitude (both are rescaled by a
    power of two from their measured maxima) and a wide spread inside one tensor; it is at least as close to float64 as the fp32-MFMA
    form's bound, and DDPM_WGRAD_F16X3=0 / ddpm_set_split_f16(0) select the fp32 form."""
    from ddpm_ood_amd import _lib
    from ddpm_ood_amd import train_ops as T

    B, cin, cout, H, W = case
    g = torch.Generator().manual_seed(sum(case))
    # a: SiLU-like (many small values, a few large ones); dy: gradients spread over four decades
    a = a_scale * torch.randn(B, cin, H, W, generator=g) * torch.exp(1.5 * torch.randn(B, cin, 1, 1, generator=g))
    dy = dy_scale * torch.randn(B, cout, H, W, generator=g) * torch.exp(2.0 * torch.randn(B, cout, 1, 1, generator=g))
    w = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, requires_grad=True)
    (ref,) = torch.autograd.grad(F.conv2d(a.double(), w, padding=1), w, dy.double())
    ad, dyd = a.to(device), dy.to(device)
    dw = T.conv_wgrad(ad, dyd, 3, 1)
    assert _rel(dw, ref) < 3e-6, _rel(dw, ref)
    assert torch.equal(dw, T.conv_wgrad(ad, dyd, 3, 1))
    monkeypatch.setenv("DDPM_WGRAD_F16X3", "0")
    f32 = T.conv_wgrad(ad, dyd, 3, 1)
    assert _rel(f32, ref) < 3e-6
    assert not torch.equal(f32, dw)  # (a different kernel ran)
    monkeypatch.delenv("DDPM_WGRAD_F16X3")
    _lib.load().ddpm_set_split_f16(0)
    try:
        assert torch.equal(T.conv_wgrad(ad, dyd, 3, 1), f32)
    finally:
        _lib.load().ddpm_set_split_f16(1)
    # operand maxima handed in (as the GroupNorm kernels emit them: partial maxima, any number): the same scale, the same bits
    amax = ad.abs().view(B, -1).amax(dim=1).view(torch.int32)
    dmax = dyd.abs().view(B * 4, -1).amax(dim=1).view(torch.int32)
    assert torch.equal(T.conv_wgrad(ad, dyd, 3, 1, a_absmax=amax, dy_absmax=dmax), dw)
    assert torch.equal(T.conv_wgrad(ad, dyd, 3, 1, dy_absmax=dmax), dw)
    # a non-finite gradient stays non-finite (the scale is taken from the maximum's bit pattern, NaN included)
    bad = dyd.clone()
    bad[0, 0, 0, 0] = float("nan")
    assert not bool(torch.isfinite(T.conv_wgrad(ad, bad, 3, 1)).all())
    # an all-zero gradient gives an all-zero weight gradient
    assert float(T.conv_wgrad(ad, torch.zeros_like(dyd), 3, 1).abs().max()) == 0.0


@pytest.mark.parametrize("case", [(2, 128, 128, 8, 1), (3, 64, 128, 4, 1), (2, 128, 64, 8, 2), (2, 64, 64, 4, 2), (1, 64, 64, 2, 1)])
def test_conv3d_wgrad_and_input_gradient_vs_autograd(device, case):
    """F.conv3d(k3, padding 1, stride 1 / 2) on NCDHW latents: the per-depth-tap weight gradient and the input gradient through
    ops.conv3d with rotated / transposed weights (stride 2: through a zero-stuffed dy)."""
    from ddpm_ood_amd import ops
    from ddpm_ood_amd import train_ops as T

    B, cin, cout, S, stride = case
    g = torch.Generator().manual_seed(sum(case))
    a = torch.randn(B, cin, S, S, S, generator=g)
    w0 = torch.randn(cout, cin, 3, 3, 3, generator=g) / math.sqrt(27 * cin)
    ad, wd = a.double().requires_grad_(True), w0.double().requires_grad_(True)
    y = F.conv3d(ad, wd, stride=stride, padding=1)
    dy = torch.randn(y.shape, generator=g)
    ra, rw = torch.autograd.grad(y, (ad, wd), dy.double())
    dw = T.conv3d_wgrad(a.to(device), dy.to(device), stride)
    assert dw.shape == rw.shape and _rel(dw, rw) < 3e-6, _rel(dw, rw)
    assert torch.equal(dw, T.conv3d_wgrad(a.to(device), dy.to(device), stride))
    if cin % 128 == 0 and cout % 4 == 0:  # the input-gradient convolution is cout -> cin: needs an MFMA tiling of its own
        wt = T.conv_weight_rot180t(w0.to(device))
        assert torch.equal(wt.cpu(), w0.flip(2, 3, 4).transpose(0, 1).contiguous())
        d = dy.to(device) if stride == 1 else T.zero_stuff2(dy.to(device))
        dx = ops.conv3d(d, wt, wino=ops.pack_wino3d_weight(wt))
        assert dx.shape == ra.shape and _rel(dx, ra) < 2e-5, _rel(dx, ra)


def test_volume_resampling_kernels(device):
    from ddpm_ood_amd import train_ops as T

    g = torch.Generator().manual_seed(3)
    s = torch.randn(2, 3, 4, 5, 6, generator=g)
    assert torch.equal(T.upsample2(s.to(device)).cpu(), F.interpolate(s, scale_factor=2.0, mode="nearest"))
    big = torch.randn(2, 3, 8, 10, 12, generator=g)
    assert _rel(T.sumpool2(big.to(device)), 8 * F.avg_pool3d(big.double(), 2)) < 1e-6
    z = T.zero_stuff2(s.to(device)).cpu()
    assert torch.equal(z[:, :, ::2, ::2, ::2], s) and float(z.abs().sum()) == float(s.abs().sum())


# ---- input gradients through ddpm_conv_f32 with rotated / transposed weights ------------------------------------------------
@pytest.mark.parametrize("case", [(4, 128, 128, 32, "s1"), (4, 128, 256, 16, "s1"), (4, 128, 128, 32, "s2"), (3, 256, 256, 8, "up"),
                                  (4, 128, 1, 32, "s1")])
def test_conv_input_gradient_forms_vs_autograd(device, case):
    from ddpm_ood_amd import ops
    from ddpm_ood_amd import train_ops as T

    B, cin, cout, H, kind = case
    g = torch.Generator().manual_seed(H + cin)
    w = torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(9 * cin)
    x = torch.zeros(B, cin, H, H, dtype=torch.float64, requires_grad=True)
    if kind == "s1":
        y = F.conv2d(x, w.double(), padding=1)
    elif kind == "s2":
        y = F.conv2d(x, w.double(), stride=2, padding=1)
    else:
        y = F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest"), w.double(), padding=1)
    dy = torch.randn(y.shape, generator=g)
    (ref,) = torch.autograd.grad(y, x, dy.double())
    wt = T.conv_weight_rot180t(w.to(device))
    assert torch.equal(wt.cpu(), w.flip(2, 3).transpose(0, 1).contiguous())
    d = dy.to(device)
    if kind == "s2":
        d = T.zero_stuff2(d)
    dx = ops.conv(d, wt, wino44h=ops.pack_wino44h_weight(wt), wino=ops.pack_wino_weight(wt))
    if kind == "up":
        dx = T.sumpool2(dx)
    assert dx.shape == ref.shape and _rel(dx, ref) < 2e-5, _rel(dx, ref)


# ---- GroupNorm (+ SiLU) training form -----------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,C,H,act", [(4, 128, 32, 1), (3, 256, 8, 1), (2, 384, 16, 0), (2, 64, 4, 1), (2, 64, 5, 1), (5, 512, 8, 1),
                                       (2, 96, 7, 0), (2, 256, 32, 1), (1, 512, 32, 1), (2, 384, 32, 0), (1, 128, 64, 1)])
def test_group_norm_forward_and_backward_vs_autograd(device, B, C, H, act):
    from ddpm_ood_amd import train_ops as T

    g = torch.Generator().manual_seed(C + H)
    x = (torch.randn(B, C, H, H, generator=g) * 1.7 + 0.3)
    gamma, beta = 1 + 0.2 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    xd = x.double().requires_grad_(True)
    gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    y = F.group_norm(xd, 32, gd, bd, eps=1e-6)
    if act:
        y = F.silu(y)
    dy = torch.randn(y.shape, generator=g)
    rx, rg, rb = torch.autograd.grad(y, (xd, gd, bd), dy.double())
    dev = lambda t: t.to(device)  # noqa: E731
    mr = T.gn_stats(dev(x), 32, 1e-6)
    yh = T.gn_apply(dev(x), mr, dev(gamma), dev(beta), 32, act)
    assert _rel(yh, y.detach()) < 3e-6
    yf, mrf = T.gn_forward(dev(x), dev(gamma), dev(beta), 32, 1e-6, act)  # statistics + apply in one call (one kernel where it fits)
    assert _rel(yf, y.detach()) < 3e-6 and _rel(mrf, mr.cpu()) < 3e-6
    assert torch.equal(yf, T.gn_forward(dev(x), dev(gamma), dev(beta), 32, 1e-6, act)[0])
    dgam, dbet = torch.empty(C, device=device), torch.empty(C, device=device)
    dx = T.gn_backward(dev(x), dev(dy), mr, dev(gamma), dev(beta), 32, act, dgam, dbet)
    assert _rel(dx, rx) < 1e-5 and _rel(dgam, rg) < 1e-5 and _rel(dbet, rb) < 1e-5, (_rel(dx, rx), _rel(dgam, rg), _rel(dbet, rb))
    base = torch.randn(x.shape, generator=g)
    acc = T.gn_backward(dev(x), dev(dy), mr, dev(gamma), dev(beta), 32, act, dgam, dbet, dx=dev(base), accumulate=True)
    assert _rel(acc, rx + base.double()) < 1e-5
    # the optional outputs: maxima per (image, group) as float bit patterns, plane sums -- of the FINAL dx / y
    yf, _, ymax = T.gn_forward(dev(x), dev(gamma), dev(beta), 32, 1e-6, act, want_absmax=True)
    want = yf.abs().view(B, 32, -1).amax(dim=2).reshape(-1)
    assert torch.equal(ymax.view(torch.float32), want)
    acc2, dmax, rows = T.gn_backward(dev(x), dev(dy), mr, dev(gamma), dev(beta), 32, act, dgam, dbet, dx=dev(base), accumulate=True,
                                     want_absmax=True, want_rowsum=True)
    assert torch.equal(acc2, acc)
    assert torch.equal(dmax.view(torch.float32), acc.abs().view(B, 32, -1).amax(dim=2).reshape(-1))
    assert rows.shape == (B, C) and _rel(rows, acc.double().cpu().sum(dim=(2, 3))) < 3e-6


# ---- the small ones -------------------------------------------------------------------------------------------------------------
def test_elementwise_and_reduction_kernels(device):
    from ddpm_ood_amd import train_ops as T

    g = torch.Generator().manual_seed(9)
    x = torch.randn(6, 40, 7, 9, generator=g)
    xd = x.to(device)
    assert _rel(T.row_sum(xd, 6 * 40, 63).view(6, 40), x.double().sum(dim=(2, 3))) < 2e-6
    x4 = torch.randn(37, 256, generator=g)  # (the 16-byte-load form: row length a multiple of 4; rows not a multiple of the 4 waves)
    assert _rel(T.row_sum(x4.to(device), 37, 256), x4.double().sum(1)) < 2e-6
    m = torch.randn(13, 50, generator=g)
    assert _rel(T.col_sum(m.to(device), 13, 50), m.double().sum(0)) < 2e-6
    for rows, cols in ((1000, 1), (300, 700), (256, 3), (5, 64)):  # one column (the loss's partial sums), many rows, several workgroups
        mm = torch.randn(rows, cols, generator=g)
        got = T.col_sum(mm.to(device), rows, cols, alpha=0.25)
        assert _rel(got, 0.25 * mm.double().sum(0)) < 2e-6, (rows, cols)
        assert torch.equal(got, T.col_sum(mm.to(device), rows, cols, alpha=0.25))
    acc = torch.randn(50, generator=g)
    out = T.col_sum(m.to(device), 13, 50, out=acc.clone().to(device), alpha=0.5, accumulate=True)
    assert _rel(out, acc.double() + 0.5 * m.double().sum(0)) < 2e-6
    xs = x.double().requires_grad_(True)
    ys = F.silu(xs)
    dy = torch.randn(x.shape, generator=g)
    (rs,) = torch.autograd.grad(ys, xs, dy.double())
    assert _rel(T.silu(xd), ys.detach()) < 2e-6 and _rel(T.silu_backward(xd, dy.to(device)), rs) < 3e-6
    assert _rel(T.axpby(xd, dy.to(device), 0.5, -2.0), 0.5 * x.double() - 2.0 * dy.double()) < 1e-6
    assert torch.equal(T.axpby(xd, None, 1.0, 0.0).cpu(), x)
    # torch.cat and its split
    a, b = torch.randn(3, 5, 4, 4, generator=g), torch.randn(3, 7, 4, 4, generator=g)
    cat = torch.empty(3, 12, 4, 4, device=device)
    T.chan_copy(a.to(device), cat, 5, 0, 0)
    T.chan_copy(b.to(device), cat, 7, 0, 5)
    assert torch.equal(cat.cpu(), torch.cat([a, b], 1))
    part = torch.empty(3, 7, 4, 4, device=device)
    T.chan_copy(cat, part, 7, 5, 0)
    assert torch.equal(part.cpu(), b)
    T.chan_copy(cat, part, 7, 5, 0, accumulate=True)
    assert torch.equal(part.cpu(), b + b)
    # nearest x2, its adjoint, zero stuffing
    s = torch.randn(2, 3, 5, 6, generator=g)
    up = T.upsample2(s.to(device))
    assert torch.equal(up.cpu(), F.interpolate(s, scale_factor=2.0, mode="nearest"))
    big = torch.randn(2, 3, 10, 12, generator=g)
    assert _rel(T.sumpool2(big.to(device)), 4 * F.avg_pool2d(big.double(), 2)) < 1e-6
    z = T.zero_stuff2(s.to(device)).cpu()
    assert torch.equal(z[:, :, ::2, ::2], s) and float(z.abs().sum()) == float(s.abs().sum())
    # softmax rows and their backward
    sc = torch.randn(37, 70, generator=g) * 3
    sd = sc.double().requires_grad_(True)
    p = torch.softmax(sd, -1)
    dp = torch.randn(sc.shape, generator=g)
    (rsm,) = torch.autograd.grad(p, sd, dp.double())
    ph = T.softmax_rows_(sc.clone().to(device), 37, 70)
    assert _rel(ph, p.detach()) < 2e-6
    assert _rel(T.softmax_backward_rows_(ph, dp.clone().to(device), 37, 70), rsm) < 5e-6
    # MSE and its gradient
    pred, tgt = torch.randn(5, 1, 32, 32, generator=g), torch.randn(5, 1, 32, 32, generator=g)
    pdd = pred.double().requires_grad_(True)
    lr = F.mse_loss(pdd, tgt.double())
    (rg,) = torch.autograd.grad(lr, pdd)
    loss, dpred = T.mse_loss_grad(pred.to(device), tgt.to(device))
    assert abs(float(loss.cpu()) - lr.item()) < 2e-6 * lr.item() and _rel(dpred, rg) < 2e-6
    t = T.fill_(torch.empty(1000, device=device), 2.5)
    assert bool((t == 2.5).all())


def test_adam_kernel_matches_torch_adam_over_five_steps(device):
    from ddpm_ood_amd import train_ops as T

    g = torch.Generator().manual_seed(4)
    p0 = torch.randn(5000, generator=g)
    ref = torch.nn.Parameter(p0.clone().double())
    opt = torch.optim.Adam([ref], lr=2.5e-5)
    p, m, v = p0.clone().to(device), torch.zeros(5000, device=device), torch.zeros(5000, device=device)
    for step in range(1, 6):
        grad = torch.randn(5000, generator=g) * 10 ** float(torch.randint(-4, 2, (1,), generator=g))
        ref.grad = grad.double()
        opt.step()
        T.adam_step_(p, (2.0 * grad).to(device), m, v, 2.5e-5, 0.9, 0.999, 1e-8, step, grad_scale=0.5)
        assert float((p.cpu().double() - ref.detach()).abs().max()) < 6e-7  # (fp32 parameters of magnitude <= 4: ulp 2.4e-7)


def test_randn_is_standard_normal_and_a_pure_function_of_its_counters(device):
    from ddpm_ood_amd import train_ops as T

    a = T.randn((1 << 20,), device, seed=11, stream_id=3)
    assert torch.equal(a, T.randn((1 << 20,), device, seed=11, stream_id=3))
    b, c = T.randn((1 << 20,), device, seed=11, stream_id=4), T.randn((1 << 20,), device, seed=12, stream_id=3)
    x = a.double().cpu()
    assert abs(float(x.mean())) < 4e-3 and abs(float(x.var()) - 1) < 6e-3
    assert abs(float((x ** 3).mean())) < 2e-2 and abs(float((x ** 4).mean()) - 3) < 5e-2
    assert 4.0 < float(x.abs().max()) < 7.0
    for other in (b, c):  # different stream / seed: uncorrelated
        assert abs(float((x * other.double().cpu()).mean())) < 4e-3
    assert abs(float((x[:-1] * x[1:]).mean())) < 4e-3  # neighbours (the Box-Muller pair included)
    odd = T.randn((7, 3), device, seed=1, stream_id=0)
    assert odd.shape == (7, 3) and bool(torch.isfinite(odd).all())
