"""-m gpu: per-kernel parity of the HIP operators (through the C ABI) against plain PyTorch
CPU fp32 ops -- the per-op ground truth named in SURVEY.md 8(c).

Tolerances (fp32): conv / GEMM outputs are compared with
    |hip - ref| <= 2e-5 * (1 + max|ref|)     [k-ordered fp32 FMA chain vs oneDNN blocking; the
    f32 MFMA is exact fp32, so the only difference is summation order, ~1e-7 * sum|a b|]
elementwise kernels with 1e-6 absolute.
"""

import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _close(a, ref, tol=2e-5):
    a = a.detach().cpu().double()
    ref = ref.detach().cpu().double()
    assert a.shape == ref.shape, (a.shape, ref.shape)
    err = (a - ref).abs().max().item()
    bound = tol * (1.0 + ref.abs().max().item())
    assert math.isfinite(err) and err <= bound, f"max err {err:.3e} > {bound:.3e}"


def _ref_conv(x, x2, w, b, gn, act, mode, chan_add, residual):
    xin = x if x2 is None else torch.cat([x, x2], 1)
    if gn is not None:
        gamma, beta, G, eps = gn
        xin = F.group_norm(xin, G, gamma, beta, eps)
    if act:
        xin = F.silu(xin)
    if mode == 2:
        xin = F.interpolate(xin, scale_factor=2.0, mode="nearest")
    k = w.shape[-1] if w.ndim == 4 else 1
    y = F.conv2d(xin, w if w.ndim == 4 else w[:, :, None, None], b, stride=2 if mode == 1 else 1,
                 padding=1 if k == 3 else 0)
    if chan_add is not None:
        y = y + chan_add[:, :, None, None]
    if residual is not None:
        y = y + residual
    return y


CONV_CASES = [
    # B, C1, C2, Cout, H, k, mode, gn, act, chan_add, residual
    (2, 128, 0, 128, 32, 3, 0, True, True, True, False),    # down0 conv1
    (2, 128, 0, 128, 32, 3, 0, True, True, False, True),    # down0 conv2 + identity skip
    (2, 128, 0, 128, 32, 3, 1, False, False, False, False),  # downsampler 32 -> 16
    (2, 128, 0, 256, 16, 3, 0, True, True, True, False),    # down1 conv1
    (3, 256, 0, 256, 16, 3, 1, False, False, False, False),  # downsampler 16 -> 8
    (3, 256, 256, 256, 8, 3, 0, True, True, True, False),   # up0 conv1 on a virtual concat
    (2, 256, 128, 256, 16, 3, 0, True, True, True, False),  # 384 ch: GroupNorm group straddles the seam
    (2, 256, 0, 256, 8, 3, 2, False, False, False, False),   # upsampler 8 -> 16
    (2, 256, 0, 256, 16, 3, 2, False, False, False, False),  # upsampler 16 -> 32
    (2, 256, 128, 128, 32, 1, 0, False, False, False, True),  # 1x1 skip + residual
    (1, 256, 0, 768, 8, 1, 0, True, False, False, False),    # fused QKV (affine, no SiLU), partial tile
    (5, 256, 256, 256, 8, 3, 0, True, True, True, False),   # odd batch: ragged last tile
    (1, 256, 0, 512, 64, 3, 0, True, True, False, False),   # W = 64 (two staging positions)
    (3, 128, 0, 128, 28, 3, 0, True, True, True, False),    # native FashionMNIST 28x28: ragged 4-row tiles (112 / 128 px)
    (3, 128, 0, 128, 28, 3, 1, False, False, False, False),  # 28 -> 14 downsample
    (2, 256, 128, 256, 14, 3, 0, True, True, True, False),  # 14x14: 7-row tiles (98 px)
    (5, 256, 256, 256, 7, 3, 0, True, True, True, False),   # 7x7: 2 images per tile (98 px), odd batch
    (3, 256, 0, 256, 7, 3, 2, False, False, False, False),   # 7 -> 14 upsample (unfolded path)
    (3, 256, 128, 128, 28, 1, 0, False, False, False, True),  # 1x1 skip at 28x28
]


@pytest.mark.parametrize("case", CONV_CASES)
@pytest.mark.parametrize("force_direct", [False, True])
def test_conv(device, case, force_direct):
    from ddpm_ood_amd import ops

    B, C1, C2, Cout, H, k, mode, gn, act, chan, res = case
    if force_direct and H == 64:
        pytest.skip("direct kernel at 64x64x512 is only slow, not different")
    g = torch.Generator().manual_seed(hash(case) & 0xFFFF)
    Cin = C1 + C2
    x = torch.randn(B, C1, H, H, generator=g)
    x2 = torch.randn(B, C2, H, H, generator=g) * 1.5 + 0.3 if C2 else None
    w = torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k)
    b = torch.randn(Cout, generator=g)
    Ho = H // 2 if mode == 1 else (2 * H if mode == 2 else H)
    gamma = torch.randn(Cin, generator=g) * 0.2 + 1
    beta = torch.randn(Cin, generator=g) * 0.2
    chan_add = torch.randn(B, Cout + 64, generator=g) if chan else None
    residual = torch.randn(B, Cout, Ho, Ho, generator=g) if res else None
    ref = _ref_conv(x, x2, w, b, (gamma, beta, 32, 1e-6) if gn else None, act, mode,
                    chan_add[:, 32:32 + Cout] if chan else None, residual)

    d = lambda t: None if t is None else t.to(device)
    gs = gh = None
    if gn:
        gs, gh = ops.gn_scale_shift(d(x), d(gamma), d(beta), 32, 1e-6, x2=d(x2))
    y = ops.conv(d(x), d(w), d(b), x2=d(x2), gscale=gs, gshift=gh, act=int(act), mode=mode, chan_add=d(chan_add),
                 chan_add_offset=32, residual=d(residual), force_direct=force_direct)
    torch.cuda.synchronize()
    _close(y, ref)


@pytest.mark.parametrize("B,C1,C2,Cout,H", [(2, 256, 0, 256, 8), (3, 256, 0, 256, 16), (1, 128, 128, 128, 32),
                                            (5, 256, 0, 128, 4)])
def test_conv_upsample_folded(device, B, C1, C2, Cout, H):
    """nearest-x2 + 3x3 as four 2x2-tap convs over the low-res image (weights pre-summed on the device): same
    result as F.interpolate + F.conv2d up to the rounding of the tap sums."""
    from ddpm_ood_amd import ops

    g = torch.Generator().manual_seed(31)
    Cin = C1 + C2
    x = torch.randn(B, C1, H, H, generator=g)
    x2 = torch.randn(B, C2, H, H, generator=g) if C2 else None
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
    b = torch.randn(Cout, generator=g)
    res = torch.randn(B, Cout, 2 * H, 2 * H, generator=g)
    ref = _ref_conv(x, x2, w, b, None, False, 2, None, res)
    d = lambda t: None if t is None else t.to(device)
    folded = ops.fold_upsample_weight(d(w))
    assert folded is not None
    y = ops.conv(d(x), d(w), d(b), x2=d(x2), mode=2, residual=d(res), folded=folded)
    _close(y, ref)
    y0 = ops.conv(d(x), d(w), d(b), x2=d(x2), mode=2, residual=d(res))  # unfolded path, same op
    _close(y0, ref)


@pytest.mark.parametrize("shape", [(2, 1, 128, 32), (2, 3, 128, 32), (2, 128, 1, 32), (2, 128, 3, 32),
                                   (2, 1, 128, 28), (1, 32, 64, 14), (2, 64, 64, 7)])
def test_conv_direct_shapes(device, shape):
    """conv_in / conv_out and extents with no MFMA tiling (28, 14, 7) take the direct kernel."""
    from ddpm_ood_amd import ops

    B, Cin, Cout, H = shape
    g = torch.Generator().manual_seed(7)
    x = torch.randn(B, Cin, H, H, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
    b = torch.randn(Cout, generator=g)
    y = ops.conv(x.to(device), w.to(device), b.to(device))
    _close(y, F.conv2d(x, w, b, padding=1))
    if H % 2 == 0:
        y = ops.conv(x.to(device), w.to(device), b.to(device), mode=1)
        _close(y, F.conv2d(x, w, b, padding=1, stride=2))


@pytest.mark.parametrize("B,Cin,Cout,act", [(16, 128, 512, 0), (256, 512, 512, 1), (3, 512, 2432, 1),
                                            (130, 512, 256, 0), (4, 32, 96, 1), (1, 128, 512, 1), (1024, 64, 32, 0),
                                            (1025, 512, 128, 1), (33, 160, 64, 1)])
def test_linear(device, B, Cin, Cout, act):
    """<= 1024 rows: linear_skinny_kernel (32 x 32 tiles, K split over four waves, ragged last row tile); more rows or
    channel counts it cannot split: the tiled MFMA kernel."""
    from ddpm_ood_amd import ops

    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, Cin, generator=g)
    w = torch.randn(Cout, Cin, generator=g) / math.sqrt(Cin)
    b = torch.randn(Cout, generator=g)
    y = ops.conv(x.to(device), w.to(device), b.to(device), act=act)
    _close(y, F.linear(F.silu(x) if act else x, w, b))


@pytest.mark.parametrize("B,C1,C2,HW", [(2, 128, 0, 1024), (3, 256, 128, 256), (2, 256, 256, 64), (1, 64, 0, 49),
                                        (2, 768, 0, 4096), (2, 256, 512, 1024), (1, 96, 0, 1026)])
def test_gn_scale_shift(device, B, C1, C2, HW):
    from ddpm_ood_amd import ops

    g = torch.Generator().manual_seed(11)
    x = torch.randn(B, C1, HW, generator=g) * 2 + 0.7
    x2 = torch.randn(B, C2, HW, generator=g) - 1 if C2 else None
    C = C1 + C2
    gamma, beta = torch.randn(C, generator=g), torch.randn(C, generator=g)
    sc, sh = ops.gn_scale_shift(x.to(device), gamma.to(device), beta.to(device), 32, 1e-6,
                                x2=None if x2 is None else x2.to(device))
    xin = x if x2 is None else torch.cat([x, x2], 1)
    ref = F.group_norm(xin, 32, gamma, beta, 1e-6)
    got = xin * sc.cpu()[:, :, None] + sh.cpu()[:, :, None]
    _close(got, ref, tol=5e-6)


@pytest.mark.parametrize("B,heads,N", [(3, 1, 64), (2, 2, 256), (1, 3, 100), (2, 1, 8), (1, 1, 1024)])
@pytest.mark.parametrize("with_res", [True, False])
@pytest.mark.parametrize("use_scratch", [True, False])
def test_attention(device, B, heads, N, with_res, use_scratch, monkeypatch):
    """use_scratch: with caller scratch the multiples of 64 take the register-resident kernel (attention_fa.hip; by default only
    from 1 024 tokens, DDPM_ATTN_FA=2 lifts that), without it (and for every other N) the LDS-exchange kernels of attention.hip."""
    monkeypatch.setenv("DDPM_ATTN_FA", "2")
    from ddpm_ood_amd import ops

    C = 256 * heads
    g = torch.Generator().manual_seed(5)
    qkv = torch.randn(B, 3 * C, N, generator=g)
    qkv[:, :C] *= 1.5  # make the softmax peaky enough to exercise the running max
    res = torch.randn(B, C, N, generator=g) if with_res else None
    scale = 1 / math.sqrt(C / heads)
    out = ops.attention(qkv.to(device), None if res is None else res.to(device), heads, scale, use_scratch=use_scratch)
    q, k, v = (t.reshape(B, heads, 256, N) for t in qkv.split(C, dim=1))
    s = torch.einsum("bhdi,bhdj->bhij", q, k) * scale
    o = torch.einsum("bhij,bhdj->bhdi", s.softmax(-1), v).reshape(B, C, N)
    _close(out, o + res if with_res else o, tol=1e-5)


@pytest.mark.parametrize("B,heads,N,qs,vs", [(3, 1, 64, 1.0, 1.0), (2, 3, 256, 0.05, 20.0), (2, 2, 1024, 3.0, 0.01),
                                             (32, 1, 1024, 1.0, 1.0), (8, 1, 4096, 1.5, 1.0)])
def test_attention_register_resident_vs_float64(device, B, heads, N, qs, vs, monkeypatch):
    """attention_fa.hip (f16 planes of q / k / v from a pre-pass, S^T and O^T in registers, 16-query waves; the last two
    shapes fill the chip with the eight-wave 128-query workgroups, the others run the four-wave form) against a float64
    attention on the first images: the error stays of the order of an fp32 attention's own from flat to peaky softmaxes, and
    the result differs from the LDS-exchange kernel's only in the last bits."""
    monkeypatch.setenv("DDPM_ATTN_FA", "2")  # (by default the small token counts stay on attention.hip: they are faster there)
    from ddpm_ood_amd import ops

    C = 256 * heads
    g = torch.Generator().manual_seed(21 + N)
    qkv = torch.randn(B, 3 * C, N, generator=g)
    qkv[:, :2 * C] *= qs
    qkv[:, 2 * C:] *= vs
    res = torch.randn(B, C, N, generator=g)
    scale = 1 / 16.0
    out = ops.attention(qkv.to(device), res.to(device), heads, scale)
    old = ops.attention(qkv.to(device), res.to(device), heads, scale, use_scratch=False)
    assert not torch.equal(out, old)  # (another kernel ran)
    n = min(B, 2 if N < 4096 else 1)

    def ref(t, r):
        q, k, v = (x.reshape(n, heads, 256, N) for x in t.split(C, dim=1))
        s = torch.einsum("bhdi,bhdj->bhij", q, k) * scale
        return torch.einsum("bhij,bhdj->bhdi", s.softmax(-1), v).reshape(n, C, N) + r

    r64 = ref(qkv[:n].double(), res[:n].double())
    norm = r64.abs().max().item()
    err = (out[:n].cpu().double() - r64).abs().max().item() / norm
    err_old = (old[:n].cpu().double() - r64).abs().max().item() / norm
    err32 = (ref(qkv[:n], res[:n]).double() - r64).abs().max().item() / norm
    print(f"N = {N}: register-resident {err:.2e}, LDS-exchange {err_old:.2e}, fp32 on the host {err32:.2e}")
    assert math.isfinite(err) and err <= max(4 * err32, 2e-6), (err, err32)
    # every image, not only the ones priced in float64: the two kernels agree
    assert (out - old).abs().max().item() <= 1e-5 * (1 + old.abs().max().item())


def test_attention_online_softmax_rescale(device):
    """Force the running-max rescale: the largest logit of every query sits in the LAST key block."""
    from ddpm_ood_amd import ops

    B, C, N = 1, 256, 256
    g = torch.Generator().manual_seed(9)
    qkv = torch.randn(B, 3 * C, N, generator=g) * 0.3
    qkv[:, C:2 * C, 200] = qkv[:, :C, :].mean(-1) * 0 + 3.0 * torch.sign(qkv[:, :C, 5])  # spike key 200
    out = ops.attention(qkv.to(device), None, 1, 1 / 16.0)
    q, k, v = qkv.split(C, dim=1)
    s = torch.einsum("bdi,bdj->bij", q, k) / 16.0
    _close(out, torch.einsum("bij,bdj->bdi", s.softmax(-1), v), tol=1e-5)


def test_attention_register_resident_is_bit_reproducible_under_load(device):
    """Race detector for the LDS-DMA ring of attention_fa.hip: the cfg4 launch (B = 16, n = 4096: two waves of workgroups over the
    chip) repeated back to back with other work in flight must return the same bits every time -- a K / V block multiplied before
    its DMA has landed shows up as run-to-run differences (round 4: hipcc's __syncthreads() does not wait for LDS-DMA writes
    issued in the previous loop iteration; the kernel waits with an explicit s_waitcnt vmcnt(0))."""
    from ddpm_ood_amd import ops

    B, C, N = 16, 256, 4096
    g = torch.Generator(device=device).manual_seed(5)
    qkv = torch.randn(B, 3 * C, N, device=device, generator=g)
    res = torch.randn(B, C, N, device=device, generator=g)
    junk = torch.randn(4096, 4096, device=device)
    first = ops.attention(qkv, res, 1, 1 / 16.0)
    for i in range(6):
        if i & 1:
            junk = junk @ junk * 1e-4  # other kernels between the launches: different arrival times
        again = ops.attention(qkv, res, 1, 1 / 16.0)
        assert torch.equal(first, again), i
    q, k, v = (t.reshape(1, 1, 256, N) for t in qkv[3:4].split(C, dim=1))
    ref = torch.einsum("bhij,bhdj->bhdi", (torch.einsum("bhdi,bhdj->bhij", q, k) / 16.0).softmax(-1), v).reshape(1, C, N) + res[3:4]
    assert (first[3:4] - ref).abs().max().item() < 2e-5


@pytest.mark.parametrize("qs,vs", [(1.0, 1.0), (0.05, 20.0), (3.0, 0.01)])
def test_attention_split_f16_error(device, qs, vs):
    """Both contractions run as split-f16 MFMA products (22-bit products, fp32 accumulate): against a float64
    attention the error is of the order of an fp32 attention's own, over operand scales from flat to peaky softmax."""
    from ddpm_ood_amd import ops

    B, heads, N = 2, 2, 300  # five key blocks, the last one (and the last query block) ragged
    C = 256 * heads
    g = torch.Generator().manual_seed(21)
    qkv = torch.randn(B, 3 * C, N, generator=g)
    qkv[:, :2 * C] *= qs
    qkv[:, 2 * C:] *= vs
    scale = 1 / 16.0
    out = ops.attention(qkv.to(device), None, heads, scale).cpu().double()

    def ref(t):
        q, k, v = (x.reshape(B, heads, 256, N) for x in t.split(C, dim=1))
        s = torch.einsum("bhdi,bhdj->bhij", q, k) * scale
        return torch.einsum("bhij,bhdj->bhdi", s.softmax(-1), v).reshape(B, C, N)

    r64 = ref(qkv.double())
    norm = r64.abs().max().item()
    err = (out - r64).abs().max().item() / norm
    err32 = (ref(qkv).double() - r64).abs().max().item() / norm
    assert math.isfinite(err) and err <= max(4 * err32, 2e-6), (err, err32)


@pytest.mark.parametrize("env", [{"DDPM_ATTN_WAVES": "4"}, {"DDPM_ATTN_F16X3": "0"}, {"DDPM_CONV1X1_F16X3": "0"}])
def test_split_f16_switches_select_working_kernels(device, env):
    """The A/B switches are read once per process: a child process with each of them set runs the attention core and a
    DMA-fed 1x1 (plain and GroupNorm-ed) against torch, so the fallback forms (four-wave split-f16 attention, f32 MFMA
    attention, f32 MFMA 1x1) stay tested while the defaults move on."""
    import os
    import subprocess
    import sys

    code = r"""
import math, torch
from ddpm_ood_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(3)
B, heads, N = 2, 2, 200
C = 256 * heads
qkv = torch.randn(B, 3 * C, N, generator=g)
res = torch.randn(B, C, N, generator=g)
out = ops.attention(qkv.to(dev), res.to(dev), heads, 1 / 16.0).cpu()
q, k, v = (t.reshape(B, heads, 256, N) for t in qkv.split(C, dim=1))
s = torch.einsum("bhdi,bhdj->bhij", q, k) / 16.0
ref = torch.einsum("bhij,bhdj->bhdi", s.softmax(-1), v).reshape(B, C, N) + res
assert (out - ref).abs().max().item() <= 1e-5 * (1 + ref.abs().max().item())
x = torch.randn(160, 256, 16, 16, generator=g)
w = torch.randn(256, 256, 1, 1, generator=g) / 16.0
b = torch.randn(256, generator=g)
gamma, beta = torch.randn(256, generator=g), torch.randn(256, generator=g)
y = ops.conv(x.to(dev), w.to(dev), b.to(dev)).cpu()
ref = torch.nn.functional.conv2d(x, w, b)
assert (y - ref).abs().max().item() <= 2e-5 * (1 + ref.abs().max().item())
gs, gh = ops.gn_scale_shift(x.to(dev), gamma.to(dev), beta.to(dev), 32, 1e-6)
y = ops.conv(x.to(dev), w.to(dev), b.to(dev), gscale=gs, gshift=gh).cpu()
ref = torch.nn.functional.conv2d(torch.nn.functional.group_norm(x, 32, gamma, beta, 1e-6), w, b)
assert (y - ref).abs().max().item() <= 2e-5 * (1 + ref.abs().max().item())
print("ok")
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PYTHONPATH=root, **env), cwd=root,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-2000:]


def test_timestep_embedding_and_known_answer(device):
    from ddpm_ood_amd import ops
    from oracle.unet import get_timestep_embedding

    t = torch.tensor([0, 10, 650, 990], dtype=torch.int64)
    half = 64
    freqs = torch.exp(-math.log(10000) * torch.arange(0, half, dtype=torch.float32) / half)
    out = ops.timestep_embedding(t.to(device), freqs.to(device), 128).cpu()
    assert torch.equal(out[0], torch.cat([torch.ones(64), torch.zeros(64)]))  # t = 0 -> [1..1, 0..0]
    assert (out - get_timestep_embedding(t, 128)).abs().max() < 2e-6


def test_add_noise_plms_clamp_mse(device):
    from ddpm_ood_amd import ops
    from ddpm_ood_amd.scheduler import PNDMScheduler
    import oracle

    g = torch.Generator().manual_seed(2)
    x0 = torch.rand(6, 1, 32, 32, generator=g)
    noise = torch.randn(6, 1, 32, 32, generator=g)
    kw = dict(num_train_timesteps=1000, skip_prk_steps=True, schedule="scaled_linear_beta", beta_start=0.0015,
              beta_end=0.0195)
    hs, osch = PNDMScheduler(**kw), oracle.PNDMScheduler(**kw)
    hs.set_timesteps(100)
    osch.set_timesteps(100)
    assert torch.equal(hs.timesteps, osch.timesteps) and torch.equal(hs.alphas_cumprod, osch.alphas_cumprod)
    t = torch.tensor([650] * 6)
    xh = hs.add_noise(x0.to(device), noise.to(device), t, b_scale=1.7)
    xo = osch.add_noise(x0 * 1.7, noise, t)
    assert (xh.cpu() - xo).abs().max() < 1e-6
    # walk 7 PLMS steps with synthetic eps: exercises every multistep formula incl. the Heun-style start
    xh, xo_ = xh, xo
    for i, step in enumerate(hs.timesteps[hs.timesteps <= 650][:7]):
        eps = torch.randn(6, 1, 32, 32, generator=g)
        xh, _ = hs.step(eps.to(device), step, xh)
        xo_, _ = osch.step(eps, step, xo_)
        assert (xh.cpu() - xo_).abs().max() < 2e-6, f"step {i}"
    rec = (xh * 0.4 + 0.3).contiguous()
    mse = ops.clamp_mse_(x0.to(device), rec, 1.7)
    ref = ((xo_ * 0.4 + 0.3) / 1.7).clamp(0, 1)
    assert (rec.cpu() - ref).abs().max() < 1e-6
    assert (mse.cpu() - torch.square(x0 - ref).mean(dim=(1, 2, 3))).abs().max() < 1e-7


def test_no_cpu_fallback():
    from ddpm_ood_amd import ops

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.conv(torch.zeros(1, 8, 8, 8), torch.zeros(8, 8, 3, 3))


@pytest.mark.parametrize("B,Cin,Cout,D,H", [(1, 128, 256, 8, 8), (2, 256, 256, 4, 16), (1, 256, 128, 6, 12),
                                            (2, 128, 128, 1, 8), (1, 256, 256, 2, 2), (3, 128, 128, 1, 1),
                                            (1, 256, 256, 32, 32)])
def test_conv3d_depth_taps_with_relu_epilogues(device, B, Cin, Cout, D, H):
    """F.conv3d (3x3x3, pad 1) as ONE launch whose chunk stream walks (depth tap, channel group), with the input /
    output ReLU and residual fusions the VQ-VAE residual units use; depth-1 volumes (centre tap only) and 1x1 /
    2x2 planes (capped images per tile) included."""
    from ddpm_ood_amd import ops

    g = torch.Generator().manual_seed(17)
    x = torch.randn(B, Cin, D, H, H, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, 3, generator=g) / math.sqrt(Cin * 27)
    b = torch.randn(Cout, generator=g)
    res = torch.randn(B, Cout, D, H, H, generator=g)
    d = lambda t: t.to(device)
    _close(ops.conv3d(d(x), d(w), d(b)), F.conv3d(x, w, b, padding=1))
    _close(ops.conv3d(d(x), d(w), d(b), act=ops.ACT_RELU, out_act=ops.ACT_RELU, residual=d(res)),
           F.relu(F.conv3d(F.relu(x), w, b, padding=1) + res))


@pytest.mark.parametrize("B,Cin,Cout,D,H", [(1, 256, 256, 16, 16), (2, 128, 128, 3, 32), (1, 256, 256, 8, 64), (1, 64, 128, 1, 16),
                                            (3, 8, 128, 5, 16)])
def test_conv3d_winograd_per_depth_tap(device, B, Cin, Cout, D, H):
    """The VQ-VAE residual-unit convolutions in the Winograd domain: 2-D F(2x2, 3x3) per depth tap, the three taps
    accumulated in the transform domain by ONE persistent launch whose chunk stream walks (depth tap, channel chunk);
    boundary slices read zeros for the tap that falls outside the volume; ReLU / residual epilogue.  Against
    F.conv3d and against the direct MFMA kernel."""
    from ddpm_ood_amd import ops

    g = torch.Generator().manual_seed(41)
    x = torch.randn(B, Cin, D, H, H, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, 3, generator=g) / math.sqrt(Cin * 27)
    b = torch.randn(Cout, generator=g)
    res = torch.randn(B, Cout, D, H, H, generator=g)
    d = lambda t: t.to(device)
    u = ops.pack_wino3d_weight(d(w))
    assert u is not None
    ref = F.relu(F.conv3d(x, w, b, padding=1) + res)
    y = ops.conv3d(d(x), d(w), d(b), residual=d(res), out_act=ops.ACT_RELU, wino=u)
    _close(y, ref)
    _close(ops.conv3d(d(x), d(w), d(b), wino=u), F.conv3d(x, w, b, padding=1))
    # it really is a different kernel from the direct one (rounding differs in the last bits), unless the shape fell back
    yd = ops.conv3d(d(x), d(w), d(b), residual=d(res), out_act=ops.ACT_RELU)
    assert (y - yd).abs().max().item() <= 2e-5 * (1 + ref.abs().max().item())
    if B > 1:  # batches past the kernel's 2 GiB addressing limit are walked in sub-batches: same result, bit for bit
        old = ops.WINO_MAX_TENSOR_BYTES
        try:
            ops.WINO_MAX_TENSOR_BYTES = max(Cin, Cout) * D * H * H * 4  # one volume per launch
            assert torch.equal(ops.conv3d(d(x), d(w), d(b), residual=d(res), out_act=ops.ACT_RELU, wino=u), y)
        finally:
            ops.WINO_MAX_TENSOR_BYTES = old


@pytest.mark.parametrize("B,Cin,Cout,D,H,W", [(1, 256, 256, 16, 16, 16), (1, 256, 256, 64, 64, 64), (2, 128, 256, 8, 32, 16),
                                              (1, 256, 128, 4, 4, 4), (3, 8, 128, 2, 2, 2), (1, 128, 128, 6, 10, 12)])
def test_conv3d_k4s2_matches_torch(device, B, Cin, Cout, D, H, W):
    """VQ-VAE down-convolution: F.conv3d(kernel 4, stride 2, pad 1) (+ ReLU) on the MFMA kernel: 16 in-plane taps per
    chunk, 4 depth taps in the chunk stream (reference: src/trainers/reconstruct.py:124 via the VQ-VAE encoder)."""
    from ddpm_ood_amd import ops

    g = torch.Generator().manual_seed(23)
    x = torch.randn(B, Cin, D, H, W, generator=g)
    w = torch.randn(Cout, Cin, 4, 4, 4, generator=g) / math.sqrt(Cin * 64)
    b = torch.randn(Cout, generator=g)
    d = lambda t: t.to(device)
    y = ops.conv3d(d(x), d(w), d(b), stride=2, out_act=ops.ACT_RELU)
    _close(y, F.relu(F.conv3d(x, w, b, stride=2, padding=1)))


@pytest.mark.parametrize("B,Cin,Cout,D,H,W", [(1, 256, 256, 8, 8, 8), (1, 256, 256, 32, 32, 32), (2, 128, 256, 4, 16, 8),
                                              (1, 256, 128, 2, 2, 2), (3, 8, 128, 1, 1, 1), (1, 128, 128, 3, 5, 6)])
def test_conv_transpose3d_k4s2_matches_torch(device, B, Cin, Cout, D, H, W):
    """VQ-VAE up-convolution: F.conv_transpose3d(kernel 4, stride 2, pad 1) (+ ReLU) as one launch with grid.z = the
    8 output parities, each a 2x2x2-tap convolution (reference: src/trainers/reconstruct.py:166 via the decoder)."""
    from ddpm_ood_amd import ops

    g = torch.Generator().manual_seed(29)
    x = torch.randn(B, Cin, D, H, W, generator=g)
    w = torch.randn(Cin, Cout, 4, 4, 4, generator=g) / math.sqrt(Cin * 8)
    b = torch.randn(Cout, generator=g)
    d = lambda t: t.to(device)
    _close(ops.conv_transpose(d(x), d(w), d(b), out_act=ops.ACT_RELU), F.relu(F.conv_transpose3d(x, w, b, stride=2, padding=1)))
    _close(ops.conv_transpose(d(x), d(w), None), F.conv_transpose3d(x, w, None, stride=2, padding=1))


@pytest.mark.parametrize("B,Cin,Cout,D,H,W,force", [(1, 64, 128, 4, 32, 32, True), (1, 64, 128, 4, 32, 32, False),
                                                      (1, 256, 256, 32, 32, 32, False), (3, 32, 128, 2, 32, 64, True)])
def test_conv_transpose3d_as_eight_parity_convolutions(device, monkeypatch, B, Cin, Cout, D, H, W, force):
    """The round-6 form of the VQ-VAE's largest up-convolution (reference: src/trainers/reconstruct.py:166 via the decoder): eight
    stride-1 3x3x3 convolutions over the input grid, one per output parity, on the split-f16 F(4x4) kernel walking its two
    non-zero depth taps (ddpm_conv_desc.depth_taps), then one interleave pass -- against F.conv_transpose3d.  force: the F(4x4)
    kernel whatever the launch size (a 4-slice volume does not fill the chip: it would take the fallbacks, which ignore the mask
    and multiply the zero tap -- the other branch of this test); the 256 -> 256, 32^3 case is the decoder's own launch."""
    from ddpm_ood_amd import ops

    if force:
        monkeypatch.setenv("DDPM_CONV_WINO44", "2")
    g = torch.Generator().manual_seed(31)
    x = torch.randn(B, Cin, D, H, W, generator=g)
    w = torch.randn(Cin, Cout, 4, 4, 4, generator=g) / math.sqrt(Cin * 8)
    b = torch.randn(Cout, generator=g)
    d = lambda t: t.to(device)
    assert ops.conv_transpose_parity_supported(d(x), d(w))
    if W == 32:  # (four tile rows per item: its 18 staged rows do not fit a 16-row slice)
        assert not ops.conv_transpose_parity_supported(d(x)[:, :, :, :16], d(w))
    pw = ops.pack_convT_parity_weights(d(w))
    # the parity weights are what the closed form says: 3-tap kernels (w[3], w[1], 0) / (0, w[2], w[0]) per axis
    g0 = pw[0][0].cpu()
    assert torch.equal(g0[:, :, 0, 1, 1], w[:, :, 3, 1, 1].t()) and float(g0[:, :, 2].abs().max()) == 0.0
    g7 = pw[7][0].cpu()
    assert torch.equal(g7[:, :, 2, 2, 1], w[:, :, 0, 0, 2].t()) and float(g7[:, :, 0].abs().max()) == 0.0
    prof_on = _prof(True)
    y = ops.conv_transpose_parity(d(x), pw, d(b), out_act=ops.ACT_RELU, sub_batch=2)
    prof = _prof(False, prof_on)
    _close(y, F.relu(F.conv_transpose3d(x, w, b, stride=2, padding=1)))
    _close(ops.conv_transpose_parity(d(x), pw, None), F.conv_transpose3d(x, w, None, stride=2, padding=1))
    if force or (Cin, D) == (256, 32):
        assert "conv3d_wino44h" in prof and prof["conv3d_wino44h"]["launches"] == 8 * ((B + 1) // 2), sorted(prof)
    assert "convT3d_parity_interleave" in prof


def _prof(on, token=None):
    import ctypes
    import json

    from ddpm_ood_amd import _lib

    lib = _lib.load()
    if on:
        lib.ddpm_prof_enable(1)
        return True
    torch.cuda.synchronize()
    lib.ddpm_prof_enable(0)
    buf = ctypes.create_string_buffer(1 << 18)
    n = lib.ddpm_prof_report(buf, len(buf))
    return json.loads(buf.value.decode()) if n > 0 else {}


@pytest.mark.parametrize("B,Cin,Cout,H,W", [(2, 128, 128, 8, 8), (1, 64, 256, 16, 32), (3, 8, 128, 1, 3)])
def test_conv_transpose2d_k4s2_matches_torch(device, B, Cin, Cout, H, W):
    from ddpm_ood_amd import ops

    g = torch.Generator().manual_seed(31)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cin, Cout, 4, 4, generator=g) / math.sqrt(Cin * 4)
    b = torch.randn(Cout, generator=g)
    d = lambda t: t.to(device)
    _close(ops.conv_transpose(d(x), d(w), d(b)), F.conv_transpose2d(x, w, b, stride=2, padding=1))


@pytest.mark.parametrize("B,C,D,H,W", [(1, 256, 32, 32, 32), (2, 16, 4, 6, 10), (1, 32, 2, 2, 2)])
def test_vqvae_edge_layers_match_torch(device, B, C, D, H, W):
    """conv3d_edge.hip: Conv3d 1 -> C (k4 s2 p1, + ReLU) and ConvTranspose3d C -> 1 (k4 s2 p1) against torch."""
    from ddpm_ood_amd import ops

    g = torch.Generator().manual_seed(37)
    d = lambda t: t.to(device)
    x = torch.rand(B, 1, 2 * D, 2 * H, 2 * W, generator=g)
    w = torch.randn(C, 1, 4, 4, 4, generator=g) / 8
    b = torch.randn(C, generator=g)
    _close(ops.conv3d_k4s2_cin1(d(x), d(w), d(b), relu=True), F.relu(F.conv3d(x, w, b, stride=2, padding=1)))
    _close(ops.conv3d_k4s2_cin1(d(x), d(w), None), F.conv3d(x, w, None, stride=2, padding=1))
    z = torch.randn(B, C, D, H, W, generator=g)
    wt = torch.randn(C, 1, 4, 4, 4, generator=g) / math.sqrt(C * 8)
    bt = torch.randn(1, generator=g)
    _close(ops.convT3d_k4s2_cout1(d(z), d(wt), d(bt)), F.conv_transpose3d(z, wt, bt, stride=2, padding=1))


def test_vqvae_readme_shape_all_layers_on_hip(device):
    """README.md:153-158 VQ-VAE (4 x k4-s2 levels, 256 channels, 3 residual units per level, 2 048 codes x 128) on a
    64^3 volume: every layer takes a HIP kernel (no PyTorch-ROCm fallback), encode / decode against the CPU oracle."""
    from oracle.vqvae import VQVAE as OV
    from ddpm_ood_amd import vqvae as pv

    cfg = dict(spatial_dims=3, in_channels=1, out_channels=1, num_channels=(256,) * 4, num_res_layers=3,
               num_res_channels=(256,) * 4, downsample_parameters=((2, 4, 1, 1),) * 4,
               upsample_parameters=((2, 4, 1, 1, 0),) * 4, num_embeddings=2048, embedding_dim=128)
    torch.manual_seed(1)
    o = OV(**cfg).eval()
    with torch.no_grad():
        o.quantizer.quantizer.embedding.weight.mul_(3.0)
    p = pv.VQVAE(**cfg)
    p.load_state_dict(o.state_dict())
    p = p.to(device).eval()
    x = torch.rand(1, 1, 64, 64, 64, generator=torch.Generator().manual_seed(2))
    kinds = []
    pv._FALLBACK_WARNED.clear()  # other tests' toy shapes (16 / 32 channels) do fall back, in the same process
    orig = pv._Convolution._hip_kind
    pv._Convolution._hip_kind = lambda self, t: (kinds.append(orig(self, t)), kinds[-1])[1]
    try:
        with torch.no_grad():
            zo = o.encode_stage_2_inputs(x)
            zp = p.encode_stage_2_inputs(x.to(device))
            _close(zp, zo, tol=1e-5)
            _close(p.decode_stage_2_outputs(zp), o.decode_stage_2_outputs(zo), tol=2e-5)
    finally:
        pv._Convolution._hip_kind = orig
    assert None not in kinds and not pv._FALLBACK_WARNED, (kinds, pv._FALLBACK_WARNED)
    assert kinds.count("conv_cin1") == 1 and kinds.count("convT_cout1") == 1 and kinds.count("convT") == 3


def test_vqvae_residual_units_on_hip_match_torch(device):
    """The product VQ-VAE runs its 3x3x3 residual units on the MFMA kernel when the channel counts allow it;
    result vs the CPU oracle restatement."""
    from oracle.vqvae import VQVAE as OV
    from ddpm_ood_amd.vqvae import VQVAE as PV

    cfg = dict(spatial_dims=3, in_channels=1, out_channels=1, num_channels=(128, 128), num_res_layers=2,
               num_res_channels=(128, 128), downsample_parameters=((2, 4, 1, 1), (2, 4, 1, 1)),
               upsample_parameters=((2, 4, 1, 1, 0), (2, 4, 1, 1, 0)), num_embeddings=32, embedding_dim=128)
    torch.manual_seed(1)
    o = OV(**cfg).eval()
    with torch.no_grad():
        o.quantizer.quantizer.embedding.weight.mul_(3.0)
    p = PV(**cfg)
    p.load_state_dict(o.state_dict())
    p = p.to(device).eval()
    x = torch.rand(1, 1, 16, 16, 16, generator=torch.Generator().manual_seed(2))
    with torch.no_grad():
        zo = o.encode_stage_2_inputs(x)
        zp = p.encode_stage_2_inputs(x.to(device))
        _close(zp, zo, tol=1e-5)
        _close(p.decode_stage_2_outputs(zp), o.decode_stage_2_outputs(zo), tol=2e-5)


def test_vqvae_2d_mfma_friendly_layers_match_oracle(device):
    """a 2-D VQ-VAE whose stride-1 3x3 layers have an MFMA tiling (128 channels) runs them -- residual units included --
    on the UNet's convolution kernels instead of the one-thread-per-output generic kernel; the k4-s2 down / up layers stay
    generic.  Encode + decode vs the CPU oracle."""
    from oracle.vqvae import VQVAE as OV
    from ddpm_ood_amd.vqvae import VQVAE as PV, _Convolution

    cfg = dict(spatial_dims=2, in_channels=1, out_channels=1, num_channels=(128, 128), num_res_layers=2,
               num_res_channels=(128, 128), downsample_parameters=((2, 4, 1, 1), (2, 4, 1, 1)),
               upsample_parameters=((2, 4, 1, 1, 0), (2, 4, 1, 1, 0)), num_embeddings=32, embedding_dim=128)
    torch.manual_seed(1)
    o = OV(**cfg).eval()
    with torch.no_grad():
        o.quantizer.quantizer.embedding.weight.mul_(3.0)
    p = PV(**cfg)
    p.load_state_dict(o.state_dict())
    p = p.to(device).eval()
    x = torch.rand(3, 1, 32, 32, generator=torch.Generator().manual_seed(2))
    kinds = {m._hip_kind(torch.zeros(1, m.conv.weight.shape[1], 8, 8, device=device)) for m in p.modules() if isinstance(m, _Convolution)}
    assert "conv2d" in kinds
    with torch.no_grad():
        zo = o.encode_stage_2_inputs(x)
        zp = p.encode_stage_2_inputs(x.to(device))
        _close(zp, zo, tol=1e-5)
        _close(p.decode_stage_2_outputs(zp), o.decode_stage_2_outputs(zo), tol=2e-5)


WINO_CASES = [
    # B, C1, C2, Cout, H, gn, chan_add, residual
    (2, 128, 0, 128, 32, True, True, False),
    (3, 128, 0, 128, 32, True, False, True),
    (2, 256, 128, 128, 32, True, True, False),
    (3, 256, 0, 256, 16, True, False, True),
    (5, 256, 256, 256, 8, True, True, False),     # 4 images per workgroup, ragged last workgroup
    (1, 128, 0, 64, 64, False, False, False),     # W = 64: two tile rows per workgroup, Cout = 64
    (2, 8, 0, 64, 4, False, False, True),         # 4x4 images: 16 per workgroup
]


@pytest.mark.parametrize("case", WINO_CASES)
def test_conv_winograd(device, case):
    """3x3 stride-1 conv as Winograd F(2x2, 3x3) on the fp32 MFMA pipe vs F.conv2d; fp32 rounding of the transforms
    costs ~2x the direct kernel's error (tolerance 4e-5 * (1 + max|ref|))."""
    from ddpm_ood_amd import ops

    B, C1, C2, Cout, H, gn, chan, res = case
    g = torch.Generator().manual_seed(hash(case) & 0xFFFF)
    Cin = C1 + C2
    x = torch.randn(B, C1, H, H, generator=g)
    x2 = torch.randn(B, C2, H, H, generator=g) * 1.5 + 0.3 if C2 else None
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
    b = torch.randn(Cout, generator=g)
    G = 32 if Cin % 32 == 0 else 8
    gamma = torch.randn(Cin, generator=g) * 0.2 + 1
    beta = torch.randn(Cin, generator=g) * 0.2
    chan_add = torch.randn(B, Cout + 64, generator=g) if chan else None
    residual = torch.randn(B, Cout, H, H, generator=g) if res else None
    ref = _ref_conv(x, x2, w, b, (gamma, beta, G, 1e-6) if gn else None, gn, 0,
                    chan_add[:, 32:32 + Cout] if chan else None, residual)
    d = lambda t: None if t is None else t.to(device)
    gs = gh = None
    if gn:
        gs, gh = ops.gn_scale_shift(d(x), d(gamma), d(beta), G, 1e-6, x2=d(x2))
    wino = ops.pack_wino_weight(d(w))
    assert wino is not None
    y = ops.conv(d(x), d(w), d(b), x2=d(x2), gscale=gs, gshift=gh, act=int(gn), chan_add=d(chan_add),
                 chan_add_offset=32, residual=d(residual), wino=wino)
    torch.cuda.synchronize()
    _close(y, ref, tol=4e-5)
    y_direct = ops.conv(d(x), d(w), d(b), x2=d(x2), gscale=gs, gshift=gh, act=int(gn), chan_add=d(chan_add),
                        chan_add_offset=32, residual=d(residual)) if Cout % 128 == 0 else None
    if y_direct is not None:
        assert (y - y_direct).abs().max().item() < 4e-5 * (1 + ref.abs().max().item())


@pytest.mark.parametrize("B,Cin,Cout,D,H", [(1, 256, 256, 8, 64), (2, 128, 128, 3, 32), (1, 64, 128, 1, 32), (3, 8, 128, 5, 32)])
def test_conv3d_winograd_f4x4_per_depth_tap(device, B, Cin, Cout, D, H, monkeypatch):
    """The VQ-VAE residual-unit convolutions as 2-D F(4x4, 3x3) per depth tap (conv_wino44.hip, 3-D form): slices of >= 32
    tiles, chunk stream = (depth tap, channel chunk), zeros for the taps outside the volume (D = 1: centre tap only), ReLU /
    residual epilogue.  Against F.conv3d at the F(4x4) tolerance, and really a different kernel from the F(2x2) one."""
    monkeypatch.setenv("DDPM_CONV_WINO44", "2")  # also for launches smaller than the chip
    from ddpm_ood_amd import ops

    g = torch.Generator().manual_seed(43)
    x = torch.randn(B, Cin, D, H, H, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, 3, generator=g) / math.sqrt(Cin * 27)
    b = torch.randn(Cout, generator=g)
    res = torch.randn(B, Cout, D, H, H, generator=g)
    d = lambda t: t.to(device)
    u, v = ops.pack_wino3d_weight(d(w)), ops.pack_wino44_3d_weight(d(w))
    assert v is not None and v.numel() == 3 * 36 * Cout * Cin
    ref = F.relu(F.conv3d(x, w, b, padding=1) + res)
    y = ops.conv3d(d(x), d(w), d(b), residual=d(res), out_act=ops.ACT_RELU, wino=u, wino44=v)
    y2 = ops.conv3d(d(x), d(w), d(b), residual=d(res), out_act=ops.ACT_RELU, wino=u)
    torch.cuda.synchronize()
    assert not torch.equal(y, y2)  # the F(4x4) kernel really ran
    for got, want in ((y, ref), (ops.conv3d(d(x), d(w), d(b), wino=u, wino44=v), F.conv3d(x, w, b, padding=1))):
        err = got.cpu() - want
        assert err.abs().max().item() < 2e-4 * (1 + want.abs().max().item()), err.abs().max().item()
        assert err.pow(2).mean().sqrt().item() < 1e-5 * (1 + want.pow(2).mean().sqrt().item())
    assert torch.equal(y, ops.conv3d(d(x), d(w), d(b), residual=d(res), out_act=ops.ACT_RELU, wino=u, wino44=v))
    if B > 1:  # sub-batch walk past the 2 GiB addressing limit: same result, bit for bit
        old = ops.WINO_MAX_TENSOR_BYTES
        try:
            ops.WINO_MAX_TENSOR_BYTES = max(Cin, Cout) * D * H * H * 4  # one volume per launch
            assert torch.equal(ops.conv3d(d(x), d(w), d(b), residual=d(res), out_act=ops.ACT_RELU, wino=u, wino44=v), y)
        finally:
            ops.WINO_MAX_TENSOR_BYTES = old


WINO44_CASES = [
    # B, C1, C2, Cout, H, gn, chan_add, residual
    (2, 64, 0, 64, 32, False, False, False),      # plain: 2 parts per image, one cout tile
    (3, 128, 0, 128, 32, True, True, True),
    (2, 256, 128, 128, 32, True, True, False),    # virtual concat
    (5, 256, 0, 256, 16, True, False, True),      # two images per item, ragged last item
    (19, 128, 128, 64, 8, True, True, True),      # eight images per item, ragged
    (1, 64, 0, 64, 64, True, False, False),       # W = 64: two tile rows per item, 8 parts
    (300, 128, 0, 128, 16, True, True, True),     # 2 x 150 items: several items per persistent workgroup
    (70, 64, 64, 128, 32, True, True, True),      # 2 x 2 x 70 = 280 items
]


@pytest.mark.parametrize("case", WINO44_CASES)
def test_conv_winograd_f4x4(device, case, monkeypatch):
    """3x3 stride-1 conv as Winograd F(4x4, 3x3) (conv_wino44.hip) vs F.conv2d.  The 6x6 transforms (constants up to 8,
    weights down to 1/24) cost about 10x the direct kernel's rounding error: tolerance 2e-4 * (1 + max|ref|) on
    single values, 1e-5 on the rms (the trajectory-level effect on Z-scores is 5e-6)."""
    monkeypatch.setenv("DDPM_CONV_WINO44", "2")  # also for launches smaller than the chip (read once per process)
    from ddpm_ood_amd import ops

    B, C1, C2, Cout, H, gn, chan, res = case
    g = torch.Generator().manual_seed(hash(case) & 0xFFFF)
    Cin = C1 + C2
    x = torch.randn(B, C1, H, H, generator=g)
    x2 = torch.randn(B, C2, H, H, generator=g) * 1.5 + 0.3 if C2 else None
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
    b = torch.randn(Cout, generator=g)
    gamma = torch.randn(Cin, generator=g) * 0.2 + 1
    beta = torch.randn(Cin, generator=g) * 0.2
    chan_add = torch.randn(B, Cout + 64, generator=g) if chan else None
    residual = torch.randn(B, Cout, H, H, generator=g) if res else None
    ref = _ref_conv(x, x2, w, b, (gamma, beta, 32, 1e-6) if gn else None, gn, 0,
                    chan_add[:, 32:32 + Cout] if chan else None, residual)
    d = lambda t: None if t is None else t.to(device)
    gs = gh = None
    if gn:
        gs, gh = ops.gn_scale_shift(d(x), d(gamma), d(beta), 32, 1e-6, x2=d(x2))
    w44 = ops.pack_wino44_weight(d(w))
    assert w44 is not None and w44.numel() == 36 * Cout * Cin
    kw = dict(x2=d(x2), gscale=gs, gshift=gh, act=int(gn), chan_add=d(chan_add), chan_add_offset=32, residual=d(residual))
    y = ops.conv(d(x), d(w), d(b), wino44=w44, **kw)
    y_f2 = ops.conv(d(x), d(w), d(b), wino=ops.pack_wino_weight(d(w)), **kw)
    torch.cuda.synchronize()
    assert not torch.equal(y, y_f2)  # the F(4x4) kernel really ran
    err = (y.cpu() - ref)
    assert err.abs().max().item() < 2e-4 * (1 + ref.abs().max().item()), err.abs().max().item()
    assert err.pow(2).mean().sqrt().item() < 1e-5 * (1 + ref.pow(2).mean().sqrt().item())
    assert torch.equal(y, ops.conv(d(x), d(w), d(b), wino44=w44, **kw))  # no dependence on leftover LDS state


def test_conv_mfma_split_k(device, monkeypatch):
    """Launches of the direct MFMA kernel with fewer workgroups than half the CUs (the 4^3 / 2^3 levels of the latent
    UNet, small-batch 8x8 layers): every tile goes to up to 8 workgroups, each walking a share of the chunk stream;
    partial slabs are added in a fixed order with bias / temb / residual by the reduce pass.  Against the torch op, and
    against the unsplit launch (DDPM_CONV_SPLITK=0) within fp32 summation-order noise."""
    from ddpm_ood_amd import ops

    g = torch.Generator().manual_seed(77)
    d = lambda t: None if t is None else t.to(device)

    def both(f):
        monkeypatch.delenv("DDPM_CONV_SPLITK", raising=False)
        y = f()
        y_again = f()
        monkeypatch.setenv("DDPM_CONV_SPLITK", "0")
        y0 = f()
        monkeypatch.delenv("DDPM_CONV_SPLITK")
        torch.cuda.synchronize()
        assert torch.equal(y, y_again)      # fixed slab order
        assert not torch.equal(y, y0)       # the split launch really ran
        return y, y0

    # fused QKV 1x1 with GroupNorm prologue, B = 4 at 8x8: 4 tiles x 6 cout tiles
    x = torch.randn(4, 256, 8, 8, generator=g)
    w = torch.randn(768, 256, 1, 1, generator=g) / 16
    b = torch.randn(768, generator=g)
    gamma, beta = torch.randn(256, generator=g) * 0.2 + 1, torch.randn(256, generator=g) * 0.2
    gs, gh = ops.gn_scale_shift(d(x), d(gamma), d(beta), 32, 1e-6)
    ref = F.conv2d(F.group_norm(x, 32, gamma, beta, 1e-6), w, b)
    y, y0 = both(lambda: ops.conv(d(x), d(w), d(b), gscale=gs, gshift=gh))
    _close(y, ref)
    _close(y, y0.cpu(), tol=2e-6)
    # 3x3 stride 2 with residual-free epilogue + temb, B = 16, 16x16 -> 8x8
    x = torch.randn(16, 256, 16, 16, generator=g)
    w = torch.randn(256, 256, 3, 3, generator=g) / 48
    temb = torch.randn(16, 256, generator=g)
    ref = F.conv2d(x, w, b[:256], stride=2, padding=1) + temb[:, :, None, None]
    y, y0 = both(lambda: ops.conv(d(x), d(w), d(b[:256]), mode=ops.CONV_STRIDE2, chan_add=d(temb)))
    _close(y, ref)
    # conv3d 3x3x3 with residual over 4^3 volumes (the latent UNet's second level)
    x = torch.randn(2, 256, 4, 4, 4, generator=g)
    w = torch.randn(256, 256, 3, 3, 3, generator=g) / math.sqrt(256 * 27)
    res = torch.randn(2, 256, 4, 4, 4, generator=g)
    ref = F.conv3d(x, w, b[:256], padding=1) + res
    y, y0 = both(lambda: ops.conv3d(d(x), d(w), d(b[:256]), residual=d(res)))
    _close(y, ref)
    _close(y, y0.cpu(), tol=2e-6)


WINO44_SPLIT_CASES = [
    # B, C1, C2, Cout, H, residual: fewer F(4x4) items than the 256 CUs -> the channel stream of an item is split over 2 / 4
    # workgroups (partial slabs in scratch + the reduce pass)
    (256, 256, 0, 256, 8, True),      # the 8x8 level of cfg2 at B = 256: 128 items, S = 2
    (251, 256, 256, 256, 8, False),   # virtual concat, ragged last item
    (128, 128, 0, 256, 8, True),      # 64 items, S = 4
    (127, 128, 128, 128, 16, True),   # 2 x 64 items (ragged), S = 2
]


@pytest.mark.parametrize("case", WINO44_SPLIT_CASES)
def test_conv_winograd_f4x4_channel_split(device, case, monkeypatch):
    """F(4x4) launches with fewer items than CUs: S workgroups share an item's channel stream; slabs added in a fixed
    order by the reduce pass together with bias / temb / residual.  Same tolerance as the unsplit kernel."""
    monkeypatch.delenv("DDPM_CONV_WINO44", raising=False)
    monkeypatch.delenv("DDPM_WINO44_SPLIT", raising=False)
    from ddpm_ood_amd import ops

    B, C1, C2, Cout, H, res = case
    g = torch.Generator().manual_seed(hash(case) & 0xFFFF)
    Cin = C1 + C2
    x = torch.randn(B, C1, H, H, generator=g)
    x2 = torch.randn(B, C2, H, H, generator=g) * 1.5 + 0.3 if C2 else None
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
    b = torch.randn(Cout, generator=g)
    gamma = torch.randn(Cin, generator=g) * 0.2 + 1
    beta = torch.randn(Cin, generator=g) * 0.2
    chan_add = torch.randn(B, Cout + 64, generator=g)
    residual = torch.randn(B, Cout, H, H, generator=g) if res else None
    ref = _ref_conv(x, x2, w, b, (gamma, beta, 32, 1e-6), True, 0, chan_add[:, 32:32 + Cout], residual)
    d = lambda t: None if t is None else t.to(device)
    gs, gh = ops.gn_scale_shift(d(x), d(gamma), d(beta), 32, 1e-6, x2=d(x2))
    w44, w22 = ops.pack_wino44_weight(d(w)), ops.pack_wino_weight(d(w))
    kw = dict(x2=d(x2), gscale=gs, gshift=gh, act=1, chan_add=d(chan_add), chan_add_offset=32, residual=d(residual))
    y = ops.conv(d(x), d(w), d(b), wino44=w44, wino=w22, **kw)
    monkeypatch.setenv("DDPM_WINO44_SPLIT", "0")  # no split: the launch falls to the F(2x2) kernel
    y_f2 = ops.conv(d(x), d(w), d(b), wino44=w44, wino=w22, **kw)
    monkeypatch.delenv("DDPM_WINO44_SPLIT")
    torch.cuda.synchronize()
    if torch.cuda.get_device_properties(0).multi_processor_count == 256:
        assert not torch.equal(y, y_f2)  # the split F(4x4) launch really ran
    err = (y.cpu() - ref)
    assert err.abs().max().item() < 2e-4 * (1 + ref.abs().max().item()), err.abs().max().item()
    assert err.pow(2).mean().sqrt().item() < 1e-5 * (1 + ref.pow(2).mean().sqrt().item())
    assert torch.equal(y, ops.conv(d(x), d(w), d(b), wino44=w44, wino=w22, **kw))  # fixed slab order: bit-reproducible


WINO_STREAM_CASES = [
    # B, C1, C2, Cout, H: more work items than the chip has CUs, so every persistent workgroup streams several
    # items back to back (the B <= 5 cases above run one item per workgroup)
    (40, 128, 0, 128, 32),      # 2 cout tiles x 4 parts x 40 images = 320 items
    (70, 128, 64, 256, 16),     # 4 x 1 x 70 = 280 items, virtual concat
    (301, 64, 0, 256, 8),       # 4 images per item, ragged last item: 4 x 76 = 304 items
]


def test_conv_winograd_small_batch_channel_split(device):
    """Small batches (BASELINE configs[0]: first_n = 16): fewer items than CUs, so the channel stream of each item is
    split over 2 or 4 workgroups (partial outputs in scratch slabs, added in a fixed order with bias / temb / residual
    by a second pass).  Same result as the unsplit launch up to fp32 rounding; bit-reproducible run to run."""
    import ctypes as C

    from ddpm_ood_amd import _lib, ops

    g = torch.Generator().manual_seed(5)
    B, Cin, Cout, H = 4, 256, 256, 8
    x = torch.randn(B, Cin, H, H, generator=g).to(device)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)).to(device)
    b, temb, res = (torch.randn(s_, generator=g).to(device) for s_ in ((Cout,), (B, Cout), (B, Cout, H, H)))
    gs, gh = ops.gn_scale_shift(x, torch.ones(Cin, device=device), torch.zeros(Cin, device=device), 32, 1e-6)
    pk, wn = ops.pack_conv_weight(w), ops.pack_wino_weight(w)
    kw = dict(gscale=gs, gshift=gh, act=ops.ACT_SILU, packed=pk, wino=wn, chan_add=temb, residual=res)
    y = ops.conv(x, w, b, **kw)
    assert torch.equal(y, ops.conv(x, w, b, **kw))                      # fixed reduction order
    ref = F.conv2d(F.silu(F.group_norm(x.cpu(), 32, eps=1e-6)), w.cpu(), b.cpu(), padding=1) + temb.cpu()[:, :, None, None] + res.cpu()
    _close(y, ref)
    # the launch really was split: the library asks for scratch for this shape, and for none at a chip-filling batch
    lib = _lib.load()
    d = _lib.ConvDesc()
    d.in1, d.C1, d.w_packed, d.w_wino, d.out = x.data_ptr(), Cin, pk.data_ptr(), wn.data_ptr(), y.data_ptr()
    d.gscale, d.gshift = gs.data_ptr(), gh.data_ptr()
    d.B, d.Cout, d.Hi, d.Wi, d.Ho, d.Wo, d.ksize, d.act = B, Cout, H, H, H, H, 3, ops.ACT_SILU
    need = lib.ddpm_conv_scratch_floats(C.byref(d))
    assert need >= 2 * y.numel() and need % y.numel() == 0  # the largest of the kernels' needs (MFMA split-K: up to 8 slabs)
    d.B = 256
    assert lib.ddpm_conv_scratch_floats(C.byref(d)) == 0


@pytest.mark.parametrize("case", WINO_STREAM_CASES)
def test_conv_winograd_item_stream(device, case):
    """Persistent Winograd kernel with several items per workgroup vs the CPU reference and the direct MFMA kernel."""
    from ddpm_ood_amd import ops

    B, C1, C2, Cout, H = case
    g = torch.Generator().manual_seed(B * 7 + H)
    Cin = C1 + C2
    x = torch.randn(B, C1, H, H, generator=g)
    x2 = torch.randn(B, C2, H, H, generator=g) * 1.5 + 0.3 if C2 else None
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
    b = torch.randn(Cout, generator=g)
    gamma = torch.randn(Cin, generator=g) * 0.2 + 1
    beta = torch.randn(Cin, generator=g) * 0.2
    chan_add = torch.randn(B, Cout, generator=g)
    residual = torch.randn(B, Cout, H, H, generator=g)
    ref = _ref_conv(x, x2, w, b, (gamma, beta, 32, 1e-6), True, 0, chan_add, residual)
    d = lambda t: None if t is None else t.to(device)
    gs, gh = ops.gn_scale_shift(d(x), d(gamma), d(beta), 32, 1e-6, x2=d(x2))
    wino = ops.pack_wino_weight(d(w))
    kw = dict(x2=d(x2), gscale=gs, gshift=gh, act=1, chan_add=d(chan_add), residual=d(residual))
    y = ops.conv(d(x), d(w), d(b), wino=wino, **kw)
    y_direct = ops.conv(d(x), d(w), d(b), **kw)
    torch.cuda.synchronize()
    _close(y, ref, tol=4e-5)
    assert (y - y_direct).abs().max().item() < 4e-5 * (1 + ref.abs().max().item())


def test_conv_winograd_full_batch_matches_direct(device):
    """BASELINE batch (256 images, 8 items per workgroup): Winograd vs the direct MFMA kernel on the device, and a
    second launch into the same buffers gives bit-identical output (no dependence on leftover LDS state)."""
    from ddpm_ood_amd import ops

    g = torch.Generator().manual_seed(11)
    B, C, H = 256, 128, 32
    x = torch.randn(B, C, H, H, generator=g).to(device)
    w = (torch.randn(C, C, 3, 3, generator=g) / math.sqrt(C * 9)).to(device)
    b = torch.randn(C, generator=g).to(device)
    gs, gh = ops.gn_scale_shift(x, torch.ones(C, device=device), torch.zeros(C, device=device), 32, 1e-6)
    wino = ops.pack_wino_weight(w)
    y = ops.conv(x, w, b, gscale=gs, gshift=gh, act=1, residual=x, wino=wino)
    y2 = ops.conv(x, w, b, gscale=gs, gshift=gh, act=1, residual=x, wino=wino)
    y_direct = ops.conv(x, w, b, gscale=gs, gshift=gh, act=1, residual=x)
    torch.cuda.synchronize()
    assert torch.equal(y, y2)
    assert (y - y_direct).abs().max().item() < 4e-5 * (1 + y_direct.abs().max().item())


# ---- LPIPS-AlexNet on the HIP kernels --------------------------------------------------

@pytest.mark.parametrize("case", [
    # N, Cx, Cin, H, W, Cout, k, stride, pad, affine
    (5, 1, 3, 32, 32, 64, 11, 4, 2, True),      # AlexNet conv1 on a grey image (1 -> 3 broadcast)
    (3, 3, 3, 64, 48, 64, 11, 4, 2, True),      # conv1, RGB, non-square
    (4, 64, 64, 7, 7, 192, 5, 1, 2, False),     # conv2
    (2, 10, 10, 5, 6, 7, 3, 1, 1, False),       # odd everything (Cout not a multiple of 4)
])
def test_lpips_conv(device, case):
    from ddpm_ood_amd import ops

    N, Cx, Cin, H, W, Cout, k, stride, pad, affine = case
    g = torch.Generator().manual_seed(sum(case[:9]))
    x = torch.rand(N, Cx, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k)
    b = torch.randn(Cout, generator=g)
    a = torch.rand(Cin, generator=g) + 0.5 if affine else None
    s = torch.randn(Cin, generator=g) if affine else None
    xin = x.expand(N, Cin, H, W) if Cx == 1 else x
    if affine:
        xin = xin * a[None, :, None, None] + s[None, :, None, None]
    ref = F.relu(F.conv2d(xin, w, b, stride=stride, padding=pad))
    d = lambda t: None if t is None else t.to(device)
    y = ops.lpips_conv(d(x), d(w), d(b), stride, pad, True, d(a), d(s))
    torch.cuda.synchronize()
    _close(y, ref, tol=2e-5)


@pytest.mark.parametrize("case", [
    # N, Cin, H, W, Cout, k
    (6, 64, 15, 15, 192, 5),      # AlexNet layer 2 over a 128 x 128 slice: 8 pixel tiles, 6 cout blocks = 12 waves
    (3, 64, 15, 11, 192, 5),      # non-square, ragged last pixel-tile group
    (2, 32, 23, 23, 64, 5),       # 529 pixels: 17 pixel tiles, 2 x 5 units over 12 waves
    (2, 8, 9, 9, 32, 3),          # the 3 x 3 instantiation
])
def test_lpips_conv_mfma(device, case):
    """The same-padded stride-1 LPIPS layers on the fp32 MFMA pipe (lpips.hip) vs F.conv2d."""
    from ddpm_ood_amd import ops

    N, Cin, H, W, Cout, k = case
    assert ops.lpips_conv_mfma_supported(Cin, H, W, Cout, k)
    assert not ops.lpips_conv_mfma_supported(64, 7, 7, 192, 5)  # tiny maps stay on the scalar kernel
    g = torch.Generator().manual_seed(sum(case))
    x = torch.rand(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k)
    b = torch.randn(Cout, generator=g)
    ref = F.conv2d(x, w, b, padding=k // 2)
    d = lambda t: t.to(device)
    packed = ops.lpips_pack_conv_weight(d(w))
    y = ops.lpips_conv_mfma(d(x), packed, d(b), Cout, k, relu=False)
    yr = ops.lpips_conv_mfma(d(x), packed, d(b), Cout, k, relu=True)
    torch.cuda.synchronize()
    _close(y, ref, tol=2e-5)
    _close(yr, F.relu(ref), tol=2e-5)
    _close(y, ops.lpips_conv(d(x), d(w), d(b), 1, k // 2, False).cpu(), tol=2e-5)  # and vs the scalar kernel


def test_maxpool3s2(device):
    from ddpm_ood_amd import ops

    g = torch.Generator().manual_seed(5)
    for shape in ((3, 64, 7, 7), (2, 5, 15, 12), (1, 2, 3, 3)):
        x = torch.randn(*shape, generator=g)
        y = ops.maxpool3s2(x.to(device))
        torch.cuda.synchronize()
        assert torch.equal(y.cpu(), F.max_pool2d(x, 3, 2))


def test_lpips_layer(device):
    from ddpm_ood_amd import ops

    g = torch.Generator().manual_seed(6)
    val, ref = None, 0
    for C, H, W in ((64, 7, 7), (192, 3, 3), (256, 1, 1), (32, 31, 31)):
        f0, f1 = torch.rand(6, C, H, W, generator=g), torch.rand(6, C, H, W, generator=g)
        f0[1] = 0  # an all-zero feature vector: the 1e-10 guard
        lin = torch.rand(C, generator=g) / C
        n0 = f0 / (torch.sqrt(torch.sum(f0 ** 2, dim=1, keepdim=True)) + 1e-10)
        n1 = f1 / (torch.sqrt(torch.sum(f1 ** 2, dim=1, keepdim=True)) + 1e-10)
        ref = ref + (((n0 - n1) ** 2) * lin[None, :, None, None]).sum(1).mean([1, 2])
        val = ops.lpips_layer(f0.to(device), f1.to(device), lin.to(device), val)
    torch.cuda.synchronize()
    _close(val, ref, tol=1e-5)


@pytest.mark.parametrize("shape", [(7, 1, 32, 32), (3, 3, 64, 64), (2, 1, 128, 96), (70, 1, 128, 128), (66, 1, 32, 32)])
def test_lpips_score_vs_oracle(device, shape):
    """Whole LPIPS(normalize=True) on the HIP kernels vs the CPU oracle with the same weights.  128-pixel maps take the MFMA
    form of the 5x5 layer; grey batches of >= 64 images (2.5-D LPIPS over volumes) the folded one-channel first layer."""
    import oracle
    from ddpm_ood_amd.perceptual import LPIPS

    g = torch.Generator().manual_seed(shape[2])
    x, y = torch.rand(*shape, generator=g), torch.rand(*shape, generator=g)
    hip = LPIPS(seed=3)
    ref = oracle.LPIPSAlex()
    ref.load_state_dict(hip.state_dict())
    want = ref(x, y, normalize=True)
    got = hip.to(device)(x.to(device), y.to(device), normalize=True)
    torch.cuda.synchronize()
    assert got.shape == want.shape
    _close(got, want, tol=2e-5)


def test_perceptual_loss_3d_last_view_only(device, monkeypatch):
    """2.5-D LPIPS over a volume: the reference's loop overwrites `loss` per view (quirk Q7), so only the last view is
    returned.  The default path computes that view alone; DDPM_LPIPS_ALL_VIEWS=1 (all three, as the reference literally
    does) returns the same bits; both match the oracle, which runs all three."""
    import oracle
    from ddpm_ood_amd.perceptual import PerceptualLoss

    g = torch.Generator().manual_seed(8)
    y, p = torch.rand(1, 1, 64, 48, 40, generator=g), torch.rand(1, 1, 64, 48, 40, generator=g)
    hip = PerceptualLoss(dimensions=3, lpips_kwargs={"seed": 5})
    ref = oracle.PerceptualLoss(dimensions=3)
    ref.perceptual_function.load_state_dict(hip.perceptual_function.state_dict())
    want = ref(y, p)
    hip = hip.to(device)
    monkeypatch.delenv("DDPM_LPIPS_ALL_VIEWS", raising=False)
    got = hip(y.to(device), p.to(device))
    monkeypatch.setenv("DDPM_LPIPS_ALL_VIEWS", "1")
    got_all = hip(y.to(device), p.to(device))
    torch.cuda.synchronize()
    assert torch.equal(got, got_all)
    _close(got.reshape(()), want.reshape(()), tol=2e-5)


CONV1X1_DMA_CASES = [
    # B, C1, C2, Cout, H, bias, chan_add, residual -- large enough for the DMA-fed 1x1 kernel (>= 384 workgroups)
    (96, 256, 128, 128, 32, True, False, True),     # up-path skip connection with virtual concat
    (256, 256, 256, 256, 16, True, True, False),    # one image per 256-pixel tile, two cout tiles
    (1537, 128, 0, 128, 8, False, False, True),     # four images per tile, ragged last tile (384.25 tiles)
]
CONV1X1_DMA_GN_CASES = [
    # B, C, Cout, H -- fused q / k / v projection behind the attention block's GroupNorm (split-f16 kernel's prologue)
    (512, 256, 768, 8),      # small UNet, 8x8 level: four images per 256-pixel tile
    (67, 128, 384, 16),      # one image per tile, odd batch
    (31, 128, 128, 32),      # four tiles per image; 124 workgroups: below the threshold, stays on conv_mfma (same answer)
    (100, 256, 256, 32),     # 400 x 2 workgroups
]


@pytest.mark.parametrize("case", CONV1X1_DMA_CASES)
def test_conv1x1_dma(device, case):
    """Plain 1x1 convolution with both operands fed by LDS-DMA vs F.conv2d and vs the register-staged kernel."""
    import os
    from ddpm_ood_amd import ops

    B, C1, C2, Cout, H, has_bias, chan, res = case
    g = torch.Generator().manual_seed(B + C1 + H)
    Cin = C1 + C2
    x = torch.randn(B, C1, H, H, generator=g)
    x2 = torch.randn(B, C2, H, H, generator=g) if C2 else None
    w = torch.randn(Cout, Cin, 1, 1, generator=g) / math.sqrt(Cin)
    b = torch.randn(Cout, generator=g) if has_bias else None
    chan_add = torch.randn(B, Cout, generator=g) if chan else None
    residual = torch.randn(B, Cout, H, H, generator=g) if res else None
    ref = _ref_conv(x, x2, w, b, None, False, 0, chan_add, residual)
    d = lambda t: None if t is None else t.to(device)
    y = ops.conv(d(x), d(w), d(b), x2=d(x2), chan_add=d(chan_add), residual=d(residual))
    torch.cuda.synchronize()
    _close(y, ref, tol=2e-5)
    # weights pre-split at pack time (what the UNet engine hands over): same products in the same order -> the same bits
    wh = ops.pack_conv1x1_h_weight(d(w))
    if wh is not None:
        y2 = ops.conv(d(x), d(w), d(b), x2=d(x2), chan_add=d(chan_add), residual=d(residual), wino44h=wh)
        assert torch.equal(y, y2)


@pytest.mark.parametrize("case", CONV1X1_DMA_GN_CASES)
def test_conv1x1_dma_groupnorm_prologue(device, case):
    """1x1 convolution of a GroupNorm-ed input (no activation): the scale / shift pairs ride the DMA ring and are
    applied where the split-f16 kernel splits its operands; vs F.group_norm + F.conv2d."""
    from ddpm_ood_amd import ops

    B, C, Cout, H = case
    g = torch.Generator().manual_seed(B + C + H)
    x = torch.randn(B, C, H, H, generator=g) * 1.7 + 0.3
    w = torch.randn(Cout, C, 1, 1, generator=g) / math.sqrt(C)
    b = torch.randn(Cout, generator=g)
    gamma, beta = torch.randn(C, generator=g), torch.randn(C, generator=g)
    ref = _ref_conv(x, None, w, b, (gamma, beta, 32, 1e-6), False, 0, None, None)
    d = lambda t: t.to(device)
    gs, gh = ops.gn_scale_shift(d(x), d(gamma), d(beta), 32, 1e-6)
    y = ops.conv(d(x), d(w), d(b), gscale=gs, gshift=gh)
    torch.cuda.synchronize()
    _close(y, ref, tol=2e-5)
    y2 = ops.conv(d(x), d(w), d(b), gscale=gs, gshift=gh, wino44h=ops.pack_conv1x1_h_weight(d(w)))  # pre-split weights
    assert torch.equal(y, y2)


@pytest.mark.parametrize("xs,ws,floor", [(1.0, 1.0, 1e-6), (1e-2, 30.0, 1e-6), (50.0, 1e-2, 1e-6), (1e-3, 1.0, 4e-6),
                                         (300.0, 3e-4, 4e-6)])
def test_conv1x1_dma_split_f16_error(device, xs, ws, floor):
    """The split-f16 MFMA loop of the DMA-fed 1x1 (x w ~ xh wh + xh wl + xl wh, fp32 accumulate, low halves scaled
    into the f16 normal range) against a float64 convolution over several decades of operand scale: inside the
    documented range its error stays within a few fp32 roundings of the output (2^-22 per product), the same order
    as an fp32 convolution's own error; one decade below it (last two cases) it degrades gracefully."""
    from ddpm_ood_amd import ops

    B, C1, C2, Cout, H = 96, 256, 128, 128, 32
    g = torch.Generator().manual_seed(77)
    x = torch.randn(B, C1, H, H, generator=g) * xs
    x2 = torch.randn(B, C2, H, H, generator=g) * xs
    w = torch.randn(Cout, C1 + C2, 1, 1, generator=g) / math.sqrt(C1 + C2) * ws
    n = 8  # float64 reference on a slice of the batch
    ref64 = F.conv2d(torch.cat([x[:n], x2[:n]], 1).double(), w.double())
    ref32 = F.conv2d(torch.cat([x[:n], x2[:n]], 1), w)
    y = ops.conv(x.to(device), w.to(device), None, x2=x2.to(device))
    torch.cuda.synchronize()
    scale = ref64.abs().max().item()
    err = (y[:n].cpu().double() - ref64).abs().max().item() / scale
    err32 = (ref32.double() - ref64).abs().max().item() / scale
    assert math.isfinite(err) and err <= max(4 * err32, floor), (err, err32)


UP_WINO_CASES = [
    # B, Cin, Cout, low-res H, bias/residual
    (2, 256, 256, 16, False),     # the 16 -> 32 Upsample of the small UNet
    (3, 256, 256, 8, True),       # 8 -> 16: one image per item
    (5, 64, 128, 4, True),        # 4x4 low-res: four images per item, ragged last item
    (1, 128, 64, 32, False),      # 32 -> 64: two tile rows per item
    (44, 128, 128, 16, True),     # 2 x 4 x 44 = 352 items: several per persistent workgroup
]


@pytest.mark.parametrize("case", UP_WINO_CASES)
def test_conv_upsample_winograd(device, case):
    """nearest x2 + 3x3 conv on the 9 surviving Winograd positions vs F.interpolate + F.conv2d and vs the folded kernel."""
    from ddpm_ood_amd import ops

    B, Cin, Cout, H, extra = case
    g = torch.Generator().manual_seed(B * 13 + H)
    x = torch.randn(B, Cin, H, H, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
    b = torch.randn(Cout, generator=g)
    residual = torch.randn(B, Cout, 2 * H, 2 * H, generator=g) if extra else None
    ref = _ref_conv(x, None, w, b, None, False, 2, None, residual)
    d = lambda t: None if t is None else t.to(device)
    wino = ops.pack_wino_weight(d(w))
    assert wino is not None
    y = ops.conv(d(x), d(w), d(b), mode=ops.CONV_UPSAMPLE2, residual=d(residual), wino=wino)
    torch.cuda.synchronize()
    _close(y, ref, tol=4e-5)
    if Cout % 128 == 0:
        y_folded = ops.conv(d(x), d(w), d(b), mode=ops.CONV_UPSAMPLE2, residual=d(residual),
                            folded=ops.fold_upsample_weight(d(w)))
        assert (y - y_folded).abs().max().item() < 4e-5 * (1 + ref.abs().max().item())


@pytest.mark.parametrize("case", UP_WINO_CASES)
def test_conv_upsample_winograd_emits_groupnorm_statistics(device, case):
    """desc.stats_out of the Upsample kernel: per-(image, cout, slice) {mean, M2} of the tensor it wrote."""
    from ddpm_ood_amd import ops

    B, Cin, Cout, H, extra = case
    g = torch.Generator().manual_seed(B * 13 + H)
    x = torch.randn(B, Cin, H, H, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
    b = torch.randn(Cout, generator=g)
    residual = torch.randn(B, Cout, 2 * H, 2 * H, generator=g) if extra else None
    d = lambda t: None if t is None else t.to(device)
    wino = ops.pack_wino_weight(d(w))
    y_plain = ops.conv(d(x), d(w), d(b), mode=ops.CONV_UPSAMPLE2, residual=d(residual), wino=wino)
    y, st = ops.conv(d(x), d(w), d(b), mode=ops.CONV_UPSAMPLE2, residual=d(residual), wino=wino, want_stats=True)
    assert torch.equal(y, y_plain)
    parts = {16: 8, 8: 2, 4: 1, 32: None}[H]
    if parts is None:  # 32 slices per image: more than a slab holds
        assert st is None
        return
    assert st is not None and tuple(st.shape) == (B, Cout, parts, 2)
    yd = y.double().cpu().view(B, Cout, parts, -1)
    mean = yd.mean(-1)
    m2 = (yd - mean[..., None]).pow(2).sum(-1)
    st = st.cpu().double()
    assert (st[..., 0] - mean).abs().max().item() <= 2e-6 * (1 + mean.abs().max().item() + (m2 / yd.shape[-1]).sqrt().max().item())
    assert ((st[..., 1] - m2).abs() / (m2 + 1e-3 * m2.mean())).max().item() <= 2e-5


@pytest.mark.parametrize("case", [(5, 1, 128, 32, 4), (3, 3, 128, 32, 4), (70, 1, 64, 16, 1), (2, 1, 128, 64, None), (4, 1, 128, 28, None)])
def test_conv_in_emits_groupnorm_statistics(device, case):
    """desc.stats_out of the small-cin kernel (conv_in): per-(image, cout, 256-pixel slice) {mean, M2} of the tensor it wrote;
    sizes with more than eight slices per image or a ragged last workgroup report 0 parts."""
    from ddpm_ood_amd import ops

    B, Cin, Cout, H, parts = case
    g = torch.Generator().manual_seed(B * 7 + H)
    x = torch.randn(B, Cin, H, H, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
    b = torch.randn(Cout, generator=g) * 3  # (means far from zero: the M2 must not be a difference of large numbers)
    d = lambda t: t.to(device)
    y_plain = ops.conv(d(x), d(w), d(b))
    y, st = ops.conv(d(x), d(w), d(b), want_stats=True)
    assert torch.equal(y, y_plain)
    ref = torch.nn.functional.conv2d(x, w, b, padding=1)
    assert (y.cpu() - ref).abs().max().item() <= 2e-5 * (1 + ref.abs().max().item())
    if parts is None:
        assert st is None
        return
    assert st is not None and tuple(st.shape) == (B, Cout, parts, 2)
    yd = y.double().cpu().view(B, Cout, parts, -1)
    mean = yd.mean(-1)
    m2 = (yd - mean[..., None]).pow(2).sum(-1)
    st2 = ops.conv(d(x), d(w), d(b), want_stats=True)[1]
    assert torch.equal(st, st2)  # fixed order: bit-reproducible
    st = st.cpu().double()
    assert (st[..., 0] - mean).abs().max().item() <= 2e-6 * (1 + mean.abs().max().item() + (m2 / yd.shape[-1]).sqrt().max().item())
    assert ((st[..., 1] - m2).abs() / (m2 + 1e-3 * m2.mean())).max().item() <= 2e-5


@pytest.mark.parametrize("case", [(3, 128, (8, 8, 8), 2048), (2, 32, (4, 5, 6), 100), (5, 8, (7, 9), 17)])
def test_vq_nearest(device, case):
    """VQ-VAE quantiser on the HIP kernel vs the oracle's formula (argmin of the expanded squared distance)."""
    import oracle.vqvae as ov
    from ddpm_ood_amd import ops

    B, D, spatial, K = case
    g = torch.Generator().manual_seed(K)
    x = torch.randn(B, D, *spatial, generator=g)
    ref = ov._EMAQuantizer(K, D)
    with torch.no_grad():
        ref.embedding.weight.copy_(torch.randn(K, D, generator=g))
    idx_ref = ref.quantize(x)
    out_ref = ref(x)
    idx, out = ops.vq_nearest(x.to(device), ref.embedding.weight.detach().to(device))
    torch.cuda.synchronize()
    assert idx.shape == idx_ref.shape
    # a different summation order may flip an exact near-tie: allow it only if the two codes are equally close
    diff = (idx.cpu() != idx_ref)
    if diff.any():
        flat = x.movedim(1, -1).reshape(-1, D).double()
        e = ref.embedding.weight.double()
        da = ((flat - e[idx.cpu().reshape(-1)]) ** 2).sum(1)
        db = ((flat - e[idx_ref.reshape(-1)]) ** 2).sum(1)
        assert torch.allclose(da, db, rtol=1e-5)
        assert diff.float().mean().item() < 1e-3
    same = ~diff
    assert torch.equal(out.cpu().movedim(1, -1)[same], out_ref.movedim(1, -1)[same])


@pytest.mark.parametrize("dims,cin,cout,k,stride,pad,transposed", [
    (3, 5, 7, 3, 1, 1, False), (3, 1, 6, 4, 2, 1, False), (3, 6, 3, 4, 2, 1, True), (2, 3, 8, 3, 1, 1, False),
    (2, 8, 5, 4, 2, 1, True), (3, 16, 32, 3, 1, 1, False), (2, 4, 4, 1, 1, 0, False)])
def test_convnd_generic_vs_torch(device, dims, cin, cout, k, stride, pad, transposed):
    """ddpm_convnd_generic_f32: the always-available convolution behind the MFMA kernels (any channel counts, 2-D / 3-D,
    stride 1 / 2, transposed), with the residual + ReLU epilogue -- what the VQ-VAE layers without an MFMA tiling run on
    instead of PyTorch-ROCm ops (nn.Conv3d / nn.ConvTranspose3d of the reference's VQVAE, reconstruct.py:124,166)."""
    from ddpm_ood_amd import ops

    g = torch.Generator().manual_seed(dims * 100 + cin * 10 + cout)
    sp = (6, 8, 10)[3 - dims:]
    x = torch.randn(2, cin, *sp, generator=g)
    w = torch.randn(*((cin, cout) if transposed else (cout, cin)), *([k] * dims), generator=g) / math.sqrt(cin * k ** dims)
    b = torch.randn(cout, generator=g)
    f = {(2, False): F.conv2d, (3, False): F.conv3d, (2, True): F.conv_transpose2d, (3, True): F.conv_transpose3d}[dims, transposed]
    ref = f(x, w, b, stride=stride, padding=pad)
    res = torch.randn(ref.shape, generator=g)
    y = ops.convnd_generic(x.to(device), w.to(device), b.to(device), stride=stride, padding=pad, transposed=transposed)
    assert y.shape == ref.shape
    _close(y, ref, tol=2e-6)
    y2 = ops.convnd_generic(x.to(device), w.to(device), b.to(device), stride=stride, padding=pad, transposed=transposed,
                            residual=res.to(device), relu=True)
    _close(y2, F.relu(ref + res), tol=2e-6)


def test_vq_nearest_generic_embedding_dim(device):
    """An embedding size without a register-resident instantiation (12) takes the generic quantiser kernel: same indices as
    a brute-force search, straight-through output x + (e_k - x)."""
    from ddpm_ood_amd import ops

    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 12, 4, 4, 4, generator=g)
    e = torch.randn(32, 12, generator=g) * 2
    idx, out = ops.vq_nearest(x.to(device), e.to(device))
    flat = x.movedim(1, -1).reshape(-1, 12)
    brute = ((flat[:, None, :] - e[None]) ** 2).sum(-1).argmin(1)
    assert torch.equal(idx.cpu().reshape(-1), brute)
    assert torch.allclose(out.cpu().movedim(1, -1).reshape(-1, 12), e[brute], atol=1e-6)


# ---- GroupNorm from per-channel statistics slabs (ABI 7) ------------------------------------------------------------------

@pytest.mark.parametrize("shape", [(3, 128, 0, 1024), (5, 256, 128, 256), (2, 256, 256, 64), (2, 64, 0, 4096), (3, 32, 0, 100)])
def test_gn_finalize_from_channel_stats_matches_group_norm(device, shape):
    """channel_stats + gn_finalize == the reading kernel == F.group_norm's statistics, incl. a virtual concat whose groups
    straddle the seam (384 = 256 + 128 channels, 12 per group) and slabs with different slice counts per source."""
    from ddpm_ood_amd import ops

    B, C1, C2, HW = shape
    g = torch.Generator().manual_seed(B * 1000 + C1 + HW)
    x = torch.randn(B, C1, HW, generator=g) * 1.7 + 0.6
    x2 = torch.randn(B, C2, HW, generator=g) * 0.4 - 2.0 if C2 else None
    C = C1 + C2
    gamma, beta = torch.randn(C, generator=g) * 0.3 + 1, torch.randn(C, generator=g) * 0.3
    d = lambda t: None if t is None else t.to(device)
    st1 = ops.channel_stats(d(x))
    xd = x.double()
    assert (st1[:, :, 0, 0].cpu().double() - xd.mean(-1)).abs().max().item() < 2e-6
    m2 = (xd - xd.mean(-1, keepdim=True)).pow(2).sum(-1)
    assert ((st1[:, :, 0, 1].cpu().double() - m2).abs() / m2).max().item() < 1e-5
    st2 = ops.channel_stats(d(x2)) if C2 else None
    if C2 and HW % 4 == 0:  # the second source as a 4-slice slab (what a producer with 4 items per image leaves behind)
        x2d = x2.double().view(B, C2, 4, HW // 4)
        mu = x2d.mean(-1)
        st2 = torch.stack([mu, (x2d - mu[..., None]).pow(2).sum(-1)], -1).float().to(device).contiguous()
    sc, sh = ops.gn_finalize(st1, d(gamma), d(beta), 32, 1e-6, HW, stats2=st2)
    sc_r, sh_r = ops.gn_scale_shift(d(x), d(gamma), d(beta), 32, 1e-6, x2=d(x2))
    xin = (x if x2 is None else torch.cat([x, x2], 1)).double()
    ref = torch.nn.functional.group_norm(xin, 32, gamma.double(), beta.double(), 1e-6)
    got = xin * sc.cpu().double()[:, :, None] + sh.cpu().double()[:, :, None]
    old = xin * sc_r.cpu().double()[:, :, None] + sh_r.cpu().double()[:, :, None]
    e_new, e_old = (got - ref).abs().max().item(), (old - ref).abs().max().item()
    assert e_new <= 5e-6 and e_new <= 2 * e_old + 1e-6, (e_new, e_old)
    sc2, sh2 = ops.gn_finalize(st1, d(gamma), d(beta), 32, 1e-6, HW, stats2=st2)
    assert torch.equal(sc, sc2) and torch.equal(sh, sh2)


# ---- Downsample convolution on the f16 MFMA (conv_s2h.hip) -----------------------------------------------------------------

S2H_CASES = [
    # B, Cin, Cout, input H
    (3, 128, 128, 32),    # the 32 -> 16 Downsample of the small UNet: two tiles per image
    (5, 256, 256, 16),    # 16 -> 8: two images per tile, ragged
    (2, 64, 128, 64),     # 64 -> 32: four rows per tile
    (11, 64, 64, 8),      # 8 -> 4: eight images per tile, ragged
    (300, 8, 64, 16),     # one chunk, many workgroups
    (513, 16, 128, 32),   # 1 026 tiles: the four-tile form (one workgroup per CU and more), ragged last group
    (4, 24, 64, 16),      # an odd number of chunks (three)
    (130, 40, 256, 64),   # 1 040 tiles x two cout-tile pairs in the four-tile form, five chunks
]


@pytest.mark.parametrize("case", S2H_CASES)
def test_conv_stride2_split_f16_vs_conv2d(device, case):
    """F.conv2d(x, w, b, stride=2, padding=1) (generative's Downsample, /root/reference/src/trainers/reconstruct.py:151-153)
    as a direct convolution with split-f16 operands: against float64, no worse than the fp32 MFMA kernel it replaces, and
    bit-reproducible; DDPM_DOWN_S2H=0 is covered by the child-process switch test."""
    from ddpm_ood_amd import ops

    B, Cin, Cout, H = case
    g = torch.Generator().manual_seed(B + Cin + H)
    x = torch.randn(B, Cin, H, H, generator=g) * 1.3 + 0.2
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
    b = torch.randn(Cout, generator=g)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), b.double(), stride=2, padding=1)
    d = lambda t: t.to(device)
    ws = ops.pack_conv_s2h_weight(d(w))
    assert ws is not None and ws.numel() == Cout * Cin * 18 + 64
    y = ops.conv(d(x), d(w), d(b), mode=ops.CONV_STRIDE2, wino44h=ws)
    y32 = ops.conv(d(x), d(w), d(b), mode=ops.CONV_STRIDE2)
    torch.cuda.synchronize()
    assert not torch.equal(y, y32)  # the split-f16 kernel really ran
    scale = ref.abs().max().item()
    e_new = (y.cpu().double() - ref).abs().max().item() / scale
    e_old = (y32.cpu().double() - ref).abs().max().item() / scale
    assert math.isfinite(e_new) and e_new <= max(2 * e_old, 2e-6), (e_new, e_old)
    assert torch.equal(y, ops.conv(d(x), d(w), d(b), mode=ops.CONV_STRIDE2, wino44h=ws))


@pytest.mark.parametrize("case", S2H_CASES)
def test_conv_stride2_emits_groupnorm_statistics(device, case):
    """desc.stats_out of the Downsample kernels (both forms): per-(image, cout, tile-of-the-image) {mean, M2} of the tensor they
    wrote; images smaller than a wave's 32 pixels report 0 parts."""
    from ddpm_ood_amd import ops

    B, Cin, Cout, H = case
    g = torch.Generator().manual_seed(B + Cin + H)
    x = torch.randn(B, Cin, H, H, generator=g) * 1.3 + 0.2
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
    b = torch.randn(Cout, generator=g) * 2
    d = lambda t: t.to(device)
    ws = ops.pack_conv_s2h_weight(d(w))
    y_plain = ops.conv(d(x), d(w), d(b), mode=ops.CONV_STRIDE2, wino44h=ws)
    y, st = ops.conv(d(x), d(w), d(b), mode=ops.CONV_STRIDE2, wino44h=ws, want_stats=True)
    assert torch.equal(y, y_plain)
    Ho = H // 2
    parts = {16: 2, 8: 1, 32: 8, 4: None}[Ho]  # 128-pixel tiles: 2 per 16x16 image, 8 per 32x32; whole 8x8 images; 4x4: half a wave
    if parts is None:
        assert st is None
        return
    assert st is not None and tuple(st.shape) == (B, Cout, parts, 2)
    assert torch.equal(st, ops.conv(d(x), d(w), d(b), mode=ops.CONV_STRIDE2, wino44h=ws, want_stats=True)[1])
    yd = y.double().cpu().view(B, Cout, parts, -1)
    mean = yd.mean(-1)
    m2 = (yd - mean[..., None]).pow(2).sum(-1)
    st = st.cpu().double()
    assert (st[..., 0] - mean).abs().max().item() <= 2e-6 * (1 + mean.abs().max().item() + (m2 / yd.shape[-1]).sqrt().max().item())
    assert ((st[..., 1] - m2).abs() / (m2 + 1e-3 * m2.mean())).max().item() <= 2e-5


@pytest.mark.parametrize("case", [(16, 128, 128, 32), (16, 256, 256, 16), (5, 256, 256, 16), (3, 64, 128, 32), (1, 64, 128, 16)])
def test_conv_stride2_small_launch_vs_conv2d(device, case, monkeypatch):
    """conv_d3s.hip's stride-2 form: the Downsample convolutions of a forward over a few images (32 -> 16, 16 -> 8), channel
    slices of 32 + the reduce pass, against F.conv2d(stride=2, padding=1) in float64; ragged last tile; bit-reproducible."""
    monkeypatch.setenv("DDPM_CONV_D3S", "2")
    from ddpm_ood_amd import ops

    B, Cin, Cout, H = case
    g = torch.Generator().manual_seed(B + Cin + H)
    x = torch.randn(B, Cin, H, H, generator=g) * 1.3 + 0.2
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
    b = torch.randn(Cout, generator=g)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), b.double(), stride=2, padding=1)
    d = lambda t: t.to(device)
    planes = ops.pack_conv_d3h_weight(d(w))
    y = ops.conv(d(x), d(w), d(b), mode=ops.CONV_STRIDE2, d3h=planes)
    y32 = ops.conv(d(x), d(w), d(b), mode=ops.CONV_STRIDE2)
    torch.cuda.synchronize()
    assert not torch.equal(y, y32)  # the one-shot kernel ran
    scale = ref.abs().max().item()
    err = (y.cpu().double() - ref).abs().max().item() / scale
    assert math.isfinite(err) and err < 3e-6, err
    assert torch.equal(y, ops.conv(d(x), d(w), d(b), mode=ops.CONV_STRIDE2, d3h=planes))


def test_conv_stride2_split_f16_operand_range(device):
    """Operand scales over several decades (the per-layer weight scale 2^su and the 2^3 input scale keep the lo halves normal)."""
    from ddpm_ood_amd import ops

    g = torch.Generator().manual_seed(5)
    for sx, sw in ((1e-3, 1.0), (30.0, 1.0), (1.0, 1e-2), (1.0, 20.0)):
        x = torch.randn(4, 64, 16, 16, generator=g) * sx
        w = torch.randn(64, 64, 3, 3, generator=g) / 24 * sw
        ref = torch.nn.functional.conv2d(x.double(), w.double(), None, stride=2, padding=1)
        y = ops.conv(x.to(device), w.to(device), None, mode=ops.CONV_STRIDE2, wino44h=ops.pack_conv_s2h_weight(w.to(device)))
        err = (y.cpu().double() - ref).abs().max().item() / ref.abs().max().item()
        assert err <= (4e-6 if sx < 1e-2 else 2e-6), (sx, sw, err)  # (inputs below 2^-9: their lo halves go subnormal)


D3S_CASES = [
    # B, C1, C2, Cout, H, gn, chan_add, residual
    (16, 256, 0, 256, 8, True, True, False),      # the benchmark's 8x8 layers at first_n = 16: 8 x 4 x 8 workgroups
    (5, 256, 256, 256, 8, True, False, True),     # virtual concat, ragged last tile (one image), 16 slices
    (3, 128, 0, 128, 16, True, True, True),       # 16x16: two tiles per image
    (4, 256, 128, 256, 16, True, False, False),   # concat seam inside a slice's chunks
    (2, 64, 0, 128, 8, False, False, True),       # no prologue (raw input, 2^0)
    (1, 32, 32, 128, 16, False, True, False),
    (2, 128, 0, 128, 32, True, True, True),       # 32x32: eight rows of an image per workgroup, four tiles per image
    (3, 256, 128, 128, 32, True, False, False),   # twelve slices
]


@pytest.mark.parametrize("case", D3S_CASES)
def test_conv_small_launch_split_f16_vs_conv2d(device, case, monkeypatch):
    """conv_d3s.hip: the one-shot 3x3 convolution of launches far smaller than the chip (cfg1: 8x8 / 16x16 levels at 16 images):
    channel slices of 32 into scratch slabs + the fixed-order reduce pass, against F.conv2d in float64; bit-reproducible."""
    monkeypatch.setenv("DDPM_CONV_D3S", "2")
    from ddpm_ood_amd import ops

    B, C1, C2, Cout, H, gn, chan, res = case
    g = torch.Generator().manual_seed(B * 13 + C1 + H)
    Cin = C1 + C2
    x = torch.randn(B, C1, H, H, generator=g) * 1.3 + 0.2
    x2 = torch.randn(B, C2, H, H, generator=g) if C2 else None
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
    b = torch.randn(Cout, generator=g)
    gamma, beta = (torch.randn(Cin, generator=g), torch.randn(Cin, generator=g)) if gn else (None, None)
    chan_add = torch.randn(B, Cout, generator=g) if chan else None
    residual = torch.randn(B, Cout, H, H, generator=g) if res else None
    xin = (x if x2 is None else torch.cat([x, x2], 1)).double()
    if gn:
        xin = F.silu(F.group_norm(xin, 32, gamma.double(), beta.double(), 1e-6))
    ref = F.conv2d(xin, w.double(), b.double(), padding=1)
    if chan:
        ref = ref + chan_add.double()[:, :, None, None]
    if res:
        ref = ref + residual.double()
    d = lambda t: None if t is None else t.to(device)
    gs = gh = None
    if gn:
        gs, gh = ops.gn_scale_shift(d(x), d(gamma), d(beta), 32, 1e-6, x2=d(x2))
    kw = dict(x2=d(x2), gscale=gs, gshift=gh, act=ops.ACT_SILU if gn else ops.ACT_NONE, chan_add=d(chan_add), residual=d(residual))
    planes = ops.pack_conv_d3h_weight(d(w))
    y = ops.conv(d(x), d(w), d(b), d3h=planes, **kw)
    y0 = ops.conv(d(x), d(w), d(b), **kw)  # without the planes: the fp32 MFMA kernels
    torch.cuda.synchronize()
    assert not torch.equal(y, y0)  # (the small-launch kernel ran)
    assert torch.equal(y, ops.conv(d(x), d(w), d(b), d3h=planes, **kw))
    scale = ref.abs().max().item()
    err = (y.cpu().double() - ref).abs().max().item() / scale
    err0 = (y0.cpu().double() - ref).abs().max().item() / scale
    print(f"{case}: small-launch split-f16 {err:.2e}, fp32 MFMA kernel {err0:.2e} (max relative to max |y|)")
    assert math.isfinite(err) and err < 3e-6, (err, err0)


@pytest.mark.parametrize("case", [(16, 256, 256, 8), (3, 128, 256, 8), (2, 64, 128, 16), (5, 256, 128, 16)])
def test_conv_upsample_small_launch_vs_interpolate_conv2d(device, case, monkeypatch):
    """conv_d3s.hip's upsample-on-load form: F.interpolate(nearest, x2) + conv3x3 (generative's Upsample,
    /root/reference/src/trainers/reconstruct.py:151-153) of a forward over a few images, against float64; bit-reproducible."""
    monkeypatch.setenv("DDPM_CONV_D3S", "2")
    from ddpm_ood_amd import ops

    B, Cin, Cout, H = case
    g = torch.Generator().manual_seed(B + Cin + H)
    x = torch.randn(B, Cin, H, H, generator=g) * 2.1 - 0.4
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
    b = torch.randn(Cout, generator=g)
    ref = F.conv2d(F.interpolate(x.double(), scale_factor=2.0, mode="nearest"), w.double(), b.double(), padding=1)
    d = lambda t: t.to(device)
    planes = ops.pack_conv_d3h_weight(d(w))
    y = ops.conv(d(x), d(w), d(b), mode=ops.CONV_UPSAMPLE2, d3h=planes)
    y0 = ops.conv(d(x), d(w), d(b), mode=ops.CONV_UPSAMPLE2)
    torch.cuda.synchronize()
    assert y.shape == (B, Cout, 2 * H, 2 * H) and not torch.equal(y, y0)  # the one-shot kernel ran
    scale = ref.abs().max().item()
    err = (y.cpu().double() - ref).abs().max().item() / scale
    assert math.isfinite(err) and err < 3e-6, err
    assert torch.equal(y, ops.conv(d(x), d(w), d(b), mode=ops.CONV_UPSAMPLE2, d3h=planes))


D1S_CASES = [
    # B, C1, C2, Cout, H, gn, residual
    (16, 256, 0, 768, 8, True, False),     # the fused q / k / v projection at first_n = 16 (GroupNorm prologue, no activation)
    (16, 256, 256, 256, 8, False, False),  # skip connection over a virtual concat: four channel slices
    (5, 256, 128, 256, 8, False, True),    # ragged last pixel tile (320 pixels), three slices, residual
    (3, 128, 0, 256, 16, False, False),    # a single slice
    (2, 256, 128, 128, 32, False, False),  # 32x32: sixteen tiles per image pair
]


@pytest.mark.parametrize("case", D1S_CASES)
def test_conv1x1_small_launch_split_f16_vs_conv2d(device, case, monkeypatch):
    """conv_d3s.hip's 1x1 form (skip connections, q / k / v at a few images): channel slices of 128 + the reduce pass against
    F.conv2d in float64; the planes of a fused weight packed member by member equal the planes packed at once."""
    monkeypatch.setenv("DDPM_CONV_D3S", "2")
    from ddpm_ood_amd import _lib, ops

    B, C1, C2, Cout, H, gn, res = case
    g = torch.Generator().manual_seed(B * 17 + C1 + H)
    Cin = C1 + C2
    x = torch.randn(B, C1, H, H, generator=g) * 1.7 - 0.3
    x2 = torch.randn(B, C2, H, H, generator=g) * 3 if C2 else None
    w = torch.randn(Cout, Cin, 1, 1, generator=g) / math.sqrt(Cin)
    w[: Cout // 2] *= 0.05  # members of very different magnitude: the scale is per packed member
    b = torch.randn(Cout, generator=g)
    gamma, beta = (torch.randn(Cin, generator=g), torch.randn(Cin, generator=g)) if gn else (None, None)
    residual = torch.randn(B, Cout, H, H, generator=g) if res else None
    xin = (x if x2 is None else torch.cat([x, x2], 1)).double()
    if gn:
        xin = F.group_norm(xin, 32, gamma.double(), beta.double(), 1e-6)
    ref = F.conv2d(xin, w.double(), b.double())
    if res:
        ref = ref + residual.double()
    d = lambda t: None if t is None else t.to(device)
    gs = gh = None
    if gn:
        gs, gh = ops.gn_scale_shift(d(x), d(gamma), d(beta), 32, 1e-6, x2=d(x2))
    kw = dict(x2=d(x2), gscale=gs, gshift=gh, act=ops.ACT_NONE, residual=d(residual))
    planes = ops.pack_conv_d1s_weight(d(w))
    assert planes is not None
    y = ops.conv(d(x), d(w), d(b), d3h=planes, **kw)
    y0 = ops.conv(d(x), d(w), d(b), **kw)  # without the planes: the library's other 1x1 kernels
    torch.cuda.synchronize()
    assert not torch.equal(y, y0)
    assert torch.equal(y, ops.conv(d(x), d(w), d(b), d3h=planes, **kw))
    scale = ref.abs().max().item()
    err = (y.cpu().double() - ref).abs().max().item() / scale
    err0 = (y0.cpu().double() - ref).abs().max().item() / scale
    print(f"{case}: small-launch 1x1 {err:.2e}, other kernel {err0:.2e} (max relative to max |y|)")
    assert math.isfinite(err) and err < 3e-6, (err, err0)
    # member-wise packing (what the UNet engine does for the fused q / k / v weight): two halves, each with its own scale
    lib = _lib.load()
    halves = torch.zeros_like(planes)
    wd = d(w).contiguous()
    for off in (0, Cout // 2):
        part = wd[off:off + Cout // 2].contiguous()
        ops.check(lib.ddpm_pack_conv_d1s_weight(part.data_ptr(), halves.data_ptr(), Cout // 2, Cin, off, Cout, ops.stream_ptr()), "pack")
    y2 = ops.conv(d(x), d(w), d(b), d3h=halves, **kw)
    torch.cuda.synchronize()
    err2 = (y2.cpu().double() - ref).abs().max().item() / scale
    assert err2 < 3e-6, err2
    # the small half keeps full precision only with its own scale: relative to ITS outputs
    sub = ref[:, : Cout // 2]
    rel_small = (y2.cpu().double()[:, : Cout // 2] - sub).abs().max().item() / (sub - (residual.double()[:, : Cout // 2] if res else 0)).abs().max().item()
    assert rel_small < 2e-5, rel_small
