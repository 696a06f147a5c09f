"""rbvae_conv3x3_halo (halo-resident stride-1 3x3 convolution, csrc/conv_halo.hip) against torch on the CPU.

Reference ops: torch.nn.Conv2d(cin, cout, 3, 1, 1) of the LDM ResnetBlock (ldm/modules/diffusionmodules/model.py:
82-141), with GroupNorm(32, eps 1e-6) + swish (:33-39) of the PRODUCER folded into the staging of the input and the
statistics of the NEXT GroupNorm taken from the stored tile."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DT = {"f32": (0, torch.float32, 2e-5), "bf16": (1, torch.bfloat16, 1.2e-2)}


@pytest.fixture(scope="module")
def sfv():
    import sfv_amd
    return sfv_amd


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / max(b.norm(), 1e-12))


def nhwc(t, tdt):
    return t.permute(0, 2, 3, 1).contiguous().reshape(-1, t.shape[1]).to(tdt).cuda()


def from_rows(r, N, H, W):
    return r.float().cpu().reshape(N, H, W, -1).permute(0, 3, 1, 2)


def pack_w(w, tdt):      # [co][ci][3][3] -> [co][tap][ci]
    return w.permute(0, 2, 3, 1).contiguous().reshape(w.shape[0], 9, w.shape[1]).to(tdt).cuda()


def halo(sfv, dt, A, Wp, out, bias, addend, N, H, W, cin, cout, scale=None, shift=None, swish=0, stats=None, cg=0):
    zero = torch.zeros(256, dtype=torch.uint8, device="cuda")
    sfv._lib.call("rbvae_conv3x3_halo", dt, A, Wp, out, bias, addend, zero, scale, shift, swish, stats, cg, N, H, W, H, W,
                  1, 1, cin, cout, A.shape[1], out.shape[1])


@pytest.mark.parametrize("dtype,N,C,Co,H,W", [("f32", 2, 32, 128, 16, 16), ("f32", 1, 64, 128, 24, 40),
                                              ("bf16", 2, 64, 128, 16, 16), ("bf16", 1, 128, 256, 40, 24),
                                              ("bf16", 3, 256, 128, 22, 40), ("bf16", 1, 512, 512, 32, 32),
                                              ("bf16", 2, 128, 128, 8, 16), ("f32", 1, 96, 256, 17, 33)])
def test_conv3x3_halo_matches_torch(sfv, dtype, N, C, Co, H, W):
    """plain convolution + bias + residual: full tiles, ragged tiles (24x40, 22x40, 17x33), several images, every
    channel-slice count (1, 2, 4, 8 slices) and 1 / 2 / 4 output-channel tiles"""
    dt, tdt, tol = DT[dtype]
    assert sfv._lib.query("rbvae_conv3x3_halo_ok", dt, H, W, H, W, C, Co)
    g = torch.Generator().manual_seed(7 + C + H)
    x = torch.randn(N, C, H, W, generator=g)
    w = torch.randn(Co, C, 3, 3, generator=g) / (3.0 * C ** 0.5)
    b = torch.randn(Co, generator=g)
    skip = torch.randn(N, Co, H, W, generator=g)
    xq, wq, sq = x.to(tdt).float(), w.to(tdt).float(), skip.to(tdt).float()
    ref = F.conv2d(xq, wq, b, 1, 1) + sq
    A, Wp, S = nhwc(x, tdt), pack_w(w, tdt), nhwc(skip, tdt)
    out = torch.full((N * H * W, Co), float("nan"), dtype=tdt, device="cuda")
    halo(sfv, dt, A, Wp, out, b.cuda(), S, N, H, W, C, Co)
    got = from_rows(out, N, H, W)
    assert torch.isfinite(got).all()
    assert rel(got, ref) < tol
    assert float((got - ref).abs().max()) < (5e-4 if dtype == "f32" else 8e-2) * max(1.0, float(ref.abs().max()))
    # no bias, no residual
    out2 = torch.empty_like(out)
    halo(sfv, dt, A, Wp, out2, None, None, N, H, W, C, Co)
    assert rel(from_rows(out2, N, H, W), F.conv2d(xq, wq, None, 1, 1)) < tol


@pytest.mark.parametrize("dtype,N,C,Co,H,W,swish", [("f32", 2, 64, 128, 16, 32, 1), ("bf16", 2, 128, 128, 24, 24, 1),
                                                    ("bf16", 1, 256, 256, 16, 16, 0), ("f32", 1, 128, 128, 19, 21, 1)])
def test_fused_groupnorm_input_and_output_statistics(sfv, dtype, N, C, Co, H, W, swish):
    """GroupNorm(32, eps 1e-6) + swish of the input applied while the patch is staged (padding stays zero), and the stored
    tile's per-group statistics merged by rbvae_gn_finish_tiles == torch's mean / variance of the convolution's output"""
    dt, tdt, tol = DT[dtype]
    lib = sfv._lib
    g = torch.Generator().manual_seed(11 + C + W)
    x = torch.randn(N, C, H, W, generator=g) * 0.7 + 1.5
    w = torch.randn(Co, C, 3, 3, generator=g) / (3.0 * C ** 0.5)
    b = torch.randn(Co, generator=g)
    gamma, beta = torch.randn(C, generator=g), torch.randn(C, generator=g)
    xq, wq = x.to(tdt).float(), w.to(tdt).float()
    hn = F.group_norm(xq, 32, gamma, beta, eps=1e-6)
    if swish:
        hn = hn * torch.sigmoid(hn)
    hq = hn.to(tdt).float()                                   # the kernel rounds the normalised value to the storage type
    ref = F.conv2d(hq, wq, b, 1, 1)
    # statistics of the input by the existing kernels -> scale / shift
    A = nhwc(x, tdt)
    nws = lib.query("rbvae_groupnorm_ws_floats", dt, N, H * W, C, 32)
    ws = torch.empty(nws, device="cuda")
    tmp = torch.empty_like(A)
    lib.call("rbvae_groupnorm_swish_ws", dt, A, tmp, gamma.cuda(), beta.cuda(), ws, nws, N, H * W, C, C, C, 32, 1e-6, swish)
    scale, shift = torch.empty(N, C, device="cuda"), torch.empty(N, C, device="cuda")
    lib.call("rbvae_gn_affine", ws[:N * 32], ws[N * 32:2 * N * 32], gamma.cuda(), beta.cuda(), scale, shift, N, C, 32)
    cg = Co // 32
    nst = lib.query("rbvae_conv3x3_halo_stats_floats", N, H, W, Co, cg)
    stats = torch.full((nst,), float("nan"), device="cuda")
    out = torch.empty(N * H * W, Co, dtype=tdt, device="cuda")
    halo(sfv, dt, A, pack_w(w, tdt), out, b.cuda(), None, N, H, W, C, Co, scale, shift, swish, stats, cg)
    got = from_rows(out, N, H, W)
    assert rel(got, ref) < (3e-5 if dtype == "f32" else 1.5e-2)
    # the fused apply == the standalone GroupNorm kernel feeding the plain convolution
    out_b = torch.empty_like(out)
    halo(sfv, dt, tmp, pack_w(w, tdt), out_b, b.cuda(), None, N, H, W, C, Co)
    assert rel(got, from_rows(out_b, N, H, W)) < (1e-6 if dtype == "f32" else 4e-3)
    # output statistics
    g2, b2 = torch.randn(Co, generator=g), torch.randn(Co, generator=g)
    sc2, sh2 = torch.empty(N, Co, device="cuda"), torch.empty(N, Co, device="cuda")
    mean, rstd = torch.empty(N * 32, device="cuda"), torch.empty(N * 32, device="cuda")
    lib.call("rbvae_gn_finish_tiles", stats, g2.cuda(), b2.cuda(), sc2, sh2, mean, rstd, N, H, W, Co, 32, 1e-6, 16, 16)
    grp = got.reshape(N, 32, -1)
    m_ref, v_ref = grp.mean(-1), grp.var(-1, unbiased=False)
    np.testing.assert_allclose(mean.cpu().reshape(N, 32).numpy(), m_ref.numpy(), rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(rstd.cpu().reshape(N, 32).numpy(), (v_ref + 1e-6).rsqrt().numpy(), rtol=1e-4)
    y_ref = F.group_norm(got, 32, g2, b2, eps=1e-6)
    y_got = got * sc2.cpu()[:, :, None, None] + sh2.cpu()[:, :, None, None]
    assert float((y_got - y_ref).abs().max()) < 2e-4 * max(1.0, float(y_ref.abs().max()))


def test_conv3x3_halo_rejects_uncovered_shapes(sfv):
    lib = sfv._lib
    assert not lib.query("rbvae_conv3x3_halo_ok", 1, 8, 8, 8, 8, 128, 128)       # narrower than a tile: gather_gemm
    assert not lib.query("rbvae_conv3x3_halo_ok", 1, 32, 32, 32, 32, 100, 128)
    assert not lib.query("rbvae_conv3x3_halo_ok", 1, 32, 32, 32, 32, 128, 8)
    A = torch.zeros(64, 128, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(ValueError):
        halo(sfv, 1, A, A, A, None, None, 1, 8, 8, 128, 128)
