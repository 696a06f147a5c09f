"""rbvae_deconv3x3s2_halo (csrc/deconv_halo.hip): the transposed 3x3 stride-2 convolution with its four output-parity
classes in one workgroup and the input patch resident in LDS, against torch on the CPU (ConvTranspose2d(c, c, 3, 2, 1, 1)
of models/percep_RBVAE/percep_RBVAE_model.py:76-81; as the Conv2d's input gradient: autograd of :54-57) and, element for
element, against the rbvae_gather_gemm launches it replaces (same epilogue arithmetic, same dropout element indices)."""
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


def halo(sfv, dt, A, Wp, out, bias, gate, mask, N, TH, TW, cin, cout, relu=0, drop_mode=0, drop_p=0.0, scale=1.0, seed=0,
         colsum=None):
    zero = torch.zeros(256, dtype=torch.uint8, device="cuda")
    sfv._lib.call("rbvae_deconv3x3s2_halo", dt, A, Wp, out, bias, gate, mask, zero, N, TH, TW, cin, cout, A.shape[1],
                  out.shape[1], relu, drop_mode, drop_p, scale, seed, None, colsum)


def gather(sfv, dt, A, Wp, out, bias, gate, mask, N, TH, TW, cin, cout, relu=0, drop_mode=0, drop_p=0.0, scale=1.0, seed=0,
           colsum=None):
    from importlib import import_module
    E = import_module("symbols-from-video_amd.engine")
    desc, ncls = E.dgrad_classes(3)
    d = (ctypes.c_int * len(desc))(*desc)
    zero = torch.zeros(256, dtype=torch.uint8, device="cuda")
    sfv._lib.call("rbvae_gather_gemm", dt, A, Wp, out, bias, gate, mask, None, zero, N, TH, TW, TH, TW, 1, 2 * TH, 2 * TW, 2,
                  cin, cout, A.shape[1], out.shape[1], 9, ncls, ctypes.addressof(d), relu, drop_mode, drop_p, scale, seed, None,
                  colsum)


@pytest.mark.parametrize("dtype,N,C,Co,TH,TW", [("f32", 2, 32, 64, 8, 8), ("f32", 5, 64, 64, 4, 4), ("bf16", 4, 64, 64, 8, 8),
                                                ("bf16", 40, 256, 256, 4, 4), ("bf16", 3, 128, 128, 22, 40),
                                                ("bf16", 2, 256, 64, 11, 20), ("bf16", 1, 64, 128, 16, 16),
                                                ("f32", 1, 32, 64, 17, 32), ("bf16", 2, 64, 64, 64, 64),
                                                ("bf16", 37, 256, 256, 8, 8)])
def test_deconv_halo_matches_torch_and_gather_gemm(sfv, dtype, N, C, Co, TH, TW):
    """forward form (bias + ReLU + scale) and gradient form (gate + scale + column sums): strips of 16 / 8 / 4 columns,
    tiles that pack several images and strips (4x4 and 8x8 grids), ragged last tiles, 1..4 channel slices, 1..4 output
    channel tiles"""
    dt, tdt, tol = DT[dtype]
    lib = sfv._lib
    assert lib.query("rbvae_deconv3x3s2_halo_ok", dt, N, TH, TW, C, Co)
    g = torch.Generator().manual_seed(13 + C + TW)
    x = torch.randn(N, C, TH, TW, generator=g)
    w = torch.randn(C, Co, 3, 3, generator=g) / (1.5 * C ** 0.5)          # ConvTranspose2d layout [cin][cout][kh][kw]
    b = torch.randn(Co, generator=g)
    xq, wq = x.to(tdt).float(), w.to(tdt).float()
    A = nhwc(x, tdt)
    Wp = w.permute(1, 2, 3, 0).contiguous().reshape(Co, 9, C).to(tdt).cuda()   # [cout][kh*3+kw][cin]
    OH, OW = 2 * TH, 2 * TW
    ref = F.relu(F.conv_transpose2d(xq, wq, b, stride=2, padding=1, output_padding=1)) * 1.25
    out = torch.full((N * OH * OW, Co), float("nan"), dtype=tdt, device="cuda")
    halo(sfv, dt, A, Wp, out, b.cuda(), None, None, N, TH, TW, C, Co, relu=1, scale=1.25)
    got = from_rows(out, N, OH, OW)
    assert torch.isfinite(got).all()
    assert rel(got, ref) < tol
    out_g = torch.empty_like(out)
    gather(sfv, dt, A, Wp, out_g, b.cuda(), None, None, N, TH, TW, C, Co, relu=1, scale=1.25)
    assert rel(got, from_rows(out_g, N, OH, OW)) < (2e-6 if dtype == "f32" else 4e-3)    # summation order only
    # gradient form: gate of the layer below, scale, per-tile column sums (the bias gradient)
    gate = torch.randn(N, Co, OH, OW, generator=g)
    rows = lib.query("rbvae_deconv3x3s2_halo_colsum_rows", dt, N, TH, TW, C, Co)
    ws = torch.full((rows, Co), float("nan"), device="cuda")
    out2 = torch.empty_like(out)
    halo(sfv, dt, A, Wp, out2, None, nhwc(gate, tdt), None, N, TH, TW, C, Co, scale=0.5, colsum=ws)
    ref2 = F.conv_transpose2d(xq, wq, None, stride=2, padding=1, output_padding=1) * 0.5 * (gate.to(tdt).float() > 0)
    got2 = from_rows(out2, N, OH, OW)
    assert rel(got2, ref2) < tol
    cs = ws.sum(0).cpu()
    np.testing.assert_allclose(cs.numpy(), got2.sum((0, 2, 3)).numpy(), rtol=2e-3 if dtype == "bf16" else 1e-4,
                               atol=2e-2 if dtype == "bf16" else 1e-3)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_deconv_halo_dropout_is_gather_gemms_dropout(sfv, dtype):
    """keyed dropout (drop_mode 1) zeroes exactly the elements rbvae_gather_gemm's zeroes for the same seed (the hash is
    keyed by the output element index), and the explicit-mask mode applies the reference-captured keep-mask"""
    dt, tdt, tol = DT[dtype]
    N, C, Co, TH, TW = 6, 64, 128, 8, 8
    g = torch.Generator().manual_seed(33)
    x = torch.randn(N, C, TH, TW, generator=g)
    w = torch.randn(C, Co, 3, 3, generator=g) / (1.5 * C ** 0.5)
    b = torch.randn(Co, generator=g)
    A = nhwc(x, tdt)
    Wp = w.permute(1, 2, 3, 0).contiguous().reshape(Co, 9, C).to(tdt).cuda()
    OH, OW = 2 * TH, 2 * TW
    o1 = torch.empty(N * OH * OW, Co, dtype=tdt, device="cuda")
    o2 = torch.empty_like(o1)
    halo(sfv, dt, A, Wp, o1, b.cuda(), None, None, N, TH, TW, C, Co, relu=1, drop_mode=1, drop_p=0.2, scale=1.25, seed=77)
    gather(sfv, dt, A, Wp, o2, b.cuda(), None, None, N, TH, TW, C, Co, relu=1, drop_mode=1, drop_p=0.2, scale=1.25, seed=77)
    plain = torch.empty_like(o1)
    halo(sfv, dt, A, Wp, plain, b.cuda(), None, None, N, TH, TW, C, Co, relu=1, scale=1.25)
    live = plain.float() != 0                                             # ReLU zeros say nothing about the mask
    assert torch.equal((o1.float() == 0)[live], (o2.float() == 0)[live])
    frac = float((o1.float() == 0)[live].float().mean())
    assert 0.17 < frac < 0.23
    keep = torch.rand(N * OH * OW, Co, generator=g) > 0.2
    o3 = torch.empty_like(o1)
    halo(sfv, dt, A, Wp, o3, b.cuda(), None, keep.to(torch.uint8).cuda(), N, TH, TW, C, Co, relu=1, drop_mode=2, drop_p=0.2,
         scale=1.25)
    assert torch.equal(o3.float().cpu(), plain.float().cpu() * keep)


def test_deconv_halo_rejects_uncovered_shapes(sfv):
    lib = sfv._lib
    assert not lib.query("rbvae_deconv3x3s2_halo_ok", 1, 4, 8, 6, 64, 64)        # width not a multiple of 4
    assert not lib.query("rbvae_deconv3x3s2_halo_ok", 1, 4, 8, 8, 48, 64)        # channels not a whole slice
    assert not lib.query("rbvae_deconv3x3s2_halo_ok", 1, 4, 8, 8, 64, 32)        # narrower than a channel tile
    assert not lib.query("rbvae_deconv3x3s2_halo_ok", 1, 64, 2, 16, 64, 64)      # 2-row strips: the patch rows do not fit
    A = torch.zeros(64, 64, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(ValueError):
        halo(sfv, 1, A, A, A, None, None, None, 1, 8, 6, 64, 64)
