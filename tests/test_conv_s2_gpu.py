"""rbvae_conv3x3s2_halo (csrc/conv_s2.hip): the 3x3 stride-2 pad-1 convolution with its input patch resident in LDS, against
torch on the CPU (nn.Conv2d(c, c, 3, 2, 1) of models/percep_RBVAE/percep_RBVAE_model.py:54-57; as the conv-form input
gradient of the ConvTranspose2d of :76-81) and, element for element, against the rbvae_gather_gemm launch it replaces (same
epilogue arithmetic, same dropout element indices)."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DT, TDT, TOL = 1, torch.bfloat16, 1.2e-2


@pytest.fixture(scope="module")
def sfv():
    import sfv_amd
    return sfv_amd


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / max(b.norm(), 1e-12))


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().reshape(-1, t.shape[1]).to(TDT).cuda()


def from_rows(r, N, H, W):
    return r.float().cpu().reshape(N, H, W, -1).permute(0, 3, 1, 2)


def halo(sfv, A, Wp, out, bias, gate, mask, N, IH, IW, cin, cout, relu=0, drop_mode=0, drop_p=0.0, scale=1.0, seed=0, colsum=None):
    sfv._lib.call("rbvae_conv3x3s2_halo", DT, A, Wp, out, bias, gate, mask, N, IH, IW, cin, cout, A.stride(0), out.stride(0), relu,
                  drop_mode, drop_p, scale, seed, None, colsum)


def gather(sfv, A, Wp, out, bias, gate, mask, N, IH, IW, cin, cout, relu=0, drop_mode=0, drop_p=0.0, scale=1.0, seed=0):
    from importlib import import_module
    E = import_module("symbols-from-video_amd.engine")
    desc = E.conv_classes(3)
    d = (ctypes.c_int * len(desc))(*desc)
    zero = torch.zeros(256, dtype=torch.uint8, device="cuda")
    sfv._lib.call("rbvae_gather_gemm", DT, A, Wp, out, bias, gate, mask, None, zero, N, IH, IW, IH // 2, IW // 2, 2, IH // 2, IW // 2,
                  1, cin, cout, A.shape[1], out.shape[1], 9, 1, ctypes.addressof(d), relu, drop_mode, drop_p, scale, seed, None, None)


@pytest.mark.parametrize("N,C,Co,IH,IW", [(2, 32, 128, 16, 32), (3, 64, 256, 16, 16), (2, 256, 256, 44, 80), (5, 96, 128, 8, 8),
                                          (1, 128, 256, 34, 66), (2, 256, 512, 12, 20), (9, 256, 256, 16, 16), (1, 64, 128, 2, 2)])
def test_conv_s2_matches_torch_and_gather_gemm(sfv, N, C, Co, IH, IW):
    """forward form (bias + ReLU + scale) and gradient form (gate + scale + column sums): both workgroup widths (256 / 128 output
    channels), 1..8 channel slices (odd and even counts: the loop takes two slices per turn), 1..2 channel tiles, images smaller
    than a tile, tiles that hang over the last rows / columns, images whose top / left halo is padding in every tile"""
    lib = sfv._lib
    assert lib.query("rbvae_conv3x3s2_halo_ok", DT, N, IH, IW, C, Co) in (128, 256)
    g = torch.Generator().manual_seed(17 + C + IW)
    x = torch.randn(N, C, IH, IW, generator=g)
    w = torch.randn(Co, C, 3, 3, generator=g) / (1.5 * C ** 0.5)
    b = torch.randn(Co, generator=g)
    xq, wq = x.to(TDT).float(), w.to(TDT).float()
    A = nhwc(x)
    Wp = w.permute(0, 2, 3, 1).contiguous().reshape(Co, 9, C).to(TDT).cuda()      # [cout][kh*3+kw][cin]
    OH, OW = IH // 2, IW // 2
    ref = F.relu(F.conv2d(xq, wq, b, stride=2, padding=1)) * 1.25
    out = torch.full((N * OH * OW, Co), float("nan"), dtype=TDT, device="cuda")
    halo(sfv, A, Wp, out, b.cuda(), None, None, N, IH, IW, C, Co, relu=1, scale=1.25)
    got = from_rows(out, N, OH, OW)
    assert torch.isfinite(got).all()
    assert rel(got, ref) < TOL
    if C % 64 == 0:                                                              # (rbvae_gather_gemm takes whole 64-channel slices)
        out_g = torch.empty_like(out)
        gather(sfv, A, Wp, out_g, b.cuda(), None, None, N, IH, IW, C, Co, relu=1, scale=1.25)
        assert rel(got, from_rows(out_g, N, OH, OW)) < 4e-3                      # summation order + one bf16 rounding
    # gradient form: gate of the layer below, scale, per-tile column sums (the bias gradient)
    gate = torch.randn(N, Co, OH, OW, generator=g)
    rows = lib.query("rbvae_conv3x3s2_halo_colsum_rows", N, IH, IW)
    ws = torch.full((rows, Co), float("nan"), device="cuda")
    out2 = torch.empty_like(out)
    halo(sfv, A, Wp, out2, None, nhwc(gate), None, N, IH, IW, C, Co, scale=0.5, colsum=ws)
    ref2 = F.conv2d(xq, wq, None, stride=2, padding=1) * 0.5 * (gate.to(TDT).float() > 0)
    got2 = from_rows(out2, N, OH, OW)
    assert rel(got2, ref2) < TOL
    cs = ws.sum(0).cpu()
    np.testing.assert_allclose(cs.numpy(), got2.sum((0, 2, 3)).numpy(), rtol=2e-3, atol=2e-2)
    # run to run: bit-identical
    out3 = torch.empty_like(out)
    halo(sfv, A, Wp, out3, b.cuda(), None, None, N, IH, IW, C, Co, relu=1, scale=1.25)
    assert torch.equal(out3, out)


def test_conv_s2_dropout_is_gather_gemms_dropout(sfv):
    """keyed dropout (drop_mode 1) zeroes exactly the elements rbvae_gather_gemm's zeroes for the same seed (the hash is keyed by
    the output element index), and the explicit-mask mode applies the reference-captured keep-mask"""
    N, C, Co, IH, IW = 6, 64, 256, 16, 32
    g = torch.Generator().manual_seed(34)
    x = torch.randn(N, C, IH, IW, generator=g)
    w = torch.randn(Co, C, 3, 3, generator=g) / (1.5 * C ** 0.5)
    b = torch.randn(Co, generator=g)
    A = nhwc(x)
    Wp = w.permute(0, 2, 3, 1).contiguous().reshape(Co, 9, C).to(TDT).cuda()
    OH, OW = IH // 2, IW // 2
    o1 = torch.empty(N * OH * OW, Co, dtype=TDT, device="cuda")
    o2 = torch.empty_like(o1)
    halo(sfv, A, Wp, o1, b.cuda(), None, None, N, IH, IW, C, Co, relu=1, drop_mode=1, drop_p=0.2, scale=1.25, seed=77)
    gather(sfv, A, Wp, o2, b.cuda(), None, None, N, IH, IW, C, Co, relu=1, drop_mode=1, drop_p=0.2, scale=1.25, seed=77)
    plain = torch.empty_like(o1)
    halo(sfv, A, Wp, plain, b.cuda(), None, None, N, IH, IW, C, Co, relu=1, scale=1.25)
    live = plain.float() != 0                                             # ReLU zeros say nothing about the mask
    assert torch.equal((o1.float() == 0)[live], (o2.float() == 0)[live])
    frac = float((o1.float() == 0)[live].float().mean())
    assert 0.17 < frac < 0.23
    keep = torch.rand(N * OH * OW, Co, generator=g) > 0.2
    o3 = torch.empty_like(o1)
    halo(sfv, A, Wp, o3, b.cuda(), None, keep.to(torch.uint8).cuda(), N, IH, IW, C, Co, relu=1, drop_mode=2, drop_p=0.2, scale=1.25)
    assert torch.equal(o3.float().cpu(), plain.float().cpu() * keep)


def test_conv_s2_padded_rows_and_rejections(sfv):
    """operands that are column slices of wider row buffers; uncovered shapes are refused"""
    lib = sfv._lib
    N, C, Co, IH, IW = 2, 64, 128, 16, 16
    g = torch.Generator().manual_seed(8)
    Aw = torch.randn(N * IH * IW, 192, generator=g).to(TDT).cuda()
    A = Aw[:, 64:128]
    Wp = (torch.randn(Co, 9, C, generator=g) / 24).to(TDT).cuda()
    Ow = torch.zeros(N * 64, 384, dtype=TDT, device="cuda")
    O = Ow[:, 128:256]
    halo(sfv, A, Wp, O, None, None, None, N, IH, IW, C, Co)
    Oc = torch.empty(N * 64, Co, dtype=TDT, device="cuda")
    halo(sfv, A.contiguous(), Wp, Oc, None, None, None, N, IH, IW, C, Co)
    assert torch.equal(O, Oc) and float(Ow[:, :128].abs().max()) == 0 and float(Ow[:, 256:].abs().max()) == 0
    assert not lib.query("rbvae_conv3x3s2_halo_ok", 0, N, IH, IW, C, Co)          # f32: rbvae_gather_gemm's
    assert not lib.query("rbvae_conv3x3s2_halo_ok", 1, N, 15, IW, C, Co)          # odd image side
    assert not lib.query("rbvae_conv3x3s2_halo_ok", 1, N, IH, IW, 48, Co)         # channels not whole 32-channel slices
    assert not lib.query("rbvae_conv3x3s2_halo_ok", 1, N, IH, IW, C, 64)          # narrower than a channel tile
    with pytest.raises(ValueError):
        halo(sfv, A, Wp, O, None, None, None, N, 15, IW, C, Co)
