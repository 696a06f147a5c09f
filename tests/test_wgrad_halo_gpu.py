"""rbvae_wgrad3x3s2_halo (csrc/wgrad_halo.hip): the weight gradient of Conv2d(c, c, 3, 2, 1) / ConvTranspose2d(c, c, 3, 2, 1, 1)
(autograd of models/percep_RBVAE/percep_RBVAE_model.py:51-57,76-81) with the nine taps in one workgroup, against torch's
autograd on the CPU and against the rbvae_wgrad_gemm launch it replaces (same slab layout, same sums up to order)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sfv():
    import sfv_amd
    return sfv_amd


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / max(b.norm(), 1e-12))


def rows(t):
    return t.permute(0, 2, 3, 1).contiguous().reshape(-1, t.shape[1]).to(torch.bfloat16).cuda()


@pytest.mark.parametrize("N,OH,OW,Ca,Cb,ks", [(2, 8, 8, 64, 64, 1), (3, 11, 20, 64, 128, 4), (2, 22, 40, 128, 64, 7),
                                              (5, 4, 4, 64, 64, 5), (1, 32, 32, 64, 64, 16), (2, 16, 24, 256, 128, 3),
                                              (4, 9, 7, 64, 64, 8)])
def test_wgrad_halo_matches_autograd_and_wgrad_gemm(sfv, N, OH, OW, Ca, Cb, ks):
    """blocks that hang over the image on both sides (11 x 20, 9 x 7, 4 x 4), several images, 1..4 channel tiles either way,
    K-slices that split an image and K-slices left without a block"""
    lib = sfv._lib
    assert lib.query("rbvae_wgrad3x3s2_halo_ok", 1, N, OH, OW, Ca, Cb)
    g = torch.Generator().manual_seed(100 + OH + Ca)
    x = torch.randn(N, Cb, 2 * OH, 2 * OW, generator=g)                   # the conv's input (high resolution)
    dy = torch.randn(N, Ca, OH, OW, generator=g) / 8                      # its output gradient (low resolution)
    xq, dyq = x.to(torch.bfloat16).float(), dy.to(torch.bfloat16).float()
    w = torch.zeros(Ca, Cb, 3, 3, requires_grad=True)
    F.conv2d(xq, w, None, stride=2, padding=1).backward(dyq)
    ref = w.grad.permute(0, 2, 3, 1).reshape(Ca, 9, Cb)                   # [a][t][b]
    S, G = rows(dy), rows(x)
    zero = torch.zeros(256, dtype=torch.uint8, device="cuda")
    nblk = lib.query("rbvae_wgrad3x3s2_halo_blocks", N, OH, OW)
    ks = min(ks, nblk)
    slabs = torch.full((ks, Ca, 9, Cb), float("nan"), device="cuda")
    lib.call("rbvae_wgrad3x3s2_halo", 1, S, G, slabs, zero, N, OH, OW, Ca, Cb, Ca, Cb, ks)
    got = slabs.sum(0).cpu()
    assert torch.isfinite(got).all()
    assert rel(got, ref) < 3e-3
    # the launch it replaces: gather table + one workgroup set per tap
    P = N * OH * OW
    idx = torch.empty(9 * P, dtype=torch.int32, device="cuda")
    lib.call("rbvae_conv_gather_index", idx, N, 2 * OH, 2 * OW, OH, OW, 3, 3, 2, 1)
    ks2 = max(1, -(-P // 4096))
    slabs2 = torch.empty(ks2, Ca, 9, Cb, device="cuda")
    lib.call("rbvae_wgrad_gemm", 1, S, G, slabs2, idx, zero, P, N * 4 * OH * OW, Ca, Cb, Ca, Cb, 9, ks2)
    assert rel(got, slabs2.sum(0).cpu()) < 2e-5                           # f32 accumulation of the same bf16 products


def test_wgrad_halo_padded_rows_and_rejections(sfv):
    """operands that are column slices of wider row buffers (leading dimensions > channels); uncovered shapes are refused"""
    lib = sfv._lib
    N, OH, OW, Ca, Cb = 2, 8, 12, 64, 64
    g = torch.Generator().manual_seed(7)
    Sw = (torch.randn(N * OH * OW, 192, generator=g) / 8).to(torch.bfloat16).cuda()
    Gw = torch.randn(N * 4 * OH * OW, 128, generator=g).to(torch.bfloat16).cuda()
    S, G = Sw[:, 64:128], Gw[:, 64:]
    zero = torch.zeros(256, dtype=torch.uint8, device="cuda")
    slabs = torch.empty(2, Ca, 9, Cb, device="cuda")
    lib.call("rbvae_wgrad3x3s2_halo", 1, S, G, slabs, zero, N, OH, OW, Ca, Cb, 192, 128, 2)
    slabs_c = torch.empty(2, Ca, 9, Cb, device="cuda")
    lib.call("rbvae_wgrad3x3s2_halo", 1, S.contiguous(), G.contiguous(), slabs_c, zero, N, OH, OW, Ca, Cb, Ca, Cb, 2)
    assert torch.equal(slabs, slabs_c)
    assert not lib.query("rbvae_wgrad3x3s2_halo_ok", 0, N, OH, OW, Ca, Cb)           # f32: rbvae_wgrad_gemm's job
    assert not lib.query("rbvae_wgrad3x3s2_halo_ok", 1, N, OH, OW, 32, Cb)
    assert not lib.query("rbvae_wgrad3x3s2_halo_ok", 1, N, OH, OW, Ca, 96)
    with pytest.raises(ValueError):
        lib.call("rbvae_wgrad3x3s2_halo", 1, S, G, slabs, zero, N, OH, OW, 32, Cb, 192, 128, 2)
    with pytest.raises(ValueError):
        lib.call("rbvae_wgrad3x3s2_halo", 1, S, G, slabs, zero, N, OH, OW, Ca, Cb, 192, 128, 10 ** 6)


def test_wgrad_halo_at_the_cfg3_layer_sizes(sfv):
    """BASELINE configs[2] sizes (128 frames of 256 x 256: conv2 / deconv2 = 64 x 64 pixels, 64 channels, 524 288 pixels; the
    engine's K-split of 256): the nine-tap kernel against rbvae_wgrad_gemm on the same operands, and linearity in S
    (dW(S1 + S2) = dW(S1) + dW(S2) up to f32 summation order) as a size-independent property."""
    lib = sfv._lib
    N, OH, OW, Ca, Cb = 128, 64, 64, 64, 64
    P = N * OH * OW
    g = torch.Generator(device="cuda").manual_seed(5)
    S1 = (torch.randn(P, Ca, device="cuda", generator=g) / 8).bfloat16()
    G = torch.randn(4 * P, Cb, device="cuda", generator=g).bfloat16()
    zero = torch.zeros(256, dtype=torch.uint8, device="cuda")
    nblk = lib.query("rbvae_wgrad3x3s2_halo_blocks", N, OH, OW)
    assert nblk == N * 16 * 8
    ks = 256

    def halo(S):
        slabs = torch.empty(ks, Ca, 9, Cb, device="cuda")
        lib.call("rbvae_wgrad3x3s2_halo", 1, S, G, slabs, zero, N, OH, OW, Ca, Cb, Ca, Cb, ks)
        return slabs.double().sum(0)

    d1 = halo(S1)
    idx = torch.empty(9 * P, dtype=torch.int32, device="cuda")
    lib.call("rbvae_conv_gather_index", idx, N, 2 * OH, 2 * OW, OH, OW, 3, 3, 2, 1)
    k2 = P // 4096
    slabs2 = torch.empty(k2, Ca, 9, Cb, device="cuda")
    lib.call("rbvae_wgrad_gemm", 1, S1, G, slabs2, idx, zero, P, 4 * P, Ca, Cb, Ca, Cb, 9, k2)
    ref = slabs2.double().sum(0)
    assert float((d1 - ref).norm() / ref.norm()) < 1e-5
    # linearity: operands whose sum is exact in bf16 (S2 = 3 S1 -> S1 + S2 = 4 S1)
    S2 = (S1.float() * 3).bfloat16()
    S12 = (S1.float() * 4).bfloat16()
    assert torch.equal(S12.float(), S1.float() * 4)
    lhs, rhs = halo(S12), d1 + halo(S2)
    exact3 = torch.equal(S2.float(), S1.float() * 3)          # 3 x a bf16 value may round: then compare loosely
    assert float((lhs - rhs).norm() / lhs.norm()) < (1e-6 if exact3 else 5e-3)
