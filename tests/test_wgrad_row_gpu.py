"""rbvae_wgrad3x3s2_row (csrc/wgrad_row.hip): the weight gradient of Conv2d(c, c, 3, 2, 1) / ConvTranspose2d(c, c, 3, 2, 1, 1) on
wide layers (autograd of models/percep_RBVAE/percep_RBVAE_model.py:54-57,76-81 as run by percep_RBVAE_train.py:552) with the
three taps of one kernel row per workgroup, against torch's autograd on the CPU and against the rbvae_wgrad_gemm launch it
replaces (same slab layout, same sums up to order)."""
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


def gemm_reference(lib, S, G, N, OH, OW, Ca, Cb):
    P = N * OH * OW
    zero = torch.zeros(256, dtype=torch.uint8, device="cuda")
    idx = torch.empty(9 * P, dtype=torch.int32, device="cuda")
    lib.call("rbvae_conv_gather_index", idx, N, 2 * OH, 2 * OW, OH, OW, 3, 3, 2, 1)
    ks2 = max(1, -(-P // 4096))
    slabs2 = torch.empty(ks2, Ca, 9, Cb, device="cuda")
    lib.call("rbvae_wgrad_gemm", 1, S, G, slabs2, idx, zero, P, N * 4 * OH * OW, Ca, Cb, S.stride(0), G.stride(0), 9, ks2)
    return slabs2.sum(0)


@pytest.mark.parametrize("N,OH,OW,Ca,Cb,ks", [(2, 8, 8, 128, 128, 1), (3, 11, 20, 128, 256, 4), (2, 22, 40, 256, 128, 7),
                                              (5, 4, 4, 128, 128, 2), (1, 32, 32, 128, 128, 16), (2, 16, 24, 256, 256, 3),
                                              (4, 9, 7, 128, 128, 8), (3, 5, 10, 128, 128, 5), (7, 4, 4, 256, 256, 1),
                                              (16, 8, 8, 256, 256, 11)])
def test_wgrad_row_matches_autograd_and_wgrad_gemm(sfv, N, OH, OW, Ca, Cb, ks):
    """both block widths (8: 8 x 8 / 22 x 40 / 9 x 7 images, 4: 4 x 4 / 11 x 20 / 5 x 10), blocks that span images and hang
    over the last row and the last column, 1..2 channel tiles either way, K-slices of unequal length"""
    lib = sfv._lib
    assert lib.query("rbvae_wgrad3x3s2_row_ok", 1, N, OH, OW, Ca, Cb)
    g = torch.Generator().manual_seed(100 + OH + Ca)
    x = torch.randn(N, Cb, 2 * OH, 2 * OW, generator=g)                   # the conv's input (high resolution)
    dy = torch.randn(N, Ca, OH, OW, generator=g) / 8                      # its output gradient (low resolution)
    xq, dyq = x.to(torch.bfloat16).float(), dy.to(torch.bfloat16).float()
    w = torch.zeros(Ca, Cb, 3, 3, requires_grad=True)
    F.conv2d(xq, w, None, stride=2, padding=1).backward(dyq)
    ref = w.grad.permute(0, 2, 3, 1).reshape(Ca, 9, Cb)                   # [a][t][b]
    S, G = rows(dy), rows(x)
    zero = torch.zeros(256, dtype=torch.uint8, device="cuda")
    nblk = lib.query("rbvae_wgrad3x3s2_row_blocks", N, OH, OW)
    ks = min(ks, nblk)
    slabs = torch.full((ks, Ca, 9, Cb), float("nan"), device="cuda")
    lib.call("rbvae_wgrad3x3s2_row", 1, S, G, slabs, zero, N, OH, OW, Ca, Cb, Ca, Cb, ks)
    got = slabs.sum(0)
    assert torch.isfinite(got).all()
    assert rel(got.cpu(), ref) < 3e-3
    assert rel(got, gemm_reference(lib, S, G, N, OH, OW, Ca, Cb)) < 2e-5      # f32 accumulation of the same bf16 products
    # run-to-run: bit-identical slabs (fixed summation order, no atomics)
    slabs_b = torch.empty_like(slabs)
    lib.call("rbvae_wgrad3x3s2_row", 1, S, G, slabs_b, zero, N, OH, OW, Ca, Cb, Ca, Cb, ks)
    assert torch.equal(slabs, slabs_b)


def test_wgrad_row_padded_rows_and_rejections(sfv):
    """operands that are column slices of wider row buffers (leading dimensions > channels); uncovered shapes are refused"""
    lib = sfv._lib
    N, OH, OW, Ca, Cb = 2, 8, 12, 128, 128
    g = torch.Generator().manual_seed(7)
    Sw = (torch.randn(N * OH * OW, 384, generator=g) / 8).to(torch.bfloat16).cuda()
    Gw = torch.randn(N * 4 * OH * OW, 256, generator=g).to(torch.bfloat16).cuda()
    S, G = Sw[:, 128:256], Gw[:, 128:]
    zero = torch.zeros(256, dtype=torch.uint8, device="cuda")
    slabs = torch.empty(2, Ca, 9, Cb, device="cuda")
    lib.call("rbvae_wgrad3x3s2_row", 1, S, G, slabs, zero, N, OH, OW, Ca, Cb, 384, 256, 2)
    slabs_c = torch.empty(2, Ca, 9, Cb, device="cuda")
    lib.call("rbvae_wgrad3x3s2_row", 1, S.contiguous(), G.contiguous(), slabs_c, zero, N, OH, OW, Ca, Cb, Ca, Cb, 2)
    assert torch.equal(slabs, slabs_c)
    assert not lib.query("rbvae_wgrad3x3s2_row_ok", 0, N, OH, OW, Ca, Cb)           # f32: rbvae_wgrad_gemm's job
    assert not lib.query("rbvae_wgrad3x3s2_row_ok", 1, N, OH, OW, 64, Cb)           # narrow layers: rbvae_wgrad3x3s2_halo's
    assert not lib.query("rbvae_wgrad3x3s2_row_ok", 1, N, OH, OW, Ca, 192)
    with pytest.raises(ValueError):
        lib.call("rbvae_wgrad3x3s2_row", 1, S, G, slabs, zero, N, OH, OW, 64, Cb, 384, 256, 2)
    with pytest.raises(ValueError):
        lib.call("rbvae_wgrad3x3s2_row", 1, S, G, slabs, zero, N, OH, OW, Ca, Cb, 384, 256, 10 ** 6)


@pytest.mark.parametrize("OH,OW", [(8, 8), (4, 4)])
def test_wgrad_row_at_the_bench_layer_sizes(sfv, OH, OW):
    """BASELINE configs[1] sizes (256 frames: conv2 / deconv2 = 8 x 8 low-resolution pixels, conv3 / deconv1 = 4 x 4, 256
    channels) at the engine's K-split: against rbvae_wgrad_gemm on the same operands, and linearity in S as a
    size-independent property (dW(4 S) = dW(S) + dW(3 S) up to f32 summation order)."""
    lib = sfv._lib
    N, Ca, Cb = 256, 256, 256
    P = N * OH * OW
    g = torch.Generator(device="cuda").manual_seed(5)
    S1 = (torch.randn(P, Ca, device="cuda", generator=g) / 8).bfloat16()
    G = torch.randn(4 * P, Cb, device="cuda", generator=g).bfloat16()
    zero = torch.zeros(256, dtype=torch.uint8, device="cuda")
    nblk = lib.query("rbvae_wgrad3x3s2_row_blocks", N, OH, OW)
    assert nblk == P // 64
    ks = min(21, nblk)

    def row(S):
        slabs = torch.empty(ks, Ca, 9, Cb, device="cuda")
        lib.call("rbvae_wgrad3x3s2_row", 1, S, G, slabs, zero, N, OH, OW, Ca, Cb, Ca, Cb, ks)
        return slabs.double().sum(0)

    d1 = row(S1)
    ref = gemm_reference(lib, S1, G, N, OH, OW, Ca, Cb).double()
    assert float((d1 - ref).norm() / ref.norm()) < 1e-5
    S2 = (S1.float() * 3).bfloat16()
    S12 = (S1.float() * 4).bfloat16()
    assert torch.equal(S12.float(), S1.float() * 4)
    lhs, rhs = row(S12), d1 + row(S2)
    exact3 = torch.equal(S2.float(), S1.float() * 3)          # 3 x a bf16 value may round: then compare loosely
    assert float((lhs - rhs).norm() / lhs.norm()) < (1e-6 if exact3 else 5e-3)
