"""Unit parity of the GEMM / LSTM / layout kernels against plain torch fp32 on the CPU."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import rbvae_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sfv():
    import sfv_amd
    return sfv_amd


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / max(b.norm(), 1e-12))


DT = {"f32": (0, torch.float32, 1e-5), "bf16": (1, torch.bfloat16, 1.2e-2)}


def nhwc(t, tdt):       # [N,C,H,W] -> rows [N*H*W, C] on the GPU
    return t.permute(0, 2, 3, 1).contiguous().reshape(-1, t.shape[1]).to(tdt).cuda()


def from_rows(r, N, H, W):
    return r.float().cpu().reshape(N, H, W, -1).permute(0, 3, 1, 2)


def gemm(sfv, dt, A, Wp, out, bias, gate, mask, geom, kc, nout, taps, desc, ncls, relu=0, drop_mode=0, drop_p=0.0,
         scale=1.0, seed=0, colsum_ws=None):
    zero = torch.zeros(256, dtype=torch.uint8, device="cuda")
    d = (ctypes.c_int * len(desc))(*desc)
    sfv._lib.call("rbvae_gather_gemm", dt, A, Wp, out, bias, gate, mask, None, zero, *geom, kc, nout, A.shape[1],
                  out.shape[1], taps, ncls, ctypes.addressof(d), relu, drop_mode, drop_p, scale, seed, None, colsum_ws)


@pytest.mark.parametrize("M,N,lda,with_bias,K", [(256, 4096, 64, True, 64), (256, 4096, 64, False, 64), (37, 1024, 72, True, 64),
                                                  (300, 2064, 64, False, 64), (1, 16, 64, True, 64), (256, 4096, 128, True, 128),
                                                  (70, 1040, 136, False, 128),
                                                  # N % 128 == 0 and at least 256 workgroups: fc_gemm_wide_k (64 x 64 per wave)
                                                  (128, 56320, 64, True, 64), (200, 32768, 72, False, 64), (1, 32768, 64, True, 64),
                                                  (513, 8192, 64, True, 64)])
def test_fc_gemm_equals_one_tap_gather_gemm(sfv, M, N, lda, with_bias, K):
    """rbvae_fc_gemm (the K = 64 fc products, operands straight into the MFMA layout) against rbvae_gather_gemm with one
    tap: outputs bit for bit (same MFMAs in the same k order, same rounding), the per-128-row column sums to f32
    rounding of their different summation order, and against the f32 product of the bf16 operands."""
    lib = sfv._lib
    assert lib.query("rbvae_fc_gemm_ok", 1, M, K, N, lda, N)
    g = torch.Generator().manual_seed(90 + M)
    A = torch.zeros(M, lda, dtype=torch.bfloat16)
    A[:, :K] = torch.randn(M, K, generator=g).bfloat16()
    A = A.cuda()
    W = (torch.randn(N, K, generator=g) / 8).bfloat16().cuda()
    b = torch.randn(N, generator=g).cuda() if with_bias else None
    mt = -(-M // 128)
    out0, ws0 = torch.empty(M, N, dtype=torch.bfloat16, device="cuda"), torch.zeros(mt, N, device="cuda")
    gemm(sfv, 1, A, W, out0, b, None, None, (M, 1, 1, 1, 1, 1, 1, 1, 1), K, N, 1, (1, 0, 0, 0, 0, 0), 1, colsum_ws=ws0)
    out1, ws1 = torch.full((M, N), 7.0, dtype=torch.bfloat16, device="cuda"), torch.full((mt, N), 7.0, device="cuda")
    lib.call("rbvae_fc_gemm", 1, A, W, out1, b, ws1, M, K, N, lda, N)
    torch.cuda.synchronize()
    assert torch.equal(out0, out1)
    ref = A[:, :K].float() @ W.float().t() + (b if with_bias else 0)
    assert rel(out1.float().cpu(), ref.cpu()) < 4e-3
    np.testing.assert_allclose(ws1.cpu().numpy(), ws0.cpu().numpy(), rtol=1e-5, atol=1e-4)
    sums = torch.stack([out1[128 * t:128 * (t + 1)].float().sum(0) for t in range(mt)])
    np.testing.assert_allclose(ws1.cpu().numpy(), sums.cpu().numpy(), rtol=1e-5, atol=1e-4)
    # without column sums, and shapes it does not cover
    out2 = torch.empty_like(out1)
    lib.call("rbvae_fc_gemm", 1, A, W, out2, b, None, M, K, N, lda, N)
    assert torch.equal(out2, out0)
    assert not lib.query("rbvae_fc_gemm_ok", 0, M, K, N, lda, N) and not lib.query("rbvae_fc_gemm_ok", 1, M, 192, N, 192, N)
    with pytest.raises(ValueError):
        lib.call("rbvae_fc_gemm", 1, A, W, out2, b, None, M, K, N + 8, lda, N + 8)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("k,C,Co,N,H,W", [(3, 64, 64, 3, 16, 24), (3, 256, 256, 2, 8, 8), (4, 64, 128, 2, 16, 16),
                                           (3, 192, 128, 5, 10, 6)])
def test_conv_forward_and_dgrad(sfv, dtype, k, C, Co, N, H, W):
    from importlib import import_module
    E = import_module("symbols-from-video_amd.engine")
    dt, tdt, tol = DT[dtype]
    g = torch.Generator().manual_seed(1)
    x = torch.randn(N, C, H, W, generator=g).to(tdt).float()
    w = (torch.randn(Co, C, k, k, generator=g) / (C * k * k) ** 0.5).to(tdt).float()
    b = torch.randn(Co, generator=g)
    Ho, Wo = E.conv_out(H, k), E.conv_out(W, k)
    # forward: conv + bias + relu, scaled
    wf = w.permute(0, 2, 3, 1).contiguous().to(tdt).cuda()             # [co][kh][kw][ci]
    out = torch.empty(N * Ho * Wo, Co, dtype=tdt, device="cuda")
    gemm(sfv, dt, nhwc(x, tdt), wf, out, b.cuda(), None, None, (N, H, W, Ho, Wo, 2, Ho, Wo, 1), C, Co, k * k,
         E.conv_classes(k), 1, relu=1, scale=1.25)
    ref = F.relu(F.conv2d(x, w, b, stride=2, padding=1)) * 1.25
    assert rel(from_rows(out, N, Ho, Wo), ref) < tol
    # input gradient (= conv-transpose forward) with a gate
    dy = torch.randn(N, Co, Ho, Wo, generator=g).to(tdt).float()
    gate = torch.randn(N, C, 2 * Ho, 2 * Wo, generator=g)
    wd = w.permute(1, 2, 3, 0).contiguous().to(tdt).cuda()             # [ci][kh][kw][co]
    desc, ncls = E.dgrad_classes(k)
    dx = torch.full((N * 2 * Ho * 2 * Wo, C), 7.0, dtype=tdt, device="cuda")
    gemm(sfv, dt, nhwc(dy, tdt), wd, dx, None, nhwc(gate, tdt), None, (N, Ho, Wo, Ho, Wo, 1, 2 * Ho, 2 * Wo, 2), Co, C,
         k * k, desc, ncls, scale=0.5)
    op = 1 if k == 3 else 0
    ref = F.conv_transpose2d(dy, w, None, stride=2, padding=1, output_padding=op) * 0.5
    ref = ref * (gate.to(tdt).float() > 0)
    assert rel(from_rows(dx, N, 2 * Ho, 2 * Wo), ref) < tol


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_dropout_modes(sfv, dtype):
    dt, tdt, tol = DT[dtype]
    g = torch.Generator().manual_seed(2)
    M, K, Nout = 300, 128, 136
    A = torch.randn(M, K, generator=g).to(tdt)
    Wt = (torch.randn(Nout, K, generator=g) / K ** 0.5).to(tdt)
    ref = A.float() @ Wt.float().t()
    keep = (torch.rand(M, Nout, generator=g) > 0.2)
    out = torch.empty(M, Nout, dtype=tdt, device="cuda")
    gemm(sfv, dt, A.cuda(), Wt.cuda(), out, None, None, keep.to(torch.uint8).cuda(), (M, 1, 1, 1, 1, 1, 1, 1, 1), K,
         Nout, 1, [1, 0, 0, 0, 0, 0], 1, drop_mode=2, drop_p=0.2, scale=1.25)
    assert rel(out.float().cpu(), ref * keep * 1.25) < tol
    gemm(sfv, dt, A.cuda(), Wt.cuda(), out, None, None, None, (M, 1, 1, 1, 1, 1, 1, 1, 1), K, Nout, 1,
         [1, 0, 0, 0, 0, 0], 1, drop_mode=1, drop_p=0.2, scale=1.25, seed=11)
    o = out.float().cpu()
    dropped = (o == 0)
    assert 0.17 < dropped.float().mean().item() < 0.23
    assert rel(o[~dropped], (ref * 1.25)[~dropped]) < tol
    out2 = torch.empty_like(out)
    gemm(sfv, dt, A.cuda(), Wt.cuda(), out2, None, None, None, (M, 1, 1, 1, 1, 1, 1, 1, 1), K, Nout, 1,
         [1, 0, 0, 0, 0, 0], 1, drop_mode=1, drop_p=0.2, scale=1.25, seed=11)
    assert torch.equal(out, out2)                                       # same seed -> same mask, bitwise


def test_hash_dropout_statistics(sfv):
    """Counter-hash keep-masks (drop_mode 1): Bernoulli(0.8) per element, no row / column / neighbour
    structure, and unrelated masks for different seeds (nn.Dropout(0.2), percep_RBVAE_model.py:53)."""
    g = torch.Generator().manual_seed(5)
    M, K, Nout = 4096, 64, 256
    A = (torch.rand(M, K, generator=g) + 0.5).bfloat16()
    Wt = (torch.rand(Nout, K, generator=g) + 0.5).bfloat16()         # positive products: no accidental zeros
    masks = []
    for seed in (11, 12):
        out = torch.empty(M, Nout, dtype=torch.bfloat16, device="cuda")
        gemm(sfv, DT["bf16"][0], A.cuda(), Wt.cuda(), out, None, None, None, (M, 1, 1, 1, 1, 1, 1, 1, 1), K, Nout, 1,
             [1, 0, 0, 0, 0, 0], 1, drop_mode=1, drop_p=0.2, scale=1.25, seed=seed)
        masks.append((out.float().cpu() != 0))
    k0, k1 = masks[0].float(), masks[1].float()
    n = k0.numel()
    sig = (0.16 / n) ** 0.5
    assert abs(k0.mean().item() - 0.8) < 5 * sig and abs(k1.mean().item() - 0.8) < 5 * sig
    # rows (256 draws each) and columns (4096 draws each): worst deviation stays within ~5 sigma of its own scale
    assert (k0.mean(1) - 0.8).abs().max().item() < 5.5 * (0.16 / Nout) ** 0.5
    assert (k0.mean(0) - 0.8).abs().max().item() < 5.5 * (0.16 / M) ** 0.5
    # neighbours along a row, along a column, and the same element under another seed are uncorrelated
    def corr(a, b):
        a, b = a.flatten() - a.mean(), b.flatten() - b.mean()
        return (a * b).mean().item() / 0.16
    assert abs(corr(k0[:, :-1], k0[:, 1:])) < 0.01
    assert abs(corr(k0[:-1, :], k0[1:, :])) < 0.01
    assert abs(corr(k0[:, :-8], k0[:, 8:])) < 0.01                    # across 16-byte store chunks
    assert abs(corr(k0, k1)) < 0.01


def test_keyed_dropout_mask_is_its_definition(sfv):
    """The keep-mask of drop_mode 1 restated on the host (csrc/common.h: drop_key, drop_run, drop_bits, then a xorshift32
    walk with 16 bits per element over each 16-byte store chunk; dropped iff the 16 bits < floor(p * 2^32) >> 16) against
    the zeros the bf16 GEMM epilogue stores -- element for element, so the packed 16-bit form of the decision (v_pk_sub_u16
    clamp / v_pk_min_u16 / v_pk_mul_lo_u16) cannot drift from the compare form the f32 path and the mask kernels use."""
    g = torch.Generator().manual_seed(9)
    M, K, Nout, seed, pdrop = 777, 64, 136, 11, 0.2
    A = (torch.rand(M, K, generator=g) + 0.5).bfloat16()
    Wt = (torch.rand(Nout, K, generator=g) + 0.5).bfloat16()         # positive products: a zero is a dropped element
    out = torch.empty(M, Nout, dtype=torch.bfloat16, device="cuda")
    gemm(sfv, DT["bf16"][0], A.cuda(), Wt.cuda(), out, None, None, None, (M, 1, 1, 1, 1, 1, 1, 1, 1), K, Nout, 1,
         [1, 0, 0, 0, 0, 0], 1, drop_mode=1, drop_p=pdrop, scale=1.25, seed=seed)
    got_keep = (out.float().cpu() != 0).numpy()
    u64, u32 = np.uint64, np.uint32
    with np.errstate(over="ignore"):
        x = u64(seed) * u64(0x9E3779B97F4A7C15) + u64(0xD6E8FEB86659FD93)
        x ^= x >> u64(32); x *= u64(0xD6E8FEB86659FD93); x ^= x >> u64(32)
        k0, k1 = u32(x & u64(0xFFFFFFFF)), u32(x >> u64(32))
        thresh16 = u32(int(pdrop * 4294967296.0) >> 16)
        rows, chunks = np.arange(M, dtype=np.uint64)[:, None], np.arange(0, Nout, 8, dtype=np.uint64)[None, :]
        idx = rows * u64(Nout) + chunks                                # first element of every 8-element chunk
        s = (idx & u64(0xFFFFFFFF)).astype(u32) + k0 + (idx >> u64(32)).astype(u32) * k1
        s ^= s >> u32(16); s *= u32(0x7feb352d); s ^= s >> u32(15); s *= u32(0x846ca68b); s ^= s >> u32(16)
        want_keep = np.empty((M, Nout), dtype=bool)
        for e in range(0, 8, 2):
            want_keep[:, e::8] = (s & u32(0xffff)) >= thresh16
            want_keep[:, e + 1::8] = (s >> u32(16)) >= thresh16
            s ^= s << u32(13); s ^= s >> u32(17); s ^= s << u32(5)
    assert np.array_equal(got_keep, want_keep)
    assert 0.17 < 1.0 - want_keep.mean() < 0.23


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("k,C,Co,N,H,W,ks", [(3, 64, 64, 3, 16, 24, 1), (3, 256, 256, 2, 8, 8, 2), (4, 64, 128, 2, 16, 16, 3),
                                              (3, 128, 72, 5, 10, 6, 1)])
def test_conv_wgrad(sfv, dtype, k, C, Co, N, H, W, ks):
    from importlib import import_module
    E = import_module("symbols-from-video_amd.engine")
    dt, tdt, tol = DT[dtype]
    g = torch.Generator().manual_seed(3)
    Ho, Wo = E.conv_out(H, k), E.conv_out(W, k)
    x = torch.randn(N, C, H, W, generator=g).to(tdt).float().requires_grad_(False)
    dy = torch.randn(N, Co, Ho, Wo, generator=g).to(tdt).float()
    w = torch.zeros(Co, C, k, k, requires_grad=True)
    F.conv2d(x, w, None, stride=2, padding=1).backward(dy)
    P = N * Ho * Wo
    idx = torch.empty(k * k * P, dtype=torch.int32, device="cuda")
    sfv._lib.call("rbvae_conv_gather_index", idx, N, H, W, Ho, Wo, k, k, 2, 1)
    slabs = torch.full((ks, Co, k * k, C), 9.0, device="cuda")
    zero = torch.zeros(256, dtype=torch.uint8, device="cuda")
    sfv._lib.call("rbvae_wgrad_gemm", dt, nhwc(dy, tdt), nhwc(x, tdt), slabs, idx, zero, P, N * H * W, Co, C, Co, C, k * k, ks)
    out = torch.empty(Co, C, k * k, device="cuda")
    sfv._lib.call("rbvae_permute_reduce", slabs, ks, Co * k * k * C, out, Co, C, k * k, k * k * C, 1, C, 1.0, 0)
    assert rel(out.cpu().reshape(Co, C, k, k), w.grad) < tol


def test_conv_wgrad_bad_indices_read_the_zero_row(sfv):
    """A gather table holding out-of-range rows (stale, half-built, built for another shape) must not fault: such
    entries read the zero row, exactly like the table's own -1 padding entries."""
    from importlib import import_module
    E = import_module("symbols-from-video_amd.engine")
    k, C, Co, N, H, W, ks = 3, 64, 64, 2, 8, 8, 2
    g = torch.Generator().manual_seed(33)
    Ho, Wo = E.conv_out(H, k), E.conv_out(W, k)
    P = N * Ho * Wo
    x = torch.randn(N * H * W, C, generator=g).cuda()
    dy = torch.randn(P, Co, generator=g).cuda()
    idx = torch.empty(k * k * P, dtype=torch.int32, device="cuda")
    sfv._lib.call("rbvae_conv_gather_index", idx, N, H, W, Ho, Wo, k, k, 2, 1)
    bad = idx.clone()
    sel = torch.randperm(bad.numel(), generator=g)[:40].cuda()
    bad[sel[:20]] = 2_000_000_000
    bad[sel[20:30]] = N * H * W            # first row past the end
    bad[sel[30:]] = -12345
    good = idx.clone()
    good[sel] = -1
    zero = torch.zeros(256, dtype=torch.uint8, device="cuda")
    outs = []
    for table in (bad, good):
        slabs = torch.empty(ks, Co, k * k, C, device="cuda")
        sfv._lib.call("rbvae_wgrad_gemm", 0, dy, x, slabs, table, zero, P, N * H * W, Co, C, Co, C, k * k, ks)
        outs.append(slabs)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_skinny_linear_and_colsum(sfv, dtype):
    dt, tdt, tol = DT[dtype]
    g = torch.Generator().manual_seed(4)
    for (M, Nc, K) in ((37, 25, 4096), (256, 32, 1024), (6, 100, 64)):
        A = torch.randn(M, K, generator=g).to(tdt)
        B = (torch.randn(Nc, K, generator=g) / K ** 0.5).to(tdt)
        b = torch.randn(Nc, generator=g)
        out = torch.empty(M, Nc, device="cuda")
        sfv._lib.call("rbvae_skinny_linear", dt, A.cuda(), B.cuda(), b.cuda(), out, M, Nc, K, K, K, Nc)
        assert rel(out.cpu(), A.float() @ B.float().t() + b) < tol
    # (513, 4096), (130, 2056), (128, 56320): the 16-byte-per-thread form (wide tensors; ragged last column group, rows not a multiple
    # of four, the native 4x88x160 decoder fc gradient)
    for (P, C) in ((1000, 256), (77, 3), (513, 4096), (300, 25), (130, 2056), (128, 56320)):
        X = torch.randn(P, C, generator=g).to(tdt)
        out = torch.empty(C, device="cuda")
        ws = torch.empty(sfv._lib.query("rbvae_colsum_ws_floats", P, C), device="cuda")
        sfv._lib.call("rbvae_colsum", dt, X.cuda(), P, C, C, out, ws, 1.0, 0)
        assert rel(out.cpu(), X.float().sum(0)) < 1e-5


@pytest.mark.parametrize("L,layers,S,T", [(32, 4, 5, 8), (25, 2, 3, 5), (64, 2, 2, 3), (100, 1, 2, 4), (50, 4, 3, 8),
                                          (75, 2, 2, 9), (100, 4, 2, 17), (128, 2, 3, 5), (33, 3, 2, 2), (36, 1, 1, 1)])
@pytest.mark.parametrize("use_wT", [False, True])
def test_lstm_forward_backward(sfv, L, layers, S, T, use_wT):
    g = torch.Generator().manual_seed(5)
    names = []
    p = {}
    for l in range(layers):
        for nm, shp in (("weight_ih", (4 * L, L)), ("weight_hh", (4 * L, L)), ("bias_ih", (4 * L,)), ("bias_hh", (4 * L,))):
            t = ((torch.rand(shp, generator=g) * 2 - 1) / L ** 0.5).requires_grad_()
            p[f"r.lstm.{nm}_l{l}"] = t
            names.append(f"r.lstm.{nm}_l{l}")
    x = torch.randn(S, T, L, generator=g, requires_grad=True)
    gt = torch.randn(S, T, L, generator=g)
    y = O.lstm_stack(x, p, "r", layers)
    y.backward(gt)
    wblk = torch.cat([p[n].detach().reshape(-1) for n in names]).cuda()
    hs = torch.empty(layers + 1, S, T, L, device="cuda")
    hs[0] = x.detach().cuda()
    hp, cs, acts = torch.empty(layers, S, T, L, device="cuda"), torch.empty(layers, S, T, L, device="cuda"), \
        torch.empty(layers, S, T, 4 * L, device="cuda")
    wT_arg = None
    if use_wT:       # [layers][ih | hh][L][4L]: the transposed copies the engine keeps for coalesced weight loads
        wT_arg = torch.stack([torch.stack([p[f"r.lstm.{nm}_l{l}"].detach().t().contiguous() for nm in ("weight_ih", "weight_hh")])
                              for l in range(layers)]).cuda()
    sfv._lib.call("rbvae_lstm_fwd", wblk, wT_arg, hs, hp, acts, cs, S, T, L, layers)
    np.testing.assert_allclose(hs[layers].cpu().numpy(), y.detach().numpy(), atol=2e-6)
    dG, dx = torch.empty(layers, S, T, 4 * L, device="cuda"), torch.empty(S, T, L, device="cuda")
    sfv._lib.call("rbvae_lstm_bwd", wblk, wT_arg, acts, cs, gt.cuda(), dG, dx, S, T, L, layers)
    np.testing.assert_allclose(dx.cpu().numpy(), x.grad.numpy(), atol=2e-6, rtol=1e-4)
    gb = torch.empty_like(wblk)
    sfv._lib.call("rbvae_lstm_wgrad", dG, hs, hp, gb, S, T, L, layers, 0)
    ref = torch.cat([p[n].grad.reshape(-1) for n in names])
    assert rel(gb.cpu(), ref) < 1e-5
    # the two-stack launch (encoder + decoder LSTMs of one step) computes each stack exactly as the single one
    dG2, hs2, hp2 = dG.flip(1).contiguous(), hs.flip(1).contiguous(), hp.flip(1).contiguous()
    gb2, ga, gbb = torch.empty_like(wblk), torch.empty_like(wblk), torch.empty_like(wblk)
    sfv._lib.call("rbvae_lstm_wgrad", dG2, hs2, hp2, gb2, S, T, L, layers, 0)
    sfv._lib.call("rbvae_lstm_wgrad_pair", dG, hs, hp, ga, dG2, hs2, hp2, gbb, S, T, L, layers, 0)
    assert torch.equal(ga, gb) and torch.equal(gbb, gb2)


def test_im2col_col2im(sfv):
    g = torch.Generator().manual_seed(6)
    N, C, H, W, k = 2, 4, 8, 12, 3
    x = torch.randn(N, C, H, W, generator=g)
    Ho, Wo, Kp = 4, 6, 64
    col = torch.empty(N * Ho * Wo, Kp, device="cuda")
    sfv._lib.call("rbvae_im2col", 0, x.cuda(), C * H * W, H * W, W, 1, N, C, H, W, Ho, Wo, k, k, 2, 1, Kp, col)
    ref = F.unfold(x, k, padding=1, stride=2)                         # [N, C*k*k, Ho*Wo], (c, kh, kw) order
    ref = ref.reshape(N, C, k * k, Ho * Wo).permute(0, 3, 2, 1).reshape(N * Ho * Wo, k * k * C)
    np.testing.assert_array_equal(col.cpu()[:, :k * k * C].numpy(), ref.numpy())
    assert float(col[:, k * k * C:].abs().max()) == 0.0
    # col2im + bias + sigmoid + mse + dpre against conv_transpose2d
    Cin, Co = 16, 3
    a = torch.randn(N, Cin, Ho, Wo, generator=g)
    V = torch.randn(Cin, Co, k, k, generator=g) * 0.2
    b = torch.randn(Co, generator=g)
    tgt = torch.rand(N, Co, 2 * Ho, 2 * Wo, generator=g)
    Y = torch.einsum("nchw,cotk->nhwtok", a, V.reshape(Cin, Co, k * k, 1)).reshape(N * Ho * Wo, k * k * Co)
    NY = 32
    Yp = torch.zeros(N * Ho * Wo, NY)
    Yp[:, :k * k * Co] = Y
    xr = torch.empty(N, Co, 2 * Ho, 2 * Wo, device="cuda")
    mse = torch.empty(1, device="cuda")
    ws = torch.empty(sfv._lib.query("rbvae_col2im_ws_floats"), device="cuda")
    dpre = torch.empty(N, 2 * Ho, 2 * Wo, Co, device="cuda")
    sfv._lib.call("rbvae_col2im_sigmoid", 0, Yp.cuda(), NY, b.cuda(), N, Ho, Wo, 2 * Ho, 2 * Wo, Co, k, k, 1, xr,
                  tgt.cuda(), mse, ws, dpre, 0.5, None)
    ref = torch.sigmoid(F.conv_transpose2d(a, V, b, stride=2, padding=1, output_padding=1))
    np.testing.assert_allclose(xr.cpu().numpy(), ref.numpy(), atol=2e-6)
    assert abs(mse.item() - ((ref - tgt) ** 2).mean().item()) < 1e-6
    dref = 0.5 * (ref - tgt) * ref * (1 - ref)
    np.testing.assert_allclose(dpre.cpu().permute(0, 3, 1, 2).numpy(), dref.numpy(), atol=1e-6)


def test_adam_matches_torch(sfv):
    g = torch.Generator().manual_seed(7)
    n = 10007
    w = torch.randn(n, generator=g)
    p = torch.nn.Parameter(w.clone())
    opt = torch.optim.Adam([p], lr=1e-3)
    wd, m, v = w.clone().cuda(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    for step in range(1, 4):
        gr = torch.randn(n, generator=g) * 10 ** float(torch.randint(-6, 1, (1,), generator=g))
        p.grad = gr.clone()
        opt.step()
        sfv._lib.call("rbvae_adam_step", wd, gr.cuda(), m, v, n, 1e-3, 0.9, 0.999, 1e-8, step, 1.0, None, None)
        np.testing.assert_allclose(wd.cpu().numpy(), p.detach().numpy(), atol=2e-7, rtol=0)


def test_frame_mapped_im2col_col2im(sfv):
    """rbvae_im2col_frames / rbvae_col2im_sigmoid_frames read an item batch [B, 2, T, C, H, W] in place as the
    sequences "view 0 of every item, then view 1" -- the same numbers as the plain calls on the copied batch."""
    g = torch.Generator().manual_seed(16)
    B, T, C, H, W, k = 3, 2, 4, 8, 12, 3
    item = torch.rand(B, 2, T, C, H, W, generator=g).cuda()
    x = item.transpose(0, 1).reshape(2 * B * T, C, H, W).contiguous()      # what the reference feeds, view by view
    N, Ho, Wo, Kp = 2 * B * T, H // 2, W // 2, 64
    chw = C * H * W
    fm = (B * T, T, T * chw, 2 * T * chw, chw)
    col_ref = torch.empty(N * Ho * Wo, Kp, device="cuda")
    col = torch.empty_like(col_ref)
    sfv._lib.call("rbvae_im2col", 0, x, chw, H * W, W, 1, N, C, H, W, Ho, Wo, k, k, 2, 1, Kp, col_ref)
    sfv._lib.call("rbvae_im2col_frames", 0, item, *fm, H * W, W, 1, N, C, H, W, Ho, Wo, k, k, 2, 1, Kp, col)
    assert torch.equal(col, col_ref)
    NY = 40
    Y = torch.randn(N * Ho * Wo, NY, generator=g).cuda()
    bias = torch.randn(C, generator=g).cuda()
    outs = []
    for mapped in (False, True):
        xr = torch.empty(N, C, H, W, device="cuda")
        ws = torch.zeros(sfv._lib.query("rbvae_col2im_ws_floats"), device="cuda")
        mse = torch.empty(1, device="cuda")
        dpre = torch.empty(N, H, W, C, device="cuda")
        if mapped:
            sfv._lib.call("rbvae_col2im_sigmoid_frames", 0, Y, NY, bias, N, Ho, Wo, H, W, C, k, k, 1, xr, item, *fm, mse,
                          ws, dpre, 0.25, None)
        else:
            sfv._lib.call("rbvae_col2im_sigmoid", 0, Y, NY, bias, N, Ho, Wo, H, W, C, k, k, 1, xr, x, mse, ws, dpre, 0.25,
                          None)
        outs.append((xr, mse, dpre))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    assert abs(outs[0][1].item() - ((outs[0][0] - x) ** 2).mean().item()) < 1e-6


def test_binarize_parts_and_combine(sfv):
    """Many-workgroup binarise + per-block KL sums finished by rbvae_combine_losses == the one-workgroup kernel
    (same y / z bit for bit, KL mean to f32 rounding), and the step counter / Adam terms it prepares."""
    g = torch.Generator().manual_seed(17)
    rows, L = 200, 25
    h = torch.randn(rows, L, generator=g).cuda()
    U = torch.rand(rows, L, generator=g).cuda()
    y0, z0, y1, z1 = (torch.empty(rows, L, device="cuda") for _ in range(4))
    kl0 = torch.empty(1, device="cuda")
    sfv._lib.call("rbvae_binarize_kl_fwd", h, U, y0, z0, kl0, rows, L, 0.7, 0.1, 1e-8, 0, 0.1, 1e-8, 1, 0, None)
    nparts = sfv._lib.query("rbvae_binarize_kl_nparts", rows, L)
    assert nparts == -(-rows * L // 256)
    parts = torch.empty(nparts, device="cuda")
    sfv._lib.call("rbvae_binarize_kl_fwd_parts", h, U, y1, z1, parts, rows, L, 0.7, None, 0.1, 1e-8, 0, 0.1, 1e-8, 1, 0, None)
    assert torch.equal(y0, y1) and torch.equal(z0, z1)
    # temperature from a device float (by-value argument ignored): same bits
    y2, z2, parts2 = torch.empty_like(y1), torch.empty_like(z1), torch.empty_like(parts)
    tau_dev = torch.tensor([0.7], device="cuda")
    sfv._lib.call("rbvae_binarize_kl_fwd_parts", h, U, y2, z2, parts2, rows, L, 123.0, tau_dev, 0.1, 1e-8, 0, 0.1, 1e-8, 1, 0, None)
    assert torch.equal(y1, y2) and torch.equal(z1, z2) and torch.equal(parts, parts2)
    sse = torch.rand(37, generator=g).cuda()
    pair = torch.tensor([0.375], device="cuda")
    out4 = torch.empty(4, device="cuda")
    step = torch.tensor([4], dtype=torch.int64, device="cuda")
    hyper = torch.zeros(2, device="cuda")
    sfv._lib.call("rbvae_combine_losses", sse, 37, 1.0 / 1000, None, parts, nparts, 1.0 / rows, pair, 0, 0.0, 0.0, 0.5, 2.0,
                  out4, step, 1e-3, None, 0.9, 0.999, hyper)
    recon, kl = float(sse.sum()) / 1000, kl0.item()
    got = out4.cpu().tolist()
    assert abs(got[1] - recon) < 1e-6 and abs(got[2] - kl) < 1e-5 * max(1.0, abs(kl)) and got[3] == 0.375
    assert abs(got[0] - (recon + 0.5 * kl + 2.0 * 0.375)) < 1e-5 * max(1.0, abs(kl))
    assert int(step.item()) == 5
    hy = hyper.cpu().tolist()
    assert abs(hy[0] - 1e-3 / (1 - 0.9 ** 5)) < 1e-9 and abs(hy[1] - (1 - 0.999 ** 5) ** 0.5) < 1e-7
    # Adam with prepared terms == Adam told the step number
    n = 1000
    w = torch.randn(n, generator=g).cuda()
    gr = torch.randn(n, generator=g).cuda()
    wa, ma, va = w.clone(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    wb, mb, vb = w.clone(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    sfv._lib.call("rbvae_adam_step", wa, gr, ma, va, n, 1e-3, 0.9, 0.999, 1e-8, 5, 1.0, None, None)
    sfv._lib.call("rbvae_adam_step", wb, gr, mb, vb, n, 1e-3, 0.9, 0.999, 1e-8, 0, 1.0, None, hyper)
    np.testing.assert_allclose(wb.cpu().numpy(), wa.cpu().numpy(), atol=1e-7, rtol=0)


def test_split_fc_slabs_feed_the_lstm_kernels(sfv):
    """rbvae_skinny_linear_parts + rbvae_lstm_fwd_ex / _bwd_ex == the unsplit product fed to rbvae_lstm_fwd /
    _bwd (slab sums differ from the one-pass product only by f32 summation order)."""
    g = torch.Generator().manual_seed(19)
    S, T, L, layers, K, ks = 6, 8, 32, 4, 4096, 4
    N = S * T
    A = torch.randn(N, K, generator=g).bfloat16().cuda()
    Wt = (torch.randn(L, K, generator=g) / K ** 0.5).bfloat16().cuda()
    bias = torch.randn(L, generator=g).cuda()
    one = torch.empty(N, L, device="cuda")
    parts = torch.empty(ks, N, L, device="cuda")
    sfv._lib.call("rbvae_skinny_linear", 1, A, Wt, bias, one, N, L, K, K, K, L)
    sfv._lib.call("rbvae_skinny_linear_parts", 1, A, Wt, bias, parts, N, L, K, K, K, L, ks)
    np.testing.assert_allclose(parts.sum(0).cpu().numpy(), one.cpu().numpy(), atol=2e-5)
    wblk = (torch.randn(layers * (8 * L * L + 8 * L), generator=g) * 0.2).cuda()

    def fwd(use_parts):
        hs = torch.zeros(layers + 1, S, T, L, device="cuda")
        hp, cs = torch.empty(layers, S, T, L, device="cuda"), torch.empty(layers, S, T, L, device="cuda")
        acts = torch.empty(layers, S, T, 4 * L, device="cuda")
        if use_parts:
            pad = torch.full((N, 64), 9.0, dtype=torch.bfloat16, device="cuda")
            sfv._lib.call("rbvae_lstm_fwd_ex", wblk, None, hs, hp, acts, cs, S, T, L, layers, parts, ks, N * L, pad, 1, 64)
            ref = torch.zeros(N, 64, dtype=torch.bfloat16, device="cuda")
            sfv._lib.call("rbvae_cast_pad", 1, hs[layers].contiguous(), ref, N, L, 64)
            assert torch.equal(pad, ref)                       # the cast copy == rbvae_cast_pad of the top layer
        else:
            hs[0] = parts.sum(0).view(S, T, L)
            sfv._lib.call("rbvae_lstm_fwd", wblk, None, hs, hp, acts, cs, S, T, L, layers)
        return hs, acts, cs
    (h0, a0, c0), (h1, a1, c1) = fwd(False), fwd(True)
    np.testing.assert_allclose(h1.cpu().numpy(), h0.cpu().numpy(), atol=1e-6)
    gparts = torch.randn(ks, N, L, generator=g).cuda()
    outs = []
    for use_parts in (False, True):
        dG, dx = torch.empty(layers, S, T, 4 * L, device="cuda"), torch.empty(N, L, device="cuda")
        if use_parts:
            pad = torch.full((N, 32), 9.0, device="cuda")
            sums = torch.empty(S, L, device="cuda")
            sfv._lib.call("rbvae_lstm_bwd_ex", wblk, a0, c0, gparts, ks, N * L, dG, dx, pad, 0, 32, sums, S, T, L, layers)
            assert torch.equal(pad, dx)
            np.testing.assert_allclose(sums.cpu().numpy(), dx.view(S, T, L).sum(1).cpu().numpy(), atol=1e-5)
        else:
            sfv._lib.call("rbvae_lstm_bwd", wblk, None, a0, c0, gparts.sum(0).contiguous(), dG, dx, S, T, L, layers)
        outs.append((dG, dx))
    np.testing.assert_allclose(outs[1][0].cpu().numpy(), outs[0][0].cpu().numpy(), atol=1e-5)
    np.testing.assert_allclose(outs[1][1].cpu().numpy(), outs[0][1].cpu().numpy(), atol=1e-5)
    # a latent size the wavefront kernel does not cover is refused, not silently mis-summed
    with pytest.raises(ValueError):
        hs = torch.zeros(2, 2, 3, 64, device="cuda")
        sfv._lib.call("rbvae_lstm_fwd_ex", torch.zeros(8 * 64 * 64 + 8 * 64, device="cuda"), None, hs, None, None, None, 2, 3,
                      64, 1, torch.zeros(2, 6, 64, device="cuda"), 2, 6 * 64, None, 0, 0)


@pytest.mark.parametrize("L,layers,S,T,hard", [(32, 4, 5, 8, 0), (32, 2, 3, 5, 1), (24, 2, 4, 9, 0), (25, 4, 3, 8, 0), (25, 2, 2, 5, 1),
                                               (7, 2, 2, 3, 0), (30, 4, 2, 17, 0)])
def test_lstm_pair_forward_equals_three_launches(sfv, L, layers, S, T, hard):
    """rbvae_lstm_pair_fwd (encoder stack -> binarise -> decoder stack, one wavefront) against rbvae_lstm_fwd +
    rbvae_binarize_kl_fwd_parts + rbvae_lstm_fwd: every saved tensor to f32 rounding (2e-6), hard codes equal
    except where y sits within rounding of 0.5."""
    assert sfv._lib.query("rbvae_lstm_pair_fwd_ok", T, L, layers)
    g = torch.Generator().manual_seed(40 + L + layers)
    N = S * T
    per = layers * (8 * L * L + 8 * L)
    we, wd = (torch.randn(per, generator=g) * 0.3).cuda(), (torch.randn(per, generator=g) * 0.3).cuda()
    x = torch.randn(S, T, L, generator=g).cuda()
    U = torch.rand(N, L, generator=g).cuda()
    tau, r, p = 0.6, 0.3, 0.1

    def bufs():
        hs = torch.zeros(layers + 1, S, T, L, device="cuda")
        return (hs, torch.empty(layers, S, T, L, device="cuda"), torch.empty(layers, S, T, 4 * L, device="cuda"),
                torch.empty(layers, S, T, L, device="cuda"))
    # three launches
    he, pe, ae, ce = bufs(); hd, pd, ad, cd = bufs()
    he[0] = x
    y0 = torch.empty(N, L, device="cuda")
    nkl = sfv._lib.query("rbvae_binarize_kl_nparts", N, L)
    parts0 = torch.empty(nkl, device="cuda")
    sfv._lib.call("rbvae_lstm_fwd", we, None, he, pe, ae, ce, S, T, L, layers)
    sfv._lib.call("rbvae_binarize_kl_fwd_parts", he[layers].contiguous(), U, y0, hd[0], parts0, N, L, tau, None, r, 1e-8, hard, p,
                  1e-8, 1, 0, None)
    pad0 = torch.zeros(N, 64, dtype=torch.bfloat16, device="cuda")
    sfv._lib.call("rbvae_lstm_fwd_ex", wd, None, hd, pd, ad, cd, S, T, L, layers, None, 1, 0, pad0, 1, 64)
    # one launch
    he1, pe1, ae1, ce1 = bufs(); hd1, pd1, ad1, cd1 = bufs()
    he1[0] = x
    y1 = torch.empty(N, L, device="cuda")
    parts1 = torch.empty(S, device="cuda")
    pad1 = torch.full((N, 64), 3.0, dtype=torch.bfloat16, device="cuda")
    sfv._lib.call("rbvae_lstm_pair_fwd", we, None, wd, None, he1, pe1, ae1, ce1, hd1, pd1, ad1, cd1, None, 1, 0, U, y1, parts1,
                  123.0, torch.tensor([tau], device="cuda"), r, 1e-8, hard, p, 1e-8, 1, 0, None, pad1, 1, 64, S, T, L, layers)
    names = "hs_enc hprev_enc acts_enc cs_enc hs_dec hprev_dec acts_dec cs_dec y_soft cast".split()
    pairs = ((he, he1), (pe, pe1), (ae, ae1), (ce, ce1), (hd, hd1), (pd, pd1), (ad, ad1), (cd, cd1), (y0, y1), (pad0, pad1))
    for nm, (a, b) in zip(names, pairs):
        if hard and nm in ("hs_dec", "hprev_dec", "acts_dec", "cs_dec", "cast"):
            continue                    # downstream of the hard codes: compared through the codes below
        d = (a.float() - b.float()).abs().max().item()
        assert d <= (1e-2 if nm == "cast" else 2e-6), (nm, d)
    if hard:
        assert (hd[0] != hd1[0]).float().mean().item() < 2e-3        # codes: only |y - 0.5| ~ 1e-7 cases may differ
    assert abs(parts0.sum().item() - parts1.sum().item()) < 1e-4 * max(1.0, abs(parts0.sum().item()))
    # shapes outside the fused kernel are refused
    assert not sfv._lib.query("rbvae_lstm_pair_fwd_ok", 8, 64, 2)


@pytest.mark.parametrize("layers,S,T,hard,use_wT,nparts", [(4, 5, 8, 0, True, 3), (2, 3, 5, 1, False, 1), (4, 2, 1, 0, True, 1), (3, 4, 11, 0, False, 5)])
def test_lstm_pair_forward_unit_threads_bit_identical(sfv, layers, S, T, hard, use_wT, nparts):
    """lstm_pair_fwd_unit_k (L == 32: one thread per hidden unit holds its four gate rows, one barrier per diagonal)
    against lstm_pair_fwd_k (one thread per gate row), selected through rbvae_dbg_lstm_unit_threads: the same
    accumulator chains and expressions, so EVERY output is bit for bit the same -- saved activations, cell states, codes,
    the bf16 cast of the decoder's top layer, the per-sequence KL sums -- also from K-split input slabs, with device noise."""
    L = 32
    g = torch.Generator().manual_seed(90 + layers + T)
    N = S * T
    per = layers * (8 * L * L + 8 * L)
    we, wd = (torch.randn(per, generator=g) * 0.3).cuda(), (torch.randn(per, generator=g) * 0.3).cuda()

    def wT_of(w):
        out = torch.empty(layers, 2, L, 4 * L, device="cuda")
        for l in range(layers):
            blk = w[l * (8 * L * L + 8 * L):]
            out[l, 0] = blk[:4 * L * L].view(4 * L, L).t()
            out[l, 1] = blk[4 * L * L:8 * L * L].view(4 * L, L).t()
        return out.contiguous()
    wTe, wTd = (wT_of(we), wT_of(wd)) if use_wT else (None, None)
    slabs = torch.randn(nparts, S, T, L, generator=g).cuda()
    U = torch.rand(N, L, generator=g).cuda() if hard else None           # None: counter-hash noise from (seed, step)
    seed_dev = torch.tensor([7], dtype=torch.int64, device="cuda")
    dbg = sfv._lib.dbg_lib()
    outs = []
    for unit in (1, 0):
        old = dbg.rbvae_dbg_lstm_unit_threads(unit)
        try:
            bufs = [torch.zeros(layers + 1, S, T, L, device="cuda"), torch.empty(layers, S, T, L, device="cuda"),
                    torch.empty(layers, S, T, 4 * L, device="cuda"), torch.empty(layers, S, T, L, device="cuda")]
            bufs += [torch.zeros(layers + 1, S, T, L, device="cuda"), torch.empty(layers, S, T, L, device="cuda"),
                     torch.empty(layers, S, T, 4 * L, device="cuda"), torch.empty(layers, S, T, L, device="cuda")]
            y = torch.empty(N, L, device="cuda")
            parts = torch.empty(S, device="cuda")
            pad = torch.full((N, 64), 3.0, dtype=torch.bfloat16, device="cuda")
            sfv._lib.call("rbvae_lstm_pair_fwd", we, wTe, wd, wTd, *bufs, slabs, nparts, S * T * L, U, y, parts, 0.6, None, 0.3,
                          1e-8, hard, 0.1, 1e-8, 1, 1234, seed_dev, pad, 1, 64, S, T, L, layers)
            outs.append(bufs + [y, parts, pad])
        finally:
            dbg.rbvae_dbg_lstm_unit_threads(old)
    for a, b in zip(*outs):
        assert torch.equal(a.view(torch.int16) if a.dtype == torch.bfloat16 else a.view(torch.int32),
                           b.view(torch.int16) if b.dtype == torch.bfloat16 else b.view(torch.int32))
    assert torch.isfinite(outs[0][8]).all() and float(outs[0][4][layers].abs().max()) > 0


@pytest.mark.parametrize("L,layers,S,T,nparts,klw,ghs,extra", [(32, 4, 5, 8, 4, 1.0, True, False), (32, 2, 3, 5, 1, 0.0, False, True),
                                                              (25, 4, 3, 8, 3, 0.5, True, True), (7, 2, 2, 3, 1, 1.0, False, False),
                                                              (30, 3, 2, 9, 2, 0.0, True, False),
                                                              # L == 32: lstm_pair_bwd_unit_k (one wave per layer)
                                                              (32, 3, 4, 11, 5, 0.5, True, True), (32, 4, 2, 1, 1, 1.0, False, False),
                                                              (32, 1, 3, 4, 2, 0.0, False, True)])
def test_lstm_pair_backward_equals_two_launches(sfv, L, layers, S, T, nparts, klw, ghs, extra):
    """rbvae_lstm_pair_bwd (decoder stack -> binarise backward -> encoder stack, one wavefront) against
    rbvae_lstm_bwd_ex(decoder) + rbvae_lstm_bwd_bin(encoder): gate gradients of both stacks, the input gradient, its cast
    copy and per-sequence sums, and the codes' gradient -- the same expressions in the same order, so bit for bit."""
    lib = sfv._lib
    assert lib.query("rbvae_lstm_pair_bwd_ok", T, L, layers)
    g = torch.Generator().manual_seed(70 + L + layers)
    N = S * T
    per = layers * (8 * L * L + 8 * L)
    we, wd = (torch.randn(per, generator=g) * 0.3).cuda(), (torch.randn(per, generator=g) * 0.3).cuda()

    def saved():        # gates in (0, 1) / (-1, 1) like a forward pass leaves them, cell states of moderate size
        a = torch.rand(layers, S, T, 4 * L, generator=g)
        a[..., 2 * L:3 * L] = a[..., 2 * L:3 * L] * 2 - 1
        return a.cuda(), torch.randn(layers, S, T, L, generator=g).cuda()
    ae, ce = saved(); ad, cd = saved()
    gparts = torch.randn(nparts, N, L, generator=g).cuda()
    y = torch.rand(N, L, generator=g).clamp(1e-3, 1 - 1e-3).cuda()
    z = torch.rand(N, L, generator=g).clamp(1e-3, 1 - 1e-3).cuda()
    g_hs = torch.randn(N, L, generator=g).cuda() if ghs else None
    gzx = torch.randn(N, L, generator=g).cuda() if extra else None
    tau = torch.tensor([0.7], device="cuda")
    Lp = 64
    # two launches
    dGd0, dz0 = torch.empty(layers, S, T, 4 * L, device="cuda"), torch.empty(N, L, device="cuda")
    lib.call("rbvae_lstm_bwd_ex", wd, ad, cd, gparts, nparts, N * L, dGd0, dz0, None, 0, 0, None, S, T, L, layers)
    gz = dz0 if gzx is None else dz0 + gzx
    dGe0, dx0 = torch.empty(layers, S, T, 4 * L, device="cuda"), torch.empty(N, L, device="cuda")
    pad0, sums0 = torch.full((N, Lp), 3.0, dtype=torch.bfloat16, device="cuda"), torch.empty(S, L, device="cuda")
    lib.call("rbvae_lstm_bwd_bin", we, ae, ce, gz, y, z, g_hs, 9.0, tau, klw, 0.1, 1e-8, 1, dGe0, dx0, pad0, 1, Lp, sums0,
             S, T, L, layers)
    # one launch
    dGd1, dz1 = torch.empty_like(dGd0), torch.empty_like(dz0)
    dGe1, dx1 = torch.empty_like(dGe0), torch.empty_like(dx0)
    pad1, sums1 = torch.full((N, Lp), 5.0, dtype=torch.bfloat16, device="cuda"), torch.empty(S, L, device="cuda")
    lib.call("rbvae_lstm_pair_bwd", we, wd, ae, ce, ad, cd, gparts, nparts, N * L, gzx, y, z, g_hs, 9.0, tau, klw, 0.1, 1e-8, 1,
             dGe1, dGd1, dx1, dz1, pad1, 1, Lp, sums1, S, T, L, layers)
    torch.cuda.synchronize()
    for nm, a, b in (("dG_dec", dGd0, dGd1), ("dz", dz0, dz1), ("dG_enc", dGe0, dGe1), ("dx", dx0, dx1), ("cast", pad0, pad1),
                     ("sums", sums0, sums1)):
        assert torch.equal(a, b), (nm, (a.float() - b.float()).abs().max().item())
    # dz is optional
    dx2 = torch.empty_like(dx0)
    lib.call("rbvae_lstm_pair_bwd", we, wd, ae, ce, ad, cd, gparts, nparts, N * L, gzx, y, z, g_hs, 9.0, tau, klw, 0.1, 1e-8, 1,
             torch.empty_like(dGe0), torch.empty_like(dGd0), dx2, None, None, 0, 0, None, S, T, L, layers)
    assert torch.equal(dx2, dx0)
    assert not lib.query("rbvae_lstm_pair_bwd_ok", 8, 64, 2) and not lib.query("rbvae_lstm_pair_bwd_ok", 64, 32, 4)


@pytest.mark.parametrize("N,IH,IW,C1,Cout", [(3, 16, 16, 256, 4), (2, 11, 20, 256, 4), (2, 8, 8, 64, 3), (1, 5, 37, 128, 4)])
def test_deconv_last_fused(sfv, N, IH, IW, C1, Cout):
    """rbvae_deconv_last_fused (per-tap products on the MFMA from an LDS-resident pixel block + gather + sigmoid +
    squared error + d(loss)/d(pre) + their sums) against F.conv_transpose2d on the bf16-rounded operands, incl.
    blocks that stick out of the image (11x20, 5x37) and a frame-mapped target."""
    L = sfv._lib
    g = torch.Generator().manual_seed(60 + IH + C1)
    a = (torch.randn(N, C1, IH, IW, generator=g) * 0.5).bfloat16()
    V = (torch.randn(C1, Cout, 3, 3, generator=g) * 0.1).bfloat16()
    b = torch.randn(Cout, generator=g)
    OH, OW = 2 * IH, 2 * IW
    tgt = torch.rand(N, Cout, OH, OW, generator=g)
    ref = torch.sigmoid(F.conv_transpose2d(a.float(), V.float(), b, stride=2, padding=1, output_padding=1))
    parts = L.query("rbvae_deconv_last_fused_parts", 1, N, IH, IW, C1, Cout)
    assert parts == N * -(-IH // 8) * -(-IW // 16)
    NY = -(-9 * Cout // 8) * 8
    Vp = torch.zeros(NY, C1, dtype=torch.bfloat16)
    Vp[:9 * Cout] = V.float().permute(2, 3, 1, 0).reshape(9 * Cout, C1).bfloat16()        # row = (kh*3+kw)*Cout + co
    rows = a.permute(0, 2, 3, 1).reshape(N * IH * IW, C1).contiguous().cuda()
    zero = torch.zeros(256, dtype=torch.uint8, device="cuda")
    xr = torch.empty(N, Cout, OH, OW, device="cuda")
    ws = torch.zeros(5 * parts, device="cuda")
    dpre = torch.empty(N, OH, OW, Cout, device="cuda")
    L.call("rbvae_deconv_last_fused", 1, rows, Vp.cuda(), NY, b.cuda(), zero, N, IH, IW, C1, Cout, xr, tgt.cuda(), 0, 0, 0,
           0, Cout * OH * OW, ws, dpre, 0.5)
    np.testing.assert_allclose(xr.cpu().numpy(), ref.numpy(), atol=2e-5)
    assert abs(ws[:parts].sum().item() / tgt.numel() - ((ref - tgt) ** 2).mean().item()) < 1e-5
    dref = 0.5 * (ref - tgt) * ref * (1 - ref)
    np.testing.assert_allclose(dpre.cpu().permute(0, 3, 1, 2).numpy(), dref.numpy(), atol=1e-5)
    np.testing.assert_allclose(ws[parts:].view(parts, 4).sum(0)[:Cout].cpu().numpy(), dref.sum((0, 2, 3)).numpy(), rtol=1e-4,
                               atol=1e-4)
    # without a target: x_recon only
    xr2 = torch.empty_like(xr)
    L.call("rbvae_deconv_last_fused", 1, rows, Vp.cuda(), NY, b.cuda(), zero, N, IH, IW, C1, Cout, xr2, None, 0, 0, 0, 0,
           Cout * OH * OW, None, None, 0.0)
    assert torch.equal(xr2, xr)
    assert L.query("rbvae_deconv_last_fused_parts", 0, N, IH, IW, C1, Cout) == 0            # f32 mode: two-kernel path


@pytest.mark.parametrize("N,Cin,IH,IW,Nout,drop", [(3, 4, 32, 32, 256, 0.1), (2, 3, 21, 40, 64, 0.0), (2, 4, 7, 70, 200, 0.3),
                                                    (5, 1, 16, 16, 32, 0.1)])
def test_conv_first_fused(sfv, N, Cin, IH, IW, Nout, drop):
    """rbvae_conv_first_fused (patch gather in LDS + MFMA + bias/ReLU/dropout from the accumulators) is bit-identical
    to rbvae_im2col_frames + the single-slice rbvae_gather_gemm it replaces -- col rows, outputs and dropout pattern
    (same key and element indices) -- incl. blocks that stick out of the image, odd sizes, Nout % 32 != 0 and a
    frame-mapped input; and agrees with F.conv2d on the bf16-rounded operands."""
    L = sfv._lib
    g = torch.Generator().manual_seed(70 + IH + Nout)
    OH, OW = (IH - 1) // 2 + 1, (IW - 1) // 2 + 1
    P = N * OH * OW
    x = torch.randn(N, Cin, IH, IW, generator=g).cuda()
    Wt = (torch.randn(Nout, Cin, 3, 3, generator=g) * 0.2)
    b = torch.randn(Nout, generator=g).cuda()
    Wp = torch.zeros(Nout, 64, dtype=torch.bfloat16)
    Wp[:, :9 * Cin] = Wt.permute(0, 2, 3, 1).reshape(Nout, 9 * Cin).bfloat16()               # column = (kh*3+kw)*Cin + ci
    Wp = Wp.cuda()
    zero = torch.zeros(256, dtype=torch.uint8, device="cuda")
    scale = 1.0 / (1.0 - drop)
    mode = 1 if drop > 0 else 0
    assert L.query("rbvae_conv_first_fused_ok", 1, Cin, IH, IW, Nout, N) == 1
    assert L.query("rbvae_conv_first_fused_ok", 0, Cin, IH, IW, Nout, N) == 0                  # f32 mode: two-kernel path
    assert L.query("rbvae_conv_first_fused_ok", 1, 8, IH, IW, Nout, N) == 0
    col_a = torch.empty(P, 64, dtype=torch.bfloat16, device="cuda")
    out_a = torch.empty(P, Nout, dtype=torch.bfloat16, device="cuda")
    L.call("rbvae_im2col_frames", 1, x, 0, 0, 0, 0, Cin * IH * IW, IH * IW, IW, 1, N, Cin, IH, IW, OH, OW, 3, 3, 2, 1, 64, col_a)
    gemm(sfv, 1, col_a, Wp, out_a, b, None, None, (P, 1, 1, 1, 1, 1, 1, 1, 1), 64, Nout, 1, [1, 0, 0, 0, 0, 0], 1, relu=1,
         drop_mode=mode, drop_p=float(drop), scale=float(scale), seed=77)
    col_b = torch.empty_like(col_a)
    out_b = torch.empty_like(out_a)
    L.call("rbvae_conv_first_fused", 1, x, 0, 0, 0, 0, Cin * IH * IW, Wp, b, zero, col_b, out_b, N, Cin, IH, IW, Nout, Nout, 1,
           mode, float(drop), float(scale), 77, None)
    assert torch.equal(col_a.view(torch.int16), col_b.view(torch.int16))
    assert torch.equal(out_a.view(torch.int16), out_b.view(torch.int16))
    ref = F.relu(F.conv2d(x.cpu().bfloat16().float(), Wt.bfloat16().float(), b.cpu(), stride=2, padding=1)) * scale
    got = out_b.float().cpu().view(N, OH, OW, Nout).permute(0, 3, 1, 2)
    kept = got != 0
    np.testing.assert_allclose(got[kept].numpy(), ref[kept].numpy(), rtol=1e-2, atol=1e-2)
    if drop > 0:
        frac = 1.0 - kept.sum().item() / (ref != 0).sum().item()
        assert abs(frac - drop) < 0.02
    # frame-mapped input: frame n = (s, t) of an item buffer [S][2][T] taken at view 1 -> s*s0 + t*s2 from the view's base
    S, T = 2, N
    buf = torch.randn(S, 2, T, Cin, IH, IW, generator=g).cuda()
    fsz = Cin * IH * IW
    dense = buf[:, 1].contiguous()
    L.call("rbvae_conv_first_fused", 1, dense, 0, 0, 0, 0, fsz, Wp, b, zero, col_a[:0].new_empty(S * T * OH * OW, 64),
           (o1 := torch.empty(S * T * OH * OW, Nout, dtype=torch.bfloat16, device="cuda")), S * T, Cin, IH, IW, Nout, Nout, 1, 0,
           0.0, 1.0, 0, None)
    L.call("rbvae_conv_first_fused", 1, buf[:, 1], T, T, 2 * T * fsz, 0, fsz, Wp, b, zero,
           col_a[:0].new_empty(S * T * OH * OW, 64),
           (o2 := torch.empty(S * T * OH * OW, Nout, dtype=torch.bfloat16, device="cuda")), S * T, Cin, IH, IW, Nout, Nout, 1, 0,
           0.0, 1.0, 0, None)
    assert torch.equal(o1.view(torch.int16), o2.view(torch.int16))


@pytest.mark.parametrize("N,Cout,OH,OW,C1", [(3, 4, 32, 32, 256), (2, 3, 21, 40, 64), (2, 4, 7, 70, 200), (3, 3, 64, 48, 40)])
def test_deconv_last_dgrad_fused(sfv, N, Cout, OH, OW, C1):
    """rbvae_deconv_last_dgrad_fused (MODE 1 of csrc/conv_first.hip) stores exactly what rbvae_im2col + the gated
    single-slice rbvae_gather_gemm store (col rows and outputs bit-identical); its per-workgroup column sums add up to
    the column sums of the stored output; and it agrees with autograd of F.conv_transpose2d on bf16-rounded operands."""
    L = sfv._lib
    g = torch.Generator().manual_seed(90 + OH + C1)
    IH, IW = (OH - 1) // 2 + 1, (OW - 1) // 2 + 1
    P = N * IH * IW
    dpre = (torch.randn(N, OH, OW, Cout, generator=g) * 0.1).cuda()
    V = torch.randn(C1, Cout, 3, 3, generator=g) * 0.2                       # ConvTranspose2d weight [in=C1][out=Cout][kh][kw]
    Wp = torch.zeros(C1, 64, dtype=torch.bfloat16)
    Wp[:, :9 * Cout] = V.permute(0, 2, 3, 1).reshape(C1, 9 * Cout).bfloat16()
    Wp = Wp.cuda()
    gate = torch.relu(torch.randn(P, C1, generator=g)).bfloat16().cuda()      # ~half zeros
    zero = torch.zeros(256, dtype=torch.uint8, device="cuda")
    col_a = torch.empty(P, 64, dtype=torch.bfloat16, device="cuda")
    out_a = torch.empty(P, C1, dtype=torch.bfloat16, device="cuda")
    L.call("rbvae_im2col", 1, dpre, OH * OW * Cout, 1, OW * Cout, Cout, N, Cout, OH, OW, IH, IW, 3, 3, 2, 1, 64, col_a)
    gemm(sfv, 1, col_a, Wp, out_a, None, gate, None, (P, 1, 1, 1, 1, 1, 1, 1, 1), 64, C1, 1, [1, 0, 0, 0, 0, 0], 1, scale=1.25)
    nb = L.query("rbvae_deconv_last_dgrad_blocks", 1, Cout, OH, OW, C1, N)
    assert nb == N * -(-IH // 8) * -(-IW // 16)
    assert L.query("rbvae_deconv_last_dgrad_blocks", 0, Cout, OH, OW, C1, N) == 0
    col_b = torch.empty_like(col_a)
    out_b = torch.empty_like(out_a)
    ws = torch.full((nb, C1), float("nan"), device="cuda")
    L.call("rbvae_deconv_last_dgrad_fused", 1, dpre, Wp, zero, col_b, gate, out_b, N, Cout, OH, OW, C1, C1, 1.25, ws)
    assert torch.equal(col_a.view(torch.int16), col_b.view(torch.int16))
    assert torch.equal(out_a.view(torch.int16), out_b.view(torch.int16))
    np.testing.assert_allclose(ws.sum(0).cpu().numpy(), out_b.float().sum(0).cpu().numpy(), rtol=2e-4, atol=2e-3)
    # autograd: d/d(input) of conv_transpose2d(input, V) contracted with dpre = conv2d(dpre, V as [C1][Cout][3][3], s2 p1)
    ref = F.conv2d(dpre.cpu().permute(0, 3, 1, 2).bfloat16().float(), V.bfloat16().float(), stride=2, padding=1) * 1.25
    ref = ref.permute(0, 2, 3, 1).reshape(P, C1) * (gate.float().cpu() > 0)
    np.testing.assert_allclose(out_b.float().cpu().numpy(), ref.numpy(), rtol=1e-2, atol=2e-3)
    # without the column sums
    out_c = torch.empty_like(out_a)
    L.call("rbvae_deconv_last_dgrad_fused", 1, dpre, Wp, zero, col_b, gate, out_c, N, Cout, OH, OW, C1, C1, 1.25, None)
    assert torch.equal(out_c.view(torch.int16), out_b.view(torch.int16))


@pytest.mark.parametrize("N,Cin,IH,IW,Nout,ks", [(3, 4, 32, 32, 256, 2), (2, 3, 21, 40, 64, 3), (4, 3, 64, 64, 64, 7),
                                                  (2, 4, 7, 70, 128, 1), (5, 1, 16, 16, 64, 5),
                                                  # Nout % 256 == 0: wgrad_first_wide_k (a workgroup takes every channel)
                                                  (2, 4, 21, 40, 256, 3), (1, 3, 16, 32, 256, 1), (3, 3, 30, 34, 512, 4),
                                                  (7, 1, 18, 18, 256, 7), (4, 4, 44, 80, 256, 6)])
def test_wgrad_first_rebuilds_the_im2col_rows(sfv, N, Cin, IH, IW, Nout, ks):
    """rbvae_wgrad_first (csrc/conv_first.hip): the weight gradients of the first Conv2d (mode 0: frames, also through a
    frame map) and of the last ConvTranspose2d (mode 1: NHWC f32 image) from the 3/4-channel image itself, against
    rbvae_wgrad_gemm over the [rows][64] im2col rows the fused forward kernels write -- the same bf16 products, f32 sums
    in another order -- and against autograd of F.conv2d.  With it the forward kernels take col = NULL and store the same
    outputs."""
    L = sfv._lib
    g = torch.Generator().manual_seed(170 + IH + Nout)
    OH, OW = (IH - 1) // 2 + 1, (IW - 1) // 2 + 1
    P = N * OH * OW
    zero = torch.zeros(256, dtype=torch.uint8, device="cuda")
    nblk = L.query("rbvae_wgrad_first_blocks", 1, Cin, IH, IW, Nout, N)
    assert nblk == N * -(-OH // 8) * -(-OW // 16)
    assert L.query("rbvae_wgrad_first_blocks", 0, Cin, IH, IW, Nout, N) == 0
    assert L.query("rbvae_wgrad_first_blocks", 1, Cin, IH, IW, Nout + 8, N) == 0
    ks = min(ks, nblk)
    dy = (torch.randn(P, Nout, generator=g) / 8).bfloat16().cuda()
    Nf = min(Nout, 256)                         # the forward kernels (which write the im2col rows) stop at 256 channels
    Wp = torch.zeros(Nf, 64, dtype=torch.bfloat16, device="cuda")
    b = torch.zeros(Nf, device="cuda")

    def via_col(col):
        k2 = max(1, -(-P // 4096))
        slabs = torch.empty(k2, Nout, 64, device="cuda")
        L.call("rbvae_wgrad_gemm", 1, dy, col, slabs, None, zero, P, P, Nout, 64, Nout, 64, 1, k2)
        return slabs.sum(0)

    # mode 0: frames
    x = torch.randn(N, Cin, IH, IW, generator=g).cuda()
    col = torch.empty(P, 64, dtype=torch.bfloat16, device="cuda")
    out_a = torch.empty(P, Nf, dtype=torch.bfloat16, device="cuda")
    L.call("rbvae_conv_first_fused", 1, x, 0, 0, 0, 0, Cin * IH * IW, Wp, b, zero, col, out_a, N, Cin, IH, IW, Nf, Nf, 1, 0,
           0.0, 1.0, 0, None)
    out_b = torch.full_like(out_a, float("nan"))
    L.call("rbvae_conv_first_fused", 1, x, 0, 0, 0, 0, Cin * IH * IW, Wp, b, zero, None, out_b, N, Cin, IH, IW, Nf, Nf, 1, 0,
           0.0, 1.0, 0, None)
    assert torch.equal(out_a.view(torch.int16), out_b.view(torch.int16))
    want = via_col(col)
    slabs = torch.full((ks, Nout, 64), float("nan"), device="cuda")
    L.call("rbvae_wgrad_first", 1, 0, x, 0, 0, 0, 0, Cin * IH * IW, dy, slabs, zero, N, Cin, IH, IW, Nout, Nout, ks)
    got = slabs.sum(0)
    assert torch.isfinite(got).all()
    assert float((got - want).abs().max()) <= 2e-5 * float(want.abs().max()) + 1e-6
    assert float(got[:, 9 * Cin:].abs().max()) == 0.0                       # the padding columns
    w = torch.zeros(Nout, Cin, 3, 3, requires_grad=True)
    F.conv2d(x.cpu().bfloat16().float(), w, None, stride=2, padding=1).backward(
        dy.float().cpu().view(N, OH, OW, Nout).permute(0, 3, 1, 2))
    ref = w.grad.permute(0, 2, 3, 1).reshape(Nout, 9 * Cin)                # column = (kh*3+kw)*Cin + ci
    assert float((got[:, :9 * Cin].cpu() - ref).norm() / ref.norm()) < 3e-3
    # frame-mapped frames: frame n = (s, t) of an item buffer [S][2][T] taken at view 1
    S, T = 1, N
    buf = torch.randn(S, 2, T, Cin, IH, IW, generator=g).cuda()
    buf[:, 1] = x.view(S, T, Cin, IH, IW)
    fsz = Cin * IH * IW
    slabs_m = torch.empty_like(slabs)
    L.call("rbvae_wgrad_first", 1, 0, buf[:, 1], T, T, 2 * T * fsz, 0, fsz, dy, slabs_m, zero, N, Cin, IH, IW, Nout, Nout, ks)
    assert torch.equal(slabs_m, slabs)
    # mode 1: NHWC f32 image (d(loss)/d(pre-sigmoid) of the last ConvTranspose2d)
    dpre = (torch.randn(N, IH, IW, Cin, generator=g) * 0.1).cuda()
    gate = torch.ones(P, Nf, dtype=torch.bfloat16, device="cuda")
    col3 = torch.empty(P, 64, dtype=torch.bfloat16, device="cuda")
    o3 = torch.empty(P, Nf, dtype=torch.bfloat16, device="cuda")
    L.call("rbvae_deconv_last_dgrad_fused", 1, dpre, Wp, zero, col3, gate, o3, N, Cin, IH, IW, Nf, Nf, 1.0, None)
    o4 = torch.full_like(o3, float("nan"))
    L.call("rbvae_deconv_last_dgrad_fused", 1, dpre, Wp, zero, None, gate, o4, N, Cin, IH, IW, Nf, Nf, 1.0, None)
    assert torch.equal(o3.view(torch.int16), o4.view(torch.int16))
    want3 = via_col(col3)
    slabs3 = torch.full((ks, Nout, 64), float("nan"), device="cuda")
    L.call("rbvae_wgrad_first", 1, 1, dpre, 0, 0, 0, 0, 0, dy, slabs3, zero, N, Cin, IH, IW, Nout, Nout, ks)
    got3 = slabs3.sum(0)
    assert float((got3 - want3).abs().max()) <= 2e-5 * float(want3.abs().max()) + 1e-6
    with pytest.raises(ValueError):
        L.call("rbvae_wgrad_first", 1, 0, x, 0, 0, 0, 0, fsz, dy, slabs, zero, N, Cin, IH, IW, Nout, Nout, nblk + 1)


def test_wgrad_first_at_the_cfg3_frame_size(sfv):
    """BASELINE configs[2] size: 128 frames of 3 x 256 x 256 -> 2 097 152 output pixels, 64 channels, the engine's K-split
    of 512: rbvae_wgrad_first against rbvae_wgrad_gemm over the im2col rows rbvae_conv_first_fused writes (same bf16
    products), and zero padding columns."""
    L = sfv._lib
    N, Cin, IH, IW, Nout = 128, 3, 256, 256, 64
    OH, OW = IH // 2, IW // 2
    P = N * OH * OW
    g = torch.Generator(device="cuda").manual_seed(17)
    x = torch.rand(N, Cin, IH, IW, device="cuda", generator=g)
    dy = (torch.randn(P, Nout, device="cuda", generator=g) / 8).bfloat16()
    Wp = torch.zeros(Nout, 64, dtype=torch.bfloat16, device="cuda")
    b = torch.zeros(Nout, device="cuda")
    zero = torch.zeros(256, dtype=torch.uint8, device="cuda")
    col = torch.empty(P, 64, dtype=torch.bfloat16, device="cuda")
    out = torch.empty(P, Nout, dtype=torch.bfloat16, device="cuda")
    L.call("rbvae_conv_first_fused", 1, x, 0, 0, 0, 0, Cin * IH * IW, Wp, b, zero, col, out, N, Cin, IH, IW, Nout, Nout, 1, 0, 0.0,
           1.0, 0, None)
    k2 = P // 4096
    slabs2 = torch.empty(k2, Nout, 64, device="cuda")
    L.call("rbvae_wgrad_gemm", 1, dy, col, slabs2, None, zero, P, P, Nout, 64, Nout, 64, 1, k2)
    ref = slabs2.double().sum(0)
    nblk = L.query("rbvae_wgrad_first_blocks", 1, Cin, IH, IW, Nout, N)
    assert nblk == N * 16 * 8
    slabs = torch.full((512, Nout, 64), float("nan"), device="cuda")
    L.call("rbvae_wgrad_first", 1, 0, x, 0, 0, 0, 0, Cin * IH * IW, dy, slabs, zero, N, Cin, IH, IW, Nout, Nout, 512)
    got = slabs.double().sum(0)
    assert torch.isfinite(got).all()
    assert float((got - ref).norm() / ref.norm()) < 1e-5
    assert float(got[:, 27:].abs().max()) == 0.0


def _job_row(kind, src, dst, d0, d1, d2, nslab=1, slab=0, dtype=0, accumulate=0, scale=1.0, dst2=None):
    import struct
    bits = struct.unpack("<I", struct.pack("<f", float(scale)))[0]
    return [kind, src.data_ptr(), dst.data_ptr(), d0, d1, d2, 0, 0, 0, nslab, slab, dtype, accumulate, bits, 0,
            0 if dst2 is None else dst2.data_ptr()]


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("Co,Ci,kk", [(256, 256, 9), (64, 64, 9), (128, 64, 16), (256, 128, 16), (72, 40, 9), (36, 20, 4)])
def test_conv_pack_job_both_orders(sfv, dtype, Co, Ci, kk):
    """rbvae_run_jobs kind 3: f32 weight [co][ci][kk] -> [co][t][ci] and [ci][t][co] in the storage type (16-byte row
    kernel; (36, 20, 4) takes its element-wise path)."""
    dt, tdt, _ = DT[dtype]
    g = torch.Generator().manual_seed(90)
    w = torch.randn(Co, Ci, kk, generator=g)
    src = w.cuda()
    wf = torch.full((Co, kk, Ci), 7.0, dtype=tdt, device="cuda")
    wd = torch.full((Ci, kk, Co), 7.0, dtype=tdt, device="cuda")
    tab = torch.tensor([_job_row(3, src, wf, Co, Ci, kk, dtype=dt, dst2=wd)], dtype=torch.int64).cuda()
    sfv._lib.call("rbvae_run_jobs", tab, 1, 256)
    assert torch.equal(wf.cpu(), w.permute(0, 2, 1).to(tdt)) and torch.equal(wd.cpu(), w.permute(1, 2, 0).to(tdt))


@pytest.mark.parametrize("Co,Ci,kk,ks,acc", [(256, 256, 9, 7, 0), (64, 64, 9, 3, 1), (128, 64, 16, 2, 0), (8, 520, 9, 5, 0),
                                             (16, 12, 4, 1, 0), (64, 64, 9, 19, 0), (64, 64, 9, 256, 1)])
def test_conv_reduce_job_matches_permute_reduce(sfv, Co, Ci, kk, ks, acc):
    """rbvae_run_jobs kind 4 (coalesced rows, LDS transpose) == rbvae_permute_reduce (one thread per output), bit for
    bit: both add the slabs in slab order."""
    g = torch.Generator().manual_seed(91)
    slabs = torch.randn(ks, Co, kk, Ci, generator=g).cuda()
    base = torch.randn(Co, Ci, kk, generator=g).cuda()
    ref, out = base.clone(), base.clone()
    sfv._lib.call("rbvae_permute_reduce", slabs, ks, Co * kk * Ci, ref, Co, Ci, kk, kk * Ci, 1, Ci, 0.5, acc)
    tab = torch.tensor([_job_row(4, slabs, out, Co, Ci, kk, nslab=ks, slab=Co * kk * Ci, accumulate=acc, scale=0.5)],
                       dtype=torch.int64).cuda()
    sfv._lib.call("rbvae_run_jobs", tab, 1, 256)
    want = slabs.sum(0).permute(0, 2, 1) * 0.5 + (base if acc else 0)
    assert torch.allclose(out, want, atol=1e-5, rtol=1e-5)
    assert torch.equal(out, ref)


@pytest.mark.parametrize("rows,C,acc", [(16384, 64, 0), (3840, 256, 1), (1030, 4, 0), (1000, 64, 0), (5000, 3, 0)])
def test_reduce_rows_job_over_thousands_of_partial_rows(sfv, rows, C, acc):
    """rbvae_run_jobs kind 2 (the per-tile column sums of a layer -> its bias gradient): from 1024 rows and a column count
    divisible by 4 up a workgroup owns four columns and walks the rows 16 bytes at a time; below (or with 3 columns) one
    wave per column.  Either way: the sums, in an order that does not depend on the run (bit-identical twice)."""
    g = torch.Generator().manual_seed(rows + C)
    ws = torch.randn(rows, C, generator=g).cuda()
    base = torch.randn(C, generator=g).cuda()
    outs = []
    for _ in range(2):
        out = base.clone()
        tab = torch.tensor([_job_row(2, ws, out, 1, 1, C, nslab=rows, slab=C, accumulate=acc, scale=0.25)],
                           dtype=torch.int64).cuda()
        sfv._lib.call("rbvae_run_jobs", tab, 1, 256)
        outs.append(out)
    want = ws.double().sum(0) * 0.25 + (base.double() if acc else 0)
    assert torch.allclose(outs[0].double(), want, atol=2e-4, rtol=1e-5)
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("variant,in_ch,dtype,Ld,hw", [("percep", 4, "bf16", 32, (16, 16)), ("percep", 4, "f32", 25, (16, 24)),
                                                        ("contrastive", 3, "bf16", 50, (32, 16)), ("percep", 4, "bf16", 32, (88, 160)),
                                                        ("contrastive", 3, "bf16", 25, (256, 256)), ("percep", 4, "f32", 32, (96, 128))])
def test_fused_update_jobs_equal_adam_plus_pack(sfv, variant, in_ch, dtype, Ld, hw):
    """Engine.update_jobs (optimiser step + weight repack as one batched job launch: kinds 3 / 6 / 7 with an Adam context)
    against rbvae_adam_step followed by Engine.pack: parameters, both moments and every packed copy bit for bit.  The last
    three shapes have fc weights of >= 1 M elements: their permuted copies go through the job's LDS tiles."""
    from importlib import import_module
    E = import_module("symbols-from-video_amd.engine")
    g = torch.Generator().manual_seed(95)
    packed = ("W1p", "W2f", "W2d", "W3f", "W3d", "Wfc", "WfcT", "Wdfc", "WdfcT", "bdfc", "V1f", "V1d", "V2f", "V2d", "V3p",
              "V3f", "wT_enc", "wT_dec")
    res = []
    for fused in (False, True, "sized"):
        eng = E.Engine(variant, in_ch, in_ch, Ld, hw, dtype, torch.device("cuda", 0))
        n = eng.layout.total
        gen = torch.Generator().manual_seed(96)
        flat = (torch.randn(n, generator=gen) * 0.1).cuda()
        grad = (torch.randn(n, generator=gen) * 0.01).cuda()
        m = (torch.randn(n, generator=gen) * 0.01).cuda()
        v = (torch.rand(n, generator=gen) * 1e-4).cuda()
        step = torch.tensor([6], dtype=torch.int64, device="cuda")
        hyper = torch.zeros(2, device="cuda")
        one = torch.ones(1, device="cuda")
        out4 = torch.empty(4, device="cuda")
        # the bias-correction terms as the trainer prepares them (step 7)
        sfv._lib.call("rbvae_combine_losses", None, 0, 0.0, one, one, 0, 0.0, one, 0, 0.0, 0.0, 1.0, 1.0, out4, step, 2e-3, None,
                      0.9, 0.999, hyper)
        if fused == "sized":      # the same table launched with exactly the workgroups its jobs can use (rbvae_run_jobs_sized)
            tab, nj = eng.update_jobs(flat, grad, m, v, hyper, (0.9, 0.999), 1e-8, 0.5)
            bmap, nb = eng._table_maps[tab.data_ptr()][:2]
            assert nb < nj * 256 and int(bmap.view(-1, 4)[:, 0].max()) == nj - 1
            eng.run_table(tab, nj)
        elif fused:
            tab, nj = eng.update_jobs(flat, grad, m, v, hyper, (0.9, 0.999), 1e-8, 0.5)
            sfv._lib.call("rbvae_run_jobs", tab, nj, 256)
        else:
            sfv._lib.call("rbvae_adam_step", flat, grad, m, v, n, 2e-3, 0.9, 0.999, 1e-8, 0, 0.5, None, hyper)
            eng.pack(flat)
        torch.cuda.synchronize()
        lay = eng.layout
        # (the alignment gaps between tensors belong to no parameter: the fused jobs do not touch them)
        pv = {f"{t}/{nm}": lay.view(buf, nm).clone() for t, buf in (("w", flat), ("m", m), ("v", v)) for nm in lay.names}
        res.append({**pv, **{k: getattr(eng, k).clone() for k in packed}})
    for k in res[0]:
        assert torch.equal(res[0][k], res[1][k]), k
        assert torch.equal(res[0][k], res[2][k]), k
