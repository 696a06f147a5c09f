"""Frozen LDM VAE encoder (cfg 5) on HIP vs the reference Encoder's own outputs."""
import numpy as np
import pytest
import torch

import ldm_oracle as LO
from _golden import load

pytestmark = pytest.mark.gpu


def _model(sfv, g, dtype):
    torch.manual_seed(int(g["meta/seed"]))
    m = sfv.LDMEncoder(compute_dtype=dtype)
    sd = m.state_dict()
    ref = LO.init_params(int(g["meta/seed"]))
    assert list(sd.keys()) == list(ref.keys())
    for k in ref:                                   # same construction order => same initial weights
        assert torch.equal(sd[k], ref[k]), k
    return m.cuda()


def test_f32_matches_reference_encoder():
    import sfv_amd as sfv
    g = load("ldm_encoder")
    m = _model(sfv, g, "f32")
    x = torch.from_numpy(g["x"]).cuda()
    mom = m.moments(x).float().cpu().reshape(2, 8, 8, 8).permute(0, 3, 1, 2)
    np.testing.assert_allclose(mom.numpy(), g["moments"], atol=2e-4, rtol=1e-4)
    lat = m.encode(x, eps=torch.from_numpy(g["eps"]).cuda())
    np.testing.assert_allclose(lat.cpu().numpy(), g["latent"], atol=1e-4, rtol=1e-4)
    mode = m.encode(x, sample=False)
    np.testing.assert_allclose(mode.cpu().numpy(), 0.18215 * g["moments"][:, :4], atol=1e-4, rtol=1e-4)


def test_bf16_tracks_and_checkpoint_keys():
    import sfv_amd as sfv
    g = load("ldm_encoder")
    m = _model(sfv, g, "bf16")
    x = torch.from_numpy(g["x"]).cuda()
    lat = m.encode(x, eps=torch.from_numpy(g["eps"]).cuda()).cpu().numpy()
    ref = g["latent"]
    assert np.linalg.norm(lat - ref) / np.linalg.norm(ref) < 5e-2
    # a Stable-Diffusion style checkpoint (first_stage_model.* keys, extra decoder entries) loads as is
    sd = {f"first_stage_model.{k}": v + 0.01 for k, v in m.state_dict().items()}
    sd["first_stage_model.decoder.conv_in.weight"] = torch.zeros(3)
    m2 = sfv.LDMEncoder(compute_dtype="bf16")
    m2.load_state_dict(sd)
    k0 = "encoder.conv_in.weight"
    assert torch.allclose(m2.state_dict()[k0], m.state_dict()[k0].cpu() + 0.01)
    with pytest.raises(RuntimeError):
        m.moments(x.cpu())
    with pytest.raises(ValueError):
        m.moments(torch.zeros(1, 3, 60, 64, device="cuda"))


@pytest.mark.parametrize("dtype,C,N,HW", [("f32", 128, 2, 1000), ("f32", 512, 1, 77), ("bf16", 128, 3, 4096 + 5),
                                          ("bf16", 256, 2, 300), ("bf16", 512, 2, 64), ("f32", 36, 2, 50)])
def test_groupnorm_tiled_matches_torch(dtype, C, N, HW):
    """rbvae_groupnorm_swish_ws (row-tiled statistics, parallel-variance merge, 16-byte apply) against
    torch GroupNorm(32 | 4, eps 1e-6) + swish (ldm/modules/diffusionmodules/model.py:33-39) on NHWC rows with a
    large common offset (where a one-pass E[x^2]-E[x]^2 would cancel); C = 36 takes the fallback kernels."""
    import sfv_amd as sfv
    L = sfv._lib
    groups = 32 if C % 32 == 0 else 4
    g = torch.Generator().manual_seed(50 + C)
    x = torch.randn(N, C, HW, generator=g) * 0.5 + 3.0
    gamma, beta = torch.randn(C, generator=g), torch.randn(C, generator=g)
    tdt, dt, tol = (torch.float32, 0, 2e-5) if dtype == "f32" else (torch.bfloat16, 1, 3e-2)
    xq = x.to(tdt).float()
    ref = torch.nn.functional.group_norm(xq, groups, gamma, beta, eps=1e-6)
    ref = ref * torch.sigmoid(ref)
    rows = xq.permute(0, 2, 1).reshape(N * HW, C).contiguous().to(tdt).cuda()
    y = torch.empty_like(rows)
    nws = L.query("rbvae_groupnorm_ws_floats", dt, N, HW, C, groups)
    ws = torch.empty(nws, device="cuda")
    L.call("rbvae_groupnorm_swish_ws", dt, rows, y, gamma.cuda(), beta.cuda(), ws, nws, N, HW, C, C, C, groups, 1e-6, 1)
    got = y.float().cpu().reshape(N, HW, C).permute(0, 2, 1)
    assert float((got - ref).abs().max()) < tol * max(1.0, float(ref.abs().max()))
    if dtype == "f32":
        mean = ws[:N * groups].cpu().reshape(N, groups)
        np.testing.assert_allclose(mean.numpy(), xq.reshape(N, groups, -1).mean(-1).numpy(), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("N,hw,C", [(2, 96, 512), (1, 4096, 512), (3, 64, 64), (2, 160, 256), (1, 32, 128)])
def test_attention_online_softmax_matches_torch(N, hw, C):
    """rbvae_attention (batched, tiled, online softmax; csrc/attn.hip) against AttnBlock's arithmetic
    (model.py:186-198: bmm, * C^-0.5, softmax, bmm) on bf16-rounded operands, including a 4096-token image (512x512
    frames), a query count that is not a multiple of the 64-row tile, and column-block operands of a fused projection."""
    import sfv_amd as sfv
    L = sfv._lib
    assert L.query("rbvae_attention_ok", 1, hw, C) and not L.query("rbvae_attention_ok", 0, hw, C)
    g = torch.Generator().manual_seed(80 + hw)
    qkv = (torch.randn(N * hw, 3 * C, generator=g) * 1.5).bfloat16()
    q, k, v = (qkv[:, i * C:(i + 1) * C].float().reshape(N, hw, C) for i in range(3))
    w = torch.softmax(torch.bmm(q, k.transpose(1, 2)) * (int(C) ** (-0.5)), dim=2)
    ref = torch.bmm(w, v).reshape(N * hw, C)
    dq = qkv.cuda()
    o = torch.zeros(N * hw, C, dtype=torch.bfloat16, device="cuda")
    L.call("rbvae_attention", 1, dq, dq[:, C:], dq[:, 2 * C:], o, N, hw, C, 3 * C, 3 * C, 3 * C, C, float(int(C) ** (-0.5)))
    got = o.float().cpu()
    assert torch.isfinite(got).all()
    assert float((got - ref).norm() / ref.norm()) < 1e-2
    assert float((got - ref).abs().max()) < 4e-2 * float(ref.abs().max())
    with pytest.raises(ValueError):
        L.call("rbvae_attention", 1, dq, dq, dq, o, N, hw + 1, C, 3 * C, 3 * C, 3 * C, C, 1.0)


@pytest.mark.parametrize("dtype,tol", [("f32", 2e-4), ("bf16", 4e-2)])
def test_halo_resnet_path_equals_gather_path(dtype, tol):
    """The halo-resident 3x3 kernel with GroupNorm folded into its staging / epilogue (conv_impl="halo", the default)
    against the gather-GEMM + standalone GroupNorm kernels (conv_impl="gather") on 128x128 frames: every ResnetBlock
    level of the encoder (128, 64, 32 and 16 pixel images; model.py:82-141, 368-459) takes the halo path."""
    import sfv_amd as sfv
    torch.manual_seed(5)
    a = sfv.LDMEncoder(compute_dtype=dtype, conv_impl="halo").cuda()
    b = sfv.LDMEncoder(compute_dtype=dtype, conv_impl="gather").cuda()
    b.load_state_dict(a.state_dict())
    x = (torch.rand(2, 3, 128, 128, generator=torch.Generator().manual_seed(6)) * 2 - 1).cuda()
    ma, mb = a.moments(x).float().cpu()[:, :8], b.moments(x).float().cpu()[:, :8]
    assert torch.isfinite(ma).all()
    assert float((ma - mb).norm() / mb.norm()) < tol
    with pytest.raises(ValueError):
        sfv.LDMEncoder(conv_impl="cudnn")


def test_captured_encode_equals_eager_and_follows_new_weights():
    """LDMEncoder(use_graph=True): the second call of an input shape captures the encode into a HIP graph,
    later calls replay it.  Replays on NEW frames are bit-identical to the eager launches on those frames; another shape
    gets its own graph; loading weights drops the graphs (they hold the packed copies' addresses) and the next encodes
    follow the new weights."""
    import sfv_amd as sfv
    torch.manual_seed(9)
    g = sfv.LDMEncoder(compute_dtype="bf16", use_graph=True).cuda()
    e = sfv.LDMEncoder(compute_dtype="bf16").cuda()
    e.load_state_dict(g.state_dict())
    gen = torch.Generator().manual_seed(10)
    xs = [(torch.rand(2, 3, 64, 64, generator=gen) * 2 - 1).cuda() for _ in range(4)]
    for i, x in enumerate(xs):
        assert torch.equal(g.encode(x, sample=False), e.encode(x, sample=False)), f"call {i}"
    assert len(g._graphs) == 1 and not e._graphs
    eps = torch.randn(2, 4, 8, 8, generator=gen).cuda()
    assert torch.equal(g.encode(xs[0], eps=eps), e.encode(xs[0], eps=eps))
    m1 = g.moments(xs[1])
    m2 = g.moments(xs[2])                       # public moments are copies: the second replay does not rewrite the first
    assert torch.equal(m1, e.moments(xs[1])) and torch.equal(m2, e.moments(xs[2]))
    y = (torch.rand(1, 3, 64, 128, generator=gen) * 2 - 1).cuda()
    for _ in range(3):
        assert torch.equal(g.encode(y, sample=False), e.encode(y, sample=False))
    assert len(g._graphs) == 2
    sd = {k: (v * 1.5 if k.endswith("conv_in.weight") else v) for k, v in g.state_dict().items()}
    g.load_state_dict(sd)
    e.load_state_dict(sd)
    assert not g._graphs
    for _ in range(3):
        assert torch.equal(g.encode(xs[3], sample=False), e.encode(xs[3], sample=False))
    with pytest.raises(ValueError):
        g.encode(torch.zeros(1, 3, 60, 64, device="cuda"))
    # an IN-PLACE weight change (no load_state_dict, no .to()): the frozen buffers' version counters move, the packed
    # copies and the graphs that hold their addresses are rebuilt instead of replaying against stale weights
    for _ in range(2):
        g.encode(xs[0], sample=False)
    assert g._graphs
    for m in (g, e):
        m._p("encoder.conv_in.weight").mul_(0.5)
    out_g = g.encode(xs[0], sample=False)
    assert torch.equal(out_g, e.encode(xs[0], sample=False))
    for _ in range(2):                          # ... and the re-captured graph replays the new weights
        assert torch.equal(g.encode(xs[0], sample=False), out_g)


def test_halo_equals_gather_on_a_four_tile_layer_and_standalone_groupnorm():
    """512-channel layers (level 2/3 + mid block: four 128-channel column tiles per pixel tile, GroupNorm as ONE standalone
    apply pass feeding the four tiles instead of fused into the staging) at 256 x 256 frames, where the 512-wide maps are
    64 x 64 and 32 x 32: halo path against the gather-GEMM + GroupNorm kernels."""
    import sfv_amd as sfv
    torch.manual_seed(15)
    a = sfv.LDMEncoder(compute_dtype="bf16", conv_impl="halo").cuda()
    b = sfv.LDMEncoder(compute_dtype="bf16", conv_impl="gather").cuda()
    b.load_state_dict(a.state_dict())
    x = (torch.rand(1, 3, 256, 256, generator=torch.Generator().manual_seed(16)) * 2 - 1).cuda()
    ma, mb = a.moments(x).float().cpu()[:, :8], b.moments(x).float().cpu()[:, :8]
    assert torch.isfinite(ma).all()
    assert float((ma - mb).norm() / mb.norm()) < 4e-2


@pytest.mark.parametrize("N,Cin,H,W,Nout,cg", [(2, 3, 64, 64, 128, 4), (1, 3, 21, 40, 128, 4), (3, 4, 16, 48, 64, 8), (2, 3, 24, 17, 256, 16),
                                               (1, 1, 8, 16, 32, 4)])
def test_conv_in_matches_conv2d_and_its_groupnorm_statistics(N, Cin, H, W, Nout, cg):
    """rbvae_conv_in (csrc/conv_in.hip): the encoder's conv_in (model.py:385-389) as one kernel against F.conv2d on the
    bf16-rounded operands, bit-identical to rbvae_im2col + the one-tap rbvae_gather_gemm it replaces, and the per-tile
    GroupNorm partials of its epilogue merged by rbvae_gn_finish_tiles(.., 8, 16) against the mean / variance of the stored
    output per (image, group) -- blocks that hang over the image, 1..4 input channels, 32..256 output channels."""
    import ctypes
    import sfv_amd as sfv
    import torch.nn.functional as F
    L = sfv._lib
    g = torch.Generator().manual_seed(300 + H + Nout)
    x = (torch.rand(N, Cin, H, W, generator=g) * 2 - 1).cuda()
    Wt = torch.randn(Nout, Cin, 3, 3, generator=g) * 0.2
    b = torch.randn(Nout, generator=g).cuda()
    Wp = torch.zeros(Nout, 64, dtype=torch.bfloat16)
    Wp[:, :9 * Cin] = Wt.permute(0, 2, 3, 1).reshape(Nout, 9 * Cin).bfloat16()
    Wp = Wp.cuda()
    zero = torch.zeros(256, dtype=torch.uint8, device="cuda")
    G = Nout // cg
    assert L.query("rbvae_conv_in_ok", 1, Cin, H, W, Nout, N, cg) == 1
    assert L.query("rbvae_conv_in_ok", 0, Cin, H, W, Nout, N, cg) == 0
    out = torch.full((N * H * W, Nout), float("nan"), dtype=torch.bfloat16, device="cuda")
    st = torch.full((L.query("rbvae_conv_in_stats_floats", N, H, W, Nout, cg),), float("nan"), device="cuda")
    L.call("rbvae_conv_in", 1, x, Wp, b, zero, out, st, cg, N, Cin, H, W, Nout, Nout)
    got = out.float().cpu().view(N, H, W, Nout).permute(0, 3, 1, 2)
    ref = F.conv2d(x.cpu().bfloat16().float(), Wt.bfloat16().float(), b.cpu(), stride=1, padding=1)
    assert torch.isfinite(got).all()
    np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=1e-2, atol=1e-2)
    # the two-kernel path it replaces
    col = torch.empty(N * H * W, 64, dtype=torch.bfloat16, device="cuda")
    L.call("rbvae_im2col", 1, x, Cin * H * W, H * W, W, 1, N, Cin, H, W, H, W, 3, 3, 1, 1, 64, col)
    out2 = torch.empty_like(out)
    d_one = (ctypes.c_int * 6)(1, 0, 0, 0, 0, 0)
    L.call("rbvae_gather_gemm", 1, col, Wp, out2, b, None, None, None, zero, N * H * W, 1, 1, 1, 1, 1, 1, 1, 1, 64, Nout, 64,
           Nout, 1, 1, ctypes.addressof(d_one), 0, 0, 0.0, 1.0, 0, None, None)
    assert torch.equal(out.view(torch.int16), out2.view(torch.int16))
    # statistics of the stored values
    gamma, beta = torch.ones(Nout, device="cuda"), torch.zeros(Nout, device="cuda")
    sc, sh = torch.empty(N, Nout, device="cuda"), torch.empty(N, Nout, device="cuda")
    mean, rstd = torch.empty(N * G, device="cuda"), torch.empty(N * G, device="cuda")
    L.call("rbvae_gn_finish_tiles", st, gamma, beta, sc, sh, mean, rstd, N, H, W, Nout, G, 1e-6, 8, 16)
    v = got.double().view(N, G, cg, H * W)
    m_ref = v.mean((2, 3)).reshape(-1)
    var_ref = v.var((2, 3), unbiased=False).reshape(-1)
    np.testing.assert_allclose(mean.cpu().double().numpy(), m_ref.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(rstd.cpu().double().numpy(), (var_ref + 1e-6).rsqrt().numpy(), rtol=1e-4)
    # without statistics
    out3 = torch.empty_like(out)
    L.call("rbvae_conv_in", 1, x, Wp, b, zero, out3, None, 0, N, Cin, H, W, Nout, Nout)
    assert torch.equal(out3.view(torch.int16), out.view(torch.int16))


def test_prefetching_attention_equals_the_load_wait_multiply_form_bit_for_bit():
    """C = 512 from four key tiles up: producer waves prefetch the K / V tiles three half-steps ahead (attn_flash_db_k); fewer
    tokens take the load-wait-multiply kernel (attn_flash_k).  Same arithmetic in the same order: running the long sequence
    in two ways -- whole (prefetching kernel) and as the first 96 tokens only (short kernel) -- cannot be compared directly
    (different softmax sets), so the check is the per-row definition on the CPU at both lengths with one tolerance, plus
    run-to-run bit-identity of the prefetching kernel."""
    import sfv_amd as sfv
    L = sfv._lib
    C = 512
    g = torch.Generator().manual_seed(41)
    for N, hw in ((2, 96), (2, 160), (1, 1024)):
        qkv = (torch.randn(N * hw, 3 * C, generator=g) * 1.2).to(torch.bfloat16)
        dq = qkv.cuda()
        o = torch.empty(N * hw, C, dtype=torch.bfloat16, device="cuda")
        L.call("rbvae_attention", 1, dq, dq[:, C:], dq[:, 2 * C:], o, N, hw, C, 3 * C, 3 * C, 3 * C, C, float(C ** -0.5))
        o2 = torch.empty_like(o)
        L.call("rbvae_attention", 1, dq, dq[:, C:], dq[:, 2 * C:], o2, N, hw, C, 3 * C, 3 * C, 3 * C, C, float(C ** -0.5))
        assert torch.equal(o, o2)
        q, k, v = (qkv[:, i * C:(i + 1) * C].float().reshape(N, hw, C) for i in range(3))
        ref = torch.softmax(q @ k.transpose(1, 2) * C ** -0.5, dim=-1) @ v
        got = o.float().cpu().reshape(N, hw, C)
        assert float((got - ref).norm() / ref.norm()) < 2e-2
