"""The persistent, wave-specialised bf16 instance of rbvae_conv3x3_halo (csrc/conv_halo_ws.hip: producer waves stage the
patch and stream the weights, MFMA waves multiply, one workgroup per CU walks its tiles) against the one-tile-per-workgroup
kernel of csrc/conv_halo.hip (include/rbvae_dbg.h: rbvae_dbg_conv_halo_variant) -- the same sums in the same order, so the
outputs and the GroupNorm partial statistics are compared BIT FOR BIT -- and against torch on the CPU.

Reference ops: torch.nn.Conv2d(cin, cout, 3, 1, 1) of the LDM ResnetBlock (ldm/modules/diffusionmodules/model.py:82-141)
with the producer's GroupNorm(32, eps 1e-6) + swish (:33-39) folded into the staging."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sfv():
    import sfv_amd
    return sfv_amd


def run(sfv, variant, A, Wp, out, bias, addend, N, H, W, cin, cout, scale=None, shift=None, swish=0, stats=None, cg=0, pad=(1, 1),
        OH=None, OW=None):
    zero = torch.zeros(256, dtype=torch.uint8, device="cuda")
    l = sfv._lib.dbg_lib()
    old = l.rbvae_dbg_conv_halo_variant(variant)          # 1: one tile per workgroup, 0: persistent wherever it covers
    try:
        sfv._lib.call("rbvae_conv3x3_halo", 1, A, Wp, out, bias, addend, zero, scale, shift, swish, stats, cg, N, H, W,
                      OH or H, OW or W, pad[0], pad[1], cin, cout, A.shape[1], out.shape[1])
        torch.cuda.synchronize()
    finally:
        l.rbvae_dbg_conv_halo_variant(old)


CASES = [
    # N, C, Co, H, W, gn, swish, stats, addend, bias
    (2, 64, 128, 16, 16, 0, 0, 0, 0, 1),          # one slice, one tile per workgroup
    (1, 128, 256, 40, 24, 0, 0, 0, 1, 1),         # ragged tiles, two channel tiles, residual
    (3, 256, 128, 22, 40, 1, 1, 1, 0, 1),         # ragged, GroupNorm + swish in flight, statistics out
    (1, 512, 512, 32, 32, 1, 0, 1, 1, 0),         # eight slices, four channel tiles, GroupNorm without swish
    (2, 128, 128, 8, 16, 0, 0, 1, 0, 0),          # half-height tiles
    (5, 128, 128, 112, 96, 1, 1, 1, 1, 1),        # 5 x 7 x 6 = 210 tiles ... below one per CU
    (6, 128, 256, 96, 112, 1, 1, 1, 1, 1),        # 6 x 6 x 7 x 2 = 504 items: two per workgroup with a ragged walk
    (9, 64, 128, 80, 80, 0, 0, 0, 1, 1),          # 9 x 25 = 225 tiles of ONE slice (a tile per slice: buffers alternate per tile)
    (20, 64, 128, 64, 80, 1, 1, 1, 0, 1),         # 400 one-slice tiles: walks of one and two tiles
    (4, 192, 128, 48, 48, 1, 1, 0, 0, 1),         # three slices (odd): the epilogue's buffer alternates between tiles
    (12, 64, 128, 128, 128, 1, 1, 1, 1, 1),       # 768 one-slice tiles: three per workgroup (the product dispatch's territory)
    (4, 128, 256, 256, 192, 1, 1, 1, 1, 1),       # 4 x 16 x 12 x 2 = 1536 items of two slices: six per workgroup, both channel tiles
]


@pytest.mark.parametrize("N,C,Co,H,W,gn,swish,st,add,bi", CASES)
def test_persistent_kernel_is_the_single_tile_kernel_bit_for_bit(sfv, N, C, Co, H, W, gn, swish, st, add, bi):
    lib = sfv._lib
    assert lib.query("rbvae_conv3x3_halo_ok", 1, H, W, H, W, C, Co)
    g = torch.Generator().manual_seed(3 + C + H + N)
    A = (torch.randn(N * H * W, C, generator=g) * 0.8 + 0.3).bfloat16().cuda()
    Wp = (torch.randn(Co, 9, C, generator=g) / (3.0 * C ** 0.5)).bfloat16().cuda()
    bias = torch.randn(Co, generator=g).cuda() if bi else None
    addend = torch.randn(N * H * W, Co, generator=g).bfloat16().cuda() if add else None
    scale = (torch.rand(N, C, generator=g) + 0.5).cuda() if gn else None
    shift = torch.randn(N, C, generator=g).cuda() if gn else None
    cg = Co // 32
    outs, stats = [], []
    for variant in (1, 0):
        out = torch.full((N * H * W, Co), float("nan"), dtype=torch.bfloat16, device="cuda")
        s = torch.full((lib.query("rbvae_conv3x3_halo_stats_floats", N, H, W, Co, cg),), float("nan"), device="cuda") if st else None
        run(sfv, variant, A, Wp, out, bias, addend, N, H, W, C, Co, scale, shift, swish, s, cg if st else 0)
        outs.append(out)
        stats.append(s)
    assert torch.isfinite(outs[1].float()).all()
    assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16))
    if st:
        assert torch.equal(stats[0].view(torch.int32), stats[1].view(torch.int32))
    # and a second launch over the same buffers gives the same bits (no state left in LDS / no race on the walk)
    out2 = torch.full_like(outs[1], float("nan"))
    run(sfv, 0, A, Wp, out2, bias, addend, N, H, W, C, Co, scale, shift, swish, None, 0)
    assert torch.equal(out2.view(torch.int16), outs[1].view(torch.int16))


@pytest.mark.parametrize("pad,dH,dW", [((0, 0), 2, 2), ((1, 0), 0, 2), ((2, 2), -2, -2), ((0, 1), 2, 0)])
def test_persistent_kernel_padding_variants_against_torch(sfv, pad, dH, dW):
    """pad_h / pad_w 0..2 (the output is OH x OW = IH + 2 pad - 2; the LDM Downsample's asymmetric pad is pad (0, 0) on a
    pre-padded image): the padding pixels come back as zeros from the out-of-range buffer offsets"""
    N, C, Co, IH, IW = 2, 128, 128, 36, 40
    OH, OW = IH + 2 * pad[0] - 2, IW + 2 * pad[1] - 2
    g = torch.Generator().manual_seed(5 + pad[0] * 3 + pad[1])
    x = torch.randn(N, C, IH, IW, generator=g)
    w = torch.randn(Co, C, 3, 3, generator=g) / (3.0 * C ** 0.5)
    b = torch.randn(Co, generator=g)
    xq, wq = x.bfloat16().float(), w.bfloat16().float()
    ref = F.conv2d(xq, wq, b, 1, pad)
    A = x.permute(0, 2, 3, 1).contiguous().reshape(-1, C).bfloat16().cuda()
    Wp = w.permute(0, 2, 3, 1).contiguous().reshape(Co, 9, C).bfloat16().cuda()
    outs = []
    for variant in (1, 0):
        out = torch.full((N * OH * OW, Co), float("nan"), dtype=torch.bfloat16, device="cuda")
        run(sfv, variant, A, Wp, out, b.cuda(), None, N, IH, IW, C, Co, pad=pad, OH=OH, OW=OW)
        outs.append(out)
    assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16))
    got = outs[1].float().cpu().reshape(N, OH, OW, Co).permute(0, 3, 1, 2)
    assert float((got - ref).norm() / ref.norm()) < 1.2e-2


def test_padding_stays_zero_under_groupnorm(sfv):
    """a shift that would make swish(0 * scale + shift) != 0: the border taps must still see zeros (torch pads the
    NORMALISED activation)"""
    N, C, Co, H, W = 1, 64, 128, 16, 16
    g = torch.Generator().manual_seed(21)
    x = torch.randn(N, C, H, W, generator=g)
    w = torch.randn(Co, C, 3, 3, generator=g) / (3.0 * C ** 0.5)
    scale, shift = torch.rand(N, C, generator=g) + 0.5, torch.randn(N, C, generator=g) + 2.0
    xq, wq = x.bfloat16().float(), w.bfloat16().float()
    hn = xq * scale[:, :, None, None] + shift[:, :, None, None]
    hn = (hn * torch.sigmoid(hn)).bfloat16().float()
    ref = F.conv2d(hn, wq, None, 1, 1)
    A = x.permute(0, 2, 3, 1).contiguous().reshape(-1, C).bfloat16().cuda()
    Wp = w.permute(0, 2, 3, 1).contiguous().reshape(Co, 9, C).bfloat16().cuda()
    out = torch.empty(N * H * W, Co, dtype=torch.bfloat16, device="cuda")
    run(sfv, 0, A, Wp, out, None, None, N, H, W, C, Co, scale.cuda(), shift.cuda(), 1)
    got = out.float().cpu().reshape(N, H, W, Co).permute(0, 3, 1, 2)
    assert float((got - ref).norm() / ref.norm()) < 1.5e-2
    assert float((got[:, :, 0, :] - ref[:, :, 0, :]).norm() / ref[:, :, 0, :].norm()) < 1.5e-2      # the border row
