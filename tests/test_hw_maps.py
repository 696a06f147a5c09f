"""GPU probes of the gfx950 lane maps that conv_gemm.hip is written against."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    import sfv_amd
    return sfv_amd._lib


def test_mfma_bf16_map(L):
    g = torch.Generator().manual_seed(0)
    A = torch.randint(-4, 5, (16, 32), generator=g).float()
    B = torch.randint(-4, 5, (32, 16), generator=g).float()     # asymmetric
    D = torch.empty(16, 16, device="cuda")
    L.dbg_call("rbvae_dbg_mfma_bf16", A.bfloat16().cuda(), B.bfloat16().cuda(), D)
    assert torch.equal(D.cpu(), A @ B)


def test_mfma_f32_map(L):
    g = torch.Generator().manual_seed(1)
    A = torch.randn(16, 4, generator=g)
    B = torch.randn(4, 16, generator=g)
    D = torch.empty(16, 16, device="cuda")
    L.dbg_call("rbvae_dbg_mfma_f32", A.cuda(), B.cuda(), D)
    ref = torch.zeros(16, 16)
    for k in range(4):                      # k-ordered fma chain
        ref = torch.addcmul(ref, A[:, k:k + 1].expand(16, 16), B[k:k + 1, :].expand(16, 16))
    assert torch.allclose(D.cpu(), A @ B, atol=1e-6)


def test_glds_lane_linear_destination(L):
    src = torch.arange(64 * 4 * 2, dtype=torch.int32)           # 128 chunks of 16 B
    perm = torch.randperm(128, generator=torch.Generator().manual_seed(2))[:64].to(torch.int32)
    out = torch.empty(256, dtype=torch.int32, device="cuda")
    L.dbg_call("rbvae_dbg_glds", src.cuda(), perm.cuda(), out)
    want = src.view(128, 4)[perm.long()].reshape(-1)            # lane i lands at LDS byte 16*i
    assert torch.equal(out.cpu(), want)


def test_tr16_block_transpose(L):
    """Per 16-lane group: lane 4q+p supplies the address of row q, cols 4p..4p+3 of a
    4x16 block; lane i receives column i of the block's 4 rows."""
    img = torch.arange(32 * 64, dtype=torch.int16)
    lane = torch.arange(64)
    g, i = lane // 16, lane % 16
    q, p = i // 4, i % 4
    row0 = torch.tensor([0, 8, 4, 20])[g]                       # block's first row per group
    col0 = torch.tensor([0, 16, 32, 48])[g]
    rowsel = (row0 + q).to(torch.int32)
    colsel = (col0 + 4 * p).to(torch.int32)
    out = torch.empty(64 * 4, dtype=torch.int16, device="cuda")
    L.dbg_call("rbvae_dbg_tr16", img.cuda(), rowsel.cuda(), colsel.cuda(), out)
    im = img.view(32, 64)
    want = torch.stack([im[row0 + j, col0 + i] for j in range(4)], dim=1).reshape(-1)
    assert torch.equal(out.cpu(), want)
