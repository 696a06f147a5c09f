"""Batched state-consistency evaluator on the GPU: same codes as the reference's one-frame-at-a-time loop."""
import random

import numpy as np
import pytest
import torch

import rbvae_oracle as O

pytestmark = pytest.mark.gpu


def test_batched_consistency_equals_per_frame_loop():
    import sfv_amd as sfv
    torch.manual_seed(2)
    Ld, hw = 25, (16, 16)
    m = sfv.Seq2SeqBinaryVAE(4, 4, Ld, Ld, variant="percep", input_hw=hw).cuda().eval()
    segs = [(0, 60), (70, 130), (140, 200)]
    flags = [65, 135]
    g = torch.Generator().manual_seed(3)
    table = torch.randn(200, 4, *hw, generator=g)
    random.seed(5)
    ds = sfv.DeviceStatePairDataset(table, segs, 0.1, 0.2, mode="val", device="cuda")
    val = [i for ix in ds.val_indices_per_state for i in ix]
    U = torch.rand(len(val), Ld, generator=g)
    avg, pct = sfv.state_consistency(m, ds, flags, temperature=0.3, noise_ratio=0.1, batch=7, u=U)
    # the reference's loop: one frame per call, B = T = 1 (percep_RBVAE_train.py:456-465)
    codes = []
    for j, fi in enumerate(val):
        z = m.encode(table[fi][None, None].cuda(), temperature=0.3, hard=True, noise_ratio=0.1, u=U[j:j + 1].cuda())
        codes.append(z.cpu().numpy().squeeze())
    labels = np.array([O.assign_label(i, flags) for i in val])
    ravg, rpct = O.state_consistency(np.array(codes), labels, len(flags) + 1)
    assert abs(avg - ravg) < 1e-12 and np.allclose(pct, rpct)
    b = ds.batch([0, 1])
    assert b.is_cuda and b.shape == (2, 2, 3, 4, *hw)
