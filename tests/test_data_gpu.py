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


def test_batch_gathers_into_the_trainer_input_buffer():
    """DeviceStatePairDataset.batch(out=trainer.input_buffer(...)) + trainer.step(that buffer): the same step as
    with a freshly gathered batch (which step() copies into the buffer)."""
    import random
    import sfv_amd as sfv
    from importlib import import_module
    FusedTrainer = import_module("symbols-from-video_amd.trainer").FusedTrainer
    data = import_module("symbols-from-video_amd.data")
    g = torch.Generator().manual_seed(3)
    emb = torch.randn(60, 4, 16, 16, generator=g)
    segments = [(0, 20), (20, 40), (40, 60)]
    random.seed(5)
    ds = data.DeviceStatePairDataset(emb, segments, mode="train", device="cuda")
    idx = [0, 2, 3]
    ref = ds.batch(idx)
    assert ref.shape == (3, 2, 3, 4, 16, 16)
    U = torch.rand(2, 3 * 3, 32, generator=g).cuda()
    res = []
    for in_place in (False, True):
        torch.manual_seed(8)
        m = sfv.Seq2SeqBinaryVAE(4, 4, 32, 32, variant="percep", input_hw=(16, 16), compute_dtype="f32").cuda().eval()
        tr = FusedTrainer(m, device_noise=False, use_graph=True)
        if in_place:
            x = ds.batch(idx, out=tr.input_buffer(3, 3, 4, 16, 16))
            assert x.data_ptr() == tr.input_buffer(3, 3, 4, 16, 16).data_ptr() and torch.equal(x, ref)
        else:
            x = ref
        for _ in range(2):
            losses = tr.step(x, 0.8, U=U).clone()
        res.append((losses, m._flat.clone()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    with pytest.raises(ValueError):
        ds.batch(idx, out=torch.empty(2, 2, 3, 4, 16, 16, device="cuda"))


def test_planned_epoch_trains_from_the_resident_table():
    """FusedTrainer.set_data(table, plan) + step(None): the batch of every step is gathered inside the captured step
    from the HBM-resident table by the device step counter -- same losses and weights, bit for bit, as feeding
    ds.batch(...) of the same items by hand; the plan wraps around; a new epoch's plan keeps the graph."""
    import sfv_amd as sfv
    Ld, hw, B = 32, (16, 16), 2
    segs = [(0, 40), (40, 80), (80, 120)]
    g = torch.Generator().manual_seed(13)
    table = torch.randn(120, 4, *hw, generator=g)
    res = []
    for planned in (False, True):
        random.seed(6)
        ds = sfv.DeviceStatePairDataset(table, segs, 0.1, 0.1, mode="train", device="cuda")
        order = torch.randperm(len(ds), generator=torch.Generator().manual_seed(14))
        plan = ds.plan(order, B)
        assert plan.shape == (len(ds) // B, B, 2, 3) and plan.dtype == torch.int64
        torch.manual_seed(15)
        m = sfv.Seq2SeqBinaryVAE(4, 4, Ld, Ld, variant="percep", input_hw=hw, compute_dtype="f32").cuda().train()
        tr = sfv.FusedTrainer(m, device_noise=True, use_graph=True, seed=16)
        losses = []
        nb = plan.shape[0]
        if planned:
            tr.set_data(ds.table, plan)
        for i in range(nb + 2):                                   # two steps past the end: the plan wraps
            if planned:
                losses.append(tr.step(None, 0.8).clone())
            else:
                items = order[(i % nb) * B:(i % nb + 1) * B].tolist()
                losses.append(tr.step(ds.batch(items), 0.8).clone())
        if planned:
            # next epoch: a new shuffle of the same shape goes into the same device buffer, no re-capture
            tr.set_data(ds.table, ds.plan(order.flip(0), B))
            tr.step(None, 0.8)
            assert len(tr._graphs) == 1
            with pytest.raises(ValueError):
                tr.set_data(ds.table, plan + 1000)
        res.append((torch.stack(losses), None))
    assert torch.equal(res[0][0], res[1][0])


@pytest.mark.parametrize("F,L,n_states", [(300, 6, 6), (2500, 25, 9), (700, 128, 3), (64, 33, 5)])
def test_device_majority_vote_equals_np_unique(F, L, n_states):
    """rbvae_state_vote (128-bit keys, device vote) == the oracle's np.unique + argmax (percep_RBVAE_train.py:473-497),
    including ties between equally common codes (smallest row wins), an empty state, and 128-bit codes."""
    import sfv_amd as sfv
    gen = torch.Generator().manual_seed(F + L)
    # few distinct codes per state so that counts collide: draw from a small pool of prototype rows
    pool = (torch.rand(7, L, generator=gen) > 0.5).float()
    codes = pool[torch.randint(0, 7, (F,), generator=gen)]
    flip = torch.rand(F, L, generator=gen) < 0.01
    codes = torch.where(flip, 1 - codes, codes)
    labels = torch.randint(0, n_states - 1, (F,), generator=gen).numpy()      # the last state stays empty
    # force an exact tie in state 0: two codes with the same count
    idx0 = np.where(labels == 0)[0]
    if len(idx0) >= 4:
        half = len(idx0) // 2
        codes[idx0[:half]] = pool[0]
        codes[idx0[half:2 * half]] = pool[1]
        codes[idx0[2 * half:]] = pool[2]
    avg, pct = sfv.consistency_from_codes(codes.cuda(), labels, n_states)
    ravg, rpct = O.state_consistency(codes.numpy(), labels, n_states)
    assert abs(avg - ravg) < 1e-12 and np.allclose(pct, rpct), (pct, rpct)
    assert pct[n_states - 1] == 0.0


def test_set_data_rebuilds_when_the_batch_geometry_changes():
    """a new plan with the same number of rows per batch but another (B, T) must not be copied into the old
    plan buffer (the trainer would keep stepping with the old item shape)."""
    import sfv_amd as sfv
    torch.manual_seed(1)
    m = sfv.Seq2SeqBinaryVAE(4, 4, 16, 16, variant="percep", input_hw=(16, 16), compute_dtype="f32").cuda().train()
    tr = sfv.FusedTrainer(m, device_noise=True, use_graph=True, seed=2)
    table = torch.randn(64, 4, 16, 16, device="cuda")
    g = torch.Generator().manual_seed(3)
    plan_a = torch.randint(0, 64, (3, 4, 2, 2), generator=g)          # B = 4, T = 2: 16 rows per batch
    plan_b = torch.randint(0, 64, (3, 2, 2, 4), generator=g)          # B = 2, T = 4: 16 rows per batch too
    tr.set_data(table, plan_a)
    tr.step(None, 0.7)
    assert tr._data[2:] == (4, 2)
    tr.set_data(table, plan_b)
    assert tr._data[2:] == (2, 4) and torch.equal(tr._data[1].cpu(), plan_b.reshape(3, 16))
    tr.step(None, 0.7)
    assert tr.input_buffer(2, 4, 4, 16, 16).shape == (2, 2, 4, 4, 16, 16) and bool(torch.isfinite(tr.losses).all())
    # same geometry again: the plan is copied in place, the graphs stay
    n_graphs = len(tr._graphs)
    buf = tr._data[1].data_ptr()
    tr.set_data(table, torch.randint(0, 64, (3, 2, 2, 4), generator=g))
    assert tr._data[1].data_ptr() == buf and len(tr._graphs) == n_graphs
