"""Host logic of the device-resident dataset and the consistency evaluator against the reference's own
dataset (fixtures from tools/make_golden.py) and the oracle.  CPU only."""
import random

import numpy as np
import pytest
import torch

import rbvae_oracle as O
from _golden import load


@pytest.fixture(scope="module")
def data():
    import importlib
    import sfv_amd  # noqa: F401  (puts the repo root on sys.path)
    return importlib.import_module("symbols-from-video_amd.data")


def test_split_and_pairs_match_reference(data):
    g = load("trainer")
    segs = [tuple(int(v) for v in r) for r in g["split/segs"]]
    tp, vp = (float(v) for v in g["split/pcts"])
    sp = data.split_indices(segs, tp, vp)
    for si in range(len(segs)):
        assert sp[si][0] == list(g[f"split/{si}/train"])
        assert sp[si][1] == list(g[f"split/{si}/test"])
        assert sp[si][2] == list(g[f"split/{si}/val"])
    emb = {f"{i:010d}.jpg": np.full((1, 3, 2, 2), float(i), dtype=np.float32) for i in range(200)}
    random.seed(77)
    ds = data.DeviceStatePairDataset(emb, segs, tp, vp, mode="train", device="cpu")
    assert len(ds) == int(g["pairs/train/len"])
    for si in range(len(segs)):
        assert np.array_equal(np.array(ds.pairs_per_state[si]).reshape(-1, 2), g[f"pairs/train/{si}"])
    np.testing.assert_array_equal(ds[3].numpy(), g["pairs/train/item3"])
    b = ds.batch([0, 3, 5])
    assert b.shape == (3, 2, len(segs), 3, 2, 2)
    np.testing.assert_array_equal(b[1].numpy(), g["pairs/train/item3"])
    random.seed(77)
    with pytest.raises(ValueError):                    # the 1-frame state has no validation frames (:325)
        data.DeviceStatePairDataset(emb, segs, tp, vp, mode="val", device="cpu")
    with pytest.raises(ValueError):
        data.DeviceStatePairDataset(emb, segs, mode="bogus", device="cpu")
    with pytest.raises(KeyError):
        data.DeviceStatePairDataset({}, segs, device="cpu")


def test_labels_and_consistency(data):
    g = load("trainer")
    flags = [int(v) for v in g["label/flags"]]
    assert [data.assign_label(int(i), flags) for i in g["label/idx"]] == list(g["label/out"])
    gen = torch.Generator().manual_seed(1)
    codes = (torch.rand(300, 6, generator=gen) > 0.7).float()
    labels = torch.randint(0, 5, (300,), generator=gen).numpy()
    avg, pct = data.consistency_from_codes(codes, labels, 6)
    ravg, rpct = O.state_consistency(codes.numpy(), labels, 6)
    assert abs(avg - ravg) < 1e-12 and np.allclose(pct, rpct)
