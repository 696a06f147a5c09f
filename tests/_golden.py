"""Helpers shared by the golden-fixture tests (CPU oracle and HIP path)."""
import os

import numpy as np
import torch

import rbvae_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

MODEL_CASES = [
    "contrastive_small_eval", "contrastive_small_train", "contrastive_small_hard",
    "triplet_small_eval", "percep_small_eval", "percep_small_train",
    "percep_native_eval", "contrastive_native_eval",
]


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def case_params(g):
    """Re-draw the case's parameters from its seed and check them against the
    checksums recorded when the reference ran."""
    variant = str(g["meta/variant"])
    in_ch, L = int(g["meta/in_ch"]), int(g["meta/L"])
    hw = tuple(int(v) for v in g["meta/hw"])
    torch.manual_seed(int(g["meta/seed"]))
    p = O.init_params(variant, in_ch, in_ch, L, hw)
    for k, v in p.items():
        cs = g[f"paramsum/{k}"]
        assert abs(float(v.double().sum()) - cs[0]) <= 1e-9 * max(1.0, cs[1]) and abs(float(v.double().abs().sum()) - cs[1]) <= 1e-9 * cs[1], k
    return variant, p


def case_masks(g, view):
    masks = []
    j = 0
    while f"mask{view}_{j}" in g.files:
        shp = tuple(int(v) for v in g[f"maskshape{view}_{j}"])
        n = int(np.prod(shp))
        bits = np.unpackbits(g[f"mask{view}_{j}"])[:n].reshape(shp)
        masks.append(torch.from_numpy(bits.astype(np.float32)))
        j += 1
    return masks or None


def sample_idx(n, stride=97):
    return np.arange(0, n, stride)
