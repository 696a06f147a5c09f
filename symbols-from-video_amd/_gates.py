"""ReLU-tie handling shared by the gradient parity checks (tests/ and __graft_entry__.smoke()).

Millions of pre-activations pass through a ReLU per step; the few that lie within rounding of zero may be resolved
either way by two correct implementations, and ONE tie resolved differently moves a small-norm gradient tensor (a sum of
cancelling terms) by 1/sqrt(#terms) -- far above the 1e-4 parity gate.  The tests therefore hand the oracle the
device's own ReLU decisions (rbvae_oracle._relu) after checking that they differ from sign(oracle pre-activation) only
where that pre-activation is within rounding of zero."""
import torch


def device_gates(tr, B, T):
    """The device's ReLU decisions of the last step, as the oracle wants them: [view][site] 0/1 tensors [B*T,C,H,W]
    (sites: conv1, conv2, deconv0, deconv1 outputs).  The saved activations are relu(x) * mask / 0.8 in NHWC rows,
    sequences ordered (view, item): a > 0 <=> gate open (where the dropout mask is 0 the gate is irrelevant)."""
    sv, eng = tr.last_saved, tr.eng
    (h1, w1), (h2, w2) = eng.g1, eng.g2
    out = [[], []]
    for a, (h, w) in ((sv.a1, (h1, w1)), (sv.a2, (h2, w2)), (sv.d1, (h2, w2)), (sv.d2, (h1, w1))):
        g = (a.float() > 0).view(2, B * T, h, w, -1).permute(0, 1, 4, 2, 3).cpu()
        out[0].append(g[0])
        out[1].append(g[1])
    return out


def count_ties(pre, gates, masks, bound):
    """pre / gates: [view][site]; masks: dropout keep-masks [view][site] or None.  Asserts that every differing decision
    sits on a pre-activation smaller than `bound`; returns their number."""
    ties = 0
    for vw in range(2):
        for j in range(4):
            x = pre[vw][j]
            differ = (x > 0) != gates[vw][j]
            if masks is not None:
                differ &= masks[vw][j] > 0
            ties += int(differ.sum())
            if differ.any():
                assert float(x[differ].abs().max()) < bound, (vw, j, float(x[differ].abs().max()))
    return ties
