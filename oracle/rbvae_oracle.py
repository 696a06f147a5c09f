"""CPU oracle for the RBVAE hot path (TEST INFRASTRUCTURE, not product code).

This file is a plain-torch fp32 CPU restatement of the reference's per-frame
encode -> LSTM -> binarise -> LSTM -> decode path and of its loss reductions.
It exists so that the HIP path can be checked on a box that does not hold
/root/reference.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import it; the product package never does.

Pinning: the reference ships no tests or golden vectors for this path
(SURVEY.md section 4), so the oracle is pinned by fixtures generated in the
dev container by importing the reference itself (tools/make_golden.py ->
tests/golden/*.npz; tests/test_oracle_golden.py re-checks them on CPU).

Reference citations (paths relative to the reference repo root):
  forward/encode      models/percep_RBVAE/percep_RBVAE_model.py:143-191
                      models/contrastive_RBVAE/contrastive_RBVAE_model.py:142-190
                      models/triplet_RBVAE/triplet_RBVAE_model.py:144-193
                      models/simple_RBVAE/simple_RBVAE_model.py:160-193
  binarise            models/percep_RBVAE/percep_RBVAE_model.py:17-44
                      (triplet: no noise_ratio, triplet_RBVAE_model.py:18-45;
                       simple: eps 1e-10, simple_RBVAE_model.py:17-44)
  losses              models/percep_RBVAE/percep_RBVAE_train.py:28-107
                      models/simple_RBVAE/simple_RBVAE_train.py:45-68
                      models/triplet_RBVAE/triplet_RBVAE_train.py:82-96
  step composition    models/percep_RBVAE/percep_RBVAE_train.py:525-549
                      models/triplet_RBVAE/triplet_RBVAE_train.py:452-474
  validation total    models/percep_RBVAE/percep_RBVAE_train.py:590-635
  temperature         models/percep_RBVAE/percep_RBVAE_train.py:424-437
  state consistency   models/percep_RBVAE/percep_RBVAE_train.py:473-497
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------
# model geometry
# --------------------------------------------------------------------------
@dataclass(frozen=True)
class Variant:
    """Static description of one of the reference's four model files."""
    name: str
    channels: Tuple[int, int, int]   # encoder conv widths (decoder mirrors them)
    kernel: int                      # 3 (percep/contrastive/triplet) or 4 (simple)
    lstm_layers: int
    dropout: float                   # 0.2, or 0.0 for simple
    noise_ratio_arg: bool            # False: noise is never scaled (triplet, simple)
    eps: float                       # inside the logistic-noise logs
    order: str                       # "cnn-rnn-bin-rnn-cnn" or "cnn-bin-rnn-rnn-cnn" (simple)
    default_hw: Tuple[int, int]      # spatial size the reference hard-codes its fc for


VARIANTS: Dict[str, Variant] = {
    "percep": Variant("percep", (256, 256, 256), 3, 4, 0.2, True, 1e-8,
                      "cnn-rnn-bin-rnn-cnn", (88, 160)),
    "contrastive": Variant("contrastive", (64, 64, 64), 3, 2, 0.2, True, 1e-8,
                           "cnn-rnn-bin-rnn-cnn", (256, 256)),
    "triplet": Variant("triplet", (64, 64, 64), 3, 2, 0.2, False, 1e-8,
                       "cnn-rnn-bin-rnn-cnn", (256, 256)),
    "simple": Variant("simple", (64, 128, 256), 4, 1, 0.0, False, 1e-10,
                      "cnn-bin-rnn-rnn-cnn", (64, 64)),
}


def conv_out_hw(hw: Tuple[int, int], k: int) -> Tuple[int, int]:
    """Spatial size after one stride-2, pad-1 convolution with a k x k kernel."""
    return ((hw[0] + 2 - k) // 2 + 1, (hw[1] + 2 - k) // 2 + 1)


def bottleneck_hw(v: Variant, hw: Tuple[int, int]) -> Tuple[int, int]:
    for _ in range(3):
        hw = conv_out_hw(hw, v.kernel)
    return hw


def param_shapes(v: Variant, in_ch: int, out_ch: int, L: int,
                 hw: Tuple[int, int]) -> "Dict[str, Tuple[int, ...]]":
    """state_dict keys and shapes, in the reference's registration order."""
    c1, c2, c3 = v.channels
    k = v.kernel
    bh, bw = bottleneck_hw(v, hw)
    flat = c3 * bh * bw
    if v.name == "simple":
        conv_idx, deconv_idx = (0, 2, 4), (0, 2, 4)
    else:
        conv_idx, deconv_idx = (0, 3, 6), (0, 3, 6)
    s: Dict[str, Tuple[int, ...]] = {}
    enc = [(in_ch, c1), (c1, c2), (c2, c3)]
    for i, (ci, co) in zip(conv_idx, enc):
        s[f"encoder_cnn.conv.{i}.weight"] = (co, ci, k, k)
        s[f"encoder_cnn.conv.{i}.bias"] = (co,)
    s["encoder_cnn.fc.weight"] = (L, flat)
    s["encoder_cnn.fc.bias"] = (L,)
    s["decoder_cnn.fc.weight"] = (flat, L)
    s["decoder_cnn.fc.bias"] = (flat,)
    dec = [(c3, c2), (c2, c1), (c1, out_ch)]
    for i, (ci, co) in zip(deconv_idx, dec):
        s[f"decoder_cnn.deconv.{i}.weight"] = (ci, co, k, k)   # ConvTranspose2d layout
        s[f"decoder_cnn.deconv.{i}.bias"] = (co,)
    for stack in ("encoder_rnn", "decoder_rnn"):
        for l in range(v.lstm_layers):
            s[f"{stack}.lstm.weight_ih_l{l}"] = (4 * L, L)
            s[f"{stack}.lstm.weight_hh_l{l}"] = (4 * L, L)
            s[f"{stack}.lstm.bias_ih_l{l}"] = (4 * L,)
            s[f"{stack}.lstm.bias_hh_l{l}"] = (4 * L,)
    return s


# --------------------------------------------------------------------------
# forward pieces
# --------------------------------------------------------------------------
def binarize(h: Tensor, U: Tensor, temperature: float, hard: bool,
             noise_ratio: float, eps: float) -> Tensor:
    """Binary-Concrete sample from logits `h` given uniform noise `U` in [0,1).

    Follows percep_RBVAE_model.py:33-42 with U made an explicit argument (the
    reference draws it from the CPU default generator at :33)."""
    noise = noise_ratio * (torch.log(U + eps) - torch.log(1.0 - U + eps))
    y = torch.sigmoid((h + noise) / temperature)
    if hard:
        y_hard = (y > 0.5).to(y.dtype)
        y = (y_hard - y).detach() + y
    return y


def lstm_stack(x: Tensor, p: Dict[str, Tensor], prefix: str, layers: int) -> Tensor:
    """Stacked LSTM, batch_first, zero initial state; returns the top layer's
    hidden state at every step.  Restates torch.nn.LSTM's documented cell
    (gate order i, f, g, o) as used at percep_RBVAE_model.py:100,115."""
    B, T, L = x.shape
    seq = x
    for l in range(layers):
        w_ih = p[f"{prefix}.lstm.weight_ih_l{l}"]
        w_hh = p[f"{prefix}.lstm.weight_hh_l{l}"]
        b_ih = p[f"{prefix}.lstm.bias_ih_l{l}"]
        b_hh = p[f"{prefix}.lstm.bias_hh_l{l}"]
        H = w_hh.shape[1]
        h = x.new_zeros(B, H)
        c = x.new_zeros(B, H)
        outs = []
        for t in range(T):
            g = seq[:, t] @ w_ih.t() + b_ih + h @ w_hh.t() + b_hh
            i, f, gg, o = g.chunk(4, dim=1)
            c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
            h = torch.sigmoid(o) * torch.tanh(c)
            outs.append(h)
        seq = torch.stack(outs, dim=1)
    return seq


def _drop(a: Tensor, rate: float, train: bool, mask: Optional[Tensor]) -> Tensor:
    """nn.Dropout: identity in eval; in train `mask` (1 = keep) is explicit."""
    if not train or rate == 0.0:
        return a
    if mask is None:
        mask = (torch.rand_like(a) >= rate).to(a.dtype)
    return a * mask / (1.0 - rate)


def _relu(a: Tensor, gates, j: int, pre: Optional[list]) -> Tensor:
    """ReLU, or -- parity diagnostics -- multiplication by a given 0/1 gate (the device's own ReLU decisions), so that
    a pre-activation within rounding of zero cannot make two correct implementations disagree by a whole element's
    gradient.  `pre` collects the pre-activations so a test can check that given gates differ from sign(x) only
    where |x| is within rounding of zero."""
    if pre is not None:
        pre.append(a.detach())
    if gates is None:
        return torch.relu(a)
    return a * gates[j].to(a.dtype)


def encoder_cnn(v: Variant, p: Dict[str, Tensor], X: Tensor, train: bool,
                masks: Optional[Sequence[Tensor]], gates=None, pre: Optional[list] = None) -> Tensor:
    k = v.kernel
    idx = (0, 2, 4) if v.name == "simple" else (0, 3, 6)
    a = X
    for j, i in enumerate(idx):
        a = F.conv2d(a, p[f"encoder_cnn.conv.{i}.weight"], p[f"encoder_cnn.conv.{i}.bias"],
                     stride=2, padding=1)
        last = j == 2
        if not last or v.name == "simple":
            a = _relu(a, gates if not last else None, j, pre if not last else None)
        if not last and v.dropout > 0:
            a = _drop(a, v.dropout, train, None if masks is None else masks[j])
    flat = a.flatten(1)                       # NCHW flatten order
    return flat @ p["encoder_cnn.fc.weight"].t() + p["encoder_cnn.fc.bias"]


def decoder_cnn(v: Variant, p: Dict[str, Tensor], d: Tensor, hw_b: Tuple[int, int],
                train: bool, masks: Optional[Sequence[Tensor]], gates=None, pre: Optional[list] = None) -> Tensor:
    k = v.kernel
    idx = (0, 2, 4) if v.name == "simple" else (0, 3, 6)
    op = 1 if k == 3 else 0
    f = d @ p["decoder_cnn.fc.weight"].t() + p["decoder_cnn.fc.bias"]
    a = f.reshape(d.shape[0], v.channels[2], hw_b[0], hw_b[1])
    for j, i in enumerate(idx):
        a = F.conv_transpose2d(a, p[f"decoder_cnn.deconv.{i}.weight"],
                               p[f"decoder_cnn.deconv.{i}.bias"],
                               stride=2, padding=1, output_padding=op)
        if j < 2:
            a = _relu(a, gates, 2 + j, pre)
            if v.dropout > 0:
                a = _drop(a, v.dropout, train, None if masks is None else masks[2 + j])
        else:
            a = torch.sigmoid(a)
    return a


def forward(variant: str, p: Dict[str, Tensor], x: Tensor, U: Tensor,
            temperature: float = 1.0, hard: bool = False, noise_ratio: float = 0.1,
            train: bool = False, masks: Optional[Sequence[Tensor]] = None, gates=None,
            pre: Optional[list] = None):
    """Seq2SeqBinaryVAE.forward.  x: [B,T,C,H,W]; U: [B*T, L] uniform noise.
    gates / pre: parity diagnostics, see _relu (4 gate tensors: conv1, conv2, deconv0, deconv1 outputs, NCHW).

    Returns (x_recon, h_seq, z_seq) for percep/contrastive/triplet and
    (x_recon, logits) for simple, exactly like the reference modules."""
    v = VARIANTS[variant]
    B, T, C, H, W = x.shape
    L = p["encoder_cnn.fc.bias"].shape[0]
    hw_b = bottleneck_hw(v, (H, W))
    r = noise_ratio if v.noise_ratio_arg else 1.0
    X = x.reshape(B * T, C, H, W)
    e = encoder_cnn(v, p, X, train, masks, gates, pre)          # [B*T, L]
    if v.order == "cnn-rnn-bin-rnn-cnn":
        hs = lstm_stack(e.reshape(B, T, L), p, "encoder_rnn", v.lstm_layers)
        z = binarize(hs.reshape(B * T, L), U, temperature, hard, r, v.eps)
        ds = lstm_stack(z.reshape(B, T, L), p, "decoder_rnn", v.lstm_layers)
        xr = decoder_cnn(v, p, ds.reshape(B * T, L), hw_b, train, masks, gates, pre)
        return xr.reshape(B, T, -1, H, W), hs, z.reshape(B, T, L)
    z = binarize(e, U, temperature, hard, r, v.eps)
    hs = lstm_stack(z.reshape(B, T, L), p, "encoder_rnn", v.lstm_layers)
    ds = lstm_stack(hs, p, "decoder_rnn", v.lstm_layers)
    xr = decoder_cnn(v, p, ds.reshape(B * T, L), hw_b, train, masks)
    return xr.reshape(B, T, -1, H, W), e


def encode(variant: str, p: Dict[str, Tensor], x: Tensor, U: Tensor,
           temperature: float = 0.5, hard: bool = False, noise_ratio: float = 0.1, train: bool = False,
           masks: Optional[Sequence[Tensor]] = None) -> Tensor:
    """Seq2SeqBinaryVAE.encode (percep_RBVAE_model.py:172-191).  The reference's encode() calls self.encoder_cnn in
    whatever mode the module is in: train=True = its two Dropout layers live (explicit keep-masks in `masks`)."""
    v = VARIANTS[variant]
    B, T, C, H, W = x.shape
    L = p["encoder_cnn.fc.bias"].shape[0]
    r = noise_ratio if v.noise_ratio_arg else 1.0
    e = encoder_cnn(v, p, x.reshape(B * T, C, H, W), train, masks)
    hs = lstm_stack(e.reshape(B, T, L), p, "encoder_rnn", v.lstm_layers)
    z = binarize(hs.reshape(B * T, L), U, temperature, hard, r, v.eps)
    return z.reshape(B, T, L)


# --------------------------------------------------------------------------
# losses
# --------------------------------------------------------------------------
def recon_loss(x_recon: Tensor, x: Tensor) -> Tensor:
    """percep_RBVAE_train.py:32-33."""
    return ((x_recon - x) ** 2).mean()


def kl_binary_concrete(q_logits: Tensor, p: float = 0.5, eps: float = 1e-8,
                       clamp: bool = True) -> Tensor:
    """percep_RBVAE_train.py:52-76 (clamp=True) and simple_RBVAE_train.py:45-68
    (clamp=False, eps=1e-10)."""
    q = torch.sigmoid(q_logits)
    if clamp:
        q = q.clamp(eps, 1.0 - eps)
    log_p = float(np.log(p))
    log_1mp = float(np.log(1.0 - p))
    kl = q * (torch.log(q + eps) - log_p) + (1.0 - q) * (torch.log((1.0 - q) + eps) - log_1mp)
    return kl.sum(dim=-1).mean()


def pairwise_distance(a: Tensor, b: Tensor, eps: float = 1e-6) -> Tensor:
    """F.pairwise_distance: || a - b + eps ||_2 over the last dim."""
    return torch.sqrt(((a - b + eps) ** 2).sum(dim=-1))


def contrast_loss(x1: Tensor, x2: Tensor, label: int, margin: float = 1.0) -> Tensor:
    """percep_RBVAE_train.py:79-107, 'euclidean' branch."""
    d = pairwise_distance(x1, x2)
    return ((1 - label) * d ** 2 + label * torch.clamp(margin - d, min=0.0) ** 2).mean()


def triplet_loss(a: Tensor, pos: Tensor, neg: Tensor, margin: float = 1.0,
                 eps: float = 1e-8, swap: bool = True) -> Tensor:
    """F.triplet_margin_loss(p=2, eps, swap) as called at triplet_RBVAE_train.py:82-96."""
    d_ap = pairwise_distance(a, pos, eps)
    d_an = pairwise_distance(a, neg, eps)
    if swap:
        d_an = torch.minimum(d_an, pairwise_distance(pos, neg, eps))
    return torch.clamp(margin + d_ap - d_an, min=0.0).mean()


def l1_loss(q_logits: Tensor, lamb: float) -> Tensor:
    return lamb * q_logits.abs().sum()


def contrast_term(h0: Tensor, h1: Tensor) -> Tensor:
    """percep_RBVAE_train.py:534-543: similar term over both views plus the mean
    over adjacent state pairs of view 0 (margin is never passed => 1.0)."""
    T = h0.shape[1]
    sim = contrast_loss(h0, h1, 0)
    dis = sum(contrast_loss(h0[:, s], h0[:, s + 1], 1) for s in range(T - 1)) / float(T - 1)
    return sim + dis


def triplet_term(h0: Tensor, h1: Tensor, margin: float) -> Tensor:
    """triplet_RBVAE_train.py:461-468."""
    T = h0.shape[1]
    tot = sum(triplet_loss(h0[:, s], h1[:, s], h0[:, s + 1], margin) for s in range(T - 1))
    return tot / float(T - 1)


def step_losses(variant: str, p: Dict[str, Tensor], item: Tensor, U: Sequence[Tensor],
                temperature: float, noise_ratio: float = 0.1, bernoulli_p: float = 0.5,
                alpha: float = 0.1, beta: float = 0.1, margin: float = 1.0,
                hard: bool = False, train: bool = False,
                masks: Optional[Sequence[Sequence[Tensor]]] = None,
                validation_norm: bool = False, pair_loss: Optional[str] = None, gates=None,
                pre: Optional[list] = None) -> Dict[str, Tensor]:
    """One trainer step's loss for item [B,2,T,C,H,W] (percep_RBVAE_train.py:517-549;
    validation weighting :590-635 when validation_norm).
    pair_loss: "contrast" | "triplet" overrides the variant's own pairwise term (BASELINE configs[4]: the
    percep-shaped network trained with the triplet term of triplet_RBVAE_train.py:461-468)."""
    recons, kls, hs = [], [], []
    for vw in range(2):
        pv = [] if pre is not None else None
        xr, h, z = forward(variant, p, item[:, vw], U[vw], temperature, hard, noise_ratio,
                           train, None if masks is None else masks[vw], None if gates is None else gates[vw], pv)
        if pre is not None:
            pre.append(pv)
        recons.append(recon_loss(xr, item[:, vw]))
        kls.append(kl_binary_concrete(z, bernoulli_p))
        hs.append(h)
    recon = (recons[0] + recons[1]) / 2
    kl = (kls[0] + kls[1]) / 2
    if (pair_loss or ("triplet" if variant == "triplet" else "contrast")) == "triplet":
        pair = triplet_term(hs[0], hs[1], margin)
    else:
        pair = contrast_term(hs[0], hs[1])
    if validation_norm:
        s = 1.0 + alpha + beta
        total = recon / s + (beta / s) * kl + (alpha / s) * pair
    else:
        total = recon + beta * kl + alpha * pair
    return {"total": total, "recon": recon, "kl": kl, "pair": pair}


def simple_step_loss(p: Dict[str, Tensor], x: Tensor, U: Tensor, temperature: float = 0.5,
                     beta: float = 0.1, bernoulli_p: float = 0.1) -> Dict[str, Tensor]:
    """simple_RBVAE_train.py:173-182."""
    xr, logits = forward("simple", p, x, U, temperature, False)
    recon = recon_loss(xr, x)
    kl = kl_binary_concrete(logits, bernoulli_p, eps=1e-10, clamp=False)
    return {"total": recon + beta * kl, "recon": recon, "kl": kl}


# --------------------------------------------------------------------------
# optimiser / trainer-side scalar logic
# --------------------------------------------------------------------------
def adam_step(params: Dict[str, Tensor], grads: Dict[str, Tensor], state: Dict[str, Dict],
              lr: float, step: int, betas=(0.9, 0.999), eps: float = 1e-8) -> None:
    """torch.optim.Adam defaults (no weight decay, no amsgrad), in place."""
    b1, b2 = betas
    for k, w in params.items():
        g = grads[k]
        st = state.setdefault(k, {"m": torch.zeros_like(w), "v": torch.zeros_like(w)})
        st["m"].mul_(b1).add_(g, alpha=1 - b1)
        st["v"].mul_(b2).addcmul_(g, g, value=1 - b2)
        bc1 = 1 - b1 ** step
        bc2 = 1 - b2 ** step
        denom = (st["v"].sqrt() / math.sqrt(bc2)).add_(eps)
        w.addcdiv_(st["m"], denom, value=-lr / bc1)


def temperature_schedule(global_step: int, current: float, init: float, final: float,
                         anneal_rate: float, every: int) -> float:
    """percep_RBVAE_train.py:424-437."""
    if global_step % every == 0:
        return max(final, init * float(np.exp(-anneal_rate * global_step)))
    return current


def assign_label(idx: int, flags: Sequence[int]) -> int:
    """percep_RBVAE_train.py:362-373: index of the first flag greater than idx."""
    for i, f in enumerate(flags):
        if idx < f:
            return i
    return len(flags)


def state_consistency(codes: np.ndarray, labels: np.ndarray, n_states: int):
    """percep_RBVAE_train.py:473-497: per state, share of frames whose code equals
    the state's most common code; then the count-weighted mean."""
    pct: List[float] = []
    counts: List[int] = []
    for s in range(n_states):
        rows = codes[labels == s]
        counts.append(int(rows.shape[0]))
        if rows.shape[0] == 0:
            pct.append(0.0)
            continue
        uniq, cnt = np.unique(rows, axis=0, return_counts=True)
        top = uniq[np.argmax(cnt)]
        pct.append(float(np.mean(np.all(rows == top, axis=1))))
    tot = sum(counts)
    avg = float(np.dot(pct, counts) / tot) if tot > 0 else 0.0
    return avg, pct


# --------------------------------------------------------------------------
# parameter initialisation (PyTorch module defaults, reference registration order)
# --------------------------------------------------------------------------
def init_params(variant: str, in_ch: int, out_ch: int, L: int,
                hw: Optional[Tuple[int, int]] = None, seed: Optional[int] = None) -> Dict[str, Tensor]:
    """Parameters drawn exactly as the reference's constructor draws them
    (percep_RBVAE_model.py:135-141: encoder_cnn, decoder_cnn, encoder_rnn,
    decoder_rnn, each with torch's default reset_parameters), so that under
    the same torch.manual_seed the state_dict equals the reference's bit for
    bit when `hw` is the reference's hard-coded size."""
    import torch.nn as nn
    v = VARIANTS[variant]
    hw = hw or v.default_hw
    if seed is not None:
        torch.manual_seed(seed)
    c1, c2, c3 = v.channels
    k = v.kernel
    bh, bw = bottleneck_hw(v, hw)
    flat = c3 * bh * bw
    op = 1 if k == 3 else 0
    mods = {}
    idx = (0, 2, 4) if v.name == "simple" else (0, 3, 6)
    for i, (ci, co) in zip(idx, [(in_ch, c1), (c1, c2), (c2, c3)]):
        mods[f"encoder_cnn.conv.{i}"] = nn.Conv2d(ci, co, k, 2, 1)
    mods["encoder_cnn.fc"] = nn.Linear(flat, L)
    mods["decoder_cnn.fc"] = nn.Linear(L, flat)
    for i, (ci, co) in zip(idx, [(c3, c2), (c2, c1), (c1, out_ch)]):
        mods[f"decoder_cnn.deconv.{i}"] = nn.ConvTranspose2d(ci, co, k, 2, 1, output_padding=op)
    mods["encoder_rnn.lstm"] = nn.LSTM(L, L, v.lstm_layers, batch_first=True)
    mods["decoder_rnn.lstm"] = nn.LSTM(L, L, v.lstm_layers, batch_first=True)
    out: Dict[str, Tensor] = {}
    for prefix, m in mods.items():
        for name, prm in m.named_parameters():
            out[f"{prefix}.{name}"] = prm.detach().clone()
    want = param_shapes(v, in_ch, out_ch, L, hw)
    assert list(out.keys()) == list(want.keys()), "registration order drifted"
    for k_, shp in want.items():
        assert tuple(out[k_].shape) == shp, (k_, out[k_].shape, shp)
    return out
