"""Device-resident latent table + state-pair sampler and the batched state-consistency evaluator
(SURVEY.md 8f rows 2-3): the callers on either side of the hot path.

  ShuffledStatePairDataset   models/percep_RBVAE/percep_RBVAE_train.py:181-360
      split logic :223-253, pair building :255-310, item assembly :315-335
  calculate_state_consistency / assign_label   :439-497, :362-373

The reference keeps the latents in a host dict and moves one batch per step over PCIe; the whole
table (<= 12 298 frames x 16 KB at 4x32x32) fits in HBM, so here it lives on the device and a batch
[B,2,T,C,H,W] is one gather.  Split / pad / pair logic is reproduced on the host with the same
`random` call sequence, so a given `random.seed` yields the reference's pairs.
"""
from __future__ import annotations

import random
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch


def split_indices(state_segments: Sequence[Tuple[int, int]], test_pct=0.1, val_pct=0.1):
    """Per state: (train, test, val) frame indices -- the contiguous middle chunk is test+val
    (percep_RBVAE_train.py:223-253)."""
    out = []
    for (start, end) in state_segments:
        full = list(range(start, end))
        n = len(full)
        tv = int(n * (test_pct + val_pct))
        margin = (n - tv) // 2
        mid = full[margin:margin + tv]
        train = full[:margin] + full[margin + tv:]
        if tv > 0:
            tc = int(round(test_pct / (test_pct + val_pct) * tv))
            test, val = mid[:tc], mid[tc:]
        else:
            test, val = [], []
        out.append((train, test, val))
    return out


def build_pairs(all_state_indices: Sequence[Sequence[int]]) -> List[List[Tuple[int, int]]]:
    """Pad every state to the longest one, shuffle, pair up (percep_RBVAE_train.py:268-306); uses the
    global `random` module in the reference's call order."""
    max_frames = max((len(ix) for ix in all_state_indices), default=0)
    pairs_per_state = []
    for indices in all_state_indices:
        indices = list(indices)
        if len(indices) < max_frames and len(indices) > 0:
            padded = indices.copy() + random.choices(indices, k=max_frames - len(indices))
        else:
            padded = indices.copy()
        random.shuffle(padded)
        pairs = [(padded[2 * i], padded[2 * i + 1]) for i in range(len(padded) // 2)]
        if len(padded) % 2 == 1:
            leftover = padded[-1]
            cand = random.choice([x for x in indices if x != leftover]) if len(indices) > 1 else leftover
            pairs.append((leftover, cand))
        pairs_per_state.append(pairs)
    return pairs_per_state


def assign_label(frame_index: int, flags: Sequence[int]) -> int:
    """percep_RBVAE_train.py:362-373."""
    label = 0
    for f in flags:
        if frame_index >= f:
            label += 1
        else:
            break
    return label


class DeviceStatePairDataset:
    """ShuffledStatePairDataset with the latents resident in HBM.

    `input_embeddings`: the reference's dict {"%010d.jpg": float32[1,C,H,W]} (get_percep_embeddings.py:106)
    or an [F,C,H,W] array/tensor indexed by frame number.  `batch(idxs)` returns [len(idxs),2,T,C,H,W]."""

    def __init__(self, input_embeddings, state_segments, test_pct=0.1, val_pct=0.1, mode="train", device="cuda"):
        self.mode = mode.lower().strip()
        if self.mode not in ("train", "test", "val"):
            raise ValueError(f"Unknown mode={self.mode}")
        self.state_segments = list(state_segments)
        self.num_states = len(self.state_segments)
        sp = split_indices(self.state_segments, test_pct, val_pct)
        self.train_indices_per_state = [s[0] for s in sp]
        self.test_indices_per_state = [s[1] for s in sp]
        self.val_indices_per_state = [s[2] for s in sp]
        chosen = {"train": self.train_indices_per_state, "test": self.test_indices_per_state,
                  "val": self.val_indices_per_state}[self.mode]
        self.pairs_per_state = build_pairs(chosen)
        self.num_items = max((len(p) for p in self.pairs_per_state), default=0)
        self.device = torch.device(device)
        self._load_table(input_embeddings)
        # [num_items, 2, T] frame rows of the table, with the reference's wrap-around (:323-324)
        idx = np.zeros((self.num_items, 2, self.num_states), dtype=np.int64)
        for s, pairs in enumerate(self.pairs_per_state):
            if len(pairs) == 0:
                raise ValueError(f"State {s} has no pairs")
            for i in range(self.num_items):
                a, b = pairs[i % len(pairs)]
                idx[i, 0, s], idx[i, 1, s] = self._row(a), self._row(b)
        self.index = torch.from_numpy(idx).to(self.device)

    def _load_table(self, emb):
        if isinstance(emb, dict):
            frames = sorted({i for seg in self.state_segments for i in range(seg[0], seg[1])})
            rows = {}
            buf = []
            for fi in frames:
                e = emb.get(f"{fi:010d}.jpg")
                if e is None:
                    e = emb.get(f"{fi:010d}")
                if e is None:
                    raise KeyError(f"No embedding found for frame index {fi}")
                rows[fi] = len(buf)
                buf.append(torch.as_tensor(np.asarray(e), dtype=torch.float32).squeeze())
            self._rows = rows
            self.table = torch.stack(buf).to(self.device)
        else:
            self._rows = None
            self.table = torch.as_tensor(emb, dtype=torch.float32).to(self.device)

    def _row(self, frame_index: int) -> int:
        if self._rows is None:
            if not 0 <= frame_index < self.table.shape[0]:
                raise KeyError(f"No embedding found for frame index {frame_index}")
            return frame_index
        return self._rows[frame_index]

    def __len__(self):
        return self.num_items

    def batch(self, item_indices, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """[B,2,T,C,H,W] on the device (one gather; the reference's DataLoader + .to(device), :509-518).
        `out`: gather in place into an existing [B,2,T,C,H,W] tensor -- FusedTrainer.input_buffer(), which the
        captured training step reads directly, so a step moves the batch exactly once."""
        ii = torch.as_tensor(item_indices, dtype=torch.long, device=self.device)
        rows = self.index[ii]                                    # [B,2,T]
        shape = (*rows.shape, *self.table.shape[1:])
        if out is None:
            return self.table[rows.reshape(-1)].reshape(shape)
        if tuple(out.shape) != shape or out.dtype != self.table.dtype or out.device != self.table.device or \
                not out.is_contiguous():
            raise ValueError(f"out must be a contiguous {self.table.dtype} tensor of shape {shape} on {self.table.device}")
        torch.index_select(self.table, 0, rows.reshape(-1), out=out.view(-1, *self.table.shape[1:]))
        return out

    def plan(self, order, batch_size: int) -> torch.Tensor:
        """An epoch's full batches laid out in advance: [n_batches, B, 2, T] table rows (int64, on the device) for the
        item order `order` (e.g. a torch.randperm of len(self), the DataLoader's shuffle).  The ragged last batch of
        len(order) % batch_size items is not in the plan (fetch it with batch()).  FusedTrainer.set_data(table, plan)
        trains from it without any host work per step."""
        ii = torch.as_tensor(order, dtype=torch.long, device=self.device)
        nb = ii.numel() // batch_size
        if nb == 0:
            raise ValueError(f"{ii.numel()} items do not fill one batch of {batch_size}")
        return self.index[ii[:nb * batch_size]].reshape(nb, batch_size, 2, self.num_states)

    def __getitem__(self, idx) -> torch.Tensor:
        return self.batch([idx])[0]

    def frames(self, frame_indices) -> torch.Tensor:
        rows = torch.as_tensor([self._row(int(i)) for i in frame_indices], dtype=torch.long, device=self.device)
        return self.table[rows]


def consistency_from_codes(codes: torch.Tensor, labels, n_states: int):
    """percep_RBVAE_train.py:473-497 on a device tensor of {0,1} codes [F, L]: per state the share of frames
    equal to the state's most common code (ties: lexicographically smallest, like np.unique), then the
    count-weighted mean.  Returns (weighted_avg, percentages)."""
    if codes.is_cuda:
        # the device path: 128-bit keys + vote kernels (csrc/eval.hip), one tiny result copy
        from . import _lib as L
        F, Ld = codes.shape
        if F == 0:
            return 0, [0.0] * n_states
        lab = torch.as_tensor(np.asarray(labels), dtype=torch.int32).to(codes.device)
        keys = torch.empty(F, 4, dtype=torch.int32, device=codes.device)
        cnt = torch.empty(F, dtype=torch.int32, device=codes.device)
        out = torch.empty(n_states, 2, dtype=torch.int32, device=codes.device)
        L.call("rbvae_state_vote", codes.float().contiguous(), lab, F, Ld, n_states, keys, cnt, out)
        res = out.cpu().numpy()
        pct = [float(b / n) if n > 0 else 0.0 for b, n in res]
        total = int(res[:, 1].sum())
        return (float(np.dot(pct, res[:, 1]) / total) if total > 0 else 0), pct
    # host tensors (the reference's own np.unique path, percep_RBVAE_train.py:473-497)
    labels = torch.as_tensor(np.asarray(labels), device=codes.device)
    pct: List[float] = []
    counts: List[int] = []
    for s in range(n_states):
        rows = codes[labels == s]
        counts.append(int(rows.shape[0]))
        if rows.shape[0] == 0:
            pct.append(0.0)
            continue
        uniq, cnt = torch.unique(rows, dim=0, return_counts=True)
        top = uniq[int(torch.argmax(cnt))]
        pct.append(float((rows == top).all(dim=1).double().mean()))
    total = sum(counts)
    avg = float(np.dot(pct, counts) / total) if total > 0 else 0
    return avg, pct


@torch.no_grad()
def state_consistency(model, dataset: DeviceStatePairDataset, flags: Sequence[int], temperature: float,
                      noise_ratio: float = 0.1, batch: int = 4096, u: Optional[torch.Tensor] = None):
    """calculate_state_consistency (percep_RBVAE_train.py:439-497) with every validation frame encoded in
    batches of B = `batch`, T = 1 (the reference encodes them one at a time with B = T = 1; sequences of
    one state are independent, so the codes are the same given the same noise draws)."""
    was_training = model.training
    model.eval()
    val = [i for ix in dataset.val_indices_per_state for i in ix]
    labels = np.array([assign_label(i, flags) for i in val])
    if u is None:
        u = torch.rand((len(val), model.latent_dim))          # the reference's per-frame draws, in order
    codes = []
    for s in range(0, len(val), batch):
        x = dataset.frames(val[s:s + batch])[:, None]        # [b,1,C,H,W]
        z = model.encode(x, temperature=temperature, hard=True, noise_ratio=noise_ratio,
                         u=u[s:s + batch].to(x.device))
        codes.append(z[:, 0])
    codes = torch.cat(codes) if codes else torch.zeros(0, model.latent_dim, device=dataset.device)
    model.train(was_training)
    return consistency_from_codes(codes, labels, len(flags) + 1)
