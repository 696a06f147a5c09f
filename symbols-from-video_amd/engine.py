"""Hand-scheduled forward/backward of the RBVAE path over librbvae_hip.

The engine owns no parameters: it is given the model's flat f32 parameter buffer
(the reference's registration order, see `ParamLayout`) and writes gradients into a
flat buffer of the same layout.  Activations are NHWC in the engine's storage dtype
("f32" parity mode / "bf16" performance mode); LSTM, binarise and the losses are f32.

Reference path restated by the kernels this file sequences:
  Seq2SeqBinaryVAE.forward / encode   models/percep_RBVAE/percep_RBVAE_model.py:143-191
  (contrastive / triplet / simple variants: the matching *_model.py files)
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib as L

F32, BF16 = 0, 1


@dataclass(frozen=True)
class Variant:
    name: str
    channels: Tuple[int, int, int]
    kernel: int
    lstm_layers: int
    dropout: float
    noise_ratio_arg: bool
    eps: float
    simple_order: bool          # conv -> binarise -> rnn -> rnn -> deconv, ReLU after conv3
    default_hw: Tuple[int, int]


VARIANTS = {
    # models/percep_RBVAE/percep_RBVAE_model.py:46-141
    "percep": Variant("percep", (256, 256, 256), 3, 4, 0.2, True, 1e-8, False, (88, 160)),
    # models/contrastive_RBVAE/contrastive_RBVAE_model.py:45-140
    "contrastive": Variant("contrastive", (64, 64, 64), 3, 2, 0.2, True, 1e-8, False, (256, 256)),
    # models/triplet_RBVAE/triplet_RBVAE_model.py:18-45,144-171 (noise is never scaled)
    "triplet": Variant("triplet", (64, 64, 64), 3, 2, 0.2, False, 1e-8, False, (256, 256)),
    # models/simple_RBVAE/simple_RBVAE_model.py:77-193
    "simple": Variant("simple", (64, 128, 256), 4, 1, 0.0, False, 1e-10, True, (64, 64)),
}


def conv_out(n: int, k: int) -> int:
    return (n + 2 - k) // 2 + 1


def _ru(x: int, m: int) -> int:
    return (x + m - 1) // m * m


class ParamLayout:
    """Names, shapes and flat offsets of the parameters in the reference's
    registration order (state_dict order of Seq2SeqBinaryVAE)."""

    def __init__(self, v: Variant, in_ch: int, out_ch: int, latent: int, hw: Tuple[int, int]):
        c1, c2, c3 = v.channels
        k = v.kernel
        h, w = hw
        for _ in range(3):
            h, w = conv_out(h, k), conv_out(w, k)
        self.bott = (h, w)
        flat = c3 * h * w
        idx = (0, 2, 4) if v.name == "simple" else (0, 3, 6)
        shapes: List[Tuple[str, Tuple[int, ...]]] = []
        for i, (ci, co) in zip(idx, [(in_ch, c1), (c1, c2), (c2, c3)]):
            shapes += [(f"encoder_cnn.conv.{i}.weight", (co, ci, k, k)), (f"encoder_cnn.conv.{i}.bias", (co,))]
        shapes += [("encoder_cnn.fc.weight", (latent, flat)), ("encoder_cnn.fc.bias", (latent,)),
                   ("decoder_cnn.fc.weight", (flat, latent)), ("decoder_cnn.fc.bias", (flat,))]
        for i, (ci, co) in zip(idx, [(c3, c2), (c2, c1), (c1, out_ch)]):
            shapes += [(f"decoder_cnn.deconv.{i}.weight", (ci, co, k, k)), (f"decoder_cnn.deconv.{i}.bias", (co,))]
        for stack in ("encoder_rnn", "decoder_rnn"):
            for l in range(v.lstm_layers):
                shapes += [(f"{stack}.lstm.weight_ih_l{l}", (4 * latent, latent)),
                           (f"{stack}.lstm.weight_hh_l{l}", (4 * latent, latent)),
                           (f"{stack}.lstm.bias_ih_l{l}", (4 * latent,)),
                           (f"{stack}.lstm.bias_hh_l{l}", (4 * latent,))]
        self.names = [n for n, _ in shapes]
        self.shapes = dict(shapes)
        self.offsets: Dict[str, int] = {}
        off = 0
        for n, s in shapes:
            # keep every tensor 16-byte aligned inside the flat buffer
            off = _ru(off, 4)
            self.offsets[n] = off
            off += math.prod(s)
        self.total = _ru(off, 4)
        self.conv_idx = idx
        # the LSTM blocks must be gap-free (the kernel walks w_ih, w_hh, b_ih, b_hh per layer)
        for stack in ("encoder_rnn", "decoder_rnn"):
            base = self.offsets[f"{stack}.lstm.weight_ih_l0"]
            per = 8 * latent * latent + 8 * latent
            for l in range(v.lstm_layers):
                assert self.offsets[f"{stack}.lstm.weight_ih_l{l}"] == base + l * per, "lstm block not contiguous"

    def view(self, flat: torch.Tensor, name: str) -> torch.Tensor:
        o = self.offsets[name]
        s = self.shapes[name]
        return flat[o:o + math.prod(s)].view(s)


def conv_classes(k: int) -> List[int]:
    """gather_gemm descriptor of Conv2d(k, stride 2, pad 1): one class, k*k taps."""
    d = [k * k, 0, 0]
    for kh in range(k):
        for kw in range(k):
            d += [kh * k + kw, kh - 1, kw - 1]
    return d


def dgrad_classes(k: int) -> Tuple[List[int], int]:
    """Descriptor of the conv's input gradient (= ConvTranspose2d(k,2,1) forward): four
    output-parity classes, heaviest first so the long workgroups start early."""
    per = []
    for ch in (0, 1):
        for cw in (0, 1):
            taps = []
            for kh in range(k):
                if (ch + 1 - kh) % 2:
                    continue
                for kw in range(k):
                    if (cw + 1 - kw) % 2:
                        continue
                    taps += [kh * k + kw, (ch + 1 - kh) // 2, (cw + 1 - kw) // 2]
            per.append((len(taps) // 3, ch, cw, taps))
    per.sort(key=lambda t: -t[0])
    d: List[int] = []
    for n, ch, cw, taps in per:
        d += [n, ch, cw] + taps
    return d, len(per)


ONE_TAP = [1, 0, 0, 0, 0, 0]

JOB_PACK, JOB_PERMUTE, JOB_ROWS, JOB_CONV_PACK, JOB_CONV_REDUCE, JOB_GATHER, JOB_ADAM_PACK, JOB_ADAM = 0, 1, 2, 3, 4, 5, 6, 7


class JobList:
    """Rows of the rbvae_run_jobs table (16 x int64 each), uploaded once and replayed."""

    def __init__(self):
        self.rows: List[List[int]] = []
        self.keep = []          # tensors whose addresses the table holds

    def add(self, kind, src, dst, dims, strides, nslab=1, slab=0, dtype=0, accumulate=0, scale=1.0):
        import struct
        bits = struct.unpack("<I", struct.pack("<f", float(scale)))[0]
        # consecutive threads walk the index whose stride on the strided side is 1 (coalesced on that side;
        # the contiguous side then sees short strides that the caches absorb)
        fast = 2
        for ax in (2, 1, 0):
            if strides[ax] == 1 and dims[ax] > 1:
                fast = ax
                break
        # ... unless that index is the middle one of a long row with a short last index (conv weights): then a
        # thread walks the last index itself and both sides move in runs
        inner = int(kind in (JOB_PACK, JOB_PERMUTE) and fast == 1 and dims[2] <= 16 and dims[1] >= 64)
        self.rows.append([kind, src.data_ptr(), dst.data_ptr(), dims[0], dims[1], dims[2], strides[0], strides[1],
                          strides[2], nslab, slab, dtype, int(accumulate), bits | (fast << 32), inner, 0])
        self.keep += [src, dst]

    def add_conv_pack(self, src, wf, wd, co, ci, kk, dtype):
        """f32 weight [co][ci][kk] -> [co][t][ci] (wf) and [ci][t][co] (wd) in one pass."""
        if kk > 16:
            raise ValueError("conv weight pack supports kernels up to 4x4")
        self.rows.append([JOB_CONV_PACK, src.data_ptr(), wf.data_ptr(), co, ci, kk, 0, 0, 0, 1, 0, dtype, 0, 0, 0,
                          wd.data_ptr()])
        self.keep += [src, wf, wd]

    def add_conv_reduce(self, slabs, dst, co, ci, kk, nslab, scale=1.0):
        """K-slice slabs [nslab][co][kk][ci] of a conv weight gradient -> torch layout [co][ci][kk], coalesced."""
        import struct
        bits = struct.unpack("<I", struct.pack("<f", float(scale)))[0]
        self.rows.append([JOB_CONV_REDUCE, slabs.data_ptr(), dst.data_ptr(), co, ci, kk, 0, 0, 0, nslab, co * kk * ci, 0, 0,
                          bits, 0, 0])
        self.keep += [slabs, dst]

    def upload(self, device):
        return torch.tensor(self.rows, dtype=torch.int64).to(device)

    def signature(self):
        return tuple(tuple(r) for r in self.rows)


def job_block_map(rows, device, cap):
    """Block map of a job table for rbvae_run_jobs_sized: one workgroup entry (job, index within the job, workgroups of the
    job) per workgroup the rows can use, at most `cap` per job.  -> (int32 device tensor [n][4], n)."""
    import numpy as np
    arr = np.ascontiguousarray(np.array(rows, dtype=np.int64).reshape(len(rows), 16))
    total = L.query("rbvae_job_block_map", arr.ctypes.data, len(rows), cap, None, 0)
    if total <= 0:
        raise RuntimeError("rbvae_job_block_map: " + L.lib().rbvae_last_error().decode())
    m = np.empty(4 * total, dtype=np.int32)
    if L.query("rbvae_job_block_map", arr.ctypes.data, len(rows), cap, m.ctypes.data, total) != total:
        raise RuntimeError("rbvae_job_block_map: " + L.lib().rbvae_last_error().decode())
    return torch.from_numpy(m).to(device), total


class Saved:
    """Activations one forward call keeps for its backward."""
    __slots__ = ("N", "S", "T", "hw", "train", "tau", "tau_dev", "hard", "col1", "x_in", "fm_in", "a1", "a2", "a3", "e", "hs_enc", "hp_enc",
                 "acts_enc", "cs_enc", "y", "z", "hs_dec", "hp_dec", "acts_dec", "cs_dec", "ds_pad", "f", "d1", "d2",
                 "xr", "gate_scale", "dpre3", "b3_parts")


class Engine:
    def __init__(self, variant: str, in_ch: int, out_ch: int, latent: int, hw: Tuple[int, int],
                 dtype: str = "f32", device: Optional[torch.device] = None):
        if variant not in VARIANTS:
            raise ValueError(f"unknown variant {variant!r}")
        if dtype not in ("f32", "bf16"):
            raise ValueError("dtype must be 'f32' or 'bf16'")
        self.v = VARIANTS[variant]
        self.in_ch, self.out_ch, self.latent = in_ch, out_ch, latent
        self.hw = tuple(hw)
        if self.hw[0] % 8 or self.hw[1] % 8:
            raise ValueError(f"frame size {self.hw} must be divisible by 8 (three stride-2 stages)")
        if latent > 128:
            raise ValueError("latent_dim > 128 is not supported by the LSTM kernel")
        self.dt = F32 if dtype == "f32" else BF16
        self.tdt = torch.float32 if dtype == "f32" else torch.bfloat16
        self.es = 4 if dtype == "f32" else 2
        self.ke = 128 // self.es
        self.layout = ParamLayout(self.v, in_ch, out_ch, latent, self.hw)
        self.device = device or torch.device("cuda", torch.cuda.current_device())
        k = self.v.kernel
        self.k = k
        H, W = self.hw
        self.g1 = (conv_out(H, k), conv_out(W, k))
        self.g2 = (conv_out(self.g1[0], k), conv_out(self.g1[1], k))
        self.g3 = (conv_out(self.g2[0], k), conv_out(self.g2[1], k))
        c1, c2, c3 = self.v.channels
        self.F3 = c3 * self.g3[0] * self.g3[1]
        self.K1 = _ru(k * k * in_ch, self.ke)          # im2col width of the first conv
        self.K3 = _ru(k * k * out_ch, self.ke)         # im2col width of d(loss)/d(pre) for the last deconv
        self.NY = _ru(k * k * out_ch, 8)               # per-tap product width of the last deconv
        self.Lp = _ru(latent, self.ke)
        self.zero = torch.zeros(256, dtype=torch.uint8, device=self.device)
        self._cls_conv = conv_classes(k)
        self._cls_dgrad, self._ncls_dgrad = dgrad_classes(k)
        self._desc_cache: Dict[Tuple, object] = {}
        self._idx_cache: Dict[Tuple, torch.Tensor] = {}
        self.seed_dev = None           # optional device step counter (int64 tensor) for graph-replayed steps
        self._pack_tab: Dict = {}
        self._bufs: Dict = {}          # persistent backward temporaries, keyed by (N, tag)
        self._bwd_tab: Dict = {}       # uploaded reduce-job tables, keyed by their signature
        self._jobs: Optional[JobList] = None
        self._wg_cus = 256
        self.last_wgrad_kernel = None  # kernel family the last _wgrad() call launched (bench.py's roofline leg keys its timers by it)
        self._job_blocks = 256      # workgroups per job of a batched job launch (at most)
        self.sized_jobs = 2         # job tables launched with exactly the workgroups they can use: 0 none, 1 the update table, 2 all
        self._side: Optional[torch.cuda.Stream] = None
        # Side stream (graph capture turns the fork / join into graph edges; `overlap = False`: everything in issue order on
        # one stream -- bench.py's `isolated` leg).  What rides it was settled by same-GPU sweeps in rounds 1-2 (ms/step):
        # pair term + decoder weight gradients + their reductions 0.547; also the loss bookkeeping / the LSTM weight
        # gradients as forks of their own / the final reductions 0.553-0.568; nothing 0.591.  Full-chip side kernels beside
        # full-chip main kernels only contend.  Program order at a fork: the main stream's continuation is issued BEFORE the
        # side work (graph capture hands the forking node's queue to the branch created first; created second, the main
        # chain paid a cross-queue hand-off of ~10 us at every fork).
        self.overlap = True
        self._ks_small = 3
        self.fc_gemm = True               # dedicated kernel for the K = 64 fc products
        # halo-resident kernel for the transposed 3x3 / stride-2 launches (csrc/deconv_halo.hip) from this many workgroups up
        self.deconv_halo = True
        self._dh_min_wgs = 1024
        # nine-taps-per-workgroup weight gradient of the 3x3 stride-2 layers (csrc/wgrad_halo.hip) when every workgroup
        # gets at least this many 4 x 8 pixel blocks
        # halo-resident kernel for the 3x3 stride-2 forward-type launches (csrc/conv_s2.hip) from this many workgroups up, when
        # its 8 x 16-pixel tiles cover at most this many times the image's pixels (measured, tools/time_conv_s2.py: 128 images
        # 88 x 160 -> 44 x 80, 256 channels: 662 vs 844 us for the gather GEMM; the native 44 x 80 -> 22 x 40 layer, whose 22 x 40
        # map the tiles cover 1.31 times: 229 vs 174 us, so it stays on the gather GEMM)
        self.conv_s2_halo = True
        self._cs_min_wgs = 512
        self._cs_max_waste = 1.15
        self.wgrad_halo = True
        self._wh_min_steps = 8
        self._wh_max_tiles = 2
        # three-taps-per-workgroup weight gradient of the wide (multiples of 128 channels) 3x3 stride-2 layers
        # (csrc/wgrad_row.hip): K-slices of at least _wr_min_steps 64-pixel blocks, at most _wr_slab_mb MB of f32 slabs
        self.wgrad_row = True
        self._wr_min_steps = 4
        self._wr_slab_mb = 64
        # K-slices ~ sqrt(coef * blocks): a K-slice more costs its f32 slab (2.4 MB written, read again by the reduction job:
        # ~1.2 us of memory time at 256 x 256 channels), a K-slice less lengthens every workgroup's loop by blocks / ks^2 steps
        # of ~1.1 us.  Bench shape (256 blocks), same box, ms per step: 7 slices 0.4447, 10 0.4392, 14 0.4549, 21 0.4689
        # (rbvae_wgrad_gemm: 0.4451); native 4 x 88 x 160 (1 760 blocks) wants every CU: 21 slices 2.00 (gemm 2.14)
        self._wr_ks_coef = 0.4
        self._wr_min_blocks = 128      # fewer 64-pixel blocks than this (the bench shape's 4 x 4 layers: 64): rbvae_wgrad_gemm
        # weight gradients of the two 3/4-channel ends from the image itself instead of from im2col rows in HBM
        # (rbvae_wgrad_first, csrc/conv_first.hip) when every workgroup gets at least this many 8 x 16 pixel blocks
        self.wgrad_first = True
        self._wf_min_steps = 8
        self.lstm_pair_bwd = True   # both stacks' BPTT in one launch
        self.keep_dz = False                # also store the codes' gradient
        # The fc products on either side of the LSTM stacks (M = frames, 32 outputs, K = thousands) run K-split over
        # several workgroup groups, and the LSTM kernels sum the slabs while staging their input (one group: 32 CUs busy at
        # 256 frames).  Needs the wavefront LSTM kernels (latent <= 32).  16 groups for the large fc layers (at F3 = 56 320 /
        # 65 536 four groups were 64 workgroups streaming 21-42 MB: 31-36 us per launch at 0.6 TB/s), else the largest of
        # 16 / 8 / 4 that divides F3 into whole K units
        ks_unit = 32 if dtype == "bf16" else 16
        wave_ok = latent <= 32 and self.v.lstm_layers * _ru(4 * latent, 64) <= 1024 and not self.v.simple_order
        self.fc_split = 1
        if wave_ok and self.F3 >= 2048:
            for want in ((16, 8, 4) if self.F3 >= 16384 else (4,)):
                if self.F3 % (want * ks_unit) == 0:
                    self.fc_split = want
                    break
        # ... and write the bf16 / padded copy of their output that the next GEMM reads (rbvae_cast_pad otherwise)
        self.lstm_cast = wave_ok
        # ... and, in forward passes that run both stacks, go as one launch with the binarisation between them
        self.lstm_pair = wave_ok
        self.bin_bwd_fused = wave_ok
        self.deconv_fused = True
        self.conv_first_fused = True
        self._alloc_packed()

    # ---- packed weights -------------------------------------------------------
    def _alloc_packed(self):
        c1, c2, c3 = self.v.channels
        kk = self.k * self.k
        z = lambda *s: torch.zeros(*s, dtype=self.tdt, device=self.device)
        self.W1p = z(c1, self.K1)                      # conv1 as a 1-tap GEMM over im2col columns
        self.W2f, self.W2d = z(c2, kk, c1), z(c1, kk, c2)
        self.W3f, self.W3d = z(c3, kk, c2), z(c2, kk, c3)
        self.Wfc = z(self.latent, self.F3)             # [L][NHWC-flat]
        self.WfcT = z(self.F3, self.Lp)                # [NHWC-flat][Lp]
        self.Wdfc = z(self.F3, self.Lp)                # [NHWC-flat][Lp]
        self.WdfcT = z(self.latent, self.F3)
        self.bdfc = torch.zeros(self.F3, dtype=torch.float32, device=self.device)   # NHWC order
        self.V1f, self.V1d = z(c3, kk, c2), z(c2, kk, c3)   # deconv0: c3 -> c2
        self.V2f, self.V2d = z(c2, kk, c1), z(c1, kk, c2)   # deconv1: c2 -> c1
        self.V3p = z(self.NY, c1)                      # [(t,co)][c1]: per-tap products of the last deconv
        self.V3f = z(c1, self.K3)                      # [c1][(t,co) padded]
        nl, Ld = self.v.lstm_layers, self.latent
        self.wT_enc = torch.zeros(nl, 2, Ld, 4 * Ld, dtype=torch.float32, device=self.device)   # [k][gate row]
        self.wT_dec = torch.zeros(nl, 2, Ld, 4 * Ld, dtype=torch.float32, device=self.device)

    def _register_table(self, tab: torch.Tensor, rows):
        """Remember the block map of an uploaded job table (run_table launches it with exactly the workgroups it can use)."""
        if not hasattr(self, "_table_maps"):
            self._table_maps = {}
        self._table_maps[tab.data_ptr()] = job_block_map(rows, self.device, self._job_blocks) + (tab,)

    def run_table(self, tab: torch.Tensor, n: int, kind: int = 2):
        """One batched job launch of an uploaded table: sized (rbvae_run_jobs_sized) when its block map is registered and
        sized_jobs covers its kind (1 = the optimiser update table, 2 = the pack / reduce tables)."""
        m = getattr(self, "_table_maps", {}).get(tab.data_ptr())
        if m is None or self.sized_jobs < kind:
            L.call("rbvae_run_jobs", tab, n, self._job_blocks)
        else:
            L.call("rbvae_run_jobs_sized", tab, m[0], m[1])

    def pack(self, flat: torch.Tensor):
        """f32 parameters (reference layouts) -> the packed T copies the GEMMs read (one launch)."""
        key = flat.data_ptr()
        tab = self._pack_tab.get(key)
        if tab is None:
            jl = self._pack_jobs(flat)
            tab = (jl.upload(self.device), len(jl.rows), jl)
            self._register_table(tab[0], jl.rows)
            self._pack_tab[key] = tab
        self.run_table(tab[0], tab[1])

    def update_jobs(self, flat, gflat, m, v, hyper, betas, eps, gscale, extra_rows=None, part=None):
        """Optimiser step + weight repack of a training step as ONE batched job launch: every parameter tensor is a job
        that applies torch.optim.Adam to its slice of the flat buffers (the arithmetic of rbvae_adam_step) and writes
        the tensor's packed copies from the new values while it holds them -- kind 3 (conv weights: the block updates
        the rows it packs), kind 6 (generic scatter to one or two copies), kind 7 (no packed copy: biases).
        part: None = every parameter; "tail" / "head" = the data-parallel trainer's gradient buckets (decoder CNN + both
        LSTM stacks / encoder CNN, cut at decoder_cnn.fc.weight): the tail is updated while the head is still being
        all-reduced.  Returns (device table, n_jobs); cached per buffer set."""
        import struct
        key = ("update", flat.data_ptr(), gflat.data_ptr(), m.data_ptr(), v.data_ptr(), hyper.data_ptr(), tuple(betas),
               float(eps), float(gscale), tuple(map(tuple, extra_rows)) if extra_rows else None, part)
        tab = self._pack_tab.get(key)
        if tab is not None:
            return tab[0], tab[1]
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("update job table missing during graph capture: run one eager update with the same buffers, "
                               f"hyper-parameters and part ({part!r}) first (FusedTrainer's warm-up steps do)")
        f2 = lambda a, b: struct.unpack("<q", struct.pack("<ff", float(a), float(b)))[0]
        b1, b2 = betas
        ctx = torch.tensor([flat.data_ptr(), gflat.data_ptr(), m.data_ptr(), v.data_ptr(), hyper.data_ptr(),
                            f2(1.0 - b1, b2), f2(1.0 - b2, eps), f2(gscale, 0.0)], dtype=torch.int64).to(self.device)
        jl = self._pack_jobs(flat)
        by_src: Dict[int, List[List[int]]] = {}
        for row in jl.rows:
            by_src.setdefault(row[1], []).append(row)
        lay = self.layout
        rows: List[List[int]] = [list(r) for r in extra_rows] if extra_rows else []
        cp = ctx.data_ptr()
        o_cut = lay.offsets["decoder_cnn.fc.weight"]
        for name in lay.names:
            src = lay.view(flat, name)
            n = src.numel()
            packs = by_src.pop(src.data_ptr(), [])
            if part is not None and (lay.offsets[name] >= o_cut) != (part == "tail"):
                continue
            if len(packs) == 1 and packs[0][0] == JOB_CONV_PACK:
                r = list(packs[0])
                r[14] = cp
                rows.append(r)
            elif packs and all(p_[0] == JOB_PACK for p_ in packs) and len(packs) <= 2:
                a = packs[0]
                bq = packs[1] if len(packs) == 2 else None
                if bq is not None and tuple(bq[3:6]) != tuple(a[3:6]):
                    raise RuntimeError(f"{name}: its two packed copies walk different logical shapes")
                t2 = bq[11] if bq is not None else 0
                rows.append([JOB_ADAM_PACK, a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8],
                             bq[6] if bq is not None else 0, bq[7] if bq is not None else 0, a[11] | (t2 << 8),
                             bq[8] if bq is not None else 0, 0, cp, bq[2] if bq is not None else 0])
            elif not packs:
                rows.append([JOB_ADAM, src.data_ptr(), 0, n, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0, cp, 0])
            else:
                raise RuntimeError(f"{name}: unsupported pack job combination for the fused update")
        if by_src:
            raise RuntimeError("pack jobs whose source is not a parameter tensor")
        t = torch.tensor(rows, dtype=torch.int64).to(self.device)
        self._register_table(t, rows)
        self._pack_tab[key] = (t, len(rows), ctx, jl)
        return t, len(rows)

    def _pack_jobs(self, flat: torch.Tensor) -> JobList:
        lay, dt = self.layout, self.dt
        c1, c2, c3 = self.v.channels
        kk = self.k * self.k
        g3 = self.g3[0] * self.g3[1]
        i0, i1, i2 = lay.conv_idx
        P = lambda name: lay.view(flat, name)
        jl = JobList()
        pk = lambda src, dst, dims, strides, d=None: jl.add(JOB_PACK, src, dst, dims, strides, dtype=dt if d is None else d)
        # conv1 [c1][cin][kk] -> [c1][t*cin + ci]
        pk(P(f"encoder_cnn.conv.{i0}.weight"), self.W1p, (c1, self.in_ch, kk), (self.K1, 1, self.in_ch))
        for name, wf, wd, co, ci in ((f"encoder_cnn.conv.{i1}.weight", self.W2f, self.W2d, c2, c1),
                                     (f"encoder_cnn.conv.{i2}.weight", self.W3f, self.W3d, c3, c2),
                                     (f"decoder_cnn.deconv.{i0}.weight", self.V1f, self.V1d, c3, c2),
                                     (f"decoder_cnn.deconv.{i1}.weight", self.V2f, self.V2d, c2, c1)):
            w = P(name)                                               # [co][ci][kk]
            jl.add_conv_pack(w, wf, wd, co, ci, kk, dt)               # [co][t][ci] and [ci][t][co]
        v3 = P(f"decoder_cnn.deconv.{i2}.weight")                     # [c1][out][kk]
        pk(v3, self.V3p, (c1, self.out_ch, kk), (1, c1, self.out_ch * c1))       # [(t*out+co)][c1]
        pk(v3, self.V3f, (c1, self.out_ch, kk), (self.K3, 1, self.out_ch))       # [c1][t*out+co]
        wfc = P("encoder_cnn.fc.weight")                              # [L][c3][g3]
        pk(wfc, self.Wfc, (self.latent, c3, g3), (self.F3, 1, c3))
        pk(wfc, self.WfcT, (self.latent, c3, g3), (1, self.Lp, c3 * self.Lp))
        wd = P("decoder_cnn.fc.weight")                               # [c3][g3][L]
        pk(wd, self.Wdfc, (c3, g3, self.latent), (self.Lp, c3 * self.Lp, 1))
        pk(wd, self.WdfcT, (c3, g3, self.latent), (1, c3, self.F3))
        pk(P("decoder_cnn.fc.bias"), self.bdfc, (c3, g3, 1), (1, c3, 0), F32)
        Ld = self.latent
        for stack, wT in (("encoder_rnn", self.wT_enc), ("decoder_rnn", self.wT_dec)):
            for l in range(self.v.lstm_layers):
                for m, nm in enumerate(("weight_ih", "weight_hh")):
                    pk(P(f"{stack}.lstm.{nm}_l{l}"), wT[l, m], (4 * Ld, Ld, 1), (1, 4 * Ld, 0), F32)
        return jl

    # ---- helpers ---------------------------------------------------------------
    def _desc(self, key, ints):
        d = self._desc_cache.get(key)
        if d is None:
            import ctypes
            d = (ctypes.c_int * len(ints))(*ints)
            self._desc_cache[key] = d
        return d

    def _buf(self, tag, numel, dtype=torch.float32):
        """Persistent temporary (same address every step, so uploaded job tables stay valid)."""
        key = (tag, numel, dtype)
        t = self._bufs.get(key)
        if t is None:
            t = torch.empty(numel, dtype=dtype, device=self.device)
            self._bufs[key] = t
        return t

    def _gemm(self, A, W, out, bias, gate, mask, nimg, ih, iw, th, tw, sa, oh, ow, so, kc, nout, lda, ldo, taps,
              cls_key, relu=0, drop_mode=0, drop_p=0.0, scale=1.0, seed=0, bias_grad=None, colsum_ws=None,
              tag=None):
        """bias_grad: f32 [nout] tensor that receives the column sums of the stored output (a reduce job
        over the kernel's per-tile partial sums, run with the other jobs at the end of backward)."""
        seed_dev = self.seed_dev
        if (self.fc_gemm and cls_key == "one" and gate is None and mask is None and drop_mode == 0 and not relu
                and scale == 1.0 and bias_grad is None and nimg <= 4096 and nout >= 1024
                and ih * iw * th * tw * oh * ow == 1 and L.query("rbvae_fc_gemm_ok", self.dt, nimg, kc, nout, lda, ldo)):
            # the two fc products at the latent bottleneck: 0.13 GFLOP each, latency-bound (csrc/fc_gemm.hip)
            L.call("rbvae_fc_gemm", self.dt, A, W, out, bias, colsum_ws, nimg, kc, nout, lda, ldo)
            return
        if (cls_key == "dgrad" and self.k == 3 and self.deconv_halo and sa == 1 and so == 2 and taps == 9
                and (oh, ow) == (2 * th, 2 * tw) and (ih, iw) == (th, tw) and colsum_ws is None
                and L.query("rbvae_deconv3x3s2_halo_tile_rows", self.dt, nimg, th, tw, kc, nout) == 128):
            # transposed 3x3 / stride 2 (decoder forward, encoder input gradients): the four parity classes in one
            # workgroup over an LDS-resident input patch (csrc/deconv_halo.hip).  Launches of several rounds of
            # workgroups only: a one-round launch (the 256-frame step's 8x8 grids: 512 workgroups) runs as long as the
            # four class launches of the gather GEMM (35 vs 34 us), measured 7-25 % faster from 1 000 workgroups up; the
            # 128-row form only (two workgroups per CU): the 256-row form measured slower than the class launches
            rows4 = L.query("rbvae_deconv3x3s2_halo_colsum_rows", self.dt, nimg, th, tw, kc, nout)
            wgs = (rows4 // 4) * (nout // 64)
            if wgs >= self._dh_min_wgs:
                ws = None
                if bias_grad is not None:
                    ws = self._buf(("colsum_dh", tag), rows4 * nout)
                    self._jobs.add(JOB_ROWS, ws, bias_grad, (1, 1, nout), (0, 0, 1), nslab=rows4, slab=nout)
                L.call("rbvae_deconv3x3s2_halo", self.dt, A, W, out, bias, gate, mask, self.zero, nimg, th, tw, kc, nout, lda,
                       ldo, relu, drop_mode, float(drop_p), float(scale), int(seed), seed_dev, ws)
                return
        if (cls_key == "conv" and self.k == 3 and self.conv_s2_halo and self.dt == BF16 and sa == 2 and so == 1 and taps == 9
                and (oh, ow) == (th, tw) and (ih, iw) == (2 * th, 2 * tw) and colsum_ws is None):
            # 3x3 stride-2 convolution (encoder forward, decoder input gradients) with the input patch resident in LDS
            # (csrc/conv_s2.hip): 8 x 16-pixel tiles, so small images (the bench shape's 8 x 8 / 4 x 4 maps) stay on the
            # row-gather GEMM, as do launches of less than a few rounds of workgroups
            bn = L.query("rbvae_conv3x3s2_halo_ok", self.dt, nimg, ih, iw, kc, nout)
            if bn:
                mt = L.query("rbvae_conv3x3s2_halo_colsum_rows", nimg, ih, iw)
                if mt * (nout // bn) >= self._cs_min_wgs and mt * 128 <= self._cs_max_waste * nimg * oh * ow:
                    ws = None
                    if bias_grad is not None:
                        ws = self._buf(("colsum_cs", tag), mt * nout)
                        self._jobs.add(JOB_ROWS, ws, bias_grad, (1, 1, nout), (0, 0, 1), nslab=mt, slab=nout)
                    L.call("rbvae_conv3x3s2_halo", self.dt, A, W, out, bias, gate, mask, nimg, ih, iw, kc, nout, lda, ldo, relu,
                           drop_mode, float(drop_p), float(scale), int(seed), seed_dev, ws)
                    return
        if cls_key == "one":
            desc, ncls = self._desc("one", ONE_TAP), 1
        elif cls_key == "conv":
            desc, ncls = self._desc("conv", self._cls_conv), 1
        else:
            desc, ncls = self._desc("dgrad", self._cls_dgrad), self._ncls_dgrad
        import ctypes
        ws = colsum_ws
        if bias_grad is not None:
            prow = ncls * (-(-(nimg * th * tw) // 128))
            ws = self._buf(("colsum", tag), prow * nout)
            self._jobs.add(JOB_ROWS, ws, bias_grad, (1, 1, nout), (0, 0, 1), nslab=prow, slab=nout)
        L.call("rbvae_gather_gemm", self.dt, A, W, out, bias, gate, mask, None, self.zero, nimg, ih, iw, th, tw, sa, oh, ow,
               so, kc, nout, lda, ldo, taps, ncls, ctypes.addressof(desc), relu, drop_mode, float(drop_p),
               float(scale), int(seed), seed_dev, ws)

    def _conv_idx(self, nimg, ih, iw, oh, ow):
        """Gather-index table of a stride-2 convolution (cached per shape, read by weight-gradient GEMMs on ANY
        stream).  A table is complete in device memory before it enters the cache: its kernel is followed by a
        host-side stream synchronisation (first use of a shape only), so no consumer on another stream can ever
        read a table whose kernel it has not waited for.  During graph capture nothing may be built: the trainer's
        eager warm-up steps (or prepare()) have filled the cache by then."""
        key = (nimg, ih, iw, oh, ow)
        t = self._idx_cache.get(key)
        if t is None:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("gather-index table missing during graph capture: run Engine.prepare(N) or one "
                                   "eager step of this shape first")
            t = torch.empty(self.k * self.k * nimg * oh * ow, dtype=torch.int32, device=self.device)
            L.call("rbvae_conv_gather_index", t, nimg, ih, iw, oh, ow, self.k, self.k, 2, 1)
            torch.cuda.current_stream().synchronize()
            self._idx_cache[key] = t
        return t

    def prepare(self, N: int):
        """Build every device table a pass over N frames reads from more than one stream (the gather-index tables
        of the weight-gradient GEMMs), on the current stream, before any side stream is forked."""
        (h1, w1), (h2, w2), (h3, w3) = self.g1, self.g2, self.g3
        self._conv_idx(N, h1, w1, h2, w2)
        self._conv_idx(N, h2, w2, h3, w3)

    def _wf_ksplit(self, N, Cin, IH, IW, Nout):
        """K-slices of rbvae_wgrad_first for this layer, or 0 when the im2col-row path stays (f32, a shape outside the
        kernel, or too few pixel blocks per workgroup: the bench shape keeps its rows -- 8 MB, 4 blocks per workgroup)."""
        if not (self.wgrad_first and self.dt == BF16 and self.k == 3):
            return 0
        nblk = L.query("rbvae_wgrad_first_blocks", self.dt, Cin, IH, IW, Nout, N)
        if nblk == 0:
            return 0
        if Nout % 256 == 0:
            ks = max(1, min(256 // (Nout // 256), nblk))       # all 256 channels of a block in one workgroup, one workgroup per CU
        else:
            ks = max(1, min(512 // (Nout // 64), nblk))        # two workgroups per CU
        return ks if nblk // ks >= self._wf_min_steps else 0

    def _wgrad_first(self, mode, x, fm, Dy, N, Cin, IH, IW, Nout, ldy, ks, out, dims, strides, tag=None):
        slabs = self._buf(("slabs", tag), ks * Nout * 64)
        L.call("rbvae_wgrad_first", self.dt, mode, x, *fm, Dy, slabs, self.zero, N, Cin, IH, IW, Nout, ldy, ks)
        self._wgrad_reduce(slabs, out, Nout, 64, 1, ks, dims, strides)

    def _wgrad(self, Dy, In, idx, P, Co, Ci, ldy, ldi, taps, out, dims, strides, tag=None, geom=None):
        """wgrad GEMM into K-slice slabs; their fixed-order reduction into the torch layout is a job.
        geom = (Nimg, OH, OW, IH, IW) of a 3x3 stride-2 layer (Dy rows at OH x OW, In rows at IH x IW): with the nine taps
        in one workgroup (csrc/wgrad_halo.hip) when the launch gives every workgroup enough pixel blocks."""
        if (geom is not None and self.wgrad_halo and self.dt == BF16 and self.k == 3 and taps == 9 and idx is not None
                and geom[3] == 2 * geom[1] and geom[4] == 2 * geom[2] and Co % 64 == 0 and Ci % 64 == 0
                and L.query("rbvae_wgrad3x3s2_halo_ok", self.dt, geom[0], geom[1], geom[2], Co, Ci)):
            nimg, oh, ow = geom[:3]
            nblk = L.query("rbvae_wgrad3x3s2_halo_blocks", nimg, oh, ow)
            ntile = (Co // 64) * (Ci // 64)
            # K-slices: one round of workgroups on the CUs this launch can count on; at most ~40 MB of f32 slabs
            ks = max(1, min(self._wg_cus // ntile, nblk, (40 << 20) // (Co * taps * Ci * 4)))
            # measured (tools/time_wgrad_halo.py): 64 x 64 channels 178 -> 71 us (cfg 3 conv2: the HBM time of its operands)
            # and 52 -> 27 us (conv3).  With 16 channel tiles per pixel block (256 x 256) a block's 32 KB stage is re-fetched
            # by every tile: the LDS-DMA alone takes 173 us (11 TB/s through the L2s), the MFMAs + fragment reads alone 137,
            # together 245 against rbvae_wgrad_gemm's 172 -- narrow layers only
            if nblk // ks >= self._wh_min_steps and ntile <= self._wh_max_tiles:
                slabs = self._buf(("slabs", tag), ks * Co * taps * Ci)
                self.last_wgrad_kernel = "wgrad_halo_k"
                L.call("rbvae_wgrad3x3s2_halo", self.dt, Dy, In, slabs, self.zero, nimg, oh, ow, Co, Ci, ldy, ldi, ks)
                self._wgrad_reduce(slabs, out, Co, Ci, taps, ks, dims, strides)
                return
        if (geom is not None and self.wgrad_row and self.dt == BF16 and self.k == 3 and taps == 9 and idx is not None
                and geom[3] == 2 * geom[1] and geom[4] == 2 * geom[2]
                and L.query("rbvae_wgrad3x3s2_row_ok", self.dt, geom[0], geom[1], geom[2], Co, Ci)
                and L.query("rbvae_wgrad3x3s2_row_blocks", geom[0], geom[1], geom[2]) >= self._wr_min_blocks):
            # wide layers: the three taps of one kernel row per workgroup (csrc/wgrad_row.hip) -- 50 KB of operands per
            # 6.3 MFLOP instead of rbvae_wgrad_gemm's 96, LDS-DMA issued by waves of their own
            nimg, oh, ow = geom[:3]
            nblk = L.query("rbvae_wgrad3x3s2_row_blocks", nimg, oh, ow)
            ntile = (Co // 128) * (Ci // 128) * 3
            # K-slices: one round of workgroups on the CUs this launch can count on, every workgroup >= _wr_min_steps
            # blocks, at most ~_wr_slab_mb of f32 slabs
            ks = max(1, min(self._wg_cus // ntile, nblk // self._wr_min_steps, (self._wr_slab_mb << 20) // (Co * taps * Ci * 4),
                            int(round(math.sqrt(self._wr_ks_coef * nblk)))))
            slabs = self._buf(("slabs", tag), ks * Co * taps * Ci)
            self.last_wgrad_kernel = "wgrad_row_k"
            L.call("rbvae_wgrad3x3s2_row", self.dt, Dy, In, slabs, self.zero, nimg, oh, ow, Co, Ci, ldy, ldi, ks)
            self._wgrad_reduce(slabs, out, Co, Ci, taps, ks, dims, strides)
            return
        blocks = -(-Co // 128) * -(-Ci // (128 if Ci > 64 else 64)) * taps
        # K-slices: one round of workgroups on the 256 CUs, each with >= 256 pixels, and at most ~16 MB of f32
        # slabs to reduce afterwards
        # (self._wg_cus: the CUs this launch can count on -- beside the LSTM backward kernels, which hold one CU per
        # sequence and whose registers leave no room for a second workgroup there, a 252-workgroup grid ran in two
        # rounds: 31 -> 43 us)
        ks = max(1, min(self._wg_cus // max(blocks, 1), P // 256 if P >= 256 else 1,
                        max(1, (4 << 20) // (Co * taps * Ci))))
        if self._ks_small and blocks >= 8 and P <= 4096:
            # the 4096-pixel layers (conv3 / first deconv at the bench shape): 3 K-slices of 22 steps instead of 7 of 9
            # -- 7 MB of slabs per weight instead of 16.5 MB; same GPU, 2 runs each: 0.4738 (7) / 0.4665 (4) /
            # 0.4659 (3) / 0.472 (2) ms per step
            ks = min(ks, self._ks_small)
        ks = max(ks, -(-P // 4096))   # the kernel keeps a K-slice's gather indices in LDS
        slabs = self._buf(("slabs", tag), ks * Co * taps * Ci)
        self.last_wgrad_kernel = "wgrad_gemm_k<%d>" % (2 if Ci > 64 else 1)
        L.call("rbvae_wgrad_gemm", self.dt, Dy, In, slabs, idx, self.zero, P, In.numel() // ldi, Co, Ci, ldy, ldi, taps, ks)
        self._wgrad_reduce(slabs, out, Co, Ci, taps, ks, dims, strides)

    def _wgrad_reduce(self, slabs, out, Co, Ci, taps, ks, dims, strides):
        if (taps > 1 and taps <= 16 and tuple(dims) == (Co, Ci, taps) and tuple(strides) == (taps * Ci, 1, Ci)
                and Ci % 4 == 0):
            # conv / conv-transpose weight: the coalesced row kernel (16-byte loads of the slabs' [t][ci] rows, LDS
            # transpose, 16-byte stores of the torch-layout row)
            self._jobs.add_conv_reduce(slabs, out, Co, Ci, taps, ks)
        else:
            self._jobs.add(JOB_PERMUTE, slabs, out, dims, strides, nslab=ks, slab=Co * taps * Ci)

    def _colsum(self, dt, X, P, C, ld, out, tag=None):
        """Column sums of a tensor no GEMM epilogue produced: partial kernel now, final reduction as a job."""
        nf = L.query("rbvae_colsum_ws_floats", P, C)
        ws = self._buf(("cs", tag), nf)
        L.call("rbvae_colsum_partial", dt, X, P, C, ld, ws)
        self._jobs.add(JOB_ROWS, ws, out, (1, 1, C), (0, 0, 1), nslab=nf // C, slab=C)

    # ---- side stream ------------------------------------------------------------------
    # The LSTM chains occupy 2B workgroups for ~25 us per stack; weight-gradient GEMMs that do not feed them are
    # issued on a side stream over exactly those windows (graph capture turns the fork/join into graph edges).
    def _fork(self):
        """The side stream picks up after everything queued so far on the current stream."""
        if not self.overlap:
            return False
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
        self._side.wait_stream(torch.cuda.current_stream())
        return True

    def _on_side(self):
        import contextlib
        return torch.cuda.stream(self._side) if self.overlap else contextlib.nullcontext()

    def _join(self):
        if self.overlap and self._side is not None:
            torch.cuda.current_stream().wait_stream(self._side)

    def _run_jobs(self):
        jl, self._jobs = self._jobs, None
        sig = jl.signature()
        tab = self._bwd_tab.get(sig)
        if tab is None:
            tab = (jl.upload(self.device), len(jl.rows), jl)
            self._register_table(tab[0], jl.rows)
            self._bwd_tab[sig] = tab
        self.run_table(tab[0], tab[1])

    def _E(self, *shape, dtype=None):
        return torch.empty(*shape, dtype=dtype or self.tdt, device=self.device)

    # ---- forward ---------------------------------------------------------------
    def forward(self, flat: torch.Tensor, x: torch.Tensor, U: torch.Tensor, tau: float, hard: bool,
                noise_ratio: float, train: bool, masks: Optional[Sequence[torch.Tensor]] = None,
                seed: int = 0, need_grad: bool = True, encode_only: bool = False,
                target: Optional[torch.Tensor] = None, recon_gscale: float = 0.0, kl_p: Optional[float] = None,
                after_hs=None, defer_losses: bool = False,
                frame_map: Optional[Tuple[int, int, int, int, int]] = None, tau_dev: Optional[torch.Tensor] = None):
        """x: [S,T,C,H,W] f32 NCHW frames; U: [S*T, L] uniform noise.
        tau_dev: optional device float the kernels read the temperature from instead of `tau` (graph-replayed
        steps under an annealing schedule; backward() reads the same tensor).
        masks: explicit dropout keep-masks (u8, NHWC rows) for the 4 dropout sites, else a counter hash.
        target/recon_gscale: fuse recon_loss and its gradient into the last kernel (trainer path).
        defer_losses: leave recon_loss and the KL mean as per-block partial sums ("sse": (ws, nparts, 1/n),
        "kl": (parts, nparts, 1/rows)) for rbvae_combine_losses to finish.
        frame_map: (d1, d2, s0, s1, s2) -- frame n of x (and of target) starts at element
        (n // d1) * s0 + ((n % d1) // d2) * s1 + (n % d2) * s2 of the buffer x points at (rbvae_im2col_frames).
        after_hs: optional callable(h_seq) issued on the side stream as soon as the encoder LSTM is done (the
        trainer's pairwise term runs there, beside the decoder); backward() joins it.
        Returns dict(xr, hs, z, e, kl, mse, saved)."""
        v = self.v
        S, T, C, H, W = x.shape
        if (H, W) != self.hw or C != self.in_ch:
            raise RuntimeError(f"input frames {tuple(x.shape[2:])} do not match the model's "
                               f"({self.in_ch}, {self.hw[0]}, {self.hw[1]})")
        N = S * T
        k, kk = self.k, self.k * self.k
        c1, c2, c3 = v.channels
        (h1, w1), (h2, w2), (h3, w3) = self.g1, self.g2, self.g3
        Ld, lay = self.latent, self.layout
        i0, i1, i2 = lay.conv_idx
        P = lambda name: lay.view(flat, name)
        x = x.contiguous()
        drop = v.dropout if train else 0.0
        dscale = 1.0 / (1.0 - drop) if drop > 0 else 1.0

        def dm(j):      # (drop_mode, mask) of dropout site j
            if drop == 0:
                return 0, None
            return (2, masks[j]) if masks is not None else (1, None)

        sv = Saved()
        sv.N, sv.S, sv.T, sv.hw, sv.train, sv.tau, sv.hard = N, S, T, (H, W), train, tau, hard
        sv.tau_dev = tau_dev
        sv.gate_scale = dscale
        # encoder CNN
        sv.a1 = self._E(N * h1 * w1, c1)
        m, mk = dm(0)
        fm = frame_map if frame_map is not None else (0, 0, 0, 0, C * H * W)
        fused1 = bool(self.conv_first_fused and k == 3 and self.K1 == 64 and m != 2
                      and L.query("rbvae_conv_first_fused_ok", self.dt, C, H, W, c1, N))
        # the im2col rows are written for the weight gradient unless it rebuilds them from the frames (rbvae_wgrad_first)
        keep_col = not (fused1 and train and self._wf_ksplit(N, C, H, W, c1))
        sv.col1 = self._E(N * h1 * w1, self.K1) if keep_col else None
        sv.x_in, sv.fm_in = x, fm
        if fused1:
            # im2col + GEMM + bias/ReLU/dropout of the first conv in one kernel (csrc/conv_first.hip)
            L.call("rbvae_conv_first_fused", self.dt, x, *fm, self.W1p, P(f"encoder_cnn.conv.{i0}.bias"), self.zero,
                   sv.col1, sv.a1, N, C, H, W, c1, c1, 1, m, float(drop), float(dscale), int(seed * 8 + 1), self.seed_dev)
        else:
            L.call("rbvae_im2col_frames", self.dt, x, *fm, H * W, W, 1, N, C, H, W, h1, w1, k, k, 2, 1, self.K1, sv.col1)
            self._gemm(sv.col1, self.W1p, sv.a1, P(f"encoder_cnn.conv.{i0}.bias"), None, mk, N * h1 * w1, 1, 1, 1, 1, 1,
                       1, 1, 1, self.K1, c1, self.K1, c1, 1, "one", relu=1, drop_mode=m, drop_p=drop, scale=dscale,
                       seed=seed * 8 + 1)
        sv.a2 = self._E(N * h2 * w2, c2)
        m, mk = dm(1)
        self._gemm(sv.a1, self.W2f, sv.a2, P(f"encoder_cnn.conv.{i1}.bias"), None, mk, N, h1, w1, h2, w2, 2, h2, w2,
                   1, c1, c2, c1, c2, kk, "conv", relu=1, drop_mode=m, drop_p=drop, scale=dscale, seed=seed * 8 + 2)
        sv.a3 = self._E(N * h3 * w3, c3)
        self._gemm(sv.a2, self.W3f, sv.a3, P(f"encoder_cnn.conv.{i2}.bias"), None, None, N, h2, w2, h3, w3, 2, h3,
                   w3, 1, c2, c3, c2, c3, kk, "conv", relu=1 if v.simple_order else 0)
        # fc -> logits e [N][L]
        nl = v.lstm_layers
        sv.hs_enc = self._E(nl + 1, S, T, Ld, dtype=torch.float32)
        sv.hs_dec = self._E(nl + 1, S, T, Ld, dtype=torch.float32)
        sv.e = self._E(N, Ld, dtype=torch.float32) if v.simple_order else sv.hs_enc[0].view(N, Ld)
        e_parts = None
        if self.fc_split > 1:
            e_parts = self._buf((N, "e_parts"), self.fc_split * N * Ld)
            L.call("rbvae_skinny_linear_parts", self.dt, sv.a3, self.Wfc, P("encoder_cnn.fc.bias"), e_parts, N, Ld,
                   self.F3, self.F3, self.F3, Ld, self.fc_split)
        else:
            L.call("rbvae_skinny_linear", self.dt, sv.a3, self.Wfc, P("encoder_cnn.fc.bias"), sv.e, N, Ld, self.F3,
                   self.F3, self.F3, Ld)
        keep = need_grad
        if keep:
            sv.hp_enc = self._E(nl, S, T, Ld, dtype=torch.float32)
            sv.cs_enc = self._E(nl, S, T, Ld, dtype=torch.float32)
            sv.acts_enc = self._E(nl, S, T, 4 * Ld, dtype=torch.float32)
            sv.hp_dec, sv.cs_dec, sv.acts_dec = (torch.empty_like(sv.hp_enc), torch.empty_like(sv.cs_enc),
                                                 torch.empty_like(sv.acts_enc))
        else:
            sv.hp_enc = sv.cs_enc = sv.acts_enc = sv.hp_dec = sv.cs_dec = sv.acts_dec = None
        sv.y = self._E(N, Ld, dtype=torch.float32)
        r = noise_ratio if v.noise_ratio_arg else 1.0
        kl = self._E(1, dtype=torch.float32) if kl_p is not None else None
        wenc, wdec = P("encoder_rnn.lstm.weight_ih_l0"), P("decoder_rnn.lstm.weight_ih_l0")
        pending_hs = None
        fused_pair = (not v.simple_order and not encode_only and self.lstm_pair
                      and bool(L.query("rbvae_lstm_pair_fwd_ok", T, Ld, nl)))
        if fused_pair:
            # encoder stack -> binarise (+ KL sums) -> decoder stack as one wavefront launch
            hs = sv.hs_enc[nl]
            sv.z = sv.hs_dec[0].view(N, Ld)
            parts = None
            if kl_p is not None:
                parts = self._E(S, dtype=torch.float32)
                kl = (parts, S, 1.0 / N) if defer_losses else kl
            sv.ds_pad = self._E(N, self.Lp)
            L.call("rbvae_lstm_pair_fwd", wenc, self.wT_enc, wdec, self.wT_dec, sv.hs_enc, sv.hp_enc, sv.acts_enc,
                   sv.cs_enc, sv.hs_dec, sv.hp_dec, sv.acts_dec, sv.cs_dec, e_parts, self.fc_split, N * Ld, U, sv.y,
                   parts, float(tau), tau_dev, float(r), v.eps, int(hard), float(kl_p if kl_p is not None else 0.5), 1e-8, 1,
                   int(seed) * 8 + 5, self.seed_dev, sv.ds_pad, self.dt, self.Lp, S, T, Ld, nl)
            if kl_p is not None and not defer_losses:
                kl = (parts.sum() * (1.0 / N)).reshape(1)
            if after_hs is not None:
                self._fork()
                pending_hs = hs                 # issued behind the main stream's continuation (end of forward())
        elif not v.simple_order:
            if e_parts is not None:
                L.call("rbvae_lstm_fwd_ex", wenc, self.wT_enc, sv.hs_enc, sv.hp_enc, sv.acts_enc, sv.cs_enc, S, T, Ld,
                       nl, e_parts, self.fc_split, N * Ld, None, 0, 0)
            else:
                L.call("rbvae_lstm_fwd", wenc, self.wT_enc, sv.hs_enc, sv.hp_enc, sv.acts_enc, sv.cs_enc, S, T, Ld, nl)
            hs = sv.hs_enc[nl]
            if after_hs is not None:
                self._fork()
                pending_hs = hs
            sv.z = sv.hs_dec[0].view(N, Ld)
            if defer_losses and kl_p is not None:
                nkl = L.query("rbvae_binarize_kl_nparts", N, Ld)
                parts = self._E(nkl, dtype=torch.float32)
                L.call("rbvae_binarize_kl_fwd_parts", hs, U, sv.y, sv.z, parts, N, Ld, float(tau), tau_dev, float(r), v.eps,
                       int(hard), float(kl_p), 1e-8, 1, int(seed) * 8 + 5, self.seed_dev)
                kl = (parts, nkl, 1.0 / N)
            else:
                L.call("rbvae_binarize_kl_fwd", hs, U, sv.y, sv.z, kl, N, Ld, float(tau), float(r), v.eps, int(hard),
                       float(kl_p if kl_p is not None else 0.5), 1e-8, 1, int(seed) * 8 + 5, self.seed_dev)
            if encode_only:
                return {"z": sv.z.view(S, T, Ld), "hs": hs, "saved": sv}
            if self.lstm_cast:
                sv.ds_pad = self._E(N, self.Lp)
                L.call("rbvae_lstm_fwd_ex", wdec, self.wT_dec, sv.hs_dec, sv.hp_dec, sv.acts_dec, sv.cs_dec, S, T, Ld, nl,
                       None, 1, 0, sv.ds_pad, self.dt, self.Lp)
            else:
                L.call("rbvae_lstm_fwd", wdec, self.wT_dec, sv.hs_dec, sv.hp_dec, sv.acts_dec, sv.cs_dec, S, T, Ld, nl)
        else:
            sv.z = sv.hs_enc[0].view(N, Ld)
            L.call("rbvae_binarize_kl_fwd", sv.e, U, sv.y, sv.z, None, N, Ld, float(tau), float(r), v.eps, int(hard),
                   0.5, 1e-10, 0, int(seed) * 8 + 5, self.seed_dev)
            L.call("rbvae_lstm_fwd", wenc, self.wT_enc, sv.hs_enc, sv.hp_enc, sv.acts_enc, sv.cs_enc, S, T, Ld, nl)
            hs = sv.hs_enc[nl]
            sv.hs_dec[0].copy_(hs)
            L.call("rbvae_lstm_fwd", wdec, self.wT_dec, sv.hs_dec, sv.hp_dec, sv.acts_dec, sv.cs_dec, S, T, Ld, nl)
        ds = sv.hs_dec[nl]
        # decoder CNN
        if not ((self.lstm_cast or fused_pair) and not v.simple_order):
            sv.ds_pad = self._E(N, self.Lp)
            L.call("rbvae_cast_pad", self.dt, ds, sv.ds_pad, N, Ld, self.Lp)
        sv.f = self._E(N * h3 * w3, c3)
        self._gemm(sv.ds_pad, self.Wdfc, sv.f, self.bdfc, None, None, N, 1, 1, 1, 1, 1, 1, 1, 1, self.Lp, self.F3,
                   self.Lp, self.F3, 1, "one")
        sv.d1 = self._E(N * h2 * w2, c2)
        m, mk = dm(2)
        self._gemm(sv.f, self.V1d, sv.d1, P(f"decoder_cnn.deconv.{i0}.bias"), None, mk, N, h3, w3, h3, w3, 1, h2, w2,
                   2, c3, c2, c3, c2, kk, "dgrad", relu=1, drop_mode=m, drop_p=drop, scale=dscale, seed=seed * 8 + 3)
        sv.d2 = self._E(N * h1 * w1, c1)
        m, mk = dm(3)
        self._gemm(sv.d1, self.V2d, sv.d2, P(f"decoder_cnn.deconv.{i1}.bias"), None, mk, N, h2, w2, h2, w2, 1, h1, w1,
                   2, c2, c1, c2, c1, kk, "dgrad", relu=1, drop_mode=m, drop_p=drop, scale=dscale, seed=seed * 8 + 4)
        sv.xr = self._E(S, T, self.out_ch, H, W, dtype=torch.float32)
        mse = None
        sse = None
        sv.dpre3 = None
        sv.b3_parts = None
        fparts = L.query("rbvae_deconv_last_fused_parts", self.dt, N, h1, w1, c1, self.out_ch) if (
            self.deconv_fused and k == 3 and (target is None or defer_losses)) else 0
        if fparts:
            # last deconv + sigmoid + recon loss as one kernel (products on the matrix cores from an LDS-resident pixel
            # block, kept in f32): no product matrix in HBM, no separate col2im pass
            ws = None
            fm = frame_map if frame_map is not None else (0, 0, 0, 0, self.out_ch * H * W)
            if target is not None:
                ws = self._buf((N, "col2im_ws"), max(L.query("rbvae_col2im_ws_floats"), 5 * fparts))
                sse = (ws, fparts, 1.0 / sv.xr.numel())
                if need_grad:
                    sv.dpre3 = self._E(N, H, W, self.out_ch, dtype=torch.float32)
                    sv.b3_parts = (ws, fparts)
            L.call("rbvae_deconv_last_fused", self.dt, sv.d2, self.V3p, self.NY, P(f"decoder_cnn.deconv.{i2}.bias"),
                   self.zero, N, h1, w1, c1, self.out_ch, sv.xr, None if target is None else target.contiguous(), *fm, ws,
                   sv.dpre3, float(recon_gscale))
        else:
            Y = self._E(N * h1 * w1, self.NY)
            self._gemm(sv.d2, self.V3p, Y, None, None, None, N * h1 * w1, 1, 1, 1, 1, 1, 1, 1, 1, c1, self.NY, c1,
                       self.NY, 1, "one")
            if target is not None:
                # persistent (one fused forward is in flight at a time): the backward pass's job table points into it
                ws = self._buf((N, "col2im_ws"), L.query("rbvae_col2im_ws_floats"))
                if defer_losses:    # per-block partial sums stay in ws; rbvae_combine_losses finishes the mean
                    sse = (ws, L.query("rbvae_col2im_nparts", sv.xr.numel()), 1.0 / sv.xr.numel())
                else:
                    mse = self._E(1, dtype=torch.float32)
                if need_grad:
                    sv.dpre3 = self._E(N, H, W, self.out_ch, dtype=torch.float32)
                    if L.query("rbvae_col2im_has_dcol", N, h1, w1, self.NY, H, W, self.out_ch):
                        # the kernel leaves dpre3's per-block column sums (the last deconv's bias gradient) behind the
                        # squared-error sums in ws
                        sv.b3_parts = (ws, L.query("rbvae_col2im_nparts", sv.xr.numel()))
                if frame_map is None:
                    L.call("rbvae_col2im_sigmoid", self.dt, Y, self.NY, P(f"decoder_cnn.deconv.{i2}.bias"), N, h1, w1, H, W,
                           self.out_ch, k, k, 1, sv.xr, target.contiguous(), mse, ws, sv.dpre3, float(recon_gscale), None)
                else:
                    L.call("rbvae_col2im_sigmoid_frames", self.dt, Y, self.NY, P(f"decoder_cnn.deconv.{i2}.bias"), N, h1,
                           w1, H, W, self.out_ch, k, k, 1, sv.xr, target.contiguous(), *frame_map, mse, ws, sv.dpre3,
                           float(recon_gscale), None)
            else:
                L.call("rbvae_col2im_sigmoid", self.dt, Y, self.NY, P(f"decoder_cnn.deconv.{i2}.bias"), N, h1, w1, H, W,
                       self.out_ch, k, k, 1, sv.xr, None, None, None, None, 0.0, None)
        if pending_hs is not None:
            with self._on_side():
                after_hs(pending_hs)
        return {"xr": sv.xr, "hs": hs, "z": sv.z.view(S, T, Ld), "e": sv.e, "kl": kl, "mse": mse, "sse": sse, "saved": sv}

    # ---- backward --------------------------------------------------------------
    def backward(self, flat: torch.Tensor, gflat: torch.Tensor, sv: Saved, g_xr: Optional[torch.Tensor],
                 g_hs: Optional[torch.Tensor], g_z: Optional[torch.Tensor], g_e: Optional[torch.Tensor] = None,
                 kl_weight: float = 0.0, kl_p: float = 0.5, g_hs_inplace: bool = False, side_first=None, cut=None):
        """Writes every parameter gradient into gflat (same layout as flat).
        g_xr: [S,T,C,H,W] upstream gradient of x_recon (None: use the fused dpre3 of forward()).
        g_hs / g_z: [S,T,L] upstream gradients of h_seq / z_seq (None = 0).
        g_e: upstream gradient of the conv logits (simple variant's second output).
        kl_weight: d(loss)/d(kl_mean) when the KL term was fused into forward().
        g_hs_inplace: the caller gives g_hs away (the binarise backward accumulates into it).
        side_first: optional callable issued on the side stream before anything else of this pass (the trainer's
        loss bookkeeping: everything it reads exists once forward() is done).
        cut: optional callable invoked once, on the main stream with every side stream joined, at the point where the
        gradients of decoder_cnn.* and both LSTM stacks (the contiguous tail of gflat from
        layout.offsets["decoder_cnn.fc.weight"]) are final and only the encoder CNN's remain to be computed: the
        data-parallel trainer ends one graph and starts the next there and all-reduces that tail beside the rest."""
        self._join()                       # side-stream work of forward() (after_hs)
        # the loss bookkeeping rides the side stream's fork for the decoder's weight gradients (no edge of its own)
        book_with_decoder = side_first is not None and self.overlap
        if side_first is not None and not book_with_decoder:
            side_first()
        v = self.v
        N, S, T = sv.N, sv.S, sv.T
        H, W = sv.hw
        k, kk = self.k, self.k * self.k
        c1, c2, c3 = v.channels
        (h1, w1), (h2, w2), (h3, w3) = self.g1, self.g2, self.g3
        Ld, lay, nl = self.latent, self.layout, v.lstm_layers
        i0, i1, i2 = lay.conv_idx
        P = lambda name: lay.view(flat, name)
        G = lambda name: lay.view(gflat, name)
        gs = sv.gate_scale
        P1, P2, P3 = N * h1 * w1, N * h2 * w2, N * h3 * w3
        g3 = h3 * w3
        oc = self.out_ch
        self._jobs = JobList()
        f32 = torch.float32

        def tmp(tag, *shape, dtype=None):
            n = math.prod(shape)
            return self._buf((N, tag), n, dtype or self.tdt).view(*shape)

        # --- decoder CNN, data-gradient chain first (main stream) ...
        if g_xr is not None:
            dpre3 = tmp("dpre3", N, H, W, oc, dtype=f32)
            L.call("rbvae_sigmoid_bwd_nhwc", g_xr.contiguous(), sv.xr, dpre3, N, oc, H, W)
        else:
            dpre3 = sv.dpre3
            if dpre3 is None:
                raise RuntimeError("backward without g_xr needs forward(target=..., need_grad=True)")
        dd2 = tmp("dd2", P1, c1)
        nb = (L.query("rbvae_deconv_last_dgrad_blocks", self.dt, oc, H, W, c1, N)
              if self.conv_first_fused and k == 3 and self.K3 == 64 else 0)
        wf3 = self._wf_ksplit(N, oc, H, W, c1) if nb else 0      # the last deconv's weight gradient from dpre3 itself
        col3 = None if wf3 else tmp("col3", P1, self.K3)
        if nb:
            # im2col + GEMM + gate + bias-gradient partial sums in one kernel (csrc/conv_first.hip, MODE 1)
            ws = self._buf((N, "colsum_dd2f"), nb * c1)
            L.call("rbvae_deconv_last_dgrad_fused", self.dt, dpre3, self.V3f, self.zero, col3, sv.d2, dd2, N, oc, H, W, c1,
                   c1, float(gs), ws)
            self._jobs.add(JOB_ROWS, ws, G(f"decoder_cnn.deconv.{i1}.bias"), (1, 1, c1), (0, 0, 1), nslab=nb, slab=c1)
        else:
            L.call("rbvae_im2col", self.dt, dpre3, H * W * oc, 1, W * oc, oc, N, oc, H, W, h1, w1, k, k, 2, 1, self.K3, col3)
            self._gemm(col3, self.V3f, dd2, None, sv.d2, None, P1, 1, 1, 1, 1, 1, 1, 1, 1, self.K3, c1, self.K3, c1, 1,
                       "one", scale=gs, bias_grad=G(f"decoder_cnn.deconv.{i1}.bias"), tag=(N, "dd2"))
        # deconv1 (c2 -> c1): input grad = conv forward of dd2 with the same weights
        dd1 = tmp("dd1", P2, c2)
        self._gemm(dd2, self.V2f, dd1, None, sv.d1, None, N, h1, w1, h2, w2, 2, h2, w2, 1, c1, c2, c1, c2, kk, "conv",
                   scale=gs, bias_grad=G(f"decoder_cnn.deconv.{i0}.bias"), tag=(N, "dd1"))
        # deconv0 (c3 -> c2)
        df = tmp("df", P3, c3)
        self._gemm(dd1, self.V1f, df, None, None, None, N, h2, w2, h3, w3, 2, h3, w3, 1, c2, c3, c2, c3, kk, "conv")

        # ... then its weight / bias gradients, beside the LSTM chain when overlap is on
        def decoder_wgrads():
            if self.overlap:
                self._wg_cus = max(64, 256 - min(S, 128))     # the LSTM backward kernels run beside these: S workgroups
            try:
                decoder_wgrads_()
            finally:
                self._wg_cus = 256
        self._job_blocks = 256      # workgroups per job of a batched job launch

        def decoder_wgrads_():
            if g_xr is None and sv.b3_parts is not None:
                ws_b3, nb3 = sv.b3_parts
                self._jobs.add(JOB_ROWS, ws_b3[nb3:], G(f"decoder_cnn.deconv.{i2}.bias"), (1, 1, oc), (0, 0, 1), nslab=nb3,
                               slab=4)
            else:
                self._colsum(F32, dpre3, N * H * W, oc, oc, G(f"decoder_cnn.deconv.{i2}.bias"), tag=(N, "b3"))
            if wf3:
                self._wgrad_first(1, dpre3, (0, 0, 0, 0, 0), sv.d2, N, oc, H, W, c1, c1, wf3,
                                  G(f"decoder_cnn.deconv.{i2}.weight"), (c1, oc, kk), (self.K3, 1, oc), tag=(N, "V3"))
            else:
                self._wgrad(sv.d2, col3, None, P1, c1, self.K3, c1, self.K3, 1, G(f"decoder_cnn.deconv.{i2}.weight"),
                            (c1, oc, kk), (self.K3, 1, oc), tag=(N, "V3"))
            self._wgrad(sv.d1, dd2, self._conv_idx(N, h1, w1, h2, w2), P2, c2, c1, c2, c1, kk,
                        G(f"decoder_cnn.deconv.{i1}.weight"), (c2, c1, kk), (kk * c1, 1, c1), tag=(N, "V2"),
                        geom=(N, h2, w2, h1, w1))
            self._wgrad(sv.f, dd1, self._conv_idx(N, h2, w2, h3, w3), P3, c3, c2, c3, c2, kk,
                        G(f"decoder_cnn.deconv.{i0}.weight"), (c3, c2, kk), (kk * c2, 1, c2), tag=(N, "V1"),
                        geom=(N, h3, w3, h2, w2))
            # decoder fc: bias = per (position, channel) sum over frames, permuted to the torch (c, hw) order
            nf = L.query("rbvae_colsum_ws_floats", N, self.F3)
            wsf = self._buf((N, "bdfc"), nf)
            L.call("rbvae_colsum_partial", self.dt, df, N, self.F3, self.F3, wsf)
            self._jobs.add(JOB_PERMUTE, wsf, G("decoder_cnn.fc.bias"), (c3, g3, 1), (1, c3, 0), nslab=nf // self.F3,
                           slab=self.F3)
            self._wgrad(df, sv.ds_pad, None, N, self.F3, self.Lp, self.F3, self.Lp, 1, G("decoder_cnn.fc.weight"),
                        (c3, g3, Ld), (self.Lp, c3 * self.Lp, 1), tag=(N, "Wdfc"))

        # The gather-index tables both streams' weight-gradient GEMMs read: complete before they enter the cache
        # (_conv_idx), built here on the main stream before the fork
        self.prepare(N)
        self._fork()
        # with a cut the reductions queued so far (decoder bias sums: their producers ran before the fork) go with the
        # decoder's early reduction, so that every decoder gradient is final at the cut
        fork_jobs = None
        if cut is not None:
            fork_jobs, self._jobs = self._jobs, JobList()
        side_tail = []                # (event, callable): issued on the side stream behind the decoder's weight gradients

        def issue_decoder_side():
            early = self.overlap or cut is not None
            main_jobs, self._jobs = self._jobs, (fork_jobs if cut is not None else JobList())
            with self._on_side():
                if book_with_decoder:
                    side_first()
                decoder_wgrads()
                if early:
                    # the decoder's slab / partial-sum reductions right behind them, not at the end of the pass
                    self._run_jobs()
                else:
                    main_jobs.rows += self._jobs.rows
                    main_jobs.keep += self._jobs.keep
                # late side work: launches of the main chain's own products that nothing on that chain waits for; their
                # reduction jobs join the final reduction
                self._jobs = main_jobs
                for ev, fn in side_tail:
                    self._side.wait_event(ev)
                    fn()
                main_jobs = self._jobs
            self._jobs = main_jobs

        # the main stream's continuation is issued BEFORE the side work (see __init__)
        defer_side = self.overlap
        if not defer_side:
            issue_decoder_side()
        if self.fc_split > 1:
            dds = tmp("dds", self.fc_split, N, Ld, dtype=f32)
            L.call("rbvae_skinny_linear_parts", self.dt, df, self.WdfcT, None, dds, N, Ld, self.F3, self.F3, self.F3,
                   Ld, self.fc_split)
        else:
            dds = tmp("dds", N, Ld, dtype=f32)
            L.call("rbvae_skinny_linear", self.dt, df, self.WdfcT, None, dds, N, Ld, self.F3, self.F3, self.F3, Ld)
        # --- decoder LSTM
        wenc, wdec = P("encoder_rnn.lstm.weight_ih_l0"), P("decoder_rnn.lstm.weight_ih_l0")
        dG = tmp("dG_dec", nl, S, T, 4 * Ld, dtype=f32)
        dGe = tmp("dG_enc", nl, S, T, 4 * Ld, dtype=f32)
        d_in_dec = tmp("d_in_dec", N, Ld, dtype=f32)
        de = tmp("de", N, Ld, dtype=f32)
        pair_bwd = (self.lstm_pair_bwd and not v.simple_order and self.lstm_cast and self.bin_bwd_fused
                    and L.query("rbvae_lstm_pair_bwd_ok", T, Ld, nl))
        if pair_bwd:
            # decoder stack -> binarise backward (+ fused KL) -> encoder stack: one wavefront launch
            de_pad = tmp("de_pad", N, self.Lp)
            de_sums = self._buf((N, "de_sums"), S * Ld)
            nparts = self.fc_split if self.fc_split > 1 else 1
            L.call("rbvae_lstm_pair_bwd", wenc, wdec, sv.acts_enc, sv.cs_enc, sv.acts_dec, sv.cs_dec, dds, nparts, N * Ld,
                   None if g_z is None else g_z.reshape(N, Ld).contiguous(), sv.y, sv.z,
                   None if g_hs is None else g_hs.reshape(N, Ld).contiguous(), float(sv.tau), sv.tau_dev, float(kl_weight),
                   float(kl_p), 1e-8, 1, dGe, dG, de, d_in_dec if self.keep_dz else None, de_pad, self.dt, self.Lp, de_sums,
                   S, T, Ld, nl)
        elif self.fc_split > 1:
            L.call("rbvae_lstm_bwd_ex", wdec, sv.acts_dec, sv.cs_dec, dds, self.fc_split, N * Ld, dG, d_in_dec, None, 0, 0,
                   None, S, T, Ld, nl)
        else:
            L.call("rbvae_lstm_bwd", wdec, self.wT_dec, sv.acts_dec, sv.cs_dec, dds, dG, d_in_dec, S, T, Ld, nl)
        if pair_bwd:
            pass
        elif not v.simple_order:
            # z -> binarise backward (+ fused KL) -> gradient of h_seq
            gz = d_in_dec
            if g_z is not None:
                gz = gz + g_z.reshape(N, Ld)
            de_pad = None
            de_sums = None
            if self.lstm_cast and self.bin_bwd_fused:
                # binarise backward (+ fused KL) in the prologue of the encoder stack's BPTT launch
                de_pad = tmp("de_pad", N, self.Lp)
                de_sums = self._buf((N, "de_sums"), S * Ld)
                L.call("rbvae_lstm_bwd_bin", wenc, sv.acts_enc, sv.cs_enc, gz, sv.y, sv.z,
                       None if g_hs is None else g_hs.reshape(N, Ld).contiguous(), float(sv.tau), sv.tau_dev, float(kl_weight),
                       float(kl_p), 1e-8, 1, dGe, de, de_pad, self.dt, self.Lp, de_sums, S, T, Ld, nl)
            else:
                if g_hs is not None and g_hs_inplace and g_hs.is_contiguous():
                    dh = g_hs.view(N, Ld)
                    L.call("rbvae_binarize_kl_bwd", gz, sv.y, sv.z, dh, 1, N, Ld, float(sv.tau), sv.tau_dev, float(kl_weight), None,
                           float(kl_p), 1e-8, 1)
                else:
                    dh = tmp("dh", N, Ld, dtype=f32)
                    L.call("rbvae_binarize_kl_bwd", gz, sv.y, sv.z, dh, 0, N, Ld, float(sv.tau), sv.tau_dev, float(kl_weight), None,
                           float(kl_p), 1e-8, 1)
                    if g_hs is not None:
                        dh = dh + g_hs.reshape(N, Ld)
                if self.lstm_cast:
                    de_pad = tmp("de_pad", N, self.Lp)
                    de_sums = self._buf((N, "de_sums"), S * Ld)        # per-sequence column sums of de: fc bias gradient
                    L.call("rbvae_lstm_bwd_ex", wenc, sv.acts_enc, sv.cs_enc, dh, 1, 0, dGe, de, de_pad, self.dt, self.Lp,
                           de_sums, S, T, Ld, nl)
                else:
                    L.call("rbvae_lstm_bwd", wenc, self.wT_enc, sv.acts_enc, sv.cs_enc, dh, dGe, de, S, T, Ld, nl)
        else:
            de_pad = None
            de_sums = None
            # decoder stack input = encoder stack output
            dz = tmp("dz", N, Ld, dtype=f32)
            L.call("rbvae_lstm_bwd", wenc, self.wT_enc, sv.acts_enc, sv.cs_enc, d_in_dec, dGe, dz, S, T, Ld, nl)
            L.call("rbvae_binarize_kl_bwd", dz, sv.y, sv.z, de, 0, N, Ld, float(sv.tau), None, 0.0, None, 0.5, 1e-10, 0)
            if g_e is not None:
                de = de + g_e.reshape(N, Ld)
        # Side-stream tail (behind the decoder's weight gradients, each piece behind an event on its inputs): launches
        # of the main chain's products that nothing on the chain waits for
        tail_ok = defer_side and cut is None

        def lstm_wgrads():
            L.call("rbvae_lstm_wgrad_pair", dG, sv.hs_dec, sv.hp_dec, G("decoder_rnn.lstm.weight_ih_l0"),
                   dGe, sv.hs_enc, sv.hp_enc, G("encoder_rnn.lstm.weight_ih_l0"), S, T, Ld, nl, 0)

        if tail_ok:
            ev_bptt = torch.cuda.Event()
            ev_bptt.record(torch.cuda.current_stream())
            side_tail.append((ev_bptt, lstm_wgrads))
        else:
            lstm_wgrads()
        if de_sums is not None and g_e is None:
            self._jobs.add(JOB_ROWS, de_sums, G("encoder_cnn.fc.bias"), (1, 1, Ld), (0, 0, 1), nslab=S, slab=Ld)
        else:
            self._colsum(F32, de, N, Ld, Ld, G("encoder_cnn.fc.bias"), tag=(N, "bfc"))
        # --- encoder fc
        if de_pad is None:
            de_pad = tmp("de_pad", N, self.Lp)
            L.call("rbvae_cast_pad", self.dt, de, de_pad, N, Ld, self.Lp)
        def wfc_wgrad():
            self._wgrad(de_pad, sv.a3, None, N, self.Lp, self.F3, self.Lp, self.F3, 1, G("encoder_cnn.fc.weight"),
                        (Ld, c3, g3), (self.F3, 1, c3), tag=(N, "Wfc"))

        # The encoder fc's weight gradient is a 32-workgroup launch nothing on the data-gradient chain waits for: with
        # the deferred side work it rides the side stream (behind an event on de_pad), so the main chain goes
        # straight on to the next data-gradient GEMM instead of queueing it behind a full chip (18 us in the step)
        if tail_ok:
            ev_de = torch.cuda.Event()
            ev_de.record(torch.cuda.current_stream())
            side_tail.append((ev_de, wfc_wgrad))
        else:
            wfc_wgrad()
        da3 = tmp("da3", P3, c3)
        # the GEMM sees da3 as [N][F3]; its fused column sums [m-tiles][F3] are [m-tiles*g3][c3] rows,
        # so the conv3 bias gradient is one row-reduce job over them
        mt = -(-N // 128)
        ws3 = self._buf((N, "ws3"), mt * self.F3)
        self._gemm(de_pad, self.WfcT, da3, None, sv.a3 if v.simple_order else None, None, N, 1, 1, 1, 1, 1, 1, 1, 1,
                   self.Lp, self.F3, self.Lp, self.F3, 1, "one", colsum_ws=ws3)
        self._jobs.add(JOB_ROWS, ws3, G(f"encoder_cnn.conv.{i2}.bias"), (1, 1, c3), (0, 0, 1), nslab=mt * g3, slab=c3)
        # --- conv3
        idx3, idx2 = self._conv_idx(N, h2, w2, h3, w3), self._conv_idx(N, h1, w1, h2, w2)
        # (the encoder's full-chip weight gradients stay on the main stream: on the side stream's tail they only contend
        # with the data-gradient GEMMs, 0.458 -> 0.476 / 0.487 ms per step in round 2)
        self._wgrad(da3, sv.a2, idx3, P3, c3, c2, c3, c2, kk, G(f"encoder_cnn.conv.{i2}.weight"),
                    (c3, c2, kk), (kk * c2, 1, c2), tag=(N, "W3"), geom=(N, h3, w3, h2, w2))
        da2 = tmp("da2", P2, c2)
        self._gemm(da3, self.W3d, da2, None, sv.a2, None, N, h3, w3, h3, w3, 1, h2, w2, 2, c3, c2, c3, c2, kk, "dgrad",
                   scale=gs, bias_grad=G(f"encoder_cnn.conv.{i1}.bias"), tag=(N, "da2"))
        if cut is not None:
            # here, not right behind the LSTM kernels: the decoder's weight gradients on the side stream take about as
            # long as the main chain needs to get this far, so the join costs no idle time (at the LSTM kernels it cost
            # 76 us per step, one-rank RCCL rehearsal)
            if defer_side:
                issue_decoder_side()
                defer_side = False
            self._join()
            cut()
        # --- conv2
        self._wgrad(da2, sv.a1, idx2, P2, c2, c1, c2, c1, kk, G(f"encoder_cnn.conv.{i1}.weight"),
                    (c2, c1, kk), (kk * c1, 1, c1), tag=(N, "W2"), geom=(N, h2, w2, h1, w1))
        da1 = tmp("da1", P1, c1)
        self._gemm(da2, self.W2d, da1, None, sv.a1, None, N, h2, w2, h2, w2, 1, h1, w1, 2, c2, c1, c2, c1, kk, "dgrad",
                   scale=gs, bias_grad=G(f"encoder_cnn.conv.{i0}.bias"), tag=(N, "da1"))
        # --- conv1 (1-tap GEMM over the saved im2col columns)
        if sv.col1 is None:
            wf1 = self._wf_ksplit(N, self.in_ch, H, W, c1)
            self._wgrad_first(0, sv.x_in, sv.fm_in, da1, N, self.in_ch, H, W, c1, c1, wf1, G(f"encoder_cnn.conv.{i0}.weight"),
                              (c1, self.in_ch, kk), (self.K1, 1, self.in_ch), tag=(N, "W1"))
        else:
            self._wgrad(da1, sv.col1, None, P1, c1, self.K1, c1, self.K1, 1, G(f"encoder_cnn.conv.{i0}.weight"),
                        (c1, self.in_ch, kk), (self.K1, 1, self.in_ch), tag=(N, "W1"))
        if defer_side:
            issue_decoder_side()
        # every slab / partial-sum reduction of this pass in one launch, once the side stream has caught up
        self._join()
        self._run_jobs()
