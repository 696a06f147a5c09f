"""Fused training step of the RBVAE path: the reference trainer's inner loop
(models/percep_RBVAE/percep_RBVAE_train.py:509-553; triplet variant
models/triplet_RBVAE/triplet_RBVAE_train.py:443-478) as one hand-scheduled pass.

  item [B,2,T,C,H,W] -> both views through the model as 2B sequences in ONE forward
  (the model has no cross-item op, so this equals the reference's two calls),
  recon MSE + its gradient fused into the last deconv kernel, KL fused into the
  binarise kernel, the pairwise term in one launch, hand-scheduled backward,
  gradient all-reduce (RCCL) when world_size > 1, fused Adam on the flat buffer.

The step is captured into HIP graphs (forward+backward; Adam+repack) so a step costs
two graph launches and, across ranks, one all-reduce between them.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import _lib as L
from .engine import VARIANTS


class FusedTrainer:
    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, alpha=0.1, beta_kl=0.1, bernoulli_p=0.5,
                 noise_ratio=0.1, margin=1.0, device_noise=True, use_graph=True, process_group=None):
        self.model = model
        self.v = VARIANTS[model.variant]
        if self.v.simple_order:
            raise ValueError("FusedTrainer covers the percep / contrastive / triplet trainers")
        self.lr, self.betas, self.eps = lr, betas, eps
        self.alpha, self.beta_kl, self.p, self.r, self.margin = alpha, beta_kl, bernoulli_p, noise_ratio, margin
        self.device_noise = device_noise
        self.use_graph = use_graph
        self.pg = process_group
        self.world = 1
        if process_group is not None or (torch.distributed.is_available() and torch.distributed.is_initialized()):
            self.world = torch.distributed.get_world_size(process_group)
        dev = model._flat.device
        if dev.type != "cuda":
            raise RuntimeError("FusedTrainer: move the model to the GPU first (there is no CPU fallback)")
        self.dev = dev
        n = model._flat.numel()
        self.gflat = torch.zeros(n, device=dev)
        self.m = torch.zeros(n, device=dev)
        self.vv = torch.zeros(n, device=dev)
        self.step_dev = torch.zeros(1, dtype=torch.int64, device=dev)       # device step counter
        self.hyper = torch.zeros(2, device=dev)
        self.losses = torch.zeros(4, device=dev)                               # total, recon, kl, pair
        self._pair = torch.zeros(1, device=dev)
        self.steps = 0
        self._graphs: Dict = {}
        self._static: Dict = {}
        self.eng = None
        self.instrument = None       # optional callable(name, flops) -> context manager (bench roofline leg)
        import os
        self.one_graph = os.environ.get("RBVAE_ONE_GRAPH", "1") == "1"
        # world > 1: the backward pass is cut where the decoder CNN's and the LSTM stacks' gradients are final (the
        # contiguous tail of the flat buffer, 51 % of it at the headline config); that tail is all-reduced on the
        # collective's own stream beside the encoder CNN's backward graph, the head after it.  RBVAE_DDP_OVERLAP=0:
        # one all-reduce of the whole buffer between the backward graph and the Adam graph.
        self.ddp_overlap = os.environ.get("RBVAE_DDP_OVERLAP", "1") == "1"
        self.fused_pair = os.environ.get("RBVAE_FUSED_PAIR", "1") == "1"

    # ---- the two halves of a step (plain launches; captured below) ------------------
    def _fwd_bwd(self, x, U, tau, B, T, cut=None):
        eng, model = self.eng, self.model
        numel = x.numel()
        Ld = model.latent_dim
        g_hs = torch.empty(2 * B, T, Ld, device=self.dev)
        pair = self._pair

        fused_pair = self.fused_pair and self.model.variant != "triplet"
        pair_parts = torch.empty(2 * L.query("rbvae_contrast_term_nparts", B, T), device=self.dev) if fused_pair else None

        def pair_term(hs):                   # [2B, T, L]
            h0, h1 = hs[:B], hs[B:]
            if self.model.variant == "triplet":
                L.call("rbvae_triplet_term_fwd", h0, h1, B, T, Ld, float(self.margin), pair)
                L.call("rbvae_triplet_term_bwd", h0, h1, B, T, Ld, float(self.margin), float(self.alpha), None,
                       g_hs[:B], g_hs[B:])
            elif fused_pair:
                # value (as per-block sums for the bookkeeping kernel) and gradient in one many-workgroup launch
                L.call("rbvae_contrast_term_fused", h0, h1, B, T, Ld, float(self.alpha), None, pair_parts, g_hs[:B],
                       g_hs[B:])
            else:
                L.call("rbvae_contrast_term_fwd", h0, h1, B, T, Ld, pair)
                L.call("rbvae_contrast_term_bwd", h0, h1, B, T, Ld, float(self.alpha), None, g_hs[:B], g_hs[B:])

        # dropout follows the module's mode like the reference (model.train() in train_one_epoch, :501)
        # x is the item batch [B, 2, T, C, H, W] as it is; frame (v, b, t) = sequence v*B + b, state t
        chw = numel // (2 * B * T)
        out = eng.forward(model._flat, x.view(2 * B, T, *x.shape[3:]), U, tau, False, self.r, bool(model.training), None,
                          seed=0, need_grad=True, target=x, recon_gscale=2.0 / numel, kl_p=self.p, after_hs=pair_term,
                          defer_losses=True, repack=True, frame_map=(B * T, T, T * chw, 2 * T * chw, chw))
        sse_ws, nparts, inv_n = out["sse"]
        kl_parts, nkl, kl_scale = out["kl"]
        b1, b2 = self.betas

        def bookkeeping():
            # [total, recon, kl, pair] from the partial sums; also advances the device step counter and prepares
            # Adam's bias corrections for _update().  Runs on the side stream beside the backward pass.
            if fused_pair:
                pargs = (pair_parts, pair_parts.numel() // 2, 1.0 / (B * T), 1.0 / (B * (T - 1)))
            else:
                pargs = (self._pair, 0, 0.0, 0.0)
            L.call("rbvae_combine_losses", sse_ws, nparts, inv_n, None, kl_parts, nkl, kl_scale, *pargs,
                   float(self.beta_kl), float(self.alpha), self.losses, self.step_dev, float(self.lr), float(b1),
                   float(b2), self.hyper)

        eng.backward(model._flat, self.gflat, out["saved"], None, g_hs, None, kl_weight=self.beta_kl, kl_p=self.p,
                     g_hs_inplace=True, side_first=bookkeeping, cut=cut)

    def _update(self):
        b1, b2 = self.betas
        L.call("rbvae_adam_step", self.model._flat, self.gflat, self.m, self.vv, self.gflat.numel(), float(self.lr),
               float(b1), float(b2), float(self.eps), 0, 1.0 / self.world, None, self.hyper)
        # the packed bf16 copies are refreshed at the start of the next step (beside its first kernels)

    # ---- public --------------------------------------------------------------------
    def step(self, item: torch.Tensor, temperature: float, U: Optional[torch.Tensor] = None):
        """One optimiser step on item [B,2,T,C,H,W] (device tensor).  U: optional [2, B*T, L] uniform
        noise (view 0, view 1) -- default: device-side counter-hash noise (device_noise=True) or a host
        torch.rand draw like the reference (device_noise=False).  Returns the device tensor
        [total, recon, kl, pair] of this step (no host sync)."""
        if item.dim() != 6 or item.shape[1] != 2:
            raise ValueError(f"expected item of shape [B, 2, T, C, H, W], got {tuple(item.shape)}")
        B, _, T, C, H, W = item.shape
        if T < 2:
            raise ZeroDivisionError("float division by zero")      # the reference's failure for one state
        model = self.model
        if self.eng is None:
            self.eng = model._engine_for(item)
            self.eng.seed_dev = self.step_dev
        Ld = model.latent_dim
        if U is None and not self.device_noise:
            U = torch.rand((2, B * T, Ld)).to(item.device)
        key = (B, T, float(temperature), U is not None)
        st = self._static.get(key[:2])
        if st is None:
            st = {"x": torch.empty(B, 2, T, C, H, W, device=self.dev),
                  "U": torch.empty(2 * B * T, Ld, device=self.dev)}
            self._static[key[:2]] = st
        if item.data_ptr() != st["x"].data_ptr() or not item.is_contiguous():
            st["x"].copy_(item)          # skipped when the caller fills input_buffer() in place
        if U is not None:
            st["U"].copy_(U.reshape(2 * B * T, Ld))
        Uarg = st["U"] if U is not None else None
        if not self.use_graph or self.instrument is not None:
            self._fwd_bwd(st["x"], Uarg, float(temperature), B, T)
            self._allreduce()
            self._update()
        else:
            g = self._graphs.get(key)
            if g is None:
                g = self._capture(st["x"], Uarg, float(temperature), B, T)
                self._graphs[key] = g
            g[0].replay()
            if len(g) == 3:
                # tail (decoder CNN + LSTM gradients) on the collective's stream beside the encoder CNN's backward graph
                tail, head = self._grad_buckets()
                w = torch.distributed.all_reduce(tail, group=self.pg, async_op=True)
                g[1].replay()
                torch.distributed.all_reduce(head, group=self.pg)
                w.wait()
                g[2].replay()
            elif g[1] is not None:
                self._allreduce()
                g[1].replay()
        self.steps += 1
        self.model._packed_version = None     # anything else that runs the model before the next step repacks first
        return self.losses

    def input_buffer(self, B: int, T: int, C: int, H: int, W: int) -> torch.Tensor:
        """The step's static input [B, 2, T, C, H, W] (the address the captured graphs read).  A data loader that
        writes its batch here (e.g. DeviceStatePairDataset gathering with out=) and passes the same tensor to
        step() saves the device-to-device copy of the batch."""
        st = self._static.get((B, T))
        if st is None:
            st = {"x": torch.empty(B, 2, T, C, H, W, device=self.dev),
                  "U": torch.empty(2 * B * T, self.model.latent_dim, device=self.dev)}
            self._static[(B, T)] = st
        return st["x"]

    def _allreduce(self):
        if self.world > 1:
            torch.distributed.all_reduce(self.gflat, group=self.pg)

    def _grad_buckets(self):
        """(tail, head) views of the flat gradient: tail = decoder_cnn.* and both LSTM stacks (final at the backward
        pass's cut), head = encoder_cnn.* (final at its end)."""
        o = self.eng.layout.offsets["decoder_cnn.fc.weight"]
        return self.gflat[o:], self.gflat[:o]

    def _capture(self, x, U, tau, B, T):
        # two eager warm-up steps on a side stream (allocator + lazy kernel attributes), then capture.
        # The warm-ups are real optimiser steps; their effect on the counters is rolled back.
        flat0, m0, v0, s0 = self.model._flat.clone(), self.m.clone(), self.vv.clone(), self.step_dev.clone()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):
                self._fwd_bwd(x, U, tau, B, T, cut=(lambda: None) if (self.world > 1 and self.ddp_overlap) else None)
                self._update()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g1, g2 = torch.cuda.CUDAGraph(), None
        if self.world == 1 and self.one_graph:
            # no collective between backward and Adam: the whole step is one graph launch
            with torch.cuda.graph(g1):
                self._fwd_bwd(x, U, tau, B, T)
                self._update()
        elif self.world > 1 and self.ddp_overlap:
            # three graphs: forward + backward up to the cut | the encoder CNN's backward | Adam
            g2, g3 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            cs = torch.cuda.Stream()
            cs.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(cs):
                g1.capture_begin()

                def cut():
                    g1.capture_end()
                    g2.capture_begin(pool=g1.pool())

                self._fwd_bwd(x, U, tau, B, T, cut=cut)
                g2.capture_end()
                g3.capture_begin(pool=g1.pool())
                self._update()
                g3.capture_end()
            torch.cuda.current_stream().wait_stream(cs)
            self.model._flat.copy_(flat0)
            self.m.copy_(m0)
            self.vv.copy_(v0)
            self.step_dev.copy_(s0)
            self.eng.pack(self.model._flat)
            torch.cuda.synchronize()
            return g1, g2, g3
        else:
            g2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g1):
                self._fwd_bwd(x, U, tau, B, T)
            with torch.cuda.graph(g2, pool=g1.pool()):
                self._update()
        self.model._flat.copy_(flat0)
        self.m.copy_(m0)
        self.vv.copy_(v0)
        self.step_dev.copy_(s0)
        self.eng.pack(self.model._flat)
        torch.cuda.synchronize()
        return g1, g2
