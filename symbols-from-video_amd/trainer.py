"""Fused training step of the RBVAE path: the reference trainer's inner loop
(models/percep_RBVAE/percep_RBVAE_train.py:509-553; triplet variant
models/triplet_RBVAE/triplet_RBVAE_train.py:443-478) as one hand-scheduled pass.

  item [B,2,T,C,H,W] -> both views through the model as 2B sequences in ONE forward
  (the model has no cross-item op, so this equals the reference's two calls),
  recon MSE + its gradient fused into the last deconv kernel, KL fused into the
  binarise kernel, the pairwise term in one launch, hand-scheduled backward,
  gradient all-reduce (RCCL) when world_size > 1, fused Adam on the flat buffer.

The step is captured into HIP graphs (one graph on a single GPU; across ranks the graphs are cut where a
gradient bucket is final and the all-reduces run between / beside them).  Temperature and learning rate are
read by the kernels from device scalars, so ONE graph per batch shape serves the whole annealing schedule
(percep_RBVAE_train.py:424-437).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.distributed as dist

from . import _lib as L
from .engine import VARIANTS


def noise_key(seed: int, rank: int) -> int:
    """Key of a rank's dropout / Binary-Concrete noise streams: splitmix64 of (seed, rank), 58 bits (the engine
    appends 3 bits of site index).  Different ranks and different seeds draw unrelated streams; the device step
    counter is mixed in by the kernels, so a resumed run continues the stream of its seed."""
    m = (1 << 64) - 1
    x = (int(seed) * 0x9E3779B97F4A7C15 + (int(rank) + 1) * 0xD1B54A32D192ED03) & m
    x ^= x >> 30
    x = (x * 0xBF58476D1CE4E5B9) & m
    x ^= x >> 27
    x = (x * 0x94D049BB133111EB) & m
    x ^= x >> 31
    return x >> 6


class FusedTrainer:
    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, alpha=0.1, beta_kl=0.1, bernoulli_p=0.5,
                 noise_ratio=0.1, margin=1.0, device_noise=True, use_graph=True, process_group=None, seed=None,
                 pair_loss=None):
        """seed: key of the device-side noise (dropout masks, Binary-Concrete uniforms); default torch.initial_seed(),
        so torch.manual_seed() before constructing the trainer makes a run reproducible.  The rank is mixed in:
        every data-parallel rank draws its own streams.
        pair_loss: "contrast" (percep_RBVAE_train.py:534-543) or "triplet" (triplet_RBVAE_train.py:461-468);
        default: the model variant's own trainer.  Independent of the conv widths (BASELINE configs[4]: the
        percep-shaped network on LDM latents with the triplet term)."""
        self.model = model
        self.v = VARIANTS[model.variant]
        if self.v.simple_order:
            raise ValueError("FusedTrainer covers the percep / contrastive / triplet trainers")
        if pair_loss is None:
            pair_loss = "triplet" if model.variant == "triplet" else "contrast"
        if pair_loss not in ("contrast", "triplet"):
            raise ValueError("pair_loss must be 'contrast' or 'triplet'")
        self.pair_loss = pair_loss
        self.lr, self.betas, self.eps = lr, betas, eps
        self.alpha, self.beta_kl, self.p, self.r, self.margin = alpha, beta_kl, bernoulli_p, noise_ratio, margin
        self.device_noise = device_noise
        self.use_graph = use_graph
        self.pg = process_group
        self.world, self.rank = 1, 0
        if process_group is not None or (dist.is_available() and dist.is_initialized()):
            self.world = dist.get_world_size(process_group)
            self.rank = dist.get_rank(process_group)
        self.seed = int(torch.initial_seed() if seed is None else seed)
        self._noise_key = noise_key(self.seed, self.rank)
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
        # schedule scalars the captured kernels read: temperature (f32) and learning rate (f64)
        self.tau_dev = torch.ones(1, device=dev)
        self.lr_dev = torch.full((1,), float(lr), dtype=torch.float64, device=dev)
        self._tau_host, self._lr_host = 1.0, float(lr)
        self.losses = torch.zeros(4, device=dev)                               # total, recon, kl, pair
        self._pair = torch.zeros(1, device=dev)
        self.steps = 0
        self._graphs: Dict = {}
        self._pool = None             # one memory pool shared by every captured graph of this trainer
        self._static: Dict = {}
        self._data = None             # (table, plan_dev [n_batches, B*2*T], B, T): device-resident loader, see set_data()
        self.eng = None
        self.instrument = None       # optional callable(name, flops) -> context manager (bench roofline leg)
        import os
        # single GPU: the whole step is ONE graph.  False (tools/rccl_single.py only): the multi-graph schedules of the
        # data-parallel trainer rehearsed on one rank
        self.one_graph = True
        # world > 1: the backward pass is cut where the decoder CNN's and the LSTM stacks' gradients are final (the
        # contiguous tail of the flat buffer, 51 % of it at the headline config); that tail is all-reduced on the
        # collective's own stream beside the encoder CNN's backward graph, the head after it.  RBVAE_DDP_OVERLAP=0:
        # one all-reduce of the whole buffer between the backward graph and the Adam graph.
        self.ddp_overlap = os.environ.get("RBVAE_DDP_OVERLAP", "1") == "1"
        # with the overlap schedule: update the tail bucket's parameters while the head's all-reduce is in flight.  Off: the
        # one-rank RCCL rehearsal (tools/rccl_single.py) prices the fourth graph at +17 us per step (0.505 -> 0.522 ms),
        # more than the 14 us of update it can hide; in the in-graph schedule it costs 9 us
        self.ddp_split_update = os.environ.get("RBVAE_DDP_SPLIT_UPDATE", "0") == "1"
        # RBVAE_DDP_INGRAPH=1 (opt-in; rehearsed with a one-rank RCCL group only, tools/rccl_single.py): the two all-reduces
        # are captured INTO the step's single graph on a communication stream -- no graph boundary at the cut, no
        # host-side choreography between replays
        self.ddp_ingraph = os.environ.get("RBVAE_DDP_INGRAPH", "0") == "1"
        self._packed_ver = None
        self._red = None
        # The step's boundary is ONE batched job launch (engine.update_jobs): every parameter tensor's job applies Adam to
        # its slice of the flat buffers and writes the packed copies from the new values while it holds them; in set_data
        # mode the same launch gathers the NEXT step's batch.  The step then opens directly with the first convolution: no
        # repack launch, no gather launch, no separate Adam launch.  (Measured in round 2 and not kept: per-group updates on
        # the side stream beside the backward GEMMs 0.475 vs 0.470 ms/step; the next batch gathered on the side stream
        # during the backward pass 0.4542 vs 0.4508.)
        self._primed = False          # set_data mode: the input buffer holds the batch of the coming step

    # ---- the two halves of a step (plain launches; captured below) ------------------
    def _fwd_bwd(self, x, U, tau, B, T, cut=None, masks=None):
        eng, model = self.eng, self.model
        numel = x.numel()
        Ld = model.latent_dim
        g_hs = torch.empty(2 * B, T, Ld, device=self.dev)
        pair = self._pair

        fused_pair = self.pair_loss != "triplet"    # the contrast term: value + gradient in one many-workgroup launch
        pair_parts = torch.empty(2 * L.query("rbvae_contrast_term_nparts", B, T), device=self.dev) if fused_pair else None

        def pair_term(hs):                   # [2B, T, L]
            h0, h1 = hs[:B], hs[B:]
            if self.pair_loss == "triplet":
                L.call("rbvae_triplet_term_fwd", h0, h1, B, T, Ld, float(self.margin), pair)
                L.call("rbvae_triplet_term_bwd", h0, h1, B, T, Ld, float(self.margin), float(self.alpha), None,
                       g_hs[:B], g_hs[B:])
            else:
                # value (as per-block sums for the bookkeeping kernel) and gradient in one many-workgroup launch
                L.call("rbvae_contrast_term_fused", h0, h1, B, T, Ld, float(self.alpha), None, pair_parts, g_hs[:B],
                       g_hs[B:])

        # (set_data mode: this step's batch was gathered by the previous step's update launch, or by step()'s priming)
        # dropout follows the module's mode like the reference (model.train() in train_one_epoch, :501)
        # x is the item batch [B, 2, T, C, H, W] as it is; frame (v, b, t) = sequence v*B + b, state t
        chw = numel // (2 * B * T)
        out = eng.forward(model._flat, x.view(2 * B, T, *x.shape[3:]), U, tau, False, self.r, bool(model.training), masks,
                          seed=self._noise_key, need_grad=True, target=x, recon_gscale=2.0 / numel, kl_p=self.p,
                          after_hs=pair_term, defer_losses=True,
                          frame_map=(B * T, T, T * chw, 2 * T * chw, chw), tau_dev=self.tau_dev)
        sse_ws, nparts, inv_n = out["sse"]
        kl_parts, nkl, kl_scale = out["kl"]
        self.last_saved = out["saved"]          # diagnostics (tests read the device's ReLU decisions from it)
        b1, b2 = self.betas

        def bookkeeping():
            # [total, recon, kl, pair] from the partial sums; also advances the device step counter and prepares
            # Adam's bias corrections for _update().  Runs on the side stream beside the backward pass.
            if fused_pair:
                pargs = (pair_parts, pair_parts.numel() // 2, 1.0 / (B * T), 1.0 / (B * (T - 1)))
            else:
                pargs = (self._pair, 0, 0.0, 0.0)
            L.call("rbvae_combine_losses", sse_ws, nparts, inv_n, None, kl_parts, nkl, kl_scale, *pargs,
                   float(self.beta_kl), float(self.alpha), self.losses, self.step_dev, float(self.lr), self.lr_dev,
                   float(b1), float(b2), self.hyper)

        eng.backward(model._flat, self.gflat, out["saved"], None, g_hs, None, kl_weight=self.beta_kl, kl_p=self.p,
                     g_hs_inplace=True, side_first=bookkeeping, cut=cut)

    def _gather_row(self, x):
        table, plan, _, _ = self._data
        from .engine import JOB_GATHER
        return [JOB_GATHER, table.data_ptr(), x.data_ptr(), plan.shape[1], plan.shape[0], table[0].numel() // 4,
                plan.data_ptr(), self.step_dev.data_ptr(), table.shape[0], 1, 0, 0, 0, 0, 0, 0]

    def _update(self, part=None):
        """Adam + weight repack (+ the next batch's gather in set_data mode) as one job launch.
        part: None, or "tail" / "head": one gradient bucket's parameters (data-parallel split update)."""
        extra = None
        if self._data is not None and self._data_active and part != "tail":
            B, T = self._data[2], self._data[3]
            extra = [self._gather_row(self._static[(B, T)]["x"])]      # the next step's batch (the counter has advanced)
        tab, n = self.eng.update_jobs(self.model._flat, self.gflat, self.m, self.vv, self.hyper, self.betas, self.eps,
                                      1.0 / self.world, extra, part)
        self.eng.run_table(tab, n, 1)

    # ---- public --------------------------------------------------------------------
    def step(self, item: Optional[torch.Tensor], temperature: float, U: Optional[torch.Tensor] = None,
             dropout_masks=None):
        """One optimiser step on item [B,2,T,C,H,W] (device tensor), or on the next batch of set_data()'s plan when
        item is None.  U: optional [2, B*T, L] uniform
        noise (view 0, view 1) -- default: device-side counter-hash noise (device_noise=True) or a host
        torch.rand draw like the reference (device_noise=False).
        dropout_masks: optional explicit keep-masks [view][site] of shape [B*T, C, H, W] (torch layout) for the four
        Dropout sites (reproducible parity runs; such a step runs eagerly, not from the captured graph).
        Returns the device tensor [total, recon, kl, pair] of this step (no host sync)."""
        from_data = item is None
        if from_data:
            if self._data is None:
                raise ValueError("step(None, ...) needs set_data() first")
            table, _, B, T = self._data
            C, H, W = table.shape[1:]
            item = self.input_buffer(B, T, C, H, W)
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
        self._data_active = from_data
        # the step leaves every packed weight copy current; repack here only when the weights changed behind the
        # trainer's back (first step, load_state_dict, a foreign optimiser)
        ver = tuple(p._version for p in model._params())
        if self._packed_ver != ver or model._packed_version != (id(self.eng), ver):
            self.eng.pack(model._flat)
            self._packed_ver = ver
        if U is None and not self.device_noise:
            U = torch.rand((2, B * T, Ld)).to(item.device)
        # temperature / lr travel through device scalars: they are NOT part of the graph key
        key = (B, T, U is not None, bool(model.training), from_data)
        self._set_schedule(float(temperature))
        st = self._static.get(key[:2])
        if st is None:
            # zeros, not empty: in set_data mode the capture's warm-up steps run before the first batch is gathered
            st = {"x": torch.zeros(B, 2, T, C, H, W, device=self.dev),
                  "U": torch.zeros(2 * B * T, Ld, device=self.dev)}
            self._static[key[:2]] = st
        if item.data_ptr() != st["x"].data_ptr() or not item.is_contiguous():
            st["x"].copy_(item)          # skipped when the caller fills input_buffer() in place
        if U is not None:
            st["U"].copy_(U.reshape(2 * B * T, Ld))
        Uarg = st["U"] if U is not None else None
        masks = None
        if dropout_masks is not None and model.training:
            from .model import _mask_to_rows
            masks = [torch.cat([_mask_to_rows(dropout_masks[0][j].to(self.dev)),
                                _mask_to_rows(dropout_masks[1][j].to(self.dev))]) for j in range(4)]
        graph = None
        if self.use_graph and self.instrument is None and masks is None:
            graph = self._graphs.get(key)
            if graph is None:
                graph = self._capture(st["x"], Uarg, float(temperature), B, T)
                self._graphs[key] = graph
                self._primed = False      # the capture's warm-up steps gathered ahead and were rolled back
        if from_data and not self._primed:
            # first step of a plan (or after anything else used the buffer): gather this step's batch now; from here on
            # every step gathers its successor's
            table, plan, _, _ = self._data
            L.call("rbvae_gather_frames", table, table.shape[0], plan, plan.shape[1], plan.shape[0], self.step_dev,
                   table[0].numel(), st["x"])
        self._primed = from_data
        if graph is None:
            self._fwd_bwd(st["x"], Uarg, float(temperature), B, T, masks=masks)
            self._allreduce()
            self._update()
        else:
            g = graph
            g[0].replay()
            if len(g) == 4:
                # tail (decoder CNN + LSTM gradients) all-reduced on the collective's stream beside the encoder CNN's
                # backward graph; the head's all-reduce beside the tail's optimiser update + repack
                red = self._reducer()
                wt = red.start_tail()
                g[1].replay()
                wh = red.start_head()
                red.wait(wt)
                g[2].replay()
                red.wait(wh)
                g[3].replay()
            elif len(g) == 3:
                # tail (decoder CNN + LSTM gradients) on the collective's stream beside the encoder CNN's backward graph
                red = self._reducer()
                w = red.start_tail()
                g[1].replay()
                red.finish(w)
                g[2].replay()
            elif g[1] is not None:
                self._allreduce()
                g[1].replay()
        self.steps += 1
        self.model._packed_version = (id(self.eng), self._packed_ver)      # the packed copies are current
        return self.losses

    def validate(self, item: torch.Tensor, temperature: float, U: Optional[torch.Tensor] = None) -> torch.Tensor:
        """The reference's validation step on one batch (percep_RBVAE_train.py:590-635): dropout off, hard codes at the
        given (final) temperature, total = (recon + beta*kl + alpha*pair) / (1 + alpha + beta).  No gradients, no
        optimiser state touched.  item [B,2,T,C,H,W]; U optional [2, B*T, L] (default: a host torch.rand draw like
        the reference's forward).  Returns a new device tensor [total, recon, kl, pair]."""
        if item.dim() != 6 or item.shape[1] != 2:
            raise ValueError(f"expected item of shape [B, 2, T, C, H, W], got {tuple(item.shape)}")
        B, _, T, C, H, W = item.shape
        if T < 2:
            raise ZeroDivisionError("float division by zero")
        model = self.model
        if self.eng is None:
            self.eng = model._engine_for(item)
            self.eng.seed_dev = self.step_dev
        eng, Ld = self.eng, model.latent_dim
        if U is None:
            U = torch.rand((2, B * T, Ld)).to(item.device)
        U = U.reshape(2 * B * T, Ld).float().contiguous()
        x = item.float().contiguous()
        chw = C * H * W
        model._pack()
        with torch.no_grad():
            out = eng.forward(model._flat, x.view(2 * B, T, C, H, W), U, float(temperature), True, self.r, False, None,
                              seed=self._noise_key, need_grad=False, target=x, recon_gscale=0.0, kl_p=self.p,
                              defer_losses=True, frame_map=(B * T, T, T * chw, 2 * T * chw, chw))
            hs = out["hs"]
            h0, h1 = hs[:B], hs[B:]
            pair = torch.empty(1, device=self.dev)
            if self.pair_loss == "triplet":
                L.call("rbvae_triplet_term_fwd", h0, h1, B, T, Ld, float(self.margin), pair)
            else:
                L.call("rbvae_contrast_term_fwd", h0, h1, B, T, Ld, pair)
            sse_ws, nparts, inv_n = out["sse"]
            kl_parts, nkl, kl_scale = out["kl"]
            res = torch.empty(4, device=self.dev)
            L.call("rbvae_combine_losses", sse_ws, nparts, inv_n, None, kl_parts, nkl, kl_scale, pair, 0, 0.0, 0.0,
                   float(self.beta_kl), float(self.alpha), res, None, 0.0, None, 0.0, 0.0, None)
            res[0] /= 1.0 + float(self.alpha) + float(self.beta_kl)
        return res

    # ---- checkpoint state (percep_RBVAE_train.py:691-702 stores optimizer.state_dict()) -----------------
    def state_dict(self) -> Dict:
        """{"optimizer_state_dict": torch.optim.Adam-format state over the model's parameters in registration
        order (loadable by a real torch.optim.Adam(model.parameters())), "seed", "steps"}."""
        lay = self.model._layout
        step = int(self.step_dev.item())
        state = {}
        if step > 0:
            for i, n in enumerate(lay.names):
                state[i] = {"step": torch.tensor(float(step)), "exp_avg": lay.view(self.m, n).clone(),
                            "exp_avg_sq": lay.view(self.vv, n).clone()}
        group = {"lr": float(self.lr), "betas": tuple(self.betas), "eps": float(self.eps), "weight_decay": 0,
                 "amsgrad": False, "maximize": False, "foreach": None, "capturable": False, "differentiable": False,
                 "fused": None, "decoupled_weight_decay": False, "params": list(range(len(lay.names)))}
        return {"optimizer_state_dict": {"state": state, "param_groups": [group]}, "seed": self.seed,
                "steps": self.steps}

    def load_state_dict(self, sd: Dict):
        """Accepts state_dict()'s output or a bare torch.optim.Adam state_dict (the reference's checkpoint entry)."""
        opt = sd.get("optimizer_state_dict", sd)
        lay = self.model._layout
        groups = opt["param_groups"]
        if len(groups) != 1 or len(groups[0]["params"]) != len(lay.names):
            raise ValueError("optimizer state does not match this model's parameter list")
        g = groups[0]
        if g.get("weight_decay", 0) or g.get("amsgrad", False) or g.get("maximize", False):
            raise ValueError("FusedTrainer implements plain Adam (no weight decay / amsgrad / maximize)")
        self.lr, self.betas, self.eps = float(g["lr"]), tuple(g["betas"]), float(g["eps"])
        state = opt["state"]
        steps = {int(float(st["step"])) for st in state.values()}
        if len(steps) > 1:
            raise ValueError("per-parameter step counts differ; the fused optimiser keeps one")
        self.m.zero_()
        self.vv.zero_()
        for i, n in enumerate(lay.names):
            st = state.get(g["params"][i])
            if st is not None:
                lay.view(self.m, n).copy_(st["exp_avg"])
                lay.view(self.vv, n).copy_(st["exp_avg_sq"])
        self.step_dev.fill_(steps.pop() if steps else 0)
        if "seed" in sd:
            self.seed = int(sd["seed"])
            self._noise_key = noise_key(self.seed, self.rank)
        # betas / eps / the noise key are by-value arguments of the captured launches (the update job table is keyed by
        # them and rebuilt by the next capture's warm-up steps); the static input buffer holds the batch gathered for the OLD step counter: drop every captured graph
        # and re-prime (a bare torch.optim.Adam state_dict with other betas would otherwise be ignored by the graphs)
        self._graphs.clear()
        self._primed = False
        self.steps = int(sd.get("steps", int(self.step_dev.item())))

    def set_data(self, table: torch.Tensor, plan: torch.Tensor):
        """Device-resident data loading: `table` [F,C,H,W] f32 latents in HBM
        (DeviceStatePairDataset.table), `plan` [n_batches, B, 2, T] int64 table rows -- an epoch's batches laid out in
        advance (DeviceStatePairDataset.plan()).  step(None, tau) then trains on batch (steps so far) % n_batches,
        gathered by a kernel inside the captured step: no host work and no separate copy of the batch per step.
        A new plan of the same shape (next epoch's shuffle) is copied into the same device buffer: the graph stays."""
        if table.dim() != 4 or plan.dim() != 4 or plan.shape[2] != 2:
            raise ValueError("table must be [F,C,H,W], plan [n_batches,B,2,T]")
        if table.dtype != torch.float32 or table.device != self.dev or not table.is_contiguous():
            raise ValueError("table must be a contiguous f32 tensor on the trainer's device")
        if table[0].numel() % 4:
            raise ValueError("frame size must be a multiple of 4 floats")
        plan = plan.to(torch.int64)
        if plan.numel() == 0 or int(plan.min()) < 0 or int(plan.max()) >= table.shape[0]:
            raise ValueError("plan is empty or holds rows outside the table")
        nb, B, _, T = plan.shape
        flat = plan.reshape(nb, B * 2 * T).to(self.dev).contiguous()
        old = self._data
        if old is not None and old[0].data_ptr() == table.data_ptr() and old[1].shape == flat.shape and old[2:] == (B, T):
            old[1].copy_(flat)                   # same table, same batch geometry: the captured graphs stay
        else:
            self._data = (table, flat, B, T)
            self._graphs = {k: g for k, g in self._graphs.items() if not k[4]}
        self._primed = False
    _data_active = False

    def input_buffer(self, B: int, T: int, C: int, H: int, W: int) -> torch.Tensor:
        """The step's static input [B, 2, T, C, H, W] (the address the captured graphs read).  A data loader that
        writes its batch here (e.g. DeviceStatePairDataset gathering with out=) and passes the same tensor to
        step() saves the device-to-device copy of the batch."""
        st = self._static.get((B, T))
        if st is None:
            st = {"x": torch.zeros(B, 2, T, C, H, W, device=self.dev),
                  "U": torch.zeros(2 * B * T, self.model.latent_dim, device=self.dev)}
            self._static[(B, T)] = st
        return st["x"]

    def _set_schedule(self, tau: float):
        """Write the step's temperature / learning rate into the device scalars (stream-ordered fills, only when
        the value changed: the reference updates tau every num_steps_to_update steps)."""
        if tau <= 0.0:
            raise ValueError(f"temperature must be positive, got {tau}")
        if tau != self._tau_host:
            self.tau_dev.fill_(tau)
            self._tau_host = tau
        if float(self.lr) != self._lr_host:
            self.lr_dev.fill_(float(self.lr))
            self._lr_host = float(self.lr)

    def _reducer(self):
        """The gradient exchange (ddp.GradReducer): buckets cut at decoder_cnn.fc.weight -- everything from there on
        (decoder CNN, both LSTM stacks) is final at the backward pass's cut, the encoder CNN at its end."""
        if self._red is None:
            from .ddp import GradReducer
            self._red = GradReducer(self.gflat, self.model._layout.offsets["decoder_cnn.fc.weight"], self.pg)
        return self._red

    def _allreduce(self):
        if self.world > 1:
            self._reducer().reduce_all()

    def _capture(self, x, U, tau, B, T):
        # two eager warm-up steps on a side stream (allocator + lazy kernel attributes), then capture.
        # The warm-ups are real optimiser steps; their effect on the counters is rolled back.
        flat0, m0, v0, s0 = self.model._flat.clone(), self.m.clone(), self.vv.clone(), self.step_dev.clone()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):
                self._fwd_bwd(x, U, tau, B, T, cut=(lambda: None) if (self.world > 1 and self.ddp_overlap) else None)
                if self.world > 1 and self.ddp_overlap and self.ddp_split_update:
                    self._update("tail")          # (the warm-up also builds the job tables the capture replays)
                    self._update("head")
                else:
                    self._update()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g1, g2 = torch.cuda.CUDAGraph(), None
        if self._pool is None:
            self._pool = torch.cuda.graph_pool_handle()
        pool = self._pool
        if self.world == 1 and self.one_graph:
            # no collective between backward and Adam: the whole step is one graph launch
            with torch.cuda.graph(g1, pool=pool):
                self._fwd_bwd(x, U, tau, B, T)
                self._update()
        elif self.world > 1 and self.ddp_overlap and self.ddp_ingraph:
            # ONE graph with the collectives inside: the tail bucket's all-reduce forks onto a communication stream at the
            # cut and runs beside the encoder CNN's backward pass, the head's follows it there behind the last reduction,
            # the update(s) wait for their bucket
            red = self._reducer()
            red.reduce_all()                                   # communicator up before the capture (gflat is scratch here)
            torch.cuda.synchronize()
            cs, comm = torch.cuda.Stream(), torch.cuda.Stream()
            cs.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(cs):
                g1.capture_begin(pool=pool)
                ev_tail = torch.cuda.Event()

                def cut():
                    comm.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(comm):
                        red.reduce_tail()
                        ev_tail.record(comm)

                self._fwd_bwd(x, U, tau, B, T, cut=cut)
                comm.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(comm):
                    red.reduce_head()
                if self.ddp_split_update:
                    torch.cuda.current_stream().wait_event(ev_tail)
                    self._update("tail")
                    torch.cuda.current_stream().wait_stream(comm)
                    self._update("head")
                else:
                    torch.cuda.current_stream().wait_stream(comm)
                    self._update()
                g1.capture_end()
            torch.cuda.current_stream().wait_stream(cs)
            g2 = None
        elif self.world > 1 and self.ddp_overlap:
            # three graphs: forward + backward up to the cut | the encoder CNN's backward | Adam
            g2, g3 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            cs = torch.cuda.Stream()
            cs.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(cs):
                g1.capture_begin(pool=pool)

                def cut():
                    g1.capture_end()
                    g2.capture_begin(pool=pool)

                self._fwd_bwd(x, U, tau, B, T, cut=cut)
                g2.capture_end()
                g3.capture_begin(pool=pool)
                g4 = None
                if self.ddp_split_update:
                    # the update in the two gradient buckets: the tail's runs while the head is still being all-reduced
                    self._update("tail")
                    g3.capture_end()
                    g4 = torch.cuda.CUDAGraph()
                    g4.capture_begin(pool=pool)
                    self._update("head")
                    g4.capture_end()
                else:
                    self._update()
                    g3.capture_end()
            torch.cuda.current_stream().wait_stream(cs)
            self.model._flat.copy_(flat0)
            self.m.copy_(m0)
            self.vv.copy_(v0)
            self.step_dev.copy_(s0)
            self.eng.pack(self.model._flat)
            torch.cuda.synchronize()
            return (g1, g2, g3) if g4 is None else (g1, g2, g3, g4)
        else:
            g2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g1, pool=pool):
                self._fwd_bwd(x, U, tau, B, T)
            with torch.cuda.graph(g2, pool=pool):
                self._update()
        self.model._flat.copy_(flat0)
        self.m.copy_(m0)
        self.vv.copy_(v0)
        self.step_dev.copy_(s0)
        self.eng.pack(self.model._flat)
        torch.cuda.synchronize()
        return g1, g2
