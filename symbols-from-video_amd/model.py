"""Drop-in `Seq2SeqBinaryVAE` over the HIP engine.

Mirrors the reference modules' surface (constructor arguments, forward / encode
signatures and return values, state_dict keys and shapes, train()/eval(), .to()):
  models/percep_RBVAE/percep_RBVAE_model.py:125-191        variant="percep"
  models/contrastive_RBVAE/contrastive_RBVAE_model.py:124-190  variant="contrastive"
  models/triplet_RBVAE/triplet_RBVAE_model.py:126-193      variant="triplet" (no noise_ratio argument)
  models/simple_RBVAE/simple_RBVAE_model.py:151-193        variant="simple"  (returns x_recon, logits)

Differences a caller can see, all additive:
  * `variant`, `input_hw` (the reference hard-codes the frame size into its fc
    layers; the default here is the same size, so checkpoints load unchanged) and
    `compute_dtype` ("f32" exact-parity mode, "bf16" MFMA mode) constructor keywords;
  * forward/encode accept `u=` (the uniform noise the reference draws with
    torch.rand on the host) and `dropout_masks=` for reproducible runs.
There is no PyTorch fallback: on a box without the GPU library the forward raises.
"""
from __future__ import annotations

import math
from typing import Optional, Sequence

import torch
import torch.nn as nn

from . import _lib as L
from .engine import VARIANTS, Engine, ParamLayout


def binary_concrete_logits(logits, temperature=0.5, hard=False, eps=1e-8, noise_ratio=0.1, u=None):
    """Binary-Concrete relaxation (percep_RBVAE_model.py:17-44) on the HIP kernel.

    Like the reference, the uniform noise comes from torch.rand on the host (CPU default
    generator) unless `u` is given."""
    return _BinarizeFn.apply(logits, temperature, hard, eps, noise_ratio, u)


class _BinarizeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, temperature, hard, eps, noise_ratio, u):
        if not logits.is_cuda:
            raise RuntimeError("binary_concrete_logits: the HIP path needs a CUDA/ROCm tensor (no CPU fallback)")
        shape = logits.shape
        h = logits.detach().reshape(-1, shape[-1]).float().contiguous()
        if u is None:
            u = torch.rand(shape).to(logits.device)       # same draw as the reference (:33)
        u = u.reshape(h.shape).float().contiguous()
        y = torch.empty_like(h)
        z = torch.empty_like(h)
        L.call("rbvae_binarize_kl_fwd", h, u, y, z, None, h.shape[0], h.shape[1], float(temperature),
               float(noise_ratio), float(eps), int(bool(hard)), 0.5, 1e-8, 1, 0, None)
        ctx.save_for_backward(y, z)
        ctx.tau = float(temperature)
        return z.reshape(shape)

    @staticmethod
    def backward(ctx, g):
        y, z = ctx.saved_tensors
        gz = g.reshape(y.shape).float().contiguous()
        dh = torch.empty_like(y)
        L.call("rbvae_binarize_kl_bwd", gz, y, z, dh, 0, y.shape[0], y.shape[1], ctx.tau, None, 0.0, None, 0.5, 1e-8, 1)
        return dh.reshape(g.shape), None, None, None, None, None


def _mask_to_rows(m: torch.Tensor) -> torch.Tensor:
    """[N,C,H,W] keep-mask (torch layout) -> u8 NHWC rows [N*H*W, C] for the kernels."""
    return m.permute(0, 2, 3, 1).contiguous().to(torch.uint8).reshape(-1, m.shape[1])


class _ForwardFn(torch.autograd.Function):
    """The whole forward as one autograd node; backward is the engine's hand-scheduled pass."""

    @staticmethod
    def forward(ctx, model, x, u, temperature, hard, noise_ratio, masks, need, *params):
        eng = model._engine_for(x)
        train = model.training
        model._pack()
        out = eng.forward(model._flat, x, u, temperature, hard, noise_ratio, train, masks,
                          seed=model._next_seed(), need_grad=need)
        ctx.model, ctx.eng, ctx.sv = model, eng, out["saved"]
        ctx.simple = eng.v.simple_order
        # Where the first conv's weight gradient is rebuilt from the frames themselves (rbvae_wgrad_first: no im2col rows in
        # HBM), backward re-reads x: an in-place write to x in between must fail like it does in stock torch
        ctx.x_ref, ctx.x_ver = (x, x._version) if (need and out["saved"].col1 is None) else (None, None)
        if ctx.simple:
            return out["xr"], out["e"].view(x.shape[0] * x.shape[1], -1)
        return out["xr"], out["hs"], out["z"]

    @staticmethod
    def backward(ctx, g_xr, g_b, g_c=None):
        model, eng, sv = ctx.model, ctx.eng, ctx.sv
        if sv.acts_enc is None:
            raise RuntimeError("backward through a forward that ran without gradient tracking")
        if ctx.x_ref is not None and ctx.x_ref._version != ctx.x_ver:
            raise RuntimeError("one of the variables needed for gradient computation has been modified by an inplace operation: "
                               f"the input frames are at version {ctx.x_ref._version}; expected version {ctx.x_ver} instead "
                               "(the first convolution's weight gradient re-reads them)")
        gflat = model._gflat_ws()
        if g_xr is None:
            g_xr = torch.zeros_like(sv.xr)
        if ctx.simple:
            eng.backward(model._flat, gflat, sv, g_xr.float(), None, None, g_e=None if g_b is None else g_b.float())
        else:
            eng.backward(model._flat, gflat, sv, g_xr.float(), None if g_b is None else g_b.float(),
                         None if g_c is None else g_c.float())
        grads = tuple(eng.layout.view(gflat, n).clone() for n in eng.layout.names)
        return (None, None, None, None, None, None, None, None) + grads


class Seq2SeqBinaryVAE(nn.Module):
    def __init__(self, in_channels=3, out_channels=3, latent_dim=32, hidden_dim=32, variant="percep",
                 input_hw=None, compute_dtype="f32"):
        super().__init__()
        if variant not in VARIANTS:
            raise ValueError(f"variant must be one of {sorted(VARIANTS)}")
        self.variant = variant
        self.latent_dim = latent_dim
        self.in_channels, self.out_channels = in_channels, out_channels
        self.input_hw = tuple(input_hw) if input_hw is not None else VARIANTS[variant].default_hw
        self.compute_dtype = compute_dtype
        # hidden_dim is accepted and ignored exactly like the reference (percep_RBVAE_model.py:140-141);
        # the simple variant wires hidden_dim into its LSTMs, which only works for hidden_dim == latent_dim
        # downstream of the decoder fc, so it is required to match there.
        if variant == "simple" and hidden_dim != latent_dim:
            raise ValueError("simple variant: hidden_dim must equal latent_dim")
        self._layout = ParamLayout(VARIANTS[variant], in_channels, out_channels, latent_dim, self.input_hw)
        self._names = list(self._layout.names)
        self._pnames = [n.replace(".", "__") for n in self._names]
        init = self._default_init()
        self._flat = torch.zeros(self._layout.total)
        for n, pn in zip(self._names, self._pnames):
            view = self._layout.view(self._flat, n)
            view.copy_(init[n])
            self.register_parameter(pn, nn.Parameter(view))
        self._engines = {}
        self._packed_version = None
        self._seed = 0
        self._register_state_dict_hook(self._sd_hook)
        self._register_load_state_dict_pre_hook(self._load_hook)

    # ---- parameters ----------------------------------------------------------------
    def _default_init(self):
        """torch's default initialisers, drawn in the reference's construction order
        (encoder_cnn, decoder_cnn, encoder_rnn, decoder_rnn) so that a given
        torch.manual_seed yields the reference's initial weights bit for bit."""
        v = VARIANTS[self.variant]
        c1, c2, c3 = v.channels
        k = v.kernel
        bh, bw = self._layout.bott
        flat = c3 * bh * bw
        op = 1 if k == 3 else 0
        i0, i1, i2 = self._layout.conv_idx
        mods = {}
        for i, (ci, co) in zip((i0, i1, i2), [(self.in_channels, c1), (c1, c2), (c2, c3)]):
            mods[f"encoder_cnn.conv.{i}"] = nn.Conv2d(ci, co, k, 2, 1)
        mods["encoder_cnn.fc"] = nn.Linear(flat, self.latent_dim)
        mods["decoder_cnn.fc"] = nn.Linear(self.latent_dim, flat)
        for i, (ci, co) in zip((i0, i1, i2), [(c3, c2), (c2, c1), (c1, self.out_channels)]):
            mods[f"decoder_cnn.deconv.{i}"] = nn.ConvTranspose2d(ci, co, k, 2, 1, output_padding=op)
        mods["encoder_rnn.lstm"] = nn.LSTM(self.latent_dim, self.latent_dim, v.lstm_layers, batch_first=True)
        mods["decoder_rnn.lstm"] = nn.LSTM(self.latent_dim, self.latent_dim, v.lstm_layers, batch_first=True)
        out = {}
        for prefix, m in mods.items():
            for name, p in m.named_parameters():
                out[f"{prefix}.{name}"] = p.detach()
        return out

    def _reflatten(self):
        """Re-establish 'every parameter is a view of one flat f32 buffer' after .to()/.cuda()."""
        ref = getattr(self, self._pnames[0])
        flat = torch.zeros(self._layout.total, dtype=torch.float32, device=ref.device)
        for n, pn in zip(self._names, self._pnames):
            p = getattr(self, pn)
            view = self._layout.view(flat, n)
            view.copy_(p.data.float())
            p.data = view
            if p.grad is not None:
                p.grad = None
        self._flat = flat
        self._engines = {}
        self._packed_version = None

    def _apply(self, fn, recurse=True):
        super()._apply(fn, recurse)
        self._reflatten()
        return self

    def _sd_hook(self, module, state_dict, prefix, local_metadata):
        for n, pn in zip(self._names, self._pnames):
            key = prefix + pn
            if key in state_dict:
                state_dict[prefix + n] = state_dict.pop(key)

    def _load_hook(self, state_dict, prefix, *args):
        for n, pn in zip(self._names, self._pnames):
            key = prefix + n
            if key in state_dict:
                state_dict[prefix + pn] = state_dict.pop(key)

    def _engine_for(self, x):
        if not x.is_cuda:
            raise RuntimeError("Seq2SeqBinaryVAE: the HIP path needs CUDA/ROCm tensors (there is no CPU fallback); "
                               "move the model and the input to the GPU")
        if self._flat.device != x.device:
            raise RuntimeError(f"model parameters are on {self._flat.device}, input on {x.device}")
        key = (x.device.index, self.compute_dtype)
        eng = self._engines.get(key)
        if eng is None:
            eng = Engine(self.variant, self.in_channels, self.out_channels, self.latent_dim, self.input_hw,
                         self.compute_dtype, x.device)
            self._engines[key] = eng
            self._packed_version = None
        return eng

    def _pack(self):
        """Refresh the packed weight copies when any parameter changed (optimizer step, load_state_dict)."""
        ver = tuple(getattr(self, pn)._version for pn in self._pnames)
        for eng in self._engines.values():
            if self._packed_version != (id(eng), ver):
                eng.pack(self._flat)
                self._packed_version = (id(eng), ver)

    def _gflat_ws(self):
        """Persistent flat gradient workspace (stable address: the engine's job tables point into it)."""
        g = getattr(self, "_gflat", None)
        if g is None or g.device != self._flat.device or g.numel() != self._flat.numel():
            g = torch.zeros_like(self._flat)
            self._gflat = g
        return g

    def _next_seed(self):
        self._seed += 1
        return self._seed

    def _params(self):
        return [getattr(self, pn) for pn in self._pnames]

    # ---- reference surface ------------------------------------------------------------
    def _check(self, x):
        if x.dim() != 5:
            raise ValueError(f"expected x of shape [B, T, C, H, W], got {tuple(x.shape)}")

    def _noise(self, x, u):
        B, T = x.shape[0], x.shape[1]
        if u is None:
            u = torch.rand((B * T, self.latent_dim)).to(x.device)        # host draw, as the reference (:33)
        return u.reshape(B * T, self.latent_dim).float().contiguous()

    def forward(self, x, temperature=1.0, hard=False, noise_ratio=0.1, u=None, dropout_masks=None):
        """percep_RBVAE_model.py:143-170 -> (x_recon, h_seq, z_seq); simple -> (x_recon, logits)."""
        self._check(x)
        masks = None
        if dropout_masks is not None and self.training:
            masks = [_mask_to_rows(m.to(x.device)) for m in dropout_masks]
        params = self._params()
        need = torch.is_grad_enabled() and any(p.requires_grad for p in params)
        return _ForwardFn.apply(self, x.float(), self._noise(x, u), float(temperature), bool(hard),
                                float(noise_ratio), masks, need, *params)

    def encode(self, x, temperature=0.5, hard=False, noise_ratio=0.1, u=None, dropout_masks=None):
        """percep_RBVAE_model.py:172-191 -> z_seq [B,T,L] (no gradient: the callers are eval loops).
        Like the reference, the encoder CNN runs in the module's current mode: after .train() its two Dropout
        layers are live (dropout_masks: optional explicit keep-masks for those two sites)."""
        self._check(x)
        if VARIANTS[self.variant].simple_order:
            raise AttributeError("the simple variant has no encode() (simple_RBVAE_model.py)")
        eng = self._engine_for(x)
        self._pack()
        masks = None
        if dropout_masks is not None and self.training:
            masks = [_mask_to_rows(m.to(x.device)) for m in dropout_masks]
        with torch.no_grad():
            out = eng.forward(self._flat, x.float(), self._noise(x, u), float(temperature), bool(hard),
                              float(noise_ratio), bool(self.training), masks, seed=self._next_seed(),
                              need_grad=False, encode_only=True)
        return out["z"].clone()
