"""Frozen LDM / Stable-Diffusion VAE encoder on the HIP kernels: frame -> latent, forward only.

This is what the reference runs offline in src/stable-diffusion/get_percep_embeddings.py:101-103
(model.encode_first_stage -> get_first_stage_encoding) to produce the percep_RBVAE inputs, and what
cfg 5 runs on the fly:
  Encoder.forward            src/stable-diffusion/ldm/modules/diffusionmodules/model.py:434-459
  ResnetBlock/AttnBlock/Downsample/Normalize/nonlinearity   same file :33-39,60-79,82-141,150-202
  AutoencoderKL.encode       src/stable-diffusion/ldm/models/autoencoder.py:324-328
  posterior sample x 0.18215 ldm/modules/distributions/distributions.py:24-37, ldm/models/diffusion/ddpm.py:542-549

The ResnetBlock convolutions (3x3, stride 1) run on rbvae_conv3x3_halo (csrc/conv_halo.hip): the input patch of a
16 x 16 pixel tile is staged in LDS once per channel slice and shared by the nine taps, the producer's GroupNorm + swish
is applied while the patch is staged, and the statistics of the NEXT GroupNorm come out of the epilogue -- the
GN -> swish -> conv chain of model.py:121-131 is two launches per block plus a 2 KB statistics merge, and no normalised
copy of an activation is written.  The 1x1 / stride-2 / 3- and 8-channel convolutions and the images narrower than a
tile run on rbvae_gather_gemm (stride-1 and asymmetric-pad stride-2 tap tables; skip connections through the
epilogue's `addend`) with the GroupNorm kernels of csrc/ldm.hip; attention is csrc/attn.hip.  state_dict keys are
the reference's (`encoder.*`, `quant_conv.*`), so a Stable-Diffusion `first_stage_model.*` checkpoint
loads as is; without one the weights are torch's default initialisation (the pretrained weights are not
available offline).
"""
from __future__ import annotations

import ctypes
import math
from typing import Dict, Optional

import torch
import torch.nn as nn

from . import _lib as L

F32, BF16 = 0, 1
DDCONFIG = dict(ch=128, ch_mult=(1, 2, 4, 4), num_res_blocks=2, in_channels=3, z_channels=4, embed_dim=4)
SCALE_FACTOR = 0.18215          # configs/stable-diffusion/v1-inference.yaml:17


def _ru(x, m):
    return (x + m - 1) // m * m


def _plan(cfg):
    ch, mult = cfg["ch"], cfg["ch_mult"]
    plan = [("encoder.conv_in", "conv_in", cfg["in_channels"], ch)]
    in_mult = (1,) + tuple(mult)
    block_in = ch
    for lvl in range(len(mult)):
        block_in, block_out = ch * in_mult[lvl], ch * mult[lvl]
        for b in range(cfg["num_res_blocks"]):
            plan.append((f"encoder.down.{lvl}.block.{b}", "res", block_in, block_out))
            block_in = block_out
        if lvl != len(mult) - 1:
            plan.append((f"encoder.down.{lvl}.downsample.conv", "down", block_in, block_in))
    plan += [("encoder.mid.block_1", "res", block_in, block_in), ("encoder.mid.attn_1", "attn", block_in, block_in),
             ("encoder.mid.block_2", "res", block_in, block_in), ("encoder.norm_out", "norm", block_in, block_in),
             ("encoder.conv_out", "conv_out", block_in, 2 * cfg["z_channels"]),
             ("quant_conv", "quant", 2 * cfg["z_channels"], 2 * cfg["embed_dim"])]
    return plan


def _conv_desc(k, off):
    d = [k * k, 0, 0]
    for kh in range(k):
        for kw in range(k):
            d += [kh * k + kw, kh + off, kw + off]
    return (ctypes.c_int * len(d))(*d)


class LDMEncoder(nn.Module):
    def __init__(self, compute_dtype: str = "bf16", cfg: Optional[dict] = None, conv_impl: str = "halo",
                 use_graph: bool = False):
        """use_graph: the ~100 launches of one encode are captured into a HIP graph per input shape at its second call and
        replayed from then on (the encoder is frozen: same kernels, same arguments).  Off by default: at 256 x 256 and
        512 x 512 frames the encode is bound by its kernels, not by the host (698.7 vs 714.1 frames/s on two boxes, within
        their spread); it pays for small frames / few frames per call, where ~10 us of host time per launch shows."""
        super().__init__()
        if compute_dtype not in ("f32", "bf16"):
            raise ValueError("compute_dtype must be 'f32' or 'bf16'")
        if conv_impl not in ("halo", "gather"):
            raise ValueError("conv_impl must be 'halo' (halo-resident 3x3 kernel + fused GroupNorm) or 'gather'")
        self.conv_impl = conv_impl
        self.use_graph = bool(use_graph)
        self._graphs: Dict[tuple, tuple] = {}
        self._seen: Dict[tuple, int] = {}
        self.cfg = dict(DDCONFIG if cfg is None else cfg)
        self.compute_dtype = compute_dtype
        self.plan = _plan(self.cfg)
        self._names = []
        for prefix, kind, cin, cout in self.plan:          # the reference's construction order (model.py:368-432)
            if kind in ("conv_in", "conv_out"):
                self._add(prefix, nn.Conv2d(cin, cout, 3, 1, 1))
            elif kind == "quant":
                self._add(prefix, nn.Conv2d(cin, cout, 1))
            elif kind == "down":
                self._add(prefix, nn.Conv2d(cin, cout, 3, 2, 0))
            elif kind == "norm":
                self._add(prefix, nn.GroupNorm(32, cin, eps=1e-6))
            elif kind == "res":
                self._add(f"{prefix}.norm1", nn.GroupNorm(32, cin, eps=1e-6))
                self._add(f"{prefix}.conv1", nn.Conv2d(cin, cout, 3, 1, 1))
                self._add(f"{prefix}.norm2", nn.GroupNorm(32, cout, eps=1e-6))
                self._add(f"{prefix}.conv2", nn.Conv2d(cout, cout, 3, 1, 1))
                if cin != cout:
                    self._add(f"{prefix}.nin_shortcut", nn.Conv2d(cin, cout, 1, 1, 0))
            elif kind == "attn":
                self._add(f"{prefix}.norm", nn.GroupNorm(32, cin, eps=1e-6))
                for nm in ("q", "k", "v", "proj_out"):
                    self._add(f"{prefix}.{nm}", nn.Conv2d(cin, cin, 1))
        self._packed = None
        self._register_state_dict_hook(self._sd_hook)
        self._register_load_state_dict_pre_hook(self._load_hook)

    # ---- frozen parameters, reference names -----------------------------------------------------------
    def _add(self, prefix, mod):
        for n, p in mod.named_parameters():
            name = f"{prefix}.{n}"
            self._names.append(name)
            self.register_buffer(name.replace(".", "__"), p.detach().clone())

    def _p(self, name):
        return getattr(self, name.replace(".", "__"))

    def _sd_hook(self, module, state_dict, prefix, local_metadata):
        for n in self._names:
            key = prefix + n.replace(".", "__")
            if key in state_dict:
                state_dict[prefix + n] = state_dict.pop(key)

    def _load_hook(self, state_dict, prefix, *args):
        # accept a Stable-Diffusion checkpoint's `first_stage_model.` keys; ignore its decoder / loss entries
        for k in list(state_dict.keys()):
            kk = k[len(prefix):] if k.startswith(prefix) else k
            if kk.startswith("first_stage_model."):
                kk = kk[len("first_stage_model."):]
            if kk in self._names:
                state_dict[prefix + kk.replace(".", "__")] = state_dict.pop(k)
            elif kk.replace(".", "__") not in [n.replace(".", "__") for n in self._names]:
                state_dict.pop(k)
        self._drop_packed()

    def _apply(self, fn, recurse=True):
        super()._apply(fn, recurse)
        self._drop_packed()
        return self

    def _drop_packed(self):
        """new weights / new device: the packed copies and every captured graph (it holds their addresses) go"""
        self._packed = None
        if getattr(self, "_graphs", None):
            self._graphs.clear()
        if getattr(self, "_seen", None):
            self._seen.clear()

    # ---- packed weights ---------------------------------------------------------------------------------
    def _pack(self, dev):
        dt = F32 if self.compute_dtype == "f32" else BF16
        tdt = torch.float32 if dt == F32 else torch.bfloat16
        ke = 128 // (4 if dt == F32 else 2)
        pk: Dict[str, torch.Tensor] = {}

        def pack3(w, shape_out, dims, strides):
            out = torch.zeros(shape_out, dtype=tdt, device=dev)
            L.call("rbvae_pack3", dt, w.float().contiguous(), out, dims[0], dims[1], dims[2], strides[0], strides[1],
                   strides[2])
            return out

        for name in self._names:
            if not name.endswith(".weight"):
                continue
            w = self._p(name)
            if w.dim() != 4:
                continue
            co, ci, kh, kw = w.shape
            kk = kh * kw
            if name == "encoder.conv_in.weight":
                K = _ru(kk * ci, ke)                             # im2col GEMM: [co][t*ci + c]
                pk[name] = pack3(w, (co, K), (co, ci, kk), (K, 1, ci))
            elif name == "quant_conv.weight":
                K = _ru(ci, ke)                                  # its input rows are padded to one K slice
                pk[name] = pack3(w, (co, K), (co, ci, 1), (K, 1, 0))
            else:
                pk[name] = pack3(w, (co, kk, ci), (co, ci, kk), (kk * ci, 1, ci))   # [co][t][ci]
        for prefix, kind, cin, _ in self.plan:
            if kind == "attn":        # q | k | v as one projection [3C][C] for the fused attention path
                pk[f"{prefix}.qkv.weight"] = torch.cat([pk[f"{prefix}.{n}.weight"] for n in ("q", "k", "v")]).contiguous()
                pk[f"{prefix}.qkv.bias"] = torch.cat([self._p(f"{prefix}.{n}.bias").float() for n in ("q", "k", "v")]).contiguous()
        self._packed = (dev, dt, tdt, ke, pk)
        self._zero = torch.zeros(256, dtype=torch.uint8, device=dev)
        self._d_conv = _conv_desc(3, -1)          # stride 1, pad 1
        self._d_down = _conv_desc(3, 0)           # pad (0,1,0,1) then stride 2, pad 0 (model.py:71-75)
        self._d_one = (ctypes.c_int * 6)(1, 0, 0, 0, 0, 0)

    # ---- forward ----------------------------------------------------------------------------------------
    def _gemm(self, A, W, out, bias, addend, nimg, ih, iw, th, tw, sa, oh, ow, kc, nout, lda, ldo, taps, desc,
              scale=1.0):
        dt = self._packed[1]
        L.call("rbvae_gather_gemm", dt, A, W, out, bias, None, None, addend, self._zero, nimg, ih, iw, th, tw, sa, oh,
               ow, 1, kc, nout, lda, ldo, taps, 1, ctypes.addressof(desc), 0, 0, 0.0, float(scale), 0, None, None)

    def _conv3(self, name, x, N, H, W, cin, cout, addend=None):
        tdt, pk = self._packed[2], self._packed[4]
        out = torch.empty(N * H * W, cout, dtype=tdt, device=x.device)
        self._gemm(x, pk[f"{name}.weight"], out, self._p(f"{name}.bias"), addend, N, H, W, H, W, 1, H, W, cin, cout,
                   x.shape[1], cout, 9, self._d_conv)
        return out

    def _conv1(self, name, x, rows, cin, cout, addend=None):
        tdt, pk = self._packed[2], self._packed[4]
        out = torch.empty(rows, cout, dtype=tdt, device=x.device)
        self._gemm(x, pk[f"{name}.weight"], out, self._p(f"{name}.bias"), addend, rows, 1, 1, 1, 1, 1, 1, 1, cin, cout,
                   x.shape[1], cout, 1, self._d_one)
        return out

    def _gn(self, name, x, N, HW, C, swish=True):
        dt = self._packed[1]
        y = torch.empty_like(x)
        nws = L.query("rbvae_groupnorm_ws_floats", dt, N, HW, C, 32)
        ws = torch.empty(nws, dtype=torch.float32, device=x.device)
        L.call("rbvae_groupnorm_swish_ws", dt, x, y, self._p(f"{name}.weight"), self._p(f"{name}.bias"), ws, nws, N, HW, C,
               x.shape[1], y.shape[1], 32, 1e-6, int(swish))
        return y

    def _gn_stats(self, x, xst, N, H, W, C):
        """per-(image, group) mean | rstd of x: from the producing convolution's per-tile partials xst when there are any,
        else from the statistics kernels"""
        dev = x.device
        if xst is not None:
            return xst                 # ("tiles", partials, tile_h, tile_w)
        dt = self._packed[1]
        nws = L.query("rbvae_groupnorm_ws_floats", dt, N, H * W, C, 32)
        ws = torch.empty(nws, dtype=torch.float32, device=dev)
        L.call("rbvae_groupnorm_stats", dt, x, ws, nws, N, H * W, C, x.shape[1], 32, 1e-6)
        return ("ms", ws)

    def _conv3_gn_halo(self, norm, conv, x, xst, N, H, W, cin, cout, addend=None):
        """conv(swish(GroupNorm(x))) + bias (+ addend) on the halo kernel (model.py:121-131) -> (rows, per-tile GroupNorm(32)
        partials of the result).  The normalisation is applied while the patch is staged when the convolution has one or
        two 128-channel output tiles; with four, every tile would redo it and ONE standalone apply pass costs less
        (measured at 512 -> 512, 128 x 128, 4 images: 306 us fused vs 241 + 30 us)."""
        dt, tdt, pk = self._packed[1], self._packed[2], self._packed[4]
        dev = x.device
        gamma, beta = self._p(f"{norm}.weight"), self._p(f"{norm}.bias")
        fused = cout // 128 <= 2
        sc = sh = None
        st = self._gn_stats(x, xst, N, H, W, cin)
        if fused:
            sc = torch.empty(N, cin, dtype=torch.float32, device=dev)
            sh = torch.empty(N, cin, dtype=torch.float32, device=dev)
            if st[0] == "ms":
                L.call("rbvae_gn_affine", st[1], st[1][N * 32:], gamma, beta, sc, sh, N, cin, 32)
            else:
                L.call("rbvae_gn_finish_tiles", st[1], gamma, beta, sc, sh, None, None, N, H, W, cin, 32, 1e-6, st[2], st[3])
        else:
            if st[0] == "ms":
                mean, rstd = st[1], st[1][N * 32:]
            else:
                ms = torch.empty(2 * N * 32, dtype=torch.float32, device=dev)
                sc0 = torch.empty(2, N, cin, dtype=torch.float32, device=dev)
                mean, rstd = ms, ms[N * 32:]
                L.call("rbvae_gn_finish_tiles", st[1], gamma, beta, sc0[0], sc0[1], mean, rstd, N, H, W, cin, 32, 1e-6, st[2],
                       st[3])
            y = torch.empty_like(x)
            L.call("rbvae_groupnorm_apply", dt, x, y, mean, rstd, gamma, beta, N, H * W, cin, x.shape[1], y.shape[1], 32, 1)
            x = y
        out = torch.empty(N * H * W, cout, dtype=tdt, device=dev)
        ost = torch.empty(L.query("rbvae_conv3x3_halo_stats_floats", N, H, W, cout, cout // 32), dtype=torch.float32, device=dev)
        L.call("rbvae_conv3x3_halo", dt, x, pk[f"{conv}.weight"], out, self._p(f"{conv}.bias"), addend, self._zero, sc, sh, 1,
               ost, cout // 32, N, H, W, H, W, 1, 1, cin, cout, x.shape[1], cout)
        return out, ("tiles", ost, 16, 16)

    def _halo_ok(self, N, H, W, cin, cout):
        dt = self._packed[1]
        # a launch needs enough 256 x 128 tiles to fill the chip; smaller problems keep the 128-row gather tiles
        tiles = N * ((H + 15) // 16) * ((W + 15) // 16) * (cout // 128)
        return (self.conv_impl == "halo" and cout % 32 == 0 and tiles >= self._halo_min_tiles and
                bool(L.query("rbvae_conv3x3_halo_ok", dt, H, W, H, W, cin, cout)))

    _halo_min_tiles = 1

    def _res(self, prefix, x, xst, N, H, W, cin, cout):
        """ResnetBlock (model.py:82-141) -> (output rows, its GroupNorm partial statistics or None)"""
        if self._halo_ok(N, H, W, cin, cout) and self._halo_ok(N, H, W, cout, cout):
            h, hst = self._conv3_gn_halo(f"{prefix}.norm1", f"{prefix}.conv1", x, xst, N, H, W, cin, cout)
            skip = x if cin == cout else self._conv1(f"{prefix}.nin_shortcut", x, N * H * W, cin, cout)
            return self._conv3_gn_halo(f"{prefix}.norm2", f"{prefix}.conv2", h, hst, N, H, W, cout, cout, addend=skip)
        h = self._gn(f"{prefix}.norm1", x, N, H * W, cin)
        h = self._conv3(f"{prefix}.conv1", h, N, H, W, cin, cout)
        h = self._gn(f"{prefix}.norm2", h, N, H * W, cout)
        skip = x if cin == cout else self._conv1(f"{prefix}.nin_shortcut", x, N * H * W, cin, cout)
        return self._conv3(f"{prefix}.conv2", h, N, H, W, cout, cout, addend=skip), None

    def _attn(self, prefix, x, N, H, W, C):
        dt, tdt, ke = self._packed[1], self._packed[2], self._packed[3]
        hw = H * W
        if hw % ke:
            raise ValueError(f"mid-block attention: {hw} tokens must be a multiple of {ke} "
                             f"(frame sides divisible by {8 * int(math.isqrt(ke))})")
        h = self._gn(f"{prefix}.norm", x, N, hw, C, swish=False)
        if L.query("rbvae_attention_ok", dt, hw, C):
            # one fused q|k|v projection, then the batched online-softmax kernel (csrc/attn.hip): no hw x hw scores,
            # no per-image loop, no transposed copy of V
            pk = self._packed[4]
            qkv = torch.empty(N * hw, 3 * C, dtype=tdt, device=x.device)
            self._gemm(h, pk[f"{prefix}.qkv.weight"], qkv, pk[f"{prefix}.qkv.bias"], None, N * hw, 1, 1, 1, 1, 1, 1, 1, C,
                       3 * C, h.shape[1], 3 * C, 1, self._d_one)
            o = torch.empty(N * hw, C, dtype=tdt, device=x.device)
            L.call("rbvae_attention", dt, qkv, qkv[:, C:], qkv[:, 2 * C:], o, N, hw, C, 3 * C, 3 * C, 3 * C, C,
                   float(int(C) ** (-0.5)))
            return self._conv1(f"{prefix}.proj_out", o, N * hw, C, C, addend=x)
        q = self._conv1(f"{prefix}.q", h, N * hw, C, C)
        k = self._conv1(f"{prefix}.k", h, N * hw, C, C)
        v = self._conv1(f"{prefix}.v", h, N * hw, C, C)
        o = torch.empty(N * hw, C, dtype=tdt, device=x.device)
        s = torch.empty(hw, hw, dtype=tdt, device=x.device)
        vt = torch.empty(C, hw, dtype=tdt, device=x.device)
        for n in range(N):                                    # one image at a time (hw x hw scores)
            qn, kn, vn = q[n * hw:(n + 1) * hw], k[n * hw:(n + 1) * hw], v[n * hw:(n + 1) * hw]
            self._gemm(qn, kn, s, None, None, hw, 1, 1, 1, 1, 1, 1, 1, C, hw, C, hw, 1, self._d_one,
                       scale=float(int(C) ** (-0.5)))
            L.call("rbvae_softmax_rows", dt, s, s, hw, hw, hw)
            L.call("rbvae_transpose2d", dt, vn, vt, hw, C, C, hw)
            self._gemm(s, vt, o[n * hw:(n + 1) * hw], None, None, hw, 1, 1, 1, 1, 1, 1, 1, hw, C, hw, C, 1, self._d_one)
        return self._conv1(f"{prefix}.proj_out", o, N * hw, C, C, addend=x)

    @torch.no_grad()
    def moments(self, x: torch.Tensor) -> torch.Tensor:
        """x [N,3,H,W] f32 in [-1,1] -> posterior moments as NHWC rows [N*(H/8)*(W/8)][>=8] (mean | logvar)."""
        m = self._moments(x)
        return m.clone() if self.use_graph else m      # a replayed graph writes the same rows at the next call

    def _weights_version(self) -> int:
        """sum of the frozen buffers' in-place version counters: an in-place weight change (p.data.copy_, a foreign
        optimiser step) moves it, and the packed copies + captured graphs (which hold their addresses) are rebuilt"""
        return sum(self._p(n)._version for n in self._names)

    def _check_weights(self):
        ver = self._weights_version()
        if getattr(self, "_packed_ver", None) != ver:
            if self._packed is not None:
                self._drop_packed()
            self._packed_ver = ver

    def _moments(self, x: torch.Tensor) -> torch.Tensor:
        """moments through the captured graph of this input shape (first call of a shape: eager, which also fills the
        packed weights; second call: capture; then replays).  The rows returned belong to the graph: consume them on
        the current stream before the next call of the same shape."""
        self._check_weights()
        if (not self.use_graph or not x.is_cuda or x.dim() != 4 or x.shape[1] != self.cfg["in_channels"]
                or x.shape[2] % 8 or x.shape[3] % 8 or torch.cuda.is_current_stream_capturing()):
            return self._moments_eager(x)
        key = (tuple(x.shape), x.device)
        g = self._graphs.get(key)
        if g is None:
            self._seen[key] = self._seen.get(key, 0) + 1
            if self._seen[key] < 2:
                return self._moments_eager(x)
            xs = torch.empty(x.shape, dtype=torch.float32, device=x.device)
            xs.copy_(x)
            torch.cuda.current_stream().synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                ms = self._moments_eager(xs)
            g = self._graphs[key] = (graph, xs, ms)
        graph, xs, ms = g
        xs.copy_(x)
        graph.replay()
        return ms

    def _moments_eager(self, x: torch.Tensor) -> torch.Tensor:
        if not x.is_cuda:
            raise RuntimeError("LDMEncoder: the HIP path needs a CUDA/ROCm tensor (there is no CPU fallback)")
        if x.dim() != 4 or x.shape[1] != self.cfg["in_channels"]:
            raise ValueError(f"expected x of shape [N, {self.cfg['in_channels']}, H, W], got {tuple(x.shape)}")
        N, C, H, W = x.shape
        if H % 8 or W % 8:
            raise ValueError("frame sides must be divisible by 8")
        self._check_weights()
        if self._packed is None or self._packed[0] != x.device:
            self._pack(x.device)
        dev, dt, tdt, ke, pk = self._packed
        x = x.float().contiguous()
        h, hst = None, None          # activation rows and (when a halo convolution produced them) their GroupNorm partials
        for prefix, kind, cin, cout in self.plan:
            if kind == "conv_in" and self.conv_impl == "halo" and cout % 32 == 0 and \
                    L.query("rbvae_conv_in_ok", dt, C, H, W, cout, N, cout // 32) and pk[f"{prefix}.weight"].shape[1] == 64:
                # one kernel: patch -> im2col rows in LDS -> MFMA -> bias -> store, and the first GroupNorm's partial
                # statistics out of the epilogue (csrc/conv_in.hip)
                h = torch.empty(N * H * W, cout, dtype=tdt, device=dev)
                ost = torch.empty(L.query("rbvae_conv_in_stats_floats", N, H, W, cout, cout // 32), dtype=torch.float32, device=dev)
                L.call("rbvae_conv_in", dt, x, pk[f"{prefix}.weight"], self._p(f"{prefix}.bias"), self._zero, h, ost, cout // 32,
                       N, C, H, W, cout, cout)
                hst = ("tiles", ost, 8, 16)
                continue
            if kind == "conv_in":
                K = pk[f"{prefix}.weight"].shape[1]
                col = torch.empty(N * H * W, K, dtype=tdt, device=dev)
                L.call("rbvae_im2col", dt, x, C * H * W, H * W, W, 1, N, C, H, W, H, W, 3, 3, 1, 1, K, col)
                h = torch.empty(N * H * W, cout, dtype=tdt, device=dev)
                self._gemm(col, pk[f"{prefix}.weight"], h, self._p(f"{prefix}.bias"), None, N * H * W, 1, 1, 1, 1, 1,
                           1, 1, K, cout, K, cout, 1, self._d_one)
                hst = None
            elif kind == "res":
                h, hst = self._res(prefix, h, hst, N, H, W, cin, cout)
                continue
            elif kind == "down":
                out = torch.empty(N * (H // 2) * (W // 2), cout, dtype=tdt, device=dev)
                self._gemm(h, pk[f"{prefix}.weight"], out, self._p(f"{prefix}.bias"), None, N, H, W, H // 2, W // 2, 2,
                           H // 2, W // 2, cin, cout, cin, cout, 9, self._d_down)
                h, H, W = out, H // 2, W // 2
            elif kind == "attn":
                h = self._attn(prefix, h, N, H, W, cin)
            elif kind == "norm":
                h = self._gn(prefix, h, N, H * W, cin)
            elif kind == "conv_out":
                K = pk["quant_conv.weight"].shape[1]
                out = torch.zeros(N * H * W, K, dtype=tdt, device=dev)     # padded to one K slice for quant_conv
                self._gemm(h, pk[f"{prefix}.weight"], out, self._p(f"{prefix}.bias"), None, N, H, W, H, W, 1, H, W, cin,
                           cout, cin, K, 9, self._d_conv)
                h = out
            elif kind == "quant":
                K = pk["quant_conv.weight"].shape[1]
                out = torch.empty(N * H * W, cout, dtype=tdt, device=dev)
                self._gemm(h, pk[f"{prefix}.weight"], out, self._p(f"{prefix}.bias"), None, N * H * W, 1, 1, 1, 1, 1, 1,
                           1, K, cout, K, cout, 1, self._d_one)
                h = out
            hst = None                # every kind but "res" leaves its statistics to the GroupNorm kernels
        return h

    @torch.no_grad()
    def encode(self, x: torch.Tensor, eps: Optional[torch.Tensor] = None, sample: bool = True,
               out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """frame batch -> latent [N, 4, H/8, W/8] f32 = 0.18215 * posterior sample (get_percep_embeddings.py:101-103).
        eps: the N(0,1) draw (default: torch.randn on the host like distributions.py:36); sample=False = mode.
        out: write the latents into this contiguous f32 tensor of N*4*(H/8)*(W/8) elements (e.g. a slice of
        FusedTrainer.input_buffer(): the on-the-fly pipeline of BASELINE configs[4] never copies a latent)."""
        m = self._moments(x)
        N, _, H, W = x.shape
        Z, hw = self.cfg["embed_dim"], (H // 8) * (W // 8)
        if sample and eps is None:
            eps = torch.randn((N, Z, H // 8, W // 8)).to(x.device)
        if out is not None:
            if out.dtype != torch.float32 or not out.is_contiguous() or out.numel() != N * Z * hw or out.device != x.device:
                raise ValueError(f"out must be a contiguous f32 tensor of {N * Z * hw} elements on {x.device}")
            lat = out.view(N, Z, H // 8, W // 8)
        else:
            lat = torch.empty(N, Z, H // 8, W // 8, dtype=torch.float32, device=x.device)
        L.call("rbvae_posterior_sample", self._packed[1], m, m.shape[1], eps.float().contiguous() if sample else None,
               lat, N, Z, hw, SCALE_FACTOR)
        return lat

    forward = encode
