"""CPU oracle for the frozen LDM / Stable-Diffusion VAE encoder (TEST INFRASTRUCTURE, not product code).

Plain-torch fp32 restatement of the path the reference runs to turn a frame into a latent
(src/stable-diffusion/get_percep_embeddings.py:101-103):
  Encoder.forward              src/stable-diffusion/ldm/modules/diffusionmodules/model.py:434-459
  ResnetBlock / AttnBlock / Downsample / Normalize / nonlinearity   same file :33-39,60-79,82-141,150-202
  AutoencoderKL.encode         src/stable-diffusion/ldm/models/autoencoder.py:324-328 (quant_conv 1x1)
  DiagonalGaussianDistribution src/stable-diffusion/ldm/modules/distributions/distributions.py:24-37
  get_first_stage_encoding     src/stable-diffusion/ldm/models/diffusion/ddpm.py:542-549 (x 0.18215)
Pinned by tests/golden/ldm_encoder.npz, produced by running the reference's own Encoder class
(random init: the pretrained SD weights are not available offline).  autoencoder.py / ddpm.py do not
import here (pytorch_lightning, taming absent), so encode / sample / scale are restated from their text.
"""
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor

# configs/stable-diffusion/v1-inference.yaml:46-67
DDCONFIG = dict(ch=128, ch_mult=(1, 2, 4, 4), num_res_blocks=2, in_channels=3, z_channels=4, embed_dim=4)
SCALE_FACTOR = 0.18215


def block_plan(cfg=DDCONFIG):
    """[(prefix, kind, cin, cout)] in the reference's construction / execution order."""
    ch, mult = cfg["ch"], cfg["ch_mult"]
    plan = [("encoder.conv_in", "conv3", cfg["in_channels"], ch)]
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
             ("encoder.conv_out", "conv3", block_in, 2 * cfg["z_channels"]),
             ("quant_conv", "conv1", 2 * cfg["z_channels"], 2 * cfg["embed_dim"])]
    return plan


def init_params(seed: Optional[int] = None, cfg=DDCONFIG) -> Dict[str, Tensor]:
    """torch default initialisers in the reference Encoder's construction order (model.py:368-432),
    then quant_conv (autoencoder.py:299) -- bit-identical to constructing the reference under the seed."""
    import torch.nn as nn
    if seed is not None:
        torch.manual_seed(seed)
    out: Dict[str, Tensor] = {}

    def add(prefix, m):
        for n, p in m.named_parameters():
            out[f"{prefix}.{n}"] = p.detach().clone()

    def res(prefix, cin, cout):
        add(f"{prefix}.norm1", nn.GroupNorm(32, cin, eps=1e-6))
        add(f"{prefix}.conv1", nn.Conv2d(cin, cout, 3, 1, 1))
        add(f"{prefix}.norm2", nn.GroupNorm(32, cout, eps=1e-6))
        add(f"{prefix}.conv2", nn.Conv2d(cout, cout, 3, 1, 1))
        if cin != cout:
            add(f"{prefix}.nin_shortcut", nn.Conv2d(cin, cout, 1, 1, 0))

    for prefix, kind, cin, cout in block_plan(cfg):
        if kind == "conv3":
            add(prefix, nn.Conv2d(cin, cout, 3, 1, 1))
        elif kind == "conv1":
            add(prefix, nn.Conv2d(cin, cout, 1))
        elif kind == "down":
            add(prefix, nn.Conv2d(cin, cout, 3, 2, 0))
        elif kind == "norm":
            add(prefix, nn.GroupNorm(32, cin, eps=1e-6))
        elif kind == "res":
            res(prefix, cin, cout)
        elif kind == "attn":
            add(f"{prefix}.norm", nn.GroupNorm(32, cin, eps=1e-6))
            for nm in ("q", "k", "v", "proj_out"):
                add(f"{prefix}.{nm}", nn.Conv2d(cin, cin, 1))
    return out


def _gn(p, prefix, x):
    return F.group_norm(x, 32, p[f"{prefix}.weight"], p[f"{prefix}.bias"], eps=1e-6)


def _swish(x):
    return x * torch.sigmoid(x)


def _res(p, prefix, x, cin, cout):
    h = F.conv2d(_swish(_gn(p, f"{prefix}.norm1", x)), p[f"{prefix}.conv1.weight"], p[f"{prefix}.conv1.bias"], padding=1)
    h = F.conv2d(_swish(_gn(p, f"{prefix}.norm2", h)), p[f"{prefix}.conv2.weight"], p[f"{prefix}.conv2.bias"], padding=1)
    if cin != cout:
        x = F.conv2d(x, p[f"{prefix}.nin_shortcut.weight"], p[f"{prefix}.nin_shortcut.bias"])
    return x + h


def _attn(p, prefix, x):
    h = _gn(p, f"{prefix}.norm", x)
    q = F.conv2d(h, p[f"{prefix}.q.weight"], p[f"{prefix}.q.bias"])
    k = F.conv2d(h, p[f"{prefix}.k.weight"], p[f"{prefix}.k.bias"])
    v = F.conv2d(h, p[f"{prefix}.v.weight"], p[f"{prefix}.v.bias"])
    b, c, hh, ww = q.shape
    w_ = torch.bmm(q.reshape(b, c, hh * ww).permute(0, 2, 1), k.reshape(b, c, hh * ww)) * (int(c) ** (-0.5))
    w_ = torch.softmax(w_, dim=2)
    o = torch.bmm(v.reshape(b, c, hh * ww), w_.permute(0, 2, 1)).reshape(b, c, hh, ww)
    return x + F.conv2d(o, p[f"{prefix}.proj_out.weight"], p[f"{prefix}.proj_out.bias"])


def encoder_moments(p: Dict[str, Tensor], x: Tensor, cfg=DDCONFIG) -> Tensor:
    """x [N,3,H,W] in [-1,1] -> moments [N, 2*embed, H/8, W/8] (Encoder.forward + quant_conv)."""
    h = x
    for prefix, kind, cin, cout in block_plan(cfg):
        if kind == "conv3":
            if prefix.endswith("conv_out"):
                pass
            h = F.conv2d(h, p[f"{prefix}.weight"], p[f"{prefix}.bias"], padding=1)
        elif kind == "conv1":
            h = F.conv2d(h, p[f"{prefix}.weight"], p[f"{prefix}.bias"])
        elif kind == "down":
            h = F.conv2d(F.pad(h, (0, 1, 0, 1)), p[f"{prefix}.weight"], p[f"{prefix}.bias"], stride=2)
        elif kind == "norm":
            h = _swish(_gn(p, prefix, h))
        elif kind == "res":
            h = _res(p, prefix, h, cin, cout)
        elif kind == "attn":
            h = _attn(p, prefix, h)
    return h


def posterior_sample(moments: Tensor, eps: Optional[Tensor], scale: float = SCALE_FACTOR) -> Tensor:
    mean, logvar = torch.chunk(moments, 2, dim=1)
    logvar = torch.clamp(logvar, -30.0, 20.0)
    z = mean if eps is None else mean + torch.exp(0.5 * logvar) * eps
    return scale * z


def encode(p: Dict[str, Tensor], x: Tensor, eps: Optional[Tensor], cfg=DDCONFIG) -> Tensor:
    """frame -> latent exactly as get_percep_embeddings.py:101-103 composes it."""
    return posterior_sample(encoder_moments(p, x, cfg), eps)
