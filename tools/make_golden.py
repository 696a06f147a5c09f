#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE on CPU in the dev container.

Run from the repo root:  python tools/make_golden.py
Needs /root/reference (read-only).  Nothing from the reference is copied: the
fixtures hold inputs, uniform noise, dropout keep-masks, outputs, losses, grads
and post-Adam parameters only.  Weights are NOT stored: every case records the
seed, and the parameters are re-drawn with oracle.init_params(seed) (torch's
own default initialisers in the reference's registration order); a checksum of
each tensor is stored so a drift in torch's RNG would be caught.

Import recipe: torchvision and torch.utils.tensorboard are not
installed and are unused by the model code, so they are stubbed; bytecode
writing is disabled so /root/reference stays pristine.
"""
import importlib.util
import os
import sys
import types

sys.dont_write_bytecode = True
os.environ["PYTHONDONTWRITEBYTECODE"] = "1"

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import rbvae_oracle as O  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
META = {"torch": torch.__version__, "numpy": np.__version__}


def _stub_imports():
    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")

    class _Any:
        def __init__(self, *a, **k):
            pass

        def __call__(self, x):
            return x
    tvt.Compose = tvt.Resize = tvt.ToTensor = _Any
    tv.transforms = tvt
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.transforms"] = tvt
    tb = types.ModuleType("torch.utils.tensorboard")

    class SummaryWriter:
        def __init__(self, *a, **k):
            pass

        def add_scalar(self, *a, **k):
            pass

        def close(self):
            pass
    tb.SummaryWriter = SummaryWriter
    sys.modules["torch.utils.tensorboard"] = tb


def load_ref(relpath, name):
    d = os.path.dirname(os.path.join(REF, relpath))
    if d not in sys.path:
        sys.path.insert(0, d)
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, relpath))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


class RecordRand:
    """Record every torch.rand(...) the reference draws while active."""

    def __enter__(self):
        self.draws = []
        self._orig = torch.rand

        def rec(*a, **k):
            out = self._orig(*a, **k)
            self.draws.append(out.clone())
            return out
        torch.rand = rec
        return self

    def __exit__(self, *e):
        torch.rand = self._orig


def checksums(p):
    return {k: np.array([float(v.double().sum()), float(v.double().abs().sum())]) for k, v in p.items()}


def resize_ref_model(model, variant, L, hw):
    """Let the reference's own forward run at another spatial size: swap the two
    fc layers (whose sizes the reference hard-codes) and the one hard-coded
    reshape in its decoder."""
    v = O.VARIANTS[variant]
    bh, bw = O.bottleneck_hw(v, hw)
    c3 = v.channels[2]
    model.encoder_cnn.fc = nn.Linear(c3 * bh * bw, L)
    model.decoder_cnn.fc = nn.Linear(L, c3 * bh * bw)
    dec = model.decoder_cnn

    def fwd(z):
        return dec.deconv(dec.fc(z).reshape(z.size(0), c3, bh, bw))
    dec.forward = fwd


def hook_dropout_masks(model):
    masks = []

    def hook(mod, inp, out):
        if mod.training:
            masks.append((out != 0) | (inp[0] == 0))   # keep-mask (ambiguous where input is 0: irrelevant)
    hs = [m.register_forward_hook(hook) for m in model.modules() if isinstance(m, nn.Dropout)]
    return masks, hs


def sample_idx(n, stride=97):
    return np.arange(0, n, stride)


def pack_grads(prefix, named, out, full):
    for k, g in named.items():
        g = g.detach().reshape(-1)
        out[f"{prefix}norm/{k}"] = np.float64(g.double().norm().item())
        if full:
            out[f"{prefix}full/{k}"] = g.numpy().copy()
        else:
            out[f"{prefix}samp/{k}"] = g[sample_idx(g.numel())].numpy().copy()


def model_case(name, variant, ref_mod, train_mod, in_ch, L, B, T, hw, seed, tau, r, bern_p,
               alpha, beta, margin, full_grads, train_mode=False, hard=False, x_dist="rand",
               adam=False):
    torch.manual_seed(seed)
    params = O.init_params(variant, in_ch, in_ch, L, hw)
    model = ref_mod.Seq2SeqBinaryVAE(in_channels=in_ch, out_channels=in_ch, latent_dim=L, hidden_dim=L)
    if tuple(hw) != O.VARIANTS[variant].default_hw:
        resize_ref_model(model, variant, L, hw)
    model.load_state_dict(params, strict=True)
    g = torch.Generator().manual_seed(seed + 1)
    shape = (B, 2, T, in_ch, hw[0], hw[1])
    item = torch.rand(shape, generator=g) if x_dist == "rand" else torch.randn(shape, generator=g)
    model.train(train_mode)
    masks, hooks = hook_dropout_masks(model)
    torch.manual_seed(seed + 2)
    recons, kls, hs, outs = [], [], [], []
    with RecordRand() as rr:
        for vw in range(2):
            frame = item[:, vw]
            if variant == "triplet":
                xr, h, z = model(frame, temperature=tau, hard=hard)
            else:
                xr, h, z = model(frame, temperature=tau, hard=hard, noise_ratio=r)
            recons.append(train_mod.recon_loss(xr, frame))
            kls.append(train_mod.kl_binary_concrete(z, p=bern_p))
            hs.append(h)
            outs.append((xr, h, z))
    U = [d for d in rr.draws if d.shape == (B * T, L)]
    assert len(U) == 2
    recon = sum(recons) / 2
    kl = sum(kls) / 2
    if variant == "triplet":
        pair = 0
        for s in range(T - 1):
            pair = pair + train_mod.triplet_loss(hs[0][:, s], hs[1][:, s], hs[0][:, s + 1],
                                                 margin=margin, p=2.0, swap=True)
        pair = pair / float(T - 1)
    else:
        pair = train_mod.contrast_loss(hs[0], hs[1], label=0)
        dis = 0
        for s in range(T - 1):
            dis = dis + train_mod.contrast_loss(hs[0][:, s], hs[0][:, s + 1], label=1)
        pair = pair + dis / float(T - 1)
    total = recon + beta * kl + alpha * pair
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    opt.zero_grad()
    total.backward()
    out = {
        "meta/variant": variant, "meta/seed": seed, "meta/in_ch": in_ch, "meta/L": L,
        "meta/hw": np.array(hw), "meta/tau": tau, "meta/noise_ratio": r, "meta/bern_p": bern_p,
        "meta/alpha": alpha, "meta/beta": beta, "meta/margin": margin,
        "meta/train_mode": train_mode, "meta/hard": hard, "meta/torch": META["torch"],
        "item": item.numpy(), "U0": U[0].numpy(), "U1": U[1].numpy(),
        "loss/total": np.float64(total.item()), "loss/recon": np.float64(recon.item()),
        "loss/kl": np.float64(kl.item()), "loss/pair": np.float64(pair.item()),
    }
    for k, cs in checksums(params).items():
        out[f"paramsum/{k}"] = cs
    for vw, (xr, h, z) in enumerate(outs):
        out[f"h{vw}"] = h.detach().numpy()
        out[f"z{vw}"] = z.detach().numpy()
        xr = xr.detach()
        if xr.numel() <= 200_000:
            out[f"xr{vw}"] = xr.numpy()
        else:
            out[f"xr{vw}_samp"] = xr.reshape(-1)[sample_idx(xr.numel(), 389)].numpy().copy()
            out[f"xr{vw}_sum"] = np.float64(xr.double().sum().item())
    if train_mode:
        per_view = len(masks) // 2
        for vw in range(2):
            for j in range(per_view):
                out[f"mask{vw}_{j}"] = np.packbits(masks[vw * per_view + j].numpy().reshape(-1))
                out[f"maskshape{vw}_{j}"] = np.array(masks[vw * per_view + j].shape)
    grads = {k: p.grad for k, p in model.named_parameters()}
    pack_grads("grad", grads, out, full_grads)
    if adam:
        opt.step()
        pack_grads("adam", dict(model.named_parameters()), out, full_grads)
    for h in hooks:
        h.remove()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(f"{name}: total={total.item():.6f} recon={recon.item():.6f} kl={kl.item():.6f} pair={pair.item():.6f}")


def simple_case(ref_mod, train_mod):
    seed, L = 21, 16
    torch.manual_seed(seed)
    params = O.init_params("simple", 3, 3, L, (64, 64))
    model = ref_mod.Seq2SeqBinaryVAE(in_channels=3, out_channels=3, latent_dim=L, hidden_dim=L)
    model.load_state_dict(params, strict=True)
    g = torch.Generator().manual_seed(seed + 1)
    x = torch.rand((8, 1, 3, 64, 64), generator=g)
    torch.manual_seed(seed + 2)
    with RecordRand() as rr:
        xr, logits = model(x, temperature=0.5, hard=False)
    U = rr.draws[0]
    recon = train_mod.recon_loss(xr, x)
    kl = train_mod.kl_binary_concrete(logits, p=0.1)
    total = recon + 0.1 * kl
    total.backward()
    out = {"meta/seed": seed, "meta/L": L, "x": x.numpy(), "U": U.numpy(),
           "logits": logits.detach().numpy(),
           "xr_samp": xr.detach().reshape(-1)[sample_idx(xr.numel(), 389)].numpy().copy(),
           "xr_sum": np.float64(xr.detach().double().sum().item()),
           "loss/total": np.float64(total.item()), "loss/recon": np.float64(recon.item()),
           "loss/kl": np.float64(kl.item())}
    for k, cs in checksums(params).items():
        out[f"paramsum/{k}"] = cs
    pack_grads("grad", {k: p.grad for k, p in model.named_parameters()}, out, False)
    np.savez_compressed(os.path.join(OUT, "simple_cfg1.npz"), **out)
    print(f"simple_cfg1: total={total.item():.6f}")


def init_parity_case(variant, ref_mod, in_ch, L, seed):
    """Default-size construction under one seed: reference vs oracle.init_params."""
    torch.manual_seed(seed)
    ref = ref_mod.Seq2SeqBinaryVAE(in_channels=in_ch, out_channels=in_ch, latent_dim=L, hidden_dim=L)
    mine = O.init_params(variant, in_ch, in_ch, L, seed=seed)
    sd = ref.state_dict()
    assert list(sd.keys()) == list(mine.keys()), variant
    for k in sd:
        assert torch.equal(sd[k], mine[k]), (variant, k)
    out = {"meta/seed": seed, "meta/in_ch": in_ch, "meta/L": L}
    for k, cs in checksums(sd).items():
        out[f"paramsum/{k}"] = cs
        out[f"head/{k}"] = sd[k].reshape(-1)[:8].numpy().copy()
    np.savez_compressed(os.path.join(OUT, f"init_{variant}.npz"), **out)
    print(f"init_{variant}: bit-identical to the reference constructor ({len(sd)} tensors)")


def function_cases(pm, pt, tm, tt, sm, st):
    g = torch.Generator().manual_seed(5)
    out = {}
    # G1 binarise: percep (noise_ratio), triplet (no ratio), simple (eps 1e-10)
    i = 0
    for L in (25, 32):
        logits = torch.randn((12, L), generator=g) * 2
        for tau in (1.0, 0.5, 0.2):
            for r in (0.1, 0.3):
                for hard in (False, True):
                    torch.manual_seed(100 + i)
                    with RecordRand() as rr:
                        y = pm.binary_concrete_logits(logits, temperature=tau, hard=hard, noise_ratio=r)
                    out[f"bin/{i}/logits"] = logits.numpy()
                    out[f"bin/{i}/U"] = rr.draws[0].numpy()
                    out[f"bin/{i}/y"] = y.numpy()
                    out[f"bin/{i}/cfg"] = np.array([tau, r, float(hard), 1e-8])
                    i += 1
    for mod, eps in ((tm, 1e-8), (sm, 1e-10)):
        logits = torch.randn((12, 16), generator=g) * 2
        for hard in (False, True):
            torch.manual_seed(100 + i)
            with RecordRand() as rr:
                y = mod.binary_concrete_logits(logits, temperature=0.7, hard=hard)
            out[f"bin/{i}/logits"] = logits.numpy()
            out[f"bin/{i}/U"] = rr.draws[0].numpy()
            out[f"bin/{i}/y"] = y.numpy()
            out[f"bin/{i}/cfg"] = np.array([0.7, 1.0, float(hard), eps])
            i += 1
    out["bin/count"] = i
    # G2 losses (+ their input gradients)
    z = torch.rand((4, 5, 25), generator=g).requires_grad_()
    for p in (0.1, 0.5):
        v = pt.kl_binary_concrete(z, p=p)
        (gz,) = torch.autograd.grad(v, z)
        out[f"kl/p{p}/val"] = np.float64(v.item())
        out[f"kl/p{p}/grad"] = gz.numpy()
    out["kl/z"] = z.detach().numpy()
    lg = (torch.randn((6, 16), generator=g) * 3).requires_grad_()
    v = st.kl_binary_concrete(lg, p=0.1)
    (gl,) = torch.autograd.grad(v, lg)
    out["kl_simple/logits"] = lg.detach().numpy()
    out["kl_simple/val"] = np.float64(v.item())
    out["kl_simple/grad"] = gl.numpy()
    a = (torch.randn((4, 5, 25), generator=g) * 0.3).requires_grad_()
    b = (torch.randn((4, 5, 25), generator=g) * 0.3).requires_grad_()
    for label in (0, 1):
        v = pt.contrast_loss(a, b, label=label)
        ga, gb = torch.autograd.grad(v, (a, b))
        out[f"contrast/l{label}/val"] = np.float64(v.item())
        out[f"contrast/l{label}/ga"] = ga.numpy()
        out[f"contrast/l{label}/gb"] = gb.numpy()
    out["contrast/a"] = a.detach().numpy()
    out["contrast/b"] = b.detach().numpy()
    an = (torch.randn((7, 16), generator=g) * 0.3).requires_grad_()
    po = (torch.randn((7, 16), generator=g) * 0.3).requires_grad_()
    ne = (torch.randn((7, 16), generator=g) * 0.3).requires_grad_()
    for m in (0.2, 1.0):
        v = tt.triplet_loss(an, po, ne, margin=m, p=2.0, swap=True)
        gs = torch.autograd.grad(v, (an, po, ne))
        out[f"triplet/m{m}/val"] = np.float64(v.item())
        for nm, gg in zip("apn", gs):
            out[f"triplet/m{m}/g{nm}"] = gg.numpy()
    out["triplet/a"], out["triplet/p"], out["triplet/n"] = (t.detach().numpy() for t in (an, po, ne))
    xr = torch.rand((2, 3, 4, 8, 8), generator=g)
    x = torch.rand((2, 3, 4, 8, 8), generator=g)
    out["recon/xr"], out["recon/x"] = xr.numpy(), x.numpy()
    out["recon/val"] = np.float64(pt.recon_loss(xr, x).item())
    out["l1/val"] = np.float64(pt.l1_loss(lg.detach(), 0.01).item())
    np.savez_compressed(os.path.join(OUT, "functions.npz"), **out)
    print(f"functions: {i} binarise cases + kl/contrast/triplet/recon")


def trainer_cases(pt):
    """G6: trainer-side scalar logic (temperature schedule, labels, split, consistency)."""
    out = {}
    tr = pt.ContrastiveRBVAETrainer(model=None, device="cpu", train_dataloader=None, val_dataloader=None,
                                    optimizer=None, init_temperature=2.0, final_temperature=0.3,
                                    anneal_rate=3e-3, num_steps_to_update=7)
    temps = []
    for s in range(1, 600):
        tr.global_step = s
        temps.append(tr.get_current_temperature())
    out["temp/cfg"] = np.array([2.0, 0.3, 3e-3, 7])
    out["temp/values"] = np.array(temps)
    flags = [40, 90, 91, 200]
    idx = np.arange(0, 260, 3)
    out["label/flags"] = np.array(flags)
    out["label/idx"] = idx
    out["label/out"] = np.array([pt.assign_label(int(i), flags) for i in idx])
    segs = [(0, 37), (45, 100), (110, 111), (120, 180)]
    emb = {f"{i:010d}.jpg": np.full((1, 3, 2, 2), float(i), dtype=np.float32) for i in range(200)}
    for mode in ("train", "val", "test"):
        ds = pt.ShuffledStatePairDataset(emb, segs, test_pct=0.1, val_pct=0.15, mode="train")
        for si in range(len(segs)):
            out[f"split/{si}/train"] = np.array(ds.train_indices_per_state[si], dtype=np.int64)
            out[f"split/{si}/test"] = np.array(ds.test_indices_per_state[si], dtype=np.int64)
            out[f"split/{si}/val"] = np.array(ds.val_indices_per_state[si], dtype=np.int64)
    out["split/segs"] = np.array(segs)
    out["split/pcts"] = np.array([0.1, 0.15])
    import random
    for mode in ("train", "val"):
        random.seed(77)
        ds = pt.ShuffledStatePairDataset(emb, segs, test_pct=0.1, val_pct=0.15, mode=mode)
        for si, pairs in enumerate(ds.pairs_per_state):
            out[f"pairs/{mode}/{si}"] = np.array(pairs, dtype=np.int64).reshape(-1, 2)
        out[f"pairs/{mode}/len"] = len(ds)
        if mode == "train":
            out[f"pairs/{mode}/item3"] = ds[3].numpy()
    np.savez_compressed(os.path.join(OUT, "trainer.npz"), **out)
    print("trainer: temperature/labels/splits")


def main():
    assert os.path.isdir(REF), "run in the dev container (needs /root/reference)"
    os.makedirs(OUT, exist_ok=True)
    _stub_imports()
    pm = load_ref("models/percep_RBVAE/percep_RBVAE_model.py", "percep_RBVAE_model")
    sys.modules["percep_RBVAE_model"] = pm
    pt = load_ref("models/percep_RBVAE/percep_RBVAE_train.py", "percep_RBVAE_train")
    cm = load_ref("models/contrastive_RBVAE/contrastive_RBVAE_model.py", "contrastive_RBVAE_model")
    sys.modules["contrastive_RBVAE_model"] = cm
    ct = load_ref("models/contrastive_RBVAE/contrastive_RBVAE_train.py", "contrastive_RBVAE_train")
    tm = load_ref("models/triplet_RBVAE/triplet_RBVAE_model.py", "triplet_RBVAE_model")
    tt = load_ref("models/triplet_RBVAE/triplet_RBVAE_train.py", "triplet_RBVAE_train")
    sm = load_ref("models/simple_RBVAE/simple_RBVAE_model.py", "simple_RBVAE_model")
    sys.modules["simple_RBVAE_model"] = sm
    st = load_ref("models/simple_RBVAE/simple_RBVAE_train.py", "simple_RBVAE_train")

    function_cases(pm, pt, tm, tt, sm, st)
    trainer_cases(pt)
    for variant, mod, in_ch, L in (("percep", pm, 4, 32), ("contrastive", cm, 3, 32),
                                   ("triplet", tm, 3, 16), ("simple", sm, 3, 16)):
        init_parity_case(variant, mod, in_ch, L, seed=7)

    common = dict(bern_p=0.1, alpha=1.0, beta=1.0, margin=0.2)
    # small shapes, full gradients (64-channel variants keep the files small)
    model_case("contrastive_small_eval", "contrastive", cm, ct, 3, 25, 2, 3, (32, 32), 11, 0.7, 0.1,
               full_grads=True, adam=True, **common)
    model_case("contrastive_small_train", "contrastive", cm, ct, 3, 25, 2, 3, (32, 32), 12, 0.7, 0.3,
               full_grads=True, train_mode=True, **common)
    model_case("contrastive_small_hard", "contrastive", cm, ct, 3, 32, 2, 4, (16, 24), 13, 0.5, 0.1,
               full_grads=True, hard=True, **common)
    model_case("triplet_small_eval", "triplet", tm, tt, 3, 16, 3, 3, (32, 32), 14, 0.9, 1.0,
               full_grads=True, adam=True, **common)
    # percep: 256 channels -> gradient norms + strided samples
    model_case("percep_small_eval", "percep", pm, pt, 4, 32, 2, 3, (16, 24), 15, 0.7, 0.1,
               full_grads=False, x_dist="randn", adam=True, **common)
    model_case("percep_small_train", "percep", pm, pt, 4, 50, 2, 4, (32, 32), 16, 0.7, 0.1,
               full_grads=False, x_dist="randn", train_mode=True, **common)
    model_case("percep_native_eval", "percep", pm, pt, 4, 32, 1, 2, (88, 160), 17, 0.7, 0.1,
               full_grads=False, x_dist="randn", **common)
    model_case("contrastive_native_eval", "contrastive", cm, ct, 3, 32, 1, 2, (256, 256), 18, 0.7, 0.1,
               full_grads=False, **common)
    simple_case(sm, st)
    ldm_case()




def ldm_case():
    """G7: the reference's own LDM Encoder class (random init) at [1,3,64,64] + restated quant_conv/posterior."""
    sdroot = os.path.join(REF, "src", "stable-diffusion")
    if sdroot not in sys.path:
        sys.path.insert(0, sdroot)
    import importlib
    mdl = importlib.import_module("ldm.modules.diffusionmodules.model")
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ldm_oracle as LO
    seed = 31
    cfg = LO.DDCONFIG
    torch.manual_seed(seed)
    enc = mdl.Encoder(ch=cfg["ch"], out_ch=3, ch_mult=cfg["ch_mult"], num_res_blocks=cfg["num_res_blocks"],
                      attn_resolutions=[], dropout=0.0, in_channels=3, resolution=256, z_channels=4, double_z=True)
    quant = nn.Conv2d(8, 8, 1)
    mine = LO.init_params(seed)
    sd = {f"encoder.{k}": v for k, v in enc.state_dict().items()}
    sd.update({f"quant_conv.{k}": v for k, v in quant.state_dict().items()})
    assert list(sd.keys()) == list(mine.keys()), "construction order drifted"
    for k in sd:
        assert torch.equal(sd[k], mine[k]), k
    enc.eval()
    g = torch.Generator().manual_seed(seed + 1)
    x = torch.rand((2, 3, 64, 64), generator=g) * 2 - 1
    eps = torch.randn((2, 4, 8, 8), generator=g)
    with torch.no_grad():
        hs = enc(x)
        moments = quant(hs)
    mean, logvar = torch.chunk(moments, 2, dim=1)
    latent = 0.18215 * (mean + torch.exp(0.5 * torch.clamp(logvar, -30.0, 20.0)) * eps)
    out = {"meta/seed": seed, "x": x.numpy(), "eps": eps.numpy(), "encoder_out": hs.numpy(),
           "moments": moments.numpy(), "latent": latent.numpy()}
    for k, v in sd.items():
        out[f"paramsum/{k}"] = np.array([float(v.double().sum()), float(v.double().abs().sum())])
    np.savez_compressed(os.path.join(OUT, "ldm_encoder.npz"), **out)
    print(f"ldm_encoder: {len(sd)} tensors bit-identical to the reference Encoder; moments {tuple(moments.shape)}")


if __name__ == "__main__":
    if "--ldm-only" in sys.argv:
        assert os.path.isdir(REF), "run in the dev container (needs /root/reference)"
        _stub_imports()
        ldm_case()
    else:
        main()
