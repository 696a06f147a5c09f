"""Pin the CPU oracle to outputs of the reference itself (tests/golden, made by
tools/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

import rbvae_oracle as O
from _golden import MODEL_CASES, case_masks, case_params, load, sample_idx


def test_binarize_bit_exact():
    g = load("functions")
    for i in range(int(g["bin/count"])):
        tau, r, hard, eps = g[f"bin/{i}/cfg"]
        y = O.binarize(torch.from_numpy(g[f"bin/{i}/logits"]), torch.from_numpy(g[f"bin/{i}/U"]),
                       float(tau), bool(hard), float(r), float(eps))
        assert np.array_equal(y.numpy(), g[f"bin/{i}/y"]), i


def test_loss_functions():
    g = load("functions")
    z = torch.from_numpy(g["kl/z"]).requires_grad_()
    for p in (0.1, 0.5):
        v = O.kl_binary_concrete(z, p)
        (gz,) = torch.autograd.grad(v, z)
        assert abs(v.item() - float(g[f"kl/p{p}/val"])) < 1e-6
        np.testing.assert_allclose(gz.numpy(), g[f"kl/p{p}/grad"], atol=1e-7)
    lg = torch.from_numpy(g["kl_simple/logits"]).requires_grad_()
    v = O.kl_binary_concrete(lg, 0.1, eps=1e-10, clamp=False)
    (gl,) = torch.autograd.grad(v, lg)
    assert abs(v.item() - float(g["kl_simple/val"])) < 1e-6
    np.testing.assert_allclose(gl.numpy(), g["kl_simple/grad"], atol=1e-7)
    a = torch.from_numpy(g["contrast/a"]).requires_grad_()
    b = torch.from_numpy(g["contrast/b"]).requires_grad_()
    for label in (0, 1):
        v = O.contrast_loss(a, b, label)
        ga, gb = torch.autograd.grad(v, (a, b))
        assert abs(v.item() - float(g[f"contrast/l{label}/val"])) < 1e-6
        np.testing.assert_allclose(ga.numpy(), g[f"contrast/l{label}/ga"], atol=1e-7)
        np.testing.assert_allclose(gb.numpy(), g[f"contrast/l{label}/gb"], atol=1e-7)
    an, po, ne = (torch.from_numpy(g[f"triplet/{k}"]).requires_grad_() for k in "apn")
    for m in (0.2, 1.0):
        v = O.triplet_loss(an, po, ne, m)
        gs = torch.autograd.grad(v, (an, po, ne))
        assert abs(v.item() - float(g[f"triplet/m{m}/val"])) < 1e-6
        for nm, gg in zip("apn", gs):
            np.testing.assert_allclose(gg.numpy(), g[f"triplet/m{m}/g{nm}"], atol=1e-7)
    v = O.recon_loss(torch.from_numpy(g["recon/xr"]), torch.from_numpy(g["recon/x"]))
    assert abs(v.item() - float(g["recon/val"])) < 1e-7
    assert abs(O.l1_loss(lg.detach(), 0.01).item() - float(g["l1/val"])) < 1e-5


def test_trainer_scalar_logic():
    g = load("trainer")
    init, final, rate, every = g["temp/cfg"]
    cur = float(init)
    got = []
    for s in range(1, 600):
        cur = O.temperature_schedule(s, cur, float(init), float(final), float(rate), int(every))
        got.append(cur)
    np.testing.assert_array_equal(np.array(got), g["temp/values"])
    flags = [int(v) for v in g["label/flags"]]
    assert [O.assign_label(int(i), flags) for i in g["label/idx"]] == list(g["label/out"])


@pytest.mark.parametrize("variant", ["percep", "contrastive", "triplet", "simple"])
def test_init_matches_reference_constructor(variant):
    g = load(f"init_{variant}")
    p = O.init_params(variant, int(g["meta/in_ch"]), int(g["meta/in_ch"]), int(g["meta/L"]),
                      seed=int(g["meta/seed"]))
    for k, v in p.items():
        np.testing.assert_array_equal(v.reshape(-1)[:8].numpy(), g[f"head/{k}"])
        cs = g[f"paramsum/{k}"]
        assert abs(float(v.double().sum()) - cs[0]) <= 1e-9 * max(1.0, cs[1]) and abs(float(v.double().abs().sum()) - cs[1]) <= 1e-9 * cs[1]


@pytest.mark.parametrize("name", MODEL_CASES)
def test_model_case(name):
    g = load(name)
    variant, p = case_params(g)
    for v in p.values():
        v.requires_grad_()
    item = torch.from_numpy(g["item"])
    U = [torch.from_numpy(g["U0"]), torch.from_numpy(g["U1"])]
    train = bool(g["meta/train_mode"])
    masks = [case_masks(g, 0), case_masks(g, 1)] if train else None
    res = O.step_losses(variant, p, item, U, float(g["meta/tau"]), float(g["meta/noise_ratio"]),
                        float(g["meta/bern_p"]), float(g["meta/alpha"]), float(g["meta/beta"]),
                        float(g["meta/margin"]), bool(g["meta/hard"]), train, masks)
    for k in ("total", "recon", "kl", "pair"):
        assert abs(res[k].item() - float(g[f"loss/{k}"])) < 2e-5, (k, res[k].item(), float(g[f"loss/{k}"]))
    # forward outputs per view
    for vw in range(2):
        xr, hs, z = O.forward(variant, p, item[:, vw], U[vw], float(g["meta/tau"]), bool(g["meta/hard"]),
                              float(g["meta/noise_ratio"]), train, None if masks is None else masks[vw])
        np.testing.assert_allclose(hs.detach().numpy(), g[f"h{vw}"], atol=2e-6)
        if bool(g["meta/hard"]):
            assert np.array_equal(z.detach().numpy(), g[f"z{vw}"])
        else:
            np.testing.assert_allclose(z.detach().numpy(), g[f"z{vw}"], atol=2e-6)
        if f"xr{vw}" in g.files:
            np.testing.assert_allclose(xr.detach().numpy(), g[f"xr{vw}"], atol=2e-6)
        else:
            flat = xr.detach().reshape(-1)
            np.testing.assert_allclose(flat[sample_idx(flat.numel(), 389)].numpy(), g[f"xr{vw}_samp"], atol=2e-6)
    res["total"].backward()
    for k, v in p.items():
        gr = v.grad.reshape(-1)
        n_ref = float(g[f"gradnorm/{k}"])
        assert abs(float(gr.double().norm()) - n_ref) <= 1e-4 * max(n_ref, 1e-6), k
        if f"gradfull/{k}" in g.files:
            ref = g[f"gradfull/{k}"]
            err = np.linalg.norm(gr.numpy() - ref) / max(np.linalg.norm(ref), 1e-12)
            assert err < 1e-4, (k, err)
        else:
            ref = g[f"gradsamp/{k}"]
            got = gr[sample_idx(gr.numel())].numpy()
            assert np.linalg.norm(got - ref) <= 1e-4 * max(np.linalg.norm(ref), 1e-9), k
    if f"adamnorm/{next(iter(p))}" in g.files:
        # Adam amplifies last-bit gradient noise where |g| ~ eps, so the optimiser is
        # checked on the reference's own gradients (same elements as the samples).
        for k, v in p.items():
            full = f"adamfull/{k}" in g.files
            w0 = v.detach().reshape(-1).clone()
            if not full:
                w0 = w0[sample_idx(w0.numel())]
            gr = torch.from_numpy(g[f"gradfull/{k}"] if full else g[f"gradsamp/{k}"])
            w = {k: w0.clone()}
            O.adam_step(w, {k: gr}, {}, 1e-3, 1)
            np.testing.assert_allclose(w[k].numpy(), g[f"adamfull/{k}" if full else f"adamsamp/{k}"],
                                       rtol=0, atol=1e-7)


def test_simple_cfg1():
    g = load("simple_cfg1")
    p = O.init_params("simple", 3, 3, int(g["meta/L"]), (64, 64), seed=int(g["meta/seed"]))
    for v in p.values():
        v.requires_grad_()
    x = torch.from_numpy(g["x"])
    res = O.simple_step_loss(p, x, torch.from_numpy(g["U"]))
    for k in ("total", "recon", "kl"):
        assert abs(res[k].item() - float(g[f"loss/{k}"])) < 1e-5
    xr, logits = O.forward("simple", p, x, torch.from_numpy(g["U"]), 0.5, False)
    np.testing.assert_allclose(logits.detach().numpy(), g["logits"], atol=2e-6)
    flat = xr.detach().reshape(-1)
    np.testing.assert_allclose(flat[sample_idx(flat.numel(), 389)].numpy(), g["xr_samp"], atol=2e-6)
    res["total"].backward()
    for k, v in p.items():
        n_ref = float(g[f"gradnorm/{k}"])
        assert abs(float(v.grad.double().norm()) - n_ref) <= 1e-4 * max(n_ref, 1e-6), k


def test_state_consistency_small_table():
    codes = np.array([[0, 1], [0, 1], [1, 1], [1, 0], [1, 0], [0, 0]], dtype=np.float32)
    labels = np.array([0, 0, 0, 1, 1, 2])
    avg, pct = O.state_consistency(codes, labels, 4)
    assert pct == [2 / 3, 1.0, 1.0, 0.0]
    assert abs(avg - (2 + 2 + 1) / 6) < 1e-12


def test_ldm_encoder_oracle_vs_reference_fixture():
    import ldm_oracle as LO
    g = load("ldm_encoder")
    p = LO.init_params(int(g["meta/seed"]))
    for k, v in p.items():
        cs = g[f"paramsum/{k}"]
        assert abs(float(v.double().sum()) - cs[0]) <= 1e-9 * max(1.0, cs[1]), k
    x = torch.from_numpy(g["x"])
    with torch.no_grad():
        m = LO.encoder_moments(p, x)
        lat = LO.posterior_sample(m, torch.from_numpy(g["eps"]))
    np.testing.assert_allclose(m.numpy(), g["moments"], atol=5e-6)
    np.testing.assert_allclose(lat.numpy(), g["latent"], atol=5e-6)
