"""The fused training step against the oracle's step + Adam (dropout off so both see the same network)."""
import numpy as np
import pytest
import torch

import rbvae_oracle as O

pytestmark = pytest.mark.gpu


def _oracle_step(variant, params, item, U, tau, r, p, alpha, beta, margin, steps, lr=1e-3):
    state = {}
    hist = []
    for st in range(1, steps + 1):
        for v in params.values():
            v.grad = None
        res = O.step_losses(variant, params, item, U, tau, r, p, alpha, beta, margin)
        res["total"].backward()
        hist.append({k: float(v) for k, v in res.items()})
        grads = {k: v.grad.clone() for k, v in params.items()}
        with torch.no_grad():
            O.adam_step({k: v for k, v in params.items()}, grads, state, lr, st)
    return hist, grads


@pytest.mark.parametrize("variant,use_graph", [("percep", False), ("percep", True), ("triplet", False)])
def test_fused_step_matches_oracle(variant, use_graph):
    import sfv_amd as sfv
    from importlib import import_module
    FusedTrainer = import_module("symbols-from-video_amd.trainer").FusedTrainer
    in_ch = 4 if variant == "percep" else 3
    B, T, Ld, hw = 3, 4, 32, (16, 24)
    torch.manual_seed(5)
    m = sfv.Seq2SeqBinaryVAE(in_ch, in_ch, Ld, Ld, variant=variant, input_hw=hw, compute_dtype="f32")
    params = {k: v.clone().requires_grad_() for k, v in m.state_dict().items()}
    m = m.cuda().eval()                                   # eval: no dropout on either side
    g = torch.Generator().manual_seed(6)
    item = torch.rand(B, 2, T, in_ch, *hw, generator=g)
    U = torch.rand(2, B * T, Ld, generator=g)
    tau, r, p, alpha, beta, margin = 0.8, 0.1, 0.1, 1.0, 0.5, 0.2
    steps = 3
    hist, last_grads = _oracle_step(variant, params, item, [U[0], U[1]], tau, r, p, alpha, beta, margin, steps)
    tr = FusedTrainer(m, lr=1e-3, alpha=alpha, beta_kl=beta, bernoulli_p=p, noise_ratio=r, margin=margin,
                      device_noise=False, use_graph=use_graph)
    for st in range(steps):
        losses = tr.step(item.cuda(), tau, U=U.cuda()).cpu().tolist()
        ref = hist[st]
        for got, k in zip(losses, ("total", "recon", "kl", "pair")):
            assert abs(got - ref[k]) < 2e-4 * max(1.0, abs(ref[k])), (st, k, got, ref[k])
    # gradients of the last step and the parameters after `steps` Adam updates
    lay = tr.eng.layout
    for k in lay.names:
        gr = lay.view(tr.gflat, k).cpu().double().reshape(-1)
        rf = last_grads[k].double().reshape(-1)
        assert float((gr - rf).norm()) <= 2e-3 * max(float(rf.norm()), 1e-7), k
    sd = m.state_dict()
    worst = max(float((sd[k].cpu() - params[k].detach()).abs().max()) for k in lay.names)
    assert worst < 5e-4, worst                      # 3 Adam steps of lr 1e-3 move a weight by <= 3e-3


def test_step_rejects_bad_input():
    import sfv_amd as sfv
    from importlib import import_module
    FusedTrainer = import_module("symbols-from-video_amd.trainer").FusedTrainer
    m = sfv.Seq2SeqBinaryVAE(4, 4, 32, 32, variant="percep", input_hw=(16, 16)).cuda()
    tr = FusedTrainer(m)
    with pytest.raises(ValueError):
        tr.step(torch.zeros(2, 3, 4, 4, 16, 16, device="cuda"), 1.0)
    with pytest.raises(ZeroDivisionError):
        tr.step(torch.zeros(2, 2, 1, 4, 16, 16, device="cuda"), 1.0)
    with pytest.raises(ValueError):
        FusedTrainer(sfv.Seq2SeqBinaryVAE(3, 3, 16, 16, variant="simple").cuda())


def test_input_buffer_is_read_in_place():
    """A batch written into trainer.input_buffer() and passed back to step() gives the same losses and weights as
    the same batch passed as a separate tensor (which step() copies into that buffer)."""
    import sfv_amd as sfv
    from importlib import import_module
    FusedTrainer = import_module("symbols-from-video_amd.trainer").FusedTrainer
    B, T, Ld, hw = 2, 3, 32, (16, 16)
    g = torch.Generator().manual_seed(9)
    item = torch.rand(B, 2, T, 4, *hw, generator=g).cuda()
    U = torch.rand(2, B * T, Ld, generator=g).cuda()
    res = []
    for in_place in (False, True):
        torch.manual_seed(8)
        m = sfv.Seq2SeqBinaryVAE(4, 4, Ld, Ld, variant="percep", input_hw=hw, compute_dtype="f32").cuda().eval()
        tr = FusedTrainer(m, device_noise=False, use_graph=True)
        x = item
        if in_place:
            x = tr.input_buffer(B, T, 4, *hw)
            x.copy_(item)
        for _ in range(2):
            losses = tr.step(x, 0.9, U=U).clone()
        res.append((losses, m._flat.clone()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])


# The reference's own sweeps: latent_dim 25 / 50 / 75 / 100, T = 5 / 9 / 17 states, batches of
# odd size, native 88x160 frames whose 11x20 bottleneck is not a power of two.  One fused step (exact-f32 mode,
# eval: no dropout on either side) against the oracle's step: losses 2e-4, gradients 2e-3 relative L2.
@pytest.mark.parametrize("variant,in_ch,B,T,Ld,hw", [
    ("percep", 4, 3, 5, 25, (24, 40)),
    ("percep", 4, 1, 9, 50, (16, 16)),
    ("percep", 4, 2, 17, 100, (8, 16)),
    ("percep", 4, 1, 2, 75, (88, 160)),
    ("contrastive", 3, 2, 5, 50, (32, 24)),
    ("triplet", 3, 3, 9, 16, (16, 16)),
])
def test_fused_step_parameter_ranges(variant, in_ch, B, T, Ld, hw):
    import sfv_amd as sfv
    from importlib import import_module
    FusedTrainer = import_module("symbols-from-video_amd.trainer").FusedTrainer
    torch.manual_seed(21)
    m = sfv.Seq2SeqBinaryVAE(in_ch, in_ch, Ld, Ld, variant=variant, input_hw=hw, compute_dtype="f32")
    params = {k: v.clone().requires_grad_() for k, v in m.state_dict().items()}
    m = m.cuda().eval()
    g = torch.Generator().manual_seed(22)
    item = torch.rand(B, 2, T, in_ch, *hw, generator=g)
    U = torch.rand(2, B * T, Ld, generator=g)
    tau, r, p, alpha, beta, margin = 0.6, 0.3, 0.1, 0.7, 0.4, 0.2
    hist, grads = _oracle_step(variant, params, item, [U[0], U[1]], tau, r, p, alpha, beta, margin, 1)
    tr = FusedTrainer(m, lr=1e-3, alpha=alpha, beta_kl=beta, bernoulli_p=p, noise_ratio=r, margin=margin,
                      device_noise=False, use_graph=False)
    losses = tr.step(item.cuda(), tau, U=U.cuda()).cpu().tolist()
    for got, k in zip(losses, ("total", "recon", "kl", "pair")):
        assert abs(got - hist[0][k]) < 2e-4 * max(1.0, abs(hist[0][k])), (k, got, hist[0][k])
    lay = tr.eng.layout
    for k in lay.names:
        gr = lay.view(tr.gflat, k).cpu().double().reshape(-1)
        rf = grads[k].double().reshape(-1)
        assert float((gr - rf).norm()) <= 2e-3 * max(float(rf.norm()), 1e-7), k


def test_bf16_graph_steps_track_f32_steps():
    """The bench configuration in small: bf16 storage, dropout on, device-side noise, HIP-graph replay.  Twenty
    steps must keep the loss finite, close to the exact-f32 trainer fed the same noise-free setting, and going down."""
    import sfv_amd as sfv
    from importlib import import_module
    FusedTrainer = import_module("symbols-from-video_amd.trainer").FusedTrainer
    B, T, Ld, hw = 4, 8, 32, (32, 32)
    g = torch.Generator().manual_seed(31)
    item = torch.randn(B, 2, T, 4, *hw, generator=g).cuda()
    U = torch.rand(2, B * T, Ld, generator=g).cuda()
    curves = {}
    for dt in ("f32", "bf16"):
        torch.manual_seed(30)
        m = sfv.Seq2SeqBinaryVAE(4, 4, Ld, Ld, variant="percep", input_hw=hw, compute_dtype=dt).cuda().eval()
        tr = FusedTrainer(m, lr=1e-3, alpha=1.0, beta_kl=1.0, bernoulli_p=0.1, noise_ratio=0.1, device_noise=False,
                          use_graph=True)
        curves[dt] = [tr.step(item, 0.7, U=U)[0].item() for _ in range(20)]
    f, b = curves["f32"], curves["bf16"]
    assert all(np.isfinite(b)) and b[-1] < b[0]
    assert max(abs(x - y) / abs(x) for x, y in zip(f, b)) < 2e-2, (f[-1], b[-1])
