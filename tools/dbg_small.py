import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import sfv_amd as sfv
import rbvae_oracle as O
from _gates import device_gates, count_ties
PAIR = sys.argv[1]; ALPHA = float(sys.argv[2])
B, T, Ld, hw = 2, 3, 32, (8, 8)
torch.manual_seed(71)
m = sfv.Seq2SeqBinaryVAE(4, 4, Ld, Ld, variant="percep", input_hw=hw, compute_dtype="f32")
w0 = {k: v.clone() for k, v in m.state_dict().items()}
m = m.cuda().eval()
g = torch.Generator().manual_seed(72)
item = torch.randn(B, 2, T, 4, *hw, generator=g) * 0.18
U = torch.rand(2, B * T, Ld, generator=g)
tr = sfv.FusedTrainer(m, alpha=ALPHA, beta_kl=0.5, bernoulli_p=0.1, noise_ratio=0.1, margin=0.2, device_noise=False, use_graph=False, pair_loss=PAIR)
got = tr.step(item.cuda(), 0.7, U=U.cuda()).cpu().tolist()
gates = device_gates(tr, B, T)
for dt in (torch.float64,):
    p = {k: v.detach().clone().to(dt).requires_grad_() for k, v in w0.items()}
    pre = []
    r = O.step_losses("percep", p, item.to(dt), [U[0].to(dt), U[1].to(dt)], 0.7, 0.1, 0.1, ALPHA, 0.5, 0.2, pair_loss=PAIR, gates=gates, pre=pre)
    r["total"].backward()
    print(dt, "ties", count_ties(pre, gates, None, 1.0), "loss", got[0], float(r["total"]))
    lay = tr.eng.layout
    for k in lay.names:
        if not ("conv.0.w" in k or "conv.6" in k or "fc.w" in k or "weight_ih_l0" in k): continue
        gr = lay.view(tr.gflat, k).cpu().double().reshape(-1); rf = p[k].grad.double().reshape(-1)
        print(f"  {k:36s} rel {float((gr-rf).norm()/rf.norm()):.2e} |ref| {float(rf.norm()):.2e}")
