import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
from importlib import import_module
import sfv_amd as sfv
import rbvae_oracle as O
FT = import_module("symbols-from-video_amd.trainer").FusedTrainer
dtype = sys.argv[1] if len(sys.argv) > 1 else "f32"
train = (sys.argv[2] == "train") if len(sys.argv) > 2 else True
for B in (2, 16):
    T, Ld, hw = 8, 32, (32, 32)
    g = torch.Generator().manual_seed(60)
    item = torch.randn(B, 2, T, 4, *hw, generator=g)
    U = torch.rand(2, B * T, Ld, generator=g)
    shapes = [(256, 16, 16), (256, 8, 8), (256, 8, 8), (256, 16, 16)]
    masks = [[(torch.rand(B * T, *s, generator=g) >= 0.2).float() for s in shapes] for _ in range(2)]
    torch.manual_seed(61)
    m = sfv.Seq2SeqBinaryVAE(4, 4, Ld, Ld, variant="percep", input_hw=hw, compute_dtype=dtype)
    p64 = {k: v.clone().double().requires_grad_() for k, v in m.state_dict().items()}
    m = m.cuda(); m.train(train)
    r = O.step_losses("percep", p64, item.double(), [U[0].double(), U[1].double()], 0.7, 0.1, 0.1, 1.0, 1.0, train=train,
                      masks=[[x.double() for x in mm] for mm in masks] if train else None)
    r["total"].backward()
    tr = FT(m, lr=1e-3, alpha=1.0, beta_kl=1.0, bernoulli_p=0.1, noise_ratio=0.1, device_noise=False, use_graph=False)
    got = tr.step(item.cuda(), 0.7, U=U.cuda(), dropout_masks=masks if train else None).cpu().tolist()
    print(dtype, "train" if train else "eval", "B", B, "losses", got, [float(r[k]) for k in ("total", "recon", "kl", "pair")])
    lay = tr.eng.layout
    for k in lay.names:
        if "lstm" in k and not k.endswith("l0"):
            continue
        gr = lay.view(tr.gflat, k).cpu().double().reshape(-1)
        rf = p64[k].grad.reshape(-1)
        print(f"  {k:36s} rel {float((gr - rf).norm() / rf.norm()):.2e}  |ref| {float(rf.norm()):.3e}")
