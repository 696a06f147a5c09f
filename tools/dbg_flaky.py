import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, "tests"); sys.path.insert(0, "oracle")
import torch, numpy as np
import sfv_amd as sfv
from _golden import load, case_masks
g = load("percep_small_eval")
def run(dtype):
    torch.manual_seed(int(g["meta/seed"]))
    m = sfv.Seq2SeqBinaryVAE(4, 4, 32, 32, variant="percep", input_hw=(16, 24), compute_dtype=dtype).cuda().eval()
    item = torch.from_numpy(g["item"]).cuda()
    U = [torch.from_numpy(g["U0"]).cuda(), torch.from_numpy(g["U1"]).cuda()]
    outs, res = [], []
    for rep in range(4):
        m.zero_grad()
        tot = 0
        for vw in range(2):
            xr, h, z = m(item[:, vw], temperature=0.7, hard=False, noise_ratio=0.1, u=U[vw])
            if vw == 1: outs.append((xr.detach().clone(), h.detach().clone(), z.detach().clone()))
            tot = tot + sfv.recon_loss(xr, item[:, vw]) + sfv.kl_binary_concrete(z, p=0.1) + (h * h).mean()
        tot.backward()
        res.append({k: p.grad.clone() for k, p in zip(m.state_dict().keys(), m.parameters())})
    for a in range(4):
        for b in range(a + 1, 4):
            fo = [torch.equal(outs[a][i], outs[b][i]) for i in range(3)]
            bad = [k for k in res[a] if not torch.equal(res[a][k], res[b][k])]
            print(dtype, f"rep {a} vs {b}: fwd equal {fo}; grads differing: {len(bad)} {bad[:4]}")
    return res
a = run("f32"); b = run("bf16")
for k in a[1]:
    na, nb = float(a[1][k].norm()), float(b[1][k].norm())
    if abs(na - nb) > 0.1 * max(na, 1e-9):
        print("f32 vs bf16 norm mismatch", k, na, nb, [float(b[r][k].norm()) for r in range(4)])
