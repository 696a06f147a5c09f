import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from importlib import import_module
import sfv_amd as sfv
FT = import_module("symbols-from-video_amd.trainer").FusedTrainer
B, T, Ld, hw = 2, 3, 32, (16, 16)
g = torch.Generator().manual_seed(44)
item = torch.rand(B, 2, T, 4, *hw, generator=g).cuda()
U = torch.rand(2, B * T, Ld, generator=g).cuda()
def mk(seed):
    torch.manual_seed(seed)
    return sfv.Seq2SeqBinaryVAE(4, 4, Ld, Ld, variant="percep", input_hw=hw).cuda().eval()
m = mk(45)
tr = FT(m, lr=2e-3, device_noise=False, use_graph=False, seed=77)
for _ in range(3):
    tr.step(item, 0.8, U=U)
sd = tr.state_dict(); msd = {k: v.clone() for k, v in m.state_dict().items()}
flat3, m3_, v3_ = m._flat.clone(), tr.m.clone(), tr.vv.clone()
l4 = tr.step(item, 0.8, U=U).clone(); g4 = tr.gflat.clone(); after = m._flat.clone(); hy = tr.hyper.clone()
mm = mk(46); mm.load_state_dict(msd)
print("weights equal after load:", torch.equal(mm._flat, flat3))
t3 = FT(mm, lr=1e-3, device_noise=False, use_graph=False)
t3.load_state_dict(sd)
print("m equal", torch.equal(t3.m, m3_), "v equal", torch.equal(t3.vv, v3_), "step", t3.step_dev.item())
l = t3.step(item, 0.8, U=U).clone()
print("loss", l4.tolist(), l.tolist())
print("grad equal", torch.equal(t3.gflat, g4), float((t3.gflat - g4).abs().max()))
print("hyper", hy.tolist(), t3.hyper.tolist(), "lr_dev", t3.lr_dev.item(), tr.lr_dev.item())
print("after equal", torch.equal(mm._flat, after), float((mm._flat - after).abs().max()))
