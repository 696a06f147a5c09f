"""Same-box A/B of engine settings on the bench workload (or native / cfg 3): ms per step for each setting, alternating.
usage: python tools/ab_step.py [--cfg bench|native|cfg3] [--steps 300] [--rounds 2] "name:attr=val,tr.attr=val" ...   (attr: engine attribute, tr.attr: trainer attribute)
e.g.   python tools/ab_step.py "row:wgrad_row=1" "gemm:wgrad_row=0" "row8:wgrad_row=1,_wr_min_steps=8" """
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sfv_amd as sfv

ap = argparse.ArgumentParser()
ap.add_argument("--cfg", default="bench")
ap.add_argument("--steps", type=int, default=300)
ap.add_argument("--rounds", type=int, default=2)
ap.add_argument("specs", nargs="+")
a = ap.parse_args()
dev = torch.device("cuda", 0)
CFG = {"bench": ("percep", 4, (32, 32), (16, 2, 8, 4, 32, 32), False), "native": ("percep", 4, (88, 160), (8, 2, 8, 4, 88, 160), False),
       "cfg3": ("contrastive", 3, (256, 256), (8, 2, 8, 3, 256, 256), True), "cfg5": ("percep", 4, (64, 64), (4, 2, 8, 4, 64, 64), False)}
variant, cin, hw, shape, uniform = CFG[a.cfg]


def build(spec):
    name, _, sets = spec.partition(":")
    torch.manual_seed(0)
    m = sfv.Seq2SeqBinaryVAE(cin, cin, 32, 32, variant=variant, input_hw=hw, compute_dtype="bf16").to(dev).train()
    g = torch.Generator(device="cpu").manual_seed(1234)
    item = (torch.rand(*shape, generator=g) if uniform else torch.randn(*shape, generator=g)).to(dev)
    tr = sfv.FusedTrainer(m, alpha=1.0, beta_kl=1.0, bernoulli_p=0.1, noise_ratio=0.1, device_noise=True, use_graph=True, seed=1234)
    eng = m._engine_for(item)          # the engine the trainer will pick up at its first step
    for kv in filter(None, sets.split(",")):
        k, v = kv.split("=")
        if k.startswith("hook."):               # a library switch of include/rbvae_dbg.h, e.g. hook.lstm_unit_threads=0 (set before the capture)
            getattr(sfv._lib.dbg_lib(), "rbvae_dbg_" + k[5:])(int(v))
            continue
        obj = eng
        if k.startswith("tr."):                 # a trainer attribute (e.g. tr.early_tail_update=0)
            obj, k = tr, k[3:]
        assert hasattr(obj, k), k
        setattr(obj, k, type(getattr(obj, k))(int(v)) if not isinstance(getattr(obj, k), float) else float(v))
    for _ in range(10):
        tr.step(item, 0.7)
    torch.cuda.synchronize()
    return name, tr, item


runs = [build(s) for s in a.specs]
for r in range(a.rounds):
    for name, tr, item in runs:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            tr.step(item, 0.7)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.steps
        print(f"{name:16s} {dt * 1e3:.4f} ms/step  losses {[round(float(x), 4) for x in tr.losses.tolist()]}", flush=True)
