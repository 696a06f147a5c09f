"""percep_RBVAE at the reference's native latent size 4x88x160 (fc 56320): fused step, 128 frames per step."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sfv_amd as sfv
from importlib import import_module
FusedTrainer = import_module("symbols-from-video_amd.trainer").FusedTrainer
import sys
for dtype in (sys.argv[1:] or ["bf16", "f32"]):
    torch.manual_seed(0)
    m = sfv.Seq2SeqBinaryVAE(4, 4, 32, 32, variant="percep", compute_dtype=dtype).cuda().train()
    item = torch.randn(8, 2, 8, 4, 88, 160, device="cuda")
    tr = FusedTrainer(m, alpha=1.0, beta_kl=1.0, bernoulli_p=0.1)
    for _ in range(5): tr.step(item, 0.7)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): tr.step(item, 0.7)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print("native", dtype, f"{dt*1e3:.3f} ms/step  {128/dt:.0f} frames/s  losses", [round(v, 4) for v in tr.losses.tolist()], flush=True)
