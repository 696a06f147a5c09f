"""Per-row timing of the batched job launches (weight pack, gradient reduce, fused update) of the bench workload, the
native 4x88x160 percep model or cfg 3: python3 tools/time_jobs.py [bench|native|cfg3]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sfv_amd as sfv
from importlib import import_module
trainer_mod = import_module("symbols-from-video_amd.trainer")
L = sfv._lib
dev = torch.device("cuda", 0)
torch.manual_seed(1234)
cfg = sys.argv[1] if len(sys.argv) > 1 else "bench"          # bench | native | cfg3
if cfg == "native":
    model = sfv.Seq2SeqBinaryVAE(4, 4, 32, 32, variant="percep", compute_dtype="bf16").to(dev).train()
    item = torch.randn(8, 2, 8, 4, 88, 160, device=dev)
elif cfg == "cfg3":
    model = sfv.Seq2SeqBinaryVAE(3, 3, 32, 32, variant="contrastive", compute_dtype="bf16").to(dev).train()
    item = torch.rand(8, 2, 8, 3, 256, 256, device=dev)
else:
    model = sfv.Seq2SeqBinaryVAE(4, 4, 32, 32, variant="percep", input_hw=(32, 32), compute_dtype="bf16").to(dev).train()
    item = torch.randn(16, 2, 8, 4, 32, 32, device=dev)
tr = trainer_mod.FusedTrainer(model, lr=1e-3, alpha=1.0, beta_kl=1.0, bernoulli_p=0.1, noise_ratio=0.1,
                              device_noise=True, use_graph=False)
for _ in range(3):
    tr.step(item, 0.7)
torch.cuda.synchronize()
eng = list(model._engines.values())[0]
names = {0: "pack", 1: "permute_reduce", 2: "reduce_rows", 3: "conv_pack", 4: "conv_reduce"}
def time_tab(label, tab, n, jl):
    """GPU-side duration of the whole table and of every job alone: the launches are queued behind a device-side
    sleep, so the event pairs bracket kernels, not host launch latency."""
    def timed(calls):
        torch.cuda.synchronize()
        torch.cuda._sleep(40_000_000)
        evs = []
        for ptr, cnt in calls:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); L.call("rbvae_run_jobs", ptr, cnt, int(os.environ.get("JOB_BLOCKS", "256"))); b.record()
            evs.append((a, b))
        torch.cuda.synchronize()
        return [a.elapsed_time(b) * 1e3 for a, b in evs]
    timed([(tab, n)])
    whole = timed([(tab, n)] * 3)
    # the same table with exactly the workgroups its jobs can use (rbvae_run_jobs_sized)
    from importlib import import_module as _im
    bmap, nb = _im("symbols-from-video_amd.engine").job_block_map(jl.rows, dev, 256)
    def timed_sized(reps):
        torch.cuda.synchronize()
        torch.cuda._sleep(40_000_000)
        evs = []
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); L.call("rbvae_run_jobs_sized", tab, bmap, nb); b.record()
            evs.append((a, b))
        torch.cuda.synchronize()
        return [a.elapsed_time(b) * 1e3 for a, b in evs]
    timed_sized(1)
    print(f"{label}: sized launch of {nb} workgroups (instead of {n * 256}): {min(timed_sized(3)):.1f} us")
    each = timed([(tab[i:i + 1].contiguous(), 1) for i in range(n)])
    print(f"{label}: {n} jobs, all together {min(whole):.1f} us (event pair overhead ~4.8 us included)")
    for i, r in enumerate(jl.rows):
        print(f"  job {i:2d} {names[r[0]]:15s} dims=({r[3]},{r[4]},{r[5]}) strides=({r[6]},{r[7]},{r[8]}) nslab={r[9]} fast={r[13] >> 32} inner={r[14]}: {each[i]:6.1f} us")
eng.pack(model._flat)
for key, val in eng._pack_tab.items():
    if len(val) == 3:
        time_tab("pack", *val)
for sig, (tab, n, jl) in eng._bwd_tab.items():
    time_tab("bwd reduce", tab, n, jl)

# the fused update launch (Adam + packed copies), whole and job by job
tab, nj = eng.update_jobs(model._flat, tr.gflat, tr.m, tr.vv, tr.hyper, tr.betas, tr.eps, 1.0)
rows = tab.cpu().tolist()
names.update({5: "gather", 6: "adam+pack", 7: "adam"})
class _JL: pass
jl = _JL(); jl.rows = rows
time_tab("fused update", tab, nj, jl)
