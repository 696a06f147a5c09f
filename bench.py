#!/usr/bin/env python3
"""RBVAE hot-path benchmark: frames/s of the fused training step (encode + binarise + decode,
forward + backward + losses + Adam) on synthetic LDM-latent frames.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run, one rank per GPU, RCCL all-reduce of the flat gradients)

Workload (BASELINE.json configs[1]): percep_RBVAE, item [16,2,8,4,32,32] ~ N(0,1)
(256 frames per step per GPU = the latents of 256x256 frames), latent 32, 4-layer LSTMs, bf16
storage / f32 accumulation, tau 0.7, noise ratio 0.1, p 0.1, alpha = beta = 1, dropout on.
Prints ONE JSON line on rank 0.
"""
import argparse
import contextlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch

B_ITEMS, T_STATES, C_IN, HW, LATENT = 16, 8, 4, (32, 32), 32
TAU, NOISE_R, BERN_P, ALPHA, BETA = 0.7, 0.1, 0.1, 1.0, 1.0
MFMA_BF16_PEAK_TFLOPS = 2500.0     # MI355X_MICROARCH.md: dense bf16 MFMA peak
# HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this same command (separate
# passes, FETCH doubled per the guide's gfx950 correction); filled in from profiles/ by tools/pmc_traffic.py
PMC_TRAFFIC = {}
try:
    with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as _f:
        PMC_TRAFFIC = json.load(_f)
except (OSError, ValueError):
    pass


class KernelTimer:
    """HIP events around every launch of one kernel instance, on the stream it is launched on."""

    def __init__(self):
        self.pairs = []
        self.flops = 0.0

    @contextlib.contextmanager
    def __call__(self, flops):
        st = torch.cuda.current_stream()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(st)
        yield
        b.record(st)
        self.pairs.append((a, b))
        self.flops += flops

    def result(self):
        torch.cuda.synchronize()
        ms = sum(a.elapsed_time(b) for a, b in self.pairs)
        return ms, len(self.pairs), self.flops


def roofline_leg(trainer, item, steps, tname):
    """Re-run `steps` steps eagerly on ONE stream (RBVAE_OVERLAP off for this leg, so an event pair brackets exactly
    one kernel) with HIP events around every launch of the two matrix-core kernel families, keyed by the template
    instance the library's dispatch picks.  Returns {instance: (ms, launches, algorithmic flops)}.

      gather_gemm_k<T,4,8,3>  128x128 row-gather GEMM, deep K, 129..256 workgroups: conv2 forward, first deconv
                              forward (4 parity classes) and their two input-gradient twins
      wgrad_gemm_k<T,2,3>     128x128 weight-gradient GEMM with K split over workgroups: the five 256-channel ones
    Algorithmic FLOPs: 2 * output rows * Nout * Kc * taps for the gather GEMM (the 4 parity classes of a
    transposed convolution share its k*k taps: taps/4 per output pixel), 2 * P * Co * Ci * taps for wgrad."""
    eng = trainer.eng
    timers = {}
    orig_gemm, orig_wgrad = eng._gemm, eng._wgrad
    ke = eng.ke

    def timed_gemm(A, W, out, bias, gate, mask, nimg, ih, iw, th, tw, sa, oh, ow, so, kc, nout, lda, ldo, taps, cls_key,
                   **kw):
        args = (A, W, out, bias, gate, mask, nimg, ih, iw, th, tw, sa, oh, ow, so, kc, nout, lda, ldo, taps, cls_key)
        ncls = 4 if cls_key == "dgrad" else 1
        max_taps = {"one": 1, "conv": taps, "dgrad": 4 if taps == 9 else max(1, taps // 4)}[cls_key]
        blocks = -(-(nimg * th * tw) // 128) * -(-nout // 128) * ncls
        deep = nout > 64 and max_taps * (kc // ke) > 2            # same rules as dispatch_gg()
        if not deep:
            name = "gather_gemm_k<%s, single/double buffer>" % tname
        elif blocks > 256:
            name = "gather_gemm_k<%s, 4, 8, 2>" % tname
        elif blocks <= 64:
            name = "gather_gemm_k<%s, 1, 4, 3>" % tname
        elif blocks <= 128:
            name = "gather_gemm_k<%s, 2, 4, 3>" % tname
        else:
            name = "gather_gemm_k<%s, 4, 8, 3>" % tname
        rows, t_eff = (nimg * th * tw * 4, taps / 4.0) if cls_key == "dgrad" else (nimg * th * tw, float(taps))
        with timers.setdefault(name, KernelTimer())(2.0 * rows * nout * kc * t_eff):
            return orig_gemm(*args, **kw)

    def timed_wgrad(Dy, In, idx, P, Co, Ci, ldy, ldi, taps, out, dims, strides, tag=None):
        name = "wgrad_gemm_k<%s, %d, K-split>" % (tname, 2 if Ci > 64 else 1)
        with timers.setdefault(name, KernelTimer())(2.0 * P * Co * Ci * taps):
            return orig_wgrad(Dy, In, idx, P, Co, Ci, ldy, ldi, taps, out, dims, strides, tag=tag)

    calib = KernelTimer()                  # event pairs around nothing: the event records' own cost

    eng._gemm, eng._wgrad = timed_gemm, timed_wgrad
    overlap = eng.overlap
    eng.overlap = False
    trainer.instrument = True
    try:
        for _ in range(2):                   # the eager one-stream path's own first-use work (job tables, buffers)
            trainer.step(item, TAU)
        torch.cuda.synchronize()
        timers.clear()
        for _ in range(steps):
            # Hold the stream behind a ~20 ms device-side sleep while the host enqueues the whole step: the GPU then
            # runs the launches and event records back to back, so an event pair brackets the kernel and not the
            # host's launch latency (eager launches are host bound: ~10 us of idle queue per launch otherwise).
            torch.cuda._sleep(40_000_000)
            for _ in range(4):
                with calib(0.0):
                    pass
            trainer.step(item, TAU)
            torch.cuda.synchronize()
        cms, cn, _ = calib.result()
        per_pair = cms / max(cn, 1)
        res = {}
        for k, t in timers.items():
            ms, n, fl = t.result()
            res[k] = (max(ms - n * per_pair, 1e-6), n, fl)
        res["_event_pair_overhead_us"] = per_pair * 1e3
    finally:
        eng._gemm, eng._wgrad = orig_gemm, orig_wgrad
        eng.overlap = overlap
        trainer.instrument = None
    return res


def cpu_baseline(seconds_budget=15.0):
    """The oracle (plain-torch restatement pinned to the reference by tests/golden) on this host's cores."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import rbvae_oracle as O
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))                   # the one-GPU box's CPU share
    torch.set_num_threads(cores)
    p = O.init_params("percep", C_IN, C_IN, LATENT, HW, seed=1234)
    for v in p.values():
        v.requires_grad_()
    g = torch.Generator().manual_seed(1234)
    Bc = 4                                           # bounded sample: 4 items = 64 frames per step
    item = torch.randn(Bc, 2, T_STATES, C_IN, *HW, generator=g)
    times = []
    t_start = time.perf_counter()
    state = {}
    step = 0
    while True:
        U = [torch.rand(Bc * T_STATES, LATENT), torch.rand(Bc * T_STATES, LATENT)]
        t0 = time.perf_counter()
        res = O.step_losses("percep", p, item, U, TAU, NOISE_R, BERN_P, ALPHA, BETA, train=True)
        for v in p.values():
            v.grad = None
        res["total"].backward()
        step += 1
        with torch.no_grad():
            O.adam_step({k: v for k, v in p.items()}, {k: v.grad for k, v in p.items()}, state, 1e-3, step)
        times.append(time.perf_counter() - t0)
        print(f"[cpu_baseline] step {len(times)}: {times[-1]:.2f} s", file=sys.stderr, flush=True)
        if (len(times) >= 2 and time.perf_counter() - t_start > seconds_budget) or len(times) >= 400:
            break
    times = times[1:] if len(times) > 1 else times   # first step warms the allocator / oneDNN primitives
    frames = Bc * 2 * T_STATES
    return {"value": round(frames / (sum(times) / len(times)), 2), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{len(times)} steps of {frames} frames (item [{Bc},2,{T_STATES},{C_IN},{HW[0]},{HW[1]}]), "
                      f"fwd+bwd+Adam, torch {torch.__version__} CPU, {cores} threads"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    args = ap.parse_args()

    import sfv_amd as sfv
    from importlib import import_module
    ddp = import_module("symbols-from-video_amd.ddp")
    trainer_mod = import_module("symbols-from-video_amd.trainer")
    # RCCL ("nccl") is the backend of record; RBVAE_DIST_BACKEND=gloo lets the N > 1 code path be rehearsed
    # with several ranks sharing one GPU (RCCL refuses duplicate devices)
    backend = os.environ.get("RBVAE_DIST_BACKEND", "nccl")
    ndev = max(torch.cuda.device_count(), 1)
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % ndev)
    rank, world, local = ddp.init_from_env(backend)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    local = local % ndev
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    torch.manual_seed(1234)                          # identical initial weights on every rank
    model = sfv.Seq2SeqBinaryVAE(C_IN, C_IN, LATENT, LATENT, variant="percep", input_hw=HW,
                                 compute_dtype=args.dtype).to(dev).train()
    ddp.broadcast_(model._flat)
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)          # per-rank shard of the global batch
    item = torch.randn(B_ITEMS, 2, T_STATES, C_IN, *HW, generator=g).to(dev)
    tr = trainer_mod.FusedTrainer(model, lr=1e-3, alpha=ALPHA, beta_kl=BETA, bernoulli_p=BERN_P, noise_ratio=NOISE_R,
                                  device_noise=True, use_graph=not args.no_graph)
    frames_per_step = B_ITEMS * 2 * T_STATES
    # the batch lives in the trainer's static input buffer (a device-resident data loader gathers into it in place)
    buf = tr.input_buffer(B_ITEMS, T_STATES, C_IN, *HW)
    buf.copy_(item)
    item = buf

    for _ in range(args.warmup):
        tr.step(item, TAU)
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        tr.step(item, TAU)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], device=dev)
    if world > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    dt = float(t.item())
    losses = [float(v) for v in tr.losses.tolist()]

    roof = None
    roof_all = None
    cpu = None
    # every rank runs the instrumented leg (its steps contain the gradient all-reduce: a collective)
    tname = "unsigned short" if args.dtype == "bf16" else "float"
    legs = roofline_leg(tr, item, min(args.steps, 20), tname)
    if rank == 0:
        peak = MFMA_BF16_PEAK_TFLOPS if args.dtype == "bf16" else 157.3
        roof_all = {}
        ev_us = legs.pop("_event_pair_overhead_us", 0.0)
        for name, (ms, launches, flops) in sorted(legs.items(), key=lambda kv: -kv[1][0]):
            ach = flops / (ms * 1e-3) / 1e12
            roof_all[name] = {"achieved": round(ach, 2), "frac": round(ach / peak, 4), "launches": launches,
                              "avg_us": round(ms * 1e3 / max(launches, 1), 2),
                              "us_per_step": round(ms * 1e3 / min(args.steps, 20), 1)}
        dom = next(iter(roof_all))                       # the instance with the largest total time
        d = roof_all[dom]
        roof = {"bound": "mfma", "kernel": dom, "achieved": d["achieved"], "peak": peak, "unit": "TFLOP/s",
                "frac": d["frac"], "traffic": PMC_TRAFFIC.get(dom), "launches": d["launches"], "avg_us": d["avg_us"],
                "us_per_step": d["us_per_step"], "event_pair_overhead_us_subtracted": round(ev_us, 2)}
        if world == 1 and not args.no_cpu:
            cpu = cpu_baseline()
    if world > 1:
        torch.distributed.barrier()
    if rank == 0:
        value = frames_per_step * world * args.steps / dt
        line = {"metric": "frames/sec (enc+binarise+dec fwd+bwd), batch 256x256, 1/2/4/8 MI355X",
                "value": round(value, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
                "config": {"workload": "percep_RBVAE fused train step (fwd+bwd+losses+Adam), item [16,2,8,4,32,32] "
                                       "per GPU = 256 frames/step/GPU (latents of 256x256 frames), latent 32, "
                                       "4-layer LSTMs, dropout on",
                           "frames_per_step_per_gpu": frames_per_step, "global_frames_per_step": frames_per_step * world,
                           "parallelism": f"dp{world}", "graph": not args.no_graph,
                           "last_losses": {"total": losses[0], "recon": losses[1], "kl": losses[2], "pair": losses[3]}},
                "roofline": roof, "roofline_all_mfma_kernels": roof_all, "cpu_baseline": cpu}
        print(json.dumps(line))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
