#!/usr/bin/env python3
"""RBVAE hot-path benchmark: frames/s of the fused training step (encode + binarise + decode,
forward + backward + losses + Adam) on synthetic LDM-latent frames.

  python bench.py --gpus N --steps K --warmup W

N > 1 works both ways: launched by torch.distributed.run (RANK / WORLD_SIZE in the environment), or typed as is --
then this process spawns N fresh rank processes BEFORE anything touches the GPU, waits for them and relays rank 0's
JSON line.  One rank per GPU, RCCL all-reduce of the flat gradients (RBVAE_DIST_BACKEND=gloo rehearses the N > 1
path with several ranks sharing one GPU).

Workload (BASELINE.json configs[1]): percep_RBVAE, items [16,2,8,4,32,32] ~ N(0,1)
(256 frames per step per GPU = the latents of 256x256 frames), latent 32, 4-layer LSTMs, bf16
storage / f32 accumulation, tau 0.7, noise ratio 0.1, p 0.1, alpha = beta = 1, dropout on.
A step includes the gather of its batch from the HBM-resident latent table (the reference's DataLoader + .to(device)).
Prints ONE JSON line on rank 0.
"""
import argparse
import contextlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B_ITEMS, T_STATES, C_IN, HW, LATENT = 16, 8, 4, (32, 32), 32
TAU, NOISE_R, BERN_P, ALPHA, BETA = 0.7, 0.1, 0.1, 1.0, 1.0
FRAMES_PER_STATE = 512                # synthetic latent table: 8 states x 512 frames (4 096 frames, 64 MB f32)
MFMA_BF16_PEAK_TFLOPS = 2500.0        # MI355X_MICROARCH.md: dense bf16 MFMA peak
# SURVEY.md 8d, cfg 2: 596.3 MFLOP and 1.805 MB (bf16 layer-boundary bytes) per frame
#   HBM roof  = 8.0 TB/s / 1.805 MB  = 4.43 M frames/s;  MFMA roof = 2.5 PFLOP/s / 596.3 MFLOP = 4.19 M frames/s
ROOF_HBM_FPS, ROOF_MFMA_FPS = 8.0e12 / 1.805e6, 2.5e15 / 596.3e6
ROOF_F32 = (8.0e12 / 3.611e6, 157.3e12 / 596.3e6)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(n, argv, cmd=None, deadline_s=None, poll_s=0.2):
    """`python bench.py --gpus N` typed as is: N rank processes, started before this one has made any GPU call
    (it never makes one), environment as torch.distributed.run would set it.  Returns the exit code.

    Every child is polled: the first non-zero exit (a rank that died before or after rendezvous) terminates the others
    and becomes the exit code, and so does an overall deadline (RBVAE_BENCH_DEADLINE seconds, default 540: below the
    driver's 600 s limit) -- a sibling's crash never leaves rank 0 waiting in init_process_group.  Rank 0's stdout is
    relayed; every rank's stderr goes to this process's stderr.  cmd: the rank command (tests launch a stub)."""
    port = _free_port()
    deadline_s = float(os.environ.get("RBVAE_BENCH_DEADLINE", "540")) if deadline_s is None else deadline_s
    cmd = [sys.executable, os.path.abspath(__file__)] + list(argv) if cmd is None else list(cmd)
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    import threading
    out = []
    reader = threading.Thread(target=lambda: out.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    t0, code, why = time.monotonic(), 0, None
    while True:
        codes = [p.poll() for p in procs]
        bad = next(((r, c) for r, c in enumerate(codes) if c not in (None, 0)), None)
        if bad is not None:
            code, why = bad[1], f"rank {bad[0]} exited with code {bad[1]}"
            break
        if all(c == 0 for c in codes):
            break
        if time.monotonic() - t0 > deadline_s:
            code, why = 124, f"deadline of {deadline_s:.0f} s passed"
            break
        time.sleep(poll_s)
    if why is not None:
        sys.stderr.write(f"bench.py: {why}; terminating the other ranks\n")
        for p in procs:
            if p.poll() is None:
                p.terminate()
        t1 = time.monotonic()
        while any(p.poll() is None for p in procs) and time.monotonic() - t1 < 5.0:
            time.sleep(0.05)
        for p in procs:
            if p.poll() is None:
                p.kill()
        for p in procs:
            p.wait()
    reader.join(timeout=5.0)
    sys.stdout.write(b"".join(out).decode())
    sys.stdout.flush()
    return code if code else 0


class KernelTimer:
    """HIP events around every launch of one kernel instance, on the stream it is launched on."""

    def __init__(self):
        self.pairs = []
        self.flops = 0.0

    @contextlib.contextmanager
    def __call__(self, flops):
        import torch
        st = torch.cuda.current_stream()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(st)
        yield
        b.record(st)
        self.pairs.append((a, b))
        self.flops += flops

    def result(self):
        import torch
        torch.cuda.synchronize()
        ms = sum(a.elapsed_time(b) for a, b in self.pairs)
        return ms, len(self.pairs), self.flops


def roofline_leg(trainer, steps, tname, overlap):
    """Re-run `steps` steps eagerly with HIP events around every launch of the two matrix-core kernel families, each
    pair recorded on the stream its kernel is launched on, keyed by the template instance the library's dispatch picks.

    overlap=True: the SAME multi-stream schedule as the captured graph (side streams on), so a kernel is timed beside
    whatever the graph runs beside it -- this is what `roofline.frac` is computed from.  overlap=False: everything on
    one stream (the kernel alone on the chip) -- reported as `isolated`.  Either way the streams are held behind a
    ~20 ms device-side sleep while the host enqueues the whole step, so an event pair brackets the kernel and not the
    host's launch latency, and the cost of an empty event pair is measured the same way and subtracted.

      gather_gemm_k<T,2,4,3,64-row tile>  64x64 tiles, deep K, 4096 rows: conv3 forward, first deconv's input gradient
      gather_gemm_k<T,4,8,4>  128x128 row-gather GEMM, deep K, 129..256 workgroups, ring of four LDS-DMA stages: conv2
                              forward, first deconv forward (4 parity classes) and their two input-gradient twins
      wgrad_row_k             3x3 stride-2 weight gradient, the three taps of a kernel row per workgroup (csrc/wgrad_row.hip)
      wgrad_gemm_k<T,2,3>     128x128 weight-gradient GEMM with K split over workgroups: the other 256-channel ones
    Algorithmic FLOPs: 2 * output rows * Nout * Kc * taps for the gather GEMM (the 4 parity classes of a
    transposed convolution share its k*k taps: taps/4 per output pixel), 2 * P * Co * Ci * taps for wgrad."""
    import torch
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
        fc = (getattr(eng, "fc_gemm", False) and cls_key == "one" and kc in (64, 128) and nout >= 1024 and nimg <= 4096
              and gate is None and mask is None and tname == "unsigned short" and not kw.get("relu") and not kw.get("drop_mode")
              and kw.get("scale", 1.0) == 1.0 and kw.get("bias_grad") is None)
        if fc:
            name = "fc_gemm_k"                      # the two K = 64 fc products (csrc/fc_gemm.hip)
        elif not deep:
            name = "gather_gemm_k<%s, single/double buffer>" % tname
        elif blocks > 256:
            name = "gather_gemm_k<%s, 4, 8, 2>" % tname
        elif blocks <= 64 and tname == "unsigned short" and kw.get("bias_grad") is None and kw.get("colsum_ws") is None:
            name = "gather_gemm_k<%s, 2, 4, 3, 64-row tile>" % tname
        elif blocks <= 64:
            name = "gather_gemm_k<%s, 1, 4, 3>" % tname
        elif blocks <= 128:
            name = "gather_gemm_k<%s, 2, 4, 3>" % tname
        else:
            # (one workgroup per CU: a ring of four stages when the K loop has at least eight steps, else three)
            name = "gather_gemm_k<%s, 4, 8, %d>" % (tname, 4 if max_taps * (kc // ke) >= 8 else 3)
        rows, t_eff = (nimg * th * tw * 4, taps / 4.0) if cls_key == "dgrad" else (nimg * th * tw, float(taps))
        with timers.setdefault(name, KernelTimer())(2.0 * rows * nout * kc * t_eff):
            return orig_gemm(*args, **kw)

    def timed_wgrad(Dy, In, idx, P, Co, Ci, ldy, ldi, taps, out, dims, strides, tag=None, **kw):
        # the engine picks the kernel (three-taps-per-workgroup rows / nine-tap halo / per-tap GEMM) and says which
        st = torch.cuda.current_stream()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(st)
        r = orig_wgrad(Dy, In, idx, P, Co, Ci, ldy, ldi, taps, out, dims, strides, tag=tag, **kw)
        b.record(st)
        fam = eng.last_wgrad_kernel or "wgrad_gemm_k"
        name = {"wgrad_row_k": "wgrad_row_k<3 taps per workgroup, K-split>", "wgrad_halo_k": "wgrad_halo_k<9 taps per workgroup, K-split>"}.get(
            fam, "wgrad_gemm_k<%s, %d, K-split>" % (tname, 2 if Ci > 64 else 1))
        t = timers.setdefault(name, KernelTimer())
        t.pairs.append((a, b))
        t.flops += 2.0 * P * Co * Ci * taps
        return r

    calib = KernelTimer()                  # event pairs around nothing: the event records' own cost

    eng._gemm, eng._wgrad = timed_gemm, timed_wgrad
    was_overlap = eng.overlap
    eng.overlap = overlap
    trainer.instrument = True
    try:
        for _ in range(2):                   # the eager path's own first-use work (job tables, buffers)
            trainer.step(None, TAU)
        torch.cuda.synchronize()
        timers.clear()
        for _ in range(steps):
            torch.cuda._sleep(40_000_000)
            for _ in range(4):
                with calib(0.0):
                    pass
            trainer.step(None, TAU)
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
        eng.overlap = was_overlap
        trainer.instrument = None
    return res


def cpu_baseline(seconds_budget=12.0):
    """The oracle (plain-torch restatement pinned to the reference by tests/golden) on this host's cores, with the one-GPU
    box's CPU share (min(affinity mask, 16) threads; the mask's size is stated in `sample`), median step of a bounded
    sample."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import rbvae_oracle as O
    try:
        affinity = len(os.sched_getaffinity(0))
    except AttributeError:
        affinity = os.cpu_count() or 1
    p = O.init_params("percep", C_IN, C_IN, LATENT, HW, seed=1234)
    for v in p.values():
        v.requires_grad_()
    g = torch.Generator().manual_seed(1234)
    Bc = 4                                           # bounded sample: 4 items = 64 frames per step
    item = torch.randn(Bc, 2, T_STATES, C_IN, *HW, generator=g)
    frames = Bc * 2 * T_STATES

    def run(cores, budget):
        torch.set_num_threads(cores)
        times, state, step = [], {}, 0
        t_start = time.perf_counter()
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
            if (len(times) >= 3 and time.perf_counter() - t_start > budget) or len(times) >= 400:
                break
        times = sorted(times[1:])                    # first step warms the allocator / oneDNN primitives
        return frames / times[len(times) // 2], len(times)      # median step

    share = max(1, min(affinity, 16))                # the one-GPU box's CPU share
    fps, n = run(share, seconds_budget)
    cpu_model = ""
    try:
        with open("/proc/cpuinfo") as f:
            cpu_model = next((l.split(":", 1)[1].strip() for l in f if l.startswith("model name")), "")
    except OSError:
        pass
    out = {"value": round(fps, 2), "unit": "frames/s", "cores": share, "kind": "port",
           "sample": f"median of {n} steps of {frames} frames (item [{Bc},2,{T_STATES},{C_IN},{HW[0]},{HW[1]}]), "
                     f"fwd+bwd+Adam, torch {torch.__version__} CPU, {share} threads, {cpu_model}; affinity mask: {affinity} cores"}
    # (A leg on every core of the affinity mask was tried in round 4: on a GPU box whose mask shows the whole host but whose
    # CPU quota is the 16-core share, torch's intra-op pool oversubscribed it so badly that the default run went silent for
    # 7 minutes.  The mask's size is stated above; the figure of record is the share's.)
    return out


def other_configs(dev):
    """The other configurations of BASELINE.json, measured in the SAME run on rank 0 (a few seconds each): frames/s,
    ms per step and the fraction of each one's own roof (SURVEY.md 8d / BASELINE.md 3).  A configuration that fails is
    reported with its error; the headline never depends on this leg."""
    import torch
    import sfv_amd as sfv

    def timed(fn, warm, it):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(it):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / it

    def entry(frames, dt, dtype, roofs, note):
        fps = frames / dt
        bound, roof = min(roofs.items(), key=lambda kv: kv[1])
        return {"frames_per_s": round(fps, 1), "ms_per_step": round(dt * 1e3, 4), "frames_per_step": frames, "dtype": dtype,
                "roof": {"bound": bound, "frames_per_s": round(roof, 1), "frac": round(fps / roof, 4),
                         "all": {k: round(v, 1) for k, v in roofs.items()}}, "workload": note}

    def train_cfg(variant, cin, hw, item_shape, dtype, latent, pair_loss=None, uniform=False, it=20):
        torch.manual_seed(0)
        m = sfv.Seq2SeqBinaryVAE(cin, cin, latent, latent, variant=variant, input_hw=hw, compute_dtype=dtype).to(dev).train()
        g = torch.Generator(device="cpu").manual_seed(1234)
        item = (torch.rand(*item_shape, generator=g) if uniform else torch.randn(*item_shape, generator=g)).to(dev)
        tr = sfv.FusedTrainer(m, alpha=ALPHA, beta_kl=BETA, bernoulli_p=BERN_P, noise_ratio=NOISE_R, device_noise=True,
                              use_graph=True, seed=1234, pair_loss=pair_loss)
        dt = timed(lambda: tr.step(item, TAU), 5, it)
        if not bool(torch.isfinite(tr.losses).all()):
            raise RuntimeError(f"non-finite losses {tr.losses.tolist()}")
        return dt

    out = {}

    def run(name, fn):
        try:
            out[name] = fn()
        except Exception as e:      # noqa: BLE001 -- reported, never fatal for the headline
            out[name] = {"error": f"{type(e).__name__}: {e}"[:300]}
        torch.cuda.empty_cache()

    # cfg 3: contrastive_RBVAE on raw 256x256 RGB, 128 frames per step.  2629.9 MFLOP, 62.92 MB (f32) per frame
    cfg3 = (8, 2, 8, 3, 256, 256)
    run("cfg3_contrastive_256x256_f32", lambda: entry(
        128, train_cfg("contrastive", 3, (256, 256), cfg3, "f32", 32, uniform=True, it=10), "f32",
        {"hbm": 8.0e12 / 62.92e6, "f32_compute": 157.3e12 / 2629.9e6}, "item [8,2,8,3,256,256] ~ U[0,1), latent 32, fused step"))
    run("cfg3_contrastive_256x256_bf16", lambda: entry(
        128, train_cfg("contrastive", 3, (256, 256), cfg3, "bf16", 32, uniform=True), "bf16",
        {"hbm": 8.0e12 / 31.46e6, "mfma": 2.5e15 / 2629.9e6}, "item [8,2,8,3,256,256] ~ U[0,1), latent 32, fused step"))
    # percep_RBVAE at the reference's native latent size 4x88x160: 8196.8 MFLOP, 49.57 MB (f32) per frame
    nat = (8, 2, 8, 4, 88, 160)
    run("native_percep_4x88x160_bf16", lambda: entry(
        128, train_cfg("percep", 4, (88, 160), nat, "bf16", 32), "bf16",
        {"hbm": 8.0e12 / 24.785e6, "mfma": 2.5e15 / 8196.8e6}, "item [8,2,8,4,88,160] ~ N(0,1), latent 32, fused step"))
    run("native_percep_4x88x160_f32", lambda: entry(
        128, train_cfg("percep", 4, (88, 160), nat, "f32", 32, it=10), "f32",
        {"hbm": 8.0e12 / 49.57e6, "f32_compute": 157.3e12 / 8196.8e6}, "item [8,2,8,4,88,160] ~ N(0,1), latent 32, fused step"))
    # the headline workload at the reference's largest swept latent size
    run("cfg2_latent100_bf16", lambda: entry(
        256, train_cfg("percep", C_IN, HW, (B_ITEMS, 2, T_STATES, C_IN, *HW), "bf16", 100, it=100), "bf16",
        {"hbm": ROOF_HBM_FPS, "mfma": ROOF_MFMA_FPS}, "headline item shape, latent_dim 100 (LSTM stacks of 100 units)"))

    # cfg 5 front end: the frozen LDM VAE encoder on 512x512 frames (1116.7 GFLOP per frame)
    def ldm():
        torch.manual_seed(0)
        enc = sfv.LDMEncoder(compute_dtype="bf16").to(dev)
        x = (torch.rand(4, 3, 512, 512, generator=torch.Generator().manual_seed(3)) * 2 - 1).to(dev)
        dt = timed(lambda: enc.encode(x, sample=False), 2, 5)
        return entry(4, dt, "bf16", {"mfma": 2.5e15 / 1116.7e9}, "LDMEncoder.encode, 4 frames of 3x512x512, posterior mode")
    run("ldm_encode_512x512_bf16", ldm)

    # cfg 5 composition: frames -> LDM encode on the fly -> percep-style RBVAE (4x64x64 latents) with the triplet term
    def cfg5():
        from importlib import import_module
        compose = import_module("symbols-from-video_amd.compose")
        torch.manual_seed(0)
        enc = sfv.LDMEncoder(compute_dtype="bf16").to(dev)
        m = sfv.Seq2SeqBinaryVAE(4, 4, 32, 32, variant="percep", input_hw=(64, 64), compute_dtype="bf16").to(dev).train()
        tr = sfv.FusedTrainer(m, alpha=ALPHA, beta_kl=BETA, bernoulli_p=BERN_P, device_noise=True, use_graph=True, seed=1234,
                              pair_loss="triplet", margin=0.2)
        pipe = compose.OnTheFlyLatentTrainer(enc, tr, frames_per_chunk=4)
        frames = (torch.rand(1, 2, 4, 3, 512, 512, generator=torch.Generator().manual_seed(4)) * 2 - 1).to(dev)
        dt = timed(lambda: pipe.step(frames, TAU, sample=False), 2, 4)
        if not bool(torch.isfinite(tr.losses).all()):
            raise RuntimeError(f"non-finite losses {tr.losses.tolist()}")
        # per frame: the encoder's 1116.7 GFLOP + the RBVAE part's 2384.7 MFLOP
        return entry(8, dt, "bf16", {"mfma": 2.5e15 / (1116.7e9 + 2384.7e6)},
                     "frames [1,2,4,3,512,512] -> LDM encode -> triplet-trained percep RBVAE on 4x64x64 latents, one step")
    run("cfg5_on_the_fly_512x512_bf16", cfg5)
    return out


def load_profile_json(name):
    try:
        with open(os.path.join(ROOT, "profiles", name)) as f:
            return json.load(f)
    except (OSError, ValueError):
        return {}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-roofline", action="store_true", help="skip the per-kernel event legs")
    ap.add_argument("--no-others", action="store_true", help="skip the other BASELINE configurations (rank 0, 1 GPU only)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus, sys.argv[1:]))          # no GPU call has happened in this process

    import random
    import torch
    import sfv_amd as sfv
    from importlib import import_module
    ddp = import_module("symbols-from-video_amd.ddp")
    trainer_mod = import_module("symbols-from-video_amd.trainer")
    data_mod = import_module("symbols-from-video_amd.data")
    # RCCL ("nccl") is the backend of record; RBVAE_DIST_BACKEND=gloo lets the N > 1 code path be rehearsed
    # with several ranks sharing one GPU (RCCL refuses duplicate devices)
    backend = os.environ.get("RBVAE_DIST_BACKEND", "nccl")
    ndev = max(torch.cuda.device_count(), 1)
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % ndev)
    rank, world, local = ddp.init_from_env(backend)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    local = local % ndev
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    torch.manual_seed(1234)                          # identical initial weights on every rank
    model = sfv.Seq2SeqBinaryVAE(C_IN, C_IN, LATENT, LATENT, variant="percep", input_hw=HW,
                                 compute_dtype=args.dtype).to(dev).train()
    ddp.broadcast_(model._flat)
    # per-rank shard of the data: its own synthetic latent table, pairs and shuffle (seed + rank)
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    table = torch.randn(T_STATES * FRAMES_PER_STATE, C_IN, *HW, generator=g)
    random.seed(1234 + rank)
    segs = [(s * FRAMES_PER_STATE, (s + 1) * FRAMES_PER_STATE) for s in range(T_STATES)]
    ds = data_mod.DeviceStatePairDataset(table, segs, mode="train", device=dev)
    plan = ds.plan(torch.randperm(len(ds), generator=g), B_ITEMS)
    tr = trainer_mod.FusedTrainer(model, lr=1e-3, alpha=ALPHA, beta_kl=BETA, bernoulli_p=BERN_P, noise_ratio=NOISE_R,
                                  device_noise=True, use_graph=not args.no_graph, seed=1234)
    tr.set_data(ds.table, plan)
    frames_per_step = B_ITEMS * 2 * T_STATES

    for _ in range(args.warmup):
        tr.step(None, TAU)
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        tr.step(None, TAU)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], device=dev)
    if world > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    dt = float(t.item())
    # per-step times of a second, untimed-by-the-contract run of up to 200 steps: one event behind every step, no host
    # synchronisation in between (SURVEY.md 8d quotes the median; `value` stays the mean over exactly --steps steps)
    nmed = min(args.steps, 200)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(nmed + 1)]
    evs[0].record()
    for i in range(nmed):
        tr.step(None, TAU)
        evs[i + 1].record()
    torch.cuda.synchronize()
    per_step = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(nmed))
    ms_median = per_step[len(per_step) // 2]
    losses = [float(v) for v in tr.losses.tolist()]
    graphs_captured = sum(len([g for g in gs if g is not None]) for gs in tr._graphs.values())
    if world == 1:
        schedule = "one HIP graph per step"
    elif not tr.ddp_overlap:
        schedule = "backward graph | all-reduce | update graph"
    elif tr.ddp_ingraph:
        schedule = "one HIP graph, both bucket all-reduces captured on a communication stream"
    else:
        schedule = "3 graphs: tail-bucket all-reduce (async) beside the encoder-CNN backward graph, head after it, update"
    dist_info = {"backend": (torch.distributed.get_backend() if world > 1 else None),
                 "dist_world_size": (torch.distributed.get_world_size() if world > 1 else 1), "schedule": schedule,
                 "devices_visible": torch.cuda.device_count()}

    roof = None
    roof_all = None
    cpu = None
    tname = "unsigned short" if args.dtype == "bf16" else "float"
    nleg = min(args.steps, 20)
    if not args.no_roofline:
        # every rank runs the instrumented legs (their steps contain the gradient all-reduce: a collective)
        legs = roofline_leg(tr, nleg, tname, overlap=True)
        legs_iso = roofline_leg(tr, nleg, tname, overlap=False)
    if rank == 0 and not args.no_roofline:
        peak = MFMA_BF16_PEAK_TFLOPS if args.dtype == "bf16" else 157.3
        traffic = load_profile_json("pmc_traffic.json")
        mfma_busy = load_profile_json("pmc_mfma_busy.json")
        prof_meta = load_profile_json("pmc_meta.json")
        roof_all = {}
        ev_us = legs.pop("_event_pair_overhead_us", 0.0)
        legs_iso.pop("_event_pair_overhead_us", None)
        for name, (ms, launches, flops) in sorted(legs.items(), key=lambda kv: -kv[1][0]):
            ach = flops / (ms * 1e-3) / 1e12
            e = {"achieved": round(ach, 2), "frac": round(ach / peak, 4), "launches": launches,
                 "avg_us": round(ms * 1e3 / max(launches, 1), 2), "us_per_step": round(ms * 1e3 / nleg, 1)}
            if name in legs_iso:
                ims, il, ifl = legs_iso[name]
                e["isolated"] = {"avg_us": round(ims * 1e3 / max(il, 1), 2),
                                 "frac": round(ifl / (ims * 1e-3) / 1e12 / peak, 4)}
            roof_all[name] = e
        dom = next(iter(roof_all))                       # the instance with the largest total time
        d = roof_all[dom]
        roof = {"bound": "mfma", "kernel": dom, "achieved": d["achieved"], "peak": peak, "unit": "TFLOP/s",
                "frac": d["frac"], "traffic": None, "launches": d["launches"], "avg_us": d["avg_us"],
                "us_per_step": d["us_per_step"], "isolated": d.get("isolated"),
                "timing": "HIP events on the launching stream, same multi-stream schedule as the captured graph "
                          "(side streams on); `isolated` = the kernel alone on one stream",
                "event_pair_overhead_us_subtracted": round(ev_us, 2),
                # counters cannot be read inside this process (rocprofv3 wraps it): the committed figures below come from
                # ANOTHER run of this bench (tools/refresh_profiles.sh: separate --pmc passes, FETCH_SIZE x 2 on gfx950) and
                # say which commit and box they were taken on; `traffic` above stays null in this line
                "profiles_ref": {"meta": prof_meta or None, "traffic": traffic.get(dom), "pmc": mfma_busy.get(dom),
                                 "files": ["profiles/pmc_traffic.json", "profiles/pmc_mfma_busy.json", "profiles/pmc_meta.json"]}}
    others = None
    if rank == 0 and world == 1 and not args.no_others:
        del tr, ds, plan, table                  # release the headline's buffers before the larger configurations
        torch.cuda.empty_cache()
        others = other_configs(dev)
    if rank == 0 and world == 1 and not args.no_cpu:
        cpu = cpu_baseline()
    if world > 1:
        torch.distributed.barrier()
    if rank == 0:
        value = frames_per_step * world * args.steps / dt
        hbm_roof, mfma_roof = (ROOF_HBM_FPS, ROOF_MFMA_FPS) if args.dtype == "bf16" else ROOF_F32
        per_gpu = value / world
        e2e = {"frames_per_s_per_gpu": round(per_gpu, 1), "hbm_roof_frames_per_s": round(hbm_roof, 0),
               "frac_hbm": round(per_gpu / hbm_roof, 4), "mfma_roof_frames_per_s": round(mfma_roof, 0),
               "frac_mfma": round(per_gpu / mfma_roof, 4),
               "basis": "SURVEY.md 8d: 596.3 MFLOP and 1.805 MB (bf16; 3.611 MB f32) per frame fwd+bwd; 8.0 TB/s, "
                        "2.5 PFLOP/s bf16 (157.3 TFLOP/s f32)"}
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import ceiling as ceiling_mod
            c = ceiling_mod.ceiling()
            step_us = dt / args.steps * 1e6
            c["achieved_over_critical_path"] = round(c["critical_path_us"] / step_us, 4)
            c["achieved_over_resource_sum"] = round(c["resource_sum_us"] / step_us, 4)
            e2e["ceiling"] = c
        except Exception as e:      # noqa: BLE001 -- the model is documentation, never fatal
            e2e["ceiling"] = {"error": str(e)[:200]}
        if roof is not None:
            roof["e2e"] = e2e
        else:
            roof = {"e2e": e2e}
        line = {"metric": "frames/sec (enc+binarise+dec fwd+bwd), batch 256x256, 1/2/4/8 MI355X",
                "value": round(value, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
                "ms_per_step_median": round(ms_median, 4), "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
                "config": {"workload": "percep_RBVAE fused train step (batch gather from the HBM-resident latent table "
                                       "+ fwd+bwd+losses+Adam), item [16,2,8,4,32,32] per GPU = 256 frames/step/GPU "
                                       "(latents of 256x256 frames), latent 32, 4-layer LSTMs, dropout on",
                           "frames_per_step_per_gpu": frames_per_step, "global_frames_per_step": frames_per_step * world,
                           "parallelism": f"dp{world}", "graph": not args.no_graph, **dist_info,
                           "graphs_captured": graphs_captured, "launcher": "torch.distributed.run or self-spawned ranks",
                           "last_losses": {"total": losses[0], "recon": losses[1], "kl": losses[2], "pair": losses[3]}},
                "roofline": roof, "roofline_all_mfma_kernels": roof_all, "cpu_baseline": cpu, "other_configs": others}
        print(json.dumps(line))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
