"""The data-parallel fused trainer on the GPU: two ranks share the one device over gloo (RCCL refuses duplicate
devices; the collective calls, streams and graphs are the product path's).  The overlapped schedule (backward cut
where the decoder / LSTM gradients are final, tail all-reduced beside the encoder CNN's backward graph) must give
bit-identical parameters to the one-all-reduce schedule, and both must track a single process stepping on the
global batch."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(rank, world, port, overlap, q, split=0, backend="gloo", ingraph=0):
    sys.path.insert(0, ROOT)
    own_gpu = backend == "nccl"                                 # RCCL: one device per rank; gloo: the ranks share device 0
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank if own_gpu else 0), RBVAE_DDP_OVERLAP=str(overlap),
                      RBVAE_DDP_SPLIT_UPDATE=str(split), RBVAE_DDP_INGRAPH=str(ingraph), HSA_ENABLE_IPC_MODE_LEGACY="0")
    from importlib import import_module
    import sfv_amd as sfv
    ddp = import_module("symbols-from-video_amd.ddp")
    trainer_mod = import_module("symbols-from-video_amd.trainer")
    torch.cuda.set_device(rank if own_gpu else 0)
    dev = torch.device("cuda", rank if own_gpu else 0)
    if world > 1:
        ddp.init_from_env(backend)
    torch.manual_seed(11)                                        # same initial weights on every rank
    Ld = 32
    model = sfv.Seq2SeqBinaryVAE(4, 4, Ld, Ld, variant="percep", input_hw=(16, 16), compute_dtype="bf16").to(dev).train()
    g = torch.Generator().manual_seed(5)
    Bg, T = 4, 4
    items = [torch.rand(Bg, 2, T, 4, 16, 16, generator=g) for _ in range(3)]
    Us = [torch.rand(2, Bg, T, Ld, generator=g) for _ in range(3)]
    # dropout off (eval-mode masks are rank-local draws otherwise); host-supplied noise so that shards see the global draw
    model.eval()
    tr = trainer_mod.FusedTrainer(model, lr=1e-3, alpha=1.0, beta_kl=1.0, bernoulli_p=0.1, noise_ratio=0.1,
                                  device_noise=False, use_graph=True,
                                  process_group=torch.distributed.group.WORLD if world > 1 else None)
    mine = ddp.shard_items(Bg, rank, world) if world > 1 else slice(0, Bg)
    for it, U in zip(items, Us):
        Ui = U[:, mine].reshape(2, -1, Ld).contiguous().to(dev)
        tr.step(it[mine].contiguous().to(dev), 0.7, U=Ui)
    torch.cuda.synchronize()
    if rank == 0:
        q.put((len(next(iter(tr._graphs.values()))), model._flat.detach().cpu().numpy().copy()))   # by value
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


def _launch(world, overlap, split=0, backend="gloo", ingraph=0):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_run, args=(r, world, port, overlap, q, split, backend, ingraph)) for r in range(world)]
    for p in procs:
        p.start()
    import queue
    import time
    out, t0 = None, time.time()
    while out is None:
        try:
            out = q.get(timeout=2)
        except queue.Empty:
            if any(p.exitcode not in (None, 0) for p in procs) or time.time() - t0 > 240:
                for p in procs:
                    if p.is_alive():
                        p.terminate()
                raise AssertionError("a rank died or timed out")
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return out[0], torch.from_numpy(out[1])


@pytest.mark.timeout(600)
def test_ddp_overlap_matches_single_allreduce():
    n_graphs_a, flat_a = _launch(2, 1)
    n_graphs_b, flat_b = _launch(2, 0)
    n_graphs_c, flat_c = _launch(2, 1, split=1)
    # overlapped: fwd+bwd to the cut | rest of bwd | update; with RBVAE_DDP_SPLIT_UPDATE=1: update(tail) | update(head)
    assert n_graphs_a == 3 and n_graphs_b == 2 and n_graphs_c == 4
    assert torch.equal(flat_a, flat_b) and torch.equal(flat_c, flat_b)
    # sanity against one process on the global batch (per-rank losses are shard means, their average is the global mean);
    # Adam turns bf16 accumulation-order noise on near-zero gradients into +-lr steps, hence the loose bound -- the
    # reduction semantics themselves are pinned by tests/test_ddp_cpu.py
    _, flat_1 = _launch(1, 0)
    assert torch.isfinite(flat_a).all()
    rel = float((flat_a - flat_1).norm() / flat_1.norm())
    assert rel < 2e-2, rel


@pytest.mark.timeout(900)
@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: RCCL refuses two ranks on one device")
def test_ddp_overlap_matches_single_allreduce_over_rccl():
    """The same on RCCL with one GPU per rank (skipped on a one-GPU box: there the schedules have only run over gloo with
    the ranks sharing the device, and with a one-rank RCCL group -- INTEGRATION.md says so): the asynchronous tail
    bucket beside the encoder CNN's backward graph (3 graphs), the split update (4) and the in-graph collectives (1)
    must leave the parameters of the one-all-reduce schedule (2 graphs), bit for bit."""
    n_b, flat_b = _launch(2, 0, backend="nccl")
    n_a, flat_a = _launch(2, 1, backend="nccl")
    n_c, flat_c = _launch(2, 1, split=1, backend="nccl")
    n_d, flat_d = _launch(2, 1, backend="nccl", ingraph=1)
    assert (n_b, n_a, n_c) == (2, 3, 4) and n_d in (1, 2)
    assert torch.equal(flat_a, flat_b) and torch.equal(flat_c, flat_b) and torch.equal(flat_d, flat_b)
    _, flat_1 = _launch(1, 0)
    assert float((flat_a - flat_1).norm() / flat_1.norm()) < 2e-2


@pytest.mark.timeout(600)
def test_bench_gpus2_self_launches():
    """`python bench.py --gpus 2` typed as is (no torch.distributed.run): the parent spawns the two ranks before making
    any GPU call, the ranks run the data-parallel step (gloo here: two ranks share the one GPU), rank 0's JSON line
    comes back through the parent."""
    import json
    import subprocess
    env = dict(os.environ, RBVAE_DIST_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--no-cpu", "--no-roofline"], env=env, capture_output=True, text=True, timeout=540)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["config"]["global_frames_per_step"] == 512 and line["value"] > 0
    assert line["config"]["graphs_captured"] >= 1
    assert line["config"]["backend"] == "gloo" and line["config"]["dist_world_size"] == 2 and "all-reduce" in line["config"]["schedule"]
    assert line["other_configs"] is None and line["cpu_baseline"] is None        # rank 0 of a 1-GPU run only
    assert 0 < line["roofline"]["e2e"]["frac_hbm"] < 1
    assert all(abs(v) < 1e4 for v in line["config"]["last_losses"].values())


@pytest.mark.timeout(900)
@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: RCCL refuses two ranks on one device")
@pytest.mark.parametrize("bf16_wire", [0, 1])
def test_bench_gpus2_over_rccl(bf16_wire):
    """`python bench.py --gpus 2` on the backend of record (RCCL over xGMI, one GPU per rank; skipped on a one-GPU box,
    where the gloo twin above is what runs): the driver's N = 2 scaling point, with the f32 and the bf16 gradient buckets."""
    import json
    import subprocess
    env = dict(os.environ, RBVAE_DDP_BF16=str(bf16_wire), HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "RBVAE_DIST_BACKEND"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5",
                        "--no-cpu"], env=env, capture_output=True, text=True, timeout=840)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["config"]["backend"] == "nccl" and line["config"]["dist_world_size"] == 2
    assert line["config"]["global_frames_per_step"] == 512 and line["value"] > 0
    assert all(abs(v) < 1e4 for v in line["config"]["last_losses"].values())


@pytest.mark.timeout(600)
def test_in_graph_collectives_with_one_rank_rccl():
    """RBVAE_DDP_INGRAPH=1: the two all-reduces captured INTO the step's single HIP graph on a communication stream.
    Rehearsed the only way one GPU allows: a one-rank "nccl" (RCCL) group whose collectives are really issued
    (GradReducer(force=True)); every schedule -- 2 / 3 / 4 graphs, and one graph with the collectives inside, with and
    without the split update -- must leave bit-identical parameters (tools/rccl_single.py)."""
    import subprocess
    env = dict(os.environ, RCCL_SINGLE_STEPS="20", MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rccl_single.py")], env=env, capture_output=True, text=True,
                       timeout=500)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-3000:]
    assert "parameters after 5 steps identical (f32 buckets): True" in out, out[-3000:]
    # in-graph collectives: two f32 schedules + the bf16 wire; the bf16 wire's losses stay within its rounding of the f32 ones
    assert out.count("1 graphs + RCCL") == 3 and "4 graphs + RCCL" in out and "3 graphs + RCCL" in out, out[-3000:]
    assert out.count("bf16_wire=1") == 2, out[-3000:]
