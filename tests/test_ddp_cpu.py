"""N > 1 path on the CPU (gloo, world_size 2): item sharding + the flat-gradient all-reduce give the
single-process gradient of the concatenated batch.  The oracle stands in for the per-rank compute
(tests only); the collective plumbing is the product code in symbols-from-video_amd/ddp.py."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import importlib
    import rbvae_oracle as O
    ddp = importlib.import_module("symbols-from-video_amd.ddp")
    eng = importlib.import_module("symbols-from-video_amd.engine")
    torch.set_num_threads(2)
    r, w, _ = ddp.init_from_env("gloo")
    assert (r, w) == (rank, world)
    variant, Ld, hw, Bg, T = "contrastive", 16, (16, 16), 4, 3
    lay = eng.ParamLayout(eng.VARIANTS[variant], 3, 3, Ld, hw)
    # rank 0's parameters win
    p = O.init_params(variant, 3, 3, Ld, hw, seed=100 + rank)
    flat = torch.zeros(lay.total)
    for k in lay.names:
        lay.view(flat, k).copy_(p[k])
    ddp.broadcast_(flat)
    p = {k: lay.view(flat, k).clone().requires_grad_() for k in lay.names}
    g = torch.Generator().manual_seed(7)                      # every rank draws the same GLOBAL batch
    item = torch.rand(Bg, 2, T, 3, *hw, generator=g)
    U = torch.rand(2, Bg, T, Ld, generator=g)
    mine = ddp.shard_items(Bg, rank, world)
    res = O.step_losses(variant, p, item[mine], [U[0, mine].reshape(-1, Ld), U[1, mine].reshape(-1, Ld)], 0.7, 0.1,
                        0.1, 1.0, 1.0)
    res["total"].backward()
    gflat = torch.zeros(lay.total)
    for k in lay.names:
        lay.view(gflat, k).copy_(p[k].grad)
    # the product schedule: two buckets cut where the backward pass finishes them (tail first, asynchronously), and the
    # one-all-reduce form -- both must equal the plain mean
    g_b, g_1 = gflat.clone(), gflat.clone()
    red = ddp.GradReducer(g_b, lay.offsets["decoder_cnn.fc.weight"])
    assert red.world == world and red.tail.numel() + red.head.numel() == lay.total
    wt = red.start_tail()
    wh = red.start_head()
    red.wait(wt)
    red.wait(wh)
    ddp.GradReducer(g_1, lay.offsets["decoder_cnn.fc.weight"]).reduce_all()
    ddp.allreduce_mean_(gflat)
    assert torch.equal(g_b / world, gflat) and torch.equal(g_1 / world, gflat)
    # bf16 on the wire (RBVAE_DDP_BF16=1 / wire_dtype): half the bytes; the summed gradient within bf16 rounding of the f32
    # exchange, tensor by tensor, and identical on both ranks (asynchronous buckets and the one-all-reduce form)
    g_h, g_h1 = torch.zeros(lay.total), torch.zeros(lay.total)
    for k in lay.names:
        lay.view(g_h, k).copy_(p[k].grad)
    g_h1.copy_(g_h)
    redh = ddp.GradReducer(g_h, lay.offsets["decoder_cnn.fc.weight"], wire_dtype=torch.bfloat16)
    wt, wh = redh.start_tail(), redh.start_head()
    redh.wait(wt)
    redh.wait(wh)
    ddp.GradReducer(g_h1, lay.offsets["decoder_cnn.fc.weight"], wire_dtype=torch.bfloat16).reduce_all()
    assert torch.equal(g_h, g_h1)
    for k in lay.names:
        a, b = lay.view(g_h, k).double() / world, lay.view(gflat, k).double()
        assert float((a - b).norm()) <= 6e-3 * max(float(b.norm()), 1e-12), k
    both = [torch.zeros_like(g_h) for _ in range(world)]
    dist.all_gather(both, g_h)
    assert all(torch.equal(both[0], b_) for b_ in both)
    if rank == 0:
        q.put((flat.clone(), gflat.clone()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gradient_equals_global_batch_gradient():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import importlib
    import rbvae_oracle as O
    eng = importlib.import_module("symbols-from-video_amd.engine")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p_ in procs:
        p_.start()
    flat, gflat = q.get(timeout=240)
    for p_ in procs:
        p_.join(timeout=60)
        assert p_.exitcode == 0
    variant, Ld, hw, Bg, T = "contrastive", 16, (16, 16), 4, 3
    lay = eng.ParamLayout(eng.VARIANTS[variant], 3, 3, Ld, hw)
    p = {k: lay.view(flat, k).clone().requires_grad_() for k in lay.names}
    ref0 = O.init_params(variant, 3, 3, Ld, hw, seed=100)
    assert all(torch.equal(p[k].detach(), ref0[k]) for k in lay.names)          # broadcast from rank 0
    g = torch.Generator().manual_seed(7)
    item = torch.rand(Bg, 2, T, 3, *hw, generator=g)
    U = torch.rand(2, Bg, T, Ld, generator=g)
    res = O.step_losses(variant, p, item, [U[0].reshape(-1, Ld), U[1].reshape(-1, Ld)], 0.7, 0.1, 0.1, 1.0, 1.0)
    res["total"].backward()
    for k in lay.names:
        a, b = lay.view(gflat, k).double(), p[k].grad.double()
        assert float((a - b).norm()) <= 1e-5 * max(float(b.norm()), 1e-9), k


def test_shard_items():
    import importlib
    sys.path.insert(0, ROOT)
    ddp = importlib.import_module("symbols-from-video_amd.ddp")
    assert ddp.shard_items(8, 1, 4) == [1, 5]
    assert sorted(sum((ddp.shard_items(16, r, 8) for r in range(8)), [])) == list(range(16))
    with pytest.raises(ValueError):
        ddp.shard_items(10, 0, 4)


def _bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("rbvae_bench", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_self_launch_kills_the_survivors_when_a_rank_dies(capfd):
    """bench.self_launch: rank 1 exits with code 3 before rendezvous while rank 0 would wait for it forever; the
    parent must come back non-zero within seconds, with no rank left behind, and relay what rank 0 printed."""
    import time
    bench = _bench()
    stub = ("import os, sys, time\n"
            "r = int(os.environ['RANK'])\n"
            "assert os.environ['WORLD_SIZE'] == '2' and os.environ['MASTER_ADDR'] == '127.0.0.1'\n"
            "print('hello from rank', r, flush=True)\n"
            "if r == 1: sys.exit(3)\n"
            "open(os.environ['PIDFILE'], 'w').write(str(os.getpid()))\n"
            "time.sleep(600)\n")
    pidfile = os.path.join(ROOT, ".pytest_cache", f"rank0_{os.getpid()}.pid")
    os.makedirs(os.path.dirname(pidfile), exist_ok=True)
    os.environ["PIDFILE"] = pidfile
    try:
        t0 = time.monotonic()
        code = bench.self_launch(2, [], cmd=[sys.executable, "-c", stub])
        dt = time.monotonic() - t0
    finally:
        os.environ.pop("PIDFILE", None)
    assert code == 3 and dt < 20.0
    out = capfd.readouterr()
    assert "hello from rank 0" in out.out and "rank 1 exited with code 3" in out.err
    pid = int(open(pidfile).read())
    os.remove(pidfile)
    with pytest.raises(ProcessLookupError):
        os.kill(pid, 0)                      # rank 0 is gone


def test_self_launch_deadline_and_clean_exit(capfd):
    import time
    bench = _bench()
    t0 = time.monotonic()
    code = bench.self_launch(2, [], cmd=[sys.executable, "-c", "import time; time.sleep(600)"], deadline_s=1.0)
    assert code == 124 and time.monotonic() - t0 < 15.0
    assert bench.self_launch(3, [], cmd=[sys.executable, "-c", "import os; print(os.environ['RANK'])"]) == 0
    assert capfd.readouterr().out.strip().splitlines()[-1] == "0"          # only rank 0's stdout is relayed


def test_process_group_timeout_is_set():
    """ddp.init_from_env passes a finite timeout to init_process_group (a missing rank is an error, not a hang)."""
    import importlib
    import datetime
    ddp = importlib.import_module("symbols-from-video_amd.ddp")
    seen = {}
    real = dist.init_process_group

    def fake(**kw):
        seen.update(kw)

    env = dict(os.environ)
    os.environ.update(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", RBVAE_DIST_TIMEOUT="7")
    dist.init_process_group = fake
    try:
        ddp.init_from_env("gloo")
    finally:
        dist.init_process_group = real
        os.environ.clear()
        os.environ.update(env)
    assert seen["timeout"] == datetime.timedelta(seconds=7) and seen["world_size"] == 2
