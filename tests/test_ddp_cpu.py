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
