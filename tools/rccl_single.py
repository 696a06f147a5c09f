"""One-rank RCCL rehearsal of the multi-GPU step: an initialised "nccl" process group (communicator, watchdog
thread) while the trainer captures its two HIP graphs, and a real all_reduce call on the flat gradient buffer
between their replays.  (The N > 1 path proper needs N GPUs; this covers its runtime interplay on one.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
import torch
import torch.distributed as dist
import sfv_amd as sfv
from importlib import import_module
FusedTrainer = import_module("symbols-from-video_amd.trainer").FusedTrainer

NSTEPS = int(os.environ.get("RCCL_SINGLE_STEPS", "200"))
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
item = torch.randn(16, 2, 8, 4, 32, 32, device="cuda")
flats = {}
GradReducer = import_module("symbols-from-video_amd.ddp").GradReducer
# (overlap, split_update, in_graph, bf16 buckets on the wire)
for overlap, split, ingraph, bf16 in ((0, 0, 0, 0), (1, 0, 0, 0), (1, 1, 0, 0), (1, 0, 1, 0), (1, 1, 1, 0), (1, 0, 0, 1), (1, 0, 1, 1)):
    torch.manual_seed(1)
    m = sfv.Seq2SeqBinaryVAE(4, 4, 32, 32, variant="percep", input_hw=(32, 32), compute_dtype="bf16").cuda().train()
    tr = FusedTrainer(m, alpha=1.0, beta_kl=1.0, bernoulli_p=0.1)
    tr.world = 2                       # take the multi-rank path (1/world in Adam: both runs alike)
    tr.one_graph = False
    tr.ddp_overlap = bool(overlap)     # 0: two graphs, one all_reduce; 1: three / four graphs, tail all_reduce beside the second
    tr.ddp_split_update = bool(split)  # 1: the tail bucket's update while the head's all_reduce is in flight (four graphs)
    tr.ddp_ingraph = bool(ingraph)     # 1: the collectives captured into ONE graph, on a communication stream
    # a reducer that issues its collectives although the group has one rank
    tr._red = GradReducer(tr.gflat, m._layout.offsets["decoder_cnn.fc.weight"], None, force=True,
                          wire_dtype=torch.bfloat16 if bf16 else torch.float32)
    for _ in range(5):
        tr.step(item, 0.7)
    torch.cuda.synchronize()
    if not bf16:
        flats[(overlap, split, ingraph)] = m._flat.detach().clone()
    t0 = time.perf_counter()
    for _ in range(NSTEPS):
        tr.step(item, 0.7)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / NSTEPS
    ng = len([g for g in next(iter(tr._graphs.values())) if g is not None])
    print(f"{ng} graphs + RCCL all_reduce (1 rank), overlap={overlap} split_update={split} in_graph={ingraph} "
          f"bf16_wire={bf16}: {dt * 1e3:.3f} ms/step, "
          f"losses {[round(v, 4) for v in tr.losses.tolist()]}", flush=True)
print("parameters after 5 steps identical (f32 buckets):", all(torch.equal(flats[(0, 0, 0)], f) for f in flats.values()))
dist.destroy_process_group()
