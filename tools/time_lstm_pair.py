"""rbvae_lstm_pair_fwd / rbvae_lstm_pair_bwd alone at the bench geometry (S sequences, L = 32, 4 + 4 layers) for several T: us per launch with one
wave per layer (lstm_pair_fwd_unit_k) and one thread per gate row (lstm_pair_fwd_k); the slope over T is the cost of a
diagonal, the intercept the prologue (weights into registers, input staging)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sfv_amd as sfv
Lb = sfv._lib
dbg = Lb.dbg_lib()
S, L, layers = 32, 32, 4
def timeit(fn, iters=100):
    for _ in range(5): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
per = layers * (8 * L * L + 8 * L)
we, wd = torch.randn(per, device="cuda") * 0.3, torch.randn(per, device="cuda") * 0.3
wTe, wTd = torch.randn(layers, 2, L, 4 * L, device="cuda") * 0.3, torch.randn(layers, 2, L, 4 * L, device="cuda") * 0.3
seed_dev = torch.tensor([7], dtype=torch.int64, device="cuda")
for T in (1, 4, 8, 16, 24):
    N = S * T
    bufs = []
    for _ in range(2):
        bufs += [torch.zeros(layers + 1, S, T, L, device="cuda"), torch.empty(layers, S, T, L, device="cuda"),
                 torch.empty(layers, S, T, 4 * L, device="cuda"), torch.empty(layers, S, T, L, device="cuda")]
    slabs = torch.randn(4, S, T, L, device="cuda")
    y = torch.empty(N, L, device="cuda"); parts = torch.empty(S, device="cuda")
    pad = torch.zeros(N, 64, dtype=torch.bfloat16, device="cuda")
    def run(): Lb.call("rbvae_lstm_pair_fwd", we, wTe, wd, wTd, *bufs, slabs, 4, S * T * L, None, y, parts, 0.6, None, 0.3, 1e-8, 0,
                       0.1, 1e-8, 1, 1234, seed_dev, pad, 1, 64, S, T, L, layers)
    res = []
    for unit in (1, 0):
        dbg.rbvae_dbg_lstm_unit_threads(unit)
        res.append(timeit(run))
    dbg.rbvae_dbg_lstm_unit_threads(1)
    ae, ce = torch.rand(layers, S, T, 4 * L, device="cuda"), torch.randn(layers, S, T, L, device="cuda")
    ad, cd = torch.rand(layers, S, T, 4 * L, device="cuda"), torch.randn(layers, S, T, L, device="cuda")
    gparts = torch.randn(4, N, L, device="cuda")
    yy, zz = torch.rand(N, L, device="cuda").clamp(1e-3, 1 - 1e-3), torch.rand(N, L, device="cuda").clamp(1e-3, 1 - 1e-3)
    tau = torch.tensor([0.7], device="cuda")
    dGe, dGd = torch.empty(layers, S, T, 4 * L, device="cuda"), torch.empty(layers, S, T, 4 * L, device="cuda")
    dx, sums = torch.empty(N, L, device="cuda"), torch.empty(S, L, device="cuda")
    def runb(): Lb.call("rbvae_lstm_pair_bwd", we, wd, ae, ce, ad, cd, gparts, 4, N * L, None, yy, zz, None, 9.0, tau, 1.0, 0.1, 1e-8, 1,
                        dGe, dGd, dx, None, pad, 1, 64, sums, S, T, L, layers)
    for unit in (1, 0):
        dbg.rbvae_dbg_lstm_unit_threads(unit)
        res.append(timeit(runb) if Lb.query("rbvae_lstm_pair_bwd_ok", T, L, layers) else float("nan"))
    dbg.rbvae_dbg_lstm_unit_threads(1)
    print(f"T={T:3d} ({T + 2 * layers - 1:2d} diagonals)  forward: wave per layer {res[0]:6.1f} us, thread per gate row {res[1]:6.1f} us   "
          f"backward: {res[2]:6.1f} us, {res[3]:6.1f} us", flush=True)
