import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sfv_amd as sfv
L_ = sfv._lib
S, T, L, layers = 32, 8, 32, 4
def timeit(fn, iters=50):
    for _ in range(5): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
wblk = torch.randn(layers * (8 * L * L + 8 * L), device="cuda") * 0.1
hs = torch.randn(layers + 1, S, T, L, device="cuda")
hp = torch.empty(layers, S, T, L, device="cuda"); cs = torch.empty_like(hp); acts = torch.empty(layers, S, T, 4 * L, device="cuda")
dG = torch.randn(layers, S, T, 4 * L, device="cuda"); dx = torch.empty(S, T, L, device="cuda"); gt = torch.randn(S, T, L, device="cuda")
gb = torch.empty_like(wblk)
wT = torch.randn(layers, 2, L, 4 * L, device="cuda") * 0.1
print("lstm_fwd  ", timeit(lambda: L_.call("rbvae_lstm_fwd", wblk, None, hs, hp, acts, cs, S, T, L, layers)))
print("lstm_bwd  ", timeit(lambda: L_.call("rbvae_lstm_bwd", wblk, None, acts, cs, gt, dG, dx, S, T, L, layers)))
print("lstm_wgrad", timeit(lambda: L_.call("rbvae_lstm_wgrad", dG, hs, hp, gb, S, T, L, layers, 0)))
for S2 in (1, 8, 128):
    hs2 = torch.randn(layers + 1, S2, T, L, device="cuda"); hp2 = torch.empty(layers, S2, T, L, device="cuda"); cs2 = torch.empty_like(hp2); ac2 = torch.empty(layers, S2, T, 4 * L, device="cuda")
    dG2 = torch.randn(layers, S2, T, 4 * L, device="cuda")
    print("S", S2, "fwd", timeit(lambda: L_.call("rbvae_lstm_fwd", wblk, wT, hs2, hp2, ac2, cs2, S2, T, L, layers)),
          "wgrad", timeit(lambda: L_.call("rbvae_lstm_wgrad", dG2, hs2, hp2, gb, S2, T, L, layers, 0)))
