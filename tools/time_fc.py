"""rbvae_fc_gemm and the column-sum pass alone at a wide fc (default: 128 frames x 64 -> 56 320 features, native 4x88x160)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sfv_amd as sfv
L = sfv._lib
M, N = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (128, 56320)
A = torch.randn(M, 64, device="cuda").bfloat16(); W = (torch.randn(N, 64, device="cuda") / 8).bfloat16()
b = torch.randn(N, device="cuda"); out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
ws = torch.empty(-(-M // 128), N, device="cuda")
def timed(f, it=50):
    for _ in range(3): f()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(it): f()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / it * 1e3
by = (M * N + N * 64) * 2
t = timed(lambda: L.call("rbvae_fc_gemm", 1, A, W, out, b, None, M, 64, N, 64, N)); print(f"fc_gemm {M} x 64 -> {N}: {t:6.1f} us  {by / t / 1e6:.2f} TB/s")
t = timed(lambda: L.call("rbvae_fc_gemm", 1, A, W, out, b, ws, M, 64, N, 64, N)); print(f"  with column sums:       {t:6.1f} us")
nws = L.query("rbvae_colsum_ws_floats", M, N) if hasattr(L.lib(), "rbvae_colsum_ws_floats") else 0
