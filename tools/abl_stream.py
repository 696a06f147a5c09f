"""rbvae_stream_gemm against rbvae_gather_gemm on the two bench shapes it serves (M = 65536, K = 64, N = 256):
conv1 forward (bias + relu + dropout) and the last deconv's input gradient (gate + column sums)."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sfv_amd as sfv
L = sfv._lib
zero = torch.zeros(256, dtype=torch.uint8, device="cuda")
ONE = (ctypes.c_int * 6)(1, 0, 0, 0, 0, 0)


def timeit(fn, iters=50):
    for _ in range(5):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


M, K, N = 65536, 64, 256
A = torch.randn(M, K, device="cuda").bfloat16(); W = torch.randn(N, K, device="cuda").bfloat16()
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
b = torch.randn(N, device="cuda"); g = torch.randn(M, N, device="cuda").bfloat16()
ws = torch.zeros(1024 * N, device="cuda")
tag = f"groups/wave={os.environ.get('RBVAE_SG_GROUPS', '2')} cap={os.environ.get('RBVAE_SG_CAP', '512')}"
for name, bias, gate, relu, drop, cs in (("conv1 fwd", b, None, 1, 1, None), ("dd2 bwd", None, g, 0, 0, ws), ("plain", None, None, 0, 0, None)):
    t_g = timeit(lambda: L.call("rbvae_gather_gemm", 1, A, W, out, bias, gate, None, None, zero, M, 1, 1, 1, 1, 1, 1, 1, 1, K, N, K, N, 1, 1,
                                ctypes.addressof(ONE), relu, drop, 0.2, 1.25, 3, None, cs))
    t_s = timeit(lambda: L.call("rbvae_stream_gemm", A, W, out, bias, gate, M, N, N, relu, drop, 0.2, 1.25, 3, None, cs))
    print(f"{tag}  {name:10s} gather {t_g:6.1f} us   stream {t_s:6.1f} us", flush=True)
