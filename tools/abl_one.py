"""Timing of the single-slice (K = 64) gather GEMM on the conv1-forward shape (M = 65536, N = 256) with the
epilogue features switched one at a time, next to plain device copies of the same byte counts."""
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


def run(M, K, N, drop, relu, bias, gate):
    A = torch.randn(M, K, device="cuda").bfloat16()
    W = torch.randn(N, K, device="cuda").bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    b = torch.randn(N, device="cuda") if bias else None
    g = torch.randn(M, N, device="cuda").bfloat16() if gate else None
    args = (1, A, W, out, b, g, None, None, zero, M, 1, 1, 1, 1, 1, 1, 1, 1, K, N, K, N, 1, 1, ctypes.addressof(ONE),
            relu, drop, 0.2, 1.25, 3, None, None)
    return timeit(lambda: L.call("rbvae_gather_gemm", *args))


M, K, N = 65536, 64, 256
print(f"GG_ONE={os.environ.get('RBVAE_GG_ONE', 'default')}")
for name, kw in (("plain", dict(drop=0, relu=0, bias=False, gate=False)),
                 ("bias+relu", dict(drop=0, relu=1, bias=True, gate=False)),
                 ("bias+relu+dropout", dict(drop=1, relu=1, bias=True, gate=False)),
                 ("gate", dict(drop=0, relu=0, bias=False, gate=True))):
    print(f"  M={M} K={K} N={N} {name:20s} {run(M, K, N, **kw):7.1f} us", flush=True)
print(f"  M={M // 2} plain {run(M // 2, K, N, 0, 0, False, False):7.1f} us;  N=128 plain {run(M, K, 128, 0, 0, False, False):7.1f} us;"
      f"  K=128 plain {run(M, 128, N, 0, 0, False, False):7.1f} us")
src = torch.empty(M * N, device="cuda", dtype=torch.bfloat16)
dst = torch.empty_like(src)
print(f"  copy 33.5 MB -> 33.5 MB: {timeit(lambda: dst.copy_(src)):7.1f} us;  fill 33.5 MB: {timeit(lambda: dst.zero_()):7.1f} us")
