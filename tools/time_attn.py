"""Mid-block attention at 512x512 / 256x256 frames: the online-softmax kernel against the three-launch form."""
import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sfv_amd as sfv
L = sfv._lib
C = 512
one = (ctypes.c_int * 6)(1, 0, 0, 0, 0, 0)
zero = torch.zeros(256, dtype=torch.uint8, device="cuda")
def gemm(A, W, out, rows, kc, nout, lda, ldo, scale=1.0):
    L.call("rbvae_gather_gemm", 1, A, W, out, None, None, None, None, zero, rows, 1, 1, 1, 1, 1, 1, 1, 1, kc, nout, lda, ldo, 1, 1,
           ctypes.addressof(one), 0, 0, 0.0, float(scale), 0, None, None)
def timeit(fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e6
for N, hw in ((8, 1024), (4, 4096)):
    qkv = (torch.randn(N * hw, 3 * C, device="cuda") * 1.5).bfloat16()
    o = torch.empty(N * hw, C, dtype=torch.bfloat16, device="cuda")
    t_flash = timeit(lambda: L.call("rbvae_attention", 1, qkv, qkv[:, C:], qkv[:, 2 * C:], o, N, hw, C, 3 * C, 3 * C, 3 * C, C, C ** -0.5))
    q, k, v = (qkv[:, i * C:(i + 1) * C].contiguous() for i in range(3))
    s = torch.empty(hw, hw, dtype=torch.bfloat16, device="cuda"); vt = torch.empty(C, hw, dtype=torch.bfloat16, device="cuda")
    o2 = torch.empty_like(o)
    def old():
        for n in range(N):
            qn, kn, vn = q[n * hw:(n + 1) * hw], k[n * hw:(n + 1) * hw], v[n * hw:(n + 1) * hw]
            gemm(qn, kn, s, hw, C, hw, C, hw, C ** -0.5)
            L.call("rbvae_softmax_rows", 1, s, s, hw, hw, hw)
            L.call("rbvae_transpose2d", 1, vn, vt, hw, C, C, hw)
            gemm(s, vt, o2[n * hw:(n + 1) * hw], hw, hw, C, hw, C)
    t_old = timeit(old)
    fl = 4.0 * N * hw * hw * C
    print(f"N={N} tokens={hw}: online-softmax kernel {t_flash:8.1f} us ({fl / t_flash / 1e6:6.0f} TFLOP/s), three-launch form {t_old:8.1f} us "
          f"({fl / t_old / 1e6:6.0f} TFLOP/s), max |diff| {float((o.float() - o2.float()).abs().max()):.3f}", flush=True)
