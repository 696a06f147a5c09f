"""3x3 stride-1 convolutions of the LDM encoder at 512x512 frames: rbvae_conv3x3_halo against rbvae_gather_gemm
(same packed weights, same NHWC rows), interleaved in one process; us per launch and TFLOP/s."""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sfv_amd as sfv

L = sfv._lib
zero = torch.zeros(256, dtype=torch.uint8, device="cuda")
d = [9, 0, 0]
for kh in range(3):
    for kw in range(3):
        d += [kh * 3 + kw, kh - 1, kw - 1]
desc = (ctypes.c_int * len(d))(*d)


def gather(A, Wp, out, N, H, W, cin, cout):
    L.call("rbvae_gather_gemm", 1, A, Wp, out, None, None, None, None, zero, N, H, W, H, W, 1, H, W, 1, cin, cout, cin, cout, 9, 1,
           ctypes.addressof(desc), 0, 0, 0.0, 1.0, 0, None, None)


def halo(A, Wp, out, N, H, W, cin, cout, scale=None, shift=None, stats=None):
    L.call("rbvae_conv3x3_halo", 1, A, Wp, out, None, None, zero, scale, shift, 1, stats, cout // 32 if stats is not None else 0,
           N, H, W, H, W, 1, 1, cin, cout, cin, cout)


def timeit(fn, it):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(it):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / it * 1e6


shapes = [(4, 128, 128, 512, 512), (4, 128, 256, 256, 256), (4, 256, 256, 256, 256), (4, 256, 512, 128, 128),
          (4, 512, 512, 128, 128), (4, 512, 512, 64, 64), (8, 256, 256, 44, 80), (1, 512, 512, 64, 64)]
if len(sys.argv) > 1:
    shapes = [shapes[int(i)] for i in sys.argv[1].split(",")]
for N, cin, cout, H, W in shapes:
    A = torch.randn(N * H * W, cin, device="cuda").bfloat16()
    Wp = (torch.randn(cout, 9, cin, device="cuda") / (3 * cin ** 0.5)).bfloat16()
    o1 = torch.empty(N * H * W, cout, dtype=torch.bfloat16, device="cuda")
    o2 = torch.empty_like(o1)
    scale, shift = torch.rand(N, cin, device="cuda") + 0.5, torch.randn(N, cin, device="cuda")
    stats = torch.empty(L.query("rbvae_conv3x3_halo_stats_floats", N, H, W, cout, cout // 32), device="cuda")
    fl = 2.0 * N * H * W * cout * cin * 9
    only = os.environ.get("CH_ONLY")
    res = {"halo": []} if only else {"gather": [], "halo": [], "halo+gn": [], "halo+gn+stats": []}
    for rnd in range(3):
        if not only:
            res["gather"].append(timeit(lambda: gather(A, Wp, o1, N, H, W, cin, cout), 5))
        res["halo"].append(timeit(lambda: halo(A, Wp, o2, N, H, W, cin, cout), 5))
        if only:
            continue
        res["halo+gn"].append(timeit(lambda: halo(A, Wp, o2, N, H, W, cin, cout, scale, shift), 5))
        res["halo+gn+stats"].append(timeit(lambda: halo(A, Wp, o2, N, H, W, cin, cout, scale, shift, stats), 5))
    halo(A, Wp, o2, N, H, W, cin, cout)
    diff = float((o1.float() - o2.float()).abs().max()) if not only else -1.0
    line = f"N={N} {cin:3d}->{cout:3d} {H}x{W}: "
    for k, v in res.items():
        t = min(v)
        line += f"{k} {t:8.1f} us ({fl / t / 1e6:5.0f} TF)  "
    print(line + f"max|diff| {diff:.3f}", flush=True)
