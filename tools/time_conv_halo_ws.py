"""rbvae_conv3x3_halo, bf16: the persistent wave-specialised kernel (conv_halo_ws.hip, variant 0) against the
one-tile-per-workgroup kernel (conv_halo.hip, variant 1) on the LDM encoder's layers, interleaved; us per launch, TFLOP/s."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sfv_amd as sfv

L = sfv._lib
lib = L.dbg_lib()
zero = torch.zeros(256, dtype=torch.uint8, device="cuda")


def halo(A, Wp, out, N, H, W, cin, cout, scale=None, shift=None, stats=None, addend=None):
    L.call("rbvae_conv3x3_halo", 1, A, Wp, out, None, addend, zero, scale, shift, 1, stats, cout // 32 if stats is not None else 0,
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
          (4, 512, 512, 128, 128), (4, 512, 512, 64, 64), (8, 256, 256, 44, 80), (8, 128, 128, 256, 256)]
if len(sys.argv) > 1:
    shapes = [shapes[int(i)] for i in sys.argv[1].split(",")]
for N, cin, cout, H, W in shapes:
    A = torch.randn(N * H * W, cin, device="cuda").bfloat16()
    Wp = (torch.randn(cout, 9, cin, device="cuda") / (3 * cin ** 0.5)).bfloat16()
    o = torch.empty(N * H * W, cout, dtype=torch.bfloat16, device="cuda")
    add = torch.randn(N * H * W, cout, device="cuda").bfloat16()
    scale, shift = torch.rand(N, cin, device="cuda") + 0.5, torch.randn(N, cin, device="cuda")
    stats = torch.empty(L.query("rbvae_conv3x3_halo_stats_floats", N, H, W, cout, cout // 32), device="cuda")
    fl = 2.0 * N * H * W * cout * cin * 9
    res = {}
    for rnd in range(3):
        for v in (1, 0):
            lib.rbvae_dbg_conv_halo_variant(v)
            for name, fn in (("plain", lambda: halo(A, Wp, o, N, H, W, cin, cout)),
                             ("gn+stats", lambda: halo(A, Wp, o, N, H, W, cin, cout, scale, shift, stats)),
                             ("gn+add", lambda: halo(A, Wp, o, N, H, W, cin, cout, scale, shift, None, add))):
                res.setdefault((name, v), []).append(timeit(fn, 5))
    lib.rbvae_dbg_conv_halo_variant(0)
    line = f"N={N} {cin:3d}->{cout:3d} {H}x{W}: "
    for name in ("plain", "gn+stats", "gn+add"):
        t1, t0 = min(res[(name, 1)]), min(res[(name, 0)])
        line += f"{name}: tile/WG {t1:7.1f} us  persistent {t0:7.1f} us ({fl / t0 / 1e6:5.0f} TF, x{t1 / t0:.2f})   "
    print(line, flush=True)
