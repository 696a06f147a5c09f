"""Transposed 3x3 stride-2 convolutions of the RBVAE path: rbvae_deconv3x3s2_halo against the four parity-class launches of
rbvae_gather_gemm (same operands), forward form (bias + ReLU + keyed dropout) and gradient form (gate + column sums)."""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sfv_amd as sfv
from importlib import import_module

E = import_module("symbols-from-video_amd.engine")
L = sfv._lib
zero = torch.zeros(256, dtype=torch.uint8, device="cuda")
desc, ncls = E.dgrad_classes(3)
d = (ctypes.c_int * len(desc))(*desc)


def timeit(fn, it):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(it):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / it * 1e6


shapes = [(256, 256, 256, 8, 8), (256, 256, 256, 4, 4), (128, 256, 256, 22, 40), (128, 256, 256, 11, 20),
          (128, 64, 64, 64, 64), (128, 64, 64, 32, 32), (1024, 256, 256, 8, 8)]
if len(sys.argv) > 1:
    shapes = [shapes[int(i)] for i in sys.argv[1].split(",")]
for N, cin, cout, TH, TW in shapes:
    A = torch.randn(N * TH * TW, cin, device="cuda").bfloat16()
    Wp = (torch.randn(cout, 9, cin, device="cuda") / (1.5 * cin ** 0.5)).bfloat16()
    b = torch.randn(cout, device="cuda")
    rows = N * 4 * TH * TW
    o1 = torch.empty(rows, cout, dtype=torch.bfloat16, device="cuda")
    o2 = torch.empty_like(o1)
    gate = torch.randn(rows, cout, device="cuda").bfloat16()
    ws_h = torch.empty(L.query("rbvae_deconv3x3s2_halo_colsum_rows", 1, N, TH, TW, cin, cout), cout, device="cuda")
    ws_g = torch.empty(ncls * -(-(N * TH * TW) // 128), cout, device="cuda")
    fl = 2.0 * N * TH * TW * cout * cin * 9

    def g_fwd():
        L.call("rbvae_gather_gemm", 1, A, Wp, o1, b, None, None, None, zero, N, TH, TW, TH, TW, 1, 2 * TH, 2 * TW, 2, cin, cout,
               cin, cout, 9, ncls, ctypes.addressof(d), 1, 1, 0.2, 1.25, 5, None, None)

    def h_fwd():
        L.call("rbvae_deconv3x3s2_halo", 1, A, Wp, o2, b, None, None, zero, N, TH, TW, cin, cout, cin, cout, 1, 1, 0.2, 1.25, 5,
               None, None)

    def g_bwd():
        L.call("rbvae_gather_gemm", 1, A, Wp, o1, None, gate, None, None, zero, N, TH, TW, TH, TW, 1, 2 * TH, 2 * TW, 2, cin, cout,
               cin, cout, 9, ncls, ctypes.addressof(d), 0, 0, 0.0, 0.5, 0, None, ws_g)

    def h_bwd():
        L.call("rbvae_deconv3x3s2_halo", 1, A, Wp, o2, None, gate, None, zero, N, TH, TW, cin, cout, cin, cout, 0, 0, 0.0, 0.5, 0,
               None, ws_h)

    res = {}
    it = 20 if fl < 5e10 else 5
    for rnd in range(3):
        for k, fn in (("gather fwd", g_fwd), ("halo fwd", h_fwd), ("gather bwd", g_bwd), ("halo bwd", h_bwd)):
            res.setdefault(k, []).append(timeit(fn, it))
    line = f"N={N} {cin}->{cout} grid {TH}x{TW}: "
    for k, v in res.items():
        t = min(v)
        line += f"{k} {t:7.1f} us ({fl / t / 1e6:5.0f} TF)  "
    print(line, flush=True)
