"""Timing of gather_gemm on conv2-forward-like shapes vs grid size and K depth."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sfv_amd as sfv
from importlib import import_module
E = import_module("symbols-from-video_amd.engine")
L = sfv._lib
C, H, W, Ho, Wo = 256, 16, 16, 8, 8
zero = torch.zeros(256, dtype=torch.uint8, device="cuda")
def run(N, mode, taps3=True, iters=30):
    A = torch.randn(N * H * W, C, device="cuda").bfloat16()
    Wt = torch.randn(C, 9, C, device="cuda").bfloat16()
    out = torch.empty(N * Ho * Wo, C, device="cuda", dtype=torch.bfloat16)
    desc = E.conv_classes(3) if taps3 else [1, 0, 0, 4, 0, 0]
    d = (ctypes.c_int * len(desc))(*desc)
    args = (1, A, Wt, out, None, None, None, None, zero, N, H, W, Ho, Wo, 2, Ho, Wo, 1, C, C, C, C, 9, 1, ctypes.addressof(d), 0, mode, 0.0, 1.0, 0, None, None)
    for _ in range(3):
        L.call("rbvae_gather_gemm", *args)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        L.call("rbvae_gather_gemm", *args)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
import time, subprocess
x = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16)
t0 = time.time()
while time.time() - t0 < 0.2:
    for _ in range(20): y = x @ x
    torch.cuda.synchronize()

for N in (256, 512):
    print(f"dbg={os.environ.get('RBVAE_GG_DBG','0')} N={N:5d} 9taps {run(N,0):7.1f} us")
