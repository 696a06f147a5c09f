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
    args = (1, A, Wt, out, None, None, None, zero, N, H, W, Ho, Wo, 2, Ho, Wo, 1, C, C, C, C, 9, 1, ctypes.addressof(d), 0, mode, 0.0, 1.0, 0, None, None)
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

ctr = torch.zeros(1, dtype=torch.int64, device="cuda")
def t_ctr(iters=50):
    for _ in range(3): L.call("rbvae_counter_add", ctr, 1)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): L.call("rbvae_counter_add", ctr, 1)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
print("empty kernel back-to-back", t_ctr())
def run_k(N, Kc, iters=30):
    A = torch.randn(N * 64, Kc, device="cuda").bfloat16()
    Wt = torch.randn(256, 1, Kc, device="cuda").bfloat16()
    out = torch.empty(N * 64, 256, device="cuda", dtype=torch.bfloat16)
    desc = [1, 0, 0, 0, 0, 0]
    d = (ctypes.c_int * len(desc))(*desc)
    args = (1, A, Wt, out, None, None, None, zero, N * 64, 1, 1, 1, 1, 1, 1, 1, 1, Kc, 256, Kc, 256, 1, 1, ctypes.addressof(d), 0, 0, 0.0, 1.0, 0, None, None)
    for _ in range(3): L.call("rbvae_gather_gemm", *args)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): L.call("rbvae_gather_gemm", *args)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
for N in (16, 256, 1024):
    print(f"N={N} (blocks {N}): Kc=64 {run_k(N,64):6.1f}  Kc=256 {run_k(N,256):6.1f}  Kc=1024 {run_k(N,1024):6.1f}  Kc=4096 {run_k(N,4096):6.1f} us")
