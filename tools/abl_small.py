"""Timing of gather_gemm on the under-filled deep-K shapes (conv3 forward: 256 frames, 8x8 -> 4x4, 256 ch)
for the tile variants selected by RBVAE_GG_SMALL (0: 128x128, 1: 128x64, 2: 128x32)."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sfv_amd as sfv
from importlib import import_module
E = import_module("symbols-from-video_amd.engine")
L = sfv._lib
zero = torch.zeros(256, dtype=torch.uint8, device="cuda")


def run(N, H, W, C=256, iters=50):
    Ho, Wo = H // 2, W // 2
    A = torch.randn(N * H * W, C, device="cuda").bfloat16()
    Wt = torch.randn(C, 9, C, device="cuda").bfloat16()
    out = torch.empty(N * Ho * Wo, C, device="cuda", dtype=torch.bfloat16)
    desc = E.conv_classes(3)
    d = (ctypes.c_int * len(desc))(*desc)
    args = (1, A, Wt, out, None, None, None, None, zero, N, H, W, Ho, Wo, 2, Ho, Wo, 1, C, C, C, C, 9, 1,
            ctypes.addressof(d), 0, 0, 0.0, 1.0, 0, None, None)
    for _ in range(5):
        L.call("rbvae_gather_gemm", *args)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        L.call("rbvae_gather_gemm", *args)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


for N, H in ((256, 8), (128, 8), (256, 16)):
    print(f"small={os.environ.get('RBVAE_GG_SMALL', 'default')} N={N} {H}x{H}->{H // 2}x{H // 2}: {run(N, H, H):7.1f} us", flush=True)
