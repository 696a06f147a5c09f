"""Timing of wgrad_gemm on the conv2 / conv3 weight-gradient shapes vs K-split (env RBVAE_WG_NS forces the ring)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sfv_amd as sfv
L = sfv._lib
zero = torch.zeros(256, dtype=torch.uint8, device="cuda")
C = 256
def run(N, H, ks, iters=30):
    Ho = H // 2
    P = N * Ho * Ho
    Dy = torch.randn(P, C, device="cuda").bfloat16()
    In = torch.randn(N * H * H, C, device="cuda").bfloat16()
    idx = torch.empty(9 * P, dtype=torch.int32, device="cuda")
    L.call("rbvae_conv_gather_index", idx, N, H, H, Ho, Ho, 3, 3, 2, 1)
    slabs = torch.empty(ks * C * 9 * C, dtype=torch.float32, device="cuda")
    args = (1, Dy, In, slabs, idx, zero, P, In.numel() // C, C, C, C, C, 9, ks)
    for _ in range(3):
        L.call("rbvae_wgrad_gemm", *args)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        L.call("rbvae_wgrad_gemm", *args)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
Hs = (int(sys.argv[1]),) if len(sys.argv) > 1 else (16, 8)
kss = (int(sys.argv[2]),) if len(sys.argv) > 2 else (7, 8, 16)
for H in Hs:
    for ks in kss:
        print(f"WG_NS={os.environ.get('RBVAE_WG_NS','auto')} N=256 H={H:2d} ksplit={ks:2d}: {run(256, H, ks):7.1f} us", flush=True)
