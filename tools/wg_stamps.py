"""Phase stamps of the weight-gradient GEMM on the bench's 256-channel shape (needs a library built with
-DWG_STAMPS=1: tools/ab_variants.sh wgrad_gemm.hip st:"-DWG_STAMPS=1", then RBVAE_LIB=...librbvae_hip_st.so).
Phases: 0 start, 1 gather indices in LDS, 2 staging roles set up, 3 K loop done, 4 slab stores issued, 5 stores acknowledged."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import sfv_amd as sfv
L = sfv._lib
zero = torch.zeros(256, dtype=torch.uint8, device="cuda")


def run(N, ih, oh, C=256, ks=7):
    P = N * oh * oh
    Dy = torch.randn(P, C, device="cuda").bfloat16()
    In = torch.randn(N * ih * ih, C, device="cuda").bfloat16()
    idx = torch.empty(9 * P, dtype=torch.int32, device="cuda")
    L.call("rbvae_conv_gather_index", idx, N, ih, ih, oh, oh, 3, 3, 2, 1)
    slabs = torch.empty(ks * C * 9 * C, device="cuda")
    args = (1, Dy, In, slabs, idx, zero, P, In.numel() // C, C, C, C, C, 9, ks)
    for _ in range(3):
        L.call("rbvae_wgrad_gemm", *args)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        L.call("rbvae_wgrad_gemm", *args)
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) / 20 * 1e3
    st = torch.zeros(4096 * 8, dtype=torch.int64, device="cuda")
    L.dbg_call("rbvae_dbg_wg_stamps", st)
    L.call("rbvae_wgrad_gemm", *args)
    torch.cuda.synchronize()
    L.dbg_call("rbvae_dbg_wg_stamps", None)
    s = st.cpu().numpy().reshape(-1, 8)
    s = s[s[:, 0] > 0]
    if len(s) == 0:
        print(f"P={P}: {us:.1f} us per launch (library built without -DWG_STAMPS=1: no stamps)")
        return
    rel = (s[:, :6] - s[:, 0].min()) * 0.01
    dur = np.diff(rel, axis=1)
    steps = -(-(-(-P // ks)) // 64)
    print(f"P={P} ks={ks}: {us:.1f} us per launch, {len(s)} workgroups, {steps} K steps each")
    print("   phase durations (us) median / p90: " + "  ".join(f"{i}->{i + 1}: {np.median(dur[:, i]):.2f}/{np.percentile(dur[:, i], 90):.2f}" for i in range(5)))
    print(f"   K loop per step: median {np.median(dur[:, 2]) / steps:.3f} us;  workgroup start median {np.median(rel[:, 0]):.1f} max {rel[:, 0].max():.1f} us; last end {rel[:, 5].max():.1f} us")


ks_big = int(sys.argv[1]) if len(sys.argv) > 1 else 7
ks_small = int(sys.argv[2]) if len(sys.argv) > 2 else 3
run(256, 16, 8, ks=ks_big)          # conv2 / deconv1 weight gradient: P = 16384
run(256, 8, 4, ks=ks_small)         # conv3 / deconv0: P = 4096
