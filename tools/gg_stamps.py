"""Where a gather-GEMM workgroup spends its time: phase stamps (s_memrealtime, 100 MHz) of every workgroup of
one launch, for the bench shapes.  Phases: 0 start, 1 row/tap tables ready, 2 staging roles set up, 3 K loop done,
4 tile in LDS, 5 stores issued, 6 this wave's stores acknowledged."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import sfv_amd as sfv
from importlib import import_module
E = import_module("symbols-from-video_amd.engine")
L = sfv._lib
zero = torch.zeros(256, dtype=torch.uint8, device="cuda")


def launch(kind, N=256, C=256, gate=False, bias=True, drop=1):
    if kind == "one":          # conv1-like: M = N*256 rows, K = 64
        M = N * 256
        A = torch.randn(M, 64, device="cuda").bfloat16(); W = torch.randn(C, 64, device="cuda").bfloat16()
        out = torch.empty(M, C, device="cuda", dtype=torch.bfloat16)
        desc = [1, 0, 0, 0, 0, 0]; geo = (M, 1, 1, 1, 1, 1, 1, 1, 1); kc, taps, ncls, rows = 64, 1, 1, M
    elif kind == "conv":       # conv2 forward: 16x16 -> 8x8
        A = torch.randn(N * 256, C, device="cuda").bfloat16(); W = torch.randn(C, 9, C, device="cuda").bfloat16()
        out = torch.empty(N * 64, C, device="cuda", dtype=torch.bfloat16)
        desc = E.conv_classes(3); geo = (N, 16, 16, 8, 8, 2, 8, 8, 1); kc, taps, ncls, rows = C, 9, 1, N * 64
    else:                      # deconv2 forward: 8x8 -> 16x16, 4 parity classes
        A = torch.randn(N * 64, C, device="cuda").bfloat16(); W = torch.randn(C, 9, C, device="cuda").bfloat16()
        out = torch.empty(N * 256, C, device="cuda", dtype=torch.bfloat16)
        desc, ncls = E.dgrad_classes(3); geo = (N, 8, 8, 8, 8, 1, 16, 16, 2); kc, taps, rows = C, 9, N * 256
    d = (ctypes.c_int * len(desc))(*desc)
    b = torch.randn(C, device="cuda") if bias else None
    g = torch.randn(rows, C, device="cuda").bfloat16() if gate else None
    args = (1, A, W, out, b, g, None, None, zero, *geo, kc, C, kc, C, taps, ncls, ctypes.addressof(d), 1, drop, 0.2,
            1.25, 3, None, None)
    for _ in range(3):
        L.call("rbvae_gather_gemm", *args)
    nwg = 8192
    st = torch.zeros(nwg * 8, dtype=torch.int64, device="cuda")
    L.dbg_call("rbvae_dbg_gg_stamps", st)
    L.call("rbvae_gather_gemm", *args)
    torch.cuda.synchronize()
    L.dbg_call("rbvae_dbg_gg_stamps", None)
    s = st.cpu().numpy().reshape(nwg, 8)
    s = s[s[:, 0] > 0]
    t0 = s[:, 0].min()
    rel = (s[:, :7] - t0) * 0.01          # us
    dur = np.diff(rel, axis=1)
    print(f"{kind:6s} gate={int(gate)} bias={int(bias)} drop={drop}: {len(s)} workgroups, launch span {rel[:, 6].max():.1f} us")
    print("   phase durations (us) median / p90:  " + "  ".join(f"{i}->{i + 1}: {np.median(dur[:, i]):.2f}/{np.percentile(dur[:, i], 90):.2f}" for i in range(6)))
    print(f"   workgroup start time: median {np.median(rel[:, 0]):.1f}, p90 {np.percentile(rel[:, 0], 90):.1f}, max {rel[:, 0].max():.1f} us; "
          f"workgroup life median {np.median(rel[:, 6] - rel[:, 0]):.1f} us")


launch("one", gate=False, bias=True, drop=1)
launch("one", gate=True, bias=False, drop=0)
launch("conv", gate=False, bias=True, drop=1)
launch("conv", gate=True, bias=False, drop=0)
launch("dgrad", gate=False, bias=True, drop=1)
launch("dgrad", gate=True, bias=False, drop=0)
