"""Phase stamps of rbvae_wgrad3x3s2_row (needs a library built with -DWR_STAMPS=1: tools/ab_variants.sh wgrad_row.hip
st:"-DWR_STAMPS=1", then RBVAE_LIB=...librbvae_hip_st.so).  Per workgroup: start -> loop start -> loop end -> stores
acknowledged (us), share of the loop's cycles spent waiting at the stage barriers, in-kernel clock of the loop."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import sfv_amd as sfv
L = sfv._lib
zero = torch.zeros(256, dtype=torch.uint8, device="cuda")


def run(N, OH, OW, ks, C=256):
    P = N * OH * OW
    S = (torch.randn(P, C, device="cuda") / 8).bfloat16()
    G = torch.randn(4 * P, C, device="cuda").bfloat16()
    slabs = torch.empty(ks * C * 9 * C, device="cuda")
    args = (1, S, G, slabs, zero, N, OH, OW, C, C, C, C, ks)
    for _ in range(3):
        L.call("rbvae_wgrad3x3s2_row", *args)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        L.call("rbvae_wgrad3x3s2_row", *args)
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) / 20 * 1e3
    st = torch.zeros(4096 * 16, dtype=torch.int64, device="cuda")
    L.dbg_call("rbvae_dbg_wr_stamps", st)
    L.call("rbvae_wgrad3x3s2_row", *args)
    torch.cuda.synchronize()
    L.dbg_call("rbvae_dbg_wr_stamps", None)
    both = st.cpu().numpy().reshape(-1, 2, 8)
    both = both[both[:, 0, 0] > 0]
    s = both[:, 0]
    t0 = s[:, 0].min()
    rel = (s[:, :4] - t0) * 0.01
    loop = rel[:, 2] - rel[:, 1]
    steps = s[:, 6]
    clk = s[:, 5] / np.maximum(loop, 1e-9) / 1e3          # GHz
    print(f"{N}x{OH}x{OW} ks={ks}: {us:.1f} us per launch, {len(s)} workgroups, {int(np.median(steps))} K steps")
    print(f"   start median {np.median(rel[:, 0]):.1f} max {rel[:, 0].max():.1f} | prologue {np.median(rel[:, 1] - rel[:, 0]):.2f} | loop "
          f"{np.median(loop):.2f} (p90 {np.percentile(loop, 90):.2f}) = {np.median(loop / steps):.3f} us/step | epilogue "
          f"{np.median(rel[:, 3] - rel[:, 2]):.2f} | last end {rel[:, 3].max():.1f} us")
    for wv, nm in ((0, "wave 0"), (1, "wave 4")):
        x = both[:, wv]
        cyc = np.maximum(x[:, 5], 1) / np.maximum(x[:, 6] , 1)
        print(f"   {nm}: cycles per step {np.median(cyc):.0f}, of them waiting for own LDS-DMA pieces {np.median(x[:, 4] / np.maximum(x[:, 6] - 1, 1)):.0f}, "
              f"at the barrier {np.median(x[:, 7] / np.maximum(x[:, 6] - 1, 1)):.0f}")
    print(f"   in-kernel clock {np.median(clk):.2f} GHz")


cases = {"bench2": (256, 8, 8), "bench3": (256, 4, 4), "native2": (128, 22, 40), "native3": (128, 11, 20)}
for name in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["bench2", "native2"]):
    for ks in ([int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [5, 21]):
        run(*cases[name], ks)
