"""rbvae_wgrad3x3s2_row against rbvae_wgrad_gemm (9 taps through the gather table) at the bench / native / cfg 5 layer shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sfv_amd as sfv
L = sfv._lib
dev = torch.device("cuda", 0)
CASES = [("bench conv2 / deconv2", 256, 8, 8, 256, 256), ("bench conv3 / deconv1", 256, 4, 4, 256, 256),
         ("native conv2 / deconv2", 128, 22, 40, 256, 256), ("native conv3 / deconv1", 128, 11, 20, 256, 256),
         ("cfg5 conv2 (64x64 latents)", 64, 16, 16, 256, 256), ("cfg5 conv3", 64, 8, 8, 256, 256)]
sel = [int(a) for a in sys.argv[1].split(",")] if len(sys.argv) > 1 else range(len(CASES))
KS = [int(a) for a in sys.argv[2].split(",")] if len(sys.argv) > 2 else None
zero = torch.zeros(256, dtype=torch.uint8, device=dev)


def timed(fn, it=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / it


for i in sel:
    name, N, OH, OW, Ca, Cb = CASES[i]
    P = N * OH * OW
    S = (torch.randn(P, Ca, device=dev) / 8).to(torch.bfloat16)
    G = torch.randn(4 * P, Cb, device=dev).to(torch.bfloat16)
    idx = torch.empty(9 * P, dtype=torch.int32, device=dev)
    L.call("rbvae_conv_gather_index", idx, N, 2 * OH, 2 * OW, OH, OW, 3, 3, 2, 1)
    blocks = -(-Ca // 128) * -(-Cb // 128) * 9
    ks_old = max(1, min(256 // blocks, P // 256, (4 << 20) // (Ca * 9 * Cb)))
    if P <= 4096: ks_old = min(ks_old, 3)
    ks_old = max(ks_old, -(-P // 4096))
    sl_old = torch.empty(ks_old, Ca, 9, Cb, device=dev)
    t_old = timed(lambda: L.call("rbvae_wgrad_gemm", 1, S, G, sl_old, idx, zero, P, 4 * P, Ca, Cb, Ca, Cb, 9, ks_old))
    nblk = L.query("rbvae_wgrad3x3s2_row_blocks", N, OH, OW)
    ntile = (Ca // 128) * (Cb // 128) * 3
    gf = 2.0 * P * Ca * Cb * 9 / 1e9
    print(f"{name:28s} P={P:7d} {Ca}x{Cb} blocks {nblk}: gemm ks={ks_old:3d} {t_old:7.1f} us ({gf / t_old * 1e3:6.0f} TF)", flush=True)
    ref = sl_old.sum(0)
    for ks in (KS or sorted({max(1, min(w // ntile, nblk)) for w in (64, 128, 192, 256, 512)})):
        ks = min(ks, nblk)
        sl = torch.empty(ks, Ca, 9, Cb, device=dev)
        t = timed(lambda: L.call("rbvae_wgrad3x3s2_row", 1, S, G, sl, zero, N, OH, OW, Ca, Cb, Ca, Cb, ks))
        d = float((sl.sum(0) - ref).abs().max())
        print(f"    row ks={ks:3d} wgs={ks * ntile:4d} steps={-(-nblk // ks):3d} {t:7.1f} us ({gf / t * 1e3:6.0f} TF) slabs "
              f"{ks * Ca * 9 * Cb * 4 / 1e6:5.1f} MB max|diff| {d:.3g}", flush=True)
