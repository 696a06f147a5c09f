"""Per-launch floors of the fused bench step (BASELINE configs[1]: 256 frames of 4x32x32 per GPU, bf16) and the ceiling they
add up to.  A launch's floor = max(matrix-core time at the dense bf16 peak, L2 -> LDS fill time at the measured per-CU
intake, HBM time at the measured streaming rate) + one dependent-kernel boundary; latency chains (the LSTM wavefronts) are
priced by their dependent steps.  Two sums: the CRITICAL PATH of the two-queue schedule (forward chain, data-gradient
chain, step boundary; weight gradients beside them) and the RESOURCE sum (every launch's floor back to back: what the
step costs if nothing overlaps because every GEMM already wants the whole chip).  The achievable step time lies between
the two; bench.py reports achieved / ceiling against the critical path (the optimistic one).

Constants (MI355X_MICROARCH.md + this repository's measurements): 2.5 PFLOP/s dense bf16; 6.29 TB/s streaming HBM;
70 GB/s per CU from its XCD's L2 into LDS (guide: 66-73; measured here 63-69 for rbvae_wgrad3x3s2_row's loaders, 78 for
rbvae_wgrad_gemm's fill-only ablation), 256 CUs; 1.45 us per dependent kernel boundary."""
import json
import sys

PEAK, HBM, FILL_CU, NCU, BOUNDARY = 2.5e15, 6.29e12, 70e9, 256, 1.45e-6
N, C, L = 256, 256, 32                       # frames, channels, latent
P1, P2, P3 = N * 16 * 16, N * 8 * 8, N * 4 * 4
ES = 2


def gemm(name, rows, nout, k, tile_rows, tile_cols, queue, crit, a_share=1.0):
    """row-gather GEMM: FLOPs; fill = per-tile operand bytes through LDS (A rows re-gathered per tap unless a_share < 1)"""
    fl = 2.0 * rows * nout * k
    tiles = -(-rows // tile_rows) * -(-nout // tile_cols)
    fill = tiles * k * (tile_rows * a_share + tile_cols) * ES
    hbm = (rows * k / 9 + nout * k + rows * nout) * ES          # compulsory: input pixels once, weights, output
    return dict(name=name, flops=fl, fill=fill, cus=min(tiles, NCU), hbm=hbm, queue=queue, crit=crit)


def mover(name, mb, queue, crit):
    return dict(name=name, flops=0.0, fill=0.0, cus=NCU, hbm=mb * 1e6, queue=queue, crit=crit)


def chain(name, steps, us_per_step, queue, crit):
    return dict(name=name, flops=0.0, fill=0.0, cus=32, hbm=0.0, queue=queue, crit=crit, chain_us=steps * us_per_step)


K9 = 9 * C
LAUNCHES = [
    # ---- forward (one chain)
    mover("conv1 + bias/ReLU/dropout (fused, writes a1 + im2col rows)", 4.2 + 33.5 + 8.4, 0, True),
    gemm("conv2 forward", P2, C, K9, 128, 128, 0, True),
    gemm("conv3 forward", P3, C, K9, 64, 64, 0, True),
    mover("encoder fc 4096 -> 32 (K-split)", 2.1 + 0.3, 0, True),
    chain("LSTM encoder -> binarise -> decoder, wavefront (T + 2 layers - 1 = 15 dependent cell steps)", 15, 0.25, 0, True),
    mover("decoder fc 32 -> 4096", 2.1 + 0.3, 0, True),
    gemm("deconv1 forward (4 parity classes)", P2, C, K9 / 4, 128, 128, 0, True),
    gemm("deconv2 forward (4 parity classes)", P1, C, K9 / 4, 128, 128, 0, True),
    mover("deconv3 + sigmoid + MSE (fused)", 33.5 + 4.2 + 4.2 + 4.2, 0, True),
    mover("pair term (contrast) value + gradient", 0.1, 1, False),
    # ---- backward: data-gradient chain (critical) ...
    mover("deconv3 input gradient + gate + im2col rows (fused)", 4.2 + 33.5 + 33.5 + 8.4, 0, True),
    gemm("deconv2 input gradient (conv form)", P2, C, K9, 128, 128, 0, True),
    gemm("deconv1 input gradient (conv form)", P3, C, K9, 64, 64, 0, True),
    mover("decoder fc input gradient", 2.1 + 0.3, 0, True),
    chain("LSTM decoder -> binarise -> encoder BPTT, wavefront (15 dependent cell steps)", 15, 0.35, 0, True),
    mover("encoder fc input gradient", 2.1 + 0.3, 0, True),
    gemm("conv3 input gradient (4 parity classes)", P2, C, K9 / 4, 128, 128, 0, True),
    gemm("conv2 input gradient (4 parity classes)", P1, C, K9 / 4, 128, 128, 0, True),
    # ---- ... weight gradients beside it
    mover("loss bookkeeping", 0.1, 1, False),
    mover("deconv3 weight gradient (im2col rows x d2)", 33.5 + 8.4, 1, False),
    dict(name="deconv2 weight gradient (3 taps per workgroup)", flops=2.0 * P2 * C * K9, fill=(P2 // 64) * 12 * 51200.0, cus=NCU,
         hbm=(P2 * C + P1 * C) * ES + 10 * 2.36e6, queue=1, crit=False),
    dict(name="deconv1 weight gradient", flops=2.0 * P3 * C * K9, fill=(P3 // 64) * 36 * 32768.0, cus=108,
         hbm=(P3 * C + P2 * C) * ES + 3 * 2.36e6, queue=1, crit=False),
    mover("decoder fc bias sums + weight gradient", 2.1 + 2.1, 1, False),
    mover("decoder slab / partial-sum reductions", 10 * 2.36 + 3 * 2.36 + 4.0, 1, False),
    mover("LSTM weight gradients", 0.5, 1, False),
    mover("encoder fc weight gradient", 2.1 + 0.5, 1, False),
    dict(name="conv3 weight gradient", flops=2.0 * P3 * C * K9, fill=(P3 // 64) * 36 * 32768.0, cus=108,
         hbm=(P3 * C + P2 * C) * ES + 3 * 2.36e6, queue=0, crit=False),
    dict(name="conv2 weight gradient (3 taps per workgroup)", flops=2.0 * P2 * C * K9, fill=(P2 // 64) * 12 * 51200.0, cus=NCU,
         hbm=(P2 * C + P1 * C) * ES + 10 * 2.36e6, queue=0, crit=False),
    mover("conv1 weight gradient (im2col rows x da1)", 33.5 + 8.4, 0, True),
    # ---- step boundary
    mover("encoder slab / partial-sum reductions", 10 * 2.36 + 3 * 2.36 + 6.0, 0, True),
    mover("Adam + weight repack + next batch gather (10.85 MB of parameters: w, g, m, v read; w, m, v + bf16 copies written)",
          10.85 * 7 + 11.0 + 2.1, 0, True),
]


def floor_us(l):
    if "chain_us" in l:
        return l["chain_us"] + BOUNDARY * 1e6
    t = max(l["flops"] / PEAK, l["fill"] / (l["cus"] * FILL_CU) if l["fill"] else 0.0, l["hbm"] / HBM)
    return (t + BOUNDARY) * 1e6


def bound(l):
    if "chain_us" in l:
        return "latency chain"
    c = {"mfma": l["flops"] / PEAK, "l2->lds fill": l["fill"] / (l["cus"] * FILL_CU) if l["fill"] else 0.0, "hbm": l["hbm"] / HBM}
    return max(c, key=c.get)


def ceiling():
    crit = sum(floor_us(l) for l in LAUNCHES if l["crit"])
    total = sum(floor_us(l) for l in LAUNCHES)
    return {"critical_path_us": round(crit, 1), "resource_sum_us": round(total, 1), "launches": len(LAUNCHES),
            "frames_per_s_at_critical_path": round(N / crit * 1e6, 0), "frames_per_s_at_resource_sum": round(N / total * 1e6, 0),
            "model": "tools/ceiling.py: per-launch max(MFMA at 2.5 PF, L2->LDS fill at 70 GB/s/CU, HBM at 6.29 TB/s) + 1.45 us boundary"}


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "json":
        print(json.dumps(ceiling()))
        sys.exit(0)
    print(f"{'launch':<98s} {'GFLOP':>7s} {'fill MB':>8s} {'HBM MB':>7s} {'floor us':>8s}  bound            queue crit")
    for l in LAUNCHES:
        print(f"{l['name'][:98]:<98s} {l['flops'] / 1e9:7.2f} {l['fill'] / 1e6:8.1f} {l['hbm'] / 1e6:7.1f} {floor_us(l):8.2f}  {bound(l):<16s} "
              f"{l['queue']:>3d}   {'*' if l['crit'] else ''}")
    c = ceiling()
    print(f"critical path {c['critical_path_us']} us = {c['frames_per_s_at_critical_path']:.0f} frames/s; "
          f"resource sum {c['resource_sum_us']} us = {c['frames_per_s_at_resource_sum']:.0f} frames/s ({c['launches']} launches)")
