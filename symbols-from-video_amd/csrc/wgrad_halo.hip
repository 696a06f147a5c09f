// Weight gradient of a 3x3 stride-2 pad-1 convolution (and of ConvTranspose2d(c, c, 3, 2, 1, 1), the same sum with the
// operands' roles swapped) with the NINE TAPS IN ONE WORKGROUP:
//
//   dW[ks][a][t][b] = sum_{p in K-slice ks} S[p][a] * G[hi(p, t)][b],   hi((n, r, c), (kh, kw)) = (n, 2r + kh - 1, 2c + kw - 1)
//
// S = the low-resolution operand ([Nimg*OH*OW][lds] bf16: a conv's output gradient / a transposed conv's input), G = the
// high-resolution one ([Nimg*2OH*2OW][ldg] bf16: the conv's input / the transposed conv's output gradient); autograd of
// percep_RBVAE_model.py:51-57,76-81 as run by percep_RBVAE_train.py:552.  rbvae_wgrad_gemm gives every tap its own
// workgroups, so S and the gathered G rows are fetched nine times from the L2s (cfg 3, 64 channels: 1.2 GB through the L2s
// for 335 MB of operands per weight); here a workgroup owns a 64 x 64 (a x b) tile of ALL nine taps (36 sub-tiles of
// 16 x 16, 4-5 per wave, 80 accumulators per lane) and walks 4 x 8 blocks of low-resolution pixels (one 32-pixel MFMA
// step each): per block the [32 px][64 a] tile of S and the 9 x 17 patch of G around it come in ONCE by LDS-DMA and serve
// all taps.  The stages (32 KB) go through a ring of four: two blocks in flight behind the one whose fragments are being
// read (one stage in flight against ~1.6 us of L2 / Infinity-Cache latency filled a CU at 39 GB/s), counted vmcnt waits,
// one barrier per block; the fragment reads of block s + 1 are issued before the MFMAs of block s.
//
// LDS images (128-byte pixel rows, reduction index = row, fragments by ds_read_b64_tr_b16 as in wgrad_gemm.hip):
//   S tile   row k = 8 rr + cc, 16-byte chunks XOR-swizzled by tr_swz (wgrad_gemm.hip).
//   G patch  four parity planes (row parity, column parity) with a row pitch of 12 slots: patch pixel (u, v) lives in plane
//            (u & 1, v & 1) at slot (u >> 1) * 12 + (v >> 1), so tap (kh, kw) of block pixel (rr, cc) is slot
//            plane(kh & 1, kw & 1) + (rr + (kh >> 1)) * 12 + cc + (kw >> 1): consecutive pixels of a row are consecutive
//            slots although the convolution strides by two, a tap is a constant offset, and with a pitch of 12 = 4 (mod 8)
//            slots and the chunk pair XORed by (slot >> 1) & 3 the eight pixel rows a 32-lane half of a transposing read
//            touches fall on eight distinct 32-byte bank groups (conflict-free).  Slots no tap reads are not filled.
// Workgroup -> (tile, K-slice): XCD x (= workgroup id % 8) takes the K-slices x, x + 8, .. with all their channel tiles, so
// the 16 tile workgroups of a 256 x 256 layer that walk the same pixels share them in ONE L2.
// Each K-slice writes its own f32 slab [a][t][b] (rbvae_wgrad_gemm's layout: the same fixed-order reduction jobs follow).
#include "common.h"
#include <stdlib.h>
#include <type_traits>

#ifndef WH_ABL          // timing ablations (results wrong on purpose): 1 no LDS-DMA, 2 no fragment reads / MFMAs
#define WH_ABL 0
#endif
#if WH_ABL && !defined(RBVAE_ABLATION)
#error "WH_ABL builds give wrong results: define RBVAE_ABLATION to confirm"
#endif

namespace rbvae {

typedef __attribute__((ext_vector_type(8))) short wh_bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short wh_s16x4_t;
typedef __attribute__((ext_vector_type(4))) float wh_f32x4_t;

struct WhArgs {
    const unsigned char* S;    // [Nimg*OH*OW][lds] bf16
    const unsigned char* G;    // [Nimg*2OH*2OW][ldg] bf16
    float* dW;                 // [ksplit][Ca][9][Cb] f32
    const unsigned char* zero; // >= 16 zero bytes
    int Nimg, OH, OW, Ca, Cb, lds, ldg, ksplit;
    int BR, BC, nblk, per;     // 4 x 8 blocks per image (rows, columns), blocks in all, blocks per K-slice
};

constexpr int WH_BH = 4, WH_BW = 8;                                  // block of low-resolution pixels = one 32-pixel MFMA step
constexpr int WH_PW = 12;                                            // plane row pitch (slots)
constexpr int WH_P00 = 0, WH_P01 = 60, WH_P10 = 120, WH_P11 = 168;   // plane (row parity, column parity) bases: 5, 5, 4, 4 rows
constexpr int WH_SLOTS = 224;                                        // 216 used + padding to whole LDS-DMA instructions
constexpr int WH_S_BYTES = 32 * 128, WH_G_BYTES = WH_SLOTS * 128, WH_STAGE = WH_S_BYTES + WH_G_BYTES;   // 4 + 28 = 32 KB
constexpr int WH_RING = 4;
constexpr int WH_DI = (WH_STAGE / 1024) / 8;                         // LDS-DMA instructions per wave and stage: 4
constexpr int WH_UNITS = 5;                                          // (tap, b sub-tile) units per wave: taps w/4 + 2j, sub-tile w % 4

__device__ __forceinline__ void wh_glds16(const void* g, void* lds) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}
__device__ __forceinline__ int wh_swz_s(int row) {      // tr_swz<128> of wgrad_gemm.hip
    return (((row >> 1) & 1) | (((row >> 3) & 1) << 1)) << 1;
}
__device__ __forceinline__ int wh_swz_g(int slot) { return ((slot >> 1) & 3) << 1; }
__device__ __forceinline__ int wh_plane(int ph, int pw) { return ph ? (pw ? WH_P11 : WH_P10) : (pw ? WH_P01 : WH_P00); }

template <int N> __device__ __forceinline__ void wh_wait_barrier() {
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}

__global__ __launch_bounds__(512, 1) void wgrad_halo_k(const WhArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int na = p.Ca >> 6, nb = p.Cb >> 6, ntile = na * nb;
    // XCD x takes K-slices x, x + 8, ..: all channel tiles of a K-slice run on one XCD and share its pixels in that L2
    const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
    const int tile = loc % ntile, ks = (loc / ntile) * 8 + xcd;
    if (ks >= p.ksplit) return;
    const int a0 = (tile / nb) * 64, b0 = (tile % nb) * 64;
    const int blk0 = ks * p.per, blk1 = min(blk0 + p.per, p.nblk);
    const int nsteps = max(blk1 - blk0, 0);
    const int IH = 2 * p.OH, IW = 2 * p.OW;

    // ---- producer roles (constant over the blocks): WH_DI instructions per wave; instruction i = 4 w + j of a stage is
    // S rows 8 i .. +7 for i < 4 (wave 0), patch slots 8 (i - 4) .. +7 otherwise
    const bool s_wave = w == 0;
    int d_r[WH_DI], d_c[WH_DI], d_off[WH_DI];      // pixel offset from the block's corner (d_r < -100: slot never read), byte offset in the row
#pragma unroll
    for (int j = 0; j < WH_DI; ++j) {
        if (s_wave) {
            const int k = 8 * j + (lane >> 3);                     // row of the S tile: (rr = j, cc = lane / 8)
            d_r[j] = j; d_c[j] = lane >> 3;
            d_off[j] = (((lane & 7) ^ wh_swz_s(k)) * 16) + a0 * 2;
        } else {
            const int slot = 8 * (4 * w + j - 4) + (lane >> 3);
            int ph, pw, rem;
            if (slot < WH_P01) { ph = 0; pw = 0; rem = slot - WH_P00; }
            else if (slot < WH_P10) { ph = 0; pw = 1; rem = slot - WH_P01; }
            else if (slot < WH_P11) { ph = 1; pw = 0; rem = slot - WH_P10; }
            else { ph = 1; pw = 1; rem = slot - WH_P11; }
            const int pr = rem / WH_PW, pc = rem - pr * WH_PW;
            const int u = 2 * pr + ph, v = 2 * pc + pw;
            // never-read slots inside a plane row are skipped lane by lane; the 8 padding slots behind the planes are one whole
            // instruction and take the zero row (every wave issues WH_DI instructions per stage: the vmcnt waits count them)
            const bool used = slot >= 216 || (u <= 2 * WH_BH && v <= 2 * WH_BW);
            d_r[j] = used ? (slot >= 216 ? -50 : u) : -1000; d_c[j] = v;
            d_off[j] = (((lane & 7) ^ wh_swz_g(slot)) * 16) + b0 * 2;
        }
    }
    const unsigned char* const d_base = s_wave ? p.S : p.G;
    const int d_H = s_wave ? p.OH : IH, d_W = s_wave ? p.OW : IW;
    const size_t d_ld = (size_t)(s_wave ? p.lds : p.ldg) * 2;
    auto stage = [&](int step) {
        const int blk = blk0 + step;
        const int n = blk / (p.BR * p.BC), rem = blk - n * (p.BR * p.BC);
        const int r0 = (rem / p.BC) * WH_BH, c0 = (rem % p.BC) * WH_BW;
        const int rb = s_wave ? r0 : 2 * r0 - 1, cb = s_wave ? c0 : 2 * c0 - 1;
        unsigned char* ld = smem + (step & (WH_RING - 1)) * WH_STAGE + (4 * w) * 1024;
#pragma unroll
        for (int j = 0; j < WH_DI; ++j) {
            const int r = rb + d_r[j], c = cb + d_c[j];
            const bool v = (unsigned)r < (unsigned)d_H && (unsigned)c < (unsigned)d_W;
            const unsigned char* src = d_base + ((size_t)(n * d_H + r) * d_W + c) * d_ld + d_off[j];
#if WH_ABL != 1
            if (d_r[j] > -100) wh_glds16(v ? src : p.zero, ld + j * 1024);     // slots no tap reads stay unfilled
#endif
        }
    };

    // ---- consumer roles: wave w owns b sub-tile w % 4 of the taps w / 4 + 2 j (waves 0-3: five taps, 4-7: four)
    const int fi = lane & 15, fg = lane >> 4;
    const int q = fi >> 2, pp = fi & 3;
    const int th = w >> 2, bs = w & 3;
    int offA[4];
    {
        const int row = 8 * fg + q;                      // + 4 for the second read
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int chunk = (mt * 2 + (pp >> 1)) ^ wh_swz_s(row);
            offA[mt] = row * 128 + chunk * 16 + (pp & 1) * 8;
        }
    }
    int offBl[WH_UNITS], offBh[WH_UNITS];               // pixel (rr = fg, cc = q) and (rr = fg, cc = q + 4) of unit j's tap
#pragma unroll
    for (int j = 0; j < WH_UNITS; ++j) {
        const int tap = min(th + 2 * j, 8), kh = tap / 3, kw = tap - 3 * kh;
        const int sl = wh_plane(kh & 1, kw & 1) + (fg + (kh >> 1)) * WH_PW + (kw >> 1) + q, sh = sl + 4;
        offBl[j] = WH_S_BYTES + sl * 128 + (((bs * 2 + (pp >> 1)) ^ wh_swz_g(sl)) * 16) + (pp & 1) * 8;
        offBh[j] = WH_S_BYTES + sh * 128 + (((bs * 2 + (pp >> 1)) ^ wh_swz_g(sh)) * 16) + (pp & 1) * 8;
    }
    const bool unit4 = th == 0;          // waves 4-7 have no fifth tap: its reads are issued (uniform wait counts), its MFMAs are not

    wh_f32x4_t acc[WH_UNITS][4];
#pragma unroll
    for (int j = 0; j < WH_UNITS; ++j)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[j][mt] = wh_f32x4_t{0.f, 0.f, 0.f, 0.f};

    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    // the 8 + 10 transposing reads of one block
    auto read_frags = [&](int step, wh_s16x4_t (&al)[4], wh_s16x4_t (&ah)[4], wh_s16x4_t (&bl)[WH_UNITS], wh_s16x4_t (&bh)[WH_UNITS]) {
        const unsigned lb = lds0 + (step & (WH_RING - 1)) * WH_STAGE;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const unsigned ad = lb + offA[mt];
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(al[mt]) : "v"(ad));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:512" : "=v"(ah[mt]) : "v"(ad));
        }
#pragma unroll
        for (int j = 0; j < WH_UNITS; ++j) {
            const unsigned adl = lb + offBl[j], adh = lb + offBh[j];
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(bl[j]) : "v"(adl));
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(bh[j]) : "v"(adh));
        }
    };
    // at most YOUNGER reads (of the next block) outstanding: this block's have landed.  The wait is tied to the registers it
    // guards; the fragments leave as MFMA operands.
    auto landed = [&](auto younger_tag, wh_s16x4_t (&al)[4], wh_s16x4_t (&ah)[4], wh_s16x4_t (&bl)[WH_UNITS],
                      wh_s16x4_t (&bh)[WH_UNITS], wh_bf16x8_t (&fa)[4], wh_bf16x8_t (&fb)[WH_UNITS]) {
        constexpr int YOUNGER = decltype(younger_tag)::value;
        asm volatile("s_waitcnt lgkmcnt(%18)"
                     : "+v"(al[0]), "+v"(ah[0]), "+v"(al[1]), "+v"(ah[1]), "+v"(al[2]), "+v"(ah[2]), "+v"(al[3]), "+v"(ah[3]),
                       "+v"(bl[0]), "+v"(bh[0]), "+v"(bl[1]), "+v"(bh[1]), "+v"(bl[2]), "+v"(bh[2]), "+v"(bl[3]), "+v"(bh[3]),
                       "+v"(bl[4]), "+v"(bh[4])
                     : "n"(YOUNGER));
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
            fa[mt] = wh_bf16x8_t{al[mt][0], al[mt][1], al[mt][2], al[mt][3], ah[mt][0], ah[mt][1], ah[mt][2], ah[mt][3]};
#pragma unroll
        for (int j = 0; j < WH_UNITS; ++j)
            fb[j] = wh_bf16x8_t{bl[j][0], bl[j][1], bl[j][2], bl[j][3], bh[j][0], bh[j][1], bh[j][2], bh[j][3]};
    };
    auto mma = [&](const wh_bf16x8_t (&fa)[4], const wh_bf16x8_t (&fb)[WH_UNITS]) {
#pragma unroll
        for (int j = 0; j < WH_UNITS - 1; ++j)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
                acc[j][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[mt], acc[j][mt], 0, 0, 0);
        if (unit4) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
                acc[WH_UNITS - 1][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[WH_UNITS - 1], fa[mt], acc[WH_UNITS - 1][mt], 0, 0, 0);
        }
    };

    if (nsteps > 0) {
        using Y0 = std::integral_constant<int, 0>;
        wh_s16x4_t xal[4], xah[4], xbl[WH_UNITS], xbh[WH_UNITS], yal[4], yah[4], ybl[WH_UNITS], ybh[WH_UNITS];
        wh_bf16x8_t fa[4], fb[WH_UNITS];
        stage(0);
        if (nsteps > 1) stage(1);
        if (nsteps > 2) stage(2);
        if (nsteps > 2) wh_wait_barrier<2 * WH_DI>(); else if (nsteps > 1) wh_wait_barrier<WH_DI>(); else wh_wait_barrier<0>();
        // Four blocks per loop iteration; NOTHING asynchronous crosses the loop's back edge (a fragment register still in
        // flight there is what the compiler copies when it splits a live range: isa_check rejects the listing), so the
        // first block of an iteration reads its own fragments (their latency sits under the barrier wait) and blocks
        // 2-4 find theirs issued one block ahead.  Block s: stage s + 1 landed (own LDS-DMA) -> barrier (everyone's;
        // everyone is past the reads of stage s - 1, whose ring slot stage s + 3 takes) -> issue stage s + 3 -> fragments
        // of s -> reads of s + 1 -> MFMAs of s.  (The wait for s's fragments must come BEFORE the reads of s + 1: behind
        // them it would have to let 18 younger reads pass, which the 4-bit lgkmcnt cannot say.)
        auto sync_and_stage = [&](int s) {
            if (s + 1 < nsteps) {
                if (s + 2 < nsteps) wh_wait_barrier<WH_DI>(); else wh_wait_barrier<0>();
                if (s + 3 < nsteps) stage(s + 3);
            }
        };
        for (int s0 = 0; s0 < nsteps; s0 += 4) {
#if WH_ABL != 2
            read_frags(s0, xal, xah, xbl, xbh);
#endif
            sync_and_stage(s0);
#if WH_ABL != 2
            landed(Y0{}, xal, xah, xbl, xbh, fa, fb);
            if (s0 + 1 < nsteps) read_frags(s0 + 1, yal, yah, ybl, ybh);
            mma(fa, fb);
#endif
            if (s0 + 1 < nsteps) {
                sync_and_stage(s0 + 1);
#if WH_ABL != 2
                landed(Y0{}, yal, yah, ybl, ybh, fa, fb);
                if (s0 + 2 < nsteps) read_frags(s0 + 2, xal, xah, xbl, xbh);
                mma(fa, fb);
#endif
            }
            if (s0 + 2 < nsteps) {
                sync_and_stage(s0 + 2);
#if WH_ABL != 2
                landed(Y0{}, xal, xah, xbl, xbh, fa, fb);
                if (s0 + 3 < nsteps) read_frags(s0 + 3, yal, yah, ybl, ybh);
                mma(fa, fb);
#endif
            }
            if (s0 + 3 < nsteps) {
                sync_and_stage(s0 + 3);
#if WH_ABL != 2
                landed(Y0{}, yal, yah, ybl, ybh, fa, fb);
                mma(fa, fb);
#endif
            }
        }
    }

    // ---- slab: D[row = b 4 fg + r][col = a fi]: a lane owns 4 consecutive b of one a -> one 16-byte store
    float* slab = p.dW + (size_t)ks * p.Ca * 9 * p.Cb;
#pragma unroll
    for (int j = 0; j < WH_UNITS; ++j) {
        const int tap = th + 2 * j;
        if (tap > 8) continue;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int a = a0 + mt * 16 + fi, b = b0 + bs * 16 + 4 * fg;
            *(wh_f32x4_t*)(slab + ((size_t)a * 9 + tap) * p.Cb + b) = acc[j][mt];
        }
    }
}

static int wh_ok(int dtype, int Nimg, int OH, int OW, int Ca, int Cb) {
    return dtype == RBVAE_BF16 && Nimg >= 1 && OH >= 1 && OW >= 1 && Ca >= 64 && Cb >= 64 && Ca % 64 == 0 && Cb % 64 == 0 &&
           (long)Nimg * OH * OW * 4 < (1l << 31);
}

}  // namespace rbvae

using namespace rbvae;

extern "C" {

int rbvae_wgrad3x3s2_halo_ok(int dtype, int Nimg, int OH, int OW, int Ca, int Cb) { return wh_ok(dtype, Nimg, OH, OW, Ca, Cb); }

/* 4 x 8 pixel blocks the K loop walks (the caller sizes ksplit against it) */
int rbvae_wgrad3x3s2_halo_blocks(int Nimg, int OH, int OW) { return Nimg * ((OH + WH_BH - 1) / WH_BH) * ((OW + WH_BW - 1) / WH_BW); }

int rbvae_wgrad3x3s2_halo(int dtype, const void* S, const void* G, float* dW_slabs, const void* zero_page, int Nimg, int OH,
                          int OW, int Ca, int Cb, int lds_, int ldg, int ksplit, void* stream) {
    RBVAE_CHECK_ARG(S && G && dW_slabs && zero_page, "wgrad3x3s2_halo: null pointer");
    RBVAE_CHECK_ARG(wh_ok(dtype, Nimg, OH, OW, Ca, Cb), "wgrad3x3s2_halo: shape not covered (dtype %d, %d x %d x %d, %d x %d channels): "
                    "query rbvae_wgrad3x3s2_halo_ok", dtype, Nimg, OH, OW, Ca, Cb);
    RBVAE_CHECK_ARG(lds_ >= Ca && ldg >= Cb && lds_ % 8 == 0 && ldg % 8 == 0, "wgrad3x3s2_halo: leading dimensions lds=%d ldg=%d", lds_, ldg);
    RBVAE_CHECK_ARG(((uintptr_t)S | (uintptr_t)G | (uintptr_t)dW_slabs | (uintptr_t)zero_page) % 16 == 0,
                    "wgrad3x3s2_halo: pointers must be 16-byte aligned");
    WhArgs a;
    a.S = (const unsigned char*)S; a.G = (const unsigned char*)G; a.dW = dW_slabs; a.zero = (const unsigned char*)zero_page;
    a.Nimg = Nimg; a.OH = OH; a.OW = OW; a.Ca = Ca; a.Cb = Cb; a.lds = lds_; a.ldg = ldg;
    a.BR = (OH + WH_BH - 1) / WH_BH; a.BC = (OW + WH_BW - 1) / WH_BW; a.nblk = Nimg * a.BR * a.BC;
    RBVAE_CHECK_ARG(ksplit >= 1 && ksplit <= a.nblk, "wgrad3x3s2_halo: ksplit=%d (1 .. %d blocks)", ksplit, a.nblk);
    a.per = (a.nblk + ksplit - 1) / ksplit;
    a.ksplit = ksplit;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)wgrad_halo_k, hipFuncAttributeMaxDynamicSharedMemorySize, WH_RING * WH_STAGE);
        attr_set = true;
    }
    // K-slices in groups of 8 (one per XCD); workgroups of the padding K-slices return at once
    const long blocks = (long)(Ca / 64) * (Cb / 64) * 8 * ((ksplit + 7) / 8);
    hipLaunchKernelGGL(wgrad_halo_k, dim3((unsigned)blocks), dim3(512), WH_RING * WH_STAGE, (hipStream_t)stream, a);
    RBVAE_CHECK_LAUNCH("wgrad3x3s2_halo");
    return RBVAE_OK;
}

}  // extern "C"
