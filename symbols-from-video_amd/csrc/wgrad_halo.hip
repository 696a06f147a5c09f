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
// 16 x 16, 4-5 per wave, 80 accumulators per lane) and walks 8 x 8 blocks of low-resolution pixels: per block one
// [64 px][64 a] tile of S and the 17 x 17 patch of G around it come in ONCE by LDS-DMA (double-buffered, one barrier per
// block) and serve all taps.
//
// LDS images (128-byte pixel rows, reduction index = row, fragments by ds_read_b64_tr_b16 as in wgrad_gemm.hip):
//   S tile   row k = 8 rr + cc, 16-byte chunks XOR-swizzled by tr_swz (wgrad_gemm.hip).
//   G patch  four parity planes (row parity, column parity) of 9 x 12 slots (112 with padding): patch pixel (u, v) lives
//            in plane (u & 1, v & 1) at slot (u >> 1) * 12 + (v >> 1), so tap (kh, kw) of block pixel (rr, cc) is slot
//            plane(kh & 1, kw & 1) + (rr + (kh >> 1)) * 12 + cc + (kw >> 1): consecutive pixels of a row are consecutive
//            slots although the convolution strides by two, a tap is a constant offset, and with a row pitch of
//            12 = 4 (mod 8) slots and the chunk pair XORed by (slot >> 1) & 3 the eight pixel rows a 32-lane half of
//            a transposing read touches fall on eight distinct 32-byte bank groups (conflict-free).
// Each K-slice writes its own f32 slab [a][t][b] (rbvae_wgrad_gemm's layout: the same fixed-order reduction jobs follow).
#include "common.h"
#include <stdlib.h>

#ifndef WH_ABL          // timing ablations (results wrong on purpose): 1 no LDS-DMA, 2 no fragment reads / MFMAs
#define WH_ABL 0
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
    int BR, BC, nblk, per;     // 8 x 8 blocks per image (rows, columns), blocks in all, blocks per K-slice
};

constexpr int WH_PW = 12, WH_PLANE = 112, WH_SLOTS = 4 * WH_PLANE;   // 448 slots = 56 KB
constexpr int WH_S_BYTES = 64 * 128, WH_G_BYTES = WH_SLOTS * 128, WH_STAGE = WH_S_BYTES + WH_G_BYTES;   // 64 KB
constexpr int WH_GI = WH_SLOTS / 64;        // patch LDS-DMA instructions per wave (8 slots each): 7
constexpr int WH_UNITS = 5;                 // (tap, b sub-tile) units per wave: taps w/4 + 2j, sub-tile w % 4

__device__ __forceinline__ void wh_glds16(const void* g, void* lds) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}
__device__ __forceinline__ int wh_swz_s(int row) {      // tr_swz<128> of wgrad_gemm.hip
    return (((row >> 1) & 1) | (((row >> 3) & 1) << 1)) << 1;
}
__device__ __forceinline__ int wh_swz_g(int slot) { return ((slot >> 1) & 3) << 1; }

__global__ __launch_bounds__(512, 1) void wgrad_halo_k(const WhArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int na = p.Ca >> 6, nb = p.Cb >> 6, ntile = na * nb;
    const int wg = blockIdx.x;
    const int tile = wg % ntile, ks = wg / ntile;
    const int a0 = (tile / nb) * 64, b0 = (tile % nb) * 64;
    const int blk0 = ks * p.per, blk1 = min(blk0 + p.per, p.nblk);
    const int IH = 2 * p.OH, IW = 2 * p.OW;

    // ---- producer roles (constant over the blocks): one S instruction and WH_GI patch instructions per wave
    // S: lane -> pixel k = 8 w + lane / 8 of the block, LDS chunk lane % 8 holds source chunk (lane % 8) ^ swz(k)
    const int s_k = 8 * w + (lane >> 3);
    const int s_rr = s_k >> 3, s_cc = s_k & 7;
    const int s_coff = (((lane & 7) ^ wh_swz_s(s_k)) * 16) + a0 * 2;
    // G: instruction j covers slots 8 (w * WH_GI + j) .. +7
    int g_du[WH_GI], g_dv[WH_GI], g_coff[WH_GI];      // patch pixel (u, v) of this lane's slot (u < 0: padding slot), byte offset in the row
#pragma unroll
    for (int j = 0; j < WH_GI; ++j) {
        const int slot = 8 * (w * WH_GI + j) + (lane >> 3);
        const int pl = slot / WH_PLANE, rem = slot - pl * WH_PLANE;
        const int pr = rem / WH_PW, pc = rem - pr * WH_PW;
        const int u = 2 * pr + (pl >> 1), v = 2 * pc + (pl & 1);
        const bool in_patch = rem < 9 * WH_PW && u <= 16 && v <= 16;
        g_du[j] = in_patch ? u : -1000;
        g_dv[j] = v;
        g_coff[j] = (((lane & 7) ^ wh_swz_g(slot)) * 16) + b0 * 2;
    }
    auto stage = [&](int blk, int buf) {
        const int n = blk / (p.BR * p.BC), rem = blk - n * (p.BR * p.BC);
        const int r0 = (rem / p.BC) * 8, c0 = (rem % p.BC) * 8;
        unsigned char* ls = smem + buf * WH_STAGE;
        {
            const int r = r0 + s_rr, c = c0 + s_cc;
            const bool v = r < p.OH && c < p.OW;
            const unsigned char* src = p.S + ((size_t)(n * p.OH + r) * p.OW + c) * ((size_t)p.lds * 2) + s_coff;
#if WH_ABL != 1
            wh_glds16(v ? src : p.zero, ls + w * 1024);
#endif
        }
        unsigned char* lg = ls + WH_S_BYTES + (w * WH_GI) * 1024;
        const int hr0 = 2 * r0 - 1, hc0 = 2 * c0 - 1;
#pragma unroll
        for (int j = 0; j < WH_GI; ++j) {
            const int hr = hr0 + g_du[j], hc = hc0 + g_dv[j];
            const bool v = (unsigned)hr < (unsigned)IH && (unsigned)hc < (unsigned)IW;
            const unsigned char* src = p.G + ((size_t)(n * IH + hr) * IW + hc) * ((size_t)p.ldg * 2) + g_coff[j];
#if WH_ABL != 1
            wh_glds16(v ? src : p.zero, lg + j * 1024);
#endif
        }
    };

    // ---- consumer roles: wave w owns b sub-tile w % 4 of the taps w / 4 + 2 j (waves 0-3: five taps, 4-7: four)
    const int fi = lane & 15, fg = lane >> 4;
    const int q = fi >> 2, pp = fi & 3;
    const int th = w >> 2, bs = w & 3;
    // A (S tile): sub-tile mt = a channels 16 mt .. +15; row 8 fg + q (+ 4 for the second read)
    int offA[4];
    {
        const int row = 8 * fg + q;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int chunk = (mt * 2 + (pp >> 1)) ^ wh_swz_s(row);
            offA[mt] = row * 128 + chunk * 16 + (pp & 1) * 8;
        }
    }
    // B (patch): unit j = tap th + 2 j; pixel (rr = fg (+ 4 per 32-pixel half), cc = q (+ 4 for the second read))
    int offBl[WH_UNITS], offBh[WH_UNITS];
#pragma unroll
    for (int j = 0; j < WH_UNITS; ++j) {
        const int tap = min(th + 2 * j, 8), kh = tap / 3, kw = tap - 3 * kh;
        const int base = ((kh & 1) * 2 + (kw & 1)) * WH_PLANE + (fg + (kh >> 1)) * WH_PW + (kw >> 1) + q;
        const int sl = base, sh = base + 4;
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
    // one 32-pixel half (block rows 4 half .. +3): 8 + 10 transposing reads
    auto read_half = [&](unsigned lb, int half, wh_s16x4_t (&al)[4], wh_s16x4_t (&ah)[4], wh_s16x4_t (&bl)[WH_UNITS],
                         wh_s16x4_t (&bh)[WH_UNITS]) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const unsigned ad = lb + offA[mt] + half * (32 * 128);
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(al[mt]) : "v"(ad));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:512" : "=v"(ah[mt]) : "v"(ad));
        }
#pragma unroll
        for (int j = 0; j < WH_UNITS; ++j) {
            const unsigned adl = lb + offBl[j] + half * (4 * WH_PW * 128), adh = lb + offBh[j] + half * (4 * WH_PW * 128);
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(bl[j]) : "v"(adl));
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(bh[j]) : "v"(adh));
        }
    };
    // every read of the half has landed: the wait is tied to the registers it guards, the fragments leave as MFMA operands
    auto landed = [&](wh_s16x4_t (&al)[4], wh_s16x4_t (&ah)[4], wh_s16x4_t (&bl)[WH_UNITS], wh_s16x4_t (&bh)[WH_UNITS],
                      wh_bf16x8_t (&fa)[4], wh_bf16x8_t (&fb)[WH_UNITS]) {
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(al[0]), "+v"(ah[0]), "+v"(al[1]), "+v"(ah[1]), "+v"(al[2]), "+v"(ah[2]), "+v"(al[3]), "+v"(ah[3]),
                       "+v"(bl[0]), "+v"(bh[0]), "+v"(bl[1]), "+v"(bh[1]), "+v"(bl[2]), "+v"(bh[2]), "+v"(bl[3]), "+v"(bh[3]),
                       "+v"(bl[4]), "+v"(bh[4]));
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
            fa[mt] = wh_bf16x8_t{al[mt][0], al[mt][1], al[mt][2], al[mt][3], ah[mt][0], ah[mt][1], ah[mt][2], ah[mt][3]};
#pragma unroll
        for (int j = 0; j < WH_UNITS; ++j)
            fb[j] = wh_bf16x8_t{bl[j][0], bl[j][1], bl[j][2], bl[j][3], bh[j][0], bh[j][1], bh[j][2], bh[j][3]};
    };
    auto mma_half = [&](const wh_bf16x8_t (&fa)[4], const wh_bf16x8_t (&fb)[WH_UNITS]) {
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

    if (blk0 < blk1) {
        stage(blk0, 0);
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        wh_s16x4_t a0l[4], a0h[4], b0l[WH_UNITS], b0h[WH_UNITS], a1l[4], a1h[4], b1l[WH_UNITS], b1h[WH_UNITS];
        wh_bf16x8_t fa[4], fb[WH_UNITS];
        int buf = 0;
        for (int blk = blk0; blk < blk1; ++blk) {
            const unsigned lcur = lds0 + buf * WH_STAGE;
            if (blk + 1 < blk1) stage(blk + 1, buf ^ 1);         // the other buffer was released by the previous barrier
#if WH_ABL != 2
            read_half(lcur, 0, a0l, a0h, b0l, b0h);
            landed(a0l, a0h, b0l, b0h, fa, fb);
            read_half(lcur, 1, a1l, a1h, b1l, b1h);              // in flight under the first half's MFMAs
            mma_half(fa, fb);
            landed(a1l, a1h, b1l, b1h, fa, fb);
            mma_half(fa, fb);
#endif
            // next block's operands landed (this wave's LDS-DMA), every wave done with this buffer
            asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            buf ^= 1;
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

/* 8 x 8 pixel blocks the K loop walks (the caller sizes ksplit against it) */
int rbvae_wgrad3x3s2_halo_blocks(int Nimg, int OH, int OW) { return Nimg * ((OH + 7) / 8) * ((OW + 7) / 8); }

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
    a.BR = (OH + 7) / 8; a.BC = (OW + 7) / 8; a.nblk = Nimg * a.BR * a.BC;
    RBVAE_CHECK_ARG(ksplit >= 1 && ksplit <= a.nblk, "wgrad3x3s2_halo: ksplit=%d (1 .. %d blocks)", ksplit, a.nblk);
    a.per = (a.nblk + ksplit - 1) / ksplit;
    a.ksplit = ksplit;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)wgrad_halo_k, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * WH_STAGE);
        attr_set = true;
    }
    const long blocks = (long)(Ca / 64) * (Cb / 64) * ksplit;
    hipLaunchKernelGGL(wgrad_halo_k, dim3((unsigned)blocks), dim3(512), 2 * WH_STAGE, (hipStream_t)stream, a);
    RBVAE_CHECK_LAUNCH("wgrad3x3s2_halo");
    return RBVAE_OK;
}

}  // extern "C"
