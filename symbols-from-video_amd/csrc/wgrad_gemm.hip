// Weight-gradient GEMM on the gfx950 matrix cores:
//
//   dW[ks][co][t][ci] = sum_{p in K-slice ks} Dy[p][co] * In[idx[t][p]][ci]
//
// for Conv2d / ConvTranspose2d / Linear weight gradients of the RBVAE path (autograd of
// percep_RBVAE_model.py:51-57,61,74,76-82 as run by percep_RBVAE_train.py:552).
// Both operands are pixel-major ("k-major") NHWC rows, so the reduction index is
// the LDS image's ROW: fragments are read with the transposing LDS read
// (ds_read_b64_tr_b16) for bf16 and with plain dword reads for f32.
//
// Tile: 128 x (64*NT) (co x ci) per 512-thread workgroup (8 waves, two per SIMD), 32 pixels per K
// step, a 3-deep LDS-DMA ring with counted vmcnt waits; the K-slice's gather indices are read into
// LDS once so no register load sits between the LDS-DMA issues.  grid = (co tiles, ci tiles, taps * ksplit); each
// K-slice writes its own f32 slab (summed later in a fixed order by
// rbvae_permute_reduce, so gradients are bitwise reproducible -- no float atomics).
#include "common.h"
#include <type_traits>
#include <utility>

namespace rbvae {

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

struct WgArgs {
    const unsigned char* Dy;   // [P][ldy] T
    const unsigned char* In;   // [*][ldi] T
    float* dW;                 // [ksplit][Co][taps][Ci] f32
    const int* idx;            // [taps][P] row of In per (tap, pixel), -1 = zero row; null = identity
    int in_rows;               // rows of In: indices outside [0, in_rows) read the zero row
    const unsigned char* zero; // >= 16 zero bytes
    int P, Co, Ci, ldy, ldi, taps, ksplit, Pper;
    int xcd_order;             // 1: workgroup id -> (K-slice, tile) so that one XCD's workgroups cover <= 2 K-slices
    unsigned long long* stamps;   // -DWG_STAMPS=1 builds only (rbvae_dbg_wg_stamps): [workgroup][8] phase stamps, 100 MHz
};

__device__ __forceinline__ void glds16w(const void* g, void* lds) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}

// XOR applied to the 16-B chunk index of an LDS image row so that the transposed
// reads of a 32-lane half (rows {q, 8+q} or {4+q, 12+q}) hit distinct banks.
template <int RB> __device__ __forceinline__ int tr_swz(int row) {
    if constexpr (RB >= 256) return ((row & 3) | (((row >> 3) & 1) << 2)) << 1;   // 8 chunk pairs
    else return (((row >> 1) & 1) | (((row >> 3) & 1) << 1)) << 1;                // 128-B rows: 4 pairs
}

// Measured and not kept (round 2, same-GPU A/B of the bench step; the variants are gone from the source): scheduling barriers
// around the MFMA groups 0.735 vs 0.74 us per K step, the two waves of a SIMD staging at different points of the step, the
// next stage's LDS-DMA pieces issued between the MFMAs 0.737, gather indices read one stage ahead 0.5395 vs 0.5378 ms per
// step, raised priority for few-workgroup launches 0.522 vs 0.516 ms per step.
#ifndef WG_STAMPS
#define WG_STAMPS 0
#endif
// (The timing ablations of rounds 2-3 -- fill only 0.41 us per K step, fragment reads + MFMAs only 0.48, both 0.74; slab
// stores of K-slice 0 only: -2.6 % of the step -- are recorded in DESIGN.md section 5 and profiles/r03_wgrad_noslab_ablation.txt;
// their wrong-result code paths are gone from this file.)
#if WG_STAMPS
#define WG_STAMP(slot) do { if (p.stamps && threadIdx.x == 0) p.stamps[(size_t)blockIdx.x * 8 + (slot)] = wall_clock64(); } while (0)
#else
#define WG_STAMP(slot) do {} while (0)
#endif

constexpr int WG_BM = 128;      // co per workgroup
constexpr int WG_MAXP = 4096;   // pixels of one K-slice (their gather indices live in LDS)

template <class F, int... Ks> __device__ __forceinline__ void wg_static_for(F&& f, std::integer_sequence<int, Ks...>) {
    (f(std::integral_constant<int, Ks>{}), ...);
}

template <int N> __device__ __forceinline__ void wg_wait_barrier() {
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}

// 8 waves as 2 (co) x 4 (ci); a wave owns 64 co x 16*NT ci; BN = 64*NT ci per workgroup.
// WG_NS = LDS ring depth.  3: two K steps in flight behind the one being multiplied, one workgroup per CU;
// 2: double buffer, two workgroups per CU (grids of more than 256 workgroups).
//
// Workgroup order: the grid is one-dimensional; with xcd_order (the default when ksplit > 1) every XCD's workgroups
// work on one or two pixel slices, so all taps and channel tiles of a slice re-read its Dy / In rows from that XCD's
// own L2 instead of from the Infinity Cache; otherwise the K-slice index is the fastest digit of the workgroup id.
// BM = co per workgroup: 128, or 64 (f32 only) for layers of <= 64 output channels -- with the 128-row tile half of the
// waves multiplied zero rows, and the exact-f32 kernel is bound by its matrix-core time (cfg 3, 64 channels: 45 % of the f32 step)
template <typename T, int NT, int WG_NS, int BM = WG_BM>
__global__ __launch_bounds__(512, WG_NS == 2 ? 2 : 1) void wgrad_gemm_k(const WgArgs p) {
    constexpr int ES = sizeof(T);
    constexpr int WG_BK = (ES == 2) ? 64 : 32;            // pixels per K step (one or two 32-pixel MFMA steps)
    constexpr int MT = BM / 32;
    static_assert(BM == 128 || (BM == 64 && ES == 4), "64-row tiles: the f32 path only (the bf16 fragment guards list MT = 4 operands)");
    constexpr int BN = 64 * NT;
    constexpr int RBA = BM * ES, RBB = BN * ES;           // image row bytes
    constexpr int A_BYTES = WG_BK * RBA, B_BYTES = WG_BK * RBB;
    constexpr int STAGE = A_BYTES + B_BYTES;
    constexpr int A_TOT = A_BYTES / 1024, B_TOT = B_BYTES / 1024;   // LDS-DMA instructions per stage
    constexpr int A_INSTR = A_TOT / 8;                               // per wave (A_TOT is 16)
    constexpr int B_INSTR = (B_TOT + 7) / 8;                         // waves >= B_TOT issue none when B_TOT < 8
    constexpr bool SWZ = (ES == 2);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int* s_idx = (int*)(smem + WG_NS * STAGE);            // [WG_MAXP]

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);     // provably uniform: scalar LDS addressing
    WG_STAMP(0);
    const int gx = (p.Co + BM - 1) / BM, gy = (p.Ci + BN - 1) / BN;
    int ks, wg;
    if (p.xcd_order) {
        // Workgroups are dealt to the 8 XCDs round-robin by id; XCD x takes the x-th eighth of the (K-slice, tile)
        // items in slice-major order, so its workgroups share one or two pixel slices of Dy / In and re-read them from
        // that XCD's L2 (with the K-slice as the fastest digit every XCD walks every slice).  Placement is a speed
        // matter only: every item is computed whichever XCD takes it.
        const int ntiles = gx * gy * p.taps, total = ntiles * p.ksplit, per = (total + 7) >> 3;
        const int x = blockIdx.x & 7, q = blockIdx.x >> 3;
        const int item = x * per + q;
        if (q >= per || item >= total) return;         // the grid is 8 * per workgroups (uniform exit, before any barrier)
        ks = item / ntiles;
        wg = item - ks * ntiles;
    } else {
        wg = blockIdx.x;
        ks = wg % p.ksplit; wg /= p.ksplit;
    }
    const int bx = wg % gx; wg /= gx;
    const int by = wg % gy;
    const int tap = wg / gy;
    const int co0 = bx * BM, ci0 = by * BN;
    const int pbeg = ks * p.Pper;
    const int pend = min(p.P, pbeg + p.Pper);
    const int npix = max(pend - pbeg, 0);
    const int nsteps = (npix + WG_BK - 1) / WG_BK;

    // gather indices of this K-slice -> LDS (identity when there is no table)
    {
        const int* idx = p.idx ? p.idx + (size_t)tap * p.P + pbeg : nullptr;
        const int padded = nsteps * WG_BK;
        // an index outside [0, in_rows) reads the zero row: a table that is stale, half-built or built for another shape
        // gives wrong sums (the parity tests see those), never an out-of-bounds access
        for (int i = tid; i < padded; i += 512) {
            int r = i < npix ? (idx ? idx[i] : pbeg + i) : -1;
            if ((unsigned)r >= (unsigned)p.in_rows) r = -1;
            s_idx[i] = r;
        }
    }
    __syncthreads();
    WG_STAMP(1);

    // staging roles.  One instruction = 1 KiB = (1024/RB) image rows.
    constexpr int A_LPR = RBA / 16, B_LPR = RBB / 16;     // lanes (chunks) per row
    constexpr int A_RPI = 64 / A_LPR, B_RPI = 64 / B_LPR; // rows per instruction
    int a_row[A_INSTR], a_coff[A_INSTR];
    bool a_cval[A_INSTR];
#pragma unroll
    for (int i = 0; i < A_INSTR; ++i) {
        const int r = (w * A_INSTR + i) * A_RPI + lane / A_LPR;
        const int c = (lane % A_LPR) ^ (SWZ ? tr_swz<RBA>(r) : 0);
        a_row[i] = r;
        a_coff[i] = c * 16;
        a_cval[i] = co0 + c * (16 / ES) < p.Co;
    }
    const bool b_wave = (w * B_INSTR) < B_TOT;            // does this wave stage part of B?
    int b_row[B_INSTR], b_coff[B_INSTR];
    bool b_cval[B_INSTR];
#pragma unroll
    for (int i = 0; i < B_INSTR; ++i) {
        const int r = (w * B_INSTR + i) * B_RPI + lane / B_LPR;
        const int c = (lane % B_LPR) ^ (SWZ ? tr_swz<RBB>(r & (WG_BK - 1)) : 0);
        b_row[i] = r & (WG_BK - 1);
        b_coff[i] = c * 16;
        b_cval[i] = b_wave && ci0 + c * (16 / ES) < p.Ci;
    }
    // producer state: A rows advance by a constant stride per K step (one 64-bit add each); B rows are
    // gathered through the index table in LDS
    const size_t a_stride = (size_t)WG_BK * p.ldy * ES;
    const unsigned char* acur[A_INSTR];
#pragma unroll
    for (int i = 0; i < A_INSTR; ++i)
        acur[i] = p.Dy + ((size_t)(pbeg + a_row[i]) * p.ldy + co0) * ES + a_coff[i];
    const size_t ldi_b = (size_t)p.ldi * ES;
    const unsigned char* bbase[B_INSTR];
#pragma unroll
    for (int i = 0; i < B_INSTR; ++i) bbase[i] = p.In + (size_t)ci0 * ES + b_coff[i];
    int pstep = 0, pbuf = 0;
    auto stage_next = [&]() {
        unsigned char* la = smem + pbuf * STAGE + (w * A_INSTR) * 1024;
        const int base = pstep * WG_BK;
#pragma unroll
        for (int i = 0; i < A_INSTR; ++i) {
            const bool v = a_cval[i] && base + a_row[i] < npix;
            glds16w(v ? acur[i] : p.zero, la + i * 1024);
            acur[i] += a_stride;
        }
        if (b_wave) {
            unsigned char* lb = smem + pbuf * STAGE + A_BYTES + (w * B_INSTR) * 1024;
#pragma unroll
            for (int i = 0; i < B_INSTR; ++i) {
                const int src = s_idx[base + b_row[i]];
                const bool v = b_cval[i] && src >= 0;
                glds16w(v ? bbase[i] + (size_t)src * ldi_b : p.zero, lb + i * 1024);
            }
        }
        ++pstep;
        pbuf = (pbuf + 1 == WG_NS) ? 0 : pbuf + 1;
    };
    WG_STAMP(2);
    const int wr = w >> 2, wc = w & 3;
    const int fi = lane & 15, fg = lane >> 4;
    f32x4_t acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // fragment read offsets
    int offA[MT], offB[NT];
    if constexpr (ES == 2) {
        // transposed read: lane (g, q, pp) addresses row 8g+4h+q, elements cb+4pp..+3 of a 16-channel block
        const int q = fi >> 2, pp = fi & 3;
        const int row = 8 * fg + q;                       // + 4h
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int chunk = ((wr * MT + mt) * 2 + (pp >> 1)) ^ tr_swz<RBA>(row);
            offA[mt] = row * RBA + chunk * 16 + (pp & 1) * 8;
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int chunk = ((wc * NT + nt) * 2 + (pp >> 1)) ^ tr_swz<RBB>(row);
            offB[nt] = A_BYTES + row * RBB + chunk * 16 + (pp & 1) * 8;
        }
    } else {
        // f32: lane (i, g) reads element (row 4s+g, channel cb+i)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) offA[mt] = fg * RBA + ((wr * MT + mt) * 16 + fi) * 4;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) offB[nt] = A_BYTES + fg * RBB + ((wc * NT + nt) * 16 + fi) * 4;
    }

    // this wave's LDS-DMA count per stage (waves that stage no B rows issue fewer)
    const int my_loads = A_INSTR + (b_wave ? B_INSTR : 0);
    auto wait_stage = [&](int younger_ok) {
        if (younger_ok) {
            // WG_NS-2 younger stages stay in flight; the immediate must match this wave's own load count
            if (my_loads == A_INSTR + B_INSTR) wg_wait_barrier<(WG_NS - 2) * (A_INSTR + B_INSTR)>();
            else wg_wait_barrier<(WG_NS - 2) * A_INSTR>();
        } else {
            wg_wait_barrier<0>();       // tail: drain (conservative)
        }
    };
    if constexpr (ES == 2) {
        // Transposed fragment reads are issued as inline asm: behind the ds_read_tr16 builtin the compiler
        // conservatively drains ALL LDS-DMA (s_waitcnt vmcnt(0)) before the first read that follows a
        // global_load_lds, which serialises the ring (measured: 1.1 us per K step instead of 0.55).  The asm
        // reads are invisible to its wait-count pass, so each group is followed by an explicit counted
        // s_waitcnt lgkmcnt that is tied ("+v") to the fragment registers it guards; only values that have
        // passed such a wait are packed and handed to the MFMAs.
        const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
        constexpr int RPH = 2 * (MT + NT);          // LDS reads of one 32-pixel half
        static_assert(RPH <= 15, "lgkmcnt is a 4-bit counter");
        auto read_half = [&](unsigned lb, int ksub, s16x4_t (&alo)[MT], s16x4_t (&ahi)[MT], s16x4_t (&blo)[NT],
                             s16x4_t (&bhi)[NT]) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const unsigned ad = lb + offA[mt] + ksub * 32 * RBA;
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(alo[mt]) : "v"(ad));
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(ahi[mt]) : "v"(ad), "n"(4 * RBA));
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const unsigned ad = lb + offB[nt] + ksub * 32 * RBB;
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(blo[nt]) : "v"(ad));
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(bhi[nt]) : "v"(ad), "n"(4 * RBB));
            }
        };
        // wait until at most `younger` LDS reads are outstanding; returns the guarded fragments as MFMA operands
        auto landed = [&](auto younger_tag, s16x4_t (&alo)[MT], s16x4_t (&ahi)[MT], s16x4_t (&blo)[NT],
                          s16x4_t (&bhi)[NT], bf16x8_t (&fa)[MT], bf16x8_t (&fb)[NT]) {
            constexpr int YOUNGER = decltype(younger_tag)::value;
            if constexpr (NT == 2)
                asm volatile("s_waitcnt lgkmcnt(%12)"
                             : "+v"(alo[0]), "+v"(ahi[0]), "+v"(alo[1]), "+v"(ahi[1]), "+v"(alo[2]), "+v"(ahi[2]),
                               "+v"(alo[3]), "+v"(ahi[3]), "+v"(blo[0]), "+v"(bhi[0]), "+v"(blo[NT - 1]), "+v"(bhi[NT - 1])
                             : "n"(YOUNGER));
            else
                asm volatile("s_waitcnt lgkmcnt(%10)"
                             : "+v"(alo[0]), "+v"(ahi[0]), "+v"(alo[1]), "+v"(ahi[1]), "+v"(alo[2]), "+v"(ahi[2]),
                               "+v"(alo[3]), "+v"(ahi[3]), "+v"(blo[0]), "+v"(bhi[0])
                             : "n"(YOUNGER));
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                fa[mt] = bf16x8_t{alo[mt][0], alo[mt][1], alo[mt][2], alo[mt][3], ahi[mt][0], ahi[mt][1], ahi[mt][2], ahi[mt][3]};
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                fb[nt] = bf16x8_t{blo[nt][0], blo[nt][1], blo[nt][2], blo[nt][3], bhi[nt][0], bhi[nt][1], bhi[nt][2], bhi[nt][3]};
        };
        auto mma_half = [&](const bf16x8_t (&fa)[MT], const bf16x8_t (&fb)[NT]) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[nt], fa[mt], acc[mt][nt], 0, 0, 0);
        };
        // software pipeline over the two 32-pixel halves (same structure as gather_gemm_k): the LDS serves one
        // half while the matrix pipe works on the other; the stage barrier sits between two MFMA groups.
        s16x4_t a0l[MT], a0h[MT], b0l[NT], b0h[NT], a1l[MT], a1h[MT], b1l[NT], b1h[NT];
        bf16x8_t fa[MT], fb[NT];
        using Younger = std::integral_constant<int, RPH>;
        using None = std::integral_constant<int, 0>;
        if (nsteps > 0) {
#pragma unroll
            for (int i = 0; i < WG_NS - 1; ++i)
                if (i < nsteps) stage_next();
            wait_stage(nsteps >= WG_NS - 1);
            if (WG_NS - 1 < nsteps) stage_next();
            int cbuf = 0;
            read_half(lds0, 0, a0l, a0h, b0l, b0h);
            for (int s = 0; s < nsteps; ++s) {
                const unsigned lcur = lds0 + cbuf * STAGE;
                cbuf = (cbuf + 1 == WG_NS) ? 0 : cbuf + 1;
                read_half(lcur, 1, a1l, a1h, b1l, b1h);
                landed(Younger{}, a0l, a0h, b0l, b0h, fa, fb);        // half 0 landed, half 1 in flight
                mma_half(fa, fb);
                // half 1 landed (its reads were issued a whole MFMA group ago).  One wait site per register
                // set keeps the compiler from merging two tied asm statements through register copies that
                // would read a fragment before its wait.
                landed(None{}, a1l, a1h, b1l, b1h, fa, fb);
                if (s + 1 < nsteps) {
                    wait_stage(nsteps - s - 2 >= WG_NS - 2);
                    if (s + WG_NS < nsteps) stage_next();
                    read_half(lds0 + cbuf * STAGE, 0, a0l, a0h, b0l, b0h);
                }
                mma_half(fa, fb);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // nothing is in flight here; keeps the ISA check linear
        }
    } else {
        auto multiply = [&](const unsigned char* lb) {
#pragma unroll
            for (int sub = 0; sub < WG_BK / 4; ++sub) {
                float fa[MT], fb[NT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) fa[mt] = *(const float*)(lb + offA[mt] + sub * 4 * RBA);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) fb[nt] = *(const float*)(lb + offB[nt] + sub * 4 * RBB);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[nt], fa[mt], acc[mt][nt], 0, 0, 0);
            }
        };
#pragma unroll
        for (int i = 0; i < WG_NS - 1; ++i)
            if (i < nsteps) stage_next();
        int cbuf = 0;
        for (int s = 0; s < nsteps; ++s) {
            wait_stage(nsteps - s - 1 >= WG_NS - 2);
            if (s + WG_NS - 1 < nsteps) stage_next();
            multiply(smem + cbuf * STAGE);
            cbuf = (cbuf + 1 == WG_NS) ? 0 : cbuf + 1;
        }
    }

    WG_STAMP(3);
    // D[row = ci 4g+r][col = co i]: lane owns 4 consecutive ci of one co -> one 16-B store
    float* slab = p.dW + (size_t)ks * p.Co * p.taps * p.Ci;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int co = co0 + (wr * MT + mt) * 16 + fi;
        if (co >= p.Co) continue;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int ci = ci0 + (wc * NT + nt) * 16 + 4 * fg;
            if (ci >= p.Ci) continue;
            *(f32x4_t*)(slab + ((size_t)co * p.taps + tap) * p.Ci + ci) = acc[mt][nt];
        }
    }
#if WG_STAMPS
    WG_STAMP(4);
    if (p.stamps) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); WG_STAMP(5); }
#endif
}

// idx[t][p] for a strided convolution: the In pixel row tap t reads at output pixel p.
__global__ void conv_gather_index_k(int* __restrict__ idx, int Nimg, int IH, int IW, int OH, int OW, int KH,
                                    int KW, int stride, int pad) {
    const int P = Nimg * OH * OW;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P * KH * KW) return;
    const int t = i / P, pp = i - t * P;
    const int kh = t / KW, kw = t - kh * KW;
    const int n = pp / (OH * OW), rem = pp - n * (OH * OW);
    const int oh = rem / OW, ow = rem - oh * OW;
    const int ih = oh * stride - pad + kh, iw = ow * stride - pad + kw;
    idx[i] = (ih >= 0 && ih < IH && iw >= 0 && iw < IW) ? (n * IH + ih) * IW + iw : -1;
}

template <typename T, int NT, int NS, int BM = WG_BM>
static int launch_wg_ns(const WgArgs& a, hipStream_t st) {
    constexpr int ES = sizeof(T);
    constexpr int BK = (ES == 2) ? 64 : 32;
    constexpr size_t ring = (size_t)NS * BK * (BM + 64 * NT) * ES;
    const size_t lds = ring + (size_t)a.Pper * sizeof(int);            // + this launch's index table
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)wgrad_gemm_k<T, NT, NS, BM>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)(ring + WG_MAXP * sizeof(int)));
        attr_set = true;
    }
    long blocks = (long)cdiv(a.Co, BM) * cdiv(a.Ci, 64 * NT) * a.taps * a.ksplit;
    if (a.xcd_order) blocks = 8 * ((blocks + 7) / 8);
    hipLaunchKernelGGL((wgrad_gemm_k<T, NT, NS, BM>), dim3((unsigned)blocks), dim3(512), lds, st, a);
    RBVAE_CHECK_LAUNCH("wgrad_gemm");
    return RBVAE_OK;
}

template <typename T, int NT, int BM = WG_BM>
static int launch_wg(const WgArgs& a, hipStream_t st) {
    constexpr int ES = sizeof(T);
    constexpr int BK = (ES == 2) ? 64 : 32;
    // more workgroups than CUs: two per CU (double buffer) when two rings + index tables fit the 160 KB LDS
    const long blocks = (long)cdiv(a.Co, BM) * cdiv(a.Ci, 64 * NT) * a.taps * a.ksplit;
    const size_t lds2 = (size_t)2 * BK * (BM + 64 * NT) * ES + (size_t)a.Pper * sizeof(int);
    constexpr int force = 0;
    const bool two = force ? force == 2 : (blocks > 256 && 2 * lds2 <= 160 * 1024);
    if constexpr (ES == 2) {
        if (force == 4 && a.Pper <= 2560) return launch_wg_ns<T, NT, 4>(a, st);
    }
    return two ? launch_wg_ns<T, NT, 2, BM>(a, st) : launch_wg_ns<T, NT, 3, BM>(a, st);
}

}  // namespace rbvae

using namespace rbvae;

static unsigned long long* g_wg_stamps = nullptr;

extern "C" {

#if WG_STAMPS
/* stamped builds only (include/rbvae_dbg.h): later rbvae_wgrad_gemm launches write phase stamps into buf */
int rbvae_dbg_wg_stamps(unsigned long long* buf, void* stream) {
    (void)stream;
    g_wg_stamps = buf;
    return RBVAE_OK;
}
#endif

int rbvae_conv_gather_index(int* idx, int Nimg, int IH, int IW, int OH, int OW, int KH, int KW, int stride,
                            int pad, void* stream) {
    RBVAE_CHECK_ARG(idx && Nimg > 0 && IH > 0 && IW > 0 && OH > 0 && OW > 0 && KH > 0 && KW > 0 && stride > 0,
                    "conv_gather_index: bad arguments");
    const long tot = (long)Nimg * OH * OW * KH * KW;
    RBVAE_CHECK_ARG(tot < (1l << 31) && (long)Nimg * IH * IW < (1l << 31), "conv_gather_index: too many rows");
    hipLaunchKernelGGL(conv_gather_index_k, dim3(cdiv(tot, 256)), dim3(256), 0, (hipStream_t)stream, idx, Nimg, IH,
                       IW, OH, OW, KH, KW, stride, pad);
    RBVAE_CHECK_LAUNCH("conv_gather_index");
    return RBVAE_OK;
}

int rbvae_wgrad_gemm(int dtype, const void* Dy, const void* In, float* dW_slabs, const int* idx,
                     const void* zero_page, int P, int in_rows, int Co, int Ci, int ldy, int ldi, int taps, int ksplit,
                     void* stream) {
    RBVAE_CHECK_ARG(Dy && In && dW_slabs && zero_page, "wgrad_gemm: null pointer");
    RBVAE_CHECK_ARG(in_rows > 0 && (idx || in_rows >= P), "wgrad_gemm: in_rows=%d (P=%d)", in_rows, P);
    RBVAE_CHECK_ARG(dtype == RBVAE_F32 || dtype == RBVAE_BF16, "wgrad_gemm: dtype %d", dtype);
    const int ES = dtype == RBVAE_F32 ? 4 : 2;
    RBVAE_CHECK_ARG(P > 0 && Co > 0 && Ci > 0 && taps > 0 && ksplit > 0, "wgrad_gemm: bad sizes");
    RBVAE_CHECK_ARG(Co % 8 == 0 && Ci % 8 == 0, "wgrad_gemm: Co=%d Ci=%d must be multiples of 8", Co, Ci);
    RBVAE_CHECK_ARG((ldy * ES) % 16 == 0 && (ldi * ES) % 16 == 0 && ldy >= Co && ldi >= Ci,
                    "wgrad_gemm: leading dimensions ldy=%d ldi=%d", ldy, ldi);
    RBVAE_CHECK_ARG(((uintptr_t)Dy | (uintptr_t)In | (uintptr_t)dW_slabs | (uintptr_t)zero_page) % 16 == 0,
                    "wgrad_gemm: pointers must be 16-byte aligned");
    WgArgs a;
    a.Dy = (const unsigned char*)Dy; a.In = (const unsigned char*)In; a.dW = dW_slabs; a.idx = idx; a.in_rows = in_rows;
    a.zero = (const unsigned char*)zero_page;
    a.P = P; a.Co = Co; a.Ci = Ci; a.ldy = ldy; a.ldi = ldi; a.taps = taps; a.ksplit = ksplit;
    a.Pper = ((cdiv(P, ksplit) + 63) / 64) * 64;
    a.stamps = g_wg_stamps;
    // XCD-ordered workgroups: the same step time (0.4454 on vs 0.4456 ms off, same GPU, 4 runs each -- the K step is bound
    // by the CU's intake, not by where the rows come from) but a third less traffic behind the L2s: 66.9 vs 103.1 MB
    // fetched by the 252-workgroup launch, 68.4 vs 86.3 (216), 21.1 vs 25.9 (108) (FETCH_SIZE, tools/pmc_traffic_env.sh)
    constexpr int xcd = 1;
    a.xcd_order = xcd && ksplit > 1;
    RBVAE_CHECK_ARG(a.Pper <= WG_MAXP, "wgrad_gemm: %d pixels per K-slice exceed %d: raise ksplit (>= %d)", a.Pper,
                    WG_MAXP, cdiv(P, WG_MAXP));
    hipStream_t st = (hipStream_t)stream;
    const bool wide = Ci > 64;
    if (dtype == RBVAE_F32 && Co <= 64) return wide ? launch_wg<float, 2, 64>(a, st) : launch_wg<float, 1, 64>(a, st);
    if (dtype == RBVAE_F32) return wide ? launch_wg<float, 2>(a, st) : launch_wg<float, 1>(a, st);
    // (a 128 x 256 tile -- 48 KB of operands per 4.2 MFLOP K step instead of 32 KB per 2.1 -- ran its K step 21 % faster per
    // FLOP but needs twice the K-slices for the same number of workgroups: 0.469-0.483 vs 0.461 ms per step, round 2; the wide
    // 3x3 layers now go to rbvae_wgrad3x3s2_row, whose tile shares operands across taps instead)
    return wide ? launch_wg<bf16_t, 2>(a, st) : launch_wg<bf16_t, 1>(a, st);
}

}  // extern "C"
