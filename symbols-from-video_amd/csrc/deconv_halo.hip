// Halo-resident transposed 3x3 convolution (stride 2, pad 1, output_padding 1) on the gfx950 matrix cores:
// ConvTranspose2d forward of the RBVAE decoder (models/percep_RBVAE/percep_RBVAE_model.py:76-81) and, with the
// data-gradient weight order, the input gradient of the encoder's Conv2d(3, s2, p1) (autograd of :54-57).
//
//   Out[n][2a+ch][2b+cw][co] = epi( sum over the taps (kh, kw) of parity class (ch, cw), ci of
//                                   A[n][a+dh][b+dw][ci] * W[co][kh*3+kw][ci] ),   dh = (ch+1-kh)/2, dw = (cw+1-kw)/2
//
// The four output-parity classes take 4 + 2 + 2 + 1 = 9 taps on the SAME (rows+1) x (cols+1) patch of input pixels.
// rbvae_gather_gemm runs them as four sets of workgroups that each re-gather one 128-row tile per tap and K slice (and
// whose K loops are only 4..16 steps long); here ONE workgroup owns 256 input-grid positions x 64 output channels for
// ALL FOUR classes: per 128-byte channel slice the patch is staged in LDS once (chunk-major image, as conv_halo.hip) and
// the nine taps read it with tap-constant shifts, accumulating into four accumulator sets (4 x 32 registers).
//
// Tile geometry for any image size: a tile is TR = 256/SC consecutive rows of the "strip-linear" row order -- strips
// of SC (16 / 8 / 4) columns, all rows of strip 0 of image 0, then strip 1, ... -- so small images (8 x 8 and 4 x 4 grids
// of the 256-frame step) pack several strips and images into one tile with no padding rows; every strip segment inside a
// tile gets one extra patch row (the a+1 halo), which keeps a tap's shift of the patch a constant: slot =
// (tile row + strip ordinal) * (SC+1) + column.
//
// Steps per channel slice: {2 taps of class 11}, {2 of class 11}, {2 of class 10}, {2 of class 01}, {1 of class 00}: each
// step stages a 16 KB weight stage (2 taps x 64 channels x 128 B, XOR-swizzled rows, LDS-DMA ring) and runs 32 MFMAs per
// wave behind one barrier.  8 waves as 4 (pixel rows) x 2 (channel halves).  Epilogue class by class through one LDS tile:
// bias, ReLU, scale, keyed / explicit dropout, ReLU gate of the layer below, per-tile column sums (bias gradients) --
// element for element the arithmetic of rbvae_gather_gemm's epilogue (same dropout element indices).
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace rbvae {

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

struct DhArgs {
    const unsigned char* A;        // [Nimg*TH*TW][lda] T
    const unsigned char* W;        // [Nout][9][Kc] T
    unsigned char* Out;            // [Nimg*OH*OW][ldo] T
    const float* bias;             // [Nout] or null
    const unsigned char* gate;     // [Nimg*OH*OW][ldo] T or null: zero the output where gate <= 0
    const unsigned char* mask;     // [Nimg*OH*OW][Nout] u8 keep-mask or null
    const unsigned char* zero;     // >= 128 zero bytes
    float* colsum_ws;              // null or [mtiles * 4][Nout]
    const unsigned long long* seed_dev;
    unsigned long long seed;
    int Nimg, TH, TW, OH, OW, Kc, Nout, lda, ldo;
    int relu, drop_mode;
    float scale;
    unsigned drop_thresh;
    int strips_per_img, grows, mtiles, ntn, total;
};

// the nine taps in step order: class (ch*2+cw), weight tap index kh*3+kw, input shift (dh, dw)
constexpr int DH_CLS[9] = {3, 3, 3, 3, 2, 2, 1, 1, 0};
constexpr int DH_WIDX[9] = {0, 2, 6, 8, 1, 7, 3, 5, 4};
constexpr int DH_DH[9] = {1, 1, 0, 0, 1, 0, 0, 0, 0};
constexpr int DH_DW[9] = {1, 0, 1, 0, 0, 0, 1, 0, 0};
constexpr int DH_STEP_OF[9] = {0, 0, 1, 1, 2, 2, 3, 3, 4};
constexpr int DH_IN_STEP[9] = {0, 1, 0, 1, 0, 1, 0, 1, 0};

template <int SC, int BM> struct DhGeom {
    static constexpr int TR = BM / SC;
    static constexpr int PW = SC + 1;
    // patch slots (multiple of 16).  256-row tiles: one 8-wave workgroup per CU; 128-row tiles: TWO 4-wave workgroups per
    // CU (80 KB of LDS each), so that one's epilogue -- 1024 / 512 output pixels per tile, HBM-bound -- runs under the
    // other's MFMA loop (one workgroup per CU left the epilogue exposed: 75 of 217 us at 128 x 22 x 40 x 256)
    static constexpr int NSLOT_PAD = BM == 128 ? 176 : SC == 16 ? 320 : SC == 8 ? 352 : 416;
    static constexpr int MAXROWS = NSLOT_PAD / PW;                                // patch rows that fit
    static constexpr int PLANE = NSLOT_PAD * 16;
    static constexpr int KKOFF = 4 * PLANE + 64;
    static constexpr int ABUF = 8 * PLANE + 128;
    static constexpr int NA = (NSLOT_PAD * 8 + 2 * BM - 1) / (2 * BM);          // 16-byte pieces per thread and slice
};

constexpr int DH_BN = 64;
constexpr int DH_TAPB = DH_BN * 128;          // one tap's weight tile
constexpr int DH_STAGE = 2 * DH_TAPB;         // a step's stage: two taps

__device__ __forceinline__ void dh_glds16(const void* g, void* lds) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}
__device__ __forceinline__ unsigned dh_plane_off(int chunk, int plane) { return (unsigned)(chunk * plane + (chunk >> 1) * 32); }

template <typename T> struct DhMma;
template <> struct DhMma<bf16_t> {
    static __device__ __forceinline__ void run(f32x4_t& acc, const u32x4_t& rowop, const u32x4_t& colop) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8_t*)&rowop, *(const bf16x8_t*)&colop, acc, 0, 0, 0);
    }
};
template <> struct DhMma<float> {
    static __device__ __forceinline__ void run(f32x4_t& acc, const u32x4_t& rowop, const u32x4_t& colop) {
        const f32x4_t r = *(const f32x4_t*)&rowop, c = *(const f32x4_t*)&colop;
#pragma unroll
        for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(r[q], c[q], acc, 0, 0, 0);
    }
};

template <typename T> __device__ __forceinline__ bool dh_pos(const unsigned char* p, int e);
template <> __device__ __forceinline__ bool dh_pos<float>(const unsigned char* p, int e) { return ((const float*)p)[e] > 0.f; }
template <> __device__ __forceinline__ bool dh_pos<bf16_t>(const unsigned char* p, int e) {
    const bf16_t v = ((const bf16_t*)p)[e];
    return (v & 0x8000u) == 0 && (v & 0x7fffu) != 0 && (v & 0x7fffu) <= 0x7f80u;
}

template <int I, int N, typename F> __device__ __forceinline__ void dh_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        dh_static_for<I + 1, N>(f);
    }
}
// workgroup barrier that orders LDS traffic only.  __syncthreads() also waits vmcnt(0): behind the epilogue's global stores
// every barrier then costs a full store round trip (the epilogue ran at half the HBM write rate because of it).
__device__ __forceinline__ void dh_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int N> __device__ __forceinline__ void dh_wait_barrier() {
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}

template <typename T, int SC, int WAVES> constexpr int dh_lds_main() {
    constexpr int ES = sizeof(T), BM = 32 * WAVES;
    constexpr int ring = 2 * DhGeom<SC, BM>::ABUF + (WAVES == 8 ? 3 : 2) * DH_STAGE;
    constexpr int epi = BM * (DH_BN * ES + 16) + (64 * WAVES / (DH_BN / (16 / ES))) * DH_BN * 4;
    return ring > epi ? ring : epi;
}

template <typename T, int SC, int WAVES>
__global__ __launch_bounds__(64 * WAVES, WAVES == 4 ? 2 : 1) void deconv_halo_k(const DhArgs p) {
    constexpr int THREADS = 64 * WAVES, DH_BM = 32 * WAVES;
    using G = DhGeom<SC, DH_BM>;
    constexpr int RING = WAVES == 8 ? 3 : 2;          // weight stages: two in flight ahead (one workgroup per CU) / one
    constexpr int AHEAD = RING - 1;
    constexpr int ES = sizeof(T);
    constexpr int KE = 128 / ES, EC = 16 / ES;
    constexpr int MT = 4, NTW = 2;
    constexpr int LOADS = 16 / WAVES;                 // weight LDS-DMA instructions per wave and stage
    constexpr int NA = G::NA;
    constexpr int SPP = THREADS / 8;                  // slots per staging pass
    constexpr int A_BYTES = 2 * G::ABUF;
    constexpr int MAIN = dh_lds_main<T, SC, WAVES>();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int* s_obase = (int*)(smem + MAIN);               // [256] output row of class (0,0) per tile pixel, -1 = no pixel
    int* s_pix = s_obase + DH_BM;                     // [NSLOT_PAD] input pixel row per patch slot, -1 = zero
    int* s_prow = s_pix + G::NSLOT_PAD;               // [MAXROWS + 1] (strip << 12 | a) per patch row, -1 = unused

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifndef DH_STAGGER
#define DH_STAGGER 0
#endif
#if DH_STAGGER
    // Two workgroups share a CU and every workgroup does the same work, so left alone they run in phase: both in the MFMA
    // loop, then both in the (HBM-bound) epilogue.  The second workgroup of each CU in the launch's first round starts
    // half a tile late; the phases then stay apart for the rest of the launch.  Speed only: every tile is computed whatever
    // the placement.
    if (WAVES == 4 && p.total >= 1024 && blockIdx.x >= 256 && blockIdx.x < 512) {
        for (int i = 0; i < DH_STAGGER; ++i) __builtin_amdgcn_s_sleep(127);
    }
#endif
    int item;
    {
        const int lin = blockIdx.x, xcd = lin & 7, q = p.total >> 3, r = p.total & 7;
        item = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (lin >> 3);
    }
    const int mtile = item / p.ntn, ntile = item - mtile * p.ntn;
    const int n0 = ntile * DH_BN;
    const int g0 = mtile * G::TR;
    const int s0 = g0 / p.TH;

    // ---- tables: patch rows, patch slots, output rows
    for (int i = tid; i <= G::MAXROWS; i += THREADS) s_prow[i] = -1;
    __syncthreads();
    if (tid < G::TR) {
        const int g = g0 + tid;
        if (g < p.grows) {
            const int s = g / p.TH, a = g - s * p.TH, pr = tid + (s - s0);
            s_prow[pr] = (s << 12) | a;
            if (a == p.TH - 1 || tid == G::TR - 1 || g == p.grows - 1) s_prow[pr + 1] = (s << 12) | (a + 1);
        }
    }
    __syncthreads();
    for (int sl = tid; sl < G::NSLOT_PAD; sl += THREADS) {
        int v = -1;
        const int pr = sl / G::PW, pc = sl - pr * G::PW;
        if (pr <= G::MAXROWS) {
            const int e = s_prow[pr];
            if (e >= 0) {
                const int s = e >> 12, a = e & 4095;
                const int img = s / p.strips_per_img, bb = s - img * p.strips_per_img;
                const int iw = bb * SC + pc;
                if (a < p.TH && iw < p.TW) v = (img * p.TH + a) * p.TW + iw;
            }
        }
        s_pix[sl] = v;
    }
    if (tid < DH_BM) {
        const int i = tid / SC, c = tid - i * SC;
        const int g = g0 + i;
        int o = -1;
        if (g < p.grows) {
            const int s = g / p.TH, a = g - s * p.TH;
            const int img = s / p.strips_per_img, bb = s - img * p.strips_per_img;
            o = (img * p.OH + 2 * a) * p.OW + 2 * (bb * SC + c);
        }
        s_obase[tid] = o;
    }
    __syncthreads();

    // ---- patch staging roles (see conv_halo.hip): piece i = (slot (tid>>3) + 64 i, chunk tid&7)
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const int chunk = tid & 7;
    const unsigned char* zsrc = p.zero + chunk * 16;
    // the NA pieces of a slice go through registers in two batches (NB0 first, the rest one step later): all of them
    // at once cost 28 registers beside 128 of accumulators and 48 of fragments, and the kernel spilled
    constexpr int NB0 = (NA + 1) / 2, NB1 = NA - NB0;
    u32x4_t areg[NB0];
    auto piece_pix = [&](int i) {                      // -1: zero (padding), -2: no such slot (LDS read: only behind a barrier)
        const int slot = (tid >> 3) + SPP * i;
        return slot < G::NSLOT_PAD ? s_pix[slot] : -2;
    };
    auto a_load = [&](int kc, auto b_tag, u32x4_t (&ar)[NB0]) {
        constexpr int b = decltype(b_tag)::value, i0 = b ? NB0 : 0, n = b ? NB1 : NB0;
#pragma unroll
        for (int i = 0; i < n; ++i) {
            const int pv = piece_pix(i0 + i);
            const unsigned char* src = pv >= 0 ? p.A + ((size_t)pv * p.lda) * ES + (size_t)kc * 128 + chunk * 16 : zsrc;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ar[i]) : "v"(src) : "memory");
        }
    };
    auto a_landed = [](auto younger_tag, u32x4_t (&ar)[NB0]) {
        constexpr int YOUNGER = decltype(younger_tag)::value;
        static_assert(NB0 == 3 || NB0 == 4, "operand list below");
        if constexpr (NB0 == 3)
            asm volatile("s_waitcnt vmcnt(%3)" : "+v"(ar[0]), "+v"(ar[1]), "+v"(ar[2]) : "n"(YOUNGER));
        else
            asm volatile("s_waitcnt vmcnt(%4)" : "+v"(ar[0]), "+v"(ar[1]), "+v"(ar[2]), "+v"(ar[3 % NB0]) : "n"(YOUNGER));
    };
    auto a_write = [&](int buf, auto b_tag, u32x4_t (&ar)[NB0]) {
        constexpr int b = decltype(b_tag)::value, i0 = b ? NB0 : 0, n = b ? NB1 : NB0;
        const unsigned base = lds0 + (unsigned)buf * G::ABUF + dh_plane_off(chunk, G::PLANE) + (unsigned)(tid >> 3) * 16;
#pragma unroll
        for (int i = 0; i < n; ++i)
            if ((tid >> 3) + SPP * (i0 + i) < G::NSLOT_PAD)
                asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(base), "v"(ar[i]), "n"((i0 + i) * SPP * 16) : "memory");
    };
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;

    // ---- weight staging: a stage = 16 LDS-DMA instructions (2 taps x 8 blocks of 8 rows); instruction q = w * LOADS + i
    // moves rows (q & 7) * 8 .. +7 of the step's tap q >> 3 (uniform base + 32-bit lane offset)
    const int srow = lane >> 3, schunk = lane & 7;
    unsigned blane[LOADS];
#pragma unroll
    for (int i = 0; i < LOADS; ++i) {
        const int brow = ((w * LOADS + i) & 7) * 8 + srow;
        blane[i] = (unsigned)(brow * 9 * p.Kc) * ES + (unsigned)((schunk ^ ((brow >> 1) & 7)) * 16);
    }
    const unsigned char* wtile = p.W + ((size_t)n0 * 9 * p.Kc) * ES;
    const int nkc = p.Kc / KE;
    int pslot = 0;
    auto b_issue = [&](int kc, auto s_tag) {
        constexpr int s = decltype(s_tag)::value;
        unsigned char* lb = smem + A_BYTES + pslot * DH_STAGE + (w * LOADS) * 1024;
        constexpr int t0 = 2 * s, t1 = s < 4 ? 2 * s + 1 : 8;         // step 4 has one tap: its second half repeats it
        const unsigned char* src0 = wtile + ((size_t)DH_WIDX[t0] * p.Kc) * ES + (size_t)kc * 128;
        const unsigned char* src1 = wtile + ((size_t)DH_WIDX[t1] * p.Kc) * ES + (size_t)kc * 128;
#pragma unroll
        for (int i = 0; i < LOADS; ++i) dh_glds16((((w * LOADS + i) >> 3) ? src1 : src0) + blane[i], lb + i * 1024);
        pslot = (pslot + 1 == RING) ? 0 : pslot + 1;
    };

    // ---- fragment addresses
    const int fi = lane & 15, fg = lane >> 4;
    const int wr = w >> 1, wc = w & 1;
    unsigned ta[MT];                     // patch read addresses of the slice being read (buffer 0 first)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int t = (wr * MT + mt) * 16 + fi;
        const int i = t / SC, c = t - i * SC;
        const int gg = min(g0 + i, p.grows - 1);
        const int pr = i + (gg / p.TH - s0);
        ta[mt] = lds0 + (unsigned)fg * G::PLANE + (unsigned)(fg >> 1) * 32 + (unsigned)(pr * G::PW + c) * 16;
    }
    const int fsw = (fi >> 1) & 7;
    const unsigned offB0 = lds0 + A_BYTES + (unsigned)(wc * NTW * 16 + fi) * 128 + (unsigned)(((0 + fg) ^ fsw) * 16);
    const unsigned offB1 = lds0 + A_BYTES + (unsigned)(wc * NTW * 16 + fi) * 128 + (unsigned)(((4 + fg) ^ fsw) * 16);

    f32x4_t acc[4][MT][NTW];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) acc[c][mt][nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    unsigned tb0 = 0, tb1 = 0;           // weight read addresses of the step being read (ring slot folded in)
    int cslot = 0;
    auto set_slice = [&](int kc) {           // called once per slice, in order: the read addresses flip between the buffers
        if (kc == 0) return;
        const unsigned d = (kc & 1) ? (unsigned)G::ABUF : (unsigned)-G::ABUF;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) ta[mt] += d;
    };
    auto set_step = [&]() {
        tb0 = offB0 + (unsigned)cslot * DH_STAGE;
        tb1 = offB1 + (unsigned)cslot * DH_STAGE;
    };
    // half-unit h = 2 * tap + kk
    auto read_half = [&](auto h_tag, u32x4_t (&fa)[MT], u32x4_t (&fb)[NTW]) {
        constexpr int h = decltype(h_tag)::value, t = h >> 1, kk = h & 1;
        constexpr int aoff = kk * G::KKOFF + (DH_DH[t] * G::PW + DH_DW[t]) * 16;
        constexpr int boff = DH_IN_STEP[t] * DH_TAPB;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[mt]) : "v"(ta[mt]), "n"(aoff));
        const unsigned ab_ = kk == 0 ? tb0 : tb1;
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[nt]) : "v"(ab_), "n"(boff + nt * 2048));
    };
    auto landed = [&](auto younger_tag, u32x4_t (&fa)[MT], u32x4_t (&fb)[NTW]) {
        constexpr int YOUNGER = decltype(younger_tag)::value;
        asm volatile("s_waitcnt lgkmcnt(%6)"
                     : "+v"(fa[0]), "+v"(fa[1]), "+v"(fa[2]), "+v"(fa[3]), "+v"(fb[0]), "+v"(fb[1])
                     : "n"(YOUNGER));
    };
    auto mma_half = [&](auto h_tag, const u32x4_t (&fa)[MT], const u32x4_t (&fb)[NTW]) {
        constexpr int cls = DH_CLS[decltype(h_tag)::value >> 1];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) DhMma<T>::run(acc[cls][mt][nt], fb[nt], fa[mt]);
    };
    using Younger = std::integral_constant<int, MT + NTW>;
    using None = std::integral_constant<int, 0>;

    // ---- prologue: patch of slice 0, the first AHEAD weight stages
    b_issue(0, std::integral_constant<int, 0>{});
    if constexpr (AHEAD >= 2) b_issue(0, std::integral_constant<int, 1>{});
    a_load(0, B0{}, areg);
    a_landed(std::integral_constant<int, 0>{}, areg);
    a_write(0, B0{}, areg);
    a_load(0, B1{}, areg);
    a_landed(std::integral_constant<int, 0>{}, areg);
    a_write(0, B1{}, areg);
    dh_wait_barrier<0>();
    b_issue(0, std::integral_constant<int, AHEAD>{});
    u32x4_t fa0[MT], fb0[NTW], fa1[MT], fb1[NTW];
    set_slice(0);
    set_step();
    read_half(std::integral_constant<int, 0>{}, fa0, fb0);

    // ---- main loop: 18 half-units per slice (even ones in fragment set 0, odd ones in set 1); the barrier that opens a
    // step sits between the last two MFMA groups of the step before it.  Behind it: the weight stage two steps ahead; when
    // the slice's step 1 opens the NEXT slice's patch goes to registers, when step 2 opens into the other patch buffer
    // (loads, wait and stores inside one unrolled slice body: conv_halo.hip).
    auto slice = [&](int kc, auto more_tag) {
        constexpr bool more = decltype(more_tag)::value;
        dh_static_for<0, 18>([&](auto h_tag) {
            constexpr int h = decltype(h_tag)::value, t = h >> 1;
            constexpr bool cur0 = (h & 1) == 0;
            constexpr bool last_of_slice = h == 17;
            constexpr bool boundary = (h & 1) == 1 && (last_of_slice || DH_STEP_OF[t + (t < 8 ? 1 : 0)] != DH_STEP_OF[t]);
            using HN = std::integral_constant<int, (h + 1) % 18>;
            auto& fa_c = cur0 ? fa0 : fa1;
            auto& fb_c = cur0 ? fb0 : fb1;
            auto& fa_n = cur0 ? fa1 : fa0;
            auto& fb_n = cur0 ? fb1 : fb0;
            if constexpr (last_of_slice && !more) {
                landed(None{}, fa_c, fb_c);
                __builtin_amdgcn_sched_barrier(0);
                mma_half(h_tag, fa_c, fb_c);
            } else if constexpr (!boundary) {
                read_half(HN{}, fa_n, fb_n);
                landed(Younger{}, fa_c, fb_c);
                __builtin_amdgcn_sched_barrier(0);
                mma_half(h_tag, fa_c, fb_c);
                __builtin_amdgcn_sched_barrier(0);
            } else {
                constexpr int ns = last_of_slice ? 0 : DH_STEP_OF[t] + 1;       // step being opened
                const int kn = last_of_slice ? kc + 1 : kc;
                const bool moren = last_of_slice ? kn + 1 < nkc : more;          // a slice follows the opened step's slice
                landed(None{}, fa_c, fb_c);
                // the opened step's stage has landed; younger: the stage behind it (none behind the very last step), and
                // when steps 2 / 3 open the patch batch issued behind it
                constexpr int young = (more || last_of_slice) ? AHEAD - 1 : (AHEAD - 1 < 4 - ns ? AHEAD - 1 : 4 - ns);
                if (last_of_slice && !moren && AHEAD - 1 > 4) dh_wait_barrier<0>();      // (never: AHEAD <= 2)
                else if constexpr (ns == 2 && more) dh_wait_barrier<young * LOADS + NB0>();
                else if constexpr (ns == 3 && more) dh_wait_barrier<young * LOADS + NB1>();
                else dh_wait_barrier<young * LOADS>();
                cslot = (cslot + 1 == RING) ? 0 : cslot + 1;
                if constexpr (ns + AHEAD < 5) b_issue(kn, std::integral_constant<int, (ns + AHEAD) % 5>{});
                else if (moren) b_issue(kn + 1, std::integral_constant<int, (ns + AHEAD) % 5>{});
                if constexpr (ns == 1 && more) a_load(kn + 1, B0{}, areg);
                if constexpr (ns == 2 && more) {
                    a_landed(std::integral_constant<int, LOADS>{}, areg);      // behind them: this step's stage issue
                    a_write((kn + 1) & 1, B0{}, areg);
                    a_load(kn + 1, B1{}, areg);
                }
                if constexpr (ns == 3 && more) {
                    a_landed(std::integral_constant<int, LOADS>{}, areg);
                    a_write((kn + 1) & 1, B1{}, areg);
                }
                if constexpr (ns == 0) set_slice(kn);
                set_step();
                read_half(HN{}, fa_n, fb_n);
                __builtin_amdgcn_sched_barrier(0);
                mma_half(h_tag, fa_c, fb_c);
                __builtin_amdgcn_sched_barrier(0);
            }
        });
    };
#ifndef DH_ABL
#define DH_ABL 0          // timing-only builds (tools/ab_variants.sh): 1 no epilogue, 2 no main loop, 4 no tables / prologue work
#endif
#if DH_ABL && !defined(RBVAE_ABLATION)
#error "DH_ABL builds give wrong results: define RBVAE_ABLATION to confirm"
#endif
#if !(DH_ABL & 2)
    for (int kcl = 0; kcl + 1 < nkc; ++kcl) {
        int kc = kcl;
        asm volatile("" : "+s"(kc));
        slice(kc, std::true_type{});
    }
    slice(nkc - 1, std::false_type{});
#endif
    __syncthreads();
#if DH_ABL & 1
    if (p.Nimg >= 0) return;
#endif

    // ---- epilogue.  As many of the four class tiles as fit go to LDS at once (bf16: all four), then every store of the
    // round is issued back to back with the ReLU-gate chunks fetched ahead of them: with one class per barrier pair a CU
    // had 32 KB of stores in flight and the phase ran latency-bound at half the HBM write rate.
    constexpr int PITCH = DH_BN * ES + 16;
    constexpr int CPR = DH_BN / EC;              // 16-B chunks per tile row
    constexpr int RL = THREADS / CPR;            // row lanes of the store phase
    constexpr int ITERS = DH_BM / RL;
    constexpr int RED_BYTES = RL * DH_BN * 4;
    constexpr int FIT = (MAIN - RED_BYTES) / (DH_BM * PITCH);
    constexpr int CPRD = FIT >= 4 ? 4 : FIT >= 2 ? 2 : 1;       // classes per round
    static_assert(FIT >= 1, "one class tile + the column-sum scratch must fit");
    unsigned char* tile = smem;
    float* red = (float*)(smem + CPRD * DH_BM * PITCH);          // [RL][BN]
    const int sch = tid % CPR, rl = tid / CPR;
    const int scol = n0 + sch * EC;
    DropKey dkey{0u, 0u};
    if (p.drop_mode == 1) dkey = drop_key(p.seed + (p.seed_dev ? p.seed_dev[0] * 0x9E3779B97F4A7C15ull : 0ull));
    int obase[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) obase[it] = s_obase[it * RL + rl];
    dh_static_for<0, 4 / CPRD>([&](auto r_tag) {
        constexpr int rnd = decltype(r_tag)::value;
        if (rnd) dh_lds_barrier();                 // the previous round's readers are done with the tiles
        dh_static_for<0, CPRD>([&](auto c_tag) {
            constexpr int cls = rnd * CPRD + decltype(c_tag)::value;
            unsigned char* tl = tile + decltype(c_tag)::value * DH_BM * PITCH;
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                const int cb = (wc * NTW + nt) * 16 + 4 * fg;
                float bz[4] = {0.f, 0.f, 0.f, 0.f};
                if (p.bias) {
                    const float4 b4 = *(const float4*)(p.bias + n0 + cb);
                    bz[0] = b4.x; bz[1] = b4.y; bz[2] = b4.z; bz[3] = b4.w;
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int row = (wr * MT + mt) * 16 + fi;
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float x = acc[cls][mt][nt][r] + bz[r];
                        if (p.relu) x = fmaxf(x, 0.f);
                        v[r] = x * p.scale;
                    }
                    unsigned char* dst = tl + row * PITCH + cb * ES;
                    if constexpr (ES == 4) {
                        *(float4*)dst = make_float4(v[0], v[1], v[2], v[3]);
                    } else {
                        uint2 pk;
                        pk.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
                        pk.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
                        *(uint2*)dst = pk;
                    }
                }
            }
        });
        dh_lds_barrier();
        // the round's gate chunks, all in flight before the first is used
        u32x4_t gv[CPRD][ITERS];
        if (p.gate) {
#pragma unroll
            for (int c = 0; c < CPRD; ++c) {
                const int cls = rnd * CPRD + c;
#pragma unroll
                for (int it = 0; it < ITERS; ++it) {
                    const int orow = obase[it] < 0 ? 0 : obase[it] + (cls >> 1) * p.OW + (cls & 1);
                    gv[c][it] = *(const u32x4_t*)(p.gate + ((size_t)orow * p.ldo + scol) * ES);
                }
            }
        }
        float csum[CPRD][EC];
#pragma unroll
        for (int c = 0; c < CPRD; ++c) {
            const int cls = rnd * CPRD + c;
#pragma unroll
            for (int e = 0; e < EC; ++e) csum[c][e] = 0.f;
#pragma unroll
            for (int it = 0; it < ITERS; ++it) {
                const int row = it * RL + rl;
                if (obase[it] >= 0) {
                    const int orow = obase[it] + (cls >> 1) * p.OW + (cls & 1);
                    u32x4_t val = *(const u32x4_t*)(tile + c * DH_BM * PITCH + row * PITCH + sch * 16);
                    T* ev = (T*)&val;
                    if (p.drop_mode == 1) {
                        const unsigned run = drop_run(dkey, (unsigned long long)orow * p.Nout + scol);
                        if constexpr (sizeof(T) == 2) drop_chunk_zero_b16<EC>(run, p.drop_thresh >> 16, (unsigned*)&val);
                        else drop_chunk_zero_f32<EC>(run, p.drop_thresh >> 16, (float*)&val);
                    } else if (p.drop_mode == 2) {
                        const unsigned char* mk = p.mask + (size_t)orow * p.Nout + scol;
#pragma unroll
                        for (int e = 0; e < EC; ++e)
                            if (!mk[e]) ev[e] = 0;
                    }
                    if (p.gate) {
#pragma unroll
                        for (int e = 0; e < EC; ++e)
                            if (!dh_pos<T>((const unsigned char*)&gv[c][it], e)) ev[e] = 0;
                    }
                    *(u32x4_t*)(p.Out + ((size_t)orow * p.ldo + scol) * ES) = val;
                    if (p.colsum_ws) {
#pragma unroll
                        for (int e = 0; e < EC; ++e) csum[c][e] += Elem<T>::load(ev + e);
                    }
                }
            }
        }
        if (p.colsum_ws) {
#pragma unroll
            for (int c = 0; c < CPRD; ++c) {
                if (c) dh_lds_barrier();
#pragma unroll
                for (int e = 0; e < EC; ++e) red[rl * DH_BN + sch * EC + e] = csum[c][e];
                dh_lds_barrier();
                if (tid < DH_BN) {
                    float t = 0.f;
#pragma unroll 8
                    for (int k = 0; k < RL; ++k) t += red[k * DH_BN + tid];
                    p.colsum_ws[((size_t)mtile * 4 + rnd * CPRD + c) * p.Nout + n0 + tid] = t;
                }
            }
        }
    });
}

template <typename T, int SC, int WAVES>
static int launch_dh(const DhArgs& a, hipStream_t st) {
    using G = DhGeom<SC, 32 * WAVES>;
    const size_t lds = (size_t)dh_lds_main<T, SC, WAVES>() + 32 * WAVES * 4 + G::NSLOT_PAD * 4 + (G::MAXROWS + 2) * 4 + 16;
    if (lds > (WAVES == 4 ? 80 : 160) * 1024) return fail(RBVAE_E_UNSUPPORTED, "deconv3x3s2_halo: %zu bytes of LDS", lds);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)deconv_halo_k<T, SC, WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (WAVES == 4 ? 80 : 160) * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL((deconv_halo_k<T, SC, WAVES>), dim3(a.total), dim3(64 * WAVES), lds, st, a);
    RBVAE_CHECK_LAUNCH("deconv3x3s2_halo");
    return RBVAE_OK;
}

static int dh_strip_cols(int TW) { return TW % 16 == 0 ? 16 : TW % 8 == 0 ? 8 : TW % 4 == 0 ? 4 : 0; }

// patch rows a tile can need: its TR rows + one halo row per strip it touches
static int dh_max_patch_rows(int TH, int TR) { return TR + (TR % TH == 0 ? TR / TH : (TR - 1) / TH + 2); }

// tile rows (128: two 4-wave workgroups per CU; 256: one 8-wave workgroup) the shape can run with, 0 = not covered
static int dh_tile_rows(int dtype, int Nimg, int TH, int TW, int Kc, int Nout) {
    if (dtype != RBVAE_F32 && dtype != RBVAE_BF16) return 0;
    const int KE = dtype == RBVAE_F32 ? 32 : 64;
    if (Kc <= 0 || Kc % KE || Nout <= 0 || Nout % DH_BN || TH < 1 || TH > 4095 || Nimg < 1) return 0;
    const int sc = dh_strip_cols(TW);
    if (!sc) return 0;
    if ((long)Nimg * (TW / sc) >= (1l << 19)) return 0;          // strip index in 19 bits of the patch-row table
    constexpr int force = 0;
    if (force != 256 && dh_max_patch_rows(TH, 128 / sc) <= 176 / (sc + 1)) return 128;
    if (force != 128 && dh_max_patch_rows(TH, 256 / sc) <= (sc == 16 ? 320 : sc == 8 ? 352 : 416) / (sc + 1)) return 256;
    return 0;
}

}  // namespace rbvae

using namespace rbvae;

extern "C" int rbvae_deconv3x3s2_halo_ok(int dtype, int Nimg, int TH, int TW, int Kc, int Nout) {
    return dh_tile_rows(dtype, Nimg, TH, TW, Kc, Nout) != 0;
}

extern "C" int rbvae_deconv3x3s2_halo_tile_rows(int dtype, int Nimg, int TH, int TW, int Kc, int Nout) {
    return dh_tile_rows(dtype, Nimg, TH, TW, Kc, Nout);
}

extern "C" int rbvae_deconv3x3s2_halo_colsum_rows(int dtype, int Nimg, int TH, int TW, int Kc, int Nout) {
    const int bm = dh_tile_rows(dtype, Nimg, TH, TW, Kc, Nout);
    if (!bm) return 0;
    const int sc = dh_strip_cols(TW);
    return 4 * cdiv((long)Nimg * (TW / sc) * TH, bm / sc);
}

extern "C" int rbvae_deconv3x3s2_halo(int dtype, const void* A, const void* W, void* Out, const float* bias, const void* gate,
                                      const void* mask, const void* zero_page, int Nimg, int TH, int TW, int Kc, int Nout,
                                      int lda, int ldo, int relu, int drop_mode, float drop_p, float scale,
                                      unsigned long long seed, const unsigned long long* seed_dev, float* colsum_ws,
                                      void* stream) {
    RBVAE_CHECK_ARG(A && W && Out && zero_page, "deconv3x3s2_halo: null pointer");
    const int bm = dh_tile_rows(dtype, Nimg, TH, TW, Kc, Nout);
    RBVAE_CHECK_ARG(bm != 0, "deconv3x3s2_halo: shape not covered (dtype %d, %d x %dx%d, Kc %d, Nout %d)", dtype, Nimg, TH, TW,
                    Kc, Nout);
    const int ES = dtype == RBVAE_F32 ? 4 : 2;
    RBVAE_CHECK_ARG(lda >= Kc && (lda * ES) % 16 == 0 && ldo >= Nout && (ldo * ES) % 16 == 0,
                    "deconv3x3s2_halo: leading dimensions lda=%d ldo=%d", lda, ldo);
    RBVAE_CHECK_ARG((long)Nimg * TH * TW * 4 < (1l << 30), "deconv3x3s2_halo: more than 2^30 output pixel rows");
    RBVAE_CHECK_ARG(((uintptr_t)A | (uintptr_t)W | (uintptr_t)Out | (uintptr_t)zero_page | (uintptr_t)gate | (uintptr_t)bias) % 16 == 0,
                    "deconv3x3s2_halo: pointers must be 16-byte aligned");
    RBVAE_CHECK_ARG(drop_mode >= 0 && drop_mode <= 2 && (drop_mode != 2 || mask), "deconv3x3s2_halo: drop_mode/mask");
    DhArgs a;
    a.A = (const unsigned char*)A; a.W = (const unsigned char*)W; a.Out = (unsigned char*)Out; a.bias = bias;
    a.gate = (const unsigned char*)gate; a.mask = (const unsigned char*)mask; a.zero = (const unsigned char*)zero_page;
    a.colsum_ws = colsum_ws; a.seed_dev = seed_dev; a.seed = seed;
    a.Nimg = Nimg; a.TH = TH; a.TW = TW; a.OH = 2 * TH; a.OW = 2 * TW; a.Kc = Kc; a.Nout = Nout; a.lda = lda; a.ldo = ldo;
    a.relu = relu; a.drop_mode = drop_mode; a.scale = scale; a.drop_thresh = (unsigned)((double)drop_p * 4294967296.0);
    const int sc = dh_strip_cols(TW);
    a.strips_per_img = TW / sc;
    a.grows = Nimg * a.strips_per_img * TH;
    a.mtiles = cdiv(a.grows, bm / sc);
    a.ntn = Nout / DH_BN;
    a.total = a.mtiles * a.ntn;
    hipStream_t st = (hipStream_t)stream;
#define DH_DISPATCH(TT)                                                                                                   \
    if (bm == 128) return sc == 16 ? launch_dh<TT, 16, 4>(a, st) : sc == 8 ? launch_dh<TT, 8, 4>(a, st) : launch_dh<TT, 4, 4>(a, st); \
    return sc == 16 ? launch_dh<TT, 16, 8>(a, st) : sc == 8 ? launch_dh<TT, 8, 8>(a, st) : launch_dh<TT, 4, 8>(a, st);
    if (dtype == RBVAE_F32) { DH_DISPATCH(float) }
    DH_DISPATCH(bf16_t)
#undef DH_DISPATCH
}
