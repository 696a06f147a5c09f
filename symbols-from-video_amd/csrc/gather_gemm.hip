// Row-gather GEMM on the gfx950 matrix cores: the conv / conv-transpose / linear
// forward and input-gradient kernel of the RBVAE path.
//
//   Out[orow(m)][n] = epi( sum_{j < ntaps} sum_{k < Kc} A[arow(m, j)][k] * W[n][widx_j][k] )
//
// A is an NHWC activation (pixel rows of Kc channels), W is [Nout][taps][Kc] and
// every tap contributes one gathered pixel row (or a zero row outside the image).
// With the right tap table this one kernel is
//   * Conv2d(k, s2, p1) forward                (percep_RBVAE_model.py:51-57)
//   * ConvTranspose2d(k, s2, p1, op) forward = the conv's input gradient, split
//     into the 4 output-parity classes (blockIdx.z)   (percep_RBVAE_model.py:76-82)
//   * their backward-data counterparts, and the Linear layers as a 1-tap case.
//
// Tile: 128 output rows x (32*NT) output channels per 256-thread workgroup, K in
// 128-byte slices (64 bf16 / 32 f32).  Both operand tiles are staged by LDS-DMA
// (global_load_lds, 16 B per lane, double buffered) into 128-B-row images whose
// 16-B chunk index is XOR-swizzled with (row>>1)&7 -- applied on the per-lane
// SOURCE address and again on the ds_read_b128 fragment read -- so fragment
// reads are bank-conflict free.  bf16 uses v_mfma_f32_16x16x32_bf16, f32 uses
// v_mfma_f32_16x16x4_f32 (exact f32 fma chain) on the same LDS image.
// The accumulator is produced transposed (weights as the MFMA row operand) so a
// lane owns 4 consecutive output channels of one pixel; the tile goes through
// LDS once more and leaves as whole 16-B chunks of NHWC rows.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace rbvae {

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

struct TapClass {
    int ntaps;
    int oh0, ow0;              // output-grid offset of this parity class
    signed char widx[16];      // tap index into W's tap axis
    signed char dh[16], dw[16];
};

struct GgArgs {
    const unsigned char* A;    // [*, lda] T
    const unsigned char* W;    // [Nout][taps_total][Kc] T
    unsigned char* Out;        // [*, ldo] T
    const float* bias;         // [Nout] or null
    const unsigned char* gate; // [*, ldo] T or null: zero the output where gate <= 0
    const unsigned char* mask; // [*, Nout] u8 keep-mask or null
    const unsigned char* addend; // [*, ldo] T or null: residual added to the stored value (after bias/relu/scale)
    const unsigned char* zero; // >= 128 zero bytes
    int Nimg, IH, IW;          // pixel grid of A
    int TH, TW;                // per-class row grid: m -> (n, a, b)
    int sa;                    // A pixel = (a*sa + dh, b*sa + dw)
    int OH, OW, so;            // Out pixel = (a*so + oh0, b*so + ow0)
    int Kc, Nout, lda, ldo, taps_total;
    int relu, drop_mode;       // drop_mode: 0 none, 1 counter hash, 2 explicit mask
    float scale;               // applied after bias/relu (dropout 1/(1-p) or gate scale)
    unsigned drop_thresh;      // keep iff hash >= thresh
    unsigned long long seed;
    const unsigned long long* seed_dev;   // optional device step counter mixed into seed
    float* colsum_ws;          // optional [nclass * m-tiles][Nout]: per-tile column sums of the stored values
    int nclass;
    int dephase;               // 8-wave kernels: the two waves of a SIMD stage at different points of the step
    int xcd_order;             // 1: the N tiles and tap classes of one M tile run back to back on ONE XCD (see the kernel)
    int sh_thw, sh_tw;         // log2(TH*TW), log2(TW) when those are powers of two, else -1 (row index -> (n, a, b) by shifts)
    unsigned long long* stamps; // debug (rbvae_dbg_gg_stamps): [workgroup][8] phase time stamps (100 MHz), or null
    TapClass cls[4];
};

__device__ __forceinline__ void glds16(const void* g, void* lds) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
    // one 128-B LDS row slice = 64 k: two 32-k MFMAs, lane group g reads chunk 4*kk+g
    static __device__ __forceinline__ void run(f32x4_t& acc, const u32x4_t& rowop, const u32x4_t& colop) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8_t*)&rowop, *(const bf16x8_t*)&colop, acc,
                                                      0, 0, 0);
    }
};
template <> struct Mma<float> {
    // 32 k per row slice; lane group g holds k = 16*kk + 4*g + c for MFMA c (same on both operands)
    static __device__ __forceinline__ void run(f32x4_t& acc, const u32x4_t& rowop, const u32x4_t& colop) {
        const f32x4_t r = *(const f32x4_t*)&rowop, c = *(const f32x4_t*)&colop;
#pragma unroll
        for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(r[q], c[q], acc, 0, 0, 0);
    }
};

template <typename T> __device__ __forceinline__ bool elem_pos(const unsigned char* p, int e);
template <> __device__ __forceinline__ bool elem_pos<float>(const unsigned char* p, int e) {
    return ((const float*)p)[e] > 0.f;
}
template <> __device__ __forceinline__ bool elem_pos<bf16_t>(const unsigned char* p, int e) {
    const bf16_t v = ((const bf16_t*)p)[e];
    return (v & 0x8000u) == 0 && (v & 0x7fffu) != 0 && (v & 0x7fffu) <= 0x7f80u;   // > 0 (NaN excluded)
}

constexpr int GG_BM = 128;

// build-time experiment switches (tools/ab_variants.sh builds one library per setting for same-box A/B runs).
// Measured on the bench step, same GPU, 3 runs each: both cut the workgroup's setup / store phases by 0.3-0.9 us
// in the phase stamps yet left the step 0.5-1 % SLOWER (more live registers around the K loop), so both are off.
#ifndef GG_BIAS_LDS
#define GG_BIAS_LDS 0      // 1: stage the tile's bias in LDS during the table setup
#endif
#ifndef GG_PREFETCH
#define GG_PREFETCH 0      // 1: fetch the store phase's gate / residual chunks before the register phase;
                           // 2: the gate chunks after it (accumulators dead): 0.4648 vs 0.4666 ms/step, same GPU, 3 runs: noise
#endif

// tile row m -> (image n, grid row a, grid column b); integer divisions only when the grid is not a power of two
// (the setup phases of a workgroup were ~2 us of mostly these divisions, against a 1.6 us single-slice K loop)
__device__ __forceinline__ void split_row(const GgArgs& p, int m, int& n, int& a, int& b) {
    int rem;
    if (p.sh_thw >= 0) { n = m >> p.sh_thw; rem = m & ((1 << p.sh_thw) - 1); }
    else { n = m / (p.TH * p.TW); rem = m - n * (p.TH * p.TW); }
    if (p.sh_tw >= 0) { a = rem >> p.sh_tw; b = rem & ((1 << p.sh_tw) - 1); }
    else { a = rem / p.TW; b = rem - a * p.TW; }
}

// phase stamp of a workgroup (wave 0 writes; s_memrealtime: constant 100 MHz).  Compiled in only with
// -DGG_STAMPS=1 (tools/ab_variants.sh): the conditional stores cost the product kernel ~2 % (they fence the
// compiler's scheduling of the epilogue loads).
#ifndef GG_STAMPS
#define GG_STAMPS 0
#endif
#if !GG_STAMPS
#define GG_STAMP(slot) do {} while (0)
#else
#define GG_STAMP(slot)                                                                                     \
    do {                                                                                                    \
        if (p.stamps && threadIdx.x == 0)                                                                   \
            p.stamps[((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8 + (slot)] = \
                wall_clock64();                                          \
    } while (0)
#endif

template <int N> __device__ __forceinline__ void wait_vmcnt_barrier() {
    // counted wait (LDS-DMA of the slice about to be read has landed for THIS wave), then the
    // workgroup barrier; no vmcnt(0) drain, so younger slices stay in flight across the barrier
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}

// GG_NS = LDS ring depth.  3: two K-slices in flight behind the one being multiplied (one workgroup per
// CU, deep K); 2: classic double buffer, two workgroups per CU; 1: single buffer for 1-2 slice problems
// where four workgroups per CU overlap each other's load / store latencies instead.
// OCC = waves per SIMD the register allocation must leave room for.  The single-buffer instances (1-2 slice
// problems: the 3/4-channel ends of the CNNs, the fc products) live on having four workgroups per CU in flight;
// the 128 x 128 one takes 230 registers unconstrained (2 per CU: a 1024-workgroup grid runs as two rounds) and
// spills accumulators when held to 128, the 128 x 64 one fits 128 registers without spilling.
// BM = tile rows (128; 64 for the 4-wave 64 x 64 tile of deep-K problems with few rows: at the same number of
// workgroups a square tile takes in the fewest operand bytes per K step, and the CU's intake is what bounds the loop).
template <typename T, int NT, int WAVES, int GG_NS, int OCC = 1, int BM = GG_BM>
__global__ __launch_bounds__(WAVES * 64, OCC) void gather_gemm_k(const GgArgs p) {
    constexpr int THREADS = WAVES * 64;
    static_assert(BM % 32 == 0 && BM + 16 <= THREADS, "tile rows");
    constexpr int BN = NT * 32;
    constexpr int ES = sizeof(T);
    constexpr int KE = 128 / ES;                 // k elements per staged row slice
    constexpr int EC = 16 / ES;                  // elements per 16-B chunk
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128;
    constexpr int STAGE = A_BYTES + B_BYTES;
    constexpr int PITCH = BN * ES + 16;          // epilogue tile row pitch
    // wave grid WR x WC over the 128 x BN tile; a wave owns MT x NTW MFMA tiles (16 x 16 each)
    constexpr int WR = 2, WC = WAVES / 2;
    constexpr int MT = (BM / 16) / WR, NTW = (2 * NT) / WC;
    constexpr int A_INSTR = (BM / 8) / WAVES, B_INSTR = (BN / 8) / WAVES;   // LDS-DMA instructions per wave per slice
    constexpr int LOADS = A_INSTR + B_INSTR;
    static_assert(NTW >= 1 && B_INSTR >= 1 && A_INSTR >= 1 && MT >= 1, "tile too small for this many waves");
    constexpr int RING = GG_NS * STAGE > BM * PITCH ? GG_NS * STAGE : BM * PITCH;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int* s_orow = (int*)(smem + RING);           // [128]
    int* s_tap = s_orow + BM;                 // [16][4]: widx, dh, dw of this class
    float* s_bias = (float*)(s_tap + 64);        // [BN]: this tile's bias (0 without one), loaded while the tables are built

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave id, provably uniform (scalar LDS addressing)
#ifndef GG_SMALL_PRIO
#define GG_SMALL_PRIO 0      // measured 0.516 (off) vs 0.522 ms/step (on), same GPU: off
#endif
#if GG_SMALL_PRIO
    // a launch of a few dozen workgroups is a latency-bound link of the main chain (the fc products); beside a
    // full-grid kernel of the side stream its waves take the issue slots first
    if (gridDim.x * gridDim.y * gridDim.z <= 64) __builtin_amdgcn_s_setprio(2);
#endif
    GG_STAMP(0);
    // Tile of this workgroup.  xcd_order (RBVAE_GG_XCD=1, off): the gridDim.y * gridDim.z workgroups that read the SAME
    // gathered rows (the N tiles and tap classes of one M tile) take consecutive ids of one XCD.  Measured on the bench
    // step, same GPU, 4 runs each: 0.452 ms (off) vs 0.462 ms (on), and FETCH_SIZE of the parity-class launches
    // unchanged (40.5 vs 39.8 MB: their reads are the input once per launch plus the 33.5 MB ReLU gate of the
    // backward one, not re-fetches) -- the plain order spreads the 1/2/2/4-tap classes over the CUs more evenly.
    unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (p.xcd_order) {
        const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        const unsigned nsub = gridDim.y * gridDim.z, r = lin >> 3;
        const unsigned g = r / nsub, sub = r - g * nsub;
        bx = g * 8 + (lin & 7);
        by = sub % gridDim.y;
        bz = sub / gridDim.y;
    }
    const TapClass& tc = p.cls[bz];
    const int Mc = p.Nimg * p.TH * p.TW;
    const int m0 = bx * BM, n0 = by * BN;
    const int ntaps = tc.ntaps;

    // output row of every tile row (for the store phase) and the class's tap table
#if GG_BIAS_LDS
    for (int i = tid; i < BN; i += THREADS) s_bias[i] = (p.bias && n0 + i < p.Nout) ? p.bias[n0 + i] : 0.f;
#endif
    if (tid < BM) {
        const int m = m0 + tid;
        int o = -1;
        if (m < Mc) {
            int n, a, b;
            split_row(p, m, n, a, b);
            const int oh = a * p.so + tc.oh0, ow = b * p.so + tc.ow0;
            if (oh < p.OH && ow < p.OW) o = (n * p.OH + oh) * p.OW + ow;
        }
        s_orow[tid] = o;
    } else if (tid < BM + 16) {
        const int j = tid - BM;
        s_tap[j * 4 + 0] = tc.widx[j];
        s_tap[j * 4 + 1] = tc.dh[j];
        s_tap[j * 4 + 2] = tc.dw[j];
    }
    __syncthreads();
    GG_STAMP(1);

    // staging roles: one LDS-DMA instruction moves 8 rows x 128 B; wave w issues rows
    // (w*A_INSTR+i)*8 .. +7 of A and (w*B_INSTR+i)*8 .. +7 of B
    const int srow = lane >> 3, schunk = lane & 7;
    int an[A_INSTR], aa[A_INSTR], ab[A_INSTR];
    unsigned a_sw[A_INSTR];
#pragma unroll
    for (int i = 0; i < A_INSTR; ++i) {
        const int r = (w * A_INSTR + i) * 8 + srow;
        const int m = m0 + r;
        a_sw[i] = (unsigned)((schunk ^ ((r >> 1) & 7)) * 16);
        if (m < Mc) {
            split_row(p, m, an[i], aa[i], ab[i]);
        } else {
            an[i] = -1; aa[i] = 0; ab[i] = 0;
        }
    }
    const unsigned char* bptr[B_INSTR];
    bool bval[B_INSTR];
#pragma unroll
    for (int i = 0; i < B_INSTR; ++i) {
        const int r = (w * B_INSTR + i) * 8 + srow;
        const int n = n0 + r;
        bval[i] = n < p.Nout;
        bptr[i] = p.W + ((size_t)(bval[i] ? n : 0) * p.taps_total * p.Kc) * ES + (schunk ^ ((r >> 1) & 7)) * 16;
    }
    const unsigned char* zsrc = p.zero + schunk * 16;

    const int kchunks = p.Kc / KE;
    const int nsteps = ntaps * kchunks;

    // producer state: the (tap, k-chunk) of the next slice to stage.  Per tap every row gets a current
    // source pointer and a per-chunk stride (128 B, or 0 for rows that read the zero page), so staging a
    // slice costs one 64-bit add per LDS-DMA instruction.
    const unsigned char* acur[A_INSTR];
    unsigned astep[A_INSTR];
    const unsigned char* bcur[B_INSTR];
    int pj = -1, pkc = 0, pbuf = 0;
    auto tap_setup = [&](int j) {
        const int widx = s_tap[j * 4 + 0], dh = s_tap[j * 4 + 1], dw = s_tap[j * 4 + 2];
        const size_t woff = ((size_t)widx * p.Kc) * ES;
#pragma unroll
        for (int i = 0; i < A_INSTR; ++i) {
            const int ih = aa[i] * p.sa + dh, iw = ab[i] * p.sa + dw;
            const bool v = an[i] >= 0 && ih >= 0 && ih < p.IH && iw >= 0 && iw < p.IW;
            acur[i] = v ? p.A + ((size_t)((an[i] * p.IH + ih) * p.IW + iw) * p.lda) * ES + a_sw[i] : zsrc;
            astep[i] = v ? 128u : 0u;
        }
#pragma unroll
        for (int i = 0; i < B_INSTR; ++i) bcur[i] = bval[i] ? bptr[i] + woff : zsrc;
    };
    auto stage_next = [&]() {
        if (pj < 0 || pkc == kchunks) { ++pj; pkc = 0; tap_setup(pj); }
        unsigned char* la = smem + pbuf * STAGE + (w * A_INSTR) * 1024;
#pragma unroll
        for (int i = 0; i < A_INSTR; ++i) {
            glds16(acur[i], la + i * 1024);
            acur[i] += astep[i];
        }
        unsigned char* lb = smem + pbuf * STAGE + A_BYTES + (w * B_INSTR) * 1024;
#pragma unroll
        for (int i = 0; i < B_INSTR; ++i) {
            glds16(bcur[i], lb + i * 1024);
            bcur[i] += bval[i] ? 128 : 0;
        }
        ++pkc;
        pbuf = (pbuf + 1 == GG_NS) ? 0 : pbuf + 1;
    };

    // fragment read offsets (swizzle depends on the lane only: tile rows are multiples of 16)
    const int fi = lane & 15, fg = lane >> 4;
    const int fsw = (fi >> 1) & 7;
    const int wr = w / WC, wc = w - wr * WC;
    int offA[2], offB[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int ch = ((4 * kk + fg) ^ fsw) * 16;
        offA[kk] = (wr * MT * 16 + fi) * 128 + ch;
        offB[kk] = A_BYTES + (wc * NTW * 16 + fi) * 128 + ch;
    }

    f32x4_t acc[MT][NTW];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) acc[mt][nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // One 32-k half of a slice: its MT + NTW fragment reads, and its MT x NTW MFMAs.
    //
    // The fragment reads are inline asm with explicit counted waits.  Left to the compiler, the loop got an
    // s_waitcnt lgkmcnt(0) in front of every MFMA group -- it waited for the reads of the NEXT half it had
    // just issued, so the LDS and the matrix pipe took turns (measured 1130 cycles per slice against 512 of
    // MFMA).  The asm reads are invisible to its wait-count pass; each group is guarded by a "+v"-tied
    // s_waitcnt, and tools/check_tr_asm.py proves on the ISA that nothing touches a fragment register
    // between a read and its wait.
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    constexpr int RPH = MT + NTW;                 // LDS reads of one half
    static_assert(RPH <= 15, "lgkmcnt is a 4-bit counter");
    auto read_half = [&](unsigned lbase, int kk, u32x4_t (&fa)[MT], u32x4_t (&fb)[NTW]) {
        const unsigned aa_ = lbase + offA[kk], ab_ = lbase + offB[kk];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[mt]) : "v"(aa_), "n"(mt * 2048));
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[nt]) : "v"(ab_), "n"(nt * 2048));
    };
    // wait until at most YOUNGER LDS reads are outstanding (in-order return): the guarded fragments landed
    auto landed = [&](auto younger_tag, u32x4_t (&fa)[MT], u32x4_t (&fb)[NTW]) {
        constexpr int YOUNGER = decltype(younger_tag)::value;
        static_assert((MT == 4 && (NTW == 1 || NTW == 2 || NTW == 4)) || (MT == 2 && NTW == 2) || (MT == 8 && NTW == 2), "operand list below");
        if constexpr (MT == 8)
            asm volatile("s_waitcnt lgkmcnt(%10)"
                         : "+v"(fa[0]), "+v"(fa[1 % MT]), "+v"(fa[2 % MT]), "+v"(fa[3 % MT]), "+v"(fa[4 % MT]), "+v"(fa[5 % MT]),
                           "+v"(fa[6 % MT]), "+v"(fa[7 % MT]), "+v"(fb[0]), "+v"(fb[1 % NTW])
                         : "n"(YOUNGER));
        else if constexpr (MT == 2)
            asm volatile("s_waitcnt lgkmcnt(%4)"
                         : "+v"(fa[0]), "+v"(fa[MT - 1]), "+v"(fb[0]), "+v"(fb[NTW - 1])
                         : "n"(YOUNGER));
        else if constexpr (NTW == 1)
            asm volatile("s_waitcnt lgkmcnt(%5)"
                         : "+v"(fa[0]), "+v"(fa[1]), "+v"(fa[2 % MT]), "+v"(fa[3 % MT]), "+v"(fb[0])
                         : "n"(YOUNGER));
        else if constexpr (NTW == 2)
            asm volatile("s_waitcnt lgkmcnt(%6)"
                         : "+v"(fa[0]), "+v"(fa[1]), "+v"(fa[2 % MT]), "+v"(fa[3 % MT]), "+v"(fb[0]), "+v"(fb[1 % NTW])
                         : "n"(YOUNGER));
        else
            asm volatile("s_waitcnt lgkmcnt(%8)"
                         : "+v"(fa[0]), "+v"(fa[1]), "+v"(fa[2 % MT]), "+v"(fa[3 % MT]), "+v"(fb[0]), "+v"(fb[1 % NTW]),
                           "+v"(fb[2 % NTW]), "+v"(fb[NTW - 1])
                         : "n"(YOUNGER));
    };
    auto mma_half = [&](const u32x4_t (&fa)[MT], const u32x4_t (&fb)[NTW]) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) Mma<T>::run(acc[mt][nt], fb[nt], fa[mt]);
    };
    using Younger = std::integral_constant<int, RPH>;
    using None = std::integral_constant<int, 0>;
    u32x4_t fa0[MT], fb0[NTW], fa1[MT], fb1[NTW];
    GG_STAMP(2);
    if constexpr (GG_NS == 1) {
        for (int s = 0; s < nsteps; ++s) {
            if (s > 0) __syncthreads();              // everyone is done reading the single buffer
            stage_next();
            wait_vmcnt_barrier<0>();
            // one fragment set, reused by both halves (registers, see the launch bounds)
            read_half(lds0, 0, fa0, fb0);
            landed(None{}, fa0, fb0);
            mma_half(fa0, fb0);
            __builtin_amdgcn_sched_barrier(0);
            read_half(lds0, 1, fa0, fb0);
            landed(None{}, fa0, fb0);
            mma_half(fa0, fb0);
        }
    } else {
        // Software pipeline over the 32-k halves: while the matrix pipe works on one half the LDS serves the
        // next one, and the slice barrier sits between two MFMA groups whose operands are already in
        // registers.  Ring: slice s is being read, slices s+1 .. s+GG_NS-1 are landed or in flight.
#pragma unroll
        for (int i = 0; i < GG_NS - 1; ++i)
            if (i < nsteps) stage_next();
        if (nsteps > GG_NS - 1 && GG_NS > 2) wait_vmcnt_barrier<(GG_NS > 2 ? GG_NS - 2 : 0) * LOADS>();
        else wait_vmcnt_barrier<0>();
        if (GG_NS - 1 < nsteps) stage_next();
        int cbuf = 0;
        read_half(lds0, 0, fa0, fb0);
        // Waves w and w+4 share a SIMD and leave every barrier together.  Issuing an LDS-DMA piece holds a
        // wave for ~60-180 cycles, so the two take turns: the low wave stages the next slice right after the
        // barrier while its partner multiplies, the high wave stages after that MFMA group.
        const bool late = WAVES == 8 && w >= 4 && p.dephase;
        // every step but the last: multiply slice s while slice s+1's first half is read behind it
        for (int s = 0; s + 1 < nsteps; ++s) {
            const unsigned lcur = lds0 + cbuf * STAGE;
            cbuf = (cbuf + 1 == GG_NS) ? 0 : cbuf + 1;
            read_half(lcur, 1, fa1, fb1);
            landed(Younger{}, fa0, fb0);                 // half 0 landed, half 1 in flight
            __builtin_amdgcn_sched_barrier(0);
            mma_half(fa0, fb0);
            __builtin_amdgcn_sched_barrier(0);
            landed(None{}, fa1, fb1);                    // issued a whole MFMA group ago (one wait site per set)
            // slice s+1 landed (GG_NS-2 younger ones may stay in flight); after the barrier every wave
            // holds both halves of slice s in registers, so its buffer can be refilled
            if (GG_NS > 2 && nsteps - s - 2 >= GG_NS - 2) wait_vmcnt_barrier<(GG_NS > 2 ? GG_NS - 2 : 0) * LOADS>();
            else wait_vmcnt_barrier<0>();
            if (!late && s + GG_NS < nsteps) stage_next();
            read_half(lds0 + cbuf * STAGE, 0, fa0, fb0);
            __builtin_amdgcn_sched_barrier(0);
            mma_half(fa1, fb1);
            __builtin_amdgcn_sched_barrier(0);
            if (late && s + GG_NS < nsteps) stage_next();
        }
        // the last step reads nothing ahead: when it is done no LDS read is in flight (the loop above is peeled
        // this way so that the build's ISA check can see that on every path, without a catch-all wait)
        {
            const unsigned lcur = lds0 + cbuf * STAGE;
            read_half(lcur, 1, fa1, fb1);
            landed(Younger{}, fa0, fb0);
            __builtin_amdgcn_sched_barrier(0);
            mma_half(fa0, fb0);
            __builtin_amdgcn_sched_barrier(0);
            landed(None{}, fa1, fb1);
            __builtin_amdgcn_sched_barrier(0);
            mma_half(fa1, fb1);
        }
    }
    GG_STAMP(3);
    __syncthreads();

    // ---- store roles, and the global operands of the store phase (saved-activation gate, residual) fetched now,
    // so their latency runs under the register phase and the LDS round trip instead of once per store pass
    constexpr int CPR = BN / EC;                 // 16-B chunks per tile row
    constexpr int RL = THREADS / CPR;            // row lanes of the store phase
    constexpr int ITERS = BM / RL;            // store passes: pass `it` handles tile row it * RL + rl
    constexpr bool PREFETCH = GG_PREFETCH && ITERS <= 8;
    constexpr bool PF_LATE = GG_PREFETCH == 2;   // gate chunks fetched after the register phase (accumulators dead)
    const int sch = tid % CPR, rl = tid / CPR;
    const int scol = n0 + sch * EC;
    int orow_[ITERS];
    u32x4_t gv[PREFETCH ? ITERS : 1], av[PREFETCH ? ITERS : 1];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int o = s_orow[it * RL + rl];
        orow_[it] = scol < p.Nout ? o : -1;
        if constexpr (PREFETCH && !PF_LATE) {
            const size_t off = ((size_t)(orow_[it] < 0 ? 0 : orow_[it]) * p.ldo + (orow_[it] < 0 ? 0 : scol)) * ES;
            if (p.gate) gv[it] = *(const u32x4_t*)(p.gate + off);
            if (p.addend) av[it] = *(const u32x4_t*)(p.addend + off);
        }
    }

    // ---- epilogue, register phase: bias, relu, scale; lane owns pixel fi, channels 4*fg..+3
    unsigned char* tile = smem;
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        const int cb = (wc * NTW + nt) * 16 + 4 * fg;        // tile-local channel
#if GG_BIAS_LDS
        const float4 b4 = *(const float4*)(s_bias + cb);
        const float bz[4] = {b4.x, b4.y, b4.z, b4.w};
#else
        float bz[4] = {0.f, 0.f, 0.f, 0.f};
        if (p.bias && n0 + cb < p.Nout) {
            const float4 b4 = *(const float4*)(p.bias + n0 + cb);
            bz[0] = b4.x; bz[1] = b4.y; bz[2] = b4.z; bz[3] = b4.w;
        }
        (void)s_bias;
#endif
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int row = (wr * MT + mt) * 16 + fi;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float x = acc[mt][nt][r] + bz[r];
                if (p.relu) x = fmaxf(x, 0.f);
                v[r] = x * p.scale;
            }
            unsigned char* dst = tile + row * PITCH + cb * ES;
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
    if constexpr (PREFETCH && PF_LATE) {
        // all store passes' gate chunks in flight at once, while the tile makes its LDS round trip: fetched at their use
        // each pass waits for its own load AND (vmcnt counts stores too) for the previous pass's stores
        if (p.gate) {
#pragma unroll
            for (int it = 0; it < ITERS; ++it) {
                const size_t off = ((size_t)(orow_[it] < 0 ? 0 : orow_[it]) * p.ldo + (orow_[it] < 0 ? 0 : scol)) * ES;
                gv[it] = *(const u32x4_t*)(p.gate + off);
            }
        }
    }
    __syncthreads();
    GG_STAMP(4);

    // ---- store phase: whole 16-B chunks of NHWC rows; dropout / gate zeroing happens here
    float csum[EC];
#pragma unroll
    for (int e = 0; e < EC; ++e) csum[e] = 0.f;
    DropKey dkey{0u, 0u};
    if (p.drop_mode == 1) dkey = drop_key(p.seed + (p.seed_dev ? p.seed_dev[0] * 0x9E3779B97F4A7C15ull : 0ull));
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int row = it * RL + rl;
        const int orow = orow_[it];
        if (orow >= 0) {
            const int col = scol;
            u32x4_t val = *(const u32x4_t*)(tile + row * PITCH + sch * 16);
            T* ev = (T*)&val;
            if (p.drop_mode == 1) {
                const unsigned run = drop_run(dkey, (unsigned long long)orow * p.Nout + col);
                if constexpr (sizeof(T) == 2) drop_chunk_zero_b16<EC>(run, p.drop_thresh >> 16, (unsigned*)&val);
                else drop_chunk_zero_f32<EC>(run, p.drop_thresh >> 16, (float*)&val);
            } else if (p.drop_mode == 2) {
                const unsigned char* mk = p.mask + (size_t)orow * p.Nout + col;
#pragma unroll
                for (int e = 0; e < EC; ++e)
                    if (!mk[e]) ev[e] = 0;
            }
            if (p.addend) {
                u32x4_t a4;
                if constexpr (PREFETCH && !PF_LATE) a4 = av[it];
                else a4 = *(const u32x4_t*)(p.addend + ((size_t)orow * p.ldo + col) * ES);
                const T* ae = (const T*)&a4;
#pragma unroll
                for (int e = 0; e < EC; ++e) Elem<T>::store(ev + e, Elem<T>::load(ev + e) + Elem<T>::load(ae + e));
            }
            if (p.gate) {
                u32x4_t g4;
                if constexpr (PREFETCH) g4 = gv[it];
                else g4 = *(const u32x4_t*)(p.gate + ((size_t)orow * p.ldo + col) * ES);
#pragma unroll
                for (int e = 0; e < EC; ++e)
                    if (!elem_pos<T>((const unsigned char*)&g4, e)) ev[e] = 0;
            }
            *(u32x4_t*)(p.Out + ((size_t)orow * p.ldo + col) * ES) = val;
            if (p.colsum_ws) {
#pragma unroll
                for (int e = 0; e < EC; ++e) csum[e] += Elem<T>::load(ev + e);
            }
        }
    }
    GG_STAMP(5);
    if (p.colsum_ws) {
        // bias gradient: column sums of this tile's stored rows, reduced over the row lanes in LDS.  LDS-only barriers:
        // __syncthreads() also waits vmcnt(0), i.e. for the round trip of the tile's global stores just issued
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        float* red = (float*)smem;                     // [RL][BN]
#pragma unroll
        for (int e = 0; e < EC; ++e) red[rl * BN + sch * EC + e] = csum[e];
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (tid < BN && n0 + tid < p.Nout) {
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < RL; ++k) t += red[k * BN + tid];
            p.colsum_ws[((size_t)bz * gridDim.x + bx) * p.Nout + n0 + tid] = t;
        }
    }
#if GG_STAMPS
    if (p.stamps) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's stores acknowledged
        GG_STAMP(6);
    }
#endif
}

template <typename T, int NT, int WAVES, int NS, int OCC = 1, int BM = GG_BM>
static int launch_gg(const GgArgs& a, hipStream_t st) {
    constexpr int BN = NT * 32;
    constexpr int ES = sizeof(T);
    const int Mc = a.Nimg * a.TH * a.TW;
    const size_t ring = (size_t)NS * (BM * 128 + BN * 128);
    const size_t tile = (size_t)BM * (BN * ES + 16);
    const size_t lds = (ring > tile ? ring : tile) + BM * sizeof(int) + 64 * sizeof(int) + BN * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)gather_gemm_k<T, NT, WAVES, NS, OCC, BM>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds);
        attr_set = true;
    }
    dim3 grid(cdiv(Mc, BM), cdiv(a.Nout, BN), a.nclass);
    hipLaunchKernelGGL((gather_gemm_k<T, NT, WAVES, NS, OCC, BM>), grid, dim3(WAVES * 64), lds, st, a);
    RBVAE_CHECK_LAUNCH("gather_gemm");
    return RBVAE_OK;
}

template <typename T>
static int dispatch_gg(const GgArgs& a, hipStream_t st, int max_steps) {
    const long blocks = (long)cdiv(a.Nimg * a.TH * a.TW, GG_BM) * cdiv(a.Nout, a.Nout > 64 ? 128 : 64) * a.nclass;
    constexpr int force = 0;
    constexpr int dbg = 0;
    int ns = max_steps <= 2 ? 1 : (blocks > 256 ? 2 : 3);
    if (force) ns = force;
    // one workgroup per CU and a deep K (the 256-channel convolutions at 256 frames): a ring of FOUR, three slices in flight --
    // a K step of the 128 x 128 tile is ~0.27 us, two of them are less lead than an L2 miss takes (bench step -0.6 %, same box)
    if (ns == 3 && blocks > 128 && max_steps >= 8) ns = 4;
    // a deep-K product with few rows and <= 64 columns (the LDM encoder's conv_out: 16 384 rows, K = 4608, 8 columns): 64-row
    // tiles put a workgroup on every CU instead of on half of them (55 -> 30 us)
    if (a.Nout <= 64 && blocks <= 128 && max_steps >= 16 && sizeof(T) == 2 && !a.colsum_ws && !a.xcd_order)
        return launch_gg<T, 2, 4, 3, 1, 64>(a, st);
    if (a.Nout <= 64) return ns == 1 ? launch_gg<T, 2, 4, 1>(a, st) : launch_gg<T, 2, 4, 2>(a, st);
    // deep-K problems with too few 128x128 tiles for the 256 CUs: narrower tiles (more workgroups, shorter steps)
    constexpr int small = 2;
    if (ns == 3 && !force && !dbg && small) {
        // 64 x 64 tiles for the 64-block problems (conv3 forward, first deconv's input gradient at 256 frames): the same
        // 256 workgroups as 128 x 32 tiles at 16 KB instead of 20 KB of operands per K step.  The fused column sums keep
        // the 128-row tiles (their partial-sum rows are counted in 128-row tiles by the callers); xcd_order too.
        constexpr int sq = 1;
        if (sq && small >= 2 && blocks <= 64 && !a.colsum_ws && !a.xcd_order && sizeof(T) == 2)
            return launch_gg<T, 2, 4, 3, 1, 64>(a, st);
        if (small >= 2 && blocks <= 64) return launch_gg<T, 1, 4, 3>(a, st);
        if (blocks <= 128) return launch_gg<T, 2, 4, 3>(a, st);
    }
    constexpr int one = 1;
    if (ns == 1 && one >= 1 && blocks > 512) return launch_gg<T, 2, 4, 1, 4>(a, st);
    // few 128-wide tiles (the fc products at 256 frames: 64): 128 x 32 tiles put a workgroup on every CU
    if (ns == 1 && one >= 2 && blocks <= 64) return launch_gg<T, 1, 4, 1>(a, st);
    if (ns == 1) return launch_gg<T, 4, 4, 1>(a, st);
    if (ns >= 2 && (dbg == 5 || dbg == 6)) return dbg == 5 ? launch_gg<T, 2, 4, 3>(a, st) : launch_gg<T, 2, 4, 2>(a, st);
    // (128 x 256 tiles -- <8, 8, 2>, the gathered rows read once instead of twice -- measured 37.2 us against 34.6 us for
    // the 1024-workgroup parity-class launches: one 8-wave workgroup per CU and two uneven rounds cost more than the
    // 25 % smaller ingest gains)
    // (256 x 128 tiles -- <4, 8, 2, 1, 256>, 64 accumulators per lane, one workgroup per CU -- for the launches of many
    // rounds: native 4x88x160 step 2.26-2.28 ms against 2.22, same GPU, alternating: two 128-row workgroups per CU hide each
    // other's barriers and epilogues better than the 25 % smaller ingest of the tall tile gains)
    if (ns == 2) return launch_gg<T, 4, 8, 2>(a, st);
    if (dbg == 4) return launch_gg<T, 4, 4, 3>(a, st);       // 4 waves, 64x64 wave tiles (less LDS traffic)
    if (ns >= 4) return launch_gg<T, 4, 8, 4>(a, st);
    return launch_gg<T, 4, 8, 3>(a, st);
}

}  // namespace rbvae

using namespace rbvae;

static unsigned long long* g_gg_stamps = nullptr;
#if GG_STAMPS
/* stamped builds only (include/rbvae_dbg.h): every later rbvae_gather_gemm launch writes 8 phase stamps per
 * workgroup into buf (null = off) */
extern "C" int rbvae_dbg_gg_stamps(unsigned long long* buf, void* stream) {
    (void)stream;
    g_gg_stamps = buf;
    return RBVAE_OK;
}
#endif

extern "C" int rbvae_gather_gemm(int dtype, const void* A, const void* W, void* Out, const float* bias,
                                 const void* gate, const void* mask, const void* addend, const void* zero_page, int Nimg, int IH,
                                 int IW, int TH, int TW, int sa, int OH, int OW, int so, int Kc, int Nout, int lda,
                                 int ldo, int taps_total, int nclass, const int* class_desc, int relu,
                                 int drop_mode, float drop_p, float scale, unsigned long long seed,
                                 const unsigned long long* seed_dev, float* colsum_ws, void* stream) {
    RBVAE_CHECK_ARG(A && W && Out && zero_page && class_desc, "gather_gemm: null pointer");
    RBVAE_CHECK_ARG(dtype == RBVAE_F32 || dtype == RBVAE_BF16, "gather_gemm: dtype %d", dtype);
    const int ES = dtype == RBVAE_F32 ? 4 : 2;
    const int KE = 128 / ES;
    RBVAE_CHECK_ARG(Kc > 0 && Kc % KE == 0, "gather_gemm: Kc=%d must be a multiple of %d", Kc, KE);
    RBVAE_CHECK_ARG(Nout > 0 && Nout % 8 == 0, "gather_gemm: Nout=%d must be a multiple of 8", Nout);
    RBVAE_CHECK_ARG(lda >= Kc && (lda * ES) % 16 == 0 && ldo >= Nout && (ldo * ES) % 16 == 0,
                    "gather_gemm: leading dimensions lda=%d ldo=%d", lda, ldo);
    RBVAE_CHECK_ARG(nclass >= 1 && nclass <= 4, "gather_gemm: nclass=%d", nclass);
    RBVAE_CHECK_ARG(Nimg > 0 && IH > 0 && IW > 0 && TH > 0 && TW > 0 && OH > 0 && OW > 0, "gather_gemm: bad grid");
    RBVAE_CHECK_ARG((long)Nimg * TH * TW < (1l << 30) && (long)Nimg * OH * OW < (1l << 30) &&
                        (long)Nimg * IH * IW < (1l << 30), "gather_gemm: more than 2^30 pixel rows");
    RBVAE_CHECK_ARG(((uintptr_t)A | (uintptr_t)W | (uintptr_t)Out | (uintptr_t)zero_page | (uintptr_t)gate) % 16 == 0,
                    "gather_gemm: pointers must be 16-byte aligned");
    RBVAE_CHECK_ARG(drop_mode >= 0 && drop_mode <= 2 && (drop_mode != 2 || mask), "gather_gemm: drop_mode/mask");
    GgArgs a;
    a.A = (const unsigned char*)A; a.W = (const unsigned char*)W; a.Out = (unsigned char*)Out; a.bias = bias;
    a.gate = (const unsigned char*)gate; a.mask = (const unsigned char*)mask; a.addend = (const unsigned char*)addend;
    a.zero = (const unsigned char*)zero_page;
    a.Nimg = Nimg; a.IH = IH; a.IW = IW; a.TH = TH; a.TW = TW; a.sa = sa; a.OH = OH; a.OW = OW; a.so = so;
    a.Kc = Kc; a.Nout = Nout; a.lda = lda; a.ldo = ldo; a.taps_total = taps_total;
    a.relu = relu; a.drop_mode = drop_mode; a.scale = scale; a.seed = seed; a.seed_dev = seed_dev; a.colsum_ws = colsum_ws;
    a.drop_thresh = (unsigned)((double)drop_p * 4294967296.0);
    a.nclass = nclass;
    constexpr int dephase = 1;
    a.dephase = dephase;
    constexpr int xcd = 0;
    a.xcd_order = xcd && cdiv(Nimg * TH * TW, GG_BM) % 8 == 0 && (nclass > 1 || cdiv(Nout, Nout > 64 ? 128 : 64) > 1);
    a.stamps = g_gg_stamps;
    auto log2_or_neg = [](int v) { int s = 0; while ((1 << s) < v) ++s; return (1 << s) == v ? s : -1; };
    a.sh_thw = log2_or_neg(TH * TW);
    a.sh_tw = log2_or_neg(TW);
    // class_desc (host ints): per class [ntaps, oh0, ow0, then ntaps x (widx, dh, dw)], classes back to back
    const int* d = class_desc;
    for (int c = 0; c < nclass; ++c) {
        TapClass& t = a.cls[c];
        t.ntaps = d[0]; t.oh0 = d[1]; t.ow0 = d[2];
        RBVAE_CHECK_ARG(t.ntaps >= 1 && t.ntaps <= 16, "gather_gemm: class %d has %d taps", c, t.ntaps);
        d += 3;
        for (int j = 0; j < t.ntaps; ++j) {
            RBVAE_CHECK_ARG(d[0] >= 0 && d[0] < taps_total, "gather_gemm: tap index %d outside [0,%d)", d[0], taps_total);
            t.widx[j] = (signed char)d[0]; t.dh[j] = (signed char)d[1]; t.dw[j] = (signed char)d[2];
            d += 3;
        }
    }
    hipStream_t st = (hipStream_t)stream;
    int max_taps = 0;
    for (int c = 0; c < nclass; ++c) max_taps = a.cls[c].ntaps > max_taps ? a.cls[c].ntaps : max_taps;
    const int max_steps = max_taps * (Kc / KE);
    return dtype == RBVAE_F32 ? dispatch_gg<float>(a, st, max_steps) : dispatch_gg<bf16_t>(a, st, max_steps);
}
