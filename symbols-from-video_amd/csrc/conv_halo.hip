// Halo-resident 3x3 convolution on the gfx950 matrix cores (stride 1): the LDM / Stable-Diffusion VAE encoder's
// ResnetBlock convolutions (src/stable-diffusion/ldm/modules/diffusionmodules/model.py:82-141, conv_in/conv_out
// :368-459) and any other NHWC 3x3 stride-1 convolution of the path.
//
//   Out[n][oh][ow][co] = epi( sum_{kh,kw,ci} f(A[n][oh+kh+dh0][ow+kw+dw0][ci]) * W[co][kh*3+kw][ci] )
//
// rbvae_gather_gemm re-gathers one pixel row per tap and K slice: a stride-1 tile takes its input in nine times
// through the CU's L2 -> LDS path, which is what bounds that kernel.  Here a workgroup owns a 16 x 16 patch of output
// pixels x 128 output channels; per 128-byte channel slice (64 bf16 / 32 f32 channels) the 18 x 18 input patch
// (tile + halo) is staged into LDS ONCE and all nine taps read it, so per slice the workgroup takes in 41 KB of
// input + 9 x 16 KB of weights for 256 x 128 x 576 MACs (200 FLOP per byte against 64 of the 128 x 128 gather tile).
//
// LDS images
//   * input patch, chunk-major: [8 chunks of 16 B][336 slots], slot = patch_row * 18 + patch_col.  A tap (kh, kw)
//     shifts every lane's slot by the same kh*18+kw, so a fragment read is base + tap offset: no per-lane gather, and
//     16 consecutive slots of one chunk are 256 contiguous bytes = conflict-free ds_read_b128 at every shift (planes
//     are multiples of 256 B; the (chunk>>1)*32 stagger keeps the transposing ds_write_b128 at a 2-way conflict).
//     Filled through registers (coalesced 128-B pixel rows in, transposed on the way to LDS), double buffered; the
//     register pass is also where the producer's GroupNorm + swish is applied (gn_scale / gn_shift: model.py:38-39,
//     121-131), so a normalised copy of the activation never exists in HBM.
//   * weight tap tiles [128 co][128 B], XOR-swizzled 16-B chunks, LDS-DMA ring (as gather_gemm.hip).
// 8 waves as 4 (pixel rows) x 2 (channel halves), 64 x 64 per wave on v_mfma_f32_16x16x32_bf16 / 16x16x4_f32, weights
// as the MFMA row operand so a lane owns 4 consecutive output channels of one pixel; the tile leaves through LDS as
// whole 16-B chunks of NHWC rows (bias, residual addend), optionally with the tile's GroupNorm partial statistics
// (per image and group: mean and sum of squared deviations) for the NEXT GroupNorm.
#include "conv_halo.h"

namespace rbvae {

// bytes of the staging buffers / epilogue tile + statistics scratch (the tables follow)
template <typename T, int RING> constexpr int ch_lds_main() {
    constexpr int ES = sizeof(T);
    constexpr int ring = 2 * CH_ABUF + RING * CH_BBYTES;
    constexpr int epi = CH_BM * (CH_BN * ES + 16);          // (the statistics scratch reuses the tile)
    return ring > epi ? ring : epi;
}

// build-time ablation switches for timing-only builds (tools/ab_variants.sh; results are wrong with any bit set):
// 1 no weight LDS-DMA in the loop, 2 no workgroup barrier in the loop, 4 no fragment reads in the loop, 8 no patch staging
#ifndef CH_ABL
#define CH_ABL 0
#endif
#if CH_ABL && !defined(RBVAE_ABLATION)
#error "CH_ABL builds give wrong results: define RBVAE_ABLATION to confirm"
#endif

template <int N> __device__ __forceinline__ void ch_wait_barrier() {
#if CH_ABL & 2
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)" ::"n"(N) : "memory");
#else
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
#endif
}

template <typename T, int RING, bool GN>
__global__ __launch_bounds__(512) void conv_halo_k(const ChArgs p) {
    constexpr int ES = sizeof(T);
    constexpr int KE = 128 / ES, EC = 16 / ES;
    constexpr int MT = 4, NTW = 4;
    constexpr int LOADS = 2;                                  // weight LDS-DMA instructions per wave and step
    constexpr int PITCH = CH_BN * ES + 16;
    constexpr int A_BYTES = 2 * CH_ABUF;
    constexpr int CPR = CH_BN / EC;              // 16-B chunks per tile row
    constexpr int RL = 512 / CPR;                // row lanes of the store phase
    constexpr int ITERS = CH_BM / RL;
    constexpr int RINGB = ch_lds_main<T, RING>();
    static_assert(CH_BM * PITCH <= RINGB && A_BYTES + RING * CH_BBYTES <= RINGB, "LDS carve");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int* s_orow = (int*)(smem + RINGB);            // [256]
    int* s_pix = s_orow + CH_BM;                   // [336]
    float* s_gsc = (float*)(s_pix + CH_NSLOT_PAD); // [Kc] scale, then [Kc] shift (only with gn_scale)
    float* s_gsh = s_gsc + p.Kc;

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);

    // work item: the N tiles of one pixel tile run back to back on ONE XCD (they re-read the same patch from its L2), and
    // an XCD walks a contiguous range of pixel tiles (neighbours share halo rows).  Bijective for any total.
    int item;
    {
        const int lin = blockIdx.x, xcd = lin & 7, q = p.total >> 3, r = p.total & 7;
        item = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (lin >> 3);
    }
    const int mtile = item / p.ntn, ntile = item - mtile * p.ntn;
    const int per_img = p.tiles_r * p.tiles_c;
    const int n = mtile / per_img, trc = mtile - n * per_img;
    const int tr = trc / p.tiles_c, tc = trc - tr * p.tiles_c;
    const int r0 = tr * CH_T, c0 = tc * CH_T, n0 = ntile * CH_BN;

    // ---- tables
    if (tid < CH_BM) {
        const int oh = r0 + (tid >> 4), ow = c0 + (tid & 15);
        s_orow[tid] = (oh < p.OH && ow < p.OW) ? (n * p.OH + oh) * p.OW + ow : -1;
    }
    if (tid < CH_NSLOT_PAD) {
        int v = -1;
        if (tid < CH_NSLOT) {
            const int pr = tid / CH_PW, pc = tid - pr * CH_PW;
            const int ih = r0 + pr + p.dh0, iw = c0 + pc + p.dw0;
            if (ih >= 0 && ih < p.IH && iw >= 0 && iw < p.IW) v = (n * p.IH + ih) * p.IW + iw;
        }
        s_pix[tid] = v;
    }
    if (p.gn_scale) {
        for (int i = tid; i < p.Kc; i += 512) {
            s_gsc[i] = p.gn_scale[(size_t)n * p.Kc + i];
            s_gsh[i] = p.gn_shift[(size_t)n * p.Kc + i];
        }
    }
    __syncthreads();

    // ---- input staging roles: piece i of this thread = (slot (tid>>3) + 64 i, chunk tid&7): 8 lanes read one pixel's
    // 128-byte slice (coalesced), and write it to 8 chunk planes
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const int chunk = tid & 7;
    const unsigned char* zsrc = p.zero + chunk * 16;
    // source pixel row of this thread's six pieces: -1 zero (padding), -2 no such slot.  Kept in registers: a
    // compiler-issued LDS read inside the loop waits lgkmcnt(0), i.e. for the fragment reads in flight around it
    int pixr[CH_NA];
#pragma unroll
    for (int i = 0; i < CH_NA; ++i) {
        const int slot = (tid >> 3) + 64 * i;
        pixr[i] = slot < CH_NSLOT_PAD ? s_pix[slot] : -2;
    }
    // (a runtime-indexed pick of pixr[i] becomes a scratch array: the staged steps rotate the six registers instead --
    // one full turn per slice, so every a_load sees them in order)
    // GroupNorm scale / shift of this thread's chunk for the slice being staged (loaded right behind a barrier)
    float gsc[EC], gsh[EC];
#pragma unroll
    for (int e = 0; e < EC; ++e) { gsc[e] = 1.f; gsh[e] = 0.f; }
    auto gn_coeffs = [&](int kc) {
        if constexpr (GN) {
            const int cb = kc * KE + chunk * EC;
#pragma unroll
            for (int e = 0; e < EC; ++e) { gsc[e] = s_gsc[cb + e]; gsh[e] = s_gsh[cb + e]; }
#pragma unroll
            for (int e = 0; e < EC; ++e) asm volatile("" : "+v"(gsc[e]), "+v"(gsh[e]));   // landed HERE
        }
    };
    // The six staged pieces of a slice: asm loads (the compiler would drain the weight tiles' LDS-DMA in front of a load
    // of its own), destinations tied to the counted wait that covers them.  Loads, wait and uses all sit inside ONE
    // unrolled slice body (taps 1, 3 and 3..8): a value that stayed in flight across the loop's back edge got copied by
    // the compiler's phi moves right behind the load instruction, before its data had landed.  The build's ISA check
    // proves that nothing touches a destination between its load and the wait.
    u32x4_t areg[CH_NA];
    auto a_load = [&](int kc) {
#pragma unroll
        for (int i = 0; i < CH_NA; ++i) {
            const int pv = pixr[i];
            const unsigned char* src = pv >= 0 ? p.A + ((size_t)pv * p.lda) * ES + (size_t)kc * 128 + chunk * 16 : zsrc;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(areg[i]) : "v"(src) : "memory");
        }
    };
    // wait until at most YOUNGER vector-memory operations are outstanding: the six staged pieces have landed
    auto a_landed = [](auto younger_tag, u32x4_t (&ar)[CH_NA]) {
        constexpr int YOUNGER = decltype(younger_tag)::value;
        asm volatile("s_waitcnt vmcnt(%6)"
                     : "+v"(ar[0]), "+v"(ar[1]), "+v"(ar[2]), "+v"(ar[3]), "+v"(ar[4]), "+v"(ar[5])
                     : "n"(YOUNGER));
    };
    // the producer's GroupNorm (+ swish) on one staged 16-byte piece of slice kc (padding pixels stay zero)
    auto gn_piece = [&](u32x4_t v, int kc, int pv) -> u32x4_t {
        if constexpr (!GN) return v;
        u32x4_t t = v;
        T* ev = (T*)&t;
#pragma unroll
        for (int e = 0; e < EC; ++e) {
            float x = fmaf(Elem<T>::load(ev + e), gsc[e], gsh[e]);
            if (p.gn_swish) {
                if constexpr (ES == 2) x = x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504f * x));
                else x = x * sigmoidf_(x);
            }
            Elem<T>::store(ev + e, x);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) t[q] = pv >= 0 ? t[q] : v[q];
        return t;
    };
    // (asm store: a compiler-issued LDS store would first drain the weight tiles' LDS-DMA -- it cannot tell them apart)
    auto piece_write = [&](const u32x4_t& v, int buf, int i, int pv) {
        const unsigned dst = lds0 + (unsigned)buf * CH_ABUF + ch_plane_off(chunk) + (unsigned)((tid >> 3) + 64 * i) * 16;
        if (pv >= -1) asm volatile("ds_write_b128 %0, %1" ::"v"(dst), "v"(v) : "memory");
    };
    auto a_write = [&](int kc, int buf) {                  // all six pieces at once (prologue)
        gn_coeffs(kc);
#pragma unroll
        for (int i = 0; i < CH_NA; ++i) piece_write(gn_piece(areg[i], kc, pixr[i]), buf, i, pixr[i]);
    };

    // ---- weight staging roles (LDS-DMA): instruction i of wave w moves rows (w*2+i)*8 .. +7 of the tap tile
    static_assert(RING == 3, "the unrolled slice body relies on 9 taps % RING == 0 (static ring slots)");
    const int srow = lane >> 3, schunk = lane & 7;
    // source = uniform base (tile, tap, slice: scalar registers) + a 32-bit per-lane offset (row, swizzled chunk): with
    // per-tap 64-bit lane pointers the unrolled slice body kept 18 of them live and spilled
    unsigned blane[LOADS];
#pragma unroll
    for (int i = 0; i < LOADS; ++i) {
        const int r = (w * LOADS + i) * 8 + srow;
        blane[i] = (unsigned)(r * 9 * p.Kc) * ES + (unsigned)((schunk ^ ((r >> 1) & 7)) * 16);
    }
    const unsigned char* wtile = p.W + ((size_t)n0 * 9 * p.Kc) * ES;
    const int nkc = p.Kc / KE;
    // weight tile of (slice kc, tap j) -> ring slot j % 3
    auto b_issue = [&](int kc, auto j_tag) {
        constexpr int j = decltype(j_tag)::value;
#if !(CH_ABL & 1)
        const unsigned char* src = wtile + ((size_t)j * p.Kc) * ES + (size_t)kc * 128;
        unsigned char* lb = smem + A_BYTES + (j % RING) * CH_BBYTES + (w * LOADS) * 1024;
#pragma unroll
        for (int i = 0; i < LOADS; ++i) ch_glds16(src + blane[i], lb + i * 1024);
#endif
    };

    // ---- fragment addresses
    const int fi = lane & 15, fg = lane >> 4;
    const int wr = w >> 1, wc = w & 1;
    unsigned abase[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
        abase[mt] = lds0 + (unsigned)fg * CH_PLANE + (unsigned)(fg >> 1) * 32 + (unsigned)((wr * MT + mt) * CH_PW + fi) * 16;
    const int fsw = (fi >> 1) & 7;
    const unsigned offB0 = lds0 + A_BYTES + (unsigned)(wc * NTW * 16 + fi) * 128 + (unsigned)(((0 + fg) ^ fsw) * 16);
    const unsigned offB1 = lds0 + A_BYTES + (unsigned)(wc * NTW * 16 + fi) * 128 + (unsigned)(((4 + fg) ^ fsw) * 16);

    f32x4_t acc[MT][NTW];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) acc[mt][nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // One 32-k half of a step: MT + NTW fragment reads (inline asm, counted waits: see gather_gemm.hip) and MT x NTW
    // MFMAs.  The tap and the ring slot are compile-time constants (the nine taps of a slice are unrolled): the tap's
    // shift of the patch and the weight tile's slot are immediate offsets of the reads.
    unsigned ta[MT];                         // patch read addresses of the slice being read (buffer folded in)
    auto set_slice = [&](int kc) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) ta[mt] = abase[mt] + (unsigned)((kc & 1) * CH_ABUF);
    };
    auto read_half = [&](auto j_tag, auto kk_tag, u32x4_t (&fa)[MT], u32x4_t (&fb)[NTW]) {
        constexpr int j = decltype(j_tag)::value, kk = decltype(kk_tag)::value;
        constexpr int aoff = kk * CH_KKOFF + ((j / 3) * CH_PW + (j % 3)) * 16;
        constexpr int boff = (j % RING) * CH_BBYTES;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[mt]) : "v"(ta[mt]), "n"(aoff));
        const unsigned ab_ = kk == 0 ? offB0 : offB1;
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[nt]) : "v"(ab_), "n"(boff + nt * 2048));
    };
    auto landed = [&](auto younger_tag, u32x4_t (&fa)[MT], u32x4_t (&fb)[NTW]) {
        constexpr int YOUNGER = decltype(younger_tag)::value;
        asm volatile("s_waitcnt lgkmcnt(%8)"
                     : "+v"(fa[0]), "+v"(fa[1]), "+v"(fa[2]), "+v"(fa[3]), "+v"(fb[0]), "+v"(fb[1]), "+v"(fb[2]), "+v"(fb[3])
                     : "n"(YOUNGER));
    };
    auto mma_half = [&](const u32x4_t (&fa)[MT], const u32x4_t (&fb)[NTW]) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) ChMma<T>::run(acc[mt][nt], fb[nt], fa[mt]);
    };
    // The same MFMA group with patch piece I of slice kcn riding in it: its GroupNorm + swish arithmetic fills the
    // vector-issue slots between the MFMAs (done as one block at the slice boundary it left the matrix pipe idle for
    // ~2000 cycles per slice: +8 % kernel time), then the piece goes to the other patch buffer.
    auto mma_half_stage = [&](const u32x4_t (&fa)[MT], const u32x4_t (&fb)[NTW], int kcn, auto i_tag) {
        constexpr int I = decltype(i_tag)::value;
        const u32x4_t v = gn_piece(areg[I], kcn, pixr[I]);
        mma_half(fa, fb);
        if constexpr (GN && ES == 2) {
#pragma unroll
            for (int k = 0; k < MT * NTW; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // 1 MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);     // 3 VALU
                __builtin_amdgcn_sched_group_barrier(0x400, 1, 0);     // 1 transcendental
            }
        }
        asm volatile("" ::"v"(v));                  // the piece is finished HERE, not inside the store's branch
        __builtin_amdgcn_sched_barrier(0);
        piece_write(v, kcn & 1, I, pixr[I]);
    };
    using K0 = std::integral_constant<int, 0>;
    using K1 = std::integral_constant<int, 1>;
    using Younger = std::integral_constant<int, MT + NTW>;
    using None = std::integral_constant<int, 0>;

    // ---- prologue: patch of slice 0, the first two weight tiles
    a_load(0);
    b_issue(0, std::integral_constant<int, 0>{});
    b_issue(0, std::integral_constant<int, 1>{});
    a_landed(std::integral_constant<int, 0>{}, areg);
    a_write(0, 0);
    ch_wait_barrier<0>();
    b_issue(0, std::integral_constant<int, 2>{});
    u32x4_t fa0[MT], fb0[NTW], fa1[MT], fb1[NTW];
    set_slice(0);
    read_half(K0{}, K0{}, fa0, fb0);

    // ---- main loop: slices of 128 bytes of channels, the nine taps unrolled.  Software pipeline over the 32-k halves
    // as in gather_gemm.hip: the barrier of step s+1 sits between the two MFMA groups of step s.  After that barrier
    // (every wave is done reading step s-1... and the tile of step s+1 has landed): stage the weight tile two steps
    // ahead; the NEXT slice's patch goes to registers at tap 1, its GroupNorm coefficients at tap 2, the loads are
    // waited for at tap 3, and at taps 3..8 one piece per step rides in the MFMA group into the other patch buffer
    // (whose last reader finished a slice ago; the barrier of the next slice's first step publishes the last piece).
    auto slice = [&](int kc, auto more_tag) {
        constexpr bool more = decltype(more_tag)::value;       // another slice follows (the last slice is its own instance)
        ch_static_for<0, 9>([&](auto j_tag) {
            constexpr int j = decltype(j_tag)::value;
            constexpr int nj = (j + 1) % 9;                  // tap of the next step
            using NJ = std::integral_constant<int, nj>;
            read_half(j_tag, K1{}, fa1, fb1);
            landed(Younger{}, fa0, fb0);
            __builtin_amdgcn_sched_barrier(0);
            mma_half(fa0, fb0);
            __builtin_amdgcn_sched_barrier(0);
            landed(None{}, fa1, fb1);
            if constexpr (j == 8 && !more) {                 // the very last step reads nothing ahead
                __builtin_amdgcn_sched_barrier(0);
                mma_half(fa1, fb1);
            } else {
                const int kn = j == 8 ? kc + 1 : kc;         // slice of the next step
                // does a slice follow the NEXT step's slice?  (compile-time inside a slice; at the hand-over to the
                // next slice it is the one run-time question of the body)
                const bool moren = j == 8 ? kn + 1 < nkc : more;
                // the next step's weight tile has landed; younger operations that may stay in flight: the tile behind
                // it (none behind the very last step), and at taps 2 and 3 the patch loads issued behind it
                if constexpr (nj == 8 && !more) ch_wait_barrier<0>();
                else if constexpr ((nj == 2 || nj == 3) && more) ch_wait_barrier<LOADS + CH_NA>();
                else ch_wait_barrier<LOADS>();
                // behind the barrier: the weight tile two steps ahead.  (Waves 4-7 issuing it behind the MFMA group
                // instead -- gather_gemm.hip's dephasing of the two waves of a SIMD -- measured 1-3 % slower here.)
                if constexpr (nj + 2 < 9) b_issue(kn, std::integral_constant<int, (nj + 2) % 9>{});
                else if (moren) b_issue(kn + 1, std::integral_constant<int, (nj + 2) % 9>{});
#if !(CH_ABL & 8)
                if constexpr (nj == 1 && more) a_load(kn + 1);
                if constexpr (nj == 2 && more) gn_coeffs(kn + 1);
                if constexpr (nj == 3 && more) a_landed(std::integral_constant<int, 2 * LOADS>{}, areg);   // behind them: the tiles of taps 4, 5
#endif
                if constexpr (nj == 0) set_slice(kn);
                read_half(NJ{}, K0{}, fa0, fb0);
                __builtin_amdgcn_sched_barrier(0);
#if !(CH_ABL & 8)
                if constexpr (nj >= 3 && nj < 3 + CH_NA && more) mma_half_stage(fa1, fb1, kn + 1, std::integral_constant<int, nj - 3>{});
                else
#endif
                    mma_half(fa1, fb1);
                __builtin_amdgcn_sched_barrier(0);
            }
        });
    };
    for (int kcl = 0; kcl + 1 < nkc; ++kcl) {
        int kc = kcl;
        asm volatile("" : "+s"(kc));        // opaque: no per-tap address induction variables across the slice loop
        slice(kc, std::true_type{});
    }
    slice(nkc - 1, std::false_type{});
    __syncthreads();

    // ---- epilogue, register phase: bias; lane owns pixel fi, channels 4*fg..+3 of each 16 x 16 tile
    unsigned char* tile = smem;
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
            unsigned char* dst = tile + row * PITCH + cb * ES;
            if constexpr (ES == 4) {
                *(float4*)dst = make_float4(acc[mt][nt][0] + bz[0], acc[mt][nt][1] + bz[1], acc[mt][nt][2] + bz[2],
                                            acc[mt][nt][3] + bz[3]);
            } else {
                uint2 pk;
                pk.x = (unsigned)f32_to_bf16(acc[mt][nt][0] + bz[0]) | ((unsigned)f32_to_bf16(acc[mt][nt][1] + bz[1]) << 16);
                pk.y = (unsigned)f32_to_bf16(acc[mt][nt][2] + bz[2]) | ((unsigned)f32_to_bf16(acc[mt][nt][3] + bz[3]) << 16);
                *(uint2*)dst = pk;
            }
        }
    }
    ch_lds_barrier();

    // ---- store phase: whole 16-B chunks of NHWC rows (+ residual).  The residual chunks of all eight passes are fetched
    // before the first is used and the stores go out back to back (a load -> add -> store chain per pass left one
    // round trip exposed per pass).
    const int sch = tid % CPR, rl = tid / CPR;
    const int scol = n0 + sch * EC;
    int orow_[ITERS];
    u32x4_t av[ITERS], vals[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        orow_[it] = s_orow[it * RL + rl];
        if (p.addend) av[it] = *(const u32x4_t*)(p.addend + ((size_t)(orow_[it] < 0 ? 0 : orow_[it]) * p.ldo + scol) * ES);
    }
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int row = it * RL + rl;
        u32x4_t val = *(const u32x4_t*)(tile + row * PITCH + sch * 16);
        if (p.addend) {
            T* ev = (T*)&val;
            const T* ae = (const T*)&av[it];
#pragma unroll
            for (int e = 0; e < EC; ++e) Elem<T>::store(ev + e, Elem<T>::load(ev + e) + Elem<T>::load(ae + e));
        }
        vals[it] = val;
        if (orow_[it] >= 0) *(u32x4_t*)(p.Out + ((size_t)orow_[it] * p.ldo + scol) * ES) = val;
    }
    if (p.stats) {
        // GroupNorm partial statistics of the STORED tile, per group of cg consecutive channels: the tile's mean and its sum
        // of squared deviations from that mean (gn_finish_tiles_k merges tiles with the parallel-variance formula).  Every
        // thread reduces its own rows (two passes over its registers: exact mean first), then the partials (count, sum,
        // M2) merge over the row lanes and over a group's channels by the same formula: as stable as torch's two passes.
        float sum[EC], m2[EC];
        int cnt = 0;
#pragma unroll
        for (int e = 0; e < EC; ++e) { sum[e] = 0.f; m2[e] = 0.f; }
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            if (orow_[it] >= 0) {
                ++cnt;
                const T* ev = (const T*)&vals[it];
#pragma unroll
                for (int e = 0; e < EC; ++e) sum[e] += Elem<T>::load(ev + e);
            }
        }
        const float inv = cnt ? 1.f / (float)cnt : 0.f;
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            if (orow_[it] >= 0) {
                const T* ev = (const T*)&vals[it];
#pragma unroll
                for (int e = 0; e < EC; ++e) { const float d = Elem<T>::load(ev + e) - sum[e] * inv; m2[e] += d * d; }
            }
        }
        // merge over the row lanes of a wave (bf16: lanes l, l+16, l+32, l+48 hold the same channels; f32: l, l+32) in
        // registers ...
        float fcnt = (float)cnt;
#pragma unroll
        for (int off = CPR; off <= 32; off <<= 1) {
            const float ocnt = __shfl_xor(fcnt, off, 64);
            const float ncnt = fcnt + ocnt;
            const float wgt = ncnt > 0.f ? fcnt * ocnt / ncnt : 0.f;
            const float ia = fcnt > 0.f ? 1.f / fcnt : 0.f, ib = ocnt > 0.f ? 1.f / ocnt : 0.f;
#pragma unroll
            for (int e = 0; e < EC; ++e) {
                const float osum = __shfl_xor(sum[e], off, 64), om2 = __shfl_xor(m2[e], off, 64);
                const float d = osum * ib - sum[e] * ia;
                m2[e] = m2[e] + om2 + d * d * wgt;
                sum[e] += osum;
            }
            fcnt = ncnt;
        }
        ch_lds_barrier();                                  // every thread has read its tile rows: the tile's LDS is scratch now
        // ... then over the eight waves and over a group's channels through LDS
        float* r_sum = (float*)smem;                       // [8 waves][BN]
        float* r_m2 = r_sum + 8 * CH_BN;                   // [8][BN]
        float* r_cnt = r_m2 + 8 * CH_BN;                   // [8]
        float* c_mean = r_cnt + 8;                         // [BN] channel means, then [BN] channel M2
        float* c_m2 = c_mean + CH_BN;
        static_assert(CPR == 16 || CPR == 32, "row-lane merge above: 16 (bf16) or 32 (f32) chunk columns per row");
        if (lane < CPR) {
#pragma unroll
            for (int e = 0; e < EC; ++e) {
                r_sum[w * CH_BN + sch * EC + e] = sum[e];
                r_m2[w * CH_BN + sch * EC + e] = m2[e];
            }
            if (lane == 0) r_cnt[w] = fcnt;
        }
        ch_lds_barrier();
        const int cg = p.stats_cg, ng = CH_BN / cg;
        if (tid < CH_BN) {
            float S = 0.f, N = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) { S += r_sum[k * CH_BN + tid]; N += r_cnt[k]; }
            const float mean = N > 0.f ? S / N : 0.f;
            float M2 = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float nk = r_cnt[k];
                const float d = nk > 0.f ? r_sum[k * CH_BN + tid] / nk - mean : 0.f;
                M2 += r_m2[k * CH_BN + tid] + nk * d * d;
            }
            c_mean[tid] = mean;
            c_m2[tid] = M2;
            if (tid == 0) c_m2[CH_BN] = N;
        }
        ch_lds_barrier();
        if (tid < ng) {
            const float N = c_m2[CH_BN];
            float gm = 0.f;
            for (int c = 0; c < cg; ++c) gm += c_mean[tid * cg + c];
            gm /= (float)cg;
            float gM2 = 0.f;
            for (int c = 0; c < cg; ++c) {
                const float d = c_mean[tid * cg + c] - gm;
                gM2 += c_m2[tid * cg + c] + N * d * d;
            }
            const int G = p.Nout / cg;
            ((float2*)p.stats)[(size_t)mtile * G + n0 / cg + tid] = make_float2(gm, gM2);
        }
    }
}

// Merge the per-tile (mean, M2) of rbvae_conv3x3_halo into per-(image, group) mean / rstd, and expand them with the
// affine parameters into the per-(image, channel) scale / shift the consuming convolution applies while it stages its
// input: y = x * scale + shift = (x - mean) * rstd * gamma + beta  (model.py:38-39).
__global__ __launch_bounds__(256) void gn_finish_tiles_k(const float2* __restrict__ part, int TH, int TW, int tiles_r, int tiles_c, int OH,
                                                         int OW, int cg, int G, float eps, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float* __restrict__ scale,
                                                         float* __restrict__ shift, float* __restrict__ mean_out,
                                                         float* __restrict__ rstd_out) {
    // one workgroup of four waves per (image, group): 1024-2048 tiles per image at 512 x 512 were 16-32 dependent rounds for
    // one wave (12-26 us per launch); the waves' partial sums merge in wave order (fixed: reproducible)
    __shared__ float s_part[4];
    const int n = blockIdx.x / G, g = blockIdx.x - n * G;
    const int nb = tiles_r * tiles_c;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float2* pp = part + (size_t)n * nb * G + g;
    auto count = [&](int b) {
        const int tr = b / tiles_c, tc = b - tr * tiles_c;
        return (float)(min(TH, OH - tr * TH) * min(TW, OW - tc * TW) * cg);
    };
    auto block_sum = [&](float v) {
        v = wave_sum(v);
        __syncthreads();                               // s_part free (second use)
        if (lane == 0) s_part[w] = v;
        __syncthreads();
        return ((s_part[0] + s_part[1]) + s_part[2]) + s_part[3];
    };
    const float total = (float)OH * (float)OW * (float)cg;
    float a = 0.f;
    for (int b = threadIdx.x; b < nb; b += 256) a += count(b) * pp[(size_t)b * G].x;
    const float m = block_sum(a) / total;
    float q = 0.f;
    for (int b = threadIdx.x; b < nb; b += 256) {
        const float2 pb = pp[(size_t)b * G];
        const float d = pb.x - m;
        q += pb.y + count(b) * d * d;
    }
    const float var = block_sum(q) / total;
    const float rs = rsqrtf(var + eps);
    if (threadIdx.x == 0 && mean_out) { mean_out[blockIdx.x] = m; rstd_out[blockIdx.x] = rs; }
    if ((int)threadIdx.x < cg) {
        const int c = g * cg + threadIdx.x, C = G * cg;
        const float sc = rs * gamma[c];
        scale[(size_t)n * C + c] = sc;
        shift[(size_t)n * C + c] = beta[c] - m * sc;
    }
}

// scale / shift from existing mean / rstd (statistics from rbvae_groupnorm_swish_ws's kernels)
__global__ void gn_affine_k(const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ gamma,
                            const float* __restrict__ beta, float* __restrict__ scale, float* __restrict__ shift, int N,
                            int C, int groups) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * C) return;
    const int n = i / C, c = i - n * C, sg = n * groups + c / (C / groups);
    const float sc = rstd[sg] * gamma[c];
    scale[i] = sc;
    shift[i] = beta[c] - mean[sg] * sc;
}

template <typename T, int RING, bool GN>
static int launch_ch(const ChArgs& a, hipStream_t st) {
    constexpr int ES = sizeof(T);
    const size_t lds = (size_t)ch_lds_main<T, RING>() + CH_BM * 4 + CH_NSLOT_PAD * 4 + (a.gn_scale ? (size_t)2 * a.Kc * 4 : 0) + 16;
    if (lds > 160 * 1024) return fail(RBVAE_E_UNSUPPORTED, "conv3x3_halo: %zu bytes of LDS", lds);
    (void)ES;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)conv_halo_k<T, RING, GN>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL((conv_halo_k<T, RING, GN>), dim3(a.total), dim3(512), lds, st, a);
    RBVAE_CHECK_LAUNCH("conv3x3_halo");
    return RBVAE_OK;
}

}  // namespace rbvae

using namespace rbvae;

// which kernel the bf16 dispatch takes: 2 auto (product), 1 conv_halo_k always, 0 the persistent kernel wherever it covers.
// Written only by librbvae_dbg's rbvae_dbg_conv_halo_variant (include/rbvae_dbg.h: bit-identity tests, A/B timing).
namespace rbvae { int ch_variant = 2; }

extern "C" int rbvae_conv3x3_halo_ok(int dtype, int IH, int IW, int OH, int OW, int Kc, int Nout) {
    const int KE = dtype == RBVAE_F32 ? 32 : 64;
    if (dtype != RBVAE_F32 && dtype != RBVAE_BF16) return 0;
    if (Kc <= 0 || Kc % KE || Nout <= 0 || Nout % CH_BN) return 0;
    if (Kc > 1024) return 0;                         // scale / shift table in LDS
    if (OH < 8 || OW < CH_T || IH < 1 || IW < 1) return 0;   // narrower images: rbvae_gather_gemm
    return 1;
}

extern "C" int rbvae_conv3x3_halo(int dtype, const void* A, const void* W, void* Out, const float* bias, const void* addend,
                                  const void* zero_page, const float* gn_scale, const float* gn_shift, int gn_swish,
                                  float* stats_part, int stats_cg, int Nimg, int IH, int IW, int OH, int OW, int pad_h,
                                  int pad_w, int Kc, int Nout, int lda, int ldo, void* stream) {
    RBVAE_CHECK_ARG(A && W && Out && zero_page, "conv3x3_halo: null pointer");
    RBVAE_CHECK_ARG(rbvae_conv3x3_halo_ok(dtype, IH, IW, OH, OW, Kc, Nout),
                    "conv3x3_halo: shape not covered (dtype %d, %dx%d -> %dx%d, Kc %d, Nout %d)", dtype, IH, IW, OH, OW, Kc, Nout);
    const int ES = dtype == RBVAE_F32 ? 4 : 2;
    RBVAE_CHECK_ARG(lda >= Kc && (lda * ES) % 16 == 0 && ldo >= Nout && (ldo * ES) % 16 == 0,
                    "conv3x3_halo: leading dimensions lda=%d ldo=%d", lda, ldo);
    RBVAE_CHECK_ARG(Nimg > 0 && (long)Nimg * IH * IW < (1l << 30) && (long)Nimg * OH * OW < (1l << 30),
                    "conv3x3_halo: more than 2^30 pixel rows");
    RBVAE_CHECK_ARG(((uintptr_t)A | (uintptr_t)W | (uintptr_t)Out | (uintptr_t)zero_page | (uintptr_t)addend | (uintptr_t)bias) % 16 == 0,
                    "conv3x3_halo: pointers must be 16-byte aligned");
    RBVAE_CHECK_ARG((gn_scale == nullptr) == (gn_shift == nullptr), "conv3x3_halo: gn_scale and gn_shift go together");
    RBVAE_CHECK_ARG(!stats_part || (stats_cg > 0 && CH_BN % stats_cg == 0 && Nout % stats_cg == 0),
                    "conv3x3_halo: stats_cg=%d must divide %d", stats_cg, CH_BN);
    RBVAE_CHECK_ARG(pad_h >= 0 && pad_h <= 2 && pad_w >= 0 && pad_w <= 2, "conv3x3_halo: pad %d %d", pad_h, pad_w);
    ChArgs a;
    a.A = (const unsigned char*)A; a.W = (const unsigned char*)W; a.Out = (unsigned char*)Out; a.bias = bias;
    a.addend = (const unsigned char*)addend; a.zero = (const unsigned char*)zero_page;
    a.gn_scale = gn_scale; a.gn_shift = gn_shift; a.gn_swish = gn_swish; a.stats = stats_part; a.stats_cg = stats_cg;
    a.Nimg = Nimg; a.IH = IH; a.IW = IW; a.OH = OH; a.OW = OW; a.dh0 = -pad_h; a.dw0 = -pad_w;
    a.Kc = Kc; a.Nout = Nout; a.lda = lda; a.ldo = ldo;
    a.tiles_r = cdiv(OH, CH_T); a.tiles_c = cdiv(OW, CH_T); a.ntn = Nout / CH_BN;
    const long total = (long)Nimg * a.tiles_r * a.tiles_c * a.ntn;
    RBVAE_CHECK_ARG(total < (1l << 30), "conv3x3_halo: too many tiles");
    a.total = (int)total;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == RBVAE_F32) return gn_scale ? launch_ch<float, 3, true>(a, st) : launch_ch<float, 3, false>(a, st);
    // bf16: the persistent kernel with producer / MFMA wave roles (conv_halo_ws.hip), bit-identical to conv_halo_k:
    // 2 (auto): the persistent kernel where it measured ahead -- at least two tiles per CU to walk (the per-tile prologue and
    // epilogue it overlaps are a third of a two-slice tile; with one tile per workgroup there is nothing to overlap)
    const bool ws_auto = a.total >= 512;
    if ((ch_variant == 0 || (ch_variant == 2 && ws_auto)) && ch_ws_covers(a)) return launch_ch_ws(a, st);
    return gn_scale ? launch_ch<bf16_t, 3, true>(a, st) : launch_ch<bf16_t, 3, false>(a, st);
}

extern "C" size_t rbvae_conv3x3_halo_stats_floats(int Nimg, int OH, int OW, int Nout, int cg) {
    return (size_t)2 * Nimg * cdiv(OH, CH_T) * cdiv(OW, CH_T) * (Nout / cg);
}

extern "C" int rbvae_gn_finish_tiles(const float* stats_part, const float* gamma, const float* beta, float* scale,
                                     float* shift, float* mean_out, float* rstd_out, int Nimg, int OH, int OW, int C,
                                     int groups, float eps, int tile_h, int tile_w, void* stream) {
    RBVAE_CHECK_ARG(stats_part && gamma && beta && scale && shift, "gn_finish_tiles: null pointer");
    RBVAE_CHECK_ARG(groups > 0 && C % groups == 0 && C / groups <= 64, "gn_finish_tiles: C=%d groups=%d", C, groups);
    RBVAE_CHECK_ARG((mean_out == nullptr) == (rstd_out == nullptr), "gn_finish_tiles: mean_out and rstd_out go together");
    RBVAE_CHECK_ARG(tile_h > 0 && tile_w > 0, "gn_finish_tiles: tile %d x %d (16 x 16: rbvae_conv3x3_halo, 8 x 16: rbvae_conv_in)", tile_h, tile_w);
    hipLaunchKernelGGL(gn_finish_tiles_k, dim3(Nimg * groups), dim3(256), 0, (hipStream_t)stream, (const float2*)stats_part,
                       tile_h, tile_w, cdiv(OH, tile_h), cdiv(OW, tile_w), OH, OW, C / groups, groups, eps, gamma, beta, scale, shift,
                       mean_out, rstd_out);
    RBVAE_CHECK_LAUNCH("gn_finish_tiles");
    return RBVAE_OK;
}

extern "C" int rbvae_gn_affine(const float* mean, const float* rstd, const float* gamma, const float* beta, float* scale,
                               float* shift, int N, int C, int groups, void* stream) {
    RBVAE_CHECK_ARG(mean && rstd && gamma && beta && scale && shift && N > 0 && C > 0 && groups > 0 && C % groups == 0,
                    "gn_affine: bad arguments");
    hipLaunchKernelGGL(gn_affine_k, dim3(cdiv((long)N * C, 256)), dim3(256), 0, (hipStream_t)stream, mean, rstd, gamma, beta,
                       scale, shift, N, C, groups);
    RBVAE_CHECK_LAUNCH("gn_affine");
    return RBVAE_OK;
}
