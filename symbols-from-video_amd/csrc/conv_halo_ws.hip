// Halo-resident 3x3 stride-1 convolution, bf16, as PERSISTENT workgroups with wave roles: the same sums, the same LDS images
// and the same epilogue arithmetic as conv_halo_k (conv_halo.hip; the LDM / Stable-Diffusion VAE encoder's ResnetBlock
// convolutions, src/stable-diffusion/ldm/modules/diffusionmodules/model.py:82-141), bit for bit -- rearranged around two
// measurements of round 4 (rbvae_wgrad3x3s2_row): an in-order wave that sits in the vector-memory issue (an LDS-DMA piece
// holds it ~70 cycles) or in GroupNorm's exp / rcp arithmetic feeds no MFMA, and a tile's prologue + epilogue (5.5 us) is
// a third of a 128-channel tile.
//
//   * 12 waves: waves 0-7 only read fragments and multiply (4 x 2 waves, 64 x 64 outputs each, as conv_halo_k); waves
//     8-11 are PRODUCERS: they stream the weight tap tiles by LDS-DMA through a ring of FOUR, and stage the next slice's
//     18 x 18 input patch: coalesced 128-byte pixel rows into registers (buffer loads: a padding pixel is an offset beyond
//     the descriptor and returns zeros), the producing GroupNorm + swish applied in flight, transposing ds_write_b128 into
//     the chunk-major image [8 chunks][336 slots] of the other patch buffer.
//   * One workgroup per CU walks its tiles (an XCD's contiguous tile range dealt round-robin to its workgroups: the
//     channel tiles of one pixel tile and neighbouring pixel tiles run on one XCD at the same time): while the MFMA waves
//     run a tile's epilogue -- through the patch buffer of the tile's LAST slice, 128 rows at a time -- the producers already
//     have the next tile's first patch in the other buffer and its first weight tiles in the ring.
//   * Unit = (slice of 64 channels, tap).  Barrier per unit; at barrier u every fragment read of unit u - 1 has landed
//     (the MFMA waves wait for them in front of it), so its ring slot takes tile u + 3 right behind the barrier.  The patch
//     of slice g + 1 is written during slice g's taps 0-7 (buffer (g + 1) & 1, last read by slice g - 1), its loads are
//     issued a slice ahead in two batches (tap 3: pieces 0-5 + the GroupNorm coefficients; tap 8: pieces 6-10) as the
//     registers of the pieces before them come free.
//   * The patch pieces ROLL: a piece's register is re-loaded for the slice after next right behind the instruction that
//     consumed it, so every patch load has a whole slice (nine units) to land.  (Loaded in two batches four to six units ahead,
//     the loads' latency was what the patch staging cost: with out-of-range loads the kernel ran 15 % faster, with the
//     processing alone removed not at all.)
//   * Every producer wait is a counted s_waitcnt vmcnt(N) against the PERIODIC op stream of a slice,
//       T0 C L0 L1 | T1 L2 L3 | T2 L4 L5 | T3 L6 | T4 L7 | T5 L8 | T6 L9 | T7 L10 | T8
//     (Tj = the four tile pieces of tap j's body, C = the GroupNorm coefficients, Lk = the re-load of piece k): a piece is
//     awaited with vmcnt(P - 1), P = the operations of a period.  Where a real operation does not exist (the prologue, the
//     last slices of the walk) an out-of-range load of the same count takes its place, so one loop body serves the whole walk.
#include "conv_halo.h"

// timing ablations (wrong results on purpose; -DRBVAE_ABLATION builds only, tools/ab_variants.sh): 1 no weight LDS-DMA (dummy
// loads keep the counts), 2 no patch pieces processed / written, 4 patch loads out of range (no traffic), 8 no fragment reads,
// 16 no epilogue LDS / global traffic, 32 no unit barriers
#ifndef CW_ABL
#define CW_ABL 0
#endif
#if CW_ABL && !defined(RBVAE_ABLATION)
#error "CW_ABL builds give wrong results: define RBVAE_ABLATION to confirm"
#endif

namespace rbvae {

constexpr int CW_RING = 4;
constexpr int CW_NP = 11;                                    // patch pieces per producer thread and slice: slots slot0 + 32 i
constexpr int CW_MAXN = 512;                                 // output channels whose bias the workgroup keeps in LDS
constexpr int CW_LDS = 2 * CH_ABUF + CW_RING * CH_BBYTES + CW_MAXN * 4;    // 153 856 B: patches, weight ring, the bias
constexpr int CW_PITCH = CH_BN * 2 + 16;                     // epilogue tile row: 128 bf16 + 16 B
static_assert(128 * CW_PITCH <= CH_ABUF, "half an output tile fits one patch buffer");
static_assert((2 * 8 * CH_BN + 8 + 2 * CH_BN + 1) * 4 <= CH_ABUF, "the statistics scratch fits one patch buffer");
static_assert(CW_LDS <= 160 * 1024, "LDS carve");
// patch pieces processed in the body of tap j (for the NEXT slice)
__host__ __device__ constexpr int cw_pp(int j) { return j < 3 ? 2 : j < 8 ? 1 : 0; }
__host__ __device__ constexpr int cw_pp_before(int j) { int s = 0; for (int k = 0; k < j; ++k) s += cw_pp(k); return s; }
static_assert(cw_pp_before(9) == CW_NP, "piece schedule");
// operations behind tile u's at the top of unit j: two tiles + what the three bodies before issued behind their tile pieces
__host__ __device__ constexpr int cw_ntop(int j, int nc) {
    int n = 8;
    for (int d = 1; d <= 3; ++d) { const int b = (j + 9 - d) % 9; n += cw_pp(b) + (b == 0 ? nc : 0); }
    return n;
}

template <bool GN>
__global__ __launch_bounds__(768, 1) void conv_halo_ws_k(const ChArgs p) {
    constexpr int MT = 4, NTW = 4;
    constexpr int NC = GN ? 4 : 0;                           // coefficient loads per slice
    constexpr int PERIOD = 36 + CW_NP + NC;                  // vector-memory operations of a producer wave per slice
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;

    // ---- this workgroup's tiles: XCD x owns a contiguous range of items (the channel tiles of a pixel tile adjacent), dealt
    // round-robin to the workgroups that run on it (blockIdx % 8 == x)
    const int G = gridDim.x, xcd = blockIdx.x & 7;
    const int q = p.total >> 3, r = p.total & 7;
    const int x0 = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    const int cnt = q + (xcd < r ? 1 : 0);
    const int wx = (G - xcd + 7) >> 3;
    const int s0 = blockIdx.x >> 3;
    const int ntiles = s0 < cnt ? (cnt - s0 + wx - 1) / wx : 0;
    const int nkc = p.Kc >> 6;
    const int per_img = p.tiles_r * p.tiles_c;
    if (ntiles == 0) return;
    const int nslices = ntiles * nkc;
    auto decode = [&](int k, int& n, int& r0, int& c0, int& n0, int& mtile) {
        // (opaque divisors: the compiler otherwise keeps their float reciprocals in vector registers across the tile loop,
        // spills them, and the reload at the top of a tile waits for the stores of the tile before)
        int ntn = p.ntn, pim = per_img, tcn = p.tiles_c;
        asm volatile("" : "+s"(ntn), "+s"(pim), "+s"(tcn));
        const int item = x0 + s0 + k * wx;
        mtile = item / ntn;
        const int ntile = item - mtile * ntn;
        n = mtile / pim;
        const int trc = mtile - n * pim;
        const int tr = trc / tcn, tc = trc - tr * tcn;
        r0 = tr * CH_T; c0 = tc * CH_T; n0 = ntile * CH_BN;
    };
    const int nb_epi = 4 + (p.stats ? 2 : 0);                // barriers of a tile's epilogue

    if (w >= 8) {
        // =========================== producer waves ===========================
        const int pw = w - 8, ptid = tid - 512;
        const int chunk = ptid & 7, slot0 = ptid >> 3;       // this thread's pieces: (slot slot0 + 32 i, chunk)
        int prc[CW_NP];                                      // patch row | column << 8 of piece i, -1 beyond the patch
#pragma unroll
        for (int i = 0; i < CW_NP; ++i) {
            const int slot = slot0 + 32 * i, pr = slot / CH_PW, pc = slot - pr * CH_PW;
            prc[i] = slot < CH_NSLOT ? (pr | (pc << 8)) : -1;
        }
        const bool last_ok = slot0 + 32 * (CW_NP - 1) < CH_NSLOT_PAD;     // piece 10's slot exists in the plane
        const unsigned long long a_addr = (unsigned long long)p.A, s_addr = (unsigned long long)p.gn_scale,
                                 h_addr = (unsigned long long)p.gn_shift;
        const u32x4_t rsA = {(unsigned)a_addr, (unsigned)(a_addr >> 32) & 0xffffu,
                             (unsigned)(p.Nimg * p.IH * p.IW * p.lda * 2), 0x00020000u};
        const u32x4_t rsS = {(unsigned)s_addr, (unsigned)(s_addr >> 32) & 0xffffu, (unsigned)(p.Nimg * p.Kc * 4), 0x00020000u};
        const u32x4_t rsH = {(unsigned)h_addr, (unsigned)(h_addr >> 32) & 0xffffu, (unsigned)(p.Nimg * p.Kc * 4), 0x00020000u};
        const auto rsrcW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, p.Nout * 9 * p.Kc * 2, 0x00020000);
        auto* lds = (__attribute__((address_space(3))) unsigned char*)smem;
        int w_off[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (pw * 4 + i) * 8 + (lane >> 3);
            w_off[i] = row * 9 * p.Kc * 2 + (((lane & 7) ^ ((row >> 1) & 7)) * 16);
        }
        const int oob = (int)0x80000000u;

        // slice g of the walk -> weight offset (tile, slice), patch geometry
        struct SliceD { int valid, wsoff, n, r0, c0, kc; };
        auto desc = [&](int g) {
            SliceD d;
            d.valid = g < nslices;
            const int gg = d.valid ? g : 0;
            const int k = gg / nkc;
            d.kc = gg - k * nkc;
            int n0, mtile;
            decode(k, d.n, d.r0, d.c0, n0, mtile);
            d.wsoff = (n0 * 9 * p.Kc + d.kc * 64) * 2;
            return d;
        };
        int pixoff[CW_NP];                                   // byte offset of piece i's 16 bytes in A (slice offset apart), oob = zeros
        unsigned mask_ld = 0, mask_proc = 0;                 // bit i: piece i is a real pixel (of the slice loaded / processed)
        int coff = oob, soffA = 0, soffC = 0;                // coefficient lane offset, scalar offsets of the slice being loaded
        auto set_patch = [&](const SliceD& d) {
            mask_ld = 0;
#pragma unroll
            for (int i = 0; i < CW_NP; ++i) {
                const int pr = prc[i] & 255, pc = prc[i] >> 8;
                const int ih = d.r0 + pr + p.dh0, iw = d.c0 + pc + p.dw0;
                const bool ok = d.valid && prc[i] >= 0 && ih >= 0 && ih < p.IH && iw >= 0 && iw < p.IW;
#if CW_ABL & 4
                pixoff[i] = oob;
#else
                pixoff[i] = ok ? ((d.n * p.IH + ih) * p.IW + iw) * p.lda * 2 + chunk * 16 : oob;
#endif
                mask_ld |= ok ? (1u << i) : 0u;
            }
            coff = d.valid ? chunk * 32 : oob;
            soffA = __builtin_amdgcn_readfirstlane(d.kc * 128);
            soffC = __builtin_amdgcn_readfirstlane((d.n * p.Kc + d.kc * 64) * 4);
            // The scalar offsets are made HERE and pinned (opaque read-write operands), wait states behind them: left to the
            // compiler they were produced by the instruction in front of the first asm load (s_mul_i32 / s_add_i32 / a
            // v_readfirstlane), whose soffset operand then still read the PREVIOUS slice's value -- the first coefficient load
            // of a slice fetched the slice before's scales.  The compiler pads hazards only between instructions it can see.
            asm volatile("s_nop 7" : "+s"(soffA), "+s"(soffC) : : "memory");
        };
        u32x4_t areg[CW_NP], ncoef[4];
        float gsc[8], gsh[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { gsc[e] = 1.f; gsh[e] = 0.f; }
        // (asm operands of a GENERIC lambda cannot name captured variables: registers and offsets travel as arguments)
        auto load_piece = [&](u32x4_t& dst, int off) {
            asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(dst) : "v"(off), "s"(rsA), "s"(soffA) : "memory");
        };
        auto load_coefs = [&]() {
            if constexpr (GN) {
                asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(ncoef[0]) : "v"(coff), "s"(rsS), "s"(soffC) : "memory");
                asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:16" : "=v"(ncoef[1]) : "v"(coff), "s"(rsS), "s"(soffC) : "memory");
                asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(ncoef[2]) : "v"(coff), "s"(rsH), "s"(soffC) : "memory");
                asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:16" : "=v"(ncoef[3]) : "v"(coff), "s"(rsH), "s"(soffC) : "memory");
            }
        };
        // N vector-memory operations may still be outstanding behind the awaited ones
        auto piece_landed = [](auto n_tag, u32x4_t& r) { asm volatile("s_waitcnt vmcnt(%1)" : "+v"(r) : "n"(decltype(n_tag)::value)); };
        auto coefs_landed = [](auto n_tag, u32x4_t (&nc)[4]) {
            asm volatile("s_waitcnt vmcnt(%4)" : "+v"(nc[0]), "+v"(nc[1]), "+v"(nc[2]), "+v"(nc[3]) : "n"(decltype(n_tag)::value));
        };
        auto take_coefs = [&](auto n_tag) {                  // the coefficients of the slice processed next: landed, into gsc / gsh
            if constexpr (GN) {
                coefs_landed(n_tag, ncoef);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    gsc[e] = __uint_as_float(ncoef[0][e]); gsc[4 + e] = __uint_as_float(ncoef[1][e]);
                    gsh[e] = __uint_as_float(ncoef[2][e]); gsh[4 + e] = __uint_as_float(ncoef[3][e]);
                }
            }
        };
        // four dummy operations in the place of a tile that does not exist (out of range: they return zeros).  Their
        // destination is ONE register reserved for the whole walk: a load lands whenever it lands, and a register the compiler
        // had meanwhile given to something else (the GroupNorm coefficients, once) would be zeroed under it.
        unsigned junk = 0;
        auto dummy4 = [&]() {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "+v"(junk) : "v"(oob), "s"(rsA) : "memory");
        };
        auto tile = [&](int soff, int slot) {
#if CW_ABL & 1
            dummy4();
            return;
#endif
            auto* dst = lds + 2 * CH_ABUF + slot * CH_BBYTES + (pw * 4) * 1024;
#pragma unroll
            for (int i = 0; i < 4; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcW, dst + i * 1024, 16, w_off[i], soff, 0, 0);
        };
        // piece K of the slice being processed: GroupNorm (+ swish) as conv_halo_k's gn_piece, then into patch buffer `buf`
        auto piece = [&](auto k_tag, int buf) {
            constexpr int K = decltype(k_tag)::value;
#if CW_ABL & 2
            return;
#endif
            u32x4_t t = areg[K];
            if constexpr (GN) {
                // conv_halo_k's arithmetic element for element (fma; x * rcp(1 + exp2(-log2e x)); round to bf16), written on
                // PAIRS so that it compiles to packed f32 instructions: the producers are the critical path of a GroupNorm layer
                // (without the patch work it runs 26 % faster), and what they contend for with the MFMA waves of their SIMD is
                // the vector issue port -- the instruction COUNT (sharing one reciprocal between four elements, fewer
                // quarter-rate but more plain instructions, made it slower).
                typedef float v2f __attribute__((ext_vector_type(2)));
                typedef __bf16 v2b __attribute__((ext_vector_type(2)));
                const u32x4_t v = t;
                v2f x[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const v2f in = {__uint_as_float(v[q] << 16), __uint_as_float(v[q] & 0xffff0000u)};
                    const v2f sc = {gsc[2 * q], gsc[2 * q + 1]}, sh = {gsh[2 * q], gsh[2 * q + 1]};
                    x[q] = __builtin_elementwise_fma(in, sc, sh);
                }
                if (p.gn_swish) {                                   // a branch, not eight selects (uniform)
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        v2f e = x[q] * -1.44269504f;
                        e.x = __builtin_amdgcn_exp2f(e.x); e.y = __builtin_amdgcn_exp2f(e.y);
                        v2f d = e + 1.0f;
                        d.x = __builtin_amdgcn_rcpf(d.x); d.y = __builtin_amdgcn_rcpf(d.y);
                        x[q] = x[q] * d;
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const v2b r = __builtin_convertvector(x[q], v2b);      // v_cvt_pk_bf16_f32: round to nearest even, both halves
                    t[q] = __builtin_bit_cast(unsigned, r);
                }
                const bool real = (mask_proc >> K) & 1u;
#pragma unroll
                for (int c = 0; c < 4; ++c) t[c] = real ? t[c] : v[c];
            }
            const unsigned dst = lds0 + (unsigned)buf * CH_ABUF + ch_plane_off(chunk) + (unsigned)(slot0 + 32 * K) * 16;
            if (K < CW_NP - 1 || last_ok) asm volatile("ds_write_b128 %0, %1" ::"v"(dst), "v"(t) : "memory");
        };
        using I0 = std::integral_constant<int, 0>;
        using PieceWait = std::integral_constant<int, PERIOD - 1>;
        using CoefWait = std::integral_constant<int, CW_NP + 32>;       // behind C: the eleven re-loads and T1..T8

        // ---- prologue: slice 0's patch (whole), then one period of the op stream as "slice -1" would have issued it: slice 1's
        // coefficients and pieces, out-of-range loads for the tiles that do not exist, the tiles of units 0, 1, 2
        SliceD d0 = desc(0);
        set_patch(d0);
        load_coefs();
        ch_static_for<0, CW_NP>([&](auto k_tag) { load_piece(areg[decltype(k_tag)::value], pixoff[decltype(k_tag)::value]); });
        take_coefs(std::integral_constant<int, CW_NP>{});
        ch_static_for<0, CW_NP>([&](auto k_tag) { piece_landed(I0{}, areg[decltype(k_tag)::value]); });
        mask_proc = mask_ld;
        ch_static_for<0, CW_NP>([&](auto k_tag) { piece(k_tag, 0); });
        {
            SliceD d1 = desc(1);
            set_patch(d1);
            ch_static_for<0, 9>([&](auto j_tag) {
                constexpr int j = decltype(j_tag)::value;
                if constexpr (j < 6) dummy4();
                else tile(d0.wsoff + (j - 6) * p.Kc * 2, j - 6);
                if constexpr (j == 0) load_coefs();
                ch_static_for<cw_pp_before(j), cw_pp_before(j) + cw_pp(j)>([&](auto k_tag) {
                    load_piece(areg[decltype(k_tag)::value], pixoff[decltype(k_tag)::value]);
                });
            });
            take_coefs(CoefWait{});
        }

        // ---- the walk
        for (int g = 0; g < nslices; ++g) {
            const SliceD dc = desc(g), dn = desc(g + 1), dl = desc(g + 2);
            const int v1 = dn.valid;
            const int nbuf = (g + 1) & 1;
            ch_static_for<0, 9>([&](auto j_tag) {
                constexpr int j = decltype(j_tag)::value;
                // tile u landed, this wave's patch writes done
                constexpr int NTOP = cw_ntop(j, NC);
#if CW_ABL & 32
                asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)" ::"n"(NTOP) : "memory");
#else
                asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::"n"(NTOP) : "memory");
#endif
                // T: tile u + 3 into the slot unit u - 1 has left
                const int slot = (g + j + 3) & (CW_RING - 1);
                if constexpr (j + 3 < 9) tile(dc.wsoff + (j + 3) * p.Kc * 2, slot);
                else if (v1) tile(dn.wsoff + (j + 3 - 9) * p.Kc * 2, slot);
                else dummy4();
                if constexpr (j == 0) {
                    // the pieces in the registers are slice g + 1's (mask of a slice ago); from here on they are re-loaded for g + 2
                    mask_proc = mask_ld;
                    set_patch(dl);
                    load_coefs();                            // C
                }
                // this body's pieces: landed (a period ago) -> GroupNorm -> the other patch buffer -> re-load for slice g + 2
                ch_static_for<cw_pp_before(j), cw_pp_before(j) + cw_pp(j)>([&](auto k_tag) {
                    constexpr int K = decltype(k_tag)::value;
                    piece_landed(PieceWait{}, areg[K]);
                    if (v1) piece(k_tag, nbuf);
                    load_piece(areg[K], pixoff[K]);
                });
                if constexpr (j == 8) take_coefs(CoefWait{});
            });
            if (dc.kc == nkc - 1 && g + 1 < nslices) {
                for (int b = 0; b < nb_epi; ++b) asm volatile("s_barrier" ::: "memory");
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(junk));     // (the reserved register lives to the end of the walk)
        return;
    }

    // =========================== MFMA waves ===========================
    const int fi = lane & 15, fg = lane >> 4;
    const int wr = w >> 1, wc = w & 1;
    unsigned abase[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
        abase[mt] = lds0 + (unsigned)fg * CH_PLANE + (unsigned)(fg >> 1) * 32 + (unsigned)((wr * MT + mt) * CH_PW + fi) * 16;
    const int fsw = (fi >> 1) & 7;
    const unsigned offB0 = lds0 + 2 * CH_ABUF + (unsigned)(wc * NTW * 16 + fi) * 128 + (unsigned)(((0 + fg) ^ fsw) * 16);
    const unsigned offB1 = lds0 + 2 * CH_ABUF + (unsigned)(wc * NTW * 16 + fi) * 128 + (unsigned)(((4 + fg) ^ fsw) * 16);

    f32x4_t acc[MT][NTW];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) acc[mt][nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    unsigned ta[MT];
    auto read_half = [&](auto j_tag, auto kk_tag, unsigned rs, u32x4_t (&fa)[MT], u32x4_t (&fb)[NTW]) {
        constexpr int j = decltype(j_tag)::value, kk = decltype(kk_tag)::value;
        constexpr int aoff = kk * CH_KKOFF + ((j / 3) * CH_PW + (j % 3)) * 16;
#if CW_ABL & 8
        return;
#endif
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[mt]) : "v"(ta[mt]), "n"(aoff));
        const unsigned ab_ = (kk == 0 ? offB0 : offB1) + rs;
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[nt]) : "v"(ab_), "n"(nt * 2048));
    };
    auto landed = [&](u32x4_t (&fa)[MT], u32x4_t (&fb)[NTW]) {
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(fa[0]), "+v"(fa[1]), "+v"(fa[2]), "+v"(fa[3]), "+v"(fb[0]), "+v"(fb[1]), "+v"(fb[2]), "+v"(fb[3]));
    };
    auto mma_half = [&](const u32x4_t (&fa)[MT], const u32x4_t (&fb)[NTW]) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) ChMma<bf16_t>::run(acc[mt][nt], fb[nt], fa[mt]);
    };
    using K0 = std::integral_constant<int, 0>;
    using K1 = std::integral_constant<int, 1>;
    u32x4_t xa[MT], xb[NTW], ya[MT], yb[NTW];
#pragma unroll
    for (int i = 0; i < MT; ++i) ya[i] = u32x4_t{0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < NTW; ++i) yb[i] = u32x4_t{0, 0, 0, 0};

    // Unit (slice g, tap j): barrier (its weight tile and, at tap 0, its patch are in LDS; every wave's reads of the unit
    // before have landed) -> first half's reads -> the second half of the unit before -> second half's reads -> first half
    auto slice = [&](int g) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) ta[mt] = abase[mt] + (unsigned)((g & 1) * CH_ABUF);
        ch_static_for<0, 9>([&](auto j_tag) {
            constexpr int j = decltype(j_tag)::value;
            const unsigned rs = (unsigned)(((g + j) & (CW_RING - 1)) * CH_BBYTES);
#if !(CW_ABL & 32)
            asm volatile("s_barrier" ::: "memory");
#endif
            read_half(j_tag, K0{}, rs, xa, xb);
            __builtin_amdgcn_sched_barrier(0);
            mma_half(ya, yb);
            __builtin_amdgcn_sched_barrier(0);
            landed(xa, xb);
            read_half(j_tag, K1{}, rs, ya, yb);
            __builtin_amdgcn_sched_barrier(0);
            mma_half(xa, xb);
            __builtin_amdgcn_sched_barrier(0);
            landed(ya, yb);
        });
    };

    // The store phase goes through buffer descriptors: a row outside the image is an offset beyond the descriptor (the
    // hardware drops the store, a load returns zeros), so the epilogue has no branch around a memory instruction and the
    // compiler can COUNT its waits (the residual loads of round 1 sit behind round 0's stores: with branches in between it
    // waited vmcnt(0), a full store round trip in mid-epilogue).  For the same reason nothing here may spill: a scratch reload
    // is a vector-memory load and waits for every store in flight (the first version did, three times per tile).
    const auto rsOut = __builtin_amdgcn_make_buffer_rsrc((void*)p.Out, 0, p.Nimg * p.OH * p.OW * p.ldo * 2, 0x00020000);
    const auto rsAdd = __builtin_amdgcn_make_buffer_rsrc((void*)(p.addend ? p.addend : p.Out), 0, p.Nimg * p.OH * p.OW * p.ldo * 2, 0x00020000);
    // the layer's bias, once per workgroup (loaded in every tile's epilogue it was a global round trip per tile; held in a
    // register across the slice loop it spilled)
    float* s_bias = (float*)(smem + 2 * CH_ABUF + CW_RING * CH_BBYTES);
    for (int i = threadIdx.x; i < p.Nout; i += 512) s_bias[i] = p.bias ? p.bias[i] : 0.f;
    int g = 0;
    for (int k = 0; k < ntiles; ++k) {
        int n, r0, c0, n0, mtile;
        decode(k, n, r0, c0, n0, mtile);
        // (the second half of a tile's last unit is multiplied at the next tile's first tap... no: the epilogue needs it, so
        // it runs here, and the first tap of a tile multiplies ZERO fragments: a branch between that tap's reads and their wait
        // makes the compiler copy in-flight registers, a second instance of the slice body makes it rename the accumulators)
        for (int kc = 0; kc < nkc; ++kc, ++g) slice(g);
        mma_half(ya, yb);
        // (lane constants from an opaque copy of the thread index: hoisted out of the tile loop they cost the slice loop
        // registers and get spilled)
        int te = threadIdx.x;
        asm volatile("" : "+v"(te));
        const int lane = te & 63, tid = te;
        const int fi = lane & 15, fg = lane >> 4, wr = (te >> 6) >> 1, wc = (te >> 6) & 1;
        const int sch = te & 15, rl = te >> 4;                     // store phase: 16 chunks per row x 32 row lanes
        // ---- epilogue through the patch buffer of the tile's last slice, 128 rows at a time (conv_halo_k's arithmetic)
        unsigned char* tile = smem + ((g - 1) & 1) * CH_ABUF;
        ch_lds_barrier();
        // + bias, to bf16: the accumulators are free again (lane: pixel fi, channels 4 fg .. + 3 of each 16 x 16 block)
        uint2 pk[MT][NTW];
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            const int cb = (wc * NTW + nt) * 16 + 4 * fg;
            const float4 b4 = *(const float4*)(s_bias + n0 + cb);
            const float bz[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                pk[mt][nt].x = (unsigned)f32_to_bf16(acc[mt][nt][0] + bz[0]) | ((unsigned)f32_to_bf16(acc[mt][nt][1] + bz[1]) << 16);
                pk[mt][nt].y = (unsigned)f32_to_bf16(acc[mt][nt][2] + bz[2]) | ((unsigned)f32_to_bf16(acc[mt][nt][3] + bz[3]) << 16);
            }
        }
        const int scol = n0 + sch * 8;
        int orow_[8], ooff[8];
        u32x4_t av[4], vals[8];
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int R = it * 32 + rl, oh = r0 + (R >> 4), ow = c0 + (R & 15);
            orow_[it] = (oh < p.OH && ow < p.OW) ? (n * p.OH + oh) * p.OW + ow : -1;
            ooff[it] = orow_[it] >= 0 ? (orow_[it] * p.ldo + scol) * 2 : (int)0x80000000u;
        }
        // the residual chunks of a round's four rows are fetched before the round's tile rows are written (round 1's behind
        // round 0's stores): all in flight together, a barrier or two ahead of their use
        auto addend_rows = [&](int round) {
#pragma unroll
            for (int i = 0; i < 4; ++i) av[i] = __builtin_amdgcn_raw_buffer_load_b128(rsAdd, ooff[round * 4 + i], 0, 0);
        };
        // (assigned on every path: a conditionally assigned array is live across the whole tile loop -- the slice loop spilled it)
#pragma unroll
        for (int i = 0; i < 4; ++i) av[i] = u32x4_t{0, 0, 0, 0};
        if (p.addend) addend_rows(0);
#pragma unroll
        for (int round = 0; round < 2; ++round) {
            if ((wr >> 1) == round && (!(CW_ABL & 16) || p.ldo < 0)) {
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        *(uint2*)(tile + (((wr & 1) * MT + mt) * 16 + fi) * CW_PITCH + ((wc * NTW + nt) * 16 + 4 * fg) * 2) = pk[mt][nt];
            }
            ch_lds_barrier();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int it = round * 4 + i, row = i * 32 + rl;
                u32x4_t val = *(const u32x4_t*)(tile + row * CW_PITCH + sch * 16);
                if (p.addend) {
                    bf16_t* ev = (bf16_t*)&val;
                    const bf16_t* ae = (const bf16_t*)&av[i];
#pragma unroll
                    for (int e = 0; e < 8; ++e) Elem<bf16_t>::store(ev + e, Elem<bf16_t>::load(ev + e) + Elem<bf16_t>::load(ae + e));
                }
                vals[it] = val;
                if (!(CW_ABL & 16) || p.ldo < 0) __builtin_amdgcn_raw_buffer_store_b128(val, rsOut, ooff[it], 0, 0);
            }
            if (round == 0) {
                if (p.addend) addend_rows(1);
                ch_lds_barrier();
            }
        }
        if (p.stats) {
            // GroupNorm partial statistics of the STORED tile (conv_halo_k's, operation for operation)
            float sum[8], m2[8];
            int cn = 0;
#pragma unroll
            for (int e = 0; e < 8; ++e) { sum[e] = 0.f; m2[e] = 0.f; }
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                if (orow_[it] >= 0) {
                    ++cn;
                    const bf16_t* ev = (const bf16_t*)&vals[it];
#pragma unroll
                    for (int e = 0; e < 8; ++e) sum[e] += Elem<bf16_t>::load(ev + e);
                }
            }
            const float inv = cn ? 1.f / (float)cn : 0.f;
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                if (orow_[it] >= 0) {
                    const bf16_t* ev = (const bf16_t*)&vals[it];
#pragma unroll
                    for (int e = 0; e < 8; ++e) { const float d = Elem<bf16_t>::load(ev + e) - sum[e] * inv; m2[e] += d * d; }
                }
            }
            float fcnt = (float)cn;
            // (__shfl_xor derives the lane index from v_mbcnt: hoisted out of the tile loop and spilled like the rest)
            auto lane_xor = [&](float v, int off) {
                return __int_as_float(__builtin_amdgcn_ds_bpermute((lane ^ off) << 2, __float_as_int(v)));
            };
#pragma unroll
            for (int off = 16; off <= 32; off <<= 1) {
                const float ocnt = lane_xor(fcnt, off);
                const float ncnt = fcnt + ocnt;
                const float wgt = ncnt > 0.f ? fcnt * ocnt / ncnt : 0.f;
                const float ia = fcnt > 0.f ? 1.f / fcnt : 0.f, ib = ocnt > 0.f ? 1.f / ocnt : 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float osum = lane_xor(sum[e], off), om2 = lane_xor(m2[e], off);
                    const float d = osum * ib - sum[e] * ia;
                    m2[e] = m2[e] + om2 + d * d * wgt;
                    sum[e] += osum;
                }
                fcnt = ncnt;
            }
            ch_lds_barrier();
            float* r_sum = (float*)tile;                       // [8 waves][BN]
            float* r_m2 = r_sum + 8 * CH_BN;                   // [8][BN]
            float* r_cnt = r_m2 + 8 * CH_BN;                   // [8]
            if (lane < 16) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    r_sum[w * CH_BN + sch * 8 + e] = sum[e];
                    r_m2[w * CH_BN + sch * 8 + e] = m2[e];
                }
                if (lane == 0) r_cnt[w] = fcnt;
            }
            ch_lds_barrier();
            const int cg = p.stats_cg;
            if (tid < CH_BN) {
                float S = 0.f, N = 0.f;
#pragma unroll
                for (int kq = 0; kq < 8; ++kq) { S += r_sum[kq * CH_BN + tid]; N += r_cnt[kq]; }
                const float mean = N > 0.f ? S / N : 0.f;
                float M2 = 0.f;
#pragma unroll
                for (int kq = 0; kq < 8; ++kq) {
                    const float nk = r_cnt[kq];
                    const float d = nk > 0.f ? r_sum[kq * CH_BN + tid] / nk - mean : 0.f;
                    M2 += r_m2[kq * CH_BN + tid] + nk * d * d;
                }
                // group = cg adjacent channels = cg adjacent lanes of this wave: every lane sums its group's channels in channel
                // order through lane reads (conv_halo_k's sums in conv_halo_k's order, without its third barrier and the
                // latency-bound loop of 32 threads over LDS)
                const int gl0 = (lane / cg) * cg;
                float gm = 0.f;
                for (int c = 0; c < cg; ++c) gm += __int_as_float(__builtin_amdgcn_ds_bpermute((gl0 + c) << 2, __float_as_int(mean)));
                gm /= (float)cg;
                float gM2 = 0.f;
                for (int c = 0; c < cg; ++c) {
                    const float mc = __int_as_float(__builtin_amdgcn_ds_bpermute((gl0 + c) << 2, __float_as_int(mean)));
                    const float qc = __int_as_float(__builtin_amdgcn_ds_bpermute((gl0 + c) << 2, __float_as_int(M2)));
                    const float d = mc - gm;
                    gM2 += qc + N * d * d;
                }
                const int Gn = p.Nout / cg;
                if (lane == gl0) ((float2*)p.stats)[(size_t)mtile * Gn + n0 / cg + tid / cg] = make_float2(gm, gM2);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the scratch is a patch buffer again behind the next barrier
        }
        // the next tile starts from zero accumulators and zero pending fragments (zeroed HERE: live through the epilogue they
        // cost it 96 registers and spilled the residual prefetch)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) acc[mt][nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < MT; ++i) ya[i] = u32x4_t{0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < NTW; ++i) yb[i] = u32x4_t{0, 0, 0, 0};
    }
}

bool ch_ws_covers(const ChArgs& a) {
    const long lim = 1l << 31;
    return (long)a.Nimg * a.IH * a.IW * a.lda * 2 < lim && (long)a.Nout * 9 * a.Kc * 2 < lim && (long)a.Nimg * a.Kc * 4 < lim &&
           (long)a.Nimg * a.OH * a.OW * a.ldo * 2 < lim && a.Kc % 64 == 0 && a.Nout <= CW_MAXN &&
           (!a.stats || (a.stats_cg <= 64 && 64 % a.stats_cg == 0));       // a statistics group within one wave's lanes
}

int launch_ch_ws(const ChArgs& a, hipStream_t st) {
    static int ncu = 0;
    static bool attr_set = false;
    if (!attr_set) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) ncu = 256;
        (void)hipFuncSetAttribute((const void*)conv_halo_ws_k<true>, hipFuncAttributeMaxDynamicSharedMemorySize, CW_LDS);
        (void)hipFuncSetAttribute((const void*)conv_halo_ws_k<false>, hipFuncAttributeMaxDynamicSharedMemorySize, CW_LDS);
        attr_set = true;
    }
    const int grid = a.total < ncu ? a.total : ncu;
    if (a.gn_scale) hipLaunchKernelGGL((conv_halo_ws_k<true>), dim3(grid), dim3(768), CW_LDS, st, a);
    else hipLaunchKernelGGL((conv_halo_ws_k<false>), dim3(grid), dim3(768), CW_LDS, st, a);
    RBVAE_CHECK_LAUNCH("conv3x3_halo (persistent)");
    return RBVAE_OK;
}

}  // namespace rbvae
