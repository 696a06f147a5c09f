// Halo-resident 3x3 STRIDE-2 (pad 1) convolution on the gfx950 matrix cores, bf16:
//
//   Out[n][r][c][co] = epi( sum_{kh,kw,ci} A[n][2r + kh - 1][2c + kw - 1][ci] * W[co][kh*3 + kw][ci] )
//
// nn.Conv2d(c, c, 3, 2, 1) forward (percep_RBVAE_model.py:54-57; the LDM encoder's Downsample,
// ldm/modules/diffusionmodules/model.py:60-79, is the same sum with its asymmetric pad moved into the index) and the input
// gradient of nn.ConvTranspose2d(c, c, 3, 2, 1, 1) (autograd of :76-81 as run by percep_RBVAE_train.py:552), which is this
// convolution of the output gradient.  rbvae_gather_gemm re-gathers one pixel row per tap and K slice: a 128 x 128 tile
// takes 32 KB through the CU's L2 -> LDS path per 2.1 MFLOP and runs at that path's rate (15.2 KB / MFLOP).
// Here a workgroup owns 8 x 16 output pixels x BN output channels; per 32-channel slice the 17 x 33 input patch is staged
// ONCE (36 KB) and all nine taps read it, only the weight tap tiles stream (9 x BN x 64 B): 9.7 KB / MFLOP at BN = 256.
//
// LDS images
//   * input patch, chunk-major per 32-channel slice: [4 chunks of 16 B][576 slots], four parity sub-planes so that the
//     stride-2 taps read CONSECUTIVE slots: input pixel (u, v) relative to (2 r0, 2 c0), u in -1..15, v in -1..31, lives at
//     base[u odd][v odd] + ((u + 1) >> 1) * pitch + ((v + 1) >> 1) (pitch 17 for odd columns, 16 for even ones); tap
//     (kh, kw) of output pixel (rr, cc) is base[kh != 1][kw != 1] + (rr + (kh == 2)) * pitch + cc + (kw == 2): 16
//     consecutive pixels of an output row are 256 contiguous bytes of one chunk plane = a conflict-free ds_read_b128 at
//     every shift.  Double buffered (the next slice arrives during this one's taps).
//   * weight tap tiles [BN co][64 B], the 16-byte chunk position XORed with 3 * bit 3 of the row (the four rows a
//     ds_read_b128 lane group takes from one 256-byte window then sit on distinct banks), ring of four.
// Both are filled by LDS-DMA (buffer_load .. lds: a lane whose pixel is padding points beyond the descriptor and the
// hardware writes zeros, tools/probes/buffer_lds_oob.hip) by FOUR PRODUCER WAVES that do nothing else; the EIGHT MFMA waves
// only read fragments and multiply (rbvae_wgrad3x3s2_row's measurement: an LDS-DMA piece holds its issuing wave ~70 cycles,
// and an in-order wave that sits in the vector-memory issue feeds no MFMA).  One barrier per (slice, tap) unit; the MFMAs of a
// unit are split around the next unit's barrier and fragment reads (gather_gemm.hip's software pipeline).
// Epilogue = rbvae_gather_gemm's, element for element: +bias, ReLU, *scale, keyed / explicit dropout (same element indices),
// ReLU gate, 16-byte NHWC stores through an LDS tile, per-tile column sums (bias gradients).
#include "common.h"
#include <type_traits>

#ifndef CS_ABL          // timing ablations (results wrong on purpose; -DRBVAE_ABLATION builds only): 1 no LDS-DMA, 2 no fragment reads / MFMAs
#define CS_ABL 0
#endif
#if CS_ABL && !defined(RBVAE_ABLATION)
#error "CS_ABL builds give wrong results: define RBVAE_ABLATION to confirm"
#endif

namespace rbvae {

typedef __attribute__((ext_vector_type(8))) short cs_bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float cs_f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned cs_u32x4_t;

struct CsArgs {
    const unsigned char* A;        // [Nimg*IH*IW][lda] bf16
    const unsigned char* W;        // [Nout][9][Kc] bf16
    unsigned char* Out;            // [Nimg*OH*OW][ldo] bf16
    const float* bias;             // [Nout] or null
    const unsigned char* gate;     // [Nimg*OH*OW][ldo] bf16 or null: zero the output where gate <= 0
    const unsigned char* mask;     // [Nimg*OH*OW][Nout] u8 keep-mask or null
    float* colsum_ws;              // null or [mtiles][Nout]
    const unsigned long long* seed_dev;
    unsigned long long seed;
    int Nimg, IH, IW, OH, OW, Kc, Nout, lda, ldo;
    int relu, drop_mode;
    float scale;
    unsigned drop_thresh;
    int tiles_r, tiles_c, ntn, total;
};

constexpr int CS_TR = 8, CS_TC = 16, CS_BM = CS_TR * CS_TC;       // output tile: 8 rows x 16 columns
constexpr int CS_NSLOT = 576;                                     // 561 used: 9x17 + 9x16 + 8x17 + 8x16
constexpr int CS_PLANE = CS_NSLOT * 16;                           // bytes per chunk plane
constexpr int CS_PATCH = 4 * CS_PLANE;                            // one 32-channel slice of the patch: 36 864 B
constexpr int CS_RING = 4;
constexpr int CS_PPW = 9;                                         // patch LDS-DMA pieces per producer wave and slice (4 x 9 x 1 KiB)
// sub-plane (row parity, column parity) bases and pitches: odd rows 9, even rows 8; odd columns 17, even columns 16
__host__ __device__ constexpr int cs_base(bool rodd, bool codd) { return rodd ? (codd ? 0 : 153) : (codd ? 297 : 433); }
__host__ __device__ constexpr int cs_pitch(bool codd) { return codd ? 17 : 16; }
// patch pieces issued in the LDS-DMA slot of unit j of a slice (for the NEXT slice): all nine in flight by unit 4, so that the
// wait for the next slice's first weight tile (issued at unit 6) covers them
__host__ __device__ constexpr int cs_pp(int j) { return j < 0 ? 0 : j < 4 ? 2 : j == 4 ? 1 : 0; }
__host__ __device__ constexpr int cs_pp_before(int j) { int s = 0; for (int k = 0; k < j; ++k) s += cs_pp(k); return s; }

template <bool GATE> __device__ __forceinline__ bool cs_pos(const unsigned char* p, int e) {
    const bf16_t v = ((const bf16_t*)p)[e];
    return (v & 0x8000u) == 0 && (v & 0x7fffu) != 0 && (v & 0x7fffu) <= 0x7f80u;
}

template <int I, int N, typename F> __device__ __forceinline__ void cs_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        cs_static_for<I + 1, N>(f);
    }
}
__device__ __forceinline__ void cs_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
template <int N> __device__ __forceinline__ void cs_wait_barrier() {
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}

template <int BN> constexpr int cs_lds_main() {
    constexpr int ring = 2 * CS_PATCH + CS_RING * BN * 64;
    constexpr int epi = CS_BM * (BN * 2 + 16) + 16 * BN * 4;      // output tile + column-sum scratch
    return ring > epi ? ring : epi;
}

template <int BN>
__global__ __launch_bounds__(768, 1) void conv_s2_k(const CsArgs p) {
    constexpr int MT = BN / 64;                 // 16-pixel output rows per MFMA wave: 4 (BN 256) / 2 (BN 128)
    constexpr int WCO = BN / 64;                // MFMA waves along the channels (64 each); 8 / WCO along the pixels
    constexpr int NT = 4;
    constexpr int NW = BN / 64;                 // weight LDS-DMA pieces (16 rows x 64 B) per producer wave and tap tile
    constexpr int TILE_B = BN * 64;             // one tap tile
    constexpr int MAIN = cs_lds_main<BN>();
    static_assert(MT == 4 || MT == 2, "BN = 256 or 128");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int* s_orow = (int*)(smem + MAIN);          // [128] output pixel row of every tile row, -1 outside the image

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);

    // work item: the channel tiles of one pixel tile run back to back on ONE XCD (they re-read the same patch from its L2),
    // and an XCD walks a contiguous range of pixel tiles (neighbours share halo rows).  Bijective for any total.
    int item;
    {
        const int lin = blockIdx.x, xcd = lin & 7, q = p.total >> 3, r = p.total & 7;
        item = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (lin >> 3);
    }
    const int mtile = item / p.ntn, ntile = item - mtile * p.ntn;
    const int per_img = p.tiles_r * p.tiles_c;
    const int n = mtile / per_img, trc = mtile - n * per_img;
    const int tr = trc / p.tiles_c, tc = trc - tr * p.tiles_c;
    const int r0 = tr * CS_TR, c0 = tc * CS_TC, n0 = ntile * BN;
    const int nsl = p.Kc >> 5;                  // 32-channel slices
    const int U = nsl * 9;                      // (slice, tap) units

    if (w >= 8) {
        // ================= producer waves =================
        // pw = w - 8 owns chunk plane pw of the patch (9 pieces of 64 slots per slice) and rows 16 (pw NW + i) .. + 15 of every
        // weight tap tile.  Source offsets: a per-lane constant (computed once: the tile is fixed) + the instruction's SGPR
        // offset (image / slice; tile row block, tap, slice).
        const int pw = w - 8;
        int a_off[CS_PPW];
#pragma unroll
        for (int k = 0; k < CS_PPW; ++k) {
            const int slot = 64 * k + lane;
            // slot -> (u, v): sub-planes [0,153) odd/odd 9x17, [153,297) odd/even 9x16, [297,433) even/odd 8x17, [433,561) even/even 8x16
            const bool rodd = slot < 297, codd = slot < 153 || (slot >= 297 && slot < 433);
            const int rel = slot - cs_base(rodd, codd), pitch = cs_pitch(codd);
            const int ri = rel / pitch, ci = rel - ri * pitch;
            const int u = rodd ? 2 * ri - 1 : 2 * ri, v = codd ? 2 * ci - 1 : 2 * ci;
            const int Uy = 2 * r0 + u, Vx = 2 * c0 + v;
            const bool ok = slot < 561 && Uy >= 0 && Uy < p.IH && Vx >= 0 && Vx < p.IW;
            a_off[k] = ok ? (Uy * p.IW + Vx) * p.lda * 2 + pw * 16 : (int)0x80000000u;
        }
        int w_off[4];                           // (NW <= 4 used; a template-dependent bound here kept hipcc from emitting the host stub)
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int r = lane >> 2, pos = lane & 3;
            const int row = 16 * (pw * NW + i) + r;
            w_off[i] = row * 9 * p.Kc * 2 + ((pos ^ (3 * ((r >> 3) & 1))) * 16);
        }
        const auto rsrcA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, p.Nimg * p.IH * p.IW * p.lda * 2, 0x00020000);
        const auto rsrcW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, p.Nout * 9 * p.Kc * 2, 0x00020000);
        const int a_img = n * p.IH * p.IW * p.lda * 2;
        const int w_tile = n0 * 9 * p.Kc * 2;
        auto* lds = (__attribute__((address_space(3))) unsigned char*)smem;
        // patch piece k of slice sl -> buffer sl & 1
        auto patch_piece = [&](int sl, int k_lo, int k_hi) {
#if CS_ABL != 1
            auto* dst = lds + (sl & 1) * CS_PATCH + pw * CS_PLANE;
#pragma unroll
            for (int k = 0; k < CS_PPW; ++k)
                if (k >= k_lo && k < k_hi)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, dst + k * 1024, 16, a_off[k], a_img + sl * 64, 0, 0);
#endif
        };
        // weight tile of (slice sl, tap j) -> ring slot
        auto tile = [&](int sl, int j, int slot) {
#if CS_ABL != 1
            auto* dst = lds + 2 * CS_PATCH + slot * TILE_B + (pw * NW) * 1024;
#pragma unroll
            for (int i = 0; i < NW; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcW, dst + i * 1024, 16, w_off[i], w_tile + (j * p.Kc + sl * 32) * 2, 0, 0);
#endif
        };
        // prologue: the first slice's patch, tiles 0, 1, 2 (a layer has at least one slice = 9 units)
        patch_piece(0, 0, CS_PPW);
        tile(0, 0, 0); tile(0, 1, 1); tile(0, 2, 2);
        int slot_next = 3;                      // ring slot of the tile issued next (tile u + 3 in the slot of unit u)
        // unit (sl, j): own pieces of tile (sl, j) -- and of slice sl's patch, all older -- landed -> barrier -> issue the next
        // slice's patch pieces of this unit, then tile u + 3.  Younger than tile u at that wait: tiles u + 1, u + 2 and the patch
        // pieces of the two units before this one.
        auto slice = [&](int sl, auto more_tag) {
            constexpr bool more = decltype(more_tag)::value;         // another slice follows
            cs_static_for<0, 9>([&](auto j_tag) {
                constexpr int j = decltype(j_tag)::value;
                constexpr int tiles_after = more ? 2 : (8 - j < 2 ? 8 - j : 2);
                constexpr int young = tiles_after * NW + (more ? cs_pp(j - 2) + cs_pp(j - 1) : 0);
                cs_wait_barrier<young>();
                if constexpr (more) {
                    if constexpr (cs_pp(j) > 0) patch_piece(sl + 1, cs_pp_before(j), cs_pp_before(j) + cs_pp(j));
                }
                if constexpr (j + 3 < 9) {
                    tile(sl, j + 3, slot_next);
                    slot_next = (slot_next + 1) & (CS_RING - 1);
                } else if constexpr (more) {
                    tile(sl + 1, j + 3 - 9, slot_next);
                    slot_next = (slot_next + 1) & (CS_RING - 1);
                }
            });
        };
        for (int sl = 0; sl + 1 < nsl; ++sl) slice(sl, std::true_type{});
        slice(nsl - 1, std::false_type{});
        return;
    }

    // ================= MFMA waves =================
    const int fi = lane & 15, fg = lane >> 4;
    const int wr = w / WCO, wc = w - wr * WCO;           // pixel rows wr * MT .. + MT - 1 of the tile, channels 64 wc .. + 63
    if (tid < CS_BM) {
        const int oh = r0 + (tid >> 4), ow = c0 + (tid & 15);
        s_orow[tid] = (oh < p.OH && ow < p.OW) ? (n * p.OH + oh) * p.OW + ow : -1;
    }
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    // patch fragment of output row (wr MT + mt): chunk plane fg, slots slot(tap, row, 0) + fi; the row's part of the slot index
    // depends on the tap's column parity through the pitch: one base per parity, everything else an immediate
    const unsigned vA17 = lds0 + fg * CS_PLANE + (wr * MT * 17 + fi) * 16;
    const unsigned vA16 = lds0 + fg * CS_PLANE + (wr * MT * 16 + fi) * 16;
    // weight fragment of channel sub-tile nt: row 64 wc + 16 nt + fi of the tap tile, chunk fg at its swizzled position
    const unsigned vB = lds0 + 2 * CS_PATCH + (wc * 64 + fi) * 64 + ((fg ^ (3 * ((fi >> 3) & 1))) * 16);

    cs_f32x4_t acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = cs_f32x4_t{0.f, 0.f, 0.f, 0.f};

    // fragment reads (inline asm: invisible to the compiler's wait-count pass; isa_check proves the counted waits)
    auto read_unit = [&](auto j_tag, unsigned pbuf, unsigned rslot, cs_u32x4_t (&fa)[MT], cs_u32x4_t (&fb)[NT]) {
        constexpr int j = decltype(j_tag)::value, kh = j / 3, kw = j % 3;
        constexpr bool codd = kw != 1;
        constexpr int pitch = cs_pitch(codd);
        constexpr int aimm = (cs_base(kh != 1, codd) + (kh == 2 ? pitch : 0) + (kw == 2 ? 1 : 0)) * 16;
        const unsigned ab = vB + rslot;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[nt]) : "v"(ab), "n"(nt * 1024));
        const unsigned aa = (codd ? vA17 : vA16) + pbuf;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[mt]) : "v"(aa), "n"(aimm + mt * pitch * 16));
    };
    auto landed = [&](cs_u32x4_t (&fa)[MT], cs_u32x4_t (&fb)[NT]) {
        if constexpr (MT == 4)
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(fa[0]), "+v"(fa[1]), "+v"(fa[2]), "+v"(fa[3]), "+v"(fb[0]), "+v"(fb[1]), "+v"(fb[2]), "+v"(fb[3]));
        else
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[0]), "+v"(fa[1]), "+v"(fb[0]), "+v"(fb[1]), "+v"(fb[2]), "+v"(fb[3]));
    };
    // the MFMAs of output rows m0 .. m0 + MT / 2 - 1 of a unit (weights as the row operand: a lane owns 4 consecutive output
    // channels of one pixel)
    auto mma_half = [&](auto m0_tag, const cs_u32x4_t (&fa)[MT], const cs_u32x4_t (&fb)[NT]) {
        constexpr int M0 = decltype(m0_tag)::value;
#pragma unroll
        for (int mt = M0; mt < M0 + MT / 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const cs_bf16x8_t*)&fb[nt], *(const cs_bf16x8_t*)&fa[mt],
                                                                      acc[mt][nt], 0, 0, 0);
    };
    using MLO = std::integral_constant<int, 0>;
    using MHI = std::integral_constant<int, MT / 2>;

    cs_u32x4_t xa[MT], xb[NT], ya[MT], yb[NT];
#pragma unroll
    for (int i = 0; i < MT; ++i) { xa[i] = cs_u32x4_t{0, 0, 0, 0}; ya[i] = cs_u32x4_t{0, 0, 0, 0}; }
#pragma unroll
    for (int i = 0; i < NT; ++i) { xb[i] = cs_u32x4_t{0, 0, 0, 0}; yb[i] = cs_u32x4_t{0, 0, 0, 0}; }
    unsigned rslot = 0;                         // byte offset of the ring slot of the unit being read
    // Unit u: barrier (its weight tile -- and at a slice's first tap its patch -- landed: the producer waves waited in front of
    // it; every MFMA wave is past the reads of unit u - 1, whose ring slot tile u + 3 takes) -> this unit's fragment reads ->
    // the second half of unit u - 1's MFMAs (zero fragments at u = 0) -> wait -> the first half of this unit's.
    // Fragment sets alternate (x, y) with the unit's parity; a slice has nine units, so a slice body starts on either set.
    auto slice = [&](int sl, auto par_tag) {
        constexpr int PAR = decltype(par_tag)::value;
        const unsigned pbuf = (sl & 1) * CS_PATCH;
        cs_static_for<0, 9>([&](auto j_tag) {
            constexpr int j = decltype(j_tag)::value;
            asm volatile("s_barrier" ::: "memory");
#if CS_ABL != 2
            if constexpr (((PAR + j) & 1) == 0) {
                read_unit(j_tag, pbuf, rslot, xa, xb);
                __builtin_amdgcn_sched_barrier(0);
                mma_half(MHI{}, ya, yb);
                __builtin_amdgcn_sched_barrier(0);
                landed(xa, xb);
                mma_half(MLO{}, xa, xb);
            } else {
                read_unit(j_tag, pbuf, rslot, ya, yb);
                __builtin_amdgcn_sched_barrier(0);
                mma_half(MHI{}, xa, xb);
                __builtin_amdgcn_sched_barrier(0);
                landed(ya, yb);
                mma_half(MLO{}, ya, yb);
            }
            __builtin_amdgcn_sched_barrier(0);
#endif
            rslot = rslot + TILE_B == CS_RING * TILE_B ? 0 : rslot + TILE_B;
        });
    };
    // two slices per loop iteration (18 units: the same fragment set is live at every back edge; with the set depending on the
    // slice's parity the compiler kept both alive across the loop and spilled 69 registers)
    int sl = 0;
    for (; sl + 1 < nsl; sl += 2) {
        slice(sl, std::integral_constant<int, 0>{});
        slice(sl + 1, std::integral_constant<int, 1>{});
    }
    if (sl < nsl) {
        slice(sl, std::integral_constant<int, 0>{});
#if CS_ABL != 2
        mma_half(MHI{}, xa, xb);                // the last unit's second half (tap 8 of an even slice: set x)
#endif
    } else {
#if CS_ABL != 2
        mma_half(MHI{}, ya, yb);                // (tap 8 of an odd slice: set y)
#endif
    }
    cs_lds_barrier();                           // (the producer waves have left: the barrier counts the MFMA waves only)

    // ---- epilogue: bias, ReLU, scale -> bf16 tile in LDS -> dropout / gate -> 16-byte NHWC stores, column sums
    constexpr int PITCH = BN * 2 + 16;
    constexpr int CPR = BN / 8;                  // 16-byte chunks per tile row
    constexpr int RL = 512 / CPR;                // row lanes of the store phase
    constexpr int ITERS = CS_BM / RL;
    unsigned char* tl = smem;
    float* red = (float*)(smem + CS_BM * PITCH);            // [RL][BN]
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int cb = wc * 64 + nt * 16 + 4 * fg;
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
                float x = acc[mt][nt][r] + bz[r];
                if (p.relu) x = fmaxf(x, 0.f);
                v[r] = x * p.scale;
            }
            uint2 pk;
            pk.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
            pk.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
            *(uint2*)(tl + row * PITCH + cb * 2) = pk;
        }
    }
    cs_lds_barrier();
    const int sch = tid % CPR, rl = tid / CPR;
    const int scol = n0 + sch * 8;
    DropKey dkey{0u, 0u};
    if (p.drop_mode == 1) dkey = drop_key(p.seed + (p.seed_dev ? p.seed_dev[0] * 0x9E3779B97F4A7C15ull : 0ull));
    int orow_[ITERS];
    cs_u32x4_t gv[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        orow_[it] = s_orow[it * RL + rl];
        if (p.gate) gv[it] = *(const cs_u32x4_t*)(p.gate + ((size_t)(orow_[it] < 0 ? 0 : orow_[it]) * p.ldo + scol) * 2);
    }
    float csum[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) csum[e] = 0.f;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int row = it * RL + rl, orow = orow_[it];
        if (orow >= 0) {
            cs_u32x4_t val = *(const cs_u32x4_t*)(tl + row * PITCH + sch * 16);
            bf16_t* ev = (bf16_t*)&val;
            if (p.drop_mode == 1) {
                const unsigned run = drop_run(dkey, (unsigned long long)orow * p.Nout + scol);
                drop_chunk_zero_b16<8>(run, p.drop_thresh >> 16, (unsigned*)&val);
            } else if (p.drop_mode == 2) {
                const unsigned char* mk = p.mask + (size_t)orow * p.Nout + scol;
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (!mk[e]) ev[e] = 0;
            }
            if (p.gate) {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (!cs_pos<true>((const unsigned char*)&gv[it], e)) ev[e] = 0;
            }
            *(cs_u32x4_t*)(p.Out + ((size_t)orow * p.ldo + scol) * 2) = val;
            if (p.colsum_ws) {
#pragma unroll
                for (int e = 0; e < 8; ++e) csum[e] += bf16_to_f32(ev[e]);
            }
        }
    }
    if (p.colsum_ws) {
#pragma unroll
        for (int e = 0; e < 8; ++e) red[rl * BN + sch * 8 + e] = csum[e];
        cs_lds_barrier();
        if (tid < BN) {
            float t = 0.f;
#pragma unroll 8
            for (int k = 0; k < RL; ++k) t += red[k * BN + tid];
            p.colsum_ws[(size_t)mtile * p.Nout + n0 + tid] = t;
        }
    }
}

static int cs_bn(int dtype, int Nimg, int IH, int IW, int Kc, int Nout) {
    if (dtype != RBVAE_BF16 || Nimg < 1 || IH < 2 || IW < 2 || (IH & 1) || (IW & 1) || Kc < 32 || Kc % 32) return 0;
    // 32-bit buffer offsets: both operands below 2 GiB
    if ((long)Nimg * IH * IW * Kc * 2 >= (1l << 31) || (long)Nout * 9 * Kc * 2 >= (1l << 31)) return 0;
    if (Nout >= 256 && Nout % 256 == 0) return 256;
    if (Nout >= 128 && Nout % 128 == 0) return 128;
    return 0;
}

template <int BN>
static int launch_cs(const CsArgs& a, hipStream_t st) {
    constexpr size_t lds = (size_t)cs_lds_main<BN>() + CS_BM * 4 + 16;
    static_assert(lds <= 160 * 1024, "LDS carve");
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)conv_s2_k<BN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((conv_s2_k<BN>), dim3(a.total), dim3(768), lds, st, a);
    RBVAE_CHECK_LAUNCH("conv3x3s2_halo");
    return RBVAE_OK;
}

}  // namespace rbvae

using namespace rbvae;

extern "C" {

/* output channels per workgroup (256 / 128) when the shape is covered, else 0 */
int rbvae_conv3x3s2_halo_ok(int dtype, int Nimg, int IH, int IW, int Kc, int Nout) { return cs_bn(dtype, Nimg, IH, IW, Kc, Nout); }

/* rows of colsum_ws: one per 8 x 16 pixel tile */
int rbvae_conv3x3s2_halo_colsum_rows(int Nimg, int IH, int IW) {
    return Nimg * cdiv(IH / 2, CS_TR) * cdiv(IW / 2, CS_TC);
}

int rbvae_conv3x3s2_halo(int dtype, const void* A, const void* W, void* Out, const float* bias, const void* gate, const void* mask,
                         int Nimg, int IH, int IW, int Kc, int Nout, int lda, int ldo, int relu, int drop_mode, float drop_p,
                         float scale, unsigned long long seed, const unsigned long long* seed_dev, float* colsum_ws, void* stream) {
    RBVAE_CHECK_ARG(A && W && Out, "conv3x3s2_halo: null pointer");
    const int bn = cs_bn(dtype, Nimg, IH, IW, Kc, Nout);
    RBVAE_CHECK_ARG(bn, "conv3x3s2_halo: shape not covered (dtype %d, %d x %d x %d, Kc %d, Nout %d): query rbvae_conv3x3s2_halo_ok",
                    dtype, Nimg, IH, IW, Kc, Nout);
    RBVAE_CHECK_ARG(lda >= Kc && lda % 8 == 0 && ldo >= Nout && ldo % 8 == 0, "conv3x3s2_halo: leading dimensions lda=%d ldo=%d", lda, ldo);
    RBVAE_CHECK_ARG((long)Nimg * IH * IW * lda * 2 < (1l << 31), "conv3x3s2_halo: input of 2 GiB or more (32-bit buffer offsets)");
    RBVAE_CHECK_ARG(((uintptr_t)A | (uintptr_t)W | (uintptr_t)Out | (uintptr_t)gate | (uintptr_t)bias) % 16 == 0,
                    "conv3x3s2_halo: pointers must be 16-byte aligned");
    RBVAE_CHECK_ARG(drop_mode >= 0 && drop_mode <= 2 && (drop_mode != 2 || mask), "conv3x3s2_halo: drop_mode/mask");
    CsArgs a;
    a.A = (const unsigned char*)A; a.W = (const unsigned char*)W; a.Out = (unsigned char*)Out; a.bias = bias;
    a.gate = (const unsigned char*)gate; a.mask = (const unsigned char*)mask; a.colsum_ws = colsum_ws; a.seed_dev = seed_dev;
    a.seed = seed;
    a.Nimg = Nimg; a.IH = IH; a.IW = IW; a.OH = IH / 2; a.OW = IW / 2; a.Kc = Kc; a.Nout = Nout; a.lda = lda; a.ldo = ldo;
    a.relu = relu; a.drop_mode = drop_mode; a.scale = scale; a.drop_thresh = (unsigned)((double)drop_p * 4294967296.0);
    a.tiles_r = cdiv(a.OH, CS_TR); a.tiles_c = cdiv(a.OW, CS_TC); a.ntn = Nout / bn;
    const long total = (long)Nimg * a.tiles_r * a.tiles_c * a.ntn;
    RBVAE_CHECK_ARG(total < (1l << 30), "conv3x3s2_halo: too many tiles");
    a.total = (int)total;
    hipStream_t st = (hipStream_t)stream;
    return bn == 256 ? launch_cs<256>(a, st) : launch_cs<128>(a, st);
}

}  // extern "C"
