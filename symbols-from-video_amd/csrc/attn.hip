// Single-head self-attention of the LDM VAE encoder's mid block (AttnBlock.forward,
// src/stable-diffusion/ldm/modules/diffusionmodules/model.py:186-198) as ONE batched, tiled, online-softmax kernel:
//   w = softmax_j( q_i . k_j * C^-0.5 ),   o_i = sum_j w_ij v_j        per image, hw tokens of C channels
// The reference materialises the hw x hw scores (torch.bmm, softmax, bmm); at 512x512 frames that is 4096^2 scores per
// image, at the reference's native 704x1280 frames 14 080^2 = 396 MB per image in bf16.  Here a workgroup owns 64
// query rows of one image and walks the keys in tiles of 32: scores, softmax statistics and the output accumulators
// never leave the registers.
//
// Layout (gfx950, bf16 storage, f32 accumulation, v_mfma_f32_16x16x32_bf16):
//   * a 64-query tile per 4-wave workgroup, two workgroups per CU (one's tile transfer and barriers hide behind the
//     other's products; the 128-query / 8-wave form, which feeds twice the products from every staged tile, measured
//     slower: 503 vs 387 us at 4 x 4096 tokens, tools/time_attn.py); wave w owns query rows 16w..16w+15; its Q
//     fragments (C/32 x 8 bf16 per lane) and its output accumulators (C/16 tiles x 4 f32 per lane) stay in registers
//     for the whole key loop (64 + 128 VGPRs at C = 512);
//   * K / V tiles are staged by LDS-DMA at C = 512 (one row = one 1-KiB instruction, no registers, a wave's 16 rows in
//     flight together);
//   * K and V tiles [32 keys][C] go through LDS with 16-byte-padded rows (conflict-free ds_read_b128 for the K
//     fragments; V fragments, which need 8 consecutive KEYS per lane, by ds_read_b64_tr_b16 -- no transposed copy of V
//     in memory);
//   * the probabilities change from accumulator layout to A-operand layout through a 1.25 KB per-wave LDS patch;
//   * exp2 with the scale folded in; the row sums are taken over the bf16-rounded probabilities the PV product uses;
//   * the output accumulators are only rescaled when a row maximum moved (wave-uniform test).
// ~72 KB of LDS per workgroup at C = 512.  Measured (MI355X, bf16): 4 x 4096 tokens (512x512 frames) 387 us = 355 TFLOP/s,
// the three-launch form 393 us; 8 x 1024 tokens (256x256) 104 us against 329 us.
#include "common.h"
#include <stdlib.h>

namespace rbvae {

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

constexpr int AT_BK = 32, AT_PAD = 8;

template <int D, int AT_BQ>
__global__ __launch_bounds__(4 * AT_BQ, AT_BQ == 64 ? 2 : 1) void attn_flash_k(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K,
                                                       const bf16_t* __restrict__ V, bf16_t* __restrict__ O, int hw,
                                                       int ldq, int ldk, int ldv, int ldo, float scale_log2e) {
    constexpr int AT_WAVES = AT_BQ / 16;
    constexpr int LD = D + AT_PAD;                 // LDS row stride of the K / V tiles (elements)
    constexpr int PLD = AT_BK + AT_PAD;            // row stride of the probability patch
    extern __shared__ __attribute__((aligned(16))) bf16_t smem[];
    bf16_t* sK = smem;
    bf16_t* sV = sK + AT_BK * LD;
    bf16_t* sP = sV + AT_BK * LD;                  // [AT_WAVES][16][PLD]
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, m = l & 15, g = l >> 4;
    const long img = (long)blockIdx.y * hw;
    const int q0 = blockIdx.x * AT_BQ + 16 * w;

    // Q fragments: a[j] = Q[q0 + m][32 s + 8 g + j]
    bf16x8_t qa[D / 32];
    {
        const int qr = min(q0 + m, hw - 1);
        const bf16_t* qp = Q + (img + qr) * ldq + 8 * g;
#pragma unroll
        for (int s = 0; s < D / 32; ++s) qa[s] = *(const bf16x8_t*)(qp + 32 * s);
    }
    f32x4_t o[D / 16];
#pragma unroll
    for (int c = 0; c < D / 16; ++c) o[c] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    float mrun[4], lrun[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { mrun[r] = -INFINITY; lrun[r] = 0.f; }
    bf16_t* myP = sP + w * 16 * PLD;

    for (int k0 = 0; k0 < hw; k0 += AT_BK) {
        // ---- stage the K and V tiles: a wave moves one 2*D-byte row per instruction --------------------------------
        __syncthreads();                            // everyone is done with the previous tiles
        constexpr int CPR = D / 8;                  // 16-byte chunks per row
        if constexpr (CPR == 64) {
            // a row is exactly one LDS-DMA instruction (64 lanes x 16 bytes, lane-linear destination): no registers, and
            // all of a wave's 8 rows are in flight together.  (Through registers the loop became a load-wait-store
            // chain of four dependent L2 round trips per tile: 4.3 us per 32-key tile at 4096 tokens.)
            for (int r = w; r < AT_BK; r += AT_WAVES) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(K + (img + k0 + r) * ldk + 8 * l),
                                                 (__attribute__((address_space(3))) void*)(sK + r * LD), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(V + (img + k0 + r) * ldv + 8 * l),
                                                 (__attribute__((address_space(3))) void*)(sV + r * LD), 16, 0, 0);
            }
            // every wave's own transfers have landed before it arrives at the barrier (the compiler only guards a
            // wave's OWN later LDS reads)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            for (int i = tid; i < AT_BK * CPR; i += 64 * AT_WAVES) {
                const int r = i / CPR, cch = i - r * CPR;
                const u32x4_t kv = *(const u32x4_t*)(K + (img + k0 + r) * ldk + 8 * cch);
                const u32x4_t vv = *(const u32x4_t*)(V + (img + k0 + r) * ldv + 8 * cch);
                *(u32x4_t*)(sK + r * LD + 8 * cch) = kv;
                *(u32x4_t*)(sV + r * LD + 8 * cch) = vv;
            }
        }
        __syncthreads();
        // ---- scores S[q][key] for 2 x 16 keys ------------------------------------------------------------------------
        f32x4_t sc[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int s = 0; s < D / 32; ++s) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const bf16x8_t b = *(const bf16x8_t*)(sK + (16 * h + m) * LD + 32 * s + 8 * g);
                sc[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[s], b, sc[h], 0, 0, 0);
            }
        }
        // ---- online softmax: lane holds rows 4g + r (r < 4), keys 16h + m --------------------------------------------
        float alpha[4];
        bool moved = false;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float t0 = sc[0][r] * scale_log2e, t1 = sc[1][r] * scale_log2e;
            float mx = fmaxf(t0, t1);
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
            const float mnew = fmaxf(mrun[r], mx);
            alpha[r] = exp2f(mrun[r] - mnew);       // first tile: exp2(-inf) = 0
            moved |= mnew != mrun[r];
            mrun[r] = mnew;
            // probabilities, rounded to bf16 once: the row sum and the PV product see the same numbers
            const bf16_t p0 = f32_to_bf16(exp2f(t0 - mnew)), p1 = f32_to_bf16(exp2f(t1 - mnew));
            myP[(4 * g + r) * PLD + m] = p0;
            myP[(4 * g + r) * PLD + 16 + m] = p1;
            float rs = bf16_to_f32(p0) + bf16_to_f32(p1);
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) rs += __shfl_xor(rs, off, 64);
            lrun[r] = lrun[r] * alpha[r] + rs;
        }
        if (__any(moved)) {
#pragma unroll
            for (int c = 0; c < D / 16; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) o[c][r] *= alpha[r];
        }
        // the patch is private to this wave: LDS serves a wave's accesses in issue order, the fence keeps the compiler
        // from moving the read above the writes
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const bf16x8_t pa = *(const bf16x8_t*)(myP + m * PLD + 8 * g);       // a[j] = P[m][8 g + j]
        // ---- O += P V: b[j] = V[8 g + j][16 c + m] by two transposing reads ------------------------------------------
        const int i4 = m >> 2, p4 = m & 3;
        const bf16_t* vlo = sV + (8 * g + i4) * LD + 4 * p4;
        const bf16_t* vhi = vlo + 4 * LD;
#pragma unroll
        for (int c = 0; c < D / 16; ++c) {
            const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(vlo + 16 * c));
            const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(vhi + 16 * c));
            const bf16x8_t b = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            o[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa, b, o[c], 0, 0, 0);
        }
    }
    // ---- normalise and store: O[q0 + 4g + r][16 c + m] ---------------------------------------------------------------
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int qr = q0 + 4 * g + r;
        if (qr >= hw) continue;
        const float inv = 1.0f / lrun[r];
        bf16_t* op = O + (img + qr) * ldo + m;
#pragma unroll
        for (int c = 0; c < D / 16; ++c) op[16 * c] = f32_to_bf16(o[c][r] * inv);
    }
}

// ---- C = 512: K / V tiles prefetched three half-steps ahead by waves of their own --------------------------------------
// attn_flash_k loads a tile, waits for it, multiplies, and starts the next load behind a barrier: at 4 x 4096 tokens a
// 32-key step took 2.9 us of which the transfer (64 KB per workgroup: this shape takes 15 KB through the CU's L2 -> LDS path
// per MFLOP) was never overlapped.  Here the four MFMA waves never touch global memory inside the loop; two PRODUCER waves
// stream the K tiles and the V tiles through two LDS buffers each, as half-steps (2t: K_t for the scores, 2t + 1: V_t for
// P V), always three half-steps (96 KB) ahead of the one being consumed.  One barrier per half-step: it publishes the data of
// half-step h (the producers waited for their pieces in front of it) and frees the buffer of half-step h - 1, which the
// producers refill with the data of half-step h + 3.  The arithmetic (score order, bf16-rounded probabilities, rescaling
// rule) is attn_flash_k's, so the results are bit-identical to it.  Measured (MI355X, bf16, same box): 4 x 4096 tokens 386 -> 302 us
// (454 TFLOP/s), 8 x 1024 tokens 103 -> 79 us.  What is left is the single MFMA wave per SIMD (248 registers: Q fragments +
// output accumulators): it exposes its own LDS latency -- a 128-query tile on eight MFMA waves that also issue the LDS-DMA
// needs every fragment read as inline asm (the compiler drains the LDS-DMA in front of its own LDS reads).
template <int N> __device__ __forceinline__ void at_wait_barrier() {
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}
__device__ __forceinline__ void at_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__global__ __launch_bounds__(384, 1) void attn_flash_db_k(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K,
                                                          const bf16_t* __restrict__ V, bf16_t* __restrict__ O, int hw, int ldq,
                                                          int ldk, int ldv, int ldo, float scale_log2e) {
    constexpr int D = 512, LD = D + AT_PAD, PLD = AT_BK + AT_PAD;
    constexpr int TILE = AT_BK * LD;                // elements of one staged tile
    extern __shared__ __attribute__((aligned(16))) bf16_t smem[];
    bf16_t* sK = smem;                              // [2][32][LD]
    bf16_t* sV = sK + 2 * TILE;                     // [2][32][LD]
    bf16_t* sP = sV + 2 * TILE;                     // [4][16][PLD]
    const int tid = threadIdx.x, l = tid & 63, m = l & 15, g = l >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long img = (long)blockIdx.y * hw;
    const int nt = hw / AT_BK, nh = 2 * nt;         // key tiles, half-steps

    if (w >= 4) {
        // producer wave pw: rows pw, pw + 2, .. of every tile (16 one-row LDS-DMA instructions per half-step)
        const int pw = w - 4;
        auto issue = [&](int h) {
            const int t = h >> 1;
            const bf16_t* src = (h & 1 ? V : K) + (img + (long)t * AT_BK) * (h & 1 ? ldv : ldk) + 8 * l;
            bf16_t* dst = (h & 1 ? sV : sK) + (t & 1) * TILE;
            const long ld = h & 1 ? ldv : ldk;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int r = pw + 2 * i;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + r * ld),
                                                 (__attribute__((address_space(3))) void*)(dst + r * LD), 16, 0, 0);
            }
        };
        issue(0);
        if (nh > 1) issue(1);
        if (nh > 2) issue(2);
        for (int h = 0; h < nh; ++h) {
            // own pieces of half-step h landed (h + 1, h + 2 may be in flight) -> barrier -> refill the buffer of half-step h - 1
            if (h + 2 < nh) at_wait_barrier<32>();
            else if (h + 1 < nh) at_wait_barrier<16>();
            else at_wait_barrier<0>();
            if (h + 3 < nh) issue(h + 3);
        }
        return;
    }

    const int q0 = blockIdx.x * 64 + 16 * w;
    bf16x8_t qa[D / 32];
    {
        const int qr = min(q0 + m, hw - 1);
        const bf16_t* qp = Q + (img + qr) * ldq + 8 * g;
#pragma unroll
        for (int s = 0; s < D / 32; ++s) qa[s] = *(const bf16x8_t*)(qp + 32 * s);
    }
    f32x4_t o[D / 16];
#pragma unroll
    for (int c = 0; c < D / 16; ++c) o[c] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    float mrun[4], lrun[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { mrun[r] = -INFINITY; lrun[r] = 0.f; }
    bf16_t* myP = sP + w * 16 * PLD;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the Q fragments: the loop's barriers wait for LDS traffic only

    for (int t = 0; t < nt; ++t) {
        const bf16_t* cK = sK + (t & 1) * TILE;
        const bf16_t* cV = sV + (t & 1) * TILE;
        at_lds_barrier();                                     // half-step 2t: K_t landed; everyone is past P V of tile t - 1
        f32x4_t sc[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int s = 0; s < D / 32; ++s) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const bf16x8_t b = *(const bf16x8_t*)(cK + (16 * h + m) * LD + 32 * s + 8 * g);
                sc[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[s], b, sc[h], 0, 0, 0);
            }
        }
        float alpha[4];
        bool moved = false;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float t0 = sc[0][r] * scale_log2e, t1 = sc[1][r] * scale_log2e;
            float mx = fmaxf(t0, t1);
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
            const float mnew = fmaxf(mrun[r], mx);
            alpha[r] = exp2f(mrun[r] - mnew);
            moved |= mnew != mrun[r];
            mrun[r] = mnew;
            const bf16_t p0 = f32_to_bf16(exp2f(t0 - mnew)), p1 = f32_to_bf16(exp2f(t1 - mnew));
            myP[(4 * g + r) * PLD + m] = p0;
            myP[(4 * g + r) * PLD + 16 + m] = p1;
            float rs = bf16_to_f32(p0) + bf16_to_f32(p1);
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) rs += __shfl_xor(rs, off, 64);
            lrun[r] = lrun[r] * alpha[r] + rs;
        }
        if (__any(moved)) {
#pragma unroll
            for (int c = 0; c < D / 16; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) o[c][r] *= alpha[r];
        }
        at_lds_barrier();                                     // half-step 2t + 1: V_t landed; everyone is past the scores of tile t
        const bf16x8_t pa = *(const bf16x8_t*)(myP + m * PLD + 8 * g);
        const int i4 = m >> 2, p4 = m & 3;
        const bf16_t* vlo = cV + (8 * g + i4) * LD + 4 * p4;
        const bf16_t* vhi = vlo + 4 * LD;
#pragma unroll
        for (int c = 0; c < D / 16; ++c) {
            const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(vlo + 16 * c));
            const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(vhi + 16 * c));
            const bf16x8_t b = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            o[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa, b, o[c], 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int qr = q0 + 4 * g + r;
        if (qr >= hw) continue;
        const float inv = 1.0f / lrun[r];
        bf16_t* op = O + (img + qr) * ldo + m;
#pragma unroll
        for (int c = 0; c < D / 16; ++c) op[16 * c] = f32_to_bf16(o[c][r] * inv);
    }
}

static int launch_attn_db(const void* Q, const void* K, const void* V, void* O, int N, int hw, int ldq, int ldk, int ldv, int ldo,
                          float scale, hipStream_t st) {
    const size_t lds = (size_t)(4 * AT_BK * (512 + AT_PAD) + 4 * 16 * (AT_BK + AT_PAD)) * sizeof(bf16_t);
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void*)attn_flash_db_k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return fail(RBVAE_E_LAUNCH, "attention: cannot reserve %zu bytes of LDS", lds);
        attr = true;
    }
    hipLaunchKernelGGL(attn_flash_db_k, dim3(cdiv(hw, 64), N), dim3(384), lds, st, (const bf16_t*)Q, (const bf16_t*)K,
                       (const bf16_t*)V, (bf16_t*)O, hw, ldq, ldk, ldv, ldo, scale * 1.4426950408889634f);
    return RBVAE_OK;
}

template <int D, int AT_BQ>
static int launch_attn_bq(const void* Q, const void* K, const void* V, void* O, int N, int hw, int ldq, int ldk, int ldv,
                          int ldo, float scale, hipStream_t st) {
    constexpr int AT_WAVES = AT_BQ / 16;
    const size_t lds = (size_t)(2 * AT_BK * (D + AT_PAD) + AT_WAVES * 16 * (AT_BK + AT_PAD)) * sizeof(bf16_t);
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void*)attn_flash_k<D, AT_BQ>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return fail(RBVAE_E_LAUNCH, "attention: cannot reserve %zu bytes of LDS", lds);
        attr = true;
    }
    hipLaunchKernelGGL((attn_flash_k<D, AT_BQ>), dim3(cdiv(hw, AT_BQ), N), dim3(64 * AT_WAVES), lds, st, (const bf16_t*)Q,
                       (const bf16_t*)K, (const bf16_t*)V, (bf16_t*)O, hw, ldq, ldk, ldv, ldo, scale * 1.4426950408889634f);
    return RBVAE_OK;
}

// Query tile: 64 rows (4 waves, two workgroups per CU: one's tile transfer and barriers hide behind the other's
// products) or 128 rows (8 waves, one per CU: every staged K / V tile feeds twice the products).  RBVAE_ATTN_BQ picks;
// default chosen by measurement (tools/time_attn.py).
template <int D>
static int launch_attn(const void* Q, const void* K, const void* V, void* O, int N, int hw, int ldq, int ldk, int ldv,
                       int ldo, float scale, hipStream_t st) {
    constexpr int bq = 64;
    if (bq == 128) return launch_attn_bq<D, 128>(Q, K, V, O, N, hw, ldq, ldk, ldv, ldo, scale, st);
    return launch_attn_bq<D, 64>(Q, K, V, O, N, hw, ldq, ldk, ldv, ldo, scale, st);
}

}  // namespace rbvae

using namespace rbvae;

extern "C" {

int rbvae_attention_ok(int dtype, int hw, int C) {
    return dtype == RBVAE_BF16 && hw > 0 && hw % AT_BK == 0 && (C == 64 || C == 128 || C == 256 || C == 512);
}

int rbvae_attention(int dtype, const void* Q, const void* K, const void* V, void* O, int N, int hw, int C, int ldq,
                    int ldk, int ldv, int ldo, float scale, void* stream) {
    RBVAE_CHECK_ARG(Q && K && V && O && N > 0, "attention: bad arguments");
    RBVAE_CHECK_ARG(rbvae_attention_ok(dtype, hw, C), "attention: dtype %d, %d tokens, %d channels outside the kernel's "
                    "range (bf16, tokens %% 32 == 0, channels 64/128/256/512)", dtype, hw, C);
    RBVAE_CHECK_ARG(ldq >= C && ldk >= C && ldv >= C && ldo >= C && ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0,
                    "attention: leading dimensions");
    RBVAE_CHECK_ARG(((uintptr_t)Q | (uintptr_t)K | (uintptr_t)V) % 16 == 0, "attention: pointers must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    int rc;
    switch (C) {
        case 64: rc = launch_attn<64>(Q, K, V, O, N, hw, ldq, ldk, ldv, ldo, scale, st); break;
        case 128: rc = launch_attn<128>(Q, K, V, O, N, hw, ldq, ldk, ldv, ldo, scale, st); break;
        case 256: rc = launch_attn<256>(Q, K, V, O, N, hw, ldq, ldk, ldv, ldo, scale, st); break;
        default:
            // the prefetching form from four key tiles up (its prologue has three half-steps in flight)
            rc = hw >= 4 * AT_BK ? launch_attn_db(Q, K, V, O, N, hw, ldq, ldk, ldv, ldo, scale, st)
                                 : launch_attn<512>(Q, K, V, O, N, hw, ldq, ldk, ldv, ldo, scale, st);
            break;
    }
    if (rc != RBVAE_OK) return rc;
    RBVAE_CHECK_LAUNCH("attention");
    return RBVAE_OK;
}

}  // extern "C"
