// The last ConvTranspose2d(C1 -> Cout <= 4, k3 s2 p1 op1) + Sigmoid + recon_loss of the decoder
// (percep_RBVAE_model.py:82-83, percep_RBVAE_train.py:32-33) as ONE kernel, bf16 storage:
//
//   a workgroup takes an 8 x 16 block of INPUT pixels plus a one-pixel halo below / to the right (9 x 17), stages
//   their C1-channel rows and the per-tap product matrix V[(tap, co)][C1] in LDS (LDS-DMA, swizzled 128-byte
//   slices), forms Y[pixel][(tap, co)] on the matrix cores (f32, kept in LDS), and gathers the 16 x 32 block of
//   OUTPUT pixels from it: + bias, sigmoid, x_recon (NCHW f32), squared error against the target, d(loss)/d(pre)
//   (NHWC f32) and their per-workgroup sums.
//
// It replaces the product GEMM (Y to HBM as bf16) + the col2im pass of the two-kernel path; same sums in the same
// tap order, with Y in f32 instead of bf16.
#include "common.h"
#include <stdlib.h>

namespace rbvae {

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

struct DlFrameMap { int d1, d2; long s0, s1, s2; };
__device__ __forceinline__ long dl_frame_off(const DlFrameMap& f, unsigned n) {
    if (f.d1 == 0) return (long)n * f.s2;
    const unsigned a = n / (unsigned)f.d1, r = n - a * (unsigned)f.d1;
    const unsigned b = r / (unsigned)f.d2, c = r - b * (unsigned)f.d2;
    return (long)a * f.s0 + (long)b * f.s1 + (long)c * f.s2;
}

struct DlArgs {
    const unsigned char* D2;     // [N*IH*IW][C1] bf16
    const unsigned char* V;      // [NYP][C1] bf16, row = tap*Cout + co
    const float* bias;           // [Cout]
    const unsigned char* zero;   // >= 16 zero bytes
    float* xr;                   // [N][Cout][OH][OW]
    const float* target;         // frames through tfm (or null)
    DlFrameMap tfm;
    float* ws;                   // [gridDim.x] squared-error sums, then [gridDim.x][4] column sums of dpre (or null)
    float* dpre;                 // [N][OH][OW][Cout] or null
    float gscale;
    int N, IH, IW, C1, NYP, Cout;
};

#ifndef DL_PIPE
#define DL_PIPE 1        // 1: double buffer over single slices (slice s+1 in flight under slice s's MFMAs); 0: two slices staged
                         // together, then multiplied
#endif
#ifndef DL_SL
#define DL_SL 2          // 128-byte channel slices resident at a time (build switch: 1 with DL_PIPE 0 -> five workgroups per CU)
#endif
constexpr int DL_TA = 8, DL_TB = 16;                 // input block
constexpr int DL_HA = DL_TA + 1, DL_HB = DL_TB + 1;  // with halo
constexpr int DL_PIX = DL_HA * DL_HB;                // 153
constexpr int DL_MROWS = 160;                        // padded to MFMA tiles
constexpr int DL_WROWS = 48;
constexpr int DL_YP = 37;                            // Y row pitch (floats)

__device__ __forceinline__ void dl_glds16(const void* g, void* lds) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}

template <bool ONE>
__global__ __launch_bounds__(256, (DL_SL == 1 ? 5 : 3)) void deconv_last_fused_k(const DlArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int KS = p.C1 >> 6;                                        // 128-byte slices per row
    // Two 128-byte channel slices are resident at a time, not all of C1, and the f32 products Y take the slice buffers'
    // place once the last slice is multiplied: 52 KB per workgroup, so THREE share a CU and one's loads run under the
    // others' MFMAs and gather phases.  (All of C1 resident: one workgroup per CU, every phase exposed, slower than the
    // two-kernel path.  Y beside the slices, 76 KB, two per CU: 93 us at native 4x88x160 against 74 -- and the same 93 with
    // a ring of three slices at two per CU: it is the number of workgroups whose phases interleave that counts, not the
    // slices in flight; one buffer, unpipelined, five per CU: 79.)
    constexpr int SL = DL_SL;
    // a 64-channel layer (cfg 3) has ONE slice: one buffer, 26 KB, five workgroups per CU (94 registers)
    const int NBUF = KS > 1 ? SL : 1;
    unsigned char* s_pix = smem;                                     // [NBUF][DL_MROWS][128]
    unsigned char* s_wts = s_pix + (size_t)NBUF * DL_MROWS * 128;    // [NBUF][DL_WROWS][128]
    float* s_y = (float*)smem;                                       // [DL_MROWS][DL_YP], over the slice buffers once they are dead
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int OH = 2 * p.IH, OW = 2 * p.IW;
    const int tb_n = (p.IW + DL_TB - 1) / DL_TB, ta_n = (p.IH + DL_TA - 1) / DL_TA;
    unsigned blk = blockIdx.x;
    const int tbi = blk % (unsigned)tb_n; blk /= (unsigned)tb_n;
    const int tai = blk % (unsigned)ta_n;
    const int n = blk / (unsigned)ta_n;
    const int a0 = tai * DL_TA, b0 = tbi * DL_TB;

    // the epilogue's operands that depend on nothing (targets, bias) are asked for first: their latency hides behind the
    // staging and the MFMAs instead of sitting, one channel after the other, in front of the stores
    constexpr int EPI = (2 * DL_TA * 2 * DL_TB) / 256;                // output pixels per thread
    const int OHW = OH * OW;
    float tg[EPI][4], bz[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) bz[c] = (p.bias && c < p.Cout) ? p.bias[c] : 0.f;
#pragma unroll
    for (int it = 0; it < EPI; ++it) {
        const int q = tid + 256 * it;
        const int oh = 2 * a0 + q / (2 * DL_TB), ow = 2 * b0 + q % (2 * DL_TB);
        const bool ok = p.target && oh < OH && ow < OW;
        const float* tp = p.target + (ok ? dl_frame_off(p.tfm, n) + (long)oh * OW + ow : 0);
#pragma unroll
        for (int c = 0; c < 4; ++c) tg[it][c] = (ok && c < p.Cout) ? tp[(long)c * OHW] : 0.f;
    }

    const int srow = lane >> 3, schunk = lane & 7;
    const int fi = lane & 15, fg = lane >> 4;
    const int fsw = (fi >> 1) & 7;
    f32x4_t acc[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#if DL_PIPE
    // Double buffer over single 128-byte channel slices (the same LDS as two resident slices): slice s+1 is in flight
    // while slice s is multiplied, so the memory system is never idle between the staging rounds.
    static_assert(SL == 2, "the two slice buffers of the pipelined form");
    constexpr int PIX_I = DL_MROWS / 8, WTS_I = DL_WROWS / 8;         // LDS-DMA instructions of one slice
    // The pieces a wave stages are the same for every slice but for the slice's 128-byte offset: source (with its chunk
    // swizzle; the zero page for padding rows and pixels outside the image) and LDS destination are worked out ONCE per
    // block -- per slice a piece is then two adds and the LDS-DMA (the row / 17 division, bounds tests and 64-bit address
    // arithmetic per piece and slice were a third of this kernel's vector instructions; it is half VALU-busy).
    // (a one-slice layer -- 64 channels, cfg 3 -- has nothing to reuse them for and needs its 94 registers for five workgroups
    // per CU: ONE compiles the inline form)
    constexpr int NPIECE = ONE ? 1 : (PIX_I + WTS_I + 3) / 4;
    const unsigned char* psrc[NPIECE];
    unsigned pinc[NPIECE], pdst[NPIECE], pbuf[NPIECE];
#pragma unroll
    for (int i = 0; i < (ONE ? 0 : NPIECE); ++i) {
        const int q = w + 4 * i;
        const bool is_w = q >= PIX_I;
        const int r = (is_w ? q - PIX_I : q) * 8 + srow;
        const int sw = (schunk ^ ((r >> 1) & 7)) * 16;
        const unsigned char* src = p.zero;
        unsigned inc = 0;
        if (is_w) {
            if (r < p.NYP) { src = p.V + ((size_t)r * p.C1) * 2 + sw; inc = 128; }
        } else if (r < DL_PIX) {
            const int la = r / DL_HB, lb = r - la * DL_HB;
            const int a = a0 + la, b = b0 + lb;
            if (a < p.IH && b < p.IW) { src = p.D2 + ((size_t)((n * p.IH + a) * p.IW + b) * p.C1) * 2 + sw; inc = 128; }
        }
        psrc[i] = src; pinc[i] = inc;
        pdst[i] = (unsigned)((is_w ? (size_t)NBUF * DL_MROWS * 128 : 0) + (size_t)(r - srow) * 128);
        pbuf[i] = is_w ? DL_WROWS * 128 : DL_MROWS * 128;
    }
    auto stage_inline = [&](int s, int buf) __attribute__((always_inline)) {
        for (int q = w; q < PIX_I + WTS_I; q += 4) {
            const bool is_w = q >= PIX_I;
            const int r = (is_w ? q - PIX_I : q) * 8 + srow;
            const int sw = (schunk ^ ((r >> 1) & 7)) * 16;
            const unsigned char* src = p.zero;
            if (is_w) {
                if (r < p.NYP) src = p.V + ((size_t)r * p.C1) * 2 + s * 128 + sw;
            } else if (r < DL_PIX) {
                const int la = r / DL_HB, lb = r - la * DL_HB;
                const int a = a0 + la, b = b0 + lb;
                if (a < p.IH && b < p.IW) src = p.D2 + ((size_t)((n * p.IH + a) * p.IW + b) * p.C1) * 2 + s * 128 + sw;
            }
            unsigned char* dst = (is_w ? s_wts + (size_t)buf * DL_WROWS * 128 : s_pix + (size_t)buf * DL_MROWS * 128) +
                                 (size_t)(r - srow) * 128;
            dl_glds16(src, dst);
        }
    };
    auto stage = [&](int s, int buf) __attribute__((always_inline)) {
        if constexpr (ONE) stage_inline(s, buf);
        else {
#pragma unroll
            for (int i = 0; i < NPIECE; ++i)
                if (w + 4 * i < PIX_I + WTS_I) dl_glds16(psrc[i] + (size_t)(pinc[i] * (unsigned)s), smem + pdst[i] + pbuf[i] * (unsigned)buf);
        }
    };
    // this wave's LDS-DMA count per slice: q = w, w + 4, ... < 26
    constexpr int PER0 = (PIX_I + WTS_I + 3) / 4, PER1 = (PIX_I + WTS_I) / 4;
    static_assert((PIX_I + WTS_I) % 4 == 2, "waves 0, 1 issue PER0 pieces, waves 2, 3 PER1");
    // The fragment reads are inline asm behind explicit waits: a C++ LDS read (or __syncthreads) after a
    // global_load_lds makes the compiler drain EVERY LDS-DMA first (s_waitcnt vmcnt(0)), slice s+1 included.
    const unsigned lds_pix = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)s_pix;
    const unsigned lds_wts = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)s_wts;
    stage(0, 0);
    for (int s = 0; s < KS; ++s) {
        const int buf = s & 1;
        if (s + 1 < KS) {
            stage(s + 1, buf ^ 1);
            // slice s landed for this wave (its PER pieces of slice s+1 stay in flight), then the workgroup barrier
            if (w < 2) asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::"n"(PER0) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::"n"(PER1) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int ch = ((4 * kk + fg) ^ fsw) * 16;
            u32x4_t wf[3], pf[3];
#pragma unroll
            for (int rt = 0; rt < 3; ++rt) {
                const unsigned ad = lds_wts + (unsigned)((buf * DL_WROWS + rt * 16 + fi) * 128 + ch);
                asm volatile("ds_read_b128 %0, %1" : "=v"(wf[rt]) : "v"(ad));
            }
#pragma unroll
            for (int pi = 0; pi < 3; ++pi) {
                const int pt = min(w + 4 * pi, DL_MROWS / 16 - 1);           // waves 2, 3 have no third tile: read, not used
                const unsigned ad = lds_pix + (unsigned)((buf * DL_MROWS + pt * 16 + fi) * 128 + ch);
                asm volatile("ds_read_b128 %0, %1" : "=v"(pf[pi]) : "v"(ad));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wf[0]), "+v"(wf[1]), "+v"(wf[2]), "+v"(pf[0]), "+v"(pf[1]), "+v"(pf[2]));
#pragma unroll
            for (int pi = 0; pi < 3; ++pi) {
                if (w + 4 * pi < DL_MROWS / 16) {
#pragma unroll
                    for (int rt = 0; rt < 3; ++rt)
                        acc[pi][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8_t*)&wf[rt],
                                                                              *(const bf16x8_t*)&pf[pi], acc[pi][rt], 0, 0, 0);
                }
            }
        }
        // everyone is done reading this buffer (every read above was waited for): slice s+2 may land in it
        if (s + 2 < KS) asm volatile("s_barrier" ::: "memory");
    }
#else
    for (int s0 = 0; s0 < KS; s0 += SL) {
        const int ns = min(SL, KS - s0);
        if (s0) __syncthreads();                                     // everyone is done reading the previous slices
        // ---- stage: one LDS-DMA instruction = 8 rows x 128 B of one slice
        const int pix_instr = (DL_MROWS / 8) * ns, wts_instr = (DL_WROWS / 8) * ns;
        for (int q = w; q < pix_instr + wts_instr; q += 4) {
            const bool is_w = q >= pix_instr;
            const int qq = is_w ? q - pix_instr : q;
            const int rows8 = is_w ? DL_WROWS / 8 : DL_MROWS / 8;
            const int sl = qq / rows8, r = (qq - sl * rows8) * 8 + srow;
            const int s = s0 + sl;
            const int sw = (schunk ^ ((r >> 1) & 7)) * 16;
            const unsigned char* src = p.zero;
            if (is_w) {
                if (r < p.NYP) src = p.V + ((size_t)r * p.C1) * 2 + s * 128 + sw;
            } else if (r < DL_PIX) {
                const int la = r / DL_HB, lb = r - la * DL_HB;
                const int a = a0 + la, b = b0 + lb;
                if (a < p.IH && b < p.IW) src = p.D2 + ((size_t)((n * p.IH + a) * p.IW + b) * p.C1) * 2 + s * 128 + sw;
            }
            unsigned char* dst = (is_w ? s_wts + (size_t)sl * DL_WROWS * 128 : s_pix + (size_t)sl * DL_MROWS * 128) +
                                 (size_t)(r - srow) * 128;
            dl_glds16(src, dst);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // ---- Y += pixels x V^T: wave w takes pixel tiles w, w+4, w+8 and all three (tap, co) tiles
        for (int sl = 0; sl < ns; ++sl) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int ch = ((4 * kk + fg) ^ fsw) * 16;
                u32x4_t wf[3];
#pragma unroll
                for (int rt = 0; rt < 3; ++rt)
                    wf[rt] = *(const u32x4_t*)(s_wts + ((size_t)sl * DL_WROWS + rt * 16 + fi) * 128 + ch);
#pragma unroll
                for (int pi = 0; pi < 3; ++pi) {
                    const int pt = w + 4 * pi;
                    if (pt < DL_MROWS / 16) {
                        const u32x4_t pf = *(const u32x4_t*)(s_pix + ((size_t)sl * DL_MROWS + pt * 16 + fi) * 128 + ch);
#pragma unroll
                        for (int rt = 0; rt < 3; ++rt)
                            acc[pi][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8_t*)&wf[rt],
                                                                                  *(const bf16x8_t*)&pf, acc[pi][rt], 0, 0, 0);
                    }
                }
            }
        }
    }
#endif
    // the prefetched epilogue operands landed long ago; saying so here keeps the compiler from waiting for them (and with
    // them for the x_recon stores) in the middle of the epilogue
#pragma unroll
    for (int it = 0; it < EPI; ++it)
#pragma unroll
        for (int c = 0; c < 4; ++c) asm volatile("" : "+v"(tg[it][c]));
#pragma unroll
    for (int c = 0; c < 4; ++c) asm volatile("" : "+v"(bz[c]));
    __syncthreads();                     // every wave has read its last fragments: the products go where the slices were
    // D[(tap,co) row 4g + r][pixel i]
#pragma unroll
    for (int pi = 0; pi < 3; ++pi) {
        const int pt = w + 4 * pi;
        if (pt < DL_MROWS / 16) {
#pragma unroll
            for (int rt = 0; rt < 3; ++rt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int wr = rt * 16 + 4 * fg + r;
                    if (wr < 36) s_y[(pt * 16 + fi) * DL_YP + wr] = acc[pi][rt][r];
                }
        }
    }
    __syncthreads();

    // ---- output block 16 x 32: gather the taps, bias, sigmoid, losses
    __shared__ float red5[4][5];
    const int Cout = p.Cout;
    const long plane = (long)OH * OW;
    float sse = 0.f, dsum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int it = 0; it < EPI; ++it) {
        const int q = tid + 256 * it;
        const int oy = q / (2 * DL_TB), ox = q - oy * (2 * DL_TB);
        const int oh = 2 * a0 + oy, ow = 2 * b0 + ox;
        if (oh >= OH || ow >= OW) continue;
        float v[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = bz[c];
        for (int kh = (oh + 1) & 1; kh < 3; kh += 2) {
            const int a = (oh + 1 - kh) >> 1;
            if (a < 0 || a >= p.IH) continue;
            for (int kw = (ow + 1) & 1; kw < 3; kw += 2) {
                const int b = (ow + 1 - kw) >> 1;
                if (b < 0 || b >= p.IW) continue;
                const float* yp = s_y + ((a - a0) * DL_HB + (b - b0)) * DL_YP + (kh * 3 + kw) * Cout;
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (c < Cout) v[c] += yp[c];
            }
        }
        const size_t xo = (size_t)n * Cout * plane + (size_t)oh * OW + ow;
        float d4[4] = {0.f, 0.f, 0.f, 0.f}, sg[4];
#pragma unroll
        for (int c = 0; c < 4; ++c)     // hardware exp2 / rcp (about 1 ulp each; this kernel is the bf16-storage path)
            sg[c] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v[c]));
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (c < Cout) p.xr[xo + (size_t)c * plane] = sg[c];
        if (p.target) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (c >= Cout) break;
                const float d = sg[c] - tg[it][c];
                sse += d * d;
                d4[c] = p.gscale * d * sg[c] * (1.f - sg[c]);
            }
        }
        if (p.target && p.dpre) {
            float* dp = p.dpre + ((size_t)(n * OH + oh) * OW + ow) * Cout;
            if (Cout == 4) *(float4*)dp = make_float4(d4[0], d4[1], d4[2], d4[3]);
            else
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (c < Cout) dp[c] = d4[c];
#pragma unroll
            for (int c = 0; c < 4; ++c) dsum[c] += d4[c];
        }
    }
    if (p.ws) {
        float vals[5] = {sse, dsum[0], dsum[1], dsum[2], dsum[3]};
#pragma unroll
        for (int k = 0; k < 5; ++k) vals[k] = wave_sum(vals[k]);
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < 5; ++k) red5[w][k] = vals[k];
        }
        __syncthreads();
        if (tid < 5) {
            const float t = ((red5[0][tid] + red5[1][tid]) + red5[2][tid]) + red5[3][tid];
            if (tid == 0) p.ws[blockIdx.x] = t;
            else if (p.dpre) p.ws[gridDim.x + 4 * blockIdx.x + (tid - 1)] = t;
        }
    }
}

static int dl_blocks(int N, int IH, int IW) { return N * cdiv(IH, DL_TA) * cdiv(IW, DL_TB); }
static size_t dl_lds(int C1) {
    const size_t slices = (size_t)(C1 > 64 ? DL_SL : 1) * (DL_MROWS + DL_WROWS) * 128, y = (size_t)DL_MROWS * DL_YP * 4;
    return slices > y ? slices : y;
}

}  // namespace rbvae

using namespace rbvae;

/* workgroups of the launch (= partial sums in ws) when the fused kernel covers the shape, else 0 */
extern "C" int rbvae_deconv_last_fused_parts(int dtype, int N, int IH, int IW, int C1, int Cout) {
    if (dtype != RBVAE_BF16 || Cout < 1 || Cout > 4 || C1 % 64 || C1 < 64 || dl_lds(C1) > 160 * 1024 - 256) return 0;
    if ((long)N * 2 * IH * 2 * IW * Cout >= (1l << 31) || (long)N * IH * IW * C1 >= (1l << 31)) return 0;
    return dl_blocks(N, IH, IW);
}

extern "C" int rbvae_deconv_last_fused(int dtype, const void* D2, const void* V, int NYP, const float* bias,
                                       const void* zero_page, int N, int IH, int IW, int C1, int Cout, float* xr,
                                       const float* target, int fd1, int fd2, long fs0, long fs1, long fs2, float* ws,
                                       float* dpre, float gscale, void* stream) {
    RBVAE_CHECK_ARG(D2 && V && zero_page && xr, "deconv_last_fused: null pointer");
    RBVAE_CHECK_ARG(rbvae_deconv_last_fused_parts(dtype, N, IH, IW, C1, Cout) > 0,
                    "deconv_last_fused: shape outside the fused kernel (bf16, Cout <= 4, C1 %% 64 == 0): N=%d %dx%d C1=%d Cout=%d",
                    N, IH, IW, C1, Cout);
    RBVAE_CHECK_ARG(NYP >= 9 * Cout && NYP <= DL_WROWS, "deconv_last_fused: NYP=%d", NYP);
    RBVAE_CHECK_ARG(!dpre || (target && ws), "deconv_last_fused: dpre needs target and ws");
    RBVAE_CHECK_ARG(((uintptr_t)D2 | (uintptr_t)V | (uintptr_t)zero_page | (uintptr_t)dpre) % 16 == 0,
                    "deconv_last_fused: pointers must be 16-byte aligned");
    DlArgs a;
    a.D2 = (const unsigned char*)D2; a.V = (const unsigned char*)V; a.bias = bias; a.zero = (const unsigned char*)zero_page;
    a.xr = xr; a.target = target; a.tfm = DlFrameMap{fd1, fd2, fs0, fs1, fs2}; a.ws = target ? ws : nullptr; a.dpre = dpre;
    a.gscale = gscale; a.N = N; a.IH = IH; a.IW = IW; a.C1 = C1; a.NYP = NYP; a.Cout = Cout;
    const size_t lds = dl_lds(C1);
    static size_t attr_lds[2] = {0, 0};
    const int one = C1 == 64;
    const void* kern = one ? (const void*)deconv_last_fused_k<true> : (const void*)deconv_last_fused_k<false>;
    if (lds > attr_lds[one]) {               // dynamic + the kernel's 80 static bytes must stay within the 160 KB of a CU
        hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return fail(RBVAE_E_LAUNCH, "deconv_last_fused: %s (dynamic LDS %zu)", hipGetErrorString(e), lds);
        attr_lds[one] = lds;
    }
    if (one) hipLaunchKernelGGL(deconv_last_fused_k<true>, dim3(dl_blocks(N, IH, IW)), dim3(256), lds, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(deconv_last_fused_k<false>, dim3(dl_blocks(N, IH, IW)), dim3(256), lds, (hipStream_t)stream, a);
    RBVAE_CHECK_LAUNCH("deconv_last_fused");
    return RBVAE_OK;
}
