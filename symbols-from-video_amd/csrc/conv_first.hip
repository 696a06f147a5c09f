// The two HBM-bound 3x3 stride-2 convolutions at the 3/4-channel ends of the CNNs, each as ONE kernel (bf16 storage):
//
//   MODE 0  the encoder's first Conv2d(Cin <= 4 -> Nout <= 256) + bias + ReLU + Dropout (percep_RBVAE_model.py:51-53),
//           input frames f32 NCHW through the frame map;
//   MODE 1  the input gradient of the decoder's last ConvTranspose2d (autograd of :82) = the same convolution of
//           d(loss)/d(pre-sigmoid) [N][H][W][Cin] f32 (NHWC, as rbvae_deconv_last_fused leaves it) with the deconv weight,
//           gated by the stored ReLU/dropout output of the layer below and column-summed for that layer's bias gradient.
//
// A workgroup takes an 8 x 16 block of OUTPUT pixels of one frame, loads the 17 x 33 input patch with coalesced loads,
// builds the im2col rows [128][64] (column (kh*3+kw)*Cin + ci, zero padded) as the swizzled LDS image the MFMA fragments
// read -- and writes them to `col`, which the weight gradient reads later -- stages W [Nout][64] by LDS-DMA, multiplies on
// the matrix cores and stores the epilogue's results straight from the accumulators, 32 bytes per lane and pixel.
//
// They replace rbvae_im2col(_frames) + the single-slice rbvae_gather_gemm (same arithmetic per element: one 64-deep MFMA
// chain; same dropout key and chunk indices; stored values are bit-identical).
#include "common.h"
#include <stdlib.h>

#ifndef CF_DBG      // timing experiments only: 1 no output stores, 2 no col stores, 3 stop after the staging, 4 no patch loads, 5 no gate reads
#define CF_DBG 0
#endif

namespace rbvae {

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

struct CfFrameMap { int d1, d2; long s0, s1, s2; };
__device__ __forceinline__ long cf_frame_off(const CfFrameMap& f, unsigned n) {
    if (f.d1 == 0) return (long)n * f.s2;
    const unsigned a = n / (unsigned)f.d1, r = n - a * (unsigned)f.d1;
    const unsigned b = r / (unsigned)f.d2, c = r - b * (unsigned)f.d2;
    return (long)a * f.s0 + (long)b * f.s1 + (long)c * f.s2;
}

struct CfArgs {
    const float* x;              // MODE 0: frames [Cin][IH][IW] f32 at cf_frame_off(fm, n); MODE 1: [N][IH][IW][Cin] f32
    CfFrameMap fm;
    const unsigned char* W;      // [Nout][64] bf16 (im2col column order, zero padded)
    const float* bias;           // [Nout] or null (MODE 0)
    const unsigned char* zero;   // >= 16 zero bytes
    unsigned char* col;          // [N*OH*OW][64] bf16 out
    unsigned char* out;          // [N*OH*OW][ldo] bf16 out
    const unsigned char* gate;   // MODE 1: [N*OH*OW][ldo] bf16, output kept where gate > 0
    float* colsum_ws;            // MODE 1: [blocks][Nout] column sums of the stored output, or null
    int N, Cin, IH, IW, OH, OW, Nout, ldo, relu, drop_mode;
    float scale;
    unsigned drop_thresh;
    unsigned long long seed;
    const unsigned long long* seed_dev;
};

constexpr int CF_TA = 8, CF_TB = 16;                          // output block
constexpr int CF_PA = 2 * CF_TA + 1, CF_PB = 2 * CF_TB + 1;   // input patch 17 x 33
constexpr int CF_PP = CF_PB + 1;                              // patch row pitch (pixels)

// patch offset of im2col column k = (kh*3+kw)*CIN + ci relative to the output pixel's corner; -1 = padding column.
// MODE 0 keeps the patch as [ci][row][col], MODE 1 as [row][col][ci] (the order the input arrives in).
template <int CIN, int MODE> struct CfOff {
    int v[64];
    constexpr CfOff() : v{} {
        for (int k = 0; k < 64; ++k) {
            const int t = k / CIN, ci = k % CIN, kh = t / 3, kw = t % 3;
            v[k] = k >= 9 * CIN ? -1 : MODE == 0 ? (ci * CF_PA + kh) * CF_PP + kw : (kh * CF_PP + kw) * CIN + ci;
        }
    }
};

__device__ __forceinline__ void cf_glds16(const void* g, void* lds) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}
template <int CTRL> __device__ __forceinline__ float cf_dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
// sum over the 16 lanes of a DPP row (lane & 15), every lane gets the total
__device__ __forceinline__ float cf_row_sum(float v) {
    v += cf_dpp<0x128>(v);     // row_ror:8
    v += cf_dpp<0x124>(v);     // row_ror:4
    v += cf_dpp<0x4E>(v);      // quad_perm [2,3,0,1]
    v += cf_dpp<0xB1>(v);      // quad_perm [1,0,3,2]
    return v;
}
__device__ __forceinline__ bool cf_bf16_pos(unsigned v16) { return (v16 - 1u) < 0x7f80u; }    // 0 < v <= +inf, NaN excluded

// NQ = 64-channel quads of the tile: 4 (Nout <= 256: wave = 64 pixels x one quad) or 1 (Nout <= 64, the cfg 3 widths: wave =
// 16 pixels x the one quad -- with the 256-wide tile three of four waves multiplied and staged zero weights and sat out the
// epilogue; 8 KB of weights instead of 32 and 16 accumulators per lane let four workgroups share a CU)
template <int CIN, int MODE, int NQ> __global__ __launch_bounds__(512, NQ == 1 ? 8 : 4) void conv_first_fused_k(const CfArgs p) {
    constexpr int MT = NQ == 4 ? 4 : 1;                                          // 16-pixel tiles per wave
    __shared__ __attribute__((aligned(16))) unsigned char s_a[128 * 128];        // im2col rows, swizzled chunks
    __shared__ __attribute__((aligned(16))) unsigned char s_b[64 * NQ * 128];    // weights in fragment-row order
    __shared__ float s_patch[4 * CF_PA * CF_PP];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int tb_n = (p.OW + CF_TB - 1) / CF_TB, ta_n = (p.OH + CF_TA - 1) / CF_TA;
    unsigned blk = blockIdx.x;
    const int tbi = blk % (unsigned)tb_n; blk /= (unsigned)tb_n;
    const int tai = blk % (unsigned)ta_n;
    const int n = blk / (unsigned)ta_n;
    const int oh0 = tai * CF_TA, ow0 = tbi * CF_TB;
    const int ih0 = 2 * oh0 - 1, iw0 = 2 * ow0 - 1;

    // ---- weights -> LDS by LDS-DMA: image row ct*16 + j holds channel 64*(ct/4) + 16*(j/4) + 4*(ct%4) + j%4, so that a
    // lane's accumulators of a tile quad are 16 consecutive channels (32 bytes per pixel, 128 per pixel and wave)
    {
        const int srow = lane >> 3, schunk = lane & 7;
        for (int q = w; q < 8 * NQ; q += 8) {                // 8 NQ instructions of 8 rows
            const int r = q * 8 + srow;
            const int ct = r >> 4, j = r & 15;
            const int ch = 64 * (ct >> 2) + 16 * (j >> 2) + 4 * (ct & 3) + (j & 3);
            const unsigned char* src = ch < p.Nout ? p.W + (size_t)ch * 128 + ((schunk ^ ((r >> 1) & 7)) * 16) : p.zero;
            cf_glds16(src, s_b + (size_t)(r - srow) * 128);
        }
    }
    // bias and dropout key early: their latency hides behind the patch loads, and the epilogue's stores are never waited on
    const int fi = lane & 15, fg = lane >> 4;
    const int wq = NQ == 4 ? (w & 3) : 0, wm = NQ == 4 ? (w >> 2) : w;
    const int col = 64 * wq + 16 * fg;
    const bool first = col < p.Nout, second = col + 8 < p.Nout;
    f32x4_t bz4[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        bz4[h] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        if (MODE == 0 && p.bias && col + 4 * h < p.Nout) bz4[h] = *(const f32x4_t*)(p.bias + col + 4 * h);
    }
    DropKey dkey{0u, 0u};
    int koff[8];
    static constexpr CfOff<CIN, MODE> otab{};
    // ---- input patch
    if constexpr (MODE == 0) {
        // s_patch row c*17 + r, column 0 = the halo column 2*ow0 - 1, columns 1 .. 32 = the 128-byte run from 2*ow0: one
        // instruction loads two rows (32 lanes each)
        const float* xf = p.x + cf_frame_off(p.fm, n);
        constexpr int R = CIN * CF_PA, PAIRS = (R + 1) / 2, PIT = (PAIRS + 7) / 8;
        float pv[PIT], hv = 0.f;
#pragma unroll
        for (int it = 0; it < PIT; ++it) {
            const int row = 2 * (w + 8 * it) + (lane >> 5);
            const int c = row / CF_PA, r = row - c * CF_PA;
            const int ih = ih0 + r, iw = iw0 + 1 + (lane & 31);
            pv[it] = 0.f;
            if (CF_DBG != 4 && row < R && ih >= 0 && ih < p.IH && iw < p.IW) pv[it] = xf[((size_t)c * p.IH + ih) * p.IW + iw];
        }
        if (tid < R) {
            const int c = tid / CF_PA, r = tid - c * CF_PA;
            const int ih = ih0 + r;
            if (CF_DBG != 4 && ih >= 0 && ih < p.IH && iw0 >= 0) hv = xf[((size_t)c * p.IH + ih) * p.IW + iw0];
        }
#pragma unroll
        for (int k8 = 0; k8 < 8; ++k8) koff[k8] = otab.v[(tid & 7) * 8 + k8];
        if (p.drop_mode == 1) dkey = drop_key(p.seed + (p.seed_dev ? p.seed_dev[0] * 0x9E3779B97F4A7C15ull : 0ull));
#pragma unroll
        for (int it = 0; it < PIT; ++it) {
            const int row = 2 * (w + 8 * it) + (lane >> 5);
            if (row < R) s_patch[row * CF_PP + 1 + (lane & 31)] = pv[it];
        }
        if (tid < R) s_patch[tid * CF_PP] = hv;
    } else {
        // [row][col][ci]: a patch row is one run of 33*CIN floats
        const float* xf = p.x + (size_t)n * p.IH * p.IW * CIN;
        constexpr int RUN = CF_PB * CIN, PN = CF_PA * RUN, PIT = (PN + 511) / 512;
        float pv[PIT];
#pragma unroll
        for (int it = 0; it < PIT; ++it) {
            const int i = it * 512 + tid;
            const int r = i / RUN, j = i - r * RUN;
            const int ih = ih0 + r, iw = iw0 + j / CIN;
            pv[it] = 0.f;
            if (CF_DBG != 4 && i < PN && ih >= 0 && ih < p.IH && iw >= 0 && iw < p.IW)
                pv[it] = xf[((long)ih * p.IW + iw0) * CIN + j];
        }
#pragma unroll
        for (int k8 = 0; k8 < 8; ++k8) koff[k8] = otab.v[(tid & 7) * 8 + k8];
#pragma unroll
        for (int it = 0; it < PIT; ++it) {
            const int i = it * 512 + tid;
            const int r = i / RUN, j = i - r * RUN;
            if (i < PN) s_patch[r * (CF_PP * CIN) + j] = pv[it];
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // weights (LDS-DMA), bias, key: everything this wave asked for
#pragma unroll
    for (int h = 0; h < 4; ++h) asm volatile("" : "+v"(bz4[h]));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    asm volatile("" : "+v"(dkey.k0), "+v"(dkey.k1));
    __syncthreads();
    // ---- im2col rows: 128 rows x 8 chunks of 8 columns (a thread keeps its chunk index over both rounds)
#pragma unroll
    for (int i0 = 0; i0 < 128 * 8; i0 += 512) {
        const int i = i0 + tid;
        const int r = i >> 3, c = i & 7;
        const int oy = r >> 4, ox = r & 15;
        const int corner = (2 * oy * CF_PP + 2 * ox) * (MODE == 0 ? 1 : CIN);
        unsigned short e[8];
#pragma unroll
        for (int k8 = 0; k8 < 8; ++k8) {
            const float v = koff[k8] >= 0 ? s_patch[corner + max(koff[k8], 0)] : 0.f;
            e[k8] = f32_to_bf16(v);
        }
        u32x4_t pk;
        pk[0] = (unsigned)e[0] | ((unsigned)e[1] << 16); pk[1] = (unsigned)e[2] | ((unsigned)e[3] << 16);
        pk[2] = (unsigned)e[4] | ((unsigned)e[5] << 16); pk[3] = (unsigned)e[6] | ((unsigned)e[7] << 16);
        *(u32x4_t*)(s_a + r * 128 + ((c ^ ((r >> 1) & 7)) * 16)) = pk;
        const int oh = oh0 + oy, ow = ow0 + ox;
        if (CF_DBG != 2 && p.col && oh < p.OH && ow < p.OW)     // col == null: the weight gradient rebuilds the rows (wgrad_first_k)
            *(u32x4_t*)(p.col + ((size_t)(n * p.OH + oh) * p.OW + ow) * 128 + c * 16) = pk;
    }
    __syncthreads();

    // ---- 128 x 256 x 64 on the matrix cores: wave w owns pixels 64*(w/4) .. +63 x the tile quad w%4 = channels 64*(w%4) .. +63
    if (CF_DBG == 3) return;
    const int fsw = (fi >> 1) & 7;
    f32x4_t acc[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int h = 0; h < 4; ++h) acc[mt][h] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int ch = ((4 * kk + fg) ^ fsw) * 16;
        u32x4_t wv[4], av[MT];
#pragma unroll
        for (int h = 0; h < 4; ++h) wv[h] = *(const u32x4_t*)(s_b + ((4 * wq + h) * 16 + fi) * 128 + ch);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) av[mt] = *(const u32x4_t*)(s_a + ((MT * wm + mt) * 16 + fi) * 128 + ch);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int h = 0; h < 4; ++h)
                acc[mt][h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8_t*)&wv[h], *(const bf16x8_t*)&av[mt],
                                                                     acc[mt][h], 0, 0, 0);
    }
    // ---- epilogue from registers: lane = pixel fi of tile mt, channels 64*wq + 16*fg .. +15 (two 16-byte chunks)
    if (MODE == 0 && !first) return;
    float csum[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) csum[e] = 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int r = (MT * wm + mt) * 16 + fi;
        const int oh = oh0 + (r >> 4), ow = ow0 + (r & 15);
        if (oh >= p.OH || ow >= p.OW || !first) continue;
        const size_t orow = (size_t)(n * p.OH + oh) * p.OW + ow;
        u32x4_t g4[2];
        if constexpr (MODE == 1) {
#if CF_DBG == 5
            g4[0] = u32x4_t{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}; g4[1] = g4[0];   // timing: no gate reads
#else
            g4[0] = *(const u32x4_t*)(p.gate + (orow * p.ldo + col) * 2);
            g4[1] = second ? *(const u32x4_t*)(p.gate + (orow * p.ldo + col + 8) * 2) : u32x4_t{0u, 0u, 0u, 0u};
#endif
        }
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if (half == 1 && !second) break;
            u32x4_t val;
            if constexpr (MODE == 0) {
                // packed pairs: (acc + bias) * scale, one rounding to bf16 per pair, ReLU on the 16-bit patterns (a negative
                // bf16 is a negative int16; max(x, 0) * scale with scale > 0 rounds to the same values), dropout on the
                // words -- 57 vector instructions per 8 elements where the element-wise form took about 90; this epilogue,
                // not HBM, is what the first conv waits for (no stores at all: 66 of 80 us at native 4x88x160)
                typedef float v2f __attribute__((ext_vector_type(2)));
                typedef __bf16 v2b __attribute__((ext_vector_type(2)));
                typedef short v2s __attribute__((ext_vector_type(2)));
                const short fl = p.relu ? (short)0 : (short)-32768;
                const v2s floor2 = {fl, fl};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const f32x4_t a4 = acc[mt][2 * half + (e >> 1)], b4 = bz4[2 * half + (e >> 1)];
                    v2f x = v2f{a4[2 * (e & 1)], a4[2 * (e & 1) + 1]} + v2f{b4[2 * (e & 1)], b4[2 * (e & 1) + 1]};
                    x = x * p.scale;
                    const v2b r = __builtin_convertvector(x, v2b);
                    const v2s m = __builtin_elementwise_max(*(const v2s*)&r, floor2);     // floor -32768: every value passes
                    val[e] = *(const unsigned*)&m;
                }
                if (p.drop_mode == 1)
                    drop_chunk_zero_b16<8>(drop_run(dkey, (unsigned long long)orow * p.Nout + col + 8 * half), p.drop_thresh >> 16,
                                           (unsigned*)&val);
            } else {
                // the gate in packed pairs too: a bf16 pattern g is > 0 (NaN excluded) iff g - 1 < 0x7f80 as unsigned 16-bit
                // numbers; keep = min(sat(0x7f80 - (g - 1)), 1), and the rounded product is multiplied by it
                typedef float v2f __attribute__((ext_vector_type(2)));
                typedef __bf16 v2b __attribute__((ext_vector_type(2)));
                const unsigned one2 = 0x00010001u, lim2 = 0x7f807f80u;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const f32x4_t a4 = acc[mt][2 * half + (e >> 1)];
                    const v2f x = v2f{a4[2 * (e & 1)], a4[2 * (e & 1) + 1]} * p.scale;
                    const v2b r = __builtin_convertvector(x, v2b);
                    unsigned v = *(const unsigned*)&r, a, d, k;
                    asm("v_pk_sub_u16 %0, %1, %2" : "=v"(a) : "v"(g4[half][e]), "v"(one2));
                    asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(d) : "v"(lim2), "v"(a));
                    asm("v_pk_min_u16 %0, %1, %2" : "=v"(k) : "v"(d), "v"(one2));
                    asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(v) : "v"(v), "v"(k));
                    val[e] = v;
                }
            }
            if (CF_DBG != 1 || val[0] == 0x12345678u) *(u32x4_t*)(p.out + (orow * p.ldo + col + 8 * half) * 2) = val;
            if constexpr (MODE == 1) {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    csum[8 * half + e] += __uint_as_float((e & 1) ? (val[e >> 1] & 0xffff0000u) : (val[e >> 1] << 16));
            }
        }
    }
    if constexpr (MODE == 1) {
        if (p.colsum_ws) {
            // bias gradient of the layer below: column sums of this block's stored values, pixels then wave halves
            float* red = s_patch;                              // [pixel groups][64 NQ]; the patch is dead since the second barrier
            constexpr int RW = 64 * NQ, NGRP = NQ == 4 ? 2 : 8;
#pragma unroll
            for (int e = 0; e < 16; ++e) csum[e] = cf_row_sum(csum[e]);
            if (fi == 0) {
#pragma unroll
                for (int e = 0; e < 16; ++e) red[wm * RW + col + e] = csum[e];
            }
            __syncthreads();
            if (tid < p.Nout) {
                float t = red[tid];
#pragma unroll
                for (int g2 = 1; g2 < NGRP; ++g2) t += red[g2 * RW + tid];      // fixed order
                p.colsum_ws[(size_t)blockIdx.x * p.Nout + tid] = t;
            }
        }
    }
}


// ---- weight gradient of the two 3/4-channel ends WITHOUT the im2col rows in HBM -------------------------------------------
//   dW[ks][co][k] = sum_{p in K-slice ks} dY[p][co] * col(x)[p][k],   k = (kh*3+kw)*CIN + ci  (zero padded to 64)
// MODE 0: the first Conv2d's weight (x = the input frames; dY = the gradient at its output); MODE 1: the last
// ConvTranspose2d's (x = d(loss)/d(pre-sigmoid) NHWC f32; dY = the stored activation in front of it) -- the sums
// rbvae_wgrad_gemm forms from the [rows][64] im2col rows that conv_first_fused_k writes (128 bytes per pixel for 12-16 bytes
// of image: on 256 x 256 frames 268 MB written and 268 MB re-read per weight).  Here a workgroup walks 8 x 16 blocks of
// output pixels: the block's input patch is prefetched into registers one block ahead, the im2col rows are rebuilt in
// LDS (pixel-major 128-byte rows: the reduction index is the row), the [128 px][64 co] tile of dY arrives by LDS-DMA one
// block ahead, fragments by ds_read_b64_tr_b16.  Tile 64 co x 64 k, 2 sub-tiles per wave; two workgroups per CU.
struct WfArgs {
    const float* x; CfFrameMap fm;
    const unsigned char* dY;     // [N*OH*OW][ldy] bf16
    const unsigned char* zero;
    float* dW;                   // [ksplit][Nout][64] f32
    int N, Cin, IH, IW, OH, OW, Nout, ldy, ksplit, per, nblk;
};
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
__device__ __forceinline__ int wf_swz(int row) { return (((row >> 1) & 1) | (((row >> 3) & 1) << 1)) << 1; }   // tr_swz<128> (wgrad_gemm.hip)
__device__ __forceinline__ void wf_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int CIN, int MODE> __global__ __launch_bounds__(512, 4) void wgrad_first_k(const WfArgs p) {
    __shared__ __attribute__((aligned(16))) unsigned char s_col[128 * 128];       // im2col rows [px][64 k], tr-swizzled
    __shared__ __attribute__((aligned(16))) unsigned char s_dy[2][128 * 128];     // dY tile [px][64 co], tr-swizzled, double-buffered
    __shared__ float s_patch[4 * CF_PA * CF_PP];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nco = p.Nout >> 6;
    const int cot = blockIdx.x % nco, ks = blockIdx.x / nco;
    const int co0 = cot * 64;
    const int tb_n = (p.OW + CF_TB - 1) / CF_TB, ta_n = (p.OH + CF_TA - 1) / CF_TA;
    const int blk0 = ks * p.per, blk1 = min(blk0 + p.per, p.nblk);
    static constexpr CfOff<CIN, MODE> otab{};
    int koff[8];
#pragma unroll
    for (int k8 = 0; k8 < 8; ++k8) koff[k8] = otab.v[(tid & 7) * 8 + k8];

    // patch of block blk -> registers (the loads of conv_first_fused_k)
    constexpr int R0 = CIN * CF_PA, PIT0 = ((R0 + 1) / 2 + 7) / 8;
    constexpr int RUN = CF_PB * CIN, PN = CF_PA * RUN, PIT1 = (PN + 511) / 512;
    constexpr int PIT = MODE == 0 ? PIT0 : PIT1;
    float pv[PIT], hv = 0.f;
    auto load_patch = [&](int blk) {
        unsigned b = blk;
        const int tbi = b % (unsigned)tb_n; b /= (unsigned)tb_n;
        const int tai = b % (unsigned)ta_n;
        const int n = b / (unsigned)ta_n;
        const int ih0 = 2 * tai * CF_TA - 1, iw0 = 2 * tbi * CF_TB - 1;
        if constexpr (MODE == 0) {
            const float* xf = p.x + cf_frame_off(p.fm, n);
#pragma unroll
            for (int it = 0; it < PIT; ++it) {
                const int row = 2 * (w + 8 * it) + (lane >> 5);
                const int c = row / CF_PA, r = row - c * CF_PA;
                const int ih = ih0 + r, iw = iw0 + 1 + (lane & 31);
                pv[it] = 0.f;
                if (row < R0 && ih >= 0 && ih < p.IH && iw < p.IW) pv[it] = xf[((size_t)c * p.IH + ih) * p.IW + iw];
            }
            hv = 0.f;
            if (tid < R0) {
                const int c = tid / CF_PA, r = tid - c * CF_PA;
                const int ih = ih0 + r;
                if (ih >= 0 && ih < p.IH && iw0 >= 0) hv = xf[((size_t)c * p.IH + ih) * p.IW + iw0];
            }
        } else {
            const float* xf = p.x + (size_t)n * p.IH * p.IW * CIN;
#pragma unroll
            for (int it = 0; it < PIT; ++it) {
                const int i = it * 512 + tid;
                const int r = i / RUN, j = i - r * RUN;
                const int ih = ih0 + r, iw = iw0 + j / CIN;
                pv[it] = 0.f;
                if (i < PN && ih >= 0 && ih < p.IH && iw >= 0 && iw < p.IW) pv[it] = xf[((long)ih * p.IW + iw0) * CIN + j];
            }
        }
    };
    auto store_patch = [&]() {
        if constexpr (MODE == 0) {
#pragma unroll
            for (int it = 0; it < PIT; ++it) {
                const int row = 2 * (w + 8 * it) + (lane >> 5);
                if (row < R0) s_patch[row * CF_PP + 1 + (lane & 31)] = pv[it];
            }
            if (tid < R0) s_patch[tid * CF_PP] = hv;
        } else {
#pragma unroll
            for (int it = 0; it < PIT; ++it) {
                const int i = it * 512 + tid;
                const int r = i / RUN, j = i - r * RUN;
                if (i < PN) s_patch[r * (CF_PP * CIN) + j] = pv[it];
            }
        }
    };
    // dY tile of block blk -> s_dy[buf]: 16 LDS-DMA instructions of 8 rows, two per wave
    auto stage_dy = [&](int blk, int buf) {
        unsigned b = blk;
        const int tbi = b % (unsigned)tb_n; b /= (unsigned)tb_n;
        const int tai = b % (unsigned)ta_n;
        const int n = b / (unsigned)ta_n;
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2) {
            const int r = (2 * w + i2) * 8 + (lane >> 3);
            const int oh = tai * CF_TA + (r >> 4), ow = tbi * CF_TB + (r & 15);
            const bool v = oh < p.OH && ow < p.OW;
            const unsigned char* src = p.dY + ((size_t)(n * p.OH + oh) * p.OW + ow) * ((size_t)p.ldy * 2) + co0 * 2 +
                                       (((lane & 7) ^ wf_swz(r)) * 16);
            cf_glds16(v ? src : p.zero, s_dy[buf] + (2 * w + i2) * 1024);
        }
    };

    // consumer: wave w = co sub-tile w & 3 x the k sub-tiles 2 (w >> 2), 2 (w >> 2) + 1
    const int fi = lane & 15, fg = lane >> 4;
    const int q = fi >> 2, pp = fi & 3;
    const int mt = w & 3, nt0 = (w >> 2) * 2;
    const int row0 = 8 * fg + q;
    const int offA = row0 * 128 + (((mt * 2 + (pp >> 1)) ^ wf_swz(row0)) * 16) + (pp & 1) * 8;
    const int offB0 = row0 * 128 + (((nt0 * 2 + (pp >> 1)) ^ wf_swz(row0)) * 16) + (pp & 1) * 8;
    const int offB1 = row0 * 128 + ((((nt0 + 1) * 2 + (pp >> 1)) ^ wf_swz(row0)) * 16) + (pp & 1) * 8;
    f32x4_t acc0 = f32x4_t{0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
    const unsigned l_col = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)s_col;
    const unsigned l_dy = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)s_dy[0];

    if (blk0 < blk1) {
        load_patch(blk0);
        stage_dy(blk0, 0);
    }
    int buf = 0;
    for (int blk = blk0; blk < blk1; ++blk) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // this block's patch registers and dY tile (issued a block ago)
#pragma unroll
        for (int it = 0; it < PIT; ++it) asm volatile("" : "+v"(pv[it]));
        asm volatile("" : "+v"(hv));
        store_patch();
        if (blk + 1 < blk1) load_patch(blk + 1);                         // lands under this block's work
        wf_lds_barrier();
        // im2col rows: 128 rows x 8 chunks of 8 columns
#pragma unroll
        for (int i0 = 0; i0 < 128 * 8; i0 += 512) {
            const int i = i0 + tid;
            const int r = i >> 3, c = i & 7;
            const int oy = r >> 4, ox = r & 15;
            const int corner = (2 * oy * CF_PP + 2 * ox) * (MODE == 0 ? 1 : CIN);
            unsigned short e[8];
#pragma unroll
            for (int k8 = 0; k8 < 8; ++k8) {
                const float v = koff[k8] >= 0 ? s_patch[corner + max(koff[k8], 0)] : 0.f;
                e[k8] = f32_to_bf16(v);
            }
            u32x4_t pk;
            pk[0] = (unsigned)e[0] | ((unsigned)e[1] << 16); pk[1] = (unsigned)e[2] | ((unsigned)e[3] << 16);
            pk[2] = (unsigned)e[4] | ((unsigned)e[5] << 16); pk[3] = (unsigned)e[6] | ((unsigned)e[7] << 16);
            *(u32x4_t*)(s_col + r * 128 + ((c ^ wf_swz(r)) * 16)) = pk;
        }
        wf_lds_barrier();
        // the next block's dY tile: issued HERE, behind the last compiler-visible LDS access of the block (in front of one
        // the compiler drains every LDS-DMA); the fragment reads below are asm
        if (blk + 1 < blk1) stage_dy(blk + 1, buf ^ 1);
        const unsigned la = l_dy + buf * (128 * 128);
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            s16x4_t al, ah, b0l, b0h, b1l, b1h;
            const unsigned ada = la + offA + kb * (32 * 128), adb0 = l_col + offB0 + kb * (32 * 128), adb1 = l_col + offB1 + kb * (32 * 128);
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(al) : "v"(ada));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:512" : "=v"(ah) : "v"(ada));
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(b0l) : "v"(adb0));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:512" : "=v"(b0h) : "v"(adb0));
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(b1l) : "v"(adb1));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:512" : "=v"(b1h) : "v"(adb1));
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(al), "+v"(ah), "+v"(b0l), "+v"(b0h), "+v"(b1l), "+v"(b1h));
            const bf16x8_t fa = bf16x8_t{al[0], al[1], al[2], al[3], ah[0], ah[1], ah[2], ah[3]};
            const bf16x8_t fb0 = bf16x8_t{b0l[0], b0l[1], b0l[2], b0l[3], b0h[0], b0h[1], b0h[2], b0h[3]};
            const bf16x8_t fb1 = bf16x8_t{b1l[0], b1l[1], b1l[2], b1l[3], b1h[0], b1h[1], b1h[2], b1h[3]};
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb0, fa, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb1, fa, acc1, 0, 0, 0);
        }
        wf_lds_barrier();                                                // everyone is done with s_patch / s_col / s_dy[buf]
        buf ^= 1;
    }
    // D[row = k 4 fg + r][col = co fi]: a lane owns 4 consecutive k of one co
    float* slab = p.dW + ((size_t)ks * p.Nout + co0 + mt * 16 + fi) * 64 + 4 * fg;
    *(f32x4_t*)(slab + nt0 * 16) = acc0;
    *(f32x4_t*)(slab + (nt0 + 1) * 16) = acc1;
}

// The same sums for layers of 256 (512, ...) output channels with ALL 256 channels of a pixel block in one workgroup.
// wgrad_first_k gives every 64-channel tile its own workgroup, and each of them loads the block's patch and rebuilds its
// im2col rows: at Nout = 256 four workgroups do that work for every block, and it -- LDS reads of the patch, conversions, the
// swizzled row writes -- not the HBM stream is what the kernel runs at (staging its operands two blocks ahead changed nothing,
// DESIGN.md 5).  Here the rows are built once per block and multiplied with the four [128 px][64 co] images of the dY tile
// (each in wgrad_first_k's layout: the same fragment addressing with an image offset).  Both operands travel through
// registers two blocks ahead (two register sets for the even / odd blocks of the walk; every load branch-free, so the
// compiler's counted waits hold); one workgroup per CU (153 KB of LDS); rbvae_wgrad_gemm's slab layout.
constexpr int WFW_LDS = 16384 + 2 * 4 * 16384 + 4 * CF_PA * CF_PP * 4;
template <int CIN, int MODE> __global__ __launch_bounds__(512, 2) void wgrad_first_wide_k(const WfArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char wsm[];
    unsigned char* s_col = wsm;                                   // im2col rows [px][64 k], tr-swizzled
    unsigned char* s_dy = wsm + 16384;                            // [2 buffers][4 channel tiles][128 px][64 co], tr-swizzled
    float* s_patch = (float*)(wsm + 16384 + 2 * 4 * 16384);
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nct = p.Nout >> 8;
    const int cot = blockIdx.x % nct, ks = blockIdx.x / nct;
    const int co0 = cot * 256;
    const int tb_n = (p.OW + CF_TB - 1) / CF_TB, ta_n = (p.OH + CF_TA - 1) / CF_TA;
    const int blk0 = ks * p.per, blk1 = min(blk0 + p.per, p.nblk);
    static constexpr CfOff<CIN, MODE> otab{};
    int koff[8];
#pragma unroll
    for (int k8 = 0; k8 < 8; ++k8) koff[k8] = otab.v[(tid & 7) * 8 + k8];

    constexpr int R0 = CIN * CF_PA, PIT0 = ((R0 + 1) / 2 + 7) / 8;
    constexpr int RUN = CF_PB * CIN, PN = CF_PA * RUN, PIT1 = (PN + 511) / 512;
    constexpr int PIT = MODE == 0 ? PIT0 : PIT1;
    struct Stage { float pv[PIT]; float hv; u32x4_t dy[8]; };
    auto load_block = [&](int blk_, Stage& st) {
        unsigned b = (unsigned)min(blk_, blk1 - 1);                       // (past the end: the last block again, never used)
        const int tbi = b % (unsigned)tb_n; b /= (unsigned)tb_n;
        const int tai = b % (unsigned)ta_n;
        const int n = b / (unsigned)ta_n;
        const int ih0 = 2 * tai * CF_TA - 1, iw0 = 2 * tbi * CF_TB - 1;
        if constexpr (MODE == 0) {
            const float* xf = p.x + cf_frame_off(p.fm, n);
#pragma unroll
            for (int it = 0; it < PIT; ++it) {
                const int row = 2 * (w + 8 * it) + (lane >> 5);
                const int c = row / CF_PA, r = row - c * CF_PA;
                const int ih = ih0 + r, iw = iw0 + 1 + (lane & 31);
                const bool ok = row < R0 && ih >= 0 && ih < p.IH && iw < p.IW;
                const float v = xf[ok ? ((size_t)c * p.IH + ih) * p.IW + iw : 0];
                st.pv[it] = ok ? v : 0.f;
            }
            {
                const int c = tid / CF_PA, r = tid - c * CF_PA;
                const int ih = ih0 + r;
                const bool ok = tid < R0 && ih >= 0 && ih < p.IH && iw0 >= 0;
                const float v = xf[ok ? ((size_t)c * p.IH + ih) * p.IW + iw0 : 0];
                st.hv = ok ? v : 0.f;
            }
        } else {
            const float* xf = p.x + (size_t)n * p.IH * p.IW * CIN;
#pragma unroll
            for (int it = 0; it < PIT; ++it) {
                const int i = it * 512 + tid;
                const int r = i / RUN, j = i - r * RUN;
                const int ih = ih0 + r, iw = iw0 + j / CIN;
                const bool ok = i < PN && ih >= 0 && ih < p.IH && iw >= 0 && iw < p.IW;
                const float v = xf[ok ? ((long)ih * p.IW + iw0) * CIN + j : 0];
                st.pv[it] = ok ? v : 0.f;
            }
            st.hv = 0.f;
        }
        // the dY tile: thread -> pixel rows (2 w + i2) * 8 + lane / 8, swizzled 16-byte chunk lane % 8 of channel tile q
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2) {
            const int r = (2 * w + i2) * 8 + (lane >> 3);
            const int oh = tai * CF_TA + (r >> 4), ow = tbi * CF_TB + (r & 15);
            const bool v = oh < p.OH && ow < p.OW;
            const unsigned char* src = p.dY + ((size_t)(n * p.OH + oh) * p.OW + ow) * ((size_t)p.ldy * 2) + co0 * 2 +
                                       (((lane & 7) ^ wf_swz(r)) * 16);
#pragma unroll
            for (int q = 0; q < 4; ++q) st.dy[q * 2 + i2] = *(const u32x4_t*)(v ? src + q * 128 : p.zero);
        }
    };
    auto store_block = [&](const Stage& st, int buf) {
        if constexpr (MODE == 0) {
#pragma unroll
            for (int it = 0; it < PIT; ++it) {
                const int row = 2 * (w + 8 * it) + (lane >> 5);
                if (row < R0) s_patch[row * CF_PP + 1 + (lane & 31)] = st.pv[it];
            }
            if (tid < R0) s_patch[tid * CF_PP] = st.hv;
        } else {
#pragma unroll
            for (int it = 0; it < PIT; ++it) {
                const int i = it * 512 + tid;
                const int r = i / RUN, j = i - r * RUN;
                if (i < PN) s_patch[r * (CF_PP * CIN) + j] = st.pv[it];
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int i2 = 0; i2 < 2; ++i2)
                *(u32x4_t*)(s_dy + (buf * 4 + q) * 16384 + (2 * w + i2) * 1024 + lane * 16) = st.dy[q * 2 + i2];
    };

    // consumer: wave w = co sub-tile w & 3 of EVERY channel tile x the k sub-tiles 2 (w >> 2), 2 (w >> 2) + 1
    const int fi = lane & 15, fg = lane >> 4;
    const int qq = fi >> 2, pp = fi & 3;
    const int mt = w & 3, nt0 = (w >> 2) * 2;
    const int row0 = 8 * fg + qq;
    const int offA = row0 * 128 + (((mt * 2 + (pp >> 1)) ^ wf_swz(row0)) * 16) + (pp & 1) * 8;
    const int offB0 = row0 * 128 + (((nt0 * 2 + (pp >> 1)) ^ wf_swz(row0)) * 16) + (pp & 1) * 8;
    const int offB1 = row0 * 128 + ((((nt0 + 1) * 2 + (pp >> 1)) ^ wf_swz(row0)) * 16) + (pp & 1) * 8;
    f32x4_t acc[4][2];
#pragma unroll
    for (int q = 0; q < 4; ++q) { acc[q][0] = f32x4_t{0.f, 0.f, 0.f, 0.f}; acc[q][1] = acc[q][0]; }
    const unsigned l_col = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)s_col;
    const unsigned l_dy = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)s_dy;

    auto block = [&](int blk, Stage& st, int buf) {
        store_block(st, buf);
        load_block(blk + 2, st);
        wf_lds_barrier();
        // im2col rows: 128 rows x 8 chunks of 8 columns
#pragma unroll
        for (int i0 = 0; i0 < 128 * 8; i0 += 512) {
            const int i = i0 + tid;
            const int r = i >> 3, c = i & 7;
            const int oy = r >> 4, ox = r & 15;
            const int corner = (2 * oy * CF_PP + 2 * ox) * (MODE == 0 ? 1 : CIN);
            unsigned short e[8];
#pragma unroll
            for (int k8 = 0; k8 < 8; ++k8) {
                const float v = koff[k8] >= 0 ? s_patch[corner + max(koff[k8], 0)] : 0.f;
                e[k8] = f32_to_bf16(v);
            }
            u32x4_t pk;
            pk[0] = (unsigned)e[0] | ((unsigned)e[1] << 16); pk[1] = (unsigned)e[2] | ((unsigned)e[3] << 16);
            pk[2] = (unsigned)e[4] | ((unsigned)e[5] << 16); pk[3] = (unsigned)e[6] | ((unsigned)e[7] << 16);
            *(u32x4_t*)(s_col + r * 128 + ((c ^ wf_swz(r)) * 16)) = pk;
        }
        wf_lds_barrier();
        const unsigned la = l_dy + buf * (4 * 16384);
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            s16x4_t al[4], ah[4], b0l, b0h, b1l, b1h;
            const unsigned ada = la + offA + kb * (32 * 128), adb0 = l_col + offB0 + kb * (32 * 128), adb1 = l_col + offB1 + kb * (32 * 128);
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(b0l) : "v"(adb0));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:512" : "=v"(b0h) : "v"(adb0));
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(b1l) : "v"(adb1));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:512" : "=v"(b1h) : "v"(adb1));
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(al[0]) : "v"(ada));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:512" : "=v"(ah[0]) : "v"(ada));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:16384" : "=v"(al[1]) : "v"(ada));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:16896" : "=v"(ah[1]) : "v"(ada));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:32768" : "=v"(al[2]) : "v"(ada));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:33280" : "=v"(ah[2]) : "v"(ada));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:49152" : "=v"(al[3]) : "v"(ada));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:49664" : "=v"(ah[3]) : "v"(ada));
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(b0l), "+v"(b0h), "+v"(b1l), "+v"(b1h), "+v"(al[0]), "+v"(ah[0]), "+v"(al[1]), "+v"(ah[1]), "+v"(al[2]),
                           "+v"(ah[2]), "+v"(al[3]), "+v"(ah[3]));
            const bf16x8_t fb0 = bf16x8_t{b0l[0], b0l[1], b0l[2], b0l[3], b0h[0], b0h[1], b0h[2], b0h[3]};
            const bf16x8_t fb1 = bf16x8_t{b1l[0], b1l[1], b1l[2], b1l[3], b1h[0], b1h[1], b1h[2], b1h[3]};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bf16x8_t fa = bf16x8_t{al[q][0], al[q][1], al[q][2], al[q][3], ah[q][0], ah[q][1], ah[q][2], ah[q][3]};
                acc[q][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb0, fa, acc[q][0], 0, 0, 0);
                acc[q][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb1, fa, acc[q][1], 0, 0, 0);
            }
        }
        wf_lds_barrier();                                                // everyone is done with s_patch / s_col / this dY buffer
    };
    if (blk0 < blk1) {
        Stage sa, sb;
        load_block(blk0, sa);
        load_block(blk0 + 1, sb);
        for (int blk = blk0; blk < blk1; blk += 2) {
            block(blk, sa, 0);
            if (blk + 1 < blk1) block(blk + 1, sb, 1);
        }
    }
    // D[row = k 4 fg + r][col = co fi]: a lane owns 4 consecutive k of one co
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float* slab = p.dW + ((size_t)ks * p.Nout + co0 + q * 64 + mt * 16 + fi) * 64 + 4 * fg;
        *(f32x4_t*)(slab + nt0 * 16) = acc[q][0];
        *(f32x4_t*)(slab + (nt0 + 1) * 16) = acc[q][1];
    }
}

static int wf_shape_ok(int dtype, int Cin, int IH, int IW, int Nout, int N) {
    const int OH = (IH + 2 - 3) / 2 + 1, OW = (IW + 2 - 3) / 2 + 1;
    return dtype == RBVAE_BF16 && Cin >= 1 && Cin <= 4 && Nout >= 64 && Nout % 64 == 0 && N >= 1 && IH >= 1 && IW >= 1 &&
           (long)N * OH * OW * 256 < (1l << 31) && (long)N * Cin * IH * IW < (1l << 40);
}
template <int CIN, int MODE> static void wf_launch_wide(const WfArgs& a, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)wgrad_first_wide_k<CIN, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, WFW_LDS);
        attr_set = true;
    }
    hipLaunchKernelGGL((wgrad_first_wide_k<CIN, MODE>), dim3((a.Nout / 256) * a.ksplit), dim3(512), WFW_LDS, st, a);
}
template <int MODE> static void wf_launch(const WfArgs& a, hipStream_t st) {
    if (a.Nout % 256 == 0) {           // all 256 channels of a block in one workgroup: the im2col rows are built once, not four times
        switch (a.Cin) {
            case 1: wf_launch_wide<1, MODE>(a, st); break;
            case 2: wf_launch_wide<2, MODE>(a, st); break;
            case 3: wf_launch_wide<3, MODE>(a, st); break;
            default: wf_launch_wide<4, MODE>(a, st); break;
        }
        return;
    }
    const int blocks = (a.Nout / 64) * a.ksplit;
    switch (a.Cin) {
        case 1: hipLaunchKernelGGL((wgrad_first_k<1, MODE>), dim3(blocks), dim3(512), 0, st, a); break;
        case 2: hipLaunchKernelGGL((wgrad_first_k<2, MODE>), dim3(blocks), dim3(512), 0, st, a); break;
        case 3: hipLaunchKernelGGL((wgrad_first_k<3, MODE>), dim3(blocks), dim3(512), 0, st, a); break;
        default: hipLaunchKernelGGL((wgrad_first_k<4, MODE>), dim3(blocks), dim3(512), 0, st, a); break;
    }
}

static int cf_shape_ok(int dtype, int Cin, int IH, int IW, int Nout, int N) {
    const int OH = (IH + 2 - 3) / 2 + 1, OW = (IW + 2 - 3) / 2 + 1;
    return dtype == RBVAE_BF16 && Cin >= 1 && Cin <= 4 && Nout >= 8 && Nout <= 256 && Nout % 8 == 0 && N >= 1 && IH >= 1 &&
           IW >= 1 && (long)N * OH * OW * 256 < (1l << 31) && (long)N * Cin * IH * IW < (1l << 40);
}

template <int MODE, int NQ> static void cf_launch_nq(const CfArgs& a, hipStream_t st) {
    const int blocks = a.N * cdiv(a.OH, CF_TA) * cdiv(a.OW, CF_TB);
    switch (a.Cin) {
        case 1: hipLaunchKernelGGL((conv_first_fused_k<1, MODE, NQ>), dim3(blocks), dim3(512), 0, st, a); break;
        case 2: hipLaunchKernelGGL((conv_first_fused_k<2, MODE, NQ>), dim3(blocks), dim3(512), 0, st, a); break;
        case 3: hipLaunchKernelGGL((conv_first_fused_k<3, MODE, NQ>), dim3(blocks), dim3(512), 0, st, a); break;
        default: hipLaunchKernelGGL((conv_first_fused_k<4, MODE, NQ>), dim3(blocks), dim3(512), 0, st, a); break;
    }
}
template <int MODE> static void cf_launch(const CfArgs& a, hipStream_t st) {
    if (a.Nout <= 64) cf_launch_nq<MODE, 1>(a, st); else cf_launch_nq<MODE, 4>(a, st);
}

}  // namespace rbvae

using namespace rbvae;

/* 1 when rbvae_conv_first_fused covers the shape (bf16, 3x3 stride 2 pad 1, Cin <= 4, Nout <= 256, Nout % 8 == 0) */
extern "C" int rbvae_conv_first_fused_ok(int dtype, int Cin, int IH, int IW, int Nout, int N) {
    return cf_shape_ok(dtype, Cin, IH, IW, Nout, N);
}

extern "C" int rbvae_conv_first_fused(int dtype, const float* x, int fd1, int fd2, long fs0, long fs1, long fs2, const void* W,
                                      const float* bias, const void* zero_page, void* col, void* out, int N, int Cin, int IH,
                                      int IW, int Nout, int ldo, int relu, int drop_mode, float drop_p, float scale,
                                      unsigned long long seed, const unsigned long long* seed_dev, void* stream) {
    RBVAE_CHECK_ARG(x && W && zero_page && out, "conv_first_fused: null pointer");
    RBVAE_CHECK_ARG(cf_shape_ok(dtype, Cin, IH, IW, Nout, N), "conv_first_fused: shape outside the fused kernel "
                    "(bf16, Cin <= 4, Nout <= 256): Cin=%d %dx%d Nout=%d", Cin, IH, IW, Nout);
    RBVAE_CHECK_ARG(ldo >= Nout && ldo % 8 == 0, "conv_first_fused: ldo=%d", ldo);
    RBVAE_CHECK_ARG(drop_mode == 0 || drop_mode == 1, "conv_first_fused: drop_mode %d (explicit masks: two-kernel path)", drop_mode);
    RBVAE_CHECK_ARG(((uintptr_t)W | (uintptr_t)zero_page | (uintptr_t)col | (uintptr_t)out) % 16 == 0 &&
                    (!bias || (uintptr_t)bias % 16 == 0), "conv_first_fused: pointers must be 16-byte aligned");
    CfArgs a;
    a.x = x; a.fm = CfFrameMap{fd1, fd2, fs0, fs1, fs2}; a.W = (const unsigned char*)W; a.bias = bias;
    a.zero = (const unsigned char*)zero_page; a.col = (unsigned char*)col; a.out = (unsigned char*)out;
    a.gate = nullptr; a.colsum_ws = nullptr;
    a.N = N; a.Cin = Cin; a.IH = IH; a.IW = IW; a.OH = (IH + 2 - 3) / 2 + 1; a.OW = (IW + 2 - 3) / 2 + 1;
    a.Nout = Nout; a.ldo = ldo; a.relu = relu; a.drop_mode = drop_mode; a.scale = scale;
    a.drop_thresh = (unsigned)((double)drop_p * 4294967296.0); a.seed = seed; a.seed_dev = seed_dev;
    cf_launch<0>(a, (hipStream_t)stream);
    RBVAE_CHECK_LAUNCH("conv_first_fused");
    return RBVAE_OK;
}

/* workgroups (= rows of colsum_ws) of rbvae_deconv_last_dgrad_fused; 0 when the shape is outside the fused kernel */
extern "C" int rbvae_deconv_last_dgrad_blocks(int dtype, int Cout, int OH, int OW, int C1, int N) {
    if (!cf_shape_ok(dtype, Cout, OH, OW, C1, N)) return 0;
    return N * cdiv((OH + 2 - 3) / 2 + 1, CF_TA) * cdiv((OW + 2 - 3) / 2 + 1, CF_TB);
}

extern "C" int rbvae_deconv_last_dgrad_fused(int dtype, const float* dpre, const void* W, const void* zero_page, void* col,
                                             const void* gate, void* out, int N, int Cout, int OH, int OW, int C1, int ldo,
                                             float scale, float* colsum_ws, void* stream) {
    RBVAE_CHECK_ARG(dpre && W && zero_page && gate && out, "deconv_last_dgrad_fused: null pointer");
    RBVAE_CHECK_ARG(cf_shape_ok(dtype, Cout, OH, OW, C1, N), "deconv_last_dgrad_fused: shape outside the fused kernel "
                    "(bf16, Cout <= 4, C1 <= 256): Cout=%d %dx%d C1=%d", Cout, OH, OW, C1);
    RBVAE_CHECK_ARG(ldo >= C1 && ldo % 8 == 0, "deconv_last_dgrad_fused: ldo=%d", ldo);
    RBVAE_CHECK_ARG(((uintptr_t)W | (uintptr_t)zero_page | (uintptr_t)col | (uintptr_t)out | (uintptr_t)gate) % 16 == 0,
                    "deconv_last_dgrad_fused: pointers must be 16-byte aligned");
    CfArgs a;
    a.x = dpre; a.fm = CfFrameMap{0, 0, 0, 0, 0}; a.W = (const unsigned char*)W; a.bias = nullptr;
    a.zero = (const unsigned char*)zero_page; a.col = (unsigned char*)col; a.out = (unsigned char*)out;
    a.gate = (const unsigned char*)gate; a.colsum_ws = colsum_ws;
    a.N = N; a.Cin = Cout; a.IH = OH; a.IW = OW; a.OH = (OH + 2 - 3) / 2 + 1; a.OW = (OW + 2 - 3) / 2 + 1;
    a.Nout = C1; a.ldo = ldo; a.relu = 0; a.drop_mode = 0; a.scale = scale; a.drop_thresh = 0; a.seed = 0; a.seed_dev = nullptr;
    cf_launch<1>(a, (hipStream_t)stream);
    RBVAE_CHECK_LAUNCH("deconv_last_dgrad_fused");
    return RBVAE_OK;
}

/* 8 x 16 output-pixel blocks rbvae_wgrad_first walks (the caller sizes ksplit against it); 0 when the shape is not covered
 * (bf16, Cin <= 4, Nout a multiple of 64) */
extern "C" int rbvae_wgrad_first_blocks(int dtype, int Cin, int IH, int IW, int Nout, int N) {
    if (!wf_shape_ok(dtype, Cin, IH, IW, Nout, N)) return 0;
    return N * cdiv((IH + 2 - 3) / 2 + 1, CF_TA) * cdiv((IW + 2 - 3) / 2 + 1, CF_TB);
}

extern "C" int rbvae_wgrad_first(int dtype, int mode, const float* x, int fd1, int fd2, long fs0, long fs1, long fs2, const void* dY,
                                 float* dW_slabs, const void* zero_page, int N, int Cin, int IH, int IW, int Nout, int ldy,
                                 int ksplit, void* stream) {
    RBVAE_CHECK_ARG(x && dY && dW_slabs && zero_page, "wgrad_first: null pointer");
    RBVAE_CHECK_ARG(mode == 0 || mode == 1, "wgrad_first: mode %d (0 = frames through the frame map, 1 = NHWC f32)", mode);
    RBVAE_CHECK_ARG(wf_shape_ok(dtype, Cin, IH, IW, Nout, N), "wgrad_first: shape not covered (bf16, Cin <= 4, Nout %% 64 == 0): "
                    "Cin=%d %dx%d Nout=%d", Cin, IH, IW, Nout);
    RBVAE_CHECK_ARG(ldy >= Nout && ldy % 8 == 0, "wgrad_first: ldy=%d", ldy);
    RBVAE_CHECK_ARG(((uintptr_t)dY | (uintptr_t)dW_slabs | (uintptr_t)zero_page) % 16 == 0, "wgrad_first: pointers must be 16-byte aligned");
    WfArgs a;
    a.x = x; a.fm = mode == 0 ? CfFrameMap{fd1, fd2, fs0, fs1, fs2} : CfFrameMap{0, 0, 0, 0, 0};
    a.dY = (const unsigned char*)dY; a.zero = (const unsigned char*)zero_page; a.dW = dW_slabs;
    a.N = N; a.Cin = Cin; a.IH = IH; a.IW = IW; a.OH = (IH + 2 - 3) / 2 + 1; a.OW = (IW + 2 - 3) / 2 + 1;
    a.Nout = Nout; a.ldy = ldy;
    a.nblk = N * cdiv(a.OH, CF_TA) * cdiv(a.OW, CF_TB);
    RBVAE_CHECK_ARG(ksplit >= 1 && ksplit <= a.nblk, "wgrad_first: ksplit=%d (1 .. %d blocks)", ksplit, a.nblk);
    a.ksplit = ksplit; a.per = cdiv(a.nblk, ksplit);
    if (mode == 0) wf_launch<0>(a, (hipStream_t)stream); else wf_launch<1>(a, (hipStream_t)stream);
    RBVAE_CHECK_LAUNCH("wgrad_first");
    return RBVAE_OK;
}
