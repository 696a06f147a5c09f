// The first Conv2d(Cin <= 4 -> Nout <= 256, k3 s2 p1) + bias + ReLU + Dropout of the encoder
// (percep_RBVAE_model.py:51-53) as ONE kernel, bf16 storage:
//
//   a workgroup takes an 8 x 16 block of OUTPUT pixels of one frame, loads the 17 x 33 input patch of every channel
//   with coalesced loads (f32 NCHW, frames through the frame map), builds the im2col rows [128][64] (column
//   (kh*3+kw)*Cin + ci, zero padded) as the swizzled LDS image the MFMA fragments read -- and writes them to
//   col1, which the weight gradient reads later -- stages W [Nout][64] by LDS-DMA, multiplies on the matrix cores
//   and stores bias / ReLU / scale / keyed-dropout results straight from the accumulators as 16-byte chunks.
//
// It replaces rbvae_im2col + the single-slice gather GEMM of the two-kernel path (same arithmetic per element:
// one 64-deep MFMA chain; same dropout key and chunk indices).
#include "common.h"
#include <stdlib.h>

#ifndef CF_DBG      // timing experiments only: 1 no output stores, 2 no col stores, 3 stop after the staging, 4 no patch loads
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
    const float* x;              // frames [Cin][IH][IW] f32 at cf_frame_off(fm, n)
    CfFrameMap fm;
    const unsigned char* W;      // [Nout][64] bf16 (im2col column order, zero padded)
    const float* bias;           // [Nout] or null
    const unsigned char* zero;   // >= 16 zero bytes
    unsigned char* col;          // [N*OH*OW][64] bf16 out
    unsigned char* out;          // [N*OH*OW][ldo] bf16 out
    int N, Cin, IH, IW, OH, OW, Nout, ldo, relu, drop_mode;
    float scale;
    unsigned drop_thresh;
    unsigned long long seed;
    const unsigned long long* seed_dev;
};

constexpr int CF_TA = 8, CF_TB = 16;                          // output block
constexpr int CF_PA = 2 * CF_TA + 1, CF_PB = 2 * CF_TB + 1;   // input patch 17 x 33
constexpr int CF_PP = CF_PB + 1;                              // patch row pitch (floats)

// patch offset of im2col column k = (kh*3+kw)*CIN + ci, relative to the pixel's corner (2*oy)*CF_PP + 2*ox; -1 = padding
template <int CIN> struct CfOff {
    int v[64];
    constexpr CfOff() : v{} {
        for (int k = 0; k < 64; ++k) {
            const int t = k / CIN, ci = k % CIN, kh = t / 3, kw = t % 3;
            v[k] = k < 9 * CIN ? (ci * CF_PA + kh) * CF_PP + kw : -1;
        }
    }
};

__device__ __forceinline__ void cf_glds16(const void* g, void* lds) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}

template <int CIN> __global__ __launch_bounds__(512, 4) void conv_first_fused_k(const CfArgs p) {
    __shared__ __attribute__((aligned(16))) unsigned char s_a[128 * 128];        // im2col rows, swizzled chunks
    __shared__ __attribute__((aligned(16))) unsigned char s_b[256 * 128];        // weights in fragment-row order
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
        for (int q = w; q < 32; q += 8) {                    // 32 instructions of 8 rows
            const int r = q * 8 + srow;
            const int ct = r >> 4, j = r & 15;
            const int ch = 64 * (ct >> 2) + 16 * (j >> 2) + 4 * (ct & 3) + (j & 3);
            const unsigned char* src = ch < p.Nout ? p.W + (size_t)ch * 128 + ((schunk ^ ((r >> 1) & 7)) * 16) : p.zero;
            cf_glds16(src, s_b + (size_t)(r - srow) * 128);
        }
    }
    // bias and dropout key early: their latency hides behind the patch loads, and the epilogue's stores are never waited on
    const int fi = lane & 15, fg = lane >> 4;
    const int wq = w & 3, wm = w >> 2;
    const int col = 64 * wq + 16 * fg;
    const bool second = col + 8 < p.Nout;
    f32x4_t bz4[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        bz4[h] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        if (p.bias && col + 4 * h < p.Nout) bz4[h] = *(const f32x4_t*)(p.bias + col + 4 * h);
    }
    DropKey dkey{0u, 0u};
    int koff[8];
    // ---- input patch, coalesced along iw
    const float* xf = p.x + cf_frame_off(p.fm, n);
    // s_patch row c*17 + r, column 0 = the halo column 2*ow0 - 1, columns 1 .. 32 = the 128-byte run from 2*ow0: one
    // instruction loads two rows (32 lanes each)
    {
        static constexpr CfOff<CIN> otab{};
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
        const int cchunk = tid & 7;
#pragma unroll
        for (int k8 = 0; k8 < 8; ++k8) koff[k8] = otab.v[cchunk * 8 + k8];
        if (p.drop_mode == 1) dkey = drop_key(p.seed + (p.seed_dev ? p.seed_dev[0] * 0x9E3779B97F4A7C15ull : 0ull));
#pragma unroll
        for (int it = 0; it < PIT; ++it) {
            const int row = 2 * (w + 8 * it) + (lane >> 5);
            if (row < R) s_patch[row * CF_PP + 1 + (lane & 31)] = pv[it];
        }
        if (tid < R) s_patch[tid * CF_PP] = hv;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // weights (LDS-DMA), bias, key: everything this wave asked for
#pragma unroll
    for (int h = 0; h < 4; ++h) asm volatile("" : "+v"(bz4[h]));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    asm volatile("" : "+v"(dkey.k0), "+v"(dkey.k1));
    __syncthreads();
    // ---- im2col rows: 128 rows x 8 chunks of 8 columns; column k = (kh*3+kw)*Cin + ci
#pragma unroll
    for (int i0 = 0; i0 < 128 * 8; i0 += 512) {
        const int i = i0 + tid;
        const int r = i >> 3, c = i & 7;
        const int oy = r >> 4, ox = r & 15;
        const int corner = 2 * oy * CF_PP + 2 * ox;
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
        if (CF_DBG != 2 && oh < p.OH && ow < p.OW)
            *(u32x4_t*)(p.col + ((size_t)(n * p.OH + oh) * p.OW + ow) * 128 + c * 16) = pk;
    }
    __syncthreads();

    // ---- 128 x 256 x 64 on the matrix cores: wave w owns pixels 64*(w/4) .. +63 x the tile quad w%4 = channels 64*(w%4) .. +63
    if (CF_DBG == 3) return;
    const int fsw = (fi >> 1) & 7;
    f32x4_t acc[4][4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int h = 0; h < 4; ++h) acc[mt][h] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int ch = ((4 * kk + fg) ^ fsw) * 16;
        u32x4_t wv[4], av[4];
#pragma unroll
        for (int h = 0; h < 4; ++h) wv[h] = *(const u32x4_t*)(s_b + ((4 * wq + h) * 16 + fi) * 128 + ch);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) av[mt] = *(const u32x4_t*)(s_a + ((4 * wm + mt) * 16 + fi) * 128 + ch);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int h = 0; h < 4; ++h)
                acc[mt][h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8_t*)&wv[h], *(const bf16x8_t*)&av[mt],
                                                                     acc[mt][h], 0, 0, 0);
    }
    // ---- epilogue from registers: lane = pixel fi of tile mt, channels 64*wq + 16*fg .. +15 (two 16-byte chunks)
    if (col >= p.Nout) return;
    const float floor_ = p.relu ? 0.f : -3.0e38f;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int r = (4 * wm + mt) * 16 + fi;
        const int oh = oh0 + (r >> 4), ow = ow0 + (r & 15);
        if (oh >= p.OH || ow >= p.OW) continue;
        const size_t orow = (size_t)(n * p.OH + oh) * p.OW + ow;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if (half == 1 && !second) break;
            float xv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e)
                xv[e] = fmaxf(acc[mt][2 * half + (e >> 2)][e & 3] + bz4[2 * half + (e >> 2)][e & 3], floor_) * p.scale;
            if (p.drop_mode == 1)
                drop_chunk_zero_f32<8>(drop_run(dkey, (unsigned long long)orow * p.Nout + col + 8 * half), p.drop_thresh >> 16, xv);
            u32x4_t val;
#pragma unroll
            for (int e = 0; e < 4; ++e) val[e] = (unsigned)f32_to_bf16(xv[2 * e]) | ((unsigned)f32_to_bf16(xv[2 * e + 1]) << 16);
            if (CF_DBG != 1 || val[0] == 0x12345678u) *(u32x4_t*)(p.out + (orow * p.ldo + col + 8 * half) * 2) = val;
        }
    }
}

}  // namespace rbvae

using namespace rbvae;

/* 1 when rbvae_conv_first_fused covers the shape (bf16, 3x3 stride 2 pad 1, Cin <= 4, Nout <= 256, Nout % 8 == 0) */
extern "C" int rbvae_conv_first_fused_ok(int dtype, int Cin, int IH, int IW, int Nout, int N) {
    const int OH = (IH + 2 - 3) / 2 + 1, OW = (IW + 2 - 3) / 2 + 1;
    return dtype == RBVAE_BF16 && Cin >= 1 && Cin <= 4 && Nout >= 8 && Nout <= 256 && Nout % 8 == 0 &&
           (long)N * OH * OW * 256 < (1l << 31) && (long)N * Cin * IH * IW < (1l << 40);
}

extern "C" int rbvae_conv_first_fused(int dtype, const float* x, int fd1, int fd2, long fs0, long fs1, long fs2, const void* W,
                                      const float* bias, const void* zero_page, void* col, void* out, int N, int Cin, int IH,
                                      int IW, int Nout, int ldo, int relu, int drop_mode, float drop_p, float scale,
                                      unsigned long long seed, const unsigned long long* seed_dev, void* stream) {
    RBVAE_CHECK_ARG(x && W && zero_page && col && out, "conv_first_fused: null pointer");
    RBVAE_CHECK_ARG(rbvae_conv_first_fused_ok(dtype, Cin, IH, IW, Nout, N), "conv_first_fused: shape outside the fused kernel "
                    "(bf16, Cin <= 4, Nout <= 256): Cin=%d %dx%d Nout=%d", Cin, IH, IW, Nout);
    RBVAE_CHECK_ARG(ldo >= Nout && ldo % 8 == 0, "conv_first_fused: ldo=%d", ldo);
    RBVAE_CHECK_ARG(drop_mode == 0 || drop_mode == 1, "conv_first_fused: drop_mode %d (explicit masks: two-kernel path)", drop_mode);
    RBVAE_CHECK_ARG(((uintptr_t)W | (uintptr_t)zero_page | (uintptr_t)col | (uintptr_t)out) % 16 == 0,
                    "conv_first_fused: pointers must be 16-byte aligned");
    CfArgs a;
    a.x = x; a.fm = CfFrameMap{fd1, fd2, fs0, fs1, fs2}; a.W = (const unsigned char*)W; a.bias = bias;
    a.zero = (const unsigned char*)zero_page; a.col = (unsigned char*)col; a.out = (unsigned char*)out;
    a.N = N; a.Cin = Cin; a.IH = IH; a.IW = IW; a.OH = (IH + 2 - 3) / 2 + 1; a.OW = (IW + 2 - 3) / 2 + 1;
    a.Nout = Nout; a.ldo = ldo; a.relu = relu; a.drop_mode = drop_mode; a.scale = scale;
    a.drop_thresh = (unsigned)((double)drop_p * 4294967296.0); a.seed = seed; a.seed_dev = seed_dev;
    const int blocks = N * cdiv(a.OH, CF_TA) * cdiv(a.OW, CF_TB);
    switch (Cin) {
        case 1: hipLaunchKernelGGL(conv_first_fused_k<1>, dim3(blocks), dim3(512), 0, (hipStream_t)stream, a); break;
        case 2: hipLaunchKernelGGL(conv_first_fused_k<2>, dim3(blocks), dim3(512), 0, (hipStream_t)stream, a); break;
        case 3: hipLaunchKernelGGL(conv_first_fused_k<3>, dim3(blocks), dim3(512), 0, (hipStream_t)stream, a); break;
        default: hipLaunchKernelGGL(conv_first_fused_k<4>, dim3(blocks), dim3(512), 0, (hipStream_t)stream, a); break;
    }
    RBVAE_CHECK_LAUNCH("conv_first_fused");
    return RBVAE_OK;
}
