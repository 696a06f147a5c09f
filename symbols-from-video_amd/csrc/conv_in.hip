// The LDM encoder's conv_in: Conv2d(Cin <= 4 -> Nout, 3x3, stride 1, pad 1) on f32 NCHW frames
// (src/stable-diffusion/ldm/modules/diffusionmodules/model.py:385-389, :436) as ONE kernel in bf16 storage, with the
// GroupNorm partial statistics of its output (the first ResnetBlock's norm1, model.py:121) out of the epilogue.
//
// It replaces rbvae_im2col + the one-tap rbvae_gather_gemm + the GroupNorm statistics kernels: at 4 frames of 512 x 512
// those wrote and re-read 134 MB of im2col rows and read the 268 MB output once more for its statistics (171 + 89 us for a
// layer whose own traffic is 13 MB in, 268 MB out).  Same structure as conv_first_fused_k (conv_first.hip), stride 1: a
// workgroup takes an 8 x 16 block of output pixels, loads the 10 x 18 input patch, builds the im2col rows [128][64]
// (column (kh*3+kw)*Cin + ci, zero padded) as the swizzled LDS image the MFMA fragments read, stages W [Nout][64] by
// LDS-DMA, multiplies (one 64-deep MFMA chain per output: the arithmetic of the two-kernel path) and stores straight from
// the accumulators, 32 bytes per lane and pixel.  Statistics: per workgroup tile and group of cg channels the mean and the
// sum of squared deviations of the STORED (bf16) values, two passes over the registers, merged over lanes by DPP row sums
// and over waves through LDS; rbvae_gn_finish_tiles (tile 8 x 16) merges the tiles.
#include "common.h"
#include <stdlib.h>

namespace rbvae {

typedef __attribute__((ext_vector_type(8))) short ci_bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float ci_f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned ci_u32x4_t;

struct CiArgs {
    const float* x;              // [N][Cin][H][W] f32
    const unsigned char* W;      // [Nout][64] bf16 (im2col column order, zero padded)
    const float* bias;           // [Nout] or null
    const unsigned char* zero;   // >= 16 zero bytes
    unsigned char* out;          // [N*H*W][ldo] bf16
    float2* stats;               // [N][tiles][Nout / cg] (mean, M2) or null
    int N, Cin, H, Wd, Nout, ldo, cg;
};

constexpr int CI_TA = 8, CI_TB = 16;                  // output block
constexpr int CI_PA = CI_TA + 2, CI_PB = CI_TB + 2;   // input patch 10 x 18
constexpr int CI_PP = CI_PB + 1;                      // patch row pitch (floats)

template <int CIN> struct CiOff {
    int v[64];
    constexpr CiOff() : v{} {
        for (int k = 0; k < 64; ++k) {
            const int t = k / CIN, ci = k % CIN, kh = t / 3, kw = t % 3;
            v[k] = k >= 9 * CIN ? -1 : (ci * CI_PA + kh) * CI_PP + kw;
        }
    }
};

__device__ __forceinline__ void ci_glds16(const void* g, void* lds) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}
template <int CTRL> __device__ __forceinline__ float ci_dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
// sum over the 16 lanes of a DPP row (lane & 15), every lane gets the total
__device__ __forceinline__ float ci_row_sum(float v) {
    v += ci_dpp<0x128>(v);     // row_ror:8
    v += ci_dpp<0x124>(v);     // row_ror:4
    v += ci_dpp<0x4E>(v);      // quad_perm [2,3,0,1]
    v += ci_dpp<0xB1>(v);      // quad_perm [1,0,3,2]
    return v;
}

// NQ = 64-channel quads of the tile (Nout <= 64 NQ); wave = quad w % NQ x pixel tiles NQ (w / NQ) .. + NQ - 1
template <int CIN, int NQ> __global__ __launch_bounds__(512, NQ == 1 ? 8 : 4) void conv_in_k(const CiArgs p) {
    constexpr int MT = NQ;                                                       // 16-pixel tiles per wave
    constexpr int NWM = 8 / NQ;                                                  // waves along the pixels
    __shared__ __attribute__((aligned(16))) unsigned char s_a[128 * 128];        // im2col rows, swizzled chunks
    __shared__ __attribute__((aligned(16))) unsigned char s_b[64 * NQ * 128];    // weights in fragment-row order
    __shared__ float s_patch[4 * CI_PA * CI_PP];
    __shared__ float s_red[NWM * 64 * NQ];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int tb_n = (p.Wd + CI_TB - 1) / CI_TB, ta_n = (p.H + CI_TA - 1) / CI_TA;
    unsigned blk = blockIdx.x;
    const int tbi = blk % (unsigned)tb_n; blk /= (unsigned)tb_n;
    const int tai = blk % (unsigned)ta_n;
    const int n = blk / (unsigned)ta_n;
    const int oh0 = tai * CI_TA, ow0 = tbi * CI_TB;

    // weights -> LDS by LDS-DMA: image row ct*16 + j holds channel 64*(ct/4) + 16*(j/4) + 4*(ct%4) + j%4, so that a lane's
    // accumulators of a tile quad are 16 consecutive channels (conv_first_fused_k's layout)
    {
        const int srow = lane >> 3, schunk = lane & 7;
        for (int q = w; q < 8 * NQ; q += 8) {
            const int r = q * 8 + srow;
            const int ct = r >> 4, j = r & 15;
            const int ch = 64 * (ct >> 2) + 16 * (j >> 2) + 4 * (ct & 3) + (j & 3);
            const unsigned char* src = ch < p.Nout ? p.W + (size_t)ch * 128 + ((schunk ^ ((r >> 1) & 7)) * 16) : p.zero;
            ci_glds16(src, s_b + (size_t)(r - srow) * 128);
        }
    }
    const int fi = lane & 15, fg = lane >> 4;
    const int wq = w % NQ, wm = w / NQ;
    const int col = 64 * wq + 16 * fg;
    const bool first = col < p.Nout, second = col + 8 < p.Nout;
    ci_f32x4_t bz4[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        bz4[h] = ci_f32x4_t{0.f, 0.f, 0.f, 0.f};
        if (p.bias && col + 4 * h < p.Nout) bz4[h] = *(const ci_f32x4_t*)(p.bias + col + 4 * h);
    }
    static constexpr CiOff<CIN> otab{};
    int koff[8];
#pragma unroll
    for (int k8 = 0; k8 < 8; ++k8) koff[k8] = otab.v[(tid & 7) * 8 + k8];
    // input patch [ci][10][18] (pad 1: rows / columns outside the image are zero)
    {
        const float* xf = p.x + (size_t)n * CIN * p.H * p.Wd;
        constexpr int PN = CIN * CI_PA * CI_PB, PIT = (PN + 511) / 512;
        float pv[PIT];
#pragma unroll
        for (int it = 0; it < PIT; ++it) {
            const int i = it * 512 + tid;
            const int c = i / (CI_PA * CI_PB), rr = i - c * (CI_PA * CI_PB), r = rr / CI_PB, j = rr - r * CI_PB;
            const int ih = oh0 - 1 + r, iw = ow0 - 1 + j;
            pv[it] = 0.f;
            if (i < PN && ih >= 0 && ih < p.H && iw >= 0 && iw < p.Wd) pv[it] = xf[((size_t)c * p.H + ih) * p.Wd + iw];
        }
#pragma unroll
        for (int it = 0; it < PIT; ++it) {
            const int i = it * 512 + tid;
            const int c = i / (CI_PA * CI_PB), rr = i - c * (CI_PA * CI_PB), r = rr / CI_PB, j = rr - r * CI_PB;
            if (i < PN) s_patch[(c * CI_PA + r) * CI_PP + j] = pv[it];
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // weights (LDS-DMA), bias
#pragma unroll
    for (int h = 0; h < 4; ++h) asm volatile("" : "+v"(bz4[h]));
    __syncthreads();
    // im2col rows: 128 rows x 8 chunks of 8 columns
#pragma unroll
    for (int i0 = 0; i0 < 128 * 8; i0 += 512) {
        const int i = i0 + tid;
        const int r = i >> 3, c = i & 7;
        const int oy = r >> 4, ox = r & 15;
        const int corner = oy * CI_PP + ox;
        unsigned short e[8];
#pragma unroll
        for (int k8 = 0; k8 < 8; ++k8) {
            const float v = koff[k8] >= 0 ? s_patch[corner + max(koff[k8], 0)] : 0.f;
            e[k8] = f32_to_bf16(v);
        }
        ci_u32x4_t pk;
        pk[0] = (unsigned)e[0] | ((unsigned)e[1] << 16); pk[1] = (unsigned)e[2] | ((unsigned)e[3] << 16);
        pk[2] = (unsigned)e[4] | ((unsigned)e[5] << 16); pk[3] = (unsigned)e[6] | ((unsigned)e[7] << 16);
        *(ci_u32x4_t*)(s_a + r * 128 + ((c ^ ((r >> 1) & 7)) * 16)) = pk;
    }
    __syncthreads();

    // 128 x 64 NQ x 64 on the matrix cores
    const int fsw = (fi >> 1) & 7;
    ci_f32x4_t acc[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int h = 0; h < 4; ++h) acc[mt][h] = ci_f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int ch = ((4 * kk + fg) ^ fsw) * 16;
        ci_u32x4_t wv[4], av[MT];
#pragma unroll
        for (int h = 0; h < 4; ++h) wv[h] = *(const ci_u32x4_t*)(s_b + ((4 * wq + h) * 16 + fi) * 128 + ch);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) av[mt] = *(const ci_u32x4_t*)(s_a + ((MT * wm + mt) * 16 + fi) * 128 + ch);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int h = 0; h < 4; ++h)
                acc[mt][h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const ci_bf16x8_t*)&wv[h], *(const ci_bf16x8_t*)&av[mt],
                                                                     acc[mt][h], 0, 0, 0);
    }
    // epilogue from registers: lane = pixel fi of tile mt, channels 64*wq + 16*fg .. +15 (two 16-byte chunks); the stored
    // (rounded) values stay in xs for the statistics
    float xs[MT][16];
    bool live[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int r = (MT * wm + mt) * 16 + fi;
        const int oh = oh0 + (r >> 4), ow = ow0 + (r & 15);
        live[mt] = oh < p.H && ow < p.Wd && first;
        const size_t orow = (size_t)(n * p.H + oh) * p.Wd + ow;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            ci_u32x4_t val;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i0 = 8 * half + 2 * e, i1 = i0 + 1;
                const unsigned lo = f32_to_bf16(acc[mt][i0 >> 2][i0 & 3] + bz4[i0 >> 2][i0 & 3]);
                const unsigned hi = f32_to_bf16(acc[mt][i1 >> 2][i1 & 3] + bz4[i1 >> 2][i1 & 3]);
                val[e] = lo | (hi << 16);
                xs[mt][i0] = __uint_as_float(lo << 16);
                xs[mt][i1] = __uint_as_float(hi << 16);
            }
            if (live[mt] && (half == 0 || second)) *(ci_u32x4_t*)(p.out + (orow * p.ldo + col + 8 * half) * 2) = val;
        }
    }
    if (!p.stats) return;
    // statistics: this lane's 16 channels are 16 / cg groups of cg channels (cg = 4, 8 or 16).  Pass 1 the tile's group means,
    // pass 2 the sums of squared deviations from them; per lane everything is formed per 4-channel quarter with static
    // register indices and combined by cg with selects.
    const int cg = p.cg, ngl = 16 / cg;
    auto combine = [&](const float (&q4)[4], float (&g4)[4]) {
        if (cg == 4) { g4[0] = q4[0]; g4[1] = q4[1]; g4[2] = q4[2]; g4[3] = q4[3]; }
        else if (cg == 8) { g4[0] = q4[0] + q4[1]; g4[1] = q4[2] + q4[3]; g4[2] = 0.f; g4[3] = 0.f; }
        else { g4[0] = (q4[0] + q4[1]) + (q4[2] + q4[3]); g4[1] = 0.f; g4[2] = 0.f; g4[3] = 0.f; }
    };
    const int gbase = col / cg;                                // first group of this lane within the tile (= within the layer)
    constexpr int RW = 64 * NQ;
    float q4[4], gs[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        q4[j] = 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            if (live[mt]) q4[j] += (xs[mt][4 * j] + xs[mt][4 * j + 1]) + (xs[mt][4 * j + 2] + xs[mt][4 * j + 3]);
    }
    combine(q4, gs);
#pragma unroll
    for (int g = 0; g < 4; ++g) gs[g] = ci_row_sum(gs[g]);
    if (fi == 0) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
            if (g < ngl) s_red[wm * RW + gbase + g] = gs[g];
    }
    const int vh = min(CI_TA, p.H - oh0), vw = min(CI_TB, p.Wd - ow0);
    const float cnt = (float)(vh * vw * cg);                   // values per group in this tile
    __syncthreads();
    float mean[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        float t = 0.f;
        if (g < ngl) {
#pragma unroll
            for (int m = 0; m < NWM; ++m) t += s_red[m * RW + gbase + g];
        }
        mean[g] = t / cnt;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float mj = cg == 4 ? mean[j] : (cg == 8 ? mean[j >> 1] : mean[0]);
        q4[j] = 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            if (live[mt]) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float d = xs[mt][4 * j + e] - mj; q4[j] += d * d; }
            }
        }
    }
    combine(q4, gs);
#pragma unroll
    for (int g = 0; g < 4; ++g) gs[g] = ci_row_sum(gs[g]);
    if (fi == 0) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
            if (g < ngl) s_red[wm * RW + gbase + g] = gs[g];
    }
    __syncthreads();
    const int G = p.Nout / cg;                                 // groups of the layer
    if (tid < RW / cg && tid < G) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < NWM; ++k) t += s_red[k * RW + tid];
        p.stats[(size_t)blockIdx.x * G + tid].y = t;
    }
    if (wm == 0 && fi == 0) {                                  // the means: every wave row computed the same values
#pragma unroll
        for (int g = 0; g < 4; ++g)
            if (g < ngl && gbase + g < G) p.stats[(size_t)blockIdx.x * G + gbase + g].x = mean[g];
    }
}

static int ci_ok(int dtype, int Cin, int H, int W, int Nout, int N, int cg) {
    return dtype == RBVAE_BF16 && Cin >= 1 && Cin <= 4 && Nout >= 8 && Nout <= 256 && Nout % 8 == 0 && N >= 1 && H >= 1 && W >= 1 &&
           (cg == 0 || ((cg == 4 || cg == 8 || cg == 16) && Nout % cg == 0)) && (long)N * H * W * 256 < (1l << 31) &&
           (long)N * Cin * H * W < (1l << 31);
}

}  // namespace rbvae

using namespace rbvae;

extern "C" {

/* 1 when rbvae_conv_in covers the shape (bf16, 3x3 stride 1 pad 1, Cin <= 4, Nout <= 256; cg = 0: no statistics) */
int rbvae_conv_in_ok(int dtype, int Cin, int H, int W, int Nout, int N, int cg) { return ci_ok(dtype, Cin, H, W, Nout, N, cg); }

/* floats of stats_part: (mean, M2) per image, 8 x 16 tile and group */
size_t rbvae_conv_in_stats_floats(int N, int H, int W, int Nout, int cg) {
    return (size_t)2 * N * cdiv(H, CI_TA) * cdiv(W, CI_TB) * (Nout / cg);
}

int rbvae_conv_in(int dtype, const float* x, const void* W, const float* bias, const void* zero_page, void* out, float* stats_part,
                  int cg, int N, int Cin, int H, int Wd, int Nout, int ldo, void* stream) {
    RBVAE_CHECK_ARG(x && W && zero_page && out, "conv_in: null pointer");
    RBVAE_CHECK_ARG(ci_ok(dtype, Cin, H, Wd, Nout, N, stats_part ? cg : 0), "conv_in: shape not covered (bf16, Cin <= 4, Nout <= 256, "
                    "cg in {4, 8, 16}): Cin=%d %dx%d Nout=%d cg=%d", Cin, H, Wd, Nout, cg);
    RBVAE_CHECK_ARG(ldo >= Nout && ldo % 8 == 0, "conv_in: ldo=%d", ldo);
    RBVAE_CHECK_ARG(((uintptr_t)W | (uintptr_t)zero_page | (uintptr_t)out) % 16 == 0 && (!bias || (uintptr_t)bias % 16 == 0) &&
                    (!stats_part || (uintptr_t)stats_part % 8 == 0), "conv_in: pointers must be 16-byte aligned");
    CiArgs a;
    a.x = x; a.W = (const unsigned char*)W; a.bias = bias; a.zero = (const unsigned char*)zero_page; a.out = (unsigned char*)out;
    a.stats = (float2*)stats_part; a.N = N; a.Cin = Cin; a.H = H; a.Wd = Wd; a.Nout = Nout; a.ldo = ldo; a.cg = stats_part ? cg : 0;
    const int blocks = N * cdiv(H, CI_TA) * cdiv(Wd, CI_TB);
    hipStream_t st = (hipStream_t)stream;
#define CI_LAUNCH(C, Q) hipLaunchKernelGGL((conv_in_k<C, Q>), dim3(blocks), dim3(512), 0, st, a)
#define CI_NQ(C) do { if (Nout <= 64) CI_LAUNCH(C, 1); else if (Nout <= 128) CI_LAUNCH(C, 2); else CI_LAUNCH(C, 4); } while (0)
    switch (Cin) {
        case 1: CI_NQ(1); break;
        case 2: CI_NQ(2); break;
        case 3: CI_NQ(3); break;
        default: CI_NQ(4); break;
    }
#undef CI_NQ
#undef CI_LAUNCH
    RBVAE_CHECK_LAUNCH("conv_in");
    return RBVAE_OK;
}

}  // extern "C"
