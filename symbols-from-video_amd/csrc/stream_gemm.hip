// Row-streaming GEMM with resident weights, for the 3/4-channel ends of the CNNs:
//
//   Out[m][n] = epi( sum_{k < 64} A[m][k] * W[n][k] ),   M = tens of thousands of pixel rows, N <= 256, bf16
//
// = the first Conv2d forward on its im2col matrix (percep_RBVAE_model.py:51) and the last ConvTranspose2d's
// input gradient (autograd of :82).  These products are HBM bound (64 K FLOP per 640 bytes of a row) and the
// tiled gather GEMM spends them waiting: a workgroup's one K step sits behind its table setup, its first-slice
// latency and its store phase (17-20 us per launch against ~8 us of HBM time).  Here a wave owns whole 16-row
// groups: the A fragment comes straight from global memory in MFMA operand layout (a row is one 128-byte
// line), the 32 KB weight image stays in LDS for the workgroup's life, the next group's rows are fetched
// while this one is multiplied, and results leave as 16-byte chunks without an LDS round trip.
//
// Channel order inside the MFMA tiles is permuted so that a lane ends up with 8 CONSECUTIVE channels of its
// pixel: tile pair (2c, 2c+1), accumulator row j = 4g + r  <->  channel 32c + 8g + 4h + r  (h = tile parity).
// Epilogue semantics (bias, relu, scale, keyed dropout per 16-byte chunk, gate by saved activation, per-block
// column sums for the bias gradient) are those of gather_gemm_k, element for element.
#include "common.h"
#include <stdlib.h>

namespace rbvae {

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

struct SgArgs {
    const unsigned char* A;      // [M][64] bf16
    const unsigned char* W;      // [Nout][64] bf16
    unsigned char* Out;          // [M][ldo] bf16
    const float* bias;           // [Nout] or null
    const unsigned char* gate;   // [M][ldo] bf16 or null: zero the output where gate <= 0
    float* colsum_ws;            // [gridDim.x][Nout] or null
    int M, Nout, ldo, relu, drop_mode;
    float scale;
    unsigned drop_thresh;
    unsigned long long seed;
    const unsigned long long* seed_dev;
};

__device__ __forceinline__ bool bf16_pos(unsigned short v) {
    return (v & 0x8000u) == 0 && (v & 0x7fffu) != 0 && (v & 0x7fffu) <= 0x7f80u;
}

constexpr int SG_THREADS = 256;

// NPAIR = 32-channel tile pairs (Nout <= 32 * NPAIR); GATE / COLSUM compile the gate chunks (32 registers at
// NPAIR 8) and the column-sum accumulators (64) in or out, so no instance spills
template <int NPAIR, bool GATE, bool COLSUM>
__global__ __launch_bounds__(SG_THREADS, 2) void stream_gemm_k(const SgArgs p) {
    constexpr int NT = 2 * NPAIR;
    __shared__ __attribute__((aligned(16))) unsigned char s_w[NT * 16 * 128];   // weight image, fragment-row order
    __shared__ __attribute__((aligned(16))) float s_bias[NT * 16];
    __shared__ float s_red[4][NT * 16];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int fi = lane & 15, fg = lane >> 4;

    // ---- weights -> LDS: image row (ct*16 + j) = W[channel(ct, j)], 16-B chunks XOR-swizzled by (row>>1)&7
    for (int i = tid; i < NT * 16 * 8; i += SG_THREADS) {
        const int r = i >> 3, c = i & 7;
        const int ct = r >> 4, j = r & 15;
        const int ch = 32 * (ct >> 1) + 8 * (j >> 2) + 4 * (ct & 1) + (j & 3);
        u32x4_t v = {0u, 0u, 0u, 0u};
        if (ch < p.Nout) v = *(const u32x4_t*)(p.W + (size_t)ch * 128 + c * 16);
        *(u32x4_t*)(s_w + r * 128 + ((c ^ ((r >> 1) & 7)) * 16)) = v;
    }
    for (int i = tid; i < NT * 16; i += SG_THREADS) s_bias[i] = (p.bias && i < p.Nout) ? p.bias[i] : 0.f;

    const int ngroups = (p.M + 15) >> 4;
    const int gstride = gridDim.x * 4;
    int rg = blockIdx.x * 4 + w;
    // A fragment of a row group: lane (fi, fg) holds A[16*rg + fi][32*kk + 8*fg .. +7], kk = 0, 1
    auto load_a = [&](int g, u32x4_t (&a)[2]) {
        int row = g * 16 + fi;
        row = row < p.M ? row : p.M - 1;
        const unsigned char* src = p.A + (size_t)row * 128 + fg * 16;
        a[0] = *(const u32x4_t*)src;
        a[1] = *(const u32x4_t*)(src + 64);
    };
    u32x4_t a_cur[2], a_nxt[2];
    if (rg < ngroups) load_a(rg, a_cur);
    __syncthreads();

    const int wsw = (fi >> 1) & 7;
    const unsigned char* wbase = s_w + fi * 128;
    DropKey dkey{0u, 0u};
    if (p.drop_mode == 1) dkey = drop_key(p.seed + (p.seed_dev ? p.seed_dev[0] * 0x9E3779B97F4A7C15ull : 0ull));
    float csum[COLSUM ? NPAIR : 1][8];
#pragma unroll
    for (int c = 0; c < (COLSUM ? NPAIR : 1); ++c)
#pragma unroll
        for (int e = 0; e < 8; ++e) csum[c][e] = 0.f;

    for (; rg < ngroups; rg += gstride) {
        const int nxt = rg + gstride;
        if (nxt < ngroups) load_a(nxt, a_nxt);
        const int row = rg * 16 + fi;
        const bool rowok = row < p.M;
        // the gate chunks of this lane's pixel (one per tile pair), in flight under the MFMAs
        u32x4_t gv[GATE ? NPAIR : 1];
        if constexpr (GATE) {
            const unsigned char* gp = p.gate + ((size_t)(rowok ? row : 0) * p.ldo + 8 * fg) * 2;
#pragma unroll
            for (int c = 0; c < NPAIR; ++c)
                gv[c] = (32 * c + 8 * fg < p.Nout) ? *(const u32x4_t*)(gp + 64 * c) : u32x4_t{0u, 0u, 0u, 0u};
        }
        // one tile pair at a time: 4 MFMAs, then the pair's epilogue (lane = pixel fi, channels 32c + 8*fg .. +7).
        // Fully unrolled (static register indices), with a compiler memory barrier per pair: without it every weight
        // fragment read was hoisted to the top (128 registers, spills).
#pragma unroll
        for (int c = 0; c < NPAIR; ++c) {
            asm volatile("" ::: "memory");
            f32x4_t acc[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                acc[h] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const u32x4_t wf = *(const u32x4_t*)(wbase + (2 * c + h) * 2048 + (((4 * kk + fg) ^ wsw) * 16));
                    acc[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8_t*)&wf, *(const bf16x8_t*)&a_cur[kk],
                                                                     acc[h], 0, 0, 0);
                }
            }
            const int col = 32 * c + 8 * fg;
            const float4 b0 = *(const float4*)(s_bias + col), b1 = *(const float4*)(s_bias + col + 4);
            const float bz[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
            unsigned short ev[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float x = (e < 4 ? acc[0][e] : acc[1][e - 4]) + bz[e];
                if (p.relu) x = fmaxf(x, 0.f);
                ev[e] = f32_to_bf16(x * p.scale);
            }
            if (p.drop_mode == 1) {
                const unsigned run = drop_run(dkey, (unsigned long long)row * p.Nout + col);
                const unsigned dm = drop_chunk_mask<8>(run, p.drop_thresh >> 16);
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if ((dm >> e) & 1u) ev[e] = 0;
            }
            if constexpr (GATE) {
                const unsigned short* ge = (const unsigned short*)&gv[c];
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (!bf16_pos(ge[e])) ev[e] = 0;
            }
            if (rowok && col < p.Nout) {
                u32x4_t val;
                val[0] = (unsigned)ev[0] | ((unsigned)ev[1] << 16); val[1] = (unsigned)ev[2] | ((unsigned)ev[3] << 16);
                val[2] = (unsigned)ev[4] | ((unsigned)ev[5] << 16); val[3] = (unsigned)ev[6] | ((unsigned)ev[7] << 16);
                *(u32x4_t*)(p.Out + ((size_t)row * p.ldo + col) * 2) = val;
                if constexpr (COLSUM) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) csum[c][e] += bf16_to_f32(ev[e]);
                }
            }
        }
        a_cur[0] = a_nxt[0];
        a_cur[1] = a_nxt[1];
    }

    if constexpr (COLSUM) {
        // column sums of everything this workgroup stored: over the 16 pixel lanes (shuffles), then the 4 waves (LDS)
#pragma unroll
        for (int c = 0; c < NPAIR; ++c)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float v = csum[c][e];
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) v += __shfl_xor(v, o, 64);
                if (fi == 0) s_red[w][32 * c + 8 * fg + e] = v;
            }
        __syncthreads();
        for (int n = tid; n < p.Nout; n += SG_THREADS)
            p.colsum_ws[(size_t)blockIdx.x * p.Nout + n] = ((s_red[0][n] + s_red[1][n]) + s_red[2][n]) + s_red[3][n];
    }
}

static int sg_blocks(int M) {
    static const int per_wave = getenv("RBVAE_SG_GROUPS") ? atoi(getenv("RBVAE_SG_GROUPS")) : 2;
    static const int cap = getenv("RBVAE_SG_CAP") ? atoi(getenv("RBVAE_SG_CAP")) : 512;
    const int groups = (M + 15) / 16;
    int b = (groups + 4 * per_wave - 1) / (4 * per_wave);   // >= per_wave row groups per wave: the prefetch has something to hide
    return b < 1 ? 1 : (b > cap ? cap : b);
}

}  // namespace rbvae

using namespace rbvae;

extern "C" int rbvae_stream_gemm_blocks(int M) { return sg_blocks(M); }

extern "C" int rbvae_stream_gemm(const void* A, const void* W, void* Out, const float* bias, const void* gate, int M,
                                 int Nout, int ldo, int relu, int drop_mode, float drop_p, float scale,
                                 unsigned long long seed, const unsigned long long* seed_dev, float* colsum_ws,
                                 void* stream) {
    RBVAE_CHECK_ARG(A && W && Out, "stream_gemm: null pointer");
    RBVAE_CHECK_ARG(M > 0 && Nout > 0 && Nout <= 256 && Nout % 8 == 0, "stream_gemm: M=%d Nout=%d (Nout <= 256, % 8)", M, Nout);
    RBVAE_CHECK_ARG(ldo >= Nout && ldo % 8 == 0, "stream_gemm: ldo=%d", ldo);
    RBVAE_CHECK_ARG(drop_mode == 0 || drop_mode == 1, "stream_gemm: drop_mode %d (explicit masks: use rbvae_gather_gemm)", drop_mode);
    RBVAE_CHECK_ARG(((uintptr_t)A | (uintptr_t)W | (uintptr_t)Out | (uintptr_t)gate) % 16 == 0,
                    "stream_gemm: pointers must be 16-byte aligned");
    RBVAE_CHECK_ARG((long)M * ldo < (1l << 31), "stream_gemm: more than 2^31 output elements");
    SgArgs a;
    a.A = (const unsigned char*)A; a.W = (const unsigned char*)W; a.Out = (unsigned char*)Out; a.bias = bias;
    a.gate = (const unsigned char*)gate; a.colsum_ws = colsum_ws; a.M = M; a.Nout = Nout; a.ldo = ldo; a.relu = relu;
    a.drop_mode = drop_mode; a.scale = scale; a.drop_thresh = (unsigned)((double)drop_p * 4294967296.0);
    a.seed = seed; a.seed_dev = seed_dev;
    const dim3 grid(sg_blocks(M));
    hipStream_t st = (hipStream_t)stream;
    const int npair = (Nout + 31) / 32;
#define RBVAE_SG(NP)                                                                                            \
    do {                                                                                                        \
        if (gate && colsum_ws) hipLaunchKernelGGL((stream_gemm_k<NP, true, true>), grid, dim3(SG_THREADS), 0, st, a);   \
        else if (gate) hipLaunchKernelGGL((stream_gemm_k<NP, true, false>), grid, dim3(SG_THREADS), 0, st, a);          \
        else if (colsum_ws) hipLaunchKernelGGL((stream_gemm_k<NP, false, true>), grid, dim3(SG_THREADS), 0, st, a);     \
        else hipLaunchKernelGGL((stream_gemm_k<NP, false, false>), grid, dim3(SG_THREADS), 0, st, a);                   \
    } while (0)
    if (npair <= 2) RBVAE_SG(2);
    else if (npair <= 4) RBVAE_SG(4);
    else RBVAE_SG(8);
#undef RBVAE_SG
    RBVAE_CHECK_LAUNCH("stream_gemm");
    return RBVAE_OK;
}
