// The two K = 64 (or 128) products around the latent bottleneck, bf16 storage:
//
//   Out[m][n] = sum_{k < K} A[m][k] * W[n][k] (+ bias[n]),   M = a few hundred frames, N = thousands of features
//
// = the decoder's fc forward (Linear(latent -> C3*h3*w3), percep_RBVAE_model.py:74 on the zero-padded codes) and the
// input gradient of the encoder's fc (autograd of :61).  They are 0.13 GFLOP and 2.5 MB each -- latency, not work:
// through the tiled gather GEMM (64 workgroups, index tables, an LDS ring for ONE K step, an LDS round trip for the
// stores) a launch took 10-11 us on the step's critical chain.  Here a workgroup owns 16 output columns of 256 rows:
// every operand goes from global memory straight into the MFMA operand layout (a row of A or W is one 128-byte
// line; lane (i, g) takes bytes 16 g .. 16 g + 15 of row i of each half), 8 MFMAs per wave, and the results leave
// from the accumulators.  No LDS, no barrier (column sums apart), N/16 workgroups.
//
// Arithmetic is that of gather_gemm_k element for element (two 32-deep MFMAs in k order, + bias, round to bf16);
// the optional column sums of the STORED values (the bias gradient of the layer below the encoder's fc) come out
// per 128-row tile in gather_gemm_k's layout [tile][N], summed in a different (fixed) order.
#include "common.h"

namespace rbvae {

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

struct FcArgs {
    const unsigned char* A;      // [M][lda] bf16, K used columns
    const unsigned char* W;      // [N][K] bf16
    unsigned char* Out;          // [M][ldo] bf16
    const float* bias;           // [N] or null
    float* colsum_ws;            // [ceil(M/128)][N] or null
    int M, N, lda, ldo;
};

template <int CTRL> __device__ __forceinline__ float fc_dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
// sum over the 16 lanes of a DPP row (lane & 15), every lane gets the total
__device__ __forceinline__ float fc_row_sum(float v) {
    v += fc_dpp<0x128>(v);     // row_ror:8
    v += fc_dpp<0x124>(v);     // row_ror:4
    v += fc_dpp<0x4E>(v);      // quad_perm [2,3,0,1]
    v += fc_dpp<0xB1>(v);      // quad_perm [1,0,3,2]
    return v;
}

// grid (N / 16, ceil(M / 256)), 256 threads: wave w owns rows 64 w .. 64 w + 63 of the workgroup's 256.
// KH = 32-deep MFMA steps: K = 64 (latent_dim <= 64) or 128 (latent_dim 65 .. 128).
template <int KH>
__global__ __launch_bounds__(256) void fc_gemm_k(const FcArgs p) {
    constexpr int K = 32 * KH;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int fi = lane & 15, fg = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const int r0 = blockIdx.y * 256 + 64 * w;
    // every load of the kernel is issued before the first use
    u32x4_t wf[KH], af[4][KH];
    const unsigned char* wp = p.W + ((size_t)(n0 + fi) * K + 8 * fg) * 2;
#pragma unroll
    for (int h = 0; h < KH; ++h) wf[h] = *(const u32x4_t*)(wp + 64 * h);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int row = min(r0 + 16 * mt + fi, p.M - 1);               // clamped: rows past M are computed, not stored
        const unsigned char* ap = p.A + ((size_t)row * p.lda + 8 * fg) * 2;
#pragma unroll
        for (int h = 0; h < KH; ++h) af[mt][h] = *(const u32x4_t*)(ap + 64 * h);
    }
    f32x4_t bz = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (p.bias) bz = *(const f32x4_t*)(p.bias + n0 + 4 * fg);
    f32x4_t acc[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) acc[mt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int h = 0; h < KH; ++h)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
            acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8_t*)&wf[h], *(const bf16x8_t*)&af[mt][h], acc[mt], 0, 0, 0);
    // lane = row fi of tile mt, columns n0 + 4 fg .. + 3
    float cs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int row = r0 + 16 * mt + fi;
        unsigned short e[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) e[r] = f32_to_bf16(acc[mt][r] + bz[r]);
        if (row < p.M) {
            uint2 pk;
            pk.x = (unsigned)e[0] | ((unsigned)e[1] << 16);
            pk.y = (unsigned)e[2] | ((unsigned)e[3] << 16);
            *(uint2*)(p.Out + ((size_t)row * p.ldo + n0 + 4 * fg) * 2) = pk;
#pragma unroll
            for (int r = 0; r < 4; ++r) cs[r] += __uint_as_float((unsigned)e[r] << 16);
        }
    }
    if (p.colsum_ws) {
        __shared__ float red[4][16];
#pragma unroll
        for (int r = 0; r < 4; ++r) cs[r] = fc_row_sum(cs[r]);
        if (fi == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) red[w][4 * fg + r] = cs[r];
        }
        __syncthreads();
        if (tid < 32) {
            const int t = tid >> 4, c = tid & 15;                       // 128-row tile t of this workgroup: waves 2t, 2t+1
            if (blockIdx.y * 256 + 128 * t < p.M)
                p.colsum_ws[((size_t)blockIdx.y * 2 + t) * p.N + n0 + c] = red[2 * t][c] + red[2 * t + 1][c];
        }
    }
}

// The same product for wide layers (N a multiple of 128 with at least 256 column groups: the 56 320 / 65 536 features of the
// native and 256 x 256 frame sizes).  With 16 columns per workgroup a row of the output receives 32 bytes per workgroup --
// 14 MB written as 450 000 quarter lines, 30 us where the bytes take 4.  Here a wave owns 64 rows x 64 columns: four
// column tiles whose W rows are dealt so that a lane's accumulators of the four tiles are 16 CONSECUTIVE columns (tile nt,
// MFMA row 4 g + r = column 16 g + 4 nt + r), i.e. two 16-byte stores per row and lane and one whole 128-byte line per row
// and wave.  Workgroup = 128 rows x 128 columns (wave w: rows 64 (w & 1), columns 64 (w >> 1)).  Same MFMAs in the same k
// order per output, same rounding; the column sums pair the two row waves of a 128-row tile as fc_gemm_k does.
template <int KH>
__global__ __launch_bounds__(256) void fc_gemm_wide_k(const FcArgs p) {
    constexpr int K = 32 * KH;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int fi = lane & 15, fg = lane >> 4;
    const int n0 = blockIdx.x * 128 + 64 * (w >> 1);
    const int r0 = blockIdx.y * 128 + 64 * (w & 1);
    u32x4_t wf[4][KH], af[4][KH];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int col = n0 + 16 * (fi >> 2) + 4 * nt + (fi & 3);       // the column MFMA row fi of tile nt computes
        const unsigned char* wp = p.W + ((size_t)col * K + 8 * fg) * 2;
#pragma unroll
        for (int h = 0; h < KH; ++h) wf[nt][h] = *(const u32x4_t*)(wp + 64 * h);
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int row = min(r0 + 16 * mt + fi, p.M - 1);               // clamped: rows past M are computed, not stored
        const unsigned char* ap = p.A + ((size_t)row * p.lda + 8 * fg) * 2;
#pragma unroll
        for (int h = 0; h < KH; ++h) af[mt][h] = *(const u32x4_t*)(ap + 64 * h);
    }
    // lane's columns: n0 + 16 fg + 4 nt + r
    f32x4_t bz[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        bz[nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        if (p.bias) bz[nt] = *(const f32x4_t*)(p.bias + n0 + 16 * fg + 4 * nt);
    }
    f32x4_t acc[4][4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int h = 0; h < KH; ++h)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8_t*)&wf[nt][h], *(const bf16x8_t*)&af[mt][h],
                                                                      acc[mt][nt], 0, 0, 0);
    float cs[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) cs[c] = 0.f;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int row = r0 + 16 * mt + fi;
        unsigned short e[16];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) e[4 * nt + r] = f32_to_bf16(acc[mt][nt][r] + bz[nt][r]);
        if (row < p.M) {
            u32x4_t lo, hi;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                lo[q] = (unsigned)e[2 * q] | ((unsigned)e[2 * q + 1] << 16);
                hi[q] = (unsigned)e[8 + 2 * q] | ((unsigned)e[8 + 2 * q + 1] << 16);
            }
            unsigned char* op = p.Out + ((size_t)row * p.ldo + n0 + 16 * fg) * 2;
            *(u32x4_t*)op = lo;
            *(u32x4_t*)(op + 16) = hi;
#pragma unroll
            for (int c = 0; c < 16; ++c) cs[c] += __uint_as_float((unsigned)e[c] << 16);
        }
    }
    if (p.colsum_ws) {
        __shared__ float red[4][64];
#pragma unroll
        for (int c = 0; c < 16; ++c) cs[c] = fc_row_sum(cs[c]);
        if (fi == 0) {
#pragma unroll
            for (int c = 0; c < 16; ++c) red[w][16 * fg + c] = cs[c];
        }
        __syncthreads();
        if (tid < 128 && blockIdx.y * 128 < p.M) {
            const int half = tid >> 6, c = tid & 63;                    // columns 64 half + c: waves 2 half (rows 0-63), 2 half + 1
            p.colsum_ws[(size_t)blockIdx.y * p.N + blockIdx.x * 128 + 64 * half + c] = red[2 * half][c] + red[2 * half + 1][c];
        }
    }
}

}  // namespace rbvae

using namespace rbvae;

extern "C" {

/* 1 when rbvae_fc_gemm covers the product (bf16, K = 64 or 128, N a multiple of 16) */
int rbvae_fc_gemm_ok(int dtype, int M, int K, int N, int lda, int ldo) {
    return dtype == RBVAE_BF16 && M >= 1 && (K == 64 || K == 128) && N >= 16 && N % 16 == 0 && lda >= K && lda % 8 == 0 && ldo >= N &&
           ldo % 4 == 0 && (long)cdiv(M, 256) < 65536;
}

int rbvae_fc_gemm(int dtype, const void* A, const void* W, void* Out, const float* bias, float* colsum_ws, int M, int K,
                  int N, int lda, int ldo, void* stream) {
    RBVAE_CHECK_ARG(A && W && Out, "fc_gemm: null pointer");
    RBVAE_CHECK_ARG(rbvae_fc_gemm_ok(dtype, M, K, N, lda, ldo), "fc_gemm: shape outside the kernel (bf16, K = 64 or 128, N %% 16 == 0): "
                    "dtype=%d M=%d K=%d N=%d lda=%d ldo=%d", dtype, M, K, N, lda, ldo);
    RBVAE_CHECK_ARG(((uintptr_t)A | (uintptr_t)W) % 16 == 0 && (uintptr_t)Out % 8 == 0 && (!bias || (uintptr_t)bias % 16 == 0),
                    "fc_gemm: A / W / bias must be 16-byte aligned, Out 8-byte aligned");
    FcArgs a;
    a.A = (const unsigned char*)A; a.W = (const unsigned char*)W; a.Out = (unsigned char*)Out; a.bias = bias;
    a.colsum_ws = colsum_ws; a.M = M; a.N = N; a.lda = lda; a.ldo = ldo;
    const bool wide = K == 64 && N % 128 == 0 && (long)(N / 128) * cdiv(M, 128) >= 256 && ldo % 8 == 0 && (uintptr_t)Out % 16 == 0;
    if (wide) hipLaunchKernelGGL(fc_gemm_wide_k<2>, dim3(N / 128, cdiv(M, 128)), dim3(256), 0, (hipStream_t)stream, a);
    else if (K == 64) hipLaunchKernelGGL(fc_gemm_k<2>, dim3(N / 16, cdiv(M, 256)), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(fc_gemm_k<4>, dim3(N / 16, cdiv(M, 256)), dim3(256), 0, (hipStream_t)stream, a);
    RBVAE_CHECK_LAUNCH("fc_gemm");
    return RBVAE_OK;
}

}  // extern "C"
