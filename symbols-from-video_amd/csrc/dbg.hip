// Hardware-mapping probes (rbvae_dbg_*): tiny kernels that pin the MFMA operand /
// accumulator lane maps, the LDS-DMA destination order and the transposed LDS read
// that conv_gemm.hip relies on.  Exercised by tests/test_hw_maps.py on the GPU.
#include "common.h"

namespace rbvae {
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

// A: [16][32] bf16 row-major, B: [32][16] bf16 row-major, D: [16][16] f32
__global__ void dbg_mfma_bf16_k(const bf16_t* A, const bf16_t* B, float* D) {
    const int l = threadIdx.x, m = l & 15, g = l >> 4;
    bf16x8_t a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = (short)A[m * 32 + 8 * g + j];
        b[j] = (short)B[(8 * g + j) * 16 + m];
    }
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * g + r) * 16 + m] = acc[r];
}
// A: [16][4] f32, B: [4][16] f32
__global__ void dbg_mfma_f32_k(const float* A, const float* B, float* D) {
    const int l = threadIdx.x, m = l & 15, g = l >> 4;
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[m * 4 + g], B[g * 16 + m], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * g + r) * 16 + m] = acc[r];
}
// LDS-DMA: 64 lanes x 16 B from per-lane source addresses; dump LDS linearly.
__global__ void dbg_glds_k(const unsigned* src, const int* lane_src_chunk, unsigned* out) {
    __shared__ __attribute__((aligned(16))) unsigned lds[256];
    const int l = threadIdx.x;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 4 * lane_src_chunk[l]),
                                     (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
    __syncthreads();
    for (int i = l; i < 256; i += 64) out[i] = lds[i];
}
// Transposed read: image [32 rows][64 cols] bf16 (128-B rows) in LDS; lane l reads at
// (row = rowsel[l], col = colsel[l]) and dumps its 4 values.
__global__ void dbg_tr16_k(const bf16_t* img, const int* rowsel, const int* colsel, bf16_t* out) {
    __shared__ __attribute__((aligned(16))) bf16_t lds[32 * 64];
    const int l = threadIdx.x;
    for (int i = l; i < 32 * 64; i += 64) lds[i] = img[i];
    __syncthreads();
    s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4_t*)(lds + rowsel[l] * 64 + colsel[l]));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (bf16_t)v[j];
}
}  // namespace rbvae

using namespace rbvae;
extern "C" {
int rbvae_dbg_mfma_bf16(const void* A, const void* B, float* D, void* stream) {
    hipLaunchKernelGGL(dbg_mfma_bf16_k, dim3(1), dim3(64), 0, (hipStream_t)stream, (const bf16_t*)A,
                       (const bf16_t*)B, D);
    RBVAE_CHECK_LAUNCH("dbg_mfma_bf16");
    return RBVAE_OK;
}
int rbvae_dbg_mfma_f32(const float* A, const float* B, float* D, void* stream) {
    hipLaunchKernelGGL(dbg_mfma_f32_k, dim3(1), dim3(64), 0, (hipStream_t)stream, A, B, D);
    RBVAE_CHECK_LAUNCH("dbg_mfma_f32");
    return RBVAE_OK;
}
int rbvae_dbg_glds(const void* src, const int* lane_src_chunk, void* out, void* stream) {
    hipLaunchKernelGGL(dbg_glds_k, dim3(1), dim3(64), 0, (hipStream_t)stream, (const unsigned*)src,
                       lane_src_chunk, (unsigned*)out);
    RBVAE_CHECK_LAUNCH("dbg_glds");
    return RBVAE_OK;
}
int rbvae_dbg_tr16(const void* img, const int* rowsel, const int* colsel, void* out, void* stream) {
    hipLaunchKernelGGL(dbg_tr16_k, dim3(1), dim3(64), 0, (hipStream_t)stream, (const bf16_t*)img, rowsel, colsel,
                       (bf16_t*)out);
    RBVAE_CHECK_LAUNCH("dbg_tr16");
    return RBVAE_OK;
}
}

// which kernel rbvae_conv3x3_halo (librbvae_hip) runs for bf16: see include/rbvae_dbg.h; returns the previous value
namespace rbvae { extern int ch_variant; }
extern "C" int rbvae_dbg_conv_halo_variant(int v) {
    const int old = rbvae::ch_variant;
    rbvae::ch_variant = v;
    return old;
}

// which kernels rbvae_lstm_pair_fwd / _bwd (librbvae_hip) run at L == 32: 1 one thread per hidden unit (default), 0 one per
// gate row; returns the previous value
namespace rbvae { extern int lstm_unit_threads; }
extern "C" int rbvae_dbg_lstm_unit_threads(int v) {
    const int old = rbvae::lstm_unit_threads;
    rbvae::lstm_unit_threads = v;
    return old;
}
