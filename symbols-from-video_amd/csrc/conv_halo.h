// Shared declarations of the halo-resident 3x3 stride-1 convolution kernels (conv_halo.hip: one tile per workgroup, f32 and
// the debug comparison; conv_halo_ws.hip: persistent workgroups with producer / MFMA wave roles, bf16).
#pragma once
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace rbvae {

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

struct ChArgs {
    const unsigned char* A;        // [Nimg*IH*IW][lda] T
    const unsigned char* W;        // [Nout][9][Kc] T
    unsigned char* Out;            // [Nimg*OH*OW][ldo] T
    const float* bias;             // [Nout] or null
    const unsigned char* addend;   // [Nimg*OH*OW][ldo] T or null (residual)
    const unsigned char* zero;     // >= 128 zero bytes
    const float* gn_scale;         // [Nimg][Kc] or null: input -> swish?(x * scale + shift) while staging
    const float* gn_shift;
    float* stats;                  // null or [m tiles][Nout / cg] float2 (mean, M2) of the stored tile per group
    int gn_swish, stats_cg;        // channels per group of the output statistics
    int Nimg, IH, IW, OH, OW, dh0, dw0;
    int Kc, Nout, lda, ldo;
    int tiles_r, tiles_c, ntn, total;
};

__device__ __forceinline__ void ch_glds16(const void* g, void* lds) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}

template <typename T> struct ChMma;
template <> struct ChMma<bf16_t> {
    static __device__ __forceinline__ void run(f32x4_t& acc, const u32x4_t& rowop, const u32x4_t& colop) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8_t*)&rowop, *(const bf16x8_t*)&colop, acc, 0, 0, 0);
    }
};
template <> struct ChMma<float> {
    static __device__ __forceinline__ void run(f32x4_t& acc, const u32x4_t& rowop, const u32x4_t& colop) {
        const f32x4_t r = *(const f32x4_t*)&rowop, c = *(const f32x4_t*)&colop;
#pragma unroll
        for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(r[q], c[q], acc, 0, 0, 0);
    }
};

constexpr int CH_BM = 256, CH_BN = 128, CH_T = 16;          // tile: 16 x 16 pixels x 128 channels
constexpr int CH_PW = CH_T + 2, CH_NSLOT = CH_PW * CH_PW;    // 18 x 18 patch
constexpr int CH_NSLOT_PAD = 336;                             // multiple of 16
constexpr int CH_PLANE = CH_NSLOT_PAD * 16;                   // bytes per chunk plane
constexpr int CH_KKOFF = 4 * CH_PLANE + 64;                   // chunk 4kk+fg = kk * KKOFF + fg part
constexpr int CH_ABUF = 8 * CH_PLANE + 128;                   // 43136: one patch image (planes + staggers)
constexpr int CH_BBYTES = CH_BN * 128;
constexpr int CH_NA = 6;                                      // register-staged 16-B pieces per thread and slice

__device__ __forceinline__ unsigned ch_plane_off(int chunk) { return (unsigned)(chunk * CH_PLANE + (chunk >> 1) * 32); }


// workgroup barrier that orders LDS traffic only.  __syncthreads() also waits vmcnt(0): behind the epilogue's global stores
// every barrier then costs a full store round trip (the epilogue ran at half the HBM write rate because of it).
__device__ __forceinline__ void ch_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int I, int N, typename F> __device__ __forceinline__ void ch_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        ch_static_for<I + 1, N>(f);
    }
}

// conv_halo_ws.hip: the persistent bf16 kernel (covers = operands below 2 GiB: 32-bit buffer offsets)
bool ch_ws_covers(const ChArgs& a);
int launch_ch_ws(const ChArgs& a, hipStream_t st);

}  // namespace rbvae
