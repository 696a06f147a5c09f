// Shared host/device helpers for librbvae_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/rbvae_hip.h"

namespace rbvae {

// thread-local message behind rbvae_last_error()
char* err_buf();
int fail(int code, const char* fmt, ...);

#define RBVAE_CHECK_ARG(cond, ...) \
    do { if (!(cond)) return ::rbvae::fail(RBVAE_E_INVALID, __VA_ARGS__); } while (0)

#define RBVAE_CHECK_LAUNCH(name) \
    do { hipError_t e__ = hipGetLastError(); \
         if (e__ != hipSuccess) return ::rbvae::fail(RBVAE_E_LAUNCH, "%s: %s", name, hipGetErrorString(e__)); \
    } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ---- device helpers -------------------------------------------------------
typedef unsigned short bf16_t;   // raw bf16 bits

__device__ __forceinline__ float bf16_to_f32(bf16_t v) {
    return __uint_as_float(((unsigned)v) << 16);
}
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    // plain cast: hipcc emits v_cvt_pk_bf16_f32 (round-to-nearest-even, NaN stays NaN)
    __hip_bfloat16 b = __float2bfloat16(f);
    return *reinterpret_cast<bf16_t*>(&b);
}

template <typename T> struct Elem;
template <> struct Elem<float> {
    static __device__ __forceinline__ float load(const float* p) { return *p; }
    static __device__ __forceinline__ void store(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
    static __device__ __forceinline__ float load(const bf16_t* p) { return bf16_to_f32(*p); }
    static __device__ __forceinline__ void store(bf16_t* p, float v) { *p = f32_to_bf16(v); }
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Sum over the whole block in a FIXED order (bitwise reproducible).  `red` holds
// one float per wave; every thread returns the total.
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) red[wid] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// torch.optim.Adam on one element (percep_RBVAE_train.py:753,553): g is scaled by gscale, m / v are updated in place,
// the new weight is returned.  step_size = lr / (1 - b1^t), bc2_sqrt = sqrt(1 - b2^t).  Explicit fma placement: every
// kernel that updates parameters (adam_k, the fused update jobs) rounds identically.
__device__ __forceinline__ float adam_update(float w, float g, float& m, float& v, float one_m_b1, float b2, float one_m_b2,
                                             float eps, float gscale, float step_size, float bc2_sqrt) {
    const float gi = g * gscale;
    const float mi = fmaf(one_m_b1, gi - m, m);
    const float vi = fmaf(one_m_b2 * gi, gi, v * b2);
    m = mi;
    v = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    return fmaf(-step_size, mi / denom, w);
}

// KL(Bernoulli(q) || Bernoulli(p)) per element, q = sigmoid(v) (percep_RBVAE_train.py:59-70); lp = log p, l1p = log(1-p)
__device__ __forceinline__ float kl_elem(float v, float lp, float l1p, float eps, int clamp) {
    float q = sigmoidf_(v);
    if (clamp) q = fminf(fmaxf(q, eps), 1.0f - eps);
    return q * (logf(q + eps) - lp) + (1.0f - q) * (logf((1.0f - q) + eps) - l1p);
}

// d kl_elem / d v
__device__ __forceinline__ float kl_elem_grad(float v, float lp, float l1p, float eps, int clamp) {
    const float s = sigmoidf_(v);
    float q = s;
    bool pass = true;
    if (clamp) {
        pass = (s >= eps) && (s <= 1.0f - eps);
        q = fminf(fmaxf(s, eps), 1.0f - eps);
    }
    if (!pass) return 0.f;
    const float omq = 1.0f - q;
    const float dq = (logf(q + eps) - lp) + q / (q + eps) - (logf(omq + eps) - l1p) - omq / (omq + eps);
    return dq * s * (1.0f - s);
}

// Counter-based uniform bits for dropout: one 32-bit draw per element index.
// (squares-style mixing of (seed, index); statistical quality is ample for a keep-mask)
__device__ __forceinline__ unsigned hash_u32(unsigned long long seed, unsigned long long idx) {
    unsigned long long x = (idx + 1) * 0x9E3779B97F4A7C15ull + seed;
    x ^= x >> 32; x *= 0xD6E8FEB86659FD93ull;
    x ^= x >> 32; x *= 0xD6E8FEB86659FD93ull;
    x ^= x >> 32;
    return (unsigned)x;
}

// Dropout keep-mask bits: 32 uniform bits per element from a per-launch key and the element's index.
// The 64-bit hash above costs ~230 cycles per wave-instruction-element (three 64-bit multiplies at quarter
// rate) and, evaluated once per activation element, WAS the conv1 forward kernel's run time; this is the
// two-round 32-bit finaliser "lowbias32" (two quarter-rate multiplies) keyed by a 64-bit mix of the seed
// that is computed once per thread.
struct DropKey { unsigned k0, k1; };
__device__ __forceinline__ DropKey drop_key(unsigned long long seed) {
    unsigned long long x = seed * 0x9E3779B97F4A7C15ull + 0xD6E8FEB86659FD93ull;
    x ^= x >> 32; x *= 0xD6E8FEB86659FD93ull;
    x ^= x >> 32;
    return DropKey{(unsigned)x, (unsigned)(x >> 32)};
}
// start of a run of consecutive elements: fold the key and the high index bits in once
__device__ __forceinline__ unsigned drop_run(DropKey k, unsigned long long idx) {
    return (unsigned)idx + k.k0 + (unsigned)(idx >> 32) * k.k1;
}
__device__ __forceinline__ unsigned drop_bits(unsigned run, int e) {
    unsigned x = run + (unsigned)e;
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= x >> 15; x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}
// One 16-byte store chunk (EC consecutive elements) at a time: the chunk's first index is hashed once
// (drop_bits above: two quarter-rate multiplies) and the state then walks a xorshift32 sequence, 16 bits per
// element.  A full hash per element cost ~64 cycles per wave-instruction-element, ~8 us of the 23 us conv1
// forward at 16.7 M activations; this is ~4x cheaper.  Element e of the chunk is dropped iff its 16 bits are
// below thresh16 = round(p * 65536) (p = 0.2 -> 0.199997).
template <int EC>
__device__ __forceinline__ unsigned drop_chunk_mask(unsigned run, unsigned thresh16) {
    unsigned s = drop_bits(run, 0);
    unsigned drop = 0;                      // bit e set = drop element e
#pragma unroll
    for (int e = 0; e < EC; e += 2) {
        drop |= ((s & 0xffffu) < thresh16 ? 1u : 0u) << e;
        drop |= ((s >> 16) < thresh16 ? 1u : 0u) << (e + 1);
        if (e + 2 < EC) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; }
    }
    return drop;
}
// The same decisions applied in place, without the detour through a bit mask (compare + select per element):
// x[0..EC) are f32 values / w[0..EC/2) are words of two packed 16-bit elements.
template <int EC>
__device__ __forceinline__ void drop_chunk_zero_f32(unsigned run, unsigned thresh16, float* x) {
    unsigned s = drop_bits(run, 0);
#pragma unroll
    for (int e = 0; e < EC; e += 2) {
        x[e] = (s & 0xffffu) < thresh16 ? 0.f : x[e];
        x[e + 1] = (s >> 16) < thresh16 ? 0.f : x[e + 1];
        if (e + 2 < EC) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; }
    }
}
// Two packed 16-bit elements at a time (the same decisions: element dropped iff its 16 bits < thresh16): keep = min(sat(s -
// (thresh16 - 1)), 1) per half is 1 iff s >= thresh16, and the word is multiplied by it -- three packed instructions per
// pair where the compare / select form takes eight (inline asm: the compiler turns the C++ form back into compares).  The
// epilogues share their SIMD's issue port with the matrix-core waves of the next tile; their instruction count is run time.
template <int EC>
__device__ __forceinline__ void drop_chunk_zero_b16(unsigned run, unsigned thresh16, unsigned* w) {
    if (thresh16 == 0) return;                       // p = 0: nothing is dropped (thresh16 - 1 would wrap)
    unsigned s = drop_bits(run, 0);
    const unsigned tm1 = (thresh16 - 1u) * 0x00010001u, one2 = 0x00010001u;
#pragma unroll
    for (int e = 0; e < EC; e += 2) {
        unsigned d, k, v = w[e >> 1];
        asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(d) : "v"(s), "v"(tm1));
        asm("v_pk_min_u16 %0, %1, %2" : "=v"(k) : "v"(d), "v"(one2));
        asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(v) : "v"(v), "v"(k));
        w[e >> 1] = v;
        if (e + 2 < EC) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; }
    }
}

}  // namespace rbvae
