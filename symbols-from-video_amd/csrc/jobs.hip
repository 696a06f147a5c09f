// Batched layout jobs: the dozens of small pack / slab-reduce / row-reduce passes of one
// training step run as ONE launch each (grid.y = job), driven by a device-resident table.
#include "common.h"

namespace rbvae {

// 16 x int64 per job
struct Job {
    long type;        // 0 pack3 (f32 -> T, strided scatter), 1 permute_reduce (thread per output), 2 reduce_rows (wave per
                      // output), 3 conv weight [co][ci][kk] f32 -> both GEMM orders [co][t][ci] (dst) and [ci][t][co] (dst2)
    const float* src;
    void* dst;
    long d0, d1, d2;  // logical extents [d0][d1][d2] (the contiguous side is laid out in this order)
    long s0, s1, s2;  // strides on the strided side
    long nslab, slab; // slabs summed in fixed order (types 1, 2)
    long dtype;       // destination type of pack3
    long accumulate;
    float scale;
    int fast;         // which logical index consecutive threads walk (the one whose strided-side stride is 1)
    long inner;       // types 0/1 with fast == 1: a thread walks the whole (short) last index itself
    void* dst2;       // type 3: the second destination
};
static_assert(sizeof(Job) == 16 * 8, "job table stride");

// Type 3: one read of a conv / conv-transpose weight [co][ci][kk] (contiguous f32 rows of TCI*kk values)
// through an LDS tile, written out in both orders the GEMMs use -- [co][t][ci] for the forward GEMM and
// [ci][t][co] for the backward-data GEMM -- as 64- to 128-byte runs on both sides.
constexpr int CP_TCO = 16;                             // co per tile: 128 tiles for a 256 x 256 weight
constexpr int CP_LDSF = CP_TCO * (32 * 9 + 1);        // floats: 32 ci x 9 taps, or 16 ci x 16 taps, +1 pad
template <typename T, unsigned KK>      // KK = taps as a compile-time constant (index math by constant), 0 = runtime
__device__ __forceinline__ void conv_pack_tile(const Job& j, float* tile) {
    const unsigned Co = (unsigned)j.d0, Ci = (unsigned)j.d1, kk = KK ? KK : (unsigned)j.d2;
    const unsigned TCI = kk <= 9 ? 32u : 16u;
    const unsigned row = TCI * kk, pitch = row + 1;
    const unsigned tco = (Co + CP_TCO - 1) / CP_TCO, tci = (Ci + TCI - 1) / TCI;
    T* wf = (T*)j.dst;
    T* wd = (T*)j.dst2;
    for (unsigned t = blockIdx.x; t < tco * tci; t += gridDim.x) {
        const unsigned co0 = (t / tci) * CP_TCO, ci0 = (t % tci) * TCI;
        __syncthreads();
        // the tile's rows, six loads per thread in flight at a time (one load per iteration paid a memory round
        // trip each: 18 of them per tile)
        constexpr unsigned LB = 6;
        for (unsigned i0 = threadIdx.x; i0 < CP_TCO * row; i0 += LB * 256) {
            float v[LB];
#pragma unroll
            for (unsigned u = 0; u < LB; ++u) {
                const unsigned i = i0 + u * 256;
                const unsigned ic = i < CP_TCO * row ? i : 0;
                const unsigned r = ic / row, e = ic - r * row;
                const unsigned c = e / kk;
                const bool ok = i < CP_TCO * row && co0 + r < Co && ci0 + c < Ci;
                const float x = j.src[ok ? ((size_t)(co0 + r) * Ci + ci0) * kk + e : 0];
                v[u] = ok ? x : 0.f;
            }
#pragma unroll
            for (unsigned u = 0; u < LB; ++u) {
                const unsigned i = i0 + u * 256;
                if (i < CP_TCO * row) {
                    const unsigned r = i / row, e = i - r * row;
                    tile[r * pitch + e] = v[u];
                }
            }
        }
        __syncthreads();
        for (unsigned i = threadIdx.x; i < CP_TCO * row; i += 256) {
            // [co][t][ci]: ci fastest
            const unsigned c = i % TCI, q = i / TCI;
            const unsigned tp = q % kk, r = q / kk;
            if (co0 + r < Co && ci0 + c < Ci)
                Elem<T>::store(wf + ((size_t)(co0 + r) * kk + tp) * Ci + ci0 + c, tile[r * pitch + c * kk + tp]);
        }
        for (unsigned i = threadIdx.x; i < CP_TCO * row; i += 256) {
            // [ci][t][co]: co fastest
            const unsigned r = i % CP_TCO, q = i / CP_TCO;
            const unsigned tp = q % kk, c = q / kk;
            if (co0 + r < Co && ci0 + c < Ci)
                Elem<T>::store(wd + ((size_t)(ci0 + c) * kk + tp) * Co + co0 + r, tile[r * pitch + c * kk + tp]);
        }
    }
}

__global__ __launch_bounds__(256) void run_jobs_k(const Job* __restrict__ jobs) {
    __shared__ float cp_tile[CP_LDSF];
    const Job j = jobs[blockIdx.y];
    if (j.type == 3) {
        const bool f32 = j.dtype == RBVAE_F32;
        if (j.d2 == 9) { if (f32) conv_pack_tile<float, 9>(j, cp_tile); else conv_pack_tile<bf16_t, 9>(j, cp_tile); }
        else if (j.d2 == 16) { if (f32) conv_pack_tile<float, 16>(j, cp_tile); else conv_pack_tile<bf16_t, 16>(j, cp_tile); }
        else { if (f32) conv_pack_tile<float, 0>(j, cp_tile); else conv_pack_tile<bf16_t, 0>(j, cp_tile); }
        return;
    }
    const unsigned d0 = (unsigned)j.d0, d1 = (unsigned)j.d1, d2 = (unsigned)j.d2;
    const unsigned n = d0 * d1 * d2;
    // this job may need fewer blocks than the launch provides
    if (blockIdx.x * (j.type == 2 ? 4u : 256u) >= n) return;
    if (j.type == 2) {
        // out[c] = scale * sum_k src[k*slab + c]; one wave per output, lanes stride the slabs, shuffle-reduce
        const int lane = threadIdx.x & 63;
        const unsigned wave = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4;
        float* out = (float*)j.dst;
        const unsigned ns = (unsigned)j.nslab, slab = (unsigned)j.slab;
        for (unsigned c = wave; c < n; c += nw) {
            // eight loads in flight per lane (fixed order of additions: reproducible); one per iteration was a
            // memory round trip each, 64 of them for the 4096 partial rows of a 4-channel bias gradient
            float a = 0.f;
            unsigned k = lane;
            for (; k + 7 * 64 < ns; k += 8 * 64) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = j.src[(size_t)(k + u * 64) * slab + c];
#pragma unroll
                for (int u = 0; u < 8; ++u) a += v[u];
            }
            for (; k < ns; k += 64) a += j.src[(size_t)k * slab + c];
            a = wave_sum(a) * j.scale;
            if (lane == 0) out[c] = j.accumulate ? out[c] + a : a;
        }
        return;
    }
    if (j.inner) {
        // fast == 1 with a short last index (conv weights: [co][ci][kk] on the contiguous side, ci-major rows on
        // the strided side): a thread owns (i0, i1) and walks i2 itself, so the strided side is read / written
        // as whole rows of consecutive i1 and the contiguous side as d2-element runs per thread
        const unsigned ns = (unsigned)j.nslab;
        for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < d0 * d1; i += gridDim.x * 256) {
            const unsigned i1 = i % d1, i0 = i / d1;
            const size_t lin0 = (size_t)i * d2;
            const size_t so0 = i0 * (size_t)j.s0 + i1 * (size_t)j.s1;
            if (j.type == 0) {
                for (unsigned i2 = 0; i2 < d2; ++i2) {
                    const size_t so = so0 + i2 * (size_t)j.s2;
                    if (j.dtype == RBVAE_F32) ((float*)j.dst)[so] = j.src[lin0 + i2];
                    else ((bf16_t*)j.dst)[so] = f32_to_bf16(j.src[lin0 + i2]);
                }
            } else {
                // slabs in fixed order, four at a time: all 4 * d2 loads of a group are independent and in flight
                // together (one slab per iteration was a memory round trip per slab: 7 for the 256-channel weights)
                float a[16];
#pragma unroll
                for (int i2 = 0; i2 < 16; ++i2) a[i2] = 0.f;
                const float* p = j.src + so0;
                unsigned k = 0;
                if (d2 <= 9) {
                    for (; k + 4 <= ns; k += 4, p += 4 * j.slab) {
                        float v[4][9];
#pragma unroll
                        for (int q = 0; q < 4; ++q)
#pragma unroll
                            for (int i2 = 0; i2 < 9; ++i2)
                                v[q][i2] = (unsigned)i2 < d2 ? p[(size_t)q * j.slab + (size_t)i2 * j.s2] : 0.f;
#pragma unroll
                        for (int q = 0; q < 4; ++q)
#pragma unroll
                            for (int i2 = 0; i2 < 9; ++i2) a[i2] += v[q][i2];
                    }
                }
                for (; k < ns; ++k, p += j.slab) {
                    float v[16];
#pragma unroll
                    for (int i2 = 0; i2 < 16; ++i2) v[i2] = (unsigned)i2 < d2 ? p[(size_t)i2 * j.s2] : 0.f;
#pragma unroll
                    for (int i2 = 0; i2 < 16; ++i2) a[i2] += v[i2];
                }
                float* out = (float*)j.dst + lin0;
#pragma unroll
                for (int i2 = 0; i2 < 16; ++i2)
                    if ((unsigned)i2 < d2) out[i2] = j.accumulate ? out[i2] + a[i2] * j.scale : a[i2] * j.scale;
            }
        }
        return;
    }
    // thread index -> (i0, i1, i2) with index `fast` varying fastest
    const int f = j.fast;
    const unsigned ef = f == 0 ? d0 : (f == 1 ? d1 : d2);       // extent of the fastest index
    const unsigned em = f == 2 ? d1 : d2;                        // extent of the middle one
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const unsigned fv = i % ef, r = i / ef;
        const unsigned mid = r % em, slow = r / em;
        const unsigned i0 = f == 0 ? fv : slow;
        const unsigned i1 = f == 1 ? fv : (f == 0 ? slow : mid);
        const unsigned i2 = f == 2 ? fv : mid;
        const size_t lin = ((size_t)i0 * d1 + i1) * d2 + i2;
        const size_t so = i0 * (size_t)j.s0 + i1 * (size_t)j.s1 + i2 * (size_t)j.s2;
        if (j.type == 0) {
            if (j.dtype == RBVAE_F32) ((float*)j.dst)[so] = j.src[lin];
            else ((bf16_t*)j.dst)[so] = f32_to_bf16(j.src[lin]);
        } else {
            const float* p = j.src + so;
            float a = 0.f;
            const unsigned ns = (unsigned)j.nslab;
            unsigned k = 0;
            for (; k + 16 <= ns; k += 16) {         // 16 independent loads in flight, fixed summation order
                float v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) v[u] = p[(size_t)(k + u) * j.slab];
#pragma unroll
                for (int u = 0; u < 16; ++u) a += v[u];
            }
            for (; k + 4 <= ns; k += 4) {           // 4 independent loads in flight, fixed summation order
                const float v0 = p[(size_t)k * j.slab], v1 = p[(size_t)(k + 1) * j.slab];
                const float v2 = p[(size_t)(k + 2) * j.slab], v3 = p[(size_t)(k + 3) * j.slab];
                a = (((a + v0) + v1) + v2) + v3;
            }
            for (; k < ns; ++k) a += p[(size_t)k * j.slab];
            a *= j.scale;
            float* out = (float*)j.dst;
            out[lin] = j.accumulate ? out[lin] + a : a;
        }
    }
}

}  // namespace rbvae

using namespace rbvae;

extern "C" int rbvae_run_jobs(const void* jobs_dev, int njobs, int blocks_per_job, void* stream) {
    RBVAE_CHECK_ARG(jobs_dev && njobs > 0 && blocks_per_job > 0, "run_jobs: bad arguments");
    RBVAE_CHECK_ARG(njobs <= 65535, "run_jobs: too many jobs");
    hipLaunchKernelGGL(run_jobs_k, dim3(blocks_per_job, njobs), dim3(256), 0, (hipStream_t)stream, (const Job*)jobs_dev);
    RBVAE_CHECK_LAUNCH("run_jobs");
    return RBVAE_OK;
}
