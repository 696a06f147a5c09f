// Batched layout jobs: the dozens of small pack / slab-reduce / row-reduce passes of one
// training step run as ONE launch each (grid.y = job), driven by a device-resident table.
#include "common.h"

namespace rbvae {

// 16 x int64 per job
struct Job {
    long type;        // 0 pack3 (f32 -> T, strided scatter), 1 permute_reduce (thread per output), 2 reduce_rows (wave per output)
    const float* src;
    void* dst;
    long d0, d1, d2;  // logical extents [d0][d1][d2] (the contiguous side is laid out in this order)
    long s0, s1, s2;  // strides on the strided side
    long nslab, slab; // slabs summed in fixed order (types 1, 2)
    long dtype;       // destination type of pack3
    long accumulate;
    float scale;
    int fast;         // which logical index consecutive threads walk (the one whose strided-side stride is 1)
    long pad1, pad2;
};
static_assert(sizeof(Job) == 16 * 8, "job table stride");

__global__ __launch_bounds__(256) void run_jobs_k(const Job* __restrict__ jobs) {
    const Job j = jobs[blockIdx.y];
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
            float a = 0.f;
            for (unsigned k = lane; k < ns; k += 64) a += j.src[(size_t)k * slab + c];
            a = wave_sum(a) * j.scale;
            if (lane == 0) out[c] = j.accumulate ? out[c] + a : a;
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
