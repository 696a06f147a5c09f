// Batched layout jobs: the dozens of small pack / slab-reduce / row-reduce passes of one
// training step run as ONE launch each (grid.y = job), driven by a device-resident table.
#include "common.h"

namespace rbvae {

// 16 x int64 per job
struct Job {
    long type;        // 0 pack3 (f32 -> T, strided scatter), 1 permute_reduce (thread per output), 2 reduce_rows (wave per
                      // output), 3 conv weight [co][ci][kk] f32 -> both GEMM orders [co][t][ci] (dst) and [ci][t][co] (dst2),
                      // 4 conv weight-gradient slabs [ks][co][t][ci] -> [co][ci][kk] (d0 = co, d1 = ci, d2 = kk),
                      // 5 batch gather: dst[r] = src[plan[(*counter % d1) * d0 + r]] (d0 rows of d2 float4; s0 = plan, s1 = counter
                      //   pointer or 0, s2 = table rows),
                      // 6 / 7 Adam update of a parameter tensor (+ its packed copies for 6), `inner` = AdamCtx*; kind 3 with
                      //   `inner` set updates the conv weight rows it packs
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

// Optimiser context of the fused update jobs (kinds 3 with `inner` set, 6, 7): torch.optim.Adam on the flat buffers, the
// arithmetic of adam_k (layout.hip) element for element.  w / g / m / v are the BASES of the flat parameter, gradient
// and moment buffers; a job's src points into w and addresses the other three at the same offset.
struct AdamCtx {
    float* w; const float* g; float* m; float* v;
    const float* hyper;                 // [step size lr / (1 - b1^t), sqrt(1 - b2^t)] (rbvae_combine_losses)
    float one_m_b1, b2, one_m_b2, eps, gscale, pad_;
};
static_assert(sizeof(AdamCtx) == 8 * 8, "adam context layout (8 x int64 on the host side)");

__device__ __forceinline__ float adam_value(float w, float g, float& m, float& v, const AdamCtx& c, float step_size,
                                            float bc2_sqrt) {
    return adam_update(w, g, m, v, c.one_m_b1, c.b2, c.one_m_b2, c.eps, c.gscale, step_size, bc2_sqrt);
}
// update element `off` of the flat buffers in place, return the new weight
__device__ __forceinline__ float adam_at(const AdamCtx& c, long off, float step_size, float bc2_sqrt) {
    float m = c.m[off], v = c.v[off];
    const float w = adam_value(c.w[off], c.g[off], m, v, c, step_size, bc2_sqrt);
    c.m[off] = m; c.v[off] = v; c.w[off] = w;
    return w;
}

// Type 3: a conv / conv-transpose weight [co][ci][kk] (f32 master, rows of ci*kk contiguous values) -> both GEMM orders,
// [co][t][ci] (dst, forward GEMM) and [ci][t][co] (dst2, backward-data GEMM), in the storage type T.
// A workgroup owns NCO = 16 / sizeof(T) output channels and ALL of ci x kk: the rows come in by 16-byte loads, sit in
// LDS in T, and leave as 16-byte stores on both sides -- [co][t][ci] as whole 16-byte runs of ci, [ci][t][co] as the
// 16-byte run of the workgroup's NCO channels (the per-element 2-byte stores of the first version moved 30 MB in
// 15 us).  Weights with more than CPK_MAXROW values per channel take the workgroup's rows in pieces of ci.
#ifndef CPK_CIB
#define CPK_CIB 64
#endif
constexpr int CPK_MAXROW = 2304;                       // 256 ci x 9 taps (or 144 ci x 16 taps)
constexpr int CPK_LDS_BYTES = 16 * (CPK_MAXROW + 8);  // NCO rows of T, NCO * sizeof(T) = 16; +8 elements of pad per row
template <typename T>
__device__ __forceinline__ void conv_pack_rows(const Job& j, unsigned char* lds_raw, unsigned bidx, unsigned nblk) {
    constexpr unsigned NCO = 16 / sizeof(T), NV = 16 / sizeof(T);
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    T* tile = (T*)lds_raw;
    const unsigned Co = (unsigned)j.d0, Ci = (unsigned)j.d1, kk = (unsigned)j.d2;
    // ci block: a multiple of NV such that cib * kk <= CPK_MAXROW
    unsigned cib = (CPK_MAXROW / kk) / NV * NV;
    if (cib > Ci) cib = Ci;
    // pieces of 64 input channels: 4x the workgroups for a 256-channel weight (128 per weight; with whole rows the
    // fused optimiser update ran on 32 workgroups per weight and took 35 us longer than the unfused step)
    if (cib > CPK_CIB && Ci % CPK_CIB == 0) cib = CPK_CIB;
    const unsigned ncb = (Ci + cib - 1) / cib, ngr = (Co + NCO - 1) / NCO;
    const unsigned pitch = CPK_MAXROW + 8;
    T* wf = (T*)j.dst;
    T* wd = (T*)j.dst2;
    const bool vec_ok = (Ci % NV == 0) && (Co % NCO == 0) && ((cib * kk) % 4 == 0) && ((Ci * kk) % 4 == 0);
    // fused optimiser update (j.inner = AdamCtx*): the block updates exactly the master rows it packs, so the packed
    // copies are written from the new values without a second pass (and without a launch of their own)
    const AdamCtx* actx = (const AdamCtx*)j.inner;
    float step_size = 0.f, bc2_sqrt = 1.f;
    if (actx) { step_size = actx->hyper[0]; bc2_sqrt = actx->hyper[1]; }
    for (unsigned b = bidx; b < ngr * ncb; b += nblk) {
        const unsigned co0 = (b / ncb) * NCO, ci0 = (b % ncb) * cib;
        const unsigned cn = min(cib, Ci - ci0), row = cn * kk;       // this piece: cn input channels
        __syncthreads();
        // ---- in: NCO rows of `row` consecutive floats each
        if (vec_ok && actx) {
            const unsigned r4 = row / 4, tot = NCO * r4;
            constexpr unsigned AB = 3;                                     // 3 x (w, g, m, v) float4 in flight per thread
            for (unsigned i0 = threadIdx.x; i0 < tot; i0 += AB * 256) {
                float4 w4[AB], g4[AB], m4[AB], v4[AB];
                long off[AB];
#pragma unroll
                for (unsigned u = 0; u < AB; ++u) {
                    const unsigned i = i0 + u * 256, ic = i < tot ? i : 0;
                    const unsigned r = ic / r4, q = ic - r * r4;
                    off[u] = (j.src - actx->w) + ((long)(co0 + r) * Ci + ci0) * kk + 4 * q;
                    w4[u] = *(const float4*)(actx->w + off[u]); g4[u] = *(const float4*)(actx->g + off[u]);
                    m4[u] = *(const float4*)(actx->m + off[u]); v4[u] = *(const float4*)(actx->v + off[u]);
                }
#pragma unroll
                for (unsigned u = 0; u < AB; ++u) {
                    const unsigned i = i0 + u * 256;
                    if (i < tot) {
                        float4 w = w4[u], m = m4[u], v = v4[u];
                        w.x = adam_value(w.x, g4[u].x, m.x, v.x, *actx, step_size, bc2_sqrt);
                        w.y = adam_value(w.y, g4[u].y, m.y, v.y, *actx, step_size, bc2_sqrt);
                        w.z = adam_value(w.z, g4[u].z, m.z, v.z, *actx, step_size, bc2_sqrt);
                        w.w = adam_value(w.w, g4[u].w, m.w, v.w, *actx, step_size, bc2_sqrt);
                        *(float4*)(actx->w + off[u]) = w; *(float4*)(actx->m + off[u]) = m; *(float4*)(actx->v + off[u]) = v;
                        const unsigned r = i / r4, q = i - r * r4;
                        T* d = tile + r * pitch + 4 * q;
                        Elem<T>::store(d, w.x); Elem<T>::store(d + 1, w.y); Elem<T>::store(d + 2, w.z); Elem<T>::store(d + 3, w.w);
                    }
                }
            }
        } else if (vec_ok) {
            // every load of the thread in flight before the first LDS store (a load per loop iteration paid a memory
            // round trip each: 18 of them per workgroup)
            const unsigned r4 = row / 4, tot = NCO * r4;
            constexpr unsigned LB = 9;                                     // loads in flight per thread and batch
            for (unsigned i0 = threadIdx.x; i0 < tot; i0 += LB * 256) {   // 2 batches (bf16) / 1 (f32) at 256 x 9
                float4 v[LB];
#pragma unroll
                for (unsigned u = 0; u < LB; ++u) {
                    const unsigned i = i0 + u * 256, ic = i < tot ? i : 0;
                    const unsigned r = ic / r4, q = ic - r * r4;
                    v[u] = *(const float4*)(j.src + ((size_t)(co0 + r) * Ci + ci0) * kk + 4 * q);
                }
#pragma unroll
                for (unsigned u = 0; u < LB; ++u) {
                    const unsigned i = i0 + u * 256;
                    if (i < tot) {
                        const unsigned r = i / r4, q = i - r * r4;
                        T* d = tile + r * pitch + 4 * q;
                        Elem<T>::store(d, v[u].x); Elem<T>::store(d + 1, v[u].y); Elem<T>::store(d + 2, v[u].z);
                        Elem<T>::store(d + 3, v[u].w);
                    }
                }
            }
        } else {
            for (unsigned i = threadIdx.x; i < NCO * row; i += 256) {
                const unsigned r = i / row, e = i - r * row;
                float v = 0.f;
                if (co0 + r < Co) {
                    const size_t so = ((size_t)(co0 + r) * Ci + ci0) * kk + e;
                    v = actx ? adam_at(*actx, (j.src - actx->w) + (long)so, step_size, bc2_sqrt) : j.src[so];
                }
                Elem<T>::store(tile + r * pitch + e, v);
            }
        }
        __syncthreads();
        if (vec_ok) {
            // ---- out 1: wf[co][t][ci0 + c .. + NV): 16-byte chunks, consecutive threads along ci
            const unsigned cch = cn / NV;
            for (unsigned i = threadIdx.x; i < NCO * kk * cch; i += 256) {
                const unsigned cc = i % cch, q = i / cch, tp = q % kk, r = q / kk;
                __attribute__((aligned(16))) T v[NV];
#pragma unroll
                for (unsigned u = 0; u < NV; ++u) v[u] = tile[r * pitch + (cc * NV + u) * kk + tp];
                *(u32x4*)(wf + ((size_t)(co0 + r) * kk + tp) * Ci + ci0 + cc * NV) = *(const u32x4*)v;
            }
            // ---- out 2: wd[ci][t][co0 .. co0 + NCO): one 16-byte chunk per (ci, t)
            for (unsigned i = threadIdx.x; i < row; i += 256) {           // i = c * kk + tp, the torch order of a row
                __attribute__((aligned(16))) T v[NCO];
#pragma unroll
                for (unsigned u = 0; u < NCO; ++u) v[u] = tile[u * pitch + i];
                const unsigned c = i / kk, tp = i - c * kk;
                *(u32x4*)(wd + ((size_t)(ci0 + c) * kk + tp) * Co + co0) = *(const u32x4*)v;
            }
        } else {
            for (unsigned i = threadIdx.x; i < NCO * row; i += 256) {
                const unsigned r = i / row, e = i - r * row, c = e / kk, tp = e - c * kk;
                if (co0 + r < Co) {
                    const T v = tile[r * pitch + e];
                    wf[((size_t)(co0 + r) * kk + tp) * Ci + ci0 + c] = v;
                    wd[((size_t)(ci0 + c) * kk + tp) * Co + co0 + r] = v;
                }
            }
        }
    }
}

// Type 4: the K-slice slabs of a conv / conv-transpose weight gradient, [ks][co][t][ci] f32 (rbvae_wgrad_gemm), summed
// in slab order into the torch layout [co][ci][t].  A workgroup owns one output channel (and a block of <= CRD_CI input
// channels): every slab's [t][ci] rows come in by 16-byte loads, all of a thread's loads in flight together, the sums
// change order through LDS and leave as 16-byte stores of the contiguous [ci][t] row.
constexpr int CRD_CI = 256, CRD_MAXKK = 16, CRD_ACC = CRD_CI * CRD_MAXKK / 4 / 256;      // float4 accumulators per thread
__device__ __forceinline__ void conv_reduce_rows(const Job& j, float* tile, unsigned bidx, unsigned nblk) {
#if defined(JOBS_ABL_ONE_SLAB) && !defined(RBVAE_ABLATION)
#error "JOBS_ABL_ONE_SLAB gives wrong sums (timing ablation): define RBVAE_ABLATION to confirm"
#endif
#ifdef JOBS_ABL_ONE_SLAB
    const unsigned Co = (unsigned)j.d0, Ci = (unsigned)j.d1, kk = (unsigned)j.d2, ns = 1;      // ablation partner of WG_ABL=5
#else
    const unsigned Co = (unsigned)j.d0, Ci = (unsigned)j.d1, kk = (unsigned)j.d2, ns = (unsigned)j.nslab;
#endif
    const unsigned ncb = (Ci + CRD_CI - 1) / CRD_CI;
    float* out = (float*)j.dst;
    for (unsigned b = bidx; b < Co * ncb; b += nblk) {
        const unsigned co = b / ncb, ci0 = (b % ncb) * CRD_CI;
        const unsigned cn = min((unsigned)CRD_CI, Ci - ci0);       // multiple of 4 (host-checked)
        const unsigned c4 = cn / 4, nch = kk * c4, pitch = cn + 1;
        float4 acc[CRD_ACC];
#pragma unroll
        for (int a = 0; a < CRD_ACC; ++a) acc[a] = make_float4(0.f, 0.f, 0.f, 0.f);
        const float* base = j.src + ((size_t)co * kk) * Ci + ci0;
        // this thread's chunk offsets inside a slab (the same for every slab)
        unsigned off[CRD_ACC];
        const int na = (int)((nch + 255) / 256);                     // uniform: accumulators in use
#pragma unroll
        for (int a = 0; a < CRD_ACC; ++a) {
            const unsigned i = threadIdx.x + a * 256;
            const unsigned ic = i < nch ? i : 0, tp = ic / c4, q = ic - tp * c4;
            off[a] = tp * Ci + 4 * q;
        }
        unsigned k = 0;
        if (na == 1) {
            // narrow weights (64 x 64 x 9: 144 chunks per slab and output channel): eight slabs' loads in flight -- with two,
            // the 128-256 slabs of a 2 M-pixel layer were 64-128 dependent round trips (35-40 us for 19 MB)
            for (; k + 8 <= ns; k += 8) {
                float4 v[8];
#pragma unroll
                for (int s8 = 0; s8 < 8; ++s8) v[s8] = *(const float4*)(base + (size_t)(k + s8) * j.slab + off[0]);
#pragma unroll
                for (int s8 = 0; s8 < 8; ++s8) { acc[0].x += v[s8].x; acc[0].y += v[s8].y; acc[0].z += v[s8].z; acc[0].w += v[s8].w; }
            }
        }
        for (; k + 2 <= ns; k += 2) {                              // two slabs' loads in flight; added in slab order
            float4 v[2][CRD_ACC];
#pragma unroll
            for (int s4 = 0; s4 < 2; ++s4)
#pragma unroll
                for (int a = 0; a < CRD_ACC; ++a)
                    if (a < na) v[s4][a] = *(const float4*)(base + (size_t)(k + s4) * j.slab + off[a]);
#pragma unroll
            for (int s4 = 0; s4 < 2; ++s4)
#pragma unroll
                for (int a = 0; a < CRD_ACC; ++a)
                    if (a < na) { acc[a].x += v[s4][a].x; acc[a].y += v[s4][a].y; acc[a].z += v[s4][a].z; acc[a].w += v[s4][a].w; }
        }
        for (; k < ns; ++k) {                                      // slab order: fixed, reproducible
            const float* p = base + (size_t)k * j.slab;
            float4 v[CRD_ACC];
#pragma unroll
            for (int a = 0; a < CRD_ACC; ++a)
                if (a < na) v[a] = *(const float4*)(p + off[a]);
#pragma unroll
            for (int a = 0; a < CRD_ACC; ++a)
                if (a < na) { acc[a].x += v[a].x; acc[a].y += v[a].y; acc[a].z += v[a].z; acc[a].w += v[a].w; }
        }
        __syncthreads();
#pragma unroll
        for (int a = 0; a < CRD_ACC; ++a) {
            const unsigned i = threadIdx.x + a * 256;
            if (i < nch) {
                const unsigned tp = i / c4, q = i - tp * c4;
                float* d = tile + tp * pitch + 4 * q;
                d[0] = acc[a].x * j.scale; d[1] = acc[a].y * j.scale; d[2] = acc[a].z * j.scale; d[3] = acc[a].w * j.scale;
            }
        }
        __syncthreads();
        // the output row [co][ci0 .. ci0 + cn)[kk] is cn * kk contiguous floats
        float* orow = out + ((size_t)co * Ci + ci0) * kk;
        const unsigned n4 = cn * kk / 4;
        for (unsigned i = threadIdx.x; i < n4; i += 256) {
            float4 o;
            float* op = (float*)&o;
#pragma unroll
            for (unsigned u = 0; u < 4; ++u) {
                const unsigned e = 4 * i + u, c = e / kk, tp = e - c * kk;
                op[u] = tile[tp * pitch + c];
            }
            if (j.accumulate) {
                const float4 w = *(const float4*)(orow + 4 * i);
                o.x += w.x; o.y += w.y; o.z += w.z; o.w += w.w;
            }
            *(float4*)(orow + 4 * i) = o;
        }
    }
}

// Launch shapes: grid (blocks per job, jobs) -- every job gets the same number of workgroups, most of which leave at once --
// or, with a block map (rbvae_run_jobs_sized), a 1-D grid of exactly the workgroups the jobs need: map[b] = (job, index of
// the workgroup within the job, workgroups of the job).  A step's update launch was 12 544 workgroups of which ~1 400 had work.
__global__ __launch_bounds__(256) void run_jobs_k(const Job* __restrict__ jobs, const int4* __restrict__ map) {
    __shared__ __attribute__((aligned(16))) unsigned char lds_raw[CPK_LDS_BYTES];
    static_assert(CPK_LDS_BYTES >= (int)sizeof(float) * CRD_MAXKK * (CRD_CI + 1), "the reduce tile fits the pack tile");
    static_assert(CPK_LDS_BYTES >= 256 * 16, "the wide row reduction's tree fits");
    unsigned bidx = blockIdx.x, nblk = gridDim.x, jidx = blockIdx.y;
    if (map) {
        const int4 m = map[blockIdx.x];
        jidx = (unsigned)m.x; bidx = (unsigned)m.y; nblk = (unsigned)m.z;
    }
    const Job j = jobs[jidx];
    if (j.type == 3) {
        if (j.dtype == RBVAE_F32) conv_pack_rows<float>(j, lds_raw, bidx, nblk); else conv_pack_rows<bf16_t>(j, lds_raw, bidx, nblk);
        return;
    }
    if (j.type == 4) {
        conv_reduce_rows(j, (float*)lds_raw, bidx, nblk);
        return;
    }
    if (j.type == 6 || j.type == 7) {
        // 6: optimiser update of a parameter tensor [d0][d1][d2] + its packed copies: dst (strides s0, s1, s2, type
        //    dtype & 255) and optionally dst2 (strides nslab, slab, accumulate; type dtype >> 8); 7: update only.
        // One thread per master element, consecutive threads on consecutive elements (w, g, m, v move coalesced).
        const AdamCtx c = *(const AdamCtx*)j.inner;
        const float step_size = c.hyper[0], bc2_sqrt = c.hyper[1];
        const unsigned d1 = (unsigned)j.d1, d2 = (unsigned)j.d2;
        const long n = j.d0 * j.d1 * j.d2, base = j.src - c.w;
        const int t1 = (int)(j.dtype & 255), t2 = (int)((j.dtype >> 8) & 255);
#ifndef JOBS_TILED_ADAM
#define JOBS_TILED_ADAM 1
#endif
        if (JOBS_TILED_ADAM && j.type == 6 && n >= (1l << 20)) {
            // A big tensor (>= 1 M elements = 128 tiles: the 131 072-element fc weight of the bench shape would keep 16 workgroups
            // busy for 40 us: 0.458 -> 0.49 ms per step) whose packed copies are PERMUTED (the fc weights: 56 320 x 32 in
            // NHWC-flatten order and transposed): with one thread per master element every 2-byte store of a transposed copy went to its own
            // 64-byte sector (30-40 us per fc weight).  Tiles of 8192 elements through LDS instead: the master side moves in
            // runs along i2, each copy in runs along ITS contiguous index; the update arithmetic is adam_at's, element for
            // element (bit-identical: only the order of the memory operations changes).
            // (scalars and selects, no indexed local arrays: those are promoted to LDS and cost every job of the launch a
            // workgroup per CU -- 40 064 instead of 36 992 bytes: +5 us per bench step)
            const long a0 = j.s0, a1 = j.s1, a2 = j.s2, c0 = j.nslab, c1 = j.slab, c2 = j.accumulate;
            const int fA = a0 == 1 ? 0 : (a1 == 1 ? 1 : 2);
            const int fB = !j.dst2 ? fA : (c0 == 1 ? 0 : (c1 == 1 ? 1 : 2));
            if (fA != 2 || fB != 2) {
                const unsigned D0 = (unsigned)j.d0;
                unsigned T0, T1, T2;
                if (fA != 2 && fB != 2 && fA != fB) { T2 = 16; T0 = fB == 0 ? 32 : 16; T1 = fB == 1 ? 32 : 16; }   // three contiguous indices
                else { const int f = fA != 2 ? fA : fB; T2 = 32; T0 = f == 0 ? 32 : 8; T1 = f == 1 ? 32 : 8; }
                T0 = min(T0, D0); T1 = min(T1, d1); T2 = min(T2, d2);
                const unsigned nt0 = (D0 + T0 - 1) / T0, nt1 = (d1 + T1 - 1) / T1, nt2 = (d2 + T2 - 1) / T2;
                const unsigned pitch = T2 + 1, tel = T0 * T1 * T2;
                float* tile = (float*)lds_raw;                                           // [T0 * T1][T2 + 1]
                for (unsigned t = bidx; t < nt0 * nt1 * nt2; t += nblk) {
                    const unsigned b2 = (t % nt2) * T2, b1 = ((t / nt2) % nt1) * T1, b0 = (t / (nt2 * nt1)) * T0;
                    // master order (i2 fastest), four elements per thread and round: their 16 loads are in flight together
                    // (one element per round was 32 dependent round trips per tile: the stores of one may alias the loads of
                    // the next as far as the compiler knows)
                    for (unsigned e0 = threadIdx.x; e0 < tel; e0 += 1024) {
                        long off[4]; unsigned li[4]; bool ok[4];
                        float pw[4], pg[4], pm[4], pv[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const unsigned e = e0 + 256 * u;
                            const unsigned l2 = e % T2, l1 = (e / T2) % T1, l0 = e / (T2 * T1);
                            ok[u] = e < tel && b0 + l0 < D0 && b1 + l1 < d1 && b2 + l2 < d2;
                            off[u] = base + ((long)(b0 + l0) * d1 + (b1 + l1)) * d2 + (b2 + l2);
                            li[u] = (l0 * T1 + l1) * pitch + l2;
                            if (ok[u]) { pw[u] = c.w[off[u]]; pg[u] = c.g[off[u]]; pm[u] = c.m[off[u]]; pv[u] = c.v[off[u]]; }
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            if (ok[u]) {
                                const float w = adam_value(pw[u], pg[u], pm[u], pv[u], c, step_size, bc2_sqrt);
                                c.m[off[u]] = pm[u]; c.v[off[u]] = pv[u]; c.w[off[u]] = w;
                                tile[li[u]] = w;
                            }
                        }
                    }
                    __syncthreads();
                    for (int copy = 0; copy < (j.dst2 ? 2 : 1); ++copy) {
                        const int f = copy ? fB : fA;                                    // this copy's contiguous index
                        const long q0 = copy ? c0 : a0, q1 = copy ? c1 : a1, q2 = copy ? c2 : a2;
                        void* dst = copy ? j.dst2 : j.dst;
                        const int ty = copy ? t2 : t1;
                        // e -> (fastest = index f, then the other two in order)
                        const unsigned Tf = f == 0 ? T0 : (f == 1 ? T1 : T2);
                        const unsigned Tm = f == 2 ? T1 : T2;                            // the faster of the other two
                        for (unsigned e = threadIdx.x; e < tel; e += 256) {
                            const unsigned lf = e % Tf, lm = (e / Tf) % Tm, ls = e / (Tf * Tm);
                            const unsigned l0 = f == 0 ? lf : ls, l1 = f == 1 ? lf : (f == 0 ? ls : lm), l2 = f == 2 ? lf : lm;
                            if (b0 + l0 < D0 && b1 + l1 < d1 && b2 + l2 < d2) {
                                const float w = tile[(l0 * T1 + l1) * pitch + l2];
                                const size_t o = (b0 + l0) * (size_t)q0 + (b1 + l1) * (size_t)q1 + (b2 + l2) * (size_t)q2;
                                if (ty == RBVAE_F32) ((float*)dst)[o] = w; else ((bf16_t*)dst)[o] = f32_to_bf16(w);
                            }
                        }
                    }
                    __syncthreads();
                }
                return;
            }
        }
        for (long i = (long)bidx * 256 + threadIdx.x; i < n; i += (long)nblk * 256) {
            const float w = adam_at(c, base + i, step_size, bc2_sqrt);
            if (j.type == 6) {
                const unsigned i2 = (unsigned)(i % d2), r = (unsigned)(i / d2), i1 = r % d1, i0 = r / d1;
                const size_t o1 = i0 * (size_t)j.s0 + i1 * (size_t)j.s1 + i2 * (size_t)j.s2;
                if (t1 == RBVAE_F32) ((float*)j.dst)[o1] = w; else ((bf16_t*)j.dst)[o1] = f32_to_bf16(w);
                if (j.dst2) {
                    const size_t o2 = i0 * (size_t)j.nslab + i1 * (size_t)j.slab + i2 * (size_t)j.accumulate;
                    if (t2 == RBVAE_F32) ((float*)j.dst2)[o2] = w; else ((bf16_t*)j.dst2)[o2] = f32_to_bf16(w);
                }
            }
        }
        return;
    }
    if (j.type == 5) {
        // batch gather from the HBM-resident latent table (rbvae_gather_frames as a job, so that it shares the launch of
        // the step's weight repack): dst[r] = src[plan[(counter % n_batches) * rows + r]], 16 bytes per lane
        const long rows = j.d0, nb = j.d1, vec = j.d2, table_rows = j.s2;
        const long* plan = (const long*)j.s0;
        const unsigned long long* counter = (const unsigned long long*)j.s1;
        const long b = counter ? (long)(counter[0] % (unsigned long long)nb) : 0;
        const float4* table = (const float4*)j.src;
        float4* out = (float4*)j.dst;
        for (long r = bidx; r < rows; r += nblk) {
            long src = plan[b * rows + r];
            if (src < 0 || src >= table_rows) src = 0;
            for (long i = threadIdx.x; i < vec; i += 256) out[r * vec + i] = table[src * vec + i];
        }
        return;
    }
    const unsigned d0 = (unsigned)j.d0, d1 = (unsigned)j.d1, d2 = (unsigned)j.d2;
    const unsigned n = d0 * d1 * d2;
    // this job may need fewer blocks than the launch provides
    if (bidx * (j.type == 2 ? 4u : 256u) >= n) return;
    if (j.type == 2) {
        const unsigned nsl = (unsigned)j.nslab, slb = (unsigned)j.slab;
        if (nsl >= 1024 && (n & 3) == 0 && (slb & 3) == 0 && ((size_t)j.src & 15) == 0) {
            // thousands of partial rows (the per-tile column sums of a 2 M-pixel layer): a workgroup owns FOUR columns, its
            // 256 threads walk the rows 16 bytes at a time (eight loads in flight each), then a fixed-order tree over the
            // threads -- one wave per column walked 16 384 rows with 64-way uncoalesced dword loads (50 us for 4 MB)
            float4* red = (float4*)lds_raw;
            float* out = (float*)j.dst;
            for (unsigned cb = bidx; cb < n / 4; cb += nblk) {
                float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
                const float* col = j.src + 4 * cb;
                unsigned k = threadIdx.x;
                for (; k + 7 * 256 < nsl; k += 8 * 256) {
                    float4 v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = *(const float4*)(col + (size_t)(k + u * 256) * slb);
#pragma unroll
                    for (int u = 0; u < 8; ++u) { a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w; }
                }
                for (; k < nsl; k += 256) {
                    const float4 v = *(const float4*)(col + (size_t)k * slb);
                    a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
                }
                red[threadIdx.x] = a;
                __syncthreads();
                for (unsigned s = 128; s > 0; s >>= 1) {
                    if (threadIdx.x < s) {
                        const float4 o = red[threadIdx.x + s];
                        float4 m = red[threadIdx.x];
                        m.x += o.x; m.y += o.y; m.z += o.z; m.w += o.w;
                        red[threadIdx.x] = m;
                    }
                    __syncthreads();
                }
                if (threadIdx.x < 4) {
                    const float* r = (const float*)red;
                    const float t = r[threadIdx.x] * j.scale;
                    out[4 * cb + threadIdx.x] = j.accumulate ? out[4 * cb + threadIdx.x] + t : t;
                }
                __syncthreads();
            }
            return;
        }
        // out[c] = scale * sum_k src[k*slab + c]; one wave per output, lanes stride the slabs, shuffle-reduce
        const int lane = threadIdx.x & 63;
        const unsigned wave = bidx * 4 + (threadIdx.x >> 6), nw = nblk * 4;
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
        for (unsigned i = bidx * 256 + threadIdx.x; i < d0 * d1; i += nblk * 256) {
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
    for (unsigned i = bidx * 256 + threadIdx.x; i < n; i += nblk * 256) {
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
    hipLaunchKernelGGL(run_jobs_k, dim3(blocks_per_job, njobs), dim3(256), 0, (hipStream_t)stream, (const Job*)jobs_dev,
                       (const int4*)nullptr);
    RBVAE_CHECK_LAUNCH("run_jobs");
    return RBVAE_OK;
}

// workgroups job `r` (16 x int64, host copy of a table row) can use: the loop bounds of run_jobs_k's branches
static long job_blocks_of(const long* r) {
    const long type = r[0], d0 = r[3], d1 = r[4], d2 = r[5], n = d0 * d1 * d2;
    const long dtype = r[11];
    if (type == 3) {
        const long es = (dtype & 255) == RBVAE_F32 ? 4 : 2, nco = 16 / es, nv = 16 / es, kk = d2;
        long cib = (CPK_MAXROW / kk) / nv * nv;
        if (cib > d1) cib = d1;
        if (cib > CPK_CIB && d1 % CPK_CIB == 0) cib = CPK_CIB;
        return ((d0 + nco - 1) / nco) * ((d1 + cib - 1) / cib);
    }
    if (type == 4) return d0 * ((d1 + CRD_CI - 1) / CRD_CI);
    if (type == 5) return d0;
    if (type == 6 || type == 7) {
        if (type == 6 && n >= (1l << 20)) {
            const long a0 = r[6], a1 = r[7], c0 = r[9], c1 = r[10];
            const int fA = a0 == 1 ? 0 : (a1 == 1 ? 1 : 2);
            const int fB = !r[15] ? fA : (c0 == 1 ? 0 : (c1 == 1 ? 1 : 2));
            if (fA != 2 || fB != 2) {
                long T0, T1, T2;
                if (fA != 2 && fB != 2 && fA != fB) { T2 = 16; T0 = fB == 0 ? 32 : 16; T1 = fB == 1 ? 32 : 16; }
                else { const int f = fA != 2 ? fA : fB; T2 = 32; T0 = f == 0 ? 32 : 8; T1 = f == 1 ? 32 : 8; }
                if (T0 > d0) T0 = d0;
                if (T1 > d1) T1 = d1;
                if (T2 > d2) T2 = d2;
                return ((d0 + T0 - 1) / T0) * ((d1 + T1 - 1) / T1) * ((d2 + T2 - 1) / T2);
            }
        }
        return (n + 255) / 256;
    }
    if (type == 2) return (n + 3) / 4;              // four columns (wide rows) or four waves = four outputs per workgroup
    if (r[14]) return (d0 * d1 + 255) / 256;        // a thread per (i0, i1) row
    return (n + 255) / 256;
}

extern "C" int rbvae_job_block_map(const long* rows_host, int njobs, int max_blocks_per_job, int* map_host, int map_capacity) {
    if (!rows_host || njobs <= 0 || max_blocks_per_job <= 0) return fail(RBVAE_E_INVALID, "job_block_map: bad arguments");
    long total = 0;
    for (int j = 0; j < njobs; ++j) {
        long nb = job_blocks_of(rows_host + 16l * j);
        if (nb < 1) nb = 1;
        if (nb > max_blocks_per_job) nb = max_blocks_per_job;
        if (map_host) {
            if (total + nb > map_capacity) return fail(RBVAE_E_INVALID, "job_block_map: map of %d entries is too small", map_capacity);
            for (long b = 0; b < nb; ++b) {
                int* m = map_host + 4 * (total + b);
                m[0] = j; m[1] = (int)b; m[2] = (int)nb; m[3] = 0;
            }
        }
        total += nb;
    }
    if (total >= (1l << 31)) return fail(RBVAE_E_INVALID, "job_block_map: too many workgroups");
    return (int)total;
}

extern "C" int rbvae_run_jobs_sized(const void* jobs_dev, const void* block_map_dev, int total_blocks, void* stream) {
    RBVAE_CHECK_ARG(jobs_dev && block_map_dev && total_blocks > 0, "run_jobs_sized: bad arguments");
    RBVAE_CHECK_ARG((uintptr_t)block_map_dev % 16 == 0, "run_jobs_sized: the block map must be 16-byte aligned");
    hipLaunchKernelGGL(run_jobs_k, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, (const Job*)jobs_dev,
                       (const int4*)block_map_dev);
    RBVAE_CHECK_LAUNCH("run_jobs_sized");
    return RBVAE_OK;
}
