// Kernels of the frozen LDM/Stable-Diffusion VAE encoder that are not GEMMs
// (src/stable-diffusion/ldm/modules/diffusionmodules/model.py): GroupNorm(32, eps 1e-6) fused with
// swish (:33-39), the softmax of the single-head mid-block attention (:186-192), a 2-D transpose for
// its value operand, and the posterior sample of AutoencoderKL.encode
// (ldm/modules/distributions/distributions.py:24-37; ldm/models/diffusion/ddpm.py:542-549).
// Activations are NHWC rows [N*H*W][C] in the storage type T.
#include "common.h"

namespace rbvae {

// ---- GroupNorm statistics: one workgroup per (image, group) -------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void gn_stats_k(const T* __restrict__ x, int HW, int C, int ld, int groups,
                                                  float eps, float* __restrict__ mean, float* __restrict__ rstd) {
    __shared__ float red[4];
    const int n = blockIdx.x / groups, g = blockIdx.x - n * groups;
    const int cg = C / groups;
    const T* base = x + (size_t)n * HW * ld + g * cg;
    // pass 1: mean
    float s = 0.f;
    for (int i = threadIdx.x; i < HW * cg; i += 256) {
        const int p = i / cg, c = i - p * cg;
        s += Elem<T>::load(base + (size_t)p * ld + c);
    }
    const float m = block_sum(s, red) / (float)(HW * cg);
    // pass 2: centred second moment (no cancellation)
    float q = 0.f;
    for (int i = threadIdx.x; i < HW * cg; i += 256) {
        const int p = i / cg, c = i - p * cg;
        const float d = Elem<T>::load(base + (size_t)p * ld + c) - m;
        q += d * d;
    }
    const float var = block_sum(q, red) / (float)(HW * cg);
    if (threadIdx.x == 0) {
        mean[blockIdx.x] = m;
        rstd[blockIdx.x] = rsqrtf(var + eps);
    }
}

// y = (x - mean) * rstd * gamma[c] + beta[c], optionally * sigmoid(.) (swish)
template <typename T>
__global__ void gn_apply_k(const T* __restrict__ x, T* __restrict__ y, const float* __restrict__ mean,
                           const float* __restrict__ rstd, const float* __restrict__ gamma,
                           const float* __restrict__ beta, long rows, int HW, int C, int ldx, int ldy, int groups,
                           int swish) {
    const int cg = C / groups;
    const long tot = rows * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const long r = i / C;
        const int n = (int)(r / HW);
        const int sg = n * groups + c / cg;
        float v = (Elem<T>::load(x + r * ldx + c) - mean[sg]) * rstd[sg] * gamma[c] + beta[c];
        if (swish) v = v * sigmoidf_(v);
        Elem<T>::store(y + r * ldy + c, v);
    }
}

// ---- GroupNorm, tiled: statistics from whole pixel rows --------------------------------------------------
// gn_stats_k above gives one workgroup a whole (image, group): N*32 workgroups that walk 8-32 byte pieces of
// every pixel row twice -- 69 % of the VAE encoder's time at 512x512 frames.  Here a workgroup takes a block of
// RB pixel rows of one image with 16-byte loads of whole rows, keeps them in registers, and leaves for every
// group the block's (mean, M2 = sum of squared deviations from that mean); gn_finish_k merges the blocks with
// the parallel-variance formula (as stable as two passes); gn_apply_vec_k normalises 16 bytes per thread.
template <typename T> struct GnVec;
template <> struct GnVec<bf16_t> { static constexpr int V = 8; };
template <> struct GnVec<float> { static constexpr int V = 4; };
constexpr int GN_PASSES = 8;

template <typename T>
__global__ __launch_bounds__(256) void gn_partial_k(const T* __restrict__ x, int HW, int C, int ld, int groups,
                                                    float2* __restrict__ part) {
    constexpr int V = GnVec<T>::V;
    __shared__ float chs[256 * V];            // [row lanes][C] per-channel sums (row lanes * C == 256 * V)
    __shared__ float gm[64];
    __shared__ float chc[256 * V];            // [C] per-channel totals (C <= 256 * V)
    const int tpr = C / V, rl_n = 256 / tpr;  // threads per row, row lanes
    const int rl = threadIdx.x / tpr, cv = threadIdx.x - rl * tpr;
    const int rb = rl_n * GN_PASSES;
    const int n = blockIdx.y, r0 = blockIdx.x * rb;
    const int cg = C / groups;
    const T* base = x + ((size_t)n * HW) * ld + cv * V;
    float v[GN_PASSES][V];
#pragma unroll
    for (int ps = 0; ps < GN_PASSES; ++ps) {
        const int r = r0 + ps * rl_n + rl;
        const bool ok = r < HW;
        const uint4 raw = *(const uint4*)(base + (size_t)(ok ? r : HW - 1) * ld);
        const T* e = (const T*)&raw;
#pragma unroll
        for (int k = 0; k < V; ++k) v[ps][k] = ok ? Elem<T>::load(e + k) : 0.f;
    }
    const int rows_b = min(rb, HW - r0);
    auto group_reduce = [&](const float (&cs)[V]) -> float {     // -> this thread's group total (threads < groups)
#pragma unroll
        for (int k = 0; k < V; ++k) chs[rl * C + cv * V + k] = cs[k];
        __syncthreads();
        // per-channel totals over the row lanes by C threads, then the group's cg channels: rl_n + cg dependent LDS
        // reads instead of rl_n * cg by the first `groups` threads alone
        for (int c = threadIdx.x; c < C; c += 256) {
            float a = 0.f;
            for (int q = 0; q < rl_n; ++q) a += chs[q * C + c];
            chc[c] = a;
        }
        __syncthreads();
        float t = 0.f;
        if ((int)threadIdx.x < groups) {
            for (int c = 0; c < cg; ++c) t += chc[threadIdx.x * cg + c];
        }
        __syncthreads();
        return t;
    };
    float cs[V];
#pragma unroll
    for (int k = 0; k < V; ++k) {
        cs[k] = 0.f;
#pragma unroll
        for (int ps = 0; ps < GN_PASSES; ++ps) cs[k] += v[ps][k];
    }
    const float gsum = group_reduce(cs);
    if ((int)threadIdx.x < groups) gm[threadIdx.x] = gsum / (float)(rows_b * cg);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < V; ++k) {
        const float m = gm[(cv * V + k) / cg];
        cs[k] = 0.f;
#pragma unroll
        for (int ps = 0; ps < GN_PASSES; ++ps) {
            const float d = v[ps][k] - m;
            const bool ok = r0 + ps * rl_n + rl < HW;
            cs[k] += ok ? d * d : 0.f;
        }
    }
    const float gm2 = group_reduce(cs);
    if ((int)threadIdx.x < groups)
        part[((size_t)n * gridDim.x + blockIdx.x) * groups + threadIdx.x] = make_float2(gm[threadIdx.x], gm2);
}

// one wave per (image, group): merge the blocks' (mean, M2)
__global__ __launch_bounds__(64) void gn_finish_k(const float2* __restrict__ part, int nb, int rb, int HW, int cg,
                                                  int groups, float eps, float* __restrict__ mean,
                                                  float* __restrict__ rstd) {
    const int n = blockIdx.x / groups, g = blockIdx.x - n * groups;
    const float2* p = part + (size_t)n * nb * groups + g;
    const float total = (float)HW * (float)cg;
    float a = 0.f;
    for (int b = threadIdx.x; b < nb; b += 64) a += (float)(min(rb, HW - b * rb) * cg) * p[(size_t)b * groups].x;
    const float m = wave_sum(a) / total;
    float q = 0.f;
    for (int b = threadIdx.x; b < nb; b += 64) {
        const float2 pb = p[(size_t)b * groups];
        const float d = pb.x - m;
        q += pb.y + (float)(min(rb, HW - b * rb) * cg) * d * d;
    }
    const float var = wave_sum(q) / total;
    if (threadIdx.x == 0) {
        mean[blockIdx.x] = m;
        rstd[blockIdx.x] = rsqrtf(var + eps);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void gn_apply_vec_k(const T* __restrict__ x, T* __restrict__ y,
                                                      const float* __restrict__ mean, const float* __restrict__ rstd,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      unsigned rows, int HW, int C, int ldx, int ldy, int groups,
                                                      int swish) {
    constexpr int V = GnVec<T>::V;
    const unsigned tpr = C / V;
    const int cg = C / groups;
    const unsigned tot = rows * tpr;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < tot; i += gridDim.x * 256u) {
        const unsigned r = i / tpr, cv = i - r * tpr;
        const unsigned n = r / (unsigned)HW;
        const int c0 = cv * V;
        uint4 raw = *(const uint4*)(x + (size_t)r * ldx + c0);
        T* e = (T*)&raw;
        float ga[V], be[V];
        if constexpr (V == 8) {
            const float4 g0 = *(const float4*)(gamma + c0), g1 = *(const float4*)(gamma + c0 + 4);
            const float4 b0 = *(const float4*)(beta + c0), b1 = *(const float4*)(beta + c0 + 4);
            ga[0] = g0.x; ga[1] = g0.y; ga[2] = g0.z; ga[3] = g0.w; ga[4] = g1.x; ga[5] = g1.y; ga[6] = g1.z; ga[7] = g1.w;
            be[0] = b0.x; be[1] = b0.y; be[2] = b0.z; be[3] = b0.w; be[4] = b1.x; be[5] = b1.y; be[6] = b1.z; be[7] = b1.w;
        } else {
            const float4 g0 = *(const float4*)(gamma + c0), b0 = *(const float4*)(beta + c0);
            ga[0] = g0.x; ga[1] = g0.y; ga[2] = g0.z; ga[3] = g0.w;
            be[0] = b0.x; be[1] = b0.y; be[2] = b0.z; be[3] = b0.w;
        }
        if ((cg & 3) == 0) {
            // a run of 4 channels (c0 is a multiple of V) lies in one group: V/4 statistics loads instead of 2 V, one
            // fma per element; bf16 storage takes the swish through the hardware exp2 / rcp (error far below its
            // rounding): with a division and two statistics loads per element this pass was ALU / latency bound
            float mu[V / 4], rs[V / 4];
#pragma unroll
            for (int h = 0; h < V / 4; ++h) {
                const int sg = n * groups + (c0 + 4 * h) / cg;
                mu[h] = mean[sg]; rs[h] = rstd[sg];
            }
#pragma unroll
            for (int k = 0; k < V; ++k) {
                const float sc = rs[k >> 2] * ga[k];
                float v = fmaf(Elem<T>::load(e + k), sc, be[k] - mu[k >> 2] * sc);
                if (swish) {
                    if constexpr (sizeof(T) == 2)
                        v = v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504f * v));
                    else
                        v = v * sigmoidf_(v);
                }
                Elem<T>::store(e + k, v);
            }
        } else {
#pragma unroll
            for (int k = 0; k < V; ++k) {
                const int sg = n * groups + (c0 + k) / cg;
                float v = (Elem<T>::load(e + k) - mean[sg]) * rstd[sg] * ga[k] + be[k];
                if (swish) v = v * sigmoidf_(v);
                Elem<T>::store(e + k, v);
            }
        }
        *(uint4*)(y + (size_t)r * ldy + c0) = raw;
    }
}

// ---- row softmax (in place capable): one wave per row -------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void softmax_rows_k(const T* __restrict__ x, T* __restrict__ y, long rows, int n,
                                                      int ld) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const T* xr = x + r * ld;
    float mx = -3.4e38f;
    for (int i = lane; i < n; i += 64) mx = fmaxf(mx, Elem<T>::load(xr + i));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float s = 0.f;
    for (int i = lane; i < n; i += 64) s += expf(Elem<T>::load(xr + i) - mx);
    s = wave_sum(s);
    const float inv = 1.0f / s;
    T* yr = y + r * ld;
    for (int i = lane; i < n; i += 64) Elem<T>::store(yr + i, expf(Elem<T>::load(xr + i) - mx) * inv);
}

// ---- out[c][r] = in[r][c]  (32x32 LDS tiles) -----------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void transpose_k(const T* __restrict__ in, T* __restrict__ out, int R, int C,
                                                   int ldi, int ldo) {
    __shared__ T tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    for (int j = ty; j < 32; j += 8)
        if (r0 + j < R && c0 + tx < C) tile[j][tx] = in[(size_t)(r0 + j) * ldi + c0 + tx];
    __syncthreads();
    for (int j = ty; j < 32; j += 8)
        if (c0 + j < C && r0 + tx < R) out[(size_t)(c0 + j) * ldo + r0 + tx] = tile[tx][j];
}

// ---- posterior sample: latent[n][c][h][w] = scale * (mean + exp(0.5*clamp(logvar,-30,20)) * eps) -----
template <typename T>
__global__ void posterior_sample_k(const T* __restrict__ moments, int ld, const float* __restrict__ eps,
                                   float* __restrict__ latent, int N, int Z, int HW, float scale) {
    const long tot = (long)N * Z * HW;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += (long)gridDim.x * blockDim.x) {
        const int p = (int)(i % HW);
        const long r = i / HW;
        const int c = (int)(r % Z), n = (int)(r / Z);
        const T* m = moments + ((size_t)n * HW + p) * ld;
        const float mean = Elem<T>::load(m + c);
        const float lv = fminf(fmaxf(Elem<T>::load(m + Z + c), -30.f), 20.f);
        latent[i] = scale * (mean + expf(0.5f * lv) * (eps ? eps[i] : 0.f));
    }
}

static inline int grid_n(long n, int cap = 8192) {
    long b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

}  // namespace rbvae

using namespace rbvae;

#define DISPATCH_T(dtype, CALL_F32, CALL_BF16, name)                       \
    if ((dtype) == RBVAE_F32) { CALL_F32; }                                \
    else if ((dtype) == RBVAE_BF16) { CALL_BF16; }                         \
    else return fail(RBVAE_E_INVALID, name ": dtype %d", (dtype));

extern "C" {


static bool gn_tiled_ok(int dtype, int C, int ldx, int ldy, int groups) {
    const int V = dtype == RBVAE_F32 ? 4 : 8;
    if (C % V || groups > 64 || C % groups) return false;
    const int tpr = C / V;
    return tpr <= 256 && 256 % tpr == 0 && ldx % V == 0 && ldy % V == 0;
}
static int gn_rows_per_block(int dtype, int C) { return (256 / (C / (dtype == RBVAE_F32 ? 4 : 8))) * GN_PASSES; }

size_t rbvae_groupnorm_ws_floats(int dtype, int N, int HW, int C, int groups) {
    size_t n = (size_t)2 * N * groups;
    if (gn_tiled_ok(dtype, C, C, C, groups)) n += (size_t)2 * N * cdiv(HW, gn_rows_per_block(dtype, C)) * groups + 4;
    return n;
}

int rbvae_groupnorm_swish(int dtype, const void* x, void* y, const float* gamma, const float* beta, float* stats_ws,
                          int N, int HW, int C, int ldx, int ldy, int groups, float eps, int swish, void* stream) {
    return rbvae_groupnorm_swish_ws(dtype, x, y, gamma, beta, stats_ws, (size_t)2 * N * groups, N, HW, C, ldx, ldy, groups,
                                    eps, swish, stream);
}

int rbvae_groupnorm_swish_ws(int dtype, const void* x, void* y, const float* gamma, const float* beta, float* stats_ws,
                             size_t ws_floats, int N, int HW, int C, int ldx, int ldy, int groups, float eps, int swish,
                             void* stream) {
    RBVAE_CHECK_ARG(x && y && gamma && beta && stats_ws && N > 0 && HW > 0 && C > 0, "groupnorm_swish: bad arguments");
    RBVAE_CHECK_ARG(groups > 0 && C % groups == 0 && ldx >= C && ldy >= C, "groupnorm_swish: C=%d groups=%d", C, groups);
    RBVAE_CHECK_ARG(ws_floats >= (size_t)2 * N * groups, "groupnorm_swish: workspace of %zu floats < %zu", ws_floats,
                    (size_t)2 * N * groups);
    hipStream_t st = (hipStream_t)stream;
    float* mean = stats_ws;
    float* rstd = stats_ws + (size_t)N * groups;
    const long rows = (long)N * HW;
    const bool tiled = gn_tiled_ok(dtype, C, ldx, ldy, groups) && ws_floats >= rbvae_groupnorm_ws_floats(dtype, N, HW, C, groups) &&
                       rows * (C / (dtype == RBVAE_F32 ? 4 : 8)) < (1l << 32) &&
                       ((uintptr_t)x | (uintptr_t)y | (uintptr_t)gamma | (uintptr_t)beta) % 16 == 0;
    if (tiled) {
        const int rb = gn_rows_per_block(dtype, C), nb = cdiv(HW, rb);
        float* pbase = stats_ws + (size_t)2 * N * groups;
        float2* part = (float2*)(pbase + (((uintptr_t)pbase % 8) ? 1 : 0));      // 8-byte aligned
        const dim3 pgrid(nb, N);
        DISPATCH_T(dtype,
                   hipLaunchKernelGGL(gn_partial_k<float>, pgrid, dim3(256), 0, st, (const float*)x, HW, C, ldx, groups, part);
                   hipLaunchKernelGGL(gn_finish_k, dim3(N * groups), dim3(64), 0, st, part, nb, rb, HW, C / groups, groups, eps,
                                      mean, rstd);
                   hipLaunchKernelGGL(gn_apply_vec_k<float>, dim3(grid_n(rows * (C / 4), 16384)), dim3(256), 0, st,
                                      (const float*)x, (float*)y, mean, rstd, gamma, beta, (unsigned)rows, HW, C, ldx, ldy,
                                      groups, swish),
                   hipLaunchKernelGGL(gn_partial_k<bf16_t>, pgrid, dim3(256), 0, st, (const bf16_t*)x, HW, C, ldx, groups, part);
                   hipLaunchKernelGGL(gn_finish_k, dim3(N * groups), dim3(64), 0, st, part, nb, rb, HW, C / groups, groups, eps,
                                      mean, rstd);
                   hipLaunchKernelGGL(gn_apply_vec_k<bf16_t>, dim3(grid_n(rows * (C / 8), 16384)), dim3(256), 0, st,
                                      (const bf16_t*)x, (bf16_t*)y, mean, rstd, gamma, beta, (unsigned)rows, HW, C, ldx, ldy,
                                      groups, swish),
                   "groupnorm_swish")
        RBVAE_CHECK_LAUNCH("groupnorm_swish");
        return RBVAE_OK;
    }
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(gn_stats_k<float>, dim3(N * groups), dim3(256), 0, st, (const float*)x, HW, C, ldx,
                                  groups, eps, mean, rstd);
               hipLaunchKernelGGL(gn_apply_k<float>, dim3(grid_n(rows * C)), dim3(256), 0, st, (const float*)x,
                                  (float*)y, mean, rstd, gamma, beta, rows, HW, C, ldx, ldy, groups, swish),
               hipLaunchKernelGGL(gn_stats_k<bf16_t>, dim3(N * groups), dim3(256), 0, st, (const bf16_t*)x, HW, C, ldx,
                                  groups, eps, mean, rstd);
               hipLaunchKernelGGL(gn_apply_k<bf16_t>, dim3(grid_n(rows * C)), dim3(256), 0, st, (const bf16_t*)x,
                                  (bf16_t*)y, mean, rstd, gamma, beta, rows, HW, C, ldx, ldy, groups, swish),
               "groupnorm_swish")
    RBVAE_CHECK_LAUNCH("groupnorm_swish");
    return RBVAE_OK;
}

/* statistics only: mean = stats_ws[0 : N*groups], rstd = stats_ws[N*groups : 2*N*groups] (for rbvae_gn_affine) */
int rbvae_groupnorm_stats(int dtype, const void* x, float* stats_ws, size_t ws_floats, int N, int HW, int C, int ldx,
                          int groups, float eps, void* stream) {
    RBVAE_CHECK_ARG(x && stats_ws && N > 0 && HW > 0 && C > 0, "groupnorm_stats: bad arguments");
    RBVAE_CHECK_ARG(groups > 0 && C % groups == 0 && ldx >= C, "groupnorm_stats: C=%d groups=%d", C, groups);
    RBVAE_CHECK_ARG(ws_floats >= (size_t)2 * N * groups, "groupnorm_stats: workspace of %zu floats < %zu", ws_floats,
                    (size_t)2 * N * groups);
    hipStream_t st = (hipStream_t)stream;
    float* mean = stats_ws;
    float* rstd = stats_ws + (size_t)N * groups;
    const bool tiled = gn_tiled_ok(dtype, C, ldx, ldx, groups) && ws_floats >= rbvae_groupnorm_ws_floats(dtype, N, HW, C, groups) &&
                       (uintptr_t)x % 16 == 0;
    if (tiled) {
        const int rb = gn_rows_per_block(dtype, C), nb = cdiv(HW, rb);
        float* pbase = stats_ws + (size_t)2 * N * groups;
        float2* part = (float2*)(pbase + (((uintptr_t)pbase % 8) ? 1 : 0));
        const dim3 pgrid(nb, N);
        DISPATCH_T(dtype,
                   hipLaunchKernelGGL(gn_partial_k<float>, pgrid, dim3(256), 0, st, (const float*)x, HW, C, ldx, groups, part),
                   hipLaunchKernelGGL(gn_partial_k<bf16_t>, pgrid, dim3(256), 0, st, (const bf16_t*)x, HW, C, ldx, groups, part),
                   "groupnorm_stats")
        hipLaunchKernelGGL(gn_finish_k, dim3(N * groups), dim3(64), 0, st, part, nb, rb, HW, C / groups, groups, eps, mean, rstd);
    } else {
        DISPATCH_T(dtype,
                   hipLaunchKernelGGL(gn_stats_k<float>, dim3(N * groups), dim3(256), 0, st, (const float*)x, HW, C, ldx,
                                      groups, eps, mean, rstd),
                   hipLaunchKernelGGL(gn_stats_k<bf16_t>, dim3(N * groups), dim3(256), 0, st, (const bf16_t*)x, HW, C, ldx,
                                      groups, eps, mean, rstd),
                   "groupnorm_stats")
    }
    RBVAE_CHECK_LAUNCH("groupnorm_stats");
    return RBVAE_OK;
}

/* the apply pass alone, from given statistics: y = swish?((x - mean) * rstd * gamma + beta) */
int rbvae_groupnorm_apply(int dtype, const void* x, void* y, const float* mean, const float* rstd, const float* gamma,
                          const float* beta, int N, int HW, int C, int ldx, int ldy, int groups, int swish, void* stream) {
    RBVAE_CHECK_ARG(x && y && mean && rstd && gamma && beta && N > 0 && HW > 0 && C > 0, "groupnorm_apply: bad arguments");
    RBVAE_CHECK_ARG(groups > 0 && C % groups == 0 && ldx >= C && ldy >= C, "groupnorm_apply: C=%d groups=%d", C, groups);
    hipStream_t st = (hipStream_t)stream;
    const long rows = (long)N * HW;
    const bool vec = gn_tiled_ok(dtype, C, ldx, ldy, groups) && rows * (C / (dtype == RBVAE_F32 ? 4 : 8)) < (1l << 32) &&
                     ((uintptr_t)x | (uintptr_t)y | (uintptr_t)gamma | (uintptr_t)beta) % 16 == 0;
    if (vec) {
        DISPATCH_T(dtype,
                   hipLaunchKernelGGL(gn_apply_vec_k<float>, dim3(grid_n(rows * (C / 4), 16384)), dim3(256), 0, st,
                                      (const float*)x, (float*)y, mean, rstd, gamma, beta, (unsigned)rows, HW, C, ldx, ldy,
                                      groups, swish),
                   hipLaunchKernelGGL(gn_apply_vec_k<bf16_t>, dim3(grid_n(rows * (C / 8), 16384)), dim3(256), 0, st,
                                      (const bf16_t*)x, (bf16_t*)y, mean, rstd, gamma, beta, (unsigned)rows, HW, C, ldx, ldy,
                                      groups, swish),
                   "groupnorm_apply")
    } else {
        DISPATCH_T(dtype,
                   hipLaunchKernelGGL(gn_apply_k<float>, dim3(grid_n(rows * C)), dim3(256), 0, st, (const float*)x,
                                      (float*)y, mean, rstd, gamma, beta, rows, HW, C, ldx, ldy, groups, swish),
                   hipLaunchKernelGGL(gn_apply_k<bf16_t>, dim3(grid_n(rows * C)), dim3(256), 0, st, (const bf16_t*)x,
                                      (bf16_t*)y, mean, rstd, gamma, beta, rows, HW, C, ldx, ldy, groups, swish),
                   "groupnorm_apply")
    }
    RBVAE_CHECK_LAUNCH("groupnorm_apply");
    return RBVAE_OK;
}

int rbvae_softmax_rows(int dtype, const void* x, void* y, long rows, int n, int ld, void* stream) {
    RBVAE_CHECK_ARG(x && y && rows > 0 && n > 0 && ld >= n, "softmax_rows: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const int blocks = cdiv(rows, 4);
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(softmax_rows_k<float>, dim3(blocks), dim3(256), 0, st, (const float*)x, (float*)y,
                                  rows, n, ld),
               hipLaunchKernelGGL(softmax_rows_k<bf16_t>, dim3(blocks), dim3(256), 0, st, (const bf16_t*)x,
                                  (bf16_t*)y, rows, n, ld),
               "softmax_rows")
    RBVAE_CHECK_LAUNCH("softmax_rows");
    return RBVAE_OK;
}

int rbvae_transpose2d(int dtype, const void* in, void* out, int R, int C, int ldi, int ldo, void* stream) {
    RBVAE_CHECK_ARG(in && out && R > 0 && C > 0 && ldi >= C && ldo >= R, "transpose2d: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(cdiv(C, 32), cdiv(R, 32));
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(transpose_k<float>, grid, dim3(256), 0, st, (const float*)in, (float*)out, R, C, ldi,
                                  ldo),
               hipLaunchKernelGGL(transpose_k<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)in, (bf16_t*)out, R, C,
                                  ldi, ldo),
               "transpose2d")
    RBVAE_CHECK_LAUNCH("transpose2d");
    return RBVAE_OK;
}

int rbvae_posterior_sample(int dtype, const void* moments, int ld, const float* eps, float* latent, int N, int Z,
                           int HW, float scale, void* stream) {
    RBVAE_CHECK_ARG(moments && latent && N > 0 && Z > 0 && HW > 0 && ld >= 2 * Z, "posterior_sample: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const long tot = (long)N * Z * HW;
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(posterior_sample_k<float>, dim3(grid_n(tot)), dim3(256), 0, st,
                                  (const float*)moments, ld, eps, latent, N, Z, HW, scale),
               hipLaunchKernelGGL(posterior_sample_k<bf16_t>, dim3(grid_n(tot)), dim3(256), 0, st,
                                  (const bf16_t*)moments, ld, eps, latent, N, Z, HW, scale),
               "posterior_sample")
    RBVAE_CHECK_LAUNCH("posterior_sample");
    return RBVAE_OK;
}

}  // extern "C"
