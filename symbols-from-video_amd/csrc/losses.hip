// Binarise + KL, pairwise-distance losses, MSE: the small f32 reductions of the
// RBVAE step.  All sums run in a fixed order (bitwise reproducible run to run).
#include "common.h"

namespace rbvae {

thread_local char g_err[512] = {0};
char* err_buf() { return g_err; }
int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

constexpr int RED_THREADS = 1024;

// kl_elem: common.h
// kl_elem_grad: common.h

__global__ __launch_bounds__(RED_THREADS) void binarize_kl_fwd_k(
    const float* __restrict__ h, const float* __restrict__ U, float* __restrict__ y_soft,
    float* __restrict__ z, float* __restrict__ kl_mean, int rows, int L, float tau, float ratio,
    float neps, int hard, float lp, float l1p, float keps, int clamp, unsigned long long seed,
    const unsigned long long* __restrict__ seed_dev) {
    __shared__ float red[RED_THREADS / 64];
    const int n = rows * L;
    float acc = 0.f;
    if (seed_dev) seed += seed_dev[0] * 0x9E3779B97F4A7C15ull;
    // batches of UB elements per thread: every global load of a batch is in flight before the first use (the
    // one-element-per-iteration loop paid a memory round trip per element: 8 of them at 256 x 32)
    constexpr int UB = 8;
    for (int base = threadIdx.x; base < n; base += UB * RED_THREADS) {
        float hv[UB], uv[UB];
#pragma unroll
        for (int b = 0; b < UB; ++b) {
            const int i = base + b * RED_THREADS, ic = i < n ? i : n - 1;
            hv[b] = h[ic];
            // U == null: 24-bit uniform in [0,1) from the counter hash (device-side noise)
            uv[b] = U ? U[ic] : (float)(hash_u32(seed, (unsigned long long)ic) >> 8) * (1.0f / 16777216.0f);
        }
#pragma unroll
        for (int b = 0; b < UB; ++b) {
            const int i = base + b * RED_THREADS;
            if (i >= n) break;
            const float u = uv[b];
            const float noise = ratio * (logf(u + neps) - logf(1.0f - u + neps));
            const float y = sigmoidf_((hv[b] + noise) / tau);
            const float zz = hard ? (y > 0.5f ? 1.0f : 0.0f) : y;
            y_soft[i] = y;
            z[i] = zz;
            if (kl_mean) acc += kl_elem(zz, lp, l1p, keps, clamp);
        }
    }
    if (kl_mean) {
        const float tot = block_sum(acc, red);
        if (threadIdx.x == 0) kl_mean[0] = tot / (float)rows;
    }
}

// many-workgroup form: one element per thread, per-block partial KL sums (fixed order inside the block)
__global__ __launch_bounds__(256) void binarize_kl_fwd_parts_k(
    const float* __restrict__ h, const float* __restrict__ U, float* __restrict__ y_soft,
    float* __restrict__ z, float* __restrict__ kl_parts, int n, float tau, const float* __restrict__ tau_dev,
    float ratio, float neps, int hard, float lp, float l1p, float keps, int clamp, unsigned long long seed,
    const unsigned long long* __restrict__ seed_dev) {
    __shared__ float red[4];
    if (tau_dev) tau = tau_dev[0];
    const int i = blockIdx.x * 256 + threadIdx.x;
    float acc = 0.f;
    if (i < n) {
        if (seed_dev) seed += seed_dev[0] * 0x9E3779B97F4A7C15ull;
        const float u = U ? U[i] : (float)(hash_u32(seed, (unsigned long long)i) >> 8) * (1.0f / 16777216.0f);
        const float noise = ratio * (logf(u + neps) - logf(1.0f - u + neps));
        const float y = sigmoidf_((h[i] + noise) / tau);
        const float zz = hard ? (y > 0.5f ? 1.0f : 0.0f) : y;
        y_soft[i] = y;
        z[i] = zz;
        if (kl_parts) acc = kl_elem(zz, lp, l1p, keps, clamp);
    }
    if (kl_parts) {
        const float tot = block_sum(acc, red);
        if (threadIdx.x == 0) kl_parts[blockIdx.x] = tot;
    }
}

__global__ void binarize_kl_bwd_k(const float* __restrict__ g_z, const float* __restrict__ y_soft,
                                  const float* __restrict__ z, float* __restrict__ dh, int accumulate,
                                  int n, int rows, float tau, const float* __restrict__ tau_dev, float klw,
                                  const float* __restrict__ gs, float lp, float l1p, float keps, int clamp) {
    __builtin_amdgcn_s_setprio(3);               // sits between the two LSTM backward launches of the chain
    if (tau_dev) tau = tau_dev[0];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float w = klw / (float)rows;
    if (gs) w *= gs[0];
    float g = g_z ? g_z[i] : 0.f;
    if (w != 0.f) g += w * kl_elem_grad(z[i], lp, l1p, keps, clamp);
    const float y = y_soft[i];
    const float v = g * y * (1.0f - y) / tau;
    dh[i] = accumulate ? dh[i] + v : v;
}

__global__ __launch_bounds__(RED_THREADS) void kl_fwd_k(const float* __restrict__ q, float* out, int rows,
                                                        int L, float lp, float l1p, float eps, int clamp) {
    __shared__ float red[RED_THREADS / 64];
    float acc = 0.f;
    for (int i = threadIdx.x; i < rows * L; i += RED_THREADS) acc += kl_elem(q[i], lp, l1p, eps, clamp);
    const float tot = block_sum(acc, red);
    if (threadIdx.x == 0) out[0] = tot / (float)rows;
}

__global__ void kl_bwd_k(const float* __restrict__ q, float* __restrict__ dq, int n, int rows, float lp,
                         float l1p, float eps, int clamp, float scale, const float* __restrict__ gs) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float w = scale / (float)rows;
    if (gs) w *= gs[0];
    dq[i] = w * kl_elem_grad(q[i], lp, l1p, eps, clamp);
}

// One wave per row: sum_L (a - b + eps)^2 with a lane-strided loop + shuffle reduce.
__device__ __forceinline__ float row_sqdist(const float* a, const float* b, int L, float eps, int lane) {
    float s = 0.f;
    for (int k = lane; k < L; k += 64) {
        const float d = a[k] - b[k] + eps;
        s += d * d;
    }
    return wave_sum(s);
}

__global__ __launch_bounds__(RED_THREADS) void pairdist_fwd_k(
    const float* __restrict__ x1, const float* __restrict__ x2, long s1, long s2, int rows, int L,
    int label, float margin, float eps, float* out) {
    __shared__ float red[RED_THREADS / 64];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    float acc = 0.f;
    for (int r = wid; r < rows; r += RED_THREADS / 64) {
        const float d = sqrtf(row_sqdist(x1 + r * s1, x2 + r * s2, L, eps, lane));
        const float m = fmaxf(margin - d, 0.f);
        acc += label ? m * m : d * d;
    }
    acc = (lane == 0) ? acc : 0.f;
    const float tot = block_sum(acc, red);
    if (threadIdx.x == 0) out[0] = tot / (float)rows;
}

// dx1 = w * dloss/dx1 for one row; dx2 = -dx1.
__global__ void pairdist_bwd_k(const float* __restrict__ x1, const float* __restrict__ x2, long s1, long s2,
                               int rows, int L, int label, float margin, float eps, float scale,
                               const float* __restrict__ gs, float* dx1, float* dx2, long ds1, long ds2,
                               int accumulate) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= rows) return;
    const float* a = x1 + r * s1;
    const float* b = x2 + r * s2;
    float w = scale / (float)rows;
    if (gs) w *= gs[0];
    float coef;
    if (label == 0) {
        coef = 2.f * w;                        // d(d^2)/da = 2 (a - b + eps)
    } else {
        const float d = sqrtf(row_sqdist(a, b, L, eps, lane));
        const float m = fmaxf(margin - d, 0.f);
        coef = (m > 0.f && d > 0.f) ? -2.f * w * m / d : 0.f;
    }
    for (int k = lane; k < L; k += 64) {
        const float g = coef * (a[k] - b[k] + eps);
        if (dx1) { float* p = dx1 + r * ds1 + k; *p = accumulate ? *p + g : g; }
        if (dx2) { float* p = dx2 + r * ds2 + k; *p = accumulate ? *p - g : -g; }
    }
}

// ---- contrast_loss, 'cosine' branch (percep_RBVAE_train.py:94-96): dist = 1 - cosine_similarity(x1, x2) ----------
// torch.nn.functional.cosine_similarity: x1.x2 / (max(|x1|, eps) * max(|x2|, eps)), eps 1e-8.  One wave per row.
__device__ __forceinline__ void row_cos(const float* a, const float* b, int L, int lane, float& dot, float& na, float& nb) {
    float d = 0.f, sa = 0.f, sb = 0.f;
    for (int k = lane; k < L; k += 64) {
        d += a[k] * b[k];
        sa += a[k] * a[k];
        sb += b[k] * b[k];
    }
    dot = wave_sum(d);
    na = sqrtf(wave_sum(sa));
    nb = sqrtf(wave_sum(sb));
}

__global__ __launch_bounds__(RED_THREADS) void paircos_fwd_k(const float* __restrict__ x1, const float* __restrict__ x2,
                                                             long s1, long s2, int rows, int L, int label, float margin,
                                                             float eps, float* out) {
    __shared__ float red[RED_THREADS / 64];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    float acc = 0.f;
    for (int r = wid; r < rows; r += RED_THREADS / 64) {
        float dot, na, nb;
        row_cos(x1 + r * s1, x2 + r * s2, L, lane, dot, na, nb);
        const float d = 1.f - dot / (fmaxf(na, eps) * fmaxf(nb, eps));
        const float m = fmaxf(margin - d, 0.f);
        acc += label ? m * m : d * d;
    }
    acc = (lane == 0) ? acc : 0.f;
    const float tot = block_sum(acc, red);
    if (threadIdx.x == 0) out[0] = tot / (float)rows;
}

__global__ void paircos_bwd_k(const float* __restrict__ x1, const float* __restrict__ x2, long s1, long s2, int rows, int L,
                              int label, float margin, float eps, float scale, const float* __restrict__ gs, float* dx1,
                              float* dx2, long ds1, long ds2) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= rows) return;
    const float* a = x1 + r * s1;
    const float* b = x2 + r * s2;
    float dot, na, nb;
    row_cos(a, b, L, lane, dot, na, nb);
    const float ca = fmaxf(na, eps), cb = fmaxf(nb, eps);
    const float cs = dot / (ca * cb), d = 1.f - cs;
    float w = scale / (float)rows;
    if (gs) w *= gs[0];
    // dloss/dcos: label 0: d(d^2) = -2 d;  label 1: d(max(m - d, 0)^2) = +2 max(m - d, 0)
    const float g = label ? 2.f * w * fmaxf(margin - d, 0.f) : -2.f * w * d;
    // dcos/da = b / (ca cb) - [na > eps] cos a / na^2   (the clamped norm is a constant below eps)
    const float ia = na > eps ? cs / (na * na) : 0.f, ib = nb > eps ? cs / (nb * nb) : 0.f;
    const float inv = 1.f / (ca * cb);
    for (int k = lane; k < L; k += 64) {
        if (dx1) dx1[r * ds1 + k] = g * (b[k] * inv - ia * a[k]);
        if (dx2) dx2[r * ds2 + k] = g * (a[k] * inv - ib * b[k]);
    }
}

// ---- trainer's contrastive term, one launch --------------------------------
__global__ __launch_bounds__(RED_THREADS) void contrast_term_fwd_k(const float* __restrict__ h0,
                                                                   const float* __restrict__ h1, int B, int T,
                                                                   int L, float* out) {
    __shared__ float red[RED_THREADS / 64];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const float eps = 1e-6f;
    float sim = 0.f, dis = 0.f;
    for (int r = wid; r < B * T; r += RED_THREADS / 64) {
        const float d = sqrtf(row_sqdist(h0 + (long)r * L, h1 + (long)r * L, L, eps, lane));
        sim += d * d;
        const int t = r % T;
        if (t < T - 1) {
            const float d2 = sqrtf(row_sqdist(h0 + (long)r * L, h0 + (long)(r + 1) * L, L, eps, lane));
            const float m = fmaxf(1.0f - d2, 0.f);
            dis += m * m;
        }
    }
    sim = (lane == 0) ? sim : 0.f;
    dis = (lane == 0) ? dis : 0.f;
    const float ts = block_sum(sim, red);
    const float td = block_sum(dis, red);
    if (threadIdx.x == 0) out[0] = ts / (float)(B * T) + td / ((float)B * (float)(T - 1));
}

// One wave per (b,t) row of h0/h1.  Row t of h0 takes: the similar term, the
// dissimilar pair (t,t+1) as first operand and the pair (t-1,t) as second.
__global__ void contrast_term_bwd_k(const float* __restrict__ h0, const float* __restrict__ h1, int B, int T,
                                    int L, float scale, const float* __restrict__ gs, float* __restrict__ dh0,
                                    float* __restrict__ dh1) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= B * T) return;
    const float eps = 1e-6f;
    float w = scale;
    if (gs) w *= gs[0];
    const float wsim = 2.f * w / (float)(B * T);
    const float wdis = w / ((float)B * (float)(T - 1));
    const int t = r % T;
    const float* a = h0 + (long)r * L;
    const float* b = h1 + (long)r * L;
    float cn = 0.f, cp = 0.f;   // coefficients of the (t,t+1) and (t-1,t) pairs
    if (t < T - 1) {
        const float d = sqrtf(row_sqdist(a, a + L, L, eps, lane));
        const float m = fmaxf(1.0f - d, 0.f);
        cn = (m > 0.f && d > 0.f) ? -2.f * wdis * m / d : 0.f;
    }
    if (t > 0) {
        const float d = sqrtf(row_sqdist(a - L, a, L, eps, lane));
        const float m = fmaxf(1.0f - d, 0.f);
        cp = (m > 0.f && d > 0.f) ? -2.f * wdis * m / d : 0.f;
    }
    for (int k = lane; k < L; k += 64) {
        const float gsim = wsim * (a[k] - b[k] + eps);
        float g0 = gsim;
        if (t < T - 1) g0 += cn * (a[k] - a[k + L] + eps);
        if (t > 0) g0 -= cp * (a[k - L] - a[k] + eps);
        dh0[(long)r * L + k] = g0;
        dh1[(long)r * L + k] = -gsim;
    }
}

// Forward value (as per-block partial sums) and gradient of the contrastive term in ONE many-workgroup launch:
// one wave per (b, t) row.  parts[2*block] = sum of d(h0,h1)^2 over the block's rows, parts[2*block+1] = sum of
// max(1 - d(h0[t], h0[t+1]), 0)^2; rbvae_combine_losses finishes  sum0/(B*T) + sum1/(B*(T-1)).  The two-launch
// form above runs its forward in a single workgroup (10-12 us); this one is ~4 us and needs no side stream.
__global__ __launch_bounds__(256) void contrast_term_fused_k(const float* __restrict__ h0, const float* __restrict__ h1,
                                                             int B, int T, int L, float scale,
                                                             const float* __restrict__ gs, float* __restrict__ parts,
                                                             float* __restrict__ dh0, float* __restrict__ dh1) {
    __shared__ float red[4][2];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int r = blockIdx.x * 4 + wid;
    const float eps = 1e-6f;
    float sim = 0.f, dis = 0.f;
    if (r < B * T) {
        float w = scale;
        if (gs) w *= gs[0];
        const float wsim = 2.f * w / (float)(B * T);
        const float wdis = w / ((float)B * (float)(T - 1));
        const int t = r % T;
        const float* a = h0 + (long)r * L;
        const float* b = h1 + (long)r * L;
        const float d = sqrtf(row_sqdist(a, b, L, eps, lane));
        sim = d * d;
        float cn = 0.f, cp = 0.f;
        if (t < T - 1) {
            const float d2 = sqrtf(row_sqdist(a, a + L, L, eps, lane));
            const float m = fmaxf(1.0f - d2, 0.f);
            dis = m * m;
            cn = (m > 0.f && d2 > 0.f) ? -2.f * wdis * m / d2 : 0.f;
        }
        if (t > 0) {
            const float d2 = sqrtf(row_sqdist(a - L, a, L, eps, lane));
            const float m = fmaxf(1.0f - d2, 0.f);
            cp = (m > 0.f && d2 > 0.f) ? -2.f * wdis * m / d2 : 0.f;
        }
        for (int k = lane; k < L; k += 64) {
            const float gsim = wsim * (a[k] - b[k] + eps);
            float g0 = gsim;
            if (t < T - 1) g0 += cn * (a[k] - a[k + L] + eps);
            if (t > 0) g0 -= cp * (a[k - L] - a[k] + eps);
            dh0[(long)r * L + k] = g0;
            dh1[(long)r * L + k] = -gsim;
        }
    }
    if (lane == 0) { red[wid][0] = sim; red[wid][1] = dis; }
    __syncthreads();
    if (threadIdx.x < 2)
        parts[2 * blockIdx.x + threadIdx.x] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}

// ---- triplet ---------------------------------------------------------------
__global__ __launch_bounds__(RED_THREADS) void triplet_fwd_k(
    const float* __restrict__ a, const float* __restrict__ p, const float* __restrict__ n, long sa, long sp,
    long sn, int rows, int L, float margin, float eps, int swap, float* out) {
    __shared__ float red[RED_THREADS / 64];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    float acc = 0.f;
    for (int r = wid; r < rows; r += RED_THREADS / 64) {
        const float dap = sqrtf(row_sqdist(a + r * sa, p + r * sp, L, eps, lane));
        float dan = sqrtf(row_sqdist(a + r * sa, n + r * sn, L, eps, lane));
        if (swap) dan = fminf(dan, sqrtf(row_sqdist(p + r * sp, n + r * sn, L, eps, lane)));
        acc += fmaxf(margin + dap - dan, 0.f);
    }
    acc = (lane == 0) ? acc : 0.f;
    const float tot = block_sum(acc, red);
    if (threadIdx.x == 0) out[0] = tot / (float)rows;
}

__device__ __forceinline__ void acc_store(float* p, float g, int accumulate) { *p = accumulate ? *p + g : g; }

// grads of mean_r max(margin + d(a,p) - min(d(a,n), d(p,n)), 0) for one row per wave
__device__ __forceinline__ void triplet_row_bwd(const float* a, const float* p, const float* n, int L,
                                                float margin, float eps, int swap, float w, int lane,
                                                float* da, float* dp, float* dn, int accumulate) {
    const float dap = sqrtf(row_sqdist(a, p, L, eps, lane));
    const float dan = sqrtf(row_sqdist(a, n, L, eps, lane));
    float dpn = 0.f;
    bool use_pn = false;
    if (swap) {
        dpn = sqrtf(row_sqdist(p, n, L, eps, lane));
        use_pn = dpn < dan;            // torch.minimum sends the gradient to the smaller one
    }
    const float dneg = use_pn ? dpn : dan;
    const bool active = (margin + dap - dneg) > 0.f;
    const float cap = (active && dap > 0.f) ? w / dap : 0.f;
    const float cng = (active && dneg > 0.f) ? -w / dneg : 0.f;
    for (int k = lane; k < L; k += 64) {
        const float gap = cap * (a[k] - p[k] + eps);           // d dap: +a, -p
        float ga = gap, gp = -gap, gn = 0.f;
        if (use_pn) { const float g = cng * (p[k] - n[k] + eps); gp += g; gn -= g; }
        else        { const float g = cng * (a[k] - n[k] + eps); ga += g; gn -= g; }
        if (da) acc_store(da + k, ga, accumulate);
        if (dp) acc_store(dp + k, gp, accumulate);
        if (dn) acc_store(dn + k, gn, accumulate);
    }
}

__global__ void triplet_bwd_k(const float* __restrict__ a, const float* __restrict__ p,
                              const float* __restrict__ n, long sa, long sp, long sn, int rows, int L,
                              float margin, float eps, int swap, float scale, const float* __restrict__ gs,
                              float* da, float* dp, float* dn, long dsa, long dsp, long dsn, int accumulate) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= rows) return;
    float w = scale / (float)rows;
    if (gs) w *= gs[0];
    triplet_row_bwd(a + r * sa, p + r * sp, n + r * sn, L, margin, eps, swap, w, lane,
                    da ? da + r * dsa : nullptr, dp ? dp + r * dsp : nullptr, dn ? dn + r * dsn : nullptr,
                    accumulate);
}

__global__ __launch_bounds__(RED_THREADS) void triplet_term_fwd_k(const float* __restrict__ h0,
                                                                  const float* __restrict__ h1, int B, int T,
                                                                  int L, float margin, float* out) {
    __shared__ float red[RED_THREADS / 64];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const float eps = 1e-8f;
    float acc = 0.f;
    for (int r = wid; r < B * T; r += RED_THREADS / 64) {
        if (r % T == T - 1) continue;
        const float* a = h0 + (long)r * L;
        const float* p = h1 + (long)r * L;
        const float* n = a + L;
        const float dap = sqrtf(row_sqdist(a, p, L, eps, lane));
        const float dan = fminf(sqrtf(row_sqdist(a, n, L, eps, lane)), sqrtf(row_sqdist(p, n, L, eps, lane)));
        acc += fmaxf(margin + dap - dan, 0.f);
    }
    acc = (lane == 0) ? acc : 0.f;
    const float tot = block_sum(acc, red);
    if (threadIdx.x == 0) out[0] = tot / ((float)B * (float)(T - 1));
}

// One block per sequence b; the T-1 triplets of a sequence touch overlapping rows,
// so a single wave walks them in order (T <= 17: negligible work).
__global__ void triplet_term_bwd_k(const float* __restrict__ h0, const float* __restrict__ h1, int B, int T,
                                   int L, float margin, float scale, const float* __restrict__ gs,
                                   float* __restrict__ dh0, float* __restrict__ dh1) {
    const int lane = threadIdx.x;
    const int b = blockIdx.x;
    float w = scale / ((float)B * (float)(T - 1));
    if (gs) w *= gs[0];
    for (int t = 0; t < T; ++t)
        for (int k = lane; k < L; k += 64) {
            dh0[((long)b * T + t) * L + k] = 0.f;
            dh1[((long)b * T + t) * L + k] = 0.f;
        }
    for (int s = 0; s < T - 1; ++s) {
        const long o = ((long)b * T + s) * L;
        triplet_row_bwd(h0 + o, h1 + o, h0 + o + L, L, margin, 1e-8f, 1, w, lane, dh0 + o, dh1 + o,
                        dh0 + o + L, 1);
    }
}

// ---- MSE -------------------------------------------------------------------
constexpr int MSE_BLOCKS = 512;
__global__ __launch_bounds__(256) void mse_partial_k(const float* __restrict__ a, const float* __restrict__ b,
                                                     long n, float* __restrict__ ws) {
    __shared__ float red[4];
    float acc = 0.f;
    const long n4 = n >> 2;
    const float4* a4 = reinterpret_cast<const float4*>(a);
    const float4* b4 = reinterpret_cast<const float4*>(b);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 x = a4[i], y = b4[i];
        const float d0 = x.x - y.x, d1 = x.y - y.y, d2 = x.z - y.z, d3 = x.w - y.w;
        acc += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
    }
    if (blockIdx.x == 0)
        for (long i = (n4 << 2) + threadIdx.x; i < n; i += 256) { const float d = a[i] - b[i]; acc += d * d; }
    const float tot = block_sum(acc, red);
    if (threadIdx.x == 0) ws[blockIdx.x] = tot;
}
__global__ __launch_bounds__(MSE_BLOCKS) void mse_final_k(const float* __restrict__ ws, int nb, long n,
                                                          float* out) {
    __shared__ float red[MSE_BLOCKS / 64];
    const float v = (int)threadIdx.x < nb ? ws[threadIdx.x] : 0.f;
    const float tot = block_sum(v, red);
    if (threadIdx.x == 0) out[0] = tot / (float)n;
}
__global__ void mse_bwd_k(const float* __restrict__ a, const float* __restrict__ b, long n, float scale,
                          const float* __restrict__ gs, float* __restrict__ da) {
    float w = 2.f * scale / (float)n;
    if (gs) w *= gs[0];
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        da[i] = w * (a[i] - b[i]);
}

}  // namespace rbvae

using namespace rbvae;

extern "C" {

int rbvae_version(void) { return 101; }

__global__ void counter_add_k(unsigned long long* c, unsigned long long inc) { c[0] += inc; }
int rbvae_counter_add(unsigned long long* counter, unsigned long long inc, void* stream) {
    RBVAE_CHECK_ARG(counter, "counter_add: null pointer");
    hipLaunchKernelGGL(counter_add_k, dim3(1), dim3(1), 0, (hipStream_t)stream, counter, inc);
    RBVAE_CHECK_LAUNCH("counter_add");
    return RBVAE_OK;
}
const char* rbvae_last_error(void) { return err_buf(); }

int rbvae_binarize_kl_fwd(const float* h, const float* U, float* y_soft, float* z, float* kl_mean, int rows,
                          int L, float tau, float noise_ratio, float noise_eps, int hard, float kl_p,
                          float kl_eps, int kl_clamp, unsigned long long seed, const unsigned long long* seed_dev,
                          void* stream) {
    RBVAE_CHECK_ARG(h && y_soft && z, "binarize_kl_fwd: null pointer");
    RBVAE_CHECK_ARG(rows > 0 && L > 0 && tau > 0.f, "binarize_kl_fwd: rows=%d L=%d tau=%g", rows, L, tau);
    RBVAE_CHECK_ARG(kl_p > 0.f && kl_p < 1.f, "binarize_kl_fwd: p=%g outside (0,1)", kl_p);
    hipLaunchKernelGGL(binarize_kl_fwd_k, dim3(1), dim3(RED_THREADS), 0, (hipStream_t)stream, h, U, y_soft, z,
                       kl_mean, rows, L, tau, noise_ratio, noise_eps, hard, logf(kl_p), logf(1.0f - kl_p),
                       kl_eps, kl_clamp, seed, seed_dev);
    RBVAE_CHECK_LAUNCH("binarize_kl_fwd");
    return RBVAE_OK;
}

int rbvae_binarize_kl_nparts(int rows, int L) { return cdiv((long)rows * L, 256); }

int rbvae_binarize_kl_fwd_parts(const float* h, const float* U, float* y_soft, float* z, float* kl_parts, int rows,
                                int L, float tau, const float* tau_dev, float noise_ratio, float noise_eps, int hard,
                                float kl_p, float kl_eps, int kl_clamp, unsigned long long seed,
                                const unsigned long long* seed_dev, void* stream) {
    RBVAE_CHECK_ARG(h && y_soft && z, "binarize_kl_fwd_parts: null pointer");
    RBVAE_CHECK_ARG(rows > 0 && L > 0 && (tau_dev || tau > 0.f), "binarize_kl_fwd_parts: rows=%d L=%d tau=%g", rows, L, tau);
    RBVAE_CHECK_ARG(!kl_parts || (kl_p > 0.f && kl_p < 1.f), "binarize_kl_fwd_parts: kl_p=%g outside (0,1)", kl_p);
    const int n = rows * L;
    hipLaunchKernelGGL(binarize_kl_fwd_parts_k, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, h, U, y_soft, z,
                       kl_parts, n, tau, tau_dev, noise_ratio, noise_eps, hard, logf(kl_p), logf(1.0f - kl_p), kl_eps,
                       kl_clamp, seed, seed_dev);
    RBVAE_CHECK_LAUNCH("binarize_kl_fwd_parts");
    return RBVAE_OK;
}

int rbvae_binarize_kl_bwd(const float* g_z, const float* y_soft, const float* z, float* dh, int accumulate,
                          int rows, int L, float tau, const float* tau_dev, float kl_weight, const float* gscale_dev,
                          float kl_p, float kl_eps, int kl_clamp, void* stream) {
    RBVAE_CHECK_ARG(y_soft && z && dh, "binarize_kl_bwd: null pointer");
    RBVAE_CHECK_ARG(rows > 0 && L > 0 && (tau_dev || tau > 0.f), "binarize_kl_bwd: rows=%d L=%d tau=%g", rows, L, tau);
    const int n = rows * L;
    hipLaunchKernelGGL(binarize_kl_bwd_k, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, g_z, y_soft, z,
                       dh, accumulate, n, rows, tau, tau_dev, kl_weight, gscale_dev, logf(kl_p), logf(1.0f - kl_p),
                       kl_eps, kl_clamp);
    RBVAE_CHECK_LAUNCH("binarize_kl_bwd");
    return RBVAE_OK;
}

int rbvae_kl_fwd(const float* q, float* out, int rows, int L, float p, float eps, int clamp, void* stream) {
    RBVAE_CHECK_ARG(q && out && rows > 0 && L > 0, "kl_fwd: bad arguments");
    RBVAE_CHECK_ARG(p > 0.f && p < 1.f, "kl_fwd: p=%g outside (0,1)", p);
    hipLaunchKernelGGL(kl_fwd_k, dim3(1), dim3(RED_THREADS), 0, (hipStream_t)stream, q, out, rows, L, logf(p),
                       logf(1.0f - p), eps, clamp);
    RBVAE_CHECK_LAUNCH("kl_fwd");
    return RBVAE_OK;
}

int rbvae_kl_bwd(const float* q, float* dq, int rows, int L, float p, float eps, int clamp, float scale,
                 const float* gscale_dev, void* stream) {
    RBVAE_CHECK_ARG(q && dq && rows > 0 && L > 0, "kl_bwd: bad arguments");
    const int n = rows * L;
    hipLaunchKernelGGL(kl_bwd_k, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, q, dq, n, rows, logf(p),
                       logf(1.0f - p), eps, clamp, scale, gscale_dev);
    RBVAE_CHECK_LAUNCH("kl_bwd");
    return RBVAE_OK;
}

int rbvae_pairdist_fwd(const float* x1, const float* x2, long s1, long s2, int rows, int L, int label,
                       float margin, float eps, float* out, void* stream) {
    RBVAE_CHECK_ARG(x1 && x2 && out && rows > 0 && L > 0, "pairdist_fwd: bad arguments");
    hipLaunchKernelGGL(pairdist_fwd_k, dim3(1), dim3(RED_THREADS), 0, (hipStream_t)stream, x1, x2, s1, s2, rows,
                       L, label, margin, eps, out);
    RBVAE_CHECK_LAUNCH("pairdist_fwd");
    return RBVAE_OK;
}

int rbvae_pairdist_bwd(const float* x1, const float* x2, long s1, long s2, int rows, int L, int label,
                       float margin, float eps, float scale, const float* gscale_dev, float* dx1, float* dx2,
                       long ds1, long ds2, int accumulate, void* stream) {
    RBVAE_CHECK_ARG(x1 && x2 && rows > 0 && L > 0, "pairdist_bwd: bad arguments");
    hipLaunchKernelGGL(pairdist_bwd_k, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x1, x2, s1, s2,
                       rows, L, label, margin, eps, scale, gscale_dev, dx1, dx2, ds1, ds2, accumulate);
    RBVAE_CHECK_LAUNCH("pairdist_bwd");
    return RBVAE_OK;
}

int rbvae_paircos_fwd(const float* x1, const float* x2, long s1, long s2, int rows, int L, int label, float margin,
                      float eps, float* out, void* stream) {
    RBVAE_CHECK_ARG(x1 && x2 && out && rows > 0 && L > 0, "paircos_fwd: bad arguments");
    hipLaunchKernelGGL(paircos_fwd_k, dim3(1), dim3(RED_THREADS), 0, (hipStream_t)stream, x1, x2, s1, s2, rows, L, label,
                       margin, eps, out);
    RBVAE_CHECK_LAUNCH("paircos_fwd");
    return RBVAE_OK;
}

int rbvae_paircos_bwd(const float* x1, const float* x2, long s1, long s2, int rows, int L, int label, float margin,
                      float eps, float scale, const float* gscale_dev, float* dx1, float* dx2, long ds1, long ds2,
                      void* stream) {
    RBVAE_CHECK_ARG(x1 && x2 && rows > 0 && L > 0, "paircos_bwd: bad arguments");
    hipLaunchKernelGGL(paircos_bwd_k, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x1, x2, s1, s2, rows, L,
                       label, margin, eps, scale, gscale_dev, dx1, dx2, ds1, ds2);
    RBVAE_CHECK_LAUNCH("paircos_bwd");
    return RBVAE_OK;
}

int rbvae_contrast_term_fwd(const float* h0, const float* h1, int B, int T, int L, float* out, void* stream) {
    RBVAE_CHECK_ARG(h0 && h1 && out && B > 0 && L > 0, "contrast_term_fwd: bad arguments");
    RBVAE_CHECK_ARG(T >= 2, "contrast_term_fwd: needs T >= 2 states (got %d)", T);
    hipLaunchKernelGGL(contrast_term_fwd_k, dim3(1), dim3(RED_THREADS), 0, (hipStream_t)stream, h0, h1, B, T, L,
                       out);
    RBVAE_CHECK_LAUNCH("contrast_term_fwd");
    return RBVAE_OK;
}

int rbvae_contrast_term_bwd(const float* h0, const float* h1, int B, int T, int L, float scale,
                            const float* gscale_dev, float* dh0, float* dh1, void* stream) {
    RBVAE_CHECK_ARG(h0 && h1 && dh0 && dh1 && B > 0 && L > 0, "contrast_term_bwd: bad arguments");
    RBVAE_CHECK_ARG(T >= 2, "contrast_term_bwd: needs T >= 2 states (got %d)", T);
    hipLaunchKernelGGL(contrast_term_bwd_k, dim3(cdiv(B * T, 4)), dim3(256), 0, (hipStream_t)stream, h0, h1, B,
                       T, L, scale, gscale_dev, dh0, dh1);
    RBVAE_CHECK_LAUNCH("contrast_term_bwd");
    return RBVAE_OK;
}

int rbvae_contrast_term_nparts(int B, int T) { return cdiv((long)B * T, 4); }

int rbvae_contrast_term_fused(const float* h0, const float* h1, int B, int T, int L, float scale,
                              const float* gscale_dev, float* parts, float* dh0, float* dh1, void* stream) {
    RBVAE_CHECK_ARG(h0 && h1 && parts && dh0 && dh1 && B > 0 && L > 0, "contrast_term_fused: bad arguments");
    RBVAE_CHECK_ARG(T >= 2, "contrast_term_fused: needs T >= 2 states (got %d)", T);
    hipLaunchKernelGGL(contrast_term_fused_k, dim3(cdiv(B * T, 4)), dim3(256), 0, (hipStream_t)stream, h0, h1, B, T, L,
                       scale, gscale_dev, parts, dh0, dh1);
    RBVAE_CHECK_LAUNCH("contrast_term_fused");
    return RBVAE_OK;
}

int rbvae_triplet_fwd(const float* a, const float* p, const float* n, long sa, long sp, long sn, int rows, int L,
                      float margin, float eps, int swap, float* out, void* stream) {
    RBVAE_CHECK_ARG(a && p && n && out && rows > 0 && L > 0, "triplet_fwd: bad arguments");
    hipLaunchKernelGGL(triplet_fwd_k, dim3(1), dim3(RED_THREADS), 0, (hipStream_t)stream, a, p, n, sa, sp, sn,
                       rows, L, margin, eps, swap, out);
    RBVAE_CHECK_LAUNCH("triplet_fwd");
    return RBVAE_OK;
}

int rbvae_triplet_bwd(const float* a, const float* p, const float* n, long sa, long sp, long sn, int rows, int L,
                      float margin, float eps, int swap, float scale, const float* gscale_dev, float* da,
                      float* dp, float* dn, long dsa, long dsp, long dsn, int accumulate, void* stream) {
    RBVAE_CHECK_ARG(a && p && n && rows > 0 && L > 0, "triplet_bwd: bad arguments");
    hipLaunchKernelGGL(triplet_bwd_k, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, a, p, n, sa, sp,
                       sn, rows, L, margin, eps, swap, scale, gscale_dev, da, dp, dn, dsa, dsp, dsn, accumulate);
    RBVAE_CHECK_LAUNCH("triplet_bwd");
    return RBVAE_OK;
}

int rbvae_triplet_term_fwd(const float* h0, const float* h1, int B, int T, int L, float margin, float* out,
                           void* stream) {
    RBVAE_CHECK_ARG(h0 && h1 && out && B > 0 && L > 0, "triplet_term_fwd: bad arguments");
    RBVAE_CHECK_ARG(T >= 2, "triplet_term_fwd: needs T >= 2 states (got %d)", T);
    hipLaunchKernelGGL(triplet_term_fwd_k, dim3(1), dim3(RED_THREADS), 0, (hipStream_t)stream, h0, h1, B, T, L,
                       margin, out);
    RBVAE_CHECK_LAUNCH("triplet_term_fwd");
    return RBVAE_OK;
}

int rbvae_triplet_term_bwd(const float* h0, const float* h1, int B, int T, int L, float margin, float scale,
                           const float* gscale_dev, float* dh0, float* dh1, void* stream) {
    RBVAE_CHECK_ARG(h0 && h1 && dh0 && dh1 && B > 0 && L > 0, "triplet_term_bwd: bad arguments");
    RBVAE_CHECK_ARG(T >= 2, "triplet_term_bwd: needs T >= 2 states (got %d)", T);
    hipLaunchKernelGGL(triplet_term_bwd_k, dim3(B), dim3(64), 0, (hipStream_t)stream, h0, h1, B, T, L, margin,
                       scale, gscale_dev, dh0, dh1);
    RBVAE_CHECK_LAUNCH("triplet_term_bwd");
    return RBVAE_OK;
}

size_t rbvae_mse_ws_floats(long n) { (void)n; return MSE_BLOCKS; }

int rbvae_mse_fwd(const float* a, const float* b, long n, float* out, float* ws, void* stream) {
    RBVAE_CHECK_ARG(a && b && out && ws && n > 0, "mse_fwd: bad arguments");
    RBVAE_CHECK_ARG(((uintptr_t)a % 16 == 0) && ((uintptr_t)b % 16 == 0), "mse_fwd: inputs must be 16-byte aligned");
    int nb = cdiv(n >> 2, 256 * 4);
    nb = nb < 1 ? 1 : (nb > MSE_BLOCKS ? MSE_BLOCKS : nb);
    hipLaunchKernelGGL(mse_partial_k, dim3(nb), dim3(256), 0, (hipStream_t)stream, a, b, n, ws);
    hipLaunchKernelGGL(mse_final_k, dim3(1), dim3(MSE_BLOCKS), 0, (hipStream_t)stream, ws, nb, n, out);
    RBVAE_CHECK_LAUNCH("mse_fwd");
    return RBVAE_OK;
}

int rbvae_mse_bwd(const float* a, const float* b, long n, float scale, const float* gscale_dev, float* da,
                  void* stream) {
    RBVAE_CHECK_ARG(a && b && da && n > 0, "mse_bwd: bad arguments");
    int nb = cdiv(n, 256 * 4);
    nb = nb > 2048 ? 2048 : nb;
    hipLaunchKernelGGL(mse_bwd_k, dim3(nb), dim3(256), 0, (hipStream_t)stream, a, b, n, scale, gscale_dev, da);
    RBVAE_CHECK_LAUNCH("mse_bwd");
    return RBVAE_OK;
}

}  // extern "C"
