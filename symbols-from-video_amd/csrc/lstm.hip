// Stacked LSTM (nn.LSTM(L, L, layers, batch_first), zero initial state) forward and
// BPTT for the encoder/decoder RNNs (percep_RBVAE_model.py:94-122, called at :155,:163).
//
// Sequences are independent, so one workgroup owns one sequence and walks
// layer -> time with the layer's weight rows held in registers (4L threads, one gate
// row each: 2L weights per thread).  The whole stack is one launch.  Weight
// gradients are a separate batched reduction over the saved gate gradients
// (fixed summation order: bitwise reproducible).
//
// Weight block layout (= the reference's registration order, per layer):
//   w_ih [4L][L], w_hh [4L][L], b_ih [4L], b_hh [4L]      -> 8*L*L + 8*L floats per layer
#include "common.h"

namespace rbvae {

__device__ __forceinline__ long lstm_layer_floats(int L) { return 8l * L * L + 8l * L; }

// hs_all : [layers+1][S][T][L]  slot 0 = stack input, slot l+1 = output of layer l
// hprev  : [layers][S][T][L]    h_{t-1} of layer l (zeros at t = 0)      (training only)
// acts   : [layers][S][T][4L]   post-activation gates i, f, g, o          (training only)
// cs     : [layers][S][T][L]    cell state                                (training only)
template <int LMAX>
__global__ __launch_bounds__(512) void lstm_fwd_k(const float* __restrict__ wblk, float* __restrict__ hs_all,
                                                  float* __restrict__ hprev, float* __restrict__ acts,
                                                  float* __restrict__ cs, int S, int T, int L, int layers) {
    extern __shared__ float sm[];
    float* xin = sm;                 // [T][L]
    float* hout = xin + T * L;       // [T][L]
    float* hcur = hout + T * L;      // [L]
    float* gates = hcur + L;         // [4L]
    const int j = threadIdx.x;
    const int s = blockIdx.x;
    const bool row = j < 4 * L;
    for (int i = j; i < T * L; i += blockDim.x) xin[i] = hs_all[((long)s * T) * L + i];
    for (int l = 0; l < layers; ++l) {
        const float* wl = wblk + l * lstm_layer_floats(L);
        float wih[LMAX > 0 ? LMAX : 1], whh[LMAX > 0 ? LMAX : 1];
        float bsum = 0.f;
        if (row) {
            bsum = wl[8l * L * L + j] + wl[8l * L * L + 4 * L + j];
            if constexpr (LMAX > 0) {
                // clamped index: every load is unconditional (no branch per element)
                const float* pi = wl + j * L;
                const float* ph = pi + 4 * L * L;
#pragma unroll
                for (int k = 0; k < LMAX; ++k) {
                    const int kk = k < L ? k : L - 1;
                    const float a = pi[kk], b = ph[kk];
                    wih[k] = k < L ? a : 0.f;
                    whh[k] = k < L ? b : 0.f;
                }
            }
        }
        if (j < L) hcur[j] = 0.f;
        float c = 0.f;
        __syncthreads();
        for (int t = 0; t < T; ++t) {
            if (row) {
                float a = bsum;
                const float* xt = xin + t * L;
                if constexpr (LMAX > 0) {
#pragma unroll
                    for (int k = 0; k < LMAX; ++k)
                        if (k < L) a = fmaf(wih[k], xt[k], a);
#pragma unroll
                    for (int k = 0; k < LMAX; ++k)
                        if (k < L) a = fmaf(whh[k], hcur[k], a);
                } else {
                    const float* wi = wl + (long)j * L;
                    const float* wh = wl + 4l * L * L + (long)j * L;
                    for (int k = 0; k < L; ++k) a = fmaf(wi[k], xt[k], a);
                    for (int k = 0; k < L; ++k) a = fmaf(wh[k], hcur[k], a);
                }
                gates[j] = a;
            }
            __syncthreads();
            if (j < L) {
                const float ig = sigmoidf_(gates[j]), fg = sigmoidf_(gates[L + j]);
                const float gg = tanhf(gates[2 * L + j]), og = sigmoidf_(gates[3 * L + j]);
                const float hp = hcur[j];
                c = fg * c + ig * gg;
                const float h = og * tanhf(c);
                const long o = (((long)l * S + s) * T + t);
                if (acts) {
                    float* ap = acts + o * 4 * L;
                    ap[j] = ig; ap[L + j] = fg; ap[2 * L + j] = gg; ap[3 * L + j] = og;
                    cs[o * L + j] = c;
                    hprev[o * L + j] = hp;
                }
                hout[t * L + j] = h;
                hcur[j] = h;          // only this thread touched hcur[j] in this phase
                hs_all[(((long)(l + 1) * S + s) * T + t) * L + j] = h;
            }
            __syncthreads();
        }
        float* tmp = xin; xin = hout; hout = tmp;
    }
}

// g_top : [S][T][L]  gradient wrt the top layer's outputs
// dG    : [layers][S][T][4L]  gradient wrt the pre-activation gates (for the weight grads)
// dx    : [S][T][L]  gradient wrt the stack input
template <int LMAX>
__global__ __launch_bounds__(512) void lstm_bwd_k(const float* __restrict__ wblk, const float* __restrict__ acts,
                                                  const float* __restrict__ cs, const float* __restrict__ g_top,
                                                  float* __restrict__ dG, float* __restrict__ dx, int S, int T, int L,
                                                  int layers) {
    extern __shared__ float sm[];
    float* dhout = sm;               // [T][L] grad wrt this layer's outputs
    float* dxin = dhout + T * L;     // [T][L] grad wrt this layer's inputs
    float* dg = dxin + T * L;        // [4L]
    float* dhrec = dg + 4 * L;       // [L]
    float* part = dhrec + L;         // [4][2][L] partial column sums
    const int j = threadIdx.x;
    const int s = blockIdx.x;
    const bool row = j < 4 * L;
    const int kcol = j % L, prt = j / L;     // column role: column kcol, gate rows prt*L .. prt*L+L-1
    for (int i = j; i < T * L; i += blockDim.x) dhout[i] = g_top[((long)s * T) * L + i];
    for (int l = layers - 1; l >= 0; --l) {
        const float* wl = wblk + l * lstm_layer_floats(L);
        float wic[LMAX > 0 ? LMAX : 1], whc[LMAX > 0 ? LMAX : 1];
        if constexpr (LMAX > 0) {
            if (row) {
                const float* pi = wl + prt * L * L + kcol;
                const float* ph = pi + 4 * L * L;
#pragma unroll
                for (int jj = 0; jj < LMAX; ++jj) {
                    const int o = (jj < L ? jj : L - 1) * L;
                    const float a = pi[o], b = ph[o];
                    wic[jj] = jj < L ? a : 0.f;
                    whc[jj] = jj < L ? b : 0.f;
                }
            }
        }
        if (j < L) dhrec[j] = 0.f;
        float dc_next = 0.f;
        __syncthreads();
        for (int t = T - 1; t >= 0; --t) {
            const long o = (((long)l * S + s) * T + t);
            if (j < L) {
                const float* ap = acts + o * 4 * L;
                const float ig = ap[j], fg = ap[L + j], gg = ap[2 * L + j], og = ap[3 * L + j];
                const float c = cs[o * L + j];
                const float cprev = t > 0 ? cs[(o - 1) * L + j] : 0.f;
                const float tc = tanhf(c);
                const float dh = dhout[t * L + j] + dhrec[j];
                const float dc = dc_next + dh * og * (1.f - tc * tc);
                const float d_o = dh * tc * og * (1.f - og);
                const float d_i = dc * gg * ig * (1.f - ig);
                const float d_f = dc * cprev * fg * (1.f - fg);
                const float d_g = dc * ig * (1.f - gg * gg);
                dc_next = dc * fg;
                dg[j] = d_i; dg[L + j] = d_f; dg[2 * L + j] = d_g; dg[3 * L + j] = d_o;
                float* gp = dG + o * 4 * L;
                gp[j] = d_i; gp[L + j] = d_f; gp[2 * L + j] = d_g; gp[3 * L + j] = d_o;
            }
            __syncthreads();
            if (row) {
                float ax = 0.f, ah = 0.f;
                const float* dgp = dg + prt * L;
                if constexpr (LMAX > 0) {
#pragma unroll
                    for (int jj = 0; jj < LMAX; ++jj)
                        if (jj < L) { ax = fmaf(wic[jj], dgp[jj], ax); ah = fmaf(whc[jj], dgp[jj], ah); }
                } else {
                    for (int jj = 0; jj < L; ++jj) {
                        const float d = dgp[jj];
                        ax = fmaf(wl[(long)(prt * L + jj) * L + kcol], d, ax);
                        ah = fmaf(wl[4l * L * L + (long)(prt * L + jj) * L + kcol], d, ah);
                    }
                }
                part[(prt * 2 + 0) * L + kcol] = ax;
                part[(prt * 2 + 1) * L + kcol] = ah;
            }
            __syncthreads();
            if (j < L) {
                dxin[t * L + j] = part[0 * L + j] + part[2 * L + j] + part[4 * L + j] + part[6 * L + j];
                dhrec[j] = part[1 * L + j] + part[3 * L + j] + part[5 * L + j] + part[7 * L + j];
            }
            __syncthreads();
        }
        float* tmp = dhout; dhout = dxin; dxin = tmp;
    }
    for (int i = j; i < T * L; i += blockDim.x) dx[((long)s * T) * L + i] = dhout[i];
}

// Weight gradients of every layer in one launch.
//   grad block (same layout as the weight block): dW_ih = dG^T X, dW_hh = dG^T Hprev, db_ih = db_hh = colsum(dG)
// grid = (ceil(4L*(L+1)/256), 2, layers): y = 0 -> ih (+ b_ih), y = 1 -> hh (+ b_hh); column L is the bias.
__global__ __launch_bounds__(256) void lstm_wgrad_k(const float* __restrict__ dG, const float* __restrict__ hs_all,
                                                    const float* __restrict__ hprev, float* __restrict__ gblk,
                                                    int S, int T, int L, int accumulate) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= 4 * L * (L + 1)) return;
    const int l = blockIdx.z, hh = blockIdx.y;
    const int k = e / (4 * L), jrow = e - k * (4 * L);      // consecutive threads -> consecutive gate rows
    const long R = (long)S * T;
    const float* g = dG + (long)l * R * 4 * L + jrow;
    const float* x = (hh ? hprev + (long)l * R * L : hs_all + (long)l * R * L) + k;
    float acc = 0.f;
    if (k < L)
        for (long r = 0; r < R; ++r) acc = fmaf(g[r * 4 * L], x[r * L], acc);
    else
        for (long r = 0; r < R; ++r) acc += g[r * 4 * L];
    float* out = gblk + l * (8l * L * L + 8l * L);
    float* dst = k < L ? out + (hh ? 4l * L * L : 0) + (long)jrow * L + k
                       : out + 8l * L * L + (hh ? 4 * L : 0) + jrow;
    *dst = accumulate ? *dst + acc : acc;
}

}  // namespace rbvae

using namespace rbvae;

extern "C" {

int rbvae_lstm_fwd(const float* wblk, float* hs_all, float* hprev, float* acts, float* cs, int S, int T, int L,
                   int layers, void* stream) {
    RBVAE_CHECK_ARG(wblk && hs_all && S > 0 && T > 0 && L > 0 && layers > 0, "lstm_fwd: bad arguments");
    RBVAE_CHECK_ARG(L <= 128, "lstm_fwd: latent_dim %d > 128 is not supported", L);
    RBVAE_CHECK_ARG((acts == nullptr) == (cs == nullptr) && (acts == nullptr) == (hprev == nullptr),
                    "lstm_fwd: hprev/acts/cs must be given together");
    const int threads = ((4 * L + 63) / 64) * 64;
    const size_t lds = (size_t)(2 * T * L + 5 * L) * sizeof(float);
    RBVAE_CHECK_ARG(lds <= 64 * 1024, "lstm_fwd: T*L=%d too large", T * L);
    hipStream_t st = (hipStream_t)stream;
    if (L <= 32)
        hipLaunchKernelGGL(lstm_fwd_k<32>, dim3(S), dim3(threads), lds, st, wblk, hs_all, hprev, acts, cs, S, T, L, layers);
    else if (L <= 64)
        hipLaunchKernelGGL(lstm_fwd_k<64>, dim3(S), dim3(threads), lds, st, wblk, hs_all, hprev, acts, cs, S, T, L, layers);
    else
        hipLaunchKernelGGL(lstm_fwd_k<0>, dim3(S), dim3(threads), lds, st, wblk, hs_all, hprev, acts, cs, S, T, L, layers);
    RBVAE_CHECK_LAUNCH("lstm_fwd");
    return RBVAE_OK;
}

int rbvae_lstm_bwd(const float* wblk, const float* acts, const float* cs, const float* g_top, float* dG, float* dx,
                   int S, int T, int L, int layers, void* stream) {
    RBVAE_CHECK_ARG(wblk && acts && cs && g_top && dG && dx && S > 0 && T > 0 && L > 0 && layers > 0,
                    "lstm_bwd: bad arguments");
    RBVAE_CHECK_ARG(L <= 128, "lstm_bwd: latent_dim %d > 128 is not supported", L);
    const int threads = ((4 * L + 63) / 64) * 64;
    const size_t lds = (size_t)(2 * T * L + 13 * L) * sizeof(float);
    RBVAE_CHECK_ARG(lds <= 64 * 1024, "lstm_bwd: T*L=%d too large", T * L);
    hipStream_t st = (hipStream_t)stream;
    if (L <= 32)
        hipLaunchKernelGGL(lstm_bwd_k<32>, dim3(S), dim3(threads), lds, st, wblk, acts, cs, g_top, dG, dx, S, T, L, layers);
    else if (L <= 64)
        hipLaunchKernelGGL(lstm_bwd_k<64>, dim3(S), dim3(threads), lds, st, wblk, acts, cs, g_top, dG, dx, S, T, L, layers);
    else
        hipLaunchKernelGGL(lstm_bwd_k<0>, dim3(S), dim3(threads), lds, st, wblk, acts, cs, g_top, dG, dx, S, T, L, layers);
    RBVAE_CHECK_LAUNCH("lstm_bwd");
    return RBVAE_OK;
}

int rbvae_lstm_wgrad(const float* dG, const float* hs_all, const float* hprev, float* gblk, int S, int T, int L,
                     int layers, int accumulate, void* stream) {
    RBVAE_CHECK_ARG(dG && hs_all && hprev && gblk && S > 0 && T > 0 && L > 0 && layers > 0, "lstm_wgrad: bad arguments");
    dim3 grid(cdiv(4 * L * (L + 1), 256), 2, layers);
    hipLaunchKernelGGL(lstm_wgrad_k, grid, dim3(256), 0, (hipStream_t)stream, dG, hs_all, hprev, gblk, S, T, L,
                       accumulate);
    RBVAE_CHECK_LAUNCH("lstm_wgrad");
    return RBVAE_OK;
}

}  // extern "C"
