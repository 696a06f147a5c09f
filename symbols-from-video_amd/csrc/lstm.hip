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
#include <stdlib.h>

// Wave priority of the latency-bound kernels that share the chip with full-grid GEMMs on the side stream (the LSTM
// backward chain ran 18 + 25 us beside the decoder weight gradients against 14 + 14 us alone): s_setprio 3 gives
// their few waves the issue slots first.  -DLSTM_PRIO=0 builds without it (tools/ab_variants.sh).
#ifndef LSTM_PRIO
#define LSTM_PRIO 3
#endif
#if LSTM_PRIO
#define RBVAE_RAISE_PRIO() __builtin_amdgcn_s_setprio(LSTM_PRIO)
#else
#define RBVAE_RAISE_PRIO() do {} while (0)
#endif

namespace rbvae {

typedef __attribute__((ext_vector_type(2))) float f32x2_t;
int lstm_unit_threads = 1;       // 0: the gate-row kernels also at L == 32 (rbvae_dbg_lstm_unit_threads: identity tests, A/B timing)

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


// ---------------------------------------------------------------------------------------------
// Wavefront variants: every layer gets its own group of G = roundup64(4L) threads with its weight
// rows in registers, and the (layer, time) cells run along anti-diagonals: layer l works on time
// d - l at diagonal d.  The dependent chain shrinks from layers*T cells to T + layers - 1 steps,
// barriers wait on LDS only (global stores of the saved state stay in flight).
// ---------------------------------------------------------------------------------------------
// sigmoid / tanh on the hardware exp2 and reciprocal (about 1 ulp each): the wavefront kernels sit on a
// short dependent chain where the library expf/tanhf sequences would dominate.
__device__ __forceinline__ float fast_sigmoid(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float fast_tanh(float x) {
    return 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.8853900817779268f * x)) - 1.0f;
}

__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// ---------------------------------------------------------------------------------------------
// 32 < L <= 128 (the reference's sweeps and best models use latent_dim 50 / 75 / 100): one workgroup per sequence,
// layer by layer.  A layer's two weight matrices do not fit one thread's registers any more (2L values per gate row),
// and all layers' do not fit a CU, so the work is cut the other way:
//   * the input half  W_ih x_t  does not depend on the recurrence: it is computed for ALL time steps first
//     (T x 4L dot products, the weights of that half in registers), then the registers are reloaded with W_hh
//     and only  W_hh h_{t-1}  sits on the dependent chain;
//   * one lane per gate row, L <= 128 weights in its registers, at most 512 threads (two waves per SIMD: a 256-register
//     budget -- with two lanes per row and 1024 threads the 128-register budget spilled, and every spill reload in
//     the time loop waited, through the shared vmcnt counter, for the step's global stores: 2.3 us per step).
// The layer-sequential kernel this replaces for these sizes read its weights from memory inside the time loop
// (L > 64) or spilled them (L <= 64): 3.76 ms per fused step at L = 100 against 0.46 ms at L = 32.
// Vector rows (x_t, h_t) live in LDS at a stride of 128 floats, zero-filled, so 16-byte reads past L see zeros (the
// weights there are zero too).
// ---------------------------------------------------------------------------------------------
constexpr int BIG_VS = 128;      // LDS stride of a vector row
constexpr int BIG_W = 128;       // weights per lane

// NCH = 16-byte chunks per dot product, a compile-time constant: with a run-time bound every chunk sat in its own basic
// block and its LDS read was waited for on the spot (a dependent LDS round trip per chunk); now the reads of eight
// chunks are in flight together.
template <int NCH>
__device__ __forceinline__ float big_dot(const float (&w)[BIG_W], const float* __restrict__ vec) {
    // four accumulator chains as two float2 pairs: a pair's fused multiply-adds issue as one packed instruction
    // (v_pk_fma_f32); each chain sees the operands of the scalar form in the same order (bit-identical sums)
    f32x2_t a01 = f32x2_t{0.f, 0.f}, a23 = f32x2_t{0.f, 0.f};
    constexpr int CH = NCH > 28 ? 4 : 8;           // reads in flight (the full 128-weight rows leave fewer registers)
#pragma unroll
    for (int i0 = 0; i0 < NCH; i0 += CH) {
        float4 v[CH];
#pragma unroll
        for (int i = 0; i < CH; ++i)
            if (i0 + i < NCH) v[i] = *(const float4*)(vec + 4 * (i0 + i));
#pragma unroll
        for (int i = 0; i < CH; ++i)
            if (i0 + i < NCH) {
                a01 = __builtin_elementwise_fma(f32x2_t{w[4 * (i0 + i)], w[4 * (i0 + i) + 1]}, f32x2_t{v[i].x, v[i].y}, a01);
                a23 = __builtin_elementwise_fma(f32x2_t{w[4 * (i0 + i) + 2], w[4 * (i0 + i) + 3]}, f32x2_t{v[i].z, v[i].w}, a23);
            }
        __builtin_amdgcn_sched_barrier(0);
    }
    return (a01[0] + a01[1]) + (a23[0] + a23[1]);
}

// w[i] = base[i * step] for i < n (else 0): unconditional loads on clamped (always valid) indices, eight in flight,
// then a select -- a predicated load per element became a chain of dependent memory round trips
template <int NCH>
__device__ __forceinline__ void big_load_w(float (&w)[BIG_W], const float* __restrict__ base, int step, int n, bool on) {
    int n1 = max(n, 1) - 1;
    // opaque to the optimiser: otherwise the 128 clamped 64-bit addresses are computed once outside the layer loop and
    // live (spilled) across the time loops
    asm volatile("" : "+v"(n1));
    // every load goes straight into its weight register and nothing is used before the last one is issued: the whole
    // row is in flight at once (with a select inside each batch of eight the batches were 13 dependent round trips per
    // reload, two reloads per layer).  The scheduling barriers keep the address arithmetic to eight loads at a time.
#pragma unroll
    for (int i0 = 0; i0 < 4 * NCH; i0 += 8) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (i0 + i < 4 * NCH) w[i0 + i] = base[min(i0 + i, n1) * step];
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = 0; i < 4 * NCH; ++i) w[i] = (on && i < n) ? w[i] : 0.f;
}

template <int NCH>
__global__ __launch_bounds__(512) void lstm_fwd_big_k(const float* __restrict__ wblk, const float* __restrict__ wT,
                                                      float* __restrict__ hs_all, float* __restrict__ hprev,
                                                      float* __restrict__ acts, float* __restrict__ cs, int S, int T,
                                                      int L, int layers, int RP) {
    RBVAE_RAISE_PRIO();
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* xin = sm;                         // [T][BIG_VS]
    float* hout = xin + T * BIG_VS;          // [T][BIG_VS]
    float* hcur = hout + T * BIG_VS;         // [BIG_VS]
    float* gates = hcur + BIG_VS;            // [RP]
    float* gx = gates + RP;                  // [T][RP]: W_ih x_t + b_ih + b_hh
    const int tid = threadIdx.x, j = tid, s = blockIdx.x;
    const bool row = j < 4 * L;
    const int jc = row ? j : 0;
    const bool is_g = j >= 2 * L && j < 3 * L;
    for (int i = tid; i < (2 * T + 1) * BIG_VS; i += blockDim.x) sm[i] = 0.f;
    __syncthreads();
    for (int i = tid; i < T * L; i += blockDim.x) xin[(i / L) * BIG_VS + i % L] = hs_all[((long)s * T) * L + i];
    float w[BIG_W];
    for (int l = 0; l < layers; ++l) {
        const float* wl = wblk + l * lstm_layer_floats(L);
        const float bsum = row ? wl[8l * L * L + j] + wl[8l * L * L + 4 * L + j] : 0.f;
        auto load_w = [&](int which) {
            const float* base = wT ? wT + ((long)(l * 2 + which) * L) * 4 * L + jc
                                   : wl + (long)which * 4 * L * L + (long)jc * L;
            big_load_w<NCH>(w, base, wT ? 4 * L : 1, L, row);
        };
        load_w(0);
        __syncthreads();                     // xin complete (the input, or the layer below)
#pragma unroll 1
        for (int t = 0; t < T; ++t) {
            const float a = big_dot<NCH>(w, xin + t * BIG_VS);
            if (row) gx[t * RP + j] = a + bsum;
        }
        load_w(1);
        if (tid < BIG_VS) hcur[tid] = 0.f;
        float c = 0.f;
        __syncthreads();
#pragma unroll 1
        for (int t = 0; t < T; ++t) {
            const float a = big_dot<NCH>(w, hcur);
            const long o = (((long)l * S + s) * T + t);
            if (row) {
                const float pre = gx[t * RP + j] + a;
                const float av = is_g ? fast_tanh(pre) : fast_sigmoid(pre);
                gates[j] = av;
                if (acts) acts[o * 4 * L + j] = av;
            }
            lds_barrier();
            if (tid < L) {
                const float ig = gates[tid], fg = gates[L + tid], gg = gates[2 * L + tid], og = gates[3 * L + tid];
                const float hp = hcur[tid];
                c = fmaf(fg, c, ig * gg);
                const float h = og * fast_tanh(c);
                if (acts) {
                    cs[o * L + tid] = c;
                    hprev[o * L + tid] = hp;
                }
                hout[t * BIG_VS + tid] = h;
                hcur[tid] = h;
                hs_all[(((long)(l + 1) * S + s) * T + t) * L + tid] = h;
            }
            lds_barrier();
        }
        float* tmp = xin; xin = hout; hout = tmp;
    }
}

// BPTT of the same: a hidden unit (column of W) is shared by FOUR adjacent lanes (L <= 128 gate rows each).  Per time
// step only  W_hh^T dG_{t+1}  is on the dependent chain; the input gradients  W_ih^T dG_t  of all time steps follow in
// one batch per layer.  dG rows sit in LDS in four 132-float segments (one per lane of a group: conflict-free 16-byte
// reads).
constexpr int BIG_SEG = 132;
template <int CTRL> __device__ __forceinline__ float big_quad(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
template <int NCH>
__global__ __launch_bounds__(512) void lstm_bwd_big_k(const float* __restrict__ wblk, const float* __restrict__ wT,
                                                      const float* __restrict__ acts, const float* __restrict__ cs,
                                                      const float* __restrict__ g_top, float* __restrict__ dG,
                                                      float* __restrict__ dx, int S, int T, int L, int layers) {
    RBVAE_RAISE_PRIO();
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* dhout = sm;                       // [T][BIG_VS] gradient wrt this layer's outputs
    float* dxin = dhout + T * BIG_VS;        // [T][BIG_VS] gradient wrt this layer's inputs
    float* dhrec = dxin + T * BIG_VS;        // [BIG_VS]
    float* dgall = dhrec + BIG_VS;           // [T][4 * BIG_SEG] gate gradients of every time step
    constexpr int DS = 4 * BIG_SEG;
    const int tid = threadIdx.x, k = tid >> 2, p = tid & 3, s = blockIdx.x;
    const int RPP = (L + 3) & ~3;            // gate rows per lane of a group: 4L / 4, rounded up to a multiple of 4
    const int r0 = p * RPP;
    const int nrow = min(max(4 * L - r0, 0), RPP);
    const bool col = k < L;
    const int kc = col ? k : 0;
    auto seg = [&](int r) { return (r / RPP) * BIG_SEG + r % RPP; };
    for (int i = tid; i < (2 * T + 1) * BIG_VS + T * DS; i += blockDim.x) sm[i] = 0.f;
    __syncthreads();
    for (int i = tid; i < T * L; i += blockDim.x) dhout[(i / L) * BIG_VS + i % L] = g_top[((long)s * T) * L + i];
    float w[BIG_W];
    for (int l = layers - 1; l >= 0; --l) {
        const float* wl = wblk + l * lstm_layer_floats(L);
        auto load_w = [&](int which) {       // w[i] = W_which[r0 + i][k]
            // from the row-major block, not from wT: a wave's 16 hidden units x 4 row groups then read four 64-byte runs
            // per instruction; through wT every lane would walk its own cache line (64 lines per instruction: the
            // weight reloads alone took ~50 us of a 160 us launch at L = 100)
            const float* base = wl + (long)which * 4 * L * L + (long)min(r0, 4 * L - 1) * L + kc;
            big_load_w<NCH>(w, base, L, nrow, col);
        };
        load_w(1);
        if (tid < BIG_VS) dhrec[tid] = 0.f;
        float dc_next = 0.f;
        // the saved gates / cell states of step t are loaded one step ahead (as loads inside the step they put a
        // memory round trip on the dependent chain of every time step)
        const int uu = tid < L ? tid : 0;
        const long ob = ((long)l * S + s) * T;
        float n_ig, n_fg, n_gg, n_og, n_c, n_cp;
        {
            const float* ap = acts + (ob + T - 1) * 4 * L;
            n_ig = ap[uu]; n_fg = ap[L + uu]; n_gg = ap[2 * L + uu]; n_og = ap[3 * L + uu];
            n_c = cs[(ob + T - 1) * L + uu];
            n_cp = cs[(ob + max(T - 2, 0)) * L + uu];
        }
        __syncthreads();
#pragma unroll 1
        for (int t = T - 1; t >= 0; --t) {
            const long o = ob + t;
            const float ig = n_ig, fg = n_fg, gg = n_gg, og = n_og, c = n_c, cprev = t > 0 ? n_cp : 0.f;
            {
                const int tn = max(t - 1, 0);
                const float* ap = acts + (ob + tn) * 4 * L;
                n_ig = ap[uu]; n_fg = ap[L + uu]; n_gg = ap[2 * L + uu]; n_og = ap[3 * L + uu];
                n_c = n_cp;
                n_cp = cs[(ob + max(tn - 1, 0)) * L + uu];
            }
            if (tid < L) {
                const int u = tid;
                const float tc = fast_tanh(c);
                const float dh = dhout[t * BIG_VS + u] + dhrec[u];
                const float dc = dc_next + dh * og * (1.f - tc * tc);
                const float d_o = dh * tc * og * (1.f - og);
                const float d_i = dc * gg * ig * (1.f - ig);
                const float d_f = dc * cprev * fg * (1.f - fg);
                const float d_g = dc * ig * (1.f - gg * gg);
                dc_next = dc * fg;
                float* dl = dgall + t * DS;
                dl[seg(u)] = d_i; dl[seg(L + u)] = d_f; dl[seg(2 * L + u)] = d_g; dl[seg(3 * L + u)] = d_o;
                float* gp = dG + o * 4 * L;
                gp[u] = d_i; gp[L + u] = d_f; gp[2 * L + u] = d_g; gp[3 * L + u] = d_o;
            }
            lds_barrier();
            float a = big_dot<NCH>(w, dgall + t * DS + p * BIG_SEG);
            a += big_quad<0xB1>(a);          // + lane ^ 1   (DPP quad_perm [1,0,3,2]; a + b == b + a: the sums of __shfl_xor)
            a += big_quad<0x4E>(a);          // + lane ^ 2   (quad_perm [2,3,0,1])
            if (p == 0 && col) dhrec[k] = a;
            lds_barrier();
        }
        load_w(0);
#pragma unroll 1
        for (int t = 0; t < T; ++t) {
            float a = big_dot<NCH>(w, dgall + t * DS + p * BIG_SEG);
            a += big_quad<0xB1>(a);          // + lane ^ 1   (DPP quad_perm [1,0,3,2]; a + b == b + a: the sums of __shfl_xor)
            a += big_quad<0x4E>(a);          // + lane ^ 2   (quad_perm [2,3,0,1])
            if (p == 0 && col) dxin[t * BIG_VS + k] = a;
        }
        __syncthreads();
        float* tmp = dhout; dhout = dxin; dxin = tmp;
    }
    for (int i = tid; i < T * L; i += blockDim.x) dx[((long)s * T) * L + i] = dhout[(i / L) * BIG_VS + i % L];
}

template <int LMAX, bool VEC, bool EXACT>       // EXACT: L == LMAX (every bound check folds away)
__global__ __launch_bounds__(1024) void lstm_fwd_wave_k(const float* __restrict__ wblk, const float* __restrict__ wT,
                                                        float* __restrict__ hs_all,
                                                        float* __restrict__ hprev, float* __restrict__ acts,
                                                        float* __restrict__ cs, int S, int T, int L_, int layers,
                                                        int G, const float* __restrict__ in_parts, int nparts,
                                                        long part_stride, void* __restrict__ cast_out, int cast_bf16,
                                                        int cast_ld) {
    RBVAE_RAISE_PRIO();
    const int L = EXACT ? LMAX : L_;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* hbuf = sm;                               // [layers+1][T][L]
    float* gates = hbuf + (layers + 1) * T * L;     // [layers][4L]  ACTIVATED gates i, f, g, o
    const int l = threadIdx.x / G, j = threadIdx.x - l * G;
    const int s = blockIdx.x;
    const bool row = j < 4 * L;
    const bool is_g = j >= 2 * L && j < 3 * L;      // the tanh gate
    if (in_parts) {
        // the stack input arrives as K-split slabs of the fc product (rbvae_skinny_linear_parts): summed here in
        // slab order and written to slot 0, where the backward pass expects the input
        for (int i = threadIdx.x; i < T * L; i += blockDim.x) {
            const long o = ((long)s * T) * L + i;
            float v = in_parts[o];
            for (int q = 1; q < nparts; ++q) v += in_parts[q * part_stride + o];
            hbuf[i] = v;
            hs_all[o] = v;
        }
    } else {
        for (int i = threadIdx.x; i < T * L; i += blockDim.x) hbuf[i] = hs_all[((long)s * T) * L + i];
    }
    if (cast_out) {
        // padding columns [L, cast_ld) of this sequence's rows of the cast copy (the consumer GEMM reads whole
        // 128-byte K slices)
        const int pw = cast_ld - L;
        for (int i = threadIdx.x; i < T * pw; i += blockDim.x) {
            const long o = ((long)s * T + i / pw) * cast_ld + L + i % pw;
            if (cast_bf16) ((bf16_t*)cast_out)[o] = 0; else ((float*)cast_out)[o] = 0.f;
        }
    }
    const float* wl = wblk + l * lstm_layer_floats(L);
    float wih[LMAX], whh[LMAX];
    float bsum = 0.f;
    {
        const int jc = row ? j : 0;
        bsum = wl[8l * L * L + jc] + wl[8l * L * L + 4 * L + jc];
        // transposed copies ([k][gate row]): consecutive threads read consecutive addresses
        const float* pi = wT ? wT + (long)(l * 2) * L * 4 * L + jc : wl + jc * L;
        const float* ph = wT ? pi + L * 4 * L : pi + 4 * L * L;
        const int kstride = wT ? 4 * L : 1;
#pragma unroll
        for (int k = 0; k < LMAX; ++k) {
            const int kk = (k < L ? k : L - 1) * kstride;
            const float a = pi[kk], b = ph[kk];
            wih[k] = k < L ? a : 0.f;
            whh[k] = k < L ? b : 0.f;
        }
    }
    float c = 0.f;
    // slots 1.. are read (times a zero factor) before they are written: keep them finite
    for (int i = T * L + threadIdx.x; i < (layers + 1) * T * L; i += blockDim.x) hbuf[i] = 0.f;
    __syncthreads();
    const int ndiag = T + layers - 1;
    for (int d = 0; d < ndiag; ++d) {
        const int t = d - l;
        const bool active = t >= 0 && t < T;
        if (active && row) {
            const float* xt = hbuf + (l * T + t) * L;
            const float* hp = hbuf + ((l + 1) * T + (t > 0 ? t - 1 : 0)) * L;
            const float hscale = t > 0 ? 1.f : 0.f;      // h_{-1} = 0
            float a0 = bsum, a1 = 0.f, a2 = 0.f, a3 = 0.f;
            float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;
            if constexpr (VEC) {
#pragma unroll
                for (int k = 0; k < LMAX; k += 4) {
                    if (k < L) {
                        const float4 xv = *(const float4*)(xt + k), hv = *(const float4*)(hp + k);
                        a0 = fmaf(wih[k], xv.x, a0); a1 = fmaf(wih[k + 1], xv.y, a1);
                        a2 = fmaf(wih[k + 2], xv.z, a2); a3 = fmaf(wih[k + 3], xv.w, a3);
                        b0 = fmaf(whh[k], hv.x, b0); b1 = fmaf(whh[k + 1], hv.y, b1);
                        b2 = fmaf(whh[k + 2], hv.z, b2); b3 = fmaf(whh[k + 3], hv.w, b3);
                    }
                }
            } else {
#pragma unroll
                for (int k = 0; k < LMAX; ++k)
                    if (k < L) { a0 = fmaf(wih[k], xt[k], a0); b0 = fmaf(whh[k], hp[k], b0); }
            }
            const float pre = ((a0 + a1) + (a2 + a3)) + hscale * ((b0 + b1) + (b2 + b3));
            const float av = is_g ? fast_tanh(pre) : fast_sigmoid(pre);
            gates[l * 4 * L + j] = av;
            if (acts) acts[(((long)l * S + s) * T + t) * 4 * L + j] = av;
        }
        lds_barrier();
        if (active && j < L) {
            const float* gl = gates + l * 4 * L;
            const float ig = gl[j], fg = gl[L + j], gg = gl[2 * L + j], og = gl[3 * L + j];
            const float hp = t > 0 ? hbuf[((l + 1) * T + t - 1) * L + j] : 0.f;
            c = fmaf(fg, c, ig * gg);        // explicit: the same rounding in every kernel that runs this cell
            const float h = og * fast_tanh(c);
            hbuf[((l + 1) * T + t) * L + j] = h;
            const long o = (((long)l * S + s) * T + t);
            if (acts) {
                cs[o * L + j] = c;
                hprev[o * L + j] = hp;
            }
            hs_all[(((long)(l + 1) * S + s) * T + t) * L + j] = h;
            if (cast_out && l == layers - 1) {
                // the top layer's output again in the next GEMM's operand type (what rbvae_cast_pad would make of it)
                const long o = ((long)s * T + t) * cast_ld + j;
                if (cast_bf16) ((bf16_t*)cast_out)[o] = f32_to_bf16(h); else ((float*)cast_out)[o] = h;
            }
        }
        lds_barrier();
    }
}

// ---------------------------------------------------------------------------------------------
// Encoder stack -> Binary-Concrete binarise -> decoder stack as ONE wavefront (percep_RBVAE_model.py:155-163):
// 2*layers thread groups, group q works on time d - q at diagonal d; the encoder's top group turns its h_t into
// z_t = binarise(h_t, U_t) in the same pointwise phase, so the decoder's bottom group reads it one diagonal later.
// T + 2*layers - 1 dependent steps and one launch instead of two stacks of T + layers - 1 plus the binarise
// kernel between them.  Arithmetic per element is that of lstm_fwd_wave_k and binarize_kl_fwd_parts_k (same
// functions, same order): results are identical to the three-launch path (explicit fma in the cell update, so not even the
// compiler's contraction choices differ).
// ---------------------------------------------------------------------------------------------
struct PairArgs {
    const float *wblk_e, *wT_e, *wblk_d, *wT_d;
    float *hs_e, *hp_e, *acts_e, *cs_e;          // encoder stack: hs [layers+1][S][T][L] (slot 0 = input), saved state
    float *hs_d, *hp_d, *acts_d, *cs_d;          // decoder stack: slot 0 receives z
    const float* in_parts; int nparts; long part_stride;     // encoder input as K-split slabs (or null)
    const float* U; float* y_soft; float* kl_parts;          // uniform noise (null: counter hash), y, per-sequence KL sums
    float tau; const float* tau_dev;                         // temperature by value, or (when set) from a device float
    float ratio, neps, lp, l1p, keps; int hard, clamp;
    unsigned long long seed; const unsigned long long* seed_dev;
    void* cast_out; int cast_bf16, cast_ld;                   // decoder top layer once more, cast / padded
    int S, T, L, layers, G;
};

// LC: compile-time length of the dot products (a multiple of 4 with roundup4(L) == LC), 0 = the run-time L: with a
// run-time bound every 16-byte chunk sits in its own basic block
template <int LMAX, bool EXACT, int LC = 0>
__global__ __launch_bounds__(1024) void lstm_pair_fwd_k(const PairArgs p) {
    RBVAE_RAISE_PRIO();
    const int L = EXACT ? LMAX : p.L;
    const int T = p.T, S = p.S, layers = p.layers, G = p.G;
    const float inv_tau_src = p.tau_dev ? p.tau_dev[0] : p.tau;   // uniform: one scalar load at the top of the kernel
    extern __shared__ __attribute__((aligned(16))) float sm[];
    // LDS rows of L values sit at a stride of LS = roundup4(L) with zero padding, so the 16-byte reads of the dot
    // products stay aligned for any L (the reference's most common latent_dim is 25)
    const int LS = EXACT ? LMAX : ((L + 3) & ~3);
    float* hbuf = sm;                                    // [2][layers+1][T][LS]: stack, slot, time
    float* gates = hbuf + 2 * (layers + 1) * T * LS;     // [2*layers][4L] activated gates
    float* nbuf = gates + 2 * layers * 4 * L;            // [T][LS] noise term of the binarisation
    float* red = nbuf + T * LS;                          // [16] block reduction
    const int q = threadIdx.x / G, j = threadIdx.x - q * G;         // global layer index, gate row
    const int stack = q >= layers, l = q - (stack ? layers : 0);
    const int s = blockIdx.x;
    const bool row = j < 4 * L;
    const bool is_g = j >= 2 * L && j < 3 * L;
    float* hs_all = stack ? p.hs_d : p.hs_e;
    float* hprev = stack ? p.hp_d : p.hp_e;
    float* acts = stack ? p.acts_d : p.acts_e;
    float* cs = stack ? p.cs_d : p.cs_e;
    float* hb = hbuf + stack * (layers + 1) * T * LS;    // this stack's slots
    // this thread's weight rows first: their loads are in flight while the input is staged
    const float* wblk = stack ? p.wblk_d : p.wblk_e;
    const float* wT = stack ? p.wT_d : p.wT_e;
    const float* wl = wblk + l * lstm_layer_floats(L);
    float wih[LMAX], whh[LMAX];
    float bsum = 0.f;
    {
        const int jc = row ? j : 0;
        bsum = wl[8l * L * L + jc] + wl[8l * L * L + 4 * L + jc];
        const float* pi = wT ? wT + (long)(l * 2) * L * 4 * L + jc : wl + jc * L;
        const float* ph = wT ? pi + L * 4 * L : pi + 4 * L * L;
        const int kstride = wT ? 4 * L : 1;
#pragma unroll
        for (int k = 0; k < LMAX; ++k) {
            const int kk = (k < L ? k : L - 1) * kstride;
            const float a = pi[kk], b = ph[kk];
            wih[k] = k < L ? a : 0.f;
            whh[k] = k < L ? b : 0.f;
        }
    }
    // ---- encoder input (plain or K-split slabs) and the noise term, staged by all threads.  The slab loads of an
    // element are issued together (summed in slab order all the same) and the step counter is read once: as a loop of
    // dependent loads this prologue cost five memory round trips before the first time step.
    for (int i = threadIdx.x; i < T * L; i += blockDim.x) {
        const long o = ((long)s * T) * L + i;
        // unconditional loads on always-valid addresses (absent operands alias the input element / a weight): nothing
        // here branches, so all of them are in flight before the first is used
        const bool parts = p.in_parts != nullptr;
        const float* ip = parts ? p.in_parts + o : p.hs_e + o;
        const long st = parts ? p.part_stride : 0;
        const int np = parts ? p.nparts : 1;
        const float a0 = ip[0];
        const float a1 = ip[(np > 1 ? 1 : 0) * st];
        const float a2 = ip[(np > 2 ? 2 : 0) * st];
        const float a3 = ip[(np > 3 ? 3 : 0) * st];
        const float uin = *(p.U ? p.U + o : ip);
        const unsigned long long sdev = *(p.seed_dev ? p.seed_dev : (const unsigned long long*)p.wblk_e);
        float v = a0;
        v += np > 1 ? a1 : 0.f;
        v += np > 2 ? a2 : 0.f;
        v += np > 3 ? a3 : 0.f;
        for (int k = 4; k < np; ++k) v += ip[k * st];
        if (parts) p.hs_e[o] = v;
        const int li = EXACT ? i : (i / L) * LS + i % L;
        hbuf[li] = v;
        const unsigned long long seed = p.seed + (p.seed_dev ? sdev * 0x9E3779B97F4A7C15ull : 0ull);
        const float u = p.U ? uin : (float)(hash_u32(seed, (unsigned long long)o) >> 8) * (1.0f / 16777216.0f);
        nbuf[li] = p.ratio * (logf(u + p.neps) - logf(1.0f - u + p.neps));
    }
    if (p.cast_out) {
        const int pw = p.cast_ld - L;
        for (int i = threadIdx.x; i < T * pw; i += blockDim.x) {
            const long o = ((long)s * T + i / pw) * p.cast_ld + L + i % pw;
            if (p.cast_bf16) ((bf16_t*)p.cast_out)[o] = 0; else ((float*)p.cast_out)[o] = 0.f;
        }
    }
    float c = 0.f;
    // every slot but the encoder input is read (times a zero factor) before it is written: keep them finite
    // (and the padding of the input rows: read by the last 16-byte chunk of a dot product)
    for (int i = threadIdx.x; i < 2 * (layers + 1) * T * LS; i += blockDim.x)
        if (i >= T * LS || (!EXACT && i % LS >= L)) hbuf[i] = 0.f;
    __syncthreads();
    const int ndiag = T + 2 * layers - 1;
    for (int d = 0; d < ndiag; ++d) {
        const int t = d - q;
        const bool active = t >= 0 && t < T;
        if (active && row) {
            const float* xt = hb + (l * T + t) * LS;
            const float* hp = hb + ((l + 1) * T + (t > 0 ? t - 1 : 0)) * LS;
            const float hscale = t > 0 ? 1.f : 0.f;
            // four accumulator chains per dot product, kept as two float2 pairs: the fused multiply-adds of a pair issue
            // as ONE packed instruction (v_pk_fma_f32, two FMAs per lane at the issue cost of one); every chain sees
            // the same operands in the same order as the scalar form, so the sums are bit for bit the same
            f32x2_t a01 = f32x2_t{bsum, 0.f}, a23 = f32x2_t{0.f, 0.f}, b01 = f32x2_t{0.f, 0.f}, b23 = f32x2_t{0.f, 0.f};
#pragma unroll
            for (int k = 0; k < LMAX; k += 4) {
                if (LC ? k < LC : k < L) {
                    const float4 xv = *(const float4*)(xt + k), hv = *(const float4*)(hp + k);
                    a01 = __builtin_elementwise_fma(f32x2_t{wih[k], wih[k + 1]}, f32x2_t{xv.x, xv.y}, a01);
                    a23 = __builtin_elementwise_fma(f32x2_t{wih[k + 2], wih[k + 3]}, f32x2_t{xv.z, xv.w}, a23);
                    b01 = __builtin_elementwise_fma(f32x2_t{whh[k], whh[k + 1]}, f32x2_t{hv.x, hv.y}, b01);
                    b23 = __builtin_elementwise_fma(f32x2_t{whh[k + 2], whh[k + 3]}, f32x2_t{hv.z, hv.w}, b23);
                }
            }
            const float pre = ((a01[0] + a01[1]) + (a23[0] + a23[1])) + hscale * ((b01[0] + b01[1]) + (b23[0] + b23[1]));
            const float av = is_g ? fast_tanh(pre) : fast_sigmoid(pre);
            gates[q * 4 * L + j] = av;
            if (acts) acts[(((long)l * S + s) * T + t) * 4 * L + j] = av;
        }
        lds_barrier();
        if (active && j < L) {
            const float* gl = gates + q * 4 * L;
            const float ig = gl[j], fg = gl[L + j], gg = gl[2 * L + j], og = gl[3 * L + j];
            const float hpv = t > 0 ? hb[((l + 1) * T + t - 1) * LS + j] : 0.f;
            c = fmaf(fg, c, ig * gg);        // explicit: the same rounding in every kernel that runs this cell
            const float h = og * fast_tanh(c);
            hb[((l + 1) * T + t) * LS + j] = h;
            const long o = (((long)l * S + s) * T + t);
            if (acts) {
                cs[o * L + j] = c;
                hprev[o * L + j] = hpv;
            }
            hs_all[(((long)(l + 1) * S + s) * T + t) * L + j] = h;
            if (q == layers - 1) {
                // binary_concrete_logits on the encoder's output (percep_RBVAE_model.py:17-44): z feeds the decoder stack
                const long e = ((long)s * T + t) * L + j;
                const float y = sigmoidf_((h + nbuf[t * LS + j]) / inv_tau_src);
                const float zz = p.hard ? (y > 0.5f ? 1.0f : 0.0f) : y;
                p.y_soft[e] = y;
                p.hs_d[e] = zz;
                hbuf[(layers + 1) * T * LS + t * LS + j] = zz;        // decoder stack, slot 0
            }
            if (p.cast_out && q == 2 * layers - 1) {
                const long oc = ((long)s * T + t) * p.cast_ld + j;
                if (p.cast_bf16) ((bf16_t*)p.cast_out)[oc] = f32_to_bf16(h); else ((float*)p.cast_out)[oc] = h;
            }
        }
        lds_barrier();
    }
    if (p.kl_parts) {
        // KL of this sequence's T*L codes (kl_binary_concrete on the sample, percep_RBVAE_train.py:528), off the
        // dependent chain: one element per thread, fixed-order block sum
        float a = 0.f;
        for (int i = threadIdx.x; i < T * L; i += blockDim.x)
            a += kl_elem(hbuf[(layers + 1) * T * LS + (EXACT ? i : (i / L) * LS + i % L)], p.lp, p.l1p, p.keps, p.clamp);
        const float tot = block_sum(a, red);
        if (threadIdx.x == 0) p.kl_parts[s] = tot;
    }
}

// lanes 0-31 receive the value of lane + 32 (v_permlane32_swap_b32 with both operands the same register: the second result
// holds [upper half, upper half]); one vector instruction where __shfl_xor(v, 32) is an LDS-crossbar round trip
__device__ __forceinline__ float from_upper_half(float v) {
    const unsigned a = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane32_swap(a, a, false, false);
    return __uint_as_float(r[1]);
}

// The same wavefront with one WAVE per layer (L == 32): lane (unit j, half hf) holds two gate rows of its unit -- i, f in
// lanes 0-31, g, o in lanes 32-63 (128 weights in registers) -- forms their pre-activations with the same four accumulator
// chains per dot product, and lanes 0-31 fetch g, o from their partner lanes (two v_permlane32_swap) and update the cell at
// once: the activated gates never pass through LDS, so a diagonal needs ONE workgroup barrier of 2 * layers waves instead
// of two barriers of 16 waves, and the x_t / h_{t-1} vectors are read by half the lanes (the gate-row form spends a third
// of a step in those broadcast reads: bench step 0.4269 -> 0.4223 ms with three quarters of them removed).  Same
// expressions in the same order as lstm_pair_fwd_k: identical results.  A diagonal writes h_t of (layer, t) and reads rows
// (layer, t-1), (layer-1, t): distinct LDS rows, so the single barrier per diagonal orders everything.  (All four gate rows
// in one thread, 256 weights: the compiler parks a quarter of them in accumulation registers and moves them back every
// diagonal -- slower than the gate-row form.)
__global__ __launch_bounds__(512) void lstm_pair_fwd_unit_k(const PairArgs p) {
    RBVAE_RAISE_PRIO();
    constexpr int L = 32, LS = 32;
    const int T = p.T, S = p.S, layers = p.layers;
    const float inv_tau_src = p.tau_dev ? p.tau_dev[0] : p.tau;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* hbuf = sm;                                    // [2][layers+1][T][LS]: stack, slot, time
    float* nbuf = hbuf + 2 * (layers + 1) * T * LS;      // [T][LS] noise term of the binarisation
    float* red = nbuf + T * LS;                          // [16] block reduction
    const int q = threadIdx.x >> 6, j = threadIdx.x & 31, hf = (threadIdx.x >> 5) & 1;      // global layer, unit, gate pair
    const int stack = q >= layers, l = q - (stack ? layers : 0);
    const int s = blockIdx.x;
    float* hs_all = stack ? p.hs_d : p.hs_e;
    float* hprev = stack ? p.hp_d : p.hp_e;
    float* acts = stack ? p.acts_d : p.acts_e;
    float* cs = stack ? p.cs_d : p.cs_e;
    float* hb = hbuf + stack * (layers + 1) * T * LS;
    const float* wblk = stack ? p.wblk_d : p.wblk_e;
    const float* wT = stack ? p.wT_d : p.wT_e;
    const float* wl = wblk + l * lstm_layer_floats(L);
    float wih[2][L], whh[2][L], bsum[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int r = (2 * hf + g) * L + j;
        bsum[g] = wl[8l * L * L + r] + wl[8l * L * L + 4 * L + r];
        const float* pi = wT ? wT + (long)(l * 2) * L * 4 * L + r : wl + r * L;
        const float* ph = wT ? pi + L * 4 * L : pi + 4 * L * L;
        const int kstride = wT ? 4 * L : 1;
#pragma unroll
        for (int k = 0; k < L; ++k) { wih[g][k] = pi[k * kstride]; whh[g][k] = ph[k * kstride]; }
    }
    for (int i = threadIdx.x; i < T * L; i += blockDim.x) {
        const long o = ((long)s * T) * L + i;
        const bool parts = p.in_parts != nullptr;
        const float* ip = parts ? p.in_parts + o : p.hs_e + o;
        const long st = parts ? p.part_stride : 0;
        const int np = parts ? p.nparts : 1;
        const float a0 = ip[0];
        const float a1 = ip[(np > 1 ? 1 : 0) * st];
        const float a2 = ip[(np > 2 ? 2 : 0) * st];
        const float a3 = ip[(np > 3 ? 3 : 0) * st];
        const float uin = *(p.U ? p.U + o : ip);
        const unsigned long long sdev = *(p.seed_dev ? p.seed_dev : (const unsigned long long*)p.wblk_e);
        float v = a0;
        v += np > 1 ? a1 : 0.f;
        v += np > 2 ? a2 : 0.f;
        v += np > 3 ? a3 : 0.f;
        for (int k = 4; k < np; ++k) v += ip[k * st];
        if (parts) p.hs_e[o] = v;
        hbuf[i] = v;
        const unsigned long long seed = p.seed + (p.seed_dev ? sdev * 0x9E3779B97F4A7C15ull : 0ull);
        const float u = p.U ? uin : (float)(hash_u32(seed, (unsigned long long)o) >> 8) * (1.0f / 16777216.0f);
        nbuf[i] = p.ratio * (logf(u + p.neps) - logf(1.0f - u + p.neps));
    }
    if (p.cast_out) {
        const int pw = p.cast_ld - L;
        for (int i = threadIdx.x; i < T * pw; i += blockDim.x) {
            const long o = ((long)s * T + i / pw) * p.cast_ld + L + i % pw;
            if (p.cast_bf16) ((bf16_t*)p.cast_out)[o] = 0; else ((float*)p.cast_out)[o] = 0.f;
        }
    }
    float c = 0.f;
    for (int i = threadIdx.x; i < 2 * (layers + 1) * T * LS; i += blockDim.x)
        if (i >= T * LS) hbuf[i] = 0.f;
    __syncthreads();
    const int ndiag = T + 2 * layers - 1;
    for (int d = 0; d < ndiag; ++d) {
        const int t = d - q;                              // wave-uniform
        if (t >= 0 && t < T) {
            const float* xt = hb + (l * T + t) * LS;
            const float* hp = hb + ((l + 1) * T + (t > 0 ? t - 1 : 0)) * LS;
            const float hscale = t > 0 ? 1.f : 0.f;
            f32x2_t a01[2], a23[2], b01[2], b23[2];
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                a01[g] = f32x2_t{bsum[g], 0.f}; a23[g] = f32x2_t{0.f, 0.f}; b01[g] = f32x2_t{0.f, 0.f}; b23[g] = f32x2_t{0.f, 0.f};
            }
            // every read of the two vectors is issued before the first multiply (one LDS latency per diagonal: left to itself
            // the compiler issues each 16-byte read just ahead of its use and the wave waits eight times)
            float4 xs[L / 4], hv4[L / 4];
#pragma unroll
            for (int k = 0; k < L; k += 4) { xs[k / 4] = *(const float4*)(xt + k); hv4[k / 4] = *(const float4*)(hp + k); }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < L; k += 4) {
                const float4 xv = xs[k / 4], hv = hv4[k / 4];
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    a01[g] = __builtin_elementwise_fma(f32x2_t{wih[g][k], wih[g][k + 1]}, f32x2_t{xv.x, xv.y}, a01[g]);
                    a23[g] = __builtin_elementwise_fma(f32x2_t{wih[g][k + 2], wih[g][k + 3]}, f32x2_t{xv.z, xv.w}, a23[g]);
                    b01[g] = __builtin_elementwise_fma(f32x2_t{whh[g][k], whh[g][k + 1]}, f32x2_t{hv.x, hv.y}, b01[g]);
                    b23[g] = __builtin_elementwise_fma(f32x2_t{whh[g][k + 2], whh[g][k + 3]}, f32x2_t{hv.z, hv.w}, b23[g]);
                }
            }
            float av[2];
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const float pre = ((a01[g][0] + a01[g][1]) + (a23[g][0] + a23[g][1])) +
                                  hscale * ((b01[g][0] + b01[g][1]) + (b23[g][0] + b23[g][1]));
                av[g] = (hf == 1 && g == 0) ? fast_tanh(pre) : fast_sigmoid(pre);      // gate g of the cell is the tanh one
            }
            const long o = (((long)l * S + s) * T + t);
            if (acts) {
#pragma unroll
                for (int g = 0; g < 2; ++g) acts[o * 4 * L + (2 * hf + g) * L + j] = av[g];
            }
            // lanes 0-31 (i, f) take g, o from lanes 32-63
            const float gg = from_upper_half(av[0]), og = from_upper_half(av[1]);
            if (hf == 0) {
                const float ig = av[0], fg = av[1];
                const float hpv = t > 0 ? hp[j] : 0.f;
                c = fmaf(fg, c, ig * gg);        // explicit: the same rounding in every kernel that runs this cell
                const float h = og * fast_tanh(c);
                hb[((l + 1) * T + t) * LS + j] = h;
                if (acts) {
                    cs[o * L + j] = c;
                    hprev[o * L + j] = hpv;
                }
                hs_all[(((long)(l + 1) * S + s) * T + t) * L + j] = h;
                if (q == layers - 1) {
                    const long e = ((long)s * T + t) * L + j;
                    const float y = sigmoidf_((h + nbuf[t * LS + j]) / inv_tau_src);
                    const float zz = p.hard ? (y > 0.5f ? 1.0f : 0.0f) : y;
                    p.y_soft[e] = y;
                    p.hs_d[e] = zz;
                    hbuf[(layers + 1) * T * LS + t * LS + j] = zz;        // decoder stack, slot 0
                }
                if (p.cast_out && q == 2 * layers - 1) {
                    const long oc = ((long)s * T + t) * p.cast_ld + j;
                    if (p.cast_bf16) ((bf16_t*)p.cast_out)[oc] = f32_to_bf16(h); else ((float*)p.cast_out)[oc] = h;
                }
            }
        }
        lds_barrier();
    }
    if (p.kl_parts) {
        float a = 0.f;
        for (int i = threadIdx.x; i < T * L; i += blockDim.x)
            a += kl_elem(hbuf[(layers + 1) * T * LS + i], p.lp, p.l1p, p.keps, p.clamp);
        const float tot = block_sum(a, red);
        if (threadIdx.x == 0) p.kl_parts[s] = tot;
    }
}

// Optional binarise backward in the prologue of the ENCODER stack's BPTT launch (rbvae_binarize_kl_bwd fused):
//   g_top = g_hs + (gz + klw * dKL/dz(z)) * y (1 - y) / tau      (straight-through: the same with hard codes)
struct BinBwd {
    const float *gz, *y, *z, *g_hs;      // gz: gradient of the codes (decoder stack's input gradient); g_hs may be null
    float tau; const float* tau_dev;     // temperature by value, or (when set) from a device float
    float klw, lp, l1p, keps;
    int clamp, on;
};

template <int LMAX, bool EXACT>
__global__ __launch_bounds__(1024) void lstm_bwd_wave_k(const float* __restrict__ wblk, const float* __restrict__ acts,
                                                        const float* __restrict__ cs, const float* __restrict__ g_top,
                                                        float* __restrict__ dG, float* __restrict__ dx, int S, int T,
                                                        int L_, int layers, int G, int nparts, long part_stride,
                                                        void* __restrict__ cast_out, int cast_bf16, int cast_ld,
                                                        float* __restrict__ dx_colsum, const BinBwd bb) {
    RBVAE_RAISE_PRIO();
    const int L = EXACT ? LMAX : L_;
    const float bin_tau = (bb.on && bb.tau_dev) ? bb.tau_dev[0] : bb.tau;     // uniform scalar load, hoisted out of the staging loop
    extern __shared__ float sm[];
    float* gtop = sm;                              // [T][L]
    float* sacts = gtop + T * L;                   // [layers][T][4L]
    float* scs = sacts + layers * T * 4 * L;       // [layers][T][L]
    float* dg = scs + layers * T * L;              // [layers][4L]
    float* part = dg + layers * 4 * L;             // [layers][4][2][L]
    const int l = threadIdx.x / G, j = threadIdx.x - l * G;
    const int s = blockIdx.x;
    const bool row = j < 4 * L;
    const int kcol = j % L, prt = j / L;
    {
        // Stage g_top, the saved gates and cell states of this sequence: gtop | sacts | scs are consecutive in LDS, so
        // one flat index covers all three.  Every load is issued before the first value is used or stored: the saved
        // tensors by unconditional loads on clamped indices (a branch per element made the compiler wait for each load
        // in turn -- 12 memory round trips in front of the first time step), then the operands of g_top (plain, K-split
        // slabs, or the fused binarise backward), then the arithmetic, then the LDS stores.
        constexpr int U = 12;
        const int n0 = T * L, na = T * 4 * L, n1 = n0 + layers * na, ntot = n1 + layers * n0;
        const int nth = blockDim.x;
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int i = n0 + (int)threadIdx.x + u * nth;
            i = i < ntot ? i : ntot - 1;
            const bool in_acts = i < n1;
            const int rel = in_acts ? i - n0 : i - n1;
            const int per = in_acts ? na : n0;
            const int ll = rel / per, r = rel - ll * per;
            const float* src = (in_acts ? acts : cs) + (((long)ll * S + s) * T) * (in_acts ? 4 * L : L) + r;
            v[u] = *src;
        }
        // g_top: one element per thread and round (T * L <= the block size in every shipped configuration)
        for (int i0 = 0; i0 < n0; i0 += nth) {
            const int i = i0 + (int)threadIdx.x;
            const long e = ((long)s * T) * L + (i < n0 ? i : n0 - 1);
            float gt;
            if (bb.on) {
                const float g = bb.gz[e], zv = bb.z[e], yv = bb.y[e];
                const float hsv = bb.g_hs ? bb.g_hs[e] : 0.f;
                float gg = g;
                if (bb.klw != 0.f) gg += bb.klw * kl_elem_grad(zv, bb.lp, bb.l1p, bb.keps, bb.clamp);
                gt = hsv + gg * yv * (1.0f - yv) / bin_tau;
            } else {
                // plain, or K-split slabs (rbvae_skinny_linear_parts) summed in slab order, four loads in flight
                const float* gp = g_top + e;
                gt = gp[0];
                int q = 1;
                for (; q + 3 <= nparts; q += 3) {
                    const float a = gp[q * part_stride], b = gp[(q + 1) * part_stride], c = gp[(q + 2) * part_stride];
                    gt = ((gt + a) + b) + c;
                }
                for (; q < nparts; ++q) gt += gp[q * part_stride];
            }
            if (i < n0) sm[i] = gt;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = n0 + (int)threadIdx.x + u * nth;
            if (i < ntot) sm[i] = v[u];
        }
        // (more than U rounds of the saved tensors: long sequences)
        for (int base = n0 + (int)threadIdx.x + U * nth; base < ntot; base += nth) {
            const bool in_acts = base < n1;
            const int rel = in_acts ? base - n0 : base - n1;
            const int per = in_acts ? na : n0;
            const int ll = rel / per, r = rel - ll * per;
            sm[base] = ((in_acts ? acts : cs) + (((long)ll * S + s) * T) * (in_acts ? 4 * L : L))[r];
        }
    }
    if (cast_out) {
        const int pw = cast_ld - L;
        for (int i = threadIdx.x; i < T * pw; i += blockDim.x) {
            const long o = ((long)s * T + i / pw) * cast_ld + L + i % pw;
            if (cast_bf16) ((bf16_t*)cast_out)[o] = 0; else ((float*)cast_out)[o] = 0.f;
        }
    }
    const float* wl = wblk + l * lstm_layer_floats(L);
    float wic[LMAX], whc[LMAX];
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
    float dc_next = 0.f;
    float dx_sum = 0.f;                  // layer 0, j < L: sum over t of this sequence's input gradient (dx_colsum)
    __syncthreads();
    const int top = layers - 1;
    const int ndiag = T + layers - 1;
    for (int e = 0; e < ndiag; ++e) {
        const int q = e - (top - l);
        const int t = T - 1 - q;
        const bool active = q >= 0 && q < T;
        if (j < L) {
            // layer 0: the input gradient of the step finished at the previous diagonal (time t+1)
            if (l == 0 && q >= 1 && q <= T) {
                const float* p0 = part;
                const float dv = p0[0 * L + j] + p0[2 * L + j] + p0[4 * L + j] + p0[6 * L + j];
                dx[((long)s * T + t + 1) * L + j] = dv;
                dx_sum += dv;
                if (cast_out) {
                    const long o = ((long)s * T + t + 1) * cast_ld + j;
                    if (cast_bf16) ((bf16_t*)cast_out)[o] = f32_to_bf16(dv); else ((float*)cast_out)[o] = dv;
                }
            }
            if (active) {
                float dh;
                if (l == top) dh = gtop[t * L + j];
                else {
                    const float* pu = part + (l + 1) * 8 * L;
                    dh = pu[0 * L + j] + pu[2 * L + j] + pu[4 * L + j] + pu[6 * L + j];
                }
                if (q > 0) {
                    const float* pm = part + l * 8 * L;
                    dh += pm[1 * L + j] + pm[3 * L + j] + pm[5 * L + j] + pm[7 * L + j];
                }
                const float* ap = sacts + (l * T + t) * 4 * L;
                const float ig = ap[j], fg = ap[L + j], gg = ap[2 * L + j], og = ap[3 * L + j];
                const float c = scs[(l * T + t) * L + j];
                const float cprev = t > 0 ? scs[(l * T + t - 1) * L + j] : 0.f;
                const float tc = fast_tanh(c);
                const float dc = dc_next + dh * og * (1.f - tc * tc);
                const float d_o = dh * tc * og * (1.f - og);
                const float d_i = dc * gg * ig * (1.f - ig);
                const float d_f = dc * cprev * fg * (1.f - fg);
                const float d_g = dc * ig * (1.f - gg * gg);
                dc_next = dc * fg;
                float* dl = dg + l * 4 * L;
                dl[j] = d_i; dl[L + j] = d_f; dl[2 * L + j] = d_g; dl[3 * L + j] = d_o;
                float* gp = dG + (((long)l * S + s) * T + t) * 4 * L;
                gp[j] = d_i; gp[L + j] = d_f; gp[2 * L + j] = d_g; gp[3 * L + j] = d_o;
            }
        }
        lds_barrier();
        if (active && row) {
            float ax0 = 0.f, ax1 = 0.f, ah0 = 0.f, ah1 = 0.f;
            const float* dgp = dg + l * 4 * L + prt * L;
#pragma unroll
            for (int jj = 0; jj < LMAX; jj += 2) {
                if (jj + 1 < L) {
                    const float d0 = dgp[jj], d1 = dgp[jj + 1];
                    ax0 = fmaf(wic[jj], d0, ax0); ax1 = fmaf(wic[jj + 1], d1, ax1);
                    ah0 = fmaf(whc[jj], d0, ah0); ah1 = fmaf(whc[jj + 1], d1, ah1);
                } else if (jj < L) {
                    const float d0 = dgp[jj];
                    ax0 = fmaf(wic[jj], d0, ax0); ah0 = fmaf(whc[jj], d0, ah0);
                }
            }
            part[l * 8 * L + (prt * 2 + 0) * L + kcol] = ax0 + ax1;
            part[l * 8 * L + (prt * 2 + 1) * L + kcol] = ah0 + ah1;
        }
        lds_barrier();
    }
    if (l == 0 && j < L) {
        const float dv = part[0 * L + j] + part[2 * L + j] + part[4 * L + j] + part[6 * L + j];
        dx[((long)s * T) * L + j] = dv;
        dx_sum += dv;
        if (cast_out) {
            const long o = ((long)s * T) * cast_ld + j;
            if (cast_bf16) ((bf16_t*)cast_out)[o] = f32_to_bf16(dv); else ((float*)cast_out)[o] = dv;
        }
        // per-sequence column sums of dx: the bias gradient of the Linear that feeds this stack, minus one launch
        if (dx_colsum) dx_colsum[(long)s * L + j] = dx_sum;
    }
}

// Both stacks' BPTT as ONE launch: decoder stack -> binarise backward (+ fused KL) -> encoder stack, the 2 * layers
// layers along one anti-diagonal wavefront (T + 2*layers - 1 dependent steps instead of 2 * (T + layers - 1), one launch
// and one staging prologue fewer).  Virtual layer vl: 0 .. layers-1 = encoder, layers .. 2*layers-1 = decoder.  The seam
// is the encoder's top layer: its dh is the decoder's input gradient (the partial sums of decoder layer 0, summed in
// the order lstm_bwd_wave_k writes dx) through the binarise backward of struct BinBwd -- same expressions, same order.
struct PairBwdArgs {
    const float *wblk_e, *wblk_d, *acts_e, *cs_e, *acts_d, *cs_d;
    const float* g_top; int nparts; long part_stride;       // decoder top: gradient of its output (K-split slabs summed in order)
    float *dG_e, *dG_d, *dx, *dz;                           // dz: the decoder stack's input gradient (optional copy)
    const float* gz_extra;                                  // optional second gradient of the codes
    void* cast_out; int cast_bf16, cast_ld;
    float* dx_colsum;
    BinBwd bb;                                              // gz unused (it never leaves the chip)
    int S, T, L, layers, G;
};

template <int LMAX, bool EXACT>
__global__ __launch_bounds__(1024) void lstm_pair_bwd_k(const PairBwdArgs p) {
    RBVAE_RAISE_PRIO();
    const int L = EXACT ? LMAX : p.L;
    const int T = p.T, S = p.S, layers = p.layers, G = p.G, VL = 2 * layers;
    const BinBwd& bb = p.bb;
    const float bin_tau = bb.tau_dev ? bb.tau_dev[0] : bb.tau;
    extern __shared__ float sm[];
    float* gtop = sm;                              // [T][L]            decoder top
    float* sy = gtop + T * L;                      // [T][L]            seam: y, dKL/dz, g_hs, extra code gradient
    float* sk = sy + T * L;
    float* sh = sk + T * L;
    float* sx = sh + T * L;
    float* sacts = sx + T * L;                     // [VL][T][4L]
    float* scs = sacts + VL * T * 4 * L;           // [VL][T][L]
    float* dg = scs + VL * T * L;                  // [VL][4L]
    float* part = dg + VL * 4 * L;                 // [VL][4][2][L]
    const int vl = threadIdx.x / G, j = threadIdx.x - vl * G;
    const bool dec = vl >= layers;
    const int l = dec ? vl - layers : vl;
    const int s = blockIdx.x;
    const bool row = j < 4 * L;
    const int kcol = j % L, prt = j / L;
    {
        // every global load of the prologue in flight before the first use (see lstm_bwd_wave_k)
        constexpr int U = 12;
        const int n0 = T * L, na = T * 4 * L, n1 = VL * na, ntot = n1 + VL * n0;
        const int nth = blockDim.x;
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int i = (int)threadIdx.x + u * nth;
            i = i < ntot ? i : ntot - 1;
            const bool in_acts = i < n1;
            const int rel = in_acts ? i : i - n1;
            const int per = in_acts ? na : n0;
            const int vll = rel / per, r = rel - vll * per;
            const bool d = vll >= layers;
            const int ll = d ? vll - layers : vll;
            const float* base = in_acts ? (d ? p.acts_d : p.acts_e) : (d ? p.cs_d : p.cs_e);
            v[u] = base[(((long)ll * S + s) * T) * (in_acts ? 4 * L : L) + r];
        }
        for (int i0 = 0; i0 < n0; i0 += nth) {
            const int i = i0 + (int)threadIdx.x;
            const long e = ((long)s * T) * L + (i < n0 ? i : n0 - 1);
            const float* gp = p.g_top + e;
            float gt = gp[0];
            int q = 1;
            for (; q + 3 <= p.nparts; q += 3) {
                const float a = gp[q * p.part_stride], b = gp[(q + 1) * p.part_stride], c = gp[(q + 2) * p.part_stride];
                gt = ((gt + a) + b) + c;
            }
            for (; q < p.nparts; ++q) gt += gp[q * p.part_stride];
            const float yv = bb.y[e], zv = bb.z[e];
            const float hsv = bb.g_hs ? bb.g_hs[e] : 0.f;
            const float xv = p.gz_extra ? p.gz_extra[e] : 0.f;
            if (i < n0) {
                gtop[i] = gt;
                sy[i] = yv;
                sk[i] = bb.klw != 0.f ? kl_elem_grad(zv, bb.lp, bb.l1p, bb.keps, bb.clamp) : 0.f;
                sh[i] = hsv;
                sx[i] = xv;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = (int)threadIdx.x + u * nth;
            if (i < ntot) sacts[i] = v[u];
        }
        for (int i = (int)threadIdx.x + U * nth; i < ntot; i += nth) {
            const bool in_acts = i < n1;
            const int rel = in_acts ? i : i - n1;
            const int per = in_acts ? na : n0;
            const int vll = rel / per, r = rel - vll * per;
            const bool d = vll >= layers;
            const int ll = d ? vll - layers : vll;
            const float* base = in_acts ? (d ? p.acts_d : p.acts_e) : (d ? p.cs_d : p.cs_e);
            sacts[i] = base[(((long)ll * S + s) * T) * (in_acts ? 4 * L : L) + r];
        }
    }
    if (p.cast_out) {
        const int pw = p.cast_ld - L;
        for (int i = threadIdx.x; i < T * pw; i += blockDim.x) {
            const long o = ((long)s * T + i / pw) * p.cast_ld + L + i % pw;
            if (p.cast_bf16) ((bf16_t*)p.cast_out)[o] = 0; else ((float*)p.cast_out)[o] = 0.f;
        }
    }
    const float* wl = (dec ? p.wblk_d : p.wblk_e) + l * lstm_layer_floats(L);
    float wic[LMAX], whc[LMAX];
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
    float dc_next = 0.f;
    float dx_sum = 0.f;
    float* dG = dec ? p.dG_d : p.dG_e;
    __syncthreads();
    const int top = VL - 1;
    const int ndiag = T + VL - 1;
    for (int e = 0; e < ndiag; ++e) {
        const int q = e - (top - vl);
        const int t = T - 1 - q;
        const bool active = q >= 0 && q < T;
        if (j < L) {
            // encoder layer 0: the input gradient of the step finished at the previous diagonal (time t+1)
            if (vl == 0 && q >= 1 && q <= T) {
                const float* p0 = part;
                const float dv = p0[0 * L + j] + p0[2 * L + j] + p0[4 * L + j] + p0[6 * L + j];
                p.dx[((long)s * T + t + 1) * L + j] = dv;
                dx_sum += dv;
                if (p.cast_out) {
                    const long o = ((long)s * T + t + 1) * p.cast_ld + j;
                    if (p.cast_bf16) ((bf16_t*)p.cast_out)[o] = f32_to_bf16(dv); else ((float*)p.cast_out)[o] = dv;
                }
            }
            if (active) {
                float dh;
                if (vl == top) dh = gtop[t * L + j];
                else {
                    const float* pu = part + (vl + 1) * 8 * L;
                    dh = pu[0 * L + j] + pu[2 * L + j] + pu[4 * L + j] + pu[6 * L + j];
                    if (vl == layers - 1) {
                        // the seam: dh so far is the decoder stack's input gradient = the gradient of the codes
                        if (p.dz) p.dz[((long)s * T + t) * L + j] = dh;
                        float gg = p.gz_extra ? dh + sx[t * L + j] : dh;
                        if (bb.klw != 0.f) gg += bb.klw * sk[t * L + j];
                        const float yv = sy[t * L + j];
                        dh = sh[t * L + j] + gg * yv * (1.0f - yv) / bin_tau;
                    }
                }
                if (q > 0) {
                    const float* pm = part + vl * 8 * L;
                    dh += pm[1 * L + j] + pm[3 * L + j] + pm[5 * L + j] + pm[7 * L + j];
                }
                const float* ap = sacts + (vl * T + t) * 4 * L;
                const float ig = ap[j], fg = ap[L + j], gg = ap[2 * L + j], og = ap[3 * L + j];
                const float c = scs[(vl * T + t) * L + j];
                const float cprev = t > 0 ? scs[(vl * T + t - 1) * L + j] : 0.f;
                const float tc = fast_tanh(c);
                const float dc = dc_next + dh * og * (1.f - tc * tc);
                const float d_o = dh * tc * og * (1.f - og);
                const float d_i = dc * gg * ig * (1.f - ig);
                const float d_f = dc * cprev * fg * (1.f - fg);
                const float d_g = dc * ig * (1.f - gg * gg);
                dc_next = dc * fg;
                float* dl = dg + vl * 4 * L;
                dl[j] = d_i; dl[L + j] = d_f; dl[2 * L + j] = d_g; dl[3 * L + j] = d_o;
                float* gp = dG + (((long)l * S + s) * T + t) * 4 * L;
                gp[j] = d_i; gp[L + j] = d_f; gp[2 * L + j] = d_g; gp[3 * L + j] = d_o;
            }
        }
        lds_barrier();
        if (active && row) {
            float ax0 = 0.f, ax1 = 0.f, ah0 = 0.f, ah1 = 0.f;
            const float* dgp = dg + vl * 4 * L + prt * L;
#pragma unroll
            for (int jj = 0; jj < LMAX; jj += 2) {
                if (jj + 1 < L) {
                    const float d0 = dgp[jj], d1 = dgp[jj + 1];
                    ax0 = fmaf(wic[jj], d0, ax0); ax1 = fmaf(wic[jj + 1], d1, ax1);
                    ah0 = fmaf(whc[jj], d0, ah0); ah1 = fmaf(whc[jj + 1], d1, ah1);
                } else if (jj < L) {
                    const float d0 = dgp[jj];
                    ax0 = fmaf(wic[jj], d0, ax0); ah0 = fmaf(whc[jj], d0, ah0);
                }
            }
            part[vl * 8 * L + (prt * 2 + 0) * L + kcol] = ax0 + ax1;
            part[vl * 8 * L + (prt * 2 + 1) * L + kcol] = ah0 + ah1;
        }
        lds_barrier();
    }
    if (vl == 0 && j < L) {
        const float dv = part[0 * L + j] + part[2 * L + j] + part[4 * L + j] + part[6 * L + j];
        p.dx[((long)s * T) * L + j] = dv;
        dx_sum += dv;
        if (p.cast_out) {
            const long o = ((long)s * T) * p.cast_ld + j;
            if (p.cast_bf16) ((bf16_t*)p.cast_out)[o] = f32_to_bf16(dv); else ((float*)p.cast_out)[o] = dv;
        }
        if (p.dx_colsum) p.dx_colsum[(long)s * L + j] = dx_sum;
    }
}

// The same BPTT wavefront with one WAVE per virtual layer (L == 32), the backward partner of lstm_pair_fwd_unit_k: lanes
// 0-31 (hidden unit j) do the pointwise gate gradients and write them to LDS; then lane (column kcol, half hf) multiplies
// the two gate blocks 2*hf, 2*hf+1 of both transposed weight matrices with them (128 weights in registers, the same two
// fused-multiply-add chains per block as lstm_pair_bwd_k), lanes 0-31 fetch the other half's block sums (four cross-lane
// moves) and add the four blocks in the gate-row kernel's order.  The recurrent part dh_{t-1} never leaves its lane; only
// the input gradient goes through LDS to the wave of the layer below (double-buffered by diagonal parity), and the gate
// gradients are read back by the wave that wrote them (LDS operations of one wave complete in issue order): ONE workgroup
// barrier of 2 * layers waves per diagonal instead of two of 16 waves.  Identical results.
__global__ __launch_bounds__(512) void lstm_pair_bwd_unit_k(const PairBwdArgs p) {
    RBVAE_RAISE_PRIO();
    constexpr int L = 32;
    const int T = p.T, S = p.S, layers = p.layers, VL = 2 * layers;
    const BinBwd& bb = p.bb;
    const float bin_tau = bb.tau_dev ? bb.tau_dev[0] : bb.tau;
    extern __shared__ float sm[];
    float* gtop = sm;                              // [T][L]            decoder top
    float* sy = gtop + T * L;                      // [T][L]            seam: y, dKL/dz, g_hs, extra code gradient
    float* sk = sy + T * L;
    float* sh = sk + T * L;
    float* sx = sh + T * L;
    float* sacts = sx + T * L;                     // [VL][T][4L]
    float* scs = sacts + VL * T * 4 * L;           // [VL][T][L]
    float* dg = scs + VL * T * L;                  // [VL][4L]
    float* partx = dg + VL * 4 * L;                // [2][VL][L]: input gradients on their way down, by diagonal parity
    const int vl = threadIdx.x >> 6, j = threadIdx.x & 31, hf = (threadIdx.x >> 5) & 1;
    const bool dec = vl >= layers;
    const int l = dec ? vl - layers : vl;
    const int s = blockIdx.x;
    {
        // every global load of the prologue in flight before the first use (see lstm_bwd_wave_k)
        constexpr int U = 12;
        const int n0 = T * L, na = T * 4 * L, n1 = VL * na, ntot = n1 + VL * n0;
        const int nth = blockDim.x;
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int i = (int)threadIdx.x + u * nth;
            i = i < ntot ? i : ntot - 1;
            const bool in_acts = i < n1;
            const int rel = in_acts ? i : i - n1;
            const int per = in_acts ? na : n0;
            const int vll = rel / per, r = rel - vll * per;
            const bool d = vll >= layers;
            const int ll = d ? vll - layers : vll;
            const float* base = in_acts ? (d ? p.acts_d : p.acts_e) : (d ? p.cs_d : p.cs_e);
            v[u] = base[(((long)ll * S + s) * T) * (in_acts ? 4 * L : L) + r];
        }
        for (int i0 = 0; i0 < n0; i0 += nth) {
            const int i = i0 + (int)threadIdx.x;
            const long e = ((long)s * T) * L + (i < n0 ? i : n0 - 1);
            const float* gp = p.g_top + e;
            float gt = gp[0];
            int q = 1;
            for (; q + 3 <= p.nparts; q += 3) {
                const float a = gp[q * p.part_stride], b = gp[(q + 1) * p.part_stride], c = gp[(q + 2) * p.part_stride];
                gt = ((gt + a) + b) + c;
            }
            for (; q < p.nparts; ++q) gt += gp[q * p.part_stride];
            const float yv = bb.y[e], zv = bb.z[e];
            const float hsv = bb.g_hs ? bb.g_hs[e] : 0.f;
            const float xv = p.gz_extra ? p.gz_extra[e] : 0.f;
            if (i < n0) {
                gtop[i] = gt;
                sy[i] = yv;
                sk[i] = bb.klw != 0.f ? kl_elem_grad(zv, bb.lp, bb.l1p, bb.keps, bb.clamp) : 0.f;
                sh[i] = hsv;
                sx[i] = xv;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = (int)threadIdx.x + u * nth;
            if (i < ntot) sacts[i] = v[u];
        }
        for (int i = (int)threadIdx.x + U * nth; i < ntot; i += nth) {
            const bool in_acts = i < n1;
            const int rel = in_acts ? i : i - n1;
            const int per = in_acts ? na : n0;
            const int vll = rel / per, r = rel - vll * per;
            const bool d = vll >= layers;
            const int ll = d ? vll - layers : vll;
            const float* base = in_acts ? (d ? p.acts_d : p.acts_e) : (d ? p.cs_d : p.cs_e);
            sacts[i] = base[(((long)ll * S + s) * T) * (in_acts ? 4 * L : L) + r];
        }
    }
    if (p.cast_out) {
        const int pw = p.cast_ld - L;
        for (int i = threadIdx.x; i < T * pw; i += blockDim.x) {
            const long o = ((long)s * T + i / pw) * p.cast_ld + L + i % pw;
            if (p.cast_bf16) ((bf16_t*)p.cast_out)[o] = 0; else ((float*)p.cast_out)[o] = 0.f;
        }
    }
    const float* wl = (dec ? p.wblk_d : p.wblk_e) + l * lstm_layer_floats(L);
    float wic[2][L], whc[2][L];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const float* pi = wl + (2 * hf + g) * L * L + j;
        const float* ph = pi + 4 * L * L;
#pragma unroll
        for (int jj = 0; jj < L; ++jj) { wic[g][jj] = pi[jj * L]; whc[g][jj] = ph[jj * L]; }
    }
    float dc_next = 0.f, dx_sum = 0.f, dh_rec = 0.f;
    float* dG = dec ? p.dG_d : p.dG_e;
    __syncthreads();
    const int top = VL - 1;
    const int ndiag = T + VL - 1;
    for (int e = 0; e < ndiag; ++e) {
        const int q = e - (top - vl);
        const int t = T - 1 - q;
        if (q >= 0 && q < T) {                             // wave-uniform
            float* dl = dg + vl * 4 * L;
            if (hf == 0) {
                float dh;
                if (vl == top) dh = gtop[t * L + j];
                else {
                    dh = partx[(((e - 1) & 1) * VL + vl + 1) * L + j];
                    if (vl == layers - 1) {
                        // the seam: dh so far is the decoder stack's input gradient = the gradient of the codes
                        if (p.dz) p.dz[((long)s * T + t) * L + j] = dh;
                        float gg = p.gz_extra ? dh + sx[t * L + j] : dh;
                        if (bb.klw != 0.f) gg += bb.klw * sk[t * L + j];
                        const float yv = sy[t * L + j];
                        dh = sh[t * L + j] + gg * yv * (1.0f - yv) / bin_tau;
                    }
                }
                if (q > 0) dh += dh_rec;
                const float* ap = sacts + (vl * T + t) * 4 * L;
                const float ig = ap[j], fg = ap[L + j], gg = ap[2 * L + j], og = ap[3 * L + j];
                const float c = scs[(vl * T + t) * L + j];
                const float cprev = t > 0 ? scs[(vl * T + t - 1) * L + j] : 0.f;
                const float tc = fast_tanh(c);
                const float dc = dc_next + dh * og * (1.f - tc * tc);
                const float d_o = dh * tc * og * (1.f - og);
                const float d_i = dc * gg * ig * (1.f - ig);
                const float d_f = dc * cprev * fg * (1.f - fg);
                const float d_g = dc * ig * (1.f - gg * gg);
                dc_next = dc * fg;
                dl[j] = d_i; dl[L + j] = d_f; dl[2 * L + j] = d_g; dl[3 * L + j] = d_o;
                float* gp = dG + (((long)l * S + s) * T + t) * 4 * L;
                gp[j] = d_i; gp[L + j] = d_f; gp[2 * L + j] = d_g; gp[3 * L + j] = d_o;
            }
            // the wave's own gate gradients back from LDS (same wave: in issue order behind the writes above)
            const float* dgp = dl + 2 * hf * L;
            float4 dv[2 * L / 4];
#pragma unroll
            for (int i = 0; i < 2 * L / 4; ++i) dv[i] = *(const float4*)(dgp + 4 * i);
            __builtin_amdgcn_sched_barrier(0);
            f32x2_t axp[2], ahp[2];
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                axp[g] = f32x2_t{0.f, 0.f}; ahp[g] = f32x2_t{0.f, 0.f};
#pragma unroll
                for (int jj = 0; jj < L; jj += 4) {
                    const float4 d4 = dv[(g * L + jj) / 4];
                    axp[g] = __builtin_elementwise_fma(f32x2_t{wic[g][jj], wic[g][jj + 1]}, f32x2_t{d4.x, d4.y}, axp[g]);
                    ahp[g] = __builtin_elementwise_fma(f32x2_t{whc[g][jj], whc[g][jj + 1]}, f32x2_t{d4.x, d4.y}, ahp[g]);
                    axp[g] = __builtin_elementwise_fma(f32x2_t{wic[g][jj + 2], wic[g][jj + 3]}, f32x2_t{d4.z, d4.w}, axp[g]);
                    ahp[g] = __builtin_elementwise_fma(f32x2_t{whc[g][jj + 2], whc[g][jj + 3]}, f32x2_t{d4.z, d4.w}, ahp[g]);
                }
            }
            const float px0 = axp[0][0] + axp[0][1], px1 = axp[1][0] + axp[1][1];
            const float ph0 = ahp[0][0] + ahp[0][1], ph1 = ahp[1][0] + ahp[1][1];
            // lanes 0-31 (gate blocks 0, 1) take the sums of blocks 2, 3 from lanes 32-63 and add in block order
            const float px2 = from_upper_half(px0), px3 = from_upper_half(px1);
            const float ph2 = from_upper_half(ph0), ph3 = from_upper_half(ph1);
            if (hf == 0) {
                const float dxv = px0 + px1 + px2 + px3;
                dh_rec = ph0 + ph1 + ph2 + ph3;
                if (vl > 0) partx[((e & 1) * VL + vl) * L + j] = dxv;
                else {
                    // encoder layer 0: the stack's input gradient at time t
                    p.dx[((long)s * T + t) * L + j] = dxv;
                    dx_sum += dxv;
                    if (p.cast_out) {
                        const long o = ((long)s * T + t) * p.cast_ld + j;
                        if (p.cast_bf16) ((bf16_t*)p.cast_out)[o] = f32_to_bf16(dxv); else ((float*)p.cast_out)[o] = dxv;
                    }
                }
            }
        }
        lds_barrier();
    }
    // per-sequence column sums of dx: the bias gradient of the Linear that feeds the encoder stack
    if (vl == 0 && hf == 0 && p.dx_colsum) p.dx_colsum[(long)s * L + j] = dx_sum;
}

// Weight gradients, LDS-tiled: block = (8 gate rows, ih|hh, layer); thread (jj, kq) owns gate row jj and
// the columns kq, kq+32, ... (column L = the bias).  Rows of dG / X stream through LDS 128 at a time.
constexpr int LW_ROWS = 128;
constexpr int LW_JT = 8;
template <int LW_KMAX>          // ceil((L + 1) / 32): 2 for L <= 32, 3 for L <= 64, 5 for L <= 128
__global__ __launch_bounds__(256) void lstm_wgrad_tiled_k(const float* __restrict__ dG, const float* __restrict__ hs_all,
                                                          const float* __restrict__ hprev, float* __restrict__ gblk,
                                                          const float* __restrict__ dG2, const float* __restrict__ hs_all2,
                                                          const float* __restrict__ hprev2, float* __restrict__ gblk2,
                                                          int S, int T, int L, int layers, int accumulate) {
    extern __shared__ float sm[];
    float* sg = sm;                         // [LW_ROWS][LW_JT gate rows]
    float* sx = sg + LW_ROWS * LW_JT;       // [LW_ROWS][L + 1]   (column L holds 1.0: the bias column)
    // grid.z = layers of the first stack, then (optionally) layers of a second one: encoder and decoder
    // stacks share one launch
    int l = blockIdx.z;
    if (l >= layers) { l -= layers; dG = dG2; hs_all = hs_all2; hprev = hprev2; gblk = gblk2; }
    const int hh = blockIdx.y, j0 = blockIdx.x * LW_JT;
    const int jj = threadIdx.x & (LW_JT - 1), kq = threadIdx.x / LW_JT;     // kq in [0, 32)
    const int R = S * T, LP = L + 1;
    const float* g = dG + (long)l * R * 4 * L;
    const float* x = hh ? hprev + (long)l * R * L : hs_all + (long)l * R * L;
    float acc[LW_KMAX];
#pragma unroll
    for (int i = 0; i < LW_KMAX; ++i) acc[i] = 0.f;
    for (int r0 = 0; r0 < R; r0 += LW_ROWS) {
        const int nr = min(LW_ROWS, R - r0);
        // clamped indices: every load is unconditional, so a thread's loads are all in flight together
#pragma unroll
        for (int it = 0; it < LW_ROWS * LW_JT / 256; ++it) {
            const int i = threadIdx.x + it * 256;
            const int r = i / LW_JT, c = i & (LW_JT - 1);
            const float v = g[(long)(r0 + min(r, nr - 1)) * 4 * L + min(j0 + c, 4 * L - 1)];
            sg[i] = (r < nr && j0 + c < 4 * L) ? v : 0.f;
        }
        for (int i0 = 0; i0 < LW_ROWS * L; i0 += 256 * 8) {
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int i = i0 + threadIdx.x + it * 256;
                const float v = x[(long)r0 * L + min(i, nr * L - 1)];
                if (i < LW_ROWS * L) { const int r = i / L; sx[r * LP + (i - r * L)] = i < nr * L ? v : 0.f; }
            }
        }
        if (threadIdx.x < LW_ROWS) sx[threadIdx.x * LP + L] = threadIdx.x < nr ? 1.f : 0.f;
        __syncthreads();
#pragma unroll 8
        for (int r = 0; r < LW_ROWS; ++r) {
            const float gv = sg[r * LW_JT + jj];
            const float* xr = sx + r * LP;
#pragma unroll
            for (int i = 0; i < LW_KMAX; ++i) {
                const int k = min(kq + 32 * i, L);           // clamped: column L is the ones column
                acc[i] = fmaf(gv, xr[k], acc[i]);
            }
        }
        __syncthreads();
    }
    const int jrow = j0 + jj;
    if (jrow >= 4 * L) return;
    float* out = gblk + l * (8l * L * L + 8l * L);
#pragma unroll
    for (int i = 0; i < LW_KMAX; ++i) {
        const int k = kq + 32 * i;
        if (k > L) continue;
        float* dst = k < L ? out + (hh ? 4l * L * L : 0) + (long)jrow * L + k : out + 8l * L * L + (hh ? 4 * L : 0) + jrow;
        *dst = accumulate ? *dst + acc[i] : acc[i];
    }
}

// Weight gradients on the f32 matrix cores (v_mfma_f32_16x16x4_f32: an exact f32 fma chain).  One workgroup per
// 16 x 16 tile of one (stack, layer, ih|hh) gradient [4L gate rows][L inputs + bias column]; its four waves take a
// quarter of the S*T rows each, operands straight from global memory in MFMA layout (a lane's loads are all
// independent: one batch per 16 rows), partial tiles meet in LDS in wave order.  The LDS-tiled kernel above spends
// ~20 us on this at the bench shape (256 rows): it is all latency, and this form has a tenth of the dependent steps.
__global__ __launch_bounds__(256) void lstm_wgrad_mfma_k(const float* __restrict__ dG, const float* __restrict__ hs_all,
                                                         const float* __restrict__ hprev, float* __restrict__ gblk,
                                                         const float* __restrict__ dG2, const float* __restrict__ hs_all2,
                                                         const float* __restrict__ hprev2, float* __restrict__ gblk2,
                                                         int S, int T, int L, int layers, int accumulate) {
    typedef __attribute__((ext_vector_type(4))) float f32x4_t;
    RBVAE_RAISE_PRIO();
    __shared__ float part[4][256];
    int l = blockIdx.z;
    if (l >= layers) { l -= layers; dG = dG2; hs_all = hs_all2; hprev = hprev2; gblk = gblk2; }
    const int hh = blockIdx.y;
    const int ktiles = (L + 1 + 15) / 16;
    const int j0 = (blockIdx.x / ktiles) * 16, k0 = (blockIdx.x % ktiles) * 16;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int i = lane & 15, g = lane >> 4;
    const int R = S * T;
    const float* gp = dG + (long)l * R * 4 * L + min(j0 + i, 4 * L - 1);
    const float* xp = (hh ? hprev + (long)l * R * L : hs_all + (long)l * R * L) + min(k0 + i, L - 1);
    const bool jv = j0 + i < 4 * L;
    const int kcol = k0 + i;                       // < L: an input column, == L: the bias (ones) column, > L: padding
    const int rq = (R + 3) / 4;                    // rows of this wave's quarter, in steps of 4
    const int rbeg = w * rq, rend = min(R, rbeg + rq);
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    constexpr int UB = 16;
    for (int r0 = rbeg; r0 < rend; r0 += 4 * UB) {
        float a[UB], b[UB];
        // All 2 * UB loads are issued before the first value is looked at (the two asm statements below take the raw
        // values of eight steps each): left to itself the compiler keeps every pair of loads next to the select that
        // consumes it and waits there -- 16 dependent memory round trips per wave, which WAS this kernel's run time.
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int rc = min(r0 + 4 * u + g, R - 1);
            a[u] = gp[(long)rc * 4 * L];
            b[u] = xp[(long)rc * L];
        }
        static_assert(UB == 16, "operand lists below");
        asm volatile("" : "+v"(a[0]), "+v"(b[0]), "+v"(a[1]), "+v"(b[1]), "+v"(a[2]), "+v"(b[2]), "+v"(a[3]), "+v"(b[3]),
                          "+v"(a[4]), "+v"(b[4]), "+v"(a[5]), "+v"(b[5]), "+v"(a[6]), "+v"(b[6]), "+v"(a[7]), "+v"(b[7]),
                          "+v"(a[8]), "+v"(b[8]), "+v"(a[9]), "+v"(b[9]), "+v"(a[10]), "+v"(b[10]), "+v"(a[11]), "+v"(b[11]),
                          "+v"(a[12]), "+v"(b[12]), "+v"(a[13]), "+v"(b[13]), "+v"(a[14]), "+v"(b[14]));
        asm volatile("" : "+v"(a[15]), "+v"(b[15]));
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const bool rv = r0 + 4 * u + g < rend;
            a[u] = (rv && jv) ? a[u] : 0.f;
            b[u] = rv ? (kcol < L ? b[u] : (kcol == L ? 1.f : 0.f)) : 0.f;
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[u], acc, 0, 0, 0);
    }
    // D[gate row 4g + r][column i]
#pragma unroll
    for (int r = 0; r < 4; ++r) part[w][(4 * g + r) * 16 + i] = acc[r];
    __syncthreads();
    const int rr = threadIdx.x >> 4, cc = threadIdx.x & 15;
    const int jrow = j0 + rr, k = k0 + cc;
    if (jrow < 4 * L && k <= L) {
        const float v = ((part[0][threadIdx.x] + part[1][threadIdx.x]) + part[2][threadIdx.x]) + part[3][threadIdx.x];
        float* out = gblk + l * (8l * L * L + 8l * L);
        float* dst = k < L ? out + (hh ? 4l * L * L : 0) + (long)jrow * L + k : out + 8l * L * L + (hh ? 4 * L : 0) + jrow;
        *dst = accumulate ? *dst + v : v;
    }
}

}  // namespace rbvae

using namespace rbvae;

extern "C" {

static int lstm_fwd_impl(const float* wblk, const float* wT, float* hs_all, float* hprev, float* acts, float* cs, int S,
                         int T, int L, int layers, const float* in_parts, int nparts, long part_stride, void* cast_out,
                         int cast_dtype, int cast_ld, void* stream) {
    RBVAE_CHECK_ARG(wblk && hs_all && S > 0 && T > 0 && L > 0 && layers > 0, "lstm_fwd: bad arguments");
    RBVAE_CHECK_ARG(!cast_out || ((cast_dtype == RBVAE_F32 || cast_dtype == RBVAE_BF16) && cast_ld >= L),
                    "lstm_fwd: cast output dtype %d ld %d", cast_dtype, cast_ld);
    const int cast_bf16 = cast_dtype == RBVAE_BF16;
    RBVAE_CHECK_ARG(L <= 128, "lstm_fwd: latent_dim %d > 128 is not supported", L);
    RBVAE_CHECK_ARG((acts == nullptr) == (cs == nullptr) && (acts == nullptr) == (hprev == nullptr),
                    "lstm_fwd: hprev/acts/cs must be given together");
    const int threads = ((4 * L + 63) / 64) * 64;
    const size_t lds = (size_t)(2 * T * L + 5 * L) * sizeof(float);
    RBVAE_CHECK_ARG(lds <= 64 * 1024, "lstm_fwd: T*L=%d too large", T * L);
    hipStream_t st = (hipStream_t)stream;
    // wavefront kernel: one thread group per layer (weights in registers, L <= 32)
    const size_t wlds = (size_t)((layers + 1) * T * L + layers * 4 * L) * sizeof(float);
    if (L <= 32 && layers * threads <= 1024 && wlds <= 64 * 1024) {
        if (L == 32)
            hipLaunchKernelGGL((lstm_fwd_wave_k<32, true, true>), dim3(S), dim3(layers * threads), wlds, st, wblk, wT,
                               hs_all, hprev, acts, cs, S, T, L, layers, threads, in_parts, nparts, part_stride, cast_out,
                               cast_bf16, cast_ld);
        else if (L % 4 == 0)
            hipLaunchKernelGGL((lstm_fwd_wave_k<32, true, false>), dim3(S), dim3(layers * threads), wlds, st, wblk, wT,
                               hs_all, hprev, acts, cs, S, T, L, layers, threads, in_parts, nparts, part_stride, cast_out,
                               cast_bf16, cast_ld);
        else
            hipLaunchKernelGGL((lstm_fwd_wave_k<32, false, false>), dim3(S), dim3(layers * threads), wlds, st, wblk, wT,
                               hs_all, hprev, acts, cs, S, T, L, layers, threads, in_parts, nparts, part_stride, cast_out,
                               cast_bf16, cast_ld);
        RBVAE_CHECK_LAUNCH("lstm_fwd_wave");
        return RBVAE_OK;
    }
    RBVAE_CHECK_ARG(!in_parts && !cast_out, "lstm_fwd_ex: only the wavefront kernel (L <= 32, layers * roundup64(4L) <= 1024) sums input slabs / writes a cast copy");
    if (L > 32) {
        // two lanes per gate row, the input half batched over time (lstm_fwd_big_k)
        const int RP = ((4 * L + 63) / 64) * 64;
        const size_t blds = (size_t)((2 * T + 1) * BIG_VS + RP + T * RP) * sizeof(float);
        RBVAE_CHECK_ARG(blds <= 64 * 1024, "lstm_fwd: T=%d too long for L=%d", T, L);
        const int nch = (L + 3) / 4;                              // 16-byte chunks of a weight row
#define FWD_BIG(N) hipLaunchKernelGGL(lstm_fwd_big_k<N>, dim3(S), dim3(RP), blds, st, wblk, wT, hs_all, hprev, acts, cs, \
                                      S, T, L, layers, RP)
        if (nch <= 10) FWD_BIG(10); else if (nch <= 13) FWD_BIG(13); else if (nch <= 16) FWD_BIG(16);
        else if (nch <= 19) FWD_BIG(19); else if (nch <= 22) FWD_BIG(22); else if (nch <= 25) FWD_BIG(25);
        else if (nch <= 28) FWD_BIG(28); else FWD_BIG(32);
#undef FWD_BIG
        RBVAE_CHECK_LAUNCH("lstm_fwd_big");
        return RBVAE_OK;
    }
    if (L <= 32)
        hipLaunchKernelGGL(lstm_fwd_k<32>, dim3(S), dim3(threads), lds, st, wblk, hs_all, hprev, acts, cs, S, T, L, layers);
    else if (L <= 64)
        hipLaunchKernelGGL(lstm_fwd_k<64>, dim3(S), dim3(threads), lds, st, wblk, hs_all, hprev, acts, cs, S, T, L, layers);
    else
        hipLaunchKernelGGL(lstm_fwd_k<0>, dim3(S), dim3(threads), lds, st, wblk, hs_all, hprev, acts, cs, S, T, L, layers);
    RBVAE_CHECK_LAUNCH("lstm_fwd");
    return RBVAE_OK;
}

int rbvae_lstm_fwd(const float* wblk, const float* wT, float* hs_all, float* hprev, float* acts, float* cs, int S,
                   int T, int L, int layers, void* stream) {
    return lstm_fwd_impl(wblk, wT, hs_all, hprev, acts, cs, S, T, L, layers, nullptr, 1, 0, nullptr, 0, 0, stream);
}

int rbvae_lstm_fwd_ex(const float* wblk, const float* wT, float* hs_all, float* hprev, float* acts, float* cs, int S,
                      int T, int L, int layers, const float* in_parts, int nparts, long part_stride, void* cast_out,
                      int cast_dtype, int cast_ld, void* stream) {
    RBVAE_CHECK_ARG(!in_parts || (nparts >= 1 && part_stride >= (long)S * T * L), "lstm_fwd_ex: bad slabs");
    return lstm_fwd_impl(wblk, wT, hs_all, hprev, acts, cs, S, T, L, layers, in_parts, nparts, part_stride, cast_out,
                         cast_dtype, cast_ld, stream);
}

int rbvae_lstm_pair_fwd_ok(int T, int L, int layers) {
    const int threads = ((4 * L + 63) / 64) * 64;
    const int LS = (L + 3) & ~3;
    const size_t lds = (size_t)(2 * (layers + 1) * T * LS + 2 * layers * 4 * L + T * LS + 16) * sizeof(float);
    return L <= 32 && 2 * layers * threads <= 1024 && lds <= 64 * 1024;
}

int rbvae_lstm_pair_fwd(const float* wblk_enc, const float* wT_enc, const float* wblk_dec, const float* wT_dec,
                        float* hs_enc, float* hprev_enc, float* acts_enc, float* cs_enc, float* hs_dec,
                        float* hprev_dec, float* acts_dec, float* cs_dec, const float* in_parts, int nparts,
                        long part_stride, const float* U, float* y_soft, float* kl_parts, float tau, const float* tau_dev,
                        float noise_ratio, float noise_eps, int hard, float kl_p, float kl_eps, int kl_clamp, unsigned long long seed,
                        const unsigned long long* seed_dev, void* cast_out, int cast_dtype, int cast_ld, int S, int T,
                        int L, int layers, void* stream) {
    RBVAE_CHECK_ARG(wblk_enc && wblk_dec && hs_enc && hs_dec && y_soft && S > 0 && T > 0 && L > 0 && layers > 0,
                    "lstm_pair_fwd: bad arguments");
    RBVAE_CHECK_ARG(rbvae_lstm_pair_fwd_ok(T, L, layers), "lstm_pair_fwd: T=%d L=%d layers=%d outside the fused kernel's range "
                    "(rbvae_lstm_pair_fwd_ok)", T, L, layers);
    RBVAE_CHECK_ARG((acts_enc == nullptr) == (cs_enc == nullptr) && (acts_enc == nullptr) == (hprev_enc == nullptr) &&
                    (acts_dec == nullptr) == (acts_enc == nullptr) && (cs_dec == nullptr) == (acts_enc == nullptr) &&
                    (hprev_dec == nullptr) == (acts_enc == nullptr), "lstm_pair_fwd: saved-state buffers must be given together");
    RBVAE_CHECK_ARG((tau_dev || tau > 0.f) && (!kl_parts || (kl_p > 0.f && kl_p < 1.f)), "lstm_pair_fwd: tau=%g kl_p=%g", tau, kl_p);
    RBVAE_CHECK_ARG(!in_parts || (nparts >= 1 && part_stride >= (long)S * T * L), "lstm_pair_fwd: bad slabs");
    RBVAE_CHECK_ARG(!cast_out || ((cast_dtype == RBVAE_F32 || cast_dtype == RBVAE_BF16) && cast_ld >= L),
                    "lstm_pair_fwd: cast output dtype %d ld %d", cast_dtype, cast_ld);
    PairArgs a;
    a.wblk_e = wblk_enc; a.wT_e = wT_enc; a.wblk_d = wblk_dec; a.wT_d = wT_dec;
    a.hs_e = hs_enc; a.hp_e = hprev_enc; a.acts_e = acts_enc; a.cs_e = cs_enc;
    a.hs_d = hs_dec; a.hp_d = hprev_dec; a.acts_d = acts_dec; a.cs_d = cs_dec;
    a.in_parts = in_parts; a.nparts = nparts; a.part_stride = part_stride;
    a.U = U; a.y_soft = y_soft; a.kl_parts = kl_parts;
    a.tau = tau; a.tau_dev = tau_dev; a.ratio = noise_ratio; a.neps = noise_eps; a.lp = kl_parts ? logf(kl_p) : 0.f;
    a.l1p = kl_parts ? logf(1.0f - kl_p) : 0.f; a.keps = kl_eps; a.hard = hard; a.clamp = kl_clamp;
    a.seed = seed; a.seed_dev = seed_dev;
    a.cast_out = cast_out; a.cast_bf16 = cast_dtype == RBVAE_BF16; a.cast_ld = cast_ld;
    const int threads = ((4 * L + 63) / 64) * 64;
    a.S = S; a.T = T; a.L = L; a.layers = layers; a.G = threads;
    const int LS = (L + 3) & ~3;
    const size_t lds = (size_t)(2 * (layers + 1) * T * LS + 2 * layers * 4 * L + T * LS + 16) * sizeof(float);
    if (L == 32 && 2 * layers * 64 <= 512 && lstm_unit_threads)
        hipLaunchKernelGGL(lstm_pair_fwd_unit_k, dim3(S), dim3(2 * layers * 64), lds, (hipStream_t)stream, a);
    else if (L == 32)
        hipLaunchKernelGGL((lstm_pair_fwd_k<32, true>), dim3(S), dim3(2 * layers * threads), lds, (hipStream_t)stream, a);
    else if (LS == 28)      // latent_dim 25 (the reference's most common) .. 28
        hipLaunchKernelGGL((lstm_pair_fwd_k<32, false, 28>), dim3(S), dim3(2 * layers * threads), lds, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL((lstm_pair_fwd_k<32, false>), dim3(S), dim3(2 * layers * threads), lds, (hipStream_t)stream, a);
    RBVAE_CHECK_LAUNCH("lstm_pair_fwd");
    return RBVAE_OK;
}

static int lstm_bwd_impl(const float* wblk, const float* wT, const float* acts, const float* cs, const float* g_top,
                         float* dG, float* dx, int S, int T, int L, int layers, int nparts, long part_stride, void* cast_out, int cast_dtype,
                         int cast_ld, float* dx_colsum, const BinBwd& bb, void* stream) {
    RBVAE_CHECK_ARG(wblk && acts && cs && g_top && dG && dx && S > 0 && T > 0 && L > 0 && layers > 0,
                    "lstm_bwd: bad arguments");
    RBVAE_CHECK_ARG(!cast_out || ((cast_dtype == RBVAE_F32 || cast_dtype == RBVAE_BF16) && cast_ld >= L),
                    "lstm_bwd: cast output dtype %d ld %d", cast_dtype, cast_ld);
    const int cast_bf16 = cast_dtype == RBVAE_BF16;
    RBVAE_CHECK_ARG(L <= 128, "lstm_bwd: latent_dim %d > 128 is not supported", L);
    const int threads = ((4 * L + 63) / 64) * 64;
    const size_t lds = (size_t)(2 * T * L + 13 * L) * sizeof(float);
    RBVAE_CHECK_ARG(lds <= 64 * 1024, "lstm_bwd: T*L=%d too large", T * L);
    hipStream_t st = (hipStream_t)stream;
    const size_t wlds = (size_t)(T * L + layers * T * 5 * L + layers * 12 * L) * sizeof(float);
    if (L <= 32 && layers * threads <= 1024 && wlds <= 64 * 1024) {
        if (L == 32)
            hipLaunchKernelGGL((lstm_bwd_wave_k<32, true>), dim3(S), dim3(layers * threads), wlds, st, wblk, acts, cs,
                               g_top, dG, dx, S, T, L, layers, threads, nparts, part_stride, cast_out, cast_bf16, cast_ld,
                               dx_colsum, bb);
        else
            hipLaunchKernelGGL((lstm_bwd_wave_k<32, false>), dim3(S), dim3(layers * threads), wlds, st, wblk, acts, cs,
                               g_top, dG, dx, S, T, L, layers, threads, nparts, part_stride, cast_out, cast_bf16, cast_ld,
                               dx_colsum, bb);
        RBVAE_CHECK_LAUNCH("lstm_bwd_wave");
        return RBVAE_OK;
    }
    RBVAE_CHECK_ARG(nparts == 1 && !cast_out && !dx_colsum && !bb.on, "lstm_bwd_ex: only the wavefront kernel (L <= 32, layers * roundup64(4L) <= 1024) sums gradient slabs / writes a cast copy");
    if (L > 32) {
        const int KP = ((L + 15) / 16) * 16;
        const size_t blds = (size_t)((2 * T + 1) * BIG_VS + T * 4 * BIG_SEG) * sizeof(float);
        RBVAE_CHECK_ARG(blds <= 64 * 1024, "lstm_bwd: T=%d too long for L=%d", T, L);
        const int nch = (L + 3) / 4;                              // 16-byte chunks of a lane's gate rows
#define BWD_BIG(N) hipLaunchKernelGGL(lstm_bwd_big_k<N>, dim3(S), dim3(4 * KP), blds, st, wblk, wT, acts, cs, g_top, dG, \
                                      dx, S, T, L, layers)
        if (nch <= 10) BWD_BIG(10); else if (nch <= 13) BWD_BIG(13); else if (nch <= 16) BWD_BIG(16);
        else if (nch <= 19) BWD_BIG(19); else if (nch <= 22) BWD_BIG(22); else if (nch <= 25) BWD_BIG(25);
        else if (nch <= 28) BWD_BIG(28); else BWD_BIG(32);
#undef BWD_BIG
        RBVAE_CHECK_LAUNCH("lstm_bwd_big");
        return RBVAE_OK;
    }
    if (L <= 32)
        hipLaunchKernelGGL(lstm_bwd_k<32>, dim3(S), dim3(threads), lds, st, wblk, acts, cs, g_top, dG, dx, S, T, L, layers);
    else if (L <= 64)
        hipLaunchKernelGGL(lstm_bwd_k<64>, dim3(S), dim3(threads), lds, st, wblk, acts, cs, g_top, dG, dx, S, T, L, layers);
    else
        hipLaunchKernelGGL(lstm_bwd_k<0>, dim3(S), dim3(threads), lds, st, wblk, acts, cs, g_top, dG, dx, S, T, L, layers);
    RBVAE_CHECK_LAUNCH("lstm_bwd");
    return RBVAE_OK;
}

int rbvae_lstm_bwd(const float* wblk, const float* wT, const float* acts, const float* cs, const float* g_top, float* dG,
                   float* dx, int S, int T, int L, int layers, void* stream) {
    return lstm_bwd_impl(wblk, wT, acts, cs, g_top, dG, dx, S, T, L, layers, 1, 0, nullptr, 0, 0, nullptr, BinBwd{}, stream);
}

int rbvae_lstm_bwd_ex(const float* wblk, const float* acts, const float* cs, const float* g_top_parts, int nparts,
                      long part_stride, float* dG, float* dx, void* cast_out, int cast_dtype, int cast_ld,
                      float* dx_colsum, int S, int T, int L, int layers, void* stream) {
    RBVAE_CHECK_ARG(nparts >= 1 && (nparts == 1 || part_stride >= (long)S * T * L), "lstm_bwd_ex: bad slabs");
    return lstm_bwd_impl(wblk, nullptr, acts, cs, g_top_parts, dG, dx, S, T, L, layers, nparts, part_stride, cast_out, cast_dtype,
                         cast_ld, dx_colsum, BinBwd{}, stream);
}

int rbvae_lstm_bwd_bin(const float* wblk, const float* acts, const float* cs, const float* g_z, const float* y_soft,
                       const float* z, const float* g_hs, float tau, const float* tau_dev, float kl_weight, float kl_p,
                       float kl_eps, int kl_clamp, float* dG, float* dx, void* cast_out, int cast_dtype, int cast_ld, float* dx_colsum,
                       int S, int T, int L, int layers, void* stream) {
    RBVAE_CHECK_ARG(g_z && y_soft && z && (tau_dev || tau > 0.f), "lstm_bwd_bin: bad arguments");
    RBVAE_CHECK_ARG(kl_weight == 0.f || (kl_p > 0.f && kl_p < 1.f), "lstm_bwd_bin: kl_p=%g outside (0,1)", kl_p);
    BinBwd bb;
    bb.gz = g_z; bb.y = y_soft; bb.z = z; bb.g_hs = g_hs; bb.tau = tau; bb.tau_dev = tau_dev;
    bb.klw = kl_weight / (float)((long)S * T);
    bb.lp = kl_weight != 0.f ? logf(kl_p) : 0.f; bb.l1p = kl_weight != 0.f ? logf(1.0f - kl_p) : 0.f;
    bb.keps = kl_eps; bb.clamp = kl_clamp; bb.on = 1;
    return lstm_bwd_impl(wblk, nullptr, acts, cs, g_z, dG, dx, S, T, L, layers, 1, 0, cast_out, cast_dtype, cast_ld, dx_colsum, bb,
                         stream);
}

static size_t pair_bwd_lds(int T, int L, int layers) {
    return (size_t)(5 * T * L + 2 * layers * (T * 5 * L + 12 * L)) * sizeof(float);
}

int rbvae_lstm_pair_bwd_ok(int T, int L, int layers) {
    const int threads = ((4 * L + 63) / 64) * 64;
    return L <= 32 && 2 * layers * threads <= 1024 && pair_bwd_lds(T, L, layers) <= 64 * 1024;
}

int rbvae_lstm_pair_bwd(const float* wblk_enc, const float* wblk_dec, const float* acts_enc, const float* cs_enc,
                        const float* acts_dec, const float* cs_dec, const float* g_top_parts, int nparts, long part_stride,
                        const float* gz_extra, const float* y_soft, const float* z, const float* g_hs, float tau,
                        const float* tau_dev, float kl_weight, float kl_p, float kl_eps, int kl_clamp, float* dG_enc,
                        float* dG_dec, float* dx, float* dz, void* cast_out, int cast_dtype, int cast_ld, float* dx_colsum,
                        int S, int T, int L, int layers, void* stream) {
    RBVAE_CHECK_ARG(wblk_enc && wblk_dec && acts_enc && cs_enc && acts_dec && cs_dec && g_top_parts && y_soft && z &&
                    dG_enc && dG_dec && dx && S > 0 && T > 0 && L > 0 && layers > 0, "lstm_pair_bwd: bad arguments");
    RBVAE_CHECK_ARG(rbvae_lstm_pair_bwd_ok(T, L, layers), "lstm_pair_bwd: T=%d L=%d layers=%d outside the fused kernel's "
                    "range (rbvae_lstm_pair_bwd_ok)", T, L, layers);
    RBVAE_CHECK_ARG(nparts >= 1 && (nparts == 1 || part_stride >= (long)S * T * L), "lstm_pair_bwd: bad slabs");
    RBVAE_CHECK_ARG(tau_dev || tau > 0.f, "lstm_pair_bwd: tau=%g", tau);
    RBVAE_CHECK_ARG(kl_weight == 0.f || (kl_p > 0.f && kl_p < 1.f), "lstm_pair_bwd: kl_p=%g outside (0,1)", kl_p);
    RBVAE_CHECK_ARG(!cast_out || ((cast_dtype == RBVAE_F32 || cast_dtype == RBVAE_BF16) && cast_ld >= L),
                    "lstm_pair_bwd: cast output dtype %d ld %d", cast_dtype, cast_ld);
    PairBwdArgs a;
    a.wblk_e = wblk_enc; a.wblk_d = wblk_dec; a.acts_e = acts_enc; a.cs_e = cs_enc; a.acts_d = acts_dec; a.cs_d = cs_dec;
    a.g_top = g_top_parts; a.nparts = nparts; a.part_stride = part_stride;
    a.dG_e = dG_enc; a.dG_d = dG_dec; a.dx = dx; a.dz = dz; a.gz_extra = gz_extra;
    a.cast_out = cast_out; a.cast_bf16 = cast_dtype == RBVAE_BF16; a.cast_ld = cast_ld; a.dx_colsum = dx_colsum;
    a.bb.gz = nullptr; a.bb.y = y_soft; a.bb.z = z; a.bb.g_hs = g_hs; a.bb.tau = tau; a.bb.tau_dev = tau_dev;
    a.bb.klw = kl_weight / (float)((long)S * T);
    a.bb.lp = kl_weight != 0.f ? logf(kl_p) : 0.f; a.bb.l1p = kl_weight != 0.f ? logf(1.0f - kl_p) : 0.f;
    a.bb.keps = kl_eps; a.bb.clamp = kl_clamp; a.bb.on = 1;
    const int threads = ((4 * L + 63) / 64) * 64;
    a.S = S; a.T = T; a.L = L; a.layers = layers; a.G = threads;
    const size_t lds = pair_bwd_lds(T, L, layers);
    if (L == 32 && 2 * layers * 64 <= 512 && lstm_unit_threads)
        hipLaunchKernelGGL(lstm_pair_bwd_unit_k, dim3(S), dim3(2 * layers * 64), lds, (hipStream_t)stream, a);
    else if (L == 32)
        hipLaunchKernelGGL((lstm_pair_bwd_k<32, true>), dim3(S), dim3(2 * layers * threads), lds, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL((lstm_pair_bwd_k<32, false>), dim3(S), dim3(2 * layers * threads), lds, (hipStream_t)stream, a);
    RBVAE_CHECK_LAUNCH("lstm_pair_bwd");
    return RBVAE_OK;
}

static int launch_lstm_wgrad(const float* dG, const float* hs_all, const float* hprev, float* gblk, const float* dG2,
                             const float* hs_all2, const float* hprev2, float* gblk2, int S, int T, int L, int layers,
                             int accumulate, void* stream) {
    constexpr int use_mfma = 1;
    if (use_mfma) {
        dim3 mgrid(cdiv(4 * L, 16) * cdiv(L + 1, 16), 2, dG2 ? 2 * layers : layers);
        hipLaunchKernelGGL(lstm_wgrad_mfma_k, mgrid, dim3(256), 0, (hipStream_t)stream, dG, hs_all, hprev, gblk, dG2,
                           hs_all2, hprev2, gblk2, S, T, L, layers, accumulate);
        RBVAE_CHECK_LAUNCH("lstm_wgrad_mfma");
        return RBVAE_OK;
    }
    dim3 grid(cdiv(4 * L, LW_JT), 2, dG2 ? 2 * layers : layers);
    const size_t lds = (size_t)(LW_ROWS * LW_JT + LW_ROWS * (L + 1)) * sizeof(float);
    RBVAE_CHECK_ARG(lds <= 64 * 1024, "lstm_wgrad: L=%d too large", L);
    if (L <= 32)
        hipLaunchKernelGGL(lstm_wgrad_tiled_k<2>, grid, dim3(256), lds, (hipStream_t)stream, dG, hs_all, hprev, gblk, dG2,
                           hs_all2, hprev2, gblk2, S, T, L, layers, accumulate);
    else if (L <= 64)
        hipLaunchKernelGGL(lstm_wgrad_tiled_k<3>, grid, dim3(256), lds, (hipStream_t)stream, dG, hs_all, hprev, gblk, dG2,
                           hs_all2, hprev2, gblk2, S, T, L, layers, accumulate);
    else
        hipLaunchKernelGGL(lstm_wgrad_tiled_k<5>, grid, dim3(256), lds, (hipStream_t)stream, dG, hs_all, hprev, gblk, dG2,
                           hs_all2, hprev2, gblk2, S, T, L, layers, accumulate);
    RBVAE_CHECK_LAUNCH("lstm_wgrad");
    return RBVAE_OK;
}

int rbvae_lstm_wgrad(const float* dG, const float* hs_all, const float* hprev, float* gblk, int S, int T, int L,
                     int layers, int accumulate, void* stream) {
    RBVAE_CHECK_ARG(dG && hs_all && hprev && gblk && S > 0 && T > 0 && L > 0 && layers > 0, "lstm_wgrad: bad arguments");
    return launch_lstm_wgrad(dG, hs_all, hprev, gblk, nullptr, nullptr, nullptr, nullptr, S, T, L, layers, accumulate,
                             stream);
}

int rbvae_lstm_wgrad_pair(const float* dG_a, const float* hs_a, const float* hprev_a, float* gblk_a, const float* dG_b,
                          const float* hs_b, const float* hprev_b, float* gblk_b, int S, int T, int L, int layers,
                          int accumulate, void* stream) {
    RBVAE_CHECK_ARG(dG_a && hs_a && hprev_a && gblk_a && dG_b && hs_b && hprev_b && gblk_b && S > 0 && T > 0 && L > 0 &&
                        layers > 0, "lstm_wgrad_pair: bad arguments");
    return launch_lstm_wgrad(dG_a, hs_a, hprev_a, gblk_a, dG_b, hs_b, hprev_b, gblk_b, S, T, L, layers, accumulate, stream);
}

}  // extern "C"
