// State-consistency metric of the reference's validation loop (calculate_state_consistency,
// models/percep_RBVAE/percep_RBVAE_train.py:473-497): per state, the share of validation frames whose binary code equals
// the state's most common code (np.unique(..., axis=0, return_counts=True) + argmax: ties go to the lexicographically
// smallest code).  The reference does this on the host after encoding one frame per call; here the codes of all frames
// are already on the device (one batched encode), are packed to 128-bit keys (element 0 = most significant bit, so key
// order = row order of np.unique) and voted on there.
#include "common.h"

namespace rbvae {

struct Key128 { unsigned w[4]; };

__device__ __forceinline__ bool key_eq(const uint4 a, const uint4 b) { return a.x == b.x && a.y == b.y && a.z == b.z && a.w == b.w; }
__device__ __forceinline__ bool key_lt(const uint4 a, const uint4 b) {
    if (a.x != b.x) return a.x < b.x;
    if (a.y != b.y) return a.y < b.y;
    if (a.z != b.z) return a.z < b.z;
    return a.w < b.w;
}

// keys[f] = bits of codes[f][0..L) (> 0.5), element 0 first
__global__ void vote_pack_k(const float* __restrict__ codes, int F, int L, uint4* __restrict__ keys) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    unsigned w[4] = {0u, 0u, 0u, 0u};
    for (int j = 0; j < L; ++j)
        if (codes[(long)f * L + j] > 0.5f) w[j >> 5] |= 1u << (31 - (j & 31));
    keys[f] = make_uint4(w[0], w[1], w[2], w[3]);
}

// counts[f] = number of frames of f's state with f's key
__global__ __launch_bounds__(256) void vote_count_k(const uint4* __restrict__ keys, const int* __restrict__ labels, int F,
                                                    int* __restrict__ counts) {
    __shared__ uint4 sk[256];
    __shared__ int sl[256];
    const int f = blockIdx.x * 256 + threadIdx.x;
    const uint4 mine = f < F ? keys[f] : make_uint4(0, 0, 0, 0);
    const int lab = f < F ? labels[f] : -1;
    int c = 0;
    for (int j0 = 0; j0 < F; j0 += 256) {
        __syncthreads();
        const int j = j0 + threadIdx.x;
        sk[threadIdx.x] = j < F ? keys[j] : make_uint4(0, 0, 0, 0);
        sl[threadIdx.x] = j < F ? labels[j] : -2;
        __syncthreads();
        const int n = min(256, F - j0);
        for (int i = 0; i < n; ++i) c += (sl[i] == lab && key_eq(sk[i], mine)) ? 1 : 0;
    }
    if (f < F) counts[f] = c;
}

// out[s] = {count of the winning code, frames of state s}: winner = highest count, then smallest key
__global__ __launch_bounds__(256) void vote_pick_k(const uint4* __restrict__ keys, const int* __restrict__ labels,
                                                   const int* __restrict__ counts, int F, int* __restrict__ out) {
    __shared__ int s_cnt[256], s_n[256];
    __shared__ uint4 s_key[256];
    const int s = blockIdx.x;
    int best = 0, n = 0;
    uint4 bk = make_uint4(~0u, ~0u, ~0u, ~0u);
    for (int f = threadIdx.x; f < F; f += 256) {
        if (labels[f] != s) continue;
        ++n;
        const int c = counts[f];
        const uint4 k = keys[f];
        if (c > best || (c == best && key_lt(k, bk))) { best = c; bk = k; }
    }
    s_cnt[threadIdx.x] = best; s_n[threadIdx.x] = n; s_key[threadIdx.x] = bk;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            const int c = s_cnt[threadIdx.x + o];
            const uint4 k = s_key[threadIdx.x + o];
            if (c > s_cnt[threadIdx.x] || (c == s_cnt[threadIdx.x] && key_lt(k, s_key[threadIdx.x]))) {
                s_cnt[threadIdx.x] = c; s_key[threadIdx.x] = k;
            }
            s_n[threadIdx.x] += s_n[threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out[2 * s] = s_cnt[0]; out[2 * s + 1] = s_n[0]; }
}

}  // namespace rbvae

using namespace rbvae;

extern "C" int rbvae_state_vote(const float* codes, const int* labels, int F, int L, int n_states, void* keys_ws,
                                int* counts_ws, int* out, void* stream) {
    RBVAE_CHECK_ARG(codes && labels && keys_ws && counts_ws && out && F > 0 && n_states > 0, "state_vote: bad arguments");
    RBVAE_CHECK_ARG(L > 0 && L <= 128, "state_vote: code length %d outside 1..128", L);
    RBVAE_CHECK_ARG((uintptr_t)keys_ws % 16 == 0, "state_vote: keys workspace must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(vote_pack_k, dim3(cdiv(F, 256)), dim3(256), 0, st, codes, F, L, (uint4*)keys_ws);
    hipLaunchKernelGGL(vote_count_k, dim3(cdiv(F, 256)), dim3(256), 0, st, (const uint4*)keys_ws, labels, F, counts_ws);
    hipLaunchKernelGGL(vote_pick_k, dim3(n_states), dim3(256), 0, st, (const uint4*)keys_ws, labels, counts_ws, F, out);
    RBVAE_CHECK_LAUNCH("state_vote");
    return RBVAE_OK;
}
