// Layout and small dense helpers around the two GEMM kernels: weight packing,
// slab reduction back to torch layouts, im2col for the few-channel ends of the
// network, the col2im + sigmoid + MSE epilogue of the last ConvTranspose2d, column
// sums (bias gradients), f32 -> T casts, the skinny Linear (N <= 128) and Adam.
#include "common.h"
#include <stdlib.h>

namespace rbvae {

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

// out[i0*s0 + i1*s1 + i2*s2] = in[i0][i1][i2]   (in: contiguous f32)
template <typename T>
__global__ void pack3_k(const float* __restrict__ in, T* __restrict__ out, int d0, int d1, int d2, long s0,
                        long s1, long s2) {
    const long n = (long)d0 * d1 * d2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int i2 = (int)(i % d2);
        const long r = i / d2;
        const int i1 = (int)(r % d1), i0 = (int)(r / d1);
        Elem<T>::store(out + i0 * s0 + i1 * s1 + i2 * s2, in[i]);
    }
}

// out[i0][i1][i2] (+)= scale * sum_k in[k*slab + i0*s0 + i1*s1 + i2*s2]   (fixed k order)
__global__ void permute_reduce_k(const float* __restrict__ in, int nslab, long slab, float* __restrict__ out,
                                 int d0, int d1, int d2, long s0, long s1, long s2, float scale, int accumulate) {
    const long n = (long)d0 * d1 * d2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int i2 = (int)(i % d2);
        const long r = i / d2;
        const int i1 = (int)(r % d1), i0 = (int)(r / d1);
        const float* p = in + i0 * s0 + i1 * s1 + i2 * s2;
        float acc = 0.f;
        for (int k = 0; k < nslab; ++k) acc += p[k * slab];
        acc *= scale;
        out[i] = accumulate ? out[i] + acc : acc;
    }
}

template <typename T>
__global__ void cast_pad_k(const float* __restrict__ in, T* __restrict__ out, int rows, int L, int Lpad) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * Lpad) return;
    const int r = i / Lpad, c = i - r * Lpad;
    Elem<T>::store(out + i, c < L ? in[(long)r * L + c] : 0.f);
}

// ---- column sums: X [P][ld] -> partial [nblk][C] -> out[C] -------------------
// grid = (row blocks of `rpb` rows, 256-column groups); thread -> (column, row lane)
template <typename T>
__global__ __launch_bounds__(256) void colsum_partial_k(const T* __restrict__ X, int P, int C, int ld, int rpb,
                                                        float* __restrict__ ws) {
    __shared__ float red[256];
    const int cw = C < 256 ? C : 256;
    const int rl = 256 / cw;
    const int tc = threadIdx.x % cw, tr = threadIdx.x / cw;
    const int r0 = blockIdx.x * rpb;
    const int r1 = min(P, r0 + rpb);
    const int c = blockIdx.y * 256 + tc;
    float acc = 0.f;
    if (tr < rl && c < C)
        for (int r = r0 + tr; r < r1; r += rl) acc += Elem<T>::load(X + (long)r * ld + c);
    red[threadIdx.x] = acc;
    __syncthreads();
    if (tr == 0 && c < C) {
        float t = 0.f;
        for (int k = 0; k < rl; ++k) t += red[k * cw + tc];
        ws[(long)blockIdx.x * C + c] = t;
    }
}
// The same partial sums for wide tensors (C a multiple of 16 / sizeof(T), 16-byte aligned rows): a thread owns one 16-byte
// chunk of columns and walks the block's rows with 16-byte loads, four in flight.  The one-element-per-thread form above
// moved the decoder fc's 128 x 56 320 bf16 gradient (14 MB, native 4x88x160) in 89 us; the sums are the same f32 additions
// in the same row order.
template <typename T>
__global__ __launch_bounds__(256) void colsum_partial_vec_k(const T* __restrict__ X, int P, int C, int ld, int rpb,
                                                            float* __restrict__ ws) {
    constexpr int EC = 16 / sizeof(T);
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    const int r0 = blockIdx.x * rpb, r1 = min(P, r0 + rpb);
    const int c = (blockIdx.y * 256 + threadIdx.x) * EC;
    if (c >= C) return;
    float acc[EC];
#pragma unroll
    for (int e = 0; e < EC; ++e) acc[e] = 0.f;
    const T* px = X + (long)r0 * ld + c;
    int r = r0;
    for (; r + 4 <= r1; r += 4) {
        u32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *(const u32x4*)(px + (long)u * ld);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const T* ev = (const T*)&v[u];
#pragma unroll
            for (int e = 0; e < EC; ++e) acc[e] += Elem<T>::load(ev + e);
        }
        px += 4l * ld;
    }
    for (; r < r1; ++r) {
        const u32x4 v = *(const u32x4*)px;
        const T* ev = (const T*)&v;
#pragma unroll
        for (int e = 0; e < EC; ++e) acc[e] += Elem<T>::load(ev + e);
        px += ld;
    }
    float* o = ws + (long)blockIdx.x * C + c;
#pragma unroll
    for (int e = 0; e < EC; e += 4) *(float4*)(o + e) = make_float4(acc[e], acc[e + 1], acc[e + 2], acc[e + 3]);
}
template <typename T>
static void launch_colsum_partial(const T* X, int P, int C, int ld, int rpb, float* ws, hipStream_t st) {
    constexpr int EC = 16 / sizeof(T);
    if (C >= 2048 && C % EC == 0 && ld % EC == 0 && (uintptr_t)X % 16 == 0 && (uintptr_t)ws % 16 == 0) {
        hipLaunchKernelGGL(colsum_partial_vec_k<T>, dim3(cdiv(P, rpb), cdiv(C, 256 * EC)), dim3(256), 0, st, X, P, C, ld, rpb, ws);
    } else {
        hipLaunchKernelGGL(colsum_partial_k<T>, dim3(cdiv(P, rpb), cdiv(C, 256)), dim3(256), 0, st, X, P, C, ld, rpb, ws);
    }
}
// out[c] (+)= scale * sum_b ws[b][c]: 64 columns per block, 4 row lanes, fixed order
__global__ __launch_bounds__(256) void colsum_final_k(const float* __restrict__ ws, int nblk, int C,
                                                      float* __restrict__ out, float scale, int accumulate) {
    __shared__ float red[256];
    const int tc = threadIdx.x & 63, tr = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + tc;
    float t = 0.f;
    if (c < C)
        for (int b = tr; b < nblk; b += 4) t += ws[(long)b * C + c];
    red[threadIdx.x] = t;
    __syncthreads();
    if (tr == 0 && c < C) {
        t = (red[tc] + red[64 + tc] + red[128 + tc] + red[192 + tc]) * scale;
        out[c] = accumulate ? out[c] + t : t;
    }
}

// Frame n of a batch -> element offset of its first value.  d1 == 0: n * s2.  Otherwise n is read as the
// mixed-radix number (n / d1, (n % d1) / d2, n % d2) with strides (s0, s1, s2): the fused trainer runs both
// views of an item batch [B][2][T] as frames v*(B*T) + b*T + t without first copying them into that order.
struct FrameMap {
    int d1, d2;
    long s0, s1, s2;
};
template <typename I> __device__ __forceinline__ I frame_off(const FrameMap& f, I n) {
    if (f.d1 == 0) return n * (I)f.s2;
    const I a = n / (I)f.d1, r = n - a * (I)f.d1;
    const I b = r / (I)f.d2, c = r - b * (I)f.d2;
    return a * (I)f.s0 + b * (I)f.s1 + c * (I)f.s2;
}
static long frame_span(const FrameMap& f, int N) {      // largest frame offset (non-negative strides)
    if (f.d1 == 0) return (long)(N - 1) * f.s2;
    return (long)((N - 1) / f.d1) * f.s0 + (long)(f.d1 / f.d2 - 1) * f.s1 + (long)(f.d2 - 1) * f.s2;
}

// ---- im2col: strided f32 source -> col[P][Kpad] (column (kh*KW+kw)*C + c) ----
// one thread = 8 consecutive columns of one row (Kpad % 8 == 0): one 16-B (bf16) / two 16-B (f32) stores
template <typename T>
__global__ void im2col_k(const float* __restrict__ src, FrameMap fm, long sc, long sh, long sw, int N, int C, int IH,
                         int IW, int OH, int OW, int KH, int KW, int stride, int pad, int Kpad,
                         T* __restrict__ col) {
    const int gpr = Kpad >> 3;                          // column groups per row
    const long tot = (long)N * OH * OW * gpr;
    const int kreal = KH * KW * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += (long)gridDim.x * blockDim.x) {
        const int kg = (int)(i % gpr);
        const long pp = i / gpr;
        const int ow = (int)(pp % OW);
        const long r = pp / OW;
        const int oh = (int)(r % OH), n = (int)(r / OH);
        const int ih0 = oh * stride - pad, iw0 = ow * stride - pad;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int kx = kg * 8 + e;
            const int kxc = kx < kreal ? kx : 0;
            const int t = kxc / C, c = kxc - t * C;
            const int kh = t / KW, kw = t - kh * KW;
            const int ih = ih0 + kh, iw = iw0 + kw;
            const bool ok = kx < kreal && ih >= 0 && ih < IH && iw >= 0 && iw < IW;
            // clamped address: the load is unconditional (all 8 in flight), the select zeroes padding
            const int ihc = min(max(ih, 0), IH - 1), iwc = min(max(iw, 0), IW - 1);
            const float x = src[frame_off<long>(fm, n) + c * sc + ihc * sh + iwc * sw];
            v[e] = ok ? x : 0.f;
        }
        T* dst = col + pp * Kpad + kg * 8;
        if constexpr (sizeof(T) == 2) {
            uint4 pk;
            pk.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
            pk.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
            pk.z = (unsigned)f32_to_bf16(v[4]) | ((unsigned)f32_to_bf16(v[5]) << 16);
            pk.w = (unsigned)f32_to_bf16(v[6]) | ((unsigned)f32_to_bf16(v[7]) << 16);
            *(uint4*)dst = pk;
        } else {
            *(float4*)dst = make_float4(v[0], v[1], v[2], v[3]);
            *(float4*)(dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
        }
    }
}

// 32-bit-index variant for the shapes of this path (the 64-bit divisions and the per-element runtime divisions
// of the general kernel above made it ALU bound: ~1000 instructions per 16-byte store).  CC / KWC > 0 fix the
// channel count and kernel width at compile time (divisions become shifts / multiplies); 0 = runtime values.
template <typename T, int CC, int KWC>
__global__ __launch_bounds__(256) void im2col_fast_k(const float* __restrict__ src, FrameMap fm, int sc, int sh, int sw,
                                                     int N, int C_, int IH, int IW, int OH, int OW, int KH, int KW_,
                                                     int stride, int pad, int Kpad, T* __restrict__ col) {
    const unsigned C = CC > 0 ? CC : C_, KW = KWC > 0 ? KWC : KW_;
    const unsigned gpr = Kpad >> 3;
    const unsigned tot = (unsigned)N * OH * OW * gpr;
    const unsigned kreal = KH * KW * C;
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= tot) return;
    const unsigned pp = i / gpr, kg = i - pp * gpr;
    const unsigned r = pp / OW, ow = pp - r * OW;
    const unsigned n = r / OH, oh = r - n * OH;
    const int ih0 = (int)oh * stride - pad, iw0 = (int)ow * stride - pad;
    const float* base = src + frame_off<unsigned>(fm, n);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const unsigned kx = kg * 8 + e;
        const unsigned kxc = kx < kreal ? kx : 0;
        const unsigned t = kxc / C, c = kxc - t * C;
        const unsigned kh = t / KW, kw = t - kh * KW;
        const int ih = ih0 + (int)kh, iw = iw0 + (int)kw;
        const bool ok = kx < kreal && ih >= 0 && ih < IH && iw >= 0 && iw < IW;
        const int ihc = min(max(ih, 0), IH - 1), iwc = min(max(iw, 0), IW - 1);
        const float x = base[(int)c * sc + ihc * sh + iwc * sw];      // unconditional load, the select zeroes padding
        v[e] = ok ? x : 0.f;
    }
    T* dst = col + (size_t)pp * Kpad + kg * 8;
    if constexpr (sizeof(T) == 2) {
        uint4 pk;
        pk.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
        pk.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
        pk.z = (unsigned)f32_to_bf16(v[4]) | ((unsigned)f32_to_bf16(v[5]) << 16);
        pk.w = (unsigned)f32_to_bf16(v[6]) | ((unsigned)f32_to_bf16(v[7]) << 16);
        *(uint4*)dst = pk;
    } else {
        *(float4*)dst = make_float4(v[0], v[1], v[2], v[3]);
        *(float4*)(dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
}

// LDS-staged variant for stride 2, pad 1, 3x3, C <= 4 (both ends of the RBVAE CNNs): a workgroup takes IM_R output
// rows of one frame, loads the (2*IM_R + 1) input rows it needs with coalesced loads into LDS (halo columns
// zeroed), and writes the column matrix as consecutive 16-byte chunks.  The gather form above issues 8 scattered
// 4-byte loads per 16-byte store; this one reads every input value once -- and measured SLOWER in the step (its
// two phases serialise inside 1024 small workgroups), so it is an experiment switch (RBVAE_IM2COL_LDS=1).
constexpr int IM_R = 4;
template <typename T>
__global__ __launch_bounds__(256) void im2col_lds_k(const float* __restrict__ src, FrameMap fm, int sc, int sh, int sw,
                                                    int N, int C, int IH, int IW, int OH, int OW, int Kpad,
                                                    T* __restrict__ col) {
    extern __shared__ float tile[];                 // [C][2*IM_R+1][IW+2], column 0 / IW+1 = padding
    const int rows_in = 2 * IM_R + 1, pitch = IW + 2;
    const int rb = (OH + IM_R - 1) / IM_R;
    const int n = blockIdx.x / rb, oh0 = (blockIdx.x - n * rb) * IM_R;
    const int ih0 = 2 * oh0 - 1;
    const float* base = src + frame_off<long>(fm, (long)n);
    const int cnt = C * rows_in * pitch;
    // element order of the load loop follows the source's fastest stride: (c, r, col) for NCHW, (r, col, c) for NHWC
    for (int i = threadIdx.x; i < cnt; i += 256) {
        int c, r, cc;
        if (sc == 1) { c = i % C; const int t = i / C; cc = t % pitch; r = t / pitch; }
        else { cc = i % pitch; const int t = i / pitch; r = t % rows_in; c = t / rows_in; }
        const int ih = ih0 + r, iw = cc - 1;
        float v = 0.f;
        if (ih >= 0 && ih < IH && iw >= 0 && iw < IW) v = base[(long)c * sc + (long)ih * sh + (long)iw * sw];
        tile[(c * rows_in + r) * pitch + cc] = v;
    }
    __syncthreads();
    const int gpr = Kpad >> 3, kreal = 9 * C;
    const int nchunk = IM_R * OW * gpr;
    for (int i = threadIdx.x; i < nchunk; i += 256) {
        const int kg = i % gpr, pp = i / gpr;
        const int ow = pp % OW, orow = pp / OW;
        const int oh = oh0 + orow;
        if (oh >= OH) break;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int kx = kg * 8 + e;
            float x = 0.f;
            if (kx < kreal) {
                const int t = kx / C, c = kx - t * C;
                const int kh = t / 3, kw = t - kh * 3;
                x = tile[(c * rows_in + 2 * orow + kh) * pitch + 2 * ow + kw];
            }
            v[e] = x;
        }
        T* dst = col + ((size_t)(n * OH + oh) * OW + ow) * Kpad + kg * 8;
        if constexpr (sizeof(T) == 2) {
            uint4 pk;
            pk.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
            pk.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
            pk.z = (unsigned)f32_to_bf16(v[4]) | ((unsigned)f32_to_bf16(v[5]) << 16);
            pk.w = (unsigned)f32_to_bf16(v[6]) | ((unsigned)f32_to_bf16(v[7]) << 16);
            *(uint4*)dst = pk;
        } else {
            *(float4*)dst = make_float4(v[0], v[1], v[2], v[3]);
            *(float4*)(dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
        }
    }
}

// ---- last ConvTranspose2d: col2im gather + bias + sigmoid (+ MSE, + d(loss)/d(pre)) ----
// Y[(n,a,b)][t*Cout + co] holds each input pixel's contribution to every tap.
// IDX = unsigned for outputs below 2^31 elements (64-bit divisions cost ~100 instructions each), else long
template <typename T, typename IDX>
__global__ __launch_bounds__(256) void col2im_sigmoid_k(
    const T* __restrict__ Y, int ldy, const float* __restrict__ bias, int N, int IH, int IW, int OH, int OW,
    int Cout, int KH, int KW, int pad, float* __restrict__ xr, const float* __restrict__ target, FrameMap tfm,
    float* __restrict__ sse_ws, float* __restrict__ dpre, float gscale, const float* __restrict__ gs_dev) {
    __shared__ float red[4];
    const IDX tot = (IDX)N * OH * OW * Cout;
    float sse = 0.f;
    float gsc = gscale;
    if (gs_dev) gsc *= gs_dev[0];
    for (IDX i = (IDX)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += (IDX)gridDim.x * blockDim.x) {
        // i enumerates the NCHW output: (n, co, oh, ow), ow fastest -> coalesced xr/target accesses
        IDX r = i / (IDX)OW;
        const int ow = (int)(i - r * (IDX)OW);
        IDX r2 = r / (IDX)OH;
        const int oh = (int)(r - r2 * (IDX)OH);
        const int n = (int)(r2 / (IDX)Cout), co = (int)(r2 - (IDX)n * (IDX)Cout);
        float v = bias ? bias[co] : 0.f;
        for (int kh = (oh + pad) & 1; kh < KH; kh += 2) {
            const int a = (oh + pad - kh) >> 1;
            if (a < 0 || a >= IH) continue;
            for (int kw = (ow + pad) & 1; kw < KW; kw += 2) {
                const int b = (ow + pad - kw) >> 1;
                if (b < 0 || b >= IW) continue;
                v += Elem<T>::load(Y + ((long)(n * IH + a) * IW + b) * ldy + (kh * KW + kw) * Cout + co);
            }
        }
        const float s = sigmoidf_(v);
        xr[i] = s;
        if (target) {
            // frame n of the target sits at its mapped offset; inside a frame the layout is xr's (co, oh, ow)
            const float d = s - target[frame_off<IDX>(tfm, (IDX)n) + (i - (IDX)n * (IDX)Cout * (IDX)OH * (IDX)OW)];
            sse += d * d;
            if (dpre) dpre[((long)(n * OH + oh) * OW + ow) * Cout + co] = gsc * d * s * (1.f - s);
        }
    }
    if (sse_ws) {
        const float tot_b = block_sum(sse, red);
        if (threadIdx.x == 0) sse_ws[blockIdx.x] = tot_b;
    }
}
// The same for Cout <= 4 with one thread per output PIXEL: a tap's Cout products are one contiguous 6-8 byte
// read (the element-per-thread form above issues Cout scattered 2-byte reads per tap), dpre leaves as one
// 16-byte store, xr / target stay coalesced along ow per channel plane.  Same sums in the same order.
template <typename T>
__global__ __launch_bounds__(256) void col2im_sigmoid_pix_k(
    const T* __restrict__ Y, int ldy, const float* __restrict__ bias, int N, int IH, int IW, int OH, int OW,
    int Cout, int KH, int KW, int pad, float* __restrict__ xr, const float* __restrict__ target, FrameMap tfm,
    float* __restrict__ sse_ws, float* __restrict__ dpre, float gscale, const float* __restrict__ gs_dev) {
    __shared__ float red[4];
    const unsigned npix = (unsigned)N * OH * OW;
    float sse = 0.f;
    float gsc = gscale;
    if (gs_dev) gsc *= gs_dev[0];
    const unsigned plane = (unsigned)OH * OW;
    float dsum[4] = {0.f, 0.f, 0.f, 0.f};
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < npix; i += gridDim.x * 256u) {
        const unsigned r = i / (unsigned)OW, ow = i - r * (unsigned)OW;
        const unsigned n = r / (unsigned)OH, oh = r - n * (unsigned)OH;
        float v[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = (bias && c < Cout) ? bias[c] : 0.f;
        for (int kh = (oh + pad) & 1; kh < KH; kh += 2) {
            const int a = ((int)oh + pad - kh) >> 1;
            if (a < 0 || a >= IH) continue;
            for (int kw = (ow + pad) & 1; kw < KW; kw += 2) {
                const int b = ((int)ow + pad - kw) >> 1;
                if (b < 0 || b >= IW) continue;
                const T* yp = Y + ((size_t)(n * IH + a) * IW + b) * ldy + (kh * KW + kw) * Cout;
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (c < Cout) v[c] += Elem<T>::load(yp + c);
            }
        }
        const size_t xo = (size_t)n * Cout * plane + (size_t)oh * OW + ow;
        const float* tp = target ? target + frame_off<unsigned>(tfm, n) + (size_t)oh * OW + ow : nullptr;
        float d4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (c >= Cout) break;
            const float sg = sigmoidf_(v[c]);
            xr[xo + (size_t)c * plane] = sg;
            if (target) {
                const float d = sg - tp[(size_t)c * plane];
                sse += d * d;
                d4[c] = gsc * d * sg * (1.f - sg);
            }
        }
        if (target && dpre) {
            float* dp = dpre + (size_t)i * Cout;
            if (Cout == 4) *(float4*)dp = make_float4(d4[0], d4[1], d4[2], d4[3]);
            else
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (c < Cout) dp[c] = d4[c];
#pragma unroll
            for (int c = 0; c < 4; ++c) dsum[c] += d4[c];
        }
    }
    if (sse_ws) {
        // squared-error sum and the four column sums of dpre in ONE pass: wave shuffles, then the four waves' values
        // meet in LDS in wave order (fixed order: reproducible); five separate block sums cost ten barriers
        __shared__ float red5[4][5];
        float vals[5] = {sse, dsum[0], dsum[1], dsum[2], dsum[3]};
#pragma unroll
        for (int k = 0; k < 5; ++k) vals[k] = wave_sum(vals[k]);
        const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < 5; ++k) red5[wid][k] = vals[k];
        }
        __syncthreads();
        if (threadIdx.x < 5) {
            const int k = threadIdx.x;
            const float t = ((red5[0][k] + red5[1][k]) + red5[2][k]) + red5[3][k];
            // sse_ws[block] = squared-error sum; behind the gridDim.x of those, per-block column sums of dpre (the
            // last deconv's bias gradient): sse_ws[gridDim.x + 4*block + c]
            if (k == 0) sse_ws[blockIdx.x] = t;
            else if (dpre) sse_ws[gridDim.x + 4 * blockIdx.x + (k - 1)] = t;
        }
        (void)red;
    }
}
__global__ __launch_bounds__(1024) void sum_partials_k(const float* __restrict__ ws, int n, float scale,
                                                       float* out, int accumulate) {
    __shared__ float red[16];
    float a = 0.f;
    for (int i = threadIdx.x; i < n; i += 1024) a += ws[i];
    const float t = block_sum(a, red) * scale;
    if (threadIdx.x == 0) out[0] = accumulate ? out[0] + t : t;
}
// dpre[n][h][w][c] = g[n][c][h][w] * xr * (1 - xr)
__global__ void sigmoid_bwd_nhwc_k(const float* __restrict__ g, const float* __restrict__ xr,
                                   float* __restrict__ dpre, int N, int C, int H, int W) {
    const long tot = (long)N * C * H * W;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += (long)gridDim.x * blockDim.x) {
        const int w = (int)(i % W);
        long r = i / W;
        const int h = (int)(r % H);
        r /= H;
        const int c = (int)(r % C), n = (int)(r / C);
        const float s = xr[i];
        dpre[((long)(n * H + h) * W + w) * C + c] = g[i] * s * (1.f - s);
    }
}

// ---- skinny Linear: out[M][Nc] = A[M][K] * B[Nc][K]^T + bias, Nc <= 128 -------------
// One 16x16 output tile per 512-thread workgroup; the 8 waves split K, operands go
// straight from global to the MFMA (each is read once), partials meet in LDS.
template <typename T>
__global__ __launch_bounds__(512) void skinny_linear_k(const T* __restrict__ A, const T* __restrict__ B,
                                                       const float* __restrict__ bias, float* __restrict__ out,
                                                       int M, int Nc, int K, int lda, int ldb, int ldo, int kper) {
    // gridDim.z > 1: K split over workgroups, part z covers [z*kper, (z+1)*kper) and writes its own [M][ldo] slab
    // (the bias goes into part 0); the consumer sums the slabs in order (rbvae_lstm_fwd_parts / _bwd_parts)
    const int kbeg = blockIdx.z * kper;
    A += kbeg; B += kbeg;
    K = min(K - kbeg, kper);
    out += (size_t)blockIdx.z * M * ldo;
    if (blockIdx.z) bias = nullptr;
    constexpr int ES = sizeof(T);
    constexpr int KS = (ES == 2) ? 32 : 16;      // k per step (16 B per lane per operand)
    constexpr int EC = 16 / ES;
    __builtin_amdgcn_s_setprio(3);               // latency-bound link of the fc / LSTM chain: issue before the side stream's GEMM waves
    __shared__ float part[8][256];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int i = lane & 15, g = lane >> 4;
    const int m = blockIdx.x * 16 + i, n = blockIdx.y * 16 + i;
    const bool mv = m < M, nv = n < Nc;
    const T* ap = A + (long)(mv ? m : 0) * lda + g * EC;
    const T* bp = B + (long)(nv ? n : 0) * ldb + g * EC;
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    const u32x4_t z = {0u, 0u, 0u, 0u};
    // batches of UB K-steps: all 2*UB operand loads are in flight before the first MFMA (one step per iteration
    // paid a memory round trip per step: 16 of them at K = 4096)
    constexpr int UB = 8;
    for (int k0 = w * KS; k0 < K; k0 += UB * 8 * KS) {
        u32x4_t a[UB], b[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int k = k0 + u * 8 * KS;
            const bool kv = k < K;
            a[u] = (mv && kv) ? *(const u32x4_t*)(ap + k) : z;
            b[u] = (nv && kv) ? *(const u32x4_t*)(bp + k) : z;
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            if (k0 + u * 8 * KS >= K) break;
            if constexpr (ES == 2) {
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8_t*)&a[u], *(const bf16x8_t*)&b[u], acc, 0, 0, 0);
            } else {
                const f32x4_t af = *(const f32x4_t*)&a[u], bf = *(const f32x4_t*)&b[u];
#pragma unroll
                for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[q], bf[q], acc, 0, 0, 0);
            }
        }
    }
    // D[row = m-local 4g+r][col = n-local i]
#pragma unroll
    for (int r = 0; r < 4; ++r) part[w][(4 * g + r) * 16 + i] = acc[r];
    __syncthreads();
    if (threadIdx.x < 256) {
        const int rm = threadIdx.x >> 4, cn = threadIdx.x & 15;
        const int mm = blockIdx.x * 16 + rm, nn = blockIdx.y * 16 + cn;
        if (mm < M && nn < Nc) {
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) t += part[k][threadIdx.x];
            out[(long)mm * ldo + nn] = t + (bias ? bias[nn] : 0.f);
        }
    }
}

// ---- Adam (torch.optim.Adam defaults: no weight decay, no amsgrad) ------------------
// Same operation order as torch's single-tensor path: m.lerp_(g, 1-b1); v = v*b2 + ((1-b2)*g)*g;
// denom = sqrt(v)/sqrt(bc2) + eps; w += -(lr/bc1) * (m/denom).
__global__ void adam_k(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ m,
                       float* __restrict__ v, long n, float step_size, float one_m_b1, float b2, float one_m_b2,
                       float eps, float bc2_sqrt, float gscale, const float* __restrict__ hyper) {
    if (hyper) { step_size = hyper[0]; bc2_sqrt = hyper[1]; }
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float mi = m[i], vi = v[i];
        w[i] = adam_update(w[i], g[i], mi, vi, one_m_b1, b2, one_m_b2, eps, gscale, step_size, bc2_sqrt);
        m[i] = mi;
        v[i] = vi;
    }
}

static inline int grid_for(long n, int block = 256, int cap = 4096) {
    long b = (n + block - 1) / block;
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

}  // namespace rbvae

using namespace rbvae;

extern "C" {

int rbvae_pack3(int dtype, const float* in, void* out, int d0, int d1, int d2, long s0, long s1, long s2,
                void* stream) {
    RBVAE_CHECK_ARG(in && out && d0 > 0 && d1 > 0 && d2 > 0, "pack3: bad arguments");
    const long n = (long)d0 * d1 * d2;
    if (dtype == RBVAE_F32)
        hipLaunchKernelGGL(pack3_k<float>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, in, (float*)out, d0,
                           d1, d2, s0, s1, s2);
    else if (dtype == RBVAE_BF16)
        hipLaunchKernelGGL(pack3_k<bf16_t>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, in, (bf16_t*)out,
                           d0, d1, d2, s0, s1, s2);
    else
        return fail(RBVAE_E_INVALID, "pack3: dtype %d", dtype);
    RBVAE_CHECK_LAUNCH("pack3");
    return RBVAE_OK;
}

int rbvae_permute_reduce(const float* in, int nslab, long slab_stride, float* out, int d0, int d1, int d2, long s0,
                         long s1, long s2, float scale, int accumulate, void* stream) {
    RBVAE_CHECK_ARG(in && out && nslab > 0 && d0 > 0 && d1 > 0 && d2 > 0, "permute_reduce: bad arguments");
    const long n = (long)d0 * d1 * d2;
    hipLaunchKernelGGL(permute_reduce_k, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, in, nslab,
                       slab_stride, out, d0, d1, d2, s0, s1, s2, scale, accumulate);
    RBVAE_CHECK_LAUNCH("permute_reduce");
    return RBVAE_OK;
}

int rbvae_cast_pad(int dtype, const float* in, void* out, int rows, int L, int Lpad, void* stream) {
    RBVAE_CHECK_ARG(in && out && rows > 0 && L > 0 && Lpad >= L, "cast_pad: bad arguments");
    const int n = rows * Lpad;
    if (dtype == RBVAE_F32)
        hipLaunchKernelGGL(cast_pad_k<float>, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, in, (float*)out,
                           rows, L, Lpad);
    else if (dtype == RBVAE_BF16)
        hipLaunchKernelGGL(cast_pad_k<bf16_t>, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, in,
                           (bf16_t*)out, rows, L, Lpad);
    else
        return fail(RBVAE_E_INVALID, "cast_pad: dtype %d", dtype);
    RBVAE_CHECK_LAUNCH("cast_pad");
    return RBVAE_OK;
}

// rows per block: at most 256 row blocks, at least 16 rows each (short tensors still fan out over the chip)
static inline int colsum_rpb(int P) { return max(16, ((cdiv(P, 256) + 15) / 16) * 16); }
size_t rbvae_colsum_ws_floats(int P, int C) { return (size_t)cdiv(P, colsum_rpb(P)) * C; }

int rbvae_colsum(int dtype, const void* X, int P, int C, int ld, float* out, float* ws, float scale, int accumulate,
                 void* stream) {
    RBVAE_CHECK_ARG(X && out && ws && P > 0 && C > 0 && ld >= C, "colsum: bad arguments");
    const int rpb = colsum_rpb(P);
    const int nblk = cdiv(P, rpb);
    if (dtype == RBVAE_F32)
        launch_colsum_partial<float>((const float*)X, P, C, ld, rpb, ws, (hipStream_t)stream);
    else if (dtype == RBVAE_BF16)
        launch_colsum_partial<bf16_t>((const bf16_t*)X, P, C, ld, rpb, ws, (hipStream_t)stream);
    else
        return fail(RBVAE_E_INVALID, "colsum: dtype %d", dtype);
    hipLaunchKernelGGL(colsum_final_k, dim3(cdiv(C, 64)), dim3(256), 0, (hipStream_t)stream, ws, nblk, C, out, scale,
                       accumulate);
    RBVAE_CHECK_LAUNCH("colsum");
    return RBVAE_OK;
}

int rbvae_colsum_partial(int dtype, const void* X, int P, int C, int ld, float* ws, void* stream) {
    RBVAE_CHECK_ARG(X && ws && P > 0 && C > 0 && ld >= C, "colsum_partial: bad arguments");
    const int rpb = colsum_rpb(P);
    if (dtype == RBVAE_F32)
        launch_colsum_partial<float>((const float*)X, P, C, ld, rpb, ws, (hipStream_t)stream);
    else if (dtype == RBVAE_BF16)
        launch_colsum_partial<bf16_t>((const bf16_t*)X, P, C, ld, rpb, ws, (hipStream_t)stream);
    else
        return fail(RBVAE_E_INVALID, "colsum_partial: dtype %d", dtype);
    RBVAE_CHECK_LAUNCH("colsum_partial");
    return RBVAE_OK;
}

int rbvae_reduce_rows(const float* ws, int rows, int C, float* out, float scale, int accumulate, void* stream) {
    RBVAE_CHECK_ARG(ws && out && rows > 0 && C > 0, "reduce_rows: bad arguments");
    hipLaunchKernelGGL(colsum_final_k, dim3(cdiv(C, 64)), dim3(256), 0, (hipStream_t)stream, ws, rows, C, out, scale,
                       accumulate);
    RBVAE_CHECK_LAUNCH("reduce_rows");
    return RBVAE_OK;
}

static int im2col_impl(int dtype, const float* src, FrameMap fm, long sc, long sh, long sw, int N, int C, int IH, int IW,
                       int OH, int OW, int KH, int KW, int stride, int pad, int Kpad, void* col, void* stream) {
    const long sn = fm.s2;
    RBVAE_CHECK_ARG(src && col && N > 0 && C > 0 && Kpad >= KH * KW * C, "im2col: bad arguments");
    RBVAE_CHECK_ARG(fm.d1 == 0 || (fm.d2 > 0 && fm.d1 % fm.d2 == 0 && fm.s0 >= 0 && fm.s1 >= 0),
                    "im2col: frame map d1=%d d2=%d", fm.d1, fm.d2);
    RBVAE_CHECK_ARG(Kpad % 8 == 0 && (uintptr_t)col % 16 == 0, "im2col: Kpad=%d must be a multiple of 8, col 16-byte aligned", Kpad);
    const long tot = (long)N * OH * OW * (Kpad / 8);
    RBVAE_CHECK_ARG(dtype == RBVAE_F32 || dtype == RBVAE_BF16, "im2col: dtype %d", dtype);
    hipStream_t st = (hipStream_t)stream;
    // largest source offset the kernel forms (all strides non-negative in this library's callers)
    const long span = frame_span(fm, N) + (long)(C - 1) * sc + (long)(IH - 1) * sh + (long)(IW - 1) * sw;
    const bool fast = tot < (1l << 31) - 256 && span < (1l << 31) && (long)N * OH * OW * Kpad < (1l << 40) &&
                      sn >= 0 && sc >= 0 && sh >= 0 && sw >= 0;
    // LDS-staged form: measured on the bench step, same GPU, 3 runs each: 0.516 ms (gather form) vs 0.527 ms: off
    constexpr int lds_on = 0;
    const size_t tile_bytes = (size_t)C * (2 * IM_R + 1) * (IW + 2) * sizeof(float);
    if (lds_on && fast && KH == 3 && KW == 3 && stride == 2 && pad == 1 && C <= 4 && tile_bytes <= 48 * 1024 &&
        OH == (IH + 2 - 3) / 2 + 1 && OW == (IW + 2 - 3) / 2 + 1) {
        const dim3 grid(N * cdiv(OH, IM_R));
        if (dtype == RBVAE_F32)
            hipLaunchKernelGGL(im2col_lds_k<float>, grid, dim3(256), tile_bytes, st, src, fm, (int)sc, (int)sh, (int)sw, N, C,
                               IH, IW, OH, OW, Kpad, (float*)col);
        else
            hipLaunchKernelGGL(im2col_lds_k<bf16_t>, grid, dim3(256), tile_bytes, st, src, fm, (int)sc, (int)sh, (int)sw, N,
                               C, IH, IW, OH, OW, Kpad, (bf16_t*)col);
    } else if (fast) {
        const dim3 grid(cdiv(tot, 256));
#define RBVAE_IM2COL(TT, CC, KK)                                                                                      \
    hipLaunchKernelGGL((im2col_fast_k<TT, CC, KK>), grid, dim3(256), 0, st, src, fm, (int)sc, (int)sh, (int)sw, N,      \
                       C, IH, IW, OH, OW, KH, KW, stride, pad, Kpad, (TT*)col)
        if (dtype == RBVAE_F32) {
            if (C == 4 && KW == 3) RBVAE_IM2COL(float, 4, 3);
            else if (C == 3 && KW == 3) RBVAE_IM2COL(float, 3, 3);
            else RBVAE_IM2COL(float, 0, 0);
        } else {
            if (C == 4 && KW == 3) RBVAE_IM2COL(bf16_t, 4, 3);
            else if (C == 3 && KW == 3) RBVAE_IM2COL(bf16_t, 3, 3);
            else RBVAE_IM2COL(bf16_t, 0, 0);
        }
#undef RBVAE_IM2COL
    } else if (dtype == RBVAE_F32)
        hipLaunchKernelGGL(im2col_k<float>, dim3(grid_for(tot, 256, 8192)), dim3(256), 0, st, src,
                           fm, sc, sh, sw, N, C, IH, IW, OH, OW, KH, KW, stride, pad, Kpad, (float*)col);
    else
        hipLaunchKernelGGL(im2col_k<bf16_t>, dim3(grid_for(tot, 256, 8192)), dim3(256), 0, st, src,
                           fm, sc, sh, sw, N, C, IH, IW, OH, OW, KH, KW, stride, pad, Kpad, (bf16_t*)col);
    RBVAE_CHECK_LAUNCH("im2col");
    return RBVAE_OK;
}

int rbvae_im2col(int dtype, const float* src, long sn, long sc, long sh, long sw, int N, int C, int IH, int IW,
                 int OH, int OW, int KH, int KW, int stride, int pad, int Kpad, void* col, void* stream) {
    return im2col_impl(dtype, src, FrameMap{0, 0, 0, 0, sn}, sc, sh, sw, N, C, IH, IW, OH, OW, KH, KW, stride, pad, Kpad,
                       col, stream);
}

int rbvae_im2col_frames(int dtype, const float* src, int fd1, int fd2, long fs0, long fs1, long fs2, long sc, long sh,
                        long sw, int N, int C, int IH, int IW, int OH, int OW, int KH, int KW, int stride, int pad,
                        int Kpad, void* col, void* stream) {
    return im2col_impl(dtype, src, FrameMap{fd1, fd2, fs0, fs1, fs2}, sc, sh, sw, N, C, IH, IW, OH, OW, KH, KW, stride,
                       pad, Kpad, col, stream);
}

size_t rbvae_col2im_ws_floats(void) { return 5 * 4096; }
int rbvae_col2im_nparts(long n_out) { return grid_for(n_out, 256, 4096); }

static bool col2im_pix_ok(bool small, int Cout) {
    constexpr int pix_on = 1;
    return pix_on && small && Cout <= 4;
}

/* 1 when rbvae_col2im_sigmoid (with target, ws and a 16-byte aligned dpre) also leaves the per-block column sums
 * of dpre at ws[nparts + 4*b + c] (the pixel-major kernel: Cout <= 4 and fewer than 2^31 outputs). */
extern "C" int rbvae_col2im_has_dcol(int N, int IH, int IW, int ldy, int OH, int OW, int Cout) {
    const long tot = (long)N * OH * OW * Cout;
    const bool small = tot < (1l << 31) - (1l << 20) && (long)N * IH * IW * ldy < (1l << 31);
    return col2im_pix_ok(small, Cout) ? 1 : 0;
}

static int col2im_impl(int dtype, const void* Y, int ldy, const float* bias, int N, int IH, int IW, int OH,
                       int OW, int Cout, int KH, int KW, int pad, float* xr, const float* target, FrameMap tfm,
                       float* sse_mean, float* ws, float* dpre, float gscale, const float* gscale_dev,
                       void* stream) {
    RBVAE_CHECK_ARG(Y && xr && N > 0 && Cout > 0 && ldy >= KH * KW * Cout, "col2im_sigmoid: bad arguments");
    RBVAE_CHECK_ARG(tfm.d1 == 0 || (tfm.d2 > 0 && tfm.d1 % tfm.d2 == 0 && tfm.s0 >= 0 && tfm.s1 >= 0),
                    "col2im_sigmoid: frame map d1=%d d2=%d", tfm.d1, tfm.d2);
    RBVAE_CHECK_ARG(!sse_mean || (target && ws), "col2im_sigmoid: sse_mean needs target and ws");
    RBVAE_CHECK_ARG(!dpre || target, "col2im_sigmoid: dpre needs target");
    const long tot = (long)N * OH * OW * Cout;
    RBVAE_CHECK_ARG(dtype == RBVAE_F32 || dtype == RBVAE_BF16, "col2im_sigmoid: dtype %d", dtype);
    const int nb = grid_for(tot, 256, 4096);
    // sse_mean == NULL with ws given: leave the nb per-block partial sums in ws (rbvae_combine_losses finishes them)
    float* sws = (sse_mean || ws) && target ? ws : nullptr;
    hipStream_t st = (hipStream_t)stream;
    const bool small = tot < (1l << 31) - (1l << 20) && (long)N * IH * IW * ldy < (1l << 31) &&
                       frame_span(tfm, N) + (long)Cout * OH * OW < (1l << 31);
    const bool pix = col2im_pix_ok(small, Cout) && (!dpre || (uintptr_t)dpre % 16 == 0);
#define RBVAE_COL2IM(TT, II)                                                                                       \
    hipLaunchKernelGGL((col2im_sigmoid_k<TT, II>), dim3(nb), dim3(256), 0, st, (const TT*)Y, ldy, bias, N, IH, IW, OH, \
                       OW, Cout, KH, KW, pad, xr, target, tfm, sws, dpre, gscale, gscale_dev)
#define RBVAE_COL2IM_PIX(TT)                                                                                    \
    hipLaunchKernelGGL((col2im_sigmoid_pix_k<TT>), dim3(nb), dim3(256), 0, st, (const TT*)Y, ldy, bias, N, IH, IW, OH, \
                       OW, Cout, KH, KW, pad, xr, target, tfm, sws, dpre, gscale, gscale_dev)
    if (pix) { if (dtype == RBVAE_F32) RBVAE_COL2IM_PIX(float); else RBVAE_COL2IM_PIX(bf16_t); }
    else if (dtype == RBVAE_F32) { if (small) RBVAE_COL2IM(float, unsigned); else RBVAE_COL2IM(float, long); }
    else { if (small) RBVAE_COL2IM(bf16_t, unsigned); else RBVAE_COL2IM(bf16_t, long); }
#undef RBVAE_COL2IM_PIX
#undef RBVAE_COL2IM
    if (sse_mean)
        hipLaunchKernelGGL(sum_partials_k, dim3(1), dim3(1024), 0, st, ws, nb, 1.0f / (float)tot, sse_mean, 0);
    RBVAE_CHECK_LAUNCH("col2im_sigmoid");
    return RBVAE_OK;
}

int rbvae_col2im_sigmoid(int dtype, const void* Y, int ldy, const float* bias, int N, int IH, int IW, int OH,
                         int OW, int Cout, int KH, int KW, int pad, float* xr, const float* target,
                         float* sse_mean, float* ws, float* dpre, float gscale, const float* gscale_dev,
                         void* stream) {
    return col2im_impl(dtype, Y, ldy, bias, N, IH, IW, OH, OW, Cout, KH, KW, pad, xr, target,
                       FrameMap{0, 0, 0, 0, (long)Cout * OH * OW}, sse_mean, ws, dpre, gscale, gscale_dev, stream);
}

int rbvae_col2im_sigmoid_frames(int dtype, const void* Y, int ldy, const float* bias, int N, int IH, int IW, int OH,
                                int OW, int Cout, int KH, int KW, int pad, float* xr, const float* target, int fd1,
                                int fd2, long fs0, long fs1, long fs2, float* sse_mean, float* ws, float* dpre,
                                float gscale, const float* gscale_dev, void* stream) {
    return col2im_impl(dtype, Y, ldy, bias, N, IH, IW, OH, OW, Cout, KH, KW, pad, xr, target,
                       FrameMap{fd1, fd2, fs0, fs1, fs2}, sse_mean, ws, dpre, gscale, gscale_dev, stream);
}

// total = recon + beta*kl + alpha*pair, with recon finished from the col2im kernel's partial sums
__global__ __launch_bounds__(1024) void combine_losses_k(const float* __restrict__ sse_ws, int nparts, float inv_n,
                                                         const float* __restrict__ recon_in,
                                                         const float* __restrict__ kl, int kl_parts, float kl_scale,
                                                         const float* __restrict__ pair, int pair_parts,
                                                         float w_sim, float w_dis,
                                                         float beta, float alpha, float* __restrict__ out4,
                                                         unsigned long long* __restrict__ step_dev, double lr,
                                                         const double* __restrict__ lr_dev,
                                                         double b1, double b2, float* __restrict__ hyper) {
    __shared__ float red[16];
    if (step_dev && threadIdx.x == 64) {
        if (lr_dev) lr = lr_dev[0];
        // the optimiser's bias corrections for the step that follows (what adam_hyper_k does as its own launch)
        const unsigned long long s1 = step_dev[0] + 1;
        step_dev[0] = s1;
        const double st = (double)s1;
        hyper[0] = (float)(lr / (1.0 - pow(b1, st)));
        hyper[1] = (float)sqrt(1.0 - pow(b2, st));
    }
    float recon;
    if (sse_ws) {
        float a = 0.f;
        for (int i = threadIdx.x; i < nparts; i += 1024) a += sse_ws[i];
        recon = block_sum(a, red) * inv_n;
    } else {
        recon = recon_in[0];
    }
    float k;
    if (kl_parts > 0) {
        float a = 0.f;
        for (int i = threadIdx.x; i < kl_parts; i += 1024) a += kl[i];
        k = block_sum(a, red) * kl_scale;
    } else {
        k = kl[0];
    }
    float pr;
    if (pair_parts > 0) {
        // rbvae_contrast_term_fused's per-block (sum d^2, sum max(1-d,0)^2) pairs
        float a0 = 0.f, a1 = 0.f;
        for (int i = threadIdx.x; i < pair_parts; i += 1024) { a0 += pair[2 * i]; a1 += pair[2 * i + 1]; }
        const float s0 = block_sum(a0, red);
        const float s1 = block_sum(a1, red);
        pr = w_sim * s0 + w_dis * s1;
    } else {
        pr = pair[0];
    }
    if (threadIdx.x == 0) {
        out4[0] = recon + beta * k + alpha * pr;
        out4[1] = recon;
        out4[2] = k;
        out4[3] = pr;
    }
}

int rbvae_combine_losses(const float* sse_ws, int nparts, float inv_n, const float* recon, const float* kl,
                         int kl_parts, float kl_scale, const float* pair, int pair_parts, float w_sim, float w_dis,
                         float beta, float alpha, float* out4, unsigned long long* step_dev, double lr,
                         const double* lr_dev, double beta1, double beta2, float* hyper_ws, void* stream) {
    RBVAE_CHECK_ARG(kl && pair && out4 && (sse_ws || recon) && kl_parts >= 0 && pair_parts >= 0, "combine_losses: bad arguments");
    RBVAE_CHECK_ARG(!step_dev || hyper_ws, "combine_losses: step_dev needs hyper_ws");
    hipLaunchKernelGGL(combine_losses_k, dim3(1), dim3(1024), 0, (hipStream_t)stream, sse_ws, nparts, inv_n, recon, kl,
                       kl_parts, kl_scale, pair, pair_parts, w_sim, w_dis, beta, alpha, out4, step_dev, lr, lr_dev, beta1,
                       beta2, hyper_ws);
    RBVAE_CHECK_LAUNCH("combine_losses");
    return RBVAE_OK;
}

int rbvae_sigmoid_bwd_nhwc(const float* g_nchw, const float* xr_nchw, float* dpre_nhwc, int N, int C, int H, int W,
                           void* stream) {
    RBVAE_CHECK_ARG(g_nchw && xr_nchw && dpre_nhwc && N > 0 && C > 0 && H > 0 && W > 0, "sigmoid_bwd_nhwc: bad arguments");
    const long tot = (long)N * C * H * W;
    hipLaunchKernelGGL(sigmoid_bwd_nhwc_k, dim3(grid_for(tot)), dim3(256), 0, (hipStream_t)stream, g_nchw, xr_nchw,
                       dpre_nhwc, N, C, H, W);
    RBVAE_CHECK_LAUNCH("sigmoid_bwd_nhwc");
    return RBVAE_OK;
}

static int skinny_impl(int dtype, const void* A, const void* B, const float* bias, float* out, int M, int Nc,
                       int K, int lda, int ldb, int ldo, int ksplit, void* stream) {
    RBVAE_CHECK_ARG(A && B && out && M > 0 && Nc > 0 && K > 0 && ksplit >= 1, "skinny_linear: bad arguments");
    RBVAE_CHECK_ARG(dtype == RBVAE_F32 || dtype == RBVAE_BF16, "skinny_linear: dtype %d", dtype);
    const int ES = dtype == RBVAE_F32 ? 4 : 2;
    const int KS = dtype == RBVAE_F32 ? 16 : 32;
    RBVAE_CHECK_ARG(K % KS == 0, "skinny_linear: K=%d must be a multiple of %d", K, KS);
    RBVAE_CHECK_ARG((lda * ES) % 16 == 0 && (ldb * ES) % 16 == 0 && lda >= K && ldb >= K && ldo >= Nc,
                    "skinny_linear: leading dimensions");
    RBVAE_CHECK_ARG(((uintptr_t)A | (uintptr_t)B) % 16 == 0, "skinny_linear: operands must be 16-byte aligned");
    RBVAE_CHECK_ARG(K % (ksplit * KS) == 0, "skinny_linear: K=%d not divisible into %d parts of whole %d-steps", K, ksplit, KS);
    dim3 grid(cdiv(M, 16), cdiv(Nc, 16), ksplit);
    const int kper = K / ksplit;
    if (dtype == RBVAE_F32)
        hipLaunchKernelGGL(skinny_linear_k<float>, grid, dim3(512), 0, (hipStream_t)stream, (const float*)A,
                           (const float*)B, bias, out, M, Nc, K, lda, ldb, ldo, kper);
    else
        hipLaunchKernelGGL(skinny_linear_k<bf16_t>, grid, dim3(512), 0, (hipStream_t)stream, (const bf16_t*)A,
                           (const bf16_t*)B, bias, out, M, Nc, K, lda, ldb, ldo, kper);
    RBVAE_CHECK_LAUNCH("skinny_linear");
    return RBVAE_OK;
}

int rbvae_skinny_linear(int dtype, const void* A, const void* B, const float* bias, float* out, int M, int Nc,
                        int K, int lda, int ldb, int ldo, void* stream) {
    return skinny_impl(dtype, A, B, bias, out, M, Nc, K, lda, ldb, ldo, 1, stream);
}

int rbvae_skinny_linear_parts(int dtype, const void* A, const void* B, const float* bias, float* out_parts, int M,
                              int Nc, int K, int lda, int ldb, int ldo, int ksplit, void* stream) {
    return skinny_impl(dtype, A, B, bias, out_parts, M, Nc, K, lda, ldb, ldo, ksplit, stream);
}

// hyper[0] = lr / (1 - b1^step), hyper[1] = sqrt(1 - b2^step) from a DEVICE step counter, which this
// kernel advances first (so a graph-replayed step needs no separate counter launch)
__global__ void adam_hyper_k(unsigned long long* __restrict__ step_dev, double lr, double b1, double b2,
                             float* __restrict__ hyper) {
    const unsigned long long s1 = step_dev[0] + 1;
    step_dev[0] = s1;
    const double st = (double)s1;
    hyper[0] = (float)(lr / (1.0 - pow(b1, st)));
    hyper[1] = (float)sqrt(1.0 - pow(b2, st));
}

// Batch gather of the device-resident latent table: out[r] = table[plan[batch][r]], 16 bytes per lane.
__global__ __launch_bounds__(256) void gather_frames_k(const float4* __restrict__ table, const long* __restrict__ plan,
                                                       int rows, int n_batches,
                                                       const unsigned long long* __restrict__ counter, long table_rows,
                                                       int vec_per_row, float4* __restrict__ out) {
    const int r = blockIdx.y;
    const long b = counter ? (long)(counter[0] % (unsigned long long)n_batches) : 0;
    long src = plan[b * rows + r];
    if (src < 0 || src >= table_rows) src = 0;       // set_data() validated the plan; never read outside the table
    const float4* s = table + src * vec_per_row;
    float4* d = out + (long)r * vec_per_row;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < vec_per_row; i += gridDim.x * 256) d[i] = s[i];
}

int rbvae_gather_frames(const float* table, long table_rows, const long* plan, int rows, int n_batches,
                        const unsigned long long* counter_dev, long frame_elems, float* out, void* stream) {
    RBVAE_CHECK_ARG(table && plan && out && rows > 0 && n_batches > 0 && table_rows > 0, "gather_frames: bad arguments");
    RBVAE_CHECK_ARG(frame_elems > 0 && frame_elems % 4 == 0 && frame_elems / 4 < (1l << 31),
                    "gather_frames: frame of %ld floats (must be a multiple of 4)", frame_elems);
    RBVAE_CHECK_ARG(((uintptr_t)table | (uintptr_t)out) % 16 == 0, "gather_frames: pointers must be 16-byte aligned");
    const int vec = (int)(frame_elems / 4);
    const int gx = vec >= 4096 ? 8 : (vec >= 1024 ? 4 : 1);
    hipLaunchKernelGGL(gather_frames_k, dim3(gx, rows), dim3(256), 0, (hipStream_t)stream, (const float4*)table, plan,
                       rows, n_batches, counter_dev, table_rows, vec, (float4*)out);
    RBVAE_CHECK_LAUNCH("gather_frames");
    return RBVAE_OK;
}

int rbvae_adam_step(float* w, const float* g, float* m, float* v, long n, double lr, double beta1, double beta2,
                    double eps, int step, float gscale, unsigned long long* step_dev, float* hyper_ws,
                    void* stream) {
    RBVAE_CHECK_ARG(w && g && m && v && n > 0, "adam_step: bad arguments");
    RBVAE_CHECK_ARG(step_dev ? hyper_ws != nullptr : (step >= 1 || hyper_ws),
                    "adam_step: needs step >= 1, or step_dev + hyper_ws, or a prepared hyper_ws");
    double bc1 = 1.0, bc2 = 1.0;
    if (!step_dev && hyper_ws) {
        // hyper_ws was prepared by rbvae_combine_losses (device step counter already advanced)
    } else if (step_dev) {
        hipLaunchKernelGGL(adam_hyper_k, dim3(1), dim3(1), 0, (hipStream_t)stream, step_dev, lr, beta1, beta2,
                           hyper_ws);
    } else {
        bc1 = 1.0 - pow(beta1, (double)step);
        bc2 = 1.0 - pow(beta2, (double)step);
    }
    hipLaunchKernelGGL(adam_k, dim3(grid_for(n, 256, 2048)), dim3(256), 0, (hipStream_t)stream, w, g, m, v, n,
                       (float)(lr / bc1), (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps,
                       (float)sqrt(bc2), gscale, hyper_ws ? hyper_ws : (const float*)nullptr);
    RBVAE_CHECK_LAUNCH("adam_step");
    return RBVAE_OK;
}

}  // extern "C"
