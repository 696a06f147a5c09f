/* librbvae_hip -- C ABI of the MI355X (gfx950) RBVAE hot path.
 *
 * The reference (matt-suncy/symbols-from-video) has no FFI: its hot path is the
 * torch ops issued by Seq2SeqBinaryVAE.forward/encode and by the trainer's loss
 * functions.  Each entry point below replaces one of those op groups; the
 * reference call site it stands in for is cited as file:line (relative to the
 * reference root).  INTEGRATION.md shows the ctypes stub a maintainer adds.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer valid for the duration of the call in
 *    stream order; the library never allocates, frees or retains them;
 *  - `stream` is a hipStream_t (NULL = default stream); calls are asynchronous
 *    and never synchronise;
 *  - return value 0 = ok, negative = RBVAE_E_*; rbvae_last_error() gives a
 *    thread-local message; nothing throws or exits across the boundary;
 *  - `dtype` selects the activation/weight storage type of the conv/linear
 *    kernels: RBVAE_F32 (exact-fp32 parity mode, f32 MFMA) or RBVAE_BF16
 *    (bf16 storage, f32 accumulation, bf16 MFMA).  LSTM, binarise and the loss
 *    reductions are always f32;
 *  - activations are NHWC ("pixel rows of C channels"), weights are the packed
 *    layouts written by the rbvae_run_jobs pack jobs (kinds 0 and 3, see that entry
 *    point) from the reference-layout f32 parameters.
 * The hardware-map probes and phase-stamp hooks of debug builds are NOT part of this
 * ABI: include/rbvae_dbg.h, librbvae_dbg.so.
 */
#ifndef RBVAE_HIP_H
#define RBVAE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RBVAE_OK 0
#define RBVAE_E_INVALID (-1)     /* bad argument (shape, alignment, null pointer) */
#define RBVAE_E_LAUNCH (-2)      /* HIP reported a launch error */
#define RBVAE_E_UNSUPPORTED (-3) /* configuration outside what the kernels cover */

#define RBVAE_F32 0
#define RBVAE_BF16 1

int rbvae_version(void);
/* counter[0] += inc: the device step counter the dropout / noise hashes mix in (graph replay safe). */
int rbvae_counter_add(unsigned long long* counter, unsigned long long inc, void* stream);
const char* rbvae_last_error(void);

/* ---- binarise + KL -------------------------------------------------------
 * binary_concrete_logits (models/percep_RBVAE/percep_RBVAE_model.py:17-44; triplet
 * variant triplet_RBVAE_model.py:18-45 = noise_ratio 1; simple variant
 * simple_RBVAE_model.py:17-44 = noise_eps 1e-10) fused with kl_binary_concrete
 * applied to the SAMPLE z (models/percep_RBVAE/percep_RBVAE_train.py:52-76 as
 * called at :528).  U is the uniform noise the reference draws with torch.rand
 * on the host (:33); the caller supplies it so codes are reproducible.
 *   y_soft = sigmoid((h + r*(log(U+e) - log(1-U+e))) / tau)
 *   z      = hard ? (y_soft > 0.5) : y_soft
 *   kl_mean[0] = mean_rows sum_L KL(clamp(sigmoid(z)) || Bernoulli(p))
 * kl_mean may be NULL (encode path).  U may be NULL: the kernel then draws 24-bit uniforms from a
 * counter hash of (seed + *seed_dev, element index) -- device-side noise for the fused trainer. */
int rbvae_binarize_kl_fwd(const float* h, const float* U, float* y_soft, float* z, float* kl_mean,
                          int rows, int L, float tau, float noise_ratio, float noise_eps, int hard,
                          float kl_p, float kl_eps, int kl_clamp, unsigned long long seed,
                          const unsigned long long* seed_dev, void* stream);
/* The same over many workgroups (256 elements each): kl_parts[b] = sum over block b of the per-element KL terms,
 * b < rbvae_binarize_kl_nparts(rows, L); kl_mean = sum(kl_parts) / rows is left to rbvae_combine_losses.  The
 * one-workgroup form above is ALU-latency bound at the trainer's 256 x 32 logits; this one is not. */
int rbvae_binarize_kl_nparts(int rows, int L);
int rbvae_binarize_kl_fwd_parts(const float* h, const float* U, float* y_soft, float* z, float* kl_parts,
                                int rows, int L, float tau, const float* tau_dev, float noise_ratio, float noise_eps, int hard,
                                float kl_p, float kl_eps, int kl_clamp, unsigned long long seed,
                                const unsigned long long* seed_dev, void* stream);
/* dh (+)= (g_z + kl_weight * gscale * dKL/dz) * y_soft*(1-y_soft)/tau (straight-through when hard).
 * g_z may be NULL; gscale_dev (device scalar, may be NULL = 1) multiplies kl_weight.
 * tau_dev (here and in every entry point that has it): when not NULL the kernel reads the temperature from that
 * device float instead of `tau`, so a captured HIP graph follows the reference's annealing schedule
 * (percep_RBVAE_train.py:424-437) without being re-captured. */
int rbvae_binarize_kl_bwd(const float* g_z, const float* y_soft, const float* z, float* dh, int accumulate,
                          int rows, int L, float tau, const float* tau_dev, float kl_weight, const float* gscale_dev,
                          float kl_p, float kl_eps, int kl_clamp, void* stream);

/* kl_binary_concrete as a free function (percep_RBVAE_train.py:52-76; simple
 * variant simple_RBVAE_train.py:45-68 = clamp 0, eps 1e-10). */
int rbvae_kl_fwd(const float* q_logits, float* out_mean, int rows, int L, float p, float eps, int clamp,
                 void* stream);
int rbvae_kl_bwd(const float* q_logits, float* dq, int rows, int L, float p, float eps, int clamp,
                 float scale, const float* gscale_dev, void* stream);

/* ---- pairwise-distance losses -------------------------------------------
 * contrast_loss 'euclidean' branch (percep_RBVAE_train.py:79-107):
 *   d = ||x1 - x2 + eps||_2 over L;  label 0: mean(d^2);  label 1: mean(max(margin-d,0)^2)
 * Rows are addressed as base + row*stride so h_seq[:, s] slices need no copy. */
int rbvae_pairdist_fwd(const float* x1, const float* x2, long stride1, long stride2, int rows, int L,
                       int label, float margin, float eps, float* out_mean, void* stream);
int rbvae_pairdist_bwd(const float* x1, const float* x2, long stride1, long stride2, int rows, int L,
                       int label, float margin, float eps, float scale, const float* gscale_dev,
                       float* dx1, float* dx2, long dstride1, long dstride2, int accumulate, void* stream);
/* contrast_loss 'cosine' branch (percep_RBVAE_train.py:94-96; never taken by the reference's trainers):
 *   d = 1 - x1.x2 / (max(|x1|, eps) * max(|x2|, eps)),  eps = 1e-8 (torch.nn.functional.cosine_similarity), then as above. */
int rbvae_paircos_fwd(const float* x1, const float* x2, long stride1, long stride2, int rows, int L, int label,
                      float margin, float eps, float* out_mean, void* stream);
int rbvae_paircos_bwd(const float* x1, const float* x2, long stride1, long stride2, int rows, int L, int label,
                      float margin, float eps, float scale, const float* gscale_dev, float* dx1, float* dx2, long dstride1,
                      long dstride2, void* stream);
/* The trainer's whole contrastive term in one launch (percep_RBVAE_train.py:534-543):
 *   mean_{b,t} d(h0,h1)^2 + 1/(T-1) sum_s mean_b max(1 - d(h0[:,s],h0[:,s+1]),0)^2, eps 1e-6.
 * h0,h1: [B,T,L] contiguous.  bwd WRITES dh0,dh1 (scale * d term/dh). */
int rbvae_contrast_term_fwd(const float* h0, const float* h1, int B, int T, int L, float* out, void* stream);
int rbvae_contrast_term_bwd(const float* h0, const float* h1, int B, int T, int L, float scale,
                            const float* gscale_dev, float* dh0, float* dh1, void* stream);
/* Value and gradient of that term in ONE many-workgroup launch: parts[2k], parts[2k+1] (k < contrast_term_nparts)
 * = per-block sums of d(h0,h1)^2 and of max(1 - d(h0[:,s],h0[:,s+1]),0)^2; the term is
 * sum(parts[2k])/(B*T) + sum(parts[2k+1])/(B*(T-1)) (rbvae_combine_losses finishes it); dh0/dh1 as the _bwd form. */
int rbvae_contrast_term_nparts(int B, int T);
int rbvae_contrast_term_fused(const float* h0, const float* h1, int B, int T, int L, float scale,
                              const float* gscale_dev, float* parts, float* dh0, float* dh1, void* stream);
/* F.triplet_margin_loss(p=2, eps, swap) (triplet_RBVAE_train.py:82-96) on strided rows. */
int rbvae_triplet_fwd(const float* a, const float* p, const float* n, long sa, long sp, long sn, int rows, int L,
                      float margin, float eps, int swap, float* out_mean, void* stream);
int rbvae_triplet_bwd(const float* a, const float* p, const float* n, long sa, long sp, long sn, int rows, int L,
                      float margin, float eps, int swap, float scale, const float* gscale_dev,
                      float* da, float* dp, float* dn, long dsa, long dsp, long dsn, int accumulate, void* stream);
/* The triplet trainer's term (triplet_RBVAE_train.py:461-468): anchor h0[:,s], positive
 * h1[:,s], negative h0[:,s+1], averaged over s < T-1; eps 1e-8, swap on. bwd WRITES. */
int rbvae_triplet_term_fwd(const float* h0, const float* h1, int B, int T, int L, float margin, float* out,
                           void* stream);
int rbvae_triplet_term_bwd(const float* h0, const float* h1, int B, int T, int L, float margin, float scale,
                           const float* gscale_dev, float* dh0, float* dh1, void* stream);

/* recon_loss = F.mse_loss (percep_RBVAE_train.py:32-33).  ws: >= rbvae_mse_ws_floats(n) floats. */
size_t rbvae_mse_ws_floats(long n);
int rbvae_mse_fwd(const float* a, const float* b, long n, float* out_mean, float* ws, void* stream);
int rbvae_mse_bwd(const float* a, const float* b, long n, float scale, const float* gscale_dev, float* da,
                  void* stream);

/* ---- row-gather GEMM on the matrix cores ---------------------------------------
 * Out[orow(m)][n] = epi( sum_{taps j} sum_{k<Kc} A[arow(m,j)][k] * W[n][widx_j][k] ), NHWC rows.
 * Replaces nn.Conv2d(k,2,1) forward (percep_RBVAE_model.py:51-57), nn.ConvTranspose2d(k,2,1,op)
 * forward (:76-82, as the conv's input gradient over 4 output-parity classes), the
 * backward-data passes of both (autograd of percep_RBVAE_train.py:552) and, with one tap,
 * the wide Linear products (:61,:74).
 *   m -> (n, a, b) over Nimg x TH x TW;  A pixel (a*sa+dh_j, b*sa+dw_j) of an IH x IW grid (zero
 *   outside);  Out pixel (a*so+oh0, b*so+ow0) of an OH x OW grid;  grid.z = parity class.
 *   class_desc is a HOST int array: per class [ntaps, oh0, ow0, ntaps x (widx, dh, dw)].
 *   epilogue: +bias, relu, *scale, dropout (drop_mode 1: counter hash of (seed, element index),
 *   2: explicit u8 keep-mask [rows][Nout]), + addend (residual, same indexing as Out; LDM ResnetBlock /
 *   AttnBlock skip connections, ldm/modules/diffusionmodules/model.py:141,202), then zero where gate <= 0 (saved activation:
 *   ReLU/dropout backward).  zero_page: >= 128 zero bytes.  Kc % (128/sizeof T) == 0, Nout % 8 == 0.
 *   colsum_ws (optional, [nclass * ceil(rows/128)][Nout] f32): per-tile column sums of the stored
 *   values -- the bias gradient, finished by rbvae_reduce_rows. */
int rbvae_gather_gemm(int dtype, const void* A, const void* W, void* Out, const float* bias, const void* gate,
                      const void* mask, const void* addend, const void* zero_page, int Nimg, int IH, int IW, int TH, int TW, int sa,
                      int OH, int OW, int so, int Kc, int Nout, int lda, int ldo, int taps_total, int nclass,
                      const int* class_desc, int relu, int drop_mode, float drop_p, float scale,
                      unsigned long long seed, const unsigned long long* seed_dev, float* colsum_ws,
                      void* stream);

/* ---- 3x3 stride-2 pad-1 convolution with the input patch resident in LDS (bf16) -----------------------------------
 * Out[n][r][c][co] = epi( sum_{kh,kw,ci} A[n][2r + kh - 1][2c + kw - 1][ci] * W[co][kh*3 + kw][ci] ), NHWC rows, IH and IW even,
 * OH = IH / 2, OW = IW / 2: rbvae_gather_gemm's result for the one-class descriptor of nn.Conv2d(c, c, 3, 2, 1)
 * (percep_RBVAE_model.py:54-57; the conv-form input gradient of nn.ConvTranspose2d(c, c, 3, 2, 1, 1), :76-81, autograd as run by
 * percep_RBVAE_train.py:552; the LDM encoder's Downsample, ldm/modules/diffusionmodules/model.py:60-79), with the same
 * epilogue element for element (+bias, relu, *scale, dropout by key / mask with the same element indices, ReLU gate) --
 * but a workgroup stages the 17 x 33 input patch of its 8 x 16 output pixels ONCE per 32-channel slice for all nine taps
 * (9.7 KB through the CU's L2 -> LDS path per MFLOP at 256 output channels per workgroup instead of 15.2).
 * rbvae_conv3x3s2_halo_ok: output channels per workgroup (256 / 128) when covered (bf16, Kc % 32 == 0, Nout % 128 == 0,
 * even IH / IW, operands below 2 GiB), else 0.  colsum_ws (optional): [rbvae_conv3x3s2_halo_colsum_rows(..)][Nout] f32
 * column sums of the stored values per pixel tile -- the bias gradient, finished by rbvae_reduce_rows / a row-reduce job. */
int rbvae_conv3x3s2_halo_ok(int dtype, int Nimg, int IH, int IW, int Kc, int Nout);
int rbvae_conv3x3s2_halo_colsum_rows(int Nimg, int IH, int IW);
int rbvae_conv3x3s2_halo(int dtype, const void* A, const void* W, void* Out, const float* bias, const void* gate, const void* mask,
                         int Nimg, int IH, int IW, int Kc, int Nout, int lda, int ldo, int relu, int drop_mode, float drop_p,
                         float scale, unsigned long long seed, const unsigned long long* seed_dev, float* colsum_ws, void* stream);

/* ---- the K = 64 / 128 products around the latent bottleneck (bf16) -------------------------------
 * Out[M][ldo] = A[M][lda] (K used columns) * W[N][K]^T (+ bias[N]) for a few hundred rows and thousands of columns:
 * the decoder's fc forward (Linear(latent_dim -> C3*h3*w3), percep_RBVAE_model.py:74, on the zero-padded codes) and
 * the input gradient of the encoder's fc (autograd of :61).  The arithmetic of rbvae_gather_gemm with one tap,
 * element for element; operands go from global memory straight into the MFMA layout (no tables, no LDS ring).
 * colsum_ws (may be NULL): [ceil(M/128)][N] column sums of the stored values per 128-row tile (rbvae_gather_gemm's
 * layout; the bias gradient of the layer below).  rbvae_fc_gemm_ok: 1 when the shape is covered (bf16, K == 64 or
 * 128 = latent_dim padded to the GEMMs' 64-column slices, N % 16 == 0). */
int rbvae_fc_gemm_ok(int dtype, int M, int K, int N, int lda, int ldo);
int rbvae_fc_gemm(int dtype, const void* A, const void* W, void* Out, const float* bias, float* colsum_ws, int M, int K,
                  int N, int lda, int ldo, void* stream);

/* ---- weight-gradient GEMM ---------------------------------------------------------
 * dW[ks][co][t][ci] = sum over K-slice ks of Dy[p][co] * In[idx[t][p]][ci]  (f32 slabs, one per
 * K-slice; sum them with rbvae_permute_reduce).  idx = rbvae_conv_gather_index table or NULL
 * (identity, 1 tap: Linear).  in_rows = rows of In: an index outside [0, in_rows) reads the zero row, so a
 * wrong table can give wrong sums but never an out-of-bounds access.  Autograd of the Conv2d/ConvTranspose2d/Linear weights
 * (percep_RBVAE_model.py:51-61,74-82). */
int rbvae_conv_gather_index(int* idx, int Nimg, int IH, int IW, int OH, int OW, int KH, int KW, int stride,
                            int pad, void* stream);
int rbvae_wgrad_gemm(int dtype, const void* Dy, const void* In, float* dW_slabs, const int* idx,
                     const void* zero_page, int P, int in_rows, int Co, int Ci, int ldy, int ldi, int taps, int ksplit,
                     void* stream);

/* ---- weight gradient of the 3x3 stride-2 pad-1 convolutions, nine taps per workgroup (bf16) ----------------------
 * dW[ks][a][t][b] = sum over the K-slice's pixels p = (n, r, c) of S[p][a] * G[(n, 2r + kh - 1, 2c + kw - 1)][b],
 * t = 3 kh + kw, rows outside the image zero: rbvae_wgrad_gemm's sum for idx = rbvae_conv_gather_index(.., 3, 3, 2, 1)
 * and taps = 9, in the same slab layout (the same reduction jobs follow), with S and the 9 x 17 patch of G around a
 * 4 x 8 pixel block fetched once for all nine taps instead of once per tap.  Conv2d(c, c, 3, 2, 1) weights: S = the output
 * gradient [Nimg*OH*OW][lds], G = the layer's input [Nimg*2OH*2OW][ldg] (percep_RBVAE_model.py:51-57);
 * ConvTranspose2d(c, c, 3, 2, 1, 1) weights: S = the layer's input, G = its output gradient (:76-81); autograd as run by
 * percep_RBVAE_train.py:552.  Covered (rbvae_wgrad3x3s2_halo_ok): bf16, Ca and Cb multiples of 64.  K-slices are runs
 * of rbvae_wgrad3x3s2_halo_blocks(..) / ksplit pixel blocks; grid = (Ca/64) * (Cb/64) * ksplit workgroups (ksplit rounded up
 * to a multiple of 8: one K-slice group per XCD). */
int rbvae_wgrad3x3s2_halo_ok(int dtype, int Nimg, int OH, int OW, int Ca, int Cb);
int rbvae_wgrad3x3s2_halo_blocks(int Nimg, int OH, int OW);
int rbvae_wgrad3x3s2_halo(int dtype, const void* S, const void* G, float* dW_slabs, const void* zero_page, int Nimg, int OH,
                          int OW, int Ca, int Cb, int lds, int ldg, int ksplit, void* stream);

/* ---- the same sum for wide layers (Ca, Cb multiples of 128, bf16): the three taps of ONE kernel row per workgroup ------
 * A workgroup owns a 128 (a) x 128 (b) tile of the taps kw = 0, 1, 2 of one kh; per block of 64 low-resolution pixels the
 * [64][128] tile of S arrives once and of G the rows 2r + kh - 1 with their 2W + 1 columns (kw = 0 and kw = 2 share the odd
 * columns): 50 KB per 6.3 MFLOP instead of rbvae_wgrad_gemm's 96 KB (three taps, three workgroups).  Replaces
 * rbvae_wgrad_gemm for the 256-channel Conv2d(.., 3, 2, 1) / ConvTranspose2d(.., 3, 2, 1, 1) weights of
 * percep_RBVAE_model.py:54-57,76-81 (autograd as run by percep_RBVAE_train.py:552); same operands, same slab layout
 * [ksplit][Ca][9][Cb] and the same reduction jobs as rbvae_wgrad3x3s2_halo.  K-slices are balanced runs of the
 * rbvae_wgrad3x3s2_row_blocks(..) pixel blocks; grid = (Ca/128) * (Cb/128) * 3 * ksplit workgroups (rounded up to a multiple
 * of 8: every XCD takes a contiguous eighth of the (K-slice, tile) items). */
int rbvae_wgrad3x3s2_row_ok(int dtype, int Nimg, int OH, int OW, int Ca, int Cb);
int rbvae_wgrad3x3s2_row_blocks(int Nimg, int OH, int OW);
int rbvae_wgrad3x3s2_row(int dtype, const void* S, const void* G, float* dW_slabs, const void* zero_page, int Nimg, int OH,
                         int OW, int Ca, int Cb, int lds, int ldg, int ksplit, void* stream);

/* ---- layout helpers --------------------------------------------------------------
 * pack3:          out[i0*s0+i1*s1+i2*s2] = (T) in[i0][i1][i2]        (torch f32 weight -> packed T)
 * permute_reduce: out[i0][i1][i2] (+)= scale * sum_k in[k*slab+i0*s0+i1*s1+i2*s2]  (slabs -> torch grad)
 * cast_pad:       f32 [rows][L] -> T [rows][Lpad], zero padded
 * colsum:         out[C] (+)= scale * sum_p X[p][c]  (bias gradients); ws >= colsum_ws_floats */
int rbvae_pack3(int dtype, const float* in, void* out, int d0, int d1, int d2, long s0, long s1, long s2,
                void* stream);
int rbvae_permute_reduce(const float* in, int nslab, long slab_stride, float* out, int d0, int d1, int d2, long s0,
                         long s1, long s2, float scale, int accumulate, void* stream);
int rbvae_cast_pad(int dtype, const float* in, void* out, int rows, int L, int Lpad, void* stream);
int rbvae_reduce_rows(const float* ws, int rows, int C, float* out, float scale, int accumulate, void* stream);
/* first half of colsum: ws[ceil(P/rpb)][C] partial sums (rbvae_colsum_ws_floats(P, C) floats) */
int rbvae_colsum_partial(int dtype, const void* X, int P, int C, int ld, float* ws, void* stream);
size_t rbvae_colsum_ws_floats(int P, int C);
int rbvae_colsum(int dtype, const void* X, int P, int C, int ld, float* out, float* ws, float scale, int accumulate,
                 void* stream);

/* Batched layout jobs: njobs rows of 16 x int64 in DEVICE memory
 *   [type, src, dst, d0, d1, d2, s0, s1, s2, nslab, slab_stride, dtype, accumulate,
 *    scale (f32 bits, low word) | fast index (high word), inner, dst2]
 * type 0 = pack3, 1 = permute_reduce, 2 = reduce_rows (out[c] = scale*sum_k src[k*slab+c], c < d0*d1*d2),
 * 3 = conv weight pack: src f32 [d0=co][d1=ci][d2=kk<=16] -> dst [co][t][ci] and dst2 [ci][t][co] of `dtype`
 * (nn.Conv2d / nn.ConvTranspose2d weights, percep_RBVAE_model.py:51-57,76-82, in the two GEMM operand orders),
 * 4 = conv weight-gradient reduce: src = rbvae_wgrad_gemm's K-slice slabs [nslab][d0=co][d2=kk<=16][d1=ci] (f32,
 * ci % 4 == 0), summed in slab order into dst [co][ci][kk] (the torch layout of the weight; scale, accumulate honoured).
 * 5 = batch gather (rbvae_gather_frames as a job: src = table, dst = out, d0 = rows, d1 = n_batches, d2 = float4 per frame,
 * s0 = plan pointer, s1 = counter pointer or 0, s2 = table rows) -- shares the launch of the step's weight repack.
 * 6 / 7 = torch.optim.Adam update (percep_RBVAE_train.py:553) of one parameter tensor inside the job launch: src points
 * into the flat parameter buffer, `inner` holds a pointer to the optimiser context (8 x int64: w, g, m, v base pointers,
 * hyper pointer [lr/(1-b1^t), sqrt(1-b2^t)], then f32 pairs (1-b1, b2), (1-b2, eps), (gscale, 0)); kind 6 also writes the
 * tensor's packed copies from the new values: dst with strides (s0, s1, s2) in type dtype & 255 and optionally dst2 with
 * strides (nslab, slab, accumulate) in type dtype >> 8.  Kind 3 with `inner` set updates the conv weight rows it packs.
 * A table of these jobs is the optimiser step AND the weight repack of a training step in one launch.
 * fast: the logical index consecutive threads walk; inner != 0 (fast == 1, short d2): a thread walks d2 itself.
 * One launch (grid.y = job) replaces the per-tensor launches of a step. */
int rbvae_run_jobs(const void* jobs_dev, int njobs, int blocks_per_job, void* stream);
/* The same table launched with exactly the workgroups its jobs can use (a step's update launch: ~1 400 instead of 49 x 256).
 * rbvae_job_block_map (host): for the njobs rows at rows_host (host copy of the table) write map_host[4 * b] = (job, index of
 * workgroup b within its job, workgroups of the job <= max_blocks_per_job, 0) and return the number of workgroups (with
 * map_host == NULL: only count them); negative = error.  rbvae_run_jobs_sized: block_map_dev = that map in device memory
 * (16-byte aligned), total_blocks = its entries. */
int rbvae_job_block_map(const long* rows_host, int njobs, int max_blocks_per_job, int* map_host, int map_capacity);
int rbvae_run_jobs_sized(const void* jobs_dev, const void* block_map_dev, int total_blocks, void* stream);

/* im2col of a strided f32 image (element strides sn,sc,sh,sw) into col[N*OH*OW][Kpad], column
 * (kh*KW+kw)*C + c: the 3/4-channel first Conv2d (percep_RBVAE_model.py:51) and the last
 * ConvTranspose2d's backward run as plain GEMMs on it. */
int rbvae_im2col(int dtype, const float* src, long sn, long sc, long sh, long sw, int N, int C, int IH, int IW,
                 int OH, int OW, int KH, int KW, int stride, int pad, int Kpad, void* col, void* stream);
/* The first Conv2d(3x3, stride 2, pad 1; Cin <= 4 -> Nout <= 256) + bias + ReLU + Dropout (percep_RBVAE_model.py:51-53)
 * as ONE kernel, bf16: rbvae_im2col_frames (col [N*OH*OW][64] is written for rbvae_wgrad_gemm unless col == NULL) + the
 * single-slice rbvae_gather_gemm, with the patch gathered in LDS and results stored from the accumulators.  W is the
 * packed [Nout][64] im2col-order weight; x frames f32 [Cin][IH][IW] through the frame map (fd1 == 0: frame n at
 * n*fs2); drop_mode 0 / 1 with rbvae_gather_gemm's key and element indices. */
int rbvae_conv_first_fused_ok(int dtype, int Cin, int IH, int IW, int Nout, int N);
int rbvae_conv_first_fused(int dtype, const float* x, int fd1, int fd2, long fs0, long fs1, long fs2, const void* W,
                           const float* bias, const void* zero_page, void* col, void* out, int N, int Cin, int IH, int IW,
                           int Nout, int ldo, int relu, int drop_mode, float drop_p, float scale, unsigned long long seed,
                           const unsigned long long* seed_dev, void* stream);
/* Input gradient of the last ConvTranspose2d (autograd of percep_RBVAE_model.py:82; k3 s2 p1) as ONE kernel, bf16: the
 * 3x3 stride-2 convolution of dpre [N][OH][OW][Cout] f32 (rbvae_deconv_last_fused's d(loss)/d(pre-sigmoid)) with the packed
 * weight W [C1][64] (column (kh*3+kw)*Cout + co), times scale, kept where gate [rows][ldo] (the stored ReLU/dropout output
 * of the layer below) is > 0 -- rbvae_im2col + the single-slice rbvae_gather_gemm, bit-identical stored values.  col
 * [rows][64] receives the im2col rows (the last deconv's weight gradient reads them); colsum_ws
 * [rbvae_deconv_last_dgrad_blocks][C1] (optional) the per-workgroup column sums of the stored output (bias gradient of
 * the layer below).  rows = N * ceil(OH/2) * ceil(OW/2). */
int rbvae_deconv_last_dgrad_blocks(int dtype, int Cout, int OH, int OW, int C1, int N);
int rbvae_deconv_last_dgrad_fused(int dtype, const float* dpre, const void* W, const void* zero_page, void* col,
                                  const void* gate, void* out, int N, int Cout, int OH, int OW, int C1, int ldo, float scale,
                                  float* colsum_ws, void* stream);

/* Weight gradient of those two layers WITHOUT the im2col rows: dW[ks][co][k] = sum over the K-slice's output pixels p of
 * dY[p][co] * col(x)[p][k], k = (kh*3+kw)*Cin + ci zero padded to 64 -- rbvae_wgrad_gemm's sums over the [rows][64] im2col rows
 * (one tap), in its slab layout [ksplit][Nout][64], with the rows rebuilt in LDS from the 3/4-channel image instead of read
 * back (128 bytes per pixel for 12-16 bytes of image).  mode 0: x = the input frames through the frame map (as
 * rbvae_conv_first_fused), dY = the gradient at the first Conv2d's output (percep_RBVAE_model.py:51, autograd as run by
 * percep_RBVAE_train.py:552); mode 1: x = d(loss)/d(pre-sigmoid) [N][IH][IW][Cin] f32, dY = the stored activation in front of
 * the last ConvTranspose2d (:82).  With it the two kernels above take col = NULL.  K-slices are runs of
 * rbvae_wgrad_first_blocks(..) / ksplit blocks of 8 x 16 output pixels; grid = (Nout / 64) * ksplit workgroups. */
int rbvae_wgrad_first_blocks(int dtype, int Cin, int IH, int IW, int Nout, int N);
int rbvae_wgrad_first(int dtype, int mode, const float* x, int fd1, int fd2, long fs0, long fs1, long fs2, const void* dY,
                      float* dW_slabs, const void* zero_page, int N, int Cin, int IH, int IW, int Nout, int ldy, int ksplit,
                      void* stream);
/* Last ConvTranspose2d + Sigmoid (percep_RBVAE_model.py:82-83) fused with recon_loss
 * (percep_RBVAE_train.py:32-33): Y[(n,a,b)][t*Cout+co] = per-tap products; gathers them (col2im),
 * adds bias, applies sigmoid, writes x_recon NCHW f32; with target: sse_mean[0] = mse and
 * dpre[n][oh][ow][co] = gscale*gscale_dev*(xr-x)*xr*(1-xr).  ws >= col2im_ws_floats floats.
 * sse_mean == NULL with ws and target given: the rbvae_col2im_nparts(N*OH*OW*Cout) per-block partial sums of
 * squared error stay in ws for rbvae_combine_losses to finish (one launch fewer per training step). */
size_t rbvae_col2im_ws_floats(void);
int rbvae_col2im_nparts(long n_out);
/* 1 when the call (with target, ws and a 16-byte aligned dpre) also leaves per-block column sums of dpre -- the
 * last deconv's bias gradient -- at ws[nparts + 4*b + c], b < nparts, c < Cout (Cout <= 4, < 2^31 outputs). */
int rbvae_col2im_has_dcol(int N, int IH, int IW, int ldy, int OH, int OW, int Cout);
int rbvae_col2im_sigmoid(int dtype, const void* Y, int ldy, const float* bias, int N, int IH, int IW, int OH,
                         int OW, int Cout, int KH, int KW, int pad, float* xr, const float* target,
                         float* sse_mean, float* ws, float* dpre, float gscale, const float* gscale_dev,
                         void* stream);
/* The same with the frames addressed through a map instead of one stride: frame n starts at element
 * (n / fd1) * fs0 + ((n % fd1) / fd2) * fs1 + (n % fd2) * fs2 of src / target.  The fused trainer reads an item
 * batch [B][2][T][C][H][W] (percep_RBVAE_train.py:509-526: x_t = item[:, 0], x_t1 = item[:, 1]) as the 2B
 * sequences "all of view 0, then all of view 1" with fd1 = B*T, fd2 = T, fs0 = T*CHW, fs1 = 2*T*CHW, fs2 = CHW,
 * without first copying it into that order. */
int rbvae_im2col_frames(int dtype, const float* src, int fd1, int fd2, long fs0, long fs1, long fs2, long sc, long sh,
                        long sw, int N, int C, int IH, int IW, int OH, int OW, int KH, int KW, int stride, int pad,
                        int Kpad, void* col, void* stream);
int rbvae_col2im_sigmoid_frames(int dtype, const void* Y, int ldy, const float* bias, int N, int IH, int IW, int OH,
                                int OW, int Cout, int KH, int KW, int pad, float* xr, const float* target, int fd1,
                                int fd2, long fs0, long fs1, long fs2, float* sse_mean, float* ws, float* dpre,
                                float gscale, const float* gscale_dev, void* stream);
/* The same layer as ONE kernel (bf16, Cout <= 4, C1 % 64 == 0): per-tap products on the matrix cores from an
 * LDS-resident 9x17-pixel block of D2 [N*IH*IW][C1] and V [NYP][C1] (row = tap*Cout + co), f32 products kept in LDS,
 * then the gather / bias / sigmoid / squared error / d(loss)/d(pre) of rbvae_col2im_sigmoid_frames.  ws receives
 * `parts` squared-error sums and, behind them, parts x 4 column sums of dpre (parts = rbvae_deconv_last_fused_parts,
 * 0 = shape not covered: use rbvae_gather_gemm + rbvae_col2im_sigmoid_frames).  fd1 == 0: frames at n*fs2. */
int rbvae_deconv_last_fused_parts(int dtype, int N, int IH, int IW, int C1, int Cout);
int rbvae_deconv_last_fused(int dtype, const void* D2, const void* V, int NYP, const float* bias, const void* zero_page,
                            int N, int IH, int IW, int C1, int Cout, float* xr, const float* target, int fd1, int fd2,
                            long fs0, long fs1, long fs2, float* ws, float* dpre, float gscale, void* stream);
int rbvae_sigmoid_bwd_nhwc(const float* g_nchw, const float* xr_nchw, float* dpre_nhwc, int N, int C, int H, int W,
                           void* stream);
/* Linear with few outputs (percep_RBVAE_model.py:61 forward; :74 backward-data):
 * out[M][Nc] f32 = A[M][K] * B[Nc][K]^T + bias. */
int rbvae_skinny_linear(int dtype, const void* A, const void* B, const float* bias, float* out, int M, int Nc,
                        int K, int lda, int ldb, int ldo, void* stream);
/* K split over `ksplit` workgroup groups (the one-group form keeps 32 CUs busy at M = 256): slab q =
 * out_parts + q*M*ldo holds the partial product over K range q (+ bias in slab 0); the consumer sums the slabs. */
int rbvae_skinny_linear_parts(int dtype, const void* A, const void* B, const float* bias, float* out_parts, int M,
                              int Nc, int K, int lda, int ldb, int ldo, int ksplit, void* stream);

/* ---- stacked LSTM (percep_RBVAE_model.py:94-122) ------------------------------------
 * wblk: per layer w_ih[4L][L], w_hh[4L][L], b_ih[4L], b_hh[4L] (the reference's registration
 * order).  hs_all [layers+1][S][T][L]: slot 0 = input (caller fills), slot l+1 = layer l output.
 * Training also saves hprev/cs [layers][S][T][L] and acts [layers][S][T][4L].
 * wT (optional): transposed weight copies [layers][ih|hh][L][4L] for coalesced loads. */
int rbvae_lstm_fwd(const float* wblk, const float* wT, float* hs_all, float* hprev, float* acts, float* cs, int S,
                   int T, int L, int layers, void* stream);
int rbvae_lstm_bwd(const float* wblk, const float* wT, const float* acts, const float* cs, const float* g_top, float* dG,
                   float* dx, int S, int T, int L, int layers, void* stream);
/* Kernels by size: L <= 32 -- one thread group per LAYER with its weight rows in registers, cells along anti-diagonals
 * (T + layers - 1 dependent steps); 32 < L <= 128 (the reference's latent_dim 50 / 75 / 100, best_models.txt) -- layer by
 * layer, a gate row shared by two lanes (eight lanes per hidden unit in the backward pass), the input half W_ih x_t of
 * every time step computed in one batch off the dependent chain. */
/* Extended forms that take over the small kernels around the stacks (wavefront kernel only: L <= 32,
 * layers * roundup64(4L) <= 1024, else RBVAE_E_INVALID):
 *  - in_parts / g_top_parts: the stack input (forward) / top-layer gradient (backward) as `nparts` K-split slabs of
 *    the fc product that feeds them (rbvae_skinny_linear_parts; slab q at + q*part_stride floats), summed in slab
 *    order in the kernel's prologue; the forward also writes the sum to slot 0 of hs_all.  in_parts NULL /
 *    nparts 1: plain input as in rbvae_lstm_fwd / _bwd;
 *  - cast_out (may be NULL): the top layer's outputs (forward) / the input gradient dx (backward) once more as
 *    [S*T][cast_ld] rows of cast_dtype, zero padded -- the operand of the GEMM that follows (rbvae_cast_pad);
 *  - dx_colsum (backward, may be NULL): [S][L] per-sequence sums over t of dx -- summed over S they are the bias
 *    gradient of the Linear that feeds the stack (percep_RBVAE_model.py:61). */
int rbvae_lstm_fwd_ex(const float* wblk, const float* wT, float* hs_all, float* hprev, float* acts, float* cs, int S,
                      int T, int L, int layers, const float* in_parts, int nparts, long part_stride, void* cast_out,
                      int cast_dtype, int cast_ld, void* stream);
int rbvae_lstm_bwd_ex(const float* wblk, const float* acts, const float* cs, const float* g_top_parts, int nparts,
                      long part_stride, float* dG, float* dx, void* cast_out, int cast_dtype, int cast_ld,
                      float* dx_colsum, int S, int T, int L, int layers, void* stream);
/* Encoder stack -> binary_concrete_logits -> decoder stack (percep_RBVAE_model.py:155-163) as ONE wavefront
 * launch: the arithmetic of rbvae_lstm_fwd(enc) + rbvae_binarize_kl_fwd_parts + rbvae_lstm_fwd(dec), with the same
 * optional slab input / cast output as the _ex forms.  hs_dec slot 0 receives z; kl_parts[s] (may be NULL) = KL sum
 * of sequence s (S parts; mean = sum / (S*T)).  rbvae_lstm_pair_fwd_ok tells whether the shape is covered
 * (L <= 32, 2 * layers * roundup64(4L) <= 1024). */
int rbvae_lstm_pair_fwd_ok(int T, int L, int layers);
int rbvae_lstm_pair_fwd(const float* wblk_enc, const float* wT_enc, const float* wblk_dec, const float* wT_dec,
                        float* hs_enc, float* hprev_enc, float* acts_enc, float* cs_enc, float* hs_dec,
                        float* hprev_dec, float* acts_dec, float* cs_dec, const float* in_parts, int nparts,
                        long part_stride, const float* U, float* y_soft, float* kl_parts, float tau, const float* tau_dev,
                        float noise_ratio, float noise_eps, int hard, float kl_p, float kl_eps, int kl_clamp, unsigned long long seed,
                        const unsigned long long* seed_dev, void* cast_out, int cast_dtype, int cast_ld, int S, int T,
                        int L, int layers, void* stream);
/* The encoder stack's BPTT with rbvae_binarize_kl_bwd fused into its prologue (wavefront kernel only):
 *   g_top = g_hs + (g_z + kl_weight/(S*T) * dKL/dz(z)) * y_soft*(1-y_soft)/tau,   g_hs may be NULL;
 * cast_out / dx_colsum as in rbvae_lstm_bwd_ex. */
int rbvae_lstm_bwd_bin(const float* wblk, const float* acts, const float* cs, const float* g_z, const float* y_soft,
                       const float* z, const float* g_hs, float tau, const float* tau_dev, float kl_weight, float kl_p, float kl_eps,
                       int kl_clamp, float* dG, float* dx, void* cast_out, int cast_dtype, int cast_ld, float* dx_colsum,
                       int S, int T, int L, int layers, void* stream);
/* Both stacks' BPTT as ONE wavefront launch (autograd of percep_RBVAE_model.py:155-163 as run by
 * percep_RBVAE_train.py:552): rbvae_lstm_bwd_ex(decoder stack) -> rbvae_binarize_kl_bwd -> rbvae_lstm_bwd_ex(encoder
 * stack), T + 2*layers - 1 dependent steps instead of 2 * (T + layers - 1).  The decoder stack's input gradient (the
 * gradient of the codes) never leaves the chip; dz (may be NULL) receives a copy, gz_extra (may be NULL) is a second
 * gradient of the codes added to it before the binarise backward.  Everything else as in rbvae_lstm_bwd_ex /
 * rbvae_lstm_bwd_bin.  rbvae_lstm_pair_bwd_ok tells whether the shape is covered (L <= 32,
 * 2 * layers * roundup64(4L) <= 1024, saved gates of both stacks within 64 KB of LDS). */
int rbvae_lstm_pair_bwd_ok(int T, int L, int layers);
int rbvae_lstm_pair_bwd(const float* wblk_enc, const float* wblk_dec, const float* acts_enc, const float* cs_enc,
                        const float* acts_dec, const float* cs_dec, const float* g_top_parts, int nparts, long part_stride,
                        const float* gz_extra, const float* y_soft, const float* z, const float* g_hs, float tau,
                        const float* tau_dev, float kl_weight, float kl_p, float kl_eps, int kl_clamp, float* dG_enc,
                        float* dG_dec, float* dx, float* dz, void* cast_out, int cast_dtype, int cast_ld, float* dx_colsum,
                        int S, int T, int L, int layers, void* stream);
int rbvae_lstm_wgrad(const float* dG, const float* hs_all, const float* hprev, float* gblk, int S, int T, int L,
                     int layers, int accumulate, void* stream);
/* the same for two stacks of equal shape (the encoder and decoder LSTMs) in one launch */
int rbvae_lstm_wgrad_pair(const float* dG_a, const float* hs_a, const float* hprev_a, float* gblk_a, const float* dG_b,
                          const float* hs_b, const float* hprev_b, float* gblk_b, int S, int T, int L, int layers,
                          int accumulate, void* stream);

/* The trainer's DataLoader + item.to(device) (percep_RBVAE_train.py:509-518, ShuffledStatePairDataset.__getitem__
 * :312-360) for a latent table resident in HBM: out[r] = table[plan[batch][r]] for r < rows, frames of frame_elems
 * floats; plan is [n_batches][rows] table rows (int64, the epoch's shuffled batches laid out in advance) and
 * batch = *counter_dev % n_batches (the device step counter: the gather can sit inside the captured step graph) or 0. */
int rbvae_gather_frames(const float* table, long table_rows, const long* plan, int rows, int n_batches,
                        const unsigned long long* counter_dev, long frame_elems, float* out, void* stream);

/* The majority vote of calculate_state_consistency (percep_RBVAE_train.py:473-497: np.unique(axis=0, return_counts) +
 * argmax per state) on the device: codes [F][L] (binary, L <= 128) are packed to 128-bit keys (element 0 most
 * significant: key order = np.unique's row order, ties go to the smallest), out[s] = {frames of state s that carry the
 * state's most common code, frames of state s}.  labels [F] int32 in [0, n_states); keys_ws: 16 F bytes (16-byte
 * aligned), counts_ws: F ints. */
int rbvae_state_vote(const float* codes, const int* labels, int F, int L, int n_states, void* keys_ws, int* counts_ws,
                     int* out, void* stream);

/* torch.optim.Adam defaults (percep_RBVAE_train.py:753,553) on a flat f32 buffer; g is scaled by gscale
 * first.  The step number comes from `step` or, when step_dev != NULL, from a device counter that the call
 * first ADVANCES by one (then hyper_ws, 2 floats, receives the bias-correction terms): graph-replay safe.
 * step_dev == NULL with hyper_ws != NULL: hyper_ws already holds the terms (rbvae_combine_losses prepared them). */
int rbvae_adam_step(float* w, const float* g, float* m, float* v, long n, double lr, double beta1, double beta2,
                    double eps, int step, float gscale, unsigned long long* step_dev, float* hyper_ws,
                    void* stream);
/* The trainer's scalar bookkeeping in one launch (percep_RBVAE_train.py:531-549):
 * out4 = [recon + beta*kl + alpha*pair, recon, kl, pair]; recon from `recon` or, when sse_ws != NULL,
 * finished here as inv_n * sum(sse_ws[0..nparts)) (the col2im kernel's partial sums); kl = kl[0] or, when
 * kl_parts > 0, kl_scale * sum(kl[0..kl_parts)) (rbvae_binarize_kl_fwd_parts' per-block sums); pair = pair[0]
 * or, when pair_parts > 0, w_sim * sum(pair[2i]) + w_dis * sum(pair[2i+1]) (rbvae_contrast_term_fused).
 * step_dev != NULL: also advances the device step counter and leaves Adam's bias-correction terms for that step
 * in hyper_ws (2 floats); rbvae_adam_step(step_dev = NULL, hyper_ws) then uses them without a launch of its own.
 * lr_dev != NULL: the learning rate is read from that device double (a captured graph follows an lr schedule). */
int rbvae_combine_losses(const float* sse_ws, int nparts, float inv_n, const float* recon, const float* kl,
                         int kl_parts, float kl_scale, const float* pair, int pair_parts, float w_sim, float w_dis,
                         float beta, float alpha, float* out4, unsigned long long* step_dev, double lr,
                         const double* lr_dev, double beta1, double beta2, float* hyper_ws, void* stream);

/* ---- frozen LDM / Stable-Diffusion VAE encoder (cfg 5: on-the-fly latents) ------------------------
 * The convolutions, 1x1 projections and both attention products run on rbvae_gather_gemm (stride-1 and
 * asymmetric-pad stride-2 tap tables, residuals through `addend`); these are the remaining pieces.
 * GroupNorm(32, eps 1e-6, affine) + swish: src/stable-diffusion/ldm/modules/diffusionmodules/model.py:33-39
 * (stats_ws: 2*N*groups floats).  softmax_rows: AttnBlock :186-192.  posterior_sample:
 * ldm/modules/distributions/distributions.py:24-37 with ldm/models/diffusion/ddpm.py:542-549's scale:
 * latent[n][c][h][w] f32 = scale * (mean + exp(0.5*clamp(logvar,-30,20)) * eps); eps NULL = posterior mode. */
int rbvae_groupnorm_swish(int dtype, const void* x, void* y, const float* gamma, const float* beta, float* stats_ws,
                          int N, int HW, int C, int ldx, int ldy, int groups, float eps, int swish, void* stream);
/* The same with a workspace of rbvae_groupnorm_ws_floats(...) floats: statistics then come from whole pixel rows
 * (16-byte loads, per-block (mean, M2) merged by the parallel-variance formula) instead of one workgroup walking
 * an (image, group) twice; needs C % 8 == 0 (bf16) / C % 4 == 0 (f32) and 16-byte aligned rows, else falls back. */
size_t rbvae_groupnorm_ws_floats(int dtype, int N, int HW, int C, int groups);
int rbvae_groupnorm_swish_ws(int dtype, const void* x, void* y, const float* gamma, const float* beta, float* stats_ws,
                             size_t ws_floats, int N, int HW, int C, int ldx, int ldy, int groups, float eps, int swish,
                             void* stream);
/* the statistics alone (no normalised copy): mean = stats_ws[0 : N*groups], rstd = stats_ws[N*groups : 2*N*groups];
 * rbvae_gn_affine turns them into the scale / shift rbvae_conv3x3_halo applies while it stages its input. */
int rbvae_groupnorm_stats(int dtype, const void* x, float* stats_ws, size_t ws_floats, int N, int HW, int C, int ldx,
                          int groups, float eps, void* stream);
/* the apply pass alone from given statistics (e.g. rbvae_gn_finish_tiles' mean_out / rstd_out): for convolutions with
 * four or more 128-channel output tiles one standalone pass costs less than re-normalising the patch in every tile. */
int rbvae_groupnorm_apply(int dtype, const void* x, void* y, const float* mean, const float* rstd, const float* gamma,
                          const float* beta, int N, int HW, int C, int ldx, int ldy, int groups, int swish, void* stream);
int rbvae_softmax_rows(int dtype, const void* x, void* y, long rows, int n, int ld, void* stream);
/* AttnBlock.forward's q k^T * C^-0.5 -> softmax -> . v (ldm/modules/diffusionmodules/model.py:186-198) for N images
 * of hw tokens x C channels as ONE batched, tiled, online-softmax kernel: the hw x hw scores are never materialised.
 * Q, K, V, O: NHWC rows [N*hw][ld*] (may be column blocks of one fused q|k|v projection).  rbvae_attention_ok tells
 * whether the shape is covered (bf16, hw % 32 == 0, C in {64,128,256,512}); otherwise use the three-launch form
 * (rbvae_gather_gemm + rbvae_softmax_rows + rbvae_transpose2d). */
int rbvae_attention_ok(int dtype, int hw, int C);
int rbvae_attention(int dtype, const void* Q, const void* K, const void* V, void* O, int N, int hw, int C, int ldq,
                    int ldk, int ldv, int ldo, float scale, void* stream);
int rbvae_transpose2d(int dtype, const void* in, void* out, int R, int C, int ldi, int ldo, void* stream);
int rbvae_posterior_sample(int dtype, const void* moments, int ld, const float* eps, float* latent, int N, int Z,
                           int HW, float scale, void* stream);

/* ---- halo-resident 3x3 convolution, stride 1 (csrc/conv_halo.hip) ---------------------------------
 * The ResnetBlock / conv_out convolutions of the LDM encoder (ldm/modules/diffusionmodules/model.py:82-141, 368-459:
 * torch.nn.Conv2d(cin, cout, 3, 1, 1)) with the 18 x 18 input patch of a 16 x 16 pixel tile staged in LDS once per
 * 128-byte channel slice and shared by all nine taps (rbvae_gather_gemm re-gathers it per tap).
 *   Out[n][oh][ow][co] = bias[co] + addend + sum_{kh,kw,ci} f(A[n][oh+kh-pad_h][ow+kw-pad_w][ci]) * W[co][kh*3+kw][ci]
 * A / Out / addend NHWC rows of the storage type, W packed [Nout][9][Kc] (rbvae_pack3), zero_page >= 128 zero bytes.
 * f = identity, or with gn_scale / gn_shift ([Nimg][Kc] f32, from rbvae_gn_finish_tiles / rbvae_gn_affine) the
 * producer's GroupNorm (+ swish when gn_swish) applied while the patch is staged: model.py:38-39 + :33-35 as called at
 * :121-131 -- x -> swish(x * scale + shift); padding pixels stay zero (the reference pads AFTER the normalisation).
 * stats_part (may be NULL; rbvae_conv3x3_halo_stats_floats floats): per pixel tile and group of stats_cg output channels
 * the (mean, sum of squared deviations) of the STORED values (after bias / addend), which rbvae_gn_finish_tiles merges
 * into the next GroupNorm's statistics -- the consumer's normalisation costs no pass over the activation.
 * rbvae_conv3x3_halo_ok: 1 when the shape is covered (OW >= 16, OH >= 8, Kc % 64 (bf16) / 32 (f32) == 0, Nout % 128 == 0);
 * other shapes run on rbvae_gather_gemm. */
int rbvae_conv3x3_halo_ok(int dtype, int IH, int IW, int OH, int OW, int Kc, int Nout);
int rbvae_conv3x3_halo(int dtype, const void* A, const void* W, void* Out, const float* bias, const void* addend,
                       const void* zero_page, const float* gn_scale, const float* gn_shift, int gn_swish,
                       float* stats_part, int stats_cg, int Nimg, int IH, int IW, int OH, int OW, int pad_h, int pad_w,
                       int Kc, int Nout, int lda, int ldo, void* stream);
size_t rbvae_conv3x3_halo_stats_floats(int Nimg, int OH, int OW, int Nout, int cg);
/* GroupNorm(groups, eps, affine) statistics -> the per-(image, channel) scale / shift rbvae_conv3x3_halo applies
 * (model.py:38-39): from the producing convolution's per-tile partials (gn_finish_tiles; mean_out / rstd_out [Nimg*groups]
 * optional), or from the mean / rstd of rbvae_groupnorm_swish_ws's statistics kernels (gn_affine). */
int rbvae_gn_finish_tiles(const float* stats_part, const float* gamma, const float* beta, float* scale, float* shift,
                          float* mean_out, float* rstd_out, int Nimg, int OH, int OW, int C, int groups, float eps,
                          int tile_h, int tile_w, void* stream);   /* tile: 16 x 16 (rbvae_conv3x3_halo), 8 x 16 (rbvae_conv_in) */
/* The LDM encoder's conv_in, Conv2d(Cin <= 4 -> Nout <= 256, 3x3, stride 1, pad 1) on f32 NCHW frames (model.py:385-389,
 * :436), as ONE kernel in bf16 storage -- rbvae_im2col + the one-tap rbvae_gather_gemm, the same 64-deep MFMA chain per
 * output -- with the GroupNorm partial statistics of its output out of the epilogue: stats_part (may be NULL;
 * rbvae_conv_in_stats_floats floats) receives per 8 x 16 pixel tile and group of cg (4, 8 or 16) channels the (mean, sum of
 * squared deviations) of the stored values, merged by rbvae_gn_finish_tiles(.., 8, 16).  W [Nout][64] bf16, column
 * (kh*3+kw)*Cin + ci zero padded; out [N*H*W][ldo] bf16. */
int rbvae_conv_in_ok(int dtype, int Cin, int H, int W, int Nout, int N, int cg);
size_t rbvae_conv_in_stats_floats(int N, int H, int W, int Nout, int cg);
int rbvae_conv_in(int dtype, const float* x, const void* W, const float* bias, const void* zero_page, void* out, float* stats_part,
                  int cg, int N, int Cin, int H, int Wd, int Nout, int ldo, void* stream);
int rbvae_gn_affine(const float* mean, const float* rstd, const float* gamma, const float* beta, float* scale,
                    float* shift, int N, int C, int groups, void* stream);

/* ---- halo-resident transposed 3x3 convolution, stride 2 (csrc/deconv_halo.hip) --------------------------
 * ConvTranspose2d(cin, cout, 3, stride 2, padding 1, output_padding 1) forward of the RBVAE decoder
 * (models/percep_RBVAE/percep_RBVAE_model.py:76-81) and, with the data-gradient weight order, the input gradient of
 * the encoder's Conv2d(3, stride 2, padding 1) (autograd of :54-57 under total_loss.backward(), percep_RBVAE_train.py:552):
 * the four output-parity classes of rbvae_gather_gemm's "dgrad" descriptor in ONE workgroup per 128 (two workgroups per
 * CU) or 256 input-grid positions x 64 output channels, the input patch staged in LDS once per channel slice and shared by the nine taps.
 *   A [Nimg*TH*TW][lda] (the TH x TW input grid), Out [Nimg*2TH*2TW][ldo], W packed [Nout][9][Kc] (tap index kh*3+kw);
 *   epilogue = rbvae_gather_gemm's (bias, relu, scale, drop_mode 0 / 1 keyed hash / 2 explicit mask with the same element
 *   indices, gate, per-tile column sums of the stored values into colsum_ws [rbvae_deconv3x3s2_halo_colsum_rows][Nout]).
 * rbvae_deconv3x3s2_halo_ok: 1 when covered (TW % 4 == 0, Kc % 64 (bf16) / 32 (f32) == 0, Nout % 64 == 0, patch fits). */
int rbvae_deconv3x3s2_halo_ok(int dtype, int Nimg, int TH, int TW, int Kc, int Nout);
/* input-grid positions per workgroup the shape runs with: 128 (two 4-wave workgroups per CU), 256 (one 8-wave workgroup:
 * grids whose patch rows do not fit the 80 KB form, e.g. 4 x 4 and 11 x 20), 0 = not covered */
int rbvae_deconv3x3s2_halo_tile_rows(int dtype, int Nimg, int TH, int TW, int Kc, int Nout);
int rbvae_deconv3x3s2_halo_colsum_rows(int dtype, int Nimg, int TH, int TW, int Kc, int Nout);
int rbvae_deconv3x3s2_halo(int dtype, const void* A, const void* W, void* Out, const float* bias, const void* gate,
                           const void* mask, const void* zero_page, int Nimg, int TH, int TW, int Kc, int Nout, int lda,
                           int ldo, int relu, int drop_mode, float drop_p, float scale, unsigned long long seed,
                           const unsigned long long* seed_dev, float* colsum_ws, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RBVAE_HIP_H */
