/* librbvae_dbg -- diagnostic probes of the gfx950 lane maps the GEMM kernels are written against
 * (tests/test_hw_maps.py) and, in stamped builds of the main library only (-DGG_STAMPS=1 / -DWG_STAMPS=1,
 * RBVAE_DEBUG=1 python symbols-from-video_amd/build.py -> librbvae_hip_debug.so), the phase-stamp hooks of the two
 * GEMM kernels.  Not part of the product ABI (include/rbvae_hip.h). */
#ifndef RBVAE_DBG_H
#define RBVAE_DBG_H
#ifdef __cplusplus
extern "C" {
#endif
/* One-wave kernels that pin the gfx950 lane maps the GEMM kernels assume (librbvae_dbg.so). */
int rbvae_dbg_mfma_bf16(const void* A, const void* B, float* D, void* stream);   /* [16x32]x[32x16] bf16 */
int rbvae_dbg_mfma_f32(const float* A, const float* B, float* D, void* stream);  /* [16x4]x[4x16] f32 */
int rbvae_dbg_glds(const void* src, const int* lane_src_chunk, void* out, void* stream);
/* stamped builds of librbvae_hip only: every later rbvae_gather_gemm launch writes 8 phase time stamps (100 MHz) per workgroup into buf (NULL = off) */
int rbvae_dbg_gg_stamps(unsigned long long* buf, void* stream);
int rbvae_dbg_wg_stamps(unsigned long long* buf, void* stream);   /* same for rbvae_wgrad_gemm (-DWG_STAMPS=1 builds) */
/* rbvae_wgrad3x3s2_row (-DWR_STAMPS=1 builds): per workgroup start, loop start, loop end, end (100 MHz), cycles waited at
 * the stage barriers, cycles of the loop, K steps */
int rbvae_dbg_wr_stamps(unsigned long long* buf, void* stream);
/* librbvae_hip: which kernel rbvae_conv3x3_halo runs for bf16: 2 (default) the product dispatch -- the persistent kernel with
 * producer / MFMA wave roles (conv_halo_ws.hip) for layers of at most 256 input channels and at least 512 tiles, else the
 * one-tile-per-workgroup kernel (conv_halo.hip); 1 the latter always; 0 the persistent kernel wherever it covers -- for the
 * bit-identity test and A/B timing; returns the previous value */
int rbvae_dbg_conv_halo_variant(int v);
/* which kernels rbvae_lstm_pair_fwd / rbvae_lstm_pair_bwd (librbvae_hip) run at L == 32: 1 (default) one thread per hidden
 * unit (lstm_pair_fwd_unit_k: four gate rows per thread, one barrier per diagonal), 0 one thread per gate row -- for the
 * bit-identity tests and A/B timing; returns the previous value */
int rbvae_dbg_lstm_unit_threads(int v);
int rbvae_dbg_tr16(const void* img, const int* rowsel, const int* colsel, void* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RBVAE_DBG_H */
