// Weight gradient of a 3x3 stride-2 pad-1 convolution (and of ConvTranspose2d(c, c, 3, 2, 1, 1): the same sum with the
// operands' roles swapped) on wide layers, with the THREE TAPS OF ONE KERNEL ROW in one workgroup:
//
//   dW[ks][a][3 kh + kw][b] = sum_{p in K-slice ks} S[p][a] * G[hi(p, kh, kw)][b],   hi((n, r, c), kh, kw) = (n, 2r + kh - 1, 2c + kw - 1)
//
// S = the low-resolution operand ([Nimg*OH*OW][lds] bf16: a conv's output gradient / a transposed conv's input), G = the
// high-resolution one ([Nimg*2OH*2OW][ldg] bf16: the conv's input / the transposed conv's output gradient); autograd of
// percep_RBVAE_model.py:51-57,76-81 as run by percep_RBVAE_train.py:552.
//
// rbvae_wgrad_gemm gives every tap its own workgroups: per 64-pixel K step a 128 x 128 tile stages 16 KB of S and 16 KB of
// gathered G rows for 2.1 MFLOP, and the step is bound by the CU's L2 -> LDS intake (0.74 us against 0.215 us of matrix-core
// time).  rbvae_wgrad3x3s2_halo (nine taps, 64 x 64 channels) re-fetches every pixel block once per
// channel tile: 16 times on a 256 x 256 layer.  Here a workgroup owns a 128 (a) x 128 (b) tile of the taps kw = 0, 1, 2 of ONE
// kernel row kh: per block of 64 low-resolution pixels (RB rows x W columns) the [64 px][128 a] tile of S arrives once
// (16 KB) and of G only the rows 2r + kh - 1 with their 2W + 1 columns (kw = 0 and kw = 2 share all odd columns but one):
// 34-36 KB instead of 3 x 16 -- 50 KB per 6.3 MFLOP instead of 96.  A wave (2 x 4 over a x b) owns 64 a x 32 b x 3 taps =
// 24 accumulator tiles (96 registers); the S fragments of a 32-pixel half serve the three taps.
//
// LDS images (256-byte pixel rows, reduction index = row, fragments by ds_read_b64_tr_b16 as in wgrad_gemm.hip):
//   S tile   row k = W rr + cc; 16-byte chunks XOR-swizzled by tr_swz<256> (wgrad_gemm.hip).
//   G patch  block row rr holds 2W + 1 slots: j = 0..W the odd columns 2 (c0 + j) - 1, j = W + 1..2W the even columns
//            2 (c0 + j - W - 1); tap kw of pixel (rr, cc) is slot j = cc | W + 1 + cc | cc + 1: consecutive pixels are
//            consecutive slots although the convolution strides by two.  Chunk pairs XORed by wr_swz_g(rr, j): the eight
//            pixel rows a 32-lane half of a transposing read touches fall on eight distinct 32-byte bank groups.
// Blocks tile the flattened rows (n, r) -> n OH + r (a block may span images: every block row looks up its own image),
// W = 8 or 4 columns (whichever pads the image width less), strips of W columns.
// Stages (50-52 KB) go through a ring of three by LDS-DMA with counted vmcnt waits, one barrier per block.
// WAVE ROLES: 12 waves per workgroup.  Waves 0-7 (two per SIMD) only read fragments and issue MFMAs; waves 8-11 (one per SIMD)
// only issue the LDS-DMA pieces.  With the pieces in the MFMA waves' own instruction streams a block took 2 850 cycles
// against 1 536 of matrix-core time -- independent of the bytes fetched (a quarter of the lanes: the same), of the
// clock and of the number of workgroups (tools/wr_stamps.py): each piece holds its issuing wave for ~70 cycles, and an
// in-order wave that sits in the vector-memory issue feeds no MFMA.  Fragment reads run one (half, tap) unit ahead of
// the MFMAs.
// Workgroup -> (tile, K-slice): XCD x (= workgroup id % 8) takes a contiguous eighth of the (K-slice, tile) items in
// slice-major order: the 12 tiles of a K-slice share its pixels in that XCD's L2.  Each K-slice writes its own f32 slab [a][t][b] (rbvae_wgrad_gemm's layout: the
// same fixed-order reduction jobs follow; bitwise reproducible, no float atomics).
#include "common.h"
#include <type_traits>

#ifndef WR_ABL          // timing ablations (results wrong on purpose; RBVAE_ABLATION builds only): 1 no LDS-DMA, 2 no fragment reads / MFMAs, 3 see stage(), 5 no fragment reads, 6 one MFMA wave per SIMD
#define WR_ABL 0
#endif
#if WR_ABL && !defined(RBVAE_ABLATION)
#error "WR_ABL builds give wrong results: define RBVAE_ABLATION to confirm"
#endif

#ifndef WR_STAMPS       // -DWR_STAMPS=1 (tools/ab_variants.sh, rbvae_dbg_wr_stamps): per-workgroup phase stamps; never in the product build
#define WR_STAMPS 0
#endif

namespace rbvae {

typedef __attribute__((ext_vector_type(8))) short wr_bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short wr_s16x4_t;
typedef __attribute__((ext_vector_type(4))) float wr_f32x4_t;

struct WrArgs {
    const unsigned char* S;    // [Nimg*OH*OW][lds] bf16
    const unsigned char* G;    // [Nimg*2OH*2OW][ldg] bf16
    float* dW;                 // [ksplit][Ca][9][Cb] f32
    const unsigned char* zero; // >= 16 zero bytes
    int Nimg, OH, OW, Ca, Cb, lds, ldg, ksplit;
    int nstrip, nblk, rows;    // column strips per row block, blocks in all, Nimg * OH
    int rbmod;                 // (block rows) % OH
    unsigned long long* stamps;   // -DWR_STAMPS=1 builds only: [workgroup][wave 0 | wave 4][8] (tools/wr_stamps.py)
};

constexpr int WR_RING = 3;
constexpr int WR_A_BYTES = 64 * 256;
constexpr int WR_MAXP = 13;                    // LDS-DMA instructions per producer wave and stage, at most

__device__ __forceinline__ void wr_glds16(const void* g, void* lds) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}
__device__ __forceinline__ int wr_swz_a(int row) { return ((row & 3) | (((row >> 3) & 1) << 2)) << 1; }   // tr_swz<256>
// chunk-pair swizzle of patch slot (block row rr, slot jj): the eight pixel rows a 32-lane half of a transposing read touches
// (four consecutive slots of two block rows: rr, rr + 1 for W = 8; rr, rr + 2 for W = 4) get eight distinct values, and the
// pixels k + 4 / k + 32 of a lane get the SAME value as pixel k (their reads are immediate offsets of one address)
template <int W> __device__ __forceinline__ int wr_swz_g(int rr, int jj) { return (jj & 3) | ((((W == 8) ? rr : rr >> 1) & 1) << 2); }

template <int N> __device__ __forceinline__ void wr_wait_barrier() {
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}

template <int W>
__global__ __launch_bounds__(768, 1) void wgrad_row_k(const WrArgs p) {
    constexpr int RB = 64 / W, SPR = 2 * W + 1, NSLOT = RB * SPR;
    constexpr int G_BYTES = NSLOT * 256, STAGE = WR_A_BYTES + G_BYTES;
    constexpr int TOT = 16 + NSLOT / 4;                               // LDS-DMA instructions per stage: 50 (W = 8) / 52 (W = 4)
    constexpr int KSUB = (32 / W) * SPR * 256;                        // second 32-pixel half of a block: 32 / W block rows on
    constexpr int HOFF = (W == 8 ? 4 : SPR) * 256;                    // pixel k + 4: four slots on (W = 8) / the next block row (W = 4)
    static_assert(NSLOT % 4 == 0 && TOT <= 4 * WR_MAXP && TOT > 4 * (WR_MAXP - 1), "instruction split");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int na = p.Ca >> 7, nb = p.Cb >> 7, ntile = na * nb * 3;
    // Workgroups are dealt to the 8 XCDs round-robin by id; XCD x takes the x-th eighth of the (K-slice, tile) items in
    // slice-major order (wgrad_gemm.hip's order): its <= 32 workgroups cover 3-4 K-slices, whose 12 tiles share the slice's
    // pixels in that XCD's L2.  Placement is a speed matter only.
    const int total = ntile * p.ksplit, per = (total + 7) >> 3;
    const int item = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per || item >= total) return;      // the grid is 8 * per workgroups (uniform exit, before any barrier)
    const int ks = item / ntile;
    int tile = item - ks * ntile;
    const int kh = tile % 3; tile /= 3;
    const int a0 = (tile / nb) * 128, b0 = (tile % nb) * 128;
    // balanced K-slices: the first nblk % ksplit slices take one block more
    const int base = p.nblk / p.ksplit, rem = p.nblk - base * p.ksplit;
    const int blk0 = ks * base + min(ks, rem);
    const int nsteps = base + (ks < rem ? 1 : 0);
    const int IW = 2 * p.OW;
#if WR_STAMPS
    unsigned long long st_t0 = wall_clock64(), st_wait = 0, st_bar = 0, st_loop0 = 0, st_loop1 = 0, st_c0 = 0, st_c1 = 0;
#endif

    if (w >= 8) {
        // ================= producer waves: pw = w - 8 issues instructions i = pw + 4 j of every stage =================
        // i < 16: S rows 4 i .. + 3, otherwise patch slots 4 (i - 16) .. + 3 (16 lanes = one 256-byte row each).  The pieces
        // are buffer loads to LDS (buffer_load_dwordx4 .. offen lds): a source offset is a per-step SCALAR (the block's
        // corner, the instruction's SGPR offset) plus a per-lane CONSTANT (its VGPR offset, computed once): low-resolution
        // pixel row (R0 OW + c0) + (rr OW + cc); high-resolution row 2 (R0 + rr) + kh - 1 of the flattened (n, row) axis
        // (IH = 2 OH: the image index drops out), column 2 c0 + dcol, against a descriptor whose base lies one row and one
        // pixel in front of G so that every part stays non-negative.  A lane whose pixel is padding gets an offset beyond
        // the descriptor's range and the hardware writes zeros for it (tools/probes/buffer_lds_oob.hip).
        const int pw = w - 8;
        const bool thirteen = pw + 4 * (WR_MAXP - 1) < TOT;   // this wave issues 13 instructions per stage (else 12)
        const int lds_b = p.lds * 2, ldg_b = p.ldg * 2;
        int d_off[WR_MAXP], d_rr[WR_MAXP], d_c[WR_MAXP], d_rm[WR_MAXP];
#pragma unroll
        for (int j = 0; j < WR_MAXP; ++j) {
            const int i = pw + 4 * j;
            if (j < 4) {
                const int k = 4 * i + (lane >> 4);
                const int rr = k / W, cc = k % W;
                d_rr[j] = rr; d_c[j] = cc; d_rm[j] = 0;
                d_off[j] = (rr * p.OW + cc) * lds_b + (((lane & 15) ^ wr_swz_a(k)) * 16) + a0 * 2;
            } else {
                const int s = min(4 * (i - 16) + (lane >> 4), NSLOT - 1);
                const int rr = s / SPR, jj = s - rr * SPR;
                const int dcol = jj <= W ? 2 * jj - 1 : 2 * (jj - W - 1);
                d_rr[j] = rr; d_c[j] = dcol; d_rm[j] = rr % p.OH;
                d_off[j] = ((2 * rr + kh) * IW + dcol + 1) * ldg_b + (((lane & 15) ^ (wr_swz_g<W>(rr, jj) << 1)) * 16) + b0 * 2;
            }
        }
        const auto rsrcS = __builtin_amdgcn_make_buffer_rsrc((void*)p.S, 0, p.rows * p.OW * lds_b, 0x00020000);
        const auto rsrcG = __builtin_amdgcn_make_buffer_rsrc((void*)(p.G - (long)(IW + 1) * ldg_b), 0,
                                                             (2 * p.rows * IW + IW + 1) * ldg_b, 0x00020000);
        constexpr int OOB = (int)0x80000000u;
        // the stage being issued (uniform, advanced block by block): column strip, first flattened row and its row within
        // the image, ring slot
        int p_strip, p_R0, p_r0, p_slot = 0;
        {
            const int rb = blk0 / p.nstrip;
            p_strip = blk0 - rb * p.nstrip;
            p_R0 = rb * RB;
            p_r0 = p_R0 % p.OH;
        }
        auto stage = [&]() {
            const int c0 = p_strip * W;
            const int sS = (p_R0 * p.OW + c0) * lds_b, sG = (2 * p_R0 * IW + 2 * c0) * ldg_b;
            const int rows_left = p.rows - p_R0, cols_left = p.OW - c0, c2 = 2 * c0, r0 = p_r0;
            const bool top = kh == 0 && (p_r0 == 0 || p_r0 + RB > p.OH);    // kh = 0 and an image's first row inside the block
            auto* pld = (__attribute__((address_space(3))) unsigned char*)smem + p_slot * STAGE + pw * 1024;
#pragma unroll
            for (int j = 0; j < WR_MAXP; ++j) {
                if (j == WR_MAXP - 1 && !thirteen) break;
                int voff = d_off[j];
                if (j < 4) {
                    voff = (d_rr[j] < rows_left && d_c[j] < cols_left) ? voff : OOB;
                } else {
                    const int c = d_c[j] + c2;
                    voff = (d_rr[j] < rows_left && c >= 0 && c < IW) ? voff : OOB;
                    if (top) { const int t = d_rm[j] + r0; voff = (t != 0 && t != p.OH) ? voff : OOB; }
                }
#if WR_ABL == 3          // every lane out of range: the pieces are issued and write zeros to LDS, nothing is fetched
                voff = OOB;
#endif
#if WR_ABL != 1
                __builtin_amdgcn_raw_ptr_buffer_load_lds(j < 4 ? rsrcS : rsrcG, pld + j * 4096, 16, voff, j < 4 ? sS : sG, 0, 0);
#endif
            }
            // advance to the next block
            if (++p_strip == p.nstrip) {
                p_strip = 0; p_R0 += RB;
                p_r0 += p.rbmod; if (p_r0 >= p.OH) p_r0 -= p.OH;
            }
            p_slot = p_slot + 1 == WR_RING ? 0 : p_slot + 1;
        };
        if (nsteps > 0) stage();
        if (nsteps > 1) stage();
        // block s: own pieces of stage s landed (stage s + 1's may be in flight) -> barrier: the MFMA waves start on stage s and
        // are past stage s - 1, whose ring slot stage s + 2 takes
        for (int s = 0; s < nsteps; ++s) {
            if (s + 1 < nsteps) { if (thirteen) wr_wait_barrier<13>(); else wr_wait_barrier<12>(); } else wr_wait_barrier<0>();
            if (s + 2 < nsteps) stage();
        }
        return;
    }

    // ================= MFMA waves: wave (wr, wc) owns a sub-tiles 4 wr .. + 3, b sub-tiles 2 wc, 2 wc + 1 of the three taps =================
    const int fi = lane & 15, fg = lane >> 4;
    const int q = fi >> 2, pp = fi & 3;
    const int wr = w >> 2, wc = w & 3;
    int offA;                                            // a sub-tile 4 wr; sub-tile 4 wr + mt: chunk index ^ 2 mt = byte address ^ 32 mt
    {
        const int row = 8 * fg + q;                      // + 4 for the second read, + 32 for the second half
        const int chunk = ((wr * 4) * 2 + (pp >> 1)) ^ wr_swz_a(row);
        offA = row * 256 + chunk * 16 + (pp & 1) * 8;
    }
    int offB[3];                                         // [kw]: pixel k = 8 fg + q of the first half, b sub-tile 2 wc
#pragma unroll                                           // (sub-tile 2 wc + 1: chunk index ^ 2 = byte address ^ 32; pixel k + 4 and the second
    for (int kw = 0; kw < 3; ++kw) {                     //  half: the same swizzle value, i.e. immediate offsets HOFF and KSUB)
        const int k = 8 * fg + q;
        const int rr = k / W, cc = k % W;
        const int jj = kw == 0 ? cc : (kw == 1 ? W + 1 + cc : cc + 1);
        const int chunk = ((wc * 2) * 2 + (pp >> 1)) ^ (wr_swz_g<W>(rr, jj) << 1);
        offB[kw] = WR_A_BYTES + (rr * SPR + jj) * 256 + chunk * 16 + (pp & 1) * 8;
    }

    wr_f32x4_t acc[3][4][2];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) acc[kw][mt][nt] = wr_f32x4_t{0.f, 0.f, 0.f, 0.f};

    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    // Fragment reads are inline asm (invisible to the compiler's wait-count pass); every group is followed by a counted
    // s_waitcnt lgkmcnt tied ("+v") to the registers it guards, and isa_check proves on the listing that nothing touches a
    // register before its wait.
    // reads of the S fragments of a sub-tiles m0, m0 + 1 (4 reads) / of one tap's G fragments (4 reads)
    auto read_a2 = [&](unsigned lb, auto ksub_tag, auto m0_tag, wr_s16x4_t (&al)[4], wr_s16x4_t (&ah)[4]) {
        constexpr int KS_ = decltype(ksub_tag)::value, M0 = decltype(m0_tag)::value;
#pragma unroll
        for (int mt = M0; mt < M0 + 2; ++mt) {
#if WR_ABL == 5
            continue;
#endif
            const unsigned ad = (lb + offA) ^ (mt * 32);
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(al[mt]) : "v"(ad), "n"(KS_ * 32 * 256));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(ah[mt]) : "v"(ad), "n"(KS_ * 32 * 256 + 4 * 256));
        }
    };
    auto read_b = [&](unsigned lb, auto ksub_tag, int kw, wr_s16x4_t (&bl)[2], wr_s16x4_t (&bh)[2]) {
        constexpr int KS_ = decltype(ksub_tag)::value;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
#if WR_ABL == 5
            continue;
#endif
            const unsigned ad = (lb + offB[kw]) ^ (nt * 32);
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(bl[nt]) : "v"(ad), "n"(KS_ * KSUB));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(bh[nt]) : "v"(ad), "n"(KS_ * KSUB + HOFF));
        }
    };
    // counted waits tied to the registers they guard: at most Y younger reads stay outstanding
    auto wait_b = [&](auto y_tag, wr_s16x4_t (&bl)[2], wr_s16x4_t (&bh)[2]) {
        constexpr int Y = decltype(y_tag)::value;
        asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(bl[0]), "+v"(bh[0]), "+v"(bl[1]), "+v"(bh[1]) : "n"(Y));
    };
    auto wait_a2 = [&](auto y_tag, auto m0_tag, wr_s16x4_t (&al)[4], wr_s16x4_t (&ah)[4]) {
        constexpr int Y = decltype(y_tag)::value, M0 = decltype(m0_tag)::value;
        asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(al[M0]), "+v"(ah[M0]), "+v"(al[M0 + 1]), "+v"(ah[M0 + 1]) : "n"(Y));
    };
    auto wait_a2b = [&](auto y_tag, auto m0_tag, wr_s16x4_t (&al)[4], wr_s16x4_t (&ah)[4], wr_s16x4_t (&bl)[2], wr_s16x4_t (&bh)[2]) {
        constexpr int Y = decltype(y_tag)::value, M0 = decltype(m0_tag)::value;
        asm volatile("s_waitcnt lgkmcnt(%8)"
                     : "+v"(al[M0]), "+v"(ah[M0]), "+v"(al[M0 + 1]), "+v"(ah[M0 + 1]), "+v"(bl[0]), "+v"(bh[0]), "+v"(bl[1]), "+v"(bh[1])
                     : "n"(Y));
    };
    // the MFMAs of tap kw on a sub-tiles m0 .. m0 + NM - 1 (fragments that have passed their wait)
    auto mma = [&](int kw, auto m0_tag, auto nm_tag, const wr_s16x4_t (&al)[4], const wr_s16x4_t (&ah)[4], const wr_s16x4_t (&bl)[2],
                   const wr_s16x4_t (&bh)[2]) {
        constexpr int M0 = decltype(m0_tag)::value, NM = decltype(nm_tag)::value;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const wr_bf16x8_t fb = {bl[nt][0], bl[nt][1], bl[nt][2], bl[nt][3], bh[nt][0], bh[nt][1], bh[nt][2], bh[nt][3]};
#pragma unroll
            for (int mt = M0; mt < M0 + NM; ++mt) {
                const wr_bf16x8_t fa = {al[mt][0], al[mt][1], al[mt][2], al[mt][3], ah[mt][0], ah[mt][1], ah[mt][2], ah[mt][3]};
                acc[kw][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb, fa, acc[kw][mt][nt], 0, 0, 0);
            }
        }
    };
    using H0 = std::integral_constant<int, 0>;
    using H1 = std::integral_constant<int, 1>;
    using M0 = std::integral_constant<int, 0>;
    using M2 = std::integral_constant<int, 2>;
    using N2 = std::integral_constant<int, 2>;
    using N4 = std::integral_constant<int, 4>;
    using Y0 = std::integral_constant<int, 0>;
    using Y4 = std::integral_constant<int, 4>;

#define WR_SB() __builtin_amdgcn_sched_barrier(0)
    if (nsteps > 0) {
        wr_s16x4_t a0l[4], a0h[4], a1l[4], a1h[4], xbl[2], xbh[2], ybl[2], ybh[2], zbl[2], zbh[2];
#pragma unroll
        for (int i = 0; i < 4; ++i) a1l[i] = a1h[i] = wr_s16x4_t{0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 2; ++i) zbl[i] = zbh[i] = wr_s16x4_t{0, 0, 0, 0};
        unsigned lb = lds0;
        // Block s: barrier (stage s landed: the producer waves waited for their pieces in front of it).  Six (half, tap) units
        // of 8 MFMAs: u0..u2 = taps 0..2 on the first 32 pixels (S fragments a0), u3..u5 on the second (a1); G fragments
        // rotate through three sets (x, y, z).  Reads are issued 8-16 MFMAs ahead of their use -- with one unit of cover the
        // wave that loses the matrix-pipe arbitration to its SIMD partner ran its last third alone at 37 % of the pipe --
        // never more than 12 outstanding (lgkmcnt is a 4-bit counter), and the last unit's MFMAs run at the top of the next
        // block behind its first reads (zero fragments at s = 0).  Nothing asynchronous crosses the loop's back edge.
#if WR_STAMPS
        st_loop0 = wall_clock64(); st_c0 = __builtin_amdgcn_s_memtime();
#endif
        for (int s = 0; s < nsteps; ++s) {
#if WR_STAMPS
            const unsigned long long st_b = __builtin_amdgcn_s_memtime();
            asm volatile("s_barrier" ::: "memory");
            if (s > 0) st_bar += __builtin_amdgcn_s_memtime() - st_b;
#else
            asm volatile("s_barrier" ::: "memory");
#endif
#if WR_ABL == 6
            if (w >= 4) continue;
#endif
#if WR_ABL != 2
            read_b(lb, H0{}, 0, xbl, xbh);
            read_a2(lb, H0{}, M0{}, a0l, a0h);
            read_a2(lb, H0{}, M2{}, a0l, a0h);
            WR_SB(); mma(2, M0{}, N4{}, a1l, a1h, zbl, zbh); WR_SB();          // u5 of block s - 1
            wait_a2b(Y4{}, M0{}, a0l, a0h, xbl, xbh);                         // x, a0[0:2]  (a0[2:4] in flight)
            read_b(lb, H0{}, 1, ybl, ybh);
            WR_SB(); mma(0, M0{}, N2{}, a0l, a0h, xbl, xbh); WR_SB();          // u0, first half
            wait_a2(Y4{}, M2{}, a0l, a0h);                                    // a0[2:4]  (y in flight)
            read_b(lb, H0{}, 2, zbl, zbh);
            WR_SB(); mma(0, M2{}, N2{}, a0l, a0h, xbl, xbh); WR_SB();          // u0, second half
            wait_b(Y4{}, ybl, ybh);                                           // y  (z in flight)
            read_a2(lb, H1{}, M0{}, a1l, a1h);
            WR_SB(); mma(1, M0{}, N4{}, a0l, a0h, ybl, ybh); WR_SB();          // u1
            wait_b(Y4{}, zbl, zbh);                                           // z  (a1[0:2] in flight)
            read_b(lb, H1{}, 0, xbl, xbh);
            read_a2(lb, H1{}, M2{}, a1l, a1h);
            WR_SB(); mma(2, M0{}, N4{}, a0l, a0h, zbl, zbh); WR_SB();          // u2
            wait_a2b(Y4{}, M0{}, a1l, a1h, xbl, xbh);                         // a1[0:2], x  (a1[2:4] in flight)
            read_b(lb, H1{}, 1, ybl, ybh);
            WR_SB(); mma(0, M0{}, N2{}, a1l, a1h, xbl, xbh); WR_SB();          // u3, first half
            wait_a2(Y4{}, M2{}, a1l, a1h);                                    // a1[2:4]  (y in flight)
            read_b(lb, H1{}, 2, zbl, zbh);
            WR_SB(); mma(0, M2{}, N2{}, a1l, a1h, xbl, xbh); WR_SB();          // u3, second half
            wait_b(Y4{}, ybl, ybh);                                           // y  (z in flight)
            WR_SB(); mma(1, M0{}, N4{}, a1l, a1h, ybl, ybh); WR_SB();          // u4
            wait_b(Y0{}, zbl, zbh);                                           // z: the stage is in registers
#endif
            lb = lb + STAGE == lds0 + WR_RING * STAGE ? lds0 : lb + STAGE;
        }
#if WR_ABL != 2
        mma(2, M0{}, N4{}, a1l, a1h, zbl, zbh);
#endif
#if WR_STAMPS
        st_loop1 = wall_clock64(); st_c1 = __builtin_amdgcn_s_memtime();
#endif
    }

    // ---- slab: D[row = b 4 fg + r][col = a fi]: a lane owns 4 consecutive b of one a -> one 16-byte store
    float* slab = p.dW + (size_t)ks * p.Ca * 9 * p.Cb;
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int a = a0 + (wr * 4 + mt) * 16 + fi;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int b = b0 + (wc * 2 + nt) * 16 + 4 * fg;
                *(wr_f32x4_t*)(slab + ((size_t)a * 9 + kh * 3 + kw) * p.Cb + b) = acc[kw][mt][nt];
            }
        }
#if WR_STAMPS
    if (p.stamps && (tid == 0 || tid == 256)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned long long* o = p.stamps + (size_t)blockIdx.x * 16 + (tid ? 8 : 0);
        o[0] = st_t0; o[1] = st_loop0; o[2] = st_loop1; o[3] = wall_clock64(); o[4] = st_wait; o[5] = st_c1 - st_c0; o[6] = nsteps;
        o[7] = st_bar;
    }
#endif
}

static int wr_width(int OW) { return ((OW + 7) / 8) * 8 == ((OW + 3) / 4) * 4 ? 8 : 4; }

static int wr_ok(int dtype, int Nimg, int OH, int OW, int Ca, int Cb) {
    return dtype == RBVAE_BF16 && Nimg >= 1 && OH >= 1 && OW >= 1 && Ca >= 128 && Cb >= 128 && Ca % 128 == 0 && Cb % 128 == 0 &&
           (long)Nimg * OH * OW * 4 < (1l << 31);       /* pixel rows; the byte sizes are checked against the leading dimensions at launch */
}

static int wr_blocks(int Nimg, int OH, int OW) {
    const int W = wr_width(OW);
    return ((OW + W - 1) / W) * ((Nimg * OH + 64 / W - 1) / (64 / W));
}

}  // namespace rbvae

using namespace rbvae;

static unsigned long long* g_wr_stamps = nullptr;

extern "C" {

#if WR_STAMPS
/* stamped builds only (include/rbvae_dbg.h): later rbvae_wgrad3x3s2_row launches write phase stamps into buf */
int rbvae_dbg_wr_stamps(unsigned long long* buf, void* stream) {
    (void)stream;
    g_wr_stamps = buf;
    return RBVAE_OK;
}
#endif

int rbvae_wgrad3x3s2_row_ok(int dtype, int Nimg, int OH, int OW, int Ca, int Cb) { return wr_ok(dtype, Nimg, OH, OW, Ca, Cb); }

/* 64-pixel blocks the K loop walks (the caller sizes ksplit against it) */
int rbvae_wgrad3x3s2_row_blocks(int Nimg, int OH, int OW) { return wr_blocks(Nimg, OH, OW); }

int rbvae_wgrad3x3s2_row(int dtype, const void* S, const void* G, float* dW_slabs, const void* zero_page, int Nimg, int OH,
                         int OW, int Ca, int Cb, int lds_, int ldg, int ksplit, void* stream) {
    RBVAE_CHECK_ARG(S && G && dW_slabs && zero_page, "wgrad3x3s2_row: null pointer");
    RBVAE_CHECK_ARG(wr_ok(dtype, Nimg, OH, OW, Ca, Cb), "wgrad3x3s2_row: shape not covered (dtype %d, %d x %d x %d, %d x %d channels): "
                    "query rbvae_wgrad3x3s2_row_ok", dtype, Nimg, OH, OW, Ca, Cb);
    RBVAE_CHECK_ARG(lds_ >= Ca && ldg >= Cb && lds_ % 8 == 0 && ldg % 8 == 0, "wgrad3x3s2_row: leading dimensions lds=%d ldg=%d", lds_, ldg);
    RBVAE_CHECK_ARG(((uintptr_t)S | (uintptr_t)G | (uintptr_t)dW_slabs | (uintptr_t)zero_page) % 16 == 0,
                    "wgrad3x3s2_row: pointers must be 16-byte aligned");
    RBVAE_CHECK_ARG(((long)Nimg * OH * 4 * OW + 2 * OW + 1) * ldg * 2 < (1l << 31) && (long)Nimg * OH * OW * lds_ * 2 < (1l << 31),
                    "wgrad3x3s2_row: operands of 2 GiB or more (32-bit buffer offsets)");
    WrArgs a;
    a.S = (const unsigned char*)S; a.G = (const unsigned char*)G; a.dW = dW_slabs; a.zero = (const unsigned char*)zero_page;
    a.Nimg = Nimg; a.OH = OH; a.OW = OW; a.Ca = Ca; a.Cb = Cb; a.lds = lds_; a.ldg = ldg;
    const int W = wr_width(OW);
    a.nstrip = (OW + W - 1) / W;
    a.nblk = wr_blocks(Nimg, OH, OW);
    a.rows = Nimg * OH;
    a.rbmod = (64 / W) % OH;
    RBVAE_CHECK_ARG(ksplit >= 1 && ksplit <= a.nblk, "wgrad3x3s2_row: ksplit=%d (1 .. %d blocks)", ksplit, a.nblk);
    a.ksplit = ksplit;
    a.stamps = g_wr_stamps;
    const int ntile = (Ca / 128) * (Cb / 128) * 3;
    const long blocks = 8 * (((long)ntile * ksplit + 7) / 8);
    hipStream_t st = (hipStream_t)stream;
    if (W == 8) {
        constexpr int LDS = WR_RING * (WR_A_BYTES + 8 * 17 * 256);
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute((const void*)wgrad_row_k<8>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
            attr_set = true;
        }
        hipLaunchKernelGGL(wgrad_row_k<8>, dim3((unsigned)blocks), dim3(768), LDS, st, a);
    } else {
        constexpr int LDS = WR_RING * (WR_A_BYTES + 16 * 9 * 256);
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute((const void*)wgrad_row_k<4>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
            attr_set = true;
        }
        hipLaunchKernelGGL(wgrad_row_k<4>, dim3((unsigned)blocks), dim3(768), LDS, st, a);
    }
    RBVAE_CHECK_LAUNCH("wgrad3x3s2_row");
    return RBVAE_OK;
}

}  // extern "C"
