// Implicit-GEMM MFMA kernel family: parameters shared by Linear / 1x1 conv / 3x3 conv launchers.
#pragma once
#include "common.h"

namespace lavie {

constexpr int IGEMM_MAX_SEG = 4;    // two conv sources + two shortcut sources
constexpr int IGEMM_BK = 64;    // K-tile in halfs: 128-B rows in LDS, one 16x16x32 MFMA pair per tile

// One K-segment of the A operand: a source tensor `src` (channels-last rows of `C` halfs) contributing
// `nchunks` 64-channel slabs, each visited at `ntaps` spatial taps (9 = the 3x3 stencil, 1 = centre only).
// K order inside a segment: slab-major, tap-minor.  A 3x3 conv over a concatenated input [x1 | x2] is two
// 9-tap segments; a fused 1x1 shortcut appends one or two 1-tap segments.
struct IgemmSeg {
    const half_t* src;
    int C;        // row length (channels) of src
    int c0;       // first channel of this segment inside src rows
    int nchunks;  // number of 64-channel slabs
    int ntaps;    // 9 (3x3) or 1 (centre tap: fused 1x1 shortcut); temporal mode: 3 or 5 frame taps
};

struct IgemmParams {
    // A operand, plain mode (GATHER = false): row-major [M, K] with leading dimension lda.
    const half_t* A;
    int lda;
    // W operand: row-major [N, Ktot] (PyTorch Linear layout; conv weights repacked [Cout][tap][Cin]).
    const half_t* W;
    int ldw;
    half_t* C;
    int ldc;
    const float* bias;    // [N] or nullptr   (GEGLU: permuted like W rows)
    const float* bias2;   // [M / rows_per_batch, ldb2] or nullptr (time-embedding projection per video)
    int ldb2;
    int rows_per_batch;
    const half_t* R;      // residual [M, N] (ldr) or nullptr; may alias C
    int ldr;
    int M, N, nk;         // nk = total number of K-tiles
    // LayerNorm folding: a producer GEMM writes per-row partial (sum, sum of squares) of its fp16 output,
    // one pair per 16*NT-column wave tile: rowstat_out[m * rowstat_slots + slot]; the consumer GEMM runs on the RAW
    // rows with gamma folded into W and finishes  y = rstd_m (acc - mean_m s_n) + bias_n  in its epilogue.
    float* rowstat_out;        // [M, N / rowstat_cols, 2] or nullptr (EPI_LINEAR, splits == 1 only)
    int rowstat_cols;          // columns per slot the caller sized rowstat_out for (igemm_rowstat_cols); launch_igemm checks that
                               // the kernel it selects writes slots of exactly this width (0 = unchecked)
    const float* ln_stats;     // [M, 2] (mean, rstd) of the A rows (launch_rowstat_finalize), or nullptr
    // ... or (round 4) the producer's partials themselves: the epilogue folds the ln_slots pairs of a row with rowstat_finalize's
    // arithmetic, and the finalize launch disappears.  Kernels with the shared epilogue only (the persistent kernel stages finished
    // (mean, rstd) rows through LDS and has no room for the partials): launch_igemm refuses the combination.
    const float* ln_partials;  // [M, ln_slots, 2] (sum, sum of squares) or nullptr (then ln_stats)
    int ln_slots;
    float ln_inv_len, ln_eps;  // 1 / row length of the normalised rows, epsilon
    const float* ln_s;         // [N]: s_n = sum_k W'[n, k]
    // GroupNorm statistics from the producer (round 4): per (row block, channel) sum and sum of squares of the ROUNDED fp16 output,
    // written by the epilogue that stores the tensor (or by the split-K reduce), so that the consuming GroupNorm needs no statistics
    // pass over the tensor (resnet.py:180,191; attention.py:369; unet.py:504).  Layout (cs_index below): per block and channel quad
    // four sums then four sums of squares.  A block = the rows of one wave tile (colstat_rows = 16 * MT of the kernel that runs, 32
    // for the split-K reduce); the parity-form upsample conv writes four sets of source-row blocks (one per output parity).
    float* colstat_out;        // [sets][ceil(rows / colstat_rows)][N / 4][2][4] or nullptr (EPI_LINEAR only)
    int colstat_rows;          // rows per block the caller planned for (igemm_colstat_rows); launch_igemm checks the kernel it picks
    int splits;           // split-K factor (1 = none); > 1 needs `slab`
    float* slab;          // [splits, M, N] fp32 partial sums
    // Gather geometry (GATHER = true): output pixel grid [NI, Ho, Wo], source grid [NI, Hi, Wi],
    // virtual input grid (Hi << ups, Wi << ups) for the folded nearest-x2 upsample.
    int Ho, Wo, Hi, Wi, stride, ups;
    // Temporal mode (tframes > 0; GATHER = true; 128-row and ping-pong kernels): rows are tokens (b, f, pixel) with tpix pixels per
    // frame, and tap t of a segment with ntaps = T reads row m + (t - T/2) * tpix, or zeros when frame f + t - T/2 falls
    // outside [0, tframes) — nn.Conv3d with kernel (T, 1, 1), padding (T/2, 0, 0) (vsr/models/resnet.py:258-259, 274).
    int tframes, tpix;
    int nseg;
    int par_ups;          // halo-patch kernel only: parity form of the 3x3 conv of a nearest-x2 upsampled image (igemm_patch.hip MODE 3):
                          // W = four [N][ldw] matrices (parity py * 2 + px), one 4-tap segment, M = OUTPUT rows, Hi x Wi = source grid
    int tap_major;        // diagnostic: K order tap > slab instead of slab > tap (needs weights packed to match)
    IgemmSeg seg[IGEMM_MAX_SEG];
    const half_t* zero;   // >= 128 B of zeros: source of out-of-image taps
};

enum IgemmEpilogue { EPI_LINEAR = 0, EPI_GEGLU = 1 };

// Picks a tile and launches.  Returns 0 or a negative status with lavie::set_error().
int launch_igemm(const IgemmParams& p, bool gather, int epilogue, hipStream_t stream);
// 160x320 two-group ping-pong kernel (igemm_pp.hip); EPI_LINEAR only, N %% 320 == 0, the caller runs the split-K reduce.
int launch_igemm_pp(const IgemmParams& p, bool gather, hipStream_t stream);
int launch_igemm_pp_geglu(const IgemmParams& p, hipStream_t stream);   // 160x256 variant, GEGLU epilogue, N %% 256 == 0
// Persistent ping-pong kernel (igemm_ppx.hip): plain A rows, no split-K; at most 256 workgroups walk the output tiles with
// the LDS-DMA stream running across tile boundaries.  EPI_LINEAR (N %% 320 == 0 or N %% 256 == 0) and EPI_GEGLU (N %% 256 == 0).
bool igemm_ppx_eligible(const IgemmParams& p, int epilogue);
int launch_igemm_ppx(const IgemmParams& p, int epilogue, hipStream_t stream);
int igemm_ppx_read_stamps(unsigned long long* out);   // diagnostic stamp build (mode 0x37): [8 waves][32] cycle sums of workgroup 0
// 320x160 halo-patch 3x3 conv kernel (igemm_patch.hip): stride 1, 9-tap segments only; the caller runs the split-K reduce.
bool igemm_patch_eligible(const IgemmParams& p);
int igemm_patch_bn(int N);                        // 160 / 128 / 0: column-tile width of the halo-patch kernel for N channels
int launch_igemm_patch(const IgemmParams& p, hipStream_t stream);
// parity form of conv3x3(nearest_x2(x)) (igemm_patch.hip MODE 3): fills p (incl. splits; the caller sets p->slab), false = not this geometry
bool igemm_setup_parity_upsample(IgemmParams* p, const half_t* x, int C, const half_t* wpar, const float* bias, half_t* y, int NI, int Hi,
                                 int Wi, const half_t* zero);
int igemm_patch_read_stamps(unsigned long long* out);   // diagnostic stamp build: [8 waves][16] cycle sums of workgroup 0
// Split-K factor the launcher would like for this problem (1 = none); slab size = splits * M * N floats.
int igemm_plan_splits(int M, int N, int nk, int epilogue);
// Same for a gathered conv whose geometry and K segments are filled in (M, N, nk, Ho, Wo, stride, ups, seg[], nseg):
// also considers the halo-patch kernel.
int igemm_plan_splits_gather(const IgemmParams& p);
// whether launch_igemm would hand this gathered conv (p.splits filled in) to the halo-patch kernel
bool igemm_patch_planned(const IgemmParams& p);
// wave-tile width (16*NT) launch_igemm will pick for a plain, unsplit EPI_LINEAR GEMM: the row-statistics slot width
int igemm_rowstat_cols(int M, int N, int nk);
// whether launch_igemm hands a plain GEMM of this shape (unsplit) to the persistent ping-pong kernel
bool igemm_takes_ppx(int M, int N, int nk, int epilogue);
// rows per column-statistics block of the kernel launch_igemm will run for `p` (geometry, segments and p.splits filled in):
// 80 (halo-patch, ping-pong and persistent kernels), 64 (128-row kernel), 32 (split-K: the reduce kernel writes them),
// 0 = this launch cannot emit them (2-D patch tiles, GEGLU)
int igemm_colstat_rows(const IgemmParams& p, bool gather, int epilogue);
// ... and the aligned run of output rows inside which the rows of one such block lie (GnColStat::span): the block height itself
// for kernels whose wave tiles are contiguous rows; one frame for the halo-patch kernel's 2-D tiles and for the parity-form
// upsample conv; one video for its temporal-conv tiles (call after igemm_colstat_rows returned > 0)
int igemm_colstat_span(const IgemmParams& p, bool gather, int rows);
constexpr int COLSTAT_REDUCE_ROWS = 32;
// float index of (block, channel c, which = 0 sum / 1 sum of squares) in a column-statistics buffer of a C-channel tensor
__host__ __device__ inline size_t cs_index(size_t block, int c, int which, int C) {
    return ((block * (size_t)(C >> 2) + (size_t)(c >> 2)) * 2 + (size_t)which) * 4 + (size_t)(c & 3);
}
// partials [M, slots, 2] (sum, sum of squares over `row_len` values per row) -> out [M, 2] = (mean, rstd); fixed order
int launch_rowstat_finalize(const float* partials, int slots, int M, int row_len, float eps, float* out, hipStream_t stream);
// Low nibble: 0 = automatic kernel / tile choice, 1 = 128-row kernel with the widest tile,
// 3 = 160x320 ping-pong kernel whenever N %% 320 == 0, 4 = automatic but never the ping-pong kernel (A/B timing),
// 5 = halo-patch conv kernel whenever the conv is eligible, 6 = automatic but never the halo-patch kernel,
// 7 = persistent ping-pong kernel for every eligible plain GEMM, 8 = automatic but never the persistent kernel.
// High nibble: diagnostic ablation build of the forced kernel (results wrong) — except 0xC (0xC0 / 0xC5): the halo-patch kernel's
// ping-pong K loop instead of the shipped software-pipelined one (same results; A/B timing).
void igemm_force_tile(int mode);
void igemm_force_splits(int s);   // 0 = automatic

}  // namespace lavie
