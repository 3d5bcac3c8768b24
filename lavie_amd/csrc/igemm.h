// Implicit-GEMM MFMA kernel family: parameters shared by Linear / 1x1 conv / 3x3 conv launchers.
#pragma once
#include "common.h"

namespace lavie {

constexpr int IGEMM_MAX_SEG = 24;
constexpr int IGEMM_BK = 64;    // K-tile in halfs: 128-B rows in LDS, one 16x16x32 MFMA pair per tile

// One K-segment of the A operand: `nchunks` K-tiles of 64 channels read from tensor `src`
// (channels-last rows of `C` halfs) at spatial tap (dy, dx).  A 3x3 conv over a concatenated
// input [x1 | x2] is 18 segments; a fused 1x1 shortcut appends 1-2 centre-tap segments.
struct IgemmSeg {
    const half_t* src;
    int C;        // row length (channels) of src
    int c0;       // first channel of this segment inside src rows
    int nchunks;  // number of 64-channel tiles
    int dy, dx;   // tap offset in the (virtual, i.e. post-upsample) input grid
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
    // Gather geometry (GATHER = true): output pixel grid [NI, Ho, Wo], source grid [NI, Hi, Wi],
    // virtual input grid (Hi << ups, Wi << ups) for the folded nearest-x2 upsample.
    int Ho, Wo, Hi, Wi, stride, ups;
    int nseg;
    IgemmSeg seg[IGEMM_MAX_SEG];
    const half_t* zero;   // >= 128 B of zeros: source of out-of-image taps
};

enum IgemmEpilogue { EPI_LINEAR = 0, EPI_GEGLU = 1 };

// Picks a tile and launches.  Returns 0 or a negative status with lavie::set_error().
int launch_igemm(const IgemmParams& p, bool gather, int epilogue, hipStream_t stream);
// 0 = automatic tile choice, 1 = 128-row tiles only, 2 = 256-row tiles whenever N %% 160 == 0 (tests, A/B timing)
void igemm_force_tile(int mode);

}  // namespace lavie
