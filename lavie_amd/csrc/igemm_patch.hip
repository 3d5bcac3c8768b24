// 3x3 convolution (stride 1, pad 1) as a 320x160 implicit-GEMM tile whose activation operand is staged ONCE per
// 64-channel slab as a halo patch and re-read nine times with shifted LDS addresses (gfx950).
//
// Why: the implicit-GEMM kernels of igemm.hip / igemm_pp.hip stage the A tile of every (slab, tap) K-tile separately,
// so each activation row travels L2 -> LDS nine times and the LDS-DMA issue rate (about one 1-KiB piece per 30 cycles
// per CU with four issuing waves) paces the loop: 60 pieces per K-tile for the 160x320 tile.  Here a tile is 320
// CONTIGUOUS output pixels = whole image rows (of one frame, or a whole number of small frames), so the nine taps of a
// slab read the same (rows + 2 halo rows) x W pixels, just shifted by (dy-1) W + (dx-1) patch rows: the patch is staged
// once per slab (<= 56 pieces for 9 K-tiles) and only the 160x64 weight tile (20 pieces) is new in every K-tile ->
// 26 pieces per K-tile for the same 2 x 320 x 160 x 64 flop.  Column wrap-around (x-1 at x = 0, x+1 at x = W-1) is the
// only case the patch cannot express; those lanes read a 128-B row of zeros instead.
//
// K loop, shipped build (round 3; template value STAMP = 6): software-pipelined.  Every wave interleaves the ten fragment reads
// of the NEXT k-step (inline-asm ds_read_b128, one per two MFMAs) with the 25 in-place inline-asm MFMAs of this k-step; one
// lgkmcnt(0) per k-step, ONE barrier per K-tile (in its middle: weight stage hand-over), weights two K-tiles ahead.
// tools/probes/pipe_probe.hip (profiles/r03_pipe_probe.txt) priced the loop structures with the same work per K-tile: without
// LDS-DMA both reach the MFMA ceiling of the held clock (0.83-0.86 of the 2.4 GHz peak); with the 26 LDS-DMA pieces per K-tile the
// ping-pong skeleton drops to 0.63 and this one to 0.67, wherever the pieces come from (L2, MALL, HBM: same) — the LDS-DMA
// writes themselves are what costs, and register-staged copies (global_load + ds_write_b128) cost more (0.53 / 0.61).
// K loop, first build (STAMP 0-5, kept for A/B and for the stamp diagnostics) = igemm_pp.hip:
// 8 waves, wave tile 80x80 (4 waves along M, 2 along N), two groups (waves 0-3 / 4-7 = the SIMD
// partners) half a k-step apart, R phase (fragment reads + LDS-DMA issue) opposite the partner's M phase (25 MFMAs).
// Group g owns output columns [80 g, 80 g + 80) and alone reads its half of the weight tile.  Per K-tile t:
//   G0 R(t,0): W rows  0-79  of K-tile t+1 (10 pieces over 4 waves)      G1 R(t,0): W rows 80-159 of K-tile t+1
//   G0 R(t,1), G1 R(t,1): one piece per wave of the NEXT slab's patch (8 per K-tile, done after 7 of the 9 K-tiles)
// LDS: 2 x (patch 56 KiB + zero row) + 2 weight stages x 20 KiB + pixel table + tap table = 159.7 KiB.
// Restrictions (the launcher falls back otherwise): every K segment has 9 taps (a fused 1x1 shortcut runs as its own
// GEMM whose result comes back through the residual operand), stride 1, no upsample, M % 320 == 0, W % 8 == 0,
// 320 % W == 0, tile = part of one frame or whole frames, patch <= 448 rows, split-K only at slab boundaries.
#include <hip/hip_ext.h>

#include <cstring>
#include <type_traits>

#include "igemm.h"
#include "igemm_epilogue.h"
#include "profile.h"

namespace lavie {

namespace pt {
constexpr int MT = 5;
constexpr int BM = 320, THREADS = 512;
constexpr int PATCH_ROWS = 448, PATCH_PIECES = PATCH_ROWS / 8;     // 56
constexpr int PATCH_STRIDE = (PATCH_ROWS + 1) * 128;                // 57,472: 448 patch rows + one row of zeros ("row 448")
constexpr int W_BYTES_MAX = 160 * 128;                              // 20,480 (the 128-wide tile uses 16,384 of it)
constexpr int W_BASE = 2 * PATCH_STRIDE;
constexpr int PTAB = W_BASE + 2 * W_BYTES_MAX;                      // int[PATCH_ROWS]: source pixel of a patch row or -1
constexpr int SEGTAB = PTAB + PATCH_ROWS * 4;
constexpr int TAPTAB = SEGTAB + IGEMM_MAX_SEG * 6 * 4;              // u16[9][320]: (u << 3) | (u & 7), u = patch row read by
constexpr int LDS_BYTES = TAPTAB + 9 * BM * 2;                      //   tile row r at that tap (448 = the zero row)
static_assert(LDS_BYTES <= 160 * 1024, "does not fit LDS");
// 2-D tiles (image rows wider than a tile: the VSR stage's 512 / 256 / 128-pixel rows): 10 image rows x 32 columns = 320
// pixels, patch = 12 x 34 = 408 rows with the halo columns staged explicitly
constexpr int T2_W = 32, T2_H = 10, T2_PW = T2_W + 2, T2_ROWS = (T2_H + 2) * T2_PW;
static_assert(T2_W * T2_H == BM && T2_ROWS <= PATCH_ROWS && T2_W % 16 == 0, "2-D tile geometry");
}  // namespace pt

// STAMP (diagnostic build, forced with lavie_debug_force_tile(0x75)): s_memtime at every phase boundary; the per-wave sums
// of workgroup 0 go to g_patch_stamps[wave][segment] (read back with lavie_debug_patch_stamps).  Segments per K-tile:
// 1 R(t,0) (issue + LDS wait), 2 barrier, 3 M(t,0), 4 barrier, 6 R(t,1), 7 barrier, 8 M(t,1) with the next tap's address
// arithmetic, 9 LDS-DMA wait, 10 barrier; [11] = K-tiles.  Read the SHARES, never the run time of this build (guide section 7, In-kernel stamps).
__device__ unsigned long long g_patch_stamps[8 * 16];

// NT = 16-column blocks per wave: 5 -> 320x160 tile (every channel count of the base model), 4 -> 320x128 (the VSR widths)
// MODE 0: tiles of whole image rows; 1: 2-D tiles (10 rows x 32 columns); 2: temporal (T,1,1) convolution, tile = every frame of
// 320 / F pixels (no halo at all: tap t of a row is the same pixel t - T/2 frames away, inside the tile or outside the clip)
// MODE 3 (round 3): 3x3 conv of a nearest-x2 upsampled image (Upsample3D, resnet.py:44-79) as FOUR 2x2 convs on the source image,
// one per output parity (py, px) = blockIdx.z: the nine taps of output pixel (2y + py, 2x + px) fall on only 2 x 2 source
// pixels {y - 1 + py, y + py} x {x - 1 + px, x + px}, so the weights of the taps that share a source pixel are summed at pack
// time (launch_pack_conv3x3_parity: fp32 sums, one rounding) and K shrinks from 9 C to 4 C — 2.25x fewer FLOP for the same
// result up to that one rounding.  Tiles run over SOURCE pixels (whole source rows, as MODE 0), four K-tiles per slab, the
// epilogue scatters a tile row to output row ((n 2H + 2y + py) 2W + 2x + px); p.M counts OUTPUT rows.
template <int EPI, int STAMP = 0, int NT = 5, int MODE = 0>
__global__ __launch_bounds__(pt::THREADS, 2) void igemm_patch_kernel(const IgemmParams p) {
    using namespace pt;
    constexpr int BN = 2 * NT * 16, W_BYTES = BN * 128;
    extern __shared__ __attribute__((aligned(128))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;                    // 0: leading group, 1: trailing group (SIMD partners)
    const int q = wave & 3;
    const int wm = q, wn = grp;

    const int n_tiles = p.N / BN;
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x;
        const int qq = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (qq + 1) : r * (qq + 1) + (xcd - r) * qq) + (bid >> 3);
    }
    int tile_m, tile_n;
    igemm_tile_of(bid, (int)gridDim.x / n_tiles, n_tiles, (long)p.N * p.nk * IGEMM_BK, &tile_m, &tile_n);
    const int m0 = tile_m * BM;
    const int n0 = tile_n * BN;
    const int split = blockIdx.y;
    const int ntap = MODE == 2 ? __builtin_amdgcn_readfirstlane(p.seg[0].ntaps) : MODE == 3 ? 4 : 9;       // K-tiles per 64-channel slab
    const int py = MODE == 3 ? (int)(blockIdx.z >> 1) : 0, px = MODE == 3 ? (int)(blockIdx.z & 1) : 0;           // output parity
    const int nslab = p.nk / ntap;
    const int slab_begin = (int)((long)nslab * split / p.splits);
    const int slab_end = (int)((long)nslab * (split + 1) / p.splits);
    const int t_begin = slab_begin * ntap, t_end = slab_end * ntap;
    // patch pieces of the NEXT slab staged per K-tile and wave: the whole patch must have landed one K-tile before the slab
    // ends.  3x3: 56 pieces, one per wave over 7 of the 9 K-tiles; temporal: 40 pieces (320 rows), two (T = 5) or three (T = 3)
    const int npieces = MODE == 2 ? BM / 8 : PATCH_PIECES;
    const int ppk = MODE == 2 ? (ntap >= 5 ? 2 : 3) : MODE == 3 ? 3 : 1;

    const int lr = lane >> 3;
    const int kofs = ((lane & 7) ^ lr) * 8;         // source K offset (halfs) after the slot swizzle
    auto sgpr = [](int v) { return __builtin_amdgcn_readfirstlane(v); };

    // ---- tile geometry: 320 pixels = `rows_seg` image rows of each of `nf` frame segments, or (t2) 10 rows x 32 columns
    const int Wd = MODE == 2 ? BM : MODE == 3 ? p.Wi : p.Wo, Hd = MODE == 2 ? 1 : MODE == 3 ? p.Hi : p.Ho, HW = Hd * Wd;      // (unused in temporal mode)
    constexpr bool t2 = MODE == 1;                  // the launcher instantiates MODE 1 exactly when BM % Wd != 0
    constexpr bool tmode = MODE == 2;
    const int seg_px = HW >= BM ? BM : HW;          // pixels of one frame segment inside the tile
    const int rows_seg = seg_px / Wd;
    const int PF = (rows_seg + 2) * Wd;             // patch rows of one segment (with the two halo image rows)
    const int frame0 = m0 / HW;
    const int y_first = HW >= BM ? (m0 - frame0 * HW) / Wd : 0;
    int ty0 = 0, tx0 = 0;                           // t2: image position of the tile's first pixel
    if (t2) {
        const int tpr = Wd / T2_W;
        const int tin = tile_m - frame0 * (HW / BM);
        ty0 = (tin / tpr) * T2_H;
        tx0 = (tin - (tin / tpr) * tpr) * T2_W;
    }

    int* ptab = reinterpret_cast<int*>(smem + PTAB);
    int* segtab = reinterpret_cast<int*>(smem + SEGTAB);
    // tmode: tile row r = (frame r / PX, pixel tp0 + r % PX) of video tb, PX = 320 / F pixels
    const int tF = tmode ? p.tframes : 1, tD = tmode ? p.tpix : 1;
    const int PX = BM / tF;
    const int tb = tmode ? tile_m / (tD / PX) : 0;
    const int tp0 = tmode ? (tile_m - tb * (tD / PX)) * PX : 0;
    auto trow = [&](int r) { return (tb * tF + r / PX) * tD + tp0 + (r - (r / PX) * PX); };
    for (int j = tid; j < PATCH_ROWS; j += THREADS) {
        if (tmode) {
            ptab[j] = j < BM ? trow(j) : -1;
        } else if (t2) {
            const int py = j / T2_PW, px = j - py * T2_PW;
            const int yy = ty0 - 1 + py, xx = tx0 - 1 + px;
            const bool ok = j < T2_ROWS && (unsigned)yy < (unsigned)Hd && (unsigned)xx < (unsigned)Wd;
            ptab[j] = ok ? (frame0 * Hd + yy) * Wd + xx : -1;
        } else {
            const int f = j / PF, jj = j - f * PF;
            const int jr = jj / Wd, x = jj - jr * Wd;
            const int yy = y_first - 1 + jr;
            const bool ok = f * seg_px < BM && (unsigned)yy < (unsigned)Hd;
            ptab[j] = ok ? ((frame0 + f) * Hd + yy) * Wd + x : -1;
        }
    }
    if (tid < 64) reinterpret_cast<float*>(smem + (tid >> 5) * PATCH_STRIDE + PATCH_ROWS * 128)[tid & 31] = 0.f;
    {   // tap table: which patch row tile row r reads at tap (dy, dx); column wrap-around -> the zero row
        unsigned short* taptab = reinterpret_cast<unsigned short*>(smem + TAPTAB);
        for (int idx = tid; idx < 9 * BM; idx += THREADS) {
            const int tap = idx / BM, r = idx - tap * BM;
            const int dy = tap / 3, dx = tap - dy * 3;
            int u;
            if (tmode) {                                    // frame tap: the same pixel (tap - T/2) frames away, or zeros
                const int ff = r / PX + tap - (ntap >> 1);
                u = (tap < ntap && (unsigned)ff < (unsigned)tF) ? ff * PX + (r - (r / PX) * PX) : PATCH_ROWS;
            } else if (t2) {                                // halo columns are part of the patch: no wrap-around case
                u = (r / T2_W + dy) * T2_PW + (r % T2_W) + dx;
            } else if (MODE == 3) {                         // 2x2 taps (a, b) of output parity (py, px): source pixel (y - 1 + py + a, x - 1 + px + b)
                const int a = tap >> 1, b = tap & 1, dyy = a - 1 + py, dxx = b - 1 + px;
                const int f = r / seg_px, rr = r - f * seg_px;
                const int x = rr % Wd;
                const bool bad = tap >= 4 || (unsigned)(x + dxx) >= (unsigned)Wd;
                u = bad ? PATCH_ROWS : f * PF + Wd + rr + dyy * Wd + dxx;
            } else {
                const int f = r / seg_px, rr = r - f * seg_px;
                const int x = rr % Wd;
                const bool bad = (dx == 0 && x == 0) || (dx == 2 && x == Wd - 1);
                u = bad ? PATCH_ROWS : f * PF + Wd + rr + (dy - 1) * Wd + (dx - 1);
            }
            taptab[idx] = (unsigned short)((u << 3) | (u & 7));
        }
    }
    if (tid == 0) {
#pragma unroll
        for (int i = 0; i < IGEMM_MAX_SEG; ++i) {
            const unsigned long long a = reinterpret_cast<unsigned long long>(p.seg[i].src);
            segtab[i * 6 + 0] = (int)(unsigned)a;
            segtab[i * 6 + 1] = (int)(unsigned)(a >> 32);
            segtab[i * 6 + 2] = p.seg[i].C;
            segtab[i * 6 + 3] = p.seg[i].c0;
            segtab[i * 6 + 4] = p.seg[i].nchunks;
            segtab[i * 6 + 5] = p.seg[i].ntaps;
        }
    }
    __syncthreads();
    auto load_seg = [&](int i) -> IgemmSeg {
        IgemmSeg r;
        const unsigned lo = (unsigned)sgpr(segtab[i * 6 + 0]), hi = (unsigned)sgpr(segtab[i * 6 + 1]);
        r.src = reinterpret_cast<const half_t*>(((unsigned long long)hi << 32) | lo);
        r.C = sgpr(segtab[i * 6 + 2]);
        r.c0 = sgpr(segtab[i * 6 + 3]);
        r.nchunks = sgpr(segtab[i * 6 + 4]);
        r.ntaps = sgpr(segtab[i * 6 + 5]);
        return r;
    };
    const half_t* const zero_page = reinterpret_cast<const half_t*>(
        ((unsigned long long)(unsigned)sgpr((int)(unsigned)(reinterpret_cast<unsigned long long>(p.zero) >> 32)) << 32) |
        (unsigned)sgpr((int)(unsigned)reinterpret_cast<unsigned long long>(p.zero)));
    const int nseg = sgpr(p.nseg);

    // ---- slab cursor: (segment, 64-channel chunk) of the slab whose patch is staged next
    int seg = 0, chunk = 0;
    IgemmSeg sg = load_seg(0);
    {
        int skip = slab_begin;
        while (skip >= sg.nchunks && seg + 1 < nseg) {
            skip -= sg.nchunks;
            sg = load_seg(++seg);
        }
        chunk = skip;
    }
    auto advance_slab = [&]() {
        if (++chunk == sg.nchunks) {
            chunk = 0;
            if (seg + 1 < nseg) sg = load_seg(++seg);
        }
    };
    // LDS-DMA of patch piece `i` (8 patch rows) of the cursor's slab into patch buffer `pb`; pix = ptab[i * 8 + lr]
    auto issue_patch = [&](int i, int pb, int pix) {
        const unsigned cofs = (unsigned)(sg.c0 + chunk * IGEMM_BK + kofs);
        const half_t* inside = sg.src + ((unsigned)pix * (unsigned)sg.C + cofs);
        const half_t* src = pix >= 0 ? inside : zero_page + kofs;
        __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(smem + pb * PATCH_STRIDE + i * 1024), 16, 0, 0);
    };

    // ---- weight pieces of this wave: rows 80 grp + (q + 4 j) * 8, j = 0..2 (j = 2 only for q < 2: 10 pieces per half)
    const half_t* wptr[3];
#pragma unroll
    for (int j = 0; j < 3; ++j)
        wptr[j] = p.W + (MODE == 3 ? (size_t)blockIdx.z * p.N * p.ldw : (size_t)0) + (size_t)(n0 + grp * (NT * 16) + (q + 4 * j) * 8 + lr) * p.ldw + kofs;
    const bool w3 = NT == 5 && q < 2;               // 80 rows per group = 10 pieces; 64 rows = 8 pieces: two per wave
    auto issue_w01 = [&](int t, int wst) {
        char* base = smem + W_BASE + wst * W_BYTES + grp * (NT * 16 * 128);
        __builtin_amdgcn_global_load_lds(GLB_PTR(wptr[0] + t * IGEMM_BK), LDS_PTR(base + q * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds(GLB_PTR(wptr[1] + t * IGEMM_BK), LDS_PTR(base + (q + 4) * 1024), 16, 0, 0);
    };
    auto issue_w2 = [&](int t, int wst) {
        char* base = smem + W_BASE + wst * W_BYTES + grp * (NT * 16 * 128);
        if (w3) __builtin_amdgcn_global_load_lds(GLB_PTR(wptr[2] + t * IGEMM_BK), LDS_PTR(base + (q + 8) * 1024), 16, 0, 0);
    };

    // ---- fragment rows of this lane: tile row r = 80 wm + 16 mt + (lane & 15); the patch row it reads at a tap comes from
    // the tap table (two VALU per fragment and K-tile instead of ~10: measured with the stamp build, the address
    // arithmetic was a quarter of the loop)
    const int frow = lane & 15, fg = lane >> 4;
    const int tap_lane = TAPTAB + (wm * (MT * 16) + frow) * 2;       // + tap * 640 + mt * 32
    const int w_frag = W_BASE + (wn * (NT * 16) + frow) * 128 + ((fg ^ (frow & 7)) << 4);   // k-step 0; k-step 1 = ^ 64

    f32x4 acc[NT][MT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    half8_t af0[MT], wf0[NT];
    int aaddr[MT];                                  // LDS byte address of this K-tile's A fragments (k-step 0)

    // A-fragment addresses of a tap in patch buffer pb: tap_read issues the table reads, tap_finish turns them into addresses
    int tv[MT];
    auto tap_read = [&](int tap) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            tv[mt] = *reinterpret_cast<const unsigned short*>(smem + tap_lane + tap * (BM * 2) + mt * 32);
    };
    auto tap_finish = [&](int pb) {
        const int base = pb * PATCH_STRIDE;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) aaddr[mt] = ((tv[mt] ^ fg) << 4) + base;
    };
    auto read_frags = [&](int wst, int ks, half8_t (&af)[MT], half8_t (&wf)[NT]) {
        const int kx = ks << 6;
        if (STAMP != 5) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) af[mt] = *reinterpret_cast<const half8_t*>(smem + (aaddr[mt] ^ kx));
        }
        if (STAMP != 4) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                wf[nt] = *reinterpret_cast<const half8_t*>(smem + ((w_frag + wst * W_BYTES + nt * 16 * 128) ^ kx));
        }
    };
    auto mfma_block = [&](half8_t (&af)[MT], half8_t (&wf)[NT]) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[nt], af[mt], acc[nt][mt], 0, 0, 0);
    };
    auto bar = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- prologue: patch of the first slab (7 pieces per wave), weight tile of the first K-tile
#pragma unroll 1
    for (int j = 0; j < PATCH_PIECES / 8; ++j) {
        const int i = wave + 8 * j;
        issue_patch(i, 0, ptab[i * 8 + lr]);
    }
    advance_slab();                                 // the cursor now names the slab to prefetch
    issue_w01(t_begin, 0);
    issue_w2(t_begin, 0);
    if constexpr (STAMP == 6) {                     // the interleaved loop keeps the weight tiles two K-tiles ahead
        if (t_begin + 1 < t_end) {
            issue_w01(t_begin + 1, 1);
            issue_w2(t_begin + 1, 1);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    bar();
    unsigned long long st_sum[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_prev = 0;
    auto stamp = [&](int seg_id) {
        if constexpr (STAMP != 0) {
            unsigned long long tnow;
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tnow)::"memory");
            __builtin_amdgcn_sched_barrier(0);
            if (seg_id >= 0) st_sum[seg_id] += tnow - st_prev;
            st_prev = tnow;
        }
    };
    if constexpr (STAMP == 6) {
        // ---- software-pipelined K loop (tools/probes/pipe_probe.hip: +5 % over the ping-pong skeleton with the same LDS-DMA load):
        // every wave interleaves the NEXT k-step's ten fragment reads (inline asm, one per two MFMAs) with the 25 in-place MFMAs
        // of this k-step; one lgkmcnt(0) per k-step, ONE barrier per K-tile, in its middle: behind it every wave's reads of this
        // tile's weight stage are over (the stage takes tile t + 2) and the weights of tile t + 1, issued one tile ago, have landed
        // for everyone (the reads of (t + 1, 0) start right behind the barrier).  Table reads (tap table, pixel table) ride in the
        // same lgkmcnt window as asm reads; compiler-visible LDS reads do not occur inside the loop.
        const unsigned lbase = (unsigned)(size_t)LDS_PTR(smem);
        half8_t fa[2][MT], fw[2][NT];
        tap_read(0);
        tap_finish(0);
        auto lds128 = [](half8_t& d, unsigned addr) { asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(addr) : "memory"); };
        auto mfma_ip = [](f32x4& c, const half8_t& a, const half8_t& b) {
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
        };
        unsigned wb = lbase + w_frag;               // weight fragment base of the current K-tile's stage (k-step 0)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) lds128(fa[0][mt], lbase + aaddr[mt]);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) lds128(fw[0][nt], wb + nt * 2048);
        int pb = 0, kt = 0, slab = slab_begin, nissue_prev = 0;
        for (int t = t_begin; t < t_end; ++t) {
            const int wst = (t - t_begin) & 1;
            const bool next_slab = slab + 1 < slab_end;
            const bool wrap = kt == ntap - 1;
            int piece[3];
            unsigned pix[3] = {0u, 0u, 0u};
            bool pissue[3];
            int nissue = 0;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                piece[j] = (kt * ppk + j) * 8 + wave;
                pissue[j] = (MODE >= 2 || j == 0) && j < ppk && next_slab && piece[j] < npieces;
                nissue += pissue[j] ? 1 : 0;
            }
            // ---- k-step 0: MFMAs on set 0, reads of (t, 1) into set 1, then the table entries of the next tap / patch pieces
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            {
                const unsigned wb1 = wb ^ 64u;
                unsigned a1[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) a1[mt] = lbase + (aaddr[mt] ^ 64);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        mfma_ip(acc[nt][mt], fw[0][nt], fa[0][mt]);
                        const int m = nt * MT + mt;
                        if (m >= 2 && m % 2 == 0 && (m - 2) / 2 < MT + NT) {
                            const int i = (m - 2) / 2;
                            if (i < MT) lds128(fa[1][i], a1[i]);
                            else lds128(fw[1][i - MT], wb1 + (i - MT) * 2048);
                        }
                    }
                const unsigned tapa = lbase + tap_lane + (wrap ? 0 : kt + 1) * (BM * 2);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) asm volatile("ds_read_u16 %0, %1 offset:%2" : "=v"(tv[mt]) : "v"(tapa), "n"(mt * 32) : "memory");
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    if (pissue[j]) asm volatile("ds_read_b32 %0, %1" : "=v"(pix[j]) : "v"(lbase + PTAB + (piece[j] * 8 + lr) * 4) : "memory");
            }
            // ---- middle of the K-tile
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(tv[0]), "+v"(tv[1]), "+v"(tv[2]), "+v"(tv[3]), "+v"(tv[4]), "+v"(pix[0]), "+v"(pix[1]), "+v"(pix[2])::"memory");
            if (wrap || nissue_prev == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (nissue_prev == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            else if (nissue_prev == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            bar();
            if (t + 2 < t_end) {
                issue_w01(t + 2, wst);
                issue_w2(t + 2, wst);
            }
#pragma unroll
            for (int j = 0; j < 3; ++j)
                if (pissue[j]) issue_patch(piece[j], pb ^ 1, (int)pix[j]);
            nissue_prev = nissue;
            tap_finish(wrap ? pb ^ 1 : pb);           // fragment addresses of K-tile t + 1
            // ---- k-step 1: MFMAs on set 1, reads of (t + 1, 0) into set 0
            {
                const unsigned wbn = lbase + w_frag + (wst ^ 1) * W_BYTES;
                unsigned a0[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) a0[mt] = lbase + aaddr[mt];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        mfma_ip(acc[nt][mt], fw[1][nt], fa[1][mt]);
                        const int m = nt * MT + mt;
                        if (m >= 2 && m % 2 == 0 && (m - 2) / 2 < MT + NT) {
                            const int i = (m - 2) / 2;
                            if (i < MT) lds128(fa[0][i], a0[i]);
                            else lds128(fw[0][i - MT], wbn + (i - MT) * 2048);
                        }
                    }
                wb = wbn;
            }
            if (++kt == ntap) {
                kt = 0;
                pb ^= 1;
                ++slab;
                advance_slab();
            }
        }
        // the accumulators pass through the wait states of the last MFMAs before compiler code (the epilogue) reads them
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            asm volatile("" : "+v"(acc[nt][0]), "+v"(acc[nt][1]), "+v"(acc[nt][2]), "+v"(acc[nt][3]), "+v"(acc[nt][4]));
    } else {
    if (grp == 1) bar();                            // the trailing group runs one barrier behind

    int pb = 0, kt = 0;                             // patch buffer of the current slab, tap index inside it
    int slab = slab_begin;
    // (Measured and rejected: reading a phase's fragments during the previous M phase of the same wave.  The reads then
    // sit beside the wave's own MFMAs and the M phase grows by more than the R phase shrinks: LDS returns and MFMA
    // operand traffic share the SIMD's register ports.)
    tap_read(0);
    tap_finish(0);
    stamp(-1);
    for (int t = t_begin; t < t_end; ++t) {
        const int wst = (t - t_begin) & 1;
        const bool more = t + 1 < t_end;
        const bool next_slab = slab + 1 < slab_end;
        // patch pieces of the next slab this wave stages during this K-tile: (kt * ppk + j) * 8 + wave, j < ppk
        int piece[3], pix[3];
        bool pissue[3];
        int nissue = 0;                             // wave-uniform AND the same for all eight waves (npieces % 8 == 0)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            piece[j] = (kt * ppk + j) * 8 + wave;
            pissue[j] = (MODE >= 2 || j == 0) && j < ppk && next_slab && piece[j] < npieces;
            nissue += pissue[j] ? 1 : 0;
        }
        // ---- R(t, 0): fragments, two weight pieces of K-tile t+1, the pixels of this K-tile's patch pieces
        read_frags(wst, 0, af0, wf0);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            pix[j] = -1;
            if (pissue[j]) pix[j] = ptab[piece[j] * 8 + lr];
        }
        if (more && STAMP != 2) issue_w01(t + 1, wst ^ 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        stamp(1);
        bar();
        stamp(2);
        // ---- M(t, 0)
        mfma_block(af0, wf0);
        stamp(3);
        bar();
        stamp(4);
        // ---- R(t, 1): fragments, the third weight piece, the patch pieces of the next slab, next tap's table entries
        read_frags(wst, 1, af0, wf0);
        const bool wrap = kt == ntap - 1;
        tap_read(wrap ? 0 : kt + 1);
        if (more && STAMP != 2) issue_w2(t + 1, wst ^ 1);
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if (pissue[j] && STAMP != 3) issue_patch(piece[j], pb ^ 1, pix[j]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        tap_finish(wrap ? pb ^ 1 : pb);
        stamp(6);
        bar();
        stamp(7);
        // ---- M(t, 1)
        mfma_block(af0, wf0);
        stamp(8);
        // this wave's weight pieces of K-tile t+1 have landed (the patch pieces issued after them may still fly)
        if (nissue == 0 || STAMP == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (nissue == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else if (nissue == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        stamp(9);
        if (!(grp == 1 && !more)) bar();
        stamp(10);
        if (++kt == ntap) {
            kt = 0;
            pb ^= 1;
            ++slab;
            advance_slab();
        }
    }

    }   // ping-pong loop
    if constexpr (STAMP != 0 && STAMP != 6) {
        if (blockIdx.x == 0 && blockIdx.y == 0 && lane == 0) {
#pragma unroll
            for (int i = 0; i < 11; ++i) g_patch_stamps[wave * 16 + i] = st_sum[i];
            g_patch_stamps[wave * 16 + 11] = (unsigned long long)(t_end - t_begin);
        }
    }
    if constexpr (MODE == 3) {         // scatter: source pixel (n, y, x) of this lane -> output pixel (n, 2y + py, 2x + px)
        igemm_epilogue_rows<MT, NT, EPI>(p, acc, [&](int mt) {
            const int m = m0 + wm * (MT * 16) + mt * 16 + (lane & 15);
            const int n = m / HW, rem = m - n * HW, y = rem / Wd, x = rem - y * Wd;
            return ((n * 2 * Hd + 2 * y + py) * 2 * Wd) + 2 * x + px;
        }, n0 + wn * (NT * 16) + (lane >> 4) * 4, n0 + wn * (NT * 16), lane, split,
        // column statistics: one set of source-row blocks per parity (the rows of a block lie in one frame: tiles are whole source rows)
        (int)blockIdx.z * (p.M / 4 / (MT * 16)) + (m0 + wm * (MT * 16)) / (MT * 16));
    } else if constexpr (tmode) {
        // column statistics: block = tile * 4 + wave row (80 rows of ONE video: valid for video-domain GroupNorms, GnColStat::span)
        igemm_epilogue_rows<MT, NT, EPI>(p, acc, [&](int mt) { return trow(wm * (MT * 16) + mt * 16 + (lane & 15)); },
                                         n0 + wn * (NT * 16) + (lane >> 4) * 4, n0 + wn * (NT * 16), lane, split, tile_m * 4 + wm);
    } else if constexpr (t2) {       // slice mt of wave row wm = 16 pixels of image row ty0 + (5 wm + mt) / 2, columns tx0 + 16 ((5 wm + mt) & 1) ..
        const int rbase = (frame0 * Hd + ty0) * Wd + tx0 + (lane & 15);
        // column statistics: block = tile * 4 + wave row (80 pixels of ONE frame; the tiles of a frame are consecutive)
        igemm_epilogue_rows<MT, NT, EPI>(p, acc, [=](int mt) { const int r16 = 5 * wm + mt; return rbase + (r16 >> 1) * Wd + (r16 & 1) * 16; },
                                         n0 + wn * (NT * 16) + (lane >> 4) * 4, n0 + wn * (NT * 16), lane, split, tile_m * 4 + wm);
    } else {
        igemm_epilogue<MT, NT, EPI>(p, acc, m0 + wm * (MT * 16) + (lane & 15), n0 + wn * (NT * 16) + (lane >> 4) * 4,
                                    n0 + wn * (NT * 16), lane, split);
    }
}

static int g_patch_stamp = 0;      // 0 shipped kernel, 1 stamps, 2 stamps without the weight LDS-DMA (wrong results), 3 stamps without the patch LDS-DMA, 6 ping-pong K loop
void igemm_patch_set_stamp(int mode) { g_patch_stamp = mode; }
int igemm_patch_read_stamps(unsigned long long* out) {
    LAVIE_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_patch_stamps), sizeof(unsigned long long) * 8 * 16));
    return 0;
}

// Column-tile width the halo-patch kernel uses for N output channels: 160 (base widths), 128 (VSR widths), 0 = none.
int igemm_patch_bn(int N) { return N % 160 == 0 ? 160 : N % 128 == 0 ? 128 : 0; }

// Temporal (T,1,1) convolution the kernel's MODE 2 accepts: every segment T = 3 or 5 frame taps, a tile = all F frames of
// 320 / F pixels (F | 320, pixels per frame a multiple of 320 / F), 128-wide column tiles.
static bool patch_temporal_ok(const IgemmParams& p) {
    if (p.tframes <= 0 || p.nseg < 1) return false;
    const int T = p.seg[0].ntaps;
    if (T != 3 && T != 5) return false;
    for (int i = 0; i < p.nseg; ++i)
        if (p.seg[i].ntaps != T) return false;
    if (pt::BM % p.tframes != 0 || p.tpix % (pt::BM / p.tframes) != 0 || p.N % 128 != 0 || p.M % pt::BM != 0 || p.nk % T != 0) return false;
    return p.splits >= 1 && p.splits <= p.nk / T;
}

// Parity form of the conv of a nearest-x2 upsampled image (MODE 3): one 4-tap segment, whole source rows per tile, 160-wide tiles
static bool patch_parity_ok(const IgemmParams& p) {
    if (p.stride != 1 || p.nseg != 1 || p.seg[0].ntaps != 4 || p.tframes > 0) return false;
    if (igemm_patch_bn(p.N) != 160 || p.M % 4 != 0 || (p.M / 4) % pt::BM != 0 || p.nk % 4 != 0) return false;
    const int W = p.Wi, HW = p.Hi * p.Wi;
    if (p.Ho != 2 * p.Hi || p.Wo != 2 * p.Wi || pt::BM % W != 0 || W % 8 != 0) return false;
    if (!(HW % pt::BM == 0 || pt::BM % HW == 0)) return false;
    const int seg_px = HW >= pt::BM ? pt::BM : HW;
    if ((pt::BM / seg_px) * (seg_px / W + 2) * W > pt::PATCH_ROWS) return false;
    return p.splits >= 1 && p.splits <= p.nk / 4;
}

// Fills `p` for y = conv3x3(nearest_x2(x)) + bias in parity form (x [NI, Hi, Wi, C] rows, y [NI, 2Hi, 2Wi, C] rows, wpar from
// launch_pack_conv3x3_parity) including the split-K factor (the caller provides p->slab when splits > 1); false = the halo-patch
// kernel's geometry does not hold and the caller runs the 9-tap gather conv with ups = 1 instead.
bool igemm_setup_parity_upsample(IgemmParams* p, const half_t* x, int C, const half_t* wpar, const float* bias, half_t* y, int NI, int Hi,
                                 int Wi, const half_t* zero) {
    memset(p, 0, sizeof(*p));
    if (C % IGEMM_BK != 0) return false;
    p->W = wpar; p->ldw = 4 * C; p->C = y; p->ldc = C; p->bias = bias; p->rows_per_batch = 1; p->ldr = C;
    p->Hi = Hi; p->Wi = Wi; p->Ho = 2 * Hi; p->Wo = 2 * Wi; p->stride = 1; p->par_ups = 1;
    p->M = NI * p->Ho * p->Wo; p->N = C; p->zero = zero;
    p->nseg = 1;
    p->seg[0].src = x; p->seg[0].C = C; p->seg[0].c0 = 0; p->seg[0].nchunks = C / IGEMM_BK; p->seg[0].ntaps = 4;
    p->nk = 4 * p->seg[0].nchunks;
    // one workgroup per CU and 85 % of a round as everywhere in the family; split-K over whole slabs when the grid is short
    const long wgs = (long)(p->M / 4 / pt::BM) * (C / 160) * 4;
    p->splits = wgs >= 218 ? 1 : (2 * wgs >= 218 && p->seg[0].nchunks >= 10) ? 2 : (p->seg[0].nchunks >= 20 ? 4 : 1);
    return patch_parity_ok(*p);
}

// Whether the halo-patch kernel can run this conv (geometry only; the caller decides on grid fill and split-K).
bool igemm_patch_eligible(const IgemmParams& p) {
    if (p.par_ups) return patch_parity_ok(p);
    if (p.stride != 1 || p.ups != 0) return false;
    if (p.tframes > 0) return patch_temporal_ok(p);
    if (igemm_patch_bn(p.N) == 0 || p.M % pt::BM != 0 || p.nk % 9 != 0) return false;
    for (int i = 0; i < p.nseg; ++i)
        if (p.seg[i].ntaps != 9) return false;
    const int W = p.Wo, HW = p.Ho * p.Wo;
    if (pt::BM % W != 0) {          // 2-D tiles: 10 image rows x 32 columns
        if (W % pt::T2_W != 0 || p.Ho % pt::T2_H != 0) return false;
    } else {
        if (W % 8 != 0) return false;
        if (!(HW % pt::BM == 0 || pt::BM % HW == 0)) return false;
        const int seg_px = HW >= pt::BM ? pt::BM : HW;
        const int patch_rows = (pt::BM / seg_px) * (seg_px / W + 2) * W;
        if (patch_rows > pt::PATCH_ROWS || patch_rows >= (1 << 16)) return false;
    }
    return p.splits >= 1 && p.splits <= p.nk / 9;
}

// MODE 3 (parity form of the upsample conv): shipped K loop only, four parities on gridDim.z
static int launch_patch_parity(const IgemmParams& p, hipStream_t stream) {
    using namespace pt;
    auto kern = igemm_patch_kernel<EPI_LINEAR, 6, 5, 3>;
    static bool attr_set = false;
    if (!attr_set) {
        LAVIE_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        attr_set = true;
    }
    const int grid = (p.M / 4 / BM) * (p.N / 160);
    // counted in the conv class by launch_igemm's scope (executed work: K = 4 Cin per output element); profile class 7 stays the
    // 3x3 instances alone, so that bench.py's `roofline` and rocprofv3's row of igemm_patch_kernel<0, 6, 5, 0> describe the same launches
    hipLaunchKernelGGL(kern, dim3(grid, p.splits, 4), dim3(THREADS), LDS_BYTES, stream, p);
    LAVIE_HIP(hipGetLastError());
    return 0;
}

template <int NT, int MODE>
static int launch_patch_nt(const IgemmParams& p, hipStream_t stream) {
    using namespace pt;
    constexpr int BN = 2 * NT * 16;
    auto kern = g_patch_stamp == 1 ? igemm_patch_kernel<EPI_LINEAR, 1, NT, MODE> : g_patch_stamp == 2 ? igemm_patch_kernel<EPI_LINEAR, 2, NT, MODE>
                : g_patch_stamp == 3 ? igemm_patch_kernel<EPI_LINEAR, 3, NT, MODE> : g_patch_stamp == 4 ? igemm_patch_kernel<EPI_LINEAR, 4, NT, MODE>
                : g_patch_stamp == 5 ? igemm_patch_kernel<EPI_LINEAR, 5, NT, MODE> : g_patch_stamp == 6 ? igemm_patch_kernel<EPI_LINEAR, 0, NT, MODE>
                : igemm_patch_kernel<EPI_LINEAR, 6, NT, MODE>;      // shipped: the software-pipelined K loop (template value 6); 6 here = the ping-pong loop
    static bool attr_set = false;
    if (!attr_set) {
        LAVIE_HIP(hipFuncSetAttribute((const void*)igemm_patch_kernel<EPI_LINEAR, 0, NT, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        LAVIE_HIP(hipFuncSetAttribute((const void*)igemm_patch_kernel<EPI_LINEAR, 1, NT, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        LAVIE_HIP(hipFuncSetAttribute((const void*)igemm_patch_kernel<EPI_LINEAR, 2, NT, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        LAVIE_HIP(hipFuncSetAttribute((const void*)igemm_patch_kernel<EPI_LINEAR, 3, NT, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        LAVIE_HIP(hipFuncSetAttribute((const void*)igemm_patch_kernel<EPI_LINEAR, 4, NT, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        LAVIE_HIP(hipFuncSetAttribute((const void*)igemm_patch_kernel<EPI_LINEAR, 5, NT, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        LAVIE_HIP(hipFuncSetAttribute((const void*)igemm_patch_kernel<EPI_LINEAR, 6, NT, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        attr_set = true;
    }
    const int grid = (p.M / BM) * (p.N / BN);
    const double K = (double)p.nk * IGEMM_BK;
    ProfileScope prof(KC_CONV_PATCH, stream, 2.0 * p.M * p.N * K, 2.0 * ((double)p.M * K / (MODE == 2 ? p.seg[0].ntaps : 9.0) + (double)p.N * K + (double)p.M * p.N),
                      /*kernel_events=*/true);
    if (prof.active()) hipExtLaunchKernelGGL(kern, dim3(grid, p.splits), dim3(THREADS), LDS_BYTES, stream, prof.start(), prof.stop(), 0, p);
    else hipLaunchKernelGGL(kern, dim3(grid, p.splits), dim3(THREADS), LDS_BYTES, stream, p);
    LAVIE_HIP(hipGetLastError());
    return 0;
}

// Launches the halo-patch conv kernel (EPI_LINEAR; the caller runs the split-K reduce).
int launch_igemm_patch(const IgemmParams& p, hipStream_t stream) {
    LAVIE_CHECK(igemm_patch_eligible(p), "igemm_patch: conv geometry not supported by the halo-patch kernel");
    if (p.par_ups) return launch_patch_parity(p, stream);
    if (p.tframes > 0) return launch_patch_nt<4, 2>(p, stream);
    const bool t2 = pt::BM % p.Wo != 0;
    if (igemm_patch_bn(p.N) == 160) return t2 ? launch_patch_nt<5, 1>(p, stream) : launch_patch_nt<5, 0>(p, stream);
    return t2 ? launch_patch_nt<4, 1>(p, stream) : launch_patch_nt<4, 0>(p, stream);
}

}  // namespace lavie
