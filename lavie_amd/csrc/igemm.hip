// Implicit-GEMM on the gfx950 matrix cores: C[M,N] = gather(A)[M,Ktot] * W[N,Ktot]^T (+ epilogue).
//
// One kernel serves every dense contraction of the denoiser: Linear layers and
// 1x1 convs (plain A rows), 3x3 convs with stride 1/2, folded nearest-x2 upsample, two-tensor
// (skip-concat) inputs and a fused 1x1 shortcut (K-segments gathered per output pixel from
// channels-last activations).  Reference ops replaced: `InflatedConv3d.forward` (resnet.py:13-21),
// `nn.Linear` in CrossAttention / FeedForward (attention.py:95-104, 479), `proj_in/proj_out`
// (attention.py:328,356), `conv_shortcut` (resnet.py:175,203), `F.interpolate` + conv (resnet.py:62-72),
// `torch.cat([h, skip])` (unet_blocks.py:538,630).
//
// Design (MI355X_MICROARCH / cdna_hip_programming §5):
//  * 64-wide waves, v_mfma_f32_16x16x32_f16, fp32 accumulation.  The WEIGHT tile is the MFMA A
//    operand and the ACTIVATION tile the B operand, so each lane ends up with 4 consecutive output
//    channels of one token: the epilogue stores 8-byte vectors along the contiguous axis.
//  * Both tiles are staged HBM -> LDS by `global_load_lds_dwordx4` (no VGPR round trip), K-tile 64
//    halfs = 128-B rows.  The LDS image is lane-linear, so the bank-conflict swizzle
//    (16-B slot ^= row & 7) is applied to the per-lane SOURCE address and again on the fragment
//    read (rule 21 of the guide); ds_read_b128 fragment reads are then conflict-free.
//  * Two LDS stages, one barrier per K-tile: loads of tile t+1 are in flight under the MFMAs of tile t.
//  * Out-of-image taps read a 128-B zero page instead of branching.
//  * blockIdx is remapped so that consecutive tiles (which share activation rows) share an XCD L2.
#include <math.h>

#include <type_traits>

#include "igemm.h"
#include "igemm_epilogue.h"
#include "profile.h"

#ifndef LAVIE_SPLITK_MFAST
#define LAVIE_SPLITK_MFAST 1      // A/B switch of the split-K tile order below
#endif

namespace lavie {

template <int WM, int WN, int MT, int NT, int NSTAGE>
struct IgemmTile {
    static constexpr int NW = WM * WN;
    static constexpr int THREADS = 64 * NW;
    static constexpr int BM = WM * MT * 16;
    static constexpr int BN = WN * NT * 16;
    static constexpr int STAGE_BYTES = (BM + BN) * 128;
    static constexpr int LDS_BYTES = NSTAGE * STAGE_BYTES;
    static constexpr int APIECES = BM / 8, WPIECES = BN / 8;          // 1-KiB pieces per stage
    static constexpr int AP = (APIECES + NW - 1) / NW;                 // pieces per wave per stage; when the count
    static constexpr int WP = (WPIECES + NW - 1) / NW;                 // does not divide, the surplus slots re-load
    static constexpr int LOADS = AP + WP;                              // pieces 0.. (same bytes, same place: benign)
    static constexpr int TAB_BYTES = BM * 9 * 4 + IGEMM_MAX_SEG * 6 * 4;  // GATHER: source-pixel table + segment table
    static_assert(LDS_BYTES <= 160 * 1024, "tile does not fit LDS");
};

// ABL (diagnostic builds only, results are wrong): 1 = no MFMA, 2 = no global loads after the prologue,
// 3 = loads + barriers only.  Used by tools/bench_ops.py to find which pipeline paces the loop.
template <int WM, int WN, int MT, int NT, int NSTAGE, bool GATHER, int EPI, int ABL = 0>
__global__ __launch_bounds__(64 * WM * WN, 2) void igemm_kernel(const IgemmParams p) {
    using T = IgemmTile<WM, WN, MT, NT, NSTAGE>;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    // ---- XCD-aware tile order (bijective for any grid size) ----
    const int n_tiles = p.N / T::BN;
    int tile_m, tile_n, split = 0;
    if (gridDim.y == 1 || !LAVIE_SPLITK_MFAST) {
        int bid = blockIdx.x;
        const int nwg = gridDim.x;
        split = blockIdx.y;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
        igemm_tile_of(bid, (int)gridDim.x / n_tiles, n_tiles, (long)p.N * p.nk * IGEMM_BK, &tile_m, &tile_n);
    } else {
        // split-K = the weight-bound convolutions of the deep levels (M 1280, K 11520 .. 23040: 30 - 59 MB of weights for
        // 3 MB of activations).  N-fastest order spread the M tiles that share a weight slice over all eight XCDs and every XCD
        // pulled most of W through the fabric (PMC: 244 / 352 MB read per launch, ~5 TB/s under the kernel).  Here the whole
        // (x, y) grid is linearised as the hardware dispatches it, made XCD-contiguous, and walked M tile fastest, then N
        // tile, then K split: the workgroups that share a weight slice are neighbours inside one XCD's L2.
        const int nwg = gridDim.x * gridDim.y;
        int lin = blockIdx.x + blockIdx.y * gridDim.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = lin & 7;
        lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (lin >> 3);
        const int m_tiles = (int)gridDim.x / n_tiles;
        tile_m = lin % m_tiles;
        const int rest = lin / m_tiles;
        tile_n = rest % n_tiles;
        split = rest / n_tiles;
    }
    const int m0 = tile_m * T::BM;
    const int n0 = tile_n * T::BN;

    // ---- staging addresses ----
    const int lr = lane >> 3;                       // row of this lane inside an 8-row piece
    const int kofs = ((lane & 7) ^ lr) * 8;         // source K offset (halfs) after the slot swizzle

    // GATHER: per-block table of source pixels, tab[row * 9 + tap] = pixel index in the source grid or -1
    // (out of image).  K order is segment > 64-channel chunk > tap: the 9 taps of one channel slab are fetched
    // back to back, so the shifted re-reads of the same cache lines hit in L1/L2 instead of going to the fabric.
    int* tab = reinterpret_cast<int*>(smem + T::LDS_BYTES);
    const half_t* aptr[T::AP];
    int arow[T::AP];
#pragma unroll
    for (int i = 0; i < T::AP; ++i) {
        arow[i] = ((wave + T::NW * i) % T::APIECES) * 8 + lr;
        int m = m0 + arow[i];
        m = m < p.M ? m : p.M - 1;
        aptr[i] = GATHER ? p.zero : p.A + (size_t)m * p.lda + kofs;
    }
    if constexpr (GATHER) {
        const int hw = p.Ho * p.Wo;
        const int Hv = p.Hi << p.ups, Wv = p.Wi << p.ups;
        if (p.tframes > 0) {          // temporal taps: slot t of row m = the same pixel, t - T/2 frames away (or -1)
            const int T_ = p.seg[0].ntaps;
            for (int idx = tid; idx < T::BM * 9; idx += T::THREADS) {
                const int row = idx / 9, tap = idx - row * 9;
                int m = m0 + row;
                m = m < p.M ? m : p.M - 1;
                const int f = (m / p.tpix) % p.tframes;
                const int ff = f + tap - (T_ >> 1);
                tab[idx] = (tap < T_ && (unsigned)ff < (unsigned)p.tframes) ? m + (tap - (T_ >> 1)) * p.tpix : -1;
            }
        } else
        for (int idx = tid; idx < T::BM * 9; idx += T::THREADS) {
            const int row = idx / 9, tap = idx - row * 9;
            int m = m0 + row;
            m = m < p.M ? m : p.M - 1;
            const int n = m / hw;
            const int rem = m - n * hw;
            const int y = rem / p.Wo, x = rem - y * p.Wo;
            const int iy = y * p.stride + tap / 3 - 1, ix = x * p.stride + tap % 3 - 1;
            const bool ok = (unsigned)iy < (unsigned)Hv && (unsigned)ix < (unsigned)Wv;
            tab[idx] = ok ? (n * p.Hi + (iy >> p.ups)) * p.Wi + (ix >> p.ups) : -1;
        }
        __syncthreads();
    }
    const half_t* wptr[T::WP];
#pragma unroll
    for (int i = 0; i < T::WP; ++i) {
        const int n = n0 + ((wave + T::NW * i) % T::WPIECES) * 8 + lr;
        wptr[i] = p.W + (size_t)n * p.ldw + kofs;
    }

    // split-K: this workgroup owns K-tiles [t_begin, t_end) of the flattened K loop
    const int t_begin = (int)((long)p.nk * split / p.splits);
    const int t_end = (int)((long)p.nk * (split + 1) / p.splits);

    // gather cursor (wave-uniform): segment, channel chunk inside it, tap — positioned at t_begin
    // constant-index selects keep the segment descriptors in kernarg SGPRs (a runtime index into the by-value
    // struct would make the compiler spill the whole parameter block to scratch)
    // Segment descriptors live in LDS (written once with constant indices, read with the runtime segment index):
    // a runtime index into the by-value kernel-parameter struct would make the compiler copy the whole parameter
    // block to scratch, and scratch loads are vmcnt-counted VMEM — every one of them inside the K loop would
    // drain the LDS-DMA pipeline (guide §5, trap (b)).  LDS reads only touch lgkmcnt.
    auto sgpr = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
    int* segtab = tab + T::BM * 9;                 // [IGEMM_MAX_SEG][6] : src lo, src hi, C, c0, nchunks, ntaps
    if constexpr (GATHER) {
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
    }
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
    const int nseg = sgpr(p.nseg), tap_major = sgpr(p.tap_major);
    int seg = 0, cchunk = 0, tap = 0;
    IgemmSeg sg = GATHER ? load_seg(0) : IgemmSeg{nullptr, 0, 0, 0, 1};
    if constexpr (GATHER) {
        int skip = t_begin;
        while (skip >= sg.nchunks * sg.ntaps && seg + 1 < nseg) {
            skip -= sg.nchunks * sg.ntaps;
            sg = load_seg(++seg);
        }
        if (tap_major) {
            tap = skip / sg.nchunks;
            cchunk = skip - tap * sg.nchunks;
        } else {
            cchunk = skip / sg.ntaps;
            tap = skip - cchunk * sg.ntaps;
        }
    }

    // Source pointers of the NEXT tile to be issued are computed one step ahead, in two halves: `prepare_read` only
    // issues the pixel-table LDS reads (together with the fragment reads of the current tile, so their latency is
    // shared), `prepare_finish` does the address arithmetic and advances the cursor (placed between the two MFMA
    // blocks of a tile, where the matrix pipe is already fed).  prepare() = both, for the prologue.
    const half_t* nptr[T::AP];
    int pv[T::AP];
    auto prepare_read = [&]() {
        if constexpr (GATHER) {
            const int tp = sg.ntaps == 1 ? 4 : tap;
#pragma unroll
            for (int i = 0; i < T::AP; ++i) pv[i] = tab[arow[i] * 9 + tp];
        }
    };
    auto prepare_finish = [&](int t) {
        if constexpr (GATHER) {
            // 32-bit element offsets: a source tensor holds < 2^31 halfs (checked by the launcher)
            const unsigned cofs = (unsigned)(sg.c0 + cchunk * IGEMM_BK + kofs);
#pragma unroll
            for (int i = 0; i < T::AP; ++i) {
                const half_t* inside = sg.src + ((unsigned)pv[i] * (unsigned)sg.C + cofs);
                nptr[i] = pv[i] >= 0 ? inside : zero_page + kofs;
            }
            if (tap_major) {                        // diagnostic K order: tap outer, channel slab inner
                if (++cchunk == sg.nchunks) {
                    cchunk = 0;
                    if (++tap == sg.ntaps) {
                        tap = 0;
                        if (seg + 1 < nseg) sg = load_seg(++seg);
                    }
                }
            } else if (++tap == sg.ntaps) {
                tap = 0;
                if (++cchunk == sg.nchunks) {
                    cchunk = 0;
                    if (seg + 1 < nseg) sg = load_seg(++seg);
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < T::AP; ++i) nptr[i] = aptr[i] + t * IGEMM_BK;
        }
    };
    auto prepare = [&](int t) { prepare_read(); prepare_finish(t); };
    auto issue = [&](int t, int buf) {
        if (ABL == 2 && t > t_begin + 1) return;
        char* base = smem + buf * T::STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < T::AP; ++i)
            __builtin_amdgcn_global_load_lds(GLB_PTR(nptr[i]), LDS_PTR(base + ((wave + T::NW * i) % T::APIECES) * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < T::WP; ++i)
            __builtin_amdgcn_global_load_lds(GLB_PTR(wptr[i] + t * IGEMM_BK),
                                             LDS_PTR(base + T::BM * 128 + ((wave + T::NW * i) % T::WPIECES) * 1024), 16, 0, 0);
    };

    f32x4 acc[NT][MT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // fragment read offsets inside a stage (bytes)
    const int frow = lane & 15;
    const int fsw = lane & 7;
    const int fg = lane >> 4;
    const int a_frag = (wm * MT * 16 + frow) * 128;
    const int w_frag = T::BM * 128 + (wn * NT * 16 + frow) * 128;

    // Fragment reads are issued for BOTH 32-deep k-steps of a tile before the first MFMA (two register sets), so the
    // LDS latency is exposed once per K-tile instead of once per fragment group; the LDS-DMA of the next tile is
    // issued between the reads and the MFMAs, under that latency.
    auto read_frags = [&](const char* base, int ks, half8_t (&af)[MT], half8_t (&wf)[NT]) {
        const int slot = ((ks * 4 + fg) ^ fsw) * 16;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) af[mt] = *reinterpret_cast<const half8_t*>(base + a_frag + mt * 16 * 128 + slot);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) wf[nt] = *reinterpret_cast<const half8_t*>(base + w_frag + nt * 16 * 128 + slot);
    };
    auto mfma_block = [&](half8_t (&af)[MT], half8_t (&wf)[NT]) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                if constexpr (ABL == 1) {
                    asm volatile("" ::"v"(wf[nt]), "v"(af[mt]));      // keep the LDS reads alive
                } else {
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[nt], af[mt], acc[nt][mt], 0, 0, 0);
                }
            }
    };
    // one K-tile: fragment + table reads -> LDS-DMA issue of a later tile -> MFMAs(k-step 0) -> pointer arithmetic for
    // the tile after that -> MFMAs(k-step 1)
    auto tile_step = [&](int buf, auto&& issue_next, int t_prep) {
        const char* base = smem + buf * T::STAGE_BYTES;
        half8_t af0[MT], wf0[NT], af1[MT], wf1[NT];
        if (ABL != 3) {
            read_frags(base, 0, af0, wf0);
            read_frags(base, 1, af1, wf1);
        }
        const bool prep = t_prep < t_end;
        if (prep) prepare_read();
        issue_next();
        if (ABL != 3) mfma_block(af0, wf0);
        if (prep) prepare_finish(t_prep);
        if (ABL != 3) mfma_block(af1, wf1);
    };
    if constexpr (NSTAGE == 2) {
        // one tile in flight: loads of tile t+1 run under the MFMAs of tile t
        prepare(t_begin);
        issue(t_begin, 0);
        if (t_begin + 1 < t_end) prepare(t_begin + 1);
        int buf = 0;
        for (int t = t_begin; t < t_end; ++t) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's LDS-DMA pieces of tile t have landed
            __syncthreads();   // ... and everybody else's; the other buffer is no longer being read
            tile_step(buf, [&]() { if (t + 1 < t_end) issue(t + 1, buf ^ 1); }, t + 2);
            buf ^= 1;
        }
    } else {
        // NSTAGE-1 tiles in flight across the barrier: counted vmcnt + raw s_barrier (a __syncthreads() would
        // drain the LDS-DMA queue, guide §5 "Pipelining across barriers").  Every wave issues exactly
        // T::LOADS LDS-DMA instructions per tile, so "all but the newest (NSTAGE-2)*LOADS" == tile t landed.
        static_assert(NSTAGE == 3, "counted-wait schedule is written for 3 stages");
        prepare(t_begin);
        issue(t_begin, 0);
        if (t_begin + 1 < t_end) { prepare(t_begin + 1); issue(t_begin + 1, 1); }
        if (t_begin + 2 < t_end) prepare(t_begin + 2);
        int buf = 0;
        for (int t = t_begin; t < t_end; ++t) {
            if (t + 1 < t_end) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(T::LOADS) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                       // tile t visible to all; tile t-1's buffer is free
            asm volatile("" ::: "memory");
            tile_step(buf, [&]() { if (t + 2 < t_end) issue(t + 2, buf == 0 ? 2 : buf - 1); }, t + 3);
            buf = buf == 2 ? 0 : buf + 1;
        }
    }

    igemm_epilogue<MT, NT, EPI>(p, acc, m0 + wm * MT * 16 + (lane & 15), n0 + wn * NT * 16 + (lane >> 4) * 4,
                                n0 + wn * NT * 16, lane, split);
}

// out[m, n] = sum_s slab[s, m, n] (+ bias[n]) (+ bias2[m / rows_per_batch, n]) (+ R[m, n]); fixed summation order.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const IgemmParams p) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const int nv = p.N >> 2;
    if (idx >= (long)p.M * nv) return;
    const int m = (int)(idx / nv), n = (int)(idx - (long)m * nv) * 4;
    f32x4 v = *reinterpret_cast<const f32x4*>(p.slab + (size_t)m * p.N + n);
    for (int s = 1; s < p.splits; ++s) v += *reinterpret_cast<const f32x4*>(p.slab + ((size_t)s * p.M + m) * p.N + n);
    if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + n);
    if (p.bias2) v += *reinterpret_cast<const f32x4*>(p.bias2 + (size_t)(m / p.rows_per_batch) * p.ldb2 + n);
    if (p.R) {
        const half4_t r = *reinterpret_cast<const half4_t*>(p.R + (size_t)m * p.ldr + n);
        v[0] += (float)r[0]; v[1] += (float)r[1]; v[2] += (float)r[2]; v[3] += (float)r[3];
    }
    const half4_t o = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
    *reinterpret_cast<half4_t*>(p.C + (size_t)m * p.ldc + n) = o;
}

// The same with GroupNorm column statistics of the rounded output (IgemmParams::colstat_out): a workgroup = 64 channel quads x 4
// row lanes over one block of COLSTAT_REDUCE_ROWS rows; the row lanes' sums (rows ascending per lane) meet in LDS in lane order.
constexpr int RCS_LANES = 16;       // row lanes: 2 rows each (4 lanes x 8 rows left the M = 1280 reduces at 200 workgroups of 256 threads: 15 us against 7)
__global__ __launch_bounds__(64 * RCS_LANES) void splitk_reduce_cs_kernel(const IgemmParams p) {
    __shared__ f32x4 s_sum[RCS_LANES][64], s_sq[RCS_LANES][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int nv = p.N >> 2;
    const int nq = blockIdx.x * 64 + tx, n = nq * 4;
    const int r0 = blockIdx.y * COLSTAT_REDUCE_ROWS;
    f32x4 cs = {0.f, 0.f, 0.f, 0.f}, cq = {0.f, 0.f, 0.f, 0.f};
    if (nq < nv) {
        const f32x4 bv = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + n) : (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int m = r0 + ty; m < r0 + COLSTAT_REDUCE_ROWS && m < p.M; m += RCS_LANES) {
            f32x4 v = *reinterpret_cast<const f32x4*>(p.slab + (size_t)m * p.N + n);
            for (int s = 1; s < p.splits; ++s) v += *reinterpret_cast<const f32x4*>(p.slab + ((size_t)s * p.M + m) * p.N + n);
            v += bv;
            if (p.bias2) v += *reinterpret_cast<const f32x4*>(p.bias2 + (size_t)(m / p.rows_per_batch) * p.ldb2 + n);
            if (p.R) {
                const half4_t r = *reinterpret_cast<const half4_t*>(p.R + (size_t)m * p.ldr + n);
                v[0] += (float)r[0]; v[1] += (float)r[1]; v[2] += (float)r[2]; v[3] += (float)r[3];
            }
            const half4_t o = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
            *reinterpret_cast<half4_t*>(p.C + (size_t)m * p.ldc + n) = o;
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float f = (float)o[r]; cs[r] += f; cq[r] += f * f; }
        }
    }
    s_sum[ty][tx] = cs;
    s_sq[ty][tx] = cq;
    __syncthreads();
    if (ty == 0 && nq < nv) {
        f32x4 a = s_sum[0][tx], b = s_sq[0][tx];
#pragma unroll
        for (int y = 1; y < RCS_LANES; ++y) { a += s_sum[y][tx]; b += s_sq[y][tx]; }
        float* dst = p.colstat_out + cs_index((size_t)blockIdx.y, n, 0, p.N);
        *reinterpret_cast<f32x4*>(dst) = a;
        *reinterpret_cast<f32x4*>(dst + 4) = b;
    }
}
// fixed-order sum of the split-K slabs + bias / residual / rounding (+ column statistics when the caller asked for them)
static int launch_splitk_reduce(const IgemmParams& p, hipStream_t stream) {
    if (p.colstat_out) {
        hipLaunchKernelGGL(splitk_reduce_cs_kernel, dim3(cdiv(p.N / 4, 64), cdiv(p.M, COLSTAT_REDUCE_ROWS)), dim3(64 * RCS_LANES), 0, stream, p);
    } else {
        const long total = (long)p.M * (p.N / 4);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, p);
    }
    LAVIE_HIP(hipGetLastError());
    return 0;
}

// Split-K factor for an under-filled grid.  The 128-row tiles run 2 workgroups per CU (512 slots); a grid of
// `blocks` tiles takes ceil(blocks*S/512) rounds of (nk/S + fixed) K-steps plus a reduce pass over S slabs.
static int g_force_tile = 0;   // see igemm_force_tile() in igemm.h
// measured: at 5 K-tiles (K = 320) the GEGLU epilogue, with no second workgroup to hide it, loses; from 8 K-tiles on the
// ping-pong variant wins or ties (K = 640 of the base model: +0.2 %, K = 512 of the VSR UNet: GEMM class -3.5 %)
static int g_geglu_pp_min_nk = 8;
static int g_force_splits = 0;
void igemm_force_splits(int s) { g_force_splits = s; }

// ---- 160x320 ping-pong kernel (igemm_pp.hip): when it is used.  Measured on this model's shapes (tools/bench_ops.py
// pp_splits, check_pp): it is 15-25 % faster than the 128-row kernel whenever its grid quantises onto the 256 CUs
// (one workgroup per CU) and the K loop is long enough to amortise its 60 KiB prologue; an under-filled grid or a
// short K loop loses.  Rule: N % 320 == 0, >= 10 K-tiles, the grid's last round at least 85 % full, and with
// split-K at least 45 K-tiles per split (the fp32 slab round trip must stay small beside the loop).
// tile width of the ping-pong kernel for this N: 320, or 256 (NT = 4) for the power-of-two widths of the VSR UNet; 0 = none
static int pp_bn(int N) { return N % 320 == 0 ? 320 : (N % 256 == 0 ? 256 : 0); }
static bool pp_fits(int M, int N, int nk, int s) {
    const int bn = pp_bn(N);
    if (bn == 0 || nk < 10 || s < 1 || s > nk) return false;      // (5 K-tiles at N = 320 measured +0.2 %: within noise, not taken)
    if (s > 1 && nk / s < 40) return false;       // (40: the L2 ff2 GEMM, 80 K-tiles, 128 tiles: 91 -> 79 us with two splits)
    const double r = (double)cdiv(M, 160) * (N / bn) * s / 256.0;
    return r / ceil(r) >= 0.85;
}
// split-K factor with which the ping-pong kernel should run this problem, 0 = do not use it
static int pp_plan(int M, int N, int nk, int epilogue) {
    if (epilogue != EPI_LINEAR || (g_force_tile & 0xF) == 4 || (g_force_tile & 0xF) == 1) return 0;
    for (int s = 1; s <= 4; ++s)
        if (pp_fits(M, N, nk, s)) return s;
    return 0;
}

// ---- 320x160 halo-patch conv kernel (igemm_patch.hip): 5-8 % faster than the ping-pong kernel on the convs it accepts
// (tools/check_patch.py); same grid rule with its own tile: one workgroup per CU, last round at least 85 % full,
// split-K (over whole slabs) only with at least 45 K-tiles per split.
static bool patch_fits(int M, int N, int nk, int s, int ntaps = 9) {
    // temporal convs (ntaps 3 / 5): the 128-wide tile only, and only from 1024 channels on — tools/bench_ops.py tconv measured
    // the temporal mode equal to the ping-pong kernel within +-5 % at 256 / 512 channels (short K loops: 12 - 40 K-tiles, the
    // activation tile is staged once per N tile instead of once) and 30 % ahead at 1024
    if (ntaps != 9 && N < 1024) return false;
    const int bn = ntaps == 9 ? igemm_patch_bn(N) : (N % 128 == 0 ? 128 : 0);
    if (M % 320 != 0 || bn == 0 || nk % ntaps != 0 || s < 1 || s > nk / ntaps) return false;
    if (s > 1 && nk / s < 45) return false;
    const double r = (double)(M / 320) * (N / bn) * s / 256.0;
    return r / ceil(r) >= 0.85;
}
static int pt_bm_rows() { return 320; }      // the halo-patch kernel's tile height (igemm_patch.hip pt::BM)
static bool patch_allowed() { const int lo = g_force_tile & 0xF; return lo == 0 || lo == 5 || lo == 8 || lo == 9; }

static int plan_splits(int M, int N, int nk, int epilogue, bool plain);
static bool ppx_plan_shape(int M, int N, int nk, int epilogue);

int igemm_plan_splits_gather(const IgemmParams& p) {
    if (g_force_splits == 0 && patch_allowed()) {
        IgemmParams q = p;
        for (int s = 1; s <= 4; ++s) {
            q.splits = s;
            if (patch_fits(p.M, p.N, p.nk, s, p.tframes > 0 ? p.seg[0].ntaps : 9) && igemm_patch_eligible(q)) return s;
        }
    }
    return plan_splits(p.M, p.N, p.nk, EPI_LINEAR, false);
}

bool igemm_patch_planned(const IgemmParams& p) {
    const int lo = ((g_force_tile & 0xF) == 8 || (g_force_tile & 0xF) == 9) ? 0 : (g_force_tile & 0xF);
    return igemm_patch_eligible(p) && (lo == 5 || (lo == 0 && patch_fits(p.M, p.N, p.nk, p.splits, p.tframes > 0 ? p.seg[0].ntaps : 9)));
}

int igemm_plan_splits(int M, int N, int nk, int epilogue) { return plan_splits(M, N, nk, epilogue, true); }
bool igemm_takes_ppx(int M, int N, int nk, int epilogue) { return ppx_plan_shape(M, N, nk, epilogue); }

static int plan_splits(int M, int N, int nk, int epilogue, bool plain) {
    if (epilogue != EPI_LINEAR || N % 64 != 0) return 1;
    if (plain && g_force_splits == 0 && ppx_plan_shape(M, N, nk, epilogue)) return 1;     // the persistent kernel never splits K
    if (g_force_splits > 0) return g_force_splits <= nk ? g_force_splits : 1;
    if (const int s = pp_plan(M, N, nk, epilogue)) return s;
    const long blocks = (long)cdiv(M, 128) * cdiv(N, 160);
    if (blocks >= 1024) return 1;
    double best = 1e30;
    int best_s = 1;
    for (int s = 1; s <= 8; ++s) {
        if (s > 1 && nk / s < 12) break;
        const double rounds = (double)((blocks * s + 511) / 512);
        double cost = rounds * ((double)nk / s + 6.0);
        if (s > 1) cost += 1.5e-6 * s * (double)M * N;     // slab write + read (~8 s M N bytes at ~4 TB/s) in K-step units of ~1.3 us
        if (cost < best * 0.93) { best = cost; best_s = s; }
    }
    return best_s;
}

// ---- persistent ping-pong kernel (igemm_ppx.hip) for plain GEMMs: forced (mode 7) or by the measured rule
static bool ppx_plan_shape(int M, int N, int nk, int epilogue) {
    const int lo = g_force_tile & 0xF;
    if (M % 160 != 0 || nk < 5 || (epilogue == EPI_GEGLU ? N % 256 != 0 : (N % 320 != 0 && N % 256 != 0))) return false;   // = igemm_ppx_eligible's shape part
    if (lo == 7) return true;
    if (lo == 9) {      // A/B switch (round 4): the automatic rule WITHOUT the GEGLU GEMMs on the persistent kernel
        const long tiles9 = (long)(M / 160) * (N / pp_bn(N));
        return tiles9 >= 256 && nk <= 10 && epilogue == EPI_LINEAR;
    }
    if (lo != 0 && lo != 6) return false;
    // Measured (tools/check_ppx.py, profiles/r02_ppx_shapes.txt): the persistent kernel wins where the K loop is short and
    // every CU gets at least one whole tile — the L0 GEMMs with K = 320 (N = 320: -14 .. -23 %, QKV -7 %, GEGLU -10 %) and
    // the L1 K = 640 linear ones (-3 .. -10 %); it loses on long K loops (the one-tile kernels' second workgroup per CU
    // hides their epilogue better) and on under-filled grids (M = 5120: 128 tiles).
    // GEGLU: -10 % in the operator benchmark but +3 % inside the UNet (rocprofv3, profiles/r02_a_kernel_summary.md: its
    // VALU-heavy epilogue stalls both groups once per tile): left to the one-tile kernels.
    // Round 4, re-measured inside the forward (tools/ab_tile.py, profiles/r04_ab_geglu_on_persistent_kernel.txt, same box,
    // interleaved, A B B A): with the level-1 GEGLU GEMM (K = 640) on the persistent kernel the forward is 0.23 ms shorter (20.18 vs
    // 20.41 ms) at an unchanged class time — its stores are spread over the launch instead of ending it in one burst: taken (mode 9
    // runs the automatic rule without it for A/B).  The same run order showed that EXACTLY one tile per CU (level-1 N = 640) still
    // belongs here and not on the one-tile kernel, although that is 11 % faster in isolation.
    const long tiles = (long)(M / 160) * (N / pp_bn(N));
    if (tiles < 256 || nk > 10) return false;
    return epilogue == EPI_LINEAR || lo == 0;
}
static bool ppx_plan(const IgemmParams& p, int epilogue) {
    return p.splits == 1 && igemm_ppx_eligible(p, epilogue) && ppx_plan_shape(p.M, p.N, p.nk, epilogue);
}

template <int WM, int WN, int MT, int NT, int NSTAGE, bool GATHER, int EPI, int ABL = 0>
static int launch_tile(const IgemmParams& p, hipStream_t stream) {
    using T = IgemmTile<WM, WN, MT, NT, NSTAGE>;
    auto kern = igemm_kernel<WM, WN, MT, NT, NSTAGE, GATHER, EPI, ABL>;
    constexpr int lds = T::LDS_BYTES + (GATHER ? T::TAB_BYTES : 0);
    static_assert(lds <= 160 * 1024, "tile + pixel table do not fit LDS");
    static bool attr_set = false;   // one per instantiation
    if (!attr_set) {
        LAVIE_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_set = true;
    }
    const int grid = cdiv(p.M, T::BM) * (p.N / T::BN);
    hipLaunchKernelGGL(kern, dim3(grid, p.splits), dim3(T::THREADS), lds, stream, p);
    LAVIE_HIP(hipGetLastError());
    if (p.splits > 1) return launch_splitk_reduce(p, stream);
    return 0;
}

// Tile width by grid quantisation: the 4-wave tiles run 2 workgroups per CU (512 slots per round); a narrower
// tile is a little less efficient per flop (exponent 0.9) but can save a whole round on short grids.
static int igemm_pick_bn(int M, int N, int splits) {
    double best = 1e30;
    int bn = 0;
    const int cand[3] = {160, 128, 64};
    for (int i = 0; i < 3; ++i) {
        if (N % cand[i] != 0) continue;
        const long blocks = (long)cdiv(M, 128) * (N / cand[i]) * splits;
        const double cost = (double)((blocks + 511) / 512) * pow(cand[i] / 160.0, 0.9);
        if (cost < best * 0.97) { best = cost; bn = cand[i]; }
    }
    const int lo = g_force_tile & 0xF;
    if ((lo == 1 || lo == 3 || g_force_tile >= 0x10) && N % 160 == 0) bn = 160;   // forced modes: widest
    return bn;
}

__global__ void rowstat_finalize_kernel(const float* __restrict__ partials, int slots, int M, float inv_len, float eps,
                                        float* __restrict__ out) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    const float* st = partials + (size_t)m * slots * 2;
    float sum = 0.f, sq = 0.f;
    for (int j = 0; j < slots; ++j) { sum += st[2 * j]; sq += st[2 * j + 1]; }
    const float mean = sum * inv_len;
    out[(size_t)m * 2] = mean;
    out[(size_t)m * 2 + 1] = rsqrtf(fmaxf(sq * inv_len - mean * mean, 0.f) + eps);
}

int launch_rowstat_finalize(const float* partials, int slots, int M, int row_len, float eps, float* out, hipStream_t stream) {
    hipLaunchKernelGGL(rowstat_finalize_kernel, dim3(cdiv(M, 256)), dim3(256), 0, stream, partials, slots, M,
                       1.0f / (float)row_len, eps, out);
    LAVIE_HIP(hipGetLastError());
    return 0;
}

// Columns per row-statistics slot (= the wave tile width 16*NT) launch_igemm uses for a plain, unsplit EPI_LINEAR
// GEMM; two waves share a tile's columns.
int igemm_rowstat_cols(int M, int N, int nk) {
    if (ppx_plan_shape(M, N, nk, EPI_LINEAR)) return pp_bn(N) / 4;                                              // persistent ping-pong wave tile
    if ((g_force_tile & 0xF) == 3 ? N % 320 == 0 : (pp_plan(M, N, nk, EPI_LINEAR) == 1)) return pp_bn(N) / 4;   // ping-pong wave tile
    return igemm_pick_bn(M, N, 1) / 2;
}

// Mirror of launch_igemm's kernel choice (same tests in the same order): rows per column-statistics block, 0 = none.
int igemm_colstat_rows(const IgemmParams& p, bool gather, int epilogue) {
    if (epilogue != EPI_LINEAR || p.N % 4 != 0) return 0;
    if (p.splits > 1) return COLSTAT_REDUCE_ROWS;                 // whatever kernel fills the slabs, the reduce kernel writes the statistics
    const int lo = ((g_force_tile & 0xF) == 8 || (g_force_tile & 0xF) == 9) ? 0 : (g_force_tile & 0xF);
    if (p.par_ups) return 80;                                     // source-row blocks, one set per parity
    if (!gather && ppx_plan(p, epilogue)) return 80;
    if (gather && igemm_patch_eligible(p) && (lo == 5 || (lo == 0 && patch_fits(p.M, p.N, p.nk, p.splits, p.tframes > 0 ? p.seg[0].ntaps : 9)))) {
        // tiles of whole image rows (MODE 0): a wave tile = 80 contiguous rows.  2-D tiles (MODE 1) and temporal-conv tiles (MODE 2):
        // 80 scattered rows of one frame / one video, numbered tile * 4 + wave (igemm_colstat_span)
        return 80;
    }
    if ((lo == 3 && p.N % 320 == 0) || ((lo == 0 || lo == 6) && pp_fits(p.M, p.N, p.nk, p.splits))) return 80;
    return igemm_pick_bn(p.M, p.N, p.splits) != 0 ? 64 : 0;
}

int igemm_colstat_span(const IgemmParams& p, bool gather, int rows) {
    if (p.splits > 1) return rows;                                 // the reduce kernel walks contiguous output rows
    const int lo = ((g_force_tile & 0xF) == 8 || (g_force_tile & 0xF) == 9) ? 0 : (g_force_tile & 0xF);
    const bool patch = gather && igemm_patch_eligible(p) &&
                       (p.par_ups || lo == 5 || (lo == 0 && patch_fits(p.M, p.N, p.nk, p.splits, p.tframes > 0 ? p.seg[0].ntaps : 9)));
    if (!patch) return rows;
    if (p.par_ups) {                                               // 80 source rows -> their 4 output parities, one set each
        const int hw = p.Hi * p.Wi;
        return 4 * (hw % rows == 0 ? hw : (rows % hw == 0 ? rows : 320));
    }
    if (p.tframes > 0) return p.tframes * p.tpix;                  // a tile = every frame of 320 / F pixels of ONE video
    if (pt_bm_rows() % p.Wo != 0) return p.Ho * p.Wo;              // 2-D tile: 10 rows x 32 columns of ONE frame
    const int hw = p.Ho * p.Wo;                                    // whole image rows: contiguous; frames smaller than a wave tile nest
    return hw % rows == 0 ? rows : (rows % hw == 0 ? rows : 320);
}

int launch_igemm(const IgemmParams& p, bool gather, int epilogue, hipStream_t stream) {
    LAVIE_CHECK(p.M > 0 && p.N > 0 && p.nk > 0, "igemm: empty problem M=%d N=%d nk=%d", p.M, p.N, p.nk);
    const double K = (double)p.nk * IGEMM_BK;
    // algorithmic work: 2 M N K flops; bytes = every operand element once (A incl. im2col reuse counted once)
    ProfileScope prof(gather ? KC_CONV3X3 : KC_LINEAR, stream, 2.0 * p.M * p.N * K,
                      2.0 * ((double)p.M * K / (gather ? 9.0 : 1.0) + (double)p.N * K + (double)p.M * p.N));
    LAVIE_CHECK(p.N % 4 == 0 && p.ldc % 4 == 0, "igemm: N and ldc must be multiples of 4");
    LAVIE_CHECK(!(p.rowstat_out || p.ln_stats || p.ln_partials) || (p.splits == 1 && !gather), "igemm: LayerNorm folding needs a plain, unsplit GEMM");
    LAVIE_CHECK(!p.ln_partials || (p.ln_slots >= 1 && !ppx_plan(p, epilogue)), "igemm: row-statistics partials cannot feed the persistent kernel (finalize them)");
    LAVIE_CHECK(p.splits >= 1 && p.splits <= p.nk && (p.splits == 1 || (p.slab && epilogue == EPI_LINEAR)),
                "igemm: bad split-K setup (splits=%d)", p.splits);
    const int lo = ((g_force_tile & 0xF) == 8 || (g_force_tile & 0xF) == 9) ? 0 : (g_force_tile & 0xF);      // 8 = automatic without the persistent kernel
    auto reduce_splits = [&]() -> int {          // fixed-order sum of the split-K slabs + bias / residual / rounding
        return p.splits > 1 ? launch_splitk_reduce(p, stream) : 0;
    };
    // the block height the caller sized the column-statistics buffer for must be the one the kernel chosen below writes
    LAVIE_CHECK(!p.colstat_out || (epilogue == EPI_LINEAR && p.colstat_rows > 0 && p.colstat_rows == igemm_colstat_rows(p, gather, epilogue)),
                "igemm: column statistics planned for %d-row blocks, this launch writes %d", p.colstat_rows, igemm_colstat_rows(p, gather, epilogue));
    if (p.par_ups) {             // parity form of an upsample conv: the halo-patch kernel is the only one that runs it
        LAVIE_CHECK(gather && epilogue == EPI_LINEAR && igemm_patch_eligible(p), "igemm: parity upsample conv outside the halo-patch kernel's geometry");
        if (int rc = launch_igemm_patch(p, stream)) return rc;
        return reduce_splits();
    }
    if (epilogue == EPI_GEGLU) {
        LAVIE_CHECK(p.N % 128 == 0, "igemm: GEGLU needs N %% 128 == 0 (N=%d)", p.N);
        LAVIE_CHECK(!p.R && !p.bias2, "igemm: GEGLU epilogue takes no residual / per-batch bias");
        LAVIE_CHECK(!gather, "igemm: GEGLU epilogue is only built for plain A rows");
        if (ppx_plan(p, epilogue)) return launch_igemm_ppx(p, epilogue, stream);
        // 160x256 ping-pong variant: same grid rule as the 160x320 kernel, from g_geglu_pp_min_nk K-tiles on
        const double r = (double)cdiv(p.M, 160) * (p.N / 256) / 256.0;
        if (p.N % 256 == 0 && (lo == 3 || ((lo == 0 || lo == 6) && p.nk >= g_geglu_pp_min_nk && r / ceil(r) >= 0.85)))
            return launch_igemm_pp_geglu(p, stream);
        return launch_tile<2, 2, 4, 4, 2, false, EPI_GEGLU>(p, stream);
    }
    // the row-statistics slot width the caller allocated for must be the wave-tile width of the kernel chosen below (the shape
    // rule of igemm_rowstat_cols and the eligibility tests here are separate code: a mismatch would write out of range)
    auto slots_ok = [&](int cols) { return !p.rowstat_out || p.rowstat_cols == 0 || p.rowstat_cols == cols; };
    if (!gather && ppx_plan(p, epilogue)) {
        LAVIE_CHECK(slots_ok(pp_bn(p.N) / 4), "igemm: row-statistics slots sized for %d columns, persistent kernel writes %d", p.rowstat_cols, pp_bn(p.N) / 4);
        return launch_igemm_ppx(p, epilogue, stream);
    }
    // halo-patch conv kernel: forced (mode 5) or whenever its grid rule holds at this split factor
    if (gather && igemm_patch_eligible(p) && (lo == 5 || (lo == 0 && patch_fits(p.M, p.N, p.nk, p.splits, p.tframes > 0 ? p.seg[0].ntaps : 9)))) {
        if (int rc = launch_igemm_patch(p, stream)) return rc;
        return reduce_splits();
    }
    // 160x320 ping-pong kernel: forced (mode 3) or whenever the planner's rule holds for this problem at its split factor
    if ((lo == 3 && p.N % 320 == 0) || ((lo == 0 || lo == 6) && pp_fits(p.M, p.N, p.nk, p.splits))) {
        LAVIE_CHECK(slots_ok(pp_bn(p.N) / 4), "igemm: row-statistics slots sized for %d columns, ping-pong kernel writes %d", p.rowstat_cols, pp_bn(p.N) / 4);
        if (int rc = launch_igemm_pp(p, gather, stream)) return rc;
        return reduce_splits();
    }
    // 128-row kernel: two independent workgroups per CU, tile width by grid quantisation
    const int bn = igemm_pick_bn(p.M, p.N, p.splits);
    LAVIE_CHECK(bn != 0, "igemm: N=%d is not a multiple of 64", p.N);
    LAVIE_CHECK(slots_ok(bn / 2), "igemm: row-statistics slots sized for %d columns, 128-row kernel writes %d", p.rowstat_cols, bn / 2);
    if (bn == 160) {
        if (g_force_tile >= 0x10 && lo == 1) {          // diagnostic ablations of the 128x160 tile (results wrong)
            const int abl = g_force_tile >> 4;
            if (gather) {
                if (abl == 1) return launch_tile<2, 2, 4, 5, 2, true, EPI_LINEAR, 1>(p, stream);
                if (abl == 2) return launch_tile<2, 2, 4, 5, 2, true, EPI_LINEAR, 2>(p, stream);
                return launch_tile<2, 2, 4, 5, 2, true, EPI_LINEAR, 3>(p, stream);
            }
            if (abl == 1) return launch_tile<2, 2, 4, 5, 2, false, EPI_LINEAR, 1>(p, stream);
            if (abl == 2) return launch_tile<2, 2, 4, 5, 2, false, EPI_LINEAR, 2>(p, stream);
            return launch_tile<2, 2, 4, 5, 2, false, EPI_LINEAR, 3>(p, stream);
        }
        return gather ? launch_tile<2, 2, 4, 5, 2, true, EPI_LINEAR>(p, stream)
                      : launch_tile<2, 2, 4, 5, 2, false, EPI_LINEAR>(p, stream);
    }
    if (bn == 128)
        return gather ? launch_tile<2, 2, 4, 4, 2, true, EPI_LINEAR>(p, stream)
                      : launch_tile<2, 2, 4, 4, 2, false, EPI_LINEAR>(p, stream);
    return gather ? launch_tile<2, 2, 4, 2, 2, true, EPI_LINEAR>(p, stream)
                  : launch_tile<2, 2, 4, 2, 2, false, EPI_LINEAR>(p, stream);
}

void igemm_pp_ablate(int a);
void igemm_ppx_ablate(int a);
void igemm_patch_set_stamp(int mode);
void igemm_force_tile(int mode) {
    g_force_tile = mode;
    igemm_ppx_ablate((mode & 0xF) == 7 ? mode >> 4 : 0);
    igemm_pp_ablate((mode & 0xF) == 3 ? mode >> 4 : 0);
    igemm_patch_set_stamp(mode == 0x75 ? 1 : mode == 0x85 ? 2 : mode == 0x95 ? 3 : mode == 0xA5 ? 4 : mode == 0xB5 ? 5 : (mode >> 4) == 0xC ? 6 : 0);
    if ((mode >> 4) == 0xC) g_force_tile = mode & 0xF;      // 0xC0 / 0xC5: the halo-patch kernel's ping-pong K loop (A/B against the shipped software-pipelined one), kernel choice as the low nibble says
}

}  // namespace lavie
