// 160x320 implicit-GEMM tile, 8 waves in two phase-alternating groups ("ping-pong"), gfx950.
//
// Why: the 128x160 / 4-wave kernel of igemm.hip moves 14 B of LDS-DMA and 0.45 fragment reads per MFMA; both the
// L2 -> LDS stream and the LDS reads sit beside the MFMAs of the SAME wave, so the matrix pipe idles while a wave
// issues its loads.  Here every channel count of the model is a multiple of 320 and every token count a multiple
// of 160, so a 160x320 block tile quantises exactly onto 256 CUs (L0: 2 tiles per CU, L1: 1) and needs 9.4 B of
// LDS-DMA and 0.4 fragment reads per MFMA.  The 8 waves (wave tile 80x80 = 5x5 MFMA tiles, 100 accumulator
// registers) form two groups, waves 0-3 and 4-7 = the two co-resident waves of every SIMD
// (MI355X_MICROARCH "Two waves per SIMD").  The groups run the same program HALF A K-STEP APART, separated by raw
// s_barriers: while one group reads fragments from LDS and issues LDS-DMA (R phase), its SIMD partner issues
// 25 MFMAs (M phase).  Per K-tile of 64 each group runs R(ks0) M(ks0) R(ks1) M(ks1); "epoch" = the interval between
// two barriers, G0 runs R(t,0) in epoch 4t, G1 one epoch later.
//
// LDS (one __shared__ array, swizzled as in igemm.hip): W 2 stages x 320 rows, A 3 stages x 160 rows, 128-B rows.
// Group g computes output columns [160 g, 160 g + 160): it alone reads the W rows of its half ("W_lo" / "W_hi"), both
// groups read A.  An LDS-DMA piece (1 KiB) costs its wave ~60-100 issue cycles, so the 60 pieces of a K-tile are
// spread 4 / 4 / 4 / 3 over the four R phases of a tile, each placed where its target is already free and at least
// two epochs before its first read (the third A stage is what makes the last row possible):
//   G0 R(t,0) epoch 4t  : W_lo(t+1) pieces 0-15                       (last read by G0 in epoch 4t-2)
//   G1 R(t,0) epoch 4t+1: W_lo(t+1) 16-19, W_hi(t+1) 16-19, A(t+2) 0-7
//   G0 R(t,1) epoch 4t+2: W_hi(t+1) pieces 0-15                       (last read by G1 in epoch 4t-1)
//   G1 R(t,1) epoch 4t+3: A(t+2) pieces 8-19                          (stage of tile t-1, last read in epoch 4t-1)
// Waits are counted (never vmcnt(0) in the loop): G0 vmcnt(4) at the end of R(t,0) [W_hi(t) landed] and of M(t,1)
// [W_lo(t+1) landed]; G1 vmcnt(5) at the end of R(t,1) [everything but the five A(t+2) pieces landed].
#include <type_traits>

#include "igemm.h"
#include "igemm_epilogue.h"

namespace lavie {

namespace pp {
constexpr int MT = 5;
constexpr int BM = 160, THREADS = 512;
constexpr int A_BYTES = BM * 128;                       // one A stage: 20,480
constexpr int A_STAGES = 3, W_STAGES = 2;
constexpr int W_BASE = A_STAGES * A_BYTES;              // A stages first, then W stages
constexpr int TAB_BYTES = BM * 9 * 4 + IGEMM_MAX_SEG * 6 * 4;
// NT = 16-wide column tiles per wave: 5 -> 160x320 block tile (every EPI_LINEAR GEMM with N % 320 == 0),
// 4 -> 160x256 (GEGLU: value / gate tile pairs need an even NT; N = 8 C is a multiple of 256)
template <int NT>
struct Geo {
    static constexpr int BN = 4 * NT * 16;
    static constexpr int HALF_ROWS = 2 * NT * 16;       // W rows read by one group
    static constexpr int HALF_PIECES = HALF_ROWS / 8;   // 20 or 16: group 0 stages 16 of each half, group 1 the rest
    static constexpr int W_BYTES = BN * 128;
    static constexpr int LDS_BYTES = W_BASE + W_STAGES * W_BYTES;
    static_assert(LDS_BYTES + TAB_BYTES <= 160 * 1024, "does not fit LDS");
};
}  // namespace pp

// ABL (diagnostic builds, wrong results): 1 = no MFMA, 2 = no LDS-DMA after the prologue, 3 = MFMAs and barriers only;
// 4 = s_setprio 1 around the MFMA blocks (results correct; measured 0-10 % slower: it starves the partner's LDS-DMA issue)
template <bool GATHER, int EPI, int ABL = 0, int NT = 5>
__global__ __launch_bounds__(pp::THREADS, 2) void igemm_pp_kernel(const IgemmParams p) {
    using namespace pp;
    using G = Geo<NT>;
    constexpr int BN = G::BN, W_BYTES = G::W_BYTES, LDS_BYTES = G::LDS_BYTES, HALF_ROWS = G::HALF_ROWS;
    constexpr bool G1_W = G::HALF_PIECES > 16;            // group 1 stages W pieces 16.. of each half (NT = 5 only)
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;                    // 0: leading group, 1: trailing group (SIMD partners)
    const int q = wave & 3;
    const int wm = q >> 1, wn = grp * 2 + (q & 1);

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
    const int t_begin = (int)((long)p.nk * split / p.splits);
    const int t_end = (int)((long)p.nk * (split + 1) / p.splits);

    const int lr = lane >> 3;
    const int kofs = ((lane & 7) ^ lr) * 8;         // source K offset (halfs) after the slot swizzle

    auto sgpr = [](int v) { return __builtin_amdgcn_readfirstlane(v); };

    // ---- GATHER tables (as igemm.hip): tab[row * 9 + tap] = source pixel or -1; segment descriptors
    int* tab = reinterpret_cast<int*>(smem + LDS_BYTES);
    int* segtab = tab + BM * 9;
    if constexpr (GATHER) {
        const int hw = p.Ho * p.Wo;
        const int Hv = p.Hi << p.ups, Wv = p.Wi << p.ups;
        if (p.tframes > 0) {          // temporal taps (IgemmParams::tframes): slot t = the same pixel, t - T/2 frames away
            const int T_ = p.seg[0].ntaps;
            for (int idx = tid; idx < BM * 9; idx += THREADS) {
                const int row = idx / 9, tap = idx - row * 9;
                int m = m0 + row;
                m = m < p.M ? m : p.M - 1;
                const int ff = (m / p.tpix) % p.tframes + tap - (T_ >> 1);
                tab[idx] = (tap < T_ && (unsigned)ff < (unsigned)p.tframes) ? m + (tap - (T_ >> 1)) * p.tpix : -1;
            }
        } else
        for (int idx = tid; idx < BM * 9; idx += THREADS) {
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
    const int nseg = sgpr(p.nseg);

    // ---- staging roles (pieces of 8 rows = 1 KiB).  Group 1 owns the A tile: piece j of wave q = rows (q + 4 j) * 8.
    int arow[5];
    const half_t* aptr[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        arow[j] = (q + 4 * j) * 8 + lr;
        int m = m0 + arow[j];
        m = m < p.M ? m : p.M - 1;
        aptr[j] = GATHER ? p.zero : p.A + (size_t)m * p.lda + kofs;
    }
    // W pieces.  Group 0 wave q: W rows (q + 4 j) * 8 in R(.,0) and 160 + (q + 4 j) * 8 in R(.,1), j = 0..3;
    // group 1 wave q: W rows (16 + q) * 8 and 160 + (16 + q) * 8 in R(.,0).
    const half_t* wp[4];
    const half_t* wp2[4];
    if (grp == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            wp[j] = p.W + (size_t)(n0 + (q + 4 * j) * 8 + lr) * p.ldw + kofs;
            wp2[j] = p.W + (size_t)(n0 + HALF_ROWS + (q + 4 * j) * 8 + lr) * p.ldw + kofs;
        }
    } else {
        const int r16 = G1_W ? (16 + q) * 8 : 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            wp[j] = p.W + (size_t)(n0 + r16 + lr) * p.ldw + kofs;
            wp2[j] = p.W + (size_t)(n0 + HALF_ROWS + r16 + lr) * p.ldw + kofs;
        }
    }

    // ---- gather cursor (group 1 only uses it): segment, channel slab, tap — positioned at t_begin
    int seg = 0, cchunk = 0, tap = 0;
    IgemmSeg sg = GATHER ? load_seg(0) : IgemmSeg{nullptr, 0, 0, 0, 1};
    if constexpr (GATHER) {
        int skip = t_begin;
        while (skip >= sg.nchunks * sg.ntaps && seg + 1 < nseg) {
            skip -= sg.nchunks * sg.ntaps;
            sg = load_seg(++seg);
        }
        cchunk = skip / sg.ntaps;
        tap = skip - cchunk * sg.ntaps;
    }
    const half_t* nptr[5];
    int pv[5];
    auto prepare_read = [&]() {
        if constexpr (GATHER) {
            const int tp = sg.ntaps == 1 ? 4 : tap;
#pragma unroll
            for (int j = 0; j < 5; ++j) pv[j] = tab[arow[j] * 9 + tp];
        }
    };
    auto prepare_finish = [&](int t) {
        if constexpr (GATHER) {
            const unsigned cofs = (unsigned)(sg.c0 + cchunk * IGEMM_BK + kofs);
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const half_t* inside = sg.src + ((unsigned)pv[j] * (unsigned)sg.C + cofs);
                nptr[j] = pv[j] >= 0 ? inside : zero_page + kofs;
            }
            if (++tap == sg.ntaps) {
                tap = 0;
                if (++cchunk == sg.nchunks) {
                    cchunk = 0;
                    if (seg + 1 < nseg) sg = load_seg(++seg);
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < 5; ++j) nptr[j] = aptr[j] + t * IGEMM_BK;
        }
    };

    bool abl_skip = false;                          // ablation builds: set after the prologue
    // LDS-DMA issue helpers (1 KiB per wave instruction; destination = wave-uniform base + lane * 16)
    auto issue_a = [&](int ast, int j0, int j1) {   // A pieces j0..j1-1 of the prepared tile -> A stage `ast`
        if ((ABL == 2 || ABL == 3) && abl_skip) return;
        char* base = smem + ast * A_BYTES;
#pragma unroll
        for (int j = 0; j < 5; ++j)
            if (j >= j0 && j < j1)
                __builtin_amdgcn_global_load_lds(GLB_PTR(nptr[j]), LDS_PTR(base + (q + 4 * j) * 1024), 16, 0, 0);
    };
    auto issue_w_g0 = [&](int t, int wst, int half) {   // group 0: four pieces of W_lo (half 0) or W_hi (half 1)
        if ((ABL == 2 || ABL == 3) && abl_skip) return;
        char* base = smem + W_BASE + wst * W_BYTES + half * (HALF_ROWS * 128);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            __builtin_amdgcn_global_load_lds(GLB_PTR((half ? wp2[j] : wp[j]) + t * IGEMM_BK), LDS_PTR(base + (q + 4 * j) * 1024), 16, 0, 0);
    };
    auto issue_w_g1 = [&](int t, int wst) {             // group 1: piece 16 + q of W_lo and of W_hi
        if (((ABL == 2 || ABL == 3) && abl_skip) || !G1_W) return;
        char* base = smem + W_BASE + wst * W_BYTES;
        __builtin_amdgcn_global_load_lds(GLB_PTR(wp[0] + t * IGEMM_BK), LDS_PTR(base + (16 + q) * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds(GLB_PTR(wp2[0] + t * IGEMM_BK), LDS_PTR(base + HALF_ROWS * 128 + (16 + q) * 1024), 16, 0, 0);
    };

    f32x4 acc[NT][MT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fsw = lane & 7, fg = lane >> 4;
    const int a_frag = (wm * (MT * 16) + frow) * 128;
    const int w_frag = W_BASE + (wn * (NT * 16) + frow) * 128;
    half8_t af[MT], wf[NT];

    auto read_frags = [&](int ast, int wst, int ks) {
        if (ABL == 3) return;
        const char* abase = smem + ast * A_BYTES + a_frag;
        const char* wbase = smem + wst * W_BYTES + w_frag;
        const int slot = ((ks * 4 + fg) ^ fsw) * 16;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) af[mt] = *reinterpret_cast<const half8_t*>(abase + mt * 16 * 128 + slot);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) wf[nt] = *reinterpret_cast<const half8_t*>(wbase + nt * 16 * 128 + slot);
    };
    auto mfma_block = [&]() {
        if constexpr (ABL == 1) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) asm volatile("" ::"v"(wf[nt]));
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) asm volatile("" ::"v"(af[mt]));
            return;
        }
        if constexpr (ABL == 4) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[nt], af[mt], acc[nt][mt], 0, 0, 0);
        if constexpr (ABL == 4) __builtin_amdgcn_s_setprio(0);
    };
    // phase boundary: nothing is scheduled across it
    auto bar = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    // counted wait: all but this wave's `N` youngest LDS-DMA pieces have landed (0 when nothing newer was issued)
    auto wait_dma = [&](bool issued, auto n_tag) {
        constexpr int N = decltype(n_tag)::value;
        if (issued) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    using I4 = std::integral_constant<int, 4>;
    using I5 = std::integral_constant<int, 5>;

    // ---- prologue: W(t_begin), A(t_begin), A(t_begin + 1) by the same roles; everything landed before the first read
    if (grp == 0) {
        issue_w_g0(t_begin, 0, 0);
        issue_w_g0(t_begin, 0, 1);
    } else {
        issue_w_g1(t_begin, 0);
        prepare_read();
        prepare_finish(t_begin);
        issue_a(0, 0, 5);
        if (t_begin + 1 < t_end) {
            prepare_read();
            prepare_finish(t_begin + 1);
            issue_a(1, 0, 5);
        }
        if (t_begin + 2 < t_end) { prepare_read(); prepare_finish(t_begin + 2); }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    bar();
    if (grp == 1) bar();                               // the trailing group runs one barrier behind
    abl_skip = true;

    int ast = 0;                                       // A stage of tile t; tile t+2 goes to stage ast2
    for (int t = t_begin; t < t_end; ++t) {
        const int wst = (t - t_begin) & 1;
        const int ast2 = ast == 0 ? 2 : ast - 1;       // (ast + 2) % 3
        const bool more1 = t + 1 < t_end, more2 = t + 2 < t_end;
        // ---- R(t, 0)
        read_frags(ast, wst, 0);
        if (grp == 0) {
            if (more1) issue_w_g0(t + 1, wst ^ 1, 0);
            wait_dma(more1, I4{});                     // W_hi(t), issued in R(t-1,1), has landed
        } else {
            if (more1) issue_w_g1(t + 1, wst ^ 1);     // W pieces first: the wait in R(t,1) leaves only A pieces behind
            if (more2) issue_a(ast2, 0, 2);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        bar();
        // ---- M(t, 0)
        mfma_block();
        bar();
        // ---- R(t, 1)
        read_frags(ast, wst, 1);
        if (grp == 0) {
            if (more1) issue_w_g0(t + 1, wst ^ 1, 1);
        } else {
            if (more2) issue_a(ast2, 2, 5);
            if (t + 3 < t_end) prepare_read();         // table reads for tile t+3 share the fragment reads' latency
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (grp == 1) {
            if (t + 3 < t_end) prepare_finish(t + 3);
            wait_dma(more2, I5{});                     // W(t+1) pieces and A(t+1) have landed
        }
        bar();
        // ---- M(t, 1)
        mfma_block();
        if (grp == 0) wait_dma(more1, I4{});           // W_lo(t+1), issued in R(t,0), has landed
        if (!(grp == 1 && !more1)) bar();
        ast = ast == 2 ? 0 : ast + 1;
    }

    igemm_epilogue<MT, NT, EPI>(p, acc, m0 + wm * (MT * 16) + (lane & 15), n0 + wn * (NT * 16) + (lane >> 4) * 4,
                                n0 + wn * (NT * 16), lane, split);
}

template <bool GATHER, int ABL = 0, int EPI = EPI_LINEAR, int NT = 5>
static int launch_pp_t(const IgemmParams& p, hipStream_t stream) {
    using namespace pp;
    constexpr int BN = Geo<NT>::BN;
    constexpr int lds = Geo<NT>::LDS_BYTES + (GATHER ? TAB_BYTES : 0);
    auto kern = igemm_pp_kernel<GATHER, EPI, ABL, NT>;
    static bool attr_set = false;
    if (!attr_set) {
        LAVIE_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_set = true;
    }
    const int grid = cdiv(p.M, BM) * (p.N / BN);
    hipLaunchKernelGGL(kern, dim3(grid, p.splits), dim3(THREADS), lds, stream, p);
    LAVIE_HIP(hipGetLastError());
    return 0;
}

static int g_pp_abl = 0;
void igemm_pp_ablate(int a) { g_pp_abl = a; }

// 160x256 variant with the GEGLU epilogue (plain A rows, no split-K).  N % 256 == 0.
int launch_igemm_pp_geglu(const IgemmParams& p, hipStream_t stream) {
    LAVIE_CHECK(p.N % 256 == 0 && p.splits == 1, "igemm_pp_geglu: N=%d must be a multiple of 256, no split-K", p.N);
    return launch_pp_t<false, 0, EPI_GEGLU, 4>(p, stream);
}

// Launches the 160x320 ping-pong kernel (EPI_LINEAR only; the caller runs the split-K reduce).  N % 320 == 0, or — the
// widths of the VSR UNet (256 / 512 / 1024) — N % 256 == 0 with the 160x256 tile (NT = 4).
int launch_igemm_pp(const IgemmParams& p, bool gather, hipStream_t stream) {
    if (p.N % 320 != 0) {
        LAVIE_CHECK(p.N % 256 == 0, "igemm_pp: N=%d is not a multiple of 320 or 256", p.N);
        return gather ? launch_pp_t<true, 0, EPI_LINEAR, 4>(p, stream) : launch_pp_t<false, 0, EPI_LINEAR, 4>(p, stream);
    }
    if (gather && g_pp_abl == 1) return launch_pp_t<true, 1>(p, stream);
    if (gather && g_pp_abl == 2) return launch_pp_t<true, 2>(p, stream);
    if (gather && g_pp_abl == 3) return launch_pp_t<true, 3>(p, stream);
    if (g_pp_abl == 4) return gather ? launch_pp_t<true, 4>(p, stream) : launch_pp_t<false, 4>(p, stream);   // correct results: with s_setprio
    return gather ? launch_pp_t<true>(p, stream) : launch_pp_t<false>(p, stream);
}

}  // namespace lavie
