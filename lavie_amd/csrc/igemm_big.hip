// 256x160 implicit-GEMM tile with phase-alternating wave groups (gfx950).
//
// Why a second kernel: the 128x160 / 4-wave kernel of igemm.hip streams 14 B of LDS-DMA per kFLOP and is paced by
// the L2 -> LDS path (~52 of ~70 GB/s per CU measured, profiles/r01_*): a 256-row tile halves the weight traffic
// (10 B/kFLOP).  A first 8-wave version ran every wave in lockstep (all read LDS, then all issue MFMAs) and lost
// more in the compute loop than it gained.  This version follows MI355X_MICROARCH "Two waves per SIMD": the
// workgroup's 8 waves form two groups (waves 0-3 / 4-7 = the two co-resident waves of each SIMD) that run HALF A
// STEP APART — while one group reads its MFMA fragments from LDS (and issues LDS-DMA for a later tile), its SIMD
// partner issues MFMAs on the fragments it read in the previous phase; a raw s_barrier separates the phases.
// The matrix pipe of every SIMD therefore always has one wave feeding it.
//
// K-tile 64, 3 LDS stages, two tiles in flight behind a counted s_waitcnt vmcnt (never 0 in the loop), one
// __shared__ array (guide §5 "Pipelining across barriers").  Fragment layout, swizzle and epilogue are those of
// igemm.hip.  GATHER: the source pixel of a row is rebuilt from a packed (image, y, x) word per row instead of a
// 9-tap table, so the whole 160 KiB of LDS stays available to the three stages.
#include <type_traits>

#include "igemm.h"
#include "igemm_epilogue.h"

namespace lavie {

namespace big {
constexpr int WM = 4, WN = 2, MT = 4, NT = 5;
constexpr int NW = WM * WN;                      // 8 waves
constexpr int THREADS = 64 * NW;
constexpr int BM = WM * MT * 16;                 // 256
constexpr int BN = WN * NT * 16;                 // 160
constexpr int NSTAGE = 3;
constexpr int STAGE_BYTES = (BM + BN) * 128;     // 53,248
constexpr int APIECES = BM / 8, WPIECES = BN / 8;
constexpr int AP = APIECES / NW;                 // 4
constexpr int WP = (WPIECES + NW - 1) / NW;      // 3 (24 slots for 20 pieces: 4 benign duplicates)
constexpr int LOADS = AP + WP;                   // LDS-DMA instructions per wave per tile
constexpr int ROWTAB_BYTES = BM * 4;             // GATHER: packed (image, y, x) per row
constexpr int SEGTAB_BYTES = IGEMM_MAX_SEG * 6 * 4;
constexpr int LDS_BYTES = NSTAGE * STAGE_BYTES;
static_assert(LDS_BYTES + ROWTAB_BYTES + SEGTAB_BYTES <= 160 * 1024, "does not fit LDS");
}  // namespace big

template <bool GATHER, int EPI, int ABL = 0>
__global__ __launch_bounds__(big::THREADS) void igemm_big_kernel(const IgemmParams p) {
    using namespace big;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;                    // 0: leading group, 1: trailing group (SIMD partners)
    const int wm = wave / WN, wn = wave % WN;

    const int n_tiles = p.N / BN;
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int m0 = (bid / n_tiles) * BM;
    const int n0 = (bid % n_tiles) * BN;
    const int split = blockIdx.y;
    const int t_begin = (int)((long)p.nk * split / p.splits);
    const int t_end = (int)((long)p.nk * (split + 1) / p.splits);
    const int nkl = t_end - t_begin;

    const int lr = lane >> 3;
    const int kofs = ((lane & 7) ^ lr) * 8;

    // ---- per-row source description
    unsigned* rowtab = reinterpret_cast<unsigned*>(smem + LDS_BYTES);
    int* segtab = reinterpret_cast<int*>(smem + LDS_BYTES + ROWTAB_BYTES);
    const half_t* aptr[AP];
    int arow[AP];
#pragma unroll
    for (int i = 0; i < AP; ++i) {
        arow[i] = (wave + NW * i) * 8 + lr;
        int m = m0 + arow[i];
        m = m < p.M ? m : p.M - 1;
        aptr[i] = GATHER ? p.zero : p.A + (size_t)m * p.lda + kofs;
    }
    auto sgpr = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
    if constexpr (GATHER) {
        const int hw = p.Ho * p.Wo;
        for (int row = tid; row < BM; row += THREADS) {
            int m = m0 + row;
            m = m < p.M ? m : p.M - 1;
            const int n = m / hw;
            const int rem = m - n * hw;
            const int y = rem / p.Wo, x = rem - y * p.Wo;
            rowtab[row] = ((unsigned)n << 20) | ((unsigned)y << 10) | (unsigned)x;      // n < 4096, y, x < 1024
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
    unsigned rowpk[AP];
#pragma unroll
    for (int i = 0; i < AP; ++i) rowpk[i] = GATHER ? rowtab[arow[i]] : 0u;
    const int Hv = p.Hi << p.ups, Wv = p.Wi << p.ups;

    const half_t* wptr[WP];
#pragma unroll
    for (int i = 0; i < WP; ++i) {
        const int n = n0 + ((wave + NW * i) % WPIECES) * 8 + lr;
        wptr[i] = p.W + (size_t)n * p.ldw + kofs;
    }

    // gather cursor at t_begin (K order: segment > 64-channel slab > tap)
    int seg = 0, cchunk = 0, tap = 0;
    IgemmSeg sg = GATHER ? load_seg(0) : IgemmSeg{nullptr, 0, 0, 0, 1};
    const int nseg = p.nseg;
    if constexpr (GATHER) {
        int skip = t_begin;
        while (skip >= sg.nchunks * sg.ntaps && seg + 1 < nseg) {
            skip -= sg.nchunks * sg.ntaps;
            sg = load_seg(++seg);
        }
        cchunk = skip / sg.ntaps;
        tap = skip - cchunk * sg.ntaps;
    }

    // issue the LDS-DMA of absolute K-tile t into stage `buf`; advances the gather cursor
    auto issue = [&](int t, int buf) {
        if (ABL == 2 && t > t_begin + 1) return;
        char* base = smem + buf * STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < AP; ++i) {
            const half_t* src;
            if constexpr (GATHER) {
                const int tp = sg.ntaps == 9 ? tap : 4;
                const int iy = (int)((rowpk[i] >> 10) & 1023u) * p.stride + tp / 3 - 1;
                const int ix = (int)(rowpk[i] & 1023u) * p.stride + tp % 3 - 1;
                const bool ok = (unsigned)iy < (unsigned)Hv && (unsigned)ix < (unsigned)Wv;
                const size_t pix = ((size_t)(rowpk[i] >> 20) * p.Hi + (iy >> p.ups)) * p.Wi + (ix >> p.ups);
                src = ok ? sg.src + pix * sg.C + (sg.c0 + cchunk * IGEMM_BK + kofs) : p.zero + kofs;
            } else {
                src = aptr[i] + t * IGEMM_BK;
            }
            __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(base + (wave + NW * i) * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < WP; ++i)
            __builtin_amdgcn_global_load_lds(GLB_PTR(wptr[i] + t * IGEMM_BK),
                                             LDS_PTR(base + BM * 128 + ((wave + NW * i) % WPIECES) * 1024), 16, 0, 0);
        if constexpr (GATHER) {
            if (++tap == sg.ntaps) {
                tap = 0;
                if (++cchunk == sg.nchunks) {
                    cchunk = 0;
                    if (seg + 1 < nseg) sg = load_seg(++seg);
                }
            }
        }
    };

    f32x4 acc[NT][MT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fsw = lane & 7, fg = lane >> 4;
    const int a_frag = (wm * MT * 16 + frow) * 128;
    const int w_frag = BM * 128 + (wn * NT * 16 + frow) * 128;
    half8_t af[MT], wf[NT];                         // fragments of the step this wave is working on

    auto read_frags = [&](int buf, int ks) {
        if (ABL == 3) return;
        const char* base = smem + buf * STAGE_BYTES;
        const int slot = ((ks * 4 + fg) ^ fsw) * 16;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) af[mt] = *reinterpret_cast<const half8_t*>(base + a_frag + mt * 16 * 128 + slot);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) wf[nt] = *reinterpret_cast<const half8_t*>(base + w_frag + nt * 16 * 128 + slot);
    };
    auto mfma_step = [&]() {
        if (ABL == 3) return;
        if constexpr (ABL == 1) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) asm volatile("" ::"v"(wf[nt]));
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) asm volatile("" ::"v"(af[mt]));
            return;
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[nt], af[mt], acc[nt][mt], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
    // a phase boundary: LDS reads of the closing phase have returned (so the stage may be overwritten once every
    // wave is past the barrier), nothing is scheduled across it
    auto phase_barrier = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- prologue: tiles 0 and 1 in flight, tile 0 landed
    issue(t_begin, 0);
    if (nkl > 1) {
        issue(t_begin + 1, 1);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS) : "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    // ---- main loop.  Every wave runs the SAME code — read fragments, barrier, MFMAs, barrier — but the trailing
    // group starts one barrier later (the stagger of MI355X_MICROARCH "Two waves per SIMD", item 9), so at any time
    // one wave of each SIMD is in its LDS/DMA phase and its partner in its MFMA phase.  Barrier counts match: the
    // trailing group takes one extra barrier before the loop, the leading group one after it.
    if (grp == 1) phase_barrier();
    const int nsteps = 2 * nkl;
    int buf = 0;
    for (int s = 0; s < nsteps; ++s) {
        const int tl = s >> 1, ks = s & 1;
        read_frags(buf, ks);
        if (ks == 0 && tl + 2 < nkl) issue(t_begin + tl + 2, buf == 0 ? 2 : buf - 1);   // stage of tile tl-1: free
        if (ks == 1 && tl + 1 < nkl) {
            // Tile tl+1 is first read by the LEADING group in the phase that follows the trailing group's read of
            // (tl, 1): because of the stagger, every wave must have its share of tile tl+1 landed before the barrier
            // that closes THIS read phase (one barrier earlier than in an unstaggered loop); tile tl+2 stays in flight.
            if (tl + 2 < nkl) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        phase_barrier();
        mfma_step();
        if (ks == 1) buf = buf == 2 ? 0 : buf + 1;
        phase_barrier();
    }
    if (grp == 0) phase_barrier();

    igemm_epilogue<MT, NT, EPI>(p, acc, m0 + wm * MT * 16 + (lane & 15), n0 + wn * NT * 16 + (lane >> 4) * 4,
                                n0 + wn * NT * 16, lane, split);
}

template <bool GATHER, int EPI, int ABL = 0>
static int launch_big_t(const IgemmParams& p, hipStream_t stream) {
    using namespace big;
    constexpr int lds = LDS_BYTES + (GATHER ? ROWTAB_BYTES + SEGTAB_BYTES : 0);
    auto kern = igemm_big_kernel<GATHER, EPI, ABL>;
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

// Launches the 256x160 kernel (no split-K reduce here: the caller runs it).  N % 160 == 0, EPI_LINEAR only.
static int g_big_abl = 0;
void igemm_big_ablate(int a) { g_big_abl = a; }

int launch_igemm_big(const IgemmParams& p, bool gather, hipStream_t stream) {
    LAVIE_CHECK(p.N % big::BN == 0, "igemm_big: N=%d is not a multiple of %d", p.N, big::BN);
    if (!gather && g_big_abl == 1) return launch_big_t<false, EPI_LINEAR, 1>(p, stream);
    if (!gather && g_big_abl == 2) return launch_big_t<false, EPI_LINEAR, 2>(p, stream);
    if (!gather && g_big_abl == 3) return launch_big_t<false, EPI_LINEAR, 3>(p, stream);
    if (gather) {
        LAVIE_CHECK(p.Ho < 1024 && p.Wo < 1024 && p.M / (p.Ho * p.Wo) < 4096, "igemm_big: image grid too large for the packed row table");
        return launch_big_t<true, EPI_LINEAR>(p, stream);
    }
    return launch_big_t<false, EPI_LINEAR>(p, stream);
}

}  // namespace lavie
