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
#include <type_traits>

#include "igemm.h"
#include "profile.h"

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
    static_assert(LDS_BYTES <= 160 * 1024, "tile does not fit LDS");
};

template <int WM, int WN, int MT, int NT, int NSTAGE, bool GATHER, int EPI>
__global__ __launch_bounds__(64 * WM * WN) void igemm_kernel(const IgemmParams p) {
    using T = IgemmTile<WM, WN, MT, NT, NSTAGE>;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    // ---- XCD-aware tile order (bijective for any grid size) ----
    const int n_tiles = p.N / T::BN;
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int m0 = (bid / n_tiles) * T::BM;
    const int n0 = (bid % n_tiles) * T::BN;

    // ---- staging addresses ----
    const int lr = lane >> 3;                       // row of this lane inside an 8-row piece
    const int kofs = ((lane & 7) ^ lr) * 8;         // source K offset (halfs) after the slot swizzle

    const half_t* aptr[T::AP];
    int astep[T::AP];
    int ay[T::AP], ax[T::AP], an[T::AP];
#pragma unroll
    for (int i = 0; i < T::AP; ++i) {
        int m = m0 + ((wave + T::NW * i) % T::APIECES) * 8 + lr;
        m = m < p.M ? m : p.M - 1;
        if constexpr (GATHER) {
            const int hw = p.Ho * p.Wo;
            an[i] = m / hw;
            const int rem = m - an[i] * hw;
            const int y = rem / p.Wo;
            ay[i] = y * p.stride;
            ax[i] = (rem - y * p.Wo) * p.stride;
            aptr[i] = p.zero;
            astep[i] = 0;
        } else {
            aptr[i] = p.A + (size_t)m * p.lda + kofs;
            astep[i] = IGEMM_BK;
            ay[i] = ax[i] = an[i] = 0;
        }
    }
    const half_t* wptr[T::WP];
#pragma unroll
    for (int i = 0; i < T::WP; ++i) {
        const int n = n0 + ((wave + T::NW * i) % T::WPIECES) * 8 + lr;
        wptr[i] = p.W + (size_t)n * p.ldw + kofs;
    }

    int seg = -1, cseg = 0, seg_chunks = 0;   // gather cursor: current segment / chunk inside it
    auto enter_segment = [&](int s) {
        seg = s;
        cseg = 0;
        if constexpr (GATHER) {
            const IgemmSeg sg = p.seg[s];
            seg_chunks = sg.nchunks;
            const int Hv = p.Hi << p.ups, Wv = p.Wi << p.ups;
#pragma unroll
            for (int i = 0; i < T::AP; ++i) {
                const int iy = ay[i] + sg.dy, ix = ax[i] + sg.dx;
                const bool ok = (unsigned)iy < (unsigned)Hv && (unsigned)ix < (unsigned)Wv;
                const size_t pix = ((size_t)an[i] * p.Hi + (iy >> p.ups)) * p.Wi + (ix >> p.ups);
                aptr[i] = ok ? sg.src + pix * sg.C + sg.c0 + kofs : p.zero + kofs;
                astep[i] = ok ? IGEMM_BK : 0;
            }
        }
    };
    if constexpr (GATHER) enter_segment(0);

    auto stage = [&](int t, int buf) {
        char* base = smem + buf * T::STAGE_BYTES;
        const int kc = GATHER ? cseg : t;
#pragma unroll
        for (int i = 0; i < T::AP; ++i)
            __builtin_amdgcn_global_load_lds(GLB_PTR(aptr[i] + kc * astep[i]),
                                             LDS_PTR(base + ((wave + T::NW * i) % T::APIECES) * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < T::WP; ++i)
            __builtin_amdgcn_global_load_lds(GLB_PTR(wptr[i] + t * IGEMM_BK),
                                             LDS_PTR(base + T::BM * 128 + ((wave + T::NW * i) % T::WPIECES) * 1024), 16, 0, 0);
        if constexpr (GATHER) {
            if (++cseg == seg_chunks && seg + 1 < p.nseg) enter_segment(seg + 1);
        }
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

    const int nk = p.nk;
    auto compute = [&](int buf) {
        const char* base = smem + buf * T::STAGE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int slot = ((ks * 4 + fg) ^ fsw) * 16;
            half8_t af[MT], wf[NT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                af[mt] = *reinterpret_cast<const half8_t*>(base + a_frag + mt * 16 * 128 + slot);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                wf[nt] = *reinterpret_cast<const half8_t*>(base + w_frag + nt * 16 * 128 + slot);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[nt], af[mt], acc[nt][mt], 0, 0, 0);
        }
    };
    if constexpr (NSTAGE == 2) {
        // one tile in flight: loads of tile t+1 run under the MFMAs of tile t
        stage(0, 0);
        for (int t = 0; t < nk; ++t) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's LDS-DMA pieces of tile t have landed
            __syncthreads();   // ... and everybody else's; buffer (t+1)&1 is no longer being read
            if (t + 1 < nk) stage(t + 1, (t + 1) & 1);
            compute(t & 1);
        }
    } else {
        // NSTAGE-1 tiles in flight across the barrier: counted vmcnt + raw s_barrier (a __syncthreads() would
        // drain the LDS-DMA queue, guide §5 "Pipelining across barriers").  Every wave issues exactly
        // T::LOADS LDS-DMA instructions per tile, so "all but the newest (NSTAGE-2)*LOADS" == tile t landed.
        static_assert(NSTAGE == 3, "counted-wait schedule is written for 3 stages");
        stage(0, 0);
        if (nk > 1) stage(1, 1);
        int buf = 0;
        for (int t = 0; t < nk; ++t) {
            if (t + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(T::LOADS) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                       // tile t visible to all; tile t-1's buffer is free
            asm volatile("" ::: "memory");
            if (t + 2 < nk) stage(t + 2, buf == 0 ? 2 : buf - 1);
            compute(buf);
            buf = buf == 2 ? 0 : buf + 1;
        }
    }

    // ---- epilogue: lane holds channels n..n+3 of token m for every (nt, mt) ----
    // Which optional operands exist is decided ONCE (wave-uniform) and the body is instantiated per
    // combination: per-element "if (ptr) load" makes hipcc wait vmcnt(0) after every load (guide §5, trap (c)).
    const int mrow = m0 + wm * MT * 16 + (lane & 15);
    const int ncol = n0 + wn * NT * 16 + (lane >> 4) * 4;
    auto epilogue = [&](auto has_bias, auto has_b2, auto has_res) {
        constexpr bool BIAS = decltype(has_bias)::value, B2 = decltype(has_b2)::value, RES = decltype(has_res)::value;
        f32x4 bv[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            bv[nt] = BIAS ? *reinterpret_cast<const f32x4*>(p.bias + ncol + nt * 16) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int m = mrow + mt * 16;
            const int mc = m < p.M ? m : p.M - 1;                  // clamp: loads stay in bounds, stores are predicated
            if constexpr (EPI == EPI_LINEAR) {
                f32x4 b2v[NT];
                half4_t rv[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    if constexpr (B2)
                        b2v[nt] = *reinterpret_cast<const f32x4*>(p.bias2 + (size_t)(mc / p.rows_per_batch) * p.ldb2 + ncol + nt * 16);
                    if constexpr (RES)
                        rv[nt] = *reinterpret_cast<const half4_t*>(p.R + (size_t)mc * p.ldr + ncol + nt * 16);
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    f32x4 v = acc[nt][mt] + bv[nt];
                    if constexpr (B2) v += b2v[nt];
                    if constexpr (RES) {
                        v[0] += (float)rv[nt][0]; v[1] += (float)rv[nt][1]; v[2] += (float)rv[nt][2]; v[3] += (float)rv[nt][3];
                    }
                    const half4_t o = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
                    if (m < p.M) *reinterpret_cast<half4_t*>(p.C + (size_t)m * p.ldc + ncol + nt * 16) = o;
                }
            } else {
                // GEGLU: W rows are stored as 16-row blocks alternating value / gate (see pack_geglu),
                // so tile nt (even) holds h and tile nt+1 the matching gate; output column = n / 2.
                static_assert(EPI != EPI_GEGLU || NT % 2 == 0, "GEGLU needs value/gate tile pairs");
#pragma unroll
                for (int nt = 0; nt < NT; nt += 2) {
                    const f32x4 h = acc[nt][mt] + bv[nt], g = acc[nt + 1][mt] + bv[nt + 1];
                    const int no = (n0 + wn * NT * 16 + nt * 16) / 2 + (lane >> 4) * 4;
                    const half4_t o = {(half_t)(h[0] * gelu_erf_f(g[0])), (half_t)(h[1] * gelu_erf_f(g[1])),
                                       (half_t)(h[2] * gelu_erf_f(g[2])), (half_t)(h[3] * gelu_erf_f(g[3]))};
                    if (m < p.M) *reinterpret_cast<half4_t*>(p.C + (size_t)m * p.ldc + no) = o;
                }
            }
        }
    };
    using T1 = std::true_type;
    using T0 = std::false_type;
    const int combo = (p.bias ? 1 : 0) | (p.bias2 ? 2 : 0) | (p.R ? 4 : 0);
    switch (combo) {
        case 0: epilogue(T0{}, T0{}, T0{}); break;
        case 1: epilogue(T1{}, T0{}, T0{}); break;
        case 2: epilogue(T0{}, T1{}, T0{}); break;
        case 3: epilogue(T1{}, T1{}, T0{}); break;
        case 4: epilogue(T0{}, T0{}, T1{}); break;
        case 5: epilogue(T1{}, T0{}, T1{}); break;
        case 6: epilogue(T0{}, T1{}, T1{}); break;
        default: epilogue(T1{}, T1{}, T1{}); break;
    }
}

static int g_force_tile = 0;   // 0 auto, 1 small tiles only, 2 big tiles whenever N allows (tests / A-B timing)

template <int WM, int WN, int MT, int NT, int NSTAGE, bool GATHER, int EPI>
static int launch_tile(const IgemmParams& p, hipStream_t stream) {
    using T = IgemmTile<WM, WN, MT, NT, NSTAGE>;
    auto kern = igemm_kernel<WM, WN, MT, NT, NSTAGE, GATHER, EPI>;
    static bool attr_set = false;   // one per instantiation
    if (!attr_set) {
        LAVIE_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, T::LDS_BYTES));
        attr_set = true;
    }
    const int grid = cdiv(p.M, T::BM) * (p.N / T::BN);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(T::THREADS), T::LDS_BYTES, stream, p);
    LAVIE_HIP(hipGetLastError());
    return 0;
}

int launch_igemm(const IgemmParams& p, bool gather, int epilogue, hipStream_t stream) {
    LAVIE_CHECK(p.M > 0 && p.N > 0 && p.nk > 0, "igemm: empty problem M=%d N=%d nk=%d", p.M, p.N, p.nk);
    const double K = (double)p.nk * IGEMM_BK;
    // algorithmic work: 2 M N K flops; bytes = every operand element once (A incl. im2col reuse counted once)
    ProfileScope prof(gather ? KC_CONV3X3 : KC_LINEAR, stream, 2.0 * p.M * p.N * K,
                      2.0 * ((double)p.M * K / (gather ? 9.0 : 1.0) + (double)p.N * K + (double)p.M * p.N));
    LAVIE_CHECK(p.N % 4 == 0 && p.ldc % 4 == 0, "igemm: N and ldc must be multiples of 4");
    // Tile choice.  "big": 256x160, 8 waves, 3 LDS stages (1 workgroup per CU, 2 waves per SIMD) — enough
    // reuse that the L2->LDS stream no longer paces the MFMAs; used when the grid still fills the chip.
    // "small": 128xBN, 4 waves, 2 stages (2 workgroups per CU) for short grids and odd N.
    const bool big = g_force_tile == 0 ? (p.M >= 256 && (long)cdiv(p.M, 256) * (p.N / 160) >= 256) : g_force_tile == 2;
    if (epilogue == EPI_GEGLU) {
        LAVIE_CHECK(p.N % 128 == 0, "igemm: GEGLU needs N %% 128 == 0 (N=%d)", p.N);
        LAVIE_CHECK(!p.R && !p.bias2, "igemm: GEGLU epilogue takes no residual / per-batch bias");
        LAVIE_CHECK(!gather, "igemm: GEGLU epilogue is only built for plain A rows");
        if (big) return launch_tile<4, 2, 4, 4, 3, false, EPI_GEGLU>(p, stream);
        return launch_tile<2, 2, 4, 4, 2, false, EPI_GEGLU>(p, stream);
    }
    if (p.N % 160 == 0) {
        if (big)
            return gather ? launch_tile<4, 2, 4, 5, 3, true, EPI_LINEAR>(p, stream)
                          : launch_tile<4, 2, 4, 5, 3, false, EPI_LINEAR>(p, stream);
        return gather ? launch_tile<2, 2, 4, 5, 2, true, EPI_LINEAR>(p, stream)
                      : launch_tile<2, 2, 4, 5, 2, false, EPI_LINEAR>(p, stream);
    }
    if (p.N % 128 == 0)
        return gather ? launch_tile<2, 2, 4, 4, 2, true, EPI_LINEAR>(p, stream)
                      : launch_tile<2, 2, 4, 4, 2, false, EPI_LINEAR>(p, stream);
    if (p.N % 64 == 0)
        return gather ? launch_tile<2, 2, 4, 2, 2, true, EPI_LINEAR>(p, stream)
                      : launch_tile<2, 2, 4, 2, 2, false, EPI_LINEAR>(p, stream);
    set_error("igemm: N=%d is not a multiple of 64", p.N);
    return -1;
}

void igemm_force_tile(int mode) { g_force_tile = mode; }

}  // namespace lavie
