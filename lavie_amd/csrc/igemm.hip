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
#include "igemm.h"
#include "profile.h"

namespace lavie {

template <int WM, int WN, int MT, int NT>
struct IgemmTile {
    static constexpr int NW = WM * WN;
    static constexpr int THREADS = 64 * NW;
    static constexpr int BM = WM * MT * 16;
    static constexpr int BN = WN * NT * 16;
    static constexpr int STAGE_BYTES = (BM + BN) * 128;
    static constexpr int LDS_BYTES = 2 * STAGE_BYTES;
    static constexpr int AP = BM / 8 / NW;   // 1-KiB A pieces per wave per stage
    static constexpr int WP = BN / 8 / NW;   // 1-KiB W pieces per wave per stage
    static_assert((BM / 8) % NW == 0 && (BN / 8) % NW == 0, "pieces must divide over waves");
};

template <int WM, int WN, int MT, int NT, bool GATHER, int EPI>
__global__ __launch_bounds__(64 * WM * WN) void igemm_kernel(const IgemmParams p) {
    using T = IgemmTile<WM, WN, MT, NT>;
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
        int m = m0 + (wave + T::NW * i) * 8 + lr;
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
        const int n = n0 + (wave + T::NW * i) * 8 + lr;
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
                                             LDS_PTR(base + (wave + T::NW * i) * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < T::WP; ++i)
            __builtin_amdgcn_global_load_lds(GLB_PTR(wptr[i] + t * IGEMM_BK),
                                             LDS_PTR(base + T::BM * 128 + (wave + T::NW * i) * 1024), 16, 0, 0);
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
    stage(0, 0);
    for (int t = 0; t < nk; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's LDS-DMA pieces of tile t have landed
        __syncthreads();   // ... and everybody else's; buffer (t+1)&1 is no longer being read
        if (t + 1 < nk) stage(t + 1, (t + 1) & 1);
        const char* base = smem + (t & 1) * T::STAGE_BYTES;
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
    }

    // ---- epilogue: lane holds channels n..n+3 of token m for every (nt, mt) ----
    const int mrow = m0 + wm * MT * 16 + (lane & 15);
    const int ncol = n0 + wn * NT * 16 + (lane >> 4) * 4;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m = mrow + mt * 16;
        if (m >= p.M) continue;
        const float* b2 = p.bias2 ? p.bias2 + (size_t)(m / p.rows_per_batch) * p.ldb2 : nullptr;
        if constexpr (EPI == EPI_LINEAR) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int n = ncol + nt * 16;
                f32x4 v = acc[nt][mt];
                if (p.bias) { const f32x4 b = *reinterpret_cast<const f32x4*>(p.bias + n); v += b; }
                if (b2) { const f32x4 b = *reinterpret_cast<const f32x4*>(b2 + n); v += b; }
                if (p.R) {
                    const half4_t r = *reinterpret_cast<const half4_t*>(p.R + (size_t)m * p.ldr + n);
                    v[0] += (float)r[0]; v[1] += (float)r[1]; v[2] += (float)r[2]; v[3] += (float)r[3];
                }
                half4_t o = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
                *reinterpret_cast<half4_t*>(p.C + (size_t)m * p.ldc + n) = o;
            }
        } else {
            // GEGLU: W rows are stored as 16-row blocks alternating value / gate (see pack_geglu),
            // so tile nt (even) holds h and tile nt+1 the matching gate; output column = n / 2.
            static_assert(EPI != EPI_GEGLU || NT % 2 == 0, "GEGLU needs value/gate tile pairs");
#pragma unroll
            for (int nt = 0; nt < NT; nt += 2) {
                const int n = ncol + nt * 16;
                f32x4 h = acc[nt][mt], g = acc[nt + 1][mt];
                if (p.bias) {
                    h += *reinterpret_cast<const f32x4*>(p.bias + n);
                    g += *reinterpret_cast<const f32x4*>(p.bias + n + 16);
                }
                const int no = (n0 + wn * NT * 16 + nt * 16) / 2 + (lane >> 4) * 4;
                half4_t o = {(half_t)(h[0] * gelu_erf_f(g[0])), (half_t)(h[1] * gelu_erf_f(g[1])),
                             (half_t)(h[2] * gelu_erf_f(g[2])), (half_t)(h[3] * gelu_erf_f(g[3]))};
                *reinterpret_cast<half4_t*>(p.C + (size_t)m * p.ldc + no) = o;
            }
        }
    }
}

template <int WM, int WN, int MT, int NT, bool GATHER, int EPI>
static int launch_tile(const IgemmParams& p, hipStream_t stream) {
    using T = IgemmTile<WM, WN, MT, NT>;
    auto kern = igemm_kernel<WM, WN, MT, NT, GATHER, EPI>;
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
    if (epilogue == EPI_GEGLU) {
        LAVIE_CHECK(p.N % 128 == 0, "igemm: GEGLU needs N %% 128 == 0 (N=%d)", p.N);
        LAVIE_CHECK(!p.R && !p.bias2, "igemm: GEGLU epilogue takes no residual / per-batch bias");
        LAVIE_CHECK(!gather, "igemm: GEGLU epilogue is only built for plain A rows");
        return launch_tile<2, 2, 4, 4, false, EPI_GEGLU>(p, stream);
    }
    if (p.N % 160 == 0)
        return gather ? launch_tile<2, 2, 4, 5, true, EPI_LINEAR>(p, stream)
                      : launch_tile<2, 2, 4, 5, false, EPI_LINEAR>(p, stream);
    if (p.N % 128 == 0)
        return gather ? launch_tile<2, 2, 4, 4, true, EPI_LINEAR>(p, stream)
                      : launch_tile<2, 2, 4, 4, false, EPI_LINEAR>(p, stream);
    if (p.N % 64 == 0)
        return gather ? launch_tile<2, 2, 4, 2, true, EPI_LINEAR>(p, stream)
                      : launch_tile<2, 2, 4, 2, false, EPI_LINEAR>(p, stream);
    set_error("igemm: N=%d is not a multiple of 64", p.N);
    return -1;
}

}  // namespace lavie
