// Row-resident fused transformer sub-blocks, gfx950 (round 3).
//
// The one-GEMM-per-launch transformer block moves every intermediate through HBM: LN(x) -> ff1 -> [T, 4C] -> ff2 -> + x is
// five tensor crossings for the 210 MB GEGLU intermediate alone at level 0, and the short-K GEMMs around it are bound by
// their stores, not by MFMA.  The kernels here keep a token's whole residual row in REGISTERS from the
// load of x to the store of x': a wave owns 16 tokens, the residual stream lives in the accumulator layout of
// v_mfma_f32_16x16x32_f16 (D[channel][token]: a lane holds 4 consecutive channels of ONE token per 16-channel tile, so
// C / 16 tiles x 4 = 80 fp32 registers at C = 320), and every product of the chain takes its activation operand straight
// from registers ("an accumulator tile as the next MFMA's operand", guide §3): the B fragment of k-step s is tiles 2s and
// 2s+1 converted pairwise to fp16, which fixes the k order inside a step to
//     k-slot 8q + j  <->  channel 32 s + 16 (j >> 2) + 4 q + (j & 3)          (q = lane >> 4)
// and the weights are repacked ONCE at load time into exactly that order.  Nothing but weights goes through LDS: they are
// streamed HBM/L2 -> LDS by LDS-DMA as 1-KiB pieces that are already the bank-conflict-free A-fragment image (one piece =
// 16 rows x 32 k: slot (4 r + (q ^ g(r >> 2))) * 16 B, g = {0, 3, 2, 1}, so a ds_read_b128 of lane (r, q) is one
// conflict-free read with an immediate offset per k-step), in the order the MFMAs consume them, through a ring the eight
// waves share.  Output accumulates INTO the residual registers (the MFMA's C operand), so "+ x" costs nothing and the
// residual sum stays in fp32 until the one rounding at the store.
//
// geglu_mlp_kernel: x' = x + W2 (h * gelu(g)) + b2, (h, g) = W1 LN(x) + b1   — FeedForward(GEGLU) of BasicTransformerBlock
//   (/root/reference/base/models/attention.py:558; spec /root/reference/vsr/models/diffusers_attention.py:801-822) with its
//   LayerNorm (attention.py:480) and residual.  Per 32 hidden units: 4 tiles x 10 k-steps of W1 (value / gate rows of the two
//   16-unit halves), GEGLU in registers, the 32 products become ONE B fragment, 20 tiles of W2 accumulate into the residual.
#include "rowfuse.h"

namespace lavie {


size_t geglu_mlp_image_bytes(int C) { return (size_t)(C / 8) * 60 * 1024; }          // 4C / 32 chunks x 60 pieces
size_t geglu_mlp_bias_floats(int C) { return (size_t)(C / 8) * 64; }
bool geglu_mlp_supported(int C) { return C == 320; }

// w1 [8C][C] (value rows first, then gate rows: `hidden_states, gate = proj(x).chunk(2)`), b1 [8C], w2 [C][4C], all fp16
// device tensors -> img (geglu_mlp_image_bytes) and b1img (geglu_mlp_bias_floats floats).  Synchronous (load time).
int pack_geglu_mlp(const half_t* w1, const half_t* b1, const half_t* w2, int C, half_t* img, float* b1img, hipStream_t stream) {
    LAVIE_CHECK(geglu_mlp_supported(C), "geglu_mlp: width %d is not built (320 only)", C);
    const int chunks = C / 8, ks_n = C / 32, nt = C / 16;
    std::vector<int2> p1, p2;
    std::vector<int> bi;
    for (int c = 0; c < chunks; ++c) {
        for (int tt = 0; tt < 4; ++tt) {        // v0 g0 v1 g1
            const int n0 = ((tt & 1) ? 4 * C : 0) + 32 * c + 16 * (tt >> 1);
            for (int ks = 0; ks < ks_n; ++ks) rf_piece_pairs(p1, c * 60 + tt * ks_n + ks, n0, C, 32 * ks);
            for (int r = 0; r < 16; ++r) bi.push_back(n0 + r);
        }
        for (int t2 = 0; t2 < nt; ++t2) rf_piece_pairs(p2, c * 60 + 4 * ks_n + t2, 16 * t2, 4 * C, 32 * c);
    }
    int2* dp = nullptr;
    int* di = nullptr;
    const size_t n1 = p1.size(), n2 = p2.size();
    LAVIE_HIP(hipMalloc(&dp, (n1 + n2) * sizeof(int2)));
    LAVIE_HIP(hipMalloc(&di, bi.size() * sizeof(int)));
    LAVIE_HIP(hipMemcpy(dp, p1.data(), n1 * sizeof(int2), hipMemcpyHostToDevice));
    LAVIE_HIP(hipMemcpy(dp + n1, p2.data(), n2 * sizeof(int2), hipMemcpyHostToDevice));
    LAVIE_HIP(hipMemcpy(di, bi.data(), bi.size() * sizeof(int), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(rf_gather8_kernel, dim3(cdiv((int)n1, 256)), dim3(256), 0, stream, (const uint2*)w1, (uint2*)img, dp, (int)n1);
    hipLaunchKernelGGL(rf_gather8_kernel, dim3(cdiv((int)n2, 256)), dim3(256), 0, stream, (const uint2*)w2, (uint2*)img, dp + n1, (int)n2);
    hipLaunchKernelGGL(rf_gather_f16_f32_kernel, dim3(cdiv((int)bi.size(), 256)), dim3(256), 0, stream, b1, b1img, di, (int)bi.size());
    LAVIE_HIP(hipGetLastError());
    LAVIE_HIP(hipStreamSynchronize(stream));
    (void)hipFree(dp);
    (void)hipFree(di);
    return 0;
}

// ------------------------------------------------------------------------------------------------ device helpers
struct GegluMlpParams {
    const half_t* x;         // [M, C] rows (ld = C)
    half_t* y;               // [M, C]; may alias x (a workgroup reads its rows before it writes them)
    const half_t* img;       // geglu_mlp_image_bytes(C)
    const float* b1img;      // geglu_mlp_bias_floats(C)
    const float* gamma;      // LayerNorm weight / bias [C]
    const float* beta;
    const float* b2;         // [C]
    int M;
    int tiles;               // 16-row tiles = ceil(M / 16), dealt to the workgroups in contiguous, near-equal runs
    float eps;
    unsigned long long* stamps;   // stamp build only: [8 waves][8] cycle sums of workgroup 0
    float* stats_out;        // optional [M, 2]: (mean, rstd) of every OUTPUT row (of its rounded fp16 values): what a LayerNorm-folded
                             // GEMM that consumes y needs (IgemmParams::ln_stats) — the interpolation block's norm_temp follows the
                             // feed-forward (interpolation/models/attention.py:592-604); nullptr = not written
};

template <int C, int PF, int ABL = 0>
__global__ __launch_bounds__(rf::THREADS, 2) void geglu_mlp_kernel(const GegluMlpParams p) {
    using namespace rf;
    unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t0 = 0;      // DMA issue, first product, geglu, second product, prologue, store, vmcnt wait, barrier
    auto mark = [&](int which) {
        if constexpr (ABL == 5) { const unsigned long long t = rf_stamp(); st[which] += t - t0; t0 = t; }
    };
    if constexpr (ABL == 5) t0 = rf_stamp();
    constexpr int NT = C / 16;                    // residual tiles per token tile (20)
    constexpr int KS = C / 32;                    // k-steps of the first product (10)
    constexpr int CHUNKS = C / 8;                 // 32-unit hidden chunks (40)
    constexpr int CH_PIECES = 4 * KS + NT;        // pieces per chunk (60)
    static_assert(2 * CH_PIECES == RING_PIECES, "two chunks fill the ring exactly");
    constexpr int PASS_GROUPS = CHUNKS * CH_PIECES / GROUP;   // 60
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const ring = smem;
    float* const b1s = reinterpret_cast<float*>(smem + RING_BYTES);           // [CHUNKS][4][16]
    float* const vec = b1s + CHUNKS * 64;                                     // gamma | beta | second-layer bias, C floats each

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, col = lane & 15;
    const int frag = ((col * 4) + (q ^ swz(col))) * 16;        // this lane's 16 bytes inside a piece

    // first-layer bias image -> LDS once (plain stores: no LDS-DMA is in flight yet)
    for (int i = tid; i < CHUNKS * 64; i += THREADS) b1s[i] = p.b1img[i];
    for (int i = tid; i < C; i += THREADS) { vec[i] = p.gamma[i]; vec[C + i] = p.beta[i]; vec[2 * C + i] = p.b2[i]; }

    // this workgroup's run of 16-row tiles: near-equal shares, 8 per pass (one per wave); the last pass may leave waves
    // without a tile (they keep moving weights and meeting barriers).  81920 rows on 256 workgroups = 20 tiles each = passes
    // of 8 + 8 + 4: the third pass runs one wave per SIMD.
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int share = p.tiles / nwg, rem = p.tiles - share * nwg;
    const int tile0 = bid * share + (bid < rem ? bid : rem);
    const int ntile = share + (bid < rem ? 1 : 0);
    const int my_passes = (ntile + WAVES - 1) / WAVES;
    const int total_groups = my_passes * PASS_GROUPS;
    // group g of the stream = image pieces (g % PASS_GROUPS) * 40 ..., ring pieces (g % 3) * 40 ...; wave w moves pieces w + 8 i
    auto issue_group = [&](int g) {
        if (g >= total_groups || (ABL == 3 && g >= 2)) return;
        const char* src = reinterpret_cast<const char*>(p.img) + ((size_t)((g % PASS_GROUPS) * GROUP + wave) << 10) + lane * 16;
        char* dst = ring + (((g % RING_GROUPS) * GROUP + wave) << 10);
#pragma unroll
        for (int i = 0; i < GROUP / WAVES; ++i) rf_dma(src + i * (WAVES << 10), dst + i * (WAVES << 10));
    };
    // before reading group g: this wave's pieces of g (issued two steps ago) have landed when at most the 5 pieces of
    // group g + 1 are outstanding; the barrier makes that true for every wave's pieces and ends everyone's reads of g - 1
    auto sync_group = [&](int g) {
        __builtin_amdgcn_sched_barrier(0);
        if (g + 1 < total_groups && ABL != 3) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        mark(6);
        if (ABL != 4) __builtin_amdgcn_s_barrier();
        mark(7);
        __builtin_amdgcn_sched_barrier(0);
        issue_group(g + 2);
        __builtin_amdgcn_sched_barrier(0);
    };
    __syncthreads();                             // bias image visible; nothing in flight yet
    issue_group(0);
    issue_group(1);

    int g = 0;
    for (int pass = 0; pass < my_passes; ++pass) {
        const int tl = pass * WAVES + wave;                      // this wave's tile within the workgroup's run
        const bool active = tl < ntile;                          // wave-uniform
        // ---- residual rows -> registers, accumulator layout: R[t][r] = x[row][16 t + 4 q + r]
        const int row = (tile0 + (active ? tl : 0)) * TOK + col;
        const int rowc = row < p.M ? row : p.M - 1;
        f32x4 R[NT];
        {
            const half_t* xr = p.x + (size_t)rowc * C + 4 * q;
            half4_t raw[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) raw[t] = *reinterpret_cast<const half4_t*>(xr + 16 * t);
#pragma unroll
            for (int t = 0; t < NT; ++t) R[t] = (f32x4){(float)raw[t][0], (float)raw[t][1], (float)raw[t][2], (float)raw[t][3]};
        }
        // ---- LayerNorm of the row (two passes over the registers, fp32), then the fp16 B fragments of LN(x)
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) sum += (R[t][0] + R[t][1]) + (R[t][2] + R[t][3]);
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float mean = sum * (1.0f / C);
        float sq = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float d = R[t][r] - mean; sq += d * d; }
        sq += __shfl_xor(sq, 16, 64);
        sq += __shfl_xor(sq, 32, 64);
        const float rstd = rsqrtf(sq * (1.0f / C) + p.eps);
        half8_t xb[KS];
        // The LayerNorm vectors and the second-layer bias come from LDS (round 3, stamp build: read from global memory they cost
        // ~35 dependent L2 round trips behind vmcnt(0) each — 21 % of the kernel).  The base is an opaque 32-bit offset renewed
        // every pass (else hipcc hoists sixty per-lane addresses out of the loop), fenced every two k-steps (else it requests
        // all sixty vectors at once and spills), and every result is pinned where it is computed (else it is sunk to its first use).
        unsigned vec_off = (unsigned)(size_t)LDS_PTR(vec + 4 * q);
        asm volatile("" : "+v"(vec_off));
        auto lds_f4 = [](unsigned off) { return *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>((size_t)off); };
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            if (s % 2 == 0 && s > 0) __builtin_amdgcn_sched_barrier(0);
            const f32x4 g0 = lds_f4(vec_off + (32 * s) * 4), g1 = lds_f4(vec_off + (32 * s + 16) * 4);
            const f32x4 e0 = lds_f4(vec_off + (C + 32 * s) * 4), e1 = lds_f4(vec_off + (C + 32 * s + 16) * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                xb[s][r] = (half_t)((R[2 * s][r] - mean) * rstd * g0[r] + e0[r]);
                xb[s][4 + r] = (half_t)((R[2 * s + 1][r] - mean) * rstd * g1[r] + e1[r]);
            }
            asm volatile("" : "+v"(xb[s]));
        }
        // second-layer bias joins the residual now: x + b2 + W2 h accumulates in place
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (t % 5 == 0 && t > 0) __builtin_amdgcn_sched_barrier(0);
            R[t] += lds_f4(vec_off + (2 * C + 16 * t) * 4);
            asm volatile("" : "+v"(R[t]));
        }
        __builtin_amdgcn_sched_barrier(0);

        // ---- 40 hidden chunks, two per ring revolution (pieces 0..59 / 60..119 of the ring).  Two accumulator sets: the
        // first product of chunk cc + 1 does not wait for the GEGLU of chunk cc.
        f32x4 accA[4], accB[4];
        half8_t hb;
        const unsigned ring_lo = (unsigned)(size_t)LDS_PTR(ring + frag);
        const unsigned ring_hi = (unsigned)(size_t)LDS_PTR(ring + (CH_PIECES << 10) + frag);
        const unsigned bias_base = (unsigned)(size_t)LDS_PTR(reinterpret_cast<char*>(b1s) + q * 16);
        // tiles T0 .. T1-1 of {v0, g0, v1, g1} of `chunk`: bias (the accumulators' initial value), then 10 k-steps each
        auto first_product = [&](f32x4 (&acc)[4], int chunk, unsigned rbase, auto t0_, auto t1_) {
            constexpr int T0 = decltype(t0_)::value, T1 = decltype(t1_)::value;
            if (!active) return;
            const unsigned ba = bias_base + chunk * 256;
            if constexpr (T0 == 0) { rf_lds_read_f32x4<0>(acc[0], ba); rf_lds_read_f32x4<64>(acc[1], ba); }
            if constexpr (T1 == 4) { rf_lds_read_f32x4<128>(acc[2], ba); rf_lds_read_f32x4<192>(acc[3], ba); }
            rf_run<(T1 - T0) * KS, PF, ABL>(rbase + ((T0 * KS) << 10), [&](auto m_, const half8_t& a) {
                constexpr int M = decltype(m_)::value;
                constexpr int TT = T0 + M / KS, S = M % KS;
                acc[TT] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, xb[S], acc[TT], 0, 0, 0);
            });
        };
        auto geglu = [&](const f32x4 (&acc)[4]) {
            if (!active) return;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                hb[r] = (half_t)(acc[0][r] * (ABL == 2 ? acc[1][r] : gelu_erf_f(acc[1][r])));
                hb[4 + r] = (half_t)(acc[2][r] * (ABL == 2 ? acc[3][r] : gelu_erf_f(acc[3][r])));
            }
        };
        auto second_product = [&](unsigned rbase) {
            if (!active) return;
            rf_run<NT, PF, ABL>(rbase + ((4 * KS) << 10), [&](auto m_, const half8_t& a) {
                constexpr int T = decltype(m_)::value;
                R[T] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, hb, R[T], 0, 0, 0);
            });
        };
        using I0 = std::integral_constant<int, 0>;
        using I2 = std::integral_constant<int, 2>;
        using I4 = std::integral_constant<int, 4>;
        mark(4);
        for (int cc = 0; cc < CHUNKS; cc += 2) {
            sync_group(g++);                              // ring pieces 0..39: W1 of chunk cc
            mark(0);
            first_product(accA, cc, ring_lo, I0{}, I4{});
            mark(1);
            sync_group(g++);                              // 40..79: W2 of chunk cc, v0 g0 of chunk cc + 1
            mark(0);
            first_product(accB, cc + 1, ring_hi, I0{}, I2{});
            mark(1);
            geglu(accA);
            mark(2);
            second_product(ring_lo);
            mark(3);
            sync_group(g++);                              // 80..119: v1 g1 and W2 of chunk cc + 1
            mark(0);
            first_product(accB, cc + 1, ring_hi, I2{}, I4{});
            mark(1);
            geglu(accB);
            mark(2);
            second_product(ring_hi);
            mark(3);
        }
        // ---- x' = residual registers, one rounding
        if (active && row < p.M) {
            half_t* yr = p.y + (size_t)row * C + 4 * q;
            float osum = 0.f, osq = 0.f;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const half4_t o = {(half_t)R[t][0], (half_t)R[t][1], (half_t)R[t][2], (half_t)R[t][3]};
                *reinterpret_cast<half4_t*>(yr + 16 * t) = o;
                if (p.stats_out) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) { const float f = (float)o[r]; osum += f; osq += f * f; }
                }
            }
            if (p.stats_out) {        // the row's 320 channels sit in the four lanes that share `col` (wave-uniform branch)
                osum += __shfl_xor(osum, 16, 64); osq += __shfl_xor(osq, 16, 64);
                osum += __shfl_xor(osum, 32, 64); osq += __shfl_xor(osq, 32, 64);
                const float omean = osum * (1.0f / C);
                const float orstd = rsqrtf(fmaxf(osq * (1.0f / C) - omean * omean, 0.f) + p.eps);      // rowstat_finalize_kernel's formula
                if (q == 0) *reinterpret_cast<float2*>(p.stats_out + (size_t)row * 2) = make_float2(omean, orstd);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (ABL == 5) {
        mark(5);
        if (blockIdx.x == 0 && lane == 0 && p.stamps) {
            for (int i = 0; i < 8; ++i) p.stamps[wave * 8 + i] = st[i];
        }
    }
}

static unsigned long long* g_rf_stamps = nullptr;
void rowfuse_set_stamp_buffer(unsigned long long* buf) { g_rf_stamps = buf; }
static int g_rf_variant = 0;            // tuning: LDS read-ahead depth of the fused kernels (0 = default)
void rowfuse_set_variant(int v) { g_rf_variant = v; }
int rowfuse_variant() { return g_rf_variant; }

template <int PF, int ABL = 0>
static int launch_geglu_mlp_t(const GegluMlpParams& p, hipStream_t stream, const ProfileScope& prof) {
    constexpr int lds = rf::RING_BYTES + (320 / 8) * 64 * 4 + 3 * 320 * 4;
    auto kern = geglu_mlp_kernel<320, PF, ABL>;
    static bool attr_set = false;
    if (!attr_set) {
        LAVIE_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_set = true;
    }
    const int grid = p.tiles < 256 ? p.tiles : 256;
    if (prof.active()) hipExtLaunchKernelGGL(kern, dim3(grid), dim3(rf::THREADS), lds, stream, prof.start(), prof.stop(), 0, p);
    else hipLaunchKernelGGL(kern, dim3(grid), dim3(rf::THREADS), lds, stream, p);
    LAVIE_HIP(hipGetLastError());
    return 0;
}

int launch_geglu_mlp(const half_t* x, half_t* y, int M, int C, const half_t* img, const float* b1img, const float* gamma,
                     const float* beta, const float* b2, float eps, hipStream_t stream, float* stats_out) {
    LAVIE_CHECK(geglu_mlp_supported(C), "geglu_mlp: width %d is not built (320 only)", C);
    LAVIE_CHECK(x && y && img && b1img && gamma && beta && b2 && M > 0, "geglu_mlp: bad arguments");
    // algorithmic work: both products; bytes: x in, x' out, the weights once
    ProfileScope prof(KC_FUSED_FF, stream, 2.0 * M * (double)C * 12.0 * C, 2.0 * (2.0 * M * C + 12.0 * C * C), /*kernel_events=*/true);
    GegluMlpParams p;
    p.x = x; p.y = y; p.img = img; p.b1img = b1img; p.gamma = gamma; p.beta = beta; p.b2 = b2;
    p.M = M; p.tiles = cdiv(M, rf::TOK); p.eps = eps; p.stamps = g_rf_stamps; p.stats_out = stats_out;
    switch (g_rf_variant) {
        case 1: return launch_geglu_mlp_t<5>(p, stream, prof);
        case 2: return launch_geglu_mlp_t<12>(p, stream, prof);
        case 3: return launch_geglu_mlp_t<8, 1>(p, stream, prof);
        case 4: return launch_geglu_mlp_t<8, 2>(p, stream, prof);
        case 5: return launch_geglu_mlp_t<8, 3>(p, stream, prof);
        case 6: return launch_geglu_mlp_t<8, 4>(p, stream, prof);
        case 7: return launch_geglu_mlp_t<8, 5>(p, stream, prof);      // stamp build
        default: return launch_geglu_mlp_t<8>(p, stream, prof);
    }
}


// ================================================================================================ temporal sub-block
// temporal_block_kernel: x' = x + Wo attn_temp(Wq LN(x), Wk LN(x), Wv LN(x)) + bo for the 16 frames of one pixel
//   (TemporalAttention, /root/reference/base/models/attention.py:580-667, with norm_temp and the residual of
//   BasicTransformerBlock :548-555) in ONE kernel: the [T, 3C] q|k|v tensor and the attention output never exist in memory,
//   and the two (b f) d c <-> (b d) f c transposes of the reference are the row addressing of one wave's 16 loads.
// A wave owns ONE pixel = 16 token rows (b*F + f) * D + pixel.  Per head pair (80 channels = five 16-channel tiles: the
// third tile holds channels 32..39 of both heads) it runs
//   q, k tiles  D[channel][frame] = W (A operand) x LN(x) (B operand)       -> scale, rotary (pairs sit in one lane), fp16
//   v tiles     D[frame][channel] = LN(x) (A operand) x W (B operand)       -> fp16 (already the A operand of V^T P^T)
//   S^T[key][query] = K^T Q + bias^T (bias = the accumulator's initial value), softmax over the key index = 4 registers + two
//                     cross-lane steps, P^T stays in registers as the B operand of O[channel][query] = V^T P^T (16x16x16)
//   x' tiles   += Wo (A) x O (B): O's accumulator tiles ARE the B fragments (k order fixed at pack time)
// Weights: 800 pieces per pass (200 per head pair: 50 q, 50 k, 50 v, 50 Wo), streamed as 40-piece units through the ring.
// Built for C = 320, 8 heads of 40 channels, F = 16, rotary over the first 32 channels of a head.
namespace tb {
constexpr int C = 320, NT = 20, KS = 10, F = 16, HEADS = 8, DH = 40;
constexpr int UNIT = 40, PASS_UNITS = 20, PAIR_PIECES = 200, PASS_PIECES = 800;
constexpr int TAB_BYTES = HEADS * F * F * 4 + 3 * C * 4;      // bias table + gamma | beta | to_out bias
}  // namespace tb

size_t temporal_block_image_bytes(int C) { return (size_t)tb::PASS_PIECES * 1024; }
bool temporal_block_supported(int C, int heads, int F, int rot_dim) {
    return C == tb::C && heads == tb::HEADS && F == tb::F && rot_dim == 32;
}

// wq / wk / wv / wo: [C][C] fp16 device tensors (attn_temp.to_q / to_k / to_v / to_out.0 weights) -> img.  Synchronous.
int pack_temporal_block(const half_t* wq, const half_t* wk, const half_t* wv, const half_t* wo, int C, half_t* img,
                        hipStream_t stream) {
    using namespace tb;
    LAVIE_CHECK(C == tb::C, "temporal_block: width %d is not built (320 only)", C);
    std::vector<int2> lists[4];
    for (int hp = 0; hp < 4; ++hp) {
        const int h0 = 2 * hp, h1 = 2 * hp + 1, P0 = PAIR_PIECES * hp;
        for (int m = 0; m < 3; ++m)
            for (int j = 0; j < 5; ++j) {
                int rows[16];
                for (int r = 0; r < 16; ++r)
                    rows[r] = j == 0 ? h0 * DH + r : j == 1 ? h0 * DH + 16 + r : j == 3 ? h1 * DH + r : j == 4 ? h1 * DH + 16 + r
                              : (r < 8 ? h0 * DH + 32 + r : h1 * DH + 32 + r - 8);
                for (int ks = 0; ks < KS; ++ks) rf_piece_pairs_rows(lists[m], P0 + 50 * m + 10 * j + ks, rows, C, 32 * ks);
            }
        for (int u = 0; u < 10; ++u) {
            for (int i = 0; i < 4; ++i)
                rf_piece_pairs(lists[3], P0 + 150 + 5 * u + i, 16 * (2 * u + (i >> 1)), C, ((i & 1) ? h1 : h0) * DH);
            const int piece = P0 + 150 + 5 * u + 4;          // the 16-deep step of two output tiles: channels 32..39 of both heads
            for (int slot = 0; slot < 64; ++slot) {
                const int r = slot >> 2, q = (slot & 3) ^ rf::swz(r);
                const int colk = q < 2 ? h0 * DH + 32 + 4 * q : h1 * DH + 32 + 4 * (q - 2);
                for (int half = 0; half < 2; ++half)
                    lists[3].push_back(make_int2(piece * 128 + slot * 2 + half, ((16 * (2 * u + half) + r) * C + colk) / 4));
            }
        }
    }
    const half_t* srcs[4] = {wq, wk, wv, wo};
    return rf_run_gathers(lists, srcs, 4, img, stream);
}

struct TemporalBlockParams {
    const half_t* x;          // [(b f) d, C] token rows
    half_t* y;                // same layout; may alias x
    const half_t* img;
    const float* gamma;       // norm_temp
    const float* beta;
    const float* bo;          // to_out.0.bias [C]
    const float* relbias;     // [heads][F][F] (query i, key j), fp32
    const float* rot_cos;     // [F][16]
    const float* rot_sin;
    int D;                    // pixels per frame
    int units;                // B * D pixels
    float scale, eps;
    float* dbg;               // development aid: workgroup 0 / wave 0 / head pair 0 dumps its register tiles (nullptr = off)
};

constexpr int g_dbg_pair_c = 1;       // development aid: which head pair the register-tile dump shows
template <int PF, int DBG = 0>      // DBG 1: every sync drains the DMA queue (vmcnt(0)) — protocol check
__global__ __launch_bounds__(rf::THREADS, 2) void temporal_block_kernel(const TemporalBlockParams p) {
    using namespace rf;
    using namespace tb;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const ring = smem;
    float* const tab = reinterpret_cast<float*>(smem + RING_BYTES);          // bias [head][query][key]
    float* const vec = tab + HEADS * F * F;                                   // gamma | beta | to_out bias, C floats each

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, col = lane & 15;
    const int frag = ((col * 4) + (q ^ swz(col))) * 16;
    for (int i = tid; i < HEADS * F * F; i += THREADS) tab[i] = p.relbias[i];
    for (int i = tid; i < C; i += THREADS) { vec[i] = p.gamma[i]; vec[C + i] = p.beta[i]; vec[2 * C + i] = p.bo[i]; }

    // rotary angles of this lane: frame = col, channel pairs (16 t + 4 q + {0,1}) / 2 and (16 t + 4 q + {2,3}) / 2 of tiles t = 0, 1
    float rc[2][2], rs[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            rc[t][e] = p.rot_cos[col * 16 + 8 * t + 2 * q + e];
            rs[t][e] = p.rot_sin[col * 16 + 8 * t + 2 * q + e];
        }

    const int nwg = gridDim.x, bid = blockIdx.x;
    const int share = p.units / nwg, rem = p.units - share * nwg;
    const int unit0 = bid * share + (bid < rem ? bid : rem);
    const int nunit = share + (bid < rem ? 1 : 0);
    const int my_passes = (nunit + WAVES - 1) / WAVES;

    auto issue_unit = [&](int u) {          // unit u of a pass: image pieces 40 u .., ring slot u mod 3; wave w moves pieces w + 8 i
        // the image base passes through an empty asm: otherwise hipcc hoists the 100 per-lane 64-bit source addresses of a pass
        // (20 units x 5 pieces) out of the pass loop and spills them (first build: 326 spilled registers)
        const char* imgp = reinterpret_cast<const char*>(p.img);
        asm volatile("" : "+s"(imgp));
        const char* src = imgp + ((size_t)(u * UNIT + wave) << 10) + lane * 16;
        char* dst = ring + (((u % RING_GROUPS) * UNIT + wave) << 10);
#pragma unroll
        for (int i = 0; i < UNIT / WAVES; ++i) rf_dma(src + i * (WAVES << 10), dst + i * (WAVES << 10));
    };
    // sync k, in front of segment k (pieces 20 k .. 20 k + 19): every wave's reads of segments < k are over (barrier), and
    // the unit that holds segment k + 1 has landed for everyone (each wave waits for its own pieces, then the barrier) — so
    // the readers may run up to one segment ahead of the barriers.  Units are issued two ahead: unit k/2 + 2 at even k, into
    // the ring slot of unit k/2 - 1 (segments k - 2, k - 1: read by nobody any more).
    auto sync = [&](auto k_) {
        constexpr int K = decltype(k_)::value;
        __builtin_amdgcn_sched_barrier(0);
        // need unit (K + 1) / 2 landed; the only younger unit this wave can have in flight is the next one
        constexpr bool younger = (K % 2 == 0) ? (K / 2 + 1 < PASS_UNITS) : ((K + 1) / 2 + 1 < PASS_UNITS);
        if constexpr (younger && !(DBG & 1)) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (K % 2 == 0 && K / 2 + 2 < PASS_UNITS) issue_unit(K / 2 + 2);
        __builtin_amdgcn_sched_barrier(0);
    };
    __syncthreads();                 // bias table visible; nothing in flight yet
    if (my_passes > 0) { issue_unit(0); issue_unit(1); }

    const unsigned ring_lo = (unsigned)(size_t)LDS_PTR(ring + frag);
    const unsigned ring_hi = (unsigned)(size_t)LDS_PTR(ring + 60 * 1024 + frag);
    constexpr float LOG2E = 1.4426950408889634f;

    for (int pass = 0; pass < my_passes; ++pass) {
        const int ul = pass * WAVES + wave;
        const bool active = ul < nunit;                          // wave-uniform
        const int unit = unit0 + (active ? ul : 0);
        const int b = unit / p.D, pix = unit - b * p.D;
        const size_t row = ((size_t)(b * F + col)) * p.D + pix;   // this lane's token row: frame = col
        // LDS bases of the small tables as opaque 32-bit offsets, renewed every pass: left as pointer arithmetic hipcc hoists some
        // eighty per-lane table addresses out of the pass loop, spills them, and reloads them behind vmcnt(0) inside the loop
        unsigned vec_off = (unsigned)(size_t)LDS_PTR(vec + 4 * q);
        unsigned tab_off = (unsigned)(size_t)LDS_PTR(tab + col * F + 4 * q);
        asm volatile("" : "+v"(vec_off), "+v"(tab_off));
        auto lds_f4 = [](unsigned off) { return *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>((size_t)off); };
        f32x4 R[NT];
        {
            const half_t* xr = p.x + row * C + 4 * q;
            half4_t raw[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) raw[t] = *reinterpret_cast<const half4_t*>(xr + 16 * t);
#pragma unroll
            for (int t = 0; t < NT; ++t) R[t] = (f32x4){(float)raw[t][0], (float)raw[t][1], (float)raw[t][2], (float)raw[t][3]};
        }
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) sum += (R[t][0] + R[t][1]) + (R[t][2] + R[t][3]);
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float mean = sum * (1.0f / C);
        float sq = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float d = R[t][r] - mean; sq += d * d; }
        sq += __shfl_xor(sq, 16, 64);
        sq += __shfl_xor(sq, 32, 64);
        const float rstd = rsqrtf(sq * (1.0f / C) + p.eps);
        half8_t xb[KS];
        // scheduling fences every two k-steps: unfenced, hipcc requests all sixty table vectors (240 registers) up front and spills
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            if (s % 2 == 0 && s > 0) __builtin_amdgcn_sched_barrier(0);
            const f32x4 g0 = lds_f4(vec_off + (32 * s) * 4), g1 = lds_f4(vec_off + (32 * s + 16) * 4);
            const f32x4 e0 = lds_f4(vec_off + (C + 32 * s) * 4), e1 = lds_f4(vec_off + (C + 32 * s + 16) * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                xb[s][r] = (half_t)((R[2 * s][r] - mean) * rstd * g0[r] + e0[r]);
                xb[s][4 + r] = (half_t)((R[2 * s + 1][r] - mean) * rstd * g1[r] + e1[r]);
            }
            asm volatile("" : "+v"(xb[s]));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (t % 5 == 0 && t > 0) __builtin_amdgcn_sched_barrier(0);
            R[t] += lds_f4(vec_off + (2 * C + 16 * t) * 4);
            asm volatile("" : "+v"(R[t]));        // the sum exists HERE: otherwise hipcc sinks it to the first use and keeps the bias vectors (spilled)
        }
        __builtin_amdgcn_sched_barrier(0);

        sync(std::integral_constant<int, 0>{});

        auto pair_body = [&](auto hp_) {
            constexpr int HP = decltype(hp_)::value;
            constexpr int P0 = PAIR_PIECES * HP;
            half4_t qp[5], kp[5], vp[5], op[5];
            if (active) {
                // One accumulator pair for the 15 tiles (5 q, 5 k, 5 v): tile T accumulates into acc[T & 1] while tile T - 1 is
                // scaled / rotated / packed four MFMAs after its last one (results complete, no wait states needed) — 8 live
                // accumulator registers instead of 40, which keeps the pair free of spills (a compiler spill of an asm-MFMA result
                // would read it before it is written: nothing pads hazards around asm).
                f32x4 acc[2];
                auto finish_tile = [&](auto t_) {
                    constexpr int T = decltype(t_)::value;
                    constexpr int W = T / 5, J = T % 5;
                    f32x4 v = acc[T & 1];
                    if constexpr (W == 0) v = v * p.scale;
                    if constexpr (W < 2 && J != 2) {             // rotary on channels 0..31 of a head: tiles 0, 1, 3, 4
                        constexpr int t = (J == 0 || J == 3) ? 0 : 1;
                        const f32x4 w = v;
                        v[0] = w[0] * rc[t][0] - w[1] * rs[t][0];
                        v[1] = w[1] * rc[t][0] + w[0] * rs[t][0];
                        v[2] = w[2] * rc[t][1] - w[3] * rs[t][1];
                        v[3] = w[3] * rc[t][1] + w[2] * rs[t][1];
                    }
                    if constexpr (W == 0) qp[J] = rf_pack(v);
                    else if constexpr (W == 1) kp[J] = rf_pack(v);
                    else vp[J] = rf_pack(v);
                    if ((DBG & 8) && HP == g_dbg_pair_c && p.dbg && bid == 0 && wave == 0 && pass == 0) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) p.dbg[(W * 20 + J * 4 + r) * 64 + lane] = W == 2 ? acc[T & 1][r] : (float)rf_pack(v)[r];
                    }
                };
                tb_run<P0, 150, PF, (DBG & 2) != 0>(ring_lo, ring_hi, ring + frag, [&](auto m_, const half8_t& a) {
                    constexpr int M = decltype(m_)::value;
                    constexpr int T = M / 10, S = M % 10;
                    if constexpr (T < 10) {                      // q, k: weights are the A operand -> D[channel][frame]
                        if constexpr (S == 0) rf_mfma32_first(acc[T & 1], a, xb[S]);
                        else rf_mfma32(acc[T & 1], a, xb[S]);
                    } else {                                     // v: LN(x) is the A operand -> D[frame][channel]
                        if constexpr (S == 0) rf_mfma32_first(acc[T & 1], xb[S], a);
                        else rf_mfma32(acc[T & 1], xb[S], a);
                    }
                    if constexpr (S == 4 && T > 0) {
                        // the previous tile passes THROUGH this point of the asm stream (five MFMAs after its last one): without it
                        // hipcc is free to schedule the VALU readers right behind that last MFMA — an asm it cannot see into —
                        // and they read the accumulator before the matrix pipe has written it
                        asm volatile("" : "+v"(acc[(T - 1) & 1]));
                        finish_tile(std::integral_constant<int, T - 1>{});
                    }
                }, sync);
                asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc[0]), "+v"(acc[1]));      // wait states, then the last tile's readers
                finish_tile(std::integral_constant<int, 14>{});
                __builtin_amdgcn_sched_barrier(0);

                // ---- attention of the two heads (everything in registers; the bias table is the only LDS access)
                f32x4 osh[2];
                auto attend = [&](auto e_) {
                    constexpr int E = decltype(e_)::value;
                    constexpr int TA = E ? 3 : 0, TB = TA + 1;
                    const int h = 2 * HP + E;
                    f32x4 S = lds_f4(tab_off + h * (F * F * 4));                 // bias[h][query = col][key = 4 q + r]
                    const bool mine = (q < 2) == (E == 0);             // rows of the shared tile that belong to this head
                    const half4_t z = {(half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};
                    const half4_t qm = mine ? qp[2] : z;
                    const half8_t kcat = rf_cat(kp[TA], kp[TB]), qcat = rf_cat(qp[TA], qp[TB]);
                    // both steps in place, in ONE statement: the operands' VALU writes are 4+ states back, the 16-deep step follows
                    // the 32-deep one after its 8 passes, and the VALU readers of S come after the trailing wait states
                    asm volatile("s_nop 4\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n\ts_nop 15\n\ts_nop 7\n\tv_mfma_f32_16x16x16_f16 %0, %3, %4, %0\n\ts_nop 15\n\ts_nop 3"
                                 : "+v"(S) : "v"(kcat), "v"(qcat), "v"(kp[2]), "v"(qm));
                    float m = fmaxf(fmaxf(S[0], S[1]), fmaxf(S[2], S[3]));
                    m = fmaxf(m, __shfl_xor(m, 16, 64));
                    m = fmaxf(m, __shfl_xor(m, 32, 64));
                    f32x4 e;
#pragma unroll
                    for (int r = 0; r < 4; ++r) e[r] = __builtin_amdgcn_exp2f((S[r] - m) * LOG2E);
                    float l = (e[0] + e[1]) + (e[2] + e[3]);
                    l += __shfl_xor(l, 16, 64);
                    l += __shfl_xor(l, 32, 64);
                    const float inv = __builtin_amdgcn_rcpf(l);
                    const half4_t pp = rf_pack(e);
                    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                    const f32x4 oa = __builtin_amdgcn_mfma_f32_16x16x16f16(vp[TA], pp, zero, 0, 0, 0);
                    const f32x4 ob = __builtin_amdgcn_mfma_f32_16x16x16f16(vp[TB], pp, zero, 0, 0, 0);
                    const f32x4 os = __builtin_amdgcn_mfma_f32_16x16x16f16(vp[2], pp, zero, 0, 0, 0);
                    op[TA] = rf_pack(oa * inv);
                    op[TB] = rf_pack(ob * inv);
                    osh[E] = os * inv;
                    if ((DBG & 8) && HP == g_dbg_pair_c && p.dbg && bid == 0 && wave == 0 && pass == 0) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            p.dbg[((3 + E) * 20 + 0 + r) * 64 + lane] = S[r];
                            p.dbg[((3 + E) * 20 + 4 + r) * 64 + lane] = e[r] * inv;
                            p.dbg[((3 + E) * 20 + 8 + r) * 64 + lane] = oa[r] * inv;
                            p.dbg[((3 + E) * 20 + 12 + r) * 64 + lane] = ob[r] * inv;
                            p.dbg[((3 + E) * 20 + 16 + r) * 64 + lane] = os[r] * inv;
                        }
                    }
                };
                attend(std::integral_constant<int, 0>{});
                attend(std::integral_constant<int, 1>{});
                op[2] = rf_pack(q < 2 ? osh[0] : osh[1]);
                __builtin_amdgcn_sched_barrier(0);

                // ---- x' += Wo O: 10 groups of five pieces = two output tiles x (head, head, 16-deep step of both)
                half8_t o0 = rf_cat(op[0], op[1]), o1 = rf_cat(op[3], op[4]);
                half8_t pend;                                    // the 16-deep fragment of the previous group of five pieces
                asm volatile("s_nop 4" : "+v"(o0), "+v"(o1), "+v"(op[2]));      // fresh VALU results: wait states before the MFMAs read them
                if ((DBG & 8) && HP == 0 && p.dbg && bid == 0 && wave == 0 && pass == 0) {       // operands and accumulators as the to_out product sees them
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        p.dbg[(420 + j) * 64 + lane] = (float)o0[j];
                        p.dbg[(428 + j) * 64 + lane] = (float)o1[j];
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) p.dbg[(436 + j) * 64 + lane] = (float)op[2][j];
#pragma unroll
                    for (int t = 0; t < NT; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) p.dbg[(440 + t * 4 + r) * 64 + lane] = R[t][r];
                }
                tb_run<P0 + 150, 50, PF, (DBG & 4) != 0>(ring_lo, ring_hi, ring + frag, [&](auto m_, const half8_t& a) {
                    constexpr int M = decltype(m_)::value;
                    constexpr int U = M / 5, I = M % 5;
                    if constexpr (I < 4) {
                        constexpr int T2 = 2 * U + (I >> 1);
                        rf_mfma32(R[T2], a, (I & 1) ? o1 : o0);
                        if constexpr (I == 3 && U > 0) {         // the previous group's 16-deep steps: its 32-deep ones are >= 4 MFMAs back
                            const half4_t alo = {pend[0], pend[1], pend[2], pend[3]}, ahi = {pend[4], pend[5], pend[6], pend[7]};
                            rf_mfma16(R[2 * U - 2], alo, op[2]);
                            rf_mfma16(R[2 * U - 1], ahi, op[2]);
                        }
                    } else {
                        pend = a;
                        if constexpr (U == 9) {
                            rf_mfma_drain();
                            const half4_t alo = {pend[0], pend[1], pend[2], pend[3]}, ahi = {pend[4], pend[5], pend[6], pend[7]};
                            rf_mfma16(R[18], alo, op[2]);
                            rf_mfma16(R[19], ahi, op[2]);
                        }
                    }
                }, sync);
                // wait states, and every residual tile passes through them: compiler code that reads R (the stores, a spill) stays behind
                asm volatile("s_nop 15\n\ts_nop 15"
                             : "+v"(R[0]), "+v"(R[1]), "+v"(R[2]), "+v"(R[3]), "+v"(R[4]), "+v"(R[5]), "+v"(R[6]), "+v"(R[7]), "+v"(R[8]), "+v"(R[9]),
                               "+v"(R[10]), "+v"(R[11]), "+v"(R[12]), "+v"(R[13]), "+v"(R[14]), "+v"(R[15]), "+v"(R[16]), "+v"(R[17]), "+v"(R[18]),
                               "+v"(R[19]));
                if ((DBG & 8) && p.dbg && bid == 0 && wave == 0 && pass == 0) {
#pragma unroll
                    for (int t = 0; t < NT; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) p.dbg[(100 + HP * 80 + t * 4 + r) * 64 + lane] = R[t][r];
                }
            } else {
                // a wave without a pixel in this pass still moves weights and meets every barrier of the pair
                auto idle = [&](auto... ks) { (sync(std::integral_constant<int, decltype(ks)::value>{}), ...); };
                if constexpr (HP == 0) idle(std::integral_constant<int, 1>{}, std::integral_constant<int, 2>{}, std::integral_constant<int, 3>{},
                                            std::integral_constant<int, 4>{}, std::integral_constant<int, 5>{}, std::integral_constant<int, 6>{},
                                            std::integral_constant<int, 7>{}, std::integral_constant<int, 8>{}, std::integral_constant<int, 9>{});
                else idle(std::integral_constant<int, 10 * HP>{}, std::integral_constant<int, 10 * HP + 1>{}, std::integral_constant<int, 10 * HP + 2>{},
                          std::integral_constant<int, 10 * HP + 3>{}, std::integral_constant<int, 10 * HP + 4>{}, std::integral_constant<int, 10 * HP + 5>{},
                          std::integral_constant<int, 10 * HP + 6>{}, std::integral_constant<int, 10 * HP + 7>{}, std::integral_constant<int, 10 * HP + 8>{},
                          std::integral_constant<int, 10 * HP + 9>{});
            }
        };
        pair_body(std::integral_constant<int, 0>{});
        pair_body(std::integral_constant<int, 1>{});
        pair_body(std::integral_constant<int, 2>{});
        pair_body(std::integral_constant<int, 3>{});

        // end of pass: every wave is past the last segment, so the ring's first two slots may take the next pass's first units
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (pass + 1 < my_passes) { issue_unit(0); issue_unit(1); }
        if (active) {
            half_t* yr = p.y + row * C + 4 * q;
#pragma unroll
            for (int t = 0; t < NT; ++t) *reinterpret_cast<half4_t*>(yr + 16 * t) = rf_pack(R[t]);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

static float* g_tb_dbg = nullptr;
void temporal_block_set_debug(float* buf) { g_tb_dbg = buf; }

int launch_temporal_block(const half_t* x, half_t* y, int B, int F, int D, int C, int heads, const half_t* img,
                          const float* gamma, const float* beta, const float* bo, const float* relbias, const float* rot_cos,
                          const float* rot_sin, int rot_dim, float scale, float eps, hipStream_t stream) {
    LAVIE_CHECK(temporal_block_supported(C, heads, F, rot_dim), "temporal_block: C=%d heads=%d F=%d rot_dim=%d is not built", C, heads, F, rot_dim);
    LAVIE_CHECK(x && y && img && gamma && beta && bo && relbias && rot_cos && rot_sin && B > 0 && D > 0, "temporal_block: bad arguments");
    const double tok = (double)B * F * D;
    // algorithmic work: four C x C projections + the attention core; bytes (fused definition): x in, x' out, weights once
    ProfileScope prof(KC_FUSED_TEMPORAL, stream, 2.0 * tok * C * 4.0 * C + 4.0 * tok * F * C, 2.0 * (2.0 * tok * C + 4.0 * C * C), /*kernel_events=*/true);
    TemporalBlockParams p;
    p.x = x; p.y = y; p.img = img; p.gamma = gamma; p.beta = beta; p.bo = bo; p.relbias = relbias; p.rot_cos = rot_cos;
    p.rot_sin = rot_sin; p.D = D; p.units = B * D; p.scale = scale; p.eps = eps; p.dbg = g_tb_dbg;
    constexpr int lds = rf::RING_BYTES + tb::TAB_BYTES;
    const int grid = p.units < 256 ? p.units : 256;
    auto go = [&](auto kern) -> int {
        if (int rc = ensure_dynamic_lds((const void*)kern, lds)) return rc;     // once per kernel address, not per launch
        if (prof.active()) hipExtLaunchKernelGGL(kern, dim3(grid), dim3(rf::THREADS), lds, stream, prof.start(), prof.stop(), 0, p);
        else hipLaunchKernelGGL(kern, dim3(grid), dim3(rf::THREADS), lds, stream, p);
        LAVIE_HIP(hipGetLastError());
        return 0;
    };
    switch (g_rf_variant) {
        case 1: return go(temporal_block_kernel<8, 1>);
        case 2: return go(temporal_block_kernel<8, 6>);      // plain reads everywhere
        case 3: return go(temporal_block_kernel<8, 4>);      // plain reads in the to_out product only
        case 4: return go(temporal_block_kernel<8, 2>);      // plain reads in the q / k / v products only
        case 5: return go(temporal_block_kernel<4, 0>);
        case 6: return go(temporal_block_kernel<8, 8>);      // with the register-tile dump (lavie_debug_temporal_block_dump)
        default: return go(temporal_block_kernel<8, 0>);
    }
}

}  // namespace lavie
