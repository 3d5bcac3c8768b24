// Row-resident fused head of the transformer block (round 4), gfx950:
//     tx  = proj_in(GroupNorm(x))                       1x1 conv on the per-frame normalised block input (attention.py:369-373)
//     qkv = [to_q | to_k | to_v](LayerNorm_1(tx))       the self-attention projections (attention.py:513-516; :154, 177-178)
// in ONE kernel at C = 320 (level 0 of the base and interpolation UNets): the normalised copy of x, the LayerNorm statistics and
// the second read of tx never exist in memory.  It replaces the GroupNorm apply pass, the proj_in GEMM (+ its row-statistics
// finalize) and the LayerNorm-folded qkv GEMM: four launches that move x / norm(x) / tx through HBM five times.
//
// Same scheme as geglu_mlp_kernel (rowfuse.hip): a wave owns 16 tokens, the rows live in the accumulator layout of
// v_mfma_f32_16x16x32_f16 (D[channel][token]), weights stream HBM / L2 -> LDS through the 120-KiB ring in consumption order.
//   * GroupNorm arrives as per-(frame, channel) scale / shift pairs (gn_affine_kernel, norm.hip: a = rstd gamma, b = beta - mean a,
//     from the producer's statistics): the B fragments of the first product are fp16(a x + b) — the same rounding point as the
//     apply pass's fp16 output.  A wave copies its frame's 2.5 KiB of pairs into a private LDS slot once per pass.
//   * first product: 20 output tiles x 10 k-steps = 5 ring groups; the accumulators start from the proj_in bias; the result is the
//     new row T (80 registers, x is dead by then), rounded ONCE to fp16 for the tx store, and LayerNorm_1 runs on those rounded
//     values in registers (what the unfused path's LayerNorm sees).
//   * second product: 60 output tiles (q | k | v) x 10 k-steps = 15 ring groups; each group's four tiles are converted and stored
//     as soon as they are complete.  Stores count in vmcnt together with the LDS-DMA pieces: a group's four stores are issued
//     BEHIND the DMA batch of its sync point's successor, so the counted wait of the next group (5 pieces + 4 stores) leaves them
//     in flight for a whole group instead of exposing their latency fifteen times per pass.
#include "rowfuse.h"

namespace lavie {

namespace pq {
constexpr int C = 320, NT = C / 16, KS = C / 32;
constexpr int TILES = 4 * NT;                               // proj_in (20) | q (20) | k (20) | v (20) output tiles
constexpr int PASS_GROUPS = TILES * KS / rf::GROUP;         // 20 ring groups of 4 tiles x 10 k-steps per pass
constexpr int GROUPS_A = NT * KS / rf::GROUP;               // 5: the proj_in product
constexpr int BIAS_BYTES = NT * 16 * 4;                     // proj_in bias in tile order
constexpr int VEC_BYTES = 2 * C * 4;                        // LayerNorm_1 gamma | beta
constexpr int AB_BYTES = C * 2 * 4;                         // one frame's (a, b) pairs, per wave
constexpr int LDS_BYTES = rf::RING_BYTES + BIAS_BYTES + VEC_BYTES + rf::WAVES * AB_BYTES;
static_assert(PASS_GROUPS * rf::GROUP == TILES * KS && LDS_BYTES <= 160 * 1024, "proj/qkv stream geometry");
}  // namespace pq

bool proj_qkv_supported(int C) { return C == pq::C; }
size_t proj_qkv_image_bytes(int C) { return C == pq::C ? (size_t)pq::TILES * pq::KS * 1024 : 0; }

// wpin [C][C] (proj_in.weight, 1x1 conv or Linear), wqkv [3C][C] (to_q rows, then to_k, then to_v), fp16 device tensors -> img.
// Synchronous (load time).
int pack_proj_qkv(const half_t* wpin, const half_t* wqkv, int C, half_t* img, hipStream_t stream) {
    LAVIE_CHECK(proj_qkv_supported(C), "proj_qkv: width %d is not built (320 only)", C);
    std::vector<int2> lists[2];
    for (int tile = 0; tile < pq::TILES; ++tile)
        for (int ks = 0; ks < pq::KS; ++ks) {
            const bool first = tile < pq::NT;
            rf_piece_pairs(lists[first ? 0 : 1], tile * pq::KS + ks, first ? 16 * tile : 16 * (tile - pq::NT), C, 32 * ks);
        }
    const half_t* srcs[2] = {wpin, wqkv};
    return rf_run_gathers(lists, srcs, 2, img, stream);
}

struct ProjQkvParams {
    const half_t* x;          // [M, C] block input rows (raw)
    const float* gn_ab;       // [M / rows_per_domain][C][2]: GroupNorm as y = a x + b per (frame, channel)
    const half_t* img;        // proj_qkv_image_bytes(C)
    const float* bpin;        // [C] proj_in bias
    const float* ln_g;        // [C] norm1 weight / bias
    const float* ln_b;
    half_t* tx;               // [M, C] out
    half_t* qkv;              // [M, 3C] out
    int M, tiles, rows_per_domain;
    float eps;
};

template <int PF>
__global__ __launch_bounds__(rf::THREADS, 2) void proj_qkv_kernel(const ProjQkvParams p) {
    using namespace rf;
    using namespace pq;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const ring = smem;
    float* const bias_s = reinterpret_cast<float*>(smem + RING_BYTES);                      // [NT][16]
    float* const vec = reinterpret_cast<float*>(smem + RING_BYTES + BIAS_BYTES);            // gamma | beta
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* const ab_s = reinterpret_cast<float*>(smem + RING_BYTES + BIAS_BYTES + VEC_BYTES + wave * AB_BYTES);   // this wave's [C][2]
    const int q = lane >> 4, col = lane & 15;
    const int frag = ((col * 4) + (q ^ swz(col))) * 16;        // this lane's 16 bytes inside a piece

    for (int i = tid; i < C; i += THREADS) { bias_s[i] = p.bpin[i]; vec[i] = p.ln_g[i]; vec[C + i] = p.ln_b[i]; }

    const int nwg = gridDim.x, bid = blockIdx.x;
    const int share = p.tiles / nwg, rem = p.tiles - share * nwg;
    const int tile0 = bid * share + (bid < rem ? bid : rem);
    const int ntile = share + (bid < rem ? 1 : 0);
    const int my_passes = (ntile + WAVES - 1) / WAVES;
    const int total_groups = my_passes * PASS_GROUPS;
    auto issue_group = [&](int g) {
        if (g >= total_groups) return;
        const char* src = reinterpret_cast<const char*>(p.img) + ((size_t)((g % PASS_GROUPS) * GROUP + wave) << 10) + lane * 16;
        char* dst = ring + (((g % RING_GROUPS) * GROUP + wave) << 10);
#pragma unroll
        for (int i = 0; i < GROUP / WAVES; ++i) rf_dma(src + i * (WAVES << 10), dst + i * (WAVES << 10));
    };
    // before reading group g: this wave's pieces of g have landed when at most the 5 pieces of group g + 1 (and `stores` younger
    // stores of this wave) are outstanding; the barrier makes that true for every wave and ends everyone's reads of g - 1
    auto sync_group = [&](int g, bool stores4) {
        __builtin_amdgcn_sched_barrier(0);
        if (g + 1 >= total_groups) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (stores4) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        issue_group(g + 2);
        __builtin_amdgcn_sched_barrier(0);
    };
    __syncthreads();                             // bias / LayerNorm vectors visible; nothing in flight yet
    issue_group(0);
    issue_group(1);

    const unsigned ring_base = (unsigned)(size_t)LDS_PTR(ring + frag);
    int g = 0;
    for (int pass = 0; pass < my_passes; ++pass) {
        const int tl = pass * WAVES + wave;
        const bool active = tl < ntile;                          // wave-uniform
        const int row = (tile0 + (active ? tl : 0)) * TOK + col;
        const int rowc = row < p.M ? row : p.M - 1;
        // ---- x rows -> registers; this tile's frame: its (a, b) pairs -> the wave's LDS slot (tracked loads + LDS writes: the one
        // point of a pass where the compiler may drain the DMA ring, two groups that have to land before the first sync anyway)
        f32x4 R[NT];
        {
            const half_t* xr = p.x + (size_t)rowc * C + 4 * q;
            half4_t raw[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) raw[t] = *reinterpret_cast<const half4_t*>(xr + 16 * t);
            const int dom = __builtin_amdgcn_readfirstlane(((tile0 + (active ? tl : 0)) * TOK < p.M ? (tile0 + (active ? tl : 0)) * TOK : p.M - 1) /
                                                           p.rows_per_domain);
            const f32x4* abg = reinterpret_cast<const f32x4*>(p.gn_ab + (size_t)dom * C * 2);
            f32x4 abv[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) abv[i] = abg[(lane + 64 * i) < C / 2 ? lane + 64 * i : 0];
#pragma unroll
            for (int i = 0; i < 3; ++i)
                if (lane + 64 * i < C / 2) reinterpret_cast<f32x4*>(ab_s)[lane + 64 * i] = abv[i];
#pragma unroll
            for (int t = 0; t < NT; ++t) R[t] = (f32x4){(float)raw[t][0], (float)raw[t][1], (float)raw[t][2], (float)raw[t][3]};
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the wave's own LDS writes (no other wave reads the slot)
        // ---- GroupNorm as a x + b -> the fp16 B fragments of the first product.  LDS vectors behind opaque offsets, fenced every two
        // k-steps (rowfuse.hip: else hipcc hoists / requests everything at once and spills)
        half8_t xb[KS];
        unsigned ab_off = (unsigned)(size_t)LDS_PTR(ab_s + 8 * q);
        asm volatile("" : "+v"(ab_off));
        auto lds_f4 = [](unsigned off) { return *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>((size_t)off); };
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            if (s % 2 == 0 && s > 0) __builtin_amdgcn_sched_barrier(0);
            // channels 32 s + 4 q .. + 3 and 32 s + 16 + 4 q .. + 3: pairs (a, b) interleaved, 8 floats per 4 channels
            const f32x4 p0 = lds_f4(ab_off + (32 * s) * 8), p1 = lds_f4(ab_off + (32 * s) * 8 + 16);
            const f32x4 p2 = lds_f4(ab_off + (32 * s + 16) * 8), p3 = lds_f4(ab_off + (32 * s + 16) * 8 + 16);
            xb[s][0] = (half_t)(R[2 * s][0] * p0[0] + p0[1]);
            xb[s][1] = (half_t)(R[2 * s][1] * p0[2] + p0[3]);
            xb[s][2] = (half_t)(R[2 * s][2] * p1[0] + p1[1]);
            xb[s][3] = (half_t)(R[2 * s][3] * p1[2] + p1[3]);
            xb[s][4] = (half_t)(R[2 * s + 1][0] * p2[0] + p2[1]);
            xb[s][5] = (half_t)(R[2 * s + 1][1] * p2[2] + p2[3]);
            xb[s][6] = (half_t)(R[2 * s + 1][2] * p3[0] + p3[1]);
            xb[s][7] = (half_t)(R[2 * s + 1][3] * p3[2] + p3[3]);
            asm volatile("" : "+v"(xb[s]));
        }
        __builtin_amdgcn_sched_barrier(0);

        // ---- first product: T = Wpin GN(x) + bpin, four output tiles per ring group
        f32x4 T[NT];
        const unsigned bias_base = (unsigned)(size_t)LDS_PTR(reinterpret_cast<char*>(bias_s) + q * 16);
        auto group_a = [&](auto g_) {
            constexpr int GA = decltype(g_)::value;
            sync_group(g, false);
            const unsigned rbase = ring_base + (unsigned)((g % RING_GROUPS) * (GROUP << 10));
            ++g;
            if (!active) return;
            rf_lds_read_f32x4<(4 * GA + 0) * 64>(T[4 * GA + 0], bias_base);
            rf_lds_read_f32x4<(4 * GA + 1) * 64>(T[4 * GA + 1], bias_base);
            rf_lds_read_f32x4<(4 * GA + 2) * 64>(T[4 * GA + 2], bias_base);
            rf_lds_read_f32x4<(4 * GA + 3) * 64>(T[4 * GA + 3], bias_base);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(T[4 * GA]), "+v"(T[4 * GA + 1]), "+v"(T[4 * GA + 2]), "+v"(T[4 * GA + 3])::"memory");
            rf_run<4 * KS, PF, 0>(rbase, [&](auto m_, const half8_t& a) {
                constexpr int M = decltype(m_)::value;
                constexpr int TT = 4 * GA + M / KS, S = M % KS;
                T[TT] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, xb[S], T[TT], 0, 0, 0);
            });
        };
        group_a(std::integral_constant<int, 0>{});
        group_a(std::integral_constant<int, 1>{});
        group_a(std::integral_constant<int, 2>{});
        group_a(std::integral_constant<int, 3>{});
        group_a(std::integral_constant<int, 4>{});

        // ---- tx = fp16(T) (the residual stream of the block); LayerNorm_1 on the rounded row, then its fp16 B fragments
        half8_t xb2[KS];
        if (active) {
            float sum = 0.f;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const half4_t o = rf_pack(T[t]);
                if (row < p.M) *reinterpret_cast<half4_t*>(p.tx + (size_t)row * C + 4 * q + 16 * t) = o;
                T[t] = (f32x4){(float)o[0], (float)o[1], (float)o[2], (float)o[3]};
                sum += (T[t][0] + T[t][1]) + (T[t][2] + T[t][3]);
            }
            sum += __shfl_xor(sum, 16, 64);
            sum += __shfl_xor(sum, 32, 64);
            const float mean = sum * (1.0f / C);
            float sq = 0.f;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float d = T[t][r] - mean; sq += d * d; }
            sq += __shfl_xor(sq, 16, 64);
            sq += __shfl_xor(sq, 32, 64);
            const float rstd = rsqrtf(sq * (1.0f / C) + p.eps);
            unsigned vec_off = (unsigned)(size_t)LDS_PTR(vec + 4 * q);
            asm volatile("" : "+v"(vec_off));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                if (s % 2 == 0 && s > 0) __builtin_amdgcn_sched_barrier(0);
                const f32x4 g0 = lds_f4(vec_off + (32 * s) * 4), g1 = lds_f4(vec_off + (32 * s + 16) * 4);
                const f32x4 e0 = lds_f4(vec_off + (C + 32 * s) * 4), e1 = lds_f4(vec_off + (C + 32 * s + 16) * 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    xb2[s][r] = (half_t)((T[2 * s][r] - mean) * rstd * g0[r] + e0[r]);
                    xb2[s][4 + r] = (half_t)((T[2 * s + 1][r] - mean) * rstd * g1[r] + e1[r]);
                }
                asm volatile("" : "+v"(xb2[s]));
            }
            __builtin_amdgcn_sched_barrier(0);
        }

        // ---- second product: q | k | v, four output tiles per ring group, stored as they complete (behind the next DMA batch)
        half_t* const orow = p.qkv + (size_t)row * (3 * C) + 4 * q;
        half4_t pend[4];
        bool have_pend = false;
        for (int gb = 0; gb < PASS_GROUPS - GROUPS_A; ++gb) {
            // (the first group of the product follows the 20 tx stores: a plain vmcnt(5) there; afterwards 5 pieces + 4 stores)
            sync_group(g, active && gb > 1);
            if (have_pend && row < p.M) {            // the previous group's four tiles: issued behind the DMA batch just above
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) *reinterpret_cast<half4_t*>(orow + ((gb - 1) * 4 + tt) * 16) = pend[tt];
            }
            const unsigned rbase = ring_base + (unsigned)((g % RING_GROUPS) * (GROUP << 10));
            ++g;
            if (!active) continue;
            f32x4 acc[4];
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) acc[tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            rf_run<4 * KS, PF, 0>(rbase, [&](auto m_, const half8_t& a) {
                constexpr int M = decltype(m_)::value;
                constexpr int TT = M / KS, S = M % KS;
                acc[TT] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, xb2[S], acc[TT], 0, 0, 0);
            });
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) pend[tt] = rf_pack(acc[tt]);
            have_pend = true;
        }
        if (have_pend && row < p.M) {
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) *reinterpret_cast<half4_t*>(orow + ((PASS_GROUPS - GROUPS_A - 1) * 4 + tt) * 16) = pend[tt];
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

int launch_proj_qkv(const half_t* x, const float* gn_ab, int rows_per_domain, const half_t* img, const float* bpin, const float* ln_g,
                    const float* ln_b, float eps, half_t* tx, half_t* qkv, int M, int C, hipStream_t stream) {
    LAVIE_CHECK(proj_qkv_supported(C), "proj_qkv: width %d is not built (320 only)", C);
    LAVIE_CHECK(x && gn_ab && img && bpin && ln_g && ln_b && tx && qkv && M > 0, "proj_qkv: bad arguments");
    LAVIE_CHECK(rows_per_domain > 0 && rows_per_domain % rf::TOK == 0 && M % rows_per_domain == 0,
                "proj_qkv: %d rows per GroupNorm domain (must be a multiple of 16 that divides M = %d)", rows_per_domain, M);
    // algorithmic work: the four C x C products; bytes: x in, tx and q|k|v out, the weights once
    ProfileScope prof(KC_LINEAR, stream, 2.0 * M * (double)C * 4.0 * C, 2.0 * (5.0 * M * C + 4.0 * C * C));
    ProjQkvParams p;
    p.x = x; p.gn_ab = gn_ab; p.img = img; p.bpin = bpin; p.ln_g = ln_g; p.ln_b = ln_b; p.tx = tx; p.qkv = qkv;
    p.M = M; p.tiles = cdiv(M, rf::TOK); p.rows_per_domain = rows_per_domain; p.eps = eps;
    auto kern = proj_qkv_kernel<8>;
    if (int rc = ensure_dynamic_lds((const void*)kern, pq::LDS_BYTES)) return rc;
    const int grid = p.tiles < 256 ? p.tiles : 256;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(rf::THREADS), pq::LDS_BYTES, stream, p);
    LAVIE_HIP(hipGetLastError());
    return 0;
}

}  // namespace lavie
