// Row-resident fused text cross-attention sub-block, gfx950 (round 3; see rowfuse.hip for the scheme).
//
// cross_block_kernel:  x' = x + Wo1 att + bo1;  x'' = x' + Wo2 attn2(Wq2 LN2(x'), K, V) + bo2
//   = the tail of attn1 (its to_out projection and residual), norm2, and the whole of attn2 of BasicTransformerBlock
//   (/root/reference/base/models/attention.py:513-534; CrossAttention.forward / _attention :253-335) in ONE kernel.  K and V
//   of the text context are per-video constants (computed once per prompt by cache_context); here they become part of the
//   weight stream: bind_cross_block() writes, per video, an image of 720 one-KiB pieces in consumption order
//       200        Wo1 (20 output tiles x 10 k-steps; k in memory order: the operand is `att` loaded straight from HBM)
//       4 x 128    per head pair:  50 Wq2 (5 tiles x 10 k-steps, k in accumulator order: the operand is LN2 of the registers)
//                                  15 K  (5 key tiles x [head 0 channels 0..31 | head 1 channels 0..31 | channels 32..39 of both])
//                                  13 V^T (5 channel tiles x keys 0..31, 32..63; then the 16-key steps of two tiles per piece)
//                                  50 Wo2 (as the temporal kernel's to_out: 10 groups of 2 tiles x (head, head) + a 16-deep piece)
//       8          padding (the stream is cut into 40-piece units)
//   and the kernel streams the image of ITS video through the ring.  A wave owns 16 consecutive token rows (one video: passes
//   never straddle a video); scores live as S^T[key][token] accumulator tiles (5 per head: up to 80 keys), the softmax runs
//   over 20 registers + two cross-lane steps, P^T is the B operand of O[channel][token] = V^T P^T, and O's tiles are the B
//   operand of the to_out product, which accumulates into the residual registers.  q, the attention output, LN2(x') and x'
//   itself never exist in memory: the sub-block reads att and x once and writes x'' once (was: 12 tensor crossings in 4 launches).
// Built for C = 320, 8 heads of 40 channels, context length <= 80.
#include <map>

#include "rowfuse.h"

namespace lavie {

namespace xb {
constexpr int C = 320, NT = 20, KS = 10, HEADS = 8, DH = 40, MAXL = 80;
constexpr int UNIT = 40, PASS_UNITS = 18, PASS_PIECES = 720, O1_PIECES = 200, PAIR_PIECES = 128;
constexpr int Q_OFF = 0, K_OFF = 50, V_OFF = 65, O_OFF = 78;       // inside a pair's 128 pieces
constexpr int VEC_BYTES = 4 * C * 4;                                 // bo1 | gamma | beta | bo2
constexpr size_t IMG_BYTES = (size_t)PASS_PIECES * 1024;
// channel of row r of tile j of head pair (h0, h1): tiles 0, 1 = channels 0..31 of h0; 3, 4 = of h1; 2 = channels 32..39 of both
inline int pair_channel(int h0, int j, int r) {
    const int h1 = h0 + 1;
    return j == 0 ? h0 * DH + r : j == 1 ? h0 * DH + 16 + r : j == 3 ? h1 * DH + r : j == 4 ? h1 * DH + 16 + r
           : (r < 8 ? h0 * DH + 32 + r : h1 * DH + 32 + r - 8);
}
}  // namespace xb

size_t cross_block_image_bytes(int C) { return xb::IMG_BYTES; }
bool cross_block_supported(int C, int heads, int ctx_len, int rows_per_batch) {
    return C == xb::C && heads == xb::HEADS && ctx_len >= 1 && ctx_len <= xb::MAXL && rows_per_batch > 0 && rows_per_batch % rf::TOK == 0;
}

// wo1 (attn1.to_out.0), wq2 (attn2.to_q), wo2 (attn2.to_out.0): [C][C] fp16 device tensors -> tmpl (cross_block_image_bytes):
// the weight pieces of the image; the K / V pieces stay zero until bind_cross_block().  Synchronous (load time).
int pack_cross_block(const half_t* wo1, const half_t* wq2, const half_t* wo2, int C, half_t* tmpl, hipStream_t stream) {
    using namespace xb;
    LAVIE_CHECK(C == xb::C, "cross_block: width %d is not built (320 only)", C);
    LAVIE_HIP(hipMemsetAsync(tmpl, 0, IMG_BYTES, stream));
    std::vector<int2> lists[3];
    for (int t = 0; t < NT; ++t)
        for (int ks = 0; ks < KS; ++ks) {
            const int piece = t * KS + ks;
            for (int slot = 0; slot < 64; ++slot) {          // k in memory order: k-slot 8 q + j <-> column 32 ks + 8 q + j
                const int r = slot >> 2, q = (slot & 3) ^ rf::swz(r);
                for (int half = 0; half < 2; ++half)
                    lists[0].push_back(make_int2(piece * 128 + slot * 2 + half, ((16 * t + r) * C + 32 * ks + 8 * q + 4 * half) / 4));
            }
        }
    for (int hp = 0; hp < 4; ++hp) {
        const int h0 = 2 * hp, h1 = h0 + 1, P0 = O1_PIECES + PAIR_PIECES * hp;
        for (int j = 0; j < 5; ++j) {
            int rows[16];
            for (int r = 0; r < 16; ++r) rows[r] = pair_channel(h0, j, r);
            for (int ks = 0; ks < KS; ++ks) rf_piece_pairs_rows(lists[1], P0 + Q_OFF + 10 * j + ks, rows, C, 32 * ks);
        }
        for (int u = 0; u < 10; ++u) {
            for (int i = 0; i < 4; ++i)
                rf_piece_pairs(lists[2], P0 + O_OFF + 5 * u + i, 16 * (2 * u + (i >> 1)), C, ((i & 1) ? h1 : h0) * DH);
            const int piece = P0 + O_OFF + 5 * u + 4;        // the 16-deep step of two output tiles: channels 32..39 of both heads
            for (int slot = 0; slot < 64; ++slot) {
                const int r = slot >> 2, q = (slot & 3) ^ rf::swz(r);
                const int colk = q < 2 ? h0 * DH + 32 + 4 * q : h1 * DH + 32 + 4 * (q - 2);
                for (int half = 0; half < 2; ++half)
                    lists[2].push_back(make_int2(piece * 128 + slot * 2 + half, ((16 * (2 * u + half) + r) * C + colk) / 4));
            }
        }
    }
    const half_t* srcs[3] = {wo1, wq2, wo2};
    return rf_run_gathers(lists, srcs, 3, tmpl, stream);
}

// Batched gathers of the K (8-byte chunks) and V^T (single halfs: a transpose) pieces: blockIdx.y = video
static __global__ void xb_gather8_kernel(const uint2* __restrict__ src, uint2* __restrict__ dst, const int2* __restrict__ pairs, int n,
                                         size_t src_stride, size_t dst_stride) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int2 pr = pairs[i];
    dst[blockIdx.y * dst_stride + pr.x] = pr.y >= 0 ? src[blockIdx.y * src_stride + pr.y] : make_uint2(0u, 0u);
}
static __global__ void xb_gather2_kernel(const half_t* __restrict__ src, half_t* __restrict__ dst, const int2* __restrict__ pairs, int n,
                                         size_t src_stride, size_t dst_stride) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int2 pr = pairs[i];
    dst[blockIdx.y * dst_stride + pr.x] = pr.y >= 0 ? src[blockIdx.y * src_stride + pr.y] : (half_t)0.f;
}

namespace {
struct BindPlan {
    int2* k_pairs = nullptr;
    int2* v_pairs = nullptr;
    int nk = 0, nv = 0;
};
std::map<int, BindPlan> g_bind_plans;      // by context length; a few hundred KiB of device memory each, kept for the process
}  // namespace

static int bind_plan(int L, BindPlan** out) {
    using namespace xb;
    auto it = g_bind_plans.find(L);
    if (it != g_bind_plans.end()) { *out = &it->second; return 0; }
    std::vector<int2> kl, vl;
    const int ld = 2 * C;            // kv rows: [k (C) | v (C)]
    for (int hp = 0; hp < 4; ++hp) {
        const int h0 = 2 * hp, h1 = h0 + 1, P0 = O1_PIECES + PAIR_PIECES * hp;
        for (int kt = 0; kt < 5; ++kt)
            for (int slot = 0; slot < 64; ++slot) {
                const int r = slot >> 2, q = (slot & 3) ^ rf::swz(r), key = 16 * kt + r;
                for (int e = 0; e < 2; ++e)            // channels 0..31 of head e: k-slot 8 q + j <-> channel 16 (j >> 2) + 4 q + (j & 3)
                    for (int half = 0; half < 2; ++half)
                        kl.push_back(make_int2((P0 + K_OFF + 3 * kt + e) * 128 + slot * 2 + half,
                                               key < L ? (key * ld + (e ? h1 : h0) * DH + 16 * half + 4 * q) / 4 : -1));
                // channels 32..39: 16-deep fragments of head 0 (low half: k-slots of q < 2) and head 1 (high half: q >= 2)
                kl.push_back(make_int2((P0 + K_OFF + 3 * kt + 2) * 128 + slot * 2 + 0, (key < L && q < 2) ? (key * ld + h0 * DH + 32 + 4 * q) / 4 : -1));
                kl.push_back(make_int2((P0 + K_OFF + 3 * kt + 2) * 128 + slot * 2 + 1, (key < L && q >= 2) ? (key * ld + h1 * DH + 32 + 4 * (q - 2)) / 4 : -1));
            }
        for (int j = 0; j < 5; ++j)
            for (int s = 0; s < 2; ++s)
                for (int slot = 0; slot < 64; ++slot) {
                    const int r = slot >> 2, q = (slot & 3) ^ rf::swz(r), ch = pair_channel(h0, j, r);
                    for (int jj = 0; jj < 8; ++jj) {
                        const int key = 32 * s + 16 * (jj >> 2) + 4 * q + (jj & 3);
                        vl.push_back(make_int2((P0 + V_OFF + 2 * j + s) * 512 + slot * 8 + jj, key < L ? key * ld + C + ch : -1));
                    }
                }
        for (int n = 0; n < 3; ++n)
            for (int slot = 0; slot < 64; ++slot) {
                const int r = slot >> 2, q = (slot & 3) ^ rf::swz(r);
                for (int jj = 0; jj < 8; ++jj) {
                    const int j = 2 * n + (jj >> 2), key = 64 + 4 * q + (jj & 3);
                    vl.push_back(make_int2((P0 + V_OFF + 10 + n) * 512 + slot * 8 + jj, (j < 5 && key < L) ? key * ld + C + pair_channel(h0, j, r) : -1));
                }
            }
    }
    BindPlan plan;
    plan.nk = (int)kl.size();
    plan.nv = (int)vl.size();
    LAVIE_HIP(hipMalloc(&plan.k_pairs, kl.size() * sizeof(int2)));
    LAVIE_HIP(hipMalloc(&plan.v_pairs, vl.size() * sizeof(int2)));
    LAVIE_HIP(hipMemcpy(plan.k_pairs, kl.data(), kl.size() * sizeof(int2), hipMemcpyHostToDevice));
    LAVIE_HIP(hipMemcpy(plan.v_pairs, vl.data(), vl.size() * sizeof(int2), hipMemcpyHostToDevice));
    *out = &(g_bind_plans[L] = plan);
    return 0;
}

// tmpl (pack_cross_block) + kv [B * L][2C] fp16 (k | v rows of attn2.to_k / to_v applied to the text context of each video)
// -> img [B] images.  Stream-ordered (no synchronisation once the plan of this L exists).
int bind_cross_block(const half_t* tmpl, const half_t* kv, int B, int L, int C, half_t* img, hipStream_t stream) {
    using namespace xb;
    LAVIE_CHECK(C == xb::C && L >= 1 && L <= MAXL && B >= 1, "cross_block: C=%d L=%d B=%d is not built", C, L, B);
    BindPlan* plan = nullptr;
    if (int rc = bind_plan(L, &plan)) return rc;
    for (int b = 0; b < B; ++b)
        LAVIE_HIP(hipMemcpyAsync(reinterpret_cast<char*>(img) + b * IMG_BYTES, tmpl, IMG_BYTES, hipMemcpyDeviceToDevice, stream));
    hipLaunchKernelGGL(xb_gather8_kernel, dim3(cdiv(plan->nk, 256), B), dim3(256), 0, stream, (const uint2*)kv, (uint2*)img, plan->k_pairs,
                       plan->nk, (size_t)L * 2 * C / 4, IMG_BYTES / 8);
    hipLaunchKernelGGL(xb_gather2_kernel, dim3(cdiv(plan->nv, 256), B), dim3(256), 0, stream, kv, img, plan->v_pairs, plan->nv,
                       (size_t)L * 2 * C, IMG_BYTES / 2);
    LAVIE_HIP(hipGetLastError());
    return 0;
}

struct CrossBlockParams {
    const half_t* att;        // [M, C] attn1 output (before its to_out projection)
    const half_t* x;          // [M, C] residual stream
    half_t* y;                // [M, C]; may alias x
    const half_t* img;        // [B] images (bind_cross_block)
    const float* bo1;         // attn1.to_out.0.bias
    const float* gamma;       // norm2
    const float* beta;
    const float* bo2;         // attn2.to_out.0.bias
    int tiles;                // 16-row tiles = M / 16
    int tiles_per_batch;      // rows_per_batch / 16
    int L;                    // context length (keys)
    float scale, eps;
};

template <class SyncFn, int... Ks>
__device__ __forceinline__ void xb_idle(SyncFn&& sync, std::integer_sequence<int, Ks...>) {
    (sync(std::integral_constant<int, Ks + 1>{}), ...);
}

template <int PF>
__global__ __launch_bounds__(rf::THREADS, 2) void cross_block_kernel(const CrossBlockParams p) {
    using namespace rf;
    using namespace xb;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const ring = smem;
    float* const vec = reinterpret_cast<float*>(smem + RING_BYTES);           // bo1 | gamma | beta | bo2

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, col = lane & 15;
    const int frag = ((col * 4) + (q ^ swz(col))) * 16;
    for (int i = tid; i < C; i += THREADS) { vec[i] = p.bo1[i]; vec[C + i] = p.gamma[i]; vec[2 * C + i] = p.beta[i]; vec[3 * C + i] = p.bo2[i]; }

    const int nwg = gridDim.x, bid = blockIdx.x;
    const int share = p.tiles / nwg, rem = p.tiles - share * nwg;
    const int tile0 = bid * share + (bid < rem ? bid : rem);
    const int tend = tile0 + share + (bid < rem ? 1 : 0);

    const char* imgb = reinterpret_cast<const char*>(p.img);       // the image of the current pass's video
    auto issue_unit = [&](int u) {
        const char* ip = imgb;
        asm volatile("" : "+s"(ip));          // keeps hipcc from hoisting (and spilling) the per-lane source addresses of a whole pass
        const char* src = ip + ((size_t)(u * UNIT + wave) << 10) + lane * 16;
        char* dst = ring + (((u % RING_GROUPS) * UNIT + wave) << 10);
#pragma unroll
        for (int i = 0; i < UNIT / WAVES; ++i) rf_dma(src + i * (WAVES << 10), dst + i * (WAVES << 10));
    };
    // sync k, in front of segment k (pieces 20 k ..): as in temporal_block_kernel
    auto sync = [&](auto k_) {
        constexpr int K = decltype(k_)::value;
        __builtin_amdgcn_sched_barrier(0);
        constexpr bool younger = (K % 2 == 0) ? (K / 2 + 1 < PASS_UNITS) : ((K + 1) / 2 + 1 < PASS_UNITS);
        if constexpr (younger) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (K % 2 == 0 && K / 2 + 2 < PASS_UNITS) issue_unit(K / 2 + 2);
        __builtin_amdgcn_sched_barrier(0);
    };
    __syncthreads();                 // vectors visible; nothing in flight yet
    int t = tile0;
    if (t < tend) {
        imgb = reinterpret_cast<const char*>(p.img) + (size_t)(t / p.tiles_per_batch) * IMG_BYTES;
        issue_unit(0);
        issue_unit(1);
    }

    const unsigned ring_lo = (unsigned)(size_t)LDS_PTR(ring + frag);
    const unsigned ring_hi = (unsigned)(size_t)LDS_PTR(ring + 60 * 1024 + frag);
    constexpr float LOG2E = 1.4426950408889634f;

    while (t < tend) {
        // a pass = up to 8 tiles of ONE video (its K / V are in the stream)
        int n = tend - t;
        const int to_end = p.tiles_per_batch - t % p.tiles_per_batch;
        n = n < WAVES ? n : WAVES;
        n = n < to_end ? n : to_end;
        const bool active = wave < n;                              // wave-uniform
        const size_t row = (size_t)(t + (active ? wave : 0)) * TOK + col;
        unsigned vec_off = (unsigned)(size_t)LDS_PTR(vec + 4 * q);
        asm volatile("" : "+v"(vec_off));
        auto lds_f4 = [](unsigned off) { return *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>((size_t)off); };
        f32x4 R[NT];
        half8_t xa[KS];
        {
            const half_t* xr = p.x + row * C + 4 * q;
            const half_t* ar = p.att + row * C + 8 * q;
            half4_t raw[NT];
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) raw[tt] = *reinterpret_cast<const half4_t*>(xr + 16 * tt);
#pragma unroll
            for (int s = 0; s < KS; ++s) xa[s] = *reinterpret_cast<const half8_t*>(ar + 32 * s);
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) R[tt] = (f32x4){(float)raw[tt][0], (float)raw[tt][1], (float)raw[tt][2], (float)raw[tt][3]};
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
            if (tt % 5 == 0 && tt > 0) __builtin_amdgcn_sched_barrier(0);
            R[tt] += lds_f4(vec_off + (16 * tt) * 4);              // + bo1
            asm volatile("" : "+v"(R[tt]));
        }
#pragma unroll
        for (int s = 0; s < KS; ++s) asm volatile("" : "+v"(xa[s]));
        __builtin_amdgcn_sched_barrier(0);

        sync(std::integral_constant<int, 0>{});

        if (active) {
            // ---- x' = x + bo1 + Wo1 att
            tb_run<0, O1_PIECES, PF>(ring_lo, ring_hi, ring + frag, [&](auto m_, const half8_t& a) {
                constexpr int M = decltype(m_)::value;
                rf_mfma32(R[M / KS], a, xa[M % KS]);
            }, sync);
            asm volatile("s_nop 15\n\ts_nop 15"
                         : "+v"(R[0]), "+v"(R[1]), "+v"(R[2]), "+v"(R[3]), "+v"(R[4]), "+v"(R[5]), "+v"(R[6]), "+v"(R[7]), "+v"(R[8]), "+v"(R[9]),
                           "+v"(R[10]), "+v"(R[11]), "+v"(R[12]), "+v"(R[13]), "+v"(R[14]), "+v"(R[15]), "+v"(R[16]), "+v"(R[17]), "+v"(R[18]),
                           "+v"(R[19]));
            // ---- LN2(x') as B fragments (accumulator order); then the to_out bias of attn2 joins the residual
            float sum = 0.f;
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) sum += (R[tt][0] + R[tt][1]) + (R[tt][2] + R[tt][3]);
            sum += __shfl_xor(sum, 16, 64);
            sum += __shfl_xor(sum, 32, 64);
            const float mean = sum * (1.0f / C);
            float sq = 0.f;
#pragma unroll
            for (int tt = 0; tt < NT; ++tt)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float d = R[tt][r] - mean; sq += d * d; }
            sq += __shfl_xor(sq, 16, 64);
            sq += __shfl_xor(sq, 32, 64);
            const float rstd = rsqrtf(sq * (1.0f / C) + p.eps);
            half8_t xbf[KS];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                if (s % 2 == 0 && s > 0) __builtin_amdgcn_sched_barrier(0);
                const f32x4 g0 = lds_f4(vec_off + (C + 32 * s) * 4), g1 = lds_f4(vec_off + (C + 32 * s + 16) * 4);
                const f32x4 e0 = lds_f4(vec_off + (2 * C + 32 * s) * 4), e1 = lds_f4(vec_off + (2 * C + 32 * s + 16) * 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    xbf[s][r] = (half_t)((R[2 * s][r] - mean) * rstd * g0[r] + e0[r]);
                    xbf[s][4 + r] = (half_t)((R[2 * s + 1][r] - mean) * rstd * g1[r] + e1[r]);
                }
                asm volatile("" : "+v"(xbf[s]));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) {
                if (tt % 5 == 0 && tt > 0) __builtin_amdgcn_sched_barrier(0);
                R[tt] += lds_f4(vec_off + (3 * C + 16 * tt) * 4);
                asm volatile("" : "+v"(R[tt]));
            }
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 4" ::: "memory");     // no LDS read of this wave in flight when a run starts

            auto pair_body = [&](auto hp_) {
                constexpr int HP = decltype(hp_)::value;
                constexpr int P0 = O1_PIECES + PAIR_PIECES * HP;
                half4_t qp[5], op[5];
                // ---- q tiles D[channel][token] = Wq2 LN2(x'), scaled: two rotating accumulators (see temporal_block_kernel)
                {
                    f32x4 acc[2];
                    auto finish_tile = [&](auto t_) {
                        constexpr int T = decltype(t_)::value;
                        qp[T] = rf_pack(acc[T & 1] * p.scale);
                    };
                    tb_run<P0 + Q_OFF, 50, PF>(ring_lo, ring_hi, ring + frag, [&](auto m_, const half8_t& a) {
                        constexpr int M = decltype(m_)::value;
                        constexpr int T = M / 10, S = M % 10;
                        if constexpr (S == 0) rf_mfma32_first(acc[T & 1], a, xbf[S]);
                        else rf_mfma32(acc[T & 1], a, xbf[S]);
                        if constexpr (S == 4 && T > 0) {
                            asm volatile("" : "+v"(acc[(T - 1) & 1]));
                            finish_tile(std::integral_constant<int, T - 1>{});
                        }
                    }, sync);
                    asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc[0]), "+v"(acc[1]));
                    finish_tile(std::integral_constant<int, 4>{});
                }
                __builtin_amdgcn_sched_barrier(0);

                // ---- S^T[key][token] of both heads: 5 key tiles each
                f32x4 S[2][5];
                {
                    const half4_t z = {(half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};
                    half8_t qc0 = rf_cat(qp[0], qp[1]), qc1 = rf_cat(qp[3], qp[4]);
                    half4_t qm0 = q < 2 ? qp[2] : z, qm1 = q < 2 ? z : qp[2];       // rows of the shared tile that belong to each head
                    half8_t pend;
                    asm volatile("s_nop 4" : "+v"(qc0), "+v"(qc1), "+v"(qm0), "+v"(qm1));
                    tb_run<P0 + K_OFF, 15, PF>(ring_lo, ring_hi, ring + frag, [&](auto m_, const half8_t& a) {
                        constexpr int M = decltype(m_)::value;
                        constexpr int KT = M / 3, I = M % 3;
                        if constexpr (I == 0) rf_mfma32_first(S[0][KT], a, qc0);
                        else if constexpr (I == 1) {
                            rf_mfma32_first(S[1][KT], a, qc1);
                            if constexpr (KT > 0) {          // the previous key tile's 16-deep steps: its 32-deep ones are three MFMAs back
                                const half4_t alo = {pend[0], pend[1], pend[2], pend[3]}, ahi = {pend[4], pend[5], pend[6], pend[7]};
                                rf_mfma16(S[0][KT - 1], alo, qm0);
                                rf_mfma16(S[1][KT - 1], ahi, qm1);
                            }
                        } else {
                            pend = a;
                            if constexpr (KT == 4) {
                                rf_mfma_drain();
                                const half4_t alo = {pend[0], pend[1], pend[2], pend[3]}, ahi = {pend[4], pend[5], pend[6], pend[7]};
                                rf_mfma16(S[0][4], alo, qm0);
                                rf_mfma16(S[1][4], ahi, qm1);
                            }
                        }
                    }, sync);
                    asm volatile("s_nop 15\n\ts_nop 15"
                                 : "+v"(S[0][0]), "+v"(S[0][1]), "+v"(S[0][2]), "+v"(S[0][3]), "+v"(S[0][4]), "+v"(S[1][0]), "+v"(S[1][1]),
                                   "+v"(S[1][2]), "+v"(S[1][3]), "+v"(S[1][4]));
                }
                // ---- softmax over the keys (registers x the four q lanes of a token), P^T packed as B fragments
                half8_t pc[2][2];
                half4_t p4[2];
                float inv[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    float m = -INFINITY;
#pragma unroll
                    for (int kt = 0; kt < 5; ++kt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            if (16 * kt + 4 * q + r >= p.L) S[e][kt][r] = -INFINITY;
                            m = fmaxf(m, S[e][kt][r]);
                        }
                    m = fmaxf(m, __shfl_xor(m, 16, 64));
                    m = fmaxf(m, __shfl_xor(m, 32, 64));
                    float l = 0.f;
                    half4_t pk[5];
#pragma unroll
                    for (int kt = 0; kt < 5; ++kt) {
                        f32x4 ev;
#pragma unroll
                        for (int r = 0; r < 4; ++r) { ev[r] = __builtin_amdgcn_exp2f((S[e][kt][r] - m) * LOG2E); l += ev[r]; }
                        pk[kt] = rf_pack(ev);
                    }
                    l += __shfl_xor(l, 16, 64);
                    l += __shfl_xor(l, 32, 64);
                    inv[e] = __builtin_amdgcn_rcpf(l);
                    pc[e][0] = rf_cat(pk[0], pk[1]);
                    pc[e][1] = rf_cat(pk[2], pk[3]);
                    p4[e] = pk[4];
                }
                __builtin_amdgcn_sched_barrier(0);

                // ---- O[channel][token] = V^T P^T: five channel tiles (the shared one once per head)
                {
                    f32x4 O[5], osh[2];
                    asm volatile("s_nop 4" : "+v"(pc[0][0]), "+v"(pc[0][1]), "+v"(pc[1][0]), "+v"(pc[1][1]), "+v"(p4[0]), "+v"(p4[1]));
                    tb_run<P0 + V_OFF, 13, PF>(ring_lo, ring_hi, ring + frag, [&](auto m_, const half8_t& a) {
                        constexpr int M = decltype(m_)::value;
                        if constexpr (M < 10) {
                            constexpr int J = M / 2, SS = M % 2;
                            if constexpr (J == 2) {
                                if constexpr (SS == 0) { rf_mfma32_first(osh[0], a, pc[0][0]); rf_mfma32_first(osh[1], a, pc[1][0]); }
                                else { rf_mfma32(osh[0], a, pc[0][1]); rf_mfma32(osh[1], a, pc[1][1]); }
                            } else {
                                constexpr int E = J < 2 ? 0 : 1;
                                if constexpr (SS == 0) rf_mfma32_first(O[J], a, pc[E][0]);
                                else rf_mfma32(O[J], a, pc[E][1]);
                            }
                        } else {                                 // keys 64..79: 16-deep steps, two tiles per piece
                            const half4_t alo = {a[0], a[1], a[2], a[3]}, ahi = {a[4], a[5], a[6], a[7]};
                            if constexpr (M == 10) { rf_mfma16(O[0], alo, p4[0]); rf_mfma16(O[1], ahi, p4[0]); }
                            else if constexpr (M == 11) { rf_mfma16(osh[0], alo, p4[0]); rf_mfma16(osh[1], alo, p4[1]); rf_mfma16(O[3], ahi, p4[1]); }
                            else rf_mfma16(O[4], alo, p4[1]);
                        }
                    }, sync);
                    asm volatile("s_nop 15\n\ts_nop 15" : "+v"(O[0]), "+v"(O[1]), "+v"(O[3]), "+v"(O[4]), "+v"(osh[0]), "+v"(osh[1]));
                    op[0] = rf_pack(O[0] * inv[0]);
                    op[1] = rf_pack(O[1] * inv[0]);
                    op[3] = rf_pack(O[3] * inv[1]);
                    op[4] = rf_pack(O[4] * inv[1]);
                    op[2] = rf_pack(q < 2 ? osh[0] * inv[0] : osh[1] * inv[1]);
                }
                __builtin_amdgcn_sched_barrier(0);

                // ---- x'' += Wo2 O (as the temporal kernel's to_out product)
                half8_t o0 = rf_cat(op[0], op[1]), o1 = rf_cat(op[3], op[4]);
                half8_t pend;
                asm volatile("s_nop 4" : "+v"(o0), "+v"(o1), "+v"(op[2]));
                tb_run<P0 + O_OFF, 50, PF>(ring_lo, ring_hi, ring + frag, [&](auto m_, const half8_t& a) {
                    constexpr int M = decltype(m_)::value;
                    constexpr int U = M / 5, I = M % 5;
                    if constexpr (I < 4) {
                        constexpr int T2 = 2 * U + (I >> 1);
                        rf_mfma32(R[T2], a, (I & 1) ? o1 : o0);
                        if constexpr (I == 3 && U > 0) {
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
                asm volatile("s_nop 15\n\ts_nop 15"
                             : "+v"(R[0]), "+v"(R[1]), "+v"(R[2]), "+v"(R[3]), "+v"(R[4]), "+v"(R[5]), "+v"(R[6]), "+v"(R[7]), "+v"(R[8]), "+v"(R[9]),
                               "+v"(R[10]), "+v"(R[11]), "+v"(R[12]), "+v"(R[13]), "+v"(R[14]), "+v"(R[15]), "+v"(R[16]), "+v"(R[17]), "+v"(R[18]),
                               "+v"(R[19]));
            };
            pair_body(std::integral_constant<int, 0>{});
            pair_body(std::integral_constant<int, 1>{});
            pair_body(std::integral_constant<int, 2>{});
            pair_body(std::integral_constant<int, 3>{});
        } else {
            // a wave without a tile in this pass still moves its share of the stream and meets every barrier
            xb_idle(sync, std::make_integer_sequence<int, PASS_PIECES / tb::SEG - 1>{});
        }

        // end of pass: every wave is past the last segment, so the ring's first two slots may take the next pass's first units
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        t += n;
        if (t < tend) {
            imgb = reinterpret_cast<const char*>(p.img) + (size_t)(t / p.tiles_per_batch) * IMG_BYTES;
            issue_unit(0);
            issue_unit(1);
        }
        if (active) {
            half_t* yr = p.y + row * C + 4 * q;
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) *reinterpret_cast<half4_t*>(yr + 16 * tt) = rf_pack(R[tt]);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

int launch_cross_block(const half_t* att, const half_t* x, half_t* y, int M, int rows_per_batch, int C, int heads, const half_t* img,
                       const float* bo1, const float* gamma, const float* beta, const float* bo2, int L, float scale, float eps,
                       hipStream_t stream) {
    LAVIE_CHECK(cross_block_supported(C, heads, L, rows_per_batch), "cross_block: C=%d heads=%d L=%d rows_per_batch=%d is not built", C, heads, L,
                rows_per_batch);
    LAVIE_CHECK(att && x && y && img && bo1 && gamma && beta && bo2 && M > 0 && M % rows_per_batch == 0, "cross_block: bad arguments");
    const double tok = (double)M;
    // algorithmic work: Wo1, Wq2, Wo2 and the two attention products over L keys; bytes: att and x in, x'' out, the image once
    ProfileScope prof(KC_FUSED_CROSS, stream, 2.0 * tok * C * 3.0 * C + 4.0 * tok * L * C,
                      2.0 * 3.0 * tok * C + (double)(M / rows_per_batch) * xb::IMG_BYTES, /*kernel_events=*/true);
    CrossBlockParams p;
    p.att = att; p.x = x; p.y = y; p.img = img; p.bo1 = bo1; p.gamma = gamma; p.beta = beta; p.bo2 = bo2;
    p.tiles = M / rf::TOK; p.tiles_per_batch = rows_per_batch / rf::TOK; p.L = L; p.scale = scale; p.eps = eps;
    constexpr int lds = rf::RING_BYTES + xb::VEC_BYTES;
    const int grid = p.tiles < 256 ? p.tiles : 256;
    auto go = [&](auto kern) -> int {
        if (int rc = ensure_dynamic_lds((const void*)kern, lds)) return rc;     // once per kernel address, not per launch
        if (prof.active()) hipExtLaunchKernelGGL(kern, dim3(grid), dim3(rf::THREADS), lds, stream, prof.start(), prof.stop(), 0, p);
        else hipLaunchKernelGGL(kern, dim3(grid), dim3(rf::THREADS), lds, stream, p);
        LAVIE_HIP(hipGetLastError());
        return 0;
    };
    switch (rowfuse_variant()) {
        case 5: return go(cross_block_kernel<4>);
        default: return go(cross_block_kernel<8>);
    }
}

}  // namespace lavie
