// GroupNorm (two reduction domains) and LayerNorm on channels-last fp16 activations.
//
// Reference ops replaced:
//   * nn.GroupNorm on the 5-D video tensor — statistics over (C/G, F, H, W), i.e. ACROSS frames
//     (resnet.py:180,191; unet.py:504), followed by SiLU (resnet.py:181,197; unet.py:505);
//   * nn.GroupNorm per frame, eps 1e-6, no activation (attention.py:324,369);
//   * nn.LayerNorm(C) (attention.py:442,459,474,480).
// Both GroupNorm domains use the same two kernels: the caller chooses what a "batch" is
// (video: NB = B, P = F*h*w rows; per frame: NB = B*F, P = h*w rows).  The input may be the
// virtual channel concatenation [x1 | x2] of the up path (unet_blocks.py:538): a group may
// straddle the two tensors, and only the normalised result is ever materialised.
//
// All three are HBM-bound: 16-byte vector accesses, fp32 statistics, wave/LDS reductions.
#include "common.h"
#include "igemm.h"
#include "ops.h"
#include "profile.h"

#include <math.h>
#include <vector>

namespace lavie {

constexpr int GN_THREADS = 256;
constexpr int GN_MAX_C = 4096;
constexpr int GN_UNROLL = 4;       // rows a thread keeps in flight
constexpr int GN_MAX_SLABS = 2048;  // NB * slabs of the statistics pass (one finalize wave folds <= 512 partials)

struct GnGeom {
    int nvec;   // (C1 + C2) / 8
    int tx;     // channel-vector lanes
    int vpt;    // vectors per thread (nvec = tx * vpt)
    int ty;     // row lanes = GN_THREADS / tx
};

static bool gn_geometry(int ctot, GnGeom* g) {
    if (ctot % 8 != 0 || ctot > GN_MAX_C) return false;
    g->nvec = ctot / 8;
    g->vpt = 1;
    while (g->nvec / g->vpt > GN_THREADS || g->nvec % g->vpt != 0) {
        if (++g->vpt > 4) return false;
    }
    g->tx = g->nvec / g->vpt;
    g->ty = GN_THREADS / g->tx;
    return true;
}

__device__ __forceinline__ const half_t* gn_src(const half_t* x1, int C1, const half_t* x2, int C2, size_t row, int c) {
    return c < C1 ? x1 + row * C1 + c : x2 + row * C2 + (c - C1);
}

// Pass 1: per-slab partial (sum, sum of squares) of every group -> partials[(nb*slabs + slab)*groups + g].
// No atomics anywhere: fixed summation order, so a forward pass is bit-reproducible.
template <int VPT>
__global__ __launch_bounds__(GN_THREADS) void gn_stats_kernel(const half_t* __restrict__ x1, int C1,
                                                             const half_t* __restrict__ x2, int C2, int P,
                                                             int rows_per_slab, int groups, int tx, int ty,
                                                             float* __restrict__ partials) {
    extern __shared__ __attribute__((aligned(16))) char gn_smem[];
    float* s_part = reinterpret_cast<float*>(gn_smem);          // [ty][ctot][2]
    const int ctot = C1 + C2;
    const int nb = blockIdx.y;
    const int tid = threadIdx.x;
    const int vx = tid % tx, vy = tid / tx;
    const int r0 = blockIdx.x * rows_per_slab;
    const int r1 = min(P, r0 + rows_per_slab);
    if (vy < ty) {
        float s[VPT][8], q[VPT][8];
#pragma unroll
        for (int v = 0; v < VPT; ++v)
#pragma unroll
            for (int j = 0; j < 8; ++j) { s[v][j] = 0.f; q[v][j] = 0.f; }
        // GN_UNROLL rows in flight per thread (all loads first): one row per iteration left the kernel latency-bound at
        // ~3.2 TB/s; the summation order per thread is unchanged (rows in ascending order)
        int r = r0 + vy;
        for (; r + (GN_UNROLL - 1) * ty < r1; r += GN_UNROLL * ty) {
            half8_t h[GN_UNROLL][VPT];
#pragma unroll
            for (int u = 0; u < GN_UNROLL; ++u)
#pragma unroll
                for (int v = 0; v < VPT; ++v)
                    h[u][v] = *reinterpret_cast<const half8_t*>(gn_src(x1, C1, x2, C2, (size_t)nb * P + r + u * ty, (vx + v * tx) * 8));
#pragma unroll
            for (int u = 0; u < GN_UNROLL; ++u)
#pragma unroll
                for (int v = 0; v < VPT; ++v)
#pragma unroll
                    for (int j = 0; j < 8; ++j) { const float f = (float)h[u][v][j]; s[v][j] += f; q[v][j] += f * f; }
        }
        for (; r < r1; r += ty) {
            const size_t row = (size_t)nb * P + r;
#pragma unroll
            for (int v = 0; v < VPT; ++v) {
                const int c = (vx + v * tx) * 8;
                const half8_t h = *reinterpret_cast<const half8_t*>(gn_src(x1, C1, x2, C2, row, c));
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float f = (float)h[j]; s[v][j] += f; q[v][j] += f * f; }
            }
        }
#pragma unroll
        for (int v = 0; v < VPT; ++v) {
            const int c = (vx + v * tx) * 8;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                s_part[((size_t)vy * ctot + c + j) * 2] = s[v][j];
                s_part[((size_t)vy * ctot + c + j) * 2 + 1] = q[v][j];
            }
        }
    }
    __syncthreads();
    const int cpg = ctot / groups;
    for (int g = tid; g < groups; g += GN_THREADS) {
        float a = 0.f, b = 0.f;
        for (int c = g * cpg; c < (g + 1) * cpg; ++c)
            for (int y = 0; y < ty; ++y) { a += s_part[((size_t)y * ctot + c) * 2]; b += s_part[((size_t)y * ctot + c) * 2 + 1]; }
        float* dst = partials + (((size_t)nb * gridDim.x + blockIdx.x) * groups + g) * 2;
        dst[0] = a;
        dst[1] = b;
    }
}

// Pass 2: one wave per (batch, group): fold the slab partials in a fixed order -> (mean, rstd).
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ partials, int slabs, int groups,
                                                         int total, float inv_count, float eps,
                                                         float* __restrict__ stats) {
    const int lane = threadIdx.x & 63;
    const int idx = blockIdx.x * 4 + (threadIdx.x >> 6);        // nb * groups + g
    if (idx >= total) return;
    const int nb = idx / groups, g = idx - nb * groups;
    // 16 partials per lane in flight (all loads, then the adds in the same ascending order): the one-load-per-iteration loop
    // took 5.9 us for 1024 slabs — sixteen dependent L2 round trips — and ran 61 times per forward
    float a = 0.f, b = 0.f;
    constexpr int FU = 16;
    for (int s0 = lane; s0 < slabs; s0 += 64 * FU) {
        float2 v[FU];
#pragma unroll
        for (int u = 0; u < FU; ++u) {
            const int sl = s0 + u * 64;
            const int sc = sl < slabs ? sl : slabs - 1;
            v[u] = *reinterpret_cast<const float2*>(partials + (((size_t)nb * slabs + sc) * groups + g) * 2);
        }
#pragma unroll
        for (int u = 0; u < FU; ++u)
            if (s0 + u * 64 < slabs) { a += v[u].x; b += v[u].y; }
    }
    a = wave_sum(a);
    b = wave_sum(b);
    if (lane == 0) {
        const float mean = a * inv_count;
        const float var = fmaxf(b * inv_count - mean * mean, 0.f);
        stats[(size_t)idx * 2] = mean;
        stats[(size_t)idx * 2 + 1] = rsqrtf(var + eps);
    }
}

// Pass 1 + 2 replaced (round 4): the producers of x1 / x2 left per-(row block, channel) sums in their epilogues (igemm.h
// colstat_out); one workgroup per (batch, group) folds the blocks of its statistics domain and the channels of its group — which may
// straddle the two tensors of a skip concatenation — in a fixed order -> (mean, rstd).  Reads sizeof(tensor) / 10 .. / 40 bytes.
struct GnFoldSrc {
    const float* partials;
    int C, c0;              // channels of the tensor; first channel of the tensor inside the concatenation
    int nsets, set_blocks;  // sets of blocks, blocks per set
    int bpd;                // blocks of one statistics domain inside a set
};
__global__ __launch_bounds__(256) void gn_fold_kernel(GnFoldSrc s1, GnFoldSrc s2, int groups, int cpg, float inv_count, float eps,
                                                     float* __restrict__ stats) {
    __shared__ float red[2][4];
    const int idx = blockIdx.x;                 // nb * groups + g
    const int nb = idx / groups, g = idx - nb * groups;
    const int glo = g * cpg, ghi = glo + cpg;   // channel range of the group in the concatenation
    float a = 0.f, b = 0.f;
    auto fold = [&](const GnFoldSrc& s) {
        const int lo = max(glo, s.c0) - s.c0, hi = min(ghi, s.c0 + s.C) - s.c0;      // the group's channels inside this tensor
        if (s.partials == nullptr || hi <= lo) return;
        const int q0 = lo >> 2, nq = ((hi + 3) >> 2) - q0;
        const int items = s.nsets * s.bpd * nq;
        // four items per thread in flight (all loads, then the adds in ascending item order): one dependent L2 round trip per item
        // made this kernel 5-6 us for the level-0 tensors (64 workgroups, 8 items per thread)
        constexpr int FU = 4;
        for (int it0 = threadIdx.x; it0 < items; it0 += 256 * FU) {
            f32x4 sv[FU], qv[FU];
            int cq[FU];
#pragma unroll
            for (int u = 0; u < FU; ++u) {
                const int it = it0 + u * 256;
                const int itc = it < items ? it : items - 1;
                const int qi = itc % nq, rest = itc / nq;
                const int blk = rest % s.bpd, set = rest / s.bpd;
                const size_t block = (size_t)set * s.set_blocks + (size_t)nb * s.bpd + blk;
                const float* src = s.partials + cs_index(block, (q0 + qi) * 4, 0, s.C);
                sv[u] = *reinterpret_cast<const f32x4*>(src);
                qv[u] = *reinterpret_cast<const f32x4*>(src + 4);
                cq[u] = it < items ? (q0 + qi) * 4 : -8;        // -8: every channel of the quad falls outside [lo, hi)
            }
#pragma unroll
            for (int u = 0; u < FU; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = cq[u] + r;
                    if (c >= lo && c < hi) { a += sv[u][r]; b += qv[u][r]; }
                }
        }
    };
    fold(s1);
    fold(s2);
    a = wave_sum(a);
    b = wave_sum(b);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float sa = ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3], sb = ((red[1][0] + red[1][1]) + red[1][2]) + red[1][3];
        const float mean = sa * inv_count;
        const float var = fmaxf(sb * inv_count - mean * mean, 0.f);
        stats[(size_t)idx * 2] = mean;
        stats[(size_t)idx * 2 + 1] = rsqrtf(var + eps);
    }
}

static long g_gn_from_producers = 0;      // GroupNorm launches that took their statistics from the producers (lavie_debug_gn_producer_count)
long gn_producer_count() { return g_gn_from_producers; }
// whether the producer statistics of a tensor can serve a GroupNorm over NB domains of P rows each
static bool gn_colstat_usable(const GnColStat* cs, int C, int P) {
    if (cs == nullptr || cs->partials == nullptr || cs->rows <= 0 || cs->C != C || cs->nsets < 1 || C % 4 != 0) return false;
    // every block inside one domain: the domain is a whole number of spans and of blocks (parity sets: P / nsets source rows per
    // domain and set)
    return cs->span > 0 && P % cs->span == 0 && P % (cs->rows * cs->nsets) == 0;
}

// y[row, :] = act((x - mean_g) * rstd_g * gamma + beta) for the slab, y is [NB*P, C1+C2] row-major.
template <int VPT, bool SILU>
__global__ __launch_bounds__(GN_THREADS) void gn_apply_kernel(const half_t* __restrict__ x1, int C1,
                                                             const half_t* __restrict__ x2, int C2, int P,
                                                             int rows_per_slab, int groups, int tx, int ty,
                                                             const float* __restrict__ stats,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta,
                                                             half_t* __restrict__ y) {
    __shared__ float s_a[GN_MAX_C];
    __shared__ float s_b[GN_MAX_C];
    const int ctot = C1 + C2;
    const int cpg = ctot / groups;
    const int nb = blockIdx.y;
    const int tid = threadIdx.x;
    for (int c = tid; c < ctot; c += GN_THREADS) {
        const float* st = stats + ((size_t)nb * groups + c / cpg) * 2;
        const float a = st[1] * gamma[c];
        s_a[c] = a;
        s_b[c] = beta[c] - st[0] * a;
    }
    __syncthreads();
    const int vx = tid % tx, vy = tid / tx;
    if (vy >= ty) return;
    const int r0 = blockIdx.x * rows_per_slab;
    const int r1 = min(P, r0 + rows_per_slab);
    auto one = [&](size_t row, int c, const half8_t& h) {
        half8_t o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float f = (float)h[j] * s_a[c + j] + s_b[c + j];
            if (SILU) f = silu_f(f);
            o[j] = (half_t)f;
        }
        *reinterpret_cast<half8_t*>(y + row * ctot + c) = o;
    };
    int r = r0 + vy;
    for (; r + (GN_UNROLL - 1) * ty < r1; r += GN_UNROLL * ty) {      // GN_UNROLL rows in flight per thread
        half8_t h[GN_UNROLL][VPT];
#pragma unroll
        for (int u = 0; u < GN_UNROLL; ++u)
#pragma unroll
            for (int v = 0; v < VPT; ++v)
                h[u][v] = *reinterpret_cast<const half8_t*>(gn_src(x1, C1, x2, C2, (size_t)nb * P + r + u * ty, (vx + v * tx) * 8));
#pragma unroll
        for (int u = 0; u < GN_UNROLL; ++u)
#pragma unroll
            for (int v = 0; v < VPT; ++v) one((size_t)nb * P + r + u * ty, (vx + v * tx) * 8, h[u][v]);
    }
    for (; r < r1; r += ty) {
        const size_t row = (size_t)nb * P + r;
#pragma unroll
        for (int v = 0; v < VPT; ++v) {
            const int c = (vx + v * tx) * 8;
            one(row, c, *reinterpret_cast<const half8_t*>(gn_src(x1, C1, x2, C2, row, c)));
        }
    }
}

static int gn_slabs(int P, int NB, int ty, int cap_total, int* rows_per_slab) {
    int slabs = cdiv(P, ty * 4);                 // at least 4 rows per row-lane
    const int cap = NB >= cap_total ? 1 : cap_total / NB;
    if (slabs > cap) slabs = cap;
    if (slabs < 1) slabs = 1;
    // whole unrolled iterations per row lane: a slab of 40 rows on 6 row lanes ran one 4-row iteration and then up to three
    // rows one dependent load at a time
    const int quantum = ty * GN_UNROLL;
    *rows_per_slab = cdiv(cdiv(P, slabs), quantum) * quantum;
    return cdiv(P, *rows_per_slab);
}

size_t gn_workspace_floats(int NB, int groups) { return ((size_t)GN_MAX_SLABS + NB) * groups * 2 + (size_t)NB * groups * 2; }

// stats_ws layout: [NB*groups*2 (mean, rstd)] [partials: NB*slabs*groups*2]
// (mean, rstd) per (batch, group) -> the normalisation as per-channel pairs: y = a x + b with a = rstd gamma, b = beta - mean a
__global__ void gn_affine_kernel(const float* __restrict__ stats, const float* __restrict__ gamma, const float* __restrict__ beta,
                                 int C, int cpg, int groups, int total, float* __restrict__ ab) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;       // nb * C + c
    if (i >= total) return;
    const int nb = i / C, c = i - nb * C;
    const float* st = stats + ((size_t)nb * groups + c / cpg) * 2;
    const float a = st[1] * gamma[c];
    ab[(size_t)i * 2] = a;
    ab[(size_t)i * 2 + 1] = beta[c] - st[0] * a;
}

int launch_group_norm(const half_t* x1, int C1, const half_t* x2, int C2, int NB, int P, int groups, const float* gamma,
                      const float* beta, float eps, bool silu, float* ws, half_t* y, hipStream_t stream, const GnColStat* cs1,
                      const GnColStat* cs2, float* ab_out) {
    GnGeom g;
    const int ctot = C1 + C2;
    LAVIE_CHECK(gn_geometry(ctot, &g), "group_norm: unsupported channel count %d", ctot);
    LAVIE_CHECK(C1 % 8 == 0 && C2 % 8 == 0 && ctot % groups == 0 && groups <= GN_THREADS, "group_norm: bad channels/groups");
    ProfileScope prof(KC_GROUPNORM, stream, 0.0, 2.0 * 3.0 * (double)NB * P * ctot);   // read, read, write (the two-pass definition, also when the producers' statistics save the first read)
    float* stats = ws;
    float* partials = ws + (size_t)NB * groups * 2;
    const bool from_producers = gn_colstat_usable(cs1, C1, P) && (C2 == 0 || gn_colstat_usable(cs2, C2, P));
    // lavie_debug_fused_mask bit 6 (debug, synchronises): both paths run and every (mean, rstd) pair of the fold is compared with the
    // statistics pass on the host — the direct check of each producer's epilogue sums (tests/test_gpu_engine.py)
    if (from_producers) ++g_gn_from_producers;
    const bool verify = from_producers && (fused_mask() & 64);
    float* const fold_stats = stats;           // what the apply pass reads, in both modes
    float* pass_stats = stats;                 // verify: the statistics pass writes a scratch copy instead
    if (verify) LAVIE_HIP(hipMalloc(&pass_stats, (size_t)NB * groups * 2 * sizeof(float)));
    if (from_producers) {
        auto src = [&](const GnColStat* cs, int c0) {
            GnFoldSrc s{nullptr, 0, c0, 1, 0, 0};
            if (cs) { s.partials = cs->partials; s.C = cs->C; s.nsets = cs->nsets; s.set_blocks = cs->set_blocks; s.bpd = P / (cs->rows * cs->nsets); }
            return s;
        };
        hipLaunchKernelGGL(gn_fold_kernel, dim3(NB * groups), dim3(256), 0, stream, src(cs1, 0), src(C2 ? cs2 : nullptr, C1), groups,
                           ctot / groups, 1.0f / ((float)P * (float)(ctot / groups)), eps, fold_stats);
        LAVIE_HIP(hipGetLastError());
    }
    if (!from_producers || verify) {
    int rps;
    const int slabs = gn_slabs(P, NB, g.ty, GN_MAX_SLABS, &rps);      // NB * slabs <= GN_MAX_SLABS + NB
    const size_t lds = (size_t)g.ty * ctot * 2 * sizeof(float);
    dim3 grid(slabs, NB);
#define LAVIE_GN_STATS(V) \
    hipLaunchKernelGGL(gn_stats_kernel<V>, grid, dim3(GN_THREADS), lds, stream, x1, C1, x2, C2, P, rps, groups, g.tx, g.ty, partials)
    switch (g.vpt) {
        case 1: LAVIE_GN_STATS(1); break;
        case 2: LAVIE_GN_STATS(2); break;
        case 3: LAVIE_GN_STATS(3); break;
        default: LAVIE_GN_STATS(4); break;
    }
#undef LAVIE_GN_STATS
    LAVIE_HIP(hipGetLastError());
    const int total = NB * groups;
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(cdiv(total, 4)), dim3(256), 0, stream, partials, slabs, groups, total,
                       1.0f / ((float)P * (float)(ctot / groups)), eps, pass_stats);
    LAVIE_HIP(hipGetLastError());
    }
    if (verify) {
        std::vector<float> a((size_t)NB * groups * 2), b(a.size());
        LAVIE_HIP(hipStreamSynchronize(stream));
        LAVIE_HIP(hipMemcpy(a.data(), fold_stats, a.size() * sizeof(float), hipMemcpyDeviceToHost));
        LAVIE_HIP(hipMemcpy(b.data(), pass_stats, b.size() * sizeof(float), hipMemcpyDeviceToHost));
        (void)hipFree(pass_stats);
        for (size_t i = 0; i < a.size(); i += 2) {
            // mean: absolute against the group's spread 1 / rstd; rstd: relative.  Both paths sum the same fp16 values in fp32.
            const float dm = fabsf(a[i] - b[i]) * b[i + 1], dr = fabsf(a[i + 1] - b[i + 1]) / b[i + 1];
            LAVIE_CHECK(dm < 1e-3f && dr < 1e-3f && a[i] == a[i] && a[i + 1] == a[i + 1],
                        "group_norm: producer statistics differ from the statistics pass at (batch, group) %zu of %d x %d (C %d+%d, P %d): mean %g vs %g, "
                        "rstd %g vs %g", i / 2, NB, groups, C1, C2, P, a[i], b[i], a[i + 1], b[i + 1]);
        }
    }
    if (ab_out) {        // statistics only: the consumer (rowfuse_pin.hip) applies the norm in registers
        const int total = NB * ctot;
        hipLaunchKernelGGL(gn_affine_kernel, dim3(cdiv(total, 256)), dim3(256), 0, stream, stats, gamma, beta, ctot, ctot / groups, groups, total, ab_out);
        LAVIE_HIP(hipGetLastError());
        return 0;
    }
    int rps2;
    const int slabs2 = gn_slabs(P, NB, g.ty, 2048, &rps2);
    dim3 grid2(slabs2, NB);
#define LAVIE_GN_APPLY(V, S)                                                                                        \
    hipLaunchKernelGGL((gn_apply_kernel<V, S>), grid2, dim3(GN_THREADS), 0, stream, x1, C1, x2, C2, P, rps2, groups, \
                       g.tx, g.ty, stats, gamma, beta, y)
#define LAVIE_GN_APPLY_V(V) \
    if (silu) LAVIE_GN_APPLY(V, true); else LAVIE_GN_APPLY(V, false)
    switch (g.vpt) {
        case 1: LAVIE_GN_APPLY_V(1); break;
        case 2: LAVIE_GN_APPLY_V(2); break;
        case 3: LAVIE_GN_APPLY_V(3); break;
        default: LAVIE_GN_APPLY_V(4); break;
    }
#undef LAVIE_GN_APPLY_V
#undef LAVIE_GN_APPLY
    LAVIE_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------- LayerNorm
// One wave per token row; the row stays in registers between the mean and the variance pass.
constexpr int LN_MAX_VEC = 4;   // C <= 4 * 64 * 8 = 2048

__global__ __launch_bounds__(256) void layernorm_kernel(const half_t* __restrict__ x, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, half_t* __restrict__ y,
                                                       int rows, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nvec = C >> 3;
    const half_t* xr = x + (size_t)row * C;
    half8_t v[LN_MAX_VEC];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAX_VEC; ++i) {
        const int vi = lane + i * 64;
        if (vi < nvec) {
            v[i] = *reinterpret_cast<const half8_t*>(xr + vi * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) sum += (float)v[i][j];
        }
    }
    const float mean = wave_sum(sum) / (float)C;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAX_VEC; ++i) {
        if (lane + i * 64 < nvec) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = (float)v[i][j] - mean; sq += d * d; }
        }
    }
    const float rstd = rsqrtf(wave_sum(sq) / (float)C + eps);
    half_t* yr = y + (size_t)row * C;
#pragma unroll
    for (int i = 0; i < LN_MAX_VEC; ++i) {
        const int vi = lane + i * 64;
        if (vi < nvec) {
            half8_t o;
            const f32x4 g0 = *reinterpret_cast<const f32x4*>(gamma + vi * 8);
            const f32x4 g1 = *reinterpret_cast<const f32x4*>(gamma + vi * 8 + 4);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(beta + vi * 8);
            const f32x4 b1 = *reinterpret_cast<const f32x4*>(beta + vi * 8 + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                o[j] = (half_t)(((float)v[i][j] - mean) * rstd * g0[j] + b0[j]);
                o[j + 4] = (half_t)(((float)v[i][j + 4] - mean) * rstd * g1[j] + b1[j]);
            }
            *reinterpret_cast<half8_t*>(yr + vi * 8) = o;
        }
    }
}

int launch_layernorm(const half_t* x, const float* gamma, const float* beta, half_t* y, int rows, int C, float eps,
                     hipStream_t stream) {
    LAVIE_CHECK(C % 8 == 0 && C <= LN_MAX_VEC * 64 * 8, "layer_norm: unsupported width %d", C);
    ProfileScope prof(KC_LAYERNORM, stream, 0.0, 2.0 * 2.0 * (double)rows * C);
    hipLaunchKernelGGL(layernorm_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, stream, x, gamma, beta, y, rows, C, eps);
    LAVIE_HIP(hipGetLastError());
    return 0;
}

}  // namespace lavie
