// Fused attention core for spatial self-attention and text cross-attention:
//   O = softmax(scale * Q K^T) V   per (frame, head), never materialising the [Lq, Lk] scores.
// Replaces `CrossAttention._attention` (attention.py:209-239: baddbmm / softmax / bmm) together with
// the head split / merge reshapes (attention.py:112-124): heads stay packed along the channel axis
// of the projection outputs, the kernel indexes them in place.
//
// Structure (per workgroup = 4 waves, one (q-block, head, frame)):
//  * each wave owns QT x 16 query rows; K/V tiles of 64 keys are staged through LDS for all waves
//    (register-staged: next tile's global loads are issued before the current tile's MFMAs and
//    written to LDS after them — T14 of the guide);
//  * S^T = K Q^T with v_mfma_f32_16x16x32_f16 (keys on the accumulator rows, the query on the lane),
//    so the row softmax is 16 in-register values + two cross-lane steps, and P^T is ALREADY the B
//    operand of the second product (guide §3 "accumulator tile as the next MFMA's operand");
//  * O^T = V^T P^T: V^T fragments come from the row-major V tile with ds_read_b64_tr_b16
//    (hardware transpose, T10); rows are padded by 32 B so both the K ds_read_b128 and the V
//    transposed reads are bank-conflict-free (stride = 32 B x odd);
//  * online softmax in fp32 with exp2, head dims that are not MFMA multiples (40, 80) are zero
//    padded in LDS only — HBM traffic stays at the true head width.
#include <utility>

#include "common.h"
#include "ops.h"
#include "profile.h"

namespace lavie {

constexpr int ATT_KEYS = 64;   // keys per tile
constexpr float RESCALE_THR = 8.0f;   // log2 units: running max moves only when a row grows by > 2^8

template <int DHP>
struct AttTile {
    static constexpr int STRIDE = DHP * 2 + 32;          // bytes per key row in LDS
    static constexpr int TILE_BYTES = ATT_KEYS * STRIDE; // one of K / V
    static constexpr int LDS_BYTES = 4 * TILE_BYTES;      // 2 buffers x (K + V)
    static constexpr int KS = DHP / 32;                  // k-steps of the QK^T contraction
    static constexpr int DT = DHP / 16;                  // 16-wide output dim tiles (upper bound)
    static constexpr int MAXPIECE = (ATT_KEYS * (DHP / 8) + 255) / 256;
};

// ABL (diagnostic, wrong results): 1 = no softmax VALU (P := S), 2 = no MFMA, 3 = staging + barriers only
// LSUM: the softmax row sums come out of the P V product itself: column `dh` of every V row in LDS (padding that the
// staging never writes; it lies inside the last 16-wide output tile whenever dh % 16 != 0) holds 1.0, so output
// dimension dh accumulates sum_k p[q, k] on the matrix pipe and is rescaled together with O — 16 v_add_f32 per query
// tile and key tile less on the VALU, which paces this kernel at head dim 40.
// SC: sparse-causal key/value addressing (AttnParams::sc_frames): key j of batch entry (b, f) is token j % D of frame
// (b, 0) for j < D and of frame (b, max(f-1, 0)) for j >= D — a per-piece row lookup in the staging loads, nothing else.
// NDT: number of 16-wide output-dimension tiles that exist ((dh + 15) / 16) as a compile-time constant for the model's head
// dims (40 -> 3, 80 -> 5, 160 -> 10); 0 = decided at run time.  The run-time test put a branch and an exposed
// ds_read -> wait -> MFMA chain around every output tile of the P V product.
template <int DHP, int QT, int ABL = 0, bool LSUM = false, bool SC = false, int NDT = 0>
__global__ __launch_bounds__(256, (DHP == 64 && QT == 2) ? 3 : 1) void attention_kernel(const AttnParams p) {
    using T = AttTile<DHP>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sK = smem;
    char* sV = smem + T::TILE_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int g = lane >> 4;       // 16-lane group
    const int li = lane & 15;
    const int head = blockIdx.y;
    const int qb = blockIdx.z;
    const int kvb = qb / p.kv_batch_div;
    const int dh = p.dh;
    const int nch = dh >> 3;       // 16-byte chunks per key row that exist in HBM
    const int q0 = blockIdx.x * (4 * QT * 16) + wave * (QT * 16);

    // zero the padding columns once (chunks nch .. DHP/8 + 1): staging never writes them
    for (int i = tid; i < ATT_KEYS * (T::STRIDE / 16 - nch); i += 256) {
        const int key = i / (T::STRIDE / 16 - nch), c = nch + i % (T::STRIDE / 16 - nch);
#pragma unroll
        for (int buf = 0; buf < 2; ++buf) {
            *reinterpret_cast<f32x4*>(sK + buf * 2 * T::TILE_BYTES + key * T::STRIDE + c * 16) = (f32x4){0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4*>(sV + buf * 2 * T::TILE_BYTES + key * T::STRIDE + c * 16) = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (LSUM && c == nch) *reinterpret_cast<half_t*>(sV + buf * 2 * T::TILE_BYTES + key * T::STRIDE + c * 16) = (half_t)1.0f;
        }
    }

    // ---- Q fragments (B operand): lane holds Q[q = li][dims 32 ks + 8 g .. +7]
    half8_t qf[QT][T::KS];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        int q = q0 + qt * 16 + li;
        q = q < p.Lq ? q : p.Lq - 1;
        const half_t* qrow = p.q + ((size_t)qb * p.Lq + q) * p.ldq + head * dh;
#pragma unroll
        for (int ks = 0; ks < T::KS; ++ks) {
            const int d = ks * 32 + g * 8;
            if (d < dh) qf[qt][ks] = *reinterpret_cast<const half8_t*>(qrow + d);
            else qf[qt][ks] = (half8_t){0, 0, 0, 0, 0, 0, 0, 0};
        }
    }

    const half_t* kbase = p.k + (SC ? (size_t)0 : (size_t)kvb * p.Lk * p.ldk) + head * dh;
    const half_t* vbase = p.v + (SC ? (size_t)0 : (size_t)kvb * p.Lk * p.ldv) + head * dh;
    // sparse-causal: first token row of the two key segments of this batch entry
    const int sc_D = p.Lk >> 1;
    const int sc_f = SC ? qb % p.sc_frames : 0;
    const int sc_row0 = (qb - sc_f) * sc_D;
    const int sc_row1 = (qb - (sc_f > 0 ? 1 : 0)) * sc_D - sc_D;      // row of key j >= D is sc_row1 + j
    // staging plan of this thread, fixed for the whole loop (no per-tile index arithmetic): piece i covers
    // 16 bytes of key `pkey[i]` of the tile; a negative key marks an unused slot
    const int npiece = ATT_KEYS * nch;
    half8_t rk[T::MAXPIECE], rv[T::MAXPIECE];
    int pkey[T::MAXPIECE], lofs[T::MAXPIECE];
    size_t kofs_g[T::MAXPIECE], vofs_g[T::MAXPIECE];
#pragma unroll
    for (int i = 0; i < T::MAXPIECE; ++i) {
        const int pc = tid + i * 256;
        const int key = pc / nch, c = pc - key * nch;
        pkey[i] = pc < npiece ? key : -(1 << 30);
        lofs[i] = key * T::STRIDE + c * 16;
        kofs_g[i] = SC ? (size_t)(c * 8) : (size_t)key * p.ldk + c * 8;
        vofs_g[i] = SC ? (size_t)(c * 8) : (size_t)key * p.ldv + c * 8;
    }

    auto load_tile = [&](int key0) {
        const half_t* kt0 = kbase + (size_t)key0 * p.ldk;
        const half_t* vt0 = vbase + (size_t)key0 * p.ldv;
#pragma unroll
        for (int i = 0; i < T::MAXPIECE; ++i) {
            if (pkey[i] >= 0) {
                if (key0 + pkey[i] < p.Lk) {
                    if constexpr (SC) {
                        const int j = key0 + pkey[i];
                        const size_t row = (size_t)(j < sc_D ? sc_row0 + j : sc_row1 + j);
                        rk[i] = *reinterpret_cast<const half8_t*>(kbase + row * p.ldk + kofs_g[i]);
                        rv[i] = *reinterpret_cast<const half8_t*>(vbase + row * p.ldv + vofs_g[i]);
                    } else {
                        rk[i] = *reinterpret_cast<const half8_t*>(kt0 + kofs_g[i]);
                        rv[i] = *reinterpret_cast<const half8_t*>(vt0 + vofs_g[i]);
                    }
                } else {
                    rk[i] = (half8_t){0, 0, 0, 0, 0, 0, 0, 0};
                    rv[i] = (half8_t){0, 0, 0, 0, 0, 0, 0, 0};
                }
            }
        }
    };
    auto store_tile = [&](int buf) {
        char* dK = sK + buf * (2 * T::TILE_BYTES);
        char* dV = sV + buf * (2 * T::TILE_BYTES);
#pragma unroll
        for (int i = 0; i < T::MAXPIECE; ++i) {
            if (pkey[i] >= 0) {
                *reinterpret_cast<half8_t*>(dK + lofs[i]) = rk[i];
                *reinterpret_cast<half8_t*>(dV + lofs[i]) = rv[i];
            }
        }
    };

    f32x4 o[T::DT][QT];
#pragma unroll
    for (int dt = 0; dt < T::DT; ++dt)
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) o[dt][qt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m_run[QT], l_run[QT];      // running max (log2 units, scaled) and per-lane partial row sums
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) { m_run[qt] = -INFINITY; l_run[qt] = 0.f; }

    const float sl2 = p.scale * 1.4426950408889634f;   // softmax scale folded with log2(e): p = exp2(s*sl2 - m)
    const int ntile = cdiv(p.Lk, ATT_KEYS);
    const int ndt = NDT ? NDT : (dh + 15) >> 4;

    // tile 0 -> LDS buffer 0, tile 1 -> registers
    load_tile(0);
    store_tile(0);
    if (ntile > 1) load_tile(ATT_KEYS);
    __syncthreads();

    for (int t = 0; t < ntile; ++t) {
        const char* cK = sK + (t & 1) * (2 * T::TILE_BYTES);
        const char* cV = sV + (t & 1) * (2 * T::TILE_BYTES);

        // ---- S^T[key, q] = K Q^T
        f32x4 s[4][QT];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) s[kt][qt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < T::KS; ++ks) {
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                const half8_t kf = *reinterpret_cast<const half8_t*>(cK + (kt * 16 + li) * T::STRIDE + (ks * 4 + g) * 16);
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) {
                    if constexpr (ABL >= 2) asm volatile("" ::"v"(kf));
                    else s[kt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[qt][ks], s[kt][qt], 0, 0, 0);
                }
            }
        }

        // ---- keys past Lk exist only in the last tile (wave-uniform branch)
        const int kleft = p.Lk - t * ATT_KEYS;
        if (kleft < ATT_KEYS) {
#pragma unroll
            for (int qt = 0; qt < QT; ++qt)
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (kt * 16 + g * 4 + r >= kleft) s[kt][qt][r] = -INFINITY;
        }

        half8_t pb[2][QT];
        if constexpr (ABL == 1 || ABL == 3) {
#pragma unroll
            for (int qt = 0; qt < QT; ++qt)
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) pb[kt >> 1][qt][(kt & 1) * 4 + r] = (half_t)s[kt][qt][r];
        } else {
        // ---- online softmax per query column (lane li of each 16-lane group), deferred rescale (T13):
        // the running max only moves when some row grew by more than 2^RESCALE_THR, so p <= 2^THR (fine in fp16,
        // sums and O stay in fp32) and the O-wide rescale runs on a handful of tiles instead of every tile.
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            float mx = fmaxf(fmaxf(s[0][qt][0], s[0][qt][1]), fmaxf(s[0][qt][2], s[0][qt][3]));
#pragma unroll
            for (int kt = 1; kt < 4; ++kt)
                mx = fmaxf(mx, fmaxf(fmaxf(s[kt][qt][0], s[kt][qt][1]), fmaxf(s[kt][qt][2], s[kt][qt][3])));
            {   // max over the four 16-lane groups holding the same query: two half-swaps instead of ds_bpermute
                const unsigned u = __float_as_uint(mx);
                const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
                mx = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
                const unsigned v = __float_as_uint(mx);
                const auto b = __builtin_amdgcn_permlane32_swap(v, v, false, false);
                mx = fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
            }
            const float mxs = mx * sl2;
            if (__builtin_amdgcn_ballot_w64(mxs > m_run[qt] + RESCALE_THR) != 0) {
                const float m_new = fmaxf(m_run[qt], mxs);
                const float alpha = __builtin_amdgcn_exp2f(m_run[qt] - m_new);   // first tile: exp2(-inf) = 0
                m_run[qt] = m_new;
                if constexpr (!LSUM) l_run[qt] *= alpha;
#pragma unroll
                for (int dt = 0; dt < T::DT; ++dt) o[dt][qt] *= alpha;
            }
            const float nm = -m_run[qt];
            float psum = 0.f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                float e[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    e[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt][qt][r], sl2, nm));
                    if constexpr (!LSUM) psum += e[r];
                }
                typedef float f32x2 __attribute__((ext_vector_type(2)));
                const half2_t h0 = __builtin_convertvector((f32x2){e[0], e[1]}, half2_t);
                const half2_t h1 = __builtin_convertvector((f32x2){e[2], e[3]}, half2_t);
                pb[kt >> 1][qt][(kt & 1) * 4 + 0] = h0[0];
                pb[kt >> 1][qt][(kt & 1) * 4 + 1] = h0[1];
                pb[kt >> 1][qt][(kt & 1) * 4 + 2] = h1[0];
                pb[kt >> 1][qt][(kt & 1) * 4 + 3] = h1[1];
            }
            if constexpr (!LSUM) l_run[qt] += psum;           // per-lane partial; reduced over g at the end
        }

        }
        // ---- O^T[dim, q] += V^T P^T  (V^T fragments by hardware-transposed LDS reads)
#pragma unroll
        for (int kt2 = 0; kt2 < 2; ++kt2) {
#pragma unroll
            for (int dt = 0; dt < T::DT; ++dt) {
                if (NDT ? dt < NDT : dt < ndt) {
                    const char* va = cV + (kt2 * 32 + g * 4 + (li >> 2)) * T::STRIDE + (dt * 16 + (li & 3) * 4) * 2;
                    const fp16x4_raw lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_raw*)(va));
                    const fp16x4_raw hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_raw*)(va + 16 * T::STRIDE));
                    half8_t vf;
                    __builtin_memcpy(&vf, &lo, 8);
                    __builtin_memcpy(reinterpret_cast<char*>(&vf) + 8, &hi, 8);
#pragma unroll
                    for (int qt = 0; qt < QT; ++qt) {
                        if constexpr (ABL >= 2) asm volatile("" ::"v"(vf), "v"(pb[kt2][qt]));
                        else o[dt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pb[kt2][qt], o[dt][qt], 0, 0, 0);
                    }
                }
            }
        }

        // ---- tile t+1 (in registers since the previous iteration) -> the other LDS buffer, which every wave
        // finished reading before the barrier that ended iteration t-1; then start fetching tile t+2
        if (t + 1 < ntile) store_tile((t + 1) & 1);
        if (t + 2 < ntile) load_tile((t + 2) * ATT_KEYS);
        __syncthreads();
    }

    // ---- normalise and store: lane holds dims dt*16 + 4g .. +3 of query li
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        float l = l_run[qt];
        if constexpr (LSUM) {
            // the row sum of query li sits in output dimension dh: tile dh / 16, lane group (dh % 16) / 4, register dh % 4
            float mine = 0.f;
#pragma unroll
            for (int dt = 0; dt < T::DT; ++dt)
#pragma unroll
                for (int r = 0; r < 4; ++r) mine = (dt * 16 + g * 4 + r == dh) ? o[dt][qt][r] : mine;
            l = __shfl(mine, ((dh & 15) >> 2) * 16 + li, 64);
        } else {
            l += __shfl_xor(l, 16, 64);
            l += __shfl_xor(l, 32, 64);
        }
        const float inv = 1.0f / l;
        const int q = q0 + qt * 16 + li;
        if (q < p.Lq) {
            half_t* orow = p.o + ((size_t)qb * p.Lq + q) * p.ldo + head * dh;
#pragma unroll
            for (int dt = 0; dt < T::DT; ++dt) {
                const int d = dt * 16 + g * 4;
                if (d < dh) {
                    const f32x4 v = o[dt][qt];
                    half4_t h = {(half_t)(v[0] * inv), (half_t)(v[1] * inv), (half_t)(v[2] * inv), (half_t)(v[3] * inv)};
                    *reinterpret_cast<half4_t*>(orow + d) = h;
                }
            }
        }
    }
}

typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));

// hardware-transposed 8-byte LDS read at a compile-time offset, untracked by the compiler (see attention_dma_kernel)
template <int OFF>
__device__ __forceinline__ u32x2_t lds_read_tr16_asm(unsigned addr) {
    u32x2_t r;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF) : "memory");
    return r;
}
template <int OFF>
__device__ __forceinline__ u32x2_t lds_read_b64_asm(unsigned addr) {
    u32x2_t r;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF) : "memory");
    return r;
}
// the V^T fragments of 32 keys: output tile dt = columns 16 dt .. 16 dt + 15, keys +0..15 (lo) and +16..31 (hi)
template <int BASE, int RS, int N, int... DTs>
__device__ __forceinline__ void att_read_v(unsigned addr, u32x2_t (&lo)[N], u32x2_t (&hi)[N], std::integer_sequence<int, DTs...>) {
    ((lo[DTs] = lds_read_tr16_asm<BASE + DTs * 32>(addr), hi[DTs] = lds_read_tr16_asm<BASE + DTs * 32 + 16 * RS>(addr)), ...);
}
// one lgkmcnt(0) that the fragments pass THROUGH, so no MFMA that reads them can be scheduled above it
template <int N, int... DTs>
__device__ __forceinline__ void att_wait_lds(u32x2_t (&lo)[N], u32x2_t (&hi)[N], std::integer_sequence<int, DTs...>) {
    if constexpr (N == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo[0]), "+v"(hi[0]), "+v"(lo[1]), "+v"(hi[1])::"memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo[0]), "+v"(hi[0]), "+v"(lo[1]), "+v"(hi[1]), "+v"(lo[2]), "+v"(hi[2]), "+v"(lo[3]), "+v"(hi[3])::"memory");
    else if constexpr (N == 8) {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo[0]), "+v"(hi[0]), "+v"(lo[1]), "+v"(hi[1]), "+v"(lo[2]), "+v"(hi[2]), "+v"(lo[3]), "+v"(hi[3])::"memory");
        asm volatile("" : "+v"(lo[4]), "+v"(hi[4]), "+v"(lo[5]), "+v"(hi[5]), "+v"(lo[6]), "+v"(hi[6]), "+v"(lo[7]), "+v"(hi[7])::"memory");
    } else if constexpr (N == 3) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo[0]), "+v"(hi[0]), "+v"(lo[1]), "+v"(hi[1]), "+v"(lo[2]), "+v"(hi[2])::"memory");
    else if constexpr (N == 5) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo[0]), "+v"(hi[0]), "+v"(lo[1]), "+v"(hi[1]), "+v"(lo[2]), "+v"(hi[2]), "+v"(lo[3]), "+v"(hi[3]), "+v"(lo[4]), "+v"(hi[4])::"memory");
    else {
        static_assert(N == 10, "head dims 32 / 40 / 64 / 80 / 128 / 160");
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo[0]), "+v"(hi[0]), "+v"(lo[1]), "+v"(hi[1]), "+v"(lo[2]), "+v"(hi[2]), "+v"(lo[3]), "+v"(hi[3]), "+v"(lo[4]), "+v"(hi[4])::"memory");
        asm volatile("" : "+v"(lo[5]), "+v"(hi[5]), "+v"(lo[6]), "+v"(hi[6]), "+v"(lo[7]), "+v"(hi[7]), "+v"(lo[8]), "+v"(hi[8]), "+v"(lo[9]), "+v"(hi[9])::"memory");
    }
}

// ------------------------------------------------------------------------------------------------------------------
// LDS-DMA variant (round 2).  tools/attn_ablate.py put ~290 of the 565 us of L0 self-attention in the register-staged
// load -> ds_write -> barrier chain: each tile's global loads had ONE tile of compute to come back in, and an L2 round
// trip under load is longer than that even with three workgroups per CU.  Here K / V tiles go HBM / L2 -> LDS with
// global_load_lds_dwordx4 (no VGPR round trip, no ds_write), THREE tile buffers deep: tile t+2 is issued while tile t is
// computed, waits are counted (vmcnt(pieces of one tile)), one barrier per tile.
//  * LDS rows hold R 16-byte chunks, R * 16 = an odd multiple of 32 bytes (the conflict-free stride of the register-staged
//    kernel), and a wave instruction writes 64 consecutive chunks, so rows are contiguous: the chunks a row has beyond
//    the head's dh / 8 real ones are fetched from a 32-byte constant page — zeros, or {1, 0, ...} for the V chunk that
//    carries the LSUM column of ones — by pointing those lanes' source address at it (the source address is per lane).
//  * K columns past dh need no zeros: the Q fragment is zero there, and 0 * finite = 0.
//  * every wave issues the same number of pieces per tile (R / 2; R is even), so one counted wait serves all.
#ifndef ATT_DMA_OCC5
#define ATT_DMA_OCC5 4      // waves per SIMD asked for at head dim 40 (128 VGPRs, 4 x 37 KiB of LDS per CU)
#endif
__device__ __attribute__((aligned(16))) half_t g_att_pad_page[16] = {(half_t)1.0f, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

// NCH = dh / 8 = 16-byte chunks a key row really has (the model's head dims: 5, 10, 20); the row holds R >= NCH (+ 1 for the
// LSUM column) chunks with R = 2 (mod 4); the MFMA loops still run over DHP = dh rounded up to 32 (K) / NDT 16-wide tiles (V)
template <int NCH, bool LSUM>
struct AttDmaTile {
    static constexpr int DHP = (NCH * 8 + 31) / 32 * 32;
    static constexpr int NEED = NCH + (LSUM ? 1 : 0);
    static constexpr int R = NEED % 4 == 2 ? NEED : NEED + (6 - NEED % 4) % 4;      // smallest R >= NEED with R % 4 == 2
    static constexpr int RS = R * 16;                                    // row stride: 32 B x odd
    static constexpr int TILE_BYTES = ATT_KEYS * RS;                     // one of K / V = R KiB
    static constexpr int PIECES = R / 2;                                 // 1-KiB pieces per wave and tile (K and V together: 2 R)
    static constexpr int KS = DHP / 32, DT = DHP / 16, NDT = (NCH * 8 + 15) / 16;
    // Q K^T contraction: KS32 k-steps of 32 dims (mfma 16x16x32) and, when at most 16 dims are left (head dim 40: dims 32..39),
    // ONE 16-deep step (mfma 16x16x16, half the matrix-pipe time of a second 32-deep step that would be 75 % padding)
    static constexpr int TAIL16 = (NCH * 8) % 32 != 0 && (NCH * 8) % 32 <= 16 ? 1 : 0;
    static constexpr int KS32 = TAIL16 ? NCH * 8 / 32 : KS;
};

// V2 (round 4): the loop is VALU-issue bound at head dim 40 (per 64-key tile and wave: 34 v_exp, 32 v_fma, 26 v_max, 16 v_cvt_pk
// beside 28 MFMAs, and an MFMA itself holds the vector issue port for 8 of its 16 cycles), so the softmax sheds instructions:
//  * the softmax scale and log2(e) are folded into the Q fragments once per wave (Q' = fp16(Q scale log2 e): one more fp16
//    rounding of Q, ~2^-12 relative per element), so the scores leave the matrix pipe in log2 units;
//  * the running maximum is SUBTRACTED BY THE MATRIX PIPE: the first MFMA of every score tile takes C = {-m, -m, -m, -m} (four
//    registers per query tile, rewritten only when the maximum moves), so p = exp2(S') with no v_fma at all: -32 VALU per tile;
//  * the per-tile maximum is only needed to DECIDE whether the running maximum must move (deferred rescale, threshold 2^8): the
//    16 in-lane values are folded with v_max3 (8 slots), the decision is one compare + ballot over the wave, and the cross-lane
//    maximum (2 permlane swaps + 2 max) moves into the rare branch that takes it: -8 VALU per tile.
// Softmax is invariant to the shift, so results differ from V1 only by the rounding of Q' and of exp2's argument.
// NW (round 4): waves per workgroup.  Every wave issues its share of a key tile's 2 R LDS-DMA pieces, and an LDS-DMA piece costs its
// wave 60 - 185 cycles of issue time (MI355X_MICROARCH.md, cycle constants): with 8 waves = 256 queries per workgroup a wave
// issues half the pieces per tile (and the K / V tiles cross L2 -> LDS half as often); the waves' piece counts may then differ
// by one (2 R = 12 pieces over 8 waves), so the counted wait takes the wave's own count.
template <int NCH, int QT, int NBUF, bool LSUM, bool SC, bool V2 = true, int NW = 4>
__global__ __launch_bounds__(64 * NW, (NCH == 5 && QT == 2) ? ATT_DMA_OCC5 : 1) void attention_dma_kernel(const AttnParams p) {
    using T = AttDmaTile<NCH, LSUM>;
    constexpr int R = T::R, RS = T::RS, NDT = T::NDT;
    constexpr int PIECES = (2 * R + NW - 1) / NW;                      // most pieces one wave issues per tile
    static_assert(R % 4 == 2 && R >= T::NEED && NDT * 32 <= RS, "row stride must be 32 B x odd and hold every output tile's columns");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // buffer b: K at b * 2 * TILE_BYTES, V right behind it

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4;       // 16-lane group
    const int li = lane & 15;
    const int head = blockIdx.y;
    const int qb = blockIdx.z;
    const int kvb = qb / p.kv_batch_div;
    constexpr int dh = NCH * 8;
    constexpr int nch = NCH;       // 16-byte chunks per key row that exist in HBM
    const int q0 = blockIdx.x * (NW * QT * 16) + wave * (QT * 16);
    const int my_pieces = (2 * R - wave + NW - 1) / NW;               // pieces x = wave, wave + NW, ... < 2 R (wave-uniform)

    // ---- Q fragments (B operand): lane holds Q[q = li][dims 32 ks + 8 g .. +7]
    half8_t qf[QT][T::KS32];
    half4_t qtail[QT];               // TAIL16: Q[q = li][dims 32 KS32 + 4 g .. +3] (zero past the head dim)
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        int q = q0 + qt * 16 + li;
        q = q < p.Lq ? q : p.Lq - 1;
        const half_t* qrow = p.q + ((size_t)qb * p.Lq + q) * p.ldq + head * dh;
#pragma unroll
        for (int ks = 0; ks < T::KS32; ++ks) {
            const int d = ks * 32 + g * 8;
            if (d < dh) qf[qt][ks] = *reinterpret_cast<const half8_t*>(qrow + d);
            else qf[qt][ks] = (half8_t){0, 0, 0, 0, 0, 0, 0, 0};
        }
        const int dtl = T::KS32 * 32 + g * 4;
        qtail[qt] = (half4_t){0, 0, 0, 0};
        if (T::TAIL16 && dtl < dh) qtail[qt] = *reinterpret_cast<const half4_t*>(qrow + dtl);
        if constexpr (V2) {       // scores in log2 units straight from the matrix pipe
            const float qs = p.scale * 1.4426950408889634f;
#pragma unroll
            for (int ks = 0; ks < T::KS32; ++ks)
#pragma unroll
                for (int j = 0; j < 8; ++j) qf[qt][ks][j] = (half_t)((float)qf[qt][ks][j] * qs);
#pragma unroll
            for (int j = 0; j < 4; ++j) qtail[qt][j] = (half_t)((float)qtail[qt][j] * qs);
        }
    }

    const half_t* kbase = p.k + (SC ? (size_t)0 : (size_t)kvb * p.Lk * p.ldk) + head * dh;
    const half_t* vbase = p.v + (SC ? (size_t)0 : (size_t)kvb * p.Lk * p.ldv) + head * dh;
    const int sc_D = p.Lk >> 1;
    const int sc_f = SC ? qb % p.sc_frames : 0;
    const int sc_row0 = (qb - sc_f) * sc_D;
    const int sc_row1 = (qb - (sc_f > 0 ? 1 : 0)) * sc_D - sc_D;      // row of key j >= D is sc_row1 + j
    // ---- this wave's pieces: piece x of 2 R (0 .. R-1 = K, R .. 2R-1 = V) for x = wave, wave + 4, ...; lane l of piece x
    // is chunk (x % R) * 64 + l of its operand's tile: key = chunk / R, column chunk = chunk % R
    int pkey[PIECES], pch[PIECES];
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
        const int x = wave + NW * i;
        const int chunk = (x % R) * 64 + lane;
        pkey[i] = chunk / R;
        pch[i] = chunk - pkey[i] * R;
    }
    const half_t* pad_zero = g_att_pad_page + 8;
    const half_t* pad_one = g_att_pad_page;
    auto issue_tile = [&](int key0, int buf) {
        char* bbase = smem + buf * (2 * T::TILE_BYTES);
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            const int x = wave + NW * i;                      // wave-uniform
            if (2 * R % NW != 0 && x >= 2 * R) break;         // (8 waves: the last round of pieces is only half full)
            const bool is_v = x >= R;
            int key = key0 + pkey[i];
            key = key < p.Lk ? key : p.Lk - 1;                // keys past Lk: any finite row (their scores are masked)
            size_t row;
            if constexpr (SC) row = (size_t)(key < sc_D ? sc_row0 + key : sc_row1 + key);
            else row = (size_t)key;
            const half_t* src = is_v ? vbase + row * p.ldv + pch[i] * 8 : kbase + row * p.ldk + pch[i] * 8;
            if (pch[i] >= nch) src = (is_v && LSUM && pch[i] == nch) ? pad_one : pad_zero;
            __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(bbase + x * 1024), 16, 0, 0);
        }
    };

    f32x4 o[T::DT][QT];
#pragma unroll
    for (int dt = 0; dt < T::DT; ++dt)
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) o[dt][qt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m_run[QT], l_run[QT];      // running max (log2 units, scaled) and per-lane partial row sums
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) { m_run[qt] = -INFINITY; l_run[qt] = 0.f; }
    f32x4 negm[QT];                  // V2: the accumulator initialiser of the score tiles, {-m, -m, -m, -m} (0 before the first tile)
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) negm[qt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const float sl2 = p.scale * 1.4426950408889634f;   // softmax scale folded with log2(e): p = exp2(s*sl2 - m)
    const int ntile = cdiv(p.Lk, ATT_KEYS);

    // the Q loads are consumed HERE, before any LDS-DMA is in flight: left to the compiler, the wait for them lands in
    // front of the loop's first MFMA as vmcnt(0) and drains the K / V stream every tile
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int ks = 0; ks < T::KS32; ++ks) asm volatile("" : "+v"(qf[qt][ks]));
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) asm volatile("" : "+v"(qtail[qt]));
    issue_tile(0, 0);
    if (ntile > 1) issue_tile(ATT_KEYS, 1 % NBUF);

    int buf = 0;
    for (int t = 0; t < ntile; ++t) {
        // tile t has landed: all but the pieces of tile t+1 (when it exists) are done; then everybody's pieces
        if (t + 1 >= ntile) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (2 * R % NW == 0 || my_pieces == PIECES) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES - 1) : "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // every wave has finished tile t-1: its buffer takes tile t+2 (NBUF = 3), i.e. two tiles of compute ahead
        if (NBUF >= 3 && t + 2 < ntile) issue_tile((t + 2) * ATT_KEYS, (buf + 2) % NBUF);
        const char* cK = smem + buf * (2 * T::TILE_BYTES);
        const char* cV = cK + T::TILE_BYTES;

        // ---- S^T[key, q] = K Q^T
        f32x4 s[4][QT];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) s[kt][qt] = V2 ? negm[qt] : (f32x4){0.f, 0.f, 0.f, 0.f};     // V2: S' = S - m, for free
#pragma unroll
        for (int ks = 0; ks < T::KS32; ++ks) {
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                const half8_t kf = *reinterpret_cast<const half8_t*>(cK + (kt * 16 + li) * RS + (ks * 4 + g) * 16);
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) s[kt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[qt][ks], s[kt][qt], 0, 0, 0);
            }
        }
        if constexpr (T::TAIL16 != 0) {
            // K[key][dims 32 KS32 + 4 g .. +3]: 8-byte reads, inline asm (a compiler-issued ds_read_b64 would drain the DMA stream)
            const unsigned kaddr = (unsigned)(size_t)LDS_PTR(cK + li * RS + T::KS32 * 64 + g * 8);
#pragma unroll
            for (int half = 0; half < 2; ++half) {      // two key blocks at a time: four fragments in flight cost 8 VGPRs = spills
                u32x2_t ka = half == 0 ? lds_read_b64_asm<0>(kaddr) : lds_read_b64_asm<32 * RS>(kaddr);
                u32x2_t kb = half == 0 ? lds_read_b64_asm<16 * RS>(kaddr) : lds_read_b64_asm<48 * RS>(kaddr);
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ka), "+v"(kb)::"memory");
                half4_t kfa, kfb;
                __builtin_memcpy(&kfa, &ka, 8);
                __builtin_memcpy(&kfb, &kb, 8);
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) {
                    s[2 * half][qt] = __builtin_amdgcn_mfma_f32_16x16x16f16(kfa, qtail[qt], s[2 * half][qt], 0, 0, 0);
                    s[2 * half + 1][qt] = __builtin_amdgcn_mfma_f32_16x16x16f16(kfb, qtail[qt], s[2 * half + 1][qt], 0, 0, 0);
                }
            }
        }

        // ---- keys past Lk exist only in the last tile (wave-uniform branch)
        const int kleft = p.Lk - t * ATT_KEYS;
        if (kleft < ATT_KEYS) {
#pragma unroll
            for (int qt = 0; qt < QT; ++qt)
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (kt * 16 + g * 4 + r >= kleft) s[kt][qt][r] = -INFINITY;
        }

        half8_t pb[2][QT];
        if constexpr (V2) {
            // s = (scores - m_run) in log2 units.  In-lane maximum of the lane's 16 keys per query tile (8 v_max3 each); whether ANY
            // row of the wave outgrew the threshold is one compare + ballot for both query tiles; the first tile always takes
            // the branch (m_run = -inf there)
            float mxl[QT];
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                float mx = fmaxf(fmaxf(s[0][qt][0], s[0][qt][1]), s[0][qt][2]);
                mx = fmaxf(fmaxf(mx, s[0][qt][3]), s[1][qt][0]);
                mx = fmaxf(fmaxf(mx, s[1][qt][1]), s[1][qt][2]);
                mx = fmaxf(fmaxf(mx, s[1][qt][3]), s[2][qt][0]);
                mx = fmaxf(fmaxf(mx, s[2][qt][1]), s[2][qt][2]);
                mx = fmaxf(fmaxf(mx, s[2][qt][3]), s[3][qt][0]);
                mx = fmaxf(fmaxf(mx, s[3][qt][1]), s[3][qt][2]);
                mxl[qt] = fmaxf(mx, s[3][qt][3]);
            }
            float mxa = mxl[0];
#pragma unroll
            for (int qt = 1; qt < QT; ++qt) mxa = fmaxf(mxa, mxl[qt]);
            if (t == 0 || __builtin_amdgcn_ballot_w64(mxa > RESCALE_THR) != 0) {
                // rare: fold the four lane groups (the query's 64 keys), move the running maximum of the rows that grew, rescale
                // O (and l), shift this tile's scores and the accumulator initialiser
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) {
                    float mx = mxl[qt];
                    const unsigned u = __float_as_uint(mx);
                    const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
                    mx = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
                    const unsigned v = __float_as_uint(mx);
                    const auto b = __builtin_amdgcn_permlane32_swap(v, v, false, false);
                    mx = fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
                    // first tile: the maximum itself (it may be far below 0: fp16 P must not underflow); later: only growth
                    // (a fully masked / -inf first tile keeps delta finite: 0)
                    float delta = t == 0 ? mx : fmaxf(mx, 0.f);
                    delta = delta > -1e30f ? delta : 0.f;
                    const float alpha = __builtin_amdgcn_exp2f(-delta);        // first tile: O and l are still 0
                    const float nmr = negm[qt][0] - delta;                      // -(running maximum); negm is 0 before the first tile
                    if constexpr (!LSUM) l_run[qt] *= alpha;
#pragma unroll
                    for (int dt = 0; dt < T::DT; ++dt) o[dt][qt] *= alpha;
#pragma unroll
                    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) s[kt][qt][r] -= delta;
                    negm[qt] = (f32x4){nmr, nmr, nmr, nmr};
                }
            }
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                float psum = 0.f;
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) {
                    float e[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        e[r] = __builtin_amdgcn_exp2f(s[kt][qt][r]);
                        if constexpr (!LSUM) psum += e[r];
                    }
                    typedef float f32x2 __attribute__((ext_vector_type(2)));
                    const half2_t h0 = __builtin_convertvector((f32x2){e[0], e[1]}, half2_t);
                    const half2_t h1 = __builtin_convertvector((f32x2){e[2], e[3]}, half2_t);
                    pb[kt >> 1][qt][(kt & 1) * 4 + 0] = h0[0];
                    pb[kt >> 1][qt][(kt & 1) * 4 + 1] = h0[1];
                    pb[kt >> 1][qt][(kt & 1) * 4 + 2] = h1[0];
                    pb[kt >> 1][qt][(kt & 1) * 4 + 3] = h1[1];
                }
                if constexpr (!LSUM) l_run[qt] += psum;
            }
        }
        // ---- online softmax per query column, deferred rescale (as the register-staged kernel)
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            if constexpr (V2) break;        // handled below, both query tiles under ONE decision

            // a chain of three-input maxima (v_max3_f32: 8 slots for the 16 scores; the pairwise tree took 13)
            float mx = fmaxf(fmaxf(s[0][qt][0], s[0][qt][1]), s[0][qt][2]);
            mx = fmaxf(fmaxf(mx, s[0][qt][3]), s[1][qt][0]);
            mx = fmaxf(fmaxf(mx, s[1][qt][1]), s[1][qt][2]);
            mx = fmaxf(fmaxf(mx, s[1][qt][3]), s[2][qt][0]);
            mx = fmaxf(fmaxf(mx, s[2][qt][1]), s[2][qt][2]);
            mx = fmaxf(fmaxf(mx, s[2][qt][3]), s[3][qt][0]);
            mx = fmaxf(fmaxf(mx, s[3][qt][1]), s[3][qt][2]);
            mx = fmaxf(mx, s[3][qt][3]);
            {
                const unsigned u = __float_as_uint(mx);
                const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
                mx = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
                const unsigned v = __float_as_uint(mx);
                const auto b = __builtin_amdgcn_permlane32_swap(v, v, false, false);
                mx = fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
            }
            const float mxs = mx * sl2;
            if (__builtin_amdgcn_ballot_w64(mxs > m_run[qt] + RESCALE_THR) != 0) {
                const float m_new = fmaxf(m_run[qt], mxs);
                const float alpha = __builtin_amdgcn_exp2f(m_run[qt] - m_new);   // first tile: exp2(-inf) = 0
                m_run[qt] = m_new;
                if constexpr (!LSUM) l_run[qt] *= alpha;
#pragma unroll
                for (int dt = 0; dt < T::DT; ++dt) o[dt][qt] *= alpha;
            }
            const float nm = -m_run[qt];
            float psum = 0.f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                float e[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    e[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt][qt][r], sl2, nm));
                    if constexpr (!LSUM) psum += e[r];
                }
                typedef float f32x2 __attribute__((ext_vector_type(2)));
                const half2_t h0 = __builtin_convertvector((f32x2){e[0], e[1]}, half2_t);
                const half2_t h1 = __builtin_convertvector((f32x2){e[2], e[3]}, half2_t);
                pb[kt >> 1][qt][(kt & 1) * 4 + 0] = h0[0];
                pb[kt >> 1][qt][(kt & 1) * 4 + 1] = h0[1];
                pb[kt >> 1][qt][(kt & 1) * 4 + 2] = h1[0];
                pb[kt >> 1][qt][(kt & 1) * 4 + 3] = h1[1];
            }
            if constexpr (!LSUM) l_run[qt] += psum;           // per-lane partial; reduced over g at the end
        }

        // ---- O^T[dim, q] += V^T P^T  (V^T fragments by hardware-transposed LDS reads)
        // The reads are inline asm with their own lgkmcnt wait: in front of a compiler-issued 8-byte LDS read hipcc drains
        // every LDS-DMA in flight (s_waitcnt vmcnt(0)), i.e. the tiles t+1 and t+2 just asked for — once per key tile.
        const unsigned vaddr = (unsigned)(size_t)LDS_PTR(cV + (g * 4 + (li >> 2)) * RS + (li & 3) * 8);
#pragma unroll
        for (int kt2 = 0; kt2 < 2; ++kt2) {
            u32x2_t lo[NDT], hi[NDT];
            if (kt2 == 0) att_read_v<0, RS>(vaddr, lo, hi, std::make_integer_sequence<int, NDT>{});
            else att_read_v<32 * RS, RS>(vaddr, lo, hi, std::make_integer_sequence<int, NDT>{});
            att_wait_lds(lo, hi, std::make_integer_sequence<int, NDT>{});
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                half8_t vf;
                __builtin_memcpy(&vf, &lo[dt], 8);
                __builtin_memcpy(reinterpret_cast<char*>(&vf) + 8, &hi[dt], 8);
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) o[dt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pb[kt2][qt], o[dt][qt], 0, 0, 0);
            }
        }
        if (NBUF < 3) {       // two buffers (large heads): the buffer of tile t is free once every wave has finished it
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (t + 2 < ntile) issue_tile((t + 2) * ATT_KEYS, buf);
        }
        buf = buf + 1 == NBUF ? 0 : buf + 1;
    }

    // ---- normalise and store: lane holds dims dt*16 + 4g .. +3 of query li
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        float l = l_run[qt];
        if constexpr (LSUM) {
            float mine = 0.f;
#pragma unroll
            for (int dt = 0; dt < T::DT; ++dt)
#pragma unroll
                for (int r = 0; r < 4; ++r) mine = (dt * 16 + g * 4 + r == dh) ? o[dt][qt][r] : mine;
            l = __shfl(mine, ((dh & 15) >> 2) * 16 + li, 64);
        } else {
            l += __shfl_xor(l, 16, 64);
            l += __shfl_xor(l, 32, 64);
        }
        const float inv = 1.0f / l;
        const int q = q0 + qt * 16 + li;
        if (q < p.Lq) {
            half_t* orow = p.o + ((size_t)qb * p.Lq + q) * p.ldo + head * dh;
#pragma unroll
            for (int dt = 0; dt < T::DT; ++dt) {
                const int d = dt * 16 + g * 4;
                if (d < dh) {
                    const f32x4 v = o[dt][qt];
                    half4_t h = {(half_t)(v[0] * inv), (half_t)(v[1] * inv), (half_t)(v[2] * inv), (half_t)(v[3] * inv)};
                    *reinterpret_cast<half4_t*>(orow + d) = h;
                }
            }
        }
    }
}

static int g_force_qt = 0;
template <int NCH, int QT, int NBUF, bool LSUM, bool SC>
static int launch_att_dma(const AttnParams& p, hipStream_t stream) {
    using T = AttDmaTile<NCH, LSUM>;
    constexpr int lds = NBUF * 2 * T::TILE_BYTES + 1024;     // + slack: the last row's MFMA-width reads run past the tile
    // 0x60: A/B switch, the round-2/3 softmax (scale + maximum applied by v_fma, cross-lane maximum every tile)
    // (head dims 128 / 160 keep it: they are MFMA-bound, and the four initialiser registers per query tile would push the
    // 32-row instantiations past 256 VGPRs = from two waves per SIMD to one)
    constexpr bool V2_OK = NCH <= 10;
    // (and the 77-key text cross-attention at head dim 40: two key tiles, the first of which always takes the rescale branch —
    // measured 41 -> 46 us at level 0, tools/ab_attn_v2.py)
    const bool short_keys = NCH == 5 && p.Lk <= 2 * ATT_KEYS;
    if constexpr (NCH == 5 && QT == 2) {
        // level-0 self-attention (2560 keys): 8 waves = 256 queries per workgroup (0x70: A/B switch back to 4 waves)
        if (g_force_qt != 0x60 && g_force_qt != 0x70 && !short_keys && p.Lq % (8 * QT * 16) == 0) {
            auto kern8 = attention_dma_kernel<NCH, QT, NBUF, LSUM, SC, true, 8>;
            if (int rc = ensure_dynamic_lds((const void*)kern8, lds)) return rc;
            hipLaunchKernelGGL(kern8, dim3(p.Lq / (8 * QT * 16), p.heads, p.NBq), dim3(512), lds, stream, p);
            LAVIE_HIP(hipGetLastError());
            return 0;
        }
    }
    auto kern = (g_force_qt == 0x60 || !V2_OK || short_keys) ? attention_dma_kernel<NCH, QT, NBUF, LSUM, SC, false>
                                                : attention_dma_kernel<NCH, QT, NBUF, LSUM, SC, V2_OK>;
    if (int rc = ensure_dynamic_lds((const void*)kern, lds)) return rc;
    dim3 grid(cdiv(p.Lq, 4 * QT * 16), p.heads, p.NBq);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, p);
    LAVIE_HIP(hipGetLastError());
    return 0;
}

template <int DHP, int QT, int ABL = 0, bool LSUM = false, bool SC = false, int NDT = 0>
static int launch_att(const AttnParams& p, hipStream_t stream) {
    using T = AttTile<DHP>;
    auto kern = attention_kernel<DHP, QT, ABL, LSUM, SC, NDT>;
    static bool attr_set = false;
    if (!attr_set) {
        LAVIE_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, T::LDS_BYTES));
        attr_set = true;
    }
    dim3 grid(cdiv(p.Lq, 4 * QT * 16), p.heads, p.NBq);
    hipLaunchKernelGGL(kern, grid, dim3(256), T::LDS_BYTES, stream, p);
    LAVIE_HIP(hipGetLastError());
    return 0;
}

// Kernel choice by head dim: the model's head dims (40, 80, 160; 64 for completeness) get the compile-time output-tile count.
void attention_force_qt(int qt) { g_force_qt = qt; }

template <bool SC>
static int dispatch_att(const AttnParams& p, bool big, bool lsum, hipStream_t stream) {
    if (g_force_qt != 0x50) {      // LDS-DMA staging for the model's head dims (0x50: A/B switch, register-staged kernels)
        if (p.dh == 40 && lsum) return big ? launch_att_dma<5, 2, 3, true, SC>(p, stream) : launch_att_dma<5, 1, 3, true, SC>(p, stream);
        if (p.dh == 80) return big ? launch_att_dma<10, 2, 3, false, SC>(p, stream) : launch_att_dma<10, 1, 3, false, SC>(p, stream);
        if (p.dh == 160) return big ? launch_att_dma<20, 2, 2, false, SC>(p, stream) : launch_att_dma<20, 1, 2, false, SC>(p, stream);
        if constexpr (!SC) {           // the VSR stage's head dims (no sparse-causal attention there)
            if (p.dh == 32) return big ? launch_att_dma<4, 2, 3, false, false>(p, stream) : launch_att_dma<4, 1, 3, false, false>(p, stream);
            if (p.dh == 64) return big ? launch_att_dma<8, 2, 3, false, false>(p, stream) : launch_att_dma<8, 1, 3, false, false>(p, stream);
            if (p.dh == 128) return big ? launch_att_dma<16, 2, 2, false, false>(p, stream) : launch_att_dma<16, 1, 2, false, false>(p, stream);
        }
    }
    if (p.dh <= 64) {
        if (lsum) {
            if (p.dh == 40) return big ? launch_att<64, 2, 0, true, SC, 3>(p, stream) : launch_att<64, 1, 0, true, SC, 3>(p, stream);
            return big ? launch_att<64, 2, 0, true, SC>(p, stream) : launch_att<64, 1, 0, true, SC>(p, stream);
        }
        if (p.dh == 64) return big ? launch_att<64, 2, 0, false, SC, 4>(p, stream) : launch_att<64, 1, 0, false, SC, 4>(p, stream);
        return big ? launch_att<64, 2, 0, false, SC>(p, stream) : launch_att<64, 1, 0, false, SC>(p, stream);
    }
    if (p.dh <= 96) {
        if (lsum) return big ? launch_att<96, 2, 0, true, SC>(p, stream) : launch_att<96, 1, 0, true, SC>(p, stream);
        if (p.dh == 80) return big ? launch_att<96, 2, 0, false, SC, 5>(p, stream) : launch_att<96, 1, 0, false, SC, 5>(p, stream);
        return big ? launch_att<96, 2, 0, false, SC>(p, stream) : launch_att<96, 1, 0, false, SC>(p, stream);
    }
    if (p.dh == 160) return big ? launch_att<160, 2, 0, false, SC, 10>(p, stream) : launch_att<160, 1, 0, false, SC, 10>(p, stream);
    return big ? launch_att<160, 2, 0, false, SC>(p, stream) : launch_att<160, 1, 0, false, SC>(p, stream);
}

int launch_attention(const AttnParams& p, hipStream_t stream) {
    LAVIE_CHECK(p.dh % 8 == 0 && p.dh >= 8 && p.dh <= 160, "attention: head dim %d unsupported (multiple of 8, <= 160)", p.dh);
    LAVIE_CHECK(p.Lq > 0 && p.Lk > 0 && p.NBq > 0 && p.heads > 0 && p.kv_batch_div > 0, "attention: empty problem");
    LAVIE_CHECK(p.ldq % 8 == 0 && p.ldk % 8 == 0 && p.ldv % 8 == 0 && p.ldo % 4 == 0, "attention: row strides must keep 16-B alignment");
    const double tok_q = (double)p.NBq * p.Lq, width = (double)p.heads * p.dh;
    ProfileScope prof(KC_ATTENTION, stream, 4.0 * tok_q * p.Lk * width,
                      2.0 * (2.0 * tok_q * width + 2.0 * ((double)p.NBq / p.kv_batch_div) * p.Lk * width));
    const bool big = p.Lq > 64 * 3;     // >= 2 full 128-row blocks: use 32 rows per wave
    if (p.sc_frames > 0) {
        LAVIE_CHECK(p.Lk == 2 * p.Lq && p.kv_batch_div == 1 && p.NBq % p.sc_frames == 0,
                    "sparse-causal attention: needs Lk = 2 Lq, kv_batch_div = 1, NB %% frames = 0 (Lq=%d Lk=%d NB=%d frames=%d)",
                    p.Lq, p.Lk, p.NBq, p.sc_frames);
        return dispatch_att<true>(p, big, p.dh % 16 != 0, stream);
    }
    if (g_force_qt == 0x12 && p.dh <= 64) return launch_att<64, 2, 1>(p, stream);
    if (g_force_qt == 0x22 && p.dh <= 64) return launch_att<64, 2, 2>(p, stream);
    if (g_force_qt == 0x32 && p.dh <= 64) return launch_att<64, 2, 3>(p, stream);
    if (g_force_qt == 1 && p.dh <= 64) return launch_att<64, 1>(p, stream);
    if (g_force_qt == 4 && p.dh <= 64) return launch_att<64, 4>(p, stream);
    const bool lsum = p.dh % 16 != 0 && g_force_qt != 0x40;     // row sums on the matrix pipe (0x40: A/B switch, VALU sums)
    return dispatch_att<false>(p, big, lsum, stream);
}

}  // namespace lavie
