// Shared epilogue of the implicit-GEMM kernels (igemm.hip, igemm_pp.hip, igemm_patch.hip).
#pragma once
#include <type_traits>

#include "igemm.h"

namespace lavie {

// Logical tile id (XCD-contiguous, in dispatch order inside an XCD) -> (M tile, N tile).
// N-fastest order makes a round of concurrently running tiles of one XCD (32 CUs) = ONE M tile x 32 N tiles when the GEMM is
// wide: every M tile then streams the whole weight matrix through the fabric again (PMC, round 2: the L2-level GEGLU GEMM
// read 869 MB for 39 MB of operands, 5.6 TB/s of fabric traffic under a 154 us kernel).  Wide GEMMs (more than 4 N tiles)
// therefore run in blocks of 32 ids = mr x nr tiles (8 x 4 or 16 x 2): a round shares nr weight slices and mr activation
// tiles (M tiles beyond the last whole block keep N-fastest order: ids >= m_full * n_tiles map to themselves either way).  Narrow ones keep N-fastest order (a round = 32 / n_tiles M tiles x all N tiles, which shares the A tiles).
#ifndef LAVIE_TILE_BLOCKS
#define LAVIE_TILE_BLOCKS 1
#endif
// `w_halfs` = elements of the whole weight matrix: up to 1 Mi halfs (2 MiB) it stays resident in every XCD's L2 whatever the
// order, and N-fastest — which streams each activation tile exactly once — is the better choice (PMC: the L0 GEGLU GEMM, 1.6 MB
// of weights, read 176 MB in blocks and 67 MB N-fastest).
__device__ __forceinline__ void igemm_tile_of(int id, int m_tiles, int n_tiles, long w_halfs, int* mt, int* nt) {
    const int nr = n_tiles % 4 == 0 ? 4 : 2, mr = 32 / nr;
    const int m_full = (m_tiles / mr) * mr;         // M tiles covered by whole blocks; the rest (< mr rows of tiles) runs N-fastest
    if (!LAVIE_TILE_BLOCKS || n_tiles <= 4 || (n_tiles & 1) || w_halfs <= (1L << 20) || id >= m_full * n_tiles) {
        *mt = id / n_tiles;
        *nt = id - *mt * n_tiles;
        return;
    }
    const int blk = id >> 5, w = id & 31;
    const int nbn = n_tiles / nr;
    const int bm = blk / nbn, bn = blk - bm * nbn;
    *mt = bm * mr + (w & (mr - 1));
    *nt = bn * nr + w / mr;
}

// Lane layout on entry (v_mfma_f32_16x16x32_f16 with the weight tile as the A operand): acc[nt][mt][r] is output
// channel ncol + nt*16 + r of token mrow + mt*16.  `nwave0` = first channel of this wave's tile (GEGLU column maths).
// `rowof(mt)` = output row of this lane in 16-row slice mt (contiguous tiles: mrow + 16 mt; the 2-D tiles of the halo-patch
// conv kernel map slices to image rows themselves).
// sum over the 16 lanes of a DPP row (lanes sharing lane >> 4): quad swaps, then the two mirror steps; every lane ends with the
// total (summation order fixed per lane: results are reproducible run to run)
__device__ __forceinline__ float row16_sum(float v) {
    auto dpp = [](float x, auto ctrl) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xF, 0xF, true));
    };
    v += dpp(v, std::integral_constant<int, 0xB1>{});      // quad_perm [1, 0, 3, 2]
    v += dpp(v, std::integral_constant<int, 0x4E>{});      // quad_perm [2, 3, 0, 1]
    v += dpp(v, std::integral_constant<int, 0x141>{});     // row_half_mirror
    v += dpp(v, std::integral_constant<int, 0x140>{});     // row_mirror
    return v;
}

// `cs_block` = column-statistics block of this wave's rows (p.colstat_out; -1 = none)
template <int MT, int NT, int EPI, class RowFn>
__device__ __forceinline__ void igemm_epilogue_rows(const IgemmParams& p, f32x4 (&acc)[NT][MT], RowFn rowof, int ncol, int nwave0,
                                                    int lane, int split, int cs_block = -1) {
    // ---- folded LayerNorm of the A rows: acc <- rstd_m * (acc - mean_m * s_n); the folded bias comes in as p.bias
    if (p.ln_stats || p.ln_partials) {
        f32x4 sv[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) sv[nt] = *reinterpret_cast<const f32x4*>(p.ln_s + ncol + nt * 16);
        float mean[MT], rstd[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int m = rowof(mt);
            const int mc = m < p.M ? m : p.M - 1;
            if (p.ln_partials) {        // fold the producer's per-wave-tile partials here (rowstat_finalize_kernel's arithmetic and order)
                const float2* st = reinterpret_cast<const float2*>(p.ln_partials) + (size_t)mc * p.ln_slots;
                float sum = 0.f, sq = 0.f;
                for (int j = 0; j < p.ln_slots; ++j) { const float2 v = st[j]; sum += v.x; sq += v.y; }
                mean[mt] = sum * p.ln_inv_len;
                rstd[mt] = rsqrtf(fmaxf(sq * p.ln_inv_len - mean[mt] * mean[mt], 0.f) + p.ln_eps);
            } else {
                mean[mt] = p.ln_stats[(size_t)mc * 2];
                rstd[mt] = p.ln_stats[(size_t)mc * 2 + 1];
            }
        }
        // One v_fma_f32 + one v_mul_f32 per element, written out: left to the compiler this becomes v_pk_fma_f32 / v_pk_mul_f32 with the
        // row's mean broadcast by op_sel from whichever register of a pair it landed in, and with three waves per SIMD (the 128 x 64
        // tile: three workgroups per CU) the LOW half of `v_pk_fma_f32 ... op_sel:[0,1,0]` (scalar taken from the HIGH register of the
        // pair) was measured reading 0 instead of the mean in lanes 48 - 63, a few launches in ten (round 4: the F = 5 chunk of the VSR
        // UNet irreproducible; tools/dbg_bn64.py isolates it: the bad elements are exactly those lanes of exactly those instructions,
        // the value is exactly the one with mean = 0, and two workgroups per CU never show it).  Same arithmetic, same rounding.
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = acc[nt][mt][r];
                    asm("v_fma_f32 %0, -%1, %2, %0\n\tv_mul_f32 %0, %3, %0" : "+v"(v) : "v"(mean[mt]), "v"(sv[nt][r]), "v"(rstd[mt]));
                    acc[nt][mt][r] = v;
                }
    }
    // ---- epilogue: lane holds channels n..n+3 of token m for every (nt, mt) ----
    // Which optional operands exist is decided ONCE (wave-uniform) and the body is instantiated per
    // combination: per-element "if (ptr) load" makes hipcc wait vmcnt(0) after every load (guide §5, trap (c)).
    auto epilogue = [&](auto has_bias, auto has_b2, auto has_res) {
        constexpr bool BIAS = decltype(has_bias)::value, B2 = decltype(has_b2)::value, RES = decltype(has_res)::value;
        const bool rowstat = p.rowstat_out != nullptr;
        f32x4 bv[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            bv[nt] = BIAS ? *reinterpret_cast<const f32x4*>(p.bias + ncol + nt * 16) : (f32x4){0.f, 0.f, 0.f, 0.f};
        if constexpr (EPI == EPI_LINEAR) {
            const bool colstat = p.colstat_out != nullptr && cs_block >= 0;
            f32x4 cs_s[NT], cs_q[NT];          // GroupNorm statistics of the rounded outputs: this lane's 4 channels per column tile
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) { cs_s[nt] = (f32x4){0.f, 0.f, 0.f, 0.f}; cs_q[nt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
            // per 16-row slice: request every optional operand first, then combine and store (measured: batching
            // the whole wave tile's loads up front buys nothing and costs ~90 VGPRs)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int m = rowof(mt);
                const int mc = m < p.M ? m : p.M - 1;              // clamp: loads stay in bounds, stores are predicated
                float rs_sum = 0.f, rs_sq = 0.f;
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
                    if (rowstat) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) { const float f = (float)o[r]; rs_sum += f; rs_sq += f * f; }
                    }
                    if (colstat && m < p.M) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) { const float f = (float)o[r]; cs_s[nt][r] += f; cs_q[nt][r] += f * f; }
                    }
                }
                if (rowstat) {
                    // this wave's 16*NT columns of row m: fold the four 16-lane groups, one pair per (row, wave tile)
                    rs_sum += __shfl_xor(rs_sum, 16, 64); rs_sq += __shfl_xor(rs_sq, 16, 64);
                    rs_sum += __shfl_xor(rs_sum, 32, 64); rs_sq += __shfl_xor(rs_sq, 32, 64);
                    if (m < p.M && (lane >> 4) == 0) {
                        float* dst = p.rowstat_out + ((size_t)m * (p.N / (NT * 16)) + nwave0 / (NT * 16)) * 2;
                        dst[0] = rs_sum;
                        dst[1] = rs_sq;
                    }
                }
            }
            if (colstat) {
                // fold the 16 rows a lane group holds per slice (rows ascending inside a lane, then the fixed DPP tree); lane 0 of
                // a group stores the four sums, lane 1 the four sums of squares: one 16-byte store per column tile (cs_index layout)
                float* dst = p.colstat_out + cs_index((size_t)cs_block, ncol, lane & 1, p.N);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    f32x4 a, b;
#pragma unroll
                    for (int r = 0; r < 4; ++r) { a[r] = row16_sum(cs_s[nt][r]); b[r] = row16_sum(cs_q[nt][r]); }
                    if ((lane & 15) < 2) *reinterpret_cast<f32x4*>(dst + nt * 32) = (lane & 1) ? b : a;
                }
            }
        } else {
            // GEGLU: W rows are stored as 16-row blocks alternating value / gate (see pack_geglu),
            // so tile nt (even) holds h and tile nt+1 the matching gate; output column = n / 2.
            static_assert(EPI != EPI_GEGLU || NT % 2 == 0, "GEGLU needs value/gate tile pairs");
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int m = rowof(mt);
#pragma unroll
                for (int nt = 0; nt < NT; nt += 2) {
                    const f32x4 h = acc[nt][mt] + bv[nt], g = acc[nt + 1][mt] + bv[nt + 1];
                    const int no = (nwave0 + nt * 16) / 2 + (lane >> 4) * 4;
                    const half4_t o = {(half_t)(h[0] * gelu_erf_f(g[0])), (half_t)(h[1] * gelu_erf_f(g[1])),
                                       (half_t)(h[2] * gelu_erf_f(g[2])), (half_t)(h[3] * gelu_erf_f(g[3]))};
                    if (m < p.M) *reinterpret_cast<half4_t*>(p.C + (size_t)m * p.ldc + no) = o;
                }
            }
        }
    };
    if (p.splits > 1) {
        // partial sums of this K range; bias / residual / rounding happen once in splitk_reduce_kernel
        float* slab = p.slab + (size_t)split * p.M * p.N;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int m = rowof(mt);
            if (m < p.M) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    *reinterpret_cast<f32x4*>(slab + (size_t)m * p.N + ncol + nt * 16) = acc[nt][mt];
            }
        }
        return;
    }
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

// contiguous rows: the wave's rows are [mrow - (lane & 15), + 16 MT): column-statistics block = that row range / (16 MT)
template <int MT, int NT, int EPI>
__device__ __forceinline__ void igemm_epilogue(const IgemmParams& p, f32x4 (&acc)[NT][MT], int mrow, int ncol, int nwave0,
                                               int lane, int split) {
    igemm_epilogue_rows<MT, NT, EPI>(p, acc, [mrow](int mt) { return mrow + mt * 16; }, ncol, nwave0, lane, split,
                                     (mrow - (lane & 15)) / (MT * 16));
}

}  // namespace lavie
