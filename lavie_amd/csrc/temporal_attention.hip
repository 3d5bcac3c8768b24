// Temporal self-attention core over the frame axis — the HBM-bound kernel of the path.
// Replaces `TemporalAttention._attention` (attention.py:634-667) AND the two layout transposes
// around it (`rearrange "(b f) d c -> (b d) f c"` + .contiguous(), attention.py:550,555): the kernel
// reads q|k|v in the token order the projections already produce, (b, f, pixel), gathers the F
// frames of a pixel itself, and writes its output back in the same order.
//
//   per (video b, pixel p, head h):   q' = rotary(scale q), k' = rotary(k)          (640, 644-646)
//                                     S = q' k'^T + rel_pos_bias[h]                  (648, 650)
//                                     O = softmax(S - rowmax) v                       (656-665)
//
// Workgroup = 4 waves = (b, PT pixels, HG heads).  Data movement:
//  * (frame, pixel)-tiles of q, k and v are read with fully coalesced 16-B/lane loads — each
//    (f, pixel) row segment is HG*dh*2 contiguous bytes and consecutive pixels are consecutive
//    rows — and staged in LDS with rows padded to 32 B x odd, which makes both the ds_read_b128
//    fragment reads (16 frames x one 16-B slot) and the ds_read_b64_tr_b16 V reads conflict-free;
//  * every (pixel, head) pair is then one wave's job: 1-5 MFMA 16x16x32 for S^T = K Q^T (keys on the
//    accumulator rows so the softmax is lane-local), bias + softmax in registers, and
//    MFMA 16x16x16 for O^T = V^T P^T with P^T taken straight from the accumulator registers;
//  * O overwrites the wave's own Q slots in LDS and the whole tile is stored coalesced.
// Algorithmic HBM bytes per launch: 4 * tokens * C * 2 (q, k, v read + o write).
#include <hip/hip_ext.h>

#include "common.h"
#include "ops.h"
#include "profile.h"

namespace lavie {

struct TemporalGeom {
    int PT, HG;        // pixels and heads per workgroup
    int RL;            // row length in halfs = HG * dh
    int RS;            // LDS row stride in bytes = RL*2 + 32
    int rows;          // LDS rows per array = PT * NT*16
    int lds_bytes;
};

template <int NT>   // frames padded to NT*16
__global__ __launch_bounds__(256) void temporal_attention_kernel(const TemporalParams p, const TemporalGeom gm) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int FP = NT * 16;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int g = lane >> 4, li = lane & 15;
    const int F = p.F, dh = p.dh;
    const int C = p.heads * dh;
    const int PT = gm.PT, HG = gm.HG, RL = gm.RL, RS = gm.RS;
    const int arr_bytes = gm.rows * RS;
    char* sQ = smem;
    char* sK = smem + arr_bytes;
    char* sV = smem + 2 * arr_bytes;

    // block -> (b, pixel tile, head group)
    const int ngroups = p.heads / HG;
    const int hg = blockIdx.x % ngroups;
    const int ptile = blockIdx.x / ngroups;
    const int tiles_per_b = cdiv(p.D, PT);
    const int b = ptile / tiles_per_b;
    const int p0 = (ptile - b * tiles_per_b) * PT;
    const int npix = min(PT, p.D - p0);
    const int col0 = hg * RL;                       // first channel of this head group inside C

    // ---- stage q | k | v : piece = (array a, pixel pp, frame f, 16-B chunk c)
    const int cpr = RL >> 3;                        // chunks per row
    const int total = 3 * PT * FP * cpr;
    // SU pieces per thread and pass: all their 16-B loads are issued before the first LDS store, so a pass costs one
    // memory latency instead of SU (the kernel has no compute to hide loads under; a one-piece loop serialised them).
    constexpr int SU = 8;                         // (16 measured slower: 121 VGPRs cost more occupancy than one latency saves)
    for (int i0 = tid; i0 < total; i0 += 256 * SU) {
        half8_t v[SU];
        int dst[SU];
        bool keep[SU];
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            // branch-free: every lane loads from a clamped (always valid) address; a conditional load would make hipcc wait
            // vmcnt(0) after each one (guide section 5, trap (c))
            const int ir = i0 + u * 256;
            const int i = ir < total ? ir : total - 1;
            const int c = i % cpr;
            int r = i / cpr;
            const int pp = r % PT; r /= PT;             // pixel fastest: consecutive rows in HBM
            const int f = r % FP;
            const int a = r / FP;
            dst[u] = ir < total ? a * arr_bytes + (pp * FP + f) * RS + c * 16 : -1;
            keep[u] = f < F && pp < npix;
            const int fc = f < F ? f : F - 1, pc = pp < npix ? pp : npix - 1;
            const size_t tok = ((size_t)b * F + fc) * p.D + p0 + pc;
            v[u] = *reinterpret_cast<const half8_t*>(p.qkv + tok * p.ld + a * C + col0 + c * 8);
        }
#pragma unroll
        for (int u = 0; u < SU; ++u)
            if (!keep[u]) v[u] = (half8_t){0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < SU; ++u)
            if (dst[u] >= 0) *reinterpret_cast<half8_t*>(smem + dst[u]) = v[u];
    }

    // rotary table entries of this lane: position = frame li (+16 per tile), pairs 4g .. 4g+3
    const int rpairs = p.rot_dim >> 1;
    float rc[NT][4], rs[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int f = t * 16 + li, k = 4 * g + j;
            const bool ok = f < F && k < rpairs;
            rc[t][j] = ok ? p.rot_cos[f * rpairs + k] : 1.f;
            rs[t][j] = ok ? p.rot_sin[f * rpairs + k] : 0.f;
        }
    // relative-position bias of this lane's (query li, keys 4g..4g+3) for the wave's first two pairs: one 16-B load each,
    // issued while the staging loads are still in flight (inside the compute loop it is a dependent L2 round trip per pair)
    const bool bias_vec = NT == 1 && (F & 3) == 0;
    f32x4 bpre0 = {0.f, 0.f, 0.f, 0.f}, bpre1 = {0.f, 0.f, 0.f, 0.f};
    if (bias_vec) {
        const int qic = li < F ? li : F - 1;
        const int kc = 4 * g < F ? 4 * g : F - 4;
        const int h0 = hg * HG + wave % HG, h1 = hg * HG + (wave + 4) % HG;
        bpre0 = *reinterpret_cast<const f32x4*>(p.bias + ((size_t)h0 * F + qic) * F + kc);
        bpre1 = *reinterpret_cast<const f32x4*>(p.bias + ((size_t)h1 * F + qic) * F + kc);
    }
    __syncthreads();

    const int KS = (dh + 31) >> 5;
    const int DT = (dh + 15) >> 4;
    const float l2e = 1.4426950408889634f;

    int pair_no = 0;
    for (int pair = wave; pair < npix * HG; pair += 4, ++pair_no) {
        const int pp = pair / HG, hl = pair - pp * HG;
        const int h = hg * HG + hl;
        const int rowbase = pp * FP;
        const int cb = hl * dh * 2;                 // byte offset of this head inside a row

        // ---- S^T[key, query] tiles
        f32x4 s[NT][NT];                            // [key tile][query tile]
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int qt = 0; qt < NT; ++qt) s[kt][qt] = (f32x4){0.f, 0.f, 0.f, 0.f};

        for (int ks = 0; ks < KS; ++ks) {
            const int d = ks * 32 + g * 8;
            const bool live = d < dh;
            half8_t qf[NT], kf[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                qf[t] = (half8_t){0, 0, 0, 0, 0, 0, 0, 0};
                kf[t] = qf[t];
                if (live) {
                    const int off = (rowbase + t * 16 + li) * RS + cb + d * 2;
                    const half8_t qraw = *reinterpret_cast<const half8_t*>(sQ + off);
                    const half8_t kraw = *reinterpret_cast<const half8_t*>(sK + off);
                    if (ks == 0) {
                        // rotary on channel pairs (2k, 2k+1), angles in fp32; q also takes the softmax scale
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float qe = (float)qraw[2 * j] * p.scale, qo = (float)qraw[2 * j + 1] * p.scale;
                            const float ke = (float)kraw[2 * j], ko = (float)kraw[2 * j + 1];
                            qf[t][2 * j] = (half_t)(qe * rc[t][j] - qo * rs[t][j]);
                            qf[t][2 * j + 1] = (half_t)(qo * rc[t][j] + qe * rs[t][j]);
                            kf[t][2 * j] = (half_t)(ke * rc[t][j] - ko * rs[t][j]);
                            kf[t][2 * j + 1] = (half_t)(ko * rc[t][j] + ke * rs[t][j]);
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < 8; ++j) qf[t][j] = (half_t)((float)qraw[j] * p.scale);
                        kf[t] = kraw;
                    }
                }
            }
#pragma unroll
            for (int kt = 0; kt < NT; ++kt)
#pragma unroll
                for (int qt = 0; qt < NT; ++qt)
                    s[kt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[kt], qf[qt], s[kt][qt], 0, 0, 0);
        }

        // ---- + bias, softmax over keys (rows of S^T), per query column li
        half4_t pb[NT][NT];
#pragma unroll
        for (int qt = 0; qt < NT; ++qt) {
            const int qi = qt * 16 + li;
            const int qic = qi < F ? qi : F - 1;
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) {
                const int kj = kt * 16 + 4 * g;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = -INFINITY;
                    if (kj + r < F) {
                        const float bias_v = (bias_vec && pair_no < 2) ? (pair_no == 0 ? bpre0[r] : bpre1[r])
                                                                     : p.bias[((size_t)h * F + qic) * F + kj + r];
                        v = (s[kt][qt][r] + bias_v) * l2e;
                    }
                    s[kt][qt][r] = v;
                    mx = fmaxf(mx, v);
                }
            }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            float sum = 0.f;
#pragma unroll
            for (int kt = 0; kt < NT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float e = exp2f(s[kt][qt][r] - mx); s[kt][qt][r] = e; sum += e; }
            sum += __shfl_xor(sum, 16, 64);
            sum += __shfl_xor(sum, 32, 64);
            const float inv = 1.0f / sum;
#pragma unroll
            for (int kt = 0; kt < NT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) pb[kt][qt][r] = (half_t)(s[kt][qt][r] * inv);
        }

        // ---- O^T[dim, query] = V^T P^T, 16 keys per MFMA; result replaces this pair's Q slots in LDS
        for (int dt = 0; dt < DT; ++dt) {
            f32x4 o[NT];
#pragma unroll
            for (int qt = 0; qt < NT; ++qt) o[qt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) {
                const char* va = sV + (rowbase + kt * 16 + 4 * g + (li >> 2)) * RS + cb + (dt * 16 + (li & 3) * 4) * 2;
                const fp16x4_raw raw = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_raw*)(va));
                half4_t vf;
                __builtin_memcpy(&vf, &raw, 8);
#pragma unroll
                for (int qt = 0; qt < NT; ++qt)
                    o[qt] = __builtin_amdgcn_mfma_f32_16x16x16f16(vf, pb[kt][qt], o[qt], 0, 0, 0);
            }
            const int d = dt * 16 + 4 * g;
            if (d < dh) {
#pragma unroll
                for (int qt = 0; qt < NT; ++qt) {
                    half4_t hv = {(half_t)o[qt][0], (half_t)o[qt][1], (half_t)o[qt][2], (half_t)o[qt][3]};
                    *reinterpret_cast<half4_t*>(sQ + (rowbase + qt * 16 + li) * RS + cb + d * 2) = hv;
                }
            }
        }
    }
    __syncthreads();

    // ---- coalesced store of the O tile (held in the Q array)
    const int ototal = PT * FP * cpr;
    for (int i = tid; i < ototal; i += 256) {
        const int c = i % cpr;
        int r = i / cpr;
        const int pp = r % PT;
        const int f = r / PT;
        if (f < F && pp < npix) {
            const size_t tok = ((size_t)b * F + f) * p.D + p0 + pp;
            *reinterpret_cast<half8_t*>(p.o + tok * p.ldo + col0 + c * 8) =
                *reinterpret_cast<const half8_t*>(sQ + (pp * FP + f) * RS + c * 16);
        }
    }
}

// LDS budget per workgroup for the three staged arrays: smaller tiles = more workgroups per CU in flight
// 0 = automatic: 33 KB for clips of <= 16 frames (measured best of 17/33/65 KB: ~4 workgroups per CU), 70 KB for the
// 64-frame tile (61-frame interpolation clips: 4 heads per row segment instead of 2; measured 33 KB 1.39, 70 KB 1.72,
// 135 KB 1.22 TB/s)
static int g_temporal_budget = 0;
void temporal_set_budget(int bytes) { g_temporal_budget = bytes; }

int launch_temporal_attention(const TemporalParams& p, hipStream_t stream) {
    LAVIE_CHECK(p.F >= 1 && p.F <= 64, "temporal attention: F=%d unsupported (1..64)", p.F);
    LAVIE_CHECK(p.dh % 8 == 0 && p.dh <= 160 && p.rot_dim <= 32 && p.rot_dim <= p.dh && p.rot_dim % 2 == 0,
                "temporal attention: dh=%d rot_dim=%d unsupported", p.dh, p.rot_dim);
    LAVIE_CHECK(p.ld % 8 == 0 && p.ldo % 8 == 0, "temporal attention: row strides must be multiples of 8 halfs");
    const double tok = (double)p.B * p.F * p.D, width = (double)p.heads * p.dh;
    // algorithmic bytes: q, k, v read + o written once: 4 * tokens * C * 2 B
    ProfileScope prof(KC_TEMPORAL, stream, 4.0 * tok * p.F * width, 4.0 * tok * width * 2.0, /*kernel_events=*/true);
    const int NT = cdiv(p.F, 16) <= 1 ? 1 : 4;
    const int FP = NT * 16;
    // pick (HG, PT): the largest tile whose three LDS arrays stay under the budget.  Rows are padded to 32 B x odd
    // (conflict-free ds_read_b128 over 16 frames and ds_read_b64_tr_b16 over 8 keys, see attention.hip).
    TemporalGeom gm;
    auto row_stride = [&](int hg) {
        int rs = hg * p.dh * 2 + 32;
        if (((rs / 32) & 1) == 0) rs += 32;
        return rs;
    };
    const int want = g_temporal_budget > 0 ? g_temporal_budget : (NT == 1 ? 33000 : 70000);
    const int budget = want > 3 * FP * 256 ? want : 3 * FP * 256;
    gm.HG = p.heads;
    while (gm.HG > 2 && gm.HG % 2 == 0 && 3 * FP * row_stride(gm.HG) > budget) gm.HG /= 2;
    LAVIE_CHECK((gm.HG * p.dh * 2) % 32 == 0, "temporal attention: head-group row (%d heads x %d) must be a multiple of 32 B", gm.HG, p.dh);
    gm.RL = gm.HG * p.dh;
    gm.RS = row_stride(gm.HG);
    gm.PT = budget / (3 * FP * gm.RS);
    if (gm.PT < 1) gm.PT = 1;
    if (gm.PT > 4) gm.PT = 4;
    gm.rows = gm.PT * FP;
    gm.lds_bytes = 3 * gm.rows * gm.RS;
    LAVIE_CHECK(gm.lds_bytes <= 160 * 1024, "temporal attention: tile needs %d B of LDS", gm.lds_bytes);
    const int grid = p.B * cdiv(p.D, gm.PT) * (p.heads / gm.HG);
    if (NT == 1) {
        static bool attr1 = false;
        if (!attr1) {
            LAVIE_HIP(hipFuncSetAttribute((const void*)temporal_attention_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr1 = true;
        }
        if (prof.active()) hipExtLaunchKernelGGL(temporal_attention_kernel<1>, dim3(grid), dim3(256), gm.lds_bytes, stream, prof.start(), prof.stop(), 0, p, gm);
        else hipLaunchKernelGGL(temporal_attention_kernel<1>, dim3(grid), dim3(256), gm.lds_bytes, stream, p, gm);
    } else {
        static bool attr4 = false;
        if (!attr4) {
            LAVIE_HIP(hipFuncSetAttribute((const void*)temporal_attention_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr4 = true;
        }
        if (prof.active()) hipExtLaunchKernelGGL(temporal_attention_kernel<4>, dim3(grid), dim3(256), gm.lds_bytes, stream, prof.start(), prof.stop(), 0, p, gm);
        else hipLaunchKernelGGL(temporal_attention_kernel<4>, dim3(grid), dim3(256), gm.lds_bytes, stream, p, gm);
    }
    LAVIE_HIP(hipGetLastError());
    return 0;
}

}  // namespace lavie
