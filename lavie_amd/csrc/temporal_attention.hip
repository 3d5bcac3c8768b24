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

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));

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

// ------------------------------------------------------------------------------------------------------------------
// Streaming variant (round 2; clips of <= 16 frames, and with NT = 4 up to 64).  The kernel above loads a tile, waits, computes, stores and exits:
// nothing of one workgroup overlaps, and ~4 small workgroups per CU reached 3.4 TB/s at L0.  Here a PERSISTENT workgroup
// (two per CU) walks (video, pixel) tiles of ONE head group of 320 channels (8 / 4 / 2 heads at dh 40 / 80 / 160: every
// level has the same 640-byte row segments) with the q | k | v rows going HBM -> LDS by global_load_lds_dwordx4 two tiles
// ahead (two tile buffers + a separate output buffer, counted vmcnt), the outputs leaving through an LDS row buffer as whole
// 640-byte rows (exactly three 16-byte stores per wave and tile, inline asm, so the counted waits know their number), and
// no ordinary global load inside the loop (rotary tables and this head group's bias rows live in registers).
// LDS rows are padded to 672 bytes = 32 B x 21 (the conflict-free stride of the kernel above); the two pad chunks of a row
// and the rows of frames >= F are fetched from a 16-byte zero page — the LDS-DMA source address is per lane.
__device__ __attribute__((aligned(16))) half_t g_tmp_zero_page[8] = {0, 0, 0, 0, 0, 0, 0, 0};
__device__ __attribute__((aligned(16))) half_t g_tmp_dump_page[64 * 8];

// Geometry per frame-block count NT (16 NT frame rows per array).  NT = 1: 320-channel head groups, 32 KiB tile buffers, two
// workgroups per CU.  NT = 4 (clips of 17..64 frames, the 61-frame interpolation model): 160-channel head groups (4 / 2 / 1
// heads), 68 KiB tile buffers, one workgroup per CU; the work items of a tile are (head, 16-query block) pairs so the four
// waves of a head share its query blocks (eight waves: 2 / 4 / 8 per head).
template <int NT, int RLv = (NT == 1 ? 320 : 160)>
struct Tdma {
    static constexpr int RL = RLv;                          // channels per row segment: 320 / 160 (base widths), 256 / 128 (VSR widths)
    static constexpr int CPR = RL / 8;                      // real 16-byte chunks per row
    static constexpr int RCH = CPR + 2;                     // chunks per LDS row: 672 / 352 B = 32 B x odd
    static constexpr int RS = RCH * 16;
    static constexpr int FP = 16 * NT;
    static constexpr int ROWS = 3 * FP;                     // q | k | v
    static constexpr int CHUNKS = ROWS * RCH;
    static constexpr int NW = NT == 1 ? 4 : 8;              // waves per workgroup
    static constexpr int PIECES = (CHUNKS + NW * 64 - 1) / (NW * 64);   // per wave (8 / 9); lanes past CHUNKS fetch zeros into the slack
    static constexpr int REAL_PIECES = (CHUNKS + 63) / 64;  // 32 / 66; every wave issues PIECES (uniform counted waits): the
    static constexpr int BUF_BYTES = REAL_PIECES * 1024;    // pieces past the buffer land in one shared 1-KiB dump slot
    static constexpr int OBUF_BYTES = FP * RS;
    static constexpr int DUMP = 2 * BUF_BYTES + OBUF_BYTES;
    static constexpr int LDS_BYTES = DUMP + 1024;           // 77,312 / 158,720
    static constexpr int STORES = (FP * CPR + NW * 64 - 1) / (NW * 64);   // 16-byte store instructions per wave and tile (3 / 3)
    static constexpr int OCC = NT == 1 ? 2 : 1;
    static constexpr int MAXH = NT == 1 ? 2 : 1;            // heads per wave (8 heads on 4 waves / at most 4 heads on 8 waves)
    static constexpr int MAXQ = NT == 1 ? 1 : 2;            // 16-query blocks per wave and head
};
static_assert(Tdma<4>::LDS_BYTES <= 160 * 1024 && Tdma<1>::LDS_BYTES * 2 <= 160 * 1024 && Tdma<4>::CHUNKS % 64 == 0 &&
              Tdma<4, 128>::LDS_BYTES <= 160 * 1024 && Tdma<1, 256>::LDS_BYTES * 2 <= 160 * 1024, "temporal stream tiles do not fit LDS");

template <int N> __device__ __forceinline__ void tdma_vmwait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// PACK = 2 (NT = 1, clips of <= 8 frames: the VSR stage's 8-frame chunks): the 16 rows of a tile are TWO neighbouring pixels x 8
// frames (row = 8 * pixel + frame) instead of one pixel x 16 frames half of which would be zero rows; scores between the two
// pixels are masked, so the probabilities are block diagonal and P V stays one MFMA.
template <int NT, int RLv, int PACK = 1>
__global__ __launch_bounds__((Tdma<NT, RLv>::NW * 64), (Tdma<NT, RLv>::OCC)) void temporal_stream_kernel(const TemporalParams p, const int ngroups, const int tiles_per_group) {
    using T = Tdma<NT, RLv>;
    constexpr int RL = T::RL, CPR = T::CPR, RCH = T::RCH, RS = T::RS, FP = T::FP, CHUNKS = T::CHUNKS, PIECES = T::PIECES,
                  BUF_BYTES = T::BUF_BYTES, STORES = T::STORES, NW = T::NW, MAXH = T::MAXH, MAXQ = T::MAXQ;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* obuf = smem + 2 * BUF_BYTES;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, li = lane & 15;
    const int F = p.F, dh = p.dh;
    static_assert(PACK == 1 || NT == 1, "pixel packing is for single-block clips");
    const int Dt = p.D / PACK;                      // tiles per video
    const int C = p.heads * dh;
    const int HG = RL / dh;
    // work split inside a tile: more heads than waves -> wave w takes heads w, w + NW, ... with every query block;
    // fewer -> NW / HG waves share a head and split its 16-query blocks
    const int wph = HG >= NW ? 1 : NW / HG;         // waves per head
    const int h0 = HG >= NW ? wave : wave / wph, hstep = HG >= NW ? NW : HG;
    const int q0 = HG >= NW ? 0 : wave % wph, qstep = wph;
    // this workgroup: one head group, tiles (b, pixel) t0, t0 + stride, ...
    const int hg = blockIdx.x % ngroups;
    const int t0 = blockIdx.x / ngroups;
    const int tstride = gridDim.x / ngroups;
    const int col0 = hg * RL;
    const int ntile = t0 < tiles_per_group ? (tiles_per_group - t0 + tstride - 1) / tstride : 0;
    if (ntile == 0) return;

    // ---- per-lane constants, loaded before any LDS-DMA is in flight
    const int rpairs = p.rot_dim >> 1;
    float rc[NT][4], rs[NT][4];                     // rotary angle of frame 16 t + li, channel pairs 4g .. 4g+3
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int f = PACK == 2 ? (li & 7) : t * 16 + li, k = 4 * g + j;
            const bool ok = f < F && k < rpairs;
            rc[t][j] = ok ? p.rot_cos[f * rpairs + k] : 1.f;
            rs[t][j] = ok ? p.rot_sin[f * rpairs + k] : 0.f;
        }
    // bias rows of this wave's (head, query block) items: query 16 qt + li, keys 16 kt + 4g .. +3
    f32x4 bias[MAXH][MAXQ][NT];
#pragma unroll
    for (int hi = 0; hi < MAXH; ++hi)
#pragma unroll
        for (int qi = 0; qi < MAXQ; ++qi) {
            const int h = h0 + hi * hstep, qt = q0 + qi * qstep;
            const bool have = h < HG && qt < NT;
            const int qidx = PACK == 2 ? (li & 7) : qt * 16 + li;
            const int qic = qidx < F ? qidx : F - 1;
#pragma unroll
            for (int kt = 0; kt < NT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int kj = PACK == 2 ? ((4 * g + r) & 7) : kt * 16 + 4 * g + r;
                    const int kc = kj < F ? kj : F - 1;
                    bias[hi][qi][kt][r] = have ? p.bias[((size_t)(hg * HG + h) * F + qic) * F + kc] : 0.f;
                }
        }
    // staging plan: piece i of this wave covers chunks (wave + NW i) * 64 + lane of the buffer: row = chunk / RCH (array a =
    // row / FP, frame f = row % FP), column chunk c = chunk % RCH; element offset from the tile's base, or -1 = zero page
    long goff[PIECES];
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
        const int chunk = (wave + NW * i) * 64 + lane;
        const int row = chunk / RCH, c = chunk - row * RCH;
        const int a = row / FP, rr = row - a * FP;
        const int f = PACK == 2 ? (rr & 7) : rr, pxl = PACK == 2 ? (rr >> 3) : 0;
        goff[i] = (chunk < CHUNKS && c < CPR && f < F) ? ((long)f * p.D * p.ld + (long)pxl * p.ld + a * C + c * 8) : -1;
    }
    // the loads above are consumed here, not inside the loop (a tracked load in flight there would cost a vmcnt(0))
#pragma unroll
    for (int hi = 0; hi < MAXH; ++hi)
#pragma unroll
        for (int qi = 0; qi < MAXQ; ++qi)
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) asm volatile("" : "+v"(bias[hi][qi][kt]));
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(rc[t][j]), "+v"(rs[t][j]));

    auto tile_base = [&](int n) -> const half_t* {          // token row of (b, frame 0, pixel) of tile n, this head group
        const int t = t0 + n * tstride;
        const int b = t / Dt, pix = (t - b * Dt) * PACK;
        return p.qkv + ((size_t)b * F * p.D + pix) * p.ld + col0;
    };
    auto issue_tile = [&](int n, int buf) {
        const half_t* base = tile_base(n);
        char* dst = smem + buf * BUF_BYTES;
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            const int x = wave + NW * i;                    // wave-uniform
            const half_t* src = goff[i] >= 0 ? base + goff[i] : g_tmp_zero_page;
            char* d = x < T::REAL_PIECES ? dst + x * 1024 : smem + T::DUMP;
            __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(d), 16, 0, 0);
        }
    };

    const int KS = (dh + 31) >> 5;
    const int DT = (dh + 15) >> 4;
    const float l2e = 1.4426950408889634f;

    issue_tile(0, 0);
    if (ntile > 1) issue_tile(1, 1);
    for (int n = 0; n < ntile; ++n) {
        // tile n has landed: younger than it are the pieces of tile n+1 and the stores of tile n-1
        if (n + 1 < ntile) { if (n > 0) tdma_vmwait<PIECES + STORES>(); else tdma_vmwait<PIECES>(); }
        else { if (n > 0) tdma_vmwait<STORES>(); else tdma_vmwait<0>(); }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const char* sQ = smem + (n & 1) * BUF_BYTES;
        const char* sK = sQ + FP * RS;
        const char* sV = sQ + 2 * FP * RS;

#pragma unroll
        for (int hi = 0; hi < MAXH; ++hi) {
            const int h = h0 + hi * hstep;
            if (h >= HG || q0 >= NT) break;
            const int cb = h * dh * 2;                  // byte offset of this head inside a row
            // rotated K fragments of the first 32 dims: once per head, reused by every query block of this wave
            half8_t kf0[NT];
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) {
                kf0[kt] = (half8_t){0, 0, 0, 0, 0, 0, 0, 0};
                if (g * 8 < dh) {
                    const half8_t kraw = *reinterpret_cast<const half8_t*>(sK + (kt * 16 + li) * RS + cb + g * 16);
                    if (rpairs == 0) { kf0[kt] = kraw; continue; }      // plain temporal attention (interpolation model): no rotary
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        const float ke = (float)kraw[2 * jj], ko = (float)kraw[2 * jj + 1];
                        kf0[kt][2 * jj] = (half_t)(ke * rc[kt][jj] - ko * rs[kt][jj]);
                        kf0[kt][2 * jj + 1] = (half_t)(ko * rc[kt][jj] + ke * rs[kt][jj]);
                    }
                }
            }
#pragma unroll
            for (int qi = 0; qi < MAXQ; ++qi) {
                const int qt = q0 + qi * qstep;
                if (qt >= NT) break;
                float qrc[4], qrs[4];                   // the query block's rotary angles (qt is not a compile-time index)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    qrc[jj] = rc[0][jj];
                    qrs[jj] = rs[0][jj];
#pragma unroll
                    for (int t = 1; t < NT; ++t) {
                        qrc[jj] = qt == t ? rc[t][jj] : qrc[jj];
                        qrs[jj] = qt == t ? rs[t][jj] : qrs[jj];
                    }
                }
                f32x4 s[NT];
#pragma unroll
                for (int kt = 0; kt < NT; ++kt) s[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                for (int ks = 0; ks < KS; ++ks) {
                    const int d = ks * 32 + g * 8;
                    half8_t qf = {0, 0, 0, 0, 0, 0, 0, 0};
                    if (d < dh) {
                        const half8_t qraw = *reinterpret_cast<const half8_t*>(sQ + (qt * 16 + li) * RS + cb + d * 2);
                        if (ks == 0 && rpairs != 0) {
#pragma unroll
                            for (int jj = 0; jj < 4; ++jj) {    // rotary on channel pairs, angles in fp32; q also takes the scale
                                const float qe = (float)qraw[2 * jj] * p.scale, qo = (float)qraw[2 * jj + 1] * p.scale;
                                qf[2 * jj] = (half_t)(qe * qrc[jj] - qo * qrs[jj]);
                                qf[2 * jj + 1] = (half_t)(qo * qrc[jj] + qe * qrs[jj]);
                            }
                        } else {
#pragma unroll
                            for (int jj = 0; jj < 8; ++jj) qf[jj] = (half_t)((float)qraw[jj] * p.scale);
                        }
                    }
#pragma unroll
                    for (int kt = 0; kt < NT; ++kt) {
                        half8_t kf = kf0[kt];
                        if (ks > 0) {
                            kf = (half8_t){0, 0, 0, 0, 0, 0, 0, 0};
                            if (d < dh) kf = *reinterpret_cast<const half8_t*>(sK + (kt * 16 + li) * RS + cb + d * 2);
                        }
                        s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf, s[kt], 0, 0, 0);
                    }
                }
                // + bias, softmax over keys (rows of S^T) per query column li — operations and order of the tile kernel
                float mx = -INFINITY;
#pragma unroll
                for (int kt = 0; kt < NT; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v = -INFINITY;
                        const int kj = kt * 16 + 4 * g + r;
                        const bool valid = PACK == 2 ? ((kj & 7) < F && (kj >> 3) == (li >> 3)) : kj < F;
                        if (valid) v = (s[kt][r] + bias[hi][qi][kt][r]) * l2e;
                        s[kt][r] = v;
                        mx = fmaxf(mx, v);
                    }
                mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                float sum = 0.f;
#pragma unroll
                for (int kt = 0; kt < NT; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { const float e = __builtin_amdgcn_exp2f(s[kt][r] - mx); s[kt][r] = e; sum += e; }   // v_exp_f32: p < 2^-126 flushes to 0
                sum += __shfl_xor(sum, 16, 64);
                sum += __shfl_xor(sum, 32, 64);
                const float inv = 1.0f / sum;
                half4_t pb[NT];
#pragma unroll
                for (int kt = 0; kt < NT; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) pb[kt][r] = (half_t)(s[kt][r] * inv);
                // O^T[dim, query] = V^T P^T, 16 keys per MFMA -> the output row buffer.  The hardware-transposed reads and the LDS
                // write are inline asm: in front of a compiler-issued 8-byte LDS read or any LDS write hipcc drains every
                // LDS-DMA in flight with vmcnt(0).  The NT reads of an output tile are issued together, one wait for all.
                for (int dt = 0; dt < DT; ++dt) {
                    u32x2_t raw[NT];
#pragma unroll
                    for (int kt = 0; kt < NT; ++kt) {
                        const char* va = sV + (kt * 16 + 4 * g + (li >> 2)) * RS + cb + (dt * 16 + (li & 3) * 4) * 2;
                        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(raw[kt]) : "v"((unsigned)(size_t)LDS_PTR(va)) : "memory");
                    }
                    if constexpr (NT == 1) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(raw[0])::"memory");
                    else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(raw[0]), "+v"(raw[1]), "+v"(raw[2]), "+v"(raw[3])::"memory");
                    f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int kt = 0; kt < NT; ++kt) {
                        half4_t vf;
                        __builtin_memcpy(&vf, &raw[kt], 8);
                        o = __builtin_amdgcn_mfma_f32_16x16x16f16(vf, pb[kt], o, 0, 0, 0);
                    }
                    const int d = dt * 16 + 4 * g;
                    if (d < dh) {
                        const half4_t hv = {(half_t)o[0], (half_t)o[1], (half_t)o[2], (half_t)o[3]};
                        u32x2_t hw;
                        __builtin_memcpy(&hw, &hv, 8);
                        const unsigned la = (unsigned)(size_t)LDS_PTR(obuf + (qt * 16 + li) * RS + cb + d * 2);
                        asm volatile("ds_write_b64 %0, %1" ::"v"(la), "v"(hw) : "memory");
                    }
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();                   // every wave has left tile n's buffer; the output rows are complete
        __builtin_amdgcn_sched_barrier(0);
        if (n + 2 < ntile) issue_tile(n + 2, n & 1);
        // ---- whole output rows (640 / 320 bytes): wave w stores chunks STORES * 64 * w .. of the FP x CPR
        {
            const int t = t0 + n * tstride;
            const int b = t / Dt, pix = (t - b * Dt) * PACK;
            half_t* obase = p.o + ((size_t)b * F * p.D + pix) * p.ldo + col0;
#pragma unroll
            for (int i = 0; i < STORES; ++i) {
                const int id = (wave * STORES + i) * 64 + lane;
                const int rr = id / CPR, c = id - rr * CPR;
                const int f = PACK == 2 ? (rr & 7) : rr, pxl = PACK == 2 ? (rr >> 3) : 0;
                const bool act = id < FP * CPR && f < F;
                const u32x4_t v = *reinterpret_cast<const u32x4_t*>(obuf + (act ? rr * RS + c * 16 : 0));
                // lanes without a row write their 16 bytes to a dump page instead: the instruction always issues, so the counted
                // waits above see exactly STORES stores per wave and tile
                half_t* dst = act ? obase + ((size_t)f * p.D + pxl) * p.ldo + c * 8 : g_tmp_dump_page + lane * 8;
                asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");
            }
        }
    }
}

// LDS budget per workgroup for the three staged arrays: smaller tiles = more workgroups per CU in flight
// 0 = automatic: 33 KB for clips of <= 16 frames (measured best of 17/33/65 KB: ~4 workgroups per CU), 70 KB for the
// 64-frame tile (61-frame interpolation clips: 4 heads per row segment instead of 2; measured 33 KB 1.39, 70 KB 1.72,
// 135 KB 1.22 TB/s); any explicit budget = the tile kernel also for <= 16 frames (the A/B switch against the streaming kernel)
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
    // streaming kernel: head groups of 320 / 256 (<= 16 frames) or 160 / 128 (<= 64 frames) channels must tile the width
    // (base widths: head dims 40 / 80 / 160; VSR widths: 32 / 64 / 128)
    const int Cw = p.heads * p.dh;
    const int wide = NT == 1 ? 320 : 160, narrow = NT == 1 ? 256 : 128;
    const int srl = (Cw % wide == 0 && wide % p.dh == 0) ? wide : (Cw % narrow == 0 && narrow % p.dh == 0) ? narrow : 0;
    if (g_temporal_budget == 0 && srl != 0 && srl / p.dh <= 8) {       // (all offsets inside the kernel are 64-bit)
        static bool attr = false;
        if (!attr) {
            LAVIE_HIP(hipFuncSetAttribute((const void*)temporal_stream_kernel<1, 320>, hipFuncAttributeMaxDynamicSharedMemorySize, Tdma<1, 320>::LDS_BYTES));
            LAVIE_HIP(hipFuncSetAttribute((const void*)temporal_stream_kernel<1, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, Tdma<1, 256>::LDS_BYTES));
            LAVIE_HIP(hipFuncSetAttribute((const void*)temporal_stream_kernel<1, 320, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, Tdma<1, 320>::LDS_BYTES));
            LAVIE_HIP(hipFuncSetAttribute((const void*)temporal_stream_kernel<1, 256, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, Tdma<1, 256>::LDS_BYTES));
            LAVIE_HIP(hipFuncSetAttribute((const void*)temporal_stream_kernel<4, 160>, hipFuncAttributeMaxDynamicSharedMemorySize, Tdma<4, 160>::LDS_BYTES));
            LAVIE_HIP(hipFuncSetAttribute((const void*)temporal_stream_kernel<4, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, Tdma<4, 128>::LDS_BYTES));
            attr = true;
        }
        const bool pack2 = NT == 1 && p.F <= 8 && p.D % 2 == 0;      // two pixels x 8 frames per 16-row tile
        const int ngroups = Cw / srl;
        const int tiles_per_group = p.B * (pack2 ? p.D / 2 : p.D);
        int per_group = (NT == 1 ? 512 : 256) / ngroups;     // two / one workgroup per CU
        if (per_group < 1) per_group = 1;
        if (per_group > tiles_per_group) per_group = tiles_per_group;
        const int grid = per_group * ngroups;
        const int lds = NT == 1 ? (srl == 320 ? Tdma<1, 320>::LDS_BYTES : Tdma<1, 256>::LDS_BYTES)
                                : (srl == 160 ? Tdma<4, 160>::LDS_BYTES : Tdma<4, 128>::LDS_BYTES);
        auto kern = pack2 ? (srl == 320 ? temporal_stream_kernel<1, 320, 2> : temporal_stream_kernel<1, 256, 2>)
                    : NT == 1 ? (srl == 320 ? temporal_stream_kernel<1, 320> : temporal_stream_kernel<1, 256>)
                            : (srl == 160 ? temporal_stream_kernel<4, 160> : temporal_stream_kernel<4, 128>);
        const int threads = NT == 1 ? Tdma<1>::NW * 64 : Tdma<4>::NW * 64;
        if (prof.active()) hipExtLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, stream, prof.start(), prof.stop(), 0, p, ngroups, tiles_per_group);
        else hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, stream, p, ngroups, tiles_per_group);
        LAVIE_HIP(hipGetLastError());
        return 0;
    }
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
