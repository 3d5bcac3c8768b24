// Shared pieces of the row-resident fused kernels (rowfuse.hip, rowfuse_cross.hip): ring geometry, image builders,
// pipelined LDS-read runs, in-place asm MFMAs.  Internal to lavie_amd/csrc.
#pragma once
#include <hip/hip_ext.h>

#include <type_traits>
#include <utility>
#include <vector>

#include "common.h"
#include "ops.h"
#include "profile.h"

namespace lavie {

namespace rf {
constexpr int THREADS = 512, WAVES = 8;
constexpr int TOK = 16;                          // tokens per wave
constexpr int PASS_ROWS = WAVES * TOK;           // 128 rows per workgroup pass
constexpr int GROUP = 40;                        // pieces (KiB) per ring group = 5 per wave
constexpr int RING_GROUPS = 3;
constexpr int RING_PIECES = GROUP * RING_GROUPS; // 120
constexpr int RING_BYTES = RING_PIECES * 1024;   // 122,880
__host__ __device__ constexpr int swz(int r) { return (4 - (r >> 2)) & 3; }   // g = {0, 3, 2, 1}
}  // namespace rf

// ------------------------------------------------------------------------------------------------ image builders (host)
// Images are described as (destination 8-byte chunk, source 8-byte chunk) pairs and produced by one gather kernel.
static __global__ void rf_gather8_kernel(const uint2* __restrict__ src, uint2* __restrict__ dst, const int2* __restrict__ pairs, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int2 pr = pairs[i];
    dst[pr.x] = pr.y >= 0 ? src[pr.y] : make_uint2(0u, 0u);
}
static __global__ void rf_gather_f16_f32_kernel(const half_t* __restrict__ src, float* __restrict__ dst, const int* __restrict__ idx, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    dst[i] = idx[i] >= 0 ? (float)src[idx[i]] : 0.f;
}

// One A-fragment piece (1 KiB): 16 rows (`rows[r]`) of a row-major [N][ld] matrix at k-step `kbase` (32 k in the register order:
// k-slot 8 q + j <-> column kbase + 16 (j >> 2) + 4 q + (j & 3)).
static void rf_piece_pairs_rows(std::vector<int2>& out, int piece, const int* rows, int ld, int kbase) {
    for (int slot = 0; slot < 64; ++slot) {
        const int r = slot >> 2, q = (slot & 3) ^ rf::swz(r);
        for (int half = 0; half < 2; ++half)
            out.push_back(make_int2(piece * 128 + slot * 2 + half, (rows[r] * ld + kbase + 16 * half + 4 * q) / 4));
    }
}
static void rf_piece_pairs(std::vector<int2>& out, int piece, int n0, int ld, int kbase) {
    int rows[16];
    for (int r = 0; r < 16; ++r) rows[r] = n0 + r;
    rf_piece_pairs_rows(out, piece, rows, ld, kbase);
}
// Uploads the pair lists and runs the gathers; synchronous (load time).
static int rf_run_gathers(const std::vector<int2>* lists, const half_t* const* srcs, int n, half_t* img, hipStream_t stream) {
    for (int i = 0; i < n; ++i) {
        if (lists[i].empty()) continue;
        int2* dp = nullptr;
        LAVIE_HIP(hipMalloc(&dp, lists[i].size() * sizeof(int2)));
        LAVIE_HIP(hipMemcpy(dp, lists[i].data(), lists[i].size() * sizeof(int2), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(rf_gather8_kernel, dim3(cdiv((int)lists[i].size(), 256)), dim3(256), 0, stream, (const uint2*)srcs[i],
                           (uint2*)img, dp, (int)lists[i].size());
        LAVIE_HIP(hipGetLastError());
        LAVIE_HIP(hipStreamSynchronize(stream));
        (void)hipFree(dp);
    }
    return 0;
}

__device__ __forceinline__ void rf_dma(const char* src, char* lds) {
    __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(lds), 16, 0, 0);
}

// A run of N MFMAs whose A fragments are N CONSECUTIVE ring pieces (the stream is laid out in consumption order): reads go
// out PF fragments ahead as inline-asm ds_read_b128 with counted lgkmcnt waits that the fragment passes THROUGH (so the MFMA
// cannot be scheduled above its wait).  Left to itself hipcc waits lgkmcnt(0) in this kernel — every wait then exposes a whole
// LDS round trip (first build: two reads in flight, 345 us for the level-0 feed-forward = slower than the GEMMs it replaces).
// `fn(integral_constant<m>, fragment)` issues MFMA m.  Nothing else of this wave may have LDS reads in flight during a run.
template <int OFF>
__device__ __forceinline__ void rf_lds_read(half8_t& dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void rf_lds_read_f32x4(f32x4& dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}
template <int CNT>
__device__ __forceinline__ void rf_lds_wait(half8_t& v) {
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(v) : "n"(CNT) : "memory");
}
template <int N, int PF, int ABL, class Fn, int... Ms>
__device__ __forceinline__ void rf_run_impl(unsigned addr, Fn&& fn, std::integer_sequence<int, Ms...>) {
    static_assert(PF >= 1 && PF <= 15 && N * 1024 <= 65536, "read-ahead depth / immediate offset range");
    half8_t fa[PF];
    auto prologue = [&](auto m_) {
        constexpr int M = decltype(m_)::value;
        if constexpr (M < PF && M < N) rf_lds_read<M * 1024>(fa[M], addr);
    };
    (prologue(std::integral_constant<int, Ms>{}), ...);
    auto step = [&](auto m_) {
        constexpr int M = decltype(m_)::value;
        constexpr int LEFT = (N - M < PF ? N - M : PF) - 1;      // reads younger than fragment M still allowed in flight
        rf_lds_wait<LEFT>(fa[M % PF]);
        fn(m_, fa[M % PF]);
        // ABL 1 (timing-only build, wrong results): every second fragment read is dropped (the stale fragment is reused)
        if constexpr (M + PF < N && !(ABL == 1 && ((M + PF) & 1))) rf_lds_read<(M + PF) * 1024>(fa[M % PF], addr);
    };
    (step(std::integral_constant<int, Ms>{}), ...);
}
template <int N, int PF, int ABL = 0, class Fn>
__device__ __forceinline__ void rf_run(unsigned addr, Fn&& fn) {
    rf_run_impl<N, PF, ABL>(addr, fn, std::make_integer_sequence<int, N>{});
}

// ABL: timing-only ablation builds (results wrong): 1 = half the LDS fragment reads, 2 = no GELU arithmetic, 3 = no LDS-DMA
// after the first two groups, 4 = no barriers inside the pass
// ABL 5: stamp build (s_memtime around the phases of the chunk loop; sums per wave of workgroup 0 go to p.stamps; read the SHARES)
__device__ __forceinline__ unsigned long long rf_stamp() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}

namespace tb { constexpr int SEG = 20; }     // pieces per ring segment: the granularity of the sync points inside a pass

// A pipelined run over pass-relative stream pieces [S0, S0 + N): like rf_run, with the ring address of every piece a
// compile-time constant ((s mod 120) KiB, two base registers) and the segment syncs issued on the way: sync k stands in front
// of the first MFMA that consumes segment k (pieces 20 k ..), while the reads run up to PF pieces ahead of it — data is
// guaranteed landed one segment ahead (see sync()).  Sync 0 belongs to the caller (nothing may be read before it).
template <int S, int OFF0>
__device__ __forceinline__ void tb_read(half8_t& dst, unsigned lo, unsigned hi) {
    constexpr int RP = S % rf::RING_PIECES;
    if constexpr (RP < 60) rf_lds_read<RP * 1024>(dst, lo);
    else rf_lds_read<(RP - 60) * 1024>(dst, hi);
}
template <int S0, int N, int PF, class Fn, class SyncFn, int... Ms>
__device__ __forceinline__ void tb_run_impl(unsigned lo, unsigned hi, Fn&& fn, SyncFn&& syncfn, std::integer_sequence<int, Ms...>) {
    static_assert(PF >= 1 && PF <= 15 && PF <= tb::SEG, "read-ahead depth");
    half8_t fa[PF];
    auto prologue = [&](auto m_) {
        constexpr int M = decltype(m_)::value;
        if constexpr (M < PF && M < N) tb_read<S0 + M, 0>(fa[M], lo, hi);
    };
    (prologue(std::integral_constant<int, Ms>{}), ...);
    auto step = [&](auto m_) {
        constexpr int M = decltype(m_)::value;
        constexpr int SP = S0 + M;
        if constexpr (SP % tb::SEG == 0 && SP != 0) syncfn(std::integral_constant<int, SP / tb::SEG>{});
        constexpr int LEFT = (N - M < PF ? N - M : PF) - 1;
        rf_lds_wait<LEFT>(fa[M % PF]);
        fn(m_, fa[M % PF]);
        if constexpr (M + PF < N) tb_read<S0 + M + PF, 0>(fa[M % PF], lo, hi);
    };
    (step(std::integral_constant<int, Ms>{}), ...);
}
// PLAIN (development aid): compiler-scheduled reads, one per MFMA, no read-ahead
template <int S0, int N, class Fn, class SyncFn, int... Ms>
__device__ __forceinline__ void tb_run_plain_impl(const char* ringp, Fn&& fn, SyncFn&& syncfn, std::integer_sequence<int, Ms...>) {
    auto step = [&](auto m_) {
        constexpr int M = decltype(m_)::value;
        constexpr int SP = S0 + M;
        if constexpr (SP % tb::SEG == 0 && SP != 0) syncfn(std::integral_constant<int, SP / tb::SEG>{});
        const half8_t a = *reinterpret_cast<const half8_t*>(ringp + ((SP % rf::RING_PIECES) << 10));
        fn(m_, a);
    };
    (step(std::integral_constant<int, Ms>{}), ...);
}
template <int S0, int N, int PF, bool PLAIN = false, class Fn, class SyncFn>
__device__ __forceinline__ void tb_run(unsigned lo, unsigned hi, const char* ringp, Fn&& fn, SyncFn&& syncfn) {
    if constexpr (PLAIN) tb_run_plain_impl<S0, N>(ringp, fn, syncfn, std::make_integer_sequence<int, N>{});
    else tb_run_impl<S0, N, PF>(lo, hi, fn, syncfn, std::make_integer_sequence<int, N>{});
}

// MFMAs of the fused temporal kernel are inline asm, accumulating IN PLACE (vDst = SrcC), in program order.  Reason (found the
// hard way): with compiler-issued MFMAs between the asm reads hipcc (ROCm 7.2) reorders and renames them freely, and where a
// 16x16x16 step reads as SrcC the register a 16x16x32 step has just written to a DIFFERENT vDst it leaves only `s_nop 0` between
// them when inline-asm statements sit in between — the dependent MFMA then reads a stale accumulator (deterministically wrong
// output tiles; every index map checked out in a CPU emulation).  In-place chains need no wait states between MFMAs of one shape;
// a 16-deep step on an accumulator follows its last 32-deep step at least four MFMAs later; compiler code that reads an
// accumulator sits behind an explicit s_nop (guide §5.7 item 2: the compiler pads nothing around asm).
__device__ __forceinline__ void rf_mfma32(f32x4& acc, const half8_t& a, const half8_t& b) {
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void rf_mfma32_first(f32x4& acc, const half8_t& a, const half8_t& b) {       // acc = A B (C = 0)
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void rf_mfma16(f32x4& acc, const half4_t& a, const half4_t& b) {
    asm volatile("v_mfma_f32_16x16x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void rf_mfma_drain() { asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); }   // 32 wait states

__device__ __forceinline__ half8_t rf_cat(const half4_t& a, const half4_t& b) {
    return (half8_t){a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}
__device__ __forceinline__ half4_t rf_pack(const f32x4& v) { return (half4_t){(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]}; }


int rowfuse_variant();      // lavie_debug_rowfuse_variant

}  // namespace lavie
