// Persistent ping-pong GEMM for plain A rows (nn.Linear / 1x1 conv / GEGLU): the 160x320 (or 160x256) two-group kernel of
// igemm_pp.hip with the TILE LOOP FLATTENED INTO THE K LOOP and an epilogue built around what the memory pipe of ONE
// workgroup per CU can take, gfx950.
//
// Why (tools/check_ppx.py, tools/stamps_ppx.py, tools/probes/store_probe.hip; profiles/r02_ppx_*.txt):
//   * the transformer blocks' GEMMs have short K loops (K = C or 4C: 5 - 80 K-tiles) and large M.  With the epilogue
//     removed (ablation 0x17) this loop runs every such shape at 1000 - 1180 TFLOP/s, also at 5 K-tiles; with an epilogue
//     in the usual place the same shapes ran at 300 - 750: half of the time of these GEMMs was their EPILOGUE.
//   * vmcnt retires in order, so every ordinary global load in an epilogue (bias, LayerNorm statistics, five serialised
//     batches of residual rows) first waits for the youngest prefetched A pieces: a full HBM round trip each.
//   * the stores are the larger half: ONE 512-thread workgroup per CU drains stores at only ~19 GB/s (a 100-KiB tile:
//     5.3 - 7.9 us whatever the shape of the store instruction: row-per-lane 8-byte, 16-byte after v_permlane16_swap,
//     whole 128-byte lines, 1 KiB contiguous — store_probe.hip), and a wave that issues into the full queue stalls:
//     a tile's 25 stores per wave issued back to back block its wave — and, at the next barrier, its SIMD partner — for
//     as long as the tile's whole K loop takes at K = 320.
//
// Structure: at most 256 workgroups (one per CU) each walk a list of output tiles; the LDS-DMA stream (W one K-tile
// ahead, A two K-tiles ahead, igemm_pp.hip's schedule) runs across tile boundaries, so only a workgroup's first tile pays
// a prologue.  The epilogue of tile i is DEFERRED into tile i+1:
//   * bias, folded-LayerNorm row sums and the (mean, rstd) rows of the tile arrive in LDS as six extra 1-KiB DMA pieces
//     ("aux", double-buffered by tile parity) issued with the tile's first K-tile — ordinary pieces of the stream;
//   * the residual rows are fetched into the output registers `ro` by inline-asm loads at the START of the tile's last
//     K-tile (before that step's DMA batch) and waited for with an exact counted vmcnt (the pieces issued since);
//   * in the first R phase of tile i+1, behind that phase's DMA batch, the accumulators are finished (fold, + bias,
//     + residual, round to fp16: operations and order of igemm_epilogue.h, results bit-identical to every other GEMM
//     kernel), regrouped with v_permlane16_swap into 16-byte row chunks, parked in `ro` (over the residual they consumed)
//     and zeroed: pure VALU, no memory instruction;
//   * the stores then TRICKLE out of `ro`, two per R phase over the next four K-tiles, each phase's counted waits raised
//     by exactly the stores issued in it: the queue never fills, no wave ever waits for a store, and the write stream
//     overlaps the next tile's MFMAs.
//
// Tile order: tile id = m_tile * n_tiles + n_tile (n fastest).  The ids are cut into 8 contiguous chunks, one per XCD
// label (blockIdx % 8: blocks that share an L2), and workgroup j of an XCD takes its chunk's tiles j, j + 32, j + 64, ...:
// the n_tiles workgroups that read one A row block run on one XCD at about the same time (A is fetched from HBM once).
#include <type_traits>

#include "igemm.h"
#include "igemm_epilogue.h"

namespace lavie {

#ifndef PPX_TRICKLE
#define PPX_TRICKLE 0      // 0: a finished tile's stores are issued at once from its finish pass (fastest measured); n > 0: parked in
                           // the output registers and issued n per R phase (measured slower: a queued store delays the wave's own loads)
#endif
#ifndef PPX_STAGGER
#define PPX_STAGGER 0      // 1: workgroups start a quarter tile period apart (measured: no effect)
#endif

namespace ppx {
constexpr int MT = 5;
constexpr int BM = 160, THREADS = 512;
constexpr int A_BYTES = BM * 128;                       // one A stage: 20,480
constexpr int A_STAGES = 3, W_STAGES = 2;
constexpr int W_BASE = A_STAGES * A_BYTES;
constexpr int AUX_BYTES = 7 * 1024;                     // bias 2 KiB | ln_s 2 KiB | (mean, rstd) rows 2 KiB | scratch 1 KiB
template <int NT>
struct Geo {
    static constexpr int BN = 4 * NT * 16;
    static constexpr int HALF_ROWS = 2 * NT * 16;       // W rows read by one group
    static constexpr int HALF_PIECES = HALF_ROWS / 8;   // 20 or 16: group 0 stages 16 of each half, group 1 the rest
    static constexpr int W_BYTES = BN * 128;
    static constexpr int AUX_BASE = W_BASE + W_STAGES * W_BYTES;
    static constexpr int LDS_BYTES = AUX_BASE + 2 * AUX_BYTES;
    static_assert(LDS_BYTES <= 160 * 1024, "does not fit LDS");
};
constexpr int MAX_WG = 256;                             // one workgroup per CU

// 8-byte global load the compiler does not track (cdna_hip_programming.md §5.7): counted and waited for by hand
template <int IMM>
__device__ __forceinline__ void asm_load_b64(unsigned long long& dst, unsigned voff, const void* sbase) {
    // s_nop 4: the scalar base may have been written by a VALU instruction just before the statement (hipcc spills SGPRs to VGPR
    // lanes and reloads them with v_readlane right in front of their use); a VMEM instruction that reads such an SGPR as its base
    // needs 5 wait states, and hipcc pads nothing inside an asm statement (guide 5.7 item 2).  Round 4: exactly that reload in
    // front of the column-statistics store of igemm_ppx_kernel<0, 4, 1, 0> sent the store to a stale base (memory aperture fault).
    asm volatile("s_nop 4\n\tglobal_load_dwordx2 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(sbase), "n"(IMM) : "memory");
}
// stamp build (ABL 3): per-wave cycle sums of workgroup 0, [wave][32]: segments 0..9 of ordinary K-tile steps, 16..25 of
// the steps that carry an epilogue, [10] K-tile steps, [11] epilogue steps, [12] whole loop.  Segments: 0 DMA batch,
// 1 epilogue, 2 R(g,0) reads + waits, 3 barrier, 4 M(g,0), 5 barrier, 6 R(g,1), 7 barrier, 8 M(g,1) + wait, 9 barrier.
__device__ unsigned long long g_ppx_stamps[8 * 32];

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
template <int IMM>
__device__ __forceinline__ void asm_store_b128(unsigned voff, u32x4 v, const void* sbase) {
    // s_nop 1: a store of more than 8 bytes must not have its data registers overwritten in the next wait state, and
    // hipcc's hazard recognizer does not look inside an asm statement (cdna_hip_programming.md §5.7)
    asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2 offset:%3\n\ts_nop 1" ::"v"(voff), "v"(v), "s"(sbase), "n"(IMM) : "memory");     // s_nop 4: see asm_load_b64
}
template <int IMM>
__device__ __forceinline__ void asm_store_b64(unsigned voff, half4_t v, const void* sbase) {
    asm volatile("s_nop 4\n\tglobal_store_dwordx2 %0, %1, %2 offset:%3" ::"v"(voff), "v"(v), "s"(sbase), "n"(IMM) : "memory");
}
}  // namespace ppx

// counted wait with a run-time count (wave-uniform): all but the wave's `n` youngest vector-memory operations are done
__device__ __forceinline__ void vmwait(int n) {
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
        case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
        case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
        case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
        case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
        case 17: asm volatile("s_waitcnt vmcnt(17)" ::: "memory"); break;
        case 18: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
        case 19: asm volatile("s_waitcnt vmcnt(19)" ::: "memory"); break;
        case 20: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
        case 21: asm volatile("s_waitcnt vmcnt(21)" ::: "memory"); break;
        case 22: asm volatile("s_waitcnt vmcnt(22)" ::: "memory"); break;
        case 23: asm volatile("s_waitcnt vmcnt(23)" ::: "memory"); break;
        case 24: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
        case 25: asm volatile("s_waitcnt vmcnt(25)" ::: "memory"); break;
        case 26: asm volatile("s_waitcnt vmcnt(26)" ::: "memory"); break;
        case 27: asm volatile("s_waitcnt vmcnt(27)" ::: "memory"); break;
        case 28: asm volatile("s_waitcnt vmcnt(28)" ::: "memory"); break;
        case 29: asm volatile("s_waitcnt vmcnt(29)" ::: "memory"); break;
        case 30: asm volatile("s_waitcnt vmcnt(30)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(31)" ::: "memory"); break;      // callers never ask for more than 31
    }
}

// MODE: 0 = bias / row statistics only, 1 = + residual rows (prefetched into the output registers), 2 = + folded LayerNorm
// of the A rows; residual and fold never meet in one GEMM of the model.
// ABL (diagnostic builds): 1 = no epilogue at all (mainloop + tile switches only; wrong results), 3 = in-kernel stamps
template <int EPI, int NT, int MODE, int ABL = 0>
__global__ __launch_bounds__(ppx::THREADS, 2) void igemm_ppx_kernel(const IgemmParams p, const int tiles_total) {
    using namespace ppx;
    using G = Geo<NT>;
    constexpr int BN = G::BN, W_BYTES = G::W_BYTES, HALF_ROWS = G::HALF_ROWS, AUX_BASE = G::AUX_BASE;
    constexpr bool G1_W = G::HALF_PIECES > 16;            // group 1 stages W pieces 16.. of each half (NT = 5 only)
    constexpr bool LIN = EPI == EPI_LINEAR;
    // output registers: per 16-row slice, 8-byte slots.  LINEAR: one per column tile; GEGLU: one per value / gate pair
    constexpr int SLOTS = LIN ? NT : NT / 2;
    // store instructions per wave and tile: per slice one 16-byte store per slot pair, one 8-byte store for an odd slot
    constexpr int SPM = SLOTS / 2 + SLOTS % 2;            // stores per slice
    constexpr int NS = MT * SPM;                          // 15 (NT = 5), 10 (NT = 4), 5 (GEGLU)
    constexpr bool DIRECT = PPX_TRICKLE == 0;             // stores leave from the finish pass itself
    constexpr int TRICKLE = DIRECT ? 1 : PPX_TRICKLE;     // parked stores per R phase
    constexpr int ND = LIN ? NS : MT * NT / 2;            // direct stores per wave and tile (GEGLU: one 8-byte store per pair)
    [[maybe_unused]] constexpr int PHASES = (NS + TRICKLE - 1) / TRICKLE;  // R phases that carry stores: 8, 5, 3 (the launcher wants nk >= 5)
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;                    // 0: leading group, 1: trailing group (SIMD partners)
    const int q = wave & 3;
    const int wm = q >> 1, wn = grp * 2 + (q & 1);
    auto sgpr = [](int v) { return __builtin_amdgcn_readfirstlane(v); };

    constexpr bool has_fold = MODE == 2, has_res = LIN && MODE == 1;
    const bool has_bias = p.bias != nullptr;
    const bool has_rs = LIN && p.rowstat_out != nullptr;
    const bool has_cs = LIN && p.colstat_out != nullptr;      // GroupNorm column statistics of the rounded output (igemm.h)

    // ---- this workgroup's tile list: ids first, first + stride, ... (count of them)
    const int n_tiles = p.N / BN;
    int first, stride, count;
    {
        const int nwg = gridDim.x, b = blockIdx.x;
        const int xcd = b & 7;
        if (tiles_total <= nwg) {                 // one tile each: the bijective XCD chunking of igemm_pp.hip
            const int qq = nwg >> 3, r = nwg & 7;
            first = (xcd < r ? xcd * (qq + 1) : r * (qq + 1) + (xcd - r) * qq) + (b >> 3);
            stride = 1;
            count = 1;
        } else {                                  // nwg is a multiple of 8 here (launcher)
            const int per = nwg >> 3, j = b >> 3;
            const int qq = tiles_total >> 3, r = tiles_total & 7;
            const int start = xcd < r ? xcd * (qq + 1) : r * (qq + 1) + (xcd - r) * qq;
            const int size = qq + (xcd < r ? 1 : 0);
            first = start + j;
            stride = per;
            count = j < size ? (size - j + per - 1) / per : 0;
        }
    }
    if (count == 0) return;
    const int nk = p.nk;
    const int total = count * nk;                 // K-tiles this workgroup walks

    const int lr = lane >> 3;
    const int kofs = ((lane & 7) ^ lr) * 8;         // source K offset (halfs) after the slot swizzle

    // ---- per-lane staging offsets (elements), constant over the kernel; the tile's scalar offset is added at issue time
    // group 1 owns the A tile: piece j of wave q = rows (q + 4 j) * 8
    // (pieces of one wave are 32 rows apart: one per-lane offset each for A and W, the piece's row block is wave-uniform)
    const unsigned a_off0 = (unsigned)(q * 8 + lr) * (unsigned)p.lda + (unsigned)kofs;
    const unsigned a_step = 32u * (unsigned)p.lda;
    // W pieces.  Group 0 wave q: W rows (q + 4 j) * 8 (+ HALF_ROWS for the second half), j = 0..3;
    // group 1 wave q: W rows (16 + q) * 8 of each half.
    const unsigned w_off0 = (unsigned)((grp == 0 ? q * 8 : (G1_W ? (16 + q) * 8 : 0)) + lr) * (unsigned)p.ldw + (unsigned)kofs;
    const unsigned w_step = 32u * (unsigned)p.ldw;
    const unsigned w_half = (unsigned)HALF_ROWS * (unsigned)p.ldw;

    // ---- stream cursors (wave-uniform): the K-tile the NEXT W / A issue belongs to, as element offsets into W / A
    struct Cursor { int ord, k; unsigned base; };            // base = row offset of the tile (n0 * ldw or m0 * lda)
    auto tile_m0 = [&](int ord) { return ((first + ord * stride) / n_tiles) * BM; };
    auto tile_n0 = [&](int ord) { return ((first + ord * stride) % n_tiles) * BN; };
    Cursor wc{0, 0, (unsigned)tile_n0(0) * (unsigned)p.ldw};
    Cursor ac{0, 0, (unsigned)tile_m0(0) * (unsigned)p.lda};
    auto advance_w = [&]() {
        if (++wc.k == nk) { wc.k = 0; ++wc.ord; wc.base = (unsigned)sgpr(tile_n0(wc.ord)) * (unsigned)p.ldw; }
    };
    auto advance_a = [&]() {
        if (++ac.k == nk) { ac.k = 0; ++ac.ord; ac.base = (unsigned)sgpr(tile_m0(ac.ord)) * (unsigned)p.lda; }
    };

    // LDS-DMA issue helpers (1 KiB per wave instruction; destination = wave-uniform base + lane * 16)
    auto issue_a = [&](int ast, int j0, int j1) {   // A pieces j0..j1-1 of the cursor's K-tile -> A stage `ast`
        char* base = smem + ast * A_BYTES;
        const half_t* src = p.A + (size_t)ac.base + (size_t)ac.k * IGEMM_BK;
#pragma unroll
        for (int j = 0; j < 5; ++j)
            if (j >= j0 && j < j1)
                __builtin_amdgcn_global_load_lds(GLB_PTR(src + (size_t)(j * a_step) + a_off0), LDS_PTR(base + (q + 4 * j) * 1024), 16, 0, 0);
    };
    auto issue_w_g0 = [&](int wst, int half) {      // group 0: four pieces of W_lo (half 0) or W_hi (half 1)
        char* base = smem + W_BASE + wst * W_BYTES + half * (HALF_ROWS * 128);
        const half_t* src = p.W + (size_t)wc.base + (size_t)wc.k * IGEMM_BK + (half ? w_half : 0u);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            __builtin_amdgcn_global_load_lds(GLB_PTR(src + (size_t)(j * w_step) + w_off0), LDS_PTR(base + (q + 4 * j) * 1024), 16, 0, 0);
    };
    auto issue_w_g1 = [&](int wst) {                // group 1: piece 16 + q of W_lo and of W_hi
        if (!G1_W) return;
        char* base = smem + W_BASE + wst * W_BYTES;
        const half_t* src = p.W + (size_t)wc.base + (size_t)wc.k * IGEMM_BK;
        __builtin_amdgcn_global_load_lds(GLB_PTR(src + w_off0), LDS_PTR(base + (16 + q) * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds(GLB_PTR(src + w_half + w_off0), LDS_PTR(base + HALF_ROWS * 128 + (16 + q) * 1024), 16, 0, 0);
    };
    // aux pieces of the tile at (m0, n0) into aux buffer `par`: ALWAYS two per group-1 wave (slots 2q, 2q+1 of 8), so
    // that every wave's vmcnt arithmetic is the same whatever operands the GEMM has.  Slot: 0,1 bias[n0 ..], 2,3
    // ln_s[n0 ..], 4,5 ln_stats[2 m0 ..] (320 floats each, the second piece clamped inside the vector), 6,7 scratch.
    auto issue_aux = [&](int par, int m0, int n0) {
        char* base = smem + AUX_BASE + par * AUX_BYTES;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int s = 2 * q + i;                            // wave-uniform
            const int which = s >> 1, piece = s & 1;
            const int limit = which == 2 ? BM * 2 : BN;         // floats in the vector
            int idx = piece * 256 + lane * 4;
            idx = idx < limit - 4 ? idx : limit - 4;
            const float* src = which == 0 && has_bias ? p.bias + n0
                             : which == 1 && has_fold ? p.ln_s + n0
                             : which == 2 && has_fold ? p.ln_stats + (size_t)m0 * 2
                             : reinterpret_cast<const float*>(p.W);          // unused slot: any readable 2 KiB
            __builtin_amdgcn_global_load_lds(GLB_PTR(src + idx), LDS_PTR(base + (which < 3 ? s : 6) * 1024), 16, 0, 0);
        }
    };

    f32x4 acc[NT][MT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // output registers: the tile's residual rows on their way in (MODE 1), its finished fp16 rows on their way out
    unsigned long long ro[MT][SLOTS];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < SLOTS; ++j) ro[mt][j] = 0;

    const int frow = lane & 15, fsw = lane & 7, fg = lane >> 4;
    const int a_frag = (wm * (MT * 16) + frow) * 128;
    const int w_frag = W_BASE + (wn * (NT * 16) + frow) * 128;
    half8_t af[MT], wf[NT];

    auto read_frags = [&](int ast, int wst, int ks) {
        const char* abase = smem + ast * A_BYTES + a_frag;
        const char* wbase = smem + wst * W_BYTES + w_frag;
        const int slot = ((ks * 4 + fg) ^ fsw) * 16;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) af[mt] = *reinterpret_cast<const half8_t*>(abase + mt * 16 * 128 + slot);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) wf[nt] = *reinterpret_cast<const half8_t*>(wbase + nt * 16 * 128 + slot);
    };
    auto mfma_block = [&]() {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[nt], af[mt], acc[nt][mt], 0, 0, 0);
    };
    auto bar = [&]() {                              // phase boundary: nothing is scheduled across it
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- residual prefetch: the wave tile's rows of R, one 8-byte load per (mt, nt), issued at the start of the tile's
    // last K-tile.  Lane address = R + ((m0 + wave row) * ldr + n0 + wave column) * 2 bytes; nt * 32 bytes as immediates.
    auto issue_residual = [&](int m0, int n0) {
        if constexpr (has_res) {
            int ln = lane;
            asm volatile("" : "+v"(ln));                 // lane-derived addresses are NOT hoisted out of the K loop: there
                                                         // they would occupy ~30 registers for the whole kernel
            const unsigned row = (unsigned)(wm * (MT * 16) + (ln & 15)), col = (unsigned)(wn * (NT * 16) + (ln >> 4) * 4);
            const half_t* sbase = p.R + ((size_t)m0 * p.ldr + n0);
            unsigned voff = (row * (unsigned)p.ldr + col) * 2u;
            const unsigned step = 16u * (unsigned)p.ldr * 2u;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                asm_load_b64<0>(ro[mt][0], voff, sbase);
                asm_load_b64<32>(ro[mt][1], voff, sbase);
                asm_load_b64<64>(ro[mt][2], voff, sbase);
                asm_load_b64<96>(ro[mt][3], voff, sbase);
                if constexpr (NT == 5) asm_load_b64<128>(ro[mt][NT - 1], voff, sbase);
                voff += step;
            }
        }
    };
    // wait for the prefetched residual rows: `counted` = every DMA batch between their issue and here was a full one
    auto wait_residual = [&](bool counted) {
        if constexpr (has_res) {
            // group 0: W_lo + W_hi of the last step + W_lo of this step = 12; group 1: (2 W +) 5 A, then (2 W +) 2 aux + 2 A
            if (!counted) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (grp == 0) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G1_W ? 13 : 9) : "memory");
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int j = 0; j < SLOTS; ++j) asm volatile("" : "+v"(ro[mt][j]));      // uses stay behind the wait
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ---- finish the tile at (m0, n0): accumulators -> fp16 rows in `ro`, store-ready, accumulators zeroed.  VALU only.
    // Per element the operations and their order are those of igemm_epilogue.h: fold, + bias, + residual, round.
    // Two adjacent column tiles' 8-byte chunks are regrouped with v_permlane16_swap — rows 1 / 3 of the first operand trade
    // places with rows 0 / 2 of the second — into 16 contiguous bytes per lane: lane group 0 ends with columns 0-7 of tile
    // k, group 1 with columns 0-7 of tile k+1, groups 2 / 3 with their columns 8-15; one 16-byte store per lane then
    // writes 64 contiguous bytes of each of 16 rows.
    auto finish_tile = [&](int m0, int n0, int par) {
        const char* aux = smem + AUX_BASE + par * AUX_BYTES;
        int ln = lane;
        asm volatile("" : "+v"(ln));                             // see issue_residual
        const int ncol_l = wn * (NT * 16) + (ln >> 4) * 4;       // this lane's first column inside the tile (nt = 0)
        const int mrow_l = wm * (MT * 16) + (ln & 15);
        float mean[has_fold ? MT : 1], rstd[has_fold ? MT : 1];
        if constexpr (has_fold) {          // folded LayerNorm of the A rows: acc <- rstd_m * (acc - mean_m * s_n)
            // (mean, rstd) rows are read as 16-byte row PAIRS: hipcc (ROCm 7.2) puts a vmcnt(0) in front of every
            // 8-byte LDS read while LDS-DMA is in flight
            const bool odd = (ln & 1) != 0;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const f32x4 st = *reinterpret_cast<const f32x4*>(aux + 4096 + ((mrow_l & ~1) + mt * 16) * 8);
                mean[mt] = odd ? st[2] : st[0];
                rstd[mt] = odd ? st[3] : st[1];
            }
        }
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
        auto load_bs = [&](int nt, f32x4& bv, f32x4& sv) {
            bv = has_bias ? *reinterpret_cast<const f32x4*>(aux + (ncol_l + nt * 16) * 4) : zero4;
            sv = zero4;
            if constexpr (has_fold) sv = *reinterpret_cast<const f32x4*>(aux + 2048 + (ncol_l + nt * 16) * 4);
        };
        // direct stores (DIRECT): scalar base of the tile, 32-bit lane offsets
        const int g4 = ln >> 4;
        const half_t* dbase = p.C + (LIN ? (size_t)m0 * p.ldc + n0 : (size_t)m0 * p.ldc + (n0 + wn * (NT * 16)) / 2);
        const unsigned drow = (unsigned)mrow_l * (unsigned)p.ldc;
        const unsigned dvoff_pair = (drow + (unsigned)(wn * (NT * 16))) * 2u + (unsigned)((g4 & 1) * 32 + (g4 >> 1) * 16);
        const unsigned dvoff_one = LIN ? (drow + (unsigned)ncol_l) * 2u : (drow + (unsigned)(g4 * 4)) * 2u;
        const unsigned dstep = 16u * (unsigned)p.ldc * 2u;
        auto swap_into = [&](int mt, auto j_tag, half4_t oa, half4_t ob) {     // slots j, j + 1 of slice mt
            constexpr int j = decltype(j_tag)::value;
            const u32x2 ua = __builtin_bit_cast(u32x2, oa), ub = __builtin_bit_cast(u32x2, ob);
            const auto sx = __builtin_amdgcn_permlane16_swap(ua[0], ub[0], false, false);
            const auto sy = __builtin_amdgcn_permlane16_swap(ua[1], ub[1], false, false);
            if constexpr (DIRECT) {
                const u32x4 w = {sx[0], sy[0], sx[1], sy[1]};
                asm_store_b128<j * 32>(dvoff_pair + mt * dstep, w, dbase);
            } else {
                ro[mt][j] = (unsigned long long)sx[0] | ((unsigned long long)sy[0] << 32);
                ro[mt][j + 1] = (unsigned long long)sx[1] | ((unsigned long long)sy[1] << 32);
            }
        };
        if constexpr (LIN) {
            float rs_sum[MT], rs_sq[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) rs_sum[mt] = rs_sq[mt] = 0.f;
            // one accumulator -> rounded half4, then zeroed (register pressure only falls during the pass)
            auto finish = [&](auto nt_tag, int mt, const f32x4& bv, const f32x4& sv) -> half4_t {
                constexpr int nt = decltype(nt_tag)::value;
                f32x4 v = acc[nt][mt];
                if constexpr (has_fold) v = (v - mean[mt] * sv) * rstd[mt];
                v = v + bv;
                if constexpr (has_res) {
                    const half4_t rv = __builtin_bit_cast(half4_t, ro[mt][nt]);
                    v[0] += (float)rv[0]; v[1] += (float)rv[1]; v[2] += (float)rv[2]; v[3] += (float)rv[3];
                }
                acc[nt][mt] = zero4;
                const half4_t o = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
                if (has_rs) {   // sums of the ROUNDED outputs, columns in ascending order per row (igemm_epilogue.h)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { const float f = (float)o[r]; rs_sum[mt] += f; rs_sq[mt] += f * f; }
                }
                return o;
            };
            // column statistics of one column tile: the 16-lane DPP tree of igemm_epilogue.h, then ONE 16-byte asm store (lane 0
            // of a lane group the four sums, lane 1 the four sums of squares) so that the step's vmcnt arithmetic knows their number: NT
            const half_t* cbase = reinterpret_cast<const half_t*>(p.colstat_out);
            const unsigned cs_voff = (unsigned)cs_index((size_t)((m0 + wm * (MT * 16)) / (MT * 16)), n0 + ncol_l, ln & 1, p.N) * 4u;
            auto cs_add = [&](f32x4& cs, f32x4& cq, const half4_t& o) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float f = (float)o[r]; cs[r] += f; cq[r] += f * f; }
            };
            auto cs_emit = [&](auto nt_tag, const f32x4& cs, const f32x4& cq) {
                constexpr int nt = decltype(nt_tag)::value;
                f32x4 a, b;
#pragma unroll
                for (int r = 0; r < 4; ++r) { a[r] = row16_sum(cs[r]); b[r] = row16_sum(cq[r]); }
                const f32x4 w = (ln & 1) ? b : a;
                if ((ln & 15) < 2) asm_store_b128<nt * 128>(cs_voff, __builtin_bit_cast(u32x4, w), cbase);
            };
            auto pair = [&](auto nt_tag) {
                constexpr int nt = decltype(nt_tag)::value;
                f32x4 bv0, sv0, bv1, sv1;
                load_bs(nt, bv0, sv0);
                load_bs(nt + 1, bv1, sv1);
                f32x4 cs0 = zero4, cq0 = zero4, cs1 = zero4, cq1 = zero4;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const half4_t oa = finish(std::integral_constant<int, nt>{}, mt, bv0, sv0);
                    const half4_t ob = finish(std::integral_constant<int, nt + 1>{}, mt, bv1, sv1);
                    if (has_cs) { cs_add(cs0, cq0, oa); cs_add(cs1, cq1, ob); }
                    swap_into(mt, std::integral_constant<int, nt>{}, oa, ob);
                }
                if (has_cs) {
                    cs_emit(std::integral_constant<int, nt>{}, cs0, cq0);
                    cs_emit(std::integral_constant<int, nt + 1>{}, cs1, cq1);
                }
                __builtin_amdgcn_sched_barrier(0);               // keep the column tiles apart: the register file is full
            };
            pair(std::integral_constant<int, 0>{});
            pair(std::integral_constant<int, 2>{});
            if constexpr (NT == 5) {
                f32x4 bv, sv;
                load_bs(4, bv, sv);
                f32x4 cs4 = zero4, cq4 = zero4;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const half4_t o = finish(std::integral_constant<int, 4>{}, mt, bv, sv);
                    if (has_cs) cs_add(cs4, cq4, o);
                    if constexpr (DIRECT) asm_store_b64<128>(dvoff_one + mt * dstep, o, dbase);
                    else ro[mt][4] = __builtin_bit_cast(unsigned long long, o);
                }
                if (has_cs) cs_emit(std::integral_constant<int, 4>{}, cs4, cq4);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (has_rs) {   // this wave's 16*NT columns of each row: fold the four 16-lane groups; ONE 8-byte asm store per
                            // slice (lane group 0), so that the step's vmcnt arithmetic knows their number: MT
                const half_t* rbase = reinterpret_cast<const half_t*>(p.rowstat_out);
                const unsigned slots = (unsigned)(p.N / (NT * 16));
                unsigned voff = (((unsigned)(m0 + mrow_l)) * slots + (unsigned)((n0 + wn * (NT * 16)) / (NT * 16))) * 8u;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    float a = rs_sum[mt], b = rs_sq[mt];
                    a += __shfl_xor(a, 16, 64); b += __shfl_xor(b, 16, 64);
                    a += __shfl_xor(a, 32, 64); b += __shfl_xor(b, 32, 64);
                    const u32x2 ab = {__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b)};
                    if ((ln >> 4) == 0) asm_store_b64<0>(voff, __builtin_bit_cast(half4_t, ab), rbase);
                    voff += 16u * slots * 8u;
                }
            }
        } else {                         // GEGLU: tile nt (even) holds h, nt + 1 the matching gate; output column = n / 2
            auto column = [&](auto nt_tag, half4_t (&o)[MT]) {
                constexpr int nt = decltype(nt_tag)::value;
                f32x4 bh, sh, bg, sg;
                load_bs(nt, bh, sh);
                load_bs(nt + 1, bg, sg);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    f32x4 h = acc[nt][mt], g = acc[nt + 1][mt];
                    if constexpr (has_fold) {
                        h = (h - mean[mt] * sh) * rstd[mt];
                        g = (g - mean[mt] * sg) * rstd[mt];
                    }
                    h = h + bh;
                    g = g + bg;
                    acc[nt][mt] = zero4;
                    acc[nt + 1][mt] = zero4;
                    o[mt] = (half4_t){(half_t)(h[0] * gelu_erf_f(g[0])), (half_t)(h[1] * gelu_erf_f(g[1])),
                                      (half_t)(h[2] * gelu_erf_f(g[2])), (half_t)(h[3] * gelu_erf_f(g[3]))};
                    if constexpr (DIRECT) asm_store_b64<nt * 16>(dvoff_one + mt * dstep, o[mt], dbase);
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            half4_t o0[MT], o1[MT];
            column(std::integral_constant<int, 0>{}, o0);
            column(std::integral_constant<int, 2>{}, o1);
            if constexpr (!DIRECT) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) swap_into(mt, std::integral_constant<int, 0>{}, o0[mt], o1[mt]);
            }
        }
    };

    // ---- the stores of the finished tile whose scalar base is st_base: store s of 0 .. NS-1 is slice s / SPM, slot
    // pair (s % SPM) of it (the odd last slot as an 8-byte store)
    const half_t* st_base = p.C;
    auto store_one = [&](auto s_tag, unsigned voff_pair, unsigned voff_single, unsigned step) {
        constexpr int s = decltype(s_tag)::value, mt = s / SPM, j = s % SPM;
        if constexpr (2 * j + 1 < SLOTS) {
            const u32x4 w = {(unsigned)ro[mt][2 * j], (unsigned)(ro[mt][2 * j] >> 32), (unsigned)ro[mt][2 * j + 1], (unsigned)(ro[mt][2 * j + 1] >> 32)};
            asm_store_b128<j * 64>(voff_pair + mt * step, w, st_base);
        } else {
            asm_store_b64<j * 64>(voff_single + mt * step, __builtin_bit_cast(half4_t, ro[mt][2 * j]), st_base);
        }
    };
    auto issue_stores = [&](int s_lo, int s_hi) {       // stores s_lo .. s_hi-1 (wave-uniform bounds)
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int g4 = ln >> 4;
        const unsigned ldo = (unsigned)p.ldc;
        // 16-byte stores: lane group g4 writes bytes (g4 & 1) * 32 + (g4 >> 1) * 16 of the slot pair's 64-byte row segment
        const unsigned row_off = (unsigned)(wm * (MT * 16) + (ln & 15)) * ldo;
        const unsigned wcol = LIN ? (unsigned)(wn * (NT * 16)) : (unsigned)(wn * (NT * 16)) / 2;
        const unsigned voff_pair = (row_off + wcol) * 2u + (unsigned)((g4 & 1) * 32 + (g4 >> 1) * 16);
        const unsigned voff_single = (row_off + wcol + (unsigned)(g4 * 4)) * 2u;
        const unsigned step = 16u * ldo * 2u;
        auto maybe = [&](auto s_tag) {
            constexpr int s = decltype(s_tag)::value;
            if constexpr (s < NS) {
                if (s >= s_lo && s < s_hi) store_one(s_tag, voff_pair, voff_single, step);
            }
        };
        maybe(std::integral_constant<int, 0>{});  maybe(std::integral_constant<int, 1>{});  maybe(std::integral_constant<int, 2>{});
        maybe(std::integral_constant<int, 3>{});  maybe(std::integral_constant<int, 4>{});  maybe(std::integral_constant<int, 5>{});
        maybe(std::integral_constant<int, 6>{});  maybe(std::integral_constant<int, 7>{});  maybe(std::integral_constant<int, 8>{});
        maybe(std::integral_constant<int, 9>{});  maybe(std::integral_constant<int, 10>{}); maybe(std::integral_constant<int, 11>{});
        maybe(std::integral_constant<int, 12>{}); maybe(std::integral_constant<int, 13>{}); maybe(std::integral_constant<int, 14>{});
    };

    // ---- stagger: the workgroups of an XCD start a quarter of a tile period apart, so that at any moment some CUs are
    // in their K loops (HBM reads) while others drain a finished tile (HBM writes) instead of the whole chip alternating
    if (PPX_STAGGER && tiles_total > (int)gridDim.x) {
        const int ph = (blockIdx.x >> 3) & 3;
        if (ph) {
            const unsigned long long ticks = (unsigned long long)ph * (unsigned long long)(nk * 40 + 75);   // 100 MHz ticks: ph/4 of ~(1.6 nk + 3) us
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
        }
    }
    // ---- prologue (first tile only): W(0), A(0), A(1) by the same roles; everything landed before the first read
    if (grp == 0) {
        issue_w_g0(0, 0);
        issue_w_g0(0, 1);
    } else {
        issue_w_g1(0);
        issue_a(0, 0, 5);
        advance_a();
        if (1 < total) issue_a(1, 0, 5);
    }
    advance_w();                                        // W cursor -> K-tile 1 of the stream
    if (grp == 0) { advance_a(); }                      // keep both groups' A cursors at K-tile 2 (group 0 never issues A)
    if (1 < total) advance_a();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    bar();
    if (grp == 1) bar();                               // the trailing group runs one barrier behind

    unsigned long long st_sum[32], st_prev = 0, st_begin = 0;
    if constexpr (ABL == 3) {
#pragma unroll
        for (int i = 0; i < 32; ++i) st_sum[i] = 0;
    }
    int st_base_idx = 0;
    auto stamp = [&](int seg_id) {
        if constexpr (ABL == 3) {
            unsigned long long tnow;
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tnow)::"memory");
            __builtin_amdgcn_sched_barrier(0);
            if (seg_id >= 0) {
#pragma unroll
                for (int i = 0; i < 10; ++i) {           // static indices only (runtime-indexed arrays go to scratch)
                    if (i == seg_id) {
                        if (st_base_idx) st_sum[16 + i] += tnow - st_prev;
                        else st_sum[i] += tnow - st_prev;
                    }
                }
            } else {
                st_begin = tnow;
            }
            st_prev = tnow;
        }
    };
    stamp(-1);
    int ast = 0;                                       // A stage of the current K-tile; K-tile g+2 goes to stage ast2
    int kcur = 0, ord = 0;                             // position of the current K-tile inside its output tile
    int cm0 = sgpr(tile_m0(0)), cn0 = sgpr(tile_n0(0));      // current tile
    bool pend = false;                                 // the previous tile's accumulators are not finished yet
    bool st_live = false;                              // `ro` holds a finished tile whose stores are trickling out
    int pm0 = 0, pn0 = 0;
    int n1_prev = 0;                                   // stores issued in the previous step's R(g-1,1) phase
    for (int g = 0; g < total; ++g) {
        const int wst = g & 1;
        const int ast2 = ast == 0 ? 2 : ast - 1;       // (ast + 2) % 3
        const bool more1 = g + 1 < total, more2 = g + 2 < total;
        const bool tfirst = kcur == 0, tlast = kcur == nk - 1;
        const bool ep_now = pend;                      // this step finishes the previous tile
        if constexpr (ABL == 3) { st_base_idx = ep_now ? 1 : 0; if (ep_now) st_sum[11] += 1; else st_sum[10] += 1; }
        // ---- R(g, 0): residual prefetch of a tile's last step, the DMA batch, the deferred finish, this phase's stores
        if (ABL != 1 && tlast && has_res) issue_residual(cm0, cn0);
        if (grp == 0) {
            if (more1) issue_w_g0(wst ^ 1, 0);
        } else {
            if (more1) issue_w_g1(wst ^ 1);            // W pieces first: the wait in R(g,1) leaves only younger pieces behind
            if (tfirst) issue_aux(ord & 1, cm0, cn0);
            if (more2) issue_a(ast2, 0, 2);
        }
        stamp(0);
        int n_extra = 0;                               // row-statistics stores of the finish (MT asm stores)
        if (ep_now) {
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (ABL != 1) {
                if (has_res) wait_residual(more2);
                finish_tile(pm0, pn0, (ord - 1) & 1);  // leaves the accumulators zeroed
                if constexpr (DIRECT) {
                    n_extra = ND;                      // its stores are out already: this step's waits skip over them
                } else {
                    st_base = p.C + (LIN ? (size_t)pm0 * p.ldc + pn0 : (size_t)pm0 * p.ldc + pn0 / 2);
                    st_live = true;
                }
                if (has_rs) n_extra += MT;
                if (has_cs) n_extra += NT;
            } else {
                if (pm0 < 0) finish_tile(pm0, pn0, 0); // never taken: keeps the accumulators live
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            pend = false;
            __builtin_amdgcn_sched_barrier(0);
        }
        // trickled stores of this step: phases 2 kcur (R0) and 2 kcur + 1 (R1) of the tile finished above / earlier
        const int ph0 = 2 * kcur;
        int n0 = 0, n1 = 0;
        if (st_live && ABL != 1) {
            const int lo0 = ph0 * TRICKLE, lo1 = lo0 + TRICKLE, lo2 = lo1 + TRICKLE;
            n0 = lo0 >= NS ? 0 : (lo1 <= NS ? TRICKLE : NS - lo0);
            n1 = lo1 >= NS ? 0 : (lo2 <= NS ? TRICKLE : NS - lo1);
            if (n0) issue_stores(lo0, lo0 + n0);
        }
        stamp(1);
        read_frags(ast, wst, 0);
        if (grp == 0) {                                // W_hi(g), issued in R(g-1,1) ahead of that phase's stores, has landed
            if (more1) vmwait(4 + n1_prev + n_extra + n0);
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        stamp(2);
        bar();
        stamp(3);
        // ---- M(g, 0)
        mfma_block();
        stamp(4);
        bar();
        stamp(5);
        // ---- R(g, 1)
        read_frags(ast, wst, 1);
        if (grp == 0) {
            if (more1) issue_w_g0(wst ^ 1, 1);
        } else {
            if (more2) issue_a(ast2, 2, 5);
        }
        if (n1) issue_stores((ph0 + 1) * TRICKLE, (ph0 + 1) * TRICKLE + n1);
        if (st_live && (ph0 + 2) * TRICKLE >= NS) st_live = false;      // all of the tile's stores are out
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (grp == 1) {                                // W(g+1) pieces, this tile's aux and A(g+1) have landed
            if (more2) vmwait(5 + n_extra + n0 + n1);
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (more1) advance_w();
        if (more2) advance_a();
        stamp(6);
        bar();
        stamp(7);
        // ---- M(g, 1)
        mfma_block();
        if (grp == 0) {                                // W_lo(g+1), issued in R(g,0), has landed
            if (more1) vmwait(4 + n_extra + n0 + n1);
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        n1_prev = n1;
        stamp(8);
        if (!(grp == 1 && !more1)) bar();
        stamp(9);
        ast = ast == 2 ? 0 : ast + 1;
        if (++kcur == nk) {                            // tile finished: its accumulators are dealt with inside the next step
            kcur = 0;
            pend = true;
            pm0 = cm0;
            pn0 = cn0;
            ++ord;
            if (more1) { cm0 = sgpr(tile_m0(ord)); cn0 = sgpr(tile_n0(ord)); }
        }
    }
    if constexpr (ABL == 3) {
        if (blockIdx.x == 0 && lane == 0) {
#pragma unroll
            for (int i = 0; i < 32; ++i) g_ppx_stamps[wave * 32 + i] = i == 12 ? st_prev - st_begin : st_sum[i];
        }
    }
    // ---- the last tile: nothing left to overlap; finish, store, done
    if constexpr (ABL != 1) {
        if (has_res) wait_residual(false);
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        finish_tile(pm0, pn0, (ord - 1) & 1);
        if constexpr (!DIRECT) {
            st_base = p.C + (LIN ? (size_t)pm0 * p.ldc + pn0 : (size_t)pm0 * p.ldc + pn0 / 2);
            issue_stores(0, NS);
        }
    } else {
        if (pm0 < 0) finish_tile(pm0, pn0, 0);
    }
}

template <int EPI, int NT, int MODE, int ABL = 0>
static int launch_ppx_t(const IgemmParams& p, hipStream_t stream) {
    using namespace ppx;
    constexpr int BN = Geo<NT>::BN;
    constexpr int lds = Geo<NT>::LDS_BYTES;
    auto kern = igemm_ppx_kernel<EPI, NT, MODE, ABL>;
    static bool attr_set = false;
    if (!attr_set) {
        LAVIE_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_set = true;
    }
    const int tiles = (p.M / BM) * (p.N / BN);
    const int grid = tiles < MAX_WG ? tiles : MAX_WG;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(THREADS), lds, stream, p, tiles);
    LAVIE_HIP(hipGetLastError());
    return 0;
}

// Shapes the persistent kernel takes: plain A rows, no split-K, no per-batch bias, whole 160-row tiles, at least five
// K-tiles (a tile's stores trickle out over the next tile's first four), N a multiple of the tile width (320, or 256 for GEGLU and the power-of-two widths of the VSR UNet), operands
// within 32-bit element offsets, 8-byte aligned residual / output rows.
bool igemm_ppx_eligible(const IgemmParams& p, int epilogue) {
    if (p.splits != 1 || p.M % ppx::BM != 0 || p.nk < 5 || p.bias2) return false;
    if ((double)p.M * p.lda >= 4.0e9 || (double)p.N * p.ldw >= 4.0e9 || (double)p.M * p.ldr * 2.0 >= 4.0e9) return false;
    if (p.ldc % 4 != 0 || (p.R && p.ldr % 4 != 0)) return false;
    if (p.R && p.ln_stats) return false;            // residual and LayerNorm fold in one GEMM: not built (never occurs in the model)
    if (epilogue == EPI_GEGLU) return p.N % 256 == 0;
    return p.N % 320 == 0 || p.N % 256 == 0;
}

static int g_ppx_abl = 0;
void igemm_ppx_ablate(int a) { g_ppx_abl = a; }
int igemm_ppx_read_stamps(unsigned long long* out) {
    LAVIE_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(ppx::g_ppx_stamps), sizeof(unsigned long long) * 8 * 32));
    return 0;
}

template <int EPI, int NT>
static int launch_ppx_mode(const IgemmParams& p, hipStream_t stream) {
    if (g_ppx_abl == 1) return launch_ppx_t<EPI, NT, 0, 1>(p, stream);
    if (g_ppx_abl == 3 && EPI == EPI_LINEAR && NT == 5) return p.R ? launch_ppx_t<EPI_LINEAR, 5, 1, 3>(p, stream) : launch_ppx_t<EPI_LINEAR, 5, 0, 3>(p, stream);
    if (p.ln_stats) return launch_ppx_t<EPI, NT, 2>(p, stream);
    if (EPI == EPI_LINEAR && p.R) return launch_ppx_t<EPI, NT, 1>(p, stream);
    return launch_ppx_t<EPI, NT, 0>(p, stream);
}

int launch_igemm_ppx(const IgemmParams& p, int epilogue, hipStream_t stream) {
    LAVIE_CHECK(igemm_ppx_eligible(p, epilogue), "igemm_ppx: shape M=%d N=%d not eligible", p.M, p.N);
    if (epilogue == EPI_GEGLU) return launch_ppx_mode<EPI_GEGLU, 4>(p, stream);
    if (p.N % 320 == 0) return launch_ppx_mode<EPI_LINEAR, 5>(p, stream);
    return launch_ppx_mode<EPI_LINEAR, 4>(p, stream);
}

}  // namespace lavie
