// Loop-structure probe for the 320x160 conv / GEMM tile (run through gpurun): the SAME work per K-tile and wave — 2 k-steps x
// (10 ds_read_b128 fragment reads + 25 v_mfma_f32_16x16x32_f16) on an 80x80 wave tile, 8 waves, one workgroup per CU, LDS-DMA of
// 26 one-KiB pieces per K-tile from an L2-resident buffer — in three loop structures:
//   V0  the shipped ping-pong: two groups of four waves half a k-step apart, R phase (reads + DMA issue, lgkmcnt(0)) opposite the
//       partner's M phase (25 compiler-issued MFMAs), four barriers per K-tile
//   V1  software-pipelined: every wave interleaves the NEXT k-step's ten reads (inline asm) between the 25 in-place asm MFMAs of
//       this k-step, one lgkmcnt(0) per k-step, ONE barrier per K-tile (stage hand-over), DMA issue right behind it
//   V2  V1 without the barrier (upper bound of the structure)
// Prints TFLOP/s of the whole chip.  No results are checked (operands are whatever the buffer holds).
// Build: hipcc -O3 --offload-arch=gfx950 tools/probes/pipe_probe.hip -o tools/probes/pipe_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

constexpr int PATCH = 57472, WST = 20480, W_BASE = 2 * PATCH, LDS_BYTES = W_BASE + 2 * WST;

__device__ __forceinline__ void lds_read(half8& d, unsigned addr) { asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(addr) : "memory"); }
__device__ __forceinline__ void mfma(f32x4& acc, const half8& a, const half8& b) {
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}

template <int V, int DMA>
__global__ __launch_bounds__(512, 2) void probe(const _Float16* g, float* out, int tiles, int gpieces) {
    extern __shared__ __attribute__((aligned(128))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wm = wave & 3, wn = grp;
    for (int i = tid; i < LDS_BYTES / 16; i += 512) reinterpret_cast<uint4*>(smem)[i] = reinterpret_cast<const uint4*>(g)[i & 4095];
    __syncthreads();
    const int frow = lane & 15, fg = lane >> 4;
    const unsigned lbase = (unsigned)(size_t)LDS_PTR(smem);
    unsigned a_frag[5], w_frag[5];
    for (int mt = 0; mt < 5; ++mt) a_frag[mt] = lbase + (wm * 80 + mt * 16 + frow) * 128 + ((fg ^ (frow & 7)) << 4);
    for (int nt = 0; nt < 5; ++nt) w_frag[nt] = lbase + W_BASE + (wn * 80 + nt * 16 + frow) * 128 + ((fg ^ (frow & 7)) << 4);
    f32x4 acc[5][5];
    for (int i = 0; i < 5; ++i)
        for (int j = 0; j < 5; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // this wave's DMA pieces of a K-tile: 2 or 3 weight pieces + 1 patch piece
    const int lr = lane >> 3, kofs = ((lane & 7) ^ lr) * 8;
    auto dma = [&](int t, int i, char* dst) {
        const _Float16* src = g + (size_t)(((blockIdx.x * 131 + t * 29 + wave * 4 + i) % gpieces) * 512) + lr * 64 + kofs;
        __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(dst), 16, 0, 0);
    };
    // register-staged copy (DMA == 2): the same pieces as plain 16-byte loads, written to LDS one K-tile later
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 stg0 = {}, stg1 = {}, stg2 = {}, stg3 = {};
#define GSRC(t, i) (g + (size_t)(((blockIdx.x * 131 + (t) * 29 + wave * 4 + (i)) % gpieces) * 512) + lr * 64 + kofs)
#define GLOAD(t, i, REG) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(REG) : "v"(GSRC(t, i)) : "memory")
#define LWRITE(REG, dst) asm volatile("ds_write_b128 %0, %1" ::"v"((unsigned)(size_t)LDS_PTR(dst) + lane * 16), "v"(REG) : "memory")
    auto bar = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };

    if constexpr (V == 0) {
        half8 af[5], wf[5];
        if (grp == 1) bar();
        for (int t = 0; t < tiles; ++t) {
            const int wst = t & 1, kt = t % 9, pb = (t / 9) & 1;
            const unsigned ashift = pb * PATCH + (kt / 3) * 64 * 128 + (kt % 3) * 128;
            for (int ks = 0; ks < 2; ++ks) {
                const unsigned kx = ks << 6;
#pragma unroll
                for (int mt = 0; mt < 5; ++mt) af[mt] = *reinterpret_cast<const half8*>(smem + (((a_frag[mt] - lbase) + ashift) ^ kx));
#pragma unroll
                for (int nt = 0; nt < 5; ++nt) wf[nt] = *reinterpret_cast<const half8*>(smem + (((w_frag[nt] - lbase) + wst * WST) ^ kx));
                if (DMA == 1) {
                    char* wdst = smem + W_BASE + (wst ^ 1) * WST + grp * 10240;
                    if (ks == 0) { dma(t, 0, wdst + wm * 1024); dma(t, 1, wdst + (wm + 4) * 1024); }
                    else {
                        if (wm < 2) dma(t, 2, wdst + (wm + 8) * 1024);
                        if (kt < 7) dma(t, 3, smem + (pb ^ 1) * PATCH + (kt * 8 + wave) * 1024);
                    }
                }
                if (DMA == 2 && ks == 0) {       // data loaded during the previous K-tile -> the stage nobody reads any more; next loads
                    char* wdst = smem + W_BASE + (wst ^ 1) * WST + grp * 10240;
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    LWRITE(stg0, wdst + wm * 1024);
                    LWRITE(stg1, wdst + (wm + 4) * 1024);
                    if (wm < 2) LWRITE(stg2, wdst + (wm + 8) * 1024);
                    if (kt < 7) LWRITE(stg3, smem + (pb ^ 1) * PATCH + (kt * 8 + wave) * 1024);
                    GLOAD(t, 0, stg0); GLOAD(t, 1, stg1);
                    if (wm < 2) GLOAD(t, 2, stg2);
                    if (kt < 7) GLOAD(t, 3, stg3);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                bar();
#pragma unroll
                for (int nt = 0; nt < 5; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 5; ++mt) acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[nt], af[mt], acc[nt][mt], 0, 0, 0);
                if (ks == 1) {
                    if (DMA == 1) { if (kt < 7) asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
                }
                if (!(ks == 1 && grp == 1 && t + 1 == tiles)) bar();
            }
        }
    } else {
        half8 fa[2][5], fw[2][5];
        auto read_set = [&](int set, int i, unsigned ashift, unsigned wsh, unsigned kx) {      // read i of 10 of a k-step into set
            if (i < 5) lds_read(fa[set][i], (a_frag[i] + ashift) ^ kx);
            else lds_read(fw[set][i - 5], (w_frag[i - 5] + wsh) ^ kx);
        };
        // prologue: k-step (0, 0)
#pragma unroll
        for (int i = 0; i < 10; ++i) read_set(0, i, 0, 0, 0);
        for (int t = 0; t < tiles; ++t) {
            const int wst = t & 1, kt = t % 9, pb = (t / 9) & 1;
            const int t1 = t + 1, kt1 = t1 % 9, pb1 = (t1 / 9) & 1;
            const unsigned ashift = pb * PATCH + (kt / 3) * 64 * 128 + (kt % 3) * 128;
            const unsigned ashift1 = pb1 * PATCH + (kt1 / 3) * 64 * 128 + (kt1 % 3) * 128;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (ks == 1) {
                    // middle of the K-tile: every wave's reads of this tile's weight stage are over, the next tile's weights (issued one
                    // tile ago) have landed for this wave -> barrier -> the stage is free for tile t + 2... (here: t + 1 into the other stage)
                    if (DMA == 1) { if (kt < 7) asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
                    if (V == 1) bar();
                    if (DMA == 2) {
                        char* wdst = smem + W_BASE + wst * WST + grp * 10240;
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        LWRITE(stg0, wdst + wm * 1024);
                        LWRITE(stg1, wdst + (wm + 4) * 1024);
                        if (wm < 2) LWRITE(stg2, wdst + (wm + 8) * 1024);
                        if (kt < 7) LWRITE(stg3, smem + (pb ^ 1) * PATCH + (kt * 8 + wave) * 1024);
                        GLOAD(t, 0, stg0); GLOAD(t, 1, stg1);
                        if (wm < 2) GLOAD(t, 2, stg2);
                        if (kt < 7) GLOAD(t, 3, stg3);
                    }
                    if (DMA == 1) {
                        char* wdst = smem + W_BASE + wst * WST + grp * 10240;      // the stage whose reads just ended
                        dma(t, 0, wdst + wm * 1024);
                        dma(t, 1, wdst + (wm + 4) * 1024);
                        if (wm < 2) dma(t, 2, wdst + (wm + 8) * 1024);
                        if (kt < 7) dma(t, 3, smem + (pb ^ 1) * PATCH + (kt * 8 + wave) * 1024);
                    }
                }
                // 25 MFMAs of this k-step with the next k-step's ten reads between them (one read per two MFMAs, from the third on)
                const unsigned nash = ks == 0 ? ashift : ashift1;
                const unsigned nwsh = (ks == 0 ? wst : (wst ^ 1)) * WST;
                const unsigned nkx = ks == 0 ? 64u : 0u;
#pragma unroll
                for (int nt = 0; nt < 5; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 5; ++mt) {
                        mfma(acc[nt][mt], fw[ks][nt], fa[ks][mt]);
                        const int m = nt * 5 + mt;
                        if (m >= 2 && m % 2 == 0 && (m - 2) / 2 < 10) read_set(ks ^ 1, (m - 2) / 2, nash, nwsh, nkx);
                    }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    float s = 0.f;
    for (int i = 0; i < 5; ++i)
        for (int j = 0; j < 5; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[blockIdx.x * 512 + tid] = s;
}

template <int V, int DMA>
static void run(const char* name, const _Float16* g, float* out, int gpieces) {
    const int tiles = 1800, grid = 256;
    hipFuncSetAttribute((const void*)probe<V, DMA>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((probe<V, DMA>), dim3(grid), dim3(512), LDS_BYTES, 0, g, out, tiles, gpieces);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double flop = (double)grid * 8 * tiles * 50 * (16.0 * 16 * 32 * 2);
        printf("%-46s %8.3f ms  %7.1f TFLOP/s  (%.2f of 2500)\n", name, ms, flop / ms / 1e9, flop / ms / 1e9 / 2500);
    }
}

int main(int argc, char** argv) {
    const int gpieces = argc > 1 ? atoi(argv[1]) : 16384;       // source pieces (KiB): 16 MiB = beyond one XCD's L2 (MALL resident); 1024 = L2 resident
    printf("source buffer: %d KiB\n", gpieces);
    _Float16* g;
    float* out;
    hipMalloc(&g, (size_t)(gpieces < 4096 ? 4096 : gpieces) * 1024);
    hipMalloc(&out, 256 * 512 * 4);
    hipMemset(g, 0x11, (size_t)(gpieces < 4096 ? 4096 : gpieces) * 1024);
    run<0, 0>("V0 ping-pong, no DMA", g, out, gpieces);
    run<0, 1>("V0 ping-pong, 26 DMA pieces / K-tile", g, out, gpieces);
    run<0, 2>("V0 ping-pong, register-staged copies", g, out, gpieces);
    run<1, 0>("V1 interleaved, 1 barrier / K-tile, no DMA", g, out, gpieces);
    run<1, 1>("V1 interleaved, 1 barrier / K-tile, DMA", g, out, gpieces);
    run<1, 2>("V1 interleaved, 1 barrier, register-staged", g, out, gpieces);
    run<2, 0>("V2 interleaved, no barrier, no DMA", g, out, gpieces);
    run<2, 1>("V2 interleaved, no barrier, DMA", g, out, gpieces);
    hipDeviceSynchronize();
    printf("%s\n", hipGetErrorString(hipGetLastError()));
    return 0;
}
