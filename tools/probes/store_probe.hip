// Store-pattern probe (MI355X): bandwidth of one wave-instruction shapes used by GEMM epilogues, on a [M, 320] fp16
// matrix (640-byte rows) far larger than the caches.  Each 512-thread workgroup owns a 160 x 320 tile like the GEMMs.
//   0: 16 rows x 32 B per instruction (8 B per lane, the MFMA accumulator layout: row-per-lane)
//   1: 16 rows x 64 B per instruction (16 B per lane, after v_permlane16_swap)
//   2:  8 rows x 128 B per instruction, line-aligned (16 B per lane, full cache lines)
//   3:  1 KiB contiguous per instruction (16 B per lane, 1.6 rows)
//   4: as 2 but 160-byte row segments (80 columns) starting at 160 B * wave column: what per-wave LDS staging could emit
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(512) void probe(unsigned short* C, int tiles) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = (wave & 3) >> 1, wn = (wave >> 2) * 2 + (wave & 1);
    for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
        char* tile = (char*)C + (size_t)t * 160 * 640;
        if (MODE == 0) {
            for (int mt = 0; mt < 5; ++mt)
                for (int nt = 0; nt < 5; ++nt) {
                    char* p = tile + (size_t)(wm * 80 + mt * 16 + (lane & 15)) * 640 + (wn * 80 + nt * 16 + (lane >> 4) * 4) * 2;
                    *(uint2*)p = make_uint2(lane, t);
                }
        } else if (MODE == 1) {
            for (int mt = 0; mt < 5; ++mt) {
                for (int pr = 0; pr < 2; ++pr) {
                    const int g = lane >> 4;
                    char* p = tile + (size_t)(wm * 80 + mt * 16 + (lane & 15)) * 640 + (wn * 80 + pr * 32) * 2 + (g & 1) * 32 + (g >> 1) * 16;
                    *(uint4*)p = make_uint4(lane, t, mt, pr);
                }
                char* p = tile + (size_t)(wm * 80 + mt * 16 + (lane & 15)) * 640 + (wn * 80 + 64 + (lane >> 4) * 4) * 2;
                *(uint2*)p = make_uint2(lane, t);
            }
        } else if (MODE == 2) {
            // 160 rows x 640 B = 800 lines of 128 B; wave w takes lines w, w + 8, ...: 8 lines (8 rows x 128 B?) per instr:
            // instruction i covers 8 consecutive lines of one row group: rows r..r+7, line column c
            for (int i = wave; i < 100; i += 8) {            // 100 units of (8 rows x 128 B)
                const int rg = i / 5, lc = i % 5;
                char* p = tile + (size_t)(rg * 8 + (lane >> 3)) * 640 + lc * 128 + (lane & 7) * 16;
                *(uint4*)p = make_uint4(lane, t, i, 0);
            }
        } else if (MODE == 3) {
            for (int i = wave; i < 100; i += 8) *(uint4*)(tile + (size_t)i * 1024 + lane * 16) = make_uint4(lane, t, i, 0);
        } else if (MODE == 5) {          // as 3 with nontemporal stores
            for (int i = wave; i < 100; i += 8) __builtin_nontemporal_store((u4){(unsigned)lane, (unsigned)t, (unsigned)i, 0u}, (u4*)(tile + (size_t)i * 1024 + lane * 16));
        } else if (MODE == 6) {          // as 1 (16 rows x 64 B) with nontemporal stores
            for (int mt = 0; mt < 5; ++mt) {
                for (int pr = 0; pr < 2; ++pr) {
                    const int g = lane >> 4;
                    char* p = tile + (size_t)(wm * 80 + mt * 16 + (lane & 15)) * 640 + (wn * 80 + pr * 32) * 2 + (g & 1) * 32 + (g >> 1) * 16;
                    __builtin_nontemporal_store((u4){(unsigned)lane, (unsigned)t, (unsigned)mt, (unsigned)pr}, (u4*)p);
                }
                char* p = tile + (size_t)(wm * 80 + mt * 16 + (lane & 15)) * 640 + (wn * 80 + 64 + (lane >> 4) * 4) * 2;
                __builtin_nontemporal_store((u2){(unsigned)lane, (unsigned)t}, (u2*)p);
            }
        } else if (MODE == 7) {          // as 3 but TWO workgroups per CU (256 threads each... same waves per CU, more blocks)
            for (int i = wave; i < 100; i += 8) *(uint4*)(tile + (size_t)i * 1024 + lane * 16) = make_uint4(lane, t, i, 0);
        } else {
            // per wave: its 80 x 80 sub-tile as 160-byte row segments: 10 lanes per row, 6.4 rows per instruction
            for (int i = 0; i < 13; ++i) {
                const int chunk = i * 64 + lane;               // 800 chunks of 16 B
                if (chunk < 800) {
                    const int row = chunk / 10, c = chunk % 10;
                    *(uint4*)(tile + (size_t)(wm * 80 + row) * 640 + wn * 160 + c * 16) = make_uint4(lane, t, i, 0);
                }
            }
        }
    }
}

int main() {
    const int tiles = 4096;                                  // 4096 x 100 KiB = 400 MiB
    unsigned short* C;
    CHECK(hipMalloc(&C, (size_t)tiles * 160 * 640));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    auto run = [&](auto kern, const char* name) {
        for (int rep = 0; rep < 2; ++rep) {
            CHECK(hipEventRecord(a));
            hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, C, tiles);
            CHECK(hipEventRecord(b));
            CHECK(hipEventSynchronize(b));
            float ms;
            CHECK(hipEventElapsedTime(&ms, a, b));
            if (rep) printf("%-44s %8.1f us  %6.2f TB/s  (%.2f us per tile per CU)\n", name, ms * 1e3, (double)tiles * 160 * 640 / ms / 1e9, ms * 1e3 / (tiles / 256.0));
        }
    };
    run(probe<0>, "0: 16 rows x 32 B (8 B/lane, row-per-lane)");
    run(probe<1>, "1: 16 rows x 64 B (16 B/lane, swapped)");
    run(probe<2>, "2: 8 rows x 128 B full lines");
    run(probe<3>, "3: 1 KiB contiguous");
    run(probe<4>, "4: 160-B row segments per wave");
    run(probe<5>, "5: 1 KiB contiguous, nontemporal");
    run(probe<6>, "6: 16 rows x 64 B, nontemporal");
    {   // more waves per CU: 4 workgroups of 512 threads per CU (32 waves)
        for (int rep = 0; rep < 2; ++rep) {
            CHECK(hipEventRecord(a));
            hipLaunchKernelGGL(probe<3>, dim3(1024), dim3(512), 0, 0, C, tiles);
            CHECK(hipEventRecord(b));
            CHECK(hipEventSynchronize(b));
            float ms;
            CHECK(hipEventElapsedTime(&ms, a, b));
            if (rep) printf("3 with 4 workgroups per CU (32 waves)        %8.1f us  %6.2f TB/s\n", ms * 1e3, (double)tiles * 160 * 640 / ms / 1e9);
        }
        for (int rep = 0; rep < 2; ++rep) {
            CHECK(hipEventRecord(a));
            hipLaunchKernelGGL(probe<0>, dim3(1024), dim3(512), 0, 0, C, tiles);
            CHECK(hipEventRecord(b));
            CHECK(hipEventSynchronize(b));
            float ms;
            CHECK(hipEventElapsedTime(&ms, a, b));
            if (rep) printf("0 with 4 workgroups per CU (32 waves)        %8.1f us  %6.2f TB/s\n", ms * 1e3, (double)tiles * 160 * 640 / ms / 1e9);
        }
    }
    return 0;
}
