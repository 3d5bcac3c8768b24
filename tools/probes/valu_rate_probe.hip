// Issue-rate probe (run through gpurun): cycles per wave64 instruction for v_exp_f32, v_fma_f32, v_pk_fma_f32, v_max3_f32,
// v_cvt_pk_f16_f32 on one wave per SIMD and on four (does the rate depend on co-resident waves?).
// Build: hipcc -O3 --offload-arch=gfx950 tools/probes/valu_rate_probe.hip -o tools/probes/valu_rate_probe
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int OP>
__global__ void probe(float* out, long long* cyc, int iters) {
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = 0.001f * (threadIdx.x + i);
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
            if (OP == 1) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a[i]));
            if (OP == 3) asm volatile("v_max3_f32 %0, %0, %0, %0" : "+v"(a[i]));
            if (OP == 5) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
        }
        if (OP == 2) {
            typedef float f2 __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int i = 0; i < 8; i += 2) {
                f2 v = {a[i], a[i + 1]};
                asm volatile("v_pk_fma_f32 %0, %0, %0, %0\n\tv_pk_fma_f32 %0, %0, %0, %0" : "+v"(v));
                a[i] = v[0]; a[i + 1] = v[1];
            }
        }
        if (OP == 4) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { unsigned r; asm volatile("v_cvt_pk_f16_f32 %0, %1, %1" : "=v"(r) : "v"(a[i])); a[i] = __uint_as_float(r | 0x3f000000u); }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 1 << 24); hipMalloc(&cyc, 8);
    const char* names[6] = {"v_exp_f32", "v_fma_f32", "v_pk_fma_f32", "v_max3_f32", "v_cvt_pk_f16_f32 (+v_or)", "v_rcp_f32"};
    const int iters = 20000;
    for (int waves = 1; waves <= 4; waves *= 4)
        for (int op = 0; op < 6; ++op) {
            // one workgroup per CU region: 256 * waves threads = `waves` waves per SIMD
            dim3 grid(256), block(256 * waves > 1024 ? 1024 : 256 * waves);
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            float ms = 0.f;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0, 0);
                switch (op) {
                    case 0: hipLaunchKernelGGL(probe<0>, grid, block, 0, 0, out, cyc, iters); break;
                    case 1: hipLaunchKernelGGL(probe<1>, grid, block, 0, 0, out, cyc, iters); break;
                    case 2: hipLaunchKernelGGL(probe<2>, grid, block, 0, 0, out, cyc, iters); break;
                    case 3: hipLaunchKernelGGL(probe<3>, grid, block, 0, 0, out, cyc, iters); break;
                    case 4: hipLaunchKernelGGL(probe<4>, grid, block, 0, 0, out, cyc, iters); break;
                    default: hipLaunchKernelGGL(probe<5>, grid, block, 0, 0, out, cyc, iters); break;
                }
                hipEventRecord(e1, 0);
                hipDeviceSynchronize();
                hipEventElapsedTime(&ms, e0, e1);
            }
            long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            const double per = (double)c / ((double)iters * 8);
            // wall: `waves` waves per SIMD each issue iters * 8 (x2 for the pk pair) instructions
            const double instr = (double)iters * 8 * (op == 2 ? 1.0 : 1.0) * waves;
            printf("%d wave(s)/SIMD  %-26s %6.2f counter ticks / instr of one wave | %6.2f ns per SIMD instruction slot (wall %.3f ms)\n", waves,
                   names[op], per, ms * 1e6 / instr, ms);
        }
    return 0;
}
