// Shared device/host helpers for the gfx950 kernels.  CDNA4 only: wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 half_t;
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __fp16 fp16x4_raw __attribute__((__vector_size__(4 * sizeof(__fp16))));

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

namespace lavie {

// Thread-local error text for lavie_last_error().
void set_error(const char* fmt, ...);
const char* get_error();
// kernel-selection debug switches (lavie_debug_*): a process-wide epoch the captured-graph key includes
void bump_debug_epoch();
unsigned long debug_epoch();
void set_fused_mask(int mask);     // which row-resident fused sub-block kernels the engine uses (default: all)
int fused_mask();

#define LAVIE_CHECK(cond, ...)                 \
    do {                                       \
        if (!(cond)) {                         \
            ::lavie::set_error(__VA_ARGS__);   \
            return -1;                         \
        }                                      \
    } while (0)

#define LAVIE_HIP(expr)                                                            \
    do {                                                                           \
        hipError_t e_ = (expr);                                                    \
        if (e_ != hipSuccess) {                                                    \
            ::lavie::set_error("%s failed: %s", #expr, hipGetErrorString(e_));     \
            return -2;                                                             \
        }                                                                          \
    } while (0)

__host__ __device__ static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// hipFuncAttributeMaxDynamicSharedMemorySize once per kernel ADDRESS (different instantiations of one kernel template share a
// function-pointer type, so a per-type static flag is not enough): keeps the call off the launch path and out of stream captures.
int ensure_dynamic_lds(const void* kernel, int bytes);     // engine.cpp

#ifdef __HIPCC__
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
// exact-erf GELU (F.gelu default).  erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, far below fp16
// resolution) with the hardware exp2/rcp: ~14 VALU slots instead of the ~40 of the libm erff.
__device__ __forceinline__ float erf_as_f(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);
    float poly = 1.061405429f;
    poly = poly * t - 1.453152027f;
    poly = poly * t + 1.421413741f;
    poly = poly * t - 0.284496736f;
    poly = poly * t + 0.254829592f;
    const float e = __builtin_amdgcn_exp2f(-ax * ax * 1.4426950408889634f);
    const float r = 1.0f - poly * t * e;
    return copysignf(r, x);
}
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erf_as_f(x * 0.70710678118654752f)); }
#endif

}  // namespace lavie
