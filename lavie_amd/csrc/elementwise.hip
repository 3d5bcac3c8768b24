// Small kernels around the contraction core: time embedding, the two 4-channel convolutions at the
// NCFHW boundary, the CFG + DDPM update, and the one-off weight repacking.
#include <math.h>

#include "common.h"
#include "ops.h"

namespace lavie {

// ------------------------------------------------------------------ time embedding
// Timesteps(dim, flip_sin_to_cos=True, freq_shift=0) (unet.py:153,428): [cos(t w_k) | sin(t w_k)].
__global__ void timestep_sinusoid_kernel(const float* __restrict__ t, float* __restrict__ out, int B, int dim) {
    const int half_dim = dim >> 1;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * half_dim) return;
    const int b = i / half_dim, k = i - b * half_dim;
    const float w = expf(-logf(10000.0f) * (float)k / (float)half_dim);
    const float a = t[b] * w;
    out[(size_t)b * dim + k] = cosf(a);
    out[(size_t)b * dim + half_dim + k] = sinf(a);
}

int launch_timestep_sinusoid(const float* t, float* out, int B, int dim, hipStream_t stream) {
    const int n = B * (dim / 2);
    hipLaunchKernelGGL(timestep_sinusoid_kernel, dim3(cdiv(n, 256)), dim3(256), 0, stream, t, out, B, dim);
    LAVIE_HIP(hipGetLastError());
    return 0;
}

// Batched GEMV for the [B <= 8, K] time-embedding vectors.
// Serves TimestepEmbedding (unet.py:434) and all ResnetBlock3D.time_emb_proj at once (resnet.py:186).
// HBM-bound on the weight matrix (19840 x 1280 halfs = 50 MB for the stacked projections): the activated input vectors
// are staged once per workgroup in LDS (fp32), each wave then streams GEMV_ROWS weight rows at a time with 16-byte loads
// (four independent loads in flight per lane) and reduces across the wave.
constexpr int GEMV_MAXB = 8;
constexpr int GEMV_ROWS = 4;           // weight rows per wave and pass
constexpr int GEMV_PASSES = 2;         // passes per wave: a workgroup covers 4 waves x 4 rows x 2 = 32 output features
__global__ __launch_bounds__(256) void gemv_kernel(const float* __restrict__ in, const half_t* __restrict__ W,
                                                  const float* __restrict__ bias, float* __restrict__ out, int B, int N,
                                                  int K, int act_in, int act_out) {
    extern __shared__ __attribute__((aligned(16))) char gemv_smem[];
    float* sx = reinterpret_cast<float*>(gemv_smem);            // [B][K]
    for (int i = threadIdx.x; i < B * K; i += 256) {
        float x = in[i];
        if (act_in) x = silu_f(x);
        sx[i] = x;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
#pragma unroll 1
    for (int pass = 0; pass < GEMV_PASSES; ++pass) {
        const int n0 = (blockIdx.x * GEMV_PASSES + pass) * (4 * GEMV_ROWS) + wave * GEMV_ROWS;
        if (n0 >= N) return;
        float acc[GEMV_ROWS][GEMV_MAXB];
#pragma unroll
        for (int r = 0; r < GEMV_ROWS; ++r)
#pragma unroll
            for (int b = 0; b < GEMV_MAXB; ++b) acc[r][b] = 0.f;
        for (int k = lane * 8; k < K; k += 64 * 8) {
            half8_t w[GEMV_ROWS];
#pragma unroll
            for (int r = 0; r < GEMV_ROWS; ++r) {
                const int n = n0 + r < N ? n0 + r : N - 1;
                w[r] = *reinterpret_cast<const half8_t*>(W + (size_t)n * K + k);
            }
#pragma unroll
            for (int b = 0; b < GEMV_MAXB; ++b) {
                if (b < B) {
                    const f32x4 x0 = *reinterpret_cast<const f32x4*>(sx + b * K + k);
                    const f32x4 x1 = *reinterpret_cast<const f32x4*>(sx + b * K + k + 4);
#pragma unroll
                    for (int r = 0; r < GEMV_ROWS; ++r) {
                        // explicit fma chain in a fixed order: every batch entry must round identically (identical CFG
                        // halves give bit-identical outputs; a contracted sum-of-products expression did not guarantee it)
                        float a = acc[r][b];
#pragma unroll
                        for (int j = 0; j < 4; ++j) a = __builtin_fmaf(x0[j], (float)w[r][j], a);
#pragma unroll
                        for (int j = 0; j < 4; ++j) a = __builtin_fmaf(x1[j], (float)w[r][4 + j], a);
                        acc[r][b] = a;
                    }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < GEMV_ROWS; ++r) {
#pragma unroll
            for (int b = 0; b < GEMV_MAXB; ++b) {
                if (b < B) {
                    float v = wave_sum(acc[r][b]);
                    if (lane == 0 && n0 + r < N) {
                        v += bias ? bias[n0 + r] : 0.f;
                        if (act_out) v = silu_f(v);
                        out[(size_t)b * N + n0 + r] = v;
                    }
                }
            }
        }
    }
}

int launch_gemv(const float* in, const half_t* W, const float* bias, float* out, int B, int N, int K, int act_in,
                int act_out, hipStream_t stream) {
    LAVIE_CHECK(B >= 1 && B <= GEMV_MAXB && K % 8 == 0, "gemv: unsupported B=%d K=%d", B, K);
    const size_t lds = (size_t)B * K * sizeof(float);
    LAVIE_CHECK(lds <= 64 * 1024, "gemv: B * K = %d does not fit the LDS staging buffer", B * K);
    hipLaunchKernelGGL(gemv_kernel, dim3(cdiv(N, 4 * GEMV_ROWS * GEMV_PASSES)), dim3(256), lds, stream, in, W, bias, out, B, N, K,
                       act_in, act_out);
    LAVIE_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ conv_in (unet.py:150,454)
// Reads the caller's NCFHW latent, writes channels-last activations.  K = 9*Cin is tiny (36), so a
// direct kernel: one thread per (pixel, 8 output channels), weights in LDS as [tap*Cin + ci][Cout].
__global__ __launch_bounds__(256) void conv_in_kernel(const half_t* __restrict__ x, const half_t* __restrict__ wp,
                                                     const float* __restrict__ bias, half_t* __restrict__ y, int B,
                                                     int Cin, int F, int H, int W, int Cout) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half_t* sw = reinterpret_cast<half_t*>(smem);               // [(tap * Cin + ci) / 2][Cout][2]: K pairs interleaved
    const int kk = 9 * Cin;
    for (int i = threadIdx.x * 8; i < kk * Cout; i += 256 * 8)
        *reinterpret_cast<half8_t*>(sw + i) = *reinterpret_cast<const half8_t*>(wp + i);
    __syncthreads();
    const int ng = Cout >> 3;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long M = (long)B * F * H * W;
    if (idx >= M * ng) return;
    const long m = idx / ng;
    const int g = (int)(idx - m * ng);
    const int xw = (int)(m % W);
    const int yh = (int)((m / W) % H);
    const int f = (int)((m / ((long)W * H)) % F);
    const int b = (int)(m / ((long)W * H * F));
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = bias[g * 8 + j];
    // the kernel is VALU-issue bound (K = 36 per output): v_dot2_f32_f16 takes two input channels per slot
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = yh + ky - 1;
        if ((unsigned)iy >= (unsigned)H) continue;
        for (int kx = 0; kx < 3; ++kx) {
            const int ix = xw + kx - 1;
            if ((unsigned)ix >= (unsigned)W) continue;
            for (int ci = 0; ci < Cin; ci += 2) {
                const half2_t v = {x[((((size_t)b * Cin + ci) * F + f) * H + iy) * W + ix],
                                   x[((((size_t)b * Cin + ci + 1) * F + f) * H + iy) * W + ix]};
                const half_t* wr = sw + ((size_t)(((ky * 3 + kx) * Cin + ci) >> 1) * Cout + g * 8) * 2;
                const half8_t w0 = *reinterpret_cast<const half8_t*>(wr);
                const half8_t w1 = *reinterpret_cast<const half8_t*>(wr + 8);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[j] = __builtin_amdgcn_fdot2(v, (half2_t){w0[2 * j], w0[2 * j + 1]}, acc[j], false);
                    acc[j + 4] = __builtin_amdgcn_fdot2(v, (half2_t){w1[2 * j], w1[2 * j + 1]}, acc[j + 4], false);
                }
            }
        }
    }
    half8_t o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (half_t)acc[j];
    *reinterpret_cast<half8_t*>(y + m * Cout + g * 8) = o;
}

int launch_conv_in(const half_t* x, const half_t* wp, const float* bias, half_t* y, int B, int Cin, int F, int H, int W,
                   int Cout, hipStream_t stream) {
    LAVIE_CHECK(Cout % 8 == 0 && Cin % 2 == 0, "conv_in: Cout must be a multiple of 8 and Cin even (Cout=%d Cin=%d)", Cout, Cin);
    const size_t lds = (size_t)9 * Cin * Cout * sizeof(half_t);
    LAVIE_CHECK(lds <= 64 * 1024, "conv_in: weights do not fit LDS (%zu B)", lds);
    const long total = (long)B * F * H * W * (Cout / 8);
    hipLaunchKernelGGL(conv_in_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), lds, stream, x, wp, bias, y, B,
                       Cin, F, H, W, Cout);
    LAVIE_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ conv_out (unet.py:290,506)
// channels-last [M, Cin] -> NCFHW [B, Cout<=8, F, H, W].  One wave per output pixel, K = 9*Cin split
// over the lanes as 16-byte vectors; weights [Cout][tap][Cin] in LDS.
constexpr int CONV_OUT_MAXC = 8;
__global__ __launch_bounds__(256) void conv_out_kernel(const half_t* __restrict__ x, const half_t* __restrict__ wp,
                                                      const float* __restrict__ bias, half_t* __restrict__ y, int B,
                                                      int Cin, int F, int H, int W, int Cout) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half_t* sw = reinterpret_cast<half_t*>(smem);
    const int kk = 9 * Cin;
    for (int i = threadIdx.x * 8; i < kk * Cout; i += 256 * 8)
        *reinterpret_cast<half8_t*>(sw + i) = *reinterpret_cast<const half8_t*>(wp + i);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const long m = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long M = (long)B * F * H * W;
    if (m >= M) return;
    const int xw = (int)(m % W);
    const int yh = (int)((m / W) % H);
    const long img = m / ((long)W * H);          // b * F + f
    const int nvec = Cin >> 3;
    float acc[CONV_OUT_MAXC];
#pragma unroll
    for (int c = 0; c < CONV_OUT_MAXC; ++c) acc[c] = 0.f;
    for (int i = lane; i < 9 * nvec; i += 64) {
        const int tap = i / nvec, vec = i - tap * nvec;
        const int iy = yh + tap / 3 - 1, ix = xw + tap % 3 - 1;
        if ((unsigned)iy >= (unsigned)H || (unsigned)ix >= (unsigned)W) continue;
        const half8_t v = *reinterpret_cast<const half8_t*>(x + ((img * H + iy) * W + ix) * Cin + vec * 8);
        // v_dot2_f32_f16: two multiply-adds per VALU slot, fp32 accumulation (the kernel is VALU-issue bound, not HBM bound)
#pragma unroll
        for (int c = 0; c < CONV_OUT_MAXC; ++c) {
            if (c < Cout) {
                const half8_t w = *reinterpret_cast<const half8_t*>(sw + (size_t)c * kk + tap * Cin + vec * 8);
#pragma unroll
                for (int j = 0; j < 8; j += 2)
                    acc[c] = __builtin_amdgcn_fdot2((half2_t){v[j], v[j + 1]}, (half2_t){w[j], w[j + 1]}, acc[c], false);
            }
        }
    }
    const int f = (int)(img % F);
    const int b = (int)(img / F);
#pragma unroll
    for (int c = 0; c < CONV_OUT_MAXC; ++c) {
        if (c < Cout) {
            const float s = wave_sum(acc[c]) + bias[c];
            if (lane == 0) y[((((size_t)b * Cout + c) * F + f) * H + yh) * W + xw] = (half_t)s;
        }
    }
}

// Register-weight variant for Cout = 4 (every model here: 4 latent channels): a lane keeps the weights of ITS (tap, 8-channel
// vector) slots for the four outputs in registers (NV slots x 4 outputs x 16 B) and walks CONV_OUT4_PPW consecutive pixels, so the
// loop has no LDS traffic at all: per pixel NV independent 16-byte loads, 16 NV v_dot2, four wave reductions.  Same per-lane
// accumulation order as the kernel above (slot lane, lane + 64, ...), then the same wave_sum: bit-identical results.
constexpr int CONV_OUT4_PPW = 16;
template <int NV>
__global__ __launch_bounds__(256) void conv_out4_kernel(const half_t* __restrict__ x, const half_t* __restrict__ wp,
                                                       const float* __restrict__ bias, half_t* __restrict__ y, int B,
                                                       int Cin, int F, int H, int W) {
    const int lane = threadIdx.x & 63;
    const long M = (long)B * F * H * W;
    const int nvec = Cin >> 3, kk = 9 * Cin;
    half8_t wr[NV][4];
    int dy[NV], dx[NV], co[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int i = lane + 64 * j;
        const bool have = i < 9 * nvec;
        const int tap = have ? i / nvec : 4, vec = have ? i - tap * nvec : 0;
        dy[j] = have ? tap / 3 - 1 : 1 << 20;       // a slot past the stencil never passes the bounds test
        dx[j] = tap % 3 - 1;
        co[j] = vec * 8;
#pragma unroll
        for (int c = 0; c < 4; ++c)
            wr[j][c] = have ? *reinterpret_cast<const half8_t*>(wp + (size_t)c * kk + tap * Cin + vec * 8) : (half8_t){0, 0, 0, 0, 0, 0, 0, 0};
    }
    const float b0 = bias[0], b1 = bias[1], b2 = bias[2], b3 = bias[3];
    const long m_begin = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * CONV_OUT4_PPW;
    for (int pp = 0; pp < CONV_OUT4_PPW; ++pp) {
        const long m = m_begin + pp;
        if (m >= M) return;
        const int xw = (int)(m % W);
        const int yh = (int)((m / W) % H);
        const long img = m / ((long)W * H);          // b * F + f
        half8_t v[NV];
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int iy = yh + dy[j], ix = xw + dx[j];
            const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
            const int iyc = ok ? iy : yh, ixc = ok ? ix : xw;           // clamped address, value discarded: no divergent load
            v[j] = *reinterpret_cast<const half8_t*>(x + ((img * H + iyc) * W + ixc) * Cin + co[j]);
            if (!ok) v[j] = (half8_t){0, 0, 0, 0, 0, 0, 0, 0};
        }
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NV; ++j)
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int e = 0; e < 8; e += 2)
                    acc[c] = __builtin_amdgcn_fdot2((half2_t){v[j][e], v[j][e + 1]}, (half2_t){wr[j][c][e], wr[j][c][e + 1]}, acc[c], false);
        const float s0 = wave_sum(acc[0]) + b0, s1 = wave_sum(acc[1]) + b1, s2 = wave_sum(acc[2]) + b2, s3 = wave_sum(acc[3]) + b3;
        if (lane == 0) {
            const int f = (int)(img % F);
            const int b = (int)(img / F);
            const size_t plane = (size_t)F * H * W;
            half_t* dst = y + (((size_t)b * 4) * F + f) * H * W + (size_t)yh * W + xw;
            dst[0] = (half_t)s0;
            dst[plane] = (half_t)s1;
            dst[2 * plane] = (half_t)s2;
            dst[3 * plane] = (half_t)s3;
        }
    }
}

int launch_conv_out(const half_t* x, const half_t* wp, const float* bias, half_t* y, int B, int Cin, int F, int H, int W,
                    int Cout, hipStream_t stream) {
    LAVIE_CHECK(Cout <= CONV_OUT_MAXC && Cin % 8 == 0, "conv_out: unsupported Cout=%d Cin=%d", Cout, Cin);
    const size_t lds = (size_t)9 * Cin * Cout * sizeof(half_t);
    LAVIE_CHECK(lds <= 64 * 1024, "conv_out: weights do not fit LDS (%zu B)", lds);
    const long M = (long)B * F * H * W;
    const int slots = 9 * (Cin / 8);
    if (Cout == 4 && slots <= 64 * 6) {            // Cin <= 336 (base 320, VSR 256): the register-weight kernel
        const unsigned grid = (unsigned)((M + 4 * CONV_OUT4_PPW - 1) / (4 * CONV_OUT4_PPW));
        if (slots <= 64 * 5) hipLaunchKernelGGL(conv_out4_kernel<5>, dim3(grid), dim3(256), 0, stream, x, wp, bias, y, B, Cin, F, H, W);
        else hipLaunchKernelGGL(conv_out4_kernel<6>, dim3(grid), dim3(256), 0, stream, x, wp, bias, y, B, Cin, F, H, W);
        LAVIE_HIP(hipGetLastError());
        return 0;
    }
    hipLaunchKernelGGL(conv_out_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), lds, stream, x, wp, bias, y, B, Cin, F,
                       H, W, Cout);
    LAVIE_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ class embedding (VSR noise level)
// emb[b, :] = silu(emb[b, :] + table[label_b, :]) (vsr/models/unet.py:494-505; the SiLU is the one every consumer of emb
// applies first, resnet.py:186).  Labels travel by value (B <= 8).
struct ClassLabels { int v[8]; };
__global__ void add_class_emb_silu_kernel(float* __restrict__ emb, const half_t* __restrict__ table, ClassLabels lab, int B, int N) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * N) return;
    const int b = i / N, n = i - b * N;
    const float x = emb[i] + (float)table[(size_t)lab.v[b] * N + n];
    emb[i] = silu_f(x);
}

int launch_add_class_emb_silu(float* emb, const half_t* table, const int* labels_host, int B, int N, hipStream_t stream) {
    LAVIE_CHECK(B >= 1 && B <= 8, "class embedding: batch %d unsupported (1..8)", B);
    ClassLabels lab;
    for (int b = 0; b < 8; ++b) lab.v[b] = b < B ? labels_host[b] : 0;
    hipLaunchKernelGGL(add_class_emb_silu_kernel, dim3((unsigned)((B * N + 255) / 256)), dim3(256), 0, stream, emb, table, lab, B, N);
    LAVIE_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ CFG + DDPM step
// pipeline_videogen.py:679-683 with the scheduler arithmetic of oracle/ddpm.py (diffusers DDPMScheduler.step).
// CFG = false: guidance_scale <= 1, the pipeline's `do_classifier_free_guidance == False` branch (:626, 666, 678) — one
// model batch entry per latent, eps used as it is, a single fp16 copy of the next model input.
template <bool CFG>
__global__ void sampler_step_kernel(const half_t* __restrict__ eps2, float* __restrict__ x,
                                    const float* __restrict__ noise, half_t* __restrict__ model_in2, long n,
                                    float guidance, float kx, float ke, float c0, float ct, float sigma, float in_scale) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float eps = (float)eps2[i];
    if (CFG) {
        const float ec = (float)eps2[n + i];
        eps = eps + guidance * (ec - eps);
    }
    const float xt = x[i];
    const float x0 = kx * xt - ke * eps;
    float xn = c0 * x0 + ct * xt;
    if (sigma != 0.f) xn += sigma * noise[i];
    x[i] = xn;
    // scheduler.scale_model_input of the NEXT step (Euler: 1 / sqrt(sigma_next^2 + 1); DDPM / DDIM: 1)
    const half_t h = (half_t)(xn * in_scale);
    model_in2[i] = h;
    if (CFG) model_in2[n + i] = h;
}

int launch_cfg_ddpm_step(const half_t* eps2, float* x, const float* noise, half_t* model_in2, int64_t n, float guidance,
                         float kx, float ke, float c0, float ct, float sigma, float in_scale, hipStream_t stream) {
    LAVIE_CHECK(sigma == 0.f || noise != nullptr, "ddpm step: sigma != 0 needs a noise tensor");
    hipLaunchKernelGGL(sampler_step_kernel<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, eps2, x, noise,
                       model_in2, (long)n, guidance, kx, ke, c0, ct, sigma, in_scale);
    LAVIE_HIP(hipGetLastError());
    return 0;
}

int launch_sampler_step(const half_t* eps, float* x, const float* noise, half_t* model_in, int64_t n, float kx, float ke,
                        float c0, float ct, float sigma, float in_scale, hipStream_t stream) {
    LAVIE_CHECK(sigma == 0.f || noise != nullptr, "sampler step: sigma != 0 needs a noise tensor");
    hipLaunchKernelGGL(sampler_step_kernel<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, eps, x, noise,
                       model_in, (long)n, 1.0f, kx, ke, c0, ct, sigma, in_scale);
    LAVIE_HIP(hipGetLastError());
    return 0;
}

template <bool DUP>
__global__ void f32_to_f16_kernel(const float* __restrict__ x, half_t* __restrict__ out2, long n, float in_scale) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const half_t h = (half_t)(x[i] * in_scale);
    out2[i] = h;
    if (DUP) out2[n + i] = h;
}

int launch_f32_to_f16_dup2(const float* x, half_t* out2, int64_t n, float in_scale, hipStream_t stream) {
    hipLaunchKernelGGL(f32_to_f16_kernel<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, x, out2, (long)n,
                       in_scale);
    LAVIE_HIP(hipGetLastError());
    return 0;
}

int launch_f32_to_f16_scaled(const float* x, half_t* out, int64_t n, float in_scale, hipStream_t stream) {
    hipLaunchKernelGGL(f32_to_f16_kernel<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, x, out, (long)n,
                       in_scale);
    LAVIE_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ relative-position bias
// RelativePositionBias (attention.py:669-707): out[h, i, j] = emb[bucket(i, j), h].
void relpos_bucket_table(int F, int num_buckets, int max_distance, int* out) {
    const int half_b = num_buckets / 2;
    const int exact = half_b / 2;
    for (int i = 0; i < F; ++i) {
        for (int j = 0; j < F; ++j) {
            int n = i - j;                         // -(k_pos - q_pos)
            int b = n < 0 ? half_b : 0;            // key in the future
            n = n < 0 ? -n : n;
            if (n < exact) {
                b += n;
            } else {
                // torch evaluates this in fp32; for every n < 128 that equals the exact value (tests/test_host_logic.py)
                int v = exact + (int)floor(log((double)n / exact) / log((double)max_distance / exact) * (half_b - exact) + 1e-9);
                b += v < half_b - 1 ? v : half_b - 1;
            }
            out[i * F + j] = b;
        }
    }
}

__global__ void fill_relpos_bias_kernel(const half_t* __restrict__ emb, const int* __restrict__ buckets,
                                        float* __restrict__ out, int heads, int F) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= heads * F * F) return;
    const int h = i / (F * F), ij = i - h * F * F;
    out[i] = (float)emb[buckets[ij] * heads + h];
}

int launch_fill_relpos_bias(const half_t* emb, const int* buckets, float* out, int heads, int F, hipStream_t stream) {
    const int n = heads * F * F;
    hipLaunchKernelGGL(fill_relpos_bias_kernel, dim3(cdiv(n, 256)), dim3(256), 0, stream, emb, buckets, out, heads, F);
    LAVIE_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ weight repacking (load time)
// [Cout][Cin][3][3] -> out[co * ld_out + col0 + k], k = ((ci/64)*9 + tap)*64 + ci%64 (chunked: slab-major,
// tap-minor, the implicit GEMM's K order) or k = tap*Cin + ci (conv_out)
// `taps` = 9 for the 3x3 convs, 3 / 5 for nn.Conv3d (T, 1, 1) weights [Cout][Cin][T][1][1] (same memory order)
__global__ void pack_conv3x3_kernel(const half_t* __restrict__ w, half_t* __restrict__ out, int Cout, int Cin, int ld_out,
                                    int col0, int chunked, int taps) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)Cout * Cin * taps;
    if (i >= total) return;
    const int ci = (int)(i % Cin);
    const int tap = (int)((i / Cin) % taps);
    const int co = (int)(i / ((long)Cin * taps));
    const int k = chunked ? ((ci >> 6) * taps + tap) * 64 + (ci & 63) : tap * Cin + ci;
    out[(size_t)co * ld_out + col0 + k] = w[((size_t)co * Cin + ci) * taps + tap];
}

int launch_pack_conv_taps(const half_t* w, half_t* out, int Cout, int Cin, int taps, int ld_out, int col0, bool chunked,
                          hipStream_t stream) {
    LAVIE_CHECK(!chunked || Cin % 64 == 0, "pack_conv: Cin=%d must be a multiple of 64", Cin);
    const long total = (long)Cout * Cin * taps;
    hipLaunchKernelGGL(pack_conv3x3_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, w, out, Cout, Cin,
                       ld_out, col0, chunked ? 1 : 0, taps);
    LAVIE_HIP(hipGetLastError());
    return 0;
}

// Parity weights of the 3x3 conv of a nearest-x2 upsampled image (igemm_patch.hip MODE 3): [Cout][Cin][3][3] ->
// out[par][co][((ci / 64) * 4 + a * 2 + b) * 64 + ci % 64], par = py * 2 + px, (a, b) = the 2x2 source taps; the value is the fp32 sum
// of the 3x3 taps (dy, dx) that land on that source pixel: parity 0: tap 0 <- {0}, tap 1 <- {1, 2}; parity 1: tap 0 <- {0, 1},
// tap 1 <- {2} (rows and columns alike), rounded once to fp16.
__global__ void pack_conv3x3_parity_kernel(const half_t* __restrict__ w, half_t* __restrict__ out, int Cout, int Cin) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long per = (long)Cout * Cin * 4;
    if (i >= 4 * per) return;
    const int par = (int)(i / per);
    const long j = i - par * per;
    const int ci = (int)(j % Cin);
    const int tap = (int)((j / Cin) % 4);
    const int co = (int)(j / ((long)Cin * 4));
    const int py = par >> 1, px = par & 1, a = tap >> 1, b = tap & 1;
    const int dy0 = py == 0 ? (a == 0 ? 0 : 1) : (a == 0 ? 0 : 2), dy1 = py == 0 ? (a == 0 ? 0 : 2) : (a == 0 ? 1 : 2);
    const int dx0 = px == 0 ? (b == 0 ? 0 : 1) : (b == 0 ? 0 : 2), dx1 = px == 0 ? (b == 0 ? 0 : 2) : (b == 0 ? 1 : 2);
    const half_t* wk = w + ((size_t)co * Cin + ci) * 9;
    float s = 0.f;
    for (int dy = dy0; dy <= dy1; ++dy)
        for (int dx = dx0; dx <= dx1; ++dx) s += (float)wk[dy * 3 + dx];
    out[((size_t)par * Cout + co) * (4 * (size_t)Cin) + ((ci >> 6) * 4 + tap) * 64 + (ci & 63)] = (half_t)s;
}

int launch_pack_conv3x3_parity(const half_t* w, half_t* out, int Cout, int Cin, hipStream_t stream) {
    LAVIE_CHECK(Cin % 64 == 0, "pack_conv_parity: Cin=%d must be a multiple of 64", Cin);
    const long total = 4L * Cout * Cin * 4;
    hipLaunchKernelGGL(pack_conv3x3_parity_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, w, out, Cout, Cin);
    LAVIE_HIP(hipGetLastError());
    return 0;
}

int launch_pack_conv3x3(const half_t* w, half_t* out, int Cout, int Cin, int ld_out, int col0, bool chunked,
                        hipStream_t stream) {
    return launch_pack_conv_taps(w, out, Cout, Cin, 9, ld_out, col0, chunked, stream);
}

__global__ void copy_rows_kernel(const half_t* __restrict__ src, int ld_src, half_t* __restrict__ dst, int ld_dst,
                                 int rows, int cols, int col0) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)rows * cols) return;
    const int r = (int)(i / cols), c = (int)(i - (long)r * cols);
    dst[(size_t)r * ld_dst + col0 + c] = src[(size_t)r * ld_src + c];
}

int launch_copy_rows(const half_t* src, int ld_src, half_t* dst, int ld_dst, int rows, int cols, int col0,
                     hipStream_t stream) {
    const long total = (long)rows * cols;
    hipLaunchKernelGGL(copy_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, src, ld_src, dst,
                       ld_dst, rows, cols, col0);
    LAVIE_HIP(hipGetLastError());
    return 0;
}

// GEGLU projection rows [value(0..N/2) ; gate(N/2..N)] -> 16-row blocks alternating value / gate, so that
// the igemm epilogue finds h and its gate in the same lane (igemm.hip, EPI_GEGLU).
__device__ __forceinline__ int geglu_src_row(int n, int N) {
    const int j = n >> 5, i = n & 31;
    return i < 16 ? 16 * j + i : N / 2 + 16 * j + (i - 16);
}

__global__ void pack_geglu_rows_kernel(const half_t* __restrict__ w, half_t* __restrict__ out, int N, int K) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)N * K) return;
    const int n = (int)(i / K), k = (int)(i - (long)n * K);
    out[i] = w[(size_t)geglu_src_row(n, N) * K + k];
}

int launch_pack_geglu_rows(const half_t* w, half_t* out, int N, int K, hipStream_t stream) {
    LAVIE_CHECK(N % 32 == 0, "geglu pack: N must be a multiple of 32");
    const long total = (long)N * K;
    hipLaunchKernelGGL(pack_geglu_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, w, out, N, K);
    LAVIE_HIP(hipGetLastError());
    return 0;
}

__global__ void pack_geglu_bias_kernel(const half_t* __restrict__ b, float* __restrict__ out, int N) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n < N) out[n] = (float)b[geglu_src_row(n, N)];
}

int launch_pack_geglu_bias(const half_t* b, float* out, int N, hipStream_t stream) {
    hipLaunchKernelGGL(pack_geglu_bias_kernel, dim3(cdiv(N, 256)), dim3(256), 0, stream, b, out, N);
    LAVIE_HIP(hipGetLastError());
    return 0;
}

__global__ void f16_to_f32_kernel(const half_t* __restrict__ a, const half_t* __restrict__ b, float* __restrict__ dst, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = (float)a[i] + (b ? (float)b[i] : 0.f);
}

int launch_f16_to_f32(const half_t* src, float* dst, int64_t n, hipStream_t stream) {
    hipLaunchKernelGGL(f16_to_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, src,
                       (const half_t*)nullptr, dst, (long)n);
    LAVIE_HIP(hipGetLastError());
    return 0;
}

int launch_add_f16_to_f32(const half_t* a, const half_t* b, float* dst, int64_t n, hipStream_t stream) {
    hipLaunchKernelGGL(f16_to_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, a, b, dst, (long)n);
    LAVIE_HIP(hipGetLastError());
    return 0;
}

// LayerNorm folding: W'[n, k] = fp16(W[n, k] * gamma[k]);  s[n] = sum_k W'[n, k];
// b'[n] = sum_k beta[k] * W[n, k] (+ bias[n]).  One wave per output row.
__global__ __launch_bounds__(256) void ln_fold_kernel(const half_t* __restrict__ W, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, const half_t* __restrict__ bias,
                                                     half_t* __restrict__ Wout, float* __restrict__ s_out,
                                                     float* __restrict__ b_out, int N, int K) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    float s = 0.f, b = 0.f;
    for (int k = lane; k < K; k += 64) {
        const float w = (float)W[(size_t)n * K + k];
        const half_t wf = (half_t)(w * gamma[k]);
        Wout[(size_t)n * K + k] = wf;
        s += (float)wf;
        b += beta[k] * w;
    }
    s = wave_sum(s);
    b = wave_sum(b);
    if (lane == 0) {
        s_out[n] = s;
        b_out[n] = b + (bias ? (float)bias[n] : 0.f);
    }
}

int launch_ln_fold(const half_t* W, const float* gamma, const float* beta, const half_t* bias, half_t* Wout, float* s_out,
                   float* b_out, int N, int K, hipStream_t stream) {
    hipLaunchKernelGGL(ln_fold_kernel, dim3(cdiv(N, 4)), dim3(256), 0, stream, W, gamma, beta, bias, Wout, s_out, b_out, N, K);
    LAVIE_HIP(hipGetLastError());
    return 0;
}

__global__ void pack_geglu_vec_kernel(const float* __restrict__ in, float* __restrict__ out, int N) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n < N) out[n] = in[geglu_src_row(n, N)];
}

int launch_pack_geglu_vec(const float* in, float* out, int N, hipStream_t stream) {
    hipLaunchKernelGGL(pack_geglu_vec_kernel, dim3(cdiv(N, 256)), dim3(256), 0, stream, in, out, N);
    LAVIE_HIP(hipGetLastError());
    return 0;
}

// conv_in weights [Cout][Cin][3][3] -> [k / 2][Cout][k & 1] with k = (ky*3+kx)*Cin + ci (K pairs interleaved for v_dot2)
__global__ void pack_conv_in_kernel(const half_t* __restrict__ w, half_t* __restrict__ out, int Cout, int Cin) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Cout * Cin * 9) return;
    const int tap = i % 9, ci = (i / 9) % Cin, co = i / (9 * Cin);
    const int k = tap * Cin + ci;
    out[((size_t)(k >> 1) * Cout + co) * 2 + (k & 1)] = w[i];
}

int launch_pack_conv_in(const half_t* w, half_t* out, int Cout, int Cin, hipStream_t stream) {
    hipLaunchKernelGGL(pack_conv_in_kernel, dim3(cdiv(Cout * Cin * 9, 256)), dim3(256), 0, stream, w, out, Cout, Cin);
    LAVIE_HIP(hipGetLastError());
    return 0;
}

}  // namespace lavie
