// extern "C" surface of liblavie_hip.so (declared in include/lavie_hip.h).
#include <string.h>

#include <new>

#include "engine.h"
#include "profile.h"

using namespace lavie;

struct lavie_unet_s {
    UNet net;
    explicit lavie_unet_s(const lavie_unet_config& c) : net(c) {}
};

static inline hipStream_t S(void* s) { return (hipStream_t)s; }

// Operator-level entry points have no workspace argument: the split-K slab comes from a grow-only scratch
// buffer owned by the library (allocated outside any stream capture; the engine uses its own workspace).
static int g_tap_major = 0;   // diagnostic K-order switch for the op-level conv (pack + launch)
static float* g_slab = nullptr;
static size_t g_slab_bytes = 0;
static int op_slab(IgemmParams& p, int epilogue, bool gather = false) {
    p.splits = gather ? igemm_plan_splits_gather(p) : igemm_plan_splits(p.M, p.N, p.nk, epilogue);
    p.slab = nullptr;
    if (p.splits <= 1) return 0;
    const size_t need = (size_t)p.splits * p.M * p.N * sizeof(float);
    if (need > g_slab_bytes) {
        if (g_slab) { LAVIE_HIP(hipDeviceSynchronize()); LAVIE_HIP(hipFree(g_slab)); g_slab = nullptr; g_slab_bytes = 0; }
        LAVIE_HIP(hipMalloc((void**)&g_slab, need));
        g_slab_bytes = need;
    }
    p.slab = g_slab;
    return 0;
}
static inline const half_t* H(const void* p) { return (const half_t*)p; }
static inline half_t* H(void* p) { return (half_t*)p; }

extern "C" {

const char* lavie_last_error(void) { return get_error(); }
int lavie_abi_version(void) { return LAVIE_ABI_VERSION; }

int lavie_linear_f16(const void* A, int lda, const void* W, const float* bias, const float* bias2, int ldb2,
                     int rows_per_batch, const void* R, int ldr, void* C, int ldc, int M, int N, int K, int geglu,
                     void* stream) {
    LAVIE_CHECK(A && W && C, "linear: null tensor");
    LAVIE_CHECK(K % IGEMM_BK == 0, "linear: K=%d must be a multiple of %d", K, IGEMM_BK);
    LAVIE_CHECK(!bias2 || rows_per_batch > 0, "linear: bias2 needs rows_per_batch > 0");
    IgemmParams p;
    memset(&p, 0, sizeof(p));
    p.A = H(A); p.lda = lda; p.W = H(W); p.ldw = K; p.C = H(C); p.ldc = ldc; p.bias = bias;
    p.bias2 = bias2; p.ldb2 = ldb2; p.rows_per_batch = rows_per_batch > 0 ? rows_per_batch : 1;
    p.R = H(R); p.ldr = ldr; p.M = M; p.N = N; p.nk = K / IGEMM_BK;
    if (int rc = op_slab(p, geglu ? EPI_GEGLU : EPI_LINEAR)) return rc;
    return launch_igemm(p, false, geglu ? EPI_GEGLU : EPI_LINEAR, S(stream));
}

// The consumer side of a folded LayerNorm at operator level (engine.cpp's `LnFold`): C = rstd_m (A W'^T - mean_m s) + bias with
// W' = W gamma, s = row sums of W', bias = W beta (+ b) prepared by the caller; stats [M, 2] = (mean, rstd) of the rows of A
int lavie_linear_lnfold_f16(const void* A, const void* Wf, const float* bias, const float* ln_s, const float* ln_stats, void* C,
                            int M, int N, int K, void* stream) {
    LAVIE_CHECK(A && Wf && C && ln_s && ln_stats, "linear_lnfold: null tensor");
    LAVIE_CHECK(K % IGEMM_BK == 0, "linear_lnfold: K=%d must be a multiple of %d", K, IGEMM_BK);
    IgemmParams p;
    memset(&p, 0, sizeof(p));
    p.A = H(A); p.lda = K; p.W = H(Wf); p.ldw = K; p.C = H(C); p.ldc = N; p.bias = bias; p.rows_per_batch = 1;
    p.M = M; p.N = N; p.nk = K / IGEMM_BK; p.splits = 1;
    p.ln_s = ln_s; p.ln_stats = ln_stats;
    return launch_igemm(p, false, EPI_LINEAR, S(stream));
}

long long lavie_geglu_mlp_image_bytes(int C) { return geglu_mlp_supported(C) ? (long long)geglu_mlp_image_bytes(C) : 0; }
long long lavie_geglu_mlp_bias_floats(int C) { return geglu_mlp_supported(C) ? (long long)geglu_mlp_bias_floats(C) : 0; }
int lavie_pack_geglu_mlp_f16(const void* w1, const void* b1_f16, const void* w2, int C, void* img, float* b1img, void* stream) {
    LAVIE_CHECK(w1 && b1_f16 && w2 && img && b1img, "pack_geglu_mlp: null tensor");
    return pack_geglu_mlp(H(w1), H(b1_f16), H(w2), C, (half_t*)img, b1img, S(stream));
}
int lavie_geglu_mlp_f16(const void* x, void* y, int M, int C, const void* img, const float* b1img, const float* gamma,
                        const float* beta, const float* b2, float eps, void* stream) {
    return launch_geglu_mlp(H(x), (half_t*)y, M, C, H(img), b1img, gamma, beta, b2, eps, S(stream));
}

long long lavie_temporal_block_image_bytes(int C, int heads, int F, int rot_dim) {
    return temporal_block_supported(C, heads, F, rot_dim) ? (long long)temporal_block_image_bytes(C) : 0;
}
int lavie_pack_temporal_block_f16(const void* wq, const void* wk, const void* wv, const void* wo, int C, void* img, void* stream) {
    LAVIE_CHECK(wq && wk && wv && wo && img, "pack_temporal_block: null tensor");
    return pack_temporal_block(H(wq), H(wk), H(wv), H(wo), C, (half_t*)img, S(stream));
}
int lavie_temporal_block_f16(const void* x, void* y, int B, int F, int D, int C, int heads, const void* img, const float* gamma,
                             const float* beta, const float* bo, const float* relbias, const float* rot_cos,
                             const float* rot_sin, int rot_dim, float scale, float eps, void* stream) {
    return launch_temporal_block(H(x), (half_t*)y, B, F, D, C, heads, H(img), gamma, beta, bo, relbias, rot_cos, rot_sin, rot_dim,
                                 scale, eps, S(stream));
}

long long lavie_cross_block_image_bytes(int C, int heads) {
    return cross_block_supported(C, heads, 1, 16) ? (long long)cross_block_image_bytes(C) : 0;
}
int lavie_pack_cross_block_f16(const void* wo1, const void* wq2, const void* wo2, int C, void* tmpl, void* stream) {
    LAVIE_CHECK(wo1 && wq2 && wo2 && tmpl, "pack_cross_block: null tensor");
    return pack_cross_block(H(wo1), H(wq2), H(wo2), C, (half_t*)tmpl, S(stream));
}
int lavie_bind_cross_block_f16(const void* tmpl, const void* kv, int B, int ctx_len, int C, void* img, void* stream) {
    LAVIE_CHECK(tmpl && kv && img, "bind_cross_block: null tensor");
    return bind_cross_block(H(tmpl), H(kv), B, ctx_len, C, (half_t*)img, S(stream));
}
int lavie_cross_block_f16(const void* att, const void* x, void* y, int M, int rows_per_batch, int C, int heads, const void* img,
                          const float* bo1, const float* gamma, const float* beta, const float* bo2, int ctx_len, float scale,
                          float eps, void* stream) {
    return launch_cross_block(H(att), H(x), (half_t*)y, M, rows_per_batch, C, heads, H(img), bo1, gamma, beta, bo2, ctx_len, scale, eps,
                              S(stream));
}

int lavie_conv3x3_f16(const void* x1, int C1, const void* x2, int C2, const void* sc1, int SC1, const void* sc2, int SC2,
                      const void* Wp, const float* bias, const float* bias2, int ldb2, int rows_per_batch, const void* R,
                      void* y, int NI, int Hi, int Wi, int Cout, int stride, int ups, const void* zero_page,
                      void* stream) {
    LAVIE_CHECK(x1 && Wp && y && zero_page, "conv3x3: null tensor");
    LAVIE_CHECK((stride == 1 || stride == 2) && (ups == 0 || ups == 1) && !(ups && stride != 1), "conv3x3: stride=%d ups=%d", stride, ups);
    LAVIE_CHECK(C1 > 0 && C1 % IGEMM_BK == 0 && C2 % IGEMM_BK == 0 && SC1 % IGEMM_BK == 0 && SC2 % IGEMM_BK == 0,
                "conv3x3: channel counts must be multiples of %d", IGEMM_BK);
    LAVIE_CHECK((!SC1 && !SC2) || (stride == 1 && !ups), "conv3x3: fused shortcut needs stride 1, no upsample");
    LAVIE_CHECK(!bias2 || rows_per_batch > 0, "conv3x3: bias2 needs rows_per_batch > 0");
    IgemmParams p;
    memset(&p, 0, sizeof(p));
    p.W = H(Wp); p.C = H(y); p.ldc = Cout; p.bias = bias; p.bias2 = bias2; p.ldb2 = ldb2;
    p.rows_per_batch = rows_per_batch > 0 ? rows_per_batch : 1;
    p.R = H(R); p.ldr = Cout;
    p.Hi = Hi; p.Wi = Wi; p.stride = stride; p.ups = ups;
    p.Ho = ups ? Hi * 2 : (Hi - 1) / stride + 1;
    p.Wo = ups ? Wi * 2 : (Wi - 1) / stride + 1;
    p.M = NI * p.Ho * p.Wo; p.N = Cout; p.zero = H(zero_page);
    int ns = 0, nk = 0;
    const half_t* src[2] = {H(x1), H(x2)};
    const int srcC[2] = {C1, x2 ? C2 : 0};
    for (int i = 0; i < 2; ++i) {
        if (!srcC[i]) continue;
        IgemmSeg& sg = p.seg[ns++];
        sg.src = src[i]; sg.C = srcC[i]; sg.c0 = 0; sg.nchunks = srcC[i] / IGEMM_BK; sg.ntaps = 9;
        nk += 9 * sg.nchunks;
    }
    const half_t* sc[2] = {H(sc1), H(sc2)};
    const int scC[2] = {sc1 ? SC1 : 0, sc2 ? SC2 : 0};
    for (int i = 0; i < 2; ++i) {
        if (!scC[i]) continue;
        IgemmSeg& sg = p.seg[ns++];
        sg.src = sc[i]; sg.C = scC[i]; sg.c0 = 0; sg.nchunks = scC[i] / IGEMM_BK; sg.ntaps = 1;
        nk += sg.nchunks;
    }
    p.nseg = ns; p.nk = nk; p.ldw = nk * IGEMM_BK; p.tap_major = g_tap_major;
    if (int rc = op_slab(p, EPI_LINEAR, true)) return rc;
    return launch_igemm(p, true, EPI_LINEAR, S(stream));
}

int lavie_pack_conv3x3_parity_f16(const void* w, void* out, int Cout, int Cin, void* stream) {
    LAVIE_CHECK(w && out && Cout > 0 && Cin > 0, "pack_conv3x3_parity: bad arguments");
    return launch_pack_conv3x3_parity(H(w), H(out), Cout, Cin, S(stream));
}
int lavie_upsample_conv3x3_supported(int NI, int Hi, int Wi, int C) {
    IgemmParams p;
    return igemm_setup_parity_upsample(&p, nullptr, C, nullptr, nullptr, nullptr, NI, Hi, Wi, nullptr) ? 1 : 0;
}
int lavie_upsample_conv3x3_f16(const void* x, const void* wpar, const float* bias, void* y, int NI, int Hi, int Wi, int C,
                               const void* zero_page, void* stream) {
    LAVIE_CHECK(x && wpar && y && zero_page, "upsample_conv3x3: null tensor");
    IgemmParams p;
    LAVIE_CHECK(igemm_setup_parity_upsample(&p, H(x), C, H(wpar), bias, H(y), NI, Hi, Wi, H(zero_page)),
                "upsample_conv3x3: NI=%d %dx%d C=%d is outside the parity kernel's geometry (lavie_upsample_conv3x3_supported)", NI, Hi, Wi, C);
    if (p.splits > 1) {
        const size_t need = (size_t)p.splits * p.M * p.N * sizeof(float);
        if (need > g_slab_bytes) {
            if (g_slab) { LAVIE_HIP(hipDeviceSynchronize()); LAVIE_HIP(hipFree(g_slab)); g_slab = nullptr; g_slab_bytes = 0; }
            LAVIE_HIP(hipMalloc((void**)&g_slab, need));
            g_slab_bytes = need;
        }
        p.slab = g_slab;
    }
    return launch_igemm(p, true, EPI_LINEAR, S(stream));
}

int lavie_pack_conv3x3_f16(const void* w, void* out, int Cout, int Cin, int ld_out, int col0, void* stream) {
    LAVIE_CHECK(w && out && ld_out >= col0 + 9 * Cin, "pack_conv3x3: bad arguments");
    return launch_pack_conv3x3(H(w), H(out), Cout, Cin, ld_out, col0, g_tap_major == 0, S(stream));
}

int lavie_temporal_conv_f16(const void* x, int C, const void* Wp, const float* bias, const float* bias2, int ldb2,
                            int rows_per_batch, const void* R, void* y, int B, int F, int D, int Cout, int taps,
                            const void* zero_page, void* stream) {
    LAVIE_CHECK(x && Wp && y && zero_page, "temporal_conv: null tensor");
    LAVIE_CHECK(taps == 3 || taps == 5, "temporal_conv: taps=%d (3 or 5)", taps);
    LAVIE_CHECK(C > 0 && C % IGEMM_BK == 0 && Cout % 64 == 0, "temporal_conv: channel counts must be multiples of %d", IGEMM_BK);
    LAVIE_CHECK(B >= 1 && F >= 1 && D >= 1 && (long long)B * F * D * (C > Cout ? C : Cout) < (1ll << 31), "temporal_conv: bad shape");
    LAVIE_CHECK(!bias2 || rows_per_batch > 0, "temporal_conv: bias2 needs rows_per_batch > 0");
    IgemmParams p;
    memset(&p, 0, sizeof(p));
    p.W = H(Wp); p.C = H(y); p.ldc = Cout; p.bias = bias; p.bias2 = bias2; p.ldb2 = ldb2;
    p.rows_per_batch = rows_per_batch > 0 ? rows_per_batch : 1;
    p.R = H(R); p.ldr = Cout;
    p.tframes = F; p.tpix = D;
    p.Hi = p.Ho = 1; p.Wi = p.Wo = 1; p.stride = 1;          // unused in temporal mode (kept valid)
    p.M = B * F * D; p.N = Cout; p.zero = H(zero_page);
    IgemmSeg& sg = p.seg[0];
    sg.src = H(x); sg.C = C; sg.c0 = 0; sg.nchunks = C / IGEMM_BK; sg.ntaps = taps;
    p.nseg = 1; p.nk = taps * sg.nchunks; p.ldw = p.nk * IGEMM_BK; p.tap_major = 0;
    if (int rc = op_slab(p, EPI_LINEAR, true)) return rc;
    return launch_igemm(p, true, EPI_LINEAR, S(stream));
}

int lavie_pack_temporal_conv_f16(const void* w, void* out, int Cout, int Cin, int taps, void* stream) {
    LAVIE_CHECK(w && out && (taps == 3 || taps == 5), "pack_temporal_conv: bad arguments");
    return launch_pack_conv_taps(H(w), H(out), Cout, Cin, taps, taps * Cin, 0, true, S(stream));
}

int lavie_pack_geglu_f16(const void* w, const void* bias_f16, void* w_out, float* bias_out, int N, int K, void* stream) {
    LAVIE_CHECK(w && w_out, "pack_geglu: null tensor");
    int rc = launch_pack_geglu_rows(H(w), H(w_out), N, K, S(stream));
    if (rc == 0 && bias_f16 && bias_out) rc = launch_pack_geglu_bias(H(bias_f16), bias_out, N, S(stream));
    return rc;
}

long long lavie_proj_qkv_image_bytes(int C) { return proj_qkv_supported(C) ? (long long)proj_qkv_image_bytes(C) : 0; }
int lavie_pack_proj_qkv_f16(const void* wpin, const void* wqkv, int C, void* img, void* stream) {
    LAVIE_CHECK(wpin && wqkv && img, "pack_proj_qkv: null tensor");
    return pack_proj_qkv(H(wpin), H(wqkv), C, (half_t*)img, S(stream));
}
int lavie_group_norm_affine_f16(const void* x, int C, int NB, int P, int groups, const float* gamma, const float* beta, float eps,
                                float* stats_ws, float* ab_out, void* stream) {
    LAVIE_CHECK(x && gamma && beta && stats_ws && ab_out, "group_norm_affine: null tensor");
    LAVIE_CHECK(NB > 0 && P > 0 && groups > 0, "group_norm_affine: empty problem");
    return launch_group_norm(H(x), C, nullptr, 0, NB, P, groups, gamma, beta, eps, false, stats_ws, nullptr, S(stream), nullptr, nullptr, ab_out);
}
int lavie_proj_qkv_f16(const void* x, const float* gn_ab, int rows_per_domain, const void* img, const float* bpin, const float* ln_gamma,
                       const float* ln_beta, float ln_eps, void* tx, void* qkv, int M, int C, void* stream) {
    return launch_proj_qkv(H(x), gn_ab, rows_per_domain, H(img), bpin, ln_gamma, ln_beta, ln_eps, (half_t*)tx, (half_t*)qkv, M, C, S(stream));
}

int lavie_group_norm_f16(const void* x1, int C1, const void* x2, int C2, int NB, int P, int groups, const float* gamma,
                         const float* beta, float eps, int silu, float* stats_ws, void* y, void* stream) {
    LAVIE_CHECK(x1 && gamma && beta && stats_ws && y, "group_norm: null tensor");
    LAVIE_CHECK(NB > 0 && P > 0 && groups > 0, "group_norm: empty problem");
    if (!x2) C2 = 0;
    return launch_group_norm(H(x1), C1, H(x2), C2, NB, P, groups, gamma, beta, eps, silu != 0, stats_ws, H(y), S(stream));
}

long long lavie_group_norm_ws_floats(int NB, int groups) { return (long long)gn_workspace_floats(NB, groups); }

int lavie_layer_norm_f16(const void* x, const float* gamma, const float* beta, void* y, int rows, int C, float eps,
                         void* stream) {
    LAVIE_CHECK(x && gamma && beta && y && rows > 0, "layer_norm: bad arguments");
    return launch_layernorm(H(x), gamma, beta, H(y), rows, C, eps, S(stream));
}

int lavie_attention_f16(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, void* o, int ldo, int NB,
                        int Lq, int Lk, int heads, int dh, int kv_batch_div, float scale, void* stream) {
    LAVIE_CHECK(q && k && v && o, "attention: null tensor");
    AttnParams a;
    a.q = H(q); a.ldq = ldq; a.k = H(k); a.ldk = ldk; a.v = H(v); a.ldv = ldv; a.o = H(o); a.ldo = ldo;
    a.NBq = NB; a.Lq = Lq; a.Lk = Lk; a.heads = heads; a.dh = dh; a.kv_batch_div = kv_batch_div; a.scale = scale;
    return launch_attention(a, S(stream));
}

int lavie_sparse_causal_attention_f16(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, void* o,
                                      int ldo, int NB, int frames, int D, int heads, int dh, float scale, void* stream) {
    LAVIE_CHECK(q && k && v && o, "sparse-causal attention: null tensor");
    LAVIE_CHECK(frames >= 1 && NB >= frames && NB % frames == 0 && D >= 1, "sparse-causal attention: NB=%d frames=%d D=%d", NB, frames, D);
    AttnParams a;
    a.q = H(q); a.ldq = ldq; a.k = H(k); a.ldk = ldk; a.v = H(v); a.ldv = ldv; a.o = H(o); a.ldo = ldo;
    a.NBq = NB; a.Lq = D; a.Lk = 2 * D; a.heads = heads; a.dh = dh; a.kv_batch_div = 1; a.scale = scale;
    a.sc_frames = frames;
    return launch_attention(a, S(stream));
}

int lavie_temporal_attention_f16(const void* qkv, int ld, void* o, int ldo, int B, int F, int D, int heads, int dh,
                                 const float* bias, const float* rot_cos, const float* rot_sin, int rot_dim, float scale,
                                 void* stream) {
    LAVIE_CHECK(qkv && o && bias && (rot_dim == 0 || (rot_cos && rot_sin)), "temporal attention: null tensor");
    TemporalParams t;
    t.qkv = H(qkv); t.ld = ld; t.o = H(o); t.ldo = ldo; t.B = B; t.F = F; t.D = D; t.heads = heads; t.dh = dh;
    t.bias = bias; t.rot_cos = rot_cos; t.rot_sin = rot_sin; t.rot_dim = rot_dim; t.scale = scale;
    return launch_temporal_attention(t, S(stream));
}

int lavie_relpos_buckets(int F, int num_buckets, int max_distance, int* out_host) {
    LAVIE_CHECK(F > 0 && num_buckets >= 4 && max_distance > num_buckets / 4 && out_host, "relpos_buckets: bad arguments");
    relpos_bucket_table(F, num_buckets, max_distance, out_host);
    return 0;
}

int lavie_cfg_ddpm_step(const void* eps2, float* x, const float* noise, void* model_in2, long long n, float guidance,
                        float k_x, float k_eps, float c_x0, float c_xt, float sigma, void* stream) {
    LAVIE_CHECK(eps2 && x && model_in2 && n > 0, "cfg_ddpm_step: bad arguments");
    return launch_cfg_ddpm_step(H(eps2), x, noise, H(model_in2), n, guidance, k_x, k_eps, c_x0, c_xt, sigma, 1.0f, S(stream));
}

int lavie_cfg_sampler_step(const void* eps2, float* x, const float* noise, void* model_in2, long long n, float guidance,
                           float k_x, float k_eps, float c_x0, float c_xt, float sigma, float next_input_scale, void* stream) {
    LAVIE_CHECK(eps2 && x && model_in2 && n > 0, "cfg_sampler_step: bad arguments");
    return launch_cfg_ddpm_step(H(eps2), x, noise, H(model_in2), n, guidance, k_x, k_eps, c_x0, c_xt, sigma, next_input_scale,
                                S(stream));
}

int lavie_sampler_step(const void* eps, float* x, const float* noise, void* model_in, long long n, float k_x, float k_eps,
                       float c_x0, float c_xt, float sigma, float next_input_scale, void* stream) {
    LAVIE_CHECK(eps && x && model_in && n > 0, "sampler_step: bad arguments");
    return launch_sampler_step(H(eps), x, noise, H(model_in), n, k_x, k_eps, c_x0, c_xt, sigma, next_input_scale, S(stream));
}

int lavie_latents_to_scaled_model_input1(const float* x, void* model_in, long long n, float input_scale, void* stream) {
    LAVIE_CHECK(x && model_in && n > 0, "latents_to_scaled_model_input1: bad arguments");
    return launch_f32_to_f16_scaled(x, H(model_in), n, input_scale, S(stream));
}

int lavie_latents_to_model_input(const float* x, void* model_in2, long long n, void* stream) {
    LAVIE_CHECK(x && model_in2 && n > 0, "latents_to_model_input: bad arguments");
    return launch_f32_to_f16_dup2(x, H(model_in2), n, 1.0f, S(stream));
}

int lavie_latents_to_scaled_model_input(const float* x, void* model_in2, long long n, float input_scale, void* stream) {
    LAVIE_CHECK(x && model_in2 && n > 0, "latents_to_scaled_model_input: bad arguments");
    return launch_f32_to_f16_dup2(x, H(model_in2), n, input_scale, S(stream));
}

// Every switch below changes which kernels a forward enqueues: each bumps the process-wide debug epoch, which is part of
// the captured graph's key (engine.h GraphKey::debug_epoch), so a replay never runs a selection made under other switches.
int lavie_debug_force_tile(int mode) { bump_debug_epoch(); igemm_force_tile(mode); return 0; }
int lavie_debug_force_splits(int s) { bump_debug_epoch(); igemm_force_splits(s); return 0; }
int lavie_debug_conv_tap_major(int on) { bump_debug_epoch(); g_tap_major = on; return 0; }
int lavie_debug_attention_qt(int qt) { bump_debug_epoch(); attention_force_qt(qt); return 0; }
int lavie_debug_rowfuse_stamps(unsigned long long* buf) { rowfuse_set_stamp_buffer(buf); return 0; }
int lavie_debug_rowfuse_variant(int v) { bump_debug_epoch(); rowfuse_set_variant(v); return 0; }
int lavie_debug_temporal_block_dump(float* buf) { temporal_block_set_debug(buf); return 0; }
int lavie_debug_fused_mask(int mask) { bump_debug_epoch(); set_fused_mask(mask); return 0; }
int lavie_debug_temporal_budget(int bytes) { bump_debug_epoch(); temporal_set_budget(bytes); return 0; }
int lavie_debug_ppx_stamps(unsigned long long* out256) {
    LAVIE_CHECK(out256, "ppx_stamps: null output");
    return igemm_ppx_read_stamps(out256);
}
long long lavie_debug_gn_producer_count(void) { return (long long)lavie::gn_producer_count(); }

int lavie_debug_patch_stamps(unsigned long long* out128) {
    LAVIE_CHECK(out128, "patch_stamps: null output");
    return igemm_patch_read_stamps(out128);
}

int lavie_profile_begin(unsigned mask, int max_events) { return profile_begin(mask, max_events); }

int lavie_profile_end(void* stream, long long* launches_host, double* ms_host, double* flops_host, double* bytes_host) {
    LAVIE_CHECK(launches_host && ms_host && flops_host && bytes_host, "profile_end: null output");
    return profile_end(S(stream), launches_host, ms_host, flops_host, bytes_host);
}

int lavie_unet_config_size(void) { return (int)sizeof(lavie_unet_config); }

int lavie_unet_create(const lavie_unet_config* cfg, lavie_unet_t* out) {
    LAVIE_CHECK(cfg && out, "unet_create: null argument");
    // read only the first int until the caller's layout is known to be this build's
    LAVIE_CHECK(cfg->struct_size == (int)sizeof(lavie_unet_config),
                "unet_create: cfg->struct_size=%d but this library's lavie_unet_config has %d bytes (ABI %d): the binding's struct "
                "layout is out of date", cfg->struct_size, (int)sizeof(lavie_unet_config), LAVIE_ABI_VERSION);
    LAVIE_CHECK(cfg->num_levels >= 1 && cfg->num_levels <= LAVIE_MAX_LEVELS, "unet_create: num_levels=%d", cfg->num_levels);
    lavie_unet_s* h = new (std::nothrow) lavie_unet_s(*cfg);
    LAVIE_CHECK(h != nullptr, "unet_create: out of host memory");
    if (int rc = h->net.validate_config()) {
        delete h;
        return rc;
    }
    *out = h;
    return 0;
}

int lavie_unet_destroy(lavie_unet_t h) {
    delete h;
    return 0;
}

int lavie_unet_num_params(lavie_unet_t h) { return h ? (int)h->net.params().size() : -1; }

int lavie_unet_param_info(lavie_unet_t h, int i, const char** name, long long* numel) {
    LAVIE_CHECK(h && i >= 0 && i < (int)h->net.params().size(), "param_info: index %d out of range", i);
    if (name) *name = h->net.params()[i].name.c_str();
    if (numel) *numel = h->net.params()[i].numel;
    return 0;
}

int lavie_unet_set_param(lavie_unet_t h, const char* name, const void* data_f16, long long numel) {
    LAVIE_CHECK(h && name, "set_param: null argument");
    return h->net.set_param(name, data_f16, numel);
}

int lavie_unet_finalize(lavie_unet_t h, void* stream) {
    LAVIE_CHECK(h, "finalize: null handle");
    return h->net.finalize(S(stream));
}

int lavie_unet_prepare(lavie_unet_t h, int B, int F, int Hh, int W, int ctx_len) {
    LAVIE_CHECK(h, "prepare: null handle");
    return h->net.prepare(B, F, Hh, W, ctx_len);
}

int lavie_unet_cache_context(lavie_unet_t h, const void* ctx, int B, int ctx_len, void* stream) {
    LAVIE_CHECK(h, "cache_context: null handle");
    return h->net.cache_context(H(ctx), B, ctx_len, S(stream));
}

int lavie_unet_set_cfg_shared_input(lavie_unet_t h, int on) {
    LAVIE_CHECK(h, "set_cfg_shared_input: null handle");
    h->net.set_cfg_shared_input(on != 0);
    return 0;
}
int lavie_unet_set_ln_fold(lavie_unet_t h, int on) {
    LAVIE_CHECK(h, "set_ln_fold: null handle");
    h->net.set_ln_fold(on != 0);
    return 0;
}

long long lavie_unet_weight_bytes(lavie_unet_t h) { return h ? h->net.weight_bytes() : -1; }
long long lavie_unet_workspace_bytes(lavie_unet_t h) { return h ? h->net.workspace_bytes() : -1; }

int lavie_unet_forward(lavie_unet_t h, const void* sample, const float* timesteps, const void* ctx, void* out, int B,
                       int F, int Hh, int W, int ctx_len, void* stream) {
    LAVIE_CHECK(h, "forward: null handle");
    return h->net.forward(H(sample), timesteps, H(ctx), H(out), B, F, Hh, W, ctx_len, S(stream), nullptr);
}

int lavie_unet_forward_graph(lavie_unet_t h, const void* sample, const float* timesteps, const void* ctx, void* out, int B,
                             int F, int Hh, int W, int ctx_len, void* stream) {
    LAVIE_CHECK(h, "forward_graph: null handle");
    return h->net.forward_graph(H(sample), timesteps, H(ctx), H(out), B, F, Hh, W, ctx_len, S(stream));
}

int lavie_unet_forward_labels(lavie_unet_t h, const void* sample, const float* timesteps, const void* ctx,
                              const int* class_labels_host, void* out, int B, int F, int Hh, int W, int ctx_len, void* stream) {
    LAVIE_CHECK(h && class_labels_host, "forward_labels: null handle / labels");
    return h->net.forward(H(sample), timesteps, H(ctx), H(out), B, F, Hh, W, ctx_len, S(stream), class_labels_host);
}

int lavie_unet_resnet_forward(lavie_unet_t h, const char* prefix, const void* x1, int C1, const void* x2, int C2,
                              const float* temb, void* y, int B, int F, int Hh, int W, void* stream) {
    LAVIE_CHECK(h && prefix && x1 && temb && y, "resnet_forward: null argument");
    return h->net.resnet_forward(prefix, H(x1), C1, H(x2), C2, temb, H(y), B, F, Hh, W, S(stream));
}

int lavie_unet_transformer_forward(lavie_unet_t h, const char* prefix, void* x_inout, const void* ctx, int B, int F, int Hh,
                                   int W, int ctx_len, void* stream) {
    LAVIE_CHECK(h && prefix && x_inout && ctx, "transformer_forward: null argument");
    return h->net.transformer_forward(prefix, H(x_inout), H(ctx), B, F, Hh, W, ctx_len, S(stream));
}

}  // extern "C"
