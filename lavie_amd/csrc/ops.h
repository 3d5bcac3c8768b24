// Internal launcher declarations (host side).  Every launcher enqueues on `stream`, performs no
// allocation or synchronisation (hipGraph-capturable), and returns 0 or a negative status with
// the message available through lavie::get_error().
#pragma once
#include "common.h"
#include "igemm.h"

namespace lavie {

// ---- norm.hip
// GroupNorm (+SiLU): statistics pass (per-slab partials), finalize (mean, rstd), apply.  `ws` is
// gn_workspace_floats(NB, groups) floats of scratch, reusable by the next call on the same stream.
size_t gn_workspace_floats(int NB, int groups);
// Producer-side statistics of one GroupNorm input tensor: what IgemmParams::colstat_out received from the kernel that stored the
// tensor (igemm.h: per (row block, channel) sum and sum of squares of the rounded fp16 values).  rows == 0 / partials == nullptr =
// none.  Parity-form upsample conv: nsets = 4 sets of set_blocks source-row blocks (a block's rows lie in ONE frame).
struct GnColStat {
    const float* partials = nullptr;
    int C = 0;            // channels of the tensor
    int rows = 0;         // rows per block
    int nsets = 1;
    int set_blocks = 0;   // blocks per set
    int span = 0;         // the rows of a block lie inside ONE aligned run of `span` tensor rows (contiguous blocks: = rows; 2-D conv
                          // tiles: one frame; temporal-conv tiles: one video): a statistics domain must be a whole number of spans
};
// cs1 / cs2 (optional): statistics of x1 / x2 from their producers.  When both tensors have them and every block lies inside one
// statistics domain (P %% (rows * nsets) == 0), the statistics pass over the tensor and its finalize launch are replaced by ONE small
// fold of the partials; otherwise the two-pass path runs.
int launch_group_norm(const half_t* x1, int C1, const half_t* x2, int C2, int NB, int P, int groups, const float* gamma,
                      const float* beta, float eps, bool silu, float* ws, half_t* y, hipStream_t stream,
                      const GnColStat* cs1 = nullptr, const GnColStat* cs2 = nullptr, float* ab_out = nullptr);
// ab_out (optional, [NB][C1 + C2][2] floats): statistics only — instead of the apply pass (y is not written) the normalisation is
// handed on as per-(batch, channel) pairs (a, b), y = a x + b, for a consumer that applies it in registers (launch_proj_qkv)
long gn_producer_count();     // GroupNorm launches so far whose statistics came from the producers' epilogues (test hook)
int launch_layernorm(const half_t* x, const float* gamma, const float* beta, half_t* y, int rows, int C, float eps,
                     hipStream_t stream);

// ---- attention.hip : softmax(scale q k^T) v, heads packed along the channel axis
struct AttnParams {
    const half_t* q; int ldq;        // [NBq * Lq, ...] rows; head h at columns h*dh
    const half_t* k; int ldk;        // [NBkv * Lk, ...]
    const half_t* v; int ldv;
    half_t* o; int ldo;              // [NBq * Lq, heads*dh]
    int NBq, Lq, Lk, heads, dh;
    int kv_batch_div;                // kv batch index = q batch index / kv_batch_div (text ctx shared by the frames)
    float scale;
    // sparse-causal self-attention (interpolation/models/attention.py:609-665): sc_frames > 0 => batch entry (b, f) reads
    // keys [0, Lk/2) from frame (b, 0) and keys [Lk/2, Lk) from frame (b, max(f-1, 0)); Lk = 2 * Lq, kv_batch_div = 1
    int sc_frames = 0;
};
int launch_attention(const AttnParams& p, hipStream_t stream);
void attention_force_qt(int qt);   // tuning knob: query tiles per wave for head dims <= 64 (0 = automatic)

// ---- temporal_attention.hip
struct TemporalParams {
    const half_t* qkv; int ld;       // [(b f) d, 3C]: q | k | v per token, token order (b, f, pixel)
    half_t* o; int ldo;              // [(b f) d, C]
    int B, F, D, heads, dh;          // D = pixels per frame
    const float* bias;               // [heads, F, F] relative-position bias (query i, key j)
    const float* rot_cos;            // [F, rot_dim/2]
    const float* rot_sin;
    int rot_dim;
    float scale;
};
int launch_temporal_attention(const TemporalParams& p, hipStream_t stream);
void temporal_set_budget(int bytes);   // tuning knob: LDS bytes per workgroup

// ---- elementwise.hip
int launch_timestep_sinusoid(const float* t, float* out, int B, int dim, hipStream_t stream);
// out[b, n] = act_out(sum_k act_in(in[b, k]) * W[n, k] + bias[n]);  act: 0 none, 1 SiLU
int launch_gemv(const float* in, const half_t* W, const float* bias, float* out, int B, int N, int K, int act_in,
                int act_out, hipStream_t stream);
// x [B, Cin, F, H, W] (NCFHW fp16) -> y [(B F) H W, Cout] channels-last, 3x3 pad 1; w packed [3*3*Cin][Cout]
int launch_conv_in(const half_t* x, const half_t* wp, const float* bias, half_t* y, int B, int Cin, int F, int H, int W,
                   int Cout, hipStream_t stream);
// x [(B F) H W, Cin] channels-last -> y [B, Cout, F, H, W] NCFHW fp16, 3x3 pad 1; w packed [Cout][3*3][Cin]
int launch_conv_out(const half_t* x, const half_t* wp, const float* bias, half_t* y, int B, int Cin, int F, int H, int W,
                    int Cout, hipStream_t stream);
// Classifier-free guidance + DDPM ancestral step (pipeline_videogen.py:679-683):
//   eps = eps_u + s (eps_c - eps_u); x0 = kx x - ke eps; x' = c0 x0 + ct x + sigma noise
//   writes x' (fp32, in place allowed) and the duplicated fp16 model input [2, n] for the next step.
int launch_cfg_ddpm_step(const half_t* eps2, float* x, const float* noise, half_t* model_in2, int64_t n, float guidance,
                         float kx, float ke, float c0, float ct, float sigma, float in_scale, hipStream_t stream);
int launch_add_class_emb_silu(float* emb, const half_t* table, const int* labels_host, int B, int N, hipStream_t stream);
int launch_f32_to_f16_dup2(const float* x, half_t* out2, int64_t n, float in_scale, hipStream_t stream);
int launch_sampler_step(const half_t* eps, float* x, const float* noise, half_t* model_in, int64_t n, float kx, float ke,
                        float c0, float ct, float sigma, float in_scale, hipStream_t stream);
int launch_f32_to_f16_scaled(const float* x, half_t* out, int64_t n, float in_scale, hipStream_t stream);
int launch_fill_relpos_bias(const half_t* emb, const int* buckets, float* out, int heads, int F, hipStream_t stream);

// ---- pack.hip : one-off weight repacking at load time
// chunked = true: K order (64-channel slab, tap, channel) for the implicit GEMM; false: (tap, channel) for conv_out
int launch_pack_conv3x3(const half_t* w, half_t* out, int Cout, int Cin, int ld_out, int col0, bool chunked,
                        hipStream_t stream);
int launch_pack_conv3x3_parity(const half_t* w, half_t* out, int Cout, int Cin, hipStream_t stream);   // [4][Cout][4 Cin], see elementwise.hip
int launch_pack_conv_taps(const half_t* w, half_t* out, int Cout, int Cin, int taps, int ld_out, int col0, bool chunked,
                          hipStream_t stream);
int launch_copy_rows(const half_t* src, int ld_src, half_t* dst, int ld_dst, int rows, int cols, int col0,
                     hipStream_t stream);
int launch_pack_geglu_rows(const half_t* w, half_t* out, int N, int K, hipStream_t stream);
int launch_pack_geglu_bias(const half_t* b, float* out, int N, hipStream_t stream);
int launch_f16_to_f32(const half_t* src, float* dst, int64_t n, hipStream_t stream);
int launch_add_f16_to_f32(const half_t* a, const half_t* b, float* dst, int64_t n, hipStream_t stream);
int launch_pack_conv_in(const half_t* w, half_t* out, int Cout, int Cin, hipStream_t stream);
// LayerNorm folding: W' = W * gamma (fp16), s = row sums of W', b' = W beta (+ bias)
int launch_ln_fold(const half_t* W, const float* gamma, const float* beta, const half_t* bias, half_t* Wout, float* s_out,
                   float* b_out, int N, int K, hipStream_t stream);
int launch_pack_geglu_vec(const float* in, float* out, int N, hipStream_t stream);

// ---- rowfuse.hip : row-resident fused transformer sub-blocks (weights streamed through LDS, rows in registers)
void rowfuse_set_stamp_buffer(unsigned long long* buf);   // stamp build (variant 7): [8 waves][8] cycle sums, or nullptr
void rowfuse_set_variant(int v);   // tuning knob: LDS read-ahead depth (0 = default)
bool geglu_mlp_supported(int C);
size_t geglu_mlp_image_bytes(int C);
size_t geglu_mlp_bias_floats(int C);
int pack_geglu_mlp(const half_t* w1, const half_t* b1, const half_t* w2, int C, half_t* img, float* b1img, hipStream_t stream);
// y = x + W2 (h * gelu(g)) + b2, (h, g) = W1 LN(x) + b1; y may alias x
// stats_out (optional, [M, 2]): (mean, rstd) of every output row for a LayerNorm-folded GEMM that consumes y (eps as the input norm's)
int launch_geglu_mlp(const half_t* x, half_t* y, int M, int C, const half_t* img, const float* b1img, const float* gamma,
                     const float* beta, const float* b2, float eps, hipStream_t stream, float* stats_out = nullptr);

// x' = x + to_out(attn_temp(LN(x))) for clips of exactly 16 frames, rows in (b f) d order; y may alias x
void temporal_block_set_debug(float* buf);   // development aid: register-tile dump of workgroup 0 (nullptr = off)
bool temporal_block_supported(int C, int heads, int F, int rot_dim);
size_t temporal_block_image_bytes(int C);
int pack_temporal_block(const half_t* wq, const half_t* wk, const half_t* wv, const half_t* wo, int C, half_t* img, hipStream_t stream);
int launch_temporal_block(const half_t* x, half_t* y, int B, int F, int D, int C, int heads, const half_t* img,
                          const float* gamma, const float* beta, const float* bo, const float* relbias, const float* rot_cos,
                          const float* rot_sin, int rot_dim, float scale, float eps, hipStream_t stream);

// ---- rowfuse_pin.hip: tx = proj_in(GroupNorm(x)), qkv = [to_q | to_k | to_v](LayerNorm(tx)) in one kernel (C = 320); GroupNorm comes in
// as the (a, b) pairs of launch_group_norm(..., ab_out), one set per `rows_per_domain` rows (a frame)
bool proj_qkv_supported(int C);
size_t proj_qkv_image_bytes(int C);
int pack_proj_qkv(const half_t* wpin, const half_t* wqkv, int C, half_t* img, hipStream_t stream);
int launch_proj_qkv(const half_t* x, const float* gn_ab, int rows_per_domain, const half_t* img, const float* bpin, const float* ln_g,
                    const float* ln_b, float eps, half_t* tx, half_t* qkv, int M, int C, hipStream_t stream);

// ---- rowfuse_cross.hip: x'' = x' + to_out2(attn2(LN2(x'), K, V)), x' = x + to_out1(att), K / V of the text context streamed
// with the weights (one image per video: pack once per model, bind once per context); y may alias x
bool cross_block_supported(int C, int heads, int ctx_len, int rows_per_batch);
size_t cross_block_image_bytes(int C);
int pack_cross_block(const half_t* wo1, const half_t* wq2, const half_t* wo2, int C, half_t* tmpl, hipStream_t stream);
int bind_cross_block(const half_t* tmpl, const half_t* kv, int B, int L, int C, half_t* img, hipStream_t stream);
int launch_cross_block(const half_t* att, const half_t* x, half_t* y, int M, int rows_per_batch, int C, int heads, const half_t* img,
                       const float* bo1, const float* gamma, const float* beta, const float* bo2, int L, float scale, float eps,
                       hipStream_t stream);

// host-only helper (no GPU): T5-style bucket of (query i, key j), attention.py:681-699
void relpos_bucket_table(int F, int num_buckets, int max_distance, int* out);

}  // namespace lavie
