/* lavie_hip.h — C ABI of liblavie_hip.so: the MI355X (gfx950) implementation of the LaVie base
 * text-to-video denoising path.
 *
 * The reference (rigelshysaj/LaVie) is pure PyTorch and has no plugin/FFI seam; its seam is the
 * Python object protocol of `UNet3DConditionModel.forward` (base/models/unet.py:366-512) and the
 * sub-module forwards below it.  Each entry point cites the reference code it replaces.  The
 * Python facade in lavie_amd/ binds these symbols with ctypes (INTEGRATION.md shows the binding a
 * reference maintainer would add).
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless its name ends in `_host`; `stream` is a hipStream_t
 *    passed as void* (NULL = default stream); kernels are enqueued, never synchronised;
 *  - activations are fp16, CHANNELS-LAST: a video tensor [b, c, f, h, w] of the reference is stored
 *    as rows (b, f, y, x) of c contiguous halfs, which is also the reference's token layout
 *    "(b f) (h w) c" (attention.py:373) — the NCFHW <-> channels-last conversion happens only in
 *    lavie_unet_forward's first and last convolution;
 *  - norm scales/biases and linear biases are fp32 device arrays; weights are fp16;
 *  - the caller owns every buffer it passes; a lavie_unet_t owns its packed weights and workspace;
 *  - return value 0 = ok, negative = error, message from lavie_last_error() (thread local);
 *  - one host thread per process per GPU; no call may run concurrently on the same handle.
 */
#ifndef LAVIE_HIP_H
#define LAVIE_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define LAVIE_ABI_VERSION 7
#define LAVIE_MAX_LEVELS 8

const char* lavie_last_error(void);
int lavie_abi_version(void);

/* ------------------------------------------------------------------------------------------------
 * Operators (the finer seam: SURVEY.md §8b "Sub-modules")
 * ---------------------------------------------------------------------------------------------- */

/* C[M,N] = A[M,K] W[N,K]^T (+ bias[N]) (+ bias2[m / rows_per_batch, N]) (+ R[M,N]).
 * Replaces nn.Linear / 1x1 Conv2d: to_q/to_k/to_v/to_out (attention.py:95-104,154,177-178,202),
 * proj_in/proj_out (attention.py:328,356,371,394), FeedForward.net.2 (attention.py:479).
 * geglu != 0: W/bias hold the GEGLU projection in lavie_pack_geglu order and
 * C[M, N/2] = h * gelu_erf(gate) (diffusers GEGLU; spec vsr/models/diffusers_attention.py:801-822).
 * K %% 64 == 0, N %% 64 == 0 (N %% 128 for geglu); R may alias C. */
int lavie_linear_f16(const void* A, int lda, const void* W, const float* bias, const float* bias2, int ldb2,
                     int rows_per_batch, const void* R, int ldr, void* C, int ldc, int M, int N, int K, int geglu,
                     void* stream);

/* Per-frame 3x3 convolution, pad 1, on channels-last images; implicit GEMM.
 * Replaces InflatedConv3d (resnet.py:13-21) as used by ResnetBlock3D.conv1/conv2 (resnet.py:183,200),
 * Downsample3D (stride 2, resnet.py:102-110), Upsample3D (nearest x2 folded into the gather: `ups`=1,
 * resnet.py:62-72), the skip concatenation torch.cat([h, skip]) (x2/C2 != 0, unet_blocks.py:538,630)
 * and the fused 1x1 conv_shortcut (sc1/sc2 appended to K, resnet.py:175,203).
 *   x1,x2 : [NI, Hi, Wi, C1|C2]          Wp : [Cout, 9*(C1+C2) + SC1 + SC2] (lavie_pack_conv3x3 order)
 *   y     : [NI, Ho, Wo, Cout],  Ho = ups ? 2*Hi : (Hi + 2 - 3)/stride + 1
 *   bias2 : per-video bias [NI/frames_per_video... ] see rows_per_batch; R: residual [M, Cout].
 * All channel counts are multiples of 64. */
int lavie_conv3x3_f16(const void* x1, int C1, const void* x2, int C2, const void* sc1, int SC1, const void* sc2, int SC2,
                      const void* Wp, const float* bias, const float* bias2, int ldb2, int rows_per_batch, const void* R,
                      void* y, int NI, int Hi, int Wi, int Cout, int stride, int ups, const void* zero_page,
                      void* stream);

/* [Cout, Cin, 3, 3] (PyTorch) -> rows of `ld_out` halfs in the implicit GEMM's K order (64-channel slab, tap,
 * channel): out[co, col0 + ((ci/64)*9 + ky*3+kx)*64 + ci%64].  Cin %% 64 == 0. */
int lavie_pack_conv3x3_f16(const void* w, void* out, int Cout, int Cin, int ld_out, int col0, void* stream);
/* Temporal convolution over the frame axis on channels-last token rows (b, f, pixel): nn.Conv3d(C, Cout, kernel (T,1,1),
 * padding (T/2,0,0)), T = 3 or 5 — conv1 / conv2 of the VSR stage's ResnetBlock3DCNN (vsr/models/resnet.py:258-259, 274,
 * 285, 309).  Same implicit GEMM as the 3x3 conv with a frame-tap table: tap t of row m reads row
 * m + (t - T/2) * D, zeros outside the clip.  x [B*F*D, C], Wp [Cout, T*C] (lavie_pack_temporal_conv_f16 order),
 * y / R [B*F*D, Cout]; bias2 [B, ldb2] is the per-video time-embedding projection (rows_per_batch = F*D). */
int lavie_temporal_conv_f16(const void* x, int C, const void* Wp, const float* bias, const float* bias2, int ldb2,
                            int rows_per_batch, const void* R, void* y, int B, int F, int D, int Cout, int taps,
                            const void* zero_page, void* stream);
/* [Cout, Cin, T, 1, 1] (PyTorch Conv3d) -> [Cout][T*Cin] in the implicit GEMM's K order (64-channel slab, tap, channel). */
int lavie_pack_temporal_conv_f16(const void* w, void* out, int Cout, int Cin, int taps, void* stream);
/* A GEMM that consumes LayerNorm(A) without the normalised copy (how the engine runs every projection behind a LayerNorm,
 * BasicTransformerBlock attention.py:513-560): C[m, n] = rstd_m (sum_k A[m, k] Wf[n, k] - mean_m s[n]) + bias[n], with
 * Wf = W * gamma (fp16), s = row sums of Wf, bias = W beta (+ the layer's bias) prepared by the caller and ln_stats [M, 2] =
 * (mean, rstd) of the rows of A.  Never splits K. */
int lavie_linear_lnfold_f16(const void* A, const void* Wf, const float* bias, const float* ln_s, const float* ln_stats, void* C,
                            int M, int N, int K, void* stream);
/* GEGLU projection [2*inner, K] (+ bias) -> 16-row value/gate interleave expected by lavie_linear_f16(geglu=1). */
int lavie_pack_geglu_f16(const void* w, const void* bias_f16, void* w_out, float* bias_out, int N, int K, void* stream);

/* Fused feed-forward sub-block (ABI 5): y = x + W2 (h * gelu(g)) + b2 with (h, g) = W1 LayerNorm(x) + b1 — the
 * `hidden_states = self.ff(self.norm3(hidden_states)) + hidden_states` line of BasicTransformerBlock
 * (/root/reference/base/models/attention.py:558; FeedForward / GEGLU spec /root/reference/vsr/models/diffusers_attention.py:
 * 734-822) as ONE kernel: the [M, 4C] GEGLU intermediate never exists in memory.  Rows stay in registers from load to store,
 * weights stream through LDS in a packed image (lavie_pack_geglu_mlp_f16).  y may alias x.  Built for C = 320:
 * lavie_geglu_mlp_image_bytes returns 0 for a width the kernel is not built for.
 *   w1 [8C, C] = ff.net.0.proj.weight (value rows, then gate rows), b1 [8C] fp16, w2 [C, 4C] = ff.net.2.weight (fp16 device);
 *   img: lavie_geglu_mlp_image_bytes(C) bytes, b1img: lavie_geglu_mlp_bias_floats(C) floats (device, written by the pack call);
 *   gamma / beta: norm3 weight / bias fp32 [C]; b2: ff.net.2.bias fp32 [C]. */
long long lavie_geglu_mlp_image_bytes(int C);
long long lavie_geglu_mlp_bias_floats(int C);
int lavie_pack_geglu_mlp_f16(const void* w1, const void* b1_f16, const void* w2, int C, void* img, float* b1img, void* stream);
int lavie_geglu_mlp_f16(const void* x, void* y, int M, int C, const void* img, const float* b1img, const float* gamma,
                        const float* beta, const float* b2, float eps, void* stream);

/* Fused temporal sub-block (ABI 5): y = x + to_out(attn_temp(norm_temp(x))) — lines 548-555 of BasicTransformerBlock.forward
 * with TemporalAttention.forward / _attention (/root/reference/base/models/attention.py:580-667) as ONE kernel on token rows in
 * (b f) d order: LayerNorm, the q / k / v projections, scale + rotary, relative-position bias, softmax over the frames of a
 * pixel, P V, to_out + bias + residual.  Neither q|k|v nor the attention output exists in memory; the two rearranges of the
 * reference are row addressing.  y may alias x.  Built for C = 320, 8 heads, exactly 16 frames, rotary over 32 channels:
 * lavie_temporal_block_image_bytes returns 0 otherwise.
 *   wq / wk / wv / wo: attn_temp.to_q / to_k / to_v / to_out.0 weights [C, C] fp16 (device); img: image bytes (device);
 *   gamma / beta: norm_temp fp32 [C]; bo: to_out.0.bias fp32 [C]; relbias fp32 [heads, F, F] (query, key);
 *   rot_cos / rot_sin fp32 [F, rot_dim / 2]; scale = dim_head^-0.5 (applied to q before the rotary embedding, :640). */
long long lavie_temporal_block_image_bytes(int C, int heads, int F, int rot_dim);
int lavie_pack_temporal_block_f16(const void* wq, const void* wk, const void* wv, const void* wo, int C, void* img, void* stream);
int lavie_temporal_block_f16(const void* x, void* y, int B, int F, int D, int C, int heads, const void* img, const float* gamma,
                             const float* beta, const float* bo, const float* relbias, const float* rot_cos,
                             const float* rot_sin, int rot_dim, float scale, float eps, void* stream);

/* Fused text cross-attention sub-block (ABI 6): with att = the output of attn1's attention core (before its to_out),
 *     x'  = x + attn1.to_out(att)                                   (attention.py:513-522, the projection and residual)
 *     y   = x' + attn2.to_out(attn2(norm2(x'), K, V))               (attention.py:524-534; CrossAttention :253-335)
 * as ONE kernel: q, the attention output, norm2(x') and x' never exist in memory.  K / V = attn2.to_k / to_v of the text
 * context are per-video constants; they travel inside the weight stream: lavie_pack_cross_block_f16 writes the weight part of
 * an image once per model, lavie_bind_cross_block_f16 completes one image per video from kv [B * ctx_len, 2C] (k | v rows,
 * fp16) once per context.  Rows [M, C] of video b are rows [b * rows_per_batch, (b + 1) * rows_per_batch) (rows_per_batch =
 * frames * pixels, a multiple of 16).  y may alias x.  Built for C = 320, 8 heads, ctx_len <= 80:
 * lavie_cross_block_image_bytes returns 0 otherwise.
 *   wo1 / wq2 / wo2: attn1.to_out.0 / attn2.to_q / attn2.to_out.0 weights [C, C] fp16 (device); tmpl: image bytes;
 *   img: B * image bytes (device); bo1 / bo2: the to_out biases fp32 [C]; gamma / beta: norm2 fp32 [C]; scale = dim_head^-0.5. */
long long lavie_cross_block_image_bytes(int C, int heads);
int lavie_pack_cross_block_f16(const void* wo1, const void* wq2, const void* wo2, int C, void* tmpl, void* stream);
int lavie_bind_cross_block_f16(const void* tmpl, const void* kv, int B, int ctx_len, int C, void* img, void* stream);
int lavie_cross_block_f16(const void* att, const void* x, void* y, int M, int rows_per_batch, int C, int heads, const void* img,
                          const float* bo1, const float* gamma, const float* beta, const float* bo2, int ctx_len, float scale,
                          float eps, void* stream);

/* Upsample3D (ABI 6, /root/reference/base/models/resnet.py:44-79: F.interpolate(scale_factor=2, mode="nearest") then the 3x3
 * conv): y = conv3x3(nearest_x2(x)) + bias as FOUR 2x2 convs on x, one per output parity — the nine taps of an output pixel
 * fall on 2 x 2 source pixels, so the weights of coinciding taps are summed once at pack time (fp32 sum, one rounding) and the
 * product needs 4 C instead of 9 C multiply-adds per output element.  Same result as lavie_conv3x3_f16(..., ups = 1) up to that
 * rounding.  x [NI * Hi * Wi, C] rows, y [NI * 2Hi * 2Wi, C] rows.
 *   lavie_pack_conv3x3_parity_f16: w [C, C, 3, 3] fp16 (PyTorch layout) -> out [4][C][4 C] fp16 (device);
 *   lavie_upsample_conv3x3_supported: 1 when the geometry fits the kernel (C % 160 == 0, whole source rows per 320-pixel tile);
 *   otherwise use lavie_conv3x3_f16 with ups = 1. */
int lavie_pack_conv3x3_parity_f16(const void* w, void* out, int Cout, int Cin, void* stream);
int lavie_upsample_conv3x3_supported(int NI, int Hi, int Wi, int C);
int lavie_upsample_conv3x3_f16(const void* x, const void* wpar, const float* bias, void* y, int NI, int Hi, int Wi, int C,
                               const void* zero_page, void* stream);

/* GroupNorm (+ optional SiLU) over channels-last rows; the "batch" is whatever shares statistics:
 *   video domain  (resnet.py:180,191; unet.py:504): NB = b,   P = f*h*w   rows per batch
 *   frame domain  (attention.py:324,369)          : NB = b*f, P = h*w
 * Input may be the virtual concat [x1 | x2].  stats_ws: lavie_group_norm_ws_floats(NB, groups) floats of
 * scratch (slab partials + mean/rstd; no atomics: results are bit-reproducible). */
long long lavie_group_norm_ws_floats(int NB, int groups);
int lavie_group_norm_f16(const void* x1, int C1, const void* x2, int C2, int NB, int P, int groups, const float* gamma,
                         const float* beta, float eps, int silu, float* stats_ws, void* y, void* stream);

/* Fused head of the transformer block (ABI 7, round 4): with GN = the per-frame GroupNorm of Transformer3DModel (attention.py:369),
 *     tx  = proj_in(GN(x))                               (attention.py:371-373: 1x1 conv, then tokens)
 *     qkv = [to_q | to_k | to_v](norm1(tx))              (attention.py:513-516 with CrossAttention :154, 177-178)
 * as ONE kernel: the normalised copy of x, norm1's statistics and the re-read of tx never exist in memory.  The GroupNorm is handed
 * over as per-(frame, channel) pairs (a, b), y = a x + b: lavie_group_norm_affine_f16 computes the statistics (as
 * lavie_group_norm_f16 does) and writes ab_out [NB][C][2] instead of a normalised tensor.  rows_per_domain = rows per frame (a
 * multiple of 16 that divides M).  Built for C = 320: lavie_proj_qkv_image_bytes returns 0 otherwise.
 *   wpin [C, C] = proj_in.weight (1x1 conv or Linear), wqkv [3C, C] = attn1.to_q / to_k / to_v rows stacked (fp16, device);
 *   bpin: proj_in.bias fp32 [C]; ln_gamma / ln_beta: norm1 fp32 [C]; tx [M, C], qkv [M, 3C] fp16 out. */
long long lavie_proj_qkv_image_bytes(int C);
int lavie_pack_proj_qkv_f16(const void* wpin, const void* wqkv, int C, void* img, void* stream);
int lavie_group_norm_affine_f16(const void* x, int C, int NB, int P, int groups, const float* gamma, const float* beta, float eps,
                                float* stats_ws, float* ab_out, void* stream);
int lavie_proj_qkv_f16(const void* x, const float* gn_ab, int rows_per_domain, const void* img, const float* bpin, const float* ln_gamma,
                       const float* ln_beta, float ln_eps, void* tx, void* qkv, int M, int C, void* stream);

/* nn.LayerNorm(C) over rows (attention.py:442,459,474,480). */
int lavie_layer_norm_f16(const void* x, const float* gamma, const float* beta, void* y, int rows, int C, float eps,
                         void* stream);

/* softmax(scale q k^T) v with heads packed along channels; replaces CrossAttention._attention
 * (attention.py:209-239) and reshape_heads_to_batch_dim / reshape_batch_dim_to_heads (112-124).
 * q: [NB*Lq, ldq], k/v: [(NB/kv_batch_div)*Lk, ld], o: [NB*Lq, ldo]; head h at columns h*dh. */
int lavie_attention_f16(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, void* o, int ldo, int NB,
                        int Lq, int Lk, int heads, int dh, int kv_batch_div, float scale, void* stream);

/* SparseCausalAttention (interpolation/models/attention.py:609-665): spatial self-attention whose keys/values for
 * frame f of a video are the D tokens of the video's FIRST frame followed by the D tokens of frame max(f-1, 0)
 * (:630-639), i.e. 2 D keys per query; the concatenation is never materialised — the kernel stages both segments
 * straight from the per-frame K/V rows.  q/k/v/o: [NB*D, ld] rows, NB = videos * frames, NB %% frames == 0. */
int lavie_sparse_causal_attention_f16(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, void* o,
                                      int ldo, int NB, int frames, int D, int heads, int dh, float scale, void* stream);

/* TemporalAttention._attention (attention.py:634-667) on tokens ordered (b, f, pixel):
 * qkv [B*F*D, ld] = q | k | v, o [B*F*D, ldo]; bias [heads, F, F] fp32 (query, key);
 * rot_cos/rot_sin [F, rot_dim/2] fp32.  rot_dim = 0 (tables may be NULL) and an all-zero bias give the plain
 * softmax(scale q k^T) v over frames of the interpolation model's attn_temp (interpolation/models/attention.py:
 * 268-289, 596-603). */
int lavie_temporal_attention_f16(const void* qkv, int ld, void* o, int ldo, int B, int F, int D, int heads, int dh,
                                 const float* bias, const float* rot_cos, const float* rot_sin, int rot_dim, float scale,
                                 void* stream);

/* T5-style relative position buckets of RelativePositionBias (attention.py:681-699), HOST function:
 * out_host[i*F + j] = bucket(query i, key j). */
int lavie_relpos_buckets(int F, int num_buckets, int max_distance, int* out_host);

/* Classifier-free guidance + DDPM ancestral step, fused (pipeline_videogen.py:679-683 and
 * diffusers DDPMScheduler.step): eps2 = [uncond | cond] fp16 (n each), x fp32 (updated in place),
 * noise fp32 (may be NULL iff sigma == 0), model_in2 = fp16 [x' | x'] for the next UNet call. */
int lavie_cfg_ddpm_step(const void* eps2, float* x, const float* noise, void* model_in2, long long n, float guidance,
                        float k_x, float k_eps, float c_x0, float c_xt, float sigma, void* stream);
int lavie_latents_to_model_input(const float* x, void* model_in2, long long n, void* stream);
/* Same two kernels for schedulers whose `scale_model_input` is not the identity (EulerDiscreteScheduler, sample_method
 * 'eulerdiscrete', base/pipelines/sample.py:50-55; pipeline_videogen.py:667): the fp16 model input written for the NEXT
 * UNet call is x' * next_input_scale (Euler: 1 / sqrt(sigma_next^2 + 1)); x itself stays unscaled in fp32. */
int lavie_cfg_sampler_step(const void* eps2, float* x, const float* noise, void* model_in2, long long n, float guidance,
                           float k_x, float k_eps, float c_x0, float c_xt, float sigma, float next_input_scale, void* stream);
int lavie_latents_to_scaled_model_input(const float* x, void* model_in2, long long n, float input_scale, void* stream);
/* The loop body without classifier-free guidance (`guidance_scale <= 1`: do_classifier_free_guidance is False,
 * pipeline_videogen.py:626, 666, 678): eps fp16 [n] is used as it is, model_in fp16 [n] is the single copy x' * scale. */
int lavie_sampler_step(const void* eps, float* x, const float* noise, void* model_in, long long n, float k_x, float k_eps,
                       float c_x0, float c_xt, float sigma, float next_input_scale, void* stream);
int lavie_latents_to_scaled_model_input1(const float* x, void* model_in, long long n, float input_scale, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Measurement hook: HIP-event timing per kernel class on the launch stream (bench.py's roofline leg).
 * Classes: 0 conv3x3 (implicit GEMM, gathered), 1 linear/1x1/GEGLU GEMM, 2 spatial+text attention core,
 * 3 temporal attention core, 4 GroupNorm, 5 LayerNorm, 6 other, 7 the halo-patch conv kernel alone (a subset of
 * class 0: the events bracket exactly that kernel's launches), 8 the fused temporal sub-block kernel, 9 the fused feed-forward
 * kernel (both with events attached to the kernel launch itself).  lavie_profile_end synchronises the
 * stream and fills four host arrays of LAVIE_PROFILE_CLASSES entries (launches, milliseconds,
 * algorithmic flops, algorithmic bytes — the per-launch figures are defined in DESIGN.md).
 * ---------------------------------------------------------------------------------------------- */
#define LAVIE_PROFILE_CLASSES 11
/* Which optional engine paths lavie_unet_forward takes (A/B timing, parity cross-checks).  Bits: 0 = fused feed-forward kernel
 * (lavie_geglu_mlp_f16), 1 = fused temporal-attention sub-block (lavie_temporal_block_f16), 2 = fused text cross-attention
 * sub-block (lavie_cross_block_f16), 3 = conv_shortcut as its own GEMM in front of a halo-patch conv2 (measured slower: OFF by
 * default), 4 = parity form of the Upsample3D convs (lavie_upsample_conv3x3_f16), 5 = GroupNorm statistics taken from the
 * producing kernel's epilogue instead of a statistics pass (round 4).  Bit 6 (debug, off): every GroupNorm that takes producer statistics ALSO runs
 * the statistics pass and compares the two on the host.  Bit 7: a LayerNorm-folded GEMM on a kernel with the shared epilogue folds its producer's row-statistics partials itself (no
 * rowstat_finalize launch; the persistent kernel's consumers still finalize, once) — measured SLOWER (every N tile of the consumer
 * repeats the fold: linear class 7.23 -> 7.89 ms per forward, profiles/r04_ab_rowstat_fold_in_consumer.txt): off.  Bit 8: the fused block head
 * (lavie_proj_qkv_f16: GroupNorm -> proj_in -> norm1 -> q|k|v in one kernel).  Default 0x137 (bits 0, 1, 2, 4, 5, 8); 0 = the one-GEMM-per-launch path of round 2. */
int lavie_debug_fused_mask(int mask);
/* Test hook: GroupNorm launches so far (process-wide) that took their statistics from the producers' epilogues.  Bit 6 of the mask
 * above makes every such launch ALSO run the statistics pass and compare the two on the host (synchronises; an error names the
 * first (batch, group) that differs). */
long long lavie_debug_gn_producer_count(void);
int lavie_debug_temporal_block_dump(float* buf);   /* development aid: device buffer of 100 * 64 floats, or NULL */
int lavie_debug_rowfuse_stamps(unsigned long long* buf);   /* stamp build (variant 7): device buffer of 64 u64, or NULL */
int lavie_debug_rowfuse_variant(int v);   /* tuning: LDS read-ahead depth of the fused kernels (0 = default) */
/* Test/tuning knob for the implicit-GEMM kernel choice.  Low nibble: 0 automatic, 1 128-row kernel with the widest tile,
 * 3 160x320 ping-pong kernel wherever N % 320 == 0, 4 automatic without the ping-pong
 * kernel, 5 halo-patch conv kernel wherever the conv is eligible, 6 automatic without the halo-patch kernel,
 * 7 persistent ping-pong kernel for every eligible plain GEMM, 8 automatic without it, 9 automatic without the GEGLU GEMMs on it.
 * High nibble: diagnostic ablation build of the forced kernel (results wrong), except 0xC: the halo-patch kernel's ping-pong K loop. */
int lavie_debug_force_tile(int mode);
/* Test/tuning knob: force the split-K factor of the implicit GEMM (0 = automatic). */
int lavie_debug_force_splits(int s);
/* Diagnostic: op-level conv3x3 + pack use the K order (tap, slab) instead of (slab, tap). */
int lavie_debug_conv_tap_major(int on);
/* Tuning knob: 16-row query tiles per wave in the attention kernel for head dims <= 64 (0 = automatic).  A/B switches: 0x50 = the
 * register-staged kernels instead of the LDS-DMA ones, 0x40 = row sums on the VALU, 0x60 = the LDS-DMA kernels with the round-3
 * softmax (scale and running maximum applied by v_fma) instead of the round-4 one (both on the matrix pipe). */
int lavie_debug_attention_qt(int qt);
/* Tuning knob: LDS bytes one temporal-attention workgroup may stage (smaller = more workgroups per CU). */
int lavie_debug_temporal_budget(int bytes);
/* Diagnostic: per-wave phase-segment cycle sums [8 waves][16] of the last halo-patch conv launched in stamp mode
 * (lavie_debug_force_tile(0x75)); layout in igemm_patch.hip. */
int lavie_debug_patch_stamps(unsigned long long* out128);
/* Diagnostic: per-wave segment cycle sums [8 waves][32] of the last persistent ping-pong GEMM launched in stamp mode
 * (lavie_debug_force_tile(0x37)); layout in igemm_ppx.hip. */
int lavie_debug_ppx_stamps(unsigned long long* out256);
int lavie_profile_begin(unsigned mask, int max_events);
int lavie_profile_end(void* stream, long long* launches_host, double* ms_host, double* flops_host, double* bytes_host);

/* ------------------------------------------------------------------------------------------------
 * Whole denoiser: UNet3DConditionModel.forward (unet.py:366-512)
 * ---------------------------------------------------------------------------------------------- */
typedef struct lavie_unet_s* lavie_unet_t;

typedef struct lavie_unet_config {
    /* sizeof(lavie_unet_config) as the CALLER compiled / declared it.  lavie_unet_create rejects any other value, so a
     * binding written against an older (shorter) layout fails with a message instead of being read past its end. */
    int struct_size;
    int in_channels, out_channels;
    int num_levels;
    int block_out_channels[LAVIE_MAX_LEVELS];
    int attn_levels[LAVIE_MAX_LEVELS];      /* 1: CrossAttn{Down,Up}Block3D, 0: {Down,Up}Block3D */
    int layers_per_block;
    int heads;
    int cross_attention_dim;
    int norm_groups;
    float norm_eps;
    int rotary_dim;
    int rel_buckets, rel_max_distance;
    /* Block variant of the frame-interpolation model (interpolation/models/attention.py:456-606; all 0 = base model):
     *   sparse_causal_attn1 : attn1 is SparseCausalAttention (use_first_frame, :493-504, 609-665)
     *   temporal_plain      : attn_temp is the plain CrossAttention over frames — no rotary embedding, no
     *                         relative-position bias, and no such tensors in the state dict (:525-533)
     *   ff_before_temporal  : block order spatial -> text -> feed-forward -> temporal (:566-606) */
    int sparse_causal_attn1, temporal_plain, ff_before_temporal;
    /* Block variant of the VSR stage's UNet3DVSRModel (vsr/models/attention.py:314-594; 0 = base model):
     *   vsr_blocks              : Transformer3DModel starts with a ResnetBlock3DCNN (3,1,1) without time embedding
     *                             (`resblock_temporal`, :350, 395-398), the temporal attention tensors are named
     *                             attn_temporal / norm_temporal, proj_in / proj_out are nn.Linear (:353, 383)
     *   only_cross_attention[l] : attn1 of level l attends to the text context instead of the frame (:465-490, 558-561) */
    int vsr_blocks;
    int only_cross_attention[LAVIE_MAX_LEVELS];
    /* UNet3DVSRModel (vsr/models/unet.py:100-600):
     *   vsr_temporal_modules : a TemporalModule3D (ResnetBlock3DCNN (5,1,1) -> ResnetBlock3D -> zero-initialised 1x1 shift
     *                          conv, residual; temporal_module.py:65-178) after every down block, the mid block and every
     *                          up block (down_temporal_idx / mid_temporal / up_temporal_idx = all levels)
     *   num_class_embeds     : > 0: emb = time_embedding + class_embedding[noise level] (:176-177, 494-505); the forward
     *                          entry is then lavie_unet_forward_labels */
    int vsr_temporal_modules;
    int num_class_embeds;
} lavie_unet_config;

/* sizeof(lavie_unet_config) in this build of the library: what cfg->struct_size must hold. */
int lavie_unet_config_size(void);
int lavie_unet_create(const lavie_unet_config* cfg, lavie_unet_t* out);
int lavie_unet_destroy(lavie_unet_t h);
/* Number of state-dict entries the model expects and the i-th name/numel (reference key names,
 * unet.py:142-295): lets a binder enumerate the checkpoint contract without Python. */
int lavie_unet_num_params(lavie_unet_t h);
int lavie_unet_param_info(lavie_unet_t h, int i, const char** name, long long* numel);
/* Hand over one fp16 state-dict tensor (borrowed until lavie_unet_finalize returns and the stream drains). */
int lavie_unet_set_param(lavie_unet_t h, const char* name, const void* data_f16, long long numel);
/* Repack all weights into the engine's own arena (fused QKV, [Cout][tap][Cin] convs, GEGLU order, fp32 biases). */
int lavie_unet_finalize(lavie_unet_t h, void* stream);
/* Size the activation workspace for inputs up to [B, *, F, H, W] (allocates; not stream-ordered). */
int lavie_unet_prepare(lavie_unet_t h, int B, int F, int H, int W, int ctx_len);
/* Optional: the text keys / values of every transformer block (attention.py:177-178 on encoder_hidden_states, which the
 * reference recomputes in every block of every denoising step, and once per frame: :364) computed ONCE for the context
 * tensor `ctx` [B, ctx_len, cross_attention_dim] and kept in the handle.  Forwards called afterwards with the same `ctx`
 * pointer, B and ctx_len read them instead of recomputing them; any other context is computed as usual.  The caller
 * promises not to change the tensor's contents while it is cached; ctx = NULL drops the cache.  Needs lavie_unet_prepare. */
int lavie_unet_cache_context(lavie_unet_t h, const void* ctx, int B, int ctx_len, void* stream);
/* A/B switch (default on): fold every LayerNorm of the transformer blocks into the epilogues of the GEMM that
 * produces its input (row statistics) and the GEMM that consumes its output (gamma folded into the weights). */
int lavie_unet_set_ln_fold(lavie_unet_t h, int on);
/* Classifier-free guidance (pipeline_videogen.py:666: `torch.cat([latents] * 2)`) runs the UNet on the SAME latents twice with
 * different text.  on = 1: the caller vouches that sample[b] == sample[b + B/2] for every b < B/2 (B even) in the forwards that
 * follow; the layers in front of the first text cross-attention (conv_in, down_blocks.0.resnets.0, and the GroupNorm / proj_in /
 * self-attention of down_blocks.0.attentions.0: unet.py:437-452, attention.py:369-373, 513-522) are then computed for the first
 * half of the batch only and copied.  Outputs equal the plain forward's to rounding (a half-batch launch may pick another tile).
 * Base UNet configuration only; ignored (plain forward) where it does not apply.  Default off. */
int lavie_unet_set_cfg_shared_input(lavie_unet_t h, int on);
long long lavie_unet_weight_bytes(lavie_unet_t h);
long long lavie_unet_workspace_bytes(lavie_unet_t h);
/* sample [B, Cin, F, H, W] fp16 (NCFHW, as the reference passes it), timesteps [B] fp32,
 * ctx [B, ctx_len, cross_attention_dim] fp16  ->  out [B, Cout, F, H, W] fp16. */
int lavie_unet_forward(lavie_unet_t h, const void* sample, const float* timesteps, const void* ctx, void* out, int B,
                       int F, int H, int W, int ctx_len, void* stream);
/* The same forward replayed from a hipGraph (the reference has no counterpart: it is the launch path of unet.py:366-512).
 * First call with a new (pointers, shape, stream) tuple: eager.  Second: the enqueue of one forward is captured on `stream`,
 * instantiated and launched.  Later calls with the same tuple: hipGraphLaunch.  Tensor contents may change between calls,
 * addresses may not (a changed address simply starts over).  Runs eagerly while lavie_profile_begin is active. */
int lavie_unet_forward_graph(lavie_unet_t h, const void* sample, const float* timesteps, const void* ctx, void* out, int B,
                             int F, int H, int W, int ctx_len, void* stream);

/* Same for a model with num_class_embeds > 0: class_labels_host[B] (host ints, the VSR noise level per video). */
int lavie_unet_forward_labels(lavie_unet_t h, const void* sample, const float* timesteps, const void* ctx,
                              const int* class_labels_host, void* out, int B, int F, int H, int W, int ctx_len, void* stream);

/* Finer engine seams for parity tests (same packed weights as the whole model):
 * ResnetBlock3D.forward (resnet.py:177-207) and Transformer3DModel.forward (attention.py:358-407)
 * of the block whose state-dict prefix is `prefix` (e.g. "down_blocks.0.resnets.0").
 * x1/x2/y channels-last; temb [B, time_embed_dim] fp32 (the output of time_embedding, pre-SiLU). */
int lavie_unet_resnet_forward(lavie_unet_t h, const char* prefix, const void* x1, int C1, const void* x2, int C2,
                              const float* temb, void* y, int B, int F, int H, int W, void* stream);
int lavie_unet_transformer_forward(lavie_unet_t h, const char* prefix, void* x_inout, const void* ctx, int B, int F, int H,
                                   int W, int ctx_len, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LAVIE_HIP_H */
