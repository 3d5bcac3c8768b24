// Host-side denoiser: owns packed weights + workspace and enqueues one UNet forward on a stream.
#pragma once
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/lavie_hip.h"
#include "common.h"
#include "ops.h"

namespace lavie {

// Bump allocator over hipMalloc'd chunks (weights) or one fixed block (workspace).
class DeviceArena {
public:
    ~DeviceArena();
    int init_fixed(size_t bytes);                 // one block, alloc() fails when exhausted
    void init_virtual() { virtual_ = true; }      // no memory: only tracks the high-water mark
    void* alloc(size_t bytes);                    // 256-B aligned; nullptr on failure
    size_t mark() const { return off_; }
    void release(size_t mark) { off_ = mark; }
    size_t peak() const { return peak_; }
    size_t total_bytes() const { return total_; }
    void free_all();

private:
    std::vector<void*> chunks_;
    char* cur_ = nullptr;
    size_t cap_ = 0, off_ = 0, peak_ = 0, total_ = 0;
    bool fixed_ = false, virtual_ = false;
    static constexpr size_t kChunk = 512ull << 20;
};

struct ParamInfo {
    std::string name;
    std::vector<int> shape;
    long long numel;
};

struct NormW { float* g = nullptr; float* b = nullptr; int C = 0; };
struct LinW { half_t* w = nullptr; float* b = nullptr; int N = 0, K = 0; };

struct ResnetW {
    std::string prefix;
    int cin = 0, cout = 0;
    bool shortcut = false;
    NormW n1, n2;
    half_t* w1 = nullptr; float* b1 = nullptr;      // [cout][9*cin]
    half_t* w2 = nullptr; float* b2 = nullptr;      // [cout][9*cout (+ cin)] ; b2 = conv2.bias (+ shortcut.bias)
    int ldw2 = 0;
    int temb_off = 0;                               // column of this block inside the fused time_emb_proj output
    float eps = 0.f;                                // GroupNorm eps; 0 = the model's norm_eps
};

// ResnetBlock3DCNN (vsr/models/resnet.py:220-315): GroupNorm + SiLU -> (T,1,1) conv -> GroupNorm + SiLU -> (3,1,1) conv
struct TemporalResW {
    bool present = false;
    int taps1 = 3;
    NormW n1, n2;
    half_t* w1 = nullptr; float* b1 = nullptr;      // [C][taps1 * C]
    half_t* w2 = nullptr; float* b2 = nullptr;      // [C][3 * C]
};

struct TransformerW {
    std::string prefix;
    int C = 0;
    // VSR variant (lavie_unet_config::vsr_blocks / only_cross_attention)
    TemporalResW tres;                              // `resblock_temporal`, runs before the block's residual is taken
    bool attn1_cross = false;                       // attn1 attends to the text context
    half_t* wq1 = nullptr; half_t* wkv1 = nullptr;  // its projections: [C][C], [2C][cross_dim]
    half_t* f_q1 = nullptr; float* s_q1 = nullptr; float* b_q1 = nullptr;
    NormW gn, ln1, ln2, lnt, ln3;
    LinW pin, pout;
    half_t* wqkv1 = nullptr; LinW o1;
    half_t* wq2 = nullptr; half_t* wkv2 = nullptr; LinW o2;
    half_t* wqkvt = nullptr; LinW ot;
    half_t* relemb = nullptr;                       // [buckets][heads] fp16 (state-dict tensor)
    LinW ff1, ff2;                                  // ff1 in GEGLU-interleaved row order
    // LayerNorm-folded copies of the four projections that consume a LayerNorm (W * gamma, row sums, W beta + bias)
    half_t* f_qkv1 = nullptr; float* s_qkv1 = nullptr; float* b_qkv1 = nullptr;
    half_t* f_q2 = nullptr; float* s_q2 = nullptr; float* b_q2 = nullptr;
    half_t* f_qkvt = nullptr; float* s_qkvt = nullptr; float* b_qkvt = nullptr;
    half_t* f_ff1 = nullptr; float* s_ff1 = nullptr; float* b_ff1 = nullptr;
    // row-resident fused sub-blocks (rowfuse.hip), built where the width has a kernel (level 0: C = 320)
    half_t* ff_img = nullptr; float* ff_b1img = nullptr;    // norm3 -> GEGLU feed-forward -> + residual in one kernel
    half_t* tb_img = nullptr;                               // norm_temp -> q|k|v -> temporal attention -> to_out -> + residual
    half_t* xb_tmpl = nullptr;                              // attn1.to_out -> norm2 -> attn2 -> + residual: weight part of the image (rowfuse_cross.hip)
    half_t* pq_img = nullptr;                               // GroupNorm -> proj_in -> norm1 -> q|k|v (rowfuse_pin.hip, round 4)
};

struct SamplerW { half_t* w = nullptr; float* b = nullptr; int C = 0;
                  half_t* wpar = nullptr; };    // upsamplers: the four parity weight sets [4][C][4 C] (igemm_patch.hip MODE 3)

// TemporalModule3D (vsr/models/temporal_module.py:65-178): ResnetBlock3DCNN (5,1,1) -> ResnetBlock3D -> 1x1 shift conv
struct TemporalModuleW {
    std::string prefix;
    int C = 0;
    TemporalResW t;
    int t_temb_off = 0;                             // column of resblocks_3d_t.time_emb_proj inside the fused projection
    ResnetW s;
    LinW shift;
};

struct FwdCtx;   // per-call state (engine.cpp)

class UNet {
public:
    explicit UNet(const lavie_unet_config& cfg);
    ~UNet();
    int validate_config();
    const std::vector<ParamInfo>& params() const { return params_; }
    int set_param(const char* name, const void* data, long long numel);
    int finalize(hipStream_t stream);
    int prepare(int B, int F, int H, int W, int ctx_len);
    int forward(const half_t* sample, const float* timesteps, const half_t* ctx, half_t* out, int B, int F, int H, int W,
                int ctx_len, hipStream_t stream, const int* class_labels_host);
    int resnet_forward(const char* prefix, const half_t* x1, int C1, const half_t* x2, int C2, const float* temb, half_t* y,
                       int B, int F, int H, int W, hipStream_t stream);
    int transformer_forward(const char* prefix, half_t* x, const half_t* ctx, int B, int F, int H, int W, int ctx_len,
                            hipStream_t stream);
    // Text K/V of every transformer block for one context tensor, computed once and reused by every forward that is called
    // with the SAME ctx pointer and shape (the denoise loop passes one context for all of its steps); ctx == nullptr clears.
    int cache_context(const half_t* ctx, int B, int ctx_len, hipStream_t stream);
    // forward() replayed from a hipGraph: the first call with a new (pointers, shape) tuple runs eagerly (tables, kernel
    // attributes), the second captures the enqueue of one forward on `stream` and launches the instantiated graph, later
    // calls with the same tuple replay it.  Tensor CONTENTS may change between calls, addresses may not.  Falls back to the
    // eager forward while kernel profiling is active (events cannot be recorded into a capture).
    int forward_graph(const half_t* sample, const float* timesteps, const half_t* ctx, half_t* out, int B, int F, int H, int W,
                      int ctx_len, hipStream_t stream);
    long long weight_bytes() const { return (long long)weights_.total_bytes(); }
    long long workspace_bytes() const { return (long long)ws_.total_bytes(); }
    void set_ln_fold(bool on) { ln_fold_ = on; ++graph_gen_; }
    void set_cfg_shared_input(bool on) { cfg_shared_input_ = on; ++graph_gen_; }

private:
    void build_param_list();
    const half_t* given(const std::string& name) const;
    int pack_norm(const std::string& prefix, int C, NormW* out, hipStream_t s);
    int pack_linear(const std::string& prefix, int N, int K, bool bias, LinW* out, hipStream_t s);
    int pack_resnet(ResnetW* r, hipStream_t s);
    int pack_transformer(TransformerW* t, hipStream_t s);
    int pack_temporal_res(const std::string& prefix, int C, int taps1, TemporalResW* out, hipStream_t s);
    int run_temporal_res(FwdCtx& c, const TemporalResW& r, const half_t* x, half_t* y, int C, int D, const float* bias2, int ldb2,
                         const GnColStat* x_cs = nullptr, float* y_csbuf = nullptr, GnColStat* y_cs = nullptr);
    int run_temporal_module(FwdCtx& c, const TemporalModuleW& m, const half_t* x, half_t* y, const float* tproj, int ld_tproj,
                            int H, int W, const GnColStat* x_cs = nullptr, float* y_csbuf = nullptr, GnColStat* y_cs = nullptr);
    int pack_sampler(const std::string& prefix, int C, SamplerW* out, hipStream_t s, bool up = false);
    int ensure_tables(int F, hipStream_t s);

    int run(FwdCtx& c, const half_t* sample, const float* timesteps, const half_t* ctx, half_t* out);
    // cs1 / cs2: GroupNorm statistics the producers of x1 / x2 left (nullptr / empty = none: the statistics pass runs); y_csbuf / y_cs:
    // where the block's last conv leaves the statistics of y for ITS consumers (engine.cpp colstat_plan)
    int run_resnet(FwdCtx& c, const ResnetW& r, const half_t* x1, int C1, const half_t* x2, int C2, const float* tproj,
                   int ld_tproj, half_t* y, int H, int W, const GnColStat* cs1 = nullptr, const GnColStat* cs2 = nullptr,
                   float* y_csbuf = nullptr, GnColStat* y_cs = nullptr);
    // x_cs: in = statistics of the block input (for its per-frame GroupNorm), out = statistics of the block output (proj_out's epilogue)
    int run_transformer(FwdCtx& c, const TransformerW& t, half_t* x, const half_t* ctx, int H, int W, bool shared_prefix = false,
                        GnColStat* x_cs = nullptr, float* x_csbuf = nullptr);
    int run_conv(FwdCtx& c, const half_t* x, int C, const SamplerW& w, half_t* y, int Hi, int Wi, int stride, int ups,
                 float* cs_buf = nullptr, GnColStat* cs_out = nullptr);

    struct GraphKey {
        const void *sample = nullptr, *t = nullptr, *ctx = nullptr, *out = nullptr, *kv_ctx = nullptr;
        int B = 0, F = 0, H = 0, W = 0, L = 0;
        unsigned long gen = 0;          // bumped by everything that changes what a forward enqueues (workspace, caches, modes)
        unsigned long debug_epoch = 0;  // process-wide: bumped by every lavie_debug_* kernel-selection switch (api.cpp)
        hipStream_t stream = nullptr;
        bool operator==(const GraphKey& o) const {
            return sample == o.sample && t == o.t && ctx == o.ctx && out == o.out && kv_ctx == o.kv_ctx && B == o.B && F == o.F &&
                   H == o.H && W == o.W && L == o.L && gen == o.gen && debug_epoch == o.debug_epoch && stream == o.stream;
        }
    };
    void drop_graph();
    GraphKey graph_seen_, graph_key_;
    hipGraph_t graph_ = nullptr;
    hipGraphExec_t graph_exec_ = nullptr;
    hipStream_t cap_stream_ = nullptr;              // capture only: nothing ever executes on it
    unsigned long graph_gen_ = 0;

    lavie_unet_config cfg_;
    std::vector<ParamInfo> params_;
    std::unordered_map<std::string, size_t> index_;
    std::vector<const half_t*> given_;
    bool finalized_ = false;
    // the caller promises sample[b] == sample[b + B/2] (classifier-free guidance on duplicated latents): the layers in front of
    // the first text cross-attention are computed for the first half and copied
    bool cfg_shared_input_ = false;
    bool ln_fold_ = true;                           // LayerNorm folded into the producer / consumer GEMM epilogues

    DeviceArena weights_, ws_;
    // packed model
    half_t* conv_in_w_ = nullptr; float* conv_in_b_ = nullptr;
    half_t* conv_out_w_ = nullptr; float* conv_out_b_ = nullptr;
    NormW norm_out_;
    LinW time1_, time2_, tproj_;                    // tproj_: all ResnetBlock3D.time_emb_proj stacked
    std::vector<ResnetW> resnets_;                  // execution order
    std::vector<TransformerW> transformers_;
    std::vector<SamplerW> downs_, ups_;
    std::vector<TemporalModuleW> tmods_;            // VSR: down 0..L-1, mid, up 0..L-1
    half_t* class_emb_ = nullptr;                   // [num_class_embeds][time_embed_dim] fp16 (state-dict tensor)
    half_t* zero_page_ = nullptr;
    // per-F tables, built once per clip length and kept (the VSR chunk driver alternates F = 8 and F = 5)
    struct FrameTables {
        float* rot_cos = nullptr; float* rot_sin = nullptr;
        std::vector<float*> relbias;                // one [heads, F, F] per transformer
    };
    std::unordered_map<int, FrameTables> tables_;
    const FrameTables* cur_tables_ = nullptr;       // the entry of the running forward's F
    // cached text K/V (cache_context): one [B * ctx_len, 2C] buffer per transformer (attn2; attn1 on VSR cross levels)
    std::vector<half_t*> kv2_cache_, kv1_cache_;
    size_t kv_cache_rows_ = 0;                      // rows the buffers were allocated for
    // fused text cross-attention: per transformer with a template image, [B] images = weights + this context's K / V
    std::vector<half_t*> xb_img_;
    int kv_cache_B_ = 0;                            // videos the image buffers were allocated for
    bool xb_bound_ = false;                         // images hold the cached context (its length fits the kernel)
    void* kv_block_ = nullptr;                      // ONE hipMalloc'd block behind every K/V cache buffer: freed and reallocated on growth
    const half_t* kv_ctx_ = nullptr;
    int kv_B_ = 0, kv_len_ = 0;
    // spatial size of the running call (set by prepare()/forward() before run())
    int prep_H_ = 0, prep_W_ = 0;
};

}  // namespace lavie
