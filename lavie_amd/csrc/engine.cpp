// UNet3DConditionModel.forward on a HIP stream: weight packing, workspace planning, kernel sequencing.
// Mirrors the wiring of /root/reference/base/models/unet.py:454-506 and unet_blocks.py
// (226-232, 320-362, 417-441, 524-574, 625-648) on channels-last activations; see DESIGN.md.
#include "engine.h"
#include "profile.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

namespace lavie {

// ------------------------------------------------------------------ error text
static thread_local char g_err[1024] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
const char* get_error() { return g_err; }

static unsigned long g_debug_epoch = 0;
static int g_fused_mask = 0x137;        // bit 0: fused feed-forward, 1: fused temporal sub-block, 2: fused text cross-attention, 3: conv_shortcut as its own GEMM in front of a halo-patch conv2 (off), 4: parity form of the upsample convs, 5: GroupNorm statistics from the producers' epilogues (A/B switches); 6: debug verification of those statistics against the statistics pass (off); 7: LayerNorm row statistics folded in the consuming GEMM's epilogue instead of a finalize launch (measured slower: off); 8: GroupNorm -> proj_in -> norm1 -> q|k|v as one row-resident kernel
void set_fused_mask(int m) { g_fused_mask = m; }
int fused_mask() { return g_fused_mask; }
void bump_debug_epoch() { ++g_debug_epoch; }
static bool debug_check_shared() {
    const char* e = getenv("LAVIE_DEBUG_CHECK_SHARED");
    return e && e[0] == '1';
}
unsigned long debug_epoch() { return g_debug_epoch; }

int ensure_dynamic_lds(const void* kernel, int bytes) {
    static std::vector<std::pair<const void*, int>> seen;      // a handful of entries; one host thread per process drives the GPU
    for (const auto& e : seen)
        if (e.first == kernel && e.second >= bytes) return 0;
    LAVIE_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    seen.emplace_back(kernel, bytes);
    return 0;
}

#define RUN(expr)                 \
    do {                          \
        int rc_ = (expr);         \
        if (rc_ != 0) return rc_; \
    } while (0)

// ------------------------------------------------------------------ arena
DeviceArena::~DeviceArena() { free_all(); }

void DeviceArena::free_all() {
    for (void* p : chunks_) (void)hipFree(p);
    chunks_.clear();
    cur_ = nullptr;
    cap_ = off_ = peak_ = total_ = 0;
}

int DeviceArena::init_fixed(size_t bytes) {
    free_all();
    fixed_ = true;
    void* p = nullptr;
    LAVIE_HIP(hipMalloc(&p, bytes));
    chunks_.push_back(p);
    cur_ = (char*)p;
    cap_ = bytes;
    total_ = bytes;
    return 0;
}

void* DeviceArena::alloc(size_t bytes) {
    bytes = (bytes + 255) & ~(size_t)255;
    if (virtual_) {
        off_ += bytes;
        if (off_ > peak_) peak_ = off_;
        return (void*)(uintptr_t)256;   // never dereferenced
    }
    if (off_ + bytes > cap_) {
        if (fixed_) return nullptr;
        const size_t sz = bytes > kChunk ? bytes : kChunk;
        void* p = nullptr;
        if (hipMalloc(&p, sz) != hipSuccess) return nullptr;
        chunks_.push_back(p);
        cur_ = (char*)p;
        cap_ = sz;
        off_ = 0;
        total_ += sz;
    }
    void* r = cur_ + off_;
    off_ += bytes;
    if (off_ > peak_) peak_ = off_;
    return r;
}

// ------------------------------------------------------------------ model structure
UNet::UNet(const lavie_unet_config& cfg) : cfg_(cfg) { build_param_list(); }
UNet::~UNet() {
    drop_graph();
    if (cap_stream_) (void)hipStreamDestroy(cap_stream_);
    if (kv_block_) (void)hipFree(kv_block_);
}

int UNet::validate_config() {
    const lavie_unet_config& c = cfg_;
    LAVIE_CHECK(c.num_levels >= 1 && c.num_levels <= LAVIE_MAX_LEVELS, "config: num_levels=%d", c.num_levels);
    LAVIE_CHECK(c.in_channels >= 1 && c.in_channels <= 16 && c.out_channels >= 1 && c.out_channels <= 8,
                "config: in/out channels %d/%d unsupported", c.in_channels, c.out_channels);
    LAVIE_CHECK(c.heads >= 1 && c.layers_per_block >= 1 && c.norm_groups >= 1, "config: heads/layers/groups");
    LAVIE_CHECK(c.cross_attention_dim % 64 == 0, "config: cross_attention_dim %d must be a multiple of 64", c.cross_attention_dim);
    for (int l = 0; l < c.num_levels; ++l) {
        const int w = c.block_out_channels[l];
        LAVIE_CHECK(w % 64 == 0 && w % c.norm_groups == 0, "config: width %d must be a multiple of 64 and of norm_groups", w);
        if (c.attn_levels[l]) {
            const int min_dh = c.temporal_plain ? 8 : c.rotary_dim;
            LAVIE_CHECK(w % c.heads == 0 && (w / c.heads) % 8 == 0 && w / c.heads >= min_dh && w / c.heads <= 160,
                        "config: width %d gives head dim %d (need multiple of 8, %d..160)", w, w / c.heads, min_dh);
        }
    }
    LAVIE_CHECK(c.rotary_dim == 32 || c.rotary_dim == 16 || c.rotary_dim == 8, "config: rotary_dim %d", c.rotary_dim);
    LAVIE_CHECK(c.in_channels % 2 == 0, "config: in_channels %d must be even (conv_in packs channel pairs)", c.in_channels);
    return 0;
}

void UNet::build_param_list() {
    const lavie_unet_config& c = cfg_;
    auto add = [&](const std::string& name, std::vector<int> shape) {
        long long n = 1;
        for (int s : shape) n *= s;
        index_[name] = params_.size();
        params_.push_back({name, shape, n});
    };
    auto affine = [&](const std::string& p, int ch) { add(p + ".weight", {ch}); add(p + ".bias", {ch}); };
    auto conv = [&](const std::string& p, int cin, int cout, int k) { add(p + ".weight", {cout, cin, k, k}); add(p + ".bias", {cout}); };
    auto lin = [&](const std::string& p, int cin, int cout, bool bias) { add(p + ".weight", {cout, cin}); if (bias) add(p + ".bias", {cout}); };
    const int temb = c.block_out_channels[0] * 4;
    int temb_total = 0;
    auto resnet = [&](const std::string& p, int cin, int cout) {
        affine(p + ".norm1", cin);
        conv(p + ".conv1", cin, cout, 3);
        lin(p + ".time_emb_proj", temb, cout, true);
        affine(p + ".norm2", cout);
        conv(p + ".conv2", cout, cout, 3);
        if (cin != cout) conv(p + ".conv_shortcut", cin, cout, 1);
        ResnetW r;
        r.prefix = p; r.cin = cin; r.cout = cout; r.shortcut = cin != cout; r.temb_off = temb_total;
        temb_total += cout;
        resnets_.push_back(r);
    };
    auto attention = [&](const std::string& p, int ch, int kv) {
        lin(p + ".to_q", ch, ch, false); lin(p + ".to_k", kv, ch, false); lin(p + ".to_v", kv, ch, false);
        lin(p + ".to_out.0", ch, ch, true);
    };
    auto conv_t = [&](const std::string& p, int cin, int cout, int taps) { add(p + ".weight", {cout, cin, taps, 1, 1}); add(p + ".bias", {cout}); };
    auto transformer = [&](const std::string& p, int ch, int level) {
        const bool vsr = c.vsr_blocks != 0;
        const bool cross1 = vsr && level >= 0 && c.only_cross_attention[level] != 0;
        const std::string tname = vsr ? "temporal" : "temp";          // attn_temporal / norm_temporal in the VSR model
        if (vsr) {                                                     // resblock_temporal: ResnetBlock3DCNN (3,1,1), no temb
            affine(p + ".resblock_temporal.norm1", ch);
            conv_t(p + ".resblock_temporal.conv1", ch, ch, 3);
            affine(p + ".resblock_temporal.norm2", ch);
            conv_t(p + ".resblock_temporal.conv2", ch, ch, 3);
        }
        affine(p + ".norm", ch);
        if (vsr) lin(p + ".proj_in", ch, ch, true); else conv(p + ".proj_in", ch, ch, 1);
        const std::string b = p + ".transformer_blocks.0";
        attention(b + ".attn1", ch, cross1 ? c.cross_attention_dim : ch);
        affine(b + ".norm1", ch);
        attention(b + ".attn2", ch, c.cross_attention_dim);
        affine(b + ".norm2", ch);
        attention(b + ".attn_" + tname, ch, ch);
        if (!c.temporal_plain) {
            add(b + ".attn_" + tname + ".time_rel_pos_bias.relative_attention_bias.weight", {c.rel_buckets, c.heads});
            add(b + ".attn_" + tname + ".rotary_emb.freqs", {c.rotary_dim / 2});
        }
        affine(b + ".norm_" + tname, ch);
        lin(b + ".ff.net.0.proj", ch, 8 * ch, true);
        lin(b + ".ff.net.2", 4 * ch, ch, true);
        affine(b + ".norm3", ch);
        if (vsr) lin(p + ".proj_out", ch, ch, true); else conv(p + ".proj_out", ch, ch, 1);
        TransformerW t;
        t.prefix = p; t.C = ch;
        t.tres.present = vsr;
        t.attn1_cross = cross1;
        transformers_.push_back(t);
    };
    auto temporal_module = [&](const std::string& p, int ch) {
        TemporalModuleW m;
        m.prefix = p; m.C = ch;
        const std::string t = p + ".resblocks_3d_t", sp = p + ".resblocks_3d_s";
        affine(t + ".norm1", ch);
        conv_t(t + ".conv1", ch, ch, 5);
        lin(t + ".time_emb_proj", temb, ch, true);
        affine(t + ".norm2", ch);
        conv_t(t + ".conv2", ch, ch, 3);
        m.t_temb_off = temb_total;
        temb_total += ch;
        affine(sp + ".norm1", ch);
        conv(sp + ".conv1", ch, ch, 3);
        lin(sp + ".time_emb_proj", temb, ch, true);
        affine(sp + ".norm2", ch);
        conv(sp + ".conv2", ch, ch, 3);
        m.s.prefix = sp; m.s.cin = ch; m.s.cout = ch; m.s.shortcut = false; m.s.temb_off = temb_total; m.s.eps = 1e-6f;
        temb_total += ch;
        conv(p + ".shift_conv", ch, ch, 1);
        tmods_.push_back(m);
    };
    const bool tmod = c.vsr_temporal_modules != 0;
    const int* widths = c.block_out_channels;
    const int L = c.num_levels;
    if (c.num_class_embeds > 0) add("class_embedding.weight", {c.num_class_embeds, temb});
    if (c.vsr_blocks && !c.temporal_plain) add("temporal_rotary_emb.freqs", {c.rotary_dim / 2});   // accepted, ignored
    conv("conv_in", c.in_channels, widths[0], 3);
    lin("time_embedding.linear_1", widths[0], temb, true);
    lin("time_embedding.linear_2", temb, temb, true);
    std::vector<int> skips{widths[0]};
    int cur = widths[0];
    for (int l = 0; l < L; ++l) {
        for (int j = 0; j < c.layers_per_block; ++j) {
            const std::string p = "down_blocks." + std::to_string(l);
            resnet(p + ".resnets." + std::to_string(j), cur, widths[l]);
            cur = widths[l];
            if (c.attn_levels[l]) transformer(p + ".attentions." + std::to_string(j), cur, l);
            skips.push_back(cur);
        }
        if (l + 1 < L) {
            conv("down_blocks." + std::to_string(l) + ".downsamplers.0.conv", cur, cur, 3);
            skips.push_back(cur);
        }
        if (tmod) temporal_module("down_temporal_blocks." + std::to_string(l), cur);
    }
    resnet("mid_block.resnets.0", cur, cur);
    transformer("mid_block.attentions.0", cur, -1);      // the mid block always self-attends (vsr/models/unet.py:248-270)
    resnet("mid_block.resnets.1", cur, cur);
    if (tmod) temporal_module("mid_temporal_block", cur);
    for (int i = 0; i < L; ++i) {
        const int l = L - 1 - i;
        const std::string p = "up_blocks." + std::to_string(i);
        for (int j = 0; j < c.layers_per_block + 1; ++j) {
            const int sk = skips.back();
            skips.pop_back();
            resnet(p + ".resnets." + std::to_string(j), cur + sk, widths[l]);
            cur = widths[l];
            if (c.attn_levels[l]) transformer(p + ".attentions." + std::to_string(j), cur, l);
        }
        if (i + 1 < L) conv(p + ".upsamplers.0.conv", cur, cur, 3);
        if (tmod) temporal_module("up_temporal_blocks." + std::to_string(i), cur);
    }
    affine("conv_norm_out", widths[0]);
    conv("conv_out", widths[0], c.out_channels, 3);
    given_.assign(params_.size(), nullptr);
    tproj_.N = temb_total;
    tproj_.K = temb;
}

int UNet::set_param(const char* name, const void* data, long long numel) {
    auto it = index_.find(name);
    LAVIE_CHECK(it != index_.end(), "set_param: unknown state-dict key '%s'", name);
    const ParamInfo& pi = params_[it->second];
    LAVIE_CHECK(pi.numel == numel, "set_param: '%s' has %lld elements, expected %lld", name, numel, pi.numel);
    LAVIE_CHECK(data != nullptr, "set_param: '%s' null data", name);
    given_[it->second] = (const half_t*)data;
    return 0;
}

const half_t* UNet::given(const std::string& name) const {
    auto it = index_.find(name);
    return it == index_.end() ? nullptr : given_[it->second];
}

// ------------------------------------------------------------------ weight packing
#define NEED(ptr, name) LAVIE_CHECK((ptr) != nullptr, "finalize: missing state-dict tensor '%s'", (name).c_str())
#define WALLOC(var, type, count)                                                                  \
    do {                                                                                          \
        (var) = (type*)weights_.alloc((size_t)(count) * sizeof(type));                            \
        LAVIE_CHECK((var) != nullptr, "finalize: out of device memory (%zu B; hip: %s)", (size_t)(count) * sizeof(type), hipGetErrorString(hipGetLastError())); \
    } while (0)

int UNet::pack_norm(const std::string& prefix, int C, NormW* out, hipStream_t s) {
    const half_t* g = given(prefix + ".weight");
    const half_t* b = given(prefix + ".bias");
    NEED(g, prefix + ".weight");
    NEED(b, prefix + ".bias");
    out->C = C;
    WALLOC(out->g, float, C);
    WALLOC(out->b, float, C);
    RUN(launch_f16_to_f32(g, out->g, C, s));
    RUN(launch_f16_to_f32(b, out->b, C, s));
    return 0;
}

int UNet::pack_linear(const std::string& prefix, int N, int K, bool bias, LinW* out, hipStream_t s) {
    const half_t* w = given(prefix + ".weight");
    NEED(w, prefix + ".weight");
    out->N = N;
    out->K = K;
    WALLOC(out->w, half_t, (size_t)N * K);
    LAVIE_HIP(hipMemcpyAsync(out->w, w, (size_t)N * K * sizeof(half_t), hipMemcpyDeviceToDevice, s));
    out->b = nullptr;
    if (bias) {
        const half_t* b = given(prefix + ".bias");
        NEED(b, prefix + ".bias");
        WALLOC(out->b, float, N);
        RUN(launch_f16_to_f32(b, out->b, N, s));
    }
    return 0;
}

int UNet::pack_resnet(ResnetW* r, hipStream_t s) {
    const std::string& p = r->prefix;
    RUN(pack_norm(p + ".norm1", r->cin, &r->n1, s));
    RUN(pack_norm(p + ".norm2", r->cout, &r->n2, s));
    const half_t* w1 = given(p + ".conv1.weight");
    const half_t* b1 = given(p + ".conv1.bias");
    const half_t* w2 = given(p + ".conv2.weight");
    const half_t* b2 = given(p + ".conv2.bias");
    NEED(w1, p + ".conv1.weight"); NEED(b1, p + ".conv1.bias"); NEED(w2, p + ".conv2.weight"); NEED(b2, p + ".conv2.bias");
    WALLOC(r->w1, half_t, (size_t)r->cout * 9 * r->cin);
    RUN(launch_pack_conv3x3(w1, r->w1, r->cout, r->cin, 9 * r->cin, 0, true, s));
    WALLOC(r->b1, float, r->cout);
    RUN(launch_f16_to_f32(b1, r->b1, r->cout, s));
    r->ldw2 = 9 * r->cout + (r->shortcut ? r->cin : 0);
    WALLOC(r->w2, half_t, (size_t)r->cout * r->ldw2);
    RUN(launch_pack_conv3x3(w2, r->w2, r->cout, r->cout, r->ldw2, 0, true, s));
    WALLOC(r->b2, float, r->cout);
    if (r->shortcut) {
        const half_t* ws = given(p + ".conv_shortcut.weight");
        const half_t* bs = given(p + ".conv_shortcut.bias");
        NEED(ws, p + ".conv_shortcut.weight"); NEED(bs, p + ".conv_shortcut.bias");
        RUN(launch_copy_rows(ws, r->cin, r->w2, r->ldw2, r->cout, r->cin, 9 * r->cout, s));
        RUN(launch_add_f16_to_f32(b2, bs, r->b2, r->cout, s));
    } else {
        RUN(launch_f16_to_f32(b2, r->b2, r->cout, s));
    }
    // fused time_emb_proj rows
    const half_t* wt = given(p + ".time_emb_proj.weight");
    const half_t* bt = given(p + ".time_emb_proj.bias");
    NEED(wt, p + ".time_emb_proj.weight"); NEED(bt, p + ".time_emb_proj.bias");
    LAVIE_HIP(hipMemcpyAsync(tproj_.w + (size_t)r->temb_off * tproj_.K, wt, (size_t)r->cout * tproj_.K * sizeof(half_t),
                             hipMemcpyDeviceToDevice, s));
    RUN(launch_f16_to_f32(bt, tproj_.b + r->temb_off, r->cout, s));
    return 0;
}

int UNet::pack_transformer(TransformerW* t, hipStream_t s) {
    const std::string& p = t->prefix;
    const std::string b = p + ".transformer_blocks.0";
    const int C = t->C, X = cfg_.cross_attention_dim;
    RUN(pack_norm(p + ".norm", C, &t->gn, s));
    RUN(pack_linear(p + ".proj_in", C, C, true, &t->pin, s));      // [C, C, 1, 1] == [C, C]
    RUN(pack_linear(p + ".proj_out", C, C, true, &t->pout, s));
    RUN(pack_norm(b + ".norm1", C, &t->ln1, s));
    RUN(pack_norm(b + ".norm2", C, &t->ln2, s));
    const std::string tname = cfg_.vsr_blocks ? "temporal" : "temp";
    RUN(pack_norm(b + ".norm_" + tname, C, &t->lnt, s));
    if (t->tres.present) RUN(pack_temporal_res(p + ".resblock_temporal", C, 3, &t->tres, s));
    RUN(pack_norm(b + ".norm3", C, &t->ln3, s));
    auto fuse = [&](const std::string& a, const char* const* names, int count, int K, half_t** out) -> int {
        WALLOC(*out, half_t, (size_t)count * C * K);
        for (int i = 0; i < count; ++i) {
            const std::string key = a + "." + names[i] + ".weight";
            const half_t* w = given(key);
            NEED(w, key);
            LAVIE_HIP(hipMemcpyAsync(*out + (size_t)i * C * K, w, (size_t)C * K * sizeof(half_t), hipMemcpyDeviceToDevice, s));
        }
        return 0;
    };
    static const char* const qkv[] = {"to_q", "to_k", "to_v"};
    static const char* const kv[] = {"to_k", "to_v"};
    static const char* const qonly[] = {"to_q"};
    if (t->attn1_cross) {
        RUN(fuse(b + ".attn1", qonly, 1, C, &t->wq1));
        RUN(fuse(b + ".attn1", kv, 2, X, &t->wkv1));
    } else {
        RUN(fuse(b + ".attn1", qkv, 3, C, &t->wqkv1));
    }
    RUN(pack_linear(b + ".attn1.to_out.0", C, C, true, &t->o1, s));
    RUN(fuse(b + ".attn2", qonly, 1, C, &t->wq2));
    RUN(fuse(b + ".attn2", kv, 2, X, &t->wkv2));
    RUN(pack_linear(b + ".attn2.to_out.0", C, C, true, &t->o2, s));
    RUN(fuse(b + ".attn_" + tname, qkv, 3, C, &t->wqkvt));
    RUN(pack_linear(b + ".attn_" + tname + ".to_out.0", C, C, true, &t->ot, s));
    if (!cfg_.temporal_plain) {
        const std::string key = b + ".attn_" + tname + ".time_rel_pos_bias.relative_attention_bias.weight";
        const half_t* e = given(key);
        NEED(e, key);
        const size_t n = (size_t)cfg_.rel_buckets * cfg_.heads;
        WALLOC(t->relemb, half_t, n);
        LAVIE_HIP(hipMemcpyAsync(t->relemb, e, n * sizeof(half_t), hipMemcpyDeviceToDevice, s));
        // `...rotary_emb.freqs` is accepted by set_param and ignored: angles are always derived in
        // fp32 from rotary_dim (SURVEY.md §8c decision; an fp16 copy of freqs would be lossy).
    }
    {
        const half_t* w = given(b + ".ff.net.0.proj.weight");
        const half_t* bias = given(b + ".ff.net.0.proj.bias");
        NEED(w, b + ".ff.net.0.proj.weight"); NEED(bias, b + ".ff.net.0.proj.bias");
        t->ff1.N = 8 * C; t->ff1.K = C;
        WALLOC(t->ff1.w, half_t, (size_t)8 * C * C);
        WALLOC(t->ff1.b, float, 8 * C);
        RUN(launch_pack_geglu_rows(w, t->ff1.w, 8 * C, C, s));
        RUN(launch_pack_geglu_bias(bias, t->ff1.b, 8 * C, s));
    }
    RUN(pack_linear(b + ".ff.net.2", C, 4 * C, true, &t->ff2, s));
    if (!cfg_.temporal_plain && !cfg_.vsr_blocks && temporal_block_supported(C, cfg_.heads, 16, cfg_.rotary_dim)) {
        // fused temporal sub-block (clips of 16 frames): q / k / v / to_out weights in MFMA-fragment order
        const half_t* wq = given(b + ".attn_" + tname + ".to_q.weight");
        const half_t* wk = given(b + ".attn_" + tname + ".to_k.weight");
        const half_t* wv = given(b + ".attn_" + tname + ".to_v.weight");
        const half_t* wo = given(b + ".attn_" + tname + ".to_out.0.weight");
        WALLOC(t->tb_img, half_t, temporal_block_image_bytes(C) / sizeof(half_t));
        RUN(pack_temporal_block(wq, wk, wv, wo, C, t->tb_img, s));
    }
    if (!t->attn1_cross && !cfg_.vsr_blocks && cross_block_supported(C, cfg_.heads, 1, 16)) {
        // fused text cross-attention sub-block: attn1.to_out / attn2.to_q / attn2.to_out in MFMA-fragment order; the K / V pieces
        // of the image are bound per context (cache_context)
        const half_t* wo1 = given(b + ".attn1.to_out.0.weight");
        const half_t* wq2 = given(b + ".attn2.to_q.weight");
        const half_t* wo2 = given(b + ".attn2.to_out.0.weight");
        NEED(wo1, b + ".attn1.to_out.0.weight"); NEED(wq2, b + ".attn2.to_q.weight"); NEED(wo2, b + ".attn2.to_out.0.weight");
        WALLOC(t->xb_tmpl, half_t, cross_block_image_bytes(C) / sizeof(half_t));
        RUN(pack_cross_block(wo1, wq2, wo2, C, t->xb_tmpl, s));
    }
    if (!t->attn1_cross && !cfg_.vsr_blocks && proj_qkv_supported(C)) {     // fused GroupNorm -> proj_in -> norm1 -> q|k|v kernel
        const half_t* wpin = given(p + ".proj_in.weight");
        NEED(wpin, p + ".proj_in.weight");
        WALLOC(t->pq_img, half_t, proj_qkv_image_bytes(C) / sizeof(half_t));
        RUN(pack_proj_qkv(wpin, t->wqkv1, C, t->pq_img, s));        // wqkv1: to_q | to_k | to_v rows, fused above
    }
    if (geglu_mlp_supported(C)) {      // fused norm3 -> feed-forward -> residual kernel: weight image in MFMA-fragment order
        const half_t* w1 = given(b + ".ff.net.0.proj.weight");
        const half_t* b1 = given(b + ".ff.net.0.proj.bias");
        const half_t* w2 = given(b + ".ff.net.2.weight");
        NEED(w2, b + ".ff.net.2.weight");
        WALLOC(t->ff_img, half_t, geglu_mlp_image_bytes(C) / sizeof(half_t));
        WALLOC(t->ff_b1img, float, geglu_mlp_bias_floats(C));
        RUN(pack_geglu_mlp(w1, b1, w2, C, t->ff_img, t->ff_b1img, s));
    }

    // LayerNorm-folded projections (norm1 -> attn1 qkv, norm2 -> attn2 q, norm_temp -> attn_temp qkv, norm3 -> GEGLU)
    auto fold = [&](const half_t* W, const NormW& ln, const half_t* bias, int N, half_t** Wf, float** sv, float** bv) -> int {
        WALLOC(*Wf, half_t, (size_t)N * C);
        WALLOC(*sv, float, N);
        WALLOC(*bv, float, N);
        return launch_ln_fold(W, ln.g, ln.b, bias, *Wf, *sv, *bv, N, C, s);
    };
    if (t->attn1_cross) RUN(fold(t->wq1, t->ln1, nullptr, C, &t->f_q1, &t->s_q1, &t->b_q1));
    else RUN(fold(t->wqkv1, t->ln1, nullptr, 3 * C, &t->f_qkv1, &t->s_qkv1, &t->b_qkv1));
    RUN(fold(t->wq2, t->ln2, nullptr, C, &t->f_q2, &t->s_q2, &t->b_q2));
    RUN(fold(t->wqkvt, t->lnt, nullptr, 3 * C, &t->f_qkvt, &t->s_qkvt, &t->b_qkvt));
    {
        // GEGLU: fold in the checkpoint's row order, then apply the value/gate interleave to W', s and b'
        const half_t* w = given(b + ".ff.net.0.proj.weight");
        const half_t* bias = given(b + ".ff.net.0.proj.bias");
        half_t* tmpW; float* tmps; float* tmpb;
        RUN(fold(w, t->ln3, bias, 8 * C, &tmpW, &tmps, &tmpb));
        WALLOC(t->f_ff1, half_t, (size_t)8 * C * C);
        WALLOC(t->s_ff1, float, 8 * C);
        WALLOC(t->b_ff1, float, 8 * C);
        RUN(launch_pack_geglu_rows(tmpW, t->f_ff1, 8 * C, C, s));
        RUN(launch_pack_geglu_vec(tmps, t->s_ff1, 8 * C, s));
        RUN(launch_pack_geglu_vec(tmpb, t->b_ff1, 8 * C, s));
    }
    return 0;
}

int UNet::pack_temporal_res(const std::string& prefix, int C, int taps1, TemporalResW* out, hipStream_t s) {
    out->present = true;
    out->taps1 = taps1;
    RUN(pack_norm(prefix + ".norm1", C, &out->n1, s));
    RUN(pack_norm(prefix + ".norm2", C, &out->n2, s));
    const half_t* w1 = given(prefix + ".conv1.weight");
    const half_t* b1 = given(prefix + ".conv1.bias");
    const half_t* w2 = given(prefix + ".conv2.weight");
    const half_t* b2 = given(prefix + ".conv2.bias");
    NEED(w1, prefix + ".conv1.weight"); NEED(b1, prefix + ".conv1.bias");
    NEED(w2, prefix + ".conv2.weight"); NEED(b2, prefix + ".conv2.bias");
    WALLOC(out->w1, half_t, (size_t)C * taps1 * C);
    WALLOC(out->w2, half_t, (size_t)C * 3 * C);
    WALLOC(out->b1, float, C);
    WALLOC(out->b2, float, C);
    RUN(launch_pack_conv_taps(w1, out->w1, C, C, taps1, taps1 * C, 0, true, s));
    RUN(launch_pack_conv_taps(w2, out->w2, C, C, 3, 3 * C, 0, true, s));
    RUN(launch_f16_to_f32(b1, out->b1, C, s));
    RUN(launch_f16_to_f32(b2, out->b2, C, s));
    return 0;
}

int UNet::pack_sampler(const std::string& prefix, int C, SamplerW* out, hipStream_t s, bool up) {
    const half_t* w = given(prefix + ".weight");
    const half_t* b = given(prefix + ".bias");
    NEED(w, prefix + ".weight"); NEED(b, prefix + ".bias");
    out->C = C;
    WALLOC(out->w, half_t, (size_t)C * 9 * C);
    WALLOC(out->b, float, C);
    RUN(launch_pack_conv3x3(w, out->w, C, C, 9 * C, 0, true, s));
    RUN(launch_f16_to_f32(b, out->b, C, s));
    if (up && C % 160 == 0) {          // conv(nearest_x2(x)) as four 2x2 convs on x: weights of coinciding taps summed once, here
        WALLOC(out->wpar, half_t, (size_t)4 * C * 4 * C);
        RUN(launch_pack_conv3x3_parity(w, out->wpar, C, C, s));
    }
    return 0;
}

int UNet::finalize(hipStream_t s) {
    RUN(validate_config());
    LAVIE_CHECK(!finalized_, "finalize: already finalized");
    const lavie_unet_config& c = cfg_;
    const int C0 = c.block_out_channels[0];
    WALLOC(zero_page_, half_t, 256);
    LAVIE_HIP(hipMemsetAsync(zero_page_, 0, 512, s));
    {   // conv_in / conv_out / conv_norm_out
        const half_t* w = given("conv_in.weight");
        const half_t* b = given("conv_in.bias");
        NEED(w, std::string("conv_in.weight")); NEED(b, std::string("conv_in.bias"));
        WALLOC(conv_in_w_, half_t, (size_t)9 * c.in_channels * C0);
        WALLOC(conv_in_b_, float, C0);
        RUN(launch_pack_conv_in(w, conv_in_w_, C0, c.in_channels, s));
        RUN(launch_f16_to_f32(b, conv_in_b_, C0, s));
        const half_t* wo = given("conv_out.weight");
        const half_t* bo = given("conv_out.bias");
        NEED(wo, std::string("conv_out.weight")); NEED(bo, std::string("conv_out.bias"));
        WALLOC(conv_out_w_, half_t, (size_t)c.out_channels * 9 * C0);
        WALLOC(conv_out_b_, float, c.out_channels);
        RUN(launch_pack_conv3x3(wo, conv_out_w_, c.out_channels, C0, 9 * C0, 0, false, s));
        RUN(launch_f16_to_f32(bo, conv_out_b_, c.out_channels, s));
        RUN(pack_norm("conv_norm_out", C0, &norm_out_, s));
    }
    RUN(pack_linear("time_embedding.linear_1", C0 * 4, C0, true, &time1_, s));
    RUN(pack_linear("time_embedding.linear_2", C0 * 4, C0 * 4, true, &time2_, s));
    WALLOC(tproj_.w, half_t, (size_t)tproj_.N * tproj_.K);
    WALLOC(tproj_.b, float, tproj_.N);
    for (ResnetW& r : resnets_) RUN(pack_resnet(&r, s));
    for (TransformerW& t : transformers_) RUN(pack_transformer(&t, s));
    for (TemporalModuleW& m : tmods_) {
        RUN(pack_temporal_res(m.prefix + ".resblocks_3d_t", m.C, 5, &m.t, s));
        const half_t* wt = given(m.prefix + ".resblocks_3d_t.time_emb_proj.weight");
        const half_t* bt = given(m.prefix + ".resblocks_3d_t.time_emb_proj.bias");
        NEED(wt, m.prefix + ".resblocks_3d_t.time_emb_proj.weight"); NEED(bt, m.prefix + ".resblocks_3d_t.time_emb_proj.bias");
        LAVIE_HIP(hipMemcpyAsync(tproj_.w + (size_t)m.t_temb_off * tproj_.K, wt, (size_t)m.C * tproj_.K * sizeof(half_t),
                                 hipMemcpyDeviceToDevice, s));
        RUN(launch_f16_to_f32(bt, tproj_.b + m.t_temb_off, m.C, s));
        RUN(pack_resnet(&m.s, s));
        RUN(pack_linear(m.prefix + ".shift_conv", m.C, m.C, true, &m.shift, s));
    }
    if (c.num_class_embeds > 0) {
        const half_t* e = given("class_embedding.weight");
        NEED(e, std::string("class_embedding.weight"));
        const size_t n = (size_t)c.num_class_embeds * C0 * 4;
        WALLOC(class_emb_, half_t, n);
        LAVIE_HIP(hipMemcpyAsync(class_emb_, e, n * sizeof(half_t), hipMemcpyDeviceToDevice, s));
    }
    const int L = c.num_levels;
    downs_.resize(L > 1 ? L - 1 : 0);
    ups_.resize(L > 1 ? L - 1 : 0);
    for (int l = 0; l + 1 < L; ++l)
        RUN(pack_sampler("down_blocks." + std::to_string(l) + ".downsamplers.0.conv", c.block_out_channels[l], &downs_[l], s));
    for (int i = 0; i + 1 < L; ++i)
        RUN(pack_sampler("up_blocks." + std::to_string(i) + ".upsamplers.0.conv", c.block_out_channels[L - 1 - i], &ups_[i], s, /*up=*/true));
    finalized_ = true;
    return 0;
}

int UNet::ensure_tables(int F, hipStream_t s) {
    auto it = tables_.find(F);
    if (it != tables_.end()) {          // built before: switching clip length costs nothing and allocates nothing
        cur_tables_ = &it->second;
        return 0;
    }
    FrameTables ft;
    const int rp = cfg_.rotary_dim / 2;
    std::vector<float> hc((size_t)F * rp), hs((size_t)F * rp);
    for (int f = 0; f < F; ++f)
        for (int k = 0; k < rp; ++k) {
            const float freq = (float)pow(10000.0, -2.0 * k / cfg_.rotary_dim);
            const float ang = (float)f * freq;
            hc[(size_t)f * rp + k] = (float)cos((double)ang);
            hs[(size_t)f * rp + k] = (float)sin((double)ang);
        }
    std::vector<int> hb((size_t)F * F);
    relpos_bucket_table(F, cfg_.rel_buckets, cfg_.rel_max_distance, hb.data());
    int* buckets_dev = nullptr;
    WALLOC(ft.rot_cos, float, (size_t)F * rp);
    WALLOC(ft.rot_sin, float, (size_t)F * rp);
    WALLOC(buckets_dev, int, (size_t)F * F);
    // pageable host memory: these copies complete before returning (once per distinct F in the handle's lifetime)
    LAVIE_HIP(hipMemcpy(ft.rot_cos, hc.data(), hc.size() * sizeof(float), hipMemcpyHostToDevice));
    LAVIE_HIP(hipMemcpy(ft.rot_sin, hs.data(), hs.size() * sizeof(float), hipMemcpyHostToDevice));
    LAVIE_HIP(hipMemcpy(buckets_dev, hb.data(), hb.size() * sizeof(int), hipMemcpyHostToDevice));
    ft.relbias.assign(transformers_.size(), nullptr);
    if (cfg_.temporal_plain) {
        // plain softmax(scale q k^T) v over frames (interpolation/models/attention.py:268-289): one shared all-zero bias
        // table (s + 0.0f == s exactly) and rot_dim = 0 in the kernel parameters
        float* zero = nullptr;
        WALLOC(zero, float, (size_t)cfg_.heads * F * F);
        LAVIE_HIP(hipMemsetAsync(zero, 0, (size_t)cfg_.heads * F * F * sizeof(float), s));
        for (size_t i = 0; i < transformers_.size(); ++i) ft.relbias[i] = zero;
    } else {
        for (size_t i = 0; i < transformers_.size(); ++i) {
            WALLOC(ft.relbias[i], float, (size_t)cfg_.heads * F * F);
            RUN(launch_fill_relpos_bias(transformers_[i].relemb, buckets_dev, ft.relbias[i], cfg_.heads, F, s));
        }
    }
    LAVIE_HIP(hipStreamSynchronize(s));
    cur_tables_ = &tables_.emplace(F, std::move(ft)).first->second;
    return 0;
}

// ------------------------------------------------------------------ forward
struct FwdCtx {
    hipStream_t s;
    DeviceArena* ws;
    bool dry;               // plan only: allocate, launch nothing
    int B, F, ctx_len;
    float* gn_ws;           // GroupNorm scratch, gn_workspace_floats(B*F, groups) floats
    const int* labels = nullptr;   // VSR noise level per video (host), num_class_embeds > 0
};

#define LAUNCH(expr)              \
    do {                          \
        if (!c.dry) RUN(expr);    \
    } while (0)

// Development aid (LAVIE_DEBUG_TRACE_HALVES=1): after the named step, wait for the stream and report whether the two batch halves of
// the rows are bit-equal (meaningful for a forward fed identical halves) — localises an irreproducible kernel
static bool trace_halves_on() {
    static const bool on = [] { const char* e = getenv("LAVIE_DEBUG_TRACE_HALVES"); return e && e[0] == '1'; }();
    return on;
}
static int trace_halves(const FwdCtx& c, const char* what, const void* p, size_t rows, size_t row_bytes, int line, int elt) {
    if (c.dry || !trace_halves_on() || rows % 2 != 0) return 0;
    LAVIE_HIP(hipStreamSynchronize(c.s));
    std::vector<unsigned char> h(rows * row_bytes);
    LAVIE_HIP(hipMemcpy(h.data(), p, h.size(), hipMemcpyDeviceToHost));
    const size_t half = rows / 2 * row_bytes;
    size_t bad = 0, first = 0, last = 0, bad_rows = 0, cmin = row_bytes, cmax = 0;
    for (size_t r = 0; r < rows / 2; ++r) {
        bool rb = false;
        for (size_t j = 0; j < row_bytes; ++j) {
            const size_t i = r * row_bytes + j;
            if (h[i] != h[half + i]) {
                if (!bad) first = i;
                last = i; ++bad; rb = true;
                if (j < cmin) cmin = j;
                if (j > cmax) cmax = j;
            }
        }
        bad_rows += rb;
    }
    fprintf(stderr, "[halves] line %d %-28s rows %zu x %zu B: %s", line, what, rows, row_bytes, bad ? "DIFFER" : "equal");
    if (bad) fprintf(stderr, " (%zu bytes in %zu rows, rows %zu .. %zu, col bytes %zu .. %zu)", bad, bad_rows, first / row_bytes, last / row_bytes, cmin, cmax);
    fprintf(stderr, "\n");
    if (bad && elt == 2) {          // the first few differing fp16 pairs
        int shown = 0;
        for (size_t i = 0; i + 1 < half && shown < 8; i += 2) {
            if (h[i] == h[half + i] && h[i + 1] == h[half + i + 1]) continue;
            _Float16 a, b;
            memcpy(&a, &h[i], 2); memcpy(&b, &h[half + i], 2);
            fprintf(stderr, "          row %zu col %zu: %.6g vs %.6g (bits %02x%02x %02x%02x)\n", i / row_bytes, (i % row_bytes) / 2, (double)a, (double)b,
                    h[i + 1], h[i], h[half + i + 1], h[half + i]);
            ++shown;
        }
    }
    return 0;
}
#define TRACE(what, ptr, rows, cols) RUN(trace_halves(c, what, ptr, (size_t)(rows), (size_t)(cols) * sizeof(*(ptr)), __LINE__, (int)sizeof(*(ptr))))

#define WS(var, type, count)                                                                                     \
    type* var = (type*)c.ws->alloc((size_t)(count) * sizeof(type));                                              \
    LAVIE_CHECK(var != nullptr, "workspace exhausted: call lavie_unet_prepare for this shape (needed %zu more B)", \
                (size_t)(count) * sizeof(type))

struct LnFold {            // consumer side of a folded LayerNorm
    float* stats;          // [M, 2] (mean, rstd) of the rows
    const float* s;        // row sums of the folded weights
    // round 4: the producer leaves its partials un-finalized; a consumer on a kernel with the shared epilogue folds them itself
    // (IgemmParams::ln_partials), the persistent kernel's consumers finalize on demand, once
    const float* partials = nullptr;   // [M, slots, 2] of the rows' latest producer, or nullptr when `stats` was written directly
    int slots = 0, row_len = 0, rows = 0;
    bool final_ok = false;             // `stats` holds the finalized (mean, rstd) of the current rows
};
struct RowStat {           // producer side: partials [M, slots, 2] -> (mean, rstd) [M, 2]
    float* partials;
    float* mean_rstd;
    int slots;
    LnFold* sink = nullptr;            // told where the partials are (no finalize launch); nullptr = finalize at once
};

// GroupNorm statistics from the producing kernel (igemm.h colstat_out, round 4).  Every tensor a GroupNorm may read gets a small
// buffer next to it (same workspace lifetime); the launcher that writes the tensor fills it and describes it in a GnColStat that
// travels WITH the tensor through the wiring below (explicitly, never keyed by address: workspace addresses are reused).
static size_t colstat_floats(size_t M, int C) { return (M / COLSTAT_REDUCE_ROWS + 8) * (size_t)C * 2; }
static bool colstat_on() { return (fused_mask() & 32) != 0; }
// plan: which block height will the launch of `p` write (0 = none), and describe the result
static void colstat_plan(IgemmParams& p, bool gather, float* buf, GnColStat* out) {
    if (out) *out = GnColStat();
    p.colstat_out = nullptr;
    p.colstat_rows = 0;
    if (!buf || !out || !colstat_on()) return;
    const int rows = igemm_colstat_rows(p, gather, EPI_LINEAR);
    if (rows <= 0) return;
    p.colstat_out = buf;
    p.colstat_rows = rows;
    out->partials = buf;
    out->C = p.N;
    out->rows = rows;
    out->span = igemm_colstat_span(p, gather, rows);
    if (p.par_ups && p.splits == 1) { out->nsets = 4; out->set_blocks = p.M / 4 / rows; }     // source-row blocks per output parity
    else { out->nsets = 1; out->set_blocks = cdiv(p.M, rows); }                               // (split-K: the reduce kernel walks output rows)
}

static int linear(FwdCtx& c, const half_t* A, int lda, const half_t* W, const float* bias, int N, int K, const half_t* R,
                  half_t* C, int ldc, int M, int epilogue = EPI_LINEAR, LnFold* fold = nullptr,
                  const RowStat* rowstat = nullptr, int ldw = 0, float* cs_buf = nullptr, GnColStat* cs_out = nullptr) {
    const bool unsplit = fold != nullptr || rowstat != nullptr;
    if (cs_out) *cs_out = GnColStat();
    if (c.dry) {   // plan the split-K slab so that prepare() sizes the workspace for it
        const int s = unsplit ? 1 : igemm_plan_splits(M, N, K / IGEMM_BK, epilogue);
        if (s > 1) {
            const size_t mark = c.ws->mark();
            (void)c.ws->alloc((size_t)s * M * N * sizeof(float));
            c.ws->release(mark);
        }
        return 0;
    }
    IgemmParams p;
    memset(&p, 0, sizeof(p));
    p.A = A; p.lda = lda; p.W = W; p.ldw = ldw > 0 ? ldw : K; p.C = C; p.ldc = ldc; p.bias = bias; p.R = R; p.ldr = ldc;
    p.M = M; p.N = N; p.nk = K / IGEMM_BK;
    LAVIE_CHECK(K % IGEMM_BK == 0, "linear: K=%d must be a multiple of %d", K, IGEMM_BK);
    p.splits = unsplit ? 1 : igemm_plan_splits(M, N, p.nk, epilogue);
    p.rowstat_out = rowstat ? rowstat->partials : nullptr;
    p.rowstat_cols = rowstat ? N / rowstat->slots : 0;
    if (fold) {
        p.ln_s = fold->s;
        const bool ppx = igemm_takes_ppx(M, N, p.nk, epilogue);
        if (fold->partials && !fold->final_ok && !ppx && fold->rows >= M && (fused_mask() & 128)) {          // fold in this GEMM's epilogue: no finalize launch
            p.ln_partials = fold->partials; p.ln_slots = fold->slots; p.ln_inv_len = 1.0f / (float)fold->row_len; p.ln_eps = 1e-5f;
        } else {
            if (fold->partials && !fold->final_ok) {       // the persistent kernel stages finished rows: finalize now, once
                RUN(launch_rowstat_finalize(fold->partials, fold->slots, fold->rows, fold->row_len, 1e-5f, fold->stats, c.s));
                fold->final_ok = true;
            }
            p.ln_stats = fold->stats;
        }
    }
    if (epilogue == EPI_LINEAR && ldc == N) colstat_plan(p, false, cs_buf, cs_out);
    const size_t mark = c.ws->mark();
    if (p.splits > 1) {
        p.slab = (float*)c.ws->alloc((size_t)p.splits * M * N * sizeof(float));
        LAVIE_CHECK(p.slab != nullptr, "workspace exhausted (split-K slab)");
    }
    int rc = launch_igemm(p, false, epilogue, c.s);
    if (rc == 0 && rowstat) {
        if (rowstat->sink) {       // deferred: the consumer decides (see above)
            rowstat->sink->partials = rowstat->partials; rowstat->sink->slots = rowstat->slots; rowstat->sink->row_len = N;
            rowstat->sink->rows = M; rowstat->sink->final_ok = false;
        } else {
            rc = launch_rowstat_finalize(rowstat->partials, rowstat->slots, M, N, 1e-5f, rowstat->mean_rstd, c.s);
        }
    }
    c.ws->release(mark);
    return rc;
}

// 3x3 conv (pad 1) over `nsrc` channel-concatenated sources, plus optional centre-tap shortcut sources.
static int conv3x3(FwdCtx& c, const half_t* const* src, const int* srcC, int nsrc, const half_t* const* sc, const int* scC,
                   int nsc, const half_t* W, int ldw, const float* bias, const float* bias2, int ldb2, int rows_per_batch,
                   const half_t* R, half_t* y, int NI, int Hi, int Wi, int Cout, int stride, int ups, const half_t* zero,
                   float* cs_buf = nullptr, GnColStat* cs_out = nullptr) {
    IgemmParams p;
    memset(&p, 0, sizeof(p));
    p.W = W; p.ldw = ldw; p.C = y; p.ldc = Cout; p.bias = bias; p.bias2 = bias2; p.ldb2 = ldb2;
    p.rows_per_batch = rows_per_batch; p.R = R; p.ldr = Cout;
    p.Hi = Hi; p.Wi = Wi; p.stride = stride; p.ups = ups;
    p.Ho = ups ? Hi * 2 : (Hi - 1) / stride + 1;
    p.Wo = ups ? Wi * 2 : (Wi - 1) / stride + 1;
    p.M = NI * p.Ho * p.Wo;
    p.N = Cout;
    p.zero = zero;
    int ns = 0, nk = 0;
    LAVIE_CHECK(nsrc + nsc <= IGEMM_MAX_SEG, "conv3x3: too many K segments");
    for (int i = 0; i < nsrc; ++i) {
        LAVIE_CHECK(srcC[i] % IGEMM_BK == 0, "conv3x3: channel count %d must be a multiple of %d", srcC[i], IGEMM_BK);
        IgemmSeg& sg = p.seg[ns++];
        sg.src = src[i]; sg.C = srcC[i]; sg.c0 = 0; sg.nchunks = srcC[i] / IGEMM_BK; sg.ntaps = 9;
        nk += 9 * sg.nchunks;
    }
    for (int i = 0; i < nsc; ++i) {
        LAVIE_CHECK(scC[i] % IGEMM_BK == 0 && stride == 1 && ups == 0, "conv3x3: bad shortcut source");
        IgemmSeg& sg = p.seg[ns++];
        sg.src = sc[i]; sg.C = scC[i]; sg.c0 = 0; sg.nchunks = scC[i] / IGEMM_BK; sg.ntaps = 1;
        nk += sg.nchunks;
    }
    LAVIE_CHECK(nk * IGEMM_BK <= ldw, "conv3x3: weight row length %d is shorter than the gathered K %d", ldw, nk * IGEMM_BK);
    p.nseg = ns;
    p.nk = nk;
    p.splits = igemm_plan_splits_gather(p);        // also picks the kernel: same inputs -> same choice in the dry run
    colstat_plan(p, true, cs_buf, cs_out);
    const size_t mark = c.ws->mark();
    if (p.splits > 1) {
        p.slab = (float*)c.ws->alloc((size_t)p.splits * p.M * p.N * sizeof(float));
        if (!c.dry) LAVIE_CHECK(p.slab != nullptr, "workspace exhausted (split-K slab)");
    }
    const int rc = c.dry ? 0 : launch_igemm(p, true, EPI_LINEAR, c.s);
    c.ws->release(mark);
    return rc;
}

// Whether a plain 3x3 conv Cin -> Cout (stride 1, one source) at this geometry goes to the halo-patch kernel (the planner's rule)
static bool conv3x3_takes_patch_kernel(int NI, int H, int W, int Cin, int Cout) {
    IgemmParams p;
    memset(&p, 0, sizeof(p));
    p.Hi = p.Ho = H; p.Wi = p.Wo = W; p.stride = 1;
    p.M = NI * H * W; p.N = Cout; p.nseg = 1;
    p.seg[0].C = Cin; p.seg[0].nchunks = Cin / IGEMM_BK; p.seg[0].ntaps = 9;
    p.nk = 9 * p.seg[0].nchunks;
    p.splits = igemm_plan_splits_gather(p);
    return igemm_patch_planned(p);
}

// (taps,1,1) temporal conv over the frame axis of token rows [(b f d), C] (IgemmParams temporal mode; 128-row kernel).
static int tconv(FwdCtx& c, const half_t* x, int C, const half_t* W, const float* bias, const float* bias2, int ldb2,
                 const half_t* R, half_t* y, int D, int Cout, int taps, const half_t* zero, float* cs_buf = nullptr,
                 GnColStat* cs_out = nullptr) {
    IgemmParams p;
    memset(&p, 0, sizeof(p));
    p.W = W; p.ldw = taps * C; p.C = y; p.ldc = Cout; p.bias = bias; p.bias2 = bias2; p.ldb2 = ldb2;
    p.rows_per_batch = c.F * D; p.R = R; p.ldr = Cout;
    p.tframes = c.F; p.tpix = D;
    p.Hi = p.Ho = 1; p.Wi = p.Wo = 1; p.stride = 1;
    p.M = c.B * c.F * D; p.N = Cout; p.zero = zero;
    LAVIE_CHECK(C % IGEMM_BK == 0, "temporal conv: channel count %d must be a multiple of %d", C, IGEMM_BK);
    IgemmSeg& sg = p.seg[0];
    sg.src = x; sg.C = C; sg.c0 = 0; sg.nchunks = C / IGEMM_BK; sg.ntaps = taps;
    p.nseg = 1;
    p.nk = taps * sg.nchunks;
    p.splits = igemm_plan_splits_gather(p);
    colstat_plan(p, true, cs_buf, cs_out);          // (round 4) the GroupNorm behind a temporal conv folds its epilogue's sums too
    const size_t mark = c.ws->mark();
    if (p.splits > 1) {
        p.slab = (float*)c.ws->alloc((size_t)p.splits * p.M * p.N * sizeof(float));
        if (!c.dry) LAVIE_CHECK(p.slab != nullptr, "workspace exhausted (split-K slab)");
    }
    const int rc = c.dry ? 0 : launch_igemm(p, true, EPI_LINEAR, c.s);
    c.ws->release(mark);
    return rc;
}

// ResnetBlock3DCNN (vsr/models/resnet.py:283-315): y = x + conv2(silu(gn(conv1(silu(gn(x))) + temb))); GroupNorm statistics
// span the whole video (5-D input), eps 1e-6; bias2 = the block's rows of the fused time-embedding projection or nullptr
// (attention.py:350: temb_channels=None).  y may alias x.
int UNet::run_temporal_res(FwdCtx& c, const TemporalResW& r, const half_t* x, half_t* y, int C, int D, const float* bias2,
                           int ldb2, const GnColStat* x_cs, float* y_csbuf, GnColStat* y_cs) {
    const size_t M = (size_t)c.B * c.F * D;
    const int P = c.F * D;
    const size_t mark = c.ws->mark();
    WS(nrm, half_t, M * C);
    WS(h1, half_t, M * C);
    WS(h1cs, float, colstat_floats(M, C));
    GnColStat h1_cs;
    // x_cs (may be null / empty): statistics the producer of x left; y_csbuf / y_cs (may be null): where conv2's go.  y may alias x
    // and y_csbuf the buffer x_cs points into: norm1's fold has read it (stream order) long before conv2's epilogue writes it
    LAUNCH(launch_group_norm(x, C, nullptr, 0, c.B, P, cfg_.norm_groups, r.n1.g, r.n1.b, 1e-6f, true, c.gn_ws, nrm, c.s, x_cs));
    TRACE("tres.norm1", nrm, M, C);
    RUN(tconv(c, nrm, C, r.w1, r.b1, bias2, ldb2, nullptr, h1, D, C, r.taps1, zero_page_, h1cs, &h1_cs));
    TRACE("tres.conv1", h1, M, C);
    LAUNCH(launch_group_norm(h1, C, nullptr, 0, c.B, P, cfg_.norm_groups, r.n2.g, r.n2.b, 1e-6f, true, c.gn_ws, nrm, c.s, &h1_cs));
    TRACE("tres.norm2", nrm, M, C);
    if (y_cs) *y_cs = GnColStat();
    RUN(tconv(c, nrm, C, r.w2, r.b2, nullptr, 0, x, y, D, C, 3, zero_page_, y_csbuf, y_cs));
    TRACE("tres.conv2", y, M, C);
    c.ws->release(mark);
    return 0;
}

// TemporalModule3D.forward (vsr/models/temporal_module.py:153-178): y = x + shift_conv(resblocks_3d_s(resblocks_3d_t(x))).
// y must not alias x when x is still referenced as a skip tensor.
int UNet::run_temporal_module(FwdCtx& c, const TemporalModuleW& m, const half_t* x, half_t* y, const float* tproj,
                              int ld_tproj, int H, int W, const GnColStat* x_cs, float* y_csbuf, GnColStat* y_cs) {
    const int C = m.C, D = H * W;
    const size_t M = (size_t)c.B * c.F * D;
    const size_t mark = c.ws->mark();
    WS(h1, half_t, M * C);
    WS(h2, half_t, M * C);
    WS(h1cs, float, colstat_floats(M, C));
    GnColStat h1_cs;
    RUN(run_temporal_res(c, m.t, x, h1, C, D, tproj + m.t_temb_off, ld_tproj, x_cs, h1cs, &h1_cs));
    RUN(run_resnet(c, m.s, h1, C, nullptr, 0, tproj + m.s.temb_off, ld_tproj, h2, H, W, &h1_cs));
    RUN(linear(c, h2, C, m.shift.w, m.shift.b, C, C, x, y, C, (int)M, EPI_LINEAR, nullptr, nullptr, 0, y_csbuf, y_cs));
    TRACE("tmodule.shift", y, M, C);
    c.ws->release(mark);
    return 0;
}

int UNet::run_resnet(FwdCtx& c, const ResnetW& r, const half_t* x1, int C1, const half_t* x2, int C2, const float* tproj,
                     int ld_tproj, half_t* y, int H, int W, const GnColStat* cs1, const GnColStat* cs2, float* y_csbuf,
                     GnColStat* y_cs) {
    LAVIE_CHECK(C1 + C2 == r.cin, "resnet %s: got %d+%d input channels, expected %d", r.prefix.c_str(), C1, C2, r.cin);
    const int G = cfg_.norm_groups;
    const int NI = c.B * c.F;
    const int P = c.F * H * W;            // rows sharing GroupNorm statistics: the whole video (resnet.py:180)
    const size_t M = (size_t)NI * H * W;
    const size_t mark = c.ws->mark();
    WS(nrm, half_t, M * r.cin);
    WS(h1, half_t, M * r.cout);
    WS(n2, half_t, M * r.cout);
    WS(h1cs, float, colstat_floats(M, r.cout));      // GroupNorm statistics of h1, written by conv1's epilogue
    GnColStat h1_cs;
    if (y_cs) *y_cs = GnColStat();
    const float eps = r.eps > 0.f ? r.eps : cfg_.norm_eps;
    // norm1: statistics from the producers of x1 / x2 where they left them (cs1 / cs2), else the statistics pass
    LAUNCH(launch_group_norm(x1, C1, x2, C2, c.B, P, G, r.n1.g, r.n1.b, eps, true, c.gn_ws, nrm, c.s, cs1, cs2));
    TRACE("resnet.norm1", nrm, M, r.cin);
    {
        const half_t* src[1] = {nrm};
        const int srcC[1] = {r.cin};
        RUN(conv3x3(c, src, srcC, 1, nullptr, nullptr, 0, r.w1, 9 * r.cin, r.b1, tproj, ld_tproj, P, nullptr, h1, NI, H, W,
                    r.cout, 1, 0, zero_page_, h1cs, &h1_cs));
    }
    TRACE("resnet.conv1", h1, M, r.cout);
    LAUNCH(launch_group_norm(h1, r.cout, nullptr, 0, c.B, P, G, r.n2.g, r.n2.b, eps, true, c.gn_ws, n2, c.s, &h1_cs));
    TRACE("resnet.norm2", n2, M, r.cout);
    {
        const half_t* src[1] = {n2};
        const int srcC[1] = {r.cout};
        const half_t* sc[2] = {x1, x2};
        const int scC[2] = {C1, C2};
        const int nsc = r.shortcut ? (x2 ? 2 : 1) : 0;
        // The 1x1 conv_shortcut rides in conv2 as extra centre-tap K segments — which keeps conv2 off the halo-patch kernel (it
        // takes 9-tap segments only; the ping-pong kernel stages 60 LDS-DMA pieces per K-tile against 26).  Where the planner
        // would give the 3x3 part to the patch kernel, the shortcut can run as its own GEMM on the same packed weight rows (a
        // column window) and come back through conv2's residual operand.  Measured (round 3, profiles/r03_ab_split_shortcut.txt):
        // forward 21.33 -> 21.41 ms, i.e. the extra [M, Cout] round trip and launch cost more than the patch kernel gains:
        // OFF by default (bit 3 of lavie_debug_fused_mask turns it on for A/B).
        const bool split_sc = r.shortcut && (fused_mask() & 8) && conv3x3_takes_patch_kernel(NI, H, W, r.cout, r.cout);
        if (split_sc) {
            WS(scy, half_t, M * r.cout);
            const half_t* wsc = r.w2 + 9 * r.cout;
            if (nsc == 1) RUN(linear(c, x1, C1, wsc, nullptr, r.cout, C1, nullptr, scy, r.cout, (int)M, EPI_LINEAR, nullptr, nullptr, r.ldw2));
            else RUN(conv3x3(c, nullptr, nullptr, 0, sc, scC, nsc, wsc, r.ldw2, nullptr, nullptr, 0, 1, nullptr, scy, NI, H, W, r.cout, 1, 0, zero_page_));
            RUN(conv3x3(c, src, srcC, 1, nullptr, nullptr, 0, r.w2, r.ldw2, r.b2, nullptr, 0, 1, scy, y, NI, H, W, r.cout, 1, 0, zero_page_,
                        y_csbuf, y_cs));
        } else {
        RUN(conv3x3(c, src, srcC, 1, sc, scC, nsc, r.w2, r.ldw2, r.b2, nullptr, 0, 1, r.shortcut ? nullptr : x1, y, NI, H, W,
                    r.cout, 1, 0, zero_page_, y_csbuf, y_cs));
        }
    }
    TRACE("resnet.conv2", y, M, r.cout);
    c.ws->release(mark);
    return 0;
}

int UNet::run_transformer(FwdCtx& c, const TransformerW& t, half_t* x, const half_t* ctx, int H, int W, bool shared_prefix,
                          GnColStat* x_cs, float* x_csbuf) {
    const int C = t.C, G = cfg_.norm_groups, heads = cfg_.heads, dh = C / heads;
    const int NI = c.B * c.F, D = H * W;
    const int T = NI * D;
    const int X = cfg_.cross_attention_dim;
    const float scale = 1.0f / sqrtf((float)dh);
    const size_t mark = c.ws->mark();
    WS(tx, half_t, (size_t)T * C);
    WS(ln, half_t, (size_t)T * C);
    WS(att, half_t, (size_t)T * C);
    WS(wide, half_t, (size_t)T * 4 * C);          // qkv [T,3C] / GEGLU output [T,4C] / q2 [T,C]
    WS(kv2, half_t, (size_t)c.B * c.ctx_len * 2 * C);
    const size_t ti = &t - transformers_.data();
    // text K/V computed once per context by cache_context(): valid for this very ctx tensor and shape only
    const bool kv_cached = !c.dry && kv_ctx_ != nullptr && kv_ctx_ == ctx && kv_B_ == c.B && kv_len_ == c.ctx_len && ti < kv2_cache_.size();

    // VSR: ResnetBlock3DCNN (3,1,1) on the block input, before the residual is taken (vsr/models/attention.py:395-400)
    if (t.tres.present) {
        // in place; conv2's epilogue leaves the new statistics where the old ones were (the per-frame GroupNorm below takes them only
        // if its frames are whole blocks: the temporal tiles of the halo-patch kernel span a video, GnColStat::span)
        GnColStat in_cs = x_cs ? *x_cs : GnColStat();
        RUN(run_temporal_res(c, t.tres, x, x, C, D, nullptr, 0, &in_cs, x_csbuf, x_cs));
    }
    // shared_prefix (set_cfg_shared_input; base block only): both halves of the batch are identical up to the text
    // cross-attention, so GroupNorm, proj_in, the qkv projection and the self-attention run on the first half (NIp frames,
    // Tp rows) and `tx` / `att` are copied to the second
    const int NIp = shared_prefix ? NI / 2 : NI, Tp = shared_prefix ? T / 2 : T;
    LAVIE_CHECK(!shared_prefix || (!t.tres.present && !t.attn1_cross && c.B % 2 == 0), "transformer: shared prefix on an unsupported block");
    // Round 4: GroupNorm -> proj_in -> norm1 -> q|k|v as ONE row-resident kernel (rowfuse_pin.hip; bit 8 of the mask): the norm's
    // statistics become per-(frame, channel) scale / shift pairs and nothing between x and (tx, qkv) touches memory
    const bool fused_pq = t.pq_img != nullptr && !t.tres.present && !t.attn1_cross && (fused_mask() & 256) && D % 16 == 0;
    WS(gn_ab, float, (size_t)NI * C * 2);        // (planned whether or not the switch is on: the plan must not depend on it)
    if (fused_pq) {
        LAUNCH(launch_group_norm(x, C, nullptr, 0, NIp, D, G, t.gn.g, t.gn.b, 1e-6f, false, c.gn_ws, nullptr, c.s, x_cs, nullptr, gn_ab));
        LAUNCH(launch_proj_qkv(x, gn_ab, D, t.pq_img, t.pin.b, t.ln1.g, t.ln1.b, 1e-5f, tx, wide, Tp, C, c.s));
    } else {
    // per-frame GroupNorm (eps 1e-6) + 1x1 proj_in (attention.py:369-373)
    LAUNCH(launch_group_norm(x, C, nullptr, 0, NIp, D, G, t.gn.g, t.gn.b, 1e-6f, false, c.gn_ws, ln, c.s, x_cs));
    }
    // LayerNorm folding: the GEMM that produces the residual stream `tx` also emits per-row (sum, sum^2) partials of
    // its fp16 output, and the projection that consumes LN(tx) runs on raw `tx` with gamma folded into its weights,
    // finishing rstd * (acc - mean * s) + b' in its epilogue — no LayerNorm kernel, no normalised copy in HBM.
    const bool fold = ln_fold_;
    // base block order: temporal -> feed-forward (attention.py:548-560); interpolation block: feed-forward -> temporal
    // (interpolation/models/attention.py:592-604)
    const bool ff_first = cfg_.ff_before_temporal != 0;
    // row-resident fused sub-blocks (rowfuse.hip), base block order only: they take their LayerNorm statistics from the rows
    // they hold, so their producers emit none
    // (round 4) the feed-forward and text cross-attention kernels do not depend on the block order: in the interpolation order the
    // feed-forward kernel ALSO writes the (mean, rstd) rows its consumer, the LayerNorm-folded temporal qkv projection, needs
    const bool fused_ff = t.ff_img != nullptr && (fused_mask() & 1) && (!ff_first || fold);
    const bool fused_t = t.tb_img != nullptr && !ff_first && c.F == 16 && (fused_mask() & 2);
    // attn1.to_out -> + residual -> norm2 -> attn2 -> to_out -> + residual in one kernel: needs this context's K / V image, and
    // (it emits no row statistics) behind it a kernel that takes its LayerNorm statistics from the rows it holds: the fused temporal
    // kernel in the base order, the fused feed-forward kernel in the interpolation order
    const bool fused_x = t.xb_tmpl != nullptr && kv_cached && xb_bound_ && xb_img_[ti] != nullptr && (ff_first ? fused_ff : fused_t) &&
                         !t.attn1_cross && (fused_mask() & 4) && cross_block_supported(C, heads, c.ctx_len, c.F * D);
    LnFold lf{nullptr, nullptr};
    RowStat rsd{nullptr, nullptr, C / igemm_rowstat_cols(T, C, C / IGEMM_BK), &lf};
    const RowStat* rowstat = nullptr;
    if (fold) {
        WS(rsp, float, (size_t)T * (C / 32) * 2);
        WS(rsm, float, (size_t)T * 2);
        rsd.partials = rsp;
        rsd.mean_rstd = rsm;
        lf.stats = rsm;
        rowstat = &rsd;
    }
    RowStat rsd_pin = rsd;                 // the producer's slot width follows the kernel the planner picks for ITS row count
    rsd_pin.slots = C / igemm_rowstat_cols(Tp, C, C / IGEMM_BK);
    if (!fused_pq) RUN(linear(c, ln, C, t.pin.w, t.pin.b, C, C, nullptr, tx, C, Tp, EPI_LINEAR, nullptr, rowstat ? &rsd_pin : nullptr));
    if (!shared_prefix) TRACE("block.proj_in", tx, T, C);

    if (t.attn1_cross) {
        // VSR only_cross_attention levels: attn1 attends to the text context (vsr/models/attention.py:558-561)
        if (fold) {
            lf.s = t.s_q1;
            RUN(linear(c, tx, C, t.f_q1, t.b_q1, C, C, nullptr, wide, C, T, EPI_LINEAR, &lf));
        } else {
            LAUNCH(launch_layernorm(tx, t.ln1.g, t.ln1.b, ln, T, C, 1e-5f, c.s));
            RUN(linear(c, ln, C, t.wq1, nullptr, C, C, nullptr, wide, C, T));
        }
        const half_t* kvc1 = kv2;
        if (kv_cached) kvc1 = kv1_cache_[ti];
        else RUN(linear(c, ctx, X, t.wkv1, nullptr, 2 * C, X, nullptr, kv2, 2 * C, c.B * c.ctx_len));
        if (!c.dry) {
            AttnParams a;
            a.q = wide; a.ldq = C; a.k = kvc1; a.ldk = 2 * C; a.v = kvc1 + C; a.ldv = 2 * C;
            a.o = att; a.ldo = C; a.NBq = NI; a.Lq = D; a.Lk = c.ctx_len; a.heads = heads; a.dh = dh; a.kv_batch_div = c.F; a.scale = scale;
            RUN(launch_attention(a, c.s));
        }
    } else {
        // spatial self-attention (attention.py:513-522)
        if (fused_pq) {
            // q | k | v already in `wide`
        } else if (fold) {
            lf.s = t.s_qkv1;
            RUN(linear(c, tx, C, t.f_qkv1, t.b_qkv1, 3 * C, C, nullptr, wide, 3 * C, Tp, EPI_LINEAR, &lf));
        } else {
            LAUNCH(launch_layernorm(tx, t.ln1.g, t.ln1.b, ln, Tp, C, 1e-5f, c.s));
            RUN(linear(c, ln, C, t.wqkv1, nullptr, 3 * C, C, nullptr, wide, 3 * C, Tp));
        }
        if (!c.dry) {
            AttnParams a;
            a.q = wide; a.ldq = 3 * C; a.k = wide + C; a.ldk = 3 * C; a.v = wide + 2 * C; a.ldv = 3 * C;
            a.o = att; a.ldo = C; a.NBq = NIp; a.Lq = D; a.Lk = D; a.heads = heads; a.dh = dh; a.kv_batch_div = 1; a.scale = scale;
            if (cfg_.sparse_causal_attn1) {      // keys/values = first frame || previous frame (interpolation attention.py:630-639)
                a.Lk = 2 * D;
                a.sc_frames = c.F;
            }
            RUN(launch_attention(a, c.s));
        }
    }
    if (!shared_prefix) TRACE("block.attn1 q(kv)", wide, T, t.attn1_cross ? C : 3 * C);
    if (!shared_prefix) TRACE("block.attn1", att, T, C);
    if (shared_prefix && !c.dry) {       // the second half of the batch: the same residual stream and attention output so far
        LAVIE_HIP(hipMemcpyAsync(tx + (size_t)Tp * C, tx, (size_t)Tp * C * sizeof(half_t), hipMemcpyDeviceToDevice, c.s));
        LAVIE_HIP(hipMemcpyAsync(att + (size_t)Tp * C, att, (size_t)Tp * C * sizeof(half_t), hipMemcpyDeviceToDevice, c.s));
    }
    if (fused_x) {
        if (!c.dry)
            RUN(launch_cross_block(att, tx, tx, T, c.F * D, C, heads, xb_img_[ti], t.o1.b, t.ln2.g, t.ln2.b, t.o2.b, c.ctx_len, scale, 1e-5f, c.s));
    } else {
    RUN(linear(c, att, C, t.o1.w, t.o1.b, C, C, tx, tx, C, T, EPI_LINEAR, nullptr, rowstat));
    TRACE("block.attn1.to_out", tx, T, C);

    // text cross-attention (attention.py:524-534); K/V once per video instead of once per frame (364)
    if (fold) {
        lf.s = t.s_q2;
        RUN(linear(c, tx, C, t.f_q2, t.b_q2, C, C, nullptr, wide, C, T, EPI_LINEAR, &lf));
    } else {
        LAUNCH(launch_layernorm(tx, t.ln2.g, t.ln2.b, ln, T, C, 1e-5f, c.s));
        RUN(linear(c, ln, C, t.wq2, nullptr, C, C, nullptr, wide, C, T));
    }
    const half_t* kvc2 = kv2;
    if (kv_cached) kvc2 = kv2_cache_[ti];
    else RUN(linear(c, ctx, X, t.wkv2, nullptr, 2 * C, X, nullptr, kv2, 2 * C, c.B * c.ctx_len));
    if (!c.dry) {
        AttnParams a;
        a.q = wide; a.ldq = C; a.k = kvc2; a.ldk = 2 * C; a.v = kvc2 + C; a.ldv = 2 * C;
        a.o = att; a.ldo = C; a.NBq = NI; a.Lq = D; a.Lk = c.ctx_len; a.heads = heads; a.dh = dh; a.kv_batch_div = c.F; a.scale = scale;
        RUN(launch_attention(a, c.s));
    }
    TRACE("block.attn2", att, T, C);
    RUN(linear(c, att, C, t.o2.w, t.o2.b, C, C, tx, tx, C, T, EPI_LINEAR, nullptr, (ff_first ? fused_ff : fused_t) ? nullptr : rowstat));
    }
    TRACE("block.after text", tx, T, C);

    // base block order: temporal -> feed-forward (attention.py:548-560); interpolation block: feed-forward -> temporal
    // (interpolation/models/attention.py:592-604)
    RowStat rsd_ff2 = rsd;
    rsd_ff2.slots = C / igemm_rowstat_cols(T, C, 4 * C / IGEMM_BK);
    const RowStat* rowstat_ff2 = rowstat ? &rsd_ff2 : nullptr;
    auto temporal = [&]() -> int {
        // temporal self-attention over frames, tokens stay in (b f) d order (attention.py:548-555)
        if (fused_t) {         // norm_temp -> q|k|v -> rotary / bias / softmax / PV -> to_out -> + residual in ONE kernel, in place
            if (!c.dry)
                RUN(launch_temporal_block(tx, tx, c.B, c.F, D, C, heads, t.tb_img, t.lnt.g, t.lnt.b, t.ot.b, cur_tables_->relbias[ti],
                                          cur_tables_->rot_cos, cur_tables_->rot_sin, cfg_.rotary_dim, scale, 1e-5f, c.s));
            return 0;
        }
        if (fold) {
            lf.s = t.s_qkvt;
            RUN(linear(c, tx, C, t.f_qkvt, t.b_qkvt, 3 * C, C, nullptr, wide, 3 * C, T, EPI_LINEAR, &lf));
            TRACE("temporal.row statistics", lf.stats, T, 2);
        } else {
            LAUNCH(launch_layernorm(tx, t.lnt.g, t.lnt.b, ln, T, C, 1e-5f, c.s));
            RUN(linear(c, ln, C, t.wqkvt, nullptr, 3 * C, C, nullptr, wide, 3 * C, T));
        }
        TRACE("temporal.qkv before", wide, T, 3 * C);
        if (!c.dry) {
            TemporalParams tp;
            tp.qkv = wide; tp.ld = 3 * C; tp.o = att; tp.ldo = C; tp.B = c.B; tp.F = c.F; tp.D = D; tp.heads = heads; tp.dh = dh;
            tp.bias = cur_tables_->relbias[ti]; tp.rot_cos = cur_tables_->rot_cos; tp.rot_sin = cur_tables_->rot_sin; tp.rot_dim = cfg_.temporal_plain ? 0 : cfg_.rotary_dim; tp.scale = scale;
            RUN(launch_temporal_attention(tp, c.s));
        }
        TRACE("temporal.qkv", wide, T, 3 * C);
        TRACE("temporal.attention", att, T, C);
        // its output feeds norm3 only in the base order; in the interpolation order proj_out follows (no LayerNorm).
        // The fused feed-forward kernel takes its LayerNorm statistics from the rows it holds: no partials needed.
        RUN(linear(c, att, C, t.ot.w, t.ot.b, C, C, tx, tx, C, T, EPI_LINEAR, nullptr, (ff_first || fused_ff) ? nullptr : rowstat));
        return 0;
    };
    auto feed_forward = [&]() -> int {
        // GEGLU feed-forward (attention.py:558)
        if (fused_ff) {        // norm3 -> ff1 -> GEGLU -> ff2 -> + residual in ONE kernel, in place on the residual stream
            // interpolation order: norm_temp consumes this output; the kernel writes its (mean, rstd) rows where ff2's epilogue +
            // rowstat_finalize would have put them
            LAUNCH(launch_geglu_mlp(tx, tx, T, C, t.ff_img, t.ff_b1img, t.ln3.g, t.ln3.b, t.ff2.b, 1e-5f, c.s, ff_first ? lf.stats : nullptr));
            if (ff_first) { lf.partials = nullptr; lf.final_ok = true; }       // finished rows, written by the kernel itself
            return 0;
        }
        if (fold && !fused_t) {       // (the fused temporal kernel emits no row statistics: explicit LayerNorm behind it)
            lf.s = t.s_ff1;
            RUN(linear(c, tx, C, t.f_ff1, t.b_ff1, 8 * C, C, nullptr, wide, 4 * C, T, EPI_GEGLU, &lf));
        } else {
            LAUNCH(launch_layernorm(tx, t.ln3.g, t.ln3.b, ln, T, C, 1e-5f, c.s));
            RUN(linear(c, ln, C, t.ff1.w, t.ff1.b, 8 * C, C, nullptr, wide, 4 * C, T, EPI_GEGLU));
        }
        // interpolation order: norm_temp consumes this GEMM's output, so it emits the row statistics (K = 4C: the
        // planner may pick another kernel than for the K = C producers, hence its own slot count)
        RUN(linear(c, wide, 4 * C, t.ff2.w, t.ff2.b, C, 4 * C, tx, tx, C, T, EPI_LINEAR, nullptr, ff_first ? rowstat_ff2 : nullptr));
        return 0;
    };
    if (ff_first) {
        RUN(feed_forward());
        TRACE("block.feed-forward", tx, T, C);
        RUN(temporal());
        TRACE("block.temporal", tx, T, C);
    } else {
        RUN(temporal());
        TRACE("block.temporal", tx, T, C);
        RUN(feed_forward());
        TRACE("block.feed-forward", tx, T, C);
    }

    // 1x1 proj_out + residual, in place on the block input (attention.py:394-401)
    // (its epilogue leaves the GroupNorm statistics of the block's output for the next resnet / skip consumer: x_cs is rewritten)
    RUN(linear(c, tx, C, t.pout.w, t.pout.b, C, C, x, x, C, T, EPI_LINEAR, nullptr, nullptr, 0, x_csbuf, x_cs));
    TRACE("block.proj_out", x, T, C);
    c.ws->release(mark);
    return 0;
}

int UNet::run_conv(FwdCtx& c, const half_t* x, int C, const SamplerW& w, half_t* y, int Hi, int Wi, int stride, int ups, float* cs_buf,
                   GnColStat* cs_out) {
    if (cs_out) *cs_out = GnColStat();
    const half_t* src[1] = {x};
    const int srcC[1] = {C};
    if (ups && stride == 1 && w.wpar && (fused_mask() & 16)) {
        // Upsample3D (resnet.py:44-79): the 3x3 conv of the nearest-x2 image as four 2x2 convs on the source image, one per
        // output parity, on the halo-patch kernel (igemm_patch.hip MODE 3): 4 C instead of 9 C multiply-adds per output element
        IgemmParams p;
        if (igemm_setup_parity_upsample(&p, x, C, w.wpar, w.b, y, c.B * c.F, Hi, Wi, zero_page_)) {
            colstat_plan(p, true, cs_buf, cs_out);
            const size_t mark = c.ws->mark();
            if (p.splits > 1) {
                p.slab = (float*)c.ws->alloc((size_t)p.splits * p.M * p.N * sizeof(float));
                if (!c.dry) LAVIE_CHECK(p.slab != nullptr, "workspace exhausted (split-K slab)");
            }
            const int rc = c.dry ? 0 : launch_igemm(p, true, EPI_LINEAR, c.s);
            c.ws->release(mark);
            return rc;
        }
    }
    return conv3x3(c, src, srcC, 1, nullptr, nullptr, 0, w.w, 9 * C, w.b, nullptr, 0, 1, nullptr, y, c.B * c.F, Hi, Wi, C, stride,
                   ups, zero_page_, cs_buf, cs_out);
}

int UNet::run(FwdCtx& c, const half_t* sample, const float* timesteps, const half_t* ctx, half_t* out) {
    const lavie_unet_config& cfg = cfg_;
    const int L = cfg.num_levels;
    const int NI = c.B * c.F;
    const int C0 = cfg.block_out_channels[0];
    const int temb = C0 * 4;
    const int G = cfg.norm_groups;

    // GroupNorm scratch (partials + mean/rstd), shared by every GroupNorm of the pass: stream order keeps it safe
    {
        WS(gnws, float, gn_workspace_floats(NI, G));
        c.gn_ws = gnws;
    }

    // time embedding (unet.py:428-434) and all 22 resnet projections in one GEMV (resnet.py:186)
    WS(tsin, float, (size_t)c.B * C0);
    WS(e1, float, (size_t)c.B * temb);
    WS(emb, float, (size_t)c.B * temb);
    WS(tproj, float, (size_t)c.B * tproj_.N);
    LAUNCH(launch_timestep_sinusoid(timesteps, tsin, c.B, C0, c.s));
    LAUNCH(launch_gemv(tsin, time1_.w, time1_.b, e1, c.B, temb, C0, 0, 1, c.s));
    // `emb` only ever reaches the resnets through SiLU (resnet.py:186): apply it once here instead of once per
    // output feature inside the stacked projection
    if (cfg.num_class_embeds > 0) {
        // emb = time_embedding + class_embedding[noise level] (vsr/models/unet.py:494-505), then the consumers' SiLU
        LAUNCH(launch_gemv(e1, time2_.w, time2_.b, emb, c.B, temb, temb, 0, 0, c.s));
        if (!c.dry) {
            LAVIE_CHECK(c.labels != nullptr, "forward: this model has a class embedding: call lavie_unet_forward_labels");
            for (int b = 0; b < c.B; ++b)
                LAVIE_CHECK(c.labels[b] >= 0 && c.labels[b] < cfg.num_class_embeds, "forward: class label %d out of range", c.labels[b]);
            RUN(launch_add_class_emb_silu(emb, class_emb_, c.labels, c.B, temb, c.s));
        }
    } else {
        LAUNCH(launch_gemv(e1, time2_.w, time2_.b, emb, c.B, temb, temb, 0, 1, c.s));
    }
    LAUNCH(launch_gemv(emb, tproj_.w, tproj_.b, tproj, c.B, tproj_.N, temb, 0, 0, c.s));

    std::vector<int> Hs(L), Ws(L);
    for (int l = 0; l < L; ++l) {
        Hs[l] = l == 0 ? prep_H_ : (Hs[l - 1] - 1) / 2 + 1;
        Ws[l] = l == 0 ? prep_W_ : (Ws[l - 1] - 1) / 2 + 1;
    }
    // prep_H_/prep_W_ hold the CURRENT call's size while run() executes (set by forward()/prepare())
    auto rows = [&](int l) { return (size_t)NI * Hs[l] * Ws[l]; };

    struct Skip { half_t* p; int C; GnColStat cs; };     // a skip tensor travels with the statistics its producer left
    GnColStat x_cs;                                      // ... and so does the running activation x
    std::vector<Skip> skips;
    size_t ri = 0, ti = 0, mi = 0;
    const bool tmod = cfg.vsr_temporal_modules != 0;

    // Classifier-free guidance runs the UNet on the latents twice with different text: up to the first text cross-attention
    // (conv_in, the first resnet, and GroupNorm / proj_in / self-attention of the first transformer block) the two halves of the
    // batch are the same computation.  With set_cfg_shared_input the caller vouches for sample[b] == sample[b + B/2]; those
    // layers then run on the first half and their outputs are copied (base UNet only: the first down block must have attention).
    const bool shared_ok = c.B % 2 == 0 && cfg.attn_levels[0] && cfg.layers_per_block >= 1 && !tmod && !cfg.vsr_blocks &&
                           cfg.num_class_embeds == 0 && !cfg.sparse_causal_attn1 && !transformers_[0].attn1_cross && !transformers_[0].tres.present;
    // a caller that left the switch on for a batch or model it cannot apply to gets an error, not a silently different amount of work
    LAVIE_CHECK(!cfg_shared_input_ || shared_ok || c.dry, "forward: set_cfg_shared_input(1) needs an even batch (B=%d) and the base UNet "
                "(attention in the first down block, no VSR / interpolation variants)", c.B);
    const bool shared = cfg_shared_input_ && shared_ok;
    if (shared && !c.dry && debug_check_shared()) {
        // LAVIE_DEBUG_CHECK_SHARED=1: verify the caller's promise (sample[b] == sample[b + B/2], equal timesteps) before trusting it
        const size_t half_bytes = (size_t)(c.B / 2) * cfg.in_channels * c.F * prep_H_ * prep_W_ * sizeof(half_t);
        std::vector<char> h0(half_bytes), h1(half_bytes);
        std::vector<float> ts(c.B);
        LAVIE_HIP(hipStreamSynchronize(c.s));
        LAVIE_HIP(hipMemcpy(h0.data(), sample, half_bytes, hipMemcpyDeviceToHost));
        LAVIE_HIP(hipMemcpy(h1.data(), (const char*)sample + half_bytes, half_bytes, hipMemcpyDeviceToHost));
        LAVIE_HIP(hipMemcpy(ts.data(), timesteps, c.B * sizeof(float), hipMemcpyDeviceToHost));
        LAVIE_CHECK(memcmp(h0.data(), h1.data(), half_bytes) == 0, "forward: set_cfg_shared_input(1) but the two halves of the sample differ");
        for (int b = 0; b < c.B / 2; ++b)
            LAVIE_CHECK(ts[b] == ts[b + c.B / 2], "forward: set_cfg_shared_input(1) but timesteps %g / %g differ", ts[b], ts[b + c.B / 2]);
    }
    FwdCtx ch = c;                       // the same stream / workspace / scratch, half the batch
    if (shared) ch.B = c.B / 2;
    auto dup_half = [&](half_t* p, size_t rows_full, int ch_count) -> int {
        if (shared && !c.dry)
            LAVIE_HIP(hipMemcpyAsync(p + rows_full / 2 * ch_count, p, rows_full / 2 * ch_count * sizeof(half_t), hipMemcpyDeviceToDevice, c.s));
        return 0;
    };

    WS(x0, half_t, rows(0) * C0);
    LAUNCH(launch_conv_in(sample, conv_in_w_, conv_in_b_, x0, ch.B, cfg.in_channels, c.F, Hs[0], Ws[0], C0, c.s));
    RUN(dup_half(x0, rows(0), C0));
    half_t* x = x0;
    int C = C0;
    skips.push_back({x, C, x_cs});                      // conv_in leaves no statistics: its two consumers run the statistics pass

    for (int l = 0; l < L; ++l) {
        for (int j = 0; j < cfg.layers_per_block; ++j) {
            const ResnetW& r = resnets_[ri++];
            WS(y, half_t, rows(l) * r.cout);
            WS(ycs, float, colstat_floats(rows(l), r.cout));
            GnColStat y_cs;
            const bool first = shared && l == 0 && j == 0;
            RUN(run_resnet(first ? ch : c, r, x, C, nullptr, 0, tproj + r.temb_off, tproj_.N, y, Hs[l], Ws[l], &x_cs, nullptr, ycs, &y_cs));
            if (first) RUN(dup_half(y, rows(l), r.cout));     // (y_cs describes the first half: all its one consumer, the half-batch GroupNorm below, reads)
            x = y; C = r.cout; x_cs = y_cs;
            if (cfg.attn_levels[l]) RUN(run_transformer(c, transformers_[ti++], x, ctx, Hs[l], Ws[l], first, &x_cs, ycs));
            else if (first) x_cs = GnColStat();
            skips.push_back({x, C, x_cs});
        }
        if (l + 1 < L) {
            WS(y, half_t, rows(l + 1) * C);
            WS(ycs, float, colstat_floats(rows(l + 1), C));
            GnColStat y_cs;
            RUN(run_conv(c, x, C, downs_[l], y, Hs[l], Ws[l], 2, 0, ycs, &y_cs));
            x = y; x_cs = y_cs;
            skips.push_back({x, C, x_cs});
        }
        if (tmod) {       // after the downsampler, into a NEW buffer: x itself stays alive as a skip (vsr/models/unet.py:523-533)
            const int ll = l + 1 < L ? l + 1 : l;
            WS(yt, half_t, rows(ll) * C);
            WS(ytcs, float, colstat_floats(rows(ll), C));
            GnColStat yt_cs;
            RUN(run_temporal_module(c, tmods_[mi++], x, yt, tproj, tproj_.N, Hs[ll], Ws[ll], &x_cs, ytcs, &yt_cs));
            x = yt; x_cs = yt_cs;
        }
    }
    {
        const int l = L - 1;
        const ResnetW& r0 = resnets_[ri++];
        WS(y0, half_t, rows(l) * r0.cout);
        WS(y0cs, float, colstat_floats(rows(l), r0.cout));
        GnColStat y_cs;
        RUN(run_resnet(c, r0, x, C, nullptr, 0, tproj + r0.temb_off, tproj_.N, y0, Hs[l], Ws[l], &x_cs, nullptr, y0cs, &y_cs));
        x = y0; C = r0.cout; x_cs = y_cs;
        RUN(run_transformer(c, transformers_[ti++], x, ctx, Hs[l], Ws[l], false, &x_cs, y0cs));
        const ResnetW& r1 = resnets_[ri++];
        WS(y1, half_t, rows(l) * r1.cout);
        WS(y1cs, float, colstat_floats(rows(l), r1.cout));
        RUN(run_resnet(c, r1, x, C, nullptr, 0, tproj + r1.temb_off, tproj_.N, y1, Hs[l], Ws[l], &x_cs, nullptr, y1cs, &y_cs));
        x = y1; C = r1.cout; x_cs = y_cs;
        if (tmod) {
            WS(yt, half_t, rows(l) * C);
            WS(ytcs, float, colstat_floats(rows(l), C));
            GnColStat yt_cs;
            RUN(run_temporal_module(c, tmods_[mi++], x, yt, tproj, tproj_.N, Hs[l], Ws[l], &x_cs, ytcs, &yt_cs));
            x = yt; x_cs = yt_cs;
        }
    }
    for (int i = 0; i < L; ++i) {
        const int l = L - 1 - i;
        for (int j = 0; j < cfg.layers_per_block + 1; ++j) {
            const Skip sk = skips.back();
            skips.pop_back();
            const ResnetW& r = resnets_[ri++];
            WS(y, half_t, rows(l) * r.cout);
            WS(ycs, float, colstat_floats(rows(l), r.cout));
            GnColStat y_cs;
            RUN(run_resnet(c, r, x, C, sk.p, sk.C, tproj + r.temb_off, tproj_.N, y, Hs[l], Ws[l], &x_cs, &sk.cs, ycs, &y_cs));
            x = y; C = r.cout; x_cs = y_cs;
            if (cfg.attn_levels[l]) RUN(run_transformer(c, transformers_[ti++], x, ctx, Hs[l], Ws[l], false, &x_cs, ycs));
        }
        if (i + 1 < L) {
            WS(y, half_t, rows(l - 1) * C);
            WS(ycs, float, colstat_floats(rows(l - 1), C));
            GnColStat y_cs;
            RUN(run_conv(c, x, C, ups_[i], y, Hs[l], Ws[l], 1, 1, ycs, &y_cs));
            x = y; x_cs = y_cs;
        }
        if (tmod) {       // after the upsampler (vsr/models/unet.py:575-590)
            const int ll = i + 1 < L ? l - 1 : l;
            WS(yt, half_t, rows(ll) * C);
            WS(ytcs, float, colstat_floats(rows(ll), C));
            GnColStat yt_cs;
            RUN(run_temporal_module(c, tmods_[mi++], x, yt, tproj, tproj_.N, Hs[ll], Ws[ll], &x_cs, ytcs, &yt_cs));
            x = yt; x_cs = yt_cs;
        }
    }
    // conv_norm_out + SiLU + conv_out (unet.py:504-506), back to the caller's NCFHW layout
    {
        WS(nrm, half_t, rows(0) * C0);
        const int P = c.F * Hs[0] * Ws[0];
        LAUNCH(launch_group_norm(x, C0, nullptr, 0, c.B, P, G, norm_out_.g, norm_out_.b, cfg.norm_eps, true, c.gn_ws, nrm, c.s, &x_cs));
        LAUNCH(launch_conv_out(nrm, conv_out_w_, conv_out_b_, out, c.B, C0, c.F, Hs[0], Ws[0], cfg.out_channels, c.s));
    }
    return 0;
}

static int check_shape(const lavie_unet_config& cfg, int B, int F, int H, int W, int ctx_len) {
    LAVIE_CHECK(B >= 1 && B <= 8 && F >= 1 && F <= 64 && H >= 1 && W >= 1 && ctx_len >= 1,
                "shape: B=%d (1..8) F=%d (1..64) H=%d W=%d ctx_len=%d unsupported", B, F, H, W, ctx_len);
    const int div = 1 << (cfg.num_levels - 1);
    LAVIE_CHECK(H % div == 0 && W % div == 0, "shape: H=%d W=%d must be multiples of %d (unet.py:393-401 upsample-size "
                "forwarding is not implemented)", H, W, div);
    return 0;
}

int UNet::prepare(int B, int F, int H, int W, int ctx_len) {
    LAVIE_CHECK(finalized_, "prepare: call lavie_unet_finalize first");
    RUN(check_shape(cfg_, B, F, H, W, ctx_len));
    // The forward may run under other switches than the ones current now (the guided loop turns the shared CFG prefix on AFTER
    // prepare(); lavie_debug_fused_mask and the text cache change which GEMMs run, and a half-batch launch may plan another
    // split-K slab): the plan is the maximum over every combination of them, so none of those switches can outgrow the workspace.
    size_t peak = 0;
    const bool shared_was = cfg_shared_input_;
    const int mask_was = fused_mask();
    int rc = 0;
    // masks: the current one; without the row-resident kernels (bits 0 - 2, 8: their GEMMs and row-statistics buffers appear); and both
    // again with every workspace-consuming option on (bit 3: the shortcut's own output, 4: the parity form's slabs, 5: statistics buffers)
    const int masks[4] = {mask_was, mask_was & ~0x107, mask_was | 0x38, (mask_was | 0x38) & ~0x107};
    for (int variant = 0; variant < 8 && rc == 0; ++variant) {
        if ((variant & 1) && B % 2 != 0) continue;
        cfg_shared_input_ = (variant & 1) != 0;
        set_fused_mask(masks[variant >> 1]);
        DeviceArena plan;
        plan.init_virtual();
        FwdCtx c{nullptr, &plan, true, B, F, ctx_len, nullptr};
        prep_H_ = H; prep_W_ = W;
        rc = run(c, nullptr, nullptr, nullptr, nullptr);
        if (plan.peak() > peak) peak = plan.peak();
    }
    cfg_shared_input_ = shared_was;
    set_fused_mask(mask_was);
    RUN(rc);
    const size_t need = peak + (1 << 20);
    if (need > ws_.total_bytes()) {
        drop_graph();                               // the captured addresses die with the old workspace
        ++graph_gen_;
        RUN(ws_.init_fixed(need));
    }
    return 0;
}

void UNet::drop_graph() {
    if (graph_exec_) (void)hipGraphExecDestroy(graph_exec_);
    if (graph_) (void)hipGraphDestroy(graph_);
    graph_exec_ = nullptr;
    graph_ = nullptr;
    graph_key_ = GraphKey();
}

int UNet::forward_graph(const half_t* sample, const float* timesteps, const half_t* ctx, half_t* out, int B, int F, int H, int W,
                        int ctx_len, hipStream_t stream) {
    bool profiling = false;
    for (int cls = 0; cls < KC_COUNT; ++cls) profiling = profiling || profile_enabled(cls);
    GraphKey key;
    key.sample = sample; key.t = timesteps; key.ctx = ctx; key.out = out; key.kv_ctx = kv_ctx_;
    key.B = B; key.F = F; key.H = H; key.W = W; key.L = ctx_len; key.gen = graph_gen_; key.debug_epoch = debug_epoch(); key.stream = stream;
    if (profiling) return forward(sample, timesteps, ctx, out, B, F, H, W, ctx_len, stream, nullptr);
    if (graph_exec_ && key == graph_key_) {
        LAVIE_HIP(hipGraphLaunch(graph_exec_, stream));
        return 0;
    }
    if (!(key == graph_seen_)) {                    // first sight of this tuple: eager (one-time setup must not be captured)
        graph_seen_ = key;
        return forward(sample, timesteps, ctx, out, B, F, H, W, ctx_len, stream, nullptr);
    }
    drop_graph();
    // captured on a private stream (the caller's may be the legacy default stream, which cannot be captured); the graph
    // itself is launched on the caller's stream
    if (!cap_stream_) LAVIE_HIP(hipStreamCreateWithFlags(&cap_stream_, hipStreamNonBlocking));
    LAVIE_HIP(hipStreamBeginCapture(cap_stream_, hipStreamCaptureModeThreadLocal));
    const int rc = forward(sample, timesteps, ctx, out, B, F, H, W, ctx_len, cap_stream_, nullptr);
    hipGraph_t g = nullptr;
    const hipError_t end = hipStreamEndCapture(cap_stream_, &g);
    if (rc != 0 || end != hipSuccess || g == nullptr) {
        if (g) (void)hipGraphDestroy(g);
        graph_seen_ = GraphKey();
        if (rc != 0) return rc;
        LAVIE_CHECK(false, "forward_graph: stream capture failed (%s)", hipGetErrorString(end));
    }
    graph_ = g;
    LAVIE_HIP(hipGraphInstantiate(&graph_exec_, graph_, nullptr, nullptr, 0));
    graph_key_ = key;
    LAVIE_HIP(hipGraphLaunch(graph_exec_, stream));
    return 0;
}

int UNet::forward(const half_t* sample, const float* timesteps, const half_t* ctx, half_t* out, int B, int F, int H, int W,
                  int ctx_len, hipStream_t stream, const int* class_labels_host) {
    LAVIE_CHECK(finalized_, "forward: call lavie_unet_finalize first");
    LAVIE_CHECK(sample && timesteps && ctx && out, "forward: null tensor");
    RUN(check_shape(cfg_, B, F, H, W, ctx_len));
    LAVIE_CHECK(ws_.total_bytes() > 0, "forward: call lavie_unet_prepare first");
    RUN(ensure_tables(F, stream));
    ws_.release(0);
    FwdCtx c{stream, &ws_, false, B, F, ctx_len, nullptr};
    c.labels = class_labels_host;
    prep_H_ = H; prep_W_ = W;
    return run(c, sample, timesteps, ctx, out);
}

int UNet::cache_context(const half_t* ctx, int B, int ctx_len, hipStream_t stream) {
    LAVIE_CHECK(finalized_, "cache_context: call lavie_unet_finalize first");
    kv_ctx_ = nullptr;                              // invalid until every buffer is written
    ++graph_gen_;
    if (ctx == nullptr) return 0;
    LAVIE_CHECK(B >= 1 && B <= 8 && ctx_len >= 1, "cache_context: B=%d ctx_len=%d unsupported", B, ctx_len);
    LAVIE_CHECK(ws_.total_bytes() > 0, "cache_context: call lavie_unet_prepare first (split-K slabs come from the workspace)");
    const size_t rows = (size_t)B * ctx_len;
    const int X = cfg_.cross_attention_dim;
    if (rows > kv_cache_rows_ || B > kv_cache_B_ || kv2_cache_.size() != transformers_.size()) {
        // the cache lives in a block of its own: a longer context frees the old block instead of stranding it in the
        // grow-only weights arena.  Earlier forwards that read the old block are ordered before the free by the sync.
        size_t total = 0;
        auto span = [&](size_t i) { return (rows * 2 * transformers_[i].C * sizeof(half_t) + 255) & ~(size_t)255; };
        auto img_span = [&](size_t i) { return transformers_[i].xb_tmpl ? (size_t)B * cross_block_image_bytes(transformers_[i].C) : 0; };
        for (size_t i = 0; i < transformers_.size(); ++i) total += span(i) * (transformers_[i].attn1_cross ? 2 : 1) + img_span(i);
        if (kv_block_) {
            LAVIE_HIP(hipStreamSynchronize(stream));
            LAVIE_HIP(hipFree(kv_block_));
            kv_block_ = nullptr;
            kv_cache_rows_ = 0;
            kv_cache_B_ = 0;
        }
        LAVIE_HIP(hipMalloc(&kv_block_, total));
        kv2_cache_.assign(transformers_.size(), nullptr);
        kv1_cache_.assign(transformers_.size(), nullptr);
        xb_img_.assign(transformers_.size(), nullptr);
        char* cur = (char*)kv_block_;
        for (size_t i = 0; i < transformers_.size(); ++i) {
            kv2_cache_[i] = (half_t*)cur; cur += span(i);
            if (transformers_[i].attn1_cross) { kv1_cache_[i] = (half_t*)cur; cur += span(i); }
            if (img_span(i)) { xb_img_[i] = (half_t*)cur; cur += img_span(i); }
        }
        kv_cache_rows_ = rows;
        kv_cache_B_ = B;
    }
    ws_.release(0);
    FwdCtx c{stream, &ws_, false, B, 1, ctx_len, nullptr};
    for (size_t i = 0; i < transformers_.size(); ++i) {
        const TransformerW& t = transformers_[i];
        RUN(linear(c, ctx, X, t.wkv2, nullptr, 2 * t.C, X, nullptr, kv2_cache_[i], 2 * t.C, (int)rows));
        if (t.attn1_cross) RUN(linear(c, ctx, X, t.wkv1, nullptr, 2 * t.C, X, nullptr, kv1_cache_[i], 2 * t.C, (int)rows));
    }
    // the fused cross-attention kernel streams K / V with its weights: one image per video, written here once per context
    xb_bound_ = false;
    if (cross_block_supported(cfg_.block_out_channels[0], cfg_.heads, ctx_len, 16)) {
        for (size_t i = 0; i < transformers_.size(); ++i)
            if (xb_img_[i]) RUN(bind_cross_block(transformers_[i].xb_tmpl, kv2_cache_[i], B, ctx_len, transformers_[i].C, xb_img_[i], stream));
        xb_bound_ = true;
    }
    kv_ctx_ = ctx;
    kv_B_ = B;
    kv_len_ = ctx_len;
    return 0;
}

int UNet::resnet_forward(const char* prefix, const half_t* x1, int C1, const half_t* x2, int C2, const float* temb, half_t* y,
                         int B, int F, int H, int W, hipStream_t stream) {
    LAVIE_CHECK(finalized_, "resnet_forward: call lavie_unet_finalize first");
    const ResnetW* r = nullptr;
    for (const ResnetW& cand : resnets_) if (cand.prefix == prefix) r = &cand;
    LAVIE_CHECK(r != nullptr, "resnet_forward: no ResnetBlock3D with prefix '%s'", prefix);
    // standalone call: private workspace sized on the fly (test seam, not the hot path)
    DeviceArena local;
    const size_t M = (size_t)B * F * H * W;
    RUN(local.init_fixed((M * (r->cin + 2 * r->cout)) * sizeof(half_t) + (size_t)B * r->cout * 4 + colstat_floats(M, r->cout) * sizeof(float) + (4 << 20)));
    FwdCtx c{stream, &local, false, B, F, 0, nullptr};
    c.gn_ws = (float*)local.alloc(gn_workspace_floats(B * F, cfg_.norm_groups) * sizeof(float));
    float* tproj = (float*)local.alloc((size_t)B * r->cout * sizeof(float));
    RUN(launch_gemv(temb, tproj_.w + (size_t)r->temb_off * tproj_.K, tproj_.b + r->temb_off, tproj, B, r->cout, tproj_.K, 1, 0, stream));
    RUN(run_resnet(c, *r, x1, C1, C2 ? x2 : nullptr, C2, tproj, r->cout, y, H, W));
    LAVIE_HIP(hipStreamSynchronize(stream));     // the private workspace dies with this frame
    return 0;
}

int UNet::transformer_forward(const char* prefix, half_t* x, const half_t* ctx, int B, int F, int H, int W, int ctx_len,
                              hipStream_t stream) {
    LAVIE_CHECK(finalized_, "transformer_forward: call lavie_unet_finalize first");
    const TransformerW* t = nullptr;
    for (const TransformerW& cand : transformers_) if (cand.prefix == prefix) t = &cand;
    LAVIE_CHECK(t != nullptr, "transformer_forward: no Transformer3DModel with prefix '%s'", prefix);
    LAVIE_CHECK(F >= 1 && F <= 64, "transformer_forward: F=%d unsupported", F);
    RUN(ensure_tables(F, stream));
    DeviceArena local;
    const size_t T = (size_t)B * F * H * W;
    RUN(local.init_fixed(T * t->C * 7 * sizeof(half_t) + (size_t)B * ctx_len * 2 * t->C * sizeof(half_t) + (4 << 20)));
    FwdCtx c{stream, &local, false, B, F, ctx_len, nullptr};
    c.gn_ws = (float*)local.alloc(gn_workspace_floats(B * F, cfg_.norm_groups) * sizeof(float));
    RUN(run_transformer(c, *t, x, ctx, H, W));
    LAVIE_HIP(hipStreamSynchronize(stream));
    return 0;
}

}  // namespace lavie
