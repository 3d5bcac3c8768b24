// `make asan` driver: walks the C ABI of the host-only sanitizer build (hip_stub.cpp) the way the Python facade does:
// create -> parameter inventory -> set_param -> finalize (weight packing, arena carving) -> prepare (workspace dry run over every
// switch combination) -> cache_context -> forward (sequencing + every launcher's host side) -> error paths -> destroy, for the
// base, interpolation and VSR variants of the engine at reduced widths.  Exit code 0 and a silent sanitizer = pass.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../../include/lavie_hip.h"

extern "C" long lavie_hostcheck_launches();

#define REQUIRE(cond)                                                                          \
    do {                                                                                       \
        if (!(cond)) {                                                                         \
            fprintf(stderr, "hostcheck: %s failed at line %d: %s\n", #cond, __LINE__, lavie_last_error()); \
            exit(2);                                                                           \
        }                                                                                      \
    } while (0)

static lavie_unet_config base_config() {
    lavie_unet_config c;
    memset(&c, 0, sizeof(c));
    c.struct_size = (int)sizeof(c);
    c.in_channels = 4; c.out_channels = 4; c.num_levels = 2;
    c.block_out_channels[0] = 320; c.block_out_channels[1] = 640;
    c.attn_levels[0] = 1; c.attn_levels[1] = 0;
    c.layers_per_block = 2; c.heads = 8; c.cross_attention_dim = 128; c.norm_groups = 32; c.norm_eps = 1e-5f;
    c.rotary_dim = 32; c.rel_buckets = 32; c.rel_max_distance = 32;
    return c;
}

static void run_model(const lavie_unet_config& cfg, int B, int F, int H, int W, bool labels) {
    lavie_unet_t h = nullptr;
    REQUIRE(lavie_unet_create(&cfg, &h) == 0);
    const int n = lavie_unet_num_params(h);
    REQUIRE(n > 0);
    std::vector<void*> bufs;
    for (int i = 0; i < n; ++i) {
        const char* name = nullptr;
        long long numel = 0;
        REQUIRE(lavie_unet_param_info(h, i, &name, &numel) == 0 && name && numel > 0);
        void* p = calloc((size_t)numel, 2);          // exact size: an over-long packing copy trips ASan
        bufs.push_back(p);
        REQUIRE(lavie_unet_set_param(h, name, p, numel) == 0);
        if (i == 0) {                                // argument checks of set_param
            REQUIRE(lavie_unet_set_param(h, name, p, numel + 1) != 0);
            REQUIRE(lavie_unet_set_param(h, "no.such.key", p, numel) != 0);
            REQUIRE(lavie_unet_set_param(h, name, nullptr, numel) != 0);
        }
    }
    REQUIRE(lavie_unet_param_info(h, n, nullptr, nullptr) != 0);
    REQUIRE(lavie_unet_prepare(h, B, F, H, W, 77) != 0);              // before finalize: refused
    REQUIRE(lavie_unet_finalize(h, nullptr) == 0);
    REQUIRE(lavie_unet_finalize(h, nullptr) != 0);                    // twice: refused
    REQUIRE(lavie_unet_prepare(h, B, F, H + 1, W, 77) != 0);          // not a multiple of 2^(levels-1)
    REQUIRE(lavie_unet_prepare(h, B, F, H, W, 77) == 0);
    REQUIRE(lavie_unet_workspace_bytes(h) > 0 && lavie_unet_weight_bytes(h) > 0);
    const size_t in_elems = (size_t)B * cfg.in_channels * F * H * W, out_elems = (size_t)B * cfg.out_channels * F * H * W;
    void* x = calloc(in_elems, 2);
    void* y = calloc(out_elems, 2);
    void* ctx = calloc((size_t)B * 77 * cfg.cross_attention_dim, 2);
    float* t = (float*)calloc(B, sizeof(float));
    std::vector<int> lab(B, 3);
    const long before = lavie_hostcheck_launches();
    if (labels) {
        REQUIRE(lavie_unet_forward(h, x, t, ctx, y, B, F, H, W, 77, nullptr) != 0);        // class-embedded model needs labels
        REQUIRE(lavie_unet_forward_labels(h, x, t, ctx, lab.data(), y, B, F, H, W, 77, nullptr) == 0);
    } else {
        REQUIRE(lavie_unet_forward(h, x, t, ctx, y, B, F, H, W, 77, nullptr) == 0);
        REQUIRE(lavie_unet_cache_context(h, ctx, B, 77, nullptr) == 0);
        REQUIRE(lavie_unet_forward(h, x, t, ctx, y, B, F, H, W, 77, nullptr) == 0);
        if (B % 2 == 0 && !cfg.sparse_causal_attn1 && !cfg.vsr_blocks) {
            REQUIRE(lavie_unet_set_cfg_shared_input(h, 1) == 0);
            REQUIRE(lavie_unet_forward(h, x, t, ctx, y, B, F, H, W, 77, nullptr) == 0);
            REQUIRE(lavie_unet_set_cfg_shared_input(h, 0) == 0);
        } else {
            REQUIRE(lavie_unet_set_cfg_shared_input(h, 1) == 0);
            REQUIRE(lavie_unet_forward(h, x, t, ctx, y, B, F, H, W, 77, nullptr) != 0);    // the switch cannot apply: an error, not a guess
            REQUIRE(lavie_unet_set_cfg_shared_input(h, 0) == 0);
        }
        REQUIRE(lavie_unet_cache_context(h, nullptr, 0, 0, nullptr) == 0);
    }
    REQUIRE(lavie_hostcheck_launches() > before + 50);
    REQUIRE(lavie_unet_forward(h, x, t, ctx, y, B, F, H * 2, W * 2, 77, nullptr) != 0);    // larger than prepared: workspace refuses
    REQUIRE(lavie_unet_forward(h, nullptr, t, ctx, y, B, F, H, W, 77, nullptr) != 0);
    REQUIRE(lavie_unet_forward(h, x, t, ctx, y, 9, F, H, W, 77, nullptr) != 0);
    REQUIRE(lavie_unet_destroy(h) == 0);
    free(x); free(y); free(ctx); free(t);
    for (void* p : bufs) free(p);
}

int main() {
    REQUIRE(lavie_abi_version() == LAVIE_ABI_VERSION);
    {   // argument checks of create
        lavie_unet_config c = base_config();
        lavie_unet_t h = nullptr;
        REQUIRE(lavie_unet_create(nullptr, &h) != 0);
        c.struct_size -= 4;
        REQUIRE(lavie_unet_create(&c, &h) != 0);
        c = base_config();
        c.num_levels = 9;
        REQUIRE(lavie_unet_create(&c, &h) != 0);
        c = base_config();
        c.block_out_channels[0] = 100;
        REQUIRE(lavie_unet_create(&c, &h) != 0);
    }
    run_model(base_config(), 2, 16, 16, 16, false);                  // base block order, fused level-0 kernels in reach (C = 320, F = 16)
    run_model(base_config(), 1, 4, 8, 8, false);                     // odd batch, ragged tiles
    {
        lavie_unet_config c = base_config();                         // interpolation variant
        c.in_channels = 8; c.sparse_causal_attn1 = 1; c.temporal_plain = 1; c.ff_before_temporal = 1;
        run_model(c, 2, 7, 8, 8, false);
    }
    {
        lavie_unet_config c = base_config();                         // VSR variant
        c.in_channels = 8; c.block_out_channels[0] = 256; c.block_out_channels[1] = 512; c.attn_levels[1] = 1;
        c.vsr_blocks = 1; c.only_cross_attention[0] = 1; c.vsr_temporal_modules = 1; c.num_class_embeds = 10;
        run_model(c, 2, 4, 8, 8, true);
    }
    // operator-level argument checks
    REQUIRE(lavie_linear_f16(nullptr, 0, nullptr, nullptr, nullptr, 0, 0, nullptr, 0, nullptr, 0, 16, 64, 63, 0, nullptr) != 0);
    REQUIRE(lavie_geglu_mlp_image_bytes(123) == 0);
    {   // round 4 operators: widths that are not built, null tensors, a frame height that is not a whole number of 16-row tiles
        REQUIRE(lavie_proj_qkv_image_bytes(256) == 0 && lavie_proj_qkv_image_bytes(320) > 0);
        std::vector<unsigned short> x(64 * 320), wq(3 * 320 * 320), wp(320 * 320), tx(64 * 320), qkv(64 * 960);
        std::vector<unsigned short> img((size_t)lavie_proj_qkv_image_bytes(320) / 2);
        std::vector<float> ab(2 * 320 * 2), v(320, 1.f), ws(1 << 16);
        REQUIRE(lavie_pack_proj_qkv_f16(wp.data(), wq.data(), 256, img.data(), nullptr) != 0);
        REQUIRE(lavie_pack_proj_qkv_f16(wp.data(), wq.data(), 320, img.data(), nullptr) == 0);
        REQUIRE(lavie_group_norm_affine_f16(x.data(), 320, 2, 32, 32, v.data(), v.data(), 1e-6f, ws.data(), ab.data(), nullptr) == 0);
        REQUIRE(lavie_group_norm_affine_f16(x.data(), 320, 2, 32, 32, v.data(), v.data(), 1e-6f, ws.data(), nullptr, nullptr) != 0);
        REQUIRE(lavie_proj_qkv_f16(x.data(), ab.data(), 32, img.data(), v.data(), v.data(), v.data(), 1e-5f, tx.data(), qkv.data(), 64, 320, nullptr) == 0);
        REQUIRE(lavie_proj_qkv_f16(x.data(), ab.data(), 24, img.data(), v.data(), v.data(), v.data(), 1e-5f, tx.data(), qkv.data(), 48, 320, nullptr) != 0);
        REQUIRE(lavie_proj_qkv_f16(x.data(), nullptr, 32, img.data(), v.data(), v.data(), v.data(), 1e-5f, tx.data(), qkv.data(), 64, 320, nullptr) != 0);
        std::vector<float> st(64 * 2, 1.f), s3(960, 0.f);
        REQUIRE(lavie_linear_lnfold_f16(x.data(), wq.data(), v.data(), s3.data(), st.data(), qkv.data(), 64, 960, 320, nullptr) == 0);
        REQUIRE(lavie_linear_lnfold_f16(x.data(), wq.data(), v.data(), nullptr, st.data(), qkv.data(), 64, 960, 320, nullptr) != 0);
        REQUIRE(lavie_linear_lnfold_f16(x.data(), wq.data(), v.data(), s3.data(), st.data(), qkv.data(), 64, 960, 300, nullptr) != 0);
    }
    REQUIRE(lavie_upsample_conv3x3_supported(320, 32, 20, 32) >= 0);
    int buckets[16 * 16];
    REQUIRE(lavie_relpos_buckets(16, 32, 32, buckets) == 0);
    REQUIRE(buckets[1] == 17 && buckets[16] == 1);                   // row q = 0 starts 0, 17; column k = 0 starts 0, 1
    printf("hostcheck: ok (%ld stubbed kernel launches)\n", lavie_hostcheck_launches());
    return 0;
}
