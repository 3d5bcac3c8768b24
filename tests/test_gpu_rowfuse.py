"""-m gpu: the row-resident fused transformer sub-blocks (lavie_amd/csrc/rowfuse.hip) through the C ABI against fp32 CPU
restatements of the reference lines they replace."""
import math

import pytest
import torch
import torch.nn.functional as F

from gpu_util import TOL_OP, f32, h16, q16, rel_l2

pytestmark = pytest.mark.gpu


def gen(seed):
    return torch.Generator().manual_seed(seed)


# 128 rows = one workgroup pass; 2608 = 20 passes + a ragged one (48 rows: five waves idle, clamped loads);
# 40960 = 320 passes on 256 persistent workgroups (some run two passes: the weight ring wraps across passes)
@pytest.mark.parametrize("M", [128, 2608, 40960])
def test_geglu_mlp_fused(M):
    """hidden_states = self.ff(self.norm3(hidden_states)) + hidden_states (attention.py:558; FeedForward / GEGLU:
    vsr/models/diffusers_attention.py:734-822) as one kernel, C = 320 (level 0 of the base UNet)."""
    from lavie_amd import ops
    C = 320
    g = gen(M)
    x = q16(torch.randn(M, C, generator=g) * 1.5 + 0.3 * torch.randn(1, C, generator=g))
    w1 = q16(torch.randn(8 * C, C, generator=g) / math.sqrt(C))
    b1 = q16(torch.randn(8 * C, generator=g) * 0.2)
    w2 = q16(torch.randn(C, 4 * C, generator=g) / math.sqrt(4 * C))
    b2 = torch.randn(C, generator=g) * 0.2
    gamma, beta = 1.0 + 0.2 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    ln = F.layer_norm(x, (C,), gamma, beta, 1e-5)
    h, gate = (ln @ w1.t() + b1).chunk(2, dim=-1)
    delta = (h * F.gelu(gate)) @ w2.t() + b2
    ref = x + delta
    img, b1img = ops.pack_geglu_mlp(h16(w1), h16(b1), h16(w2))
    xd = h16(x)
    got = ops.geglu_mlp(xd, img, b1img, f32(gamma), f32(beta), f32(b2))
    assert rel_l2(got, ref) < TOL_OP
    assert rel_l2(got.float().cpu() - x, delta) < 4e-3            # the residual must not hide an error in the product
    got2 = ops.geglu_mlp(xd, img, b1img, f32(gamma), f32(beta), f32(b2))
    assert torch.equal(got, got2)                                  # no atomics, fixed order: bit-reproducible
    ops.geglu_mlp(xd, img, b1img, f32(gamma), f32(beta), f32(b2), out=xd)      # in place, as the engine runs it
    assert torch.equal(xd, got)


def test_geglu_mlp_rejects_unbuilt_width():
    from lavie_amd import ops
    with pytest.raises(RuntimeError):
        ops.pack_geglu_mlp(torch.zeros(8 * 256, 256, dtype=torch.float16, device="cuda"),
                           torch.zeros(8 * 256, dtype=torch.float16, device="cuda"),
                           torch.zeros(256, 4 * 256, dtype=torch.float16, device="cuda"))


# grid = min(pixels, 256): 24 pixels -> 24 workgroups of one pixel (seven idle waves each); 700 pixels -> 256 workgroups with
# 2-3 pixels; 5120 = the bench's level-0 shape (20 pixels per workgroup: passes of 8 + 8 + 4, the weight ring restarts at
# every pass)
@pytest.mark.parametrize("B,D", [(2, 12), (1, 700), (2, 2560)])
def test_temporal_block_fused(B, D):
    """hidden_states = self.attn_temp(self.norm_temp(hidden_states)) + hidden_states between the two rearranges of
    BasicTransformerBlock.forward (attention.py:548-555), TemporalAttention.forward / _attention (:580-667) — as one kernel on
    token rows in (b f) d order, against the oracle's restatement (pinned to the imported reference class in
    tests/test_oracle_vs_reference.py::test_temporal_attention)."""
    from lavie_amd import ops
    from oracle import unet_fp32 as O
    C, heads, Fr = 320, 8, 16
    cfg = O.UNetConfig()
    g = gen(B * 1000 + D)
    sd = {"to_q.weight": q16(torch.randn(C, C, generator=g) / math.sqrt(C)),
          "to_k.weight": q16(torch.randn(C, C, generator=g) / math.sqrt(C)),
          "to_v.weight": q16(torch.randn(C, C, generator=g) / math.sqrt(C)),
          "to_out.0.weight": q16(torch.randn(C, C, generator=g) / math.sqrt(C)),
          "to_out.0.bias": torch.randn(C, generator=g) * 0.2,
          "time_rel_pos_bias.relative_attention_bias.weight": q16(torch.randn(cfg.rel_buckets, heads, generator=g))}
    gamma, beta = 1.0 + 0.2 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    x = q16(torch.randn(B * Fr * D, C, generator=g) * 1.5 + 0.3 * torch.randn(1, C, generator=g))
    xr = x.reshape(B, Fr, D, C).permute(0, 2, 1, 3).reshape(B * D, Fr, C)                      # (b f) d c -> (b d) f c (:550)
    delta = O.temporal_attention(sd, "", F.layer_norm(xr, (C,), gamma, beta, 1e-5), cfg)
    back = lambda t: t.reshape(B, D, Fr, C).permute(0, 2, 1, 3).reshape(B * Fr * D, C)          # (:555)
    ref, delta = back(xr + delta), back(delta)
    inv = 10000.0 ** (-torch.arange(0, 32, 2, dtype=torch.float32) / 32)
    ang = torch.arange(Fr, dtype=torch.float32).reshape(Fr, 1) * inv.reshape(1, -1)
    relbias = O.rel_pos_bias(sd, "", Fr, cfg).contiguous()
    img = ops.pack_temporal_block(h16(sd["to_q.weight"]), h16(sd["to_k.weight"]), h16(sd["to_v.weight"]), h16(sd["to_out.0.weight"]))
    xd = h16(x)
    args = (img, f32(gamma), f32(beta), f32(sd["to_out.0.bias"]), f32(relbias), f32(ang.cos()), f32(ang.sin()), B, Fr, D, heads, 32,
            (C // heads) ** -0.5)
    got = ops.temporal_block(xd, *args)
    assert rel_l2(got, ref) < TOL_OP
    assert rel_l2(got.float().cpu() - x, delta) < 4e-3
    assert torch.equal(got, ops.temporal_block(xd, *args))          # bit-reproducible
    ops.temporal_block(xd, *args, out=xd)                           # in place, as the engine runs it
    assert torch.equal(xd, got)


# (videos, rows per video, context length): one pass; ragged runs with passes cut at the video boundary (163 tiles per video on
# 256 workgroups: some workgroups own tiles of both videos); the bench's level-0 shape (20 tiles per workgroup: 8 + 8 + 4);
# short contexts (a single key tile in use; exactly 64 keys; the full 80)
@pytest.mark.parametrize("B,P,L", [(1, 128, 77), (2, 2608, 77), (2, 40960, 77), (3, 48, 5), (1, 1024, 64), (2, 512, 80)])
def test_cross_block_fused(B, P, L):
    """hidden_states = attn1(...) + hidden_states (the to_out projection and residual of attention.py:513-522), then
    hidden_states = attn2(norm2(hidden_states), encoder_hidden_states) + hidden_states (:524-534; CrossAttention.forward /
    _attention :253-335, K / V per video as the engine caches them) — as one kernel, against an fp32 restatement."""
    from lavie_amd import ops
    C, heads = 320, 8
    dh = C // heads
    M = B * P
    g = gen(B * 100000 + P * 10 + L)
    rnd = lambda *s: torch.randn(*s, generator=g)
    x = q16(rnd(M, C) * 1.5 + 0.3 * rnd(1, C))
    att = q16(rnd(M, C))
    wo1, wq2, wo2 = [q16(rnd(C, C) / math.sqrt(C)) for _ in range(3)]
    bo1, bo2 = rnd(C) * 0.2, rnd(C) * 0.2
    gamma, beta = 1.0 + 0.2 * rnd(C), 0.1 * rnd(C)
    kv = q16(rnd(B * L, 2 * C) * torch.cat([torch.full((C,), 1.5), torch.ones(C)]))      # sharper scores than unit keys give
    scale = dh ** -0.5
    x1 = x + att @ wo1.t() + bo1
    q = (F.layer_norm(x1, (C,), gamma, beta, 1e-5) @ wq2.t()).reshape(B, P, heads, dh).permute(0, 2, 1, 3)
    k = kv[:, :C].reshape(B, L, heads, dh).permute(0, 2, 1, 3)
    v = kv[:, C:].reshape(B, L, heads, dh).permute(0, 2, 1, 3)
    o = torch.softmax(q @ k.transpose(-1, -2) * scale, dim=-1) @ v
    delta2 = o.permute(0, 2, 1, 3).reshape(M, C) @ wo2.t() + bo2
    ref = x1 + delta2
    tmpl = ops.pack_cross_block(h16(wo1), h16(wq2), h16(wo2))
    img = ops.bind_cross_block(tmpl, h16(kv), B, L)
    xd, ad = h16(x), h16(att)
    args = (img, f32(bo1), f32(gamma), f32(beta), f32(bo2), P, L, heads, scale)
    got = ops.cross_block(ad, xd, *args)
    assert rel_l2(got, ref) < TOL_OP
    assert rel_l2(got.float().cpu() - x, ref - x) < 4e-3          # the residual must not hide an error in the products
    assert torch.equal(got, ops.cross_block(ad, xd, *args))        # bit-reproducible
    ops.cross_block(ad, xd, *args, out=xd)                         # in place, as the engine runs it
    assert torch.equal(xd, got)


def test_cross_block_rejects_long_context():
    from lavie_amd import ops
    z = torch.zeros(320, 320, dtype=torch.float16, device="cuda")
    tmpl = ops.pack_cross_block(z, z, z)
    with pytest.raises(RuntimeError):
        ops.bind_cross_block(tmpl, torch.zeros(81, 640, dtype=torch.float16, device="cuda"), 1, 81)


# 2560 rows per frame = the bench's level-0 frames (passes of 128 rows never straddle a frame); 256: sixteen tiles per frame;
# 48: three tiles per frame, so the waves of one pass work under different frames' scale / shift pairs, and a ragged last pass
@pytest.mark.parametrize("NB,D", [(4, 2560), (6, 256), (5, 48)])
def test_proj_qkv_fused(NB, D):
    """Transformer3DModel's per-frame GroupNorm + proj_in (attention.py:369-373) and BasicTransformerBlock's norm1 + attn1 to_q / to_k /
    to_v (attention.py:513-516; CrossAttention :154, 177-178) as one kernel at C = 320, against the oracle's own functions."""
    from lavie_amd import ops
    from oracle import unet_fp32 as O
    C, G = 320, 32
    M = NB * D
    g = gen(NB * D)
    x = q16(torch.randn(M, C, generator=g) * 1.3 + 0.4 * torch.randn(NB, 1, C, generator=g).repeat_interleave(D, 0).reshape(M, C))
    gn_g, gn_b = 1.0 + 0.2 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    wpin = q16(torch.randn(C, C, generator=g) / math.sqrt(C))
    bpin = torch.randn(C, generator=g) * 0.2
    ln_g, ln_b = 1.0 + 0.2 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    wqkv = q16(torch.randn(3 * C, C, generator=g) / math.sqrt(C))
    # reference: GroupNorm over (C / 32, D) per frame (eps 1e-6), 1x1 conv, LayerNorm (eps 1e-5), three bias-free projections
    xn = F.group_norm(x.reshape(NB, D, C).transpose(1, 2), G, gn_g, gn_b, 1e-6).transpose(1, 2).reshape(M, C)
    tx_ref = xn @ wpin.t() + bpin
    qkv_ref = F.layer_norm(tx_ref, (C,), ln_g, ln_b, 1e-5) @ wqkv.t()
    xd = h16(x)
    ab = ops.group_norm_affine(xd, f32(gn_g), f32(gn_b), NB, G, 1e-6)
    img = ops.pack_proj_qkv(h16(wpin), h16(wqkv))
    tx, qkv = ops.proj_qkv(xd, ab, D, img, f32(bpin), f32(ln_g), f32(ln_b))
    assert rel_l2(tx, tx_ref) < TOL_OP
    assert rel_l2(qkv, qkv_ref) < 2 * TOL_OP        # two chained products and two norms: the block-level tolerance of the seams
    for i in range(3):                              # q, k and v each on their own (a permutation of the row blocks would hide in the whole)
        assert rel_l2(qkv[:, i * C:(i + 1) * C], qkv_ref[:, i * C:(i + 1) * C]) < 2 * TOL_OP
    # against the unfused operators of this library on the same inputs: same rounding points up to the LayerNorm fold
    y = ops.group_norm(xd, f32(gn_g), f32(gn_b), NB, G, 1e-6, False)
    tx_u = ops.linear(y, h16(wpin), f32(bpin))
    assert rel_l2(tx, tx_u.float().cpu()) < 3e-4
    tx2, qkv2 = ops.proj_qkv(xd, ab, D, img, f32(bpin), f32(ln_g), f32(ln_b))
    assert torch.equal(tx, tx2) and torch.equal(qkv, qkv2)


def test_proj_qkv_rejects_unbuilt_width():
    from lavie_amd import ops
    with pytest.raises(RuntimeError):
        ops.pack_proj_qkv(torch.zeros(256, 256, dtype=torch.float16, device="cuda"), torch.zeros(768, 256, dtype=torch.float16, device="cuda"))
