"""-m gpu: every HIP operator through the C ABI against the fp32 CPU oracle."""
import math

import pytest
import torch
import torch.nn.functional as F

from gpu_util import TOL_OP, f32, h16, q16, rel_l2, rows, unrows

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["auto", "row128-tiles", "split-k-3", "pingpong", "pingpong-split-k-2", "ppx-persistent"])
def ops(request):
    """Every operator test runs six times: automatic choices, the 128-row GEMM kernel forced (widest tile),
    split-K = 3 forced (fp32 slabs + fixed-order reduce) on every implicit GEMM that has >= 3 K-tiles, the
    160x320 two-group ping-pong kernel forced (every GEMM with N % 320 == 0), alone and with split-K = 2, and the
    persistent ping-pong kernel (igemm_ppx.hip, mode 7) forced on every plain GEMM whose shape it takes
    (M % 160 == 0, >= 5 K-tiles, N % 320 == 0 or N % 256 == 0)."""
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    from lavie_amd import _lib, ops as o
    lib = _lib.load()
    lib.lavie_debug_force_tile({"row128-tiles": 1, "pingpong": 3, "pingpong-split-k-2": 3, "ppx-persistent": 7}.get(request.param, 0))
    lib.lavie_debug_force_splits({"split-k-3": 3, "pingpong-split-k-2": 2}.get(request.param, 0))
    yield o
    lib.lavie_debug_force_tile(0)
    lib.lavie_debug_force_splits(0)


def gen(seed):
    return torch.Generator().manual_seed(seed)


# ------------------------------------------------------------------ linear family
@pytest.mark.parametrize("M,N,K", [(256, 320, 320), (1280, 960, 320), (154, 640, 768), (384, 1280, 1280),
                                   (130, 192, 64), (2560, 320, 1280), (77, 128, 128),
                                   # wide GEMMs with a weight matrix beyond 2 MiB: tiles run in 8 x 4 / 16 x 2 blocks, with M-tile
                                   # counts that are and are not multiples of the block height, and a ragged last M tile
                                   (2048, 2560, 640), (2450, 2560, 640), (1920, 1920, 1280),
                                   # shapes the AUTOMATIC rule sends to the persistent ping-pong kernel (>= 256 tiles of 160x320,
                                   # 5..10 K-tiles): the L0 / L1 to_out + residual GEMMs of the bench shape, and a ragged tile count
                                   (40960, 320, 320), (20480, 640, 640), (41120, 320, 320),
                                   # persistent kernel only when forced (mode 7): 160-row multiples with few tiles, N % 256 == 0
                                   (1600, 960, 320), (2560, 512, 640)])
def test_linear_bias_residual(ops, M, N, K):
    g = gen(M + N + K)
    a = q16(torch.randn(M, K, generator=g))
    w = q16(torch.randn(N, K, generator=g) / math.sqrt(K))
    b = torch.randn(N, generator=g)
    r = q16(torch.randn(M, N, generator=g))
    ref = a @ w.t() + b + r
    got = ops.linear(h16(a), h16(w), bias=f32(b), residual=h16(r))
    assert rel_l2(got, ref) < TOL_OP
    got2 = ops.linear(h16(a), h16(w))
    assert rel_l2(got2, a @ w.t()) < TOL_OP


def test_linear_residual_in_place(ops):
    g = gen(5)
    a = q16(torch.randn(512, 320, generator=g))
    w = q16(torch.randn(320, 320, generator=g) / 18)
    x = q16(torch.randn(512, 320, generator=g))
    xd = h16(x)
    ops.linear(h16(a), h16(w), residual=xd, out=xd)
    assert rel_l2(xd, a @ w.t() + x) < TOL_OP


def test_linear_per_batch_bias(ops):
    g = gen(6)
    a = q16(torch.randn(4 * 96, 128, generator=g))
    w = q16(torch.randn(320, 128, generator=g) / 11)
    b2 = torch.randn(4, 320, generator=g)
    ref = a @ w.t() + b2.repeat_interleave(96, dim=0)
    got = ops.linear(h16(a), h16(w), bias2=f32(b2), rows_per_batch=96)
    assert rel_l2(got, ref) < TOL_OP


# the projections behind a LayerNorm (attention.py:513-560) run as GEMMs on the raw rows with the norm folded into the epilogue.
# Shapes: base level 0 / 1 q|k|v, a ragged one, and the VSR level-1 temporal q|k|v at 3 / 4 / 6 frames of a 64 x 64 latent, where the
# planner picks the 128 x 64 tile (three workgroups per CU) — round 4 found that launch irreproducible, see DESIGN.md
@pytest.mark.parametrize("M,N,K", [(2560, 960, 320), (1280, 1920, 640), (1234, 320, 320), (6144, 1536, 512), (8192, 1536, 512),
                                   (12288, 1536, 512), (6144, 512, 512), (40960, 960, 320)])
def test_linear_layernorm_folded(ops, M, N, K):
    g = gen(M + N + K + 1)
    a = q16(torch.randn(M, K, generator=g) * (0.5 + torch.rand(M, 1, generator=g)) + torch.randn(M, 1, generator=g))
    gamma, beta = 1 + 0.2 * torch.randn(K, generator=g), 0.1 * torch.randn(K, generator=g)
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    bias = torch.randn(N, generator=g)
    wf = q16(w * gamma)                                   # what launch_ln_fold prepares at load time
    s, bf = wf.sum(1), w @ beta + bias
    stats = torch.stack([a.mean(1), (a.var(1, unbiased=False) + 1e-5).rsqrt()], 1)
    ref = F.layer_norm(a, (K,), gamma, beta, 1e-5) @ w.t() + bias
    ad, wd, bd, sd, std = h16(a), h16(wf), f32(bf), f32(s), f32(stats)
    got = ops.linear_lnfold(ad, wd, bd, sd, std)
    assert rel_l2(got, ref) < TOL_OP
    for _ in range(20):                                   # bit-reproducible launch to launch
        assert torch.equal(ops.linear_lnfold(ad, wd, bd, sd, std), got)


@pytest.mark.parametrize("M,C", [(256, 320), (200, 640), (2720, 640), (1600, 1280)])     # the last two: blocked tile order, ragged blocks
def test_geglu(ops, M, C):
    g = gen(C)
    a = q16(torch.randn(M, C, generator=g))
    w = q16(torch.randn(8 * C, C, generator=g) / math.sqrt(C))
    b = q16(torch.randn(8 * C, generator=g) * 0.1)
    h, gate = (a @ w.t() + b).chunk(2, dim=-1)
    ref = h * F.gelu(gate)
    wp, bp = ops.pack_geglu(h16(w), h16(b))
    got = ops.linear(h16(a), wp, bias=bp, geglu=True)
    assert got.shape == (M, 4 * C)
    assert rel_l2(got, ref) < TOL_OP


# ------------------------------------------------------------------ 3x3 convolution family
def conv_ref(x, w, b, stride=1, ups=0):
    if ups:
        x = x.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)
    return F.conv2d(x, w, b, stride=stride, padding=1)


@pytest.mark.parametrize("n,cin,cout,h,w,stride,ups", [(4, 64, 128, 8, 8, 1, 0), (3, 320, 320, 10, 16, 1, 0),
                                                      (4, 128, 128, 8, 12, 2, 0), (2, 64, 64, 5, 7, 2, 0),
                                                      (4, 128, 64, 4, 6, 1, 1), (2, 640, 320, 5, 8, 1, 0)])
def test_conv3x3(ops, n, cin, cout, h, w, stride, ups):
    g = gen(n * cin + cout + h)
    x = q16(torch.randn(n, cin, h, w, generator=g))
    wt = q16(torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(9 * cin))
    b = torch.randn(cout, generator=g)
    ref = conv_ref(x, wt, b, stride, ups)
    y = ops.conv3x3(h16(rows(x)), ops.pack_conv3x3(h16(wt)), f32(b), n, h, w, stride=stride, ups=ups)
    got = unrows(y.float().cpu(), n, ref.shape[2], ref.shape[3])
    assert rel_l2(got, ref) < TOL_OP


def test_conv3x3_concat_shortcut_temb(ops):
    """conv over [x1 | x2] + fused 1x1 shortcut of raw [s1 | s2] + per-video bias (ResnetBlock3D conv2 / conv1)."""
    g = gen(11)
    n, h, w, c1, c2, cout = 4, 6, 8, 128, 64, 192
    x1, x2 = q16(torch.randn(n, c1, h, w, generator=g)), q16(torch.randn(n, c2, h, w, generator=g))
    s1, s2 = q16(torch.randn(n, 64, h, w, generator=g)), q16(torch.randn(n, 128, h, w, generator=g))
    wt = q16(torch.randn(cout, c1 + c2, 3, 3, generator=g) / 40)
    ws = q16(torch.randn(cout, 192, 1, 1, generator=g) / 14)
    b = torch.randn(cout, generator=g)
    b2 = torch.randn(2, cout, generator=g)                     # 2 videos x 2 frames
    ref = F.conv2d(torch.cat([x1, x2], 1), wt, b, padding=1) + F.conv2d(torch.cat([s1, s2], 1), ws)
    ref = ref + b2.repeat_interleave(2, dim=0)[:, :, None, None]
    wp = ops.pack_conv3x3(h16(wt), h16(ws))
    y = ops.conv3x3(h16(rows(x1)), wp, f32(b), n, h, w, x2=h16(rows(x2)), sc1=h16(rows(s1)), sc2=h16(rows(s2)),
                    bias2=f32(b2), rows_per_batch=2 * h * w)
    assert rel_l2(unrows(y.float().cpu(), n, h, w), ref) < TOL_OP


@pytest.mark.parametrize("n,c1,c2,cout,h,w,splits", [
    (5, 64, 0, 160, 8, 8, 0),        # five whole 8x8 frames per 320-pixel tile
    (10, 128, 0, 320, 8, 8, 2),      # two tiles x two column tiles, split-K over the two slabs
    (4, 64, 0, 160, 10, 16, 0),      # two whole frames per tile
    (1, 64, 0, 160, 40, 16, 0),      # tiles of 20 image rows inside one frame (halo rows are real neighbours)
    (8, 128, 0, 160, 5, 8, 0),       # eight 5x8 frames per tile (the model's deepest level)
    (2, 64, 64, 160, 20, 16, 0),     # channel concat of two sources, one frame per tile
    (1, 192, 0, 320, 10, 64, 3),     # W = 64: five image rows per tile, split-K = 3 over three slabs
    (2, 64, 0, 128, 10, 64, 0),      # 128-wide column tile (the VSR widths), whole-row tiles
    (1, 128, 0, 256, 20, 128, 0),    # 2-D tiles of 10 rows x 32 columns (rows wider than a tile), 128-wide column tiles
    (2, 64, 64, 256, 10, 96, 2),     # 2-D tiles, channel concat, split-K over the two slabs, three tiles per image row
    (1, 64, 0, 160, 30, 32, 0),      # 2-D geometry is not needed here (W = 32 divides 320): still the row tiles
    (1, 64, 0, 320, 10, 512, 0),     # the VSR stage's 512-pixel rows, 160-wide column tiles
])
@pytest.mark.parametrize("force", [5, 0xC5], ids=["pipelined-loop", "pingpong-loop"])
def test_conv3x3_halo_patch_kernel(ops, n, c1, c2, cout, h, w, splits, force):
    """The 320x160 halo-patch conv kernel (igemm_patch.hip) forced on shapes it accepts, with bias, per-video bias
    and residual, in both of its K-loop builds (ping-pong groups; software-pipelined reads between in-place MFMAs);
    the same call through the default kernels must agree with it to rounding."""
    from lavie_amd import _lib
    lib = _lib.load()
    g = gen(n * 7 + c1 + cout + h)
    x1 = q16(torch.randn(n, c1, h, w, generator=g))
    x2 = q16(torch.randn(n, c2, h, w, generator=g)) if c2 else None
    wt = q16(torch.randn(cout, c1 + c2, 3, 3, generator=g) / math.sqrt(9 * (c1 + c2)))
    b = torch.randn(cout, generator=g)
    b2 = torch.randn(n, cout, generator=g)
    r = q16(torch.randn(n, cout, h, w, generator=g))
    xin = torch.cat([x1, x2], 1) if c2 else x1
    ref = F.conv2d(xin, wt, b, padding=1) + b2[:, :, None, None] + r
    wp = ops.pack_conv3x3(h16(wt))
    args = dict(x2=h16(rows(x2)) if c2 else None, bias2=f32(b2), rows_per_batch=h * w, residual=h16(rows(r)))
    try:
        lib.lavie_debug_force_tile(force)
        lib.lavie_debug_force_splits(splits)
        y = ops.conv3x3(h16(rows(x1)), wp, f32(b), n, h, w, **args)
    finally:
        lib.lavie_debug_force_tile(0)
        lib.lavie_debug_force_splits(0)
    assert rel_l2(unrows(y.float().cpu(), n, h, w), ref) < TOL_OP


# (frames, channels, source h, w): several frames per 320-pixel tile with split-K 2 (8 x 5x8: the deepest upsampler's geometry);
# two frames per tile (10x16); one frame = two tiles (20x32, the level-1 -> level-0 upsampler of the bench); four slabs, one
# frame per tile with ragged image borders everywhere (10x32); 1280 channels (20 slabs: split-K 4 on a short grid)
@pytest.mark.parametrize("n,c,h,w", [(32, 320, 5, 8), (8, 320, 10, 16), (4, 640, 20, 32), (3, 160 * 2, 10, 32), (8, 1280, 5, 8)])
def test_upsample_conv3x3_parity_form(ops, n, c, h, w):
    """Upsample3D (resnet.py:44-79): conv3x3(nearest_x2(x)) + bias as four 2x2 convs on x (igemm_patch.hip MODE 3, weights of
    coinciding taps summed at pack time) against F.conv2d on the materialised upsampled image, and against the engine's 9-tap
    gather conv with ups = 1."""
    from lavie_amd import _lib
    lib = _lib.load()
    assert lib.lavie_upsample_conv3x3_supported(n, h, w, c) == 1
    g = gen(n * 31 + c + h)
    x = q16(torch.randn(n, c, h, w, generator=g))
    wt = q16(torch.randn(c, c, 3, 3, generator=g) / math.sqrt(9 * c))
    b = torch.randn(c, generator=g)
    ref = conv_ref(x, wt, b, 1, 1)
    y = ops.upsample_conv3x3(h16(rows(x)), ops.pack_conv3x3_parity(h16(wt)), f32(b), n, h, w)
    got = unrows(y.float().cpu(), n, 2 * h, 2 * w)
    assert rel_l2(got, ref) < TOL_OP
    # every output parity and the image border on its own (a wrong tap sum or scatter shows up in one of them only)
    for py in (0, 1):
        for px in (0, 1):
            assert rel_l2(got[:, :, py::2, px::2], ref[:, :, py::2, px::2]) < TOL_OP, (py, px)
    border = torch.ones(2 * h, 2 * w, dtype=torch.bool)
    border[1:-1, 1:-1] = False
    assert rel_l2(got[:, :, border], ref[:, :, border]) < TOL_OP
    y9 = ops.conv3x3(h16(rows(x)), ops.pack_conv3x3(h16(wt)), f32(b), n, h, w, ups=1)
    assert rel_l2(y, y9) < 2e-3


def test_upsample_conv3x3_parity_rejects_other_geometry(ops):
    from lavie_amd import _lib
    lib = _lib.load()
    assert lib.lavie_upsample_conv3x3_supported(2, 6, 12, 320) == 0          # 12-pixel rows do not tile 320
    assert lib.lavie_upsample_conv3x3_supported(2, 10, 16, 256) == 0         # 256 channels: no 160-wide column tile
    with pytest.raises(RuntimeError):
        ops.upsample_conv3x3(torch.zeros(2 * 6 * 12, 320, dtype=torch.float16, device="cuda"),
                             torch.zeros(4, 320, 1280, dtype=torch.float16, device="cuda"), torch.zeros(320, device="cuda"), 2, 6, 12)


def test_conv3x3_residual(ops):
    g = gen(12)
    n, h, w, c = 2, 8, 8, 64
    x = q16(torch.randn(n, c, h, w, generator=g))
    r = q16(torch.randn(n, c, h, w, generator=g))
    wt = q16(torch.randn(c, c, 3, 3, generator=g) / 24)
    b = torch.randn(c, generator=g)
    y = ops.conv3x3(h16(rows(x)), ops.pack_conv3x3(h16(wt)), f32(b), n, h, w, residual=h16(rows(r)))
    assert rel_l2(unrows(y.float().cpu(), n, h, w), F.conv2d(x, wt, b, padding=1) + r) < TOL_OP


# ------------------------------------------------------------------ norms
@pytest.mark.parametrize("b,c,f,h,w", [(2, 320, 4, 8, 8), (2, 256, 16, 4, 4), (1, 1280, 2, 5, 8), (2, 2560, 2, 2, 4)])
def test_group_norm_video_domain(ops, b, c, f, h, w):
    """5-D GroupNorm: statistics across frames (resnet.py:180) + SiLU."""
    g = gen(c + f)
    x = q16(torch.randn(b, c, f, h, w, generator=g) * torch.linspace(0.5, 2.0, f).reshape(1, 1, f, 1, 1) + 0.3)
    gamma, beta = 1 + 0.1 * torch.randn(c, generator=g), 0.1 * torch.randn(c, generator=g)
    ref = F.silu(F.group_norm(x, 32, gamma, beta, 1e-5))
    xr = h16(x.permute(0, 2, 3, 4, 1).reshape(-1, c))
    y = ops.group_norm(xr, f32(gamma), f32(beta), nb=b, groups=32, eps=1e-5, silu=True)
    got = y.float().cpu().reshape(b, f, h, w, c).permute(0, 4, 1, 2, 3)
    assert rel_l2(got, ref) < TOL_OP


def test_group_norm_frame_domain(ops):
    """per-frame GroupNorm eps 1e-6, no activation (attention.py:324,369); must differ from the video domain."""
    g = gen(21)
    b, c, f, h, w = 2, 320, 4, 8, 8
    x = q16(torch.randn(b, c, f, h, w, generator=g) * torch.linspace(0.5, 2.0, f).reshape(1, 1, f, 1, 1))
    gamma, beta = 1 + 0.1 * torch.randn(c, generator=g), 0.1 * torch.randn(c, generator=g)
    frames = x.permute(0, 2, 1, 3, 4).reshape(b * f, c, h, w)
    ref = F.group_norm(frames, 32, gamma, beta, 1e-6)
    y = ops.group_norm(h16(rows(frames)), f32(gamma), f32(beta), nb=b * f, groups=32, eps=1e-6, silu=False)
    got = unrows(y.float().cpu(), b * f, h, w)
    assert rel_l2(got, ref) < TOL_OP
    wrong = F.group_norm(x, 32, gamma, beta, 1e-6).permute(0, 2, 1, 3, 4).reshape(b * f, c, h, w)
    assert rel_l2(got, wrong) > 0.1


def test_group_norm_concat_straddling_group(ops):
    """[x1 | x2] with 1280 + 640 channels: 60 channels per group, group 21 straddles the two tensors."""
    g = gen(22)
    b, f, h, w, c1, c2 = 1, 2, 4, 4, 1280, 640
    x1 = q16(torch.randn(b, c1, f, h, w, generator=g) + 0.5)
    x2 = q16(torch.randn(b, c2, f, h, w, generator=g) * 2.0)
    gamma, beta = 1 + 0.1 * torch.randn(c1 + c2, generator=g), 0.1 * torch.randn(c1 + c2, generator=g)
    ref = F.silu(F.group_norm(torch.cat([x1, x2], 1), 32, gamma, beta, 1e-5))
    r1 = h16(x1.permute(0, 2, 3, 4, 1).reshape(-1, c1))
    r2 = h16(x2.permute(0, 2, 3, 4, 1).reshape(-1, c2))
    y = ops.group_norm(r1, f32(gamma), f32(beta), nb=b, groups=32, eps=1e-5, silu=True, x2=r2)
    got = y.float().cpu().reshape(b, f, h, w, c1 + c2).permute(0, 4, 1, 2, 3)
    assert rel_l2(got, ref) < TOL_OP


@pytest.mark.parametrize("rows_,c", [(1000, 320), (333, 640), (64, 1280), (5, 256)])
def test_layer_norm(ops, rows_, c):
    g = gen(c)
    x = q16(torch.randn(rows_, c, generator=g) * 2 + 0.7)
    gamma, beta = 1 + 0.1 * torch.randn(c, generator=g), 0.1 * torch.randn(c, generator=g)
    ref = F.layer_norm(x, (c,), gamma, beta, 1e-5)
    assert rel_l2(ops.layer_norm(h16(x), f32(gamma), f32(beta)), ref) < TOL_OP


# ------------------------------------------------------------------ attention cores
def attn_ref(q, k, v, heads, kv_div=1):
    nb, lq, c = q.shape
    dh = c // heads
    if kv_div > 1:
        k, v = k.repeat_interleave(kv_div, dim=0), v.repeat_interleave(kv_div, dim=0)
    qh = q.reshape(nb, lq, heads, dh).permute(0, 2, 1, 3)
    kh = k.reshape(nb, -1, heads, dh).permute(0, 2, 1, 3)
    vh = v.reshape(nb, -1, heads, dh).permute(0, 2, 1, 3)
    p = torch.softmax(qh @ kh.transpose(-1, -2) * dh ** -0.5, dim=-1)
    return (p @ vh).permute(0, 2, 1, 3).reshape(nb, lq, c)


@pytest.mark.parametrize("nb,l,c", [(3, 160, 1280), (2, 640, 640), (2, 200, 320), (4, 40, 1280), (2, 64, 320), (1, 300, 320),
                                     (2, 300, 256), (2, 200, 512), (1, 330, 1024), (2, 50, 512)])     # head dims 32 / 64 / 128 (VSR)
def test_self_attention_fused_qkv(ops, nb, l, c):
    g = gen(l + c)
    qkv = q16(torch.randn(nb * l, 3 * c, generator=g))
    q, k, v = (t.reshape(nb, l, c) for t in qkv.split(c, dim=1))
    ref = attn_ref(q, k, v, 8).reshape(nb * l, c)
    d = h16(qkv)
    got = ops.attention(d[:, :c], d[:, c:2 * c], d[:, 2 * c:], nb=nb, lq=l, lk=l, heads=8)
    assert rel_l2(got, ref) < TOL_OP


def test_self_attention_sharp_softmax(ops):
    """one dominant key per query, placed in a later tile: exercises the online-softmax rescale."""
    g = gen(31)
    nb, l, c = 1, 192, 320
    q = torch.randn(nb, l, c, generator=g)
    k = torch.randn(nb, l, c, generator=g)
    k[:, 150] = q[:, 7] * 3.0                      # key 150 (third tile) matches query 7 strongly
    v = torch.randn(nb, l, c, generator=g)
    q, k, v = q16(q), q16(k), q16(v)
    ref = attn_ref(q, k, v, 8).reshape(nb * l, c)
    got = ops.attention(h16(q.reshape(-1, c)), h16(k.reshape(-1, c)), h16(v.reshape(-1, c)), nb=nb, lq=l, lk=l, heads=8)
    assert rel_l2(got, ref) < TOL_OP


@pytest.mark.parametrize("b,f,d,c", [(2, 4, 64, 320), (2, 16, 40, 1280), (1, 2, 130, 640), (1, 3, 260, 256), (2, 2, 200, 512),
                                     (1, 4, 70, 1024)])
def test_cross_attention_text(ops, b, f, d, c):
    """77 text keys shared by the f frames of a video (attention.py:364,529-532)."""
    g = gen(d + c)
    q = q16(torch.randn(b * f, d, c, generator=g))
    kv = q16(torch.randn(b, 77, 2 * c, generator=g))
    ref = attn_ref(q, kv[..., :c], kv[..., c:], 8, kv_div=f).reshape(-1, c)
    kvd = h16(kv.reshape(-1, 2 * c))
    got = ops.attention(h16(q.reshape(-1, c)), kvd[:, :c], kvd[:, c:], nb=b * f, lq=d, lk=77, heads=8, kv_batch_div=f)
    assert rel_l2(got, ref) < TOL_OP


@pytest.mark.parametrize("b,f,d,c", [(2, 16, 24, 320), (1, 16, 10, 640), (2, 16, 5, 1280), (1, 4, 7, 320), (1, 13, 3, 256),
                                     (1, 61, 9, 320), (2, 33, 4, 640), (1, 61, 3, 1280), (1, 17, 5, 256),
                                     # several tiles per persistent workgroup of the streaming kernel, full and short clips
                                     (2, 16, 700, 320), (1, 8, 1500, 640), (2, 16, 300, 1280), (1, 3, 2000, 320),
                                     (1, 61, 300, 320), (1, 40, 400, 640), (1, 64, 100, 1280),
                                     # the VSR widths (head dims 32 / 64 / 128): 256- and 128-channel head groups
                                     (2, 8, 700, 256), (1, 8, 400, 1024), (1, 40, 300, 512), (1, 5, 640, 512)])
def test_temporal_attention(ops, b, f, d, c):
    from oracle import unet_fp32 as O
    g = gen(f * d + c)
    heads, dh = 8, c // 8
    qkv = q16(torch.randn(b * f * d, 3 * c, generator=g))
    emb = q16(torch.randn(32, heads, generator=g))
    table = O.rel_pos_bucket_table(f, 32, 32)
    bias = emb[table].permute(2, 0, 1).contiguous()                       # [heads, f, f]
    # oracle works on (b d) f c sequences: regroup the (b f d) token order
    seq = qkv.reshape(b, f, d, 3, heads, dh).permute(3, 0, 2, 4, 1, 5).reshape(3, b * d, heads, f, dh)
    ref = O.temporal_attention_core(seq[0], seq[1], seq[2], bias, 32)     # [(b d), heads, f, dh]
    ref = ref.reshape(b, d, heads, f, dh).permute(0, 3, 1, 2, 4).reshape(b * f * d, c)
    cos, sin = ops.rotary_tables(f, 32)
    got = ops.temporal_attention(h16(qkv), b, f, d, heads, f32(bias), cos, sin)
    assert rel_l2(got, ref) < TOL_OP
    # the tile kernel (explicit LDS budget) and the streaming kernel (default for <= 16 frames) do the same arithmetic up to
    # the compiler's choice of fused multiply-adds in the rotary step: a few outputs differ in their last one or two fp16 bits
    from lavie_amd import _lib
    _lib.load().lavie_debug_temporal_budget(33000)
    try:
        tiled = ops.temporal_attention(h16(qkv), b, f, d, heads, f32(bias), cos, sin)
    finally:
        _lib.load().lavie_debug_temporal_budget(0)
    assert rel_l2(got, tiled.float().cpu()) < 1e-4 and (got.float() - tiled.float()).abs().max() <= 4e-3 * tiled.float().abs().max()


@pytest.mark.parametrize("b,f,d,c", [(1, 16, 300, 320), (1, 61, 200, 320), (1, 33, 150, 1280), (1, 5, 7, 256)])
def test_temporal_attention_plain(ops, b, f, d, c):
    """The interpolation model's temporal attention: no rotary embedding, no relative-position bias
    (interpolation/models/attention.py:566-606 with use_relative_position=False) = softmax(q k^T / sqrt(dh)) v per pixel."""
    g = gen(f * d + c + 1)
    heads, dh = 8, c // 8
    qkv = q16(torch.randn(b * f * d, 3 * c, generator=g))
    seq = qkv.reshape(b, f, d, 3, heads, dh).permute(3, 0, 2, 4, 1, 5).reshape(3, b * d, heads, f, dh)
    att = torch.softmax(seq[0] @ seq[1].transpose(-1, -2) * dh ** -0.5, dim=-1) @ seq[2]
    ref = att.reshape(b, d, heads, f, dh).permute(0, 3, 1, 2, 4).reshape(b * f * d, c)
    bias = torch.zeros(heads, f, f)
    got = ops.temporal_attention(h16(qkv), b, f, d, heads, f32(bias), None, None, rot_dim=0)
    assert rel_l2(got, ref) < TOL_OP


def test_relpos_bias_matches_oracle_tables(ops):
    from oracle import unet_fp32 as O
    for f in (16, 61, 1, 9):
        assert torch.equal(ops.relpos_buckets(f), O.rel_pos_bucket_table(f, 32, 32))


# ------------------------------------------------------------------ CFG + DDPM step
def test_cfg_ddpm_step(ops):
    from oracle.ddpm import DDPMSchedule
    from lavie_amd.scheduling_ddpm import DDPMScheduler
    g = gen(41)
    n = 4 * 16 * 8 * 8
    eps = q16(torch.randn(2, n, generator=g))
    x = torch.randn(n, generator=g)
    noise = torch.randn(n, generator=g)
    osch, sch = DDPMSchedule(), DDPMScheduler()
    osch.set_timesteps(50)
    sch.set_timesteps(50)
    for t in (980, 500, 20, 0):
        guided = eps[0] + 7.5 * (eps[1] - eps[0])
        ref = osch.step(guided, t, x, noise if t > 0 else None)
        xd = f32(x)
        model_in = torch.empty(2, n, dtype=torch.float16, device="cuda")
        ops.cfg_ddpm_step(h16(eps), xd, f32(noise) if t > 0 else None, model_in, 7.5, sch.coefficients(t))
        assert rel_l2(xd, ref) < 1e-5
        assert rel_l2(model_in[0], ref) < 1e-3 and torch.equal(model_in[0], model_in[1])
