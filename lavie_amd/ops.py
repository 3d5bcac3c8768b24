"""Operator-level entry points (the finer seam of SURVEY.md §8b) over the C ABI.

All tensors are CUDA(HIP) fp16, channels-last: a reference video tensor `[b, c, f, h, w]` is the
row-major matrix `[(b f h w), c]`, which is also the reference's token layout `(b f) (h w) c`.
Every function enqueues on torch's current stream and raises RuntimeError on failure."""
import ctypes
import math
from typing import Optional

import torch

from . import _lib


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _chk16(*ts):
    for t in ts:
        if t is None:
            continue
        if not (t.is_cuda and t.dtype == torch.float16 and t.is_contiguous()):
            raise ValueError("expected contiguous fp16 device tensors")


def _chk32(*ts):
    for t in ts:
        if t is None:
            continue
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise ValueError("expected contiguous fp32 device tensors")


def to_rows(x: torch.Tensor) -> torch.Tensor:
    """[b, c, f, h, w] -> channels-last rows [(b f h w), c] (a copy)."""
    b, c, f, h, w = x.shape
    return x.permute(0, 2, 3, 4, 1).reshape(b * f * h * w, c).contiguous()


def from_rows(rows: torch.Tensor, b: int, f: int, h: int, w: int) -> torch.Tensor:
    """channels-last rows [(b f h w), c] -> [b, c, f, h, w] (a copy)."""
    return rows.reshape(b, f, h, w, -1).permute(0, 4, 1, 2, 3).contiguous()


def linear(a, weight, bias=None, residual=None, bias2=None, rows_per_batch=0, geglu=False, out=None):
    """a[M,K] @ weight[N,K]^T (+bias fp32[N]) (+bias2 fp32[M/rows_per_batch, N]) (+residual[M,N])."""
    _chk16(a, weight, residual, out)
    _chk32(bias, bias2)
    M, K = a.shape
    N = weight.shape[0]
    n_out = N // 2 if geglu else N
    if out is None:
        out = torch.empty(M, n_out, dtype=torch.float16, device=a.device)
    lib = _lib.load()
    _lib.check(lib.lavie_linear_f16(_p(a), K, _p(weight), _p(bias), _p(bias2), N, rows_per_batch, _p(residual), n_out,
                                    _p(out), n_out, M, N, K, int(geglu), _stream()), "lavie_linear_f16")
    return out


def linear_lnfold(a, weight_folded, bias, ln_s, ln_stats):
    """rstd_m (a @ weight_folded^T - mean_m ln_s) + bias: a projection behind a LayerNorm, the norm folded into the GEMM epilogue
    (weight_folded = W * gamma, ln_s = its row sums, bias = W beta (+ b), ln_stats [M, 2] = (mean, rstd) of the rows of `a`)."""
    _chk16(a, weight_folded)
    _chk32(bias, ln_s, ln_stats)
    M, K = a.shape
    N = weight_folded.shape[0]
    out = torch.empty(M, N, dtype=torch.float16, device=a.device)
    _lib.check(_lib.load().lavie_linear_lnfold_f16(_p(a), _p(weight_folded), _p(bias), _p(ln_s), _p(ln_stats), _p(out), M, N, K,
                                                   _stream()), "lavie_linear_lnfold_f16")
    return out


def pack_geglu(weight, bias):
    """GEGLU projection [2*inner, K] -> the value/gate 16-row interleave the geglu epilogue expects."""
    _chk16(weight, bias)
    N, K = weight.shape
    w_out = torch.empty_like(weight)
    b_out = torch.empty(N, dtype=torch.float32, device=weight.device)
    _lib.check(_lib.load().lavie_pack_geglu_f16(_p(weight), _p(bias), _p(w_out), _p(b_out), N, K, _stream()))
    return w_out, b_out


def pack_geglu_mlp(w1, b1, w2):
    """ff.net.0.proj.weight [8C, C], .bias [8C], ff.net.2.weight [C, 4C] (fp16, device) -> (image, bias image) of the fused
    feed-forward kernel (lavie_geglu_mlp_f16).  Raises for a width the kernel is not built for."""
    _chk16(w1, b1, w2)
    C = w2.shape[0]
    lib = _lib.load()
    nbytes = lib.lavie_geglu_mlp_image_bytes(C)
    if nbytes == 0 or tuple(w1.shape) != (8 * C, C) or tuple(w2.shape) != (C, 4 * C):
        raise RuntimeError(f"geglu_mlp: width {C} is not built (or weight shapes do not match)")
    img = torch.empty(nbytes // 2, dtype=torch.float16, device=w1.device)
    b1img = torch.empty(lib.lavie_geglu_mlp_bias_floats(C), dtype=torch.float32, device=w1.device)
    _lib.check(lib.lavie_pack_geglu_mlp_f16(_p(w1), _p(b1), _p(w2), C, _p(img), _p(b1img), _stream()), "lavie_pack_geglu_mlp_f16")
    return img, b1img


def geglu_mlp(x, img, b1img, gamma, beta, b2, eps=1e-5, out=None):
    """x + FeedForward_GEGLU(LayerNorm(x)) in one kernel (attention.py:558); `out` may be x itself."""
    _chk16(x, img, out)
    _chk32(b1img, gamma, beta, b2)
    M, C = x.shape
    if out is None:
        out = torch.empty_like(x)
    _lib.check(_lib.load().lavie_geglu_mlp_f16(_p(x), _p(out), M, C, _p(img), _p(b1img), _p(gamma), _p(beta), _p(b2), eps,
                                               _stream()), "lavie_geglu_mlp_f16")
    return out


def pack_temporal_block(wq, wk, wv, wo, heads=8, frames=16, rot_dim=32):
    """attn_temp.to_q / to_k / to_v / to_out.0 weights [C, C] (fp16, device) -> the weight image of the fused temporal
    sub-block kernel (lavie_temporal_block_f16).  Raises for a configuration the kernel is not built for."""
    _chk16(wq, wk, wv, wo)
    C = wq.shape[0]
    lib = _lib.load()
    nbytes = lib.lavie_temporal_block_image_bytes(C, heads, frames, rot_dim)
    if nbytes == 0:
        raise RuntimeError(f"temporal_block: C={C} heads={heads} frames={frames} rot_dim={rot_dim} is not built")
    img = torch.empty(nbytes // 2, dtype=torch.float16, device=wq.device)
    _lib.check(lib.lavie_pack_temporal_block_f16(_p(wq), _p(wk), _p(wv), _p(wo), C, _p(img), _stream()), "lavie_pack_temporal_block_f16")
    return img


def temporal_block(x, img, gamma, beta, bo, relbias, rot_cos, rot_sin, B, F, D, heads, rot_dim, scale, eps=1e-5, out=None):
    """x + to_out(attn_temp(norm_temp(x))) on token rows [(b f) d, C] in one kernel (attention.py:548-555, 580-667)."""
    _chk16(x, img, out)
    _chk32(gamma, beta, bo, relbias, rot_cos, rot_sin)
    C = x.shape[1]
    if out is None:
        out = torch.empty_like(x)
    _lib.check(_lib.load().lavie_temporal_block_f16(_p(x), _p(out), B, F, D, C, heads, _p(img), _p(gamma), _p(beta), _p(bo),
                                                    _p(relbias), _p(rot_cos), _p(rot_sin), rot_dim, scale, eps, _stream()),
               "lavie_temporal_block_f16")
    return out


def pack_cross_block(wo1, wq2, wo2, heads=8):
    """attn1.to_out.0 / attn2.to_q / attn2.to_out.0 weights [C, C] (fp16, device) -> the weight part of the image of the fused
    text cross-attention kernel (lavie_cross_block_f16).  Raises for a configuration the kernel is not built for."""
    _chk16(wo1, wq2, wo2)
    C = wo1.shape[0]
    lib = _lib.load()
    nbytes = lib.lavie_cross_block_image_bytes(C, heads)
    if nbytes == 0:
        raise RuntimeError(f"cross_block: C={C} heads={heads} is not built")
    tmpl = torch.empty(nbytes // 2, dtype=torch.float16, device=wo1.device)
    _lib.check(lib.lavie_pack_cross_block_f16(_p(wo1), _p(wq2), _p(wo2), C, _p(tmpl), _stream()), "lavie_pack_cross_block_f16")
    return tmpl


def bind_cross_block(tmpl, kv, B, ctx_len):
    """Completes one image per video from kv [B * ctx_len, 2C] (attn2.to_k | to_v of the text context): once per context."""
    _chk16(tmpl, kv)
    C = kv.shape[1] // 2
    if tuple(kv.shape) != (B * ctx_len, 2 * C):
        raise RuntimeError("cross_block: kv must be [B * ctx_len, 2C]")
    img = torch.empty(B * tmpl.numel(), dtype=torch.float16, device=tmpl.device)
    _lib.check(_lib.load().lavie_bind_cross_block_f16(_p(tmpl), _p(kv), B, ctx_len, C, _p(img), _stream()), "lavie_bind_cross_block_f16")
    return img


def cross_block(att, x, img, bo1, gamma, beta, bo2, rows_per_batch, ctx_len, heads, scale, eps=1e-5, out=None):
    """x + to_out1(att), then + to_out2(attn2(norm2(.), K, V)) in one kernel (attention.py:513-534); `out` may be x itself."""
    _chk16(att, x, img, out)
    _chk32(bo1, gamma, beta, bo2)
    M, C = x.shape
    if out is None:
        out = torch.empty_like(x)
    _lib.check(_lib.load().lavie_cross_block_f16(_p(att), _p(x), _p(out), M, rows_per_batch, C, heads, _p(img), _p(bo1), _p(gamma),
                                                 _p(beta), _p(bo2), ctx_len, scale, eps, _stream()), "lavie_cross_block_f16")
    return out


def pack_conv3x3(weight, shortcut_weight=None):
    """[Cout, Cin, 3, 3] (+ optional 1x1 shortcut [Cout, Csc, 1, 1]) -> [Cout, 9*Cin (+ Csc)]."""
    _chk16(weight, shortcut_weight)
    cout, cin = weight.shape[:2]
    csc = 0 if shortcut_weight is None else shortcut_weight.shape[1]
    ld = 9 * cin + csc
    out = torch.empty(cout, ld, dtype=torch.float16, device=weight.device)
    _lib.check(_lib.load().lavie_pack_conv3x3_f16(_p(weight), _p(out), cout, cin, ld, 0, _stream()))
    if csc:
        out[:, 9 * cin:] = shortcut_weight.reshape(cout, csc)
    return out


_zero_pages = {}


def _zero_page(device):
    z = _zero_pages.get(device)
    if z is None:
        z = torch.zeros(256, dtype=torch.float16, device=device)
        _zero_pages[device] = z
    return z


def conv3x3(x1, wp, bias, ni, hi, wi, x2=None, sc1=None, sc2=None, bias2=None, rows_per_batch=0, residual=None,
            stride=1, ups=0):
    """Per-frame 3x3 conv, pad 1, on channels-last rows [(ni hi wi), C]; see lavie_conv3x3_f16."""
    _chk16(x1, x2, sc1, sc2, wp, residual)
    _chk32(bias, bias2)
    cout = wp.shape[0]
    ho = hi * 2 if ups else (hi - 1) // stride + 1
    wo = wi * 2 if ups else (wi - 1) // stride + 1
    y = torch.empty(ni * ho * wo, cout, dtype=torch.float16, device=x1.device)
    c = lambda t: 0 if t is None else t.shape[1]
    lib = _lib.load()
    _lib.check(lib.lavie_conv3x3_f16(_p(x1), c(x1), _p(x2), c(x2), _p(sc1), c(sc1), _p(sc2), c(sc2), _p(wp), _p(bias),
                                     _p(bias2), cout, rows_per_batch, _p(residual), _p(y), ni, hi, wi, cout, stride, ups,
                                     _p(_zero_page(x1.device)), _stream()), "lavie_conv3x3_f16")
    return y


def pack_conv3x3_parity(weight):
    """[C, C, 3, 3] (fp16, device) -> the four parity weight sets [4, C, 4C] of the upsample conv (lavie_upsample_conv3x3_f16)."""
    _chk16(weight)
    cout, cin = weight.shape[0], weight.shape[1]
    out = torch.empty(4, cout, 4 * cin, dtype=torch.float16, device=weight.device)
    _lib.check(_lib.load().lavie_pack_conv3x3_parity_f16(_p(weight.contiguous()), _p(out), cout, cin, _stream()), "lavie_pack_conv3x3_parity_f16")
    return out


def upsample_conv3x3(x, wpar, bias, ni, hi, wi):
    """conv3x3(nearest_x2(x)) + bias (Upsample3D, resnet.py:44-79) in parity form: x [ni*hi*wi, C] rows -> [ni*2hi*2wi, C] rows.
    Raises where the kernel's geometry does not hold (use conv3x3(..., ups=1) there)."""
    _chk16(x, wpar)
    _chk32(bias)
    c = x.shape[1]
    y = torch.empty(ni * 4 * hi * wi, c, dtype=torch.float16, device=x.device)
    _lib.check(_lib.load().lavie_upsample_conv3x3_f16(_p(x), _p(wpar), _p(bias), _p(y), ni, hi, wi, c, _p(_zero_page(x.device)), _stream()),
               "lavie_upsample_conv3x3_f16")
    return y


def pack_temporal_conv(weight):
    """nn.Conv3d weight [Cout, Cin, T, 1, 1] (T = 3 or 5) -> [Cout, T*Cin] in the implicit GEMM's K order."""
    _chk16(weight)
    cout, cin, taps = weight.shape[:3]
    out = torch.empty(cout, taps * cin, dtype=torch.float16, device=weight.device)
    _lib.check(_lib.load().lavie_pack_temporal_conv_f16(_p(weight.contiguous()), _p(out), cout, cin, taps, _stream()),
               "lavie_pack_temporal_conv_f16")
    return out


def temporal_conv(x, wp, bias, b, frames, d, taps, bias2=None, residual=None):
    """Conv3d (taps, 1, 1), padding (taps // 2, 0, 0), over the frame axis of token rows [(b f d), C] (the VSR stage's
    ResnetBlock3DCNN convs); bias2 [b, Cout] = per-video time-embedding projection; see lavie_temporal_conv_f16."""
    _chk16(x, wp, residual)
    _chk32(bias, bias2)
    cout = wp.shape[0]
    y = torch.empty(b * frames * d, cout, dtype=torch.float16, device=x.device)
    _lib.check(_lib.load().lavie_temporal_conv_f16(_p(x), x.shape[1], _p(wp), _p(bias), _p(bias2), cout,
                                                   frames * d if bias2 is not None else 0, _p(residual), _p(y), b, frames, d,
                                                   cout, taps, _p(_zero_page(x.device)), _stream()),
               "lavie_temporal_conv_f16")
    return y


def group_norm(x1, gamma, beta, nb, groups, eps, silu, x2=None):
    """GroupNorm (+SiLU) over rows; `nb` batches share statistics over rows/nb rows each."""
    _chk16(x1, x2)
    _chk32(gamma, beta)
    rows = x1.shape[0]
    c1, c2 = x1.shape[1], 0 if x2 is None else x2.shape[1]
    y = torch.empty(rows, c1 + c2, dtype=torch.float16, device=x1.device)
    ws = torch.empty(_lib.load().lavie_group_norm_ws_floats(nb, groups), dtype=torch.float32, device=x1.device)
    _lib.check(_lib.load().lavie_group_norm_f16(_p(x1), c1, _p(x2), c2, nb, rows // nb, groups, _p(gamma), _p(beta),
                                                float(eps), int(silu), _p(ws), _p(y), _stream()), "lavie_group_norm_f16")
    return y


def group_norm_affine(x, gamma, beta, nb, groups, eps):
    """GroupNorm statistics only: the normalisation as per-(batch, channel) pairs (a, b) with norm(x) = a x + b -> [nb, C, 2] fp32
    (lavie_group_norm_affine_f16; consumed by proj_qkv)."""
    _chk16(x)
    _chk32(gamma, beta)
    rows, c = x.shape
    ab = torch.empty(nb, c, 2, dtype=torch.float32, device=x.device)
    ws = torch.empty(_lib.load().lavie_group_norm_ws_floats(nb, groups), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().lavie_group_norm_affine_f16(_p(x), c, nb, rows // nb, groups, _p(gamma), _p(beta), float(eps), _p(ws),
                                                       _p(ab), _stream()), "lavie_group_norm_affine_f16")
    return ab


def pack_proj_qkv(wpin, wqkv):
    """proj_in.weight [C, C] and the stacked attn1 to_q / to_k / to_v weights [3C, C] (fp16, device) -> the weight image of the fused
    block-head kernel (lavie_proj_qkv_f16).  Raises for a width the kernel is not built for."""
    _chk16(wpin, wqkv)
    C = wpin.shape[0]
    lib = _lib.load()
    nbytes = lib.lavie_proj_qkv_image_bytes(C)
    if nbytes == 0 or tuple(wpin.shape) != (C, C) or tuple(wqkv.shape) != (3 * C, C):
        raise RuntimeError(f"proj_qkv: width {C} is not built (or weight shapes do not match)")
    img = torch.empty(nbytes // 2, dtype=torch.float16, device=wpin.device)
    _lib.check(lib.lavie_pack_proj_qkv_f16(_p(wpin), _p(wqkv), C, _p(img), _stream()), "lavie_pack_proj_qkv_f16")
    return img


def proj_qkv(x, gn_ab, rows_per_domain, img, bpin, ln_gamma, ln_beta, eps=1e-5):
    """tx = proj_in(GroupNorm(x)), qkv = to_qkv(LayerNorm(tx)) in one kernel (attention.py:369-373, 513-516) -> (tx [M, C], qkv [M, 3C])."""
    _chk16(x, img)
    _chk32(gn_ab, bpin, ln_gamma, ln_beta)
    M, C = x.shape
    tx = torch.empty(M, C, dtype=torch.float16, device=x.device)
    qkv = torch.empty(M, 3 * C, dtype=torch.float16, device=x.device)
    _lib.check(_lib.load().lavie_proj_qkv_f16(_p(x), _p(gn_ab), rows_per_domain, _p(img), _p(bpin), _p(ln_gamma), _p(ln_beta), eps,
                                              _p(tx), _p(qkv), M, C, _stream()), "lavie_proj_qkv_f16")
    return tx, qkv


def layer_norm(x, gamma, beta, eps=1e-5):
    _chk16(x)
    _chk32(gamma, beta)
    y = torch.empty_like(x)
    _lib.check(_lib.load().lavie_layer_norm_f16(_p(x), _p(gamma), _p(beta), _p(y), x.shape[0], x.shape[1], float(eps),
                                                _stream()), "lavie_layer_norm_f16")
    return y


def attention(q, k, v, nb, lq, lk, heads, kv_batch_div=1, scale=None):
    """q [nb*lq, *], k/v [(nb/kv_batch_div)*lk, *] may be column slices of wider row-major tensors."""
    for t in (q, k, v):
        if not (t.is_cuda and t.dtype == torch.float16 and t.stride(1) == 1):
            raise ValueError("attention operands must be fp16 device tensors with unit column stride")
    c = q.shape[1]
    dh = c // heads
    o = torch.empty(nb * lq, c, dtype=torch.float16, device=q.device)
    scale = dh ** -0.5 if scale is None else scale
    _lib.check(_lib.load().lavie_attention_f16(_p(q), q.stride(0), _p(k), k.stride(0), _p(v), v.stride(0), _p(o), c, nb,
                                               lq, lk, heads, dh, kv_batch_div, float(scale), _stream()),
               "lavie_attention_f16")
    return o


def sparse_causal_attention(q, k, v, nb, frames, d, heads, scale=None):
    """SparseCausalAttention core (interpolation/models/attention.py:609-665): q/k/v [nb*d, *] per-frame rows (column
    slices of a wider tensor allowed); frame f of a video attends to [first frame || frame max(f-1, 0)] of that video."""
    for t in (q, k, v):
        if not (t.is_cuda and t.dtype == torch.float16 and t.stride(1) == 1):
            raise ValueError("attention operands must be fp16 device tensors with unit column stride")
    c = q.shape[1]
    dh = c // heads
    o = torch.empty(nb * d, c, dtype=torch.float16, device=q.device)
    scale = dh ** -0.5 if scale is None else scale
    _lib.check(_lib.load().lavie_sparse_causal_attention_f16(_p(q), q.stride(0), _p(k), k.stride(0), _p(v), v.stride(0),
                                                             _p(o), c, nb, frames, d, heads, dh, float(scale), _stream()),
               "lavie_sparse_causal_attention_f16")
    return o


def relpos_buckets(frames: int, num_buckets: int = 32, max_distance: int = 32) -> torch.Tensor:
    """Host-side bucket table [F, F] (query i, key j) — needs no GPU."""
    buf = (ctypes.c_int * (frames * frames))()
    _lib.check(_lib.load().lavie_relpos_buckets(frames, num_buckets, max_distance, buf), "lavie_relpos_buckets")
    return torch.tensor(list(buf), dtype=torch.int64).reshape(frames, frames)


def rotary_tables(frames: int, rot_dim: int = 32, theta: float = 10000.0, device="cuda"):
    """cos/sin [F, rot_dim/2] in fp32 (angles = frame * theta^(-2k/rot_dim))."""
    inv = theta ** (-torch.arange(0, rot_dim, 2, dtype=torch.float32) / rot_dim)
    ang = torch.arange(frames, dtype=torch.float32).reshape(-1, 1) * inv.reshape(1, -1)
    return ang.cos().to(device).contiguous(), ang.sin().to(device).contiguous()


def temporal_attention(qkv, b, frames, d, heads, bias, rot_cos, rot_sin, rot_dim=32, scale=None):
    """qkv [(b f d), 3C] (q | k | v) in (b, f, pixel) token order -> [(b f d), C]."""
    _chk16(qkv)
    _chk32(bias, *(() if rot_dim == 0 else (rot_cos, rot_sin)))      # rot_dim = 0: no rotary embedding, tables may be None
    c = qkv.shape[1] // 3
    dh = c // heads
    o = torch.empty(qkv.shape[0], c, dtype=torch.float16, device=qkv.device)
    scale = dh ** -0.5 if scale is None else scale
    _lib.check(_lib.load().lavie_temporal_attention_f16(_p(qkv), 3 * c, _p(o), c, b, frames, d, heads, dh, _p(bias),
                                                        _p(rot_cos), _p(rot_sin), rot_dim, float(scale), _stream()),
               "lavie_temporal_attention_f16")
    return o


def cfg_ddpm_step(eps2, x, noise, model_in2, guidance, coeffs, next_input_scale: float = 1.0):
    """Fused CFG + scheduler update; `coeffs` = (k_x, k_eps, c_x0, c_xt, sigma) from <scheduler>.coefficients;
    `next_input_scale` = the scheduler's scale_model_input factor of the next step (1 for DDPM / DDIM)."""
    _chk16(eps2, model_in2)
    _chk32(x, noise)
    k_x, k_e, c_x0, c_xt, sigma = coeffs
    n = x.numel()
    _lib.check(_lib.load().lavie_cfg_sampler_step(_p(eps2), _p(x), _p(noise), _p(model_in2), n, float(guidance), float(k_x),
                                                  float(k_e), float(c_x0), float(c_xt), float(sigma),
                                                  float(next_input_scale), _stream()),
               "lavie_cfg_sampler_step")


def latents_to_model_input(x, model_in2, input_scale: float = 1.0):
    _chk32(x)
    _chk16(model_in2)
    _lib.check(_lib.load().lavie_latents_to_scaled_model_input(_p(x), _p(model_in2), x.numel(), float(input_scale), _stream()))


def sampler_step(eps, x, noise, model_in, coeffs, next_input_scale: float = 1.0):
    """Scheduler update without classifier-free guidance (guidance_scale <= 1): eps / model_in are fp16 of x's size."""
    _chk16(eps, model_in)
    _chk32(x, noise)
    if eps.numel() != x.numel() or model_in.numel() != x.numel():
        raise ValueError("sampler_step: eps / model_in must have as many elements as x")
    k_x, k_e, c_x0, c_xt, sigma = coeffs
    _lib.check(_lib.load().lavie_sampler_step(_p(eps), _p(x), _p(noise), _p(model_in), x.numel(), float(k_x), float(k_e),
                                              float(c_x0), float(c_xt), float(sigma), float(next_input_scale), _stream()),
               "lavie_sampler_step")


def latents_to_model_input1(x, model_in, input_scale: float = 1.0):
    _chk32(x)
    _chk16(model_in)
    _lib.check(_lib.load().lavie_latents_to_scaled_model_input1(_p(x), _p(model_in), x.numel(), float(input_scale), _stream()))


# ------------------------------------------------------------------ engine seams (sub-module forwards)
def unet_resnet_block(net, prefix: str, x1, x2, temb, b: int, f: int, h: int, w: int):
    """ResnetBlock3D.forward of `net`'s block `prefix` on channels-last rows (x2 = skip half or None).
    temb: fp32 [b, time_embed_dim] — the output of time_embedding (the block applies SiLU + its projection)."""
    _chk16(x1, x2)
    _chk32(temb)
    handle = net.engine_handle()
    cout = dict(net.named_parameters())[prefix + ".conv2.bias"].shape[0]
    y = torch.empty(x1.shape[0], cout, dtype=torch.float16, device=x1.device)
    _lib.check(_lib.load().lavie_unet_resnet_forward(handle, prefix.encode(), _p(x1), x1.shape[1], _p(x2),
                                                     0 if x2 is None else x2.shape[1], _p(temb), _p(y), b, f, h, w,
                                                     _stream()), "lavie_unet_resnet_forward")
    return y


def unet_transformer(net, prefix: str, x, ctx, b: int, f: int, h: int, w: int):
    """Transformer3DModel.forward of `net`'s block `prefix`; x rows are updated in place and returned."""
    _chk16(x, ctx)
    handle = net.engine_handle()
    _lib.check(_lib.load().lavie_unet_transformer_forward(handle, prefix.encode(), _p(x), _p(ctx), b, f, h, w,
                                                          ctx.shape[1], _stream()), "lavie_unet_transformer_forward")
    return x
