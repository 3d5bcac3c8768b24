"""Times the fused feed-forward kernel against the two GEMM launches it replaces (level-0 shape of the bench: 81920 x 320).
Interleaved rounds in one process (guide rule 24)."""
import math
import sys

import torch

sys.path.insert(0, ".")
from lavie_amd import ops  # noqa: E402


def timeit(fn, n=20):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return 1e3 * s.elapsed_time(e) / n


def main():
    M, C = 81920, 320
    g = torch.Generator().manual_seed(0)
    x = torch.randn(M, C, generator=g).half().cuda()
    w1 = (torch.randn(8 * C, C, generator=g) / math.sqrt(C)).half().cuda()
    b1 = (torch.randn(8 * C, generator=g) * 0.1).half().cuda()
    w2 = (torch.randn(C, 4 * C, generator=g) / math.sqrt(4 * C)).half().cuda()
    b2 = torch.randn(C, generator=g).cuda()
    gamma, beta = torch.ones(C).cuda(), torch.zeros(C).cuda()
    img, b1img = ops.pack_geglu_mlp(w1, b1, w2)
    wp, bp = ops.pack_geglu(w1, b1)
    out = torch.empty_like(x)
    wide = torch.empty(M, 4 * C, dtype=torch.float16, device="cuda")

    def fused():
        ops.geglu_mlp(x, img, b1img, gamma, beta, b2, out=out)

    def unfused():
        ops.linear(x, wp, bias=bp, geglu=True, out=wide)
        ops.linear(wide, w2, bias=b2, residual=x, out=out)

    for _ in range(3):
        fused()
        unfused()
    flop = 2.0 * M * C * 12 * C
    from lavie_amd import _lib
    lib = _lib.load()
    for r in range(3):
        line = f"round {r}:"
        for v in (0, 1, 2):
            lib.lavie_debug_rowfuse_variant(v)
            fused()
            tf = timeit(fused)
            line += f"  fused[v{v}] {tf:7.1f} us ({flop / tf / 1e6:6.1f} TF/s)"
        tu = timeit(unfused)
        if r == 0:
            print("feed-forward variants: 0 = read-ahead 8 (shipped), 1 = 5, 2 = 12")
        print(line + f"   ff1 + ff2 unfused (no LN fold) {tu:7.1f} us ({flop / tu / 1e6:6.1f} TF/s)", flush=True)
    lib.lavie_debug_rowfuse_variant(0)

    # ---- temporal sub-block: fused kernel vs LN-folded-free q|k|v GEMM + temporal attention kernel + to_out GEMM (+ residual)
    B, Fr, D, heads = 2, 16, 2560, 8
    wq, wk, wv, wo = [(torch.randn(C, C, generator=g) / math.sqrt(C)).half().cuda() for _ in range(4)]
    bo = torch.randn(C, generator=g).cuda()
    relbias = torch.randn(heads, Fr, Fr, generator=g).cuda()
    inv = 10000.0 ** (-torch.arange(0, 32, 2, dtype=torch.float32) / 32)
    ang = torch.arange(Fr, dtype=torch.float32).reshape(Fr, 1) * inv.reshape(1, -1)
    rc, rs = ang.cos().cuda(), ang.sin().cuda()
    timg = ops.pack_temporal_block(wq, wk, wv, wo)
    wqkv = torch.cat([wq, wk, wv]).contiguous()
    qkv = torch.empty(M, 3 * C, dtype=torch.float16, device="cuda")
    att = torch.empty_like(x)
    scale = 40 ** -0.5

    def tfused():
        ops.temporal_block(x, timg, gamma, beta, bo, relbias, rc, rs, B, Fr, D, heads, 32, scale, out=out)

    def tunfused():
        ops.linear(x, wqkv, out=qkv)
        att.copy_(ops.temporal_attention(qkv, B, Fr, D, heads, relbias, rc, rs, 32, scale)) if False else ops.temporal_attention(qkv, B, Fr, D, heads, relbias, rc, rs, 32, scale)
        ops.linear(att, wo, bias=bo, residual=x, out=out)

    have_unfused = hasattr(ops, "temporal_attention")
    for _ in range(3):
        tfused()
        if have_unfused:
            tunfused()
    tflop = 2.0 * M * C * 4 * C
    for r in range(3):
        lib.lavie_debug_rowfuse_variant(0)
        t8 = timeit(tfused)
        lib.lavie_debug_rowfuse_variant(5)
        t4 = timeit(tfused)
        lib.lavie_debug_rowfuse_variant(0)
        line = f"temporal round {r}: fused PF8 {t8:7.1f} us ({tflop / t8 / 1e6:6.1f} TF/s)  fused PF4 {t4:7.1f} us"
        if have_unfused:
            tu = timeit(tunfused)
            line += f"   qkv GEMM + temporal kernel + to_out GEMM {tu:7.1f} us"
        print(line, flush=True)


if __name__ == "__main__" and "cross" not in sys.argv[1:]:
    main()


def cross():
    """fused text cross-attention sub-block vs the four launches it replaces (to_out GEMM, to_q GEMM with an explicit LayerNorm
    left out = the LN-folded cost, attention kernel over 77 keys, to_out GEMM)."""
    M, C, B, L, heads = 81920, 320, 2, 77, 8
    P = M // B
    g = torch.Generator().manual_seed(1)
    x = torch.randn(M, C, generator=g).half().cuda()
    att = torch.randn(M, C, generator=g).half().cuda()
    wo1, wq2, wo2 = [(torch.randn(C, C, generator=g) / math.sqrt(C)).half().cuda() for _ in range(3)]
    bo1, bo2 = torch.randn(C, generator=g).cuda(), torch.randn(C, generator=g).cuda()
    gamma, beta = torch.ones(C).cuda(), torch.zeros(C).cuda()
    kv = torch.randn(B * L, 2 * C, generator=g).half().cuda()
    img = ops.bind_cross_block(ops.pack_cross_block(wo1, wq2, wo2), kv, B, L)
    out, q, o = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
    scale = 40 ** -0.5
    k, v = kv[:, :C], kv[:, C:]

    def fused():
        ops.cross_block(att, x, img, bo1, gamma, beta, bo2, P, L, heads, scale, out=out)

    def unfused():
        ops.linear(att, wo1, bias=bo1, residual=x, out=out)
        ops.linear(out, wq2, out=q)
        a = ops.attention(q, k, v, B * 16, P // 16, L, heads, kv_batch_div=16, scale=scale)
        ops.linear(a, wo2, bias=bo2, residual=out, out=o)

    from lavie_amd import _lib
    lib = _lib.load()
    for _ in range(3):
        fused()
        unfused()
    flop = 2.0 * M * C * 3 * C + 4.0 * M * L * C
    for r in range(3):
        lib.lavie_debug_rowfuse_variant(0)
        t8 = timeit(fused)
        lib.lavie_debug_rowfuse_variant(5)
        t4 = timeit(fused)
        lib.lavie_debug_rowfuse_variant(0)
        tu = timeit(unfused)
        print(f"cross round {r}: fused PF8 {t8:7.1f} us ({flop / t8 / 1e6:6.1f} TF/s)  fused PF4 {t4:7.1f} us   to_out + to_q + attention + to_out "
              f"{tu:7.1f} us", flush=True)


if __name__ == "__main__" and "cross" in sys.argv[1:]:
    cross()
