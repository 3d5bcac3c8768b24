import sys, torch
sys.path.insert(0, '/root/repo')
from lavie_amd import _lib, ops
from tools.bench_attn_sc import timeit
lib = _lib.load()
C, D, nb, heads = 320, 2560, 32, 8
qkv = (torch.randn(nb * D, 3 * C, device="cuda") * 0.5).half()
q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
for name, mode in (("full", 0), ("no-softmax-VALU", 0x12), ("no-MFMA", 0x22), ("staging+barriers only", 0x32), ("QT=1", 1), ("QT=4", 4), ("VALU row sums", 0x40)):
    lib.lavie_debug_attention_qt(mode)
    print(f"{name:24s} {timeit(lambda: ops.attention(q, k, v, nb, D, D, heads), iters=10):8.1f} us", flush=True)
