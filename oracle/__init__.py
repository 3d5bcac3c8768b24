"""oracle/ — CPU restatement of the LaVie base T2V denoising path.  TEST INFRASTRUCTURE.

This package is the *checker*, never the product: only `tests/`, `__graft_entry__.smoke()`
and the `cpu_baseline` leg of `bench.py` may import it.  Nothing under `lavie_amd/` does
(tests/test_layout.py enforces that), and the product path raises when the HIP library is
missing instead of falling back to anything in here.

Parity status:
  * the wiring — block order, reshapes, the two GroupNorm reduction domains, attention
    math, skip concatenation, rel-pos buckets — is PINNED against the reference's own
    `base/models/{resnet,attention,unet_blocks,unet}.py`, executed in the build container
    by `tests/test_oracle_vs_reference.py` and frozen into `tests/golden/*.pt` by
    `tests/golden/make_golden.py`;
  * the arithmetic that lives in third-party packages absent from the image
    (diffusers 0.16.0: GEGLU feed-forward, Timesteps/TimestepEmbedding, DDPMScheduler;
    rotary_embedding_torch, unpinned) is restated from the published algorithms and is
    PARITY-UNPINNED — the reference holds no tests or fixtures for it.
"""
