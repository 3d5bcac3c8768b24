"""INTEGRATION.md §B is executable documentation: the ctypes stub a reference maintainer would add.
These tests extract that code block and run it verbatim, so the documented struct layout can never
drift from include/lavie_hip.h again (round-1 finding: the stub carried the ABI-v1 layout)."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def stub_source():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sect = text[text.index("## B. Bind the C ABI directly"):]
    m = re.search(r"```python\n(.*?)```", sect, re.S)
    assert m, "INTEGRATION.md §B has no python code block"
    return m.group(1)


def run_stub():
    from lavie_amd import _lib
    os.environ["LAVIE_HIP_LIB"] = _lib.LIB_PATH
    try:
        ns = {}
        exec(compile(stub_source(), "INTEGRATION.md#B", "exec"), ns)
    finally:
        os.environ.pop("LAVIE_HIP_LIB", None)
    return ns


def test_doc_stub_struct_matches_binding_and_library():
    """The documented `_Cfg` has the layout of lavie_unet_config: same size as the product's ctypes mirror and as the
    library reports, same field names in the same order; the stub's own ABI / size assertions pass."""
    from lavie_amd import _lib
    ns = run_stub()
    cfg = ns["_Cfg"]
    assert ctypes.sizeof(cfg) == ctypes.sizeof(_lib.UNetConfigC) == _lib.load().lavie_unet_config_size()
    assert [(n, ctypes.sizeof(t)) for n, t in cfg._fields_] == [(n, ctypes.sizeof(t)) for n, t in _lib.UNetConfigC._fields_]
    # and the header declares the same member names in the same order
    hdr = open(os.path.join(ROOT, "include", "lavie_hip.h")).read()
    body = hdr[hdr.index("typedef struct lavie_unet_config {"):hdr.index("} lavie_unet_config;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in re.findall(r"(?:int|float)\s+([^;]+);", body):
        names += [re.sub(r"\[.*\]", "", v).strip() for v in decl.split(",")]
    assert names == [n for n, _ in cfg._fields_]


def test_short_config_struct_is_rejected_not_overread():
    """A binding compiled against an older, shorter lavie_unet_config gets an error from lavie_unet_create."""
    from lavie_amd import _lib
    lib = _lib.load()
    c = _lib.UNetConfigC()
    c.struct_size = 27 * 4                       # the ABI-v1 layout the round-1 document described
    c.in_channels = c.out_channels = 4
    c.num_levels = 4
    h = ctypes.c_void_p()
    rc = lib.lavie_unet_create(ctypes.byref(c), ctypes.byref(h))
    assert rc != 0 and not h.value
    assert b"struct_size" in lib.lavie_last_error()


@pytest.mark.gpu
def test_doc_stub_runs_and_matches_reference_fixture():
    """HipUNet from the document, fed a state dict under the reference's key names, reproduces the output the imported
    reference UNet gave for tests/golden/unet_full_8x8.pt (909 M parameters)."""
    import golden_util as G
    from gpu_util import TOL_UNET, rel_l2
    from lavie_amd import spec
    ns = run_stub()
    sd = {k: v.to("cuda", torch.float16) for k, v in G.synth16(spec.param_shapes(), 0).items()}

    class RefUNet:                                # what the stub needs from the reference module: state_dict()
        def state_dict(self):
            return sd

    net = ns["HipUNet"](RefUNet())
    fx = G.load("unet_full_8x8.pt")
    for t, ref in fx["y"].items():
        got = net(fx["x"].cuda(), int(t), fx["ctx"].cuda())
        assert rel_l2(got, ref) < TOL_UNET, t
    ns["_lib"].lavie_unet_destroy(net.h)
