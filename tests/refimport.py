"""Loads the reference's base/models package under tests/refshim (TEST INFRASTRUCTURE).

Only used in the build container: /root/reference does not exist on the GPU box, so
every caller must skip when `available()` is False."""
import importlib.util
import os
import sys

REF = "/root/reference/base/models"
_SHIM = os.path.join(os.path.dirname(os.path.abspath(__file__)), "refshim")


def available():
    return os.path.isfile(os.path.join(REF, "unet.py"))


def load(stage="base"):
    """Returns the reference modules (resnet, attention, unet_blocks, unet) of `<stage>/models` as a namespace;
    stage is "base" (the T2V denoiser) or "interpolation" (the 16 -> 61 frame model)."""
    pkgname = "refmodels" if stage == "base" else f"refmodels_{stage}"
    if f"{pkgname}.unet" in sys.modules:
        return sys.modules[pkgname]
    if _SHIM not in sys.path:
        sys.path.insert(0, _SHIM)
    root = REF if stage == "base" else f"/root/reference/{stage}/models"
    pkg = importlib.util.module_from_spec(importlib.machinery.ModuleSpec(pkgname, None, is_package=True))
    pkg.__path__ = [root]
    sys.modules[pkgname] = pkg
    for name in ("resnet", "attention", "unet_blocks", "unet"):
        spec = importlib.util.spec_from_file_location(f"{pkgname}.{name}", os.path.join(root, f"{name}.py"))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[f"{pkgname}.{name}"] = mod
        spec.loader.exec_module(mod)
        setattr(pkg, name, mod)
    return pkg


def load_vsr_ddim():
    """The reference's vendored DDIM scheduler class (vsr/diffusion/scheduling_ddim.py) under the shim."""
    if "ref_vsr_ddim" in sys.modules:
        return sys.modules["ref_vsr_ddim"].DDIMScheduler
    if _SHIM not in sys.path:
        sys.path.insert(0, _SHIM)
    spec = importlib.util.spec_from_file_location("ref_vsr_ddim", "/root/reference/vsr/diffusion/scheduling_ddim.py")
    mod = importlib.util.module_from_spec(spec)
    sys.modules["ref_vsr_ddim"] = mod
    spec.loader.exec_module(mod)
    return mod.DDIMScheduler


def load_interp_diffusion():
    """The reference's in-tree OpenAI-style diffusion package (interpolation/diffusion: numpy + torch only)."""
    name = "ref_interp_diffusion"
    if name in sys.modules:
        return sys.modules[name]
    root = "/root/reference/interpolation/diffusion"
    spec = importlib.util.spec_from_file_location(name, os.path.join(root, "__init__.py"), submodule_search_locations=[root])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def load_vsr_blocks():
    """vsr/models/{resnet,attention,unet_blocks,diffusers_attention,temporal_module,unet}.py under the shim."""
    pkgname, root = "refmodels_vsr", "/root/reference/vsr/models"
    if f"{pkgname}.attention" in sys.modules:
        return sys.modules[pkgname]
    if _SHIM not in sys.path:
        sys.path.insert(0, _SHIM)
    pkg = importlib.util.module_from_spec(importlib.machinery.ModuleSpec(pkgname, None, is_package=True))
    pkg.__path__ = [root]
    sys.modules[pkgname] = pkg
    for name in ("resnet", "attention", "unet_blocks", "diffusers_attention", "temporal_module", "unet"):
        spec = importlib.util.spec_from_file_location(f"{pkgname}.{name}", os.path.join(root, f"{name}.py"))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[f"{pkgname}.{name}"] = mod
        spec.loader.exec_module(mod)
        setattr(pkg, name, mod)
    return pkg


VSR_TEMPORAL_MODULE_CONFIG = dict(       # vsr/configs/unet_3d_config.json "temporal_module_config" (values, not code)
    num_attention_layers=1, attention_block_types=["", ""], cross_frame_attention_mode="0_i-1_i", temporal_shift_fold_div=2,
    temporal_shift_direction="right", use_dcn_warpping=False, use_deformable_conv=True, attention_dim_div=2)
