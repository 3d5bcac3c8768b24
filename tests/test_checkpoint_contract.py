"""CPU: the checkpoint entry points real weights depend on.

`from_pretrained_2d` of the base mirror (base/models/unet.py:540-588) and of the interpolation mirror
(interpolation/models/unet.py:477-555), and `load_checkpoint` = `find_model` (base/download.py:10-18), round-tripped
through synthetic files in tmp_path: an SD-style `unet/config.json` + `diffusion_pytorch_model.bin` holding only the
2-D tensors, and a `{"ema": state_dict}` LaVie checkpoint.  Everything is loaded with weights_only=True."""
import json
import os

import pytest
import torch

from lavie_amd import weights
from lavie_amd.interpolation import UNet3DConditionModel as InterpUNet
from lavie_amd.unet import UNet3DConditionModel

# four levels (from_pretrained_2d always builds the 4-level block list, unet.py:550-561) at toy widths
SD_CONFIG = dict(sample_size=8, in_channels=4, out_channels=4, block_out_channels=[64, 64, 128, 128], layers_per_block=1,
                 norm_num_groups=32, norm_eps=1e-5, cross_attention_dim=64, attention_head_dim=2,
                 _class_name="UNet2DConditionModel", act_fn="silu")       # extra keys are ignored like from_config does


def write_sd_folder(root, state):
    unet = os.path.join(root, "unet")
    os.makedirs(unet)
    with open(os.path.join(unet, "config.json"), "w") as fh:
        json.dump(SD_CONFIG, fh)
    torch.save(state, os.path.join(unet, "diffusion_pytorch_model.bin"))
    return root


def fresh(cls, seed, **kw):
    torch.manual_seed(seed)
    cfg = {k: (tuple(v) if isinstance(v, list) else v) for k, v in SD_CONFIG.items() if not k.startswith("_") and k != "act_fn"}
    cfg.update(kw)
    return cls(**cfg)


def two_d_state(model, seed):
    """What an SD-1.x UNet file holds for this architecture: every tensor but the temporal ('_temp') ones."""
    g = torch.Generator().manual_seed(seed)
    return {k: torch.randn(v.shape, generator=g) for k, v in model.state_dict().items() if "_temp" not in k}


def test_base_from_pretrained_2d_keeps_temporal_init(tmp_path):
    probe = fresh(UNet3DConditionModel, 5)
    state2d = two_d_state(probe, 1)
    temp_keys = [k for k in probe.state_dict() if "_temp" in k]
    assert temp_keys and not any("_temp" in k for k in state2d)
    root = write_sd_folder(str(tmp_path), state2d)
    torch.manual_seed(5)                                       # the constructor's own initialisation, reproducibly
    model = UNet3DConditionModel.from_pretrained_2d(root, subfolder="unet")
    sd = model.state_dict()
    assert set(sd) == set(probe.state_dict())
    for k, v in state2d.items():                               # 2-D tensors come from the file
        assert torch.equal(sd[k], v), k
    for k in temp_keys:                                        # temporal tensors keep the constructor's values (:575-578)
        assert torch.equal(sd[k], probe.state_dict()[k]), k
    assert any(float(sd[k].abs().sum()) == 0 for k in temp_keys if k.endswith("attn_temp.to_out.0.weight"))   # attention.py:475
    assert model.config.in_channels == 4 and model.config.sample_size == 8
    with pytest.raises(RuntimeError):
        UNet3DConditionModel.from_pretrained_2d(str(tmp_path / "missing"), subfolder="unet")


@pytest.mark.parametrize("wrap", [True, False])
def test_load_checkpoint_unwraps_ema_and_loads_on_top(tmp_path, wrap):
    model = fresh(UNet3DConditionModel, 7)
    g = torch.Generator().manual_seed(2)
    ckpt = {k: torch.randn(v.shape, generator=g) for k, v in model.state_dict().items()}
    path = str(tmp_path / "lavie_base.pt")
    torch.save({"ema": ckpt, "opt": {"lr": torch.tensor(1e-4)}} if wrap else ckpt, path)
    got = weights.load_checkpoint(path)                        # download.py:14-17
    assert set(got) == set(ckpt) and all(torch.equal(got[k], ckpt[k]) for k in ckpt)
    missing, unexpected = model.load_state_dict(got)           # strict, as sample.py:28 does
    assert not missing and not unexpected
    assert all(torch.equal(model.state_dict()[k], ckpt[k]) for k in ckpt)


def test_interpolation_from_pretrained_2d_concat_widens_conv_in(tmp_path):
    probe4 = fresh(InterpUNet, 11, use_first_frame=True)
    state2d = two_d_state(probe4, 3)
    root = write_sd_folder(str(tmp_path), state2d)
    for copy_no_mask, cin in ((True, 8), (False, 9)):
        probe = fresh(InterpUNet, 11, use_first_frame=True, in_channels=cin)
        torch.manual_seed(11)
        model = InterpUNet.from_pretrained_2d(root, subfolder="unet", use_concat=True, copy_no_mask=copy_no_mask)
        sd = model.state_dict()
        assert model.cfg.sparse_causal_attn1 and model.config.in_channels == cin
        w = sd["conv_in.weight"]
        assert w.shape[1] == cin
        assert torch.equal(w[:, :4], state2d["conv_in.weight"]) and float(w[:, 4:].abs().sum()) == 0     # :524-531
        assert torch.equal(sd["conv_in.bias"], state2d["conv_in.bias"])
        for k, v in probe.state_dict().items():                # :537-543 — every other tensor keeps the constructor's value
            if "conv_in" not in k:
                assert torch.equal(sd[k], v), k


def test_interpolation_from_pretrained_2d_plain_path(tmp_path):
    probe = fresh(InterpUNet, 13, use_first_frame=True)
    state2d = two_d_state(probe, 4)
    root = write_sd_folder(str(tmp_path), state2d)
    torch.manual_seed(13)
    model = InterpUNet.from_pretrained_2d(root, subfolder="unet")          # use_concat False: :549-554
    sd = model.state_dict()
    for k, v in state2d.items():
        assert torch.equal(sd[k], v), k
    for k, v in probe.state_dict().items():
        if "_temp." in k:
            assert torch.equal(sd[k], v), k
    assert not any("rotary_emb" in k or "time_rel_pos_bias" in k for k in sd)   # plain temporal attention: no such tensors


def test_generator_lists_follow_randn_tensor_contract():
    """`generator=[g0, g1]` (one per latent) draws each latent from its own generator (pipeline_videogen.py:499-504)."""
    from lavie_amd.scheduling_ddpm import randn_tensor
    a = randn_tensor((2, 3, 4), generator=[torch.Generator().manual_seed(1), torch.Generator().manual_seed(2)])
    assert torch.equal(a[0], torch.randn(3, 4, generator=torch.Generator().manual_seed(1)))
    assert torch.equal(a[1], torch.randn(3, 4, generator=torch.Generator().manual_seed(2)))
    with pytest.raises(ValueError):
        randn_tensor((2, 3), generator=[torch.Generator()])
