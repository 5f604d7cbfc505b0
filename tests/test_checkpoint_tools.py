"""checkpoint ingestion — LoRA merge and key surgery of scripts/sampling/util.py:45-272 —
against the output of the reference's own convert_load_lora (tests/golden/lora_merge.npz)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def _lora_fixture():
    """Same construction as tests/golden/make_golden.py: lora_fixture (kept in sync by the test below failing)."""
    from ccedit_amd.utils.synth import synth_tensor
    mods = []
    for i in range(12):
        mods += [f"lora_te_text_model_encoder_layers_{i}_self_attn_{n}_proj" for n in ("q", "k", "v", "out")]
        mods += [f"lora_te_text_model_encoder_layers_{i}_mlp_{n}" for n in ("fc1", "fc2")]
    tails = ["proj_in", "proj_out"] + [f"transformer_blocks_0_{a}_to_{n}" for a in ("attn1", "attn2") for n in ("q", "k", "v", "out_0")] \
        + ["transformer_blocks_0_ff_net_0_proj", "transformer_blocks_0_ff_net_2"]
    blocks = [f"down_blocks_{i}_attentions_{j}" for i in range(3) for j in range(2)] + ["mid_block_attentions_0"] \
        + [f"up_blocks_{i}_attentions_{j}" for i in (1, 2, 3) for j in range(3)]
    mods += [f"lora_unet_{b}_{t}" for b in blocks for t in tails]
    lora = {}
    for m in mods:
        conv = m.endswith(("proj_in", "proj_out")) and "unet" in m
        lora[m + ".lora_down.weight"] = synth_tensor(m + ".lora_down.weight", (2, 8, 1, 1) if conv else (2, 8))
        lora[m + ".lora_up.weight"] = synth_tensor(m + ".lora_up.weight", (8, 2, 1, 1) if conv else (8, 2))
        lora[m + ".alpha"] = torch.tensor(2.0)
    return mods, lora


def test_lora_merge_matches_reference_function(golden_dir):
    from ccedit_amd.utils.synth import synth_tensor
    from scripts.sampling.util import convert_load_lora, lora_target_key
    z = np.load(os.path.join(golden_dir, "lora_merge.npz"))
    mods, lora = _lora_fixture()
    targets = {lora_target_key(m + ".lora_up.weight") for m in mods}
    assert targets == set(z.files), sorted(targets ^ set(z.files))[:4]          # every module lands on the reference's key
    base = {k: synth_tensor(k, z[k].shape) for k in z.files}
    merged = convert_load_lora(sd_state_dict=base, state_dict=lora, alpha=0.8)
    for k in z.files:
        assert merged[k].shape == z[k].shape
        assert np.array_equal(merged[k].numpy(), z[k]), k                       # same fp32 operations, same order: bit-exact


def test_lora_targets_exist_in_the_engine_state_dict(golden_dir):
    """The merged keys must be real parameters of the full-size engine (UNet attention weights, CLIP text layers)."""
    import json
    from scripts.sampling.util import lora_target_key
    keys = set(json.load(open(os.path.join(golden_dir, "keys_tv2v.json")))) | set(json.load(open(os.path.join(golden_dir, "keys_clip_text.json"))))
    mods, _ = _lora_fixture()
    missing = [m for m in mods if lora_target_key(m + ".lora_down.weight") not in keys]
    assert not missing, missing[:5]
    with pytest.raises(ValueError):
        lora_target_key("lora_unet_down_blocks_0_resnets_0_conv1.lora_down.weight")


def test_checkpoint_key_surgery_and_load(tmp_path):
    """model_load_ckpt: VAE copies nested under conditioner embedders, cond_stage_model renaming for a new base
    model, `state_dict` wrapper, embedded LoRA tensors merged at alpha 0.8 before loading."""
    from scripts.sampling.util import model_load_ckpt, remap_checkpoint_keys
    sd = {"conditioner.embedders.2.first_stage_model.decoder.conv_in.weight": torch.ones(1),
          "cond_stage_model.transformer.text_model.final_layer_norm.weight": torch.ones(1),
          "model.diffusion_model.out.2.weight": torch.ones(1)}
    r = remap_checkpoint_keys(sd, newbasemodel=True)
    assert set(r) == {"first_stage_model.decoder.conv_in.weight", "conditioner.embedders.0.transformer.text_model.final_layer_norm.weight",
                      "model.diffusion_model.out.2.weight"}
    assert "cond_stage_model.transformer.text_model.final_layer_norm.weight" in remap_checkpoint_keys(sd)

    class Tiny(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.model = torch.nn.Module()
            self.model.diffusion_model = torch.nn.Module()
            self.model.diffusion_model.middle_block = torch.nn.ModuleList([torch.nn.Identity(), torch.nn.Module()])
            self.model.diffusion_model.middle_block[1].proj_in = torch.nn.Conv2d(8, 8, 1, bias=False)

    m = Tiny()
    w0 = torch.randn(8, 8, 1, 1)
    up, down = torch.randn(8, 2, 1, 1), torch.randn(2, 8, 1, 1)
    path = os.path.join(tmp_path, "toy.ckpt")
    torch.save({"state_dict": {"model.diffusion_model.middle_block.1.proj_in.weight": w0.clone(),
                               "lora_unet_mid_block_attentions_0_proj_in.lora_up.weight": up,
                               "lora_unet_mid_block_attentions_0_proj_in.lora_down.weight": down,
                               "lora_unet_mid_block_attentions_0_proj_in.alpha": torch.tensor(1.0)}}, path)
    model_load_ckpt(m, path)
    want = w0 + 0.8 * (up[:, :, 0, 0] @ down[:, :, 0, 0])[:, :, None, None]
    assert torch.allclose(m.model.diffusion_model.middle_block[1].proj_in.weight, want, atol=1e-6)
