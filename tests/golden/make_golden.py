#!/usr/bin/env python3
"""Record golden vectors from the REFERENCE (RuoyuFeng/CCEdit, /root/reference) itself.

Run in the authoring container only:   python tests/golden/make_golden.py
It imports the reference's own hot-path modules (via _refshim.py), fills them with the name-keyed
synthetic weights of ccedit_amd/utils/synth.py, runs them on CPU in fp32 and writes small fixtures
(inputs + expected outputs, never reference source) into tests/golden/:

  sigmas.npz          sampler sigma tables N in {5,30,50} + the 1000-entry DiscreteDenoiser table
  keys_tv2v.json      state-dict key -> shape of the full-size TV2V network + VAE (meta device)
  net_g160.npz        one network evaluation at the reduced "G160" config: inputs, eps output,
                      digests of every block output of ControlNet2D and the 3D UNet, control digests
  sampler_g160.npz    a 5-step DPMPP2SAncestral + CFG 7.5 trajectory with injected noise:
                      timestep-index trace, per-step latent digests, final latent
  vae_g32.npz         AutoencoderKL decode at a reduced ddconfig: latent in, frames digest out
  net_tvi2v_g160.npz  one TVI2V network evaluation (controlnet_img on cond_feat + SpatialTransformer3DCA
                      anchor cross-frame attention) + keys_tvi2v_g160.json (state-dict keys/shapes)

  vae_enc_g32.npz     AutoencoderKLInferenceWrapper.encode at the same reduced ddconfig: 5-D frames and a 4-D reference
                      image in, moments digests + the posterior samples out (the CPU global-generator noise the
                      reference drew is recorded); Img2ImgDiscretizationWrapper sigma tables (SDEdit)

  clip_text.npz       HF CLIPTextModel (the network inside FrozenCLIPEmbedder; transformers is a third-party dependency
                      of the reference, not under /root/reference) at the ViT-L/14 text configuration with name-keyed
                      synthetic weights: token ids in, last_hidden_state out

A digest of a tensor = (shape, mean, std, abs-max, 256 evenly spaced samples) — enough to pin a
restatement while keeping every fixture well under 1 MB.
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))

import _refshim  # noqa: E402
from ccedit_amd.utils.synth import fill_module_  # noqa: E402

torch.set_grad_enabled(False)

NSAMP = 256


def digest(t: torch.Tensor):
    t = t.detach().float().contiguous()
    flat = t.reshape(-1)
    idx = torch.linspace(0, flat.numel() - 1, min(NSAMP, flat.numel())).long()
    return dict(shape=np.array(t.shape, dtype=np.int64),
                stats=np.array([flat.mean().item(), flat.std().item(), flat.abs().max().item()], dtype=np.float64),
                samp=flat[idx].numpy())


def put(out: dict, name: str, t: torch.Tensor):
    d = digest(t)
    out[name + "|shape"] = d["shape"]
    out[name + "|stats"] = d["stats"]
    out[name + "|samp"] = d["samp"]


# ------------------------------------------------------------------------------------------
G160 = dict(model_channels=160, num_heads=4, context_dim=128)      # => head dims 40 / 80 / 160 / 160
G160_SHAPE = dict(B=1, T=3, H=16, W=24, L=77)                       # latent 16x24 (non-square), 3 keyframes


def net_params(mc, heads, ctx):
    common = dict(use_checkpoint=False, in_channels=4, model_channels=mc, attention_resolutions=[4, 2, 1],
                  num_res_blocks=2, channel_mult=[1, 2, 4, 4], num_heads=heads, use_spatial_transformer=True,
                  transformer_depth=1, context_dim=ctx, legacy=False)
    cn = dict(common, hint_channels=3, control_scales=1.0)
    return dict(common, out_channels=4, disable_temporal_text_ca=True,
                controlnet_config=dict(target="sgm.modules.diffusionmodules.controlmodel.ControlNet2D", params=cn))


def tvi2v_params(mc, heads, ctx):
    """network_config.params of keyframe_ref_cp_no2ndca_add_cfca_depthzoe.yaml:32-90 at reduced width."""
    p = net_params(mc, heads, ctx)
    cn = dict(p["controlnet_config"]["params"])
    p.update(enable_attention3d_crossframe=True, ST3DCA_ca_type="center_self",
             controlnet_img_config=dict(target="sgm.modules.diffusionmodules.controlmodel.ControlNet2D",
                                        params=dict(cn, no_add_x=True, set_input_hint_block_as_identity=True,
                                                    disable_text_ca=True)))
    return p


def build_ref_network(mc, heads, ctx, device="cpu", tvi2v=False):
    cm = _refshim.ref("sgm.modules.diffusionmodules.controlmodel")
    wr = _refshim.ref("sgm.modules.diffusionmodules.wrappers")
    with torch.device(device):
        net = cm.ControlledUNetModel3DTV2V(**(tvi2v_params if tvi2v else net_params)(mc, heads, ctx))
    return wr.OpenAIWrapperControlLDM3DTV2V(net)


def synth_inputs(seed, B, T, H, W, L, ctx):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, 4, T, H, W, generator=g)
    cross_c = torch.randn(B, L, ctx, generator=g)
    cross_uc = torch.randn(B, L, ctx, generator=g)
    # smooth-ish depth-like hint in [-1, 1], one map per frame replicated over 3 channels
    low = torch.rand(B, 1, T, H, W, generator=g)
    hint = torch.nn.functional.interpolate(low, scale_factor=(1, 8, 8), mode="trilinear", align_corners=False)
    hint = (hint * 2 - 1).repeat(1, 3, 1, 1, 1).contiguous()
    return x, cross_c, cross_uc, hint


def gen_sigmas():
    dz = _refshim.ref("sgm.modules.diffusionmodules.discretizer")
    dn = _refshim.ref("sgm.modules.diffusionmodules.denoiser")
    out = {}
    disc = dz.LegacyDDPMDiscretization()
    for n in (5, 30, 50):
        out[f"sampler_{n}"] = disc(n, device="cpu").numpy()
    den = dn.DiscreteDenoiser(
        weighting_config=dict(target="sgm.modules.diffusionmodules.denoiser_weighting.EpsWeighting"),
        scaling_config=dict(target="sgm.modules.diffusionmodules.denoiser_scaling.EpsScaling"),
        num_idx=1000,
        discretization_config=dict(target="sgm.modules.diffusionmodules.discretizer.LegacyDDPMDiscretization"))
    out["denoiser_1000"] = den.sigmas.numpy()
    # sigma -> idx quantisation probes
    probe = torch.tensor([14.6146, 5.0878, 2.2765, 1.1606, 0.5693, 0.03, 100.0, 0.7], dtype=torch.float32)
    out["probe_sigma"] = probe.numpy()
    out["probe_idx"] = den.sigma_to_idx(probe).numpy()
    ut = _refshim.ref("sgm.modules.diffusionmodules.util")
    tt = torch.tensor([999, 799, 599, 1, 0], dtype=torch.int64)
    out["temb_t"] = tt.numpy()
    out["temb_320"] = ut.timestep_embedding(tt, 320).numpy()
    np.savez_compressed(os.path.join(HERE, "sigmas.npz"), **out)
    print("sigmas.npz", {k: v.shape for k, v in out.items()})
    return den


def gen_keys():
    ae = _refshim.ref("sgm.models.autoencoder")
    wrapper = build_ref_network(320, 8, 768, device="meta")
    keys = {"model." + k: list(v.shape) for k, v in wrapper.state_dict().items()}
    with torch.device("meta"):
        vae = ae.AutoencoderKLInferenceWrapper(
            embed_dim=4, monitor="val/rec_loss", lossconfig=dict(target="torch.nn.Identity"),
            ddconfig=dict(double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=128,
                          ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[], dropout=0.0))
    keys.update({"first_stage_model." + k: list(v.shape) for k, v in vae.state_dict().items()})
    with open(os.path.join(HERE, "keys_tv2v.json"), "w") as f:
        json.dump(keys, f, indent=0, sort_keys=True)
    n = sum(int(np.prod(s)) for k, s in keys.items() if k.startswith("model."))
    print("keys_tv2v.json", len(keys), "tensors; network params", n / 1e6, "M")


def gen_net():
    mc, heads, ctx = G160["model_channels"], G160["num_heads"], G160["context_dim"]
    S = G160_SHAPE
    wrapper = build_ref_network(mc, heads, ctx).eval()
    fill_module_(wrapper, prefix="model.")
    x, cross_c, cross_uc, hint = synth_inputs(1234, S["B"], S["T"], S["H"], S["W"], S["L"], ctx)
    # CFG-doubled batch, uc first (guiders.py:63)
    x2 = torch.cat([x, x])
    ctx2 = torch.cat([cross_uc, cross_c])
    hint2 = torch.cat([hint, hint])
    t = torch.tensor([601, 601], dtype=torch.int64)

    out = dict(x=x.numpy(), cross_c=cross_c.numpy(), cross_uc=cross_uc.numpy(), hint1=hint[:, :1].numpy(), t=t.numpy())  # hint = hint1 x3 channels
    traces = {}

    def hook(name):
        def fn(mod, inp, o):
            traces[name] = o.detach().clone()
        return fn

    net = wrapper.diffusion_model
    hs = []
    for i, m in enumerate(net.controlnet.input_blocks):
        hs.append(m.register_forward_hook(hook(f"model.diffusion_model.controlnet.input_blocks.{i}")))
    hs.append(net.controlnet.middle_block.register_forward_hook(hook("model.diffusion_model.controlnet.middle_block")))
    for i, m in enumerate(net.input_blocks):
        if i:   # block 0 goes through spatial_temporal_forward: hook sees only the spatial part
            hs.append(m.register_forward_hook(hook(f"model.diffusion_model.input_blocks.{i}")))
    for i, m in enumerate(net.output_blocks):
        hs.append(m.register_forward_hook(hook(f"model.diffusion_model.output_blocks.{i}")))

    # capture the control list (it is consumed by pop())
    orig_cn = net.controlnet.forward
    ctrl = {}

    def cn_forward(*a, **k):
        r = orig_cn(*a, **k)
        ctrl["list"] = [c.clone() for c in r]
        return r
    net.controlnet.forward = cn_forward

    eps = wrapper(x2, t, dict(crossattn=ctx2, control_hint=hint2))
    for h in hs:
        h.remove()
    out["eps"] = eps.numpy()
    # NOTE: the controlnet's input_blocks.0 hook fires BEFORE `h += guided_hint` is applied in place...
    # the in-place add mutates the hooked tensor object, and we cloned at hook time => pre-add value.
    for k, v in traces.items():
        put(out, "trace:" + k, v)
    for i, c in enumerate(ctrl["list"]):
        put(out, f"control:{i}", c)
    np.savez_compressed(os.path.join(HERE, "net_g160.npz"), **out)
    print("net_g160.npz eps", tuple(eps.shape), "rms", eps.pow(2).mean().sqrt().item(),
          "size", os.path.getsize(os.path.join(HERE, "net_g160.npz")))
    return wrapper


def gen_net_tvi2v():
    """One TVI2V network evaluation (controlnet_img + anchor cross-frame attention) at the G160 width."""
    mc, heads, ctx = G160["model_channels"], G160["num_heads"], G160["context_dim"]
    S = G160_SHAPE
    wrapper = build_ref_network(mc, heads, ctx, tvi2v=True).eval()
    fill_module_(wrapper, prefix="model.")
    x, cross_c, cross_uc, hint = synth_inputs(2468, S["B"], S["T"], S["H"], S["W"], S["L"], ctx)
    g = torch.Generator().manual_seed(97)
    cond_feat = torch.randn(S["B"], 4, S["H"], S["W"], generator=g) * 0.18215
    x2 = torch.cat([x, x])
    t = torch.tensor([401, 401], dtype=torch.int64)
    c = dict(crossattn=torch.cat([cross_uc, cross_c]), control_hint=torch.cat([hint, hint]),
             cond_feat=torch.cat([cond_feat, cond_feat]))
    eps = wrapper(x2, t, c)
    keys = {"model." + k: list(v.shape) for k, v in wrapper.state_dict().items()}
    with open(os.path.join(HERE, "keys_tvi2v_g160.json"), "w") as f:
        json.dump(keys, f, indent=0, sort_keys=True)
    out = dict(x=x.numpy(), cross_c=cross_c.numpy(), cross_uc=cross_uc.numpy(), hint1=hint[:, :1].numpy(),
               cond_feat=cond_feat.numpy(), t=t.numpy(), eps=eps.numpy())
    np.savez_compressed(os.path.join(HERE, "net_tvi2v_g160.npz"), **out)
    print("net_tvi2v_g160.npz eps", tuple(eps.shape), "rms", eps.pow(2).mean().sqrt().item(), "keys", len(keys),
          "size", os.path.getsize(os.path.join(HERE, "net_tvi2v_g160.npz")))


def gen_sampler(wrapper, den):
    sp = _refshim.ref("sgm.modules.diffusionmodules.sampling")
    mc, heads, ctx = G160["model_channels"], G160["num_heads"], G160["context_dim"]
    S = G160_SHAPE
    x, cross_c, cross_uc, hint = synth_inputs(4321, S["B"], S["T"], S["H"], S["W"], S["L"], ctx)
    g = torch.Generator().manual_seed(99)
    N = 5
    noises = [torch.randn(x.shape, generator=g) for _ in range(N)]
    sampler = sp.DPMPP2SAncestralSampler(
        eta=1.0, s_noise=1.0, num_steps=N, device="cpu", verbose=False,
        discretization_config=dict(target="sgm.modules.diffusionmodules.discretizer.LegacyDDPMDiscretization"),
        guider_config=dict(target="sgm.modules.diffusionmodules.guiders.VanillaCFGTV2V", params=dict(scale=7.5)))
    it = iter(noises)
    sampler.noise_sampler = lambda xx: next(it)
    idx_trace, sig_trace = [], []

    def network(xx, tt, cc):
        idx_trace.append(tt.clone())
        return wrapper(xx, tt, cc)

    def denoiser(inp, sigma, c):      # closure of sampling_tv2v.py:366-369
        sig_trace.append(sigma.clone())
        return den(network, inp, sigma, c)

    xs = []
    orig_step = sampler.sampler_step

    def step(*a, **k):
        r = orig_step(*a, **k)
        xs.append(r.clone())
        return r
    sampler.sampler_step = step
    c = dict(crossattn=cross_c, control_hint=hint)
    uc = dict(crossattn=cross_uc, control_hint=hint.clone())
    final = sampler(denoiser, x.clone(), c, uc=uc)
    out = dict(x=x.numpy(), cross_c=cross_c.numpy(), cross_uc=cross_uc.numpy(), hint1=hint[:, :1].numpy(),
               noises=torch.stack(noises).numpy(), final=final.numpy(),
               idx_trace=torch.stack(idx_trace).numpy(), sigma_trace=torch.stack(sig_trace).numpy())
    for i, xi in enumerate(xs):
        put(out, f"step:{i}", xi)
    np.savez_compressed(os.path.join(HERE, "sampler_g160.npz"), **out)
    print("sampler_g160.npz idx trace", torch.stack(idx_trace)[:, 0].tolist(), "final rms",
          final.pow(2).mean().sqrt().item(), "size", os.path.getsize(os.path.join(HERE, "sampler_g160.npz")))


def gen_vae():
    ae = _refshim.ref("sgm.models.autoencoder")
    vae = ae.AutoencoderKLInferenceWrapper(
        embed_dim=4, monitor="val/rec_loss", lossconfig=dict(target="torch.nn.Identity"),
        ddconfig=dict(double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=32,
                      ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[], dropout=0.0)).eval()
    fill_module_(vae, prefix="first_stage_model.")
    g = torch.Generator().manual_seed(777)
    z = torch.randn(1, 4, 3, 8, 12, generator=g) * 0.18215 * 4.0
    # decode_first_stage (diffusion.py:151-156): z / scale_factor, then first_stage_model.decode
    dec = vae.decode(1.0 / 0.18215 * z)
    out = dict(z=z.numpy(), dec=dec.numpy())
    np.savez_compressed(os.path.join(HERE, "vae_g32.npz"), **out)
    print("vae_g32.npz dec", tuple(dec.shape), "rms", dec.pow(2).mean().sqrt().item(),
          "size", os.path.getsize(os.path.join(HERE, "vae_g32.npz")))


def gen_vae_encode():
    import contextlib
    import io
    ae = _refshim.ref("sgm.models.autoencoder")
    vae = ae.AutoencoderKLInferenceWrapper(
        embed_dim=4, monitor="val/rec_loss", lossconfig=dict(target="torch.nn.Identity"),
        ddconfig=dict(double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=32,
                      ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[], dropout=0.0)).eval()
    fill_module_(vae, prefix="first_stage_model.")
    g = torch.Generator().manual_seed(778)
    # smooth-ish frames in [-1, 1]: low-resolution noise upsampled, plus a little pixel noise
    base = torch.nn.functional.interpolate(torch.rand(3, 3, 8, 12, generator=g), size=(64, 96), mode="bilinear")
    frames = ((base + 0.1 * torch.rand(3, 3, 64, 96, generator=g)).clamp(0, 1) * 2 - 1)
    x5 = frames.permute(1, 0, 2, 3)[None].contiguous()            # (1, 3, T=3, 64, 96)
    out = dict(x5=x5.numpy().astype(np.float16))                  # fp16 storage is exact enough for an input: re-read as fp32
    x5 = torch.from_numpy(out["x5"].astype(np.float32))
    # the reference draws the posterior noise with torch.randn(mean.shape) from the CPU global generator
    torch.manual_seed(4242)
    noise5 = torch.randn(3, 4, 8, 12)
    torch.manual_seed(4242)
    z5 = vae.encode(x5)                                           # (1, 4, 3, 8, 12), unscaled posterior sample
    mom5 = ae.AutoencoderKL.encode(vae, x5[0].permute(1, 0, 2, 3)).parameters      # (3, 8, 8, 12) = [mean | logvar]
    ref_img = x5[:, :, 1]                                         # (1, 3, 64, 96): the VAEEmbedder / prior_type='ref' input
    torch.manual_seed(99)
    noise4 = torch.randn(1, 4, 8, 12)
    torch.manual_seed(99)
    z4 = vae.encode(ref_img)
    out.update(noise5=noise5.numpy(), z5=z5.numpy(), moments5=mom5.numpy(), noise4=noise4.numpy(), z4=z4.numpy())
    # SDEdit sigma pruning (scripts/demo/streamlit_helpers.py:212-233) around the reference's own discretization
    disc_mod = _refshim.ref("sgm.modules.diffusionmodules.discretizer")
    Wrap = _refshim.ref_class_from_file("scripts/demo/streamlit_helpers.py", "Img2ImgDiscretizationWrapper")
    for n, strength in ((30, 0.6), (5, 0.5), (50, 1.0), (30, 0.01)):
        with contextlib.redirect_stdout(io.StringIO()):
            sig = Wrap(disc_mod.LegacyDDPMDiscretization(), strength=strength)(n, device="cpu")
        out[f"img2img_sigmas_{n}_{strength}"] = sig.numpy()
    np.savez_compressed(os.path.join(HERE, "vae_enc_g32.npz"), **out)
    print("vae_enc_g32.npz z5 rms", z5.pow(2).mean().sqrt().item(), "logvar range",
          mom5[:, 4:].min().item(), mom5[:, 4:].max().item(), "size", os.path.getsize(os.path.join(HERE, "vae_enc_g32.npz")))


CLIP_PREFIX = "conditioner.embedders.0.transformer.text_model."


def gen_clip():
    """FrozenCLIPEmbedder.forward, layer='last' (encoders/modules.py:393-413): tokens -> CLIPTextModel -> last_hidden_state."""
    import transformers
    from transformers import CLIPTextConfig, CLIPTextModel
    cfg = CLIPTextConfig(vocab_size=49408, hidden_size=768, intermediate_size=3072, num_hidden_layers=12,
                         num_attention_heads=12, max_position_embeddings=77, hidden_act="quick_gelu", projection_dim=768)
    m = CLIPTextModel(cfg).eval()
    inner = m.text_model if hasattr(m, "text_model") and any(k.startswith("text_model.") for k in m.state_dict()) else m
    fill_module_(inner, prefix=CLIP_PREFIX)
    g = torch.Generator().manual_seed(31)
    tokens = torch.full((2, 77), 49407, dtype=torch.int64)          # CLIP pads with the end-of-text token
    for b, n in enumerate((9, 40)):
        tokens[b, 0] = 49406
        tokens[b, 1:1 + n] = torch.randint(0, 49406, (n,), generator=g)
    out = m(input_ids=tokens).last_hidden_state
    keys = {CLIP_PREFIX + k: list(v.shape) for k, v in inner.state_dict().items()}
    with open(os.path.join(HERE, "keys_clip_text.json"), "w") as f:
        json.dump(keys, f, indent=0, sort_keys=True)
    np.savez_compressed(os.path.join(HERE, "clip_text.npz"), tokens=tokens.numpy(), last_hidden_state=out.numpy().astype(np.float32))
    print("clip_text.npz transformers", transformers.__version__, "out rms", out.pow(2).mean().sqrt().item(), "keys", len(keys),
          "size", os.path.getsize(os.path.join(HERE, "clip_text.npz")))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "clip":
        gen_clip()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "vae_encode":          # add this fixture without regenerating the others
        gen_vae_encode()
        sys.exit(0)
    torch.manual_seed(0)
    den = gen_sigmas()
    gen_keys()
    wrapper = gen_net()
    gen_sampler(wrapper, den)
    gen_vae()
    gen_net_tvi2v()
    gen_vae_encode()
    gen_clip()
