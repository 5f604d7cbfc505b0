"""Helper of test_fullsize_gpu.py: one network evaluation (and one VAE decode) at BASELINE.json's full size
(17 keyframes, 512x768, CFG-doubled batch) under whatever kernel-policy environment the parent test set, saved to .npz.
Run as a subprocess because the policy switches are read once per process."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)


def main_tvi2v(out_path):
    """BASELINE.json config 3 at full size: controlnet_img on the reference latent + anchor cross-frame attention."""
    from ccedit_amd.sgm_compat import build_network
    from ccedit_amd.utils.synth import fill_module_
    torch.set_grad_enabled(False)
    dev = torch.device("cuda")
    T, H, W = 17, 64, 96
    w = build_network(dev, crossframe=True)
    fill_module_(w, prefix="model.")
    w.diffusion_model.pack(dev)
    g = torch.Generator().manual_seed(321)
    x = torch.randn(1, 4, T, H, W, generator=g)
    cc, cu = torch.randn(1, 77, 768, generator=g), torch.randn(1, 77, 768, generator=g)
    hint = (torch.rand(1, 1, T, 8 * H, 8 * W, generator=g) * 2 - 1).repeat(1, 3, 1, 1, 1)
    cf, cf2 = (torch.randn(1, 4, H, W, generator=g) * 0.18215 for _ in range(2))
    t = torch.tensor([601, 601], dtype=torch.int64, device=dev)

    def run(cfa, cfb, ca, cb):
        c = dict(crossattn=torch.cat([ca, cb]).to(dev), control_hint=torch.cat([hint, hint]).to(dev), cond_feat=torch.cat([cfa, cfb]).to(dev))
        return w(torch.cat([x, x]).to(dev), t, c).float().cpu().numpy()

    np.savez(out_path, eps=run(cf, cf, cu, cc),        # the CFG pair: same latent / reference frame, two prompts
             eps_same=run(cf, cf, cc, cc),             # identical halves -> identical predictions
             eps_ref=run(cf, cf2, cu, cc))             # another reference frame in half 1: half 0 must not change, half 1 must


def main(out_path):
    from ccedit_amd.sgm_compat import build_network, build_vae
    from ccedit_amd.utils.synth import fill_module_
    torch.set_grad_enabled(False)
    dev = torch.device("cuda")
    T, H, W = 17, 64, 96
    w = build_network(dev)
    fill_module_(w, prefix="model.")
    w.diffusion_model.pack(dev)
    g = torch.Generator().manual_seed(123)
    x = torch.randn(1, 4, T, H, W, generator=g)
    xb = torch.randn(1, 4, T, H, W, generator=g)
    cc, cu = torch.randn(1, 77, 768, generator=g), torch.randn(1, 77, 768, generator=g)
    hint = (torch.rand(1, 1, T, 8 * H, 8 * W, generator=g) * 2 - 1).repeat(1, 3, 1, 1, 1)
    t = torch.tensor([601, 601], dtype=torch.int64, device=dev)

    def run(xa, xb_, ca, cb):
        c = dict(crossattn=torch.cat([ca, cb]).to(dev), control_hint=torch.cat([hint, hint]).to(dev))
        return w(torch.cat([xa, xb_]).to(dev), t, c).float().cpu().numpy()

    # the same conditioning tensors three times, as a sampler does: eager, HIP-graph capture + replay, replay with another latent
    # and timestep put into the static inputs and the first pair put back — the graph must reproduce the eager bits
    c = dict(crossattn=torch.cat([cu, cc]).to(dev), control_hint=torch.cat([hint, hint]).to(dev))
    xx, t2 = torch.cat([x, x]).to(dev), torch.tensor([333, 333], dtype=torch.int64, device=dev)
    e_eager = w(xx, t, c).clone()
    e_graph = w(xx, t, c).clone()
    e_moved = w(torch.cat([xb, xb]).to(dev), t2, c).clone()
    e_back = w(xx, t, c).clone()
    graphed = bool(getattr(w, "_graphs", None)) and any("graph" in e for e in w._graphs.values())
    assert graphed == (w.use_graph and not type(w)._graph_failed), "the HIP graph was not captured"
    assert torch.equal(e_eager, e_graph) and torch.equal(e_eager, e_back), "HIP-graph replay differs from the eager evaluation"
    assert not torch.equal(e_eager, e_moved), "the replay did not pick up the new latent / timestep"
    # a SECOND graph (other conditioning tensors, as the next clip or the sampler after the step benchmark brings): its replays must
    # not inherit scratch (zeroed statistics arenas of the main AND the ControlNet's side stream) from the first graph's capture
    c2 = dict(crossattn=torch.cat([cc, cu]).to(dev), control_hint=torch.cat([hint, hint]).to(dev))
    f = [w(xx, t, c2).clone() for _ in range(4)]                   # eager, capture + replay, replay, replay
    assert all(torch.equal(f[0], fi) for fi in f[1:]), "replays of the second HIP graph differ from its eager evaluation"
    assert torch.equal(w(xx, t, c), e_eager) and torch.equal(w(xx, t, c2), f[0]), "the two graphs disturb each other"
    if graphed:
        assert sum("graph" in e for e in w._graphs.values()) == 2

    out = dict(eps=run(x, x, cu, cc),                 # the CFG pair of the benchmark: same latent, two prompts
               eps_same=run(x, x, cc, cc),            # identical halves -> identical predictions
               eps_other=run(x, xb, cu, cc))          # clips do not interact: half 0 must not change
    del w
    torch.cuda.empty_cache()
    vae = build_vae(dev)
    fill_module_(vae, prefix="first_stage_model.")
    vae.pack(dev)
    z = torch.randn(1, 4, 3, H, W, generator=g).to(dev)           # 3 frames of the full 512x768 size
    out["frames"] = vae.decode(z).float().cpu().numpy()
    np.savez(out_path, **out)


if __name__ == "__main__":
    (main_tvi2v if len(sys.argv) > 2 and sys.argv[2] == "tvi2v" else main)(sys.argv[1])
