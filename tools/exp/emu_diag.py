"""Stage-by-stage comparison of the HIP path with the oracle's bf16-emulation mode on the first ControlNet block
(ResBlock + SpatialTransformer): every stage is fed the ORACLE's input, so a stage's number is its own mismatch."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import numpy as np, torch, torch.nn.functional as F
from ccedit_amd import ops, network
from ccedit_amd.sgm_compat import build_network, build_network_spec
from ccedit_amd.utils.synth import fill_module_, synth_state_dict
from oracle import ccedit_oracle as O

G = dict(model_channels=160, num_heads=4, context_dim=128)
w = build_network("cpu", **G); fill_module_(w, prefix="model."); w.diffusion_model.pack("cuda")
sd = synth_state_dict(build_network_spec(G))
cn = w.diffusion_model.controlnet
P = "model.diffusion_model.controlnet"
g = torch.Generator().manual_seed(0)
bt, c, hh, ww = 6, 160, 16, 24
x = (torch.randn(bt, c, hh, ww, generator=g)).to(torch.bfloat16).float()
ctx = torch.randn(2, 77, 128, generator=g).to(torch.bfloat16).float()
t = torch.tensor([601, 601])


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt()).item()


def nhwc(x4):
    return x4.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).cuda()


def nchw(y):
    return y.float().cpu().permute(0, 3, 1, 2)


with O.bf16_emulation():
    emb = O.time_embed(sd, P + ".time_embed", t, 160)
    embo = O._emb_out(sd, P + ".input_blocks.1.0", emb)
    e_hip = cn._emb_silu(t.cuda())
    rb = cn.input_blocks[1][0]
    print("emb_out", rel(e_hip.of(rb), embo))
    rp = P + ".input_blocks.1.0"
    # ResBlock stages
    a_o = O._gn(sd, rp + ".in_layers.0", x, 1e-5, silu=True)
    a_h = ops.groupnorm_spatial(nhwc(x), rb.in_layers[0].g, rb.in_layers[0].b, 1e-5, True)
    print("gn+silu", rel(nchw(a_h), a_o))
    embr = embo.repeat_interleave(3, dim=0)
    h_o = O._conv2d(sd, rp + ".in_layers.2", a_o, padding=1, add=[embr[:, :, None, None]])
    h_h = ops.conv2d(nhwc(a_o), rb.in_layers[2].pw, group_bias=e_hip.of(rb), group_rows=3 * hh * ww, gn=True)
    print("conv+emb", rel(nchw(h_h), h_o))
    a2_o = O._gn(sd, rp + ".out_layers.0", h_o, 1e-5, silu=True)
    a2_h = ops.groupnorm_spatial(nhwc(h_o), rb.out_layers[0].g, rb.out_layers[0].b, 1e-5, True)
    print("gn2+silu", rel(nchw(a2_h), a2_o))
    o_o = O._conv2d(sd, rp + ".out_layers.3", a2_o, padding=1, add=[x])
    o_h = ops.conv2d(nhwc(a2_o), rb.out_layers[3].pw, res1=nhwc(x).view(-1, c))
    print("conv+res", rel(nchw(o_h), o_o))
    print("resblock whole", rel(nchw(rb.run(nhwc(x), e_hip, network.Geometry(2, 3))), O.resblock2d(sd, rp, x, emb.repeat_interleave(3, dim=0))))
    # transformer stages
    st = cn.input_blocks[1][1]
    tp = P + ".input_blocks.1.1"
    y_o = O._conv2d(sd, tp + ".proj_in", O._gn(sd, tp + ".norm", x, 1e-6))
    an = ops.groupnorm_spatial(nhwc(x), st.norm.g, st.norm.b, 1e-6, False)
    y_h = ops.linear(an.view(-1, c), st.proj_in.pw)
    print("gn+proj_in", rel(y_h.view(bt, hh, ww, c).permute(0, 3, 1, 2), y_o))
    tok_o = y_o.flatten(2).transpose(1, 2)
    blk = st.transformer_blocks[0]
    bp = tp + ".transformer_blocks.0"
    tok_h0 = tok_o.reshape(-1, c).to(torch.bfloat16).cuda()
    n1_o = O._ln(sd, bp + ".norm1", tok_o)
    n1_h = ops.layernorm(tok_h0, blk.norm1.g, blk.norm1.b)
    print("ln1", rel(n1_h.view(bt, -1, c), n1_o))
    a1 = blk.attn1
    qkv = ops.linear(n1_o.reshape(-1, c).to(torch.bfloat16).cuda(), a1.qkv)
    q_o, k_o, v_o = (O._linear(sd, bp + ".attn1." + n, n1_o) for n in ("to_q", "to_k", "to_v"))
    print("q", rel(qkv[:, :c].view(bt, -1, c), q_o), "k", rel(qkv[:, c:2 * c].view(bt, -1, c), k_o), "v", rel(qkv[:, 2 * c:].view(bt, -1, c), v_o))
    heads, d = 4, 40
    qq, kk, vv = (z.reshape(bt, -1, heads, d).transpose(1, 2) for z in (q_o, k_o, v_o))
    o_o2 = O._sdpa(qq, kk, vv).transpose(1, 2).reshape(bt, -1, c)
    qkv_o = torch.cat([q_o, k_o, v_o], dim=-1).reshape(-1, 3 * c).to(torch.bfloat16).cuda()
    o_h2 = ops.attention(qkv_o[:, :c], qkv_o[:, c:2 * c], qkv_o[:, 2 * c:], heads, d, batches=bt, lq=hh * ww, lk=hh * ww)
    print("sdpa self", rel(o_h2.view(bt, -1, c), o_o2))
    with_fp32 = F.scaled_dot_product_attention(qq, kk, vv).transpose(1, 2).reshape(bt, -1, c)
    print("   (sdpa emu vs exact fp32 sdpa)", rel(o_o2, with_fp32), " hip vs exact", rel(o_h2.view(bt, -1, c), with_fp32))
    t1_o = O._linear(sd, bp + ".attn1.to_out.0", o_o2, add=[tok_o])
    t1_h = ops.linear(o_o2.reshape(-1, c).to(torch.bfloat16).cuda(), a1.to_out[0].pw, res1=tok_h0)
    print("to_out+res", rel(t1_h.view(bt, -1, c), t1_o))
    ff_o = O.feed_forward(sd, bp + ".ff", bp + ".norm3", tok_o)
    ff_h = blk.ff.run(tok_h0, blk.norm3)
    print("ff", rel(ff_h.view(bt, -1, c), ff_o))
    whole_o = O.spatial_transformer2d(sd, tp, x, ctx.repeat_interleave(3, dim=0), heads)
    ctx2d = ctx.reshape(-1, 128).to(torch.bfloat16).cuda()
    whole_h = st.run_spatial(nhwc(x), ctx2d, 77, 3)
    print("transformer whole", rel(nchw(whole_h), whole_o))
