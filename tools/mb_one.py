#!/usr/bin/env python3
"""Run ONE microbench case a few times (for PMC collection): python tools/mb_one.py conv|qkv|attn"""
import sys, os
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from ccedit_amd import ops
from ccedit_amd.packing import pack_weight
BF = torch.bfloat16
which = sys.argv[1]
N = 34
if which == "conv":
    x = torch.randn(N, 64, 96, 320, device="cuda").to(BF)
    pw = pack_weight(torch.randn(320, 320, 3, 3) * 0.02, torch.randn(320)).to("cuda")
    f = lambda: ops.conv2d(x, pw, tile=int(sys.argv[2]) if len(sys.argv) > 2 else 0)
elif which == "qkv":
    x = torch.randn(N * 6144, 320, device="cuda").to(BF)
    pw = pack_weight(torch.randn(960, 320) * 0.05).to("cuda")
    f = lambda: ops.linear(x, pw, tile=int(sys.argv[2]) if len(sys.argv) > 2 else 0)
elif which == "g8geglu":         # GEGLU projection of the 32x48 level with the LayerNorm folded (g8_kernel 256x256, LNF)
    from ccedit_amd.packing import fold_layernorm
    x = torch.randn(N * 1536, 640, device="cuda").to(BF)
    pw = fold_layernorm([torch.randn(5120, 640) * 0.04], [torch.randn(5120)], torch.ones(640), torch.zeros(640), device="cuda", geglu=True)
    st = ops.row_stats(x, 1e-5)
    f = lambda: ops.linear(x, pw, ln_stats=st)
elif which == "g8res":           # 52224 x 640 <- 640 + residual (g8_kernel 128ch x 512pix)
    x = torch.randn(N * 1536, 640, device="cuda").to(BF)
    r = torch.randn(N * 1536, 640, device="cuda").to(BF)
    pw = pack_weight(torch.randn(640, 640) * 0.04, torch.randn(640)).to("cuda")
    f = lambda: ops.linear(x, pw, res1=r)
elif which == "g8conv":          # 3x3 conv 1280 -> 1280 at 16x24 (g8_kernel 256x256, 3x3 tap gather)
    x = torch.randn(N, 16, 24, 1280, device="cuda").to(BF)
    pw = pack_weight(torch.randn(1280, 1280, 3, 3) * 0.01, torch.randn(1280)).to("cuda")
    f = lambda: ops.conv2d(x, pw)
elif which == "lin320":          # 208896 x 320 <- 320 + residual (lin320_kernel)
    x = torch.randn(N * 6144, 320, device="cuda").to(BF)
    r = torch.randn(N * 6144, 320, device="cuda").to(BF)
    pw = pack_weight(torch.randn(320, 320) * 0.05, torch.randn(320)).to("cuda")
    f = lambda: ops.linear(x, pw, res1=r)
elif which == "lin640":          # fused q,k,v projection of the 32x48 level, LayerNorm folded: 52224 x 1920 <- 640 (lin640s_kernel)
    from ccedit_amd.packing import fold_layernorm
    x = torch.randn(N * 1536, 640, device="cuda").to(BF)
    pw = fold_layernorm([torch.randn(1920, 640) * 0.04], [torch.randn(1920)], torch.ones(640), torch.zeros(640)).to("cuda")
    st = ops.row_stats(x, 1e-5)
    f = lambda: ops.linear(x, pw, ln_stats=st)
elif which in ("ff320", "ff320tail"):   # the dim-320 feed-forward alone / the block tail (to_out + FF + proj_out) at 34 x 6144 tokens
    from ccedit_amd.packing import pack_ff320, pack_ff320_tail
    g = torch.Generator().manual_seed(0)
    base = pack_ff320(torch.randn(2560, 320, generator=g) * 0.05, torch.randn(2560, generator=g) * 0.1, torch.randn(320, 1280, generator=g) * 0.03,
                      torch.randn(320, generator=g) * 0.1, torch.ones(320), torch.zeros(320), device="cuda")
    a, r, x2 = (torch.randn(N * 6144, 320, device="cuda").to(BF) for _ in range(3))
    if which == "ff320":
        f = lambda: ops.ff320(a, base)
    else:
        pk = pack_ff320_tail(base, torch.randn(320, 320, generator=g) * 0.05, torch.randn(320, generator=g) * 0.1,
                             torch.randn(320, 320, generator=g) * 0.05, torch.randn(320, generator=g) * 0.1, device="cuda")
        f = lambda: ops.ff320(None, pk, a=a, res=r, res2=x2)
elif which == "attnq":           # the network's call: q pre-scaled into log2 units
    q = torch.randn(N * 6144, 960, device="cuda").to(BF)
    f = lambda: ops.attention(q[:, :320], q[:, 320:640], q[:, 640:], 8, 40, batches=N, lq=6144, lk=6144, q_log2=True)
elif which == "attnq80":         # the 32x48 level: 1536 keys, d = 80, q in log2 units
    q = torch.randn(N * 1536, 1920, device="cuda").to(BF)
    f = lambda: ops.attention(q[:, :640], q[:, 640:1280], q[:, 1280:], 8, 80, batches=N, lq=1536, lk=1536, q_log2=True)
elif which == "attn":
    q = torch.randn(N * 6144, 960, device="cuda").to(BF)
    f = lambda: ops.attention(q[:, :320], q[:, 320:640], q[:, 640:], 8, 40, batches=N, lq=6144, lk=6144)
elif which in ("f32conv", "f32conv128"):   # fp32 first stage: 3x3 conv 512 -> 512 at 128x192 (f32p_gemm_kernel) / 128 -> 128 at 512x768 (f32s_gemm_kernel), 17 frames;
    from ccedit_amd import vae_f32 as V          # CCEDIT_POLICY=f32_split=0 puts both on v_mfma_f32_32x32x2_f32 (f32_gemm_kernel)
    cin, hh, ww = (512, 128, 192) if which == "f32conv" else (128, 512, 768)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(17, hh, ww, cin, generator=g).to("cuda")
    pw = V.pack_f32(torch.randn(cin, cin, 3, 3, generator=g) * (9 * cin) ** -0.5, torch.randn(cin, generator=g), "cuda")
    f = lambda: V.conv2d_f32(x, pw)
for _ in range(4):
    f()
torch.cuda.synchronize()
