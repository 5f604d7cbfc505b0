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
elif which == "attn":
    q = torch.randn(N * 6144, 960, device="cuda").to(BF)
    f = lambda: ops.attention(q[:, :320], q[:, 320:640], q[:, 640:], 8, 40, batches=N, lq=6144, lk=6144)
for _ in range(4):
    f()
torch.cuda.synchronize()
