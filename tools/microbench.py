#!/usr/bin/env python3
"""Kernel micro-benchmarks at the hot path's real shapes (17x512x768, CFG batch 2 => 34 frames).
Prints achieved TFLOP/s or GB/s per kernel; run on the GPU box:  python tools/microbench.py"""
import sys, os, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from ccedit_amd import ops
from ccedit_amd.packing import pack_weight

BF = torch.bfloat16
dev = "cuda"


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def rnd(*s):
    return torch.randn(*s, device=dev, dtype=torch.float32).to(BF)


def bench_conv(n, h, w, cin, cout, stride=1, tile=0, name=""):
    x = rnd(n, h, w, cin)
    pw = pack_weight(torch.randn(cout, cin, 3, 3) * (9 * cin) ** -0.5, torch.randn(cout)).to(dev)
    t = timeit(lambda: ops.conv2d(x, pw, stride=stride, tile=tile))
    fl = 2.0 * n * (h // stride) * (w // stride) * cout * cin * 9
    print(f"conv3x3 {name} ({n},{h},{w},{cin})->{cout} s{stride} tile{tile}: {t*1e3:8.3f} ms  {fl/t/1e12:7.1f} TF/s")


def bench_linear(m, k, nn, tile=0, geglu=False, name=""):
    x = rnd(m, k)
    pw = pack_weight(torch.randn(nn, k) * k ** -0.5, torch.randn(nn), geglu=geglu).to(dev)
    t = timeit(lambda: ops.linear(x, pw, tile=tile))
    fl = 2.0 * m * k * nn
    print(f"linear {name} M={m} K={k} N={nn} tile{tile}: {t*1e3:8.3f} ms  {fl/t/1e12:7.1f} TF/s")


def bench_temporal(b, t_, h, w, c):
    x = rnd(b * t_, h, w, c)
    pw = pack_weight(torch.randn(c, c, 3) * (3 * c) ** -0.5, torch.randn(c)).to(dev)
    t = timeit(lambda: ops.conv_temporal(x, t_, pw))
    fl = 2.0 * b * t_ * h * w * c * c * 3
    print(f"conv1d-T k3 ({b}x{t_},{h},{w},{c}): {t*1e3:8.3f} ms  {fl/t/1e12:7.1f} TF/s")


def bench_attn(n, l, lk, heads, d, name="", **kw):
    c = heads * d
    q, k, v = rnd(n * l, c), rnd(kw.get("kvrows", n * lk), c), rnd(kw.get("kvrows", n * lk), c)
    kw.pop("kvrows", None)
    t = timeit(lambda: ops.attention(q, k, v, heads, d, batches=n, lq=l, lk=lk, **kw))
    fl = 4.0 * n * heads * l * lk * d
    print(f"attention {name} n={n} Lq={l} Lk={lk} h={heads} d={d}: {t*1e3:8.3f} ms  {fl/t/1e12:7.1f} TF/s")


def bench_gn(n, h, w, c):
    x = rnd(n, h, w, c)
    g, b = torch.ones(c, device=dev), torch.zeros(c, device=dev)
    t = timeit(lambda: ops.groupnorm_spatial(x, g, b, 1e-5, True))
    by = 3.0 * x.numel() * 2
    print(f"GN+SiLU spatial ({n},{h},{w},{c}): {t*1e3:8.3f} ms  {by/t/1e9:7.0f} GB/s (2 reads + 1 write)")


def bench_gnt(b, t_, h, w, c):
    x = rnd(b * t_, h, w, c)
    g, bb = torch.ones(c, device=dev), torch.zeros(c, device=dev)
    t = timeit(lambda: ops.groupnorm_temporal(x, b, t_, g, bb, 1e-5, True))
    by = 2.0 * x.numel() * 2
    print(f"GN+SiLU temporal ({b}x{t_},{h},{w},{c}): {t*1e3:8.3f} ms  {by/t/1e9:7.0f} GB/s (algorithmic 1R+1W)")


def bench_ln(rows, c):
    x = rnd(rows, c)
    g, b = torch.ones(c, device=dev), torch.zeros(c, device=dev)
    t = timeit(lambda: ops.layernorm(x, g, b))
    print(f"LayerNorm ({rows},{c}): {t*1e3:8.3f} ms  {2.0*x.numel()*2/t/1e9:7.0f} GB/s")


if __name__ == "__main__":
    which = sys.argv[1:] or ["conv", "linear", "temporal", "attn", "norm"]
    N = 34
    if "conv" in which:
        for tile in (1, 2, 3):
            bench_conv(N, 64, 96, 320, 320, tile=tile, name="L0")
        bench_conv(N, 32, 48, 640, 640, name="L1")
        bench_conv(N, 16, 24, 1280, 1280, name="L2")
        bench_conv(N, 8, 12, 1280, 1280, name="L3")
        bench_conv(N, 64, 96, 960, 320, name="dec L0")
        bench_conv(N, 64, 96, 320, 320, stride=2, name="down")
    if "linear" in which:
        M = N * 6144
        for tile in (1, 2, 3):
            bench_linear(M, 320, 960, tile=tile, name="qkv L0")
        bench_linear(M, 320, 320, name="to_out L0")
        bench_linear(M, 320, 2560, geglu=True, name="ff.proj L0")
        bench_linear(M, 1280, 320, name="ff.out L0")
        bench_linear(N * 1536, 640, 1920, name="qkv L1")
        bench_linear(N * 384, 1280, 3840, name="qkv L2")
        bench_linear(N * 384, 1280, 10240, geglu=True, name="ff.proj L2")
    if "temporal" in which:
        bench_temporal(2, 17, 64, 96, 320)
        bench_temporal(2, 17, 32, 48, 640)
        bench_temporal(2, 17, 16, 24, 1280)
    if "attn" in which:
        bench_attn(N, 6144, 6144, 8, 40, name="spatial L0")
        bench_attn(N, 1536, 1536, 8, 80, name="spatial L1")
        bench_attn(N, 384, 384, 8, 160, name="spatial L2")
        bench_attn(N, 6144, 77, 8, 40, name="text L0", kv_div=17, kvrows=2 * 77)
        hw = 6144
        bench_attn(2 * hw, 17, 17, 8, 40, name="temporal L0", q_inner=hw, q_outer_rows=17 * hw, q_inner_rows=1,
                   q_seq_rows=hw, kv_inner=hw, kv_outer_rows=17 * hw, kv_inner_rows=1, kv_seq_rows=hw, kvrows=2 * 17 * hw)
    if "norm" in which:
        bench_gn(N, 64, 96, 320)
        bench_gn(N, 32, 48, 640)
        bench_gn(N, 64, 96, 960)
        bench_gnt(2, 17, 64, 96, 320)
        bench_gnt(2, 17, 16, 24, 1280)
        bench_ln(N * 6144, 320)
        bench_ln(N * 384, 1280)
