"""Split-K of g8_kernel at the 8x12 level (3264 rows): TF/s with and without the workspace, operands rotated (cold-ish)."""
import os, sys, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
from ccedit_amd import ops, hip
from ccedit_amd.packing import pack_weight
dev, BF = "cuda", torch.bfloat16
def timeit(fn, nb, reps=3):
    evs = []
    for rep in range(reps + 1):
        for i in range(nb):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(i); e1.record()
            if rep: evs.append((e0, e1))
    torch.cuda.synchronize()
    return sum(a.elapsed_time(b) for a, b in evs) / len(evs)
def both(name, fl, fn, nb=4):
    row = []
    for sk in (False, True):
        ops.SPLIT_K = sk
        ms = timeit(fn, nb)
        row.append(f"{'split' if sk else 'plain'} {ms * 1e3:7.1f} us {fl / ms / 1e9:6.0f} TF/s [{hip.lib().ccedit_last_kernel().decode()}]")
    print(f"{name}: " + "   ".join(row), flush=True)
NB = 4
def conv(n, h, w, cin, cout):
    pw = pack_weight(torch.randn(cout, cin, 3, 3) * (9 * cin) ** -0.5, torch.randn(cout)).to(dev)
    xs = [torch.randn(n, h, w, cin, device=dev).to(BF) for _ in range(NB)]
    rs = [torch.randn(n * h * w, cout, device=dev).to(BF) for _ in range(NB)]
    gb = torch.randn(n, cout, device=dev)
    both(f"conv3x3 {n}x{h}x{w} {cin}->{cout}", 2.0 * n * h * w * cout * cin * 9, lambda i: ops.conv2d(xs[i], pw, res1=rs[i], group_bias=gb, group_rows=h * w))
def temporal(b, t, h, w, c):
    pw = pack_weight(torch.randn(c, c, 3) * (3 * c) ** -0.5, torch.randn(c)).to(dev)
    xs = [torch.randn(b * t, h, w, c, device=dev).to(BF) for _ in range(NB)]
    rs = [torch.randn(b * t * h * w, c, device=dev).to(BF) for _ in range(NB)]
    both(f"temporal {b}x{t}x{h}x{w} {c}->{c}", 2.0 * b * t * h * w * c * c * 3, lambda i: ops.conv_temporal(xs[i], t, pw, res1=rs[i]))
def linear(m, n, k, res=True):
    pw = pack_weight(torch.randn(n, k) * k ** -0.5, torch.randn(n)).to(dev)
    xs = [torch.randn(m, k, device=dev).to(BF) for _ in range(NB)]
    rs = [torch.randn(m, n, device=dev).to(BF) for _ in range(NB)]
    both(f"linear {m}x{n}<-{k}", 2.0 * m * n * k, lambda i: ops.linear(xs[i], pw, res1=rs[i] if res else None))
conv(34, 8, 12, 1280, 1280)
conv(34, 8, 12, 2560, 1280)
temporal(2, 17, 8, 12, 1280)
linear(3264, 1280, 5120)
linear(3264, 1280, 1280)
linear(3264, 1280, 2560)
conv(17, 8, 12, 1280, 1280)
conv(17, 16, 24, 1280, 1280)
temporal(1, 17, 16, 24, 1280)
linear(6528, 1280, 5120)
