"""Does a producer -> consumer chain of HBM-bound level-0 ops run faster on HALF the batch (67 MB tensors) than on the whole CFG-doubled
batch (134 MB tensors) — i.e. is there Infinity-Cache reuse to win by evaluating the halves one after the other?"""
import sys, os
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import torch
from ccedit_amd import ops
from ccedit_amd.packing import pack_weight

BF = torch.bfloat16
dev = "cuda"
frames, h, w, c = 34, 64, 96, 320
g = torch.Generator().manual_seed(0)
x = torch.randn(frames, h, w, c, generator=g).to(BF).to(dev)
gam, bet = torch.ones(c, device=dev), torch.zeros(c, device=dev)
pw1 = pack_weight(torch.randn(c, c, generator=g) * c ** -0.5, torch.randn(c, generator=g)).to(dev)
pw2 = pack_weight(torch.randn(c, c, generator=g) * c ** -0.5, torch.randn(c, generator=g)).to(dev)


def chain(xx):
    n = xx.shape[0]
    a = ops.groupnorm_spatial(xx, gam, bet, 1e-6, False)
    b = ops.linear(a.view(-1, c), pw1)
    cc = ops.layernorm(b, gam, bet, 1e-5)
    d = ops.linear(cc, pw2, res1=b)
    e = ops.groupnorm_spatial(d.view(n, h, w, c), gam, bet, 1e-5, True)
    return e


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


for parts in (1, 2, 4):
    k = frames // parts if frames % parts == 0 else None
    if k is None:
        xs = [x[:9], x[9:18], x[18:26], x[26:]]
    else:
        xs = [x[i * k:(i + 1) * k] for i in range(parts)]
    us = timeit(lambda: [chain(t) for t in xs])
    print(f"{parts} part(s) of {[t.shape[0] for t in xs]} frames: {us:8.1f} us per whole batch", flush=True)
