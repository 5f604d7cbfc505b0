"""Block-shape sweep for the low rows of the step: the 8x12-level Linears, the stride-2 convs, the 8x12 temporal
convs.  us per launch by CcGemmDesc.tile (0 = the library's choice); operands rotate over several buffers (cold, as in the network)."""
import sys, os
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import torch
from ccedit_amd import hip, ops
from ccedit_amd.packing import pack_weight

NB = 4


def timeit(fn, reps=20):
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def rnd(*shape):
    return [torch.randn(*shape, device="cuda").to(torch.bfloat16) for _ in range(NB)]


def sweep(name, fn, tiles, flops):
    out = []
    for tile in tiles:
        try:
            us = timeit(lambda i: fn(i, tile))
            k = hip.lib().ccedit_last_kernel().decode()
            out.append(f"t{tile}:{us:6.1f}us/{flops / us / 1e6:5.0f}TF" + (f" [{k[:28]}]" if tile == 0 else ""))
        except Exception as e:
            out.append(f"t{tile}: n/a")
    print(f"{name:36s} " + "  ".join(out), flush=True)


def lin(name, m, k, n, res=False, tiles=(0, 1, 2, 3, 4, 5, 6, 7, 11, 12, 13)):
    xs, rs = rnd(m, k), rnd(m, n)
    pws = [pack_weight(torch.randn(n, k) * k ** -0.5, torch.randn(n)).to("cuda") for _ in range(NB)]
    sweep(name + (" +res" if res else ""), lambda i, t: ops.linear(xs[i % NB], pws[i % NB], res1=rs[i % NB] if res else None, tile=t), tiles, 2.0 * m * k * n)


def conv(name, n_, h, w, cin, cout, stride, tiles=(0, 1, 2, 3)):
    xs = rnd(n_, h, w, cin)
    pws = [pack_weight(torch.randn(cout, cin, 3, 3) * (9 * cin) ** -0.5, torch.randn(cout)).to("cuda") for _ in range(NB)]
    ho, wo = h // stride, w // stride
    sweep(name, lambda i, t: ops.conv2d(xs[i % NB], pws[i % NB], stride=stride, tile=t), tiles, 2.0 * n_ * ho * wo * cout * cin * 9)


def temp(name, n_, h, w, c, tiles=(0, 1, 2, 3, 6)):
    xs = rnd(n_, h, w, c)
    pws = [pack_weight(torch.randn(c, c, 3) * (3 * c) ** -0.5, torch.randn(c)).to("cuda") for _ in range(NB)]
    sweep(name, lambda i, t: ops.conv_temporal(xs[i % NB], 17, pws[i % NB], res1=xs[(i + 1) % NB].view(-1, c), tile=t), tiles, 2.0 * n_ * h * w * c * c * 3)


lin("L3 lin 3264x1280<-1280", 3264, 1280, 1280)
lin("L3 lin 3264x1280<-1280", 3264, 1280, 1280, res=True)
lin("L3 lin 3264x1280<-5120", 3264, 5120, 1280, res=True)
lin("L3 lin 3264x1280<-2560", 3264, 2560, 1280)
lin("L3 lin 3264x3840<-1280", 3264, 1280, 3840)
lin("text kv 154x24960<-768", 154, 768, 24960, tiles=(0, 1, 2, 3))
conv("down L0 52224x320<-2880 s2", 34, 64, 96, 320, 320, 2)
conv("down L1 13056x640<-5760 s2", 34, 32, 48, 640, 640, 2)
conv("down L2 3264x1280<-11520 s2", 34, 16, 24, 1280, 1280, 2)
temp("temp L3 3264x1280<-3840", 34, 8, 12, 1280)
temp("temp L2 13056x1280<-3840", 34, 16, 24, 1280)
