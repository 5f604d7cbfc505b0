"""fp32 first stage, upsample + conv 3x3: the nine-tap gather over the virtual up-sampled source against four parity convs (17 frames)"""
import sys, os, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import torch
from ccedit_amd import vae_f32 as V
from ccedit_amd.layers import Conv
dev = torch.device("cuda:0")
def timed(fn, n=3):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return min(ts)
for (c, h, w) in [(256, 256, 384), (512, 128, 192), (512, 64, 96)]:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(17, h, w, c, generator=g).to(dev)
    conv = Conv(c, c, 3)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(c, c, 3, 3, generator=g) * (9 * c) ** -0.5); conv.bias.copy_(torch.randn(c, generator=g))
    conv.pack(dev)
    t9 = timed(lambda: V.conv2d_f32(x, V._pw(conv), upsample=True))
    t4 = timed(lambda: V.upsample_conv2d_f32(x, conv))
    packs = V._pw_parities(conv)
    out = torch.empty((17, 2 * h, 2 * w, c), dtype=torch.float32, device=dev)
    t1 = timed(lambda: V.gemm_f32(x.reshape(-1, c), packs[0], m=17 * h * w, conv=(h, w, h, w, 1, 1, 2), out=out.view(-1, c)))
    fl = 2.0 * 17 * 4 * h * w * c * c * 9
    print(f"{c}->{c} {h}x{w} -> {2*h}x{2*w}: nine taps {t9*1e3:.2f} ms ({fl/t9/1e12:.0f} TF/s eq), four parity convs {t4*1e3:.2f} ms ({fl*4/9/t4/1e12:.0f} TF/s executed), one parity launch {t1*1e3:.2f} ms", flush=True)
    del x, out
