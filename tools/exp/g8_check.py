"""tile 11 (gemm8p.hip, persistent 256x256 eight-phase Linear): correctness + race screen + throughput against the vendor GEMM
(torch.matmul = hipBLASLt; calibration only) and the older block shapes, hot (operands re-read from the Infinity Cache) and cold
(every launch reads what a producer pass just wrote, rotating buffers — the setting GEMMs see inside the network)."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import torch
import torch.nn.functional as F
from ccedit_amd import ops, hip
from ccedit_amd.packing import pack_weight

BF = torch.bfloat16
T11 = int(os.environ.get("G8_TILE", "11"))
dev = "cuda"


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(BF)


def check(m, n, k, geglu=False, res=0, act=0, reps=4, T11=T11):
    x, w, b = rnd(m, k, seed=1), rnd(n, k, seed=2, scale=k ** -0.5), rnd(n, seed=3).float()
    pw = pack_weight(w.float(), b, geglu=geglu).to(dev)
    xc = x.to(dev)
    r1 = rnd(m, n, seed=4).to(dev) if res >= 1 else None
    r2 = rnd(m, n, seed=5).to(dev) if res >= 2 else None
    ref = F.linear(xc.float(), w.to(dev).float(), b.to(dev))
    if geglu:
        a, g = ref.chunk(2, dim=-1)
        ref = a * F.gelu(g)
    if act == 1:
        ref = F.silu(ref)
    if r1 is not None:
        ref = ref + r1.float()
    if r2 is not None:
        ref = ref + r2.float()
    y0 = ops.linear(xc, pw, res1=r1, res2=r2, act=act, tile=T11)
    torch.cuda.synchronize()
    err = (y0.float() - ref).abs().max().item()
    lim = 2.0 ** -7 * ref.abs().max().item() + 1e-3
    bad = 0
    for _ in range(reps):
        y = ops.linear(xc, pw, res1=r1, res2=r2, act=act, tile=T11)
        bad += int(not torch.equal(y, y0))
    ok = err <= lim and bad == 0 and bool(torch.isfinite(y0.float()).all())
    print(f"{'ok ' if ok else 'BAD'} M={m} N={n} K={k} geglu={int(geglu)} res={res} act={act}: err {err:.3g} (lim {lim:.3g}), {bad}/{reps} reruns differ",
          flush=True)
    return ok


def timeit(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def hip_ln(x, y, g, b):
    if x.shape[1] > 1536:
        y.copy_(x)
        return
    hip.check(hip.lib().ccedit_layernorm(x.data_ptr(), y.data_ptr(), g.data_ptr(), b.data_ptr(), x.shape[0], x.shape[1], 1e-5,
                                         torch.cuda.current_stream().cuda_stream), "ln")


def perf(m, n, k, geglu=False, res=False, tiles=(1, 3, 4, 6, 11)):
    a = torch.randn(m, k, device=dev, dtype=BF)
    wt = torch.randn(n, k) * k ** -0.5
    pw = pack_weight(wt, torch.randn(n), geglu=geglu).to(dev)
    wv = wt.to(dev).to(BF)
    r = torch.randn(m, n, device=dev, dtype=BF) if res else None
    fl = 2.0 * m * n * k
    row = [f"vendor {fl / timeit(lambda: torch.matmul(a, wv.t())) / 1e9:5.0f}"]
    for t in tiles:
        if t == 6 and n % 320:
            continue
        try:
            row.append(f"t{t} {fl / timeit(lambda: ops.linear(a, pw, res1=r, tile=t)) / 1e9:5.0f}")
        except Exception as e:
            row.append(f"t{t} err")
    # cold: producer (LayerNorm-sized pass) writes the activation, rotating over NB buffer sets
    NB = 6
    src = [torch.randn(m, k, device=dev).to(BF) for _ in range(NB)]
    act = [torch.empty_like(s) for s in src]
    rs = [torch.randn(m, n, device=dev).to(BF) for _ in range(NB)] if res else [None] * NB
    g1, b1 = torch.ones(k, device=dev), torch.zeros(k, device=dev)
    outs = [torch.empty(m, n // 2 if geglu else n, device=dev, dtype=BF) for _ in range(NB)]
    cold = []

    def cold_run(f):
        evs = []
        for rep in range(3):
            for i in range(NB):
                hip_ln(src[i], act[i], g1, b1)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); f(i); e1.record()
                if rep:
                    evs.append((e0, e1))
        torch.cuda.synchronize()
        return sum(x.elapsed_time(y) for x, y in evs) / len(evs)
    vouts = [torch.empty(m, n, device=dev, dtype=BF) for _ in range(NB)]
    cold.append(f"vendor {fl / cold_run(lambda i: torch.matmul(act[i], wv.t(), out=vouts[i])) / 1e9:5.0f}")
    for t in tiles:
        if t == 6 and n % 320:
            continue
        try:
            cold.append(f"t{t} {fl / cold_run(lambda i: ops.linear(act[i], pw, res1=rs[i], out=outs[i], tile=t)) / 1e9:5.0f}")
        except Exception as e:
            cold.append(f"t{t} err")
    print(f"M={m:6d} N={n:5d} K={k:5d} geglu={int(geglu)} res={int(res)} | hot: " + "  ".join(row) + " | cold: " + "  ".join(cold), flush=True)


def perf_temporal(b_, t, h, w, c, cout, res=1, tiles=(0, 1, 2, 12, 13)):
    """Conv1d k3 over T (temporal mode), cold operands, + residual(s): TF/s per block shape."""
    n = b_ * t
    pw = pack_weight(torch.randn(cout, c, 3) * (3 * c) ** -0.5, torch.randn(cout)).to(dev)
    NB = 4
    src = [torch.randn(n * h * w, c, device=dev).to(BF) for _ in range(NB)]
    act = [torch.empty_like(x) for x in src]
    rs = [torch.randn(n * h * w, cout, device=dev).to(BF) for _ in range(NB)]
    g1, b1 = torch.ones(c, device=dev), torch.zeros(c, device=dev)
    fl = 2.0 * n * h * w * cout * c * 3
    row = []
    for tl in tiles:
        try:
            evs = []
            for rep in range(3):
                for i in range(NB):
                    hip_ln(src[i], act[i], g1, b1)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    ops.conv_temporal(act[i].view(n, h, w, c), t, pw, res1=rs[i], res2=rs[(i + 1) % NB] if res > 1 else None, tile=tl)
                    e1.record()
                    if rep:
                        evs.append((e0, e1))
            torch.cuda.synchronize()
            row.append(f"t{tl} {fl / (sum(x.elapsed_time(y) for x, y in evs) / len(evs)) / 1e9:5.0f}")
        except Exception as e:
            row.append(f"t{tl} err")
    print(f"temporal B={b_} T={t} {h}x{w} {c}->{cout} res={res}: " + "  ".join(row), flush=True)


def perf_conv(n, h, w, cin, cout, res=True, tiles=(0, 8, 12, 13)):
    """Conv2d 3x3 stride 1, cold operands, + row bias + residual: TF/s per block shape (0 = auto, 8 = LDS-halo kernel)."""
    pw = pack_weight(torch.randn(cout, cin, 3, 3) * (9 * cin) ** -0.5, torch.randn(cout)).to(dev)
    NB = 4
    src = [torch.randn(n * h * w, cin, device=dev).to(BF) for _ in range(NB)]
    act = [torch.empty_like(x) for x in src]
    rs = [torch.randn(n * h * w, cout, device=dev).to(BF) for _ in range(NB)]
    gb = torch.randn(n, cout, device=dev)
    g1, b1 = torch.ones(cin, device=dev), torch.zeros(cin, device=dev)
    fl = 2.0 * n * h * w * cout * cin * 9
    row = []
    for tl in tiles:
        try:
            evs = []
            for rep in range(3):
                for i in range(NB):
                    hip_ln(src[i], act[i], g1, b1)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    ops.conv2d(act[i].view(n, h, w, cin), pw, res1=rs[i] if res else None, group_bias=gb, group_rows=h * w, gn=(h * w) % 512 == 0, tile=tl)
                    e1.record()
                    if rep:
                        evs.append((e0, e1))
            torch.cuda.synchronize()
            row.append(f"t{tl} {fl / (sum(x.elapsed_time(y) for x, y in evs) / len(evs)) / 1e9:5.0f}")
        except Exception as e:
            row.append(f"t{tl} err")
    print(f"conv3x3 n={n} {h}x{w} {cin}->{cout}: " + "  ".join(row), flush=True)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    ok = True
    if what in ("all", "check"):
        for (m, n, k) in [(256, 256, 128), (512, 512, 192), (300, 320, 320), (1000, 640, 640), (257, 1280, 2560), (6528, 1280, 1280),
                          (4096, 2048, 1024), (13056, 1280, 5120)]:
            ok &= check(m, n, k)
        ok &= check(777, 640, 640, res=1)
        ok &= check(2048, 1280, 1280, res=2)
        ok &= check(1000, 320, 128)
        ok &= check(3000, 5120, 640, geglu=True)
        ok &= check(6528, 10240, 1280, geglu=True)
        ok &= check(200, 2560, 320, geglu=True)
        ok &= check(26112, 640, 2560, res=1, reps=8)
        ok &= check(52224, 5120, 640, geglu=True, reps=8)
        for t in (12, 13):          # both block shapes on shapes that are not their natural ones
            ok &= check(1000, 640, 640, res=1, T11=t)
            ok &= check(700, 384, 192, T11=t)
            ok &= check(3000, 5120, 640, geglu=True, T11=t)
            ok &= check(26112, 640, 2560, res=2, T11=t)
        print("CHECK", "PASSED" if ok else "FAILED", flush=True)
    if what in ("all", "conv"):
        perf_conv(34, 32, 48, 640, 640)
        perf_conv(34, 32, 48, 1280, 640)
        perf_conv(34, 32, 48, 320, 640)
        perf_conv(34, 16, 24, 1280, 1280)
        perf_conv(34, 16, 24, 2560, 1280)
        perf_conv(34, 16, 24, 640, 1280)
        perf_conv(34, 64, 96, 320, 320)
        perf_conv(34, 64, 96, 640, 320)
        perf_conv(17, 32, 48, 640, 640)
        perf_conv(17, 16, 24, 1280, 1280)
        perf_conv(34, 8, 12, 1280, 1280)
    if what in ("all", "temporal"):
        perf_temporal(2, 17, 64, 96, 320, 320)
        perf_temporal(1, 17, 64, 96, 320, 320)
        perf_temporal(2, 17, 32, 48, 640, 640)
        perf_temporal(1, 17, 32, 48, 640, 640)
        perf_temporal(2, 17, 16, 24, 1280, 1280)
        perf_temporal(1, 17, 16, 24, 1280, 1280)
        perf_temporal(2, 17, 8, 12, 1280, 1280)
        perf_temporal(2, 17, 64, 96, 640, 640)
        perf_temporal(2, 17, 32, 48, 1280, 1280)
    if what in ("all", "perf"):
        perf(8192, 8192, 8192, tiles=(4, 11))
        perf(52224, 5120, 640, tiles=(1, 11))
        perf(52224, 5120, 640, geglu=True, tiles=(2, 11))
        perf(13056, 10240, 1280, tiles=(6, 11))
        perf(13056, 10240, 1280, geglu=True, tiles=(6, 11))
        perf(52224, 640, 2560, res=True, tiles=(1, 12, 13))
        perf(26112, 640, 2560, res=True, tiles=(1, 13))
        perf(13056, 1280, 5120, res=True, tiles=(4, 11))
        perf(13056, 1280, 1280, res=True, tiles=(4, 11))
        perf(13056, 1280, 1280, tiles=(4, 11))
        perf(52224, 640, 640, res=True, tiles=(1, 13))
        perf(26112, 5120, 640, geglu=True, tiles=(2, 11))
        perf(6528, 10240, 1280, geglu=True, tiles=(6, 11))
        perf(6528, 1280, 5120, res=True, tiles=(1, 11))
