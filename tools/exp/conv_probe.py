"""Per-workgroup timeline of conv_halo_kernel (probe build: tools/exp/build_conv_probe.sh, CCEDIT_HIP_LIB=build_var/libccedit_probe.so).
Stamps (s_memrealtime, 100 MHz): 0 start, 1 first operands landed, 2 K loop done, 3 epilogue done; 4 = (xcc, hw_id), 5 = first channel."""
import sys, os, ctypes
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import numpy as np
import torch
from ccedit_amd import hip, ops
from ccedit_amd.packing import pack_weight
lib = hip.lib()
lib.ccedit_conv_probe_read.restype = ctypes.c_int
lib.ccedit_conv_probe_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
NS = 8
for n, h, w, cin, cout in ((34, 64, 96, 320, 320), (34, 64, 96, 640, 320), (34, 32, 48, 640, 640), (34, 32, 48, 1280, 640)):
    pw = pack_weight(torch.randn(cout, cin, 3, 3) * (9 * cin) ** -0.5, torch.randn(cout)).to("cuda")
    xs = [torch.randn(n, h, w, cin, device="cuda").to(torch.bfloat16) for _ in range(3)]
    for x in xs:
        ops.conv2d(x, pw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    y = ops.conv2d(xs[0], pw)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3
    tiles = n * (h // 8) * (w // 16)
    ct = (cout + 127) // 128
    nwg = min(8 * ((tiles + 7) // 8) * ct, 8192)
    buf = np.zeros(nwg * NS, dtype=np.uint64)
    rc = lib.ccedit_conv_probe_read(buf.ctypes.data, buf.size)
    assert rc == 0, rc
    st = buf.reshape(nwg, NS)
    ok = st[:, 3] > 0
    st = st[ok]
    t0 = st[:, 0].min()
    tick = 0.01       # us per tick (100 MHz)
    start, pro, loop, epi = ((st[:, 0] - t0) * tick, (st[:, 1] - st[:, 0]) * tick, (st[:, 2] - st[:, 1]) * tick, (st[:, 3] - st[:, 2]) * tick)
    end = (st[:, 3] - t0) * tick
    narrow = (cout - st[:, 5].astype(np.int64)) <= 64
    nk = cin // 64 * 9
    print(f"{n}x{h}x{w} {cin}->{cout}: launch {us:.1f} us ({2.0 * n * h * w * cin * 9 * cout / us / 1e6:.0f} TF/s), {len(st)} workgroups, span {end.max():.1f} us")
    for name, sel in (("full tiles", ~narrow), ("narrow tiles", narrow)):
        if sel.sum() == 0:
            continue
        print(f"  {name:12s} n={sel.sum():5d}  prologue {pro[sel].mean():6.2f}  K loop {loop[sel].mean():6.2f} ({loop[sel].mean() / nk:.3f} / k-tile; p10 {np.percentile(loop[sel], 10):.2f} p90 {np.percentile(loop[sel], 90):.2f})"
              f"  epilogue {epi[sel].mean():5.2f}  total {(pro + loop + epi)[sel].mean():6.2f} us")
    # occupancy over time: how many workgroups are resident in each 5 % slice of the span
    edges = np.linspace(0, end.max(), 21)
    occ = [int(((start < b) & (end > a)).sum()) for a, b in zip(edges[:-1], edges[1:])]
    print("  resident workgroups per 5 % of the span:", occ)
    busy = (pro + loop + epi).sum() / (512 * end.max())
    print(f"  slot utilisation (sum of workgroup times / 512 slots x span): {busy:.3f};  last start {start.max():.1f} us; workgroups starting after 90 % of the span: {(start > 0.9 * end.max()).sum()}")
    xcc = (st[:, 4] >> np.uint64(32)).astype(np.int64)
    print("  workgroups per XCC:", np.bincount(xcc, minlength=8).tolist(), " mean end per XCC:", [round(float(end[xcc == i].max()), 1) for i in range(8)])
