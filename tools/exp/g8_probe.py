"""Per-workgroup cycle stamps of the persistent eight-phase GEMM (needs the -DG8_PROBE build: tools/exp/build_g8_probe.sh, loaded
through CCEDIT_HIP_LIB).  Stamps per workgroup: kernel entry, then per output tile: first K tile landed / K loop done / epilogue
done.  s_memtime ticks at 100 MHz; durations are printed in microseconds."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import ctypes as C
import torch
from ccedit_amd import ops, hip
from ccedit_amd.hip import CcGemmDesc
from ccedit_amd.packing import pack_weight

BF = torch.bfloat16


def run(m, n, k, geglu=False, res=False, tile=11):
    a = torch.randn(m, k, device="cuda", dtype=BF)
    pw = pack_weight(torch.randn(n, k) * k ** -0.5, torch.randn(n), geglu=geglu).to("cuda")
    r = torch.randn(m, n, device="cuda", dtype=BF) if res else None
    out = torch.empty(m, n // 2 if geglu else n, device="cuda", dtype=BF)
    probe = torch.zeros(256 * 64, dtype=torch.int64, device="cuda")
    d = CcGemmDesc()
    d.M, d.N, d.Cin, d.Cin1, d.taps, d.mode = m, pw.n, pw.cin, pw.cin, 1, 0
    d.lda, d.ldc, d.Kpad = k, out.stride(0), pw.kpad
    d.act = 2 if geglu else 0
    d.ldr1 = n if res else 0
    d.tile = tile
    d.A, d.W, d.bias, d.out = a.data_ptr(), pw.w.data_ptr(), pw.bias.data_ptr(), out.data_ptr()
    d.res1 = r.data_ptr() if res else None
    for rep in range(3):
        probe.zero_()
        d.gn_stats = probe.data_ptr()
        hip.check(hip.lib().ccedit_gemm(C.byref(d), torch.cuda.current_stream().cuda_stream), "gemm")
        torch.cuda.synchronize()
    p = probe.view(256, 64).cpu()
    t0 = p[:, 0].min()
    nst = (p > 0).sum(1)
    us = lambda x: float(x) / 100.0
    print(f"M={m} N={n} K={k} geglu={int(geglu)} res={int(res)}: stamps/wg {int(nst.min())}..{int(nst.max())}, kernel {us(p.max() - t0):.1f} us")
    # per-tile phases for the median workgroup
    for wg in (0, 100, 255):
        row = p[wg]
        ns = int(nst[wg])
        segs = []
        for i in range(1, ns, 3):
            if i + 2 < ns + 1:
                landed, kdone = us(row[i] - row[i - 1]), us(row[i + 1] - row[i])
                epi = us(row[i + 2] - row[i + 1]) if i + 2 < ns else float("nan")
                segs.append(f"[wait {landed:.1f} k {kdone:.1f} epi {epi:.1f}]")
        print(f"  wg{wg}: start +{us(row[0] - t0):.1f}  " + " ".join(segs[:6]) + (" ..." if len(segs) > 6 else ""))
    # averages over all workgroups and tiles
    w, kk, ee, cnt = 0.0, 0.0, 0.0, 0
    for wg in range(256):
        row, ns = p[wg], int(nst[wg])
        for i in range(1, ns - 2, 3):
            w += us(row[i] - row[i - 1]); kk += us(row[i + 1] - row[i]); ee += us(row[i + 2] - row[i + 1]); cnt += 1
    if cnt:
        print(f"  mean per tile over {cnt} tiles: wait-for-operands {w / cnt:.2f} us, K loop {kk / cnt:.2f} us, epilogue {ee / cnt:.2f} us")


if __name__ == "__main__":
    if len(sys.argv) > 1:            # M,N,K[,res][,geglu][,tile=T] ...
        for spec in sys.argv[1:]:
            f = spec.split(",")
            kw = dict(res="res" in f, geglu="geglu" in f)
            kw.update({"tile": int(x[5:]) for x in f if x.startswith("tile=")})
            run(int(f[0]), int(f[1]), int(f[2]), **kw)
        sys.exit(0)
    run(52224, 5120, 640)
    run(52224, 5120, 640, geglu=True)
    run(13056, 10240, 1280, geglu=True)
    run(13056, 1280, 1280, res=True)
    run(13056, 1280, 5120, res=True)
    run(52224, 640, 2560, res=True)
    run(8192, 8192, 8192)
