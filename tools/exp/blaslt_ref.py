"""Calibration only (never on the product path): what the vendor GEMM (torch.matmul -> hipBLASLt / rocBLAS) reaches on this
MI355X for plain bf16 GEMMs of the network's shapes — the practical ceiling the hand-written kernels are compared with."""
import torch

shapes = [(8192, 8192, 8192), (16384, 8192, 4096), (208896, 320, 2880), (208896, 320, 320), (208896, 2560, 320), (208896, 320, 1280),
          (52224, 5120, 640), (52224, 640, 2560), (52224, 640, 5760), (13056, 10240, 1280), (13056, 1280, 11520), (13056, 1280, 1280)]
for m, n, k in shapes:
    a = torch.randn(m, k, device="cuda", dtype=torch.bfloat16)
    b = torch.randn(n, k, device="cuda", dtype=torch.bfloat16)
    for _ in range(3):
        c = a @ b.t()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps):
        c = a @ b.t()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"M={m:7d} N={n:6d} K={k:6d}: {ms * 1e3:9.1f} us  {2.0 * m * n * k / ms / 1e9:8.1f} TF/s", flush=True)
