import ctypes, os, subprocess, sys, torch
here = os.path.dirname(os.path.abspath(__file__))
co = os.path.join(here, "readpat.co")
hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
mod = ctypes.c_void_p()
assert hip.hipModuleLoad(ctypes.byref(mod), co.encode()) == 0
M = 208896
bufs = [torch.randn(M, 320, device="cuda").to(torch.bfloat16) for _ in range(8)]
outs = [torch.empty(M, 320, device="cuda", dtype=torch.bfloat16) for _ in range(8)]
wbuf = torch.randn(512, 320, device="cuda").to(torch.bfloat16)
def run(name, rows_per_block, lds, mult=1, usew=False, useout=False):
    f = ctypes.c_void_p()
    assert hip.hipModuleGetFunction(ctypes.byref(f), mod, name.encode()) == 0
    hip.hipFuncSetAttribute  # dynamic LDS > 48K handled by hipModuleLaunchKernel's sharedMemBytes on ROCm
    grid = ((M + rows_per_block - 1) // rows_per_block) * mult
    if mult > 1: grid = 8 * ((((M + 255) // 256) + 7) // 8) * mult
    ts = []
    for it in range(3):
        for bi, b in enumerate(bufs):
            a = ctypes.c_void_p(b.data_ptr()); m = ctypes.c_int64(M); s = ctypes.c_void_p(outs[bi].data_ptr() if useout else (wbuf.data_ptr() if usew else 0))
            args = (ctypes.c_void_p * 3)(ctypes.cast(ctypes.byref(a), ctypes.c_void_p), ctypes.cast(ctypes.byref(m), ctypes.c_void_p), ctypes.cast(ctypes.byref(s), ctypes.c_void_p))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = hip.hipModuleLaunchKernel(f, grid, 1, 1, 256, 1, 1, lds, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), args, None)
            assert rc == 0, rc
            e1.record(); ts.append((e0, e1))
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ts[8:])
    t = t[len(t) // 2] * 1e-3
    print(f"{name}: {t*1e6:.1f} us  {M*640/t/1e12:.2f} TB/s")
run("pat0", 256, 65536); run("pat1", 64, 40960); run("pat2", 64, 40960); run("pat3", 256, 65536, mult=5); run("pat4", 256, 81920, usew=True); run("pat5", 128, 32768, useout=True); run("pat6", 256, 65536, mult=5); run("pat7", 256, 0, mult=5)
run("pat9", 256, 65536, mult=5); run("pat10", 256, 65536)
x = bufs[0]; y = torch.empty_like(x)
for _ in range(2): y.copy_(x)
torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for b in bufs: y.copy_(b)
e1.record(); torch.cuda.synchronize(); t = e0.elapsed_time(e1) / 8 * 1e-3
print(f"torch copy: {t*1e6:.1f} us  {2*M*640/t/1e12:.2f} TB/s (R+W)")
