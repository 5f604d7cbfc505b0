"""one 3x3 conv of the fp32 first stage at full size (17 x 512 x 768, 128 -> 128) on the library CCEDIT_HIP_LIB names"""
import sys, os, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import torch
from ccedit_amd import vae_f32 as V
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
for (cin, cout, h, w) in [(128, 128, 512, 768), (512, 512, 128, 192)]:
    x = torch.randn(17, h, w, cin, generator=g).to(dev)
    pw = V.pack_f32(torch.randn(cout, cin, 3, 3, generator=g) * (9 * cin) ** -0.5, torch.randn(cout, generator=g), dev)
    V.conv2d_f32(x, pw); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); V.conv2d_f32(x, pw); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    fl = 2.0 * 17 * h * w * cin * cout * 9
    print(f"{sys.argv[1]}: {cin}->{cout} {h}x{w}: {min(ts) * 1e6:8.0f} us  {fl / min(ts) / 1e12:6.1f} TF/s eq", flush=True)
    del x
