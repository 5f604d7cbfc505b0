import sys, os, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import torch
from ccedit_amd.sgm_compat import build_vae
from ccedit_amd.utils.synth import fill_module_
dev = torch.device("cuda")
torch.set_grad_enabled(False)
vae = build_vae(dev)
fill_module_(vae, prefix="first_stage_model.")
vae.pack(dev)
z = torch.randn(1, 4, 17, 64, 96, device=dev)
for i in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    y = vae.decode(z)
    torch.cuda.synchronize(); print(f"decode {i}: {time.perf_counter() - t0:.3f} s", flush=True)
