"""attn_text_kernel on the three text cross-attention shapes of the step (cold-ish: rotating q buffers)."""
import os, sys, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
from ccedit_amd import ops, hip
BF = torch.bfloat16
def run(heads, d, lq, fpc=17, clips=2, lk=77):
    c = heads * d
    n = clips * fpc
    NB = 4
    qs = [torch.randn(n * lq, c, device="cuda").to(BF) for _ in range(NB)]
    kv = torch.randn(clips * lk, 2 * c, device="cuda").to(BF)
    evs = []
    for rep in range(4):
        for i in range(NB):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); ops.attention(qs[i], kv[:, :c], kv[:, c:], heads, d, batches=n, lq=lq, lk=lk, kv_div=fpc); e1.record()
            if rep: evs.append((e0, e1))
    torch.cuda.synchronize()
    ms = sum(a.elapsed_time(b) for a, b in evs) / len(evs)
    print(f"{os.environ.get('TAG','')} d={d} Lq={lq}: {ms * 1e3:6.1f} us  {4.0 * n * lq * c * 1e-3 / ms / 1e6:5.2f} TB/s q+o  [{hip.lib().ccedit_last_kernel().decode()}]", flush=True)
run(8, 40, 6144); run(8, 80, 1536); run(8, 160, 384)
