"""E2: does the UNet running on one stream disturb an UNRELATED consumer chain on another stream?  Side stream: `rounds` times
(producer kernel writes a fresh buffer; consumer kernel reads it) with fixed inputs — every round must give the same checksums.
PRODUCER / CONSUMER: 'hip' = our g8 Linear -> our LayerNorm (the pair where the network first diverged), 'torch' = ATen matmul -> layer_norm."""
import os, sys, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import bench
from ccedit_amd import ops
from ccedit_amd.packing import pack_weight
KIND = os.environ.get("KIND", "hip")
MAIN = os.environ.get("MAIN", "unet")          # what runs on the main stream meanwhile: unet | none | matmul
dev = torch.device("cuda")
os.environ.setdefault("CCEDIT_OVERLAP_CONTROLNET", "0")
w = bench.build_model(dev)
x, cc, cu, hint = bench.synth_inputs(dev)
x2 = torch.cat([x, x]).contiguous()
cond = dict(crossattn=torch.cat([cu, cc]).contiguous(), control_hint=torch.cat([hint, hint]).contiguous())
t = torch.tensor([601, 601], dtype=torch.int64, device=dev)
w(x2, t, cond)
g = torch.Generator(device=dev).manual_seed(5)
m, k, n = 52224, 640, 640
a = torch.randn(m, k, device=dev, generator=g).to(torch.bfloat16)
wt = (torch.randn(n, k, device=dev, generator=g) * k ** -0.5)
pw = pack_weight(wt.cpu(), torch.randn(n)).to(dev)
wb = wt.to(torch.bfloat16)
gam, bet = torch.ones(n, device=dev), torch.zeros(n, device=dev)
big = torch.randn(8192, 8192, device=dev).to(torch.bfloat16)
side = torch.cuda.Stream()
def csum(y):
    v = y.reshape(-1).view(torch.int16).to(torch.int64)
    return (v * (torch.arange(v.numel(), device=v.device) % 8191 + 1)).sum()
def chain(rounds):
    out = []
    for _ in range(rounds):
        prod, cons = KIND.split("+") if "+" in KIND else (KIND, KIND)
        if prod == "hip":
            p = ops.linear(a, pw, tile=11)
        elif prod == "hip1":
            p = ops.linear(a, pw, tile=1)
        else:
            p = torch.matmul(a, wb.t())
        if cons == "hip":
            c = ops.layernorm(p, gam, bet, 1e-5)
        elif cons == "copy":
            c = p.clone()
        else:
            c = torch.nn.functional.layer_norm(p, (n,), gam.to(torch.bfloat16), bet.to(torch.bfloat16))
        out.append((csum(p), csum(c)))
        del p, c
    return out
for trial in range(4):
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        res = chain(int(os.environ.get("ROUNDS", "40")))
    if MAIN == "unet":
        w(x2, t, cond)
    elif MAIN == "matmul":
        for _ in range(30):
            torch.matmul(big, big)
    torch.cuda.synchronize()
    ps = {int(p) for p, _ in res}
    cs = {int(c) for _, c in res}
    print(f"KIND={KIND} MAIN={MAIN} trial {trial}: distinct producer checksums {len(ps)}, distinct consumer checksums {len(cs)}", flush=True)
