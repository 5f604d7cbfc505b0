"""E4: our LayerNorm on a side stream, FIXED input, many rounds into separate outputs, while one kind of kernel loops on the main
stream.  Counts the rounds whose output differs from the idle result.  LOADS = comma list of: none,unet,g8,t1,lin320,conv,attn,ln,gn,torch"""
import os, sys, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
from ccedit_amd import ops
from ccedit_amd.packing import pack_weight
dev = torch.device("cuda")
BF = torch.bfloat16
g = torch.Generator(device=dev).manual_seed(5)
m, n = 52224, 640
src = torch.randn(m, n, device=dev, generator=g).to(BF)
gam, bet = torch.ones(n, device=dev), torch.zeros(n, device=dev)
ROUNDS = int(os.environ.get("ROUNDS", "150"))
CONS = os.environ.get("CONS", "ln")
def consumer(out=None):
    if CONS == "ln":
        return ops.layernorm(src, gam, bet, 1e-5)
    if CONS == "silu":
        return ops.silu(src)
    if CONS == "gn":
        return ops.groupnorm_spatial(src.view(34, 32, 48, 640), gam, bet, 1e-5, False)
    return torch.nn.functional.layer_norm(src, (n,), gam.to(BF), bet.to(BF))
ref = consumer()
torch.cuda.synchronize()
side = torch.cuda.Stream()

a640 = torch.randn(52224, 640, device=dev, generator=g).to(BF)
pw_ff = pack_weight(torch.randn(5120, 640) * 0.04, torch.randn(5120)).to(dev)
a320 = torch.randn(104448, 320, device=dev, generator=g).to(BF)
pw_320 = pack_weight(torch.randn(320, 320) * 0.05, torch.randn(320)).to(dev)
xc = torch.randn(34, 64, 96, 320, device=dev, generator=g).to(BF)
pw_c = pack_weight(torch.randn(320, 320, 3, 3) * 0.02, torch.randn(320)).to(dev)
qkv = torch.randn(2 * 6144, 960, device=dev, generator=g).to(BF)
big = torch.randn(8192, 8192, device=dev, generator=g).to(BF)
unet = None
def load(kind):
    global unet
    if kind == "none":
        return
    if kind == "unet":
        if unet is None:
            import bench
            os.environ.setdefault("CCEDIT_OVERLAP_CONTROLNET", "0")
            w = bench.build_model(dev)
            x, cc, cu, hint = bench.synth_inputs(dev)
            unet = (w, torch.cat([x, x]).contiguous(), torch.tensor([601, 601], dtype=torch.int64, device=dev),
                    dict(crossattn=torch.cat([cu, cc]).contiguous(), control_hint=torch.cat([hint, hint]).contiguous()))
        unet[0](unet[1], unet[2], unet[3])
        return
    for _ in range(int(os.environ.get("REPS", "40"))):
        if kind == "g8":
            ops.linear(a640, pw_ff, tile=11)
        elif kind[0] == "t" and kind[1:].isdigit():
            ops.linear(a640, pw_ff, tile=int(kind[1:]))
        elif kind == "lin320":
            ops.linear(a320, pw_320)
        elif kind == "conv":
            ops.conv2d(xc, pw_c)
        elif kind == "attn":
            ops.attention(qkv[:, :320], qkv[:, 320:640], qkv[:, 640:], 8, 40, batches=2, lq=6144, lk=6144)
        elif kind == "ln":
            ops.layernorm(a640, gam, bet, 1e-5)
        elif kind == "gn":
            ops.groupnorm_spatial(xc, gam[:320].contiguous(), bet[:320].contiguous(), 1e-5, True)
        elif kind == "torch":
            torch.matmul(big, big)
for kind in os.environ.get("LOADS", "none,unet,g8,t1,lin320,conv,attn,ln,gn,torch").split(","):
    load(kind)
    torch.cuda.synchronize()
    tot = 0
    for trial in range(int(os.environ.get("TRIALS", "3"))):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            cs = [consumer() for _ in range(ROUNDS)]
        load(kind)
        torch.cuda.synchronize()
        bad = [r for r in range(ROUNDS) if not torch.equal(cs[r], ref)]
        tot += len(bad)
        for r in bad[:2]:
            d = cs[r].view(torch.int16) != ref.view(torch.int16)
            rows = d.any(dim=1).nonzero().flatten()
            r0 = int(rows[0])
            cols = d[r0].nonzero().flatten().tolist()
            xs = src[r0 - 1:r0 + 1].float()
            mm = xs.mean(dim=1)
            rs = (xs.var(dim=1, unbiased=False) + 1e-5).rsqrt()
            off = (cs[r][r0, cols].float() - ref[r0, cols].float()).mean()
            print(f"   {kind} trial {trial} round {r}: {int(d.sum())} values in rows {rows[:6].tolist()}, cols {cols[:3]}..{cols[-1]}; "
                  f"offset {float(off):+.4f}; (m1-m0)*r1 {float((mm[1] - mm[0]) * rs[1]):+.4f}; m1*r1 {float(mm[1] * rs[1]):+.4f}", flush=True)
        del cs
    print(f"CONS={CONS} main={kind}: {tot} bad rounds of {ROUNDS * int(os.environ.get('TRIALS', '3'))}", flush=True)
