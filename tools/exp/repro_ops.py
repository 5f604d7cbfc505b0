"""Which OP's output first differs between two evaluations with the ControlNet on its side stream?  Every ops.* result gets a
checksum (on the stream it was produced on); the per-stream checksum sequences of two runs are compared."""
import os, sys, torch, collections
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import bench
from ccedit_amd import ops, hip
dev = torch.device("cuda")
w = bench.build_model(dev)
x, cc, cu, hint = bench.synth_inputs(dev)
x2 = torch.cat([x, x]).contiguous()
cond = dict(crossattn=torch.cat([cu, cc]).contiguous(), control_hint=torch.cat([hint, hint]).contiguous())
t = torch.tensor([601, 601], dtype=torch.int64, device=dev)
w(x2, t, cond)
LOG = None
def wrap(name):
    orig = getattr(ops, name)
    def f(*a, **k):
        y = orig(*a, **k)
        if LOG is not None and torch.is_tensor(y):
            sid = torch.cuda.current_stream().cuda_stream
            kname = hip.lib().ccedit_last_kernel().decode() if name in ("gemm", "attention", "ff320") else ""
            v = y.reshape(-1).view(torch.int16 if y.dtype == torch.bfloat16 else torch.int32).to(torch.int64)
            LOG[sid].append((name, kname, tuple(y.shape), (v * (torch.arange(v.numel(), device=v.device) % 8191 + 1)).sum()))
        return y
    setattr(ops, name, f)
for n in ("gemm", "attention", "ff320", "layernorm", "groupnorm_spatial", "groupnorm_temporal", "cat_add", "add", "silu"):
    wrap(n)
ops.linear = lambda x2d, pw, **kw: ops.gemm(x2d, pw, mode=0, **kw)
def run():
    global LOG
    LOG = collections.defaultdict(list)
    y = w(x2, t, cond)
    torch.cuda.synchronize()
    out, LOG = LOG, None
    return y, {k: [(a, b, c, int(d)) for a, b, c, d in v] for k, v in out.items()}
y0, l0 = run()
for rep in range(5):
    y1, l1 = run()
    msg = []
    for sid in l0:
        for i, (p, q) in enumerate(zip(l0[sid], l1[sid])):
            if p != q:
                msg.append(f"stream {sid & 0xffff:x}: op #{i} {p[0]} [{p[1]}] {p[2]} (previous op: {l0[sid][i-1][:3]})")
                break
    print(rep, "eps equal" if torch.equal(y0, y1) else "eps DIFFERS", "|", "; ".join(msg) or "all op checksums equal", flush=True)
