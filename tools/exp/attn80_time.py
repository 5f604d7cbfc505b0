"""d = 80 spatial self-attention (32 x 48 level: 34 frames x 8 heads x 1536 tokens): general flash kernel vs attn_spatial (policy attn_spatial: 2 = general kernel for d = 80, 1 = the d = 80 instantiation)."""
import sys, os
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import torch
from ccedit_amd import hip, ops

frames, heads, d, L = 34, 8, 80, 1536
c = heads * d
NB = 3
g = torch.Generator().manual_seed(0)
qkv = [torch.randn(frames * L, 3 * c, generator=g).to(torch.bfloat16).cuda() for _ in range(NB)]
lib = hip.lib()


def run(i):
    t = qkv[i % NB]
    return ops.attention(t[:, :c], t[:, c:2 * c], t[:, 2 * c:], heads, d, batches=frames, lq=L, lk=L, q_log2=True)


def timeit(reps=20):
    for i in range(3):
        run(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        run(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


flops = 4.0 * frames * heads * L * L * d
outs = {}
for pol in (2, 1):
    assert lib.ccedit_policy_set(b"attn_spatial", pol) == 0
    outs[pol] = run(0).float()
    k = lib.ccedit_last_kernel().decode()
    us = timeit()
    print(f"attn_spatial={pol}: {k:32s} {us:7.1f} us  {flops / us / 1e6:6.0f} TF/s", flush=True)
ref = outs[2]
dlt = (outs[1] - ref)
print(f"new vs general kernel: max abs {dlt.abs().max().item():.3e}, rel rms {(dlt.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item():.3e}")
# fp32 reference on a slice (frame 0, head 0..7) for both
t = qkv[0][:L].float()
q, k_, v = (t[:, i * c:(i + 1) * c].view(L, heads, d).permute(1, 0, 2) for i in range(3))
p = torch.softmax((q @ k_.transpose(1, 2)) * 0.6931471805599453, dim=-1)          # q is in log2 units: exp2(s) = exp(s ln 2)
want = (p @ v).permute(1, 0, 2).reshape(L, c)
for pol in (1, 2):
    e = (outs[pol][:L] - want)
    print(f"policy {pol} vs fp32 reference (frame 0): rel rms {(e.pow(2).mean().sqrt() / want.pow(2).mean().sqrt()).item():.3e}")
