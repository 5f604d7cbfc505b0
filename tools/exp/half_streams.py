"""Would the DECODER of the UNet gain from evaluating the two CFG halves as B = 1 chains on two streams (the encoder has the ControlNet beside
it on a side stream; the decoder runs alone)?  One decoder block at a time, full size: the batched call against two half-batch calls on
two streams, eager launches, wall clock over 10 repetitions."""
import sys, os, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import torch
import bench
from ccedit_amd.network import Geometry
dev = torch.device("cuda:0")
torch.set_grad_enabled(False)
w = bench.build_model(dev)
net = w.diffusion_model
T = 17
t2 = torch.tensor([601, 601], dtype=torch.int64, device=dev)
ctx = torch.randn(2 * 77, 768, device=dev).to(torch.bfloat16)
emb2, kv2 = net._emb_silu(t2), net.text_kv(ctx)
embh = [net._emb_silu(t2[i:i + 1]) for i in range(2)]
kvh = [net.text_kv(ctx[77 * i:77 * (i + 1)].contiguous()) for i in range(2)]
g2, g1 = Geometry(2, T), Geometry(1, T)
s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
res = {32: (32, 48), 64: (64, 96)}
for i, (hh, ww) in ((6, (32, 48)), (7, (32, 48)), (9, (64, 96)), (10, (64, 96)), (11, (64, 96))):
    blk = net.output_blocks[i]
    cin = blk[0].in_layers[0].weight.numel()
    h = (torch.randn(2 * T, hh, ww, cin, device=dev) * 0.5).to(torch.bfloat16)
    halves = [h[:T].contiguous(), h[T:].contiguous()]
    def batched():
        return blk.run(h, emb2, g2, kv2, 77)
    def split():
        main = torch.cuda.current_stream()
        outs = []
        for st, hx, e, k in ((s0, halves[0], embh[0], kvh[0]), (s1, halves[1], embh[1], kvh[1])):
            st.wait_stream(main)
            with torch.cuda.stream(st):
                outs.append(blk.run(hx, e, g1, k, 77))
        main.wait_stream(s0); main.wait_stream(s1)
        return outs
    def serial():
        return [blk.run(halves[0], embh[0], g1, kvh[0], 77), blk.run(halves[1], embh[1], g1, kvh[1], 77)]
    def timed(f, n=10):
        f(); f(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n): f()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    tb, ts, tq = timed(batched), timed(split), timed(serial)
    print(f"output_blocks.{i} ({cin} ch in, {hh}x{ww}): batched {tb:.3f} ms, halves on two streams {ts:.3f} ms, halves one after the other {tq:.3f} ms", flush=True)
