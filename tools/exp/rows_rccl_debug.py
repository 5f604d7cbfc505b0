"""World-1 RCCL run of the row-sharded evaluation, stage by stage (where does it stop?)."""
import faulthandler, os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
faulthandler.dump_traceback_later(100, exit=True)
import torch
import torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
torch.set_grad_enabled(False)
from ccedit_amd.parallel import RowShard
from ccedit_amd.sgm_compat import build_network
from ccedit_amd.utils.synth import fill_module_
G = dict(model_channels=64, num_heads=2, context_dim=64)
T, H, W = 5, 16, 16
cross = len(sys.argv) > 1 and sys.argv[1] == "cross"
w = build_network("cpu", **(dict(G, crossframe=True) if cross else G))
fill_module_(w, prefix="model.")
w.diffusion_model.pack("cuda")
g = torch.Generator().manual_seed(21)
x = torch.randn(1, 4, T, H, W, generator=g)
x2 = torch.cat([x, x]).cuda()
c = dict(crossattn=torch.randn(2, 77, 64, generator=g).cuda(), control_hint=(torch.rand(1, 3, T, 8 * H, 8 * W, generator=g) * 2 - 1).repeat(2, 1, 1, 1, 1).cuda())
if cross:
    c["cond_feat"] = (0.18215 * torch.randn(1, 4, H, W, generator=g)).repeat(2, 1, 1, 1).cuda()
t = torch.tensor([501, 501], dtype=torch.int64).cuda()


def say(*a):
    print(*a, flush=True)


ref = w(x2, t, c).clone(); torch.cuda.synchronize(); say("unsharded ok")
w.reset_caches()
combos = [(a, g_, o) for o in (False, True) for g_ in (False, True) for a in ("heads", "gather")]
for attn, graph, overlap in combos:
    rs = RowShard(attn=attn)
    w.row_shard = rs
    if True:
        w.use_graph, w.overlap_controlnet = graph, overlap
        w.reset_caches()
        outs = []
        for i in range(3):
            t0 = time.time()
            outs.append(w(x2, t, c).clone()); torch.cuda.synchronize()
            say(f"attn={attn} graph={graph} overlap={overlap} call {i}: {time.time() - t0:.2f}s  rel vs unsharded "
                f"{((outs[-1] - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item():.5f}  equal to call 0: {torch.equal(outs[0], outs[-1])}")
say("done")
dist.destroy_process_group()
