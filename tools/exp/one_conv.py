import sys, os
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import torch
from ccedit_amd import ops, hip
from ccedit_amd.packing import pack_weight
tile = int(os.environ.get("TILE", "0"))
n, h, w, cin, cout = 34, int(os.environ.get("HH", "64")), int(os.environ.get("WW", "96")), int(os.environ.get("CIN", "320")), int(os.environ.get("COUT", "320"))
pw = pack_weight(torch.randn(cout, cin, 3, 3) * (9 * cin) ** -0.5, torch.randn(cout)).to("cuda")
NB = 4
a = [torch.randn(n, h, w, cin, device="cuda").to(torch.bfloat16) for _ in range(NB)]
for rep in range(3):
    for i in range(NB):
        ops.conv2d(a[i], pw, tile=tile)
torch.cuda.synchronize()
