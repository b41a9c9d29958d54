import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import needle_amd as na
from needle_amd.sharded import hip_local_search
c = na.Context(0)
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
n, dim = 7_000_000, 384
base = torch.randn((n, dim), generator=g, device=dev)
q = torch.randn((1000, dim), generator=g, device=dev)
ix = na.FlatIndex(c, base, na.DistanceFunction.Euclidean)
os.environ["NEEDLE_HIP_DEBUG"] = "8"
res = {}
for prec in ("bf16", "fp32"):
    c.reset_metrics()
    ids, d, cnt = hip_local_search(ix, prec)(q, 10)
    torch.cuda.synchronize()
    res[prec] = (ids.cpu().numpy(), d.cpu().numpy())
    print(prec, {k: v for k, v in c.metrics().items() if "queries" in k}, res[prec][0][0, :3], flush=True)
print("equal", np.array_equal(res["bf16"][0], res["fp32"][0]), np.array_equal(res["bf16"][1], res["fp32"][1]))
