import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import needle_amd as na
from needle_amd.sharded import hip_local_search
c = na.Context(0)
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
base = torch.randn((1_000_000, 384), generator=g, device=dev)
for m in (na.DistanceFunction.Cosine, na.DistanceFunction.Manhattan):
    ix = na.FlatIndex(c, base, m)
    for nq in (4, 64):
        q = torch.randn((nq, 384), generator=g, device=dev)
        f = hip_local_search(ix, "exact")
        f(q, 10); torch.cuda.synchronize()
        t0 = time.perf_counter(); f(q, 10); torch.cuda.synchronize()
        print(m.name, nq, "queries exact scan:", round((time.perf_counter() - t0) * 1e3, 2), "ms", flush=True)
    ix.close()
