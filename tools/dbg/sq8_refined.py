import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import needle_amd as na
from needle_amd.quantization import ScalarQuantizer, Sq8Index
c = na.Context(0)
rng = np.random.default_rng(1)
dim = 384
sample = rng.standard_normal((20000, dim)).astype(np.float32)
sq = ScalarQuantizer.train(sample)
os.environ["NEEDLE_HIP_DEBUG"] = "8"
for n in (4_500_000, 6_250_000):
    codes = rng.integers(0, 256, size=(n, dim), dtype=np.uint8)
    q = rng.standard_normal((1000, dim)).astype(np.float32)
    ix = Sq8Index(c, sq, codes)
    for prec in ("bf16",):
        c.reset_metrics()
        ids, d, cnt = ix.search(q, 10, precision=prec)
        print(n, dim, prec, {k: v for k, v in c.metrics().items() if "queries" in k}, ids[0, :3], d[0, :3], flush=True)
    ix.close()
c.close()
