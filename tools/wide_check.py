"""Prints the first-pass line of a 1024- and a 1536-dimensional search (which precision served it) and times both."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
import needle_amd as na
ctx = na.Context(0)
rng = np.random.default_rng(0)
for d in (1024, 1536):
    base = rng.standard_normal((200_000, d)).astype(np.float32)
    q = rng.standard_normal((1000, d)).astype(np.float32)
    ix = na.FlatIndex(ctx, base, na.DistanceFunction.DotProduct)
    ctx.set_tuning(debug=8)
    ix.search(q, 10, precision="bf16")
    ctx.set_tuning(debug=0)
    for prec in ("bf16", "fp32"):
        ix.search(q, 10, precision=prec)
        t = time.perf_counter()
        for _ in range(3):
            ix.search(q, 10, precision=prec)
        print(f"dim {d} first pass {prec}: {(time.perf_counter() - t) / 3 * 1e3:.2f} ms per 1K queries over 200K rows (host pointers)")
    ix.close()
