"""Development aid: the concurrent flat-search scenario of tests/test_reentrant_gpu.py with progress lines.

usage: python tools/diag_reentrant.py [threads] [rounds] [precision]
"""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import needle_amd as na  # noqa: E402


def main():
    n_threads = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    precision = sys.argv[3] if len(sys.argv) > 3 else "bf16"
    ctx = na.Context(0)
    rng = np.random.default_rng(8)
    base = rng.standard_normal((60_000, 96)).astype(np.float32)
    ix = na.FlatIndex(ctx, base, na.DistanceFunction.Cosine)
    batches = [rng.standard_normal((300 + 50 * t, 96)).astype(np.float32) for t in range(n_threads)]
    t0 = time.time()
    serial = [ix.search(b, 10, precision=precision) for b in batches]
    print(f"serial first pass over {n_threads} batches: {time.time() - t0:.3f} s", flush=True)
    t0 = time.time()
    serial = [ix.search(b, 10, precision=precision) for b in batches]
    print(f"serial again: {time.time() - t0:.3f} s", flush=True)
    bad = []
    start = threading.Barrier(n_threads)

    def work(t):
        start.wait()
        for r in range(rounds):
            got = ix.search(batches[t], 10, precision=precision)
            ok = all(np.array_equal(a, b) for a, b in zip(got, serial[t]))
            print(f"thread {t} round {r}: {'ok' if ok else 'MISMATCH'}", flush=True)
            if not ok:
                bad.append((t, r))

    ths = [threading.Thread(target=work, args=(t,)) for t in range(n_threads)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    print("metrics", ctx.metrics(), flush=True)
    ix.close()
    ctx.close()
    print("done, mismatches:", bad, flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
