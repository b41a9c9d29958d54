"""One IVF measurement with explicit shape: python tools/ivf_shape.py ROWS NLIST [NPROBE] (parity sample included)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import needle_amd as na  # noqa: E402
import tools.bench_paths as bp  # noqa: E402
from oracle import oracle as orc  # noqa: E402  (checker only)

rows, nlist = int(sys.argv[1]), int(sys.argv[2])
nprobe = int(sys.argv[3]) if len(sys.argv) > 3 else 32
orc.build()
torch.cuda.set_device(0)
ctx = na.Context(0)
bp.ivf_case(ctx, torch.device("cuda", 0), rows, 384, nlist, nprobe, 10_000, 10, 3, oracle=orc)
ctx.close()
