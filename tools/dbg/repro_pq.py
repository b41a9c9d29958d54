import ctypes, os, sys
ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libsegv_bt.so"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import needle_amd as na
from needle_amd.quantization import PqIndex, ProductQuantizer
c = na.Context(0)
rng = np.random.default_rng(0)
cb = rng.standard_normal((4, 256, 8)).astype(np.float32)
codes = rng.integers(0, 256, size=(5000, 4)).astype(np.uint8)
print("building", flush=True)
ix = PqIndex(c, ProductQuantizer(cb), codes)
print("built", flush=True)
q = rng.standard_normal((3, 32)).astype(np.float32)
print(ix.search(q, 5)[0])
ix.close(); c.close()
