"""The allocation-free top-k entry point (static device workspace) against the allocating one, through the C ABI:
two calls in a row, the second with a smaller input, every output bit-equal; oversize requests are refused."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(fn, ctx, scores, k):
    import torch

    rows, n = scores.shape
    dev = torch.device("cuda", 0)
    ts = torch.from_numpy(np.ascontiguousarray(scores)).to(dev)
    kk = max(k, 1)
    idx = torch.full((rows, kk), -7, dtype=torch.int64, device=dev)
    val = torch.full((rows, kk), -7.0, dtype=torch.float32, device=dev)
    cnt = torch.full((rows,), -7, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    st = fn(ctx._h, ts.data_ptr(), rows, n, k, idx.data_ptr(), val.data_ptr(), cnt.data_ptr())
    torch.cuda.synchronize()
    return st, idx.cpu().numpy(), val.cpu().numpy(), cnt.cpu().numpy()


def test_noalloc_top_k_matches_allocating_entry_point():
    import needle_amd as na
    from needle_amd import _lib

    L = _lib.lib()
    ctx = na.Context(0)
    try:
        rng = np.random.default_rng(123)
        first = rng.standard_normal((6, 200_000)).astype(np.float32)
        first[:, ::5] = np.round(first[:, ::5], 1)       # ties, decided by index
        first[2, 100_000:] = first[2, 100_000]           # a run of equal scores across workgroups
        second = rng.standard_normal((2, 3000)).astype(np.float32)   # smaller: stale workspace contents must not matter
        second[1, :] = 0.25
        for scores, k in ((first, 100), (second, 7), (second, 3000), (first[:1, :50], 64)):
            st_a, ia, va, ca = _run(L.needle_hip_batch_top_k_device, ctx, scores, k)
            st_b, ib, vb, cb = _run(L.needle_hip_batch_top_k_device_noalloc, ctx, scores, k)
            if k > 2048:
                assert st_a == st_b == _lib.ERR_UNSUPPORTED
                continue
            assert st_a == _lib.OK and st_b == _lib.OK, L.needle_hip_last_error()
            assert np.array_equal(ca, cb) and np.array_equal(ia, ib)
            assert np.array_equal(va.view(np.uint32), vb.view(np.uint32))
            assert np.all(ca == min(k, scores.shape[1]))
            # and both agree with a stable host sort
            for r in range(scores.shape[0]):
                want = np.argsort(scores[r], kind="stable")[: ca[r]]
                assert np.array_equal(ib[r, : ca[r]], want)
        # k = 0 and an oversize request
        st, _, _, c0 = _run(L.needle_hip_batch_top_k_device_noalloc, ctx, second, 0)
        assert st == _lib.OK and np.all(c0 == 0)
        big = np.zeros((200, 4096), dtype=np.float32)
        st, _, _, _ = _run(L.needle_hip_batch_top_k_device_noalloc, ctx, big, 2048)
        assert st == _lib.ERR_UNSUPPORTED and b"allocation-free" in L.needle_hip_last_error()
        st, ia, va, ca = _run(L.needle_hip_batch_top_k_device, ctx, big, 2048)
        assert st == _lib.OK and np.array_equal(ia[5], np.arange(2048))
    finally:
        ctx.close()
