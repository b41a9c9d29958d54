"""Host-side mirror logic that needs no device: validation, clamping, error types
(src/collection/validation.rs tests :88-195, src/collection/batch.rs tests :221-246)."""
import math

import numpy as np
import pytest


class _NoDevice:
    """Stands where a Context would; these tests never reach the device."""


def _col(dim=3):
    import needle_amd as na

    return na.Collection("test", dim, ctx=_NoDevice())


def test_validate_vector():  # validation.rs:88-119
    import needle_amd as na

    C = na.Collection
    C.validate_vector([1.0, 2.0, 3.0])
    C.validate_vector([])
    C.validate_vector([0.0, 0.0, 0.0])
    for bad in ([1.0, math.nan, 3.0], [math.inf, 0.0], [0.0, -math.inf]):
        with pytest.raises(na.InvalidVector):
            C.validate_vector(bad)


def test_validate_query():  # validation.rs:123-141
    import needle_amd as na

    c = _col(3)
    c.validate_query([1.0, 2.0, 3.0])
    with pytest.raises(na.DimensionMismatch) as e:
        _col(4).validate_query([1.0, 2.0])
    assert "expected 4, got 2" in str(e.value)
    with pytest.raises(na.InvalidVector):
        _col(2).validate_query([1.0, math.nan])


def test_clamp_k():  # validation.rs:167-195
    c = _col(4)
    assert c.clamp_k(10) == 0
    c.insert("v1", [1.0, 0.0, 0.0, 0.0])
    assert c.clamp_k(100) == 1 and c.clamp_k(0) == 0
    for i in range(9):
        c.insert(f"w{i}", [float(i), 0.0, 0.0, 0.0])
    assert c.clamp_k(5) == 5


def test_insert_validation_and_delete():
    import needle_amd as na

    c = _col(2)
    with pytest.raises(na.DimensionMismatch):
        c.insert("a", [1.0])
    with pytest.raises(na.InvalidVector):
        c.insert("a", [math.nan, 1.0])
    c.insert("a", [1.0, 2.0])
    with pytest.raises(ValueError):
        c.insert("a", [1.0, 2.0])
    assert len(c) == 1 and c.delete("a") and not c.delete("a") and len(c) == 0


def test_batch_search_short_circuits_without_device():
    """k == 0, an empty collection and an empty batch never touch the device (batch.rs:40-43, :229-246);
    a bad query rejects the whole batch before any work (batch.rs:37-39)."""
    import needle_amd as na

    c = _col(2)
    assert c.batch_search([[1.0, 0.0], [0.0, 1.0]], 5) == [[], []]  # empty collection -> clamp_k == 0
    c.insert("a", [1.0, 0.0])
    assert c.batch_search([[1.0, 0.0]], 0) == [[]]
    assert c.batch_search([], 3) == []
    with pytest.raises(na.DimensionMismatch):
        c.batch_search([[1.0, 0.0], [1.0]], 3)
    with pytest.raises(na.InvalidVector):
        c.batch_search([[1.0, 0.0], [math.inf, 0.0]], 3)


def test_pack_bitmask_layout():
    import needle_amd as na

    bits = np.zeros(70, dtype=bool)
    bits[[0, 31, 32, 69]] = True
    w = na.pack_bitmask(bits)
    assert w.dtype == np.uint32 and w.size == 3
    assert w[0] == (1 | (1 << 31)) and w[1] == 1 and w[2] == (1 << 5)


class _FakeAccelerator:
    """batch_*_distance of a GpuAccelerator that fails on demand (no device is touched)."""

    def __init__(self):
        self.fail = False
        self.calls = 0

    def _d(self, query, vectors):
        import needle_amd as na

        self.calls += 1
        if self.fail:
            raise na._lib.NeedleHipError(na._lib.ERR_HIP, "HIP error 719")
        q = np.asarray(query, dtype=np.float32)
        v = np.asarray(vectors, dtype=np.float32)
        return (1.0 - (v @ q) / (np.linalg.norm(v, axis=1) * np.linalg.norm(q))).astype(np.float32)

    batch_cosine_distance = batch_euclidean_distance = batch_dot_product = _d


def test_transparent_fallback_manager():  # gpu/mod.rs:1407-1560; tests :2228-2264
    import needle_amd as na

    now = [100.0]
    acc = _FakeAccelerator()
    host_calls = []

    def cpu_search(query, vectors, k, distance):
        host_calls.append(k)
        return [(2, 0.25)][:k]

    mgr = na.TransparentFallbackManager(acc, cpu_search=cpu_search, clock=lambda: now[0])
    assert mgr.gpu_reliability() == 1.0  # no ops yet (test_fallback_manager_reliability)
    query = [1.0, 0.0, 0.0]
    vectors = [[1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.5, 0.5, 0.0], [2.0, 0.0, 0.0]]
    r = mgr.search(query, vectors, 2, na.DistanceType.Cosine)  # test_fallback_manager_cpu
    assert len(r.results) == 2 and r.results[0][0] == 0 and r.backend_used == "Gpu"
    assert [i for i, _ in r.results] == [0, 3]  # equal distances keep their index order (stable sort, then truncate)
    assert mgr.gpu_successes == 1 and mgr.gpu_reliability() == 1.0
    # failures are answered by the caller's own path; the fourth one switches the device path off for 5 s
    acc.fail = True
    for n in range(1, 4):
        r = mgr.search(query, vectors, 1, na.DistanceType.Cosine)
        assert r.backend_used == "CpuSimd" and r.results == [(2, 0.25)] and mgr.gpu_failures == n and mgr.gpu_disabled_until is None
    r = mgr.search(query, vectors, 1, na.DistanceType.Cosine)
    assert mgr.gpu_failures == 4 and mgr.gpu_disabled_until == 105.0 and mgr.backoff_secs == 10
    assert mgr.gpu_reliability() == 1 / 5
    calls = acc.calls
    now[0] = 104.0
    assert mgr.search(query, vectors, 1, na.DistanceType.Cosine).backend_used == "CpuSimd" and acc.calls == calls  # not even tried
    now[0] = 105.0  # the period is over: tried again, fails again, twice the period
    mgr.search(query, vectors, 1, na.DistanceType.Cosine)
    assert acc.calls == calls + 1 and mgr.gpu_disabled_until == 115.0 and mgr.backoff_secs == 20
    for _ in range(10):  # the period doubles up to 300 s
        now[0] = mgr.gpu_disabled_until
        mgr.search(query, vectors, 1, na.DistanceType.Cosine)
    assert mgr.backoff_secs == 300
    acc.fail = False
    now[0] = mgr.gpu_disabled_until
    assert mgr.search(query, vectors, 2, na.DistanceType.Cosine).backend_used == "Gpu"
    mgr.reset()
    assert mgr.gpu_reliability() == 1.0 and mgr.gpu_disabled_until is None and mgr.backoff_secs == 5
    # no host path handed in: the device error reaches the caller, this package never answers from the CPU
    strict = na.TransparentFallbackManager(acc, clock=lambda: now[0])
    acc.fail = True
    for _ in range(5):
        with pytest.raises(na._lib.NeedleHipError):
            strict.search(query, vectors, 1, na.DistanceType.Cosine)
    assert strict.gpu_failures == 4  # the fifth call found the device path switched off
