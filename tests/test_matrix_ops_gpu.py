"""The queries x vectors operators (src/services/compute/gpu_kernels.rs:223-300, src/experimental/gpu/mod.rs:906-951)
through the C ABI: every matrix value bit-equal to the CPU restatement in each of the three arithmetics, the top-k of
every row equal to a stable sort, and the situations the reference's own tests set up (gpu_kernels.rs:407-440)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

F = np.float32


@pytest.fixture(scope="module")
def ctx():
    import needle_amd as na

    c = na.Context(0)
    yield c
    c.close()


def _bits(a):
    return np.ascontiguousarray(a, dtype=F).view(np.uint32)


def test_reference_distance_matrix_cases(ctx):
    from needle_amd.compute import GpuAccelerator

    acc = GpuAccelerator(ctx=ctx)
    r = acc.distance_matrix_cosine([[1.0] * 8, [0.0] * 8], [[1.0] * 8, [0.5] * 8, [0.0] * 8])  # gpu_kernels.rs:407-418
    assert (r.num_queries, r.num_vectors, r.matrix.size) == (2, 3, 6)
    assert r.get(0, 0) < r.get(0, 2)
    assert r.get(1, 0) == 1.0 and r.get(0, 2) == 1.0  # denominator below f32::EPSILON -> 1.0 (:349)
    r = acc.distance_matrix_cosine([[1.0] * 4], [[1.0] * 4, [0.5] * 4, [0.0] * 4])  # :420-429
    top2 = r.top_k(0, 2)
    assert len(top2) == 2 and top2[0][0] == 0
    vectors = [[float(i)] * 8 for i in range(50)]  # :431-440
    res = acc.multi_query_search([[1.0] * 8, [0.0, 1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0]], vectors, 3)
    assert len(res.results) == 2 and len(res.results[0]) == 3 and len(res.results[1]) == 3 and res.queries_per_second > 0
    e = acc.batch_euclidean_distance([1.0] * 16, [[1.0] * 16, [2.0] * 16])  # :376-382
    assert e.distances[0] < e.distances[1]
    assert acc.batch_cosine_distance([1.0] * 32, [[float(i)] * 32 for i in range(100)]).distances.size == 100  # :367-374
    assert acc.hnsw_neighbor_distances([1.0] * 16, [[float(i)] * 16 for i in range(10)]).distances.size == 10  # :442-449
    assert acc.multi_query_search([], vectors, 3).results == [] and acc.multi_query_search([[1.0] * 8], [], 3).results == [[]]


@pytest.mark.parametrize("dim", [1, 7, 32, 33, 96, 384])
def test_service_matrix_matches_oracle(ctx, oracle, dim):
    import needle_amd as na
    from needle_amd import _lib

    rng = np.random.default_rng(dim)
    q = rng.standard_normal((37, dim)).astype(F)
    b = rng.standard_normal((203, dim)).astype(F)
    b[5] = 0.0          # zero norm
    b[6] = 1e-20        # norms whose product falls below f32::EPSILON
    q[3] = 0.0
    b[9] = q[1]         # an exact zero distance
    for dtype, metric in ((0, na.DistanceFunction.Cosine), (1, na.DistanceFunction.Euclidean), (2, na.DistanceFunction.DotProduct)):
        got = ctx.distance_matrix(metric, q, b, _lib.ARITH_SERVICE)
        want = oracle.service_distance_matrix(q, b, dtype)
        assert np.array_equal(_bits(got), _bits(want)), (dim, dtype)


def test_matrix_in_the_other_arithmetics(ctx, oracle):
    import needle_amd as na
    from needle_amd import _lib

    rng = np.random.default_rng(11)
    q = rng.standard_normal((20, 50)).astype(F)
    b = rng.standard_normal((130, 50)).astype(F)
    b[0] = 0.0
    for metric in na.DistanceFunction:  # all seven metrics in distance.rs's order
        got = ctx.distance_matrix(metric, q, b, _lib.ARITH_DISTANCE_RS)
        want = np.array([[oracle.compute(int(metric), x, y) for y in b] for x in q], dtype=F)
        assert np.array_equal(_bits(got), _bits(want)), metric
    for dtype, metric in ((0, na.DistanceFunction.Cosine), (1, na.DistanceFunction.Euclidean), (2, na.DistanceFunction.DotProduct)):
        got = ctx.distance_matrix(metric, q, b, _lib.ARITH_ACCEL_CHUNK4)  # the accelerator's groups of four, dot negated
        want = np.stack([oracle.accel_batch_distance(x, b, dtype) for x in q])
        if dtype == 2:
            want = -want
        assert np.array_equal(_bits(got), _bits(want)), dtype
    with pytest.raises(_lib.NeedleHipError):
        ctx.distance_matrix(na.DistanceFunction.Manhattan, q, b, _lib.ARITH_SERVICE)


def test_multi_query_search_is_a_stable_sort_of_every_row(ctx, oracle):
    import needle_amd as na
    from needle_amd import _lib
    from needle_amd.accelerator import GpuAccelerator
    from needle_amd.distance import DistanceType

    rng = np.random.default_rng(4)
    q = rng.standard_normal((70, 24)).astype(F)
    b = rng.standard_normal((5000, 24)).astype(F)
    b[100:110] = b[7]  # ties: the lower index wins
    q[0] = b[7]
    for k in (1, 10, 64):
        idx, val, cnt = ctx.multi_query_search(na.DistanceFunction.Cosine, q, b, k, _lib.ARITH_SERVICE)
        m = oracle.service_distance_matrix(q, b, 0)
        for i in range(q.shape[0]):
            order = np.argsort(m[i], kind="stable")[:k]
            assert cnt[i] == k and np.array_equal(idx[i], order.astype(np.uint64)), (k, i)
            assert np.array_equal(_bits(val[i]), _bits(m[i][order]))
    acc = GpuAccelerator(ctx=ctx)
    got = acc.streaming_multi_query_search(q[:9], b, 5, DistanceType.Euclidean)
    for i in range(9):
        want = acc.streaming_batch_search(q[i], b, 5, DistanceType.Euclidean, chunk_size=1200)  # the chunked form, gpu/mod.rs:855-903
        oi, od = oracle.accel_fused_search(q[i], b, 5, 1)
        assert [h[0] for h in got[i]] == [h[0] for h in want] == oi.tolist()
        assert np.array_equal(_bits([h[1] for h in got[i]]), _bits(od))
    assert acc.streaming_multi_query_search([], b, 5, DistanceType.Cosine) == []
    assert acc.streaming_multi_query_search(q[:2], [], 5, DistanceType.Cosine) == [[], []]


def test_multi_query_search_in_blocks(ctx, oracle):
    """More score rows than one selection pass takes: the query blocks must line up."""
    import needle_amd as na
    from needle_amd import _lib

    rng = np.random.default_rng(6)
    q = rng.standard_normal((3000, 8)).astype(F)
    b = rng.standard_normal((150_000, 8)).astype(F)  # 1 GiB / (150000 * 4) = 1789 -> blocks of 1776 queries
    idx, val, cnt = ctx.multi_query_search(na.DistanceFunction.Euclidean, q, b, 3, _lib.ARITH_SERVICE)
    for i in (0, 1775, 1776, 2999):
        row = oracle.service_distance_matrix(q[i:i + 1], b, 1)[0]
        order = np.argsort(row, kind="stable")[:3]
        assert np.array_equal(idx[i], order.astype(np.uint64)) and np.array_equal(_bits(val[i]), _bits(row[order]))
    assert (cnt == 3).all()
