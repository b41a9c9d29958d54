"""The CPU restatement of the reference's HNSW index (oracle/needle_hnsw.c) against the properties the reference's
own tests pin (src/indexing/hnsw.rs:1450-1784).  The reference's graphs depend on an unseeded generator, so there is no
golden graph: parity of the graph structure is unpinned, and these are the checks the reference itself makes."""
import numpy as np
import pytest

from oracle import oracle as O


def _random_vectors(n, dim, seed):
    return np.random.default_rng(seed).uniform(-1.0, 1.0, size=(n, dim)).astype(np.float32)


def _built(vectors, metric=O.EUCLIDEAN, **kw):
    h = O.Hnsw(vectors, metric, **kw)
    h.insert_range(0, len(vectors))
    return h


def test_insert_and_search_finds_the_query_itself():
    """test_hnsw_insert_and_search (:1450-1473): 1000 x 128, Euclidean, query = row 0."""
    v = _random_vectors(1000, 128, 1)
    h = _built(v)
    assert len(h) == 1000
    ids, dists, _ = h.search(v[0], 10)
    assert len(ids) == 10 and ids[0] == 0 and dists[0] < 0.001
    assert np.all(np.diff(dists) >= 0)


def test_empty_index_returns_nothing():
    """test_hnsw_empty (:1476-1483)."""
    h = O.Hnsw(np.zeros((0, 3), np.float32), O.COSINE)
    assert len(h) == 0
    ids, _, _ = h.search(np.array([1.0, 2.0, 3.0], np.float32), 10)
    assert len(ids) == 0


def test_recall_against_the_exact_search():
    """test_hnsw_recall (:1486-1534): 500 x 32, k = 10, queries = rows 0, 10, ..., 90; mean recall > 0.8."""
    v = _random_vectors(500, 32, 2)
    h = _built(v)
    q = v[0:100:10]
    gt, _, _ = O.flat_batch_search(q, v, 10, O.EUCLIDEAN)
    total = 0.0
    for i in range(len(q)):
        ids, _, _ = h.search(q[i], 10)
        total += len(set(ids.tolist()) & set(gt[i].tolist())) / 10
    assert total / len(q) > 0.8


def test_deleted_rows_are_never_returned():
    """test_hnsw_delete (:1537-1567): a deleted row stays a routing node and leaves the results."""
    v = _random_vectors(100, 32, 3)
    h = _built(v)
    assert h.delete(5) and not h.delete(5)
    ids, _, _ = h.search(v[5], 10)
    assert 5 not in ids.tolist() and len(ids) == 10


def test_ef_below_k_and_ef_zero():
    """test_search_with_ef_less_than_k / _zero (:1756-1784): fewer results, never a failure."""
    v = _random_vectors(30, 8, 4)
    h = _built(v)
    ids, _, _ = h.search(v[0], 10, ef=2)
    assert 1 <= len(ids) <= 10
    ids, _, _ = h.search(v[0], 5, ef=0)
    assert len(ids) <= 5


def test_degrees_respect_the_connection_bounds():
    """max_connections (:592-598): 2 m on layer 0, m above, after every insert's pruning."""
    v = _random_vectors(600, 16, 5)
    h = _built(v, m=8)
    assert max(h.degree(i, 0) for i in range(600)) <= 16
    assert h.entry_level >= 1
    assert max(h.degree(i, 1) for i in range(600)) <= 8


def test_invalid_configuration_is_rejected():
    """HnswConfig::validate (:326-357)."""
    v = _random_vectors(4, 4, 6)
    with pytest.raises(ValueError):
        O.Hnsw(v, O.EUCLIDEAN, m=0)
    with pytest.raises(ValueError):
        O.Hnsw(v, O.EUCLIDEAN, m=16, m_max_0=8)
    with pytest.raises(ValueError):
        O.Hnsw(v, O.EUCLIDEAN, ef_construction=0)


def test_batch_search_equals_single_searches_and_is_thread_count_independent():
    """Collection::batch_search (src/collection/batch.rs:32-52) is one search per query."""
    v = _random_vectors(2000, 24, 7)
    h = _built(v, metric=O.COSINE)
    q = _random_vectors(40, 24, 8)
    ids1, d1, c1 = h.batch_search(q, 10, threads=1)
    ids4, d4, c4 = h.batch_search(q, 10, threads=4)
    assert np.array_equal(ids1, ids4) and np.array_equal(d1, d4) and np.array_equal(c1, c4)
    for i in (0, 17, 39):
        ids, dists, _ = h.search(q[i], 10)
        assert np.array_equal(ids, ids1[i, : c1[i]]) and np.array_equal(dists, d1[i, : c1[i]])


@pytest.mark.parametrize("metric", ["COSINE", "COSINE_NORMALIZED", "EUCLIDEAN", "DOT_PRODUCT", "MANHATTAN"])
def test_the_builds_batched_distances_leave_the_graph_unchanged(metric):
    """The build scores the unvisited neighbours of a candidate together (interleaved sums, norms kept per row): every
    adjacency list and every search result equals those of the row-by-row build, bit for bit."""
    rng = np.random.default_rng(11)
    v = rng.standard_normal((1500, 37)).astype(np.float32)
    v[40] = 0.0            # a zero norm
    v[900] = v[41]         # equal distances
    q = rng.standard_normal((30, 37)).astype(np.float32)
    m = getattr(O, metric)
    graphs = []
    for fast in (True, False):
        h = O.Hnsw(v, m, m=8, ef_construction=60, fast_build=fast)
        h.insert_range(0, 1500)
        ids, d, c = h.batch_search(q, 10, threads=2)
        graphs.append((ids.copy(), d.copy().view(np.uint32), c.copy(), [h.degree(i, 0) for i in range(1500)],
                       [h.degree(i, 1) for i in range(1500)], h.entry_level))
        h.close()
    for a, b in zip(graphs[0], graphs[1]):
        assert np.array_equal(a, b)
