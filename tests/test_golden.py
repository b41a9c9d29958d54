"""The committed fixture tests/golden/hotpath_golden.npz (expected outputs on the reference's
deterministic gen_vector inputs, src/recall_benchmark.rs:171-180, written by
tests/golden/make_golden.py).  CPU: the oracle still reproduces every array in the file.  GPU: the HIP
path, through the C ABI, returns exactly the file's ids and distance bits with no oracle in the loop."""
import importlib.util
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
F = np.float32


def _maker():
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="module")
def golden():
    with np.load(os.path.join(HERE, "golden", "hotpath_golden.npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def test_oracle_reproduces_the_fixture(oracle, golden):
    fresh = _maker().compute()
    assert sorted(fresh) == sorted(golden)
    for key, want in golden.items():
        assert fresh[key].dtype == want.dtype and np.array_equal(fresh[key], want), key


def test_fixture_inputs_are_the_reference_generator(oracle):
    """gen_vector is a pure integer recurrence: restated here in numpy, independent of the C oracle."""
    mk = _maker()
    mul, inc = np.uint64(6364136223846793005), np.uint64(1442695040888963407)
    with np.errstate(over="ignore"):
        for seed, idx in ((mk.BASE_SEED, 0), (mk.QUERY_SEED, 5)):
            state = np.uint64(seed) * mul + np.uint64(idx)
            vals = []
            for _ in range(7):
                state = state * mul + inc
                vals.append(F(F(F(int(state >> np.uint64(33))) / F(4294967295.0)) * F(2.0)) - F(1.0))
            got = oracle.gen_vectors(seed, idx, 1, 7)[0]
            assert np.array_equal(np.array(vals, dtype=F).view(np.uint32), got.view(np.uint32))


# ---------------------------------------------------------------------------------------------
# GPU: the HIP path against the file
# ---------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def ctx():
    import needle_amd as na

    c = na.Context(0)
    yield c
    c.close()


def _gen(seed, count, dim):
    """gen_vector in numpy (uint64 wrap-around), so the GPU tests need nothing from oracle/."""
    mul, inc = np.uint64(6364136223846793005), np.uint64(1442695040888963407)
    with np.errstate(over="ignore"):
        state = np.uint64(seed) * mul + np.arange(count, dtype=np.uint64)
        out = np.empty((count, dim), dtype=F)
        for j in range(dim):
            state = state * mul + inc
            out[:, j] = ((state >> np.uint64(33)).astype(F) / F(4294967295.0)) * F(2.0) - F(1.0)
    return out


def _inputs(spec):
    mk = _maker()
    return _gen(mk.BASE_SEED, spec["n"], spec["dim"]), _gen(mk.QUERY_SEED, spec["nq"], spec["dim"])


def _same(got, ids, dist_bits):
    assert np.array_equal(got[0], ids)
    assert np.array_equal(np.ascontiguousarray(got[1]).view(np.uint32), dist_bits)


def _unit(rows):  # normalize of src/distance.rs:426-434: reciprocal of the sequential-sum norm, then multiply
    out = rows.copy()
    for r in out:
        s = F(-0.0)
        for x in r:
            s = F(s + F(x * x))
        n = np.sqrt(s, dtype=F)
        if n > 0:
            r *= F(F(1.0) / n)
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_gpu_flat_matches_fixture(ctx, golden, precision):
    import needle_amd as na

    mk = _maker()
    base, queries = _inputs(mk.FLAT)
    names = {"cosine": "Cosine", "cosine_normalized": "CosineNormalized", "euclidean": "Euclidean", "dot": "DotProduct",
             "manhattan": "Manhattan", "hamming": "Hamming", "chebyshev": "Chebyshev"}
    for name, enum_name in names.items():
        b, q = (_unit(base), _unit(queries)) if name == "cosine_normalized" else (base, queries)
        ix = na.FlatIndex(ctx, b, na.DistanceFunction[enum_name])
        _same(ix.search(q, mk.FLAT["k"], precision=precision), golden[f"flat_{name}_ids"], golden[f"flat_{name}_dist_bits"])
        ix.set_valid_mask(golden["flat_valid"])
        _same(ix.search(q, mk.FLAT["k"], precision=precision, filter_mask=golden["flat_filter"]),
              golden[f"flat_{name}_masked_ids"], golden[f"flat_{name}_masked_dist_bits"])
        ix.close()
    base, queries = _inputs(mk.WIDE)
    for name, enum_name in (("cosine", "Cosine"), ("euclidean", "Euclidean")):
        ix = na.FlatIndex(ctx, base, na.DistanceFunction[enum_name])
        _same(ix.search(queries, mk.WIDE["k"], precision=precision), golden[f"wide_{name}_ids"], golden[f"wide_{name}_dist_bits"])
        ix.close()


@pytest.mark.gpu
def test_gpu_ivf_matches_fixture(ctx, golden):
    from needle_amd.ivf import IvfConfig, IvfIndex

    mk = _maker()
    base, queries = _inputs(mk.IVF)
    cents = golden["ivf_centroids"]
    assert np.array_equal(ctx.kmeans_assign(base, cents, 0), golden["ivf_assign"])
    ix = IvfIndex(mk.IVF["dim"], IvfConfig.new(cents.shape[0]).with_nprobe(mk.IVF["nprobe"]), ctx=ctx)
    ix.set_centroids(cents)
    ix.add(range(mk.IVF["n"]), base)
    assert np.array_equal(np.stack(ix.probe(queries)), golden["ivf_probe"])
    for precision in ("fp32", "bf16"):
        got = ix.batch_search_arrays(queries, mk.IVF["k"], precision=precision)
        _same(got, golden["ivf_ids"], golden["ivf_dist_bits"])
        assert np.array_equal(got[2], golden["ivf_counts"])
    ix.close()


@pytest.mark.gpu
def test_gpu_quantized_scans_match_fixture(ctx, golden):
    from needle_amd.quantization import PqIndex, ProductQuantizer, ScalarQuantizer, Sq8Index

    mk = _maker()
    base, queries = _inputs(mk.ADC)
    sq = ScalarQuantizer.train(base[:1000])
    assert np.array_equal(sq.min_vals, golden["sq_min"]) and np.array_equal(sq.scale, golden["sq_scale"])
    codes = sq.quantize(base)
    assert np.array_equal(codes[:64], golden["sq_codes_head"])
    ix = Sq8Index(ctx, sq, codes)
    for precision in ("fp32", "bf16"):
        _same(ix.search(queries, mk.ADC["k"], precision=precision), golden["sq_ids"], golden["sq_dist_bits"])
    ix.close()
    pq = ProductQuantizer(golden["pq_codebooks"])
    assert np.array_equal(pq.encode(base[:300]), golden["pq_codes"][:300])
    ix = PqIndex(ctx, pq, golden["pq_codes"])
    _same(ix.search(queries, mk.ADC["k"]), golden["pq_ids"], golden["pq_dist_bits"])
    ix.close()
