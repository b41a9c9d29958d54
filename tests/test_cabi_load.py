"""CPU-side checks of the C ABI: the library loads and exports every symbol the header declares.
No compute call is made (there is no GPU here, and the library has no CPU path)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "needle_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(needle_hip_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    from needle_amd import _lib

    assert sorted(_lib.EXPORTS) == _header_symbols()


def _struct_fields(text, opener, c_style):
    """(name, kind) of every field of a struct body; kind is 'u64' or 'f64'."""
    body = text[text.index(opener):]
    body = body[body.index("{") + 1: body.index("}")]
    body = re.sub(r"/\*.*?\*/|//[^\n]*", "", body, flags=re.S)
    if c_style:
        pairs = re.findall(r"\b(uint64_t|double)\s+([a-z_0-9]+)\s*;", body)
        return [(n, "u64" if t == "uint64_t" else "f64") for t, n in pairs]
    return [(n, t) for n, t in re.findall(r"pub\s+([a-z_0-9]+)\s*:\s*(u64|f64)", body)]


def test_metrics_struct_mirrors_agree():
    """needle_hip_get_metrics writes the whole struct: a mirror that is shorter, or ordered differently, reads
    garbage or is overrun.  The ctypes binding and the Rust crate's mirror must match the header."""
    import ctypes as C

    from needle_amd import _lib

    header = _struct_fields(open(os.path.join(ROOT, "include", "needle_hip.h")).read(), "typedef struct needle_hip_metrics", True)
    assert len(header) >= 9
    binding = [(n, "u64" if t is C.c_uint64 else "f64" if t is C.c_double else str(t)) for n, t in _lib.Metrics._fields_]
    assert binding == header
    assert C.sizeof(_lib.Metrics) == 8 * len(header)
    rust = _struct_fields(open(os.path.join(ROOT, "bindings", "rust", "needle-hip-sys", "src", "lib.rs")).read(),
                          "pub struct needle_hip_metrics", False)
    assert rust == header


def test_search_params_mirrors_agree():
    """Same for needle_search_params, which every search call reads: field names and order."""
    from needle_amd import _lib

    def names(text, opener, pattern):
        body = text[text.index(opener):]
        body = re.sub(r"/\*.*?\*/|//[^\n]*", "", body[body.index("{") + 1: body.index("}")], flags=re.S)
        return re.findall(pattern, body)

    header = names(open(os.path.join(ROOT, "include", "needle_hip.h")).read(), "typedef struct needle_search_params",
                   r"([a-z_0-9]+)\s*;")
    assert header == ["k", "nprobe", "precision", "reserved", "filter_bitmask"]
    assert [n for n, _ in _lib.SearchParams._fields_] == header
    rust = names(open(os.path.join(ROOT, "bindings", "rust", "needle-hip-sys", "src", "lib.rs")).read(),
                 "pub struct needle_search_params", r"pub\s+([a-z_0-9]+)\s*:")
    assert rust == header


def test_library_exports_every_declared_symbol():
    from needle_amd import _lib

    L = _lib.lib()
    for sym in _header_symbols():
        assert hasattr(L, sym), sym
    assert L.needle_hip_version() >= 100


def test_metric_aliases_resolved_by_library():
    import needle_amd as na

    DF = na.DistanceFunction
    for s, m in {"cosine": DF.Cosine, "L2": DF.Euclidean, "inner_product": DF.DotProduct, "l1": DF.Manhattan,
                 "linf": DF.Chebyshev, "CosineNormalized": DF.CosineNormalized, "hamming": DF.Hamming}.items():
        assert DF.from_str(s) == m
    with pytest.raises(ValueError):
        DF.from_str("nope")
    assert str(DF.DotProduct) == "DotProduct"


def test_no_device_fails_loudly():
    """Without a GPU the product path must refuse to start, not fall back to anything."""
    import torch

    import needle_amd as na

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(na.NeedleHipError) as e:
        na.Context(0)
    assert e.value.status == na._lib.ERR_NO_DEVICE


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "needle_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("the oracle's", ""), os.path.join(dirpath, f)
