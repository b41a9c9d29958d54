"""The metadata filter mirror (needle_amd/metadata.py) against the reference's own tests of
src/metadata.rs (each case cites the #[test] it restates), the bitmask producer, and the
deleted / expired row masks of the collection mirror (src/collection/ttl.rs).  Host logic only."""
import numpy as np
import pytest

from needle_amd.metadata import Filter, FilterParseError, MetadataStore, flatten_metadata, simple_regex_match


def P(j):
    return Filter.parse(j)


def test_eq_numeric_in_and_nested():  # metadata.rs:1440-1492
    assert Filter.eq("category", "greeting").matches({"category": "greeting", "priority": 1})
    assert not Filter.eq("category", "greeting").matches({"category": "other"})
    assert Filter.gt("score", 5).matches({"score": 10}) and not Filter.gt("score", 5).matches({"score": 3})
    f = Filter.is_in("status", ["active", "pending"])
    assert f.matches({"status": "active"}) and not f.matches({"status": "deleted"})
    f = Filter.and_([Filter.eq("category", "greeting"), Filter.gt("priority", 0)])
    assert f.matches({"category": "greeting", "priority": 1}) and not f.matches({"category": "greeting", "priority": 0})
    assert Filter.eq("user.name", "Alice").matches({"user": {"name": "Alice", "age": 30}})


def test_parse_forms_and_depth():  # metadata.rs:1495-1546, 2605-2635
    assert P({"category": {"$eq": "greeting"}, "priority": {"$gte": 1}}).matches({"category": "greeting", "priority": 2})
    assert P({"category": "greeting"}).matches({"category": "greeting"})
    nested = {"field": "value"}
    for _ in range(35):
        nested = {"$not": nested}
    with pytest.raises(FilterParseError, match="too deep"):
        P(nested)
    f = P({"$and": [{"$or": [{"category": "a"}, {"category": "b"}]}, {"$not": {"status": "deleted"}}]})
    assert f.matches({"category": "b", "status": "live"}) and not f.matches({"category": "b", "status": "deleted"})
    assert Filter.eq("a", 1).depth() == 1
    assert Filter.and_([Filter.eq("a", 1), Filter.eq("b", 2)]).depth() == 2
    assert Filter.not_(Filter.eq("a", 1)).depth() == 2 and Filter.and_([]).depth() == 1
    for bad, msg in (([1], "JSON object"), ({}, "Empty filter"), ({"$foo": 1}, "Unknown operator"),
                     ({"$and": 3}, "must be an array"), ({"x": {"$bogus": 1}}, "Unknown operator")):
        with pytest.raises(FilterParseError, match=msg):
            P(bad)


def test_exists():  # metadata.rs:1965-2005, 2391-2410
    t, f = Filter.exists("category", True), Filter.exists("category", False)
    assert t.matches({"category": "books"}) and not t.matches({"other": "value"}) and not t.matches(None)
    assert not f.matches({"category": "books"}) and f.matches({"other": "value"}) and f.matches(None)
    assert Filter.exists("author.name", True).matches({"author": {"name": "Alice"}})
    assert not Filter.exists("author.name", True).matches({"author": {"age": 30}})
    assert P({"tags": {"$exists": True}}).matches({"tags": ["a", "b"]})
    assert P({"deleted": {"$exists": False}}).matches({"name": "test"})
    with pytest.raises(FilterParseError, match="boolean"):
        P({"field": {"$exists": "yes"}})
    assert Filter.exists("field", True).matches({"field": None}) and Filter.exists("field", True).matches({"field": ""})


def test_regex_subset():  # metadata.rs:2008-2061, 2147-2158
    assert Filter.regex("name", ".*Alice.*").matches({"name": "Alice Smith"})
    assert not Filter.regex("name", ".*Alice.*").matches({"name": "Bob"})
    assert Filter.regex("email", "admin.*").matches({"email": "admin@example.com"})
    assert not Filter.regex("email", "admin.*").matches({"email": "user@admin.com"})
    assert Filter.regex("file", ".*.pdf").matches({"file": "report.pdf"})
    assert not Filter.regex("file", ".*.pdf").matches({"file": "report.doc"})
    f = Filter.regex("code", "A.C")
    assert f.matches({"code": "ABC"}) and f.matches({"code": "AXC"}) and not f.matches({"code": "ABBC"})
    assert Filter.regex("any", ".*").matches({"any": ""}) and not Filter.regex("count", ".*").matches({"count": 42})
    with pytest.raises(FilterParseError, match="string"):
        P({"field": {"$regex": 42}})
    assert simple_regex_match("", "") and not simple_regex_match("a", "")
    assert simple_regex_match("hello", "^hello$") and not simple_regex_match("hello!", "^hello$")


def test_string_and_membership_operators():  # metadata.rs:2066-2095, 2314-2333, 2414-2426
    assert Filter.contains("title", "World").matches({"title": "Hello World"})
    assert Filter.contains("tags", "rust").matches({"tags": ["rust", "db"]})
    assert Filter.starts_with("name", "Al").matches({"name": "Alice"}) and not Filter.starts_with("name", "Al").matches({"name": "Bob"})
    assert Filter.ends_with("file", ".rs").matches({"file": "main.rs"})
    assert Filter.not_in("status", ["deleted"]).matches({"status": "active"})
    assert not P({"status": {"$in": []}}).matches({"status": "active"}) and not P({"status": {"$in": []}}).matches(None)
    assert P({"status": {"$nin": []}}).matches({"status": "active"})
    assert not Filter.contains("field", 5).matches({"field": 5}) and not Filter.contains("field", True).matches({"field": True})


def test_all_and_elem_match():  # metadata.rs:2161-2283
    assert P({"tags": {"$all": ["rust", "db"]}}).matches({"tags": ["rust", "db", "fast"]})
    assert not P({"tags": {"$all": ["rust", "missing"]}}).matches({"tags": ["rust", "db"]})
    assert P({"tags": {"$all": []}}).matches({"tags": ["x"]}) and not P({"name": {"$all": ["x"]}}).matches({"name": "x"})
    assert P({"scores": {"$all": [1, 2, 3]}}).matches({"scores": [3, 2, 1, 0]})
    with pytest.raises(FilterParseError):
        P({"tags": {"$all": "not_array"}})
    f = P({"items": {"$elemMatch": {"price": {"$gt": 10}, "in_stock": True}}})
    assert f.matches({"items": [{"price": 5, "in_stock": True}, {"price": 20, "in_stock": True}]})
    assert not f.matches({"items": [{"price": 5, "in_stock": True}, {"price": 20, "in_stock": False}]})
    assert not f.matches({"items": "x"}) and not f.matches({"items": []})
    with pytest.raises(FilterParseError):
        P({"items": {"$elemMatch": [1]}})


def test_paths_unicode_and_number_identity():  # metadata.rs:2336-2388
    assert P({"a.b.c.d.e": "deep"}).matches({"a": {"b": {"c": {"d": {"e": "deep"}}}}})
    assert not P({"a.b.c": "value"}).matches({"a": {"x": 1}}) and not P({"a.b": "value"}).matches({"a": None})
    assert P({"名前": "テスト"}).matches({"名前": "テスト"}) and Filter.eq("tag", "🔥hot").matches({"tag": "🔥hot"})
    assert P({"score": 0.1}).matches({"score": 0.1})
    assert P({"count": {"$gte": 5}}).matches({"count": 5.0})       # comparisons go through f64
    assert not Filter.eq("count", 5).matches({"count": 5.0})        # serde_json: an integer is not a float
    assert not Filter.eq("flag", 1).matches({"flag": True})
    assert Filter.eq("items.1", "b").matches({"items": ["a", "b"]})  # arrays by index, metadata.rs:1232-1235


def test_between_size_type():  # metadata.rs:2430-2600
    f = Filter.between("price", 10, 50)
    assert [f.matches({"price": p}) for p in (10, 30, 50, 9, 51)] == [True, True, True, False, False]
    f = P({"score": {"$between": [0.5, 0.9]}})
    assert [f.matches({"score": s}) for s in (0.7, 0.5, 0.9, 0.4, 1.0)] == [True, True, True, False, False]
    for bad in ({"x": {"$between": [1, 2, 3]}}, {"x": {"$between": 5}}, {"tags": {"$size": "three"}},
                {"x": {"$type": "integer"}}, {"x": {"$type": 42}}):
        with pytest.raises(FilterParseError):
            P(bad)
    assert not Filter.between("price", 1, 10).matches(None) and not Filter.between("name", 1, 10).matches({"name": "alice"})
    assert Filter.size("tags", 3).matches({"tags": ["a", "b", "c"]}) and not Filter.size("tags", 3).matches({"tags": []})
    assert Filter.size("code", 5).matches({"code": "hello"}) and Filter.size("items", 0).matches({"items": []})
    assert not Filter.size("val", 1).matches({"val": 42}) and not Filter.size("val", 1).matches({"val": True})
    assert Filter.size("s", 3).matches({"s": "é1"})  # byte length, as str::len
    for name, good, bad in (("string", "x", 1), ("number", 1.5, "x"), ("boolean", False, 0), ("array", [1], {}),
                            ("object", {}, []), ("null", None, 0)):
        assert Filter.type_of("v", name).matches({"v": good}) and not Filter.type_of("v", name).matches({"v": bad})


def test_inverted_index_and_bitmask():  # metadata.rs:1550-1660, 1866-1960
    st = MetadataStore()
    st.insert(0, "v0", {"color": "red", "size": 10})
    st.insert(1, "v1", {"color": "blue", "size": 20})
    st.insert(2, "v2", {"color": "red", "size": 30})
    assert "color" in st.indexed_fields() and st.field_cardinality("color") == 2
    assert st.lookup_field_eq("color", "red") == [0, 2] and st.lookup_field_eq("nope", 1) is None
    assert st.resolve_filter_via_index(Filter.eq("color", "red")) == {0, 2}
    assert st.resolve_filter_via_index(Filter.and_([Filter.eq("color", "red"), Filter.eq("size", 30)])) == {2}
    assert st.resolve_filter_via_index(Filter.gt("size", 5)) is None  # falls back to the sequential scan
    st.delete(0)
    assert st.lookup_field_eq("color", "red") == [2]
    with pytest.raises(ValueError):
        st.insert(5, "v1", None)
    st.insert(3, "v3", None)  # no metadata: matches nothing but $exists:false
    assert st.filter_bitmask(Filter.eq("color", "red"), 4).tolist() == [False, False, True, False]
    assert st.filter_bitmask(Filter.gt("size", 15), 4).tolist() == [False, True, True, False]
    assert st.filter_bitmask(Filter.exists("color", False), 4).tolist() == [False, False, False, True]
    nested = MetadataStore()
    nested.insert(0, "a", {"author": {"name": "Alice"}, "tags": ["x"]})
    assert dict(flatten_metadata({"a": {"b": 1}, "c": [1]})) == {"a.b": 1, "c": [1]}
    assert nested.resolve_filter_via_index(Filter.eq("author.name", "Alice")) == {0}
    hc = MetadataStore(cardinality_threshold=3)
    for i in range(6):
        hc.insert(i, f"u{i}", {"uuid": f"id{i}", "kind": "k"})
    assert hc.is_high_cardinality("uuid") and "uuid" not in hc.indexed_fields() and not hc.is_high_cardinality("kind")
    assert hc.filter_bitmask(Filter.eq("uuid", "id4"), 6).tolist() == [False] * 4 + [True, False]  # sequential path


def test_collection_masks_without_a_device():
    """Deleted and expired rows and the filter bitmask are decided on the host (ttl.rs, search_methods.rs:205-223)."""
    from needle_amd.collection import Collection

    now = [1000]
    col = Collection("c", 4, clock=lambda: now[0])
    v = [1.0, 0.0, 0.0, 0.0]
    col.insert("keep", v, {"type": "x", "n": 1})
    col.insert_with_ttl("temp", v, {"type": "x", "n": 2}, 60)
    col.insert("other", v, {"type": "y", "n": 3})
    col.insert("bare", v)
    assert col._live_mask() is None and col.get_ttl("temp") == 1060 and col.get_ttl("keep") is None
    assert col.filter_mask(Filter.eq("type", "x")).tolist() == [True, True, False, False]
    assert col.filter_mask({"n": {"$gte": 2}}).tolist() == [False, True, True, False]
    assert col.filter_mask(lambda m: m is None).tolist() == [False, False, False, True]
    now[0] = 1060  # expired exactly at the expiration second (ttl.rs:19: now >= expiration)
    assert col.is_expired(1) and col._live_mask().tolist() == [True, False, True, True]
    assert col.ttl_stats() == (1, 1, 1060, 1060) and col.needs_expiration_sweep(0.2) and not col.needs_expiration_sweep(0.3)
    col.lazy_expiration = False
    assert col._live_mask() is None
    col.lazy_expiration = True
    col.set_ttl("temp", 100)
    assert not col.is_expired(1) and col._live_mask() is None
    col.set_ttl("temp", None)
    assert col.get_ttl("temp") is None
    with pytest.raises(KeyError):
        col.set_ttl("missing", 5)
    col.set_ttl("other", 0)
    assert col.expire_vectors() == 1 and len(col) == 3 and col.get_ttl("other") is None
    assert col._live_mask().tolist() == [True, True, False, True]
    assert col.filter_mask(Filter.eq("type", "y")).tolist() == [False] * 4  # its index entries went with it
    assert col.delete("keep") and not col.delete("keep")
    assert col._live_mask().tolist() == [False, True, False, True]
