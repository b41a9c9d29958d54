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
