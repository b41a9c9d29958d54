"""Host-side mirror of the reference's metadata filter (src/metadata.rs): `Filter` (MongoDB-style
operators, `parse`, `matches`), the equality inverted index of `MetadataStore`, and the producer of the
row bitmask the GPU scan consumes as an exact PRE-filter (the rule of `brute_force_search`,
src/collection/search_methods.rs:215-223: a row whose metadata fails the filter is never scored).

JSON values follow serde_json's equality, not Python's: `true != 1`, `1 != 1.0`, string lengths are UTF-8
byte lengths (src/metadata.rs:1246, 1317).
"""
from __future__ import annotations

import json
from typing import Any, Dict, Iterable, List, Optional, Sequence, Set, Tuple

import numpy as np

MAX_FILTER_DEPTH = 32          # src/metadata.rs:971
MAX_FILTER_ARRAY_SIZE = 1000   # src/metadata.rs:974
FLATTEN_MAX_DEPTH = 5          # src/metadata.rs:139
DEFAULT_CARDINALITY_THRESHOLD = 10_000  # src/metadata.rs:189
_TYPE_NAMES = ("string", "number", "boolean", "array", "object", "null")


class FilterParseError(ValueError):
    """`Filter::parse` returned Err(String)."""


def _is_number(v: Any) -> bool:
    return isinstance(v, (int, float)) and not isinstance(v, bool)


def json_eq(a: Any, b: Any) -> bool:
    """serde_json::Value equality: same JSON type, numbers equal only within the same representation
    (integer vs float), arrays element-wise, objects key-wise."""
    if isinstance(a, bool) or isinstance(b, bool):
        return isinstance(a, bool) and isinstance(b, bool) and a == b
    if a is None or b is None:
        return a is None and b is None
    if _is_number(a) or _is_number(b):
        return _is_number(a) and _is_number(b) and isinstance(a, float) == isinstance(b, float) and a == b
    if isinstance(a, str) or isinstance(b, str):
        return isinstance(a, str) and isinstance(b, str) and a == b
    if isinstance(a, (list, tuple)) or isinstance(b, (list, tuple)):
        return (isinstance(a, (list, tuple)) and isinstance(b, (list, tuple)) and len(a) == len(b)
                and all(json_eq(x, y) for x, y in zip(a, b)))
    if isinstance(a, dict) and isinstance(b, dict):
        return a.keys() == b.keys() and all(json_eq(a[k], b[k]) for k in a)
    return False


def _contains(arr: Iterable[Any], v: Any) -> bool:
    return any(json_eq(x, v) for x in arr)


def _num(v: Any) -> Optional[float]:  # value_to_f64, src/metadata.rs:1411-1416
    return float(v) if _is_number(v) else None


def _cmp_numeric(a: Any, b: Any, op) -> bool:  # src/metadata.rs:1397-1408
    x, y = _num(a), _num(b)
    return x is not None and y is not None and op(x, y)


def get_field_value(value: Any, path: str) -> Tuple[bool, Any]:
    """Dot-notation lookup (objects by key, arrays by index), src/metadata.rs:1223-1241 -> (found, value)."""
    cur = value
    for part in path.split("."):
        if isinstance(cur, dict):
            if part not in cur:
                return False, None
            cur = cur[part]
        elif isinstance(cur, (list, tuple)):
            digits = part[1:] if part.startswith("+") else part  # usize::from_str
            if not (digits.isascii() and digits.isdigit()):
                return False, None
            i = int(digits)
            if i >= len(cur):
                return False, None
            cur = cur[i]
        else:
            return False, None
    return True, cur


def simple_regex_match(text: str, pattern: str) -> bool:
    """The reference's regex subset (src/metadata.rs:1351-1394): optional ^ $, leading / trailing `.*`,
    otherwise equal length with `.` matching any one character."""
    if pattern == "":
        return text == ""
    if pattern in (".*", "^.*$"):
        return True
    if pattern.startswith("^"):
        pattern = pattern[1:]
    if pattern.endswith("$"):
        pattern = pattern[:-1]
    prefix_wild = pattern.startswith(".*")
    if prefix_wild:
        pattern = pattern[2:]
    suffix_wild = pattern.endswith(".*")
    if suffix_wild:
        pattern = pattern[:-2]
    if prefix_wild and suffix_wild:
        return pattern in text
    if prefix_wild:
        return text.endswith(pattern)
    if suffix_wild:
        return text.startswith(pattern)
    if len(pattern.encode()) != len(text.encode()):
        return False
    return all(p == "." or p == t for p, t in zip(pattern, text))


class Filter:
    """Filter::{Condition, And, Or, Not}, src/metadata.rs:721-938."""

    __slots__ = ("kind", "field", "op", "value", "children")

    def __init__(self, kind: str, field: str = "", op: str = "", value: Any = None, children: Sequence["Filter"] = ()):
        self.kind, self.field, self.op, self.value, self.children = kind, field, op, value, list(children)

    # ---- constructors (src/metadata.rs:734-928) ---------------------------------
    @classmethod
    def _cond(cls, field: str, op: str, value: Any) -> "Filter":
        return cls("cond", field, op, value)

    @classmethod
    def eq(cls, field, value): return cls._cond(field, "$eq", value)
    @classmethod
    def ne(cls, field, value): return cls._cond(field, "$ne", value)
    @classmethod
    def gt(cls, field, value): return cls._cond(field, "$gt", value)
    @classmethod
    def gte(cls, field, value): return cls._cond(field, "$gte", value)
    @classmethod
    def lt(cls, field, value): return cls._cond(field, "$lt", value)
    @classmethod
    def lte(cls, field, value): return cls._cond(field, "$lte", value)
    @classmethod
    def is_in(cls, field, values): return cls._cond(field, "$in", list(values))
    @classmethod
    def not_in(cls, field, values): return cls._cond(field, "$nin", list(values))
    @classmethod
    def contains(cls, field, value): return cls._cond(field, "$contains", value)
    @classmethod
    def starts_with(cls, field, prefix): return cls._cond(field, "$startsWith", prefix)
    @classmethod
    def ends_with(cls, field, suffix): return cls._cond(field, "$endsWith", suffix)
    @classmethod
    def exists(cls, field, should_exist: bool): return cls._cond(field, "$exists", bool(should_exist))
    @classmethod
    def regex(cls, field, pattern: str): return cls._cond(field, "$regex", pattern)
    @classmethod
    def all(cls, field, values): return cls._cond(field, "$all", list(values))
    @classmethod
    def elem_match(cls, field, sub_filter: dict): return cls._cond(field, "$elemMatch", sub_filter)
    @classmethod
    def between(cls, field, low, high): return cls._cond(field, "$between", [low, high])
    @classmethod
    def size(cls, field, n: int): return cls._cond(field, "$size", n)
    @classmethod
    def type_of(cls, field, type_name: str): return cls._cond(field, "$type", type_name)
    @classmethod
    def and_(cls, filters): return cls("and", children=filters)
    @classmethod
    def or_(cls, filters): return cls("or", children=filters)
    @classmethod
    def not_(cls, f): return cls("not", children=[f])

    def __repr__(self) -> str:
        if self.kind == "cond":
            return f"Filter({self.field!r} {self.op} {self.value!r})"
        return f"Filter.{self.kind}({self.children!r})"

    # ---- evaluation (src/metadata.rs:931-938, 1197-1337) --------------------------
    def matches(self, metadata: Optional[Any]) -> bool:
        if self.kind == "and":
            return all(f.matches(metadata) for f in self.children)
        if self.kind == "or":
            return any(f.matches(metadata) for f in self.children)
        if self.kind == "not":
            return not self.children[0].matches(metadata)
        if metadata is None:  # no metadata: only $exists:false matches
            return self.op == "$exists" and self.value is False
        found, fv = get_field_value(metadata, self.field)
        if self.op == "$exists":
            want = self.value if isinstance(self.value, bool) else True
            return found == want
        return found and _compare(self.op, fv, self.value)

    def depth(self) -> int:  # src/metadata.rs:944-952
        if self.kind == "cond":
            return 1
        return 1 + max((f.depth() for f in self.children), default=0)

    def equality_conditions(self) -> List[Tuple[str, str]]:  # src/metadata.rs:957-968
        if self.kind == "cond" and self.op == "$eq":
            return [(self.field, value_string(self.value))]
        if self.kind == "and":
            return [c for f in self.children for c in f.equality_conditions()]
        return []

    # ---- parse (src/metadata.rs:984-1193) ------------------------------------------
    @classmethod
    def parse(cls, value: Any) -> "Filter":
        return cls._parse(value, 0)

    @classmethod
    def _parse(cls, value: Any, depth: int) -> "Filter":
        if depth > MAX_FILTER_DEPTH:
            raise FilterParseError(f"Filter nesting too deep (max {MAX_FILTER_DEPTH} levels)")
        if not isinstance(value, dict):
            raise FilterParseError("Filter must be a JSON object")
        for key, kind in (("$and", "and"), ("$or", "or")):
            if key in value:
                arr = value[key]
                if not isinstance(arr, (list, tuple)):
                    raise FilterParseError(f"{key} must be an array")
                if len(arr) > MAX_FILTER_ARRAY_SIZE:
                    raise FilterParseError(f"{key} array too large ({len(arr)} elements, max {MAX_FILTER_ARRAY_SIZE})")
                return cls(kind, children=[cls._parse(v, depth + 1) for v in arr])
        if "$not" in value:
            return cls.not_(cls._parse(value["$not"], depth + 1))
        conds = []
        for field, fv in value.items():
            if field.startswith("$"):
                raise FilterParseError(f"Unknown operator: {field}")
            conds.append(cls._parse_field(field, fv))
        if not conds:
            raise FilterParseError("Empty filter object")
        return conds[0] if len(conds) == 1 else cls.and_(conds)

    @classmethod
    def _parse_field(cls, field: str, value: Any) -> "Filter":
        if not (isinstance(value, dict) and any(k.startswith("$") for k in value)):
            return cls.eq(field, value)  # simple equality
        conds = []
        for op, ov in value.items():
            if op in ("$eq", "$ne", "$gt", "$gte", "$lt", "$lte", "$contains", "$startsWith", "$endsWith"):
                conds.append(cls._cond(field, op, ov))
            elif op in ("$in", "$nin", "$all"):
                if not isinstance(ov, (list, tuple)):
                    raise FilterParseError(f"{op} requires an array")
                conds.append(cls._cond(field, op, list(ov)))
            elif op == "$exists":
                if not isinstance(ov, bool):
                    raise FilterParseError("$exists requires a boolean value")
                conds.append(cls.exists(field, ov))
            elif op == "$regex":
                if not isinstance(ov, str):
                    raise FilterParseError("$regex requires a string pattern")
                conds.append(cls.regex(field, ov))
            elif op == "$elemMatch":
                if not isinstance(ov, dict):
                    raise FilterParseError("$elemMatch requires an object")
                conds.append(cls.elem_match(field, ov))
            elif op == "$between":
                if not isinstance(ov, (list, tuple)):
                    raise FilterParseError("$between requires an array of [low, high]")
                if len(ov) != 2:
                    raise FilterParseError("$between requires an array of exactly [low, high]")
                conds.append(cls._cond(field, op, list(ov)))
            elif op == "$size":
                if not _is_number(ov):
                    raise FilterParseError("$size requires a number")
                conds.append(cls._cond(field, op, ov))
            elif op == "$type":
                if not isinstance(ov, str):
                    raise FilterParseError("$type requires a string")
                if ov not in _TYPE_NAMES:
                    raise FilterParseError("$type requires one of: " + ", ".join(_TYPE_NAMES))
                conds.append(cls._cond(field, op, ov))
            else:
                raise FilterParseError(f"Unknown operator: {op}")
        return conds[0] if len(conds) == 1 else cls.and_(conds)


def _compare(op: str, fv: Any, v: Any) -> bool:  # compare_values, src/metadata.rs:1244-1337
    if op == "$eq":
        return json_eq(fv, v)
    if op == "$ne":
        return not json_eq(fv, v)
    if op == "$gt":
        return _cmp_numeric(fv, v, lambda a, b: a > b)
    if op == "$gte":
        return _cmp_numeric(fv, v, lambda a, b: a >= b)
    if op == "$lt":
        return _cmp_numeric(fv, v, lambda a, b: a < b)
    if op == "$lte":
        return _cmp_numeric(fv, v, lambda a, b: a <= b)
    if op == "$in":
        return isinstance(v, (list, tuple)) and _contains(v, fv)
    if op == "$nin":
        return not (isinstance(v, (list, tuple)) and _contains(v, fv))
    if op == "$contains":
        if isinstance(fv, str) and isinstance(v, str):
            return v in fv
        return isinstance(fv, (list, tuple)) and _contains(fv, v)
    if op == "$startsWith":
        return isinstance(fv, str) and isinstance(v, str) and fv.startswith(v)
    if op == "$endsWith":
        return isinstance(fv, str) and isinstance(v, str) and fv.endswith(v)
    if op == "$regex":
        return isinstance(fv, str) and isinstance(v, str) and simple_regex_match(fv, v)
    if op == "$all":
        return isinstance(fv, (list, tuple)) and isinstance(v, (list, tuple)) and all(_contains(fv, r) for r in v)
    if op == "$elemMatch":
        if not isinstance(fv, (list, tuple)):
            return False
        try:
            sub = Filter.parse(v)
        except FilterParseError:
            return False
        return any(sub.matches(e) for e in fv)
    if op == "$between":
        return (isinstance(v, (list, tuple)) and len(v) == 2 and _cmp_numeric(fv, v[0], lambda a, b: a >= b)
                and _cmp_numeric(fv, v[1], lambda a, b: a <= b))
    if op == "$size":
        n = v if isinstance(v, int) and not isinstance(v, bool) and v >= 0 else None  # Value::as_u64
        if n is None:
            return False
        if isinstance(fv, (list, tuple)):
            return len(fv) == n
        return isinstance(fv, str) and len(fv.encode()) == n
    if op == "$type":
        return {"string": isinstance(fv, str), "number": _is_number(fv), "boolean": isinstance(fv, bool),
                "array": isinstance(fv, (list, tuple)), "object": isinstance(fv, dict), "null": fv is None}.get(v, False)
    return False  # $exists never reaches compare_values


def value_string(v: Any) -> str:
    """Value::to_string(): compact JSON, the key of the inverted index (src/metadata.rs:327, 389)."""
    return json.dumps(v, separators=(",", ":"), ensure_ascii=False)


def flatten_metadata(value: Any, prefix: str = "", depth: int = 0, out: Optional[List[Tuple[str, Any]]] = None):
    """Nested objects -> dot paths, src/metadata.rs:151-175 (arrays and scalars are leaves)."""
    out = [] if out is None else out
    if depth > FLATTEN_MAX_DEPTH:
        return out
    if isinstance(value, dict):
        for key, val in value.items():
            path = key if not prefix else f"{prefix}.{key}"
            if isinstance(val, dict):
                flatten_metadata(val, path, depth + 1, out)
            else:
                out.append((path, val))
    elif prefix:
        out.append((prefix, value))
    return out


class MetadataStore:
    """The parts of MetadataStore the filter path uses (src/metadata.rs:193-456): entries by internal id and
    the equality inverted index field -> compact JSON of the value -> internal ids."""

    def __init__(self, cardinality_threshold: int = DEFAULT_CARDINALITY_THRESHOLD):
        self.entries: Dict[int, Tuple[str, Optional[Any]]] = {}
        self.id_map: Dict[str, int] = {}
        self.field_indexes: Dict[str, Dict[str, List[int]]] = {}
        self.high_cardinality_fields: Set[str] = set()
        self.cardinality_threshold = cardinality_threshold

    def __len__(self) -> int:
        return len(self.entries)

    def insert(self, internal_id: int, external_id: str, data: Optional[Any]) -> None:  # :299-349
        if external_id in self.id_map:
            raise ValueError(f"Vector already exists: {external_id}")
        if data is not None:
            for field, value in flatten_metadata(data):
                if field in self.high_cardinality_fields:
                    continue
                idx = self.field_indexes.setdefault(field, {})
                idx.setdefault(value_string(value), []).append(internal_id)
                if len(idx) > self.cardinality_threshold:
                    self.high_cardinality_fields.add(field)
                    del self.field_indexes[field]
        self.id_map[external_id] = internal_id
        self.entries[internal_id] = (external_id, data)

    def get(self, internal_id: int) -> Optional[Tuple[str, Optional[Any]]]:
        return self.entries.get(internal_id)

    def get_internal_id(self, external_id: str) -> Optional[int]:
        return self.id_map.get(external_id)

    def delete(self, internal_id: int) -> Optional[Tuple[str, Optional[Any]]]:  # :362-384
        entry = self.entries.pop(internal_id, None)
        if entry is None:
            return None
        self.id_map.pop(entry[0], None)
        if entry[1] is not None:
            for field, value in flatten_metadata(entry[1]):
                ids = self.field_indexes.get(field, {}).get(value_string(value))
                if ids is not None:
                    ids[:] = [i for i in ids if i != internal_id]
        return entry

    def lookup_field_eq(self, field: str, value: Any) -> Optional[List[int]]:  # :388-391
        idx = self.field_indexes.get(field)
        return None if idx is None else idx.get(value_string(value))

    def indexed_fields(self) -> List[str]:
        return list(self.field_indexes)

    def field_cardinality(self, field: str) -> int:
        return len(self.field_indexes.get(field, {}))

    def is_high_cardinality(self, field: str) -> bool:
        return field in self.high_cardinality_fields

    def resolve_filter_via_index(self, f: Filter) -> Optional[Set[int]]:  # :433-456
        if f.kind == "cond" and f.op == "$eq":
            ids = self.lookup_field_eq(f.field, f.value)
            return None if ids is None else set(ids)
        if f.kind == "and":
            result: Optional[Set[int]] = None
            for sub in f.children:
                ids = self.resolve_filter_via_index(sub)
                if ids is None:
                    return None
                result = ids if result is None else (result & ids)
            return result
        return None

    def filter_bitmask(self, f: Filter, n_rows: int) -> np.ndarray:
        """bool[n_rows]: True where the row may be scored.  Equality filters on indexed fields come from
        the inverted index, everything else from Filter::matches per entry; rows without an entry fail
        (src/collection/search_methods.rs:215-223)."""
        keep = np.zeros(n_rows, dtype=np.bool_)
        ids = self.resolve_filter_via_index(f)
        if ids is not None:
            live = [i for i in ids if i in self.entries and i < n_rows]
            keep[live] = True
            return keep
        for i, (_, data) in self.entries.items():
            if i < n_rows and f.matches(data):
                keep[i] = True
        return keep
