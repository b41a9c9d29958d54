#!/usr/bin/env python3
"""Regenerates the `extern "C"` block of bindings/rust/needle-hip-sys/src/lib.rs from include/needle_hip.h, so that the
Rust FFI crate declares every export with the header's exact arity and types.  `--check` exits 1 when lib.rs is stale
(tests/test_rust_binding.py runs it)."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "needle_hip.h")
LIB_RS = os.path.join(ROOT, "bindings", "rust", "needle-hip-sys", "src", "lib.rs")
BEGIN, END = "// ---- generated from include/needle_hip.h by tools/gen_rust_sys.py: begin ----", \
             "// ---- generated from include/needle_hip.h: end ----"

SCALARS = {"needle_hip_status": "c_int", "int": "c_int", "needle_metric": "c_int", "needle_arith": "c_int",
           "uint64_t": "u64", "uint32_t": "u32", "uint16_t": "u16", "uint8_t": "u8", "float": "f32", "double": "f64", "void": "c_void",
           "char": "c_char", "needle_hip_ctx": "needle_hip_ctx", "needle_hip_index": "needle_hip_index",
           "needle_hip_comm": "needle_hip_comm", "needle_search_params": "needle_search_params",
           "needle_hip_metrics": "needle_hip_metrics"}


def prototypes():
    text = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    found = re.findall(r"\n((?:needle_hip_status|void|int|uint64_t|uint32_t|const char\*)\s+(needle_hip_[a-z0-9_]+)\s*\(([^;]*)\);)", text)
    return [(re.sub(r"\s+", " ", full).split(" needle_hip_")[0], name, re.sub(r"\s+", " ", args).strip()) for full, name, args in found]


def rust_type(c: str) -> str:
    c = c.strip()
    stars = c.count("*")
    toks = c.replace("*", " * ").split()
    base = [t for t in toks if t not in ("const", "*")][0]
    r = SCALARS[base]
    if stars == 0:
        return r
    # pointer levels, innermost first: `const T*` -> *const T, `T*` -> *mut T, `const T* const*` -> *const *const T
    consts = []
    seg = c.split("*")
    for i in range(stars):
        consts.append("const" in seg[i].split())
    for is_const in consts:
        r = ("*const " if is_const else "*mut ") + r
    return r


def rust_decl(ret: str, name: str, args: str) -> str:
    params = []
    if args != "void":
        for a in args.split(","):
            a = a.strip()
            m = re.match(r"(.*?)([A-Za-z_][A-Za-z_0-9]*)$", a)
            ctype, pname = m.group(1), m.group(2)
            params.append(f"{pname}: {rust_type(ctype)}")
    r = "" if ret == "void" else " -> " + rust_type(ret)
    return f"    pub fn {name}({', '.join(params)}){r};"


def block() -> str:
    lines = [BEGIN, 'extern "C" {']
    lines += [rust_decl(*p) for p in prototypes()]
    lines += ["}", END]
    return "\n".join(lines)


def main():
    src = open(LIB_RS).read()
    a, b = src.index(BEGIN), src.index(END) + len(END)
    fresh = src[:a] + block() + src[b:]
    if "--check" in sys.argv:
        sys.exit(0 if fresh == src else 1)
    open(LIB_RS, "w").write(fresh)


if __name__ == "__main__":
    main()
