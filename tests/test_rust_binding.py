"""The Rust FFI crate (bindings/rust/needle-hip-sys) is source only here (no Rust toolchain): these checks keep it in
step with what it binds -- the header's exports and structs, and the Makefile's translation units and flags, which
build.rs must reproduce for the library it builds to be the one the tests ran."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CRATE = os.path.join(ROOT, "bindings", "rust", "needle-hip-sys")


def _read(*parts):
    return open(os.path.join(*parts)).read()


def _rust_list(src, name):
    body = src[src.index(f"const {name}: &[&str] = &["):]
    body = body[body.index("&[", body.index("=")) + 2: body.index("];")]
    return re.findall(r'"([^"]+)"', body)


def _make_var(mk, name):
    m = re.search(rf"^{name}\s*[:?]?=\s*((?:.*\\\n)*.*)$", mk, flags=re.M)
    return m.group(1).replace("\\\n", " ").split()


def test_extern_block_is_generated_from_the_header():
    assert subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_rust_sys.py"), "--check"]).returncode == 0
    lib = _read(CRATE, "src", "lib.rs")
    header = re.sub(r"/\*.*?\*/", "", _read(ROOT, "include", "needle_hip.h"), flags=re.S)
    for sym in sorted(set(re.findall(r"\b(needle_hip_[a-z0-9_]+)\s*\(", header))):
        assert re.search(rf"pub fn {sym}\(", lib), sym


def test_build_rs_compiles_what_the_makefile_compiles():
    mk = _read(ROOT, "needle_amd", "csrc", "Makefile")
    rs = _read(CRATE, "build.rs")
    assert _rust_list(rs, "SOURCES") == _make_var(mk, "SRCS")
    want_flags = [f.replace("$(ARCH)", "gfx950") for f in _make_var(mk, "CXXFLAGS")]
    assert _rust_list(rs, "FLAGS") == want_flags
    per_file = re.search(r"^flat_scan_bf16\.o: CXXFLAGS \+= (.*)$", mk, flags=re.M).group(1).split()
    assert _rust_list(rs, "BF16_SCAN_FLAGS") == per_file
    assert re.findall(r"^(\w+)\.o: CXXFLAGS", mk, flags=re.M) == ["flat_scan_bf16"]  # no other per-file flags to mirror
    assert "-lrccl" in _make_var(mk, "LDLIBS") and '"-lrccl"' in rs and "rustc-link-lib=dylib=rccl" in rs
    assert "rustc-link-lib=dylib=amdhip64" in rs and "rustc-link-lib=dylib=needle_hip" in rs


def test_status_and_enum_constants_match_the_header():
    lib = _read(CRATE, "src", "lib.rs")
    header = _read(ROOT, "include", "needle_hip.h")
    for name, value in re.findall(r"\b(NEEDLE_(?:HIP_ERR|HIP_OK|METRIC|ARITH|PRECISION|HALF)[A-Z0-9_]*)\s*=\s*(\d+)", header):
        assert re.search(rf"pub const {name}: (?:c_int|i32) = {value};", lib), name
    assert "pub const NEEDLE_HIP_UNIQUE_ID_BYTES: usize = 128;" in lib and "#define NEEDLE_HIP_UNIQUE_ID_BYTES 128" in header
    assert "precision: NEEDLE_PRECISION_BF16" in lib  # the params constructor defaults to the fast first pass
