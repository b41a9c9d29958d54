// Builds libneedle_hip.so for gfx950 with hipcc from the HIP sources (needle_amd/csrc of the needle_amd repository,
// vendored or pointed to by NEEDLE_HIP_SRC) and links it, together with the HIP runtime and RCCL.
//
// The source list and the flags mirror needle_amd/csrc/Makefile (tests/test_rust_binding.py holds the two together):
//   -ffp-contract=off                          one rounding per multiply and per add in the exact re-scoring kernels
//   -fhip-fp32-correctly-rounded-divide-sqrt   IEEE sqrt and divide, as the reference's scalar Rust
//   flat_scan_bf16.hip additionally            -mllvm -amdgpu-mfma-vgpr-form=1 -fno-honor-nans
use std::{env, path::PathBuf, process::Command};

const SOURCES: &[&str] = &[
    "api.hip", "api_ivf.hip", "api_adc.hip", "flat_scan.hip", "flat_scan_bf16.hip", "flat_post.hip", "prep.hip", "adc.hip",
    "binary.hip", "train.hip", "adaptive.hip", "ops.hip", "comm.hip",
];
const FLAGS: &[&str] = &[
    "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt",
    "-Wall", "-Wno-unused-function",
];
const BF16_SCAN_FLAGS: &[&str] = &["-mllvm", "-amdgpu-mfma-vgpr-form=1", "-fno-honor-nans"];

fn main() {
    let rocm = env::var("ROCM_PATH").unwrap_or_else(|_| "/opt/rocm".into());
    println!("cargo:rustc-link-search=native={rocm}/lib");
    println!("cargo:rustc-link-lib=dylib=amdhip64");
    println!("cargo:rustc-link-lib=dylib=rccl");
    println!("cargo:rerun-if-env-changed=NEEDLE_HIP_SRC");
    println!("cargo:rerun-if-env-changed=NEEDLE_HIP_LIB_DIR");
    if env::var("CARGO_FEATURE_PREBUILT").is_ok() {
        let dir = env::var("NEEDLE_HIP_LIB_DIR").expect("feature `prebuilt` needs NEEDLE_HIP_LIB_DIR");
        println!("cargo:rustc-link-search=native={dir}");
        println!("cargo:rustc-link-lib=dylib=needle_hip");
        return;
    }
    let manifest = PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap());
    // default: the crate sits in bindings/rust/needle-hip-sys of the needle_amd repository
    let src = env::var("NEEDLE_HIP_SRC").map(PathBuf::from).unwrap_or_else(|_| manifest.join("../../../needle_amd/csrc"));
    let out = PathBuf::from(env::var("OUT_DIR").unwrap());
    let hipcc = env::var("HIPCC").unwrap_or_else(|_| format!("{rocm}/bin/hipcc"));
    let mut objs = vec![];
    for f in SOURCES {
        let o = out.join(f.replace(".hip", ".o"));
        let mut cmd = Command::new(&hipcc);
        cmd.args(FLAGS);
        if *f == "flat_scan_bf16.hip" {
            cmd.args(BF16_SCAN_FLAGS);
        }
        let st = cmd.arg("-c").arg(src.join(f)).arg("-o").arg(&o).status().expect("hipcc not found");
        assert!(st.success(), "hipcc failed on {f}");
        println!("cargo:rerun-if-changed={}", src.join(f).display());
        objs.push(o);
    }
    let so = out.join("libneedle_hip.so");
    let st = Command::new(&hipcc)
        .args(["--offload-arch=gfx950", "-shared", "-fPIC", "-o"])
        .arg(&so)
        .args(&objs)
        .arg(format!("-L{rocm}/lib"))
        .arg("-lrccl")
        .status()
        .expect("hipcc not found");
    assert!(st.success(), "linking libneedle_hip.so failed");
    println!("cargo:rustc-link-search=native={}", out.display());
    println!("cargo:rustc-link-lib=dylib=needle_hip");
}
