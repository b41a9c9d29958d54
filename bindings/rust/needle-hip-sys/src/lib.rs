//! FFI to `libneedle_hip.so` (C ABI: `include/needle_hip.h` of the needle_amd repository): Needle's batched
//! vector-search hot path on AMD MI355X.  Plain pointers and sizes; every call returns a status (0 = ok) and leaves
//! a thread-local message for [`needle_hip_last_error`].  The library has no CPU path: without a gfx950 device
//! [`needle_hip_init`] fails with [`NEEDLE_HIP_ERR_NO_DEVICE`] and the caller routes the request to its own CPU code
//! (the contract of `TransparentFallbackManager`, src/experimental/gpu/mod.rs:1427-1474).
#![allow(non_camel_case_types)]
use std::ffi::CStr;
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)]
pub struct needle_hip_ctx {
    _p: [u8; 0],
}
#[repr(C)]
pub struct needle_hip_index {
    _p: [u8; 0],
}
#[repr(C)]
pub struct needle_hip_comm {
    _p: [u8; 0],
}

// needle_hip_status
pub const NEEDLE_HIP_OK: c_int = 0;
pub const NEEDLE_HIP_ERR_DIMENSION_MISMATCH: c_int = 1; // NeedleError::DimensionMismatch
pub const NEEDLE_HIP_ERR_INVALID_VECTOR: c_int = 2; // NeedleError::InvalidVector
pub const NEEDLE_HIP_ERR_INVALID_ARGUMENT: c_int = 3;
pub const NEEDLE_HIP_ERR_OUT_OF_MEMORY: c_int = 4;
pub const NEEDLE_HIP_ERR_HIP: c_int = 5;
pub const NEEDLE_HIP_ERR_NOT_TRAINED: c_int = 6; // IvfError::NotTrained
pub const NEEDLE_HIP_ERR_UNSUPPORTED: c_int = 7;
pub const NEEDLE_HIP_ERR_NO_DEVICE: c_int = 8;
pub const NEEDLE_HIP_ERR_RCCL: c_int = 9;

// needle_metric: `DistanceFunction as c_int` is valid as is (same declaration order, src/distance.rs:43-73)
pub const NEEDLE_METRIC_COSINE: c_int = 0;
pub const NEEDLE_METRIC_COSINE_NORMALIZED: c_int = 1;
pub const NEEDLE_METRIC_EUCLIDEAN: c_int = 2;
pub const NEEDLE_METRIC_DOT_PRODUCT: c_int = 3;
pub const NEEDLE_METRIC_MANHATTAN: c_int = 4;
pub const NEEDLE_METRIC_HAMMING: c_int = 5;
pub const NEEDLE_METRIC_CHEBYSHEV: c_int = 6;

// needle_arith
pub const NEEDLE_ARITH_DISTANCE_RS: c_int = 0; // src/distance.rs sequential sums: Collection::search / batch_search
pub const NEEDLE_ARITH_ACCEL_CHUNK4: c_int = 1; // src/experimental/gpu/common.rs groups of four: GpuAccelerator
pub const NEEDLE_ARITH_SERVICE: c_int = 2; // src/services/compute/gpu_kernels.rs:345-358 (operator-level calls only)
pub const NEEDLE_ARITH_AVX2: c_int = 3; // src/distance.rs:481-591: what a build with AVX2 enabled computes
pub const NEEDLE_ARITH_AVX512: c_int = 4; // src/distance.rs:613-715: what a build with AVX-512 enabled computes
pub const NEEDLE_ARITH_NEON: c_int = 5; // src/distance.rs:741-834: what an aarch64 build computes

// needle_precision: all three return the same ids and distances; BF16 is the fast first pass
pub const NEEDLE_PRECISION_FP32: i32 = 0;
pub const NEEDLE_PRECISION_BF16: i32 = 1;
pub const NEEDLE_PRECISION_EXACT_SCAN: i32 = 2;

pub const NEEDLE_HALF_F16: c_int = 0;
pub const NEEDLE_HALF_BF16: c_int = 1;
pub const NEEDLE_HIP_UNIQUE_ID_BYTES: usize = 128;

#[repr(C)]
pub struct needle_search_params {
    pub k: u32,
    pub nprobe: u32,
    pub precision: i32,
    pub reserved: u32,
    pub filter_bitmask: *const u32,
}

impl needle_search_params {
    /// k nearest with the bf16 first pass (the default a maintainer wants: same results as the other two, fastest).
    pub fn new(k: u32) -> Self {
        Self { k, nprobe: 0, precision: NEEDLE_PRECISION_BF16, reserved: 0, filter_bitmask: std::ptr::null() }
    }
}

/// Field for field the struct of include/needle_hip.h: needle_hip_get_metrics writes all of it, so a shorter mirror
/// is an out-of-bounds write (tests/test_cabi_load.py checks this block against the header).
#[repr(C)]
#[derive(Default, Debug, Clone, Copy)]
pub struct needle_hip_metrics {
    pub searches: u64,
    pub queries: u64,
    pub fallback_queries: u64,
    pub scan_launches: u64,
    pub scan_ms_total: f64,
    pub last_scan_ms: f64,
    pub last_search_ms: f64,
    pub refined_queries: u64,
    pub binary_code_scans: u64,
    pub list_scans: u64,
    pub list_scan_ms_total: f64,
    pub compacted_searches: u64,
    pub compacted_rows: u64,
}

// ---- generated from include/needle_hip.h by tools/gen_rust_sys.py: begin ----
extern "C" {
    pub fn needle_hip_init(device_id: c_int, out_ctx: *mut *mut needle_hip_ctx) -> c_int;
    pub fn needle_hip_shutdown(ctx: *mut needle_hip_ctx);
    pub fn needle_hip_last_error() -> *const c_char;
    pub fn needle_hip_version() -> c_int;
    pub fn needle_hip_set_profiling(ctx: *mut needle_hip_ctx, on: c_int) -> c_int;
    pub fn needle_hip_get_metrics(ctx: *mut needle_hip_ctx, out: *mut needle_hip_metrics) -> c_int;
    pub fn needle_hip_reset_metrics(ctx: *mut needle_hip_ctx) -> c_int;
    pub fn needle_hip_set_tuning(ctx: *mut needle_hip_ctx, key: *const c_char, value: f64) -> c_int;
    pub fn needle_hip_metric_from_str(name: *const c_char) -> c_int;
    pub fn needle_hip_flat_build(ctx: *mut needle_hip_ctx, base: *const f32, n: u64, dim: u32, metric: c_int, arith: c_int, valid_bitmask: *const u32, out: *mut *mut needle_hip_index) -> c_int;
    pub fn needle_hip_flat_build_rows(ctx: *mut needle_hip_ctx, rows: *const *const f32, n: u64, dim: u32, metric: c_int, arith: c_int, valid_bitmask: *const u32, out: *mut *mut needle_hip_index) -> c_int;
    pub fn needle_hip_flat_build_device(ctx: *mut needle_hip_ctx, d_base: *const f32, n: u64, dim: u32, metric: c_int, arith: c_int, valid_bitmask_host: *const u32, out: *mut *mut needle_hip_index) -> c_int;
    pub fn needle_hip_flat_build_half(ctx: *mut needle_hip_ctx, rows: *const u16, n: u64, dim: u32, half_kind: c_int, metric: c_int, arith: c_int, valid_bitmask: *const u32, out: *mut *mut needle_hip_index) -> c_int;
    pub fn needle_hip_index_set_valid_mask(index: *mut needle_hip_index, valid_bitmask: *const u32) -> c_int;
    pub fn needle_hip_index_append(index: *mut needle_hip_index, rows: *const f32, n_new: u64, new_valid_bitmask: *const u32) -> c_int;
    pub fn needle_hip_index_append_device(index: *mut needle_hip_index, d_rows: *const f32, n_new: u64, new_valid_bitmask_host: *const u32) -> c_int;
    pub fn needle_hip_index_swap_remove(index: *mut needle_hip_index, pos: u64) -> c_int;
    pub fn needle_hip_index_destroy(index: *mut needle_hip_index);
    pub fn needle_hip_index_len(index: *const needle_hip_index) -> u64;
    pub fn needle_hip_index_dimensions(index: *const needle_hip_index) -> u32;
    pub fn needle_hip_batch_search(index: *mut needle_hip_index, queries: *const f32, q: u32, query_dim: u32, params: *const needle_search_params, out_ids: *mut u64, out_dists: *mut f32, out_counts: *mut u32) -> c_int;
    pub fn needle_hip_batch_search_device(index: *mut needle_hip_index, d_queries: *const f32, q: u32, params: *const needle_search_params, d_out_ids: *mut u64, d_out_dists: *mut f32, d_out_counts: *mut u32) -> c_int;
    pub fn needle_hip_ivf_build(ctx: *mut needle_hip_ctx, base: *const f32, n: u64, dim: u32, centroids: *const f32, nlist: u32, assignment: *const u32, valid_bitmask: *const u32, out: *mut *mut needle_hip_index) -> c_int;
    pub fn needle_hip_ivf_probe(index: *mut needle_hip_index, queries: *const f32, q: u32, query_dim: u32, nprobe: u32, out_list_ids: *mut u32, out_counts: *mut u32) -> c_int;
    pub fn needle_hip_ivf_list_sizes(index: *const needle_hip_index, out_sizes: *mut u32) -> c_int;
    pub fn needle_hip_flat_build_prefix(full: *mut needle_hip_index, dims: u32, out: *mut *mut needle_hip_index) -> c_int;
    pub fn needle_hip_adaptive_search(full: *mut needle_hip_index, coarse: *mut needle_hip_index, queries: *const f32, q: u32, query_dim: u32, k: u32, overfetch: u32, out_ids: *mut u64, out_dists: *mut f32, out_coarse_dists: *mut f32, out_counts: *mut u32) -> c_int;
    pub fn needle_hip_ivf_train(ctx: *mut needle_hip_ctx, vectors: *const f32, n: u64, dim: u32, n_clusters: u32, max_iterations: u32, tolerance: f32, seed: u64, out_centroids: *mut f32, out_k: *mut u32) -> c_int;
    pub fn needle_hip_pq_train(ctx: *mut needle_hip_ctx, vectors: *const f32, n: u64, dim: u32, m: u32, iterations: u32, seed: u64, out_codebooks: *mut f32, out_ncent: *mut u32) -> c_int;
    pub fn needle_hip_pq_encode(ctx: *mut needle_hip_ctx, codebooks: *const f32, dim: u32, m: u32, ncent: *const u32, vectors: *const f32, n: u64, out_codes: *mut u8) -> c_int;
    pub fn needle_hip_ivf_pq_build(ctx: *mut needle_hip_ctx, base: *const f32, n: u64, dim: u32, centroids: *const f32, nlist: u32, assignment: *const u32, valid_bitmask: *const u32, m: u32, codebooks: *const f32, out: *mut *mut needle_hip_index) -> c_int;
    pub fn needle_hip_pq_build(ctx: *mut needle_hip_ctx, codes: *const u8, n: u64, dim: u32, m: u32, codebooks: *const f32, valid_bitmask: *const u32, out: *mut *mut needle_hip_index) -> c_int;
    pub fn needle_hip_binary_build(ctx: *mut needle_hip_ctx, codes: *const u8, n: u64, dim: u32, thresholds: *const f32, valid_bitmask: *const u32, out: *mut *mut needle_hip_index) -> c_int;
    pub fn needle_hip_binary_quantize(ctx: *mut needle_hip_ctx, thresholds: *const f32, vectors: *const f32, n: u64, dim: u32, out_codes: *mut u8) -> c_int;
    pub fn needle_hip_sq8_build(ctx: *mut needle_hip_ctx, codes: *const u8, n: u64, dim: u32, min_vals: *const f32, scale: *const f32, valid_bitmask: *const u32, out: *mut *mut needle_hip_index) -> c_int;
    pub fn needle_hip_batch_distance(ctx: *mut needle_hip_ctx, metric: c_int, arith: c_int, query: *const f32, base: *const f32, n: u64, dim: u32, out: *mut f32) -> c_int;
    pub fn needle_hip_kmeans_assign(ctx: *mut needle_hip_ctx, arith: c_int, vectors: *const f32, n: u64, centroids: *const f32, n_centroids: u32, dim: u32, out_assignment: *mut u32) -> c_int;
    pub fn needle_hip_merge_topk_device(ctx: *mut needle_hip_ctx, d_dists: *const f32, d_ids: *const u64, d_counts: *const u32, lists: u32, q: u32, per: u32, k: u32, d_out_ids: *mut u64, d_out_dists: *mut f32, d_out_counts: *mut u32) -> c_int;
    pub fn needle_hip_merge_topk(ctx: *mut needle_hip_ctx, dists: *const f32, ids: *const u64, counts: *const u32, lists: u32, q: u32, per: u32, k: u32, out_ids: *mut u64, out_dists: *mut f32, out_counts: *mut u32) -> c_int;
    pub fn needle_hip_normalize(ctx: *mut needle_hip_ctx, arith: c_int, vectors: *mut f32, n: u64, dim: u32) -> c_int;
    pub fn needle_hip_normalize_device(ctx: *mut needle_hip_ctx, arith: c_int, d_vectors: *mut f32, n: u64, dim: u32) -> c_int;
    pub fn needle_hip_batch_top_k(ctx: *mut needle_hip_ctx, scores: *const f32, rows: u32, n: u64, k: u32, out_idx: *mut u64, out_scores: *mut f32, out_counts: *mut u32) -> c_int;
    pub fn needle_hip_batch_top_k_device(ctx: *mut needle_hip_ctx, d_scores: *const f32, rows: u32, n: u64, k: u32, d_out_idx: *mut u64, d_out_scores: *mut f32, d_out_counts: *mut u32) -> c_int;
    pub fn needle_hip_distance_matrix(ctx: *mut needle_hip_ctx, metric: c_int, arith: c_int, queries: *const f32, q: u32, base: *const f32, n: u64, dim: u32, out: *mut f32) -> c_int;
    pub fn needle_hip_distance_matrix_device(ctx: *mut needle_hip_ctx, metric: c_int, arith: c_int, d_queries: *const f32, q: u32, d_base: *const f32, n: u64, dim: u32, d_out: *mut f32) -> c_int;
    pub fn needle_hip_multi_query_search(ctx: *mut needle_hip_ctx, metric: c_int, arith: c_int, queries: *const f32, q: u32, base: *const f32, n: u64, dim: u32, k: u32, out_idx: *mut u64, out_dists: *mut f32, out_counts: *mut u32) -> c_int;
    pub fn needle_hip_multi_query_search_device(ctx: *mut needle_hip_ctx, metric: c_int, arith: c_int, d_queries: *const f32, q: u32, d_base: *const f32, n: u64, dim: u32, k: u32, d_out_idx: *mut u64, d_out_dists: *mut f32, d_out_counts: *mut u32) -> c_int;
    pub fn needle_hip_comm_unique_id(out_id: *mut u8) -> c_int;
    pub fn needle_hip_comm_init(ctx: *mut needle_hip_ctx, unique_id: *const u8, rank: c_int, world: c_int, out: *mut *mut needle_hip_comm) -> c_int;
    pub fn needle_hip_comm_destroy(comm: *mut needle_hip_comm);
    pub fn needle_hip_comm_rank(comm: *const needle_hip_comm) -> c_int;
    pub fn needle_hip_comm_world(comm: *const needle_hip_comm) -> c_int;
    pub fn needle_hip_sharded_batch_search_device(shard: *mut needle_hip_index, comm: *mut needle_hip_comm, row_offset: u64, d_queries: *const f32, q: u32, params: *const needle_search_params, d_out_ids: *mut u64, d_out_dists: *mut f32, d_out_counts: *mut u32) -> c_int;
    pub fn needle_hip_shard_slot_bytes(q: u32, k: u32) -> u64;
    pub fn needle_hip_merge_shard_slots_device(ctx: *mut needle_hip_ctx, d_slots: *const c_void, lists: u32, q: u32, k: u32, d_out_ids: *mut u64, d_out_dists: *mut f32, d_out_counts: *mut u32) -> c_int;
    pub fn needle_hip_batch_top_k_device_noalloc(ctx: *mut needle_hip_ctx, d_scores: *const f32, rows: u32, n: u64, k: u32, d_out_idx: *mut u64, d_out_scores: *mut f32, d_out_counts: *mut u32) -> c_int;
}
// ---- generated from include/needle_hip.h: end ----

/// The calling thread's last error message.
pub fn last_error() -> String {
    // SAFETY: the library returns a pointer to a NUL-terminated thread-local string that lives until the thread's
    // next failing call.
    unsafe { CStr::from_ptr(needle_hip_last_error()).to_string_lossy().into_owned() }
}

/// `Ok(())` for NEEDLE_HIP_OK, else the status with the library's message.
pub fn check(status: c_int) -> Result<(), (c_int, String)> {
    if status == NEEDLE_HIP_OK {
        Ok(())
    } else {
        Err((status, last_error()))
    }
}
