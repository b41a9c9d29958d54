/*
 * needle_hip.h -- C ABI of libneedle_hip.so: batched vector search on AMD
 * MI355X (gfx950) behind the operator boundary of the Needle vector database.
 *
 * Every entry point names the reference interface it stands in for
 * (file:line relative to the reference repository).  Plain pointers and
 * sizes only; no exceptions cross this boundary; every call returns a
 * needle_hip_status and leaves a thread-local message for
 * needle_hip_last_error().
 *
 * Ownership: inputs are borrowed for the duration of the call.  Index builds
 * COPY the caller's data into device memory (the caller keeps its VectorStore,
 * src/storage.rs:624-629).  Outputs are written into caller-allocated arrays.
 * Handles are opaque and released by the matching *_destroy / shutdown call.
 * needle_hip_shutdown takes along the indexes still bound to the context.  A
 * context or index handle that is already dead may be passed to its destroy
 * call again: the address of a destroyed handle is never reused, so the call
 * is a no-op.  A communicator is destroyed exactly once, before its context.
 *
 * Streams: every call works on a stream the context owns and returns after its
 * work completed.  Entry points that take DEVICE pointers do not order
 * themselves behind the caller's streams: whatever produces those inputs must
 * have finished (stream or device synchronised) before the call.
 *
 * Threading: needle_hip_batch_search[_device] (and the sharded form) are
 * re-entrant on one index, as Collection::batch_search is called under a read
 * lock from many threads (src/database/internal.rs:319): every call runs on a
 * search slot of its own -- the index's resident data, a private workspace and
 * a private stream -- so concurrent batches overlap on the device; at most 4
 * slots per index exist, a fifth concurrent call waits.  Builds, mask updates
 * and destroy need exclusive access, like the reference's &mut self methods.
 * The remaining per-index calls (ivf_probe, adaptive_search) serialise on an
 * internal lock.
 */
#ifndef NEEDLE_HIP_H
#define NEEDLE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum needle_hip_status {
    NEEDLE_HIP_OK = 0,
    /* NeedleError::DimensionMismatch{expected,got}, src/error.rs; raised by
     * validate_query, src/collection/validation.rs:27-35 */
    NEEDLE_HIP_ERR_DIMENSION_MISMATCH = 1,
    /* NeedleError::InvalidVector (NaN / Inf), validation.rs:8-24 */
    NEEDLE_HIP_ERR_INVALID_VECTOR = 2,
    NEEDLE_HIP_ERR_INVALID_ARGUMENT = 3,
    NEEDLE_HIP_ERR_OUT_OF_MEMORY = 4,
    NEEDLE_HIP_ERR_HIP = 5,          /* a HIP runtime call failed */
    NEEDLE_HIP_ERR_NOT_TRAINED = 6,  /* IvfError::NotTrained, src/indexing/ivf.rs:79 */
    NEEDLE_HIP_ERR_UNSUPPORTED = 7,
    NEEDLE_HIP_ERR_NO_DEVICE = 8,
    NEEDLE_HIP_ERR_RCCL = 9          /* an RCCL call failed */
} needle_hip_status;

/* DistanceFunction, src/distance.rs:43-73 (same order). */
typedef enum needle_metric {
    NEEDLE_METRIC_COSINE = 0,
    NEEDLE_METRIC_COSINE_NORMALIZED = 1,
    NEEDLE_METRIC_EUCLIDEAN = 2,
    NEEDLE_METRIC_DOT_PRODUCT = 3,
    NEEDLE_METRIC_MANHATTAN = 4,
    NEEDLE_METRIC_HAMMING = 5,
    NEEDLE_METRIC_CHEBYSHEV = 6
} needle_metric;

/* Which of the reference's arithmetics the reported distances (and therefore
 * the ranking) reproduce bit for bit:
 *   DISTANCE_RS : src/distance.rs:153-316 sequential sums -- what
 *                 Collection::search / batch_search / compute_ground_truth use
 *   ACCEL_CHUNK4: src/experimental/gpu/common.rs:515-581 groups of four --
 *                 what GpuAccelerator::{batch_*_distance, fused_search} use
 *                 (cosine / euclidean / dot only)
 *   SERVICE     : src/services/compute/gpu_kernels.rs:345-358, the compute
 *                 service's kernels -- cosine as one loop over three sums from
 *                 +0.0 with `denominator < f32::EPSILON -> 1.0`; euclidean and
 *                 dot product as DISTANCE_RS.  Operator-level calls only
 *                 (batch_distance, distance_matrix, multi_query_search).
 *   AVX2        : src/distance.rs:481-591, what DISTANCE_RS's functions compute
 *                 in a reference BUILT WITH AVX2 ENABLED (-C target-cpu=native):
 *                 eight lane sums of fused multiply-adds, folded as hsum_avx
 *                 (:489-496), the len % 8 tail added after the fold.  Cosine,
 *                 cosine-normalized, euclidean, dot product and manhattan
 *                 differ from DISTANCE_RS; hamming and chebyshev do not.  Flat
 *                 indexes, normalize and the operator-level calls.
 *   AVX512      : src/distance.rs:613-715, the same in a build with AVX-512
 *                 enabled: sixteen lane sums, folded l_i + l_(i+8) and then as
 *                 hsum_avx (:619-627), the len % 16 tail after the fold.
 *   NEON        : src/distance.rs:741-834, what an aarch64 build computes (its
 *                 only path there): four lanes of vfmaq_f32, folded by
 *                 vaddvq_f32 -- (l0 + l1) + (l2 + l3) --, the len % 4 tail
 *                 after the fold. */
typedef enum needle_arith {
    NEEDLE_ARITH_DISTANCE_RS = 0,
    NEEDLE_ARITH_ACCEL_CHUNK4 = 1,
    NEEDLE_ARITH_SERVICE = 2,
    NEEDLE_ARITH_AVX2 = 3,
    NEEDLE_ARITH_AVX512 = 4,
    NEEDLE_ARITH_NEON = 5
} needle_arith;

/* Arithmetic of the candidate-generating first pass.  The returned ids and
 * distances never depend on it: every candidate is re-scored in the
 * reference's scalar order, a per-query error-bound check proves the
 * candidate set sufficient, and queries that fail it are re-run on the exact
 * scan.  EXACT_SCAN skips the first pass (every query on the exact scan). */
typedef enum needle_precision {
    NEEDLE_PRECISION_FP32 = 0,
    NEEDLE_PRECISION_BF16 = 1,
    NEEDLE_PRECISION_EXACT_SCAN = 2
} needle_precision;

typedef struct needle_hip_ctx needle_hip_ctx;
typedef struct needle_hip_index needle_hip_index;

typedef struct needle_search_params {
    uint32_t k;          /* clamped to the live row count, validation.rs:39-41 */
    uint32_t nprobe;     /* IVF only: IvfConfig::n_probe, ivf.rs:112 */
    int32_t precision;   /* needle_precision */
    uint32_t reserved;
    /* Pre-filter bitmask over internal row ids (bit i of word i>>5 at i&31; set =
     * row may be returned), NULL = no filter.  Exact pre-filter semantics of
     * Collection::brute_force_search, src/collection/search_methods.rs:215-223,
     * and GpuAccelerator::prefiltered_batch_search, gpu/mod.rs:1033-1080.  Host
     * pointer for needle_hip_batch_search, device pointer for the *_device form.
     * A flat index whose masks (deleted rows and this filter) admit at most half
     * of its rows is searched over a gathered copy of the admitted rows -- the
     * reference scores only matching rows, too -- with the same hits; the
     * metrics count such searches (compacted_searches). */
    const uint32_t* filter_bitmask;
} needle_search_params;

/* GpuMetrics / update_metrics, src/experimental/gpu/common.rs:246-259,
 * gpu/mod.rs:825-845, restricted to what the device path can count. */
typedef struct needle_hip_metrics {
    uint64_t searches;            /* batch search calls */
    uint64_t queries;             /* queries served */
    uint64_t fallback_queries;    /* queries re-run on the exact scan */
    uint64_t scan_launches;       /* first-pass kernel launches timed */
    double scan_ms_total;         /* device time of those launches (needs profiling on) */
    double last_scan_ms;
    double last_search_ms;        /* device time of the last whole search (profiling on) */
    uint64_t refined_queries;     /* queries a bf16 first pass left undecided, re-run with the fp32 first pass */
    uint64_t binary_code_scans;   /* full passes over a binary code matrix (2 or 3 per batch: see needle_hip_binary_build) */
    uint64_t list_scans;          /* IVF batches whose inverted-list scan was timed (profiling on) */
    double list_scan_ms_total;    /* device time of those list scans: threshold sample, list-major scan launches, refreshes */
    uint64_t compacted_searches;  /* filtered flat searches served from a gathered copy of the rows the masks admit */
    uint64_t compacted_rows;      /* rows of those copies, summed */
} needle_hip_metrics;

/* ---- context --------------------------------------------------------- */

/* GpuAccelerator::new / with_cpu_fallback, src/experimental/gpu/mod.rs:103-132:
 * binds one HIP device.  There is NO CPU fallback here: without a usable
 * device the call fails with NEEDLE_HIP_ERR_NO_DEVICE. */
needle_hip_status needle_hip_init(int device_id, needle_hip_ctx** out_ctx);
void needle_hip_shutdown(needle_hip_ctx* ctx);

/* Message of the calling thread's last failing call ("" if none). */
const char* needle_hip_last_error(void);

/* Library / ABI version (major*10000 + minor*100 + patch). */
int needle_hip_version(void);

/* When on, the first-pass kernel and the whole search are bracketed by HIP
 * events on the library's stream (adds a stream sync per search); the events
 * belong to the context, so searches of all its indexes then run one at a time. */
needle_hip_status needle_hip_set_profiling(needle_hip_ctx* ctx, int on);
needle_hip_status needle_hip_get_metrics(needle_hip_ctx* ctx, needle_hip_metrics* out);
needle_hip_status needle_hip_reset_metrics(needle_hip_ctx* ctx);

/* Planning overrides for the searches run through this context; no counterpart
 * in the reference (its GpuConfig, src/experimental/gpu/common.rs:143-175, holds
 * the analogous knobs).  The library reads no environment variable: tests use
 * this call to force a code path, the A/B tools to sweep a parameter.  Overrides
 * move plans and cost, never results -- with one exception, the timing
 * experiments among the "debug" bits.  Keys: "splits", "pre_tiles", "kseed",
 * "band", "stage0".."stage3" (percent of a slice per staged launch),
 * "no_dense", "no_set_only", "merge_keys", "merge_lists", "rs_cap", "rs_dc",
 * "ivf_no_seed", "ivf_seed_rows", "ivf_split", "binary_two_scans", "no_compact",
 * "compact_pct", "compact_min_rows", "debug" (bit 1: no threshold seeding, bit 2:
 * one launch per slice, bit 3: print what the first pass listed, bit 4: no
 * candidate pruning; bits 0, 8 and 9 switch off the selection, the table refill
 * and the table lookups of a scan to time what is left: a search run under them
 * returns WRONG results), "trace" and "poison" (process-wide), "reset" (back to
 * the built-in choices).  Unknown keys fail with
 * NEEDLE_HIP_ERR_INVALID_ARGUMENT. */
needle_hip_status needle_hip_set_tuning(needle_hip_ctx* ctx, const char* key, double value);

/* FromStr for DistanceFunction, src/distance.rs:132-145; -1 if unknown. */
int needle_hip_metric_from_str(const char* name);

/* ---- exact flat index -------------------------------------------------- */

/* Upload of the base set that Collection::search_builder().distance(..) /
 * brute_force_search scan, src/collection/search_methods.rs:189-271, and that
 * GpuResidentIndex::{new,add_batch} keep resident, gpu/mod.rs:1102-1160.
 * base: n x dim row-major f32 host memory.  valid_bitmask: nullable, bit set =
 * live row (cleared = deleted / expired, search_methods.rs:205-212). */
needle_hip_status needle_hip_flat_build(needle_hip_ctx* ctx, const float* base, uint64_t n,
                                        uint32_t dim, needle_metric metric, needle_arith arith,
                                        const uint32_t* valid_bitmask, needle_hip_index** out);

/* Same, from the reference's own layout: one heap allocation per row
 * (VectorStore::as_slice() -> &[Vec<f32>], src/storage.rs:689). */
needle_hip_status needle_hip_flat_build_rows(needle_hip_ctx* ctx, const float* const* rows,
                                             uint64_t n, uint32_t dim, needle_metric metric,
                                             needle_arith arith, const uint32_t* valid_bitmask,
                                             needle_hip_index** out);

/* Same, from a device-resident n x dim row-major f32 array (copied). */
needle_hip_status needle_hip_flat_build_device(needle_hip_ctx* ctx, const float* d_base,
                                               uint64_t n, uint32_t dim, needle_metric metric,
                                               needle_arith arith, const uint32_t* valid_bitmask_host,
                                               needle_hip_index** out);

/* Same, from rows stored in half precision: F16Vector / Bf16Vector of
 * src/indexing/float16.rs (:261-275, :335-358).  rows: n x dim host 16-bit patterns;
 * they are widened on the device exactly as F16::to_f32 / Bf16::to_f32 do
 * (:101-136, :223-225), and every distance is then the f32 arithmetic on the
 * widened values, as the reference's half vectors compute theirs (:294-326). */
typedef enum needle_half_kind { NEEDLE_HALF_F16 = 0, NEEDLE_HALF_BF16 = 1 } needle_half_kind;
needle_hip_status needle_hip_flat_build_half(needle_hip_ctx* ctx, const uint16_t* rows, uint64_t n, uint32_t dim,
                                             int half_kind, needle_metric metric, needle_arith arith,
                                             const uint32_t* valid_bitmask, needle_hip_index** out);

/* HnswIndex::delete / is_deleted bookkeeping seen by the scan
 * (search_methods.rs:205): replace the live-row mask (NULL = all live). */
needle_hip_status needle_hip_index_set_valid_mask(needle_hip_index* index,
                                                  const uint32_t* valid_bitmask);

/* GpuResidentIndex::add / add_batch, src/experimental/gpu/mod.rs:1102-1160: rows
 * appended behind the resident ones of a FLAT index; row i of the batch becomes
 * internal id len + i.  Only the new rows cross PCIe: their operands, norms,
 * seeds and error bounds are derived on the device and the resident arrays grow
 * in place (capacity doubles when it runs out, moved device to device).  The
 * result of any later search equals that of an index built from all rows at
 * once.  new_valid_bitmask (host, bit j = row j of the batch may be returned;
 * NULL = all) extends the live-row mask.  NaN / Inf in the batch fails with
 * NEEDLE_HIP_ERR_INVALID_VECTOR and leaves the index unchanged.  Like `&mut
 * self` in the reference: no search of this index may run during the call.
 * Inverted-file and quantized indexes answer NEEDLE_HIP_ERR_UNSUPPORTED. */
needle_hip_status needle_hip_index_append(needle_hip_index* index, const float* rows, uint64_t n_new,
                                          const uint32_t* new_valid_bitmask);
needle_hip_status needle_hip_index_append_device(needle_hip_index* index, const float* d_rows, uint64_t n_new,
                                                 const uint32_t* new_valid_bitmask_host);
/* GpuResidentIndex::remove, src/experimental/gpu/mod.rs:1198-1207
 * (Vec::swap_remove): the last row of a FLAT index takes internal id `pos`, the
 * index shrinks by one row.  Same exclusivity as needle_hip_index_append. */
needle_hip_status needle_hip_index_swap_remove(needle_hip_index* index, uint64_t pos);
void needle_hip_index_destroy(needle_hip_index* index);

uint64_t needle_hip_index_len(const needle_hip_index* index);        /* Collection::len */
uint32_t needle_hip_index_dimensions(const needle_hip_index* index); /* config.dimensions */

/* Collection::batch_search, src/collection/batch.rs:32-75 (and, with
 * params->filter_bitmask, the exact-pre-filter form of batch_search_with_filter
 * :108-171); GpuAccelerator::batch_fused_search, gpu/mod.rs:790-823.
 * queries: q x query_dim row-major host f32.  Every query is validated first
 * (dimension, NaN, Inf) and any failure rejects the whole batch before work
 * starts, as batch.rs:37-39.  Outputs are [q][k] row-major; row i holds
 * out_counts[i] = min(k, live filtered rows) hits ascending by (distance, id),
 * the rest of the row is padded with id UINT64_MAX / distance +inf. */
needle_hip_status needle_hip_batch_search(needle_hip_index* index, const float* queries,
                                          uint32_t q, uint32_t query_dim,
                                          const needle_search_params* params,
                                          uint64_t* out_ids, float* out_dists,
                                          uint32_t* out_counts);

/* Device-resident form of the same call: queries / outputs / filter are device
 * pointers, work is enqueued on the library stream and the call returns after
 * it completed.  Values are not validated (no NaN / Inf scan). */
needle_hip_status needle_hip_batch_search_device(needle_hip_index* index, const float* d_queries,
                                                 uint32_t q, const needle_search_params* params,
                                                 uint64_t* d_out_ids, float* d_out_dists,
                                                 uint32_t* d_out_counts);

/* ---- IVF index --------------------------------------------------------- */

/* IvfIndex after train() + insert() of every row, src/indexing/ivf.rs:299-327,
 * 462-498, for GIVEN centroids (the reference trains them with an unseeded
 * k-means, ivf.rs:330-450; training output is not part of the parity contract).
 * base: n x dim host f32; centroids: nlist x dim host f32; assignment: nullable,
 * assignment[i] = list of row i; NULL -> computed as find_nearest_centroid,
 * ivf.rs:453-459 (first nearest centroid by Euclidean distance).  Rows are
 * stored list by list in insertion (row id) order, IVF-Flat, ivf.rs:181-196.
 * Distances are always Euclidean in distance.rs arithmetic, ivf.rs:67,551. */
needle_hip_status needle_hip_ivf_build(needle_hip_ctx* ctx, const float* base, uint64_t n, uint32_t dim,
                                       const float* centroids, uint32_t nlist,
                                       const uint32_t* assignment, const uint32_t* valid_bitmask,
                                       needle_hip_index** out);

/* Coarse step of IvfIndex::search, ivf.rs:521-533: for each query the
 * min(nprobe, nlist) lists nearest by Euclidean distance, ascending, ties to
 * the lower list id.  out_list_ids is [q][nprobe] (unused tail = UINT32_MAX).
 * Host pointers. */
needle_hip_status needle_hip_ivf_probe(needle_hip_index* index, const float* queries, uint32_t q,
                                       uint32_t query_dim, uint32_t nprobe, uint32_t* out_list_ids,
                                       uint32_t* out_counts);

/* IvfIndex::cluster_stats sizes, ivf.rs:788-805: rows per list, [nlist]. */
needle_hip_status needle_hip_ivf_list_sizes(const needle_hip_index* index, uint32_t* out_sizes);

/* needle_hip_batch_search[_device] on an IVF index = IvfIndex::search per query,
 * ivf.rs:509-567, with params->nprobe: the k nearest rows of the probed lists
 * by (distance, id). */

/* ---- two-phase search ----------------------------------------------------- */

/* A Euclidean flat index over the first `dims` dimensions of every row of a flat
 * index (MatryoshkaTruncation::truncate, src/indexing/quantization.rs:801-804),
 * built on the device from the resident rows; it inherits the live-row mask the
 * full index has at this moment. */
needle_hip_status needle_hip_flat_build_prefix(needle_hip_index* full, uint32_t dims, needle_hip_index** out);

/* adaptive_search, src/indexing/quantization.rs:890-943, for q host queries of
 * the full dimension: phase 1 takes the min(k * max(overfetch, 1), n) rows
 * nearest over `coarse`'s prefix (Euclidean; by (distance, row)), phase 2 ranks
 * them by the Euclidean distance over all dimensions and returns the k best with
 * both distances.  Ties in the full distance go to the candidate with the LARGER
 * (coarse distance, row), as the reference's stable sort of its farthest-first
 * candidate list does (:922-939).  Phase 1 keeps the nearest candidates; the
 * reference's heap evicts its nearest entry once full (:913-918), untested there.
 * k * overfetch <= 2048. */
needle_hip_status needle_hip_adaptive_search(needle_hip_index* full, needle_hip_index* coarse, const float* queries,
                                             uint32_t q, uint32_t query_dim, uint32_t k, uint32_t overfetch,
                                             uint64_t* out_ids, float* out_dists, float* out_coarse_dists,
                                             uint32_t* out_counts);

/* ---- training ------------------------------------------------------------- */

/* IvfIndex::train's k-means, src/indexing/ivf.rs:330-450: k-means++ seeding,
 * Lloyd iterations (stop: no assignment changed, largest centroid shift <
 * tolerance, or max_iterations), empty clusters re-seeded from a random row.
 * The reference draws from thread_rng; here every draw comes from
 * splitmix64(seed), so a run is reproducible.  vectors: n x dim host f32;
 * out_centroids: room for min(n_clusters, n) x dim; *out_k receives the number
 * of centroids produced (ivf.rs caps n_clusters at n).  Distances, cluster sums
 * (row order) and the divide keep the reference's scalar arithmetic. */
needle_hip_status needle_hip_ivf_train(needle_hip_ctx* ctx, const float* vectors, uint64_t n, uint32_t dim,
                                       uint32_t n_clusters, uint32_t max_iterations, float tolerance,
                                       uint64_t seed, float* out_centroids, uint32_t* out_k);

/* ProductQuantizer::train, src/indexing/quantization.rs:330-364 with its kmeans
 * (:597-711): per subspace up to 256 centroids, seeding by squared distance to the
 * nearest chosen centroid, `iterations` (reference: 20) Lloyd passes, on the raw
 * sub-vectors; draws from splitmix64(seed) instead of thread_rng.  out_codebooks:
 * [m][256][dim/m] host f32, zero beyond the centroids a subspace produced;
 * out_ncent (nullable): [m] centroids produced (256 unless n < 256 or the rows
 * hold fewer distinct sub-vectors). */
needle_hip_status needle_hip_pq_train(needle_hip_ctx* ctx, const float* vectors, uint64_t n, uint32_t dim, uint32_t m,
                                      uint32_t iterations, uint64_t seed, float* out_codebooks, uint32_t* out_ncent);

/* ProductQuantizer::encode, quantization.rs:377-446, of n host vectors: per
 * subspace the first centroid at minimum squared distance (strict '<' from
 * f32::MAX).  ncent (nullable): centroids in use per subspace (256 when null).
 * out_codes: n x m host bytes. */
needle_hip_status needle_hip_pq_encode(needle_hip_ctx* ctx, const float* codebooks, uint32_t dim, uint32_t m,
                                       const uint32_t* ncent, const float* vectors, uint64_t n, uint8_t* out_codes);

/* IVF-PQ (IvfConfig::use_pq, src/indexing/ivf.rs:488-491, 536-546): as
 * needle_hip_ivf_build, but a list keeps ProductQuantizer::encode(vector) and a
 * search scores the probed lists with ProductQuantizer::asymmetric_distance
 * (src/indexing/quantization.rs:449-465).  codebooks: [m][256][dim/m] host f32.
 * The rows are encoded on the device and not kept. */
needle_hip_status needle_hip_ivf_pq_build(needle_hip_ctx* ctx, const float* base, uint64_t n, uint32_t dim,
                                          const float* centroids, uint32_t nlist, const uint32_t* assignment,
                                          const uint32_t* valid_bitmask, uint32_t m, const float* codebooks,
                                          needle_hip_index** out);

/* ---- quantized (ADC) indexes ------------------------------------------- */

/* A code matrix scored with ProductQuantizer::{compute_distance_table,
 * distance_with_table}, src/indexing/quantization.rs:468-503: per query the
 * [m][256] table of squared sub-distances, per row sqrt(sum_i table[i][code_i])
 * in subspace order.  codes: n x m host bytes (ProductQuantizer::encode output,
 * quantization.rs:377-398); codebooks: [m][256][dim/m] host f32
 * (quantization.rs:322-323).  needle_hip_batch_search on this index returns the
 * k rows nearest by that distance, ties to the lower row id (k above 256:
 * every row is scored in the same arithmetic, without the table scan). */
needle_hip_status needle_hip_pq_build(needle_hip_ctx* ctx, const uint8_t* codes, uint64_t n, uint32_t dim,
                                      uint32_t m, const float* codebooks, const uint32_t* valid_bitmask,
                                      needle_hip_index** out);

/* ---- binary codes -------------------------------------------------------- */

/* A 1-bit-per-dimension code matrix scanned by Hamming distance: BinaryQuantizer,
 * src/indexing/quantization.rs:536-594, searched as its call site does
 * (src/indexing/compression_advisor.rs:658-671: all distances, stable sort, first
 * k).  codes: n x ceil(dim / 8) host bytes (BinaryQuantizer::quantize output: bit
 * i % 8 of byte i / 8 set iff v[i] > thresholds[i], :574-585); thresholds: [dim]
 * host f32 (:538).  needle_hip_batch_search on this index quantizes every f32
 * query with the same rule and returns the k rows of smallest Hamming distance
 * (reported as f32, exact), ties to the lower row id (k <= 1024). */
needle_hip_status needle_hip_binary_build(needle_hip_ctx* ctx, const uint8_t* codes, uint64_t n, uint32_t dim,
                                          const float* thresholds, const uint32_t* valid_bitmask,
                                          needle_hip_index** out);

/* BinaryQuantizer::quantize (quantization.rs:574-585) of n host vectors on the
 * device: out_codes receives n x ceil(dim / 8) bytes. */
needle_hip_status needle_hip_binary_quantize(needle_hip_ctx* ctx, const float* thresholds, const float* vectors,
                                             uint64_t n, uint32_t dim, uint8_t* out_codes);

/* A u8 code matrix scored with ScalarQuantizer::asymmetric_distance,
 * src/indexing/quantization.rs:277-285: sqrt(sum_j (q_j - (c_j / scale_j + min_j))^2).
 * codes: n x dim host bytes (ScalarQuantizer::quantize output, :219-238);
 * min_vals, scale: [dim] host f32 of the trained quantizer (:137-141).
 * needle_hip_batch_search returns the k nearest rows by that distance, bit-equal
 * to the reference arithmetic, ties to the lower row id. */
needle_hip_status needle_hip_sq8_build(needle_hip_ctx* ctx, const uint8_t* codes, uint64_t n, uint32_t dim,
                                       const float* min_vals, const float* scale,
                                       const uint32_t* valid_bitmask, needle_hip_index** out);

/* ---- operator-level calls --------------------------------------------- */

/* GpuAccelerator::batch_{cosine,euclidean}_distance / batch_dot_product,
 * gpu/mod.rs:308-448 (what the reference's three CUDA kernels compute,
 * src/experimental/gpu/cuda.rs:212-453): one query against n rows.
 * metric in {COSINE, EUCLIDEAN, DOT_PRODUCT}; the dot product is returned
 * positive (not negated), as gpu/mod.rs:432.  Host pointers. */
needle_hip_status needle_hip_batch_distance(needle_hip_ctx* ctx, needle_metric metric,
                                            needle_arith arith, const float* query,
                                            const float* base, uint64_t n, uint32_t dim,
                                            float* out);

/* GpuAccelerator::kmeans_assign, gpu/mod.rs:660-703 (arith ACCEL_CHUNK4) and
 * IvfIndex::find_nearest_centroid, ivf.rs:453-459 (arith DISTANCE_RS): index of
 * the first nearest centroid by Euclidean distance.  Host pointers. */
needle_hip_status needle_hip_kmeans_assign(needle_hip_ctx* ctx, needle_arith arith,
                                           const float* vectors, uint64_t n,
                                           const float* centroids, uint32_t n_centroids,
                                           uint32_t dim, uint32_t* out_assignment);

/* merge_shard_results, src/persistence/shard.rs:622-643, and
 * MultiGpuShardManager::search's concat-sort-truncate, gpu/mod.rs:1355-1366:
 * for each of q queries merge `lists` runs of `per` (distance, id) pairs
 * (layout [lists][q][per], counts [lists][q]) into the k best by
 * (distance, id), dropping repeated ids.  Device pointers; used after the
 * all-gather of per-GPU results. */
needle_hip_status needle_hip_merge_topk_device(needle_hip_ctx* ctx, const float* d_dists,
                                               const uint64_t* d_ids, const uint32_t* d_counts,
                                               uint32_t lists, uint32_t q, uint32_t per,
                                               uint32_t k, uint64_t* d_out_ids,
                                               float* d_out_dists, uint32_t* d_out_counts);

/* The same merge for host arrays (uploaded, merged on the device, copied back): what
 * MultiGpuShardManager::search, gpu/mod.rs:1355-1366, does with the per-shard Vecs
 * it collected, and the chunk merge of streaming_batch_search, gpu/mod.rs:876-892. */
needle_hip_status needle_hip_merge_topk(needle_hip_ctx* ctx, const float* dists, const uint64_t* ids,
                                        const uint32_t* counts, uint32_t lists, uint32_t q, uint32_t per,
                                        uint32_t k, uint64_t* out_ids, float* out_dists, uint32_t* out_counts);

/* normalize, src/distance.rs:426-434 (arith DISTANCE_RS: |v|^2 summed in order, then
 * every component times 1 / sqrt) and GpuAccelerator::batch_normalize,
 * gpu/mod.rs:451-473 with normalize_vector_simd, gpu/common.rs:628-635 (arith
 * ACCEL_CHUNK4: groups of four, every component divided by the norm): n vectors of
 * `dim` f32 normalised in place; a zero vector is left as it is.  Host pointer /
 * device pointer. */
needle_hip_status needle_hip_normalize(needle_hip_ctx* ctx, needle_arith arith, float* vectors, uint64_t n,
                                       uint32_t dim);
needle_hip_status needle_hip_normalize_device(needle_hip_ctx* ctx, needle_arith arith, float* d_vectors,
                                              uint64_t n, uint32_t dim);

/* GpuAccelerator::batch_top_k, gpu/mod.rs:534-549 (stable sort by score, first k) and
 * partial_sort_top_k, gpu/common.rs:584-626 (heap of k with strict '<', so the
 * earliest index is kept among equal scores): for each of `rows` score vectors of n
 * f32 the min(k, n) smallest by (score, index), ascending; -0.0 == +0.0, NaN sorts
 * last.  Outputs are [rows][k]; unused tails hold index UINT64_MAX / score +inf.
 * k <= 2048, n < 2^32.  Host pointers / device pointers. */
needle_hip_status needle_hip_batch_top_k(needle_hip_ctx* ctx, const float* scores, uint32_t rows, uint64_t n,
                                         uint32_t k, uint64_t* out_idx, float* out_scores,
                                         uint32_t* out_counts);
needle_hip_status needle_hip_batch_top_k_device(needle_hip_ctx* ctx, const float* d_scores, uint32_t rows,
                                                uint64_t n, uint32_t k, uint64_t* d_out_idx,
                                                float* d_out_scores, uint32_t* d_out_counts);

/* GpuAccelerator::distance_matrix_{cosine,euclidean}, src/services/compute/
 * gpu_kernels.rs:223-265: every query against every vector, out[j * n + i] =
 * distance(query j, vector i) (DistanceMatrixResult::get, :113-115).  Any metric
 * of the chosen arithmetic (SERVICE for the calls above); the dot product is
 * reported negated, as the distance it is.  q <= 1 048 560. */
needle_hip_status needle_hip_distance_matrix(needle_hip_ctx* ctx, needle_metric metric, needle_arith arith,
                                             const float* queries, uint32_t q, const float* base, uint64_t n,
                                             uint32_t dim, float* out);
needle_hip_status needle_hip_distance_matrix_device(needle_hip_ctx* ctx, needle_metric metric, needle_arith arith,
                                                    const float* d_queries, uint32_t q, const float* d_base,
                                                    uint64_t n, uint32_t dim, float* d_out);

/* GpuAccelerator::multi_query_search, gpu_kernels.rs:267-300 (per query: all
 * distances, stable sort, truncate to k -- ties keep the lower index), also
 * DistanceMatrixResult::top_k (:118-129) and, with ACCEL_CHUNK4, the result of
 * streaming_multi_query_search, src/experimental/gpu/mod.rs:906-951.  The
 * distance matrix is formed on the device a block of queries at a time and each
 * block goes through the batch_top_k selection; nothing returns to the host in
 * between.  Outputs as needle_hip_batch_search ([q][k], tails padded). */
needle_hip_status needle_hip_multi_query_search(needle_hip_ctx* ctx, needle_metric metric, needle_arith arith,
                                                const float* queries, uint32_t q, const float* base, uint64_t n,
                                                uint32_t dim, uint32_t k, uint64_t* out_idx, float* out_dists,
                                                uint32_t* out_counts);
needle_hip_status needle_hip_multi_query_search_device(needle_hip_ctx* ctx, needle_metric metric, needle_arith arith,
                                                       const float* d_queries, uint32_t q, const float* d_base,
                                                       uint64_t n, uint32_t dim, uint32_t k, uint64_t* d_out_idx,
                                                       float* d_out_dists, uint32_t* d_out_counts);

/* ---- row-sharded search over the GPUs of one node --------------------------- */

/* MultiGpuShardManager, src/experimental/gpu/mod.rs:1269-1383, as one process per
 * GPU: every process builds an index over ITS contiguous block of rows (the
 * reference places rows round-robin, :1297-1306; contiguous blocks keep
 * global id = block start + local id, so ties by id survive the merge) and joins
 * one RCCL communicator.  Rank 0 obtains a unique id, the host distributes its
 * NEEDLE_HIP_UNIQUE_ID_BYTES bytes over its own channel, every rank calls
 * needle_hip_comm_init with the same bytes (collective: returns when all joined). */
#define NEEDLE_HIP_UNIQUE_ID_BYTES 128
typedef struct needle_hip_comm needle_hip_comm;
needle_hip_status needle_hip_comm_unique_id(uint8_t* out_id);
needle_hip_status needle_hip_comm_init(needle_hip_ctx* ctx, const uint8_t* unique_id, int rank, int world,
                                       needle_hip_comm** out);
void needle_hip_comm_destroy(needle_hip_comm* comm);
int needle_hip_comm_rank(const needle_hip_comm* comm);
int needle_hip_comm_world(const needle_hip_comm* comm);

/* MultiGpuShardManager::search, gpu/mod.rs:1355-1366 (search every shard, concat,
 * sort by distance, truncate) and merge_shard_results, src/persistence/shard.rs:
 * 622-643: needle_hip_batch_search_device on this rank's shard, local ids turned
 * global by adding row_offset, ONE all-gather of the ranks' (ids | distances |
 * counts) over RCCL / xGMI, and a merge by (distance, global id) on every rank.
 * The local search runs on one of the shard's search slots (its own stream) and
 * is complete when the exchange is enqueued on the context's stream; the call
 * returns with the merged result in place.  Collective: every rank calls it with
 * the same q, k and queries; every rank receives the full result.  A rank whose
 * local search fails still joins the all-gather with a marked slot, so the other
 * ranks do not wait for it: it returns its own error, the others
 * NEEDLE_HIP_ERR_RCCL naming the failed ranks.  world * k <= 2048.  Device
 * pointers; the queries must be complete before the call (the library does not
 * order itself behind the caller's streams). */
needle_hip_status needle_hip_sharded_batch_search_device(needle_hip_index* shard, needle_hip_comm* comm,
                                                         uint64_t row_offset, const float* d_queries,
                                                         uint32_t q, const needle_search_params* params,
                                                         uint64_t* d_out_ids, float* d_out_dists,
                                                         uint32_t* d_out_counts);

/* The merge step of the call above on `lists` slots that another transport
 * gathered (a host with its own channel between processes; single-GPU rehearsal):
 * slot l, needle_hip_shard_slot_bytes(q, k) bytes, holds shard l's ids [q][k] u64
 * (already global), distances [q][k] f32 and counts [q] u32, in that order. */
uint64_t needle_hip_shard_slot_bytes(uint32_t q, uint32_t k);
needle_hip_status needle_hip_merge_shard_slots_device(needle_hip_ctx* ctx, const void* d_slots, uint32_t lists,
                                                      uint32_t q, uint32_t k, uint64_t* d_out_ids,
                                                      float* d_out_dists, uint32_t* d_out_counts);

/* needle_hip_batch_top_k_device for callers that cannot allocate device memory
 * (tight serving loops, harnesses that forbid hipMalloc).  It has no counterpart
 * in the reference.  The call synchronises the library's stream before it
 * returns, so it cannot itself be recorded under HIP graph capture.  Same arguments and
 * results; the workspace (histograms, per-workgroup tie counts, the selected keys)
 * lives in a 1 MiB array that is part of the library's code object, one copy per
 * GPU for the life of the process, so the call issues no hipMalloc / hipFree.
 * A call whose workspace would exceed 1 MiB (about rows * (8 k + 4 * (260 +
 * min(ceil(n / 4096), 1024))) bytes) fails with NEEDLE_HIP_ERR_UNSUPPORTED.
 * Every call on one GPU uses that GPU's one array: calls that overlap in time on
 * the same GPU (from several contexts or processes' threads) race and must be
 * serialised by the caller; calls through one context are serialised by it. */
needle_hip_status needle_hip_batch_top_k_device_noalloc(needle_hip_ctx* ctx, const float* d_scores,
                                                        uint32_t rows, uint64_t n, uint32_t k,
                                                        uint64_t* d_out_idx, float* d_out_scores,
                                                        uint32_t* d_out_counts);

#ifdef __cplusplus
}
#endif
#endif /* NEEDLE_HIP_H */
