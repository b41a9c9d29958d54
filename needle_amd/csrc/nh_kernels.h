// Host-side launchers of the device kernels (internal to libneedle_hip).
#pragma once
#include "nh_common.h"

namespace nh {

// ---- flat_scan.hip ---------------------------------------------------------
hipError_t launch_flat_scan(int precision, const ScanParams& p, hipStream_t stream);
// What the threshold refresh needs to size the uncertainty band of a query's scores (see score_eps).
struct TauBand {
    const double* qn2;   // [q] |q|^2 (SQ8: qconst)
    const double* qs2;   // SQ8 with bf16 query operand: [q] |q/scale - bf16(q/scale)|^2, else null
    const float* qerr;   // bf16 operands: [q] |query operand - its bf16 copy|, else null
    double berr;         // bf16 operands: max over rows of |row operand - its bf16 copy|; SQ8: max |code vector|
    double bmax, mnorm;  // max row norm; SQ8: |min_vals|
    uint32_t dim;
    int metric, precision, sq8;
    float width = 2.0f;  // thresholds stay this many score uncertainties below the k-th score they were taken from
};
#ifdef __HIPCC__
// absolute uncertainty of a first-pass score of query q against any row (operand rounding + f32 accumulation)
__device__ inline double band_eps(const TauBand& band, uint32_t q) {
    if (band.sq8) return score_eps_sq8(band.qn2[q], band.bmax, band.mnorm, band.dim, band.qs2 ? band.qs2[q] : -1.0, band.berr);
    const double qn = sqrt(band.qn2[q]);
    const double lowp = bf16_score_err(band.metric, qn, band.bmax, band.qerr ? (double)band.qerr[q] : -1.0, band.berr);
    return score_eps(band.metric, lowp, qn, band.bmax, band.dim);
}
#endif
// Between candidate selection and exact re-scoring: candidates whose first-pass score lies more than twice the score
// uncertainty below the k-th best candidate score cannot be among the k hits; they are dropped (cand_ids / cand_n
// compacted) and tau_star is raised to the best dropped score, so the bound check still covers them.
// cand_scores must hold the candidates' first-pass scores.
struct PostParams;
hipError_t launch_prune_candidates(const PostParams& p, const TauBand& band, hipStream_t stream);
// pre-pass over p.pre_tiles tiles per slice (writes p.pre_max), then d_tau0[q_pad] = the kseed-th largest group
// maximum minus twice the score uncertainty
hipError_t launch_flat_prepass(int precision, const ScanParams& p, float* d_tau0, const TauBand& band, hipStream_t stream);
// between two launches sharing lists: d_tau[q] = max(d_tau[q], kseed-th largest score listed for q so far
// minus twice the score uncertainty)
// (d_list_base as in PostParams::list_base, with n_lists = the largest number of lists any query owns)
hipError_t launch_tau_refresh(const ScanParams& p, uint32_t n_lists, float* d_tau, const TauBand& band, hipStream_t stream,
                              const uint32_t* d_list_base = nullptr);
// inverted-list scan: one workgroup per p.items[i]
hipError_t launch_ivf_scan(int precision, const ScanParams& p, uint32_t n_items, hipStream_t stream);
int flat_scan_blocks_per_cu(int precision);
uint32_t flat_scan_kstep(int precision);
// queries one workgroup of the flat scan serves (q_pad is a multiple of it)
uint32_t flat_scan_queries_per_block(int precision, uint32_t dpad);

// ---- flat_scan_bf16.hip ----------------------------------------------------
// Padded operand dimension of the bf16 scan for `dim`, 0 when the dimension is beyond what the
// query-stationary kernel holds in registers (then the fp32 scan serves the search).
uint32_t flat_scan_bf16_dpad(uint32_t dim);
uint32_t flat_scan_bf16_queries_per_block(uint32_t dpad);
uint32_t flat_scan_bf16_ivf_queries_per_block(uint32_t dpad);  // query slots of an inverted-list work item
// Candidate lists are per (query, slice, lane group): flat_scan_bf16_lists_per_slice() * n_splits lists per query.
uint32_t flat_scan_bf16_lists_per_slice();
hipError_t launch_flat_scan_bf16(const ScanParams& p, bool prepass, hipStream_t stream);
// inverted-list scan: items hold up to flat_scan_bf16_queries_per_block(dpad) entries; entry .y names a GROUP of
// flat_scan_bf16_lists_per_slice() consecutive candidate lists (one per lane group)
hipError_t launch_ivf_scan_bf16(const ScanParams& p, uint32_t n_items, hipStream_t stream);

// ---- prep.hip --------------------------------------------------------------
// Per row: |b|, |b|^2 (wave-parallel sums; used by the first pass and its error
// bound only), the operand copies of the first pass (unit-normalised when
// `normalize`), the accumulator seed -|b|^2/2 when `want_cinit`, and in d_max_norm_bits (two words, as float
// bits, norms are non-negative) the running maximum norm [0] and the running minimum of the non-zero norms [1].
hipError_t launch_row_prep(const float* d_rows, uint64_t n, uint32_t dim, uint32_t dpad,
                           bool normalize, float* d_op_f32, uint16_t* d_op_bf16, float* d_cinit,
                           float* d_norms, double* d_norm2, uint32_t* d_max_norm_bits,
                           float* d_bf16_err /* [n] |operand - bf16 operand|, nullable */,
                           uint32_t* d_max_bf16_err_bits /* running maximum of it, nullable */, hipStream_t stream);
// Cosine rows with a norm f32 cannot square (prep.hip): list and count them, seeds[i] = -inf for them and 0 otherwise.
hipError_t launch_unsafe_rows(const float* d_norms, uint64_t n, float* d_seeds, uint32_t* d_list, uint32_t cap, uint32_t* d_count,
                              hipStream_t stream);
hipError_t launch_unsafe_live(const uint32_t* d_list, uint32_t n_list, const uint32_t* d_valid, const uint32_t* d_filter, uint32_t* d_out,
                              uint32_t* d_out_n, hipStream_t stream);
// 0 if every value is finite, else 1 + index of the first row holding a NaN / Inf.
hipError_t launch_validate_rows(const float* d_rows, uint64_t n, uint32_t dim, uint32_t* d_bad_row_plus1,
                                hipStream_t stream);

// Scalar-quantized index build: per row -|v|^2/2 (v = decoded row); d_max_norm_bits[0] = running max |v|,
// d_max_norm_bits[1] = running max |code vector|.
hipError_t launch_sq8_row_prep(const uint8_t* d_codes, uint64_t n, uint32_t dim, uint32_t stride, const float* d_min,
                               const float* d_scale, float* d_cinit, uint32_t* d_max_norm_bits, hipStream_t stream);
// Scalar-quantized query upload: q_op = q / scale (zero padded to dpad; f32 and/or bf16 copy),
// qconst = sum(q^2 - 2 q min), qscaled_err2 = |q/scale - bf16(q/scale)|^2 (nullable, with the bf16 copy).
hipError_t launch_sq8_query_prep(const float* d_queries, uint32_t q, uint32_t dim, uint32_t dpad, const float* d_min,
                                 const float* d_scale, float* d_qop, uint16_t* d_qop_bf16, double* d_qconst,
                                 double* d_qscaled_err2, hipStream_t stream);
// bf16 operand copy [n][dpad] of a u8 code matrix [n][stride] (exact); d_out is zero-filled by the caller.
// ProductQuantizer::encode of device-resident rows (train.hip): out [n][m]
hipError_t launch_pq_encode(const float* d_v, uint32_t n, uint32_t dim, uint32_t m, const float* d_books, const uint32_t* d_ncent,
                            uint8_t* d_out, hipStream_t stream);
// decoded PQ rows as the bf16 operand of the matrix-core scan, with -|row|^2/2, max |row| and max rounding error (prep.hip)
hipError_t launch_pq_decode_bf16(const uint8_t* d_codes, const float* d_books, uint64_t n, uint32_t m, uint32_t dsub, uint32_t dpad,
                                 uint16_t* d_op_bf16, float* d_cinit, uint32_t* d_max_bits, hipStream_t stream);
hipError_t launch_sq8_codes_bf16(const uint8_t* d_codes, uint64_t n, uint32_t dim, uint32_t stride, uint32_t dpad,
                                 uint16_t* d_out, hipStream_t stream);

// ---- flat_post.hip ---------------------------------------------------------
struct PostParams {
    const float* base_raw;    // [n][dim] original rows
    const float* query_raw;   // [q][dim] original queries
    const double* qnorm2;     // [q] |q|^2
    const double* qaux;       // SQ8 with bf16 operands: [q] |q/scale - bf16(q/scale)|^2, else null
    const float* qerr;        // bf16 operands: [q] |query operand - its bf16 copy|, else null
    double berr;              // bf16 operands: max |row operand - its bf16 copy|; SQ8: max |code vector|
    const uint2* lists;
    const uint32_t* list_counts;
    const uint32_t* list_base; // [q + 1] or null: query q owns lists list_base[q] .. list_base[q+1]-1 (else q*n_splits ..)
    uint32_t* cand_ids;       // [q][kprime]
    uint32_t* cand_n;         // [q]
    float* cand_scores;       // [q][kprime] first-pass scores of the kept candidates, or null
    float* tau_star;          // [q]
    const float* tau0;        // [q] threshold the main pass started from, or null
    const uint32_t* pos2id;   // candidates are positions of a list-ordered copy: their row ids, or null
    const float* dense_scores;  // dense path: [q][dense_ld] first-pass scores, with the masks the scan would apply
    uint32_t dense_ld;
    const uint32_t* valid;
    const uint32_t* filter;
    // scalar-quantized rows: distances are ScalarQuantizer::asymmetric_distance on the codes
    const uint8_t* sq_codes;  // [n_pad][sq_stride] or null
    const uint8_t* pq_codes;  // PQ rows scored through their decoded bf16 copy: [n][pq_m] codes, else null
    const float* pq_books;    // [pq_m][256][pq_dsub]
    uint32_t pq_m, pq_dsub;
    int pq_asym;              // 1: asymmetric_distance's single accumulator (IVF-PQ); 0: the distance-table sums
    const float* sq_min;      // [dim]
    const float* sq_scale;    // [dim]
    uint32_t sq_stride;
    float aux_norm;           // SQ8: |min_vals| (bound check)
    const uint32_t* extra_ids;  // rows every query re-scores beside its own candidates (cosine rows whose norm the first pass
    const uint32_t* extra_n;    // cannot bound), and how many: device pointers, null = none
    uint32_t tune_merge_keys, tune_merge_lists, tune_rs_cap, tune_rs_dc;  // planning overrides (nh::Tuning), 0 = built-in
    uint32_t* flagged;        // [q] query indices failing the bound check
    uint32_t* flag_count;     // [1]
    uint64_t* out_ids;        // [q][k]
    float* out_dists;         // [q][k]
    uint32_t* out_counts;     // [q]
    uint32_t q, n, dim, k, kprime, cap, n_splits;
    int metric, arith, precision;
    float max_norm;
};
hipError_t launch_merge_select(const PostParams& p, hipStream_t stream);
// Small bases: candidates straight from the dense score matrix p.dense_scores[q][dense_ld] (the kprime best live
// rows by (score, lower id); tau* = the kprime-th score, -inf when every live row is a candidate).  n <= DENSE_MAX_ROWS.
constexpr uint32_t DENSE_MAX_ROWS = 8192;
hipError_t launch_dense_select(const PostParams& p, hipStream_t stream);
hipError_t launch_dense_scores_bf16(const ScanParams& p, hipStream_t stream);
// set_only_band (Euclidean flat rows, cand_scores present, no pruning before): only the SET of the k nearest rows is
// wanted -- candidates the first-pass scores already decide are not re-scored; ids unordered, distances unset
hipError_t launch_rescore_finalize(const PostParams& p, hipStream_t stream, const TauBand* set_only_band = nullptr);
// For scans whose scores ARE the reference distances (score = -distance): rank the merged
// candidates by (distance, id) and emit the k best.  Needs cand_scores.
hipError_t launch_emit_sorted(const PostParams& p, hipStream_t stream);

// ---- adc.hip ---------------------------------------------------------------
// ProductQuantizer table scan, src/indexing/quantization.rs:468-503: per query the distance table
// [m][256] is built in LDS, every row's distance is the table sum in subspace order, then sqrt.
struct PqScanParams {
    const uint8_t* codes_gm;   // [ceil(m / 16)][n][16]: the codes of 16 subspaces of a row in one 16-byte read, rotated by row & 15
    const float* tables;       // [ceil(q / 8)][ceil(m / 16)][256][2][16][4]: distance tables of 8 queries, one LDS image per slice
    const uint32_t* valid;
    const uint32_t* filter;
    uint2* lists;              // [q][n_splits * PQ_WAVES][cap]  (bits of -distance, row)
    uint32_t* list_counts;     // [q][n_splits * PQ_WAVES]
    const float* tau_seed;     // [q] a score at least k live rows of the query reach (a sample's k-th best), or null
    uint32_t n, m, n_sg, q, n_splits, rows_per_split, k, cap;  // split s scans rows_per_split rows from s * split_stride on
    uint32_t split_stride;     // a multiple of 16
    uint32_t xcd_slots;        // workgroups an XCD runs at a time (its CUs); a multiple of n_splits
    uint32_t dev;              // timing experiments only (results are wrong): 1 no selection, 256 tables filled once, 512 no lookups
};
#ifndef PQ_WAVES_N
#define PQ_WAVES_N 12
#endif
constexpr int PQ_WAVES = PQ_WAVES_N;   // waves per workgroup sharing the resident tables; one candidate list per wave and query
constexpr int PQ_QB = 8;       // queries per workgroup
constexpr int PQ_SG = 16;      // subspaces per resident table slice
hipError_t launch_pq_tables(const float* d_queries, const float* d_codebooks, uint32_t q, uint32_t m, uint32_t dsub, uint32_t dim,
                            float* d_tables, hipStream_t stream);
hipError_t launch_pq_group_codes(const uint8_t* codes, uint64_t n, uint32_t m, uint8_t* out, hipStream_t stream);
hipError_t launch_pq_scan(const PqScanParams& p, hipStream_t stream);
uint32_t pq_scan_block_rows();
size_t pq_table_bytes(uint32_t q, uint32_t m);  // of the tables launch_pq_tables writes for q queries

// Exact scan (every row scored in the reference's scalar order): the path for
// queries whose first-pass candidates fail the bound check, for the metrics with
// no contraction form, and for NEEDLE_PRECISION_EXACT_SCAN.
struct ExactParams {
    const float* base_raw;
    const float* query_raw;
    const uint32_t* qsel;     // [f] query index of each slot, or null (slot == query)
    const uint32_t* valid;
    const uint32_t* filter;
    const uint32_t* pos2id;   // rows are positions of a list-ordered copy: their row ids (~0 padding), or null
    const uint32_t* pos_filter;  // [f][pos_filter_words] per-slot bitmask over positions, or null
    const uint8_t* pq_codes;  // product-quantized rows [n][pq_m], or null
    const float* pq_books;
    uint32_t pq_m, pq_dsub;
    int pq_asym;              // 1: asymmetric_distance's single accumulator (IVF-PQ); 0: the distance-table sums (PQ index)
    const uint8_t* sq_codes;  // scalar-quantized rows (see PostParams) or null
    const float* sq_min;
    const float* sq_scale;
    uint32_t sq_stride;
    uint32_t pos_filter_words;
    uint64_t* keys;           // [f][n] workspace
    uint32_t* live;           // [f]
    uint32_t* k_eff;          // [f]
    uint32_t* kth_rem;        // [f]
    uint64_t* prefix;         // [f]
    uint32_t* hist;           // [f][256]
    uint64_t* sel;            // [f][k]
    uint32_t* sel_n;          // [f]
    uint64_t* out_ids;
    float* out_dists;
    uint32_t* out_counts;
    uint32_t f, n, dim, k;
    int metric, arith;
};
hipError_t launch_exact_scan(const ExactParams& p, hipStream_t stream);

// For each slot f (query qsel[f]) set the position bits of the rows of its probed lists.
hipError_t launch_ivf_mark_lists(const uint32_t* d_qsel, uint32_t f, const uint32_t* d_probe, const uint32_t* d_probe_n,
                                 uint32_t nprobe, const uint32_t* d_list_pos, const uint32_t* d_list_len,
                                 uint32_t* d_mask, uint32_t words, hipStream_t stream);

// One query against n rows, raw distances (dot product positive).
// every query against every row, [q][n] distances in the chosen arithmetic (flat_post.hip)
hipError_t launch_distance_matrix(const float* d_queries, uint32_t q, const float* d_base, uint64_t n, uint32_t dim, int metric,
                                  int arith, float* d_out, hipStream_t stream);
hipError_t launch_batch_distance(const float* d_query, const float* d_base, uint64_t n, uint32_t dim,
                                 int metric, int arith, float* d_out, hipStream_t stream);

hipError_t launch_merge_topk(const float* d_dists, const uint64_t* d_ids, const uint32_t* d_counts,
                             uint32_t lists, uint32_t q, uint32_t per, uint32_t k, uint64_t* d_out_ids,
                             float* d_out_dists, uint32_t* d_out_counts, hipStream_t stream);

// dst[i] = src[idx[i]] for rows of `dim` floats; and the way back for finished result rows.
hipError_t launch_gather_rows(const float* d_src, const uint32_t* d_idx, uint32_t n, uint32_t dim, float* d_dst,
                              hipStream_t stream);
hipError_t launch_scatter_results(const uint32_t* d_idx, uint32_t n, uint32_t k, const uint64_t* d_ids, const float* d_dists,
                                  const uint32_t* d_counts, uint64_t* d_out_ids, float* d_out_dists, uint32_t* d_out_counts,
                                  hipStream_t stream);

// 16-bit rows (F16 or Bf16 of src/indexing/float16.rs) widened exactly to f32 (prep.hip)
hipError_t launch_half_widen(const uint16_t* d_in, uint64_t total, bool bf16, float* d_out, hipStream_t stream);
// per-search accumulator seeds with dead rows (deleted, filtered out, beyond n) at -inf (prep.hip)
// Compaction of the rows a filtered search admits (prep.hip): per-block counts + total, ascending positions, gathered
// rows (rows at or beyond `live` zero), hits renamed to base row numbers.
uint32_t mask_blocks(uint32_t n);  // entries of d_block_counts for n rows
hipError_t launch_mask_count(const uint32_t* d_valid, const uint32_t* d_filter, uint32_t n, uint32_t* d_block_counts, uint32_t* d_total,
                             hipStream_t stream);
hipError_t launch_mask_positions(const uint32_t* d_valid, const uint32_t* d_filter, uint32_t n, uint32_t* d_block_counts, uint32_t live,
                                 uint32_t* d_pos, hipStream_t stream);
hipError_t launch_gather_row_bytes(const void* d_src, const uint32_t* d_pos, uint32_t live, uint32_t rows, uint32_t n_src,
                                   size_t row_bytes, void* d_dst, hipStream_t stream);
hipError_t launch_map_ids(uint64_t* d_ids, uint64_t total, const uint32_t* d_pos, uint32_t live, hipStream_t stream);
hipError_t launch_masked_seed(const float* d_seed, const uint32_t* d_valid, const uint32_t* d_filter, uint32_t n, uint32_t n_pad,
                              float* d_out, hipStream_t stream);
hipError_t launch_fill_empty(uint64_t* d_out_ids, float* d_out_dists, uint32_t* d_out_counts, uint32_t q,
                             uint32_t k, hipStream_t stream);

}  // namespace nh
