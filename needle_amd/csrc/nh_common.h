// Shared host/device helpers for libneedle_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/needle_hip.h"

namespace nh {

// ---- geometry of the first-pass (MFMA) scan ------------------------------
constexpr int TILE_ROWS = 128;     // base rows per tile (MFMA M)
constexpr int TILE_QUERIES = 128;  // queries per workgroup (MFMA N): 4 waves x 32
constexpr int KPRIME_MAX = 256;    // most candidates per query the first pass keeps
constexpr int LIST_E_MAX = 9;      // ceil((2*KPRIME_MAX + 32) / 64) entries per lane
constexpr int K_MAX = 2048;        // largest k any search accepts

// Candidate list capacity per (query, split): compaction brings a list back to
// kprime entries once it holds more than cap - 32; one MFMA tile adds at most
// 32 rows per query between two checks.
__host__ __device__ inline uint32_t list_cap(uint32_t kprime) { return 2 * kprime + 32; }

// ---- order-preserving float <-> uint maps --------------------------------
// f2ord is strictly increasing over all non-NaN floats (-0 < +0).
__host__ __device__ inline uint32_t f2ord(float f) {
#ifdef __HIP_DEVICE_COMPILE__
    uint32_t u = __float_as_uint(f);
#else
    union { float f; uint32_t u; } c; c.f = f; uint32_t u = c.u;
#endif
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ inline float ord2f(uint32_t o) {
    uint32_t u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
#ifdef __HIP_DEVICE_COMPILE__
    return __uint_as_float(u);
#else
    union { float f; uint32_t u; } c; c.u = u; return c.f;
#endif
}

// Distances are compared as OrderedFloat does (ordered-float 4.6: NaN greatest,
// -0 == +0): fold -0 into +0 and every NaN into the positive quiet NaN, whose
// f2ord image is above +inf.
__device__ inline float canon_distance(float d) {
    if (d != d) return __uint_as_float(0x7fc00000u);
    return d + 0.0f;
}

// Ascending key of a (distance, id) hit: smaller key = better hit, ties on the
// distance go to the lower id.
__device__ inline uint64_t hit_key(float canon_d, uint32_t id) {
    return ((uint64_t)f2ord(canon_d) << 32) | (uint64_t)id;
}

__device__ inline bool bit_set(const uint32_t* __restrict__ mask, uint32_t i) {
    return (mask[i >> 5] >> (i & 31)) & 1u;
}

// Rounding the operands to bf16 moves a first-pass score by at most
//     |dq| |b| + |q~| |db|  <=  qerr * B + (Q + qerr) * berr
// (dq, db = operand minus its bf16 copy, measured when the copies are made; Q, B = operand norms: 1 for the
// unit-normalised cosine operands, |q| and max |row| otherwise).  0 for f32 operands (qerr < 0).
__host__ __device__ inline double bf16_score_err(int metric, double qn, double bmax, double qerr, double berr) {
    if (qerr < 0.0) return 0.0;
    const double Q = metric == NEEDLE_METRIC_COSINE ? 1.000001 : qn, B = metric == NEEDLE_METRIC_COSINE ? 1.000001 : bmax;
    return 1.000001 * (qerr * B + (Q + qerr) * berr);
}
// Largest amount by which a first-pass score can differ from the exactly computed value of the same
// expression, in score units (u = 2^-24; a D-term product sum errs by at most D u sum|x y| in any
// order).  qn = |q|, bmax = max |row|, lowp = bf16_score_err(...).
__host__ __device__ inline double score_eps(int metric, double lowp, double qn, double bmax, uint32_t dim) {
    const double u = 5.9604644775390625e-08;
    const double gam = 2.0 * ((double)dim + 16.0) * u;
    switch (metric) {
    case NEEDLE_METRIC_COSINE: return lowp + 2.0 * gam;
    case NEEDLE_METRIC_EUCLIDEAN: return lowp + gam * qn * bmax + gam * 0.5 * bmax * bmax;
    default: return lowp + 2.0 * gam * qn * bmax;
    }
}
// the same for scalar-quantized rows (score = sum (q/scale) c - |v|^2/2): qconst = |q|^2 - 2 q.min, vmax = max |v|,
// mnorm = |min_vals|; with a bf16 query operand qs_err2 = |q/scale - bf16(q/scale)|^2 and cmax = max |code vector|
// (the codes themselves are exact in bf16), else qs_err2 < 0
__host__ __device__ inline double score_eps_sq8(double qconst, double vmax, double mnorm, uint32_t dim, double qs_err2,
                                                double cmax) {
    const double u = 5.9604644775390625e-08;
    const double gam = 2.0 * ((double)dim + 16.0) * u;
    const double qn = mnorm + sqrt(mnorm * mnorm + fabs(qconst));
    double e = gam * (qn * (vmax + mnorm) + 0.5 * vmax * vmax);
    if (qs_err2 >= 0.0) e += 1.000001 * sqrt(qs_err2) * cmax;
    return e;
}

// One unit of the inverted-list scan.
struct IvfItem {
    uint32_t tile0, tile1;   // row tiles [tile0, tile1) of the list (lists are padded to whole tiles)
    uint32_t qoff, nq;       // entries [qoff, qoff + nq) of item_queries, nq <= TILE_QUERIES
    uint32_t list_add;       // index of this run within its list: added to every entry's candidate list index
};

// ---- launch parameters of the first-pass scan ----------------------------
struct ScanParams {
    const void* base_op;      // [n_pad][dpad] operand copy (f32 or bf16), zero padded
    const void* query_op;     // [q_pad][dpad] operand copy of the queries
    const float* cinit;       // [n_pad] accumulator seed per row (-|b|^2/2 for L2) or null
    const uint32_t* valid;    // live-row bitmask or null
    const uint32_t* filter;   // pre-filter bitmask or null
    uint2* lists;             // [q_pad][n_splits][cap] (score bits, row id)
    uint32_t* list_counts;    // [q_pad][n_splits]
    uint32_t n;               // real rows
    uint32_t n_tiles;         // ceil(n / TILE_ROWS)
    uint32_t tiles_per_split; // slice s owns tiles s, s + n_splits, s + 2 n_splits, ... (at most tiles_per_split of them):
                              // rows that arrive clustered are spread over all slices
    uint32_t n_splits;
    uint32_t q_tiles;         // q_pad / queries per workgroup
    uint32_t n_queries;       // real queries; the padding slots up to q_pad never list a candidate
    uint32_t dpad;            // padded dimension, multiple of the k-step
    uint32_t kprime;
    uint32_t cap;
    uint32_t xcd_map;         // 1: n_splits % 8 == 0, splits pinned to blockIdx % 8
    uint32_t debug;           // nh::Tuning::debug: 1 = selection off (timing only)
    // threshold seeding: a pre-pass over the first pre_tiles tiles of every slice records, per
    // query, the maxima of 8 * n_splits disjoint row groups; the kprime-th largest of them (minus
    // two ulps) is a score at least kprime live rows exceed, so the main pass may start from it
    float* pre_max;           // [q_pad][8 * n_splits] (pre-pass output)
    const float* tau0;        // [q_pad] starting threshold of the main pass, or null
    uint32_t pre_tiles;
    uint32_t pre_stride;      // the pre-pass samples every pre_stride-th tile of a slice
    uint32_t kseed;           // rank of the seeded threshold among the sampled group maxima (k <= kseed <= kprime)
    // a launch may cover only tiles [tile_off, tile_off + tile_len) of every slice; `resume` continues the
    // candidate lists of the launch before it (counts are read back), with thresholds refreshed in between
    uint32_t tile_off, tile_len, resume;
    // small bases: the scan writes every score to dense_scores[query][dense_ld] (row-major per query) and a
    // selection kernel takes the candidates from there; no lists
    float* dense_scores;
    uint32_t dense_ld;
    // inverted-list scan: one workgroup per work item = (one list's row tiles) x (<= 128 of the
    // queries probing that list); rows are positions in the list-ordered copy of the base
    const IvfItem* items;     // [grid]
    const uint32_t* item_bounds;  // device: this launch covers items [item_bounds[0], item_bounds[1]); null: items[0 .. grid)
    const uint2* item_queries;  // (query index, candidate list of the entry's first run; + IvfItem::list_add) per item entry
    const uint32_t* pos2id;   // [n_pad] original row id of a position, ~0 for padding
    uint32_t base_is_u8;      // base_op is a u8 code matrix (scalar-quantized scan)
};

}  // namespace nh
