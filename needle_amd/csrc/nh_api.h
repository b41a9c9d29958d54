// Internal host-side definitions shared by the C ABI translation units.
#pragma once
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <strings.h>
#include <vector>

#include "nh_common.h"
#include "nh_kernels.h"

namespace nh {

needle_hip_status fail(needle_hip_status st, const char* fmt, ...);

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess)                                                                  \
            return nh::fail(_e == hipErrorOutOfMemory ? NEEDLE_HIP_ERR_OUT_OF_MEMORY : NEEDLE_HIP_ERR_HIP, \
                            "HIP error %d (%s) at %s:%d: %s", (int)_e, hipGetErrorString(_e), __FILE__,    \
                            __LINE__, #expr);                                                  \
    } while (0)

#define ST_TRY(expr)                              \
    do {                                          \
        needle_hip_status _s = (expr);            \
        if (_s != NEEDLE_HIP_OK) return _s;       \
    } while (0)

// Process-wide development switches, set through needle_hip_set_tuning ("trace", "poison"); the library reads no
// environment variable.
inline std::atomic<bool> g_trace{false};   // log workspace allocations and slot hand-outs to stderr
inline std::atomic<bool> g_poison{false};  // fresh workspace starts as all-ones: a kernel reading what nothing wrote shows up at once

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) {
            if (g_trace) fprintf(stderr, "[needle_hip] buf %p: grows, frees %p\n", (void*)this, p);
            (void)hipFree(p); p = nullptr; cap = 0;
        }
        size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) { p = nullptr; return e; }
        cap = want;
        if (g_poison) { (void)hipMemset(p, 0xff, want); (void)hipDeviceSynchronize(); }
        if (g_trace) fprintf(stderr, "[needle_hip] buf %p: %zu bytes at %p\n", (void*)this, want, p);
        return hipSuccess;
    }
    void release() {
        if (g_trace && p) fprintf(stderr, "[needle_hip] buf %p: released %p\n", (void*)this, p);
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
    }
    template <typename T> T* as() const { return (T*)p; }
};

// Planning overrides of a context (needle_hip_set_tuning): what tests use to force a code path and what the A/B tools
// sweep.  They move plans and cost, never results.  0 / negative = the built-in choice.
struct Tuning {
    uint32_t debug = 0;          // 1 selection off (timing only: wrong results), 2 no threshold seeding, 4 one launch per slice,
                                 // 8 print what the first pass listed, 16 no candidate pruning, 64 print the in-kernel clock
                                 // (stamp build); PQ table scan, timing only: 256 tables filled once, 512 no lookups
    uint32_t splits = 0;         // slices of the base per query block
    int32_t pre_tiles = -1;      // tiles per slice the threshold pre-pass samples
    uint32_t kseed = 0;          // rank of the seeded threshold among the sampled group maxima
    float band = 0.0f;           // width of the threshold band in score uncertainties
    uint32_t no_compact = 0;     // 1: a filtered flat search never scans a gathered copy of the admitted rows
    uint32_t compact_pct = 0;    // largest admitted share (percent of the rows) that still takes the gathered copy (0: 50)
    uint32_t compact_min_rows = 0;  // smallest base that does (0: 65536 rows)
    uint32_t stage_pct[4] = {0, 0, 0, 0};  // share of a slice per staged launch (the last launch takes the rest)
    uint32_t n_stage_pct = 0;
    uint32_t no_dense = 0;       // small bases take the candidate lists instead of the dense score matrix
    uint32_t no_set_only = 0;    // the IVF probe re-scores every centroid candidate
    uint32_t merge_keys = 0, merge_lists = 0;  // LDS keys / one-shot lists of the candidate merge
    uint32_t rs_cap = 0, rs_dc = 0;            // exact re-score: rows staged at a time / dimensions per block
    uint32_t ivf_no_seed = 0;    // inverted-list scan: thresholds start at -inf
    uint32_t ivf_seed_rows = 0;  // rows of the nearest list sampled per query for the starting threshold
    uint32_t ivf_split = 0;      // probe slots scanned by the first list-scan launch (the threshold refresh follows it)
    uint32_t binary_two_scans = 0;  // binary codes: always the ranked two-scan path
};

inline uint32_t round_up(uint32_t v, uint32_t m) { return (v + m - 1) / m * m; }

}  // namespace nh

struct needle_hip_ctx {
    int device = 0;
    int n_cu = 256;
    hipStream_t stream = nullptr;
    bool profiling = false;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    // a second stream (created on first use) for work of one search that does not depend on what `stream` is doing:
    // forked and joined with events, never synchronised on its own
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    needle_hip_metrics m{};
    nh::Tuning tune;
    std::recursive_mutex mu;  // recursive: a host-pointer operator holds it across staging, its _device form and the copy back
    std::mutex prof_mu;  // held by a whole search while profiling is on (the events ev[] are shared by the context's indexes)
    nh::DevBuf scratch[8];  // operator-level calls
    void* topk_static = nullptr;   // device address of the static top-k workspace on this context's device (ops.hip)
    size_t topk_static_bytes = 0;
};

namespace nh { struct SearchSlot; }

enum IndexKind { KIND_FLAT = 0, KIND_IVF = 1, KIND_PQ = 2, KIND_SQ8 = 3, KIND_BINARY = 4 };

struct needle_hip_index {
    needle_hip_ctx* ctx = nullptr;
    int kind = KIND_FLAT;
    uint64_t n = 0;                // rows
    uint32_t dim = 0;
    uint32_t n_pad = 0;            // rows of the device copies (flat: n rounded up to 128; IVF: padded positions)
    int metric = 0, arith = 0;
    float* d_raw = nullptr;        // [n_pad][dim], padding rows zero (IVF: list-ordered)
    float* d_op_f32 = nullptr;     // [n_pad][dpad_f32] (may alias d_raw)
    bool op_f32_alias = false;
    uint16_t* d_op_bf16 = nullptr; // [n_pad][dpad_bf16], built on first use
    uint32_t dpad_f32 = 0, dpad_bf16 = 0;
    float* d_cinit = nullptr;      // [n_pad] (Euclidean only)
    float* d_norms = nullptr;      // [n_pad]
    uint32_t* d_valid = nullptr;   // live-row bitmask (by row id) or null
    float max_norm = 0.0f;
    float min_norm = 3.40282347e+38f;  // smallest non-zero row norm (cosine: decides cos_norm_unsafe; kept for appends)
    uint32_t cap_pad = 0;          // flat index: rows the resident arrays are allocated for (>= n_pad; grows by doubling on append)
    bool cos_norm_unsafe = false;  // cosine index holding a row whose norm the reference's f32 arithmetic cannot represent
    uint32_t* d_unsafe = nullptr;  // those rows (up to UNSAFE_ROWS_MAX of them): re-scored for every query beside its candidates,
    uint32_t n_unsafe = 0;         // and kept out of every first pass by -inf accumulator seeds; a search that cannot afford
                                   // n_unsafe extra candidates (or an index with more of them) takes the exact scan
    uint32_t bf16_extra_slack = 0; // candidates kept per query beyond the default bf16 slack (grows when batches leave queries undecided)
    float bf16_err = 0.0f;         // max over rows of |row operand - its bf16 copy| (set with d_op_bf16); SQ8: max |code vector|
    std::mutex mu;
    // IVF
    needle_hip_index* centroids = nullptr;  // flat Euclidean index over the centroids (probing)
    uint32_t nlist = 0;
    std::vector<uint32_t> list_pos, list_len;  // start position (multiple of 128) and rows of every list
    std::vector<uint32_t> list_tiles_desc;     // row tiles of every list, largest first (bounds a query's candidate lists)
    uint32_t* d_list_pos = nullptr;
    uint32_t* d_list_len = nullptr;
    uint32_t* d_pos2id = nullptr;  // [n_pad] row id of a position, ~0 for padding
    // quantized code matrices
    uint8_t* d_codes = nullptr;    // PQ: [n][pq_m]; SQ8: [n_pad][dpad_f32] zero padded
    uint8_t* d_codes_gm = nullptr; // PQ index: the codes again as [ceil(pq_m / 16)][n][16] for the distance-table scan
    float* d_codebooks = nullptr;  // PQ: [pq_m][256][pq_dsub]
    uint32_t pq_m = 0, pq_dsub = 0;
    uint32_t bin_words = 0;        // binary codes: 64-bit words per row; d_codes holds [bin_words][n_pad] words, d_sq_min the thresholds
    float* d_sq_min = nullptr;     // SQ8: [dim] min_vals
    float* d_sq_scale = nullptr;   // SQ8: [dim] scale
    float* d_sq_inv = nullptr;     // SQ8: [dpad] 1/scale, zero padded (first-pass query scaling)
    double sq_min_norm = 0.0;      // SQ8: |min_vals|
    nh::DevBuf w_scores;
    // per-search workspace (grow-only)
    nh::DevBuf w_qraw, w_qop, w_qn2, w_lists, w_lcnt, w_cand, w_candn, w_tau, w_flagged, w_flagcnt, w_filter, w_premax,
        w_tau0;
    nh::DevBuf w_extra;  // [UNSAFE_ROWS_MAX + 1]: the unsafe rows this search may return, then their count
    nh::DevBuf w_cscore; // [q][kprime] first-pass scores of the selected candidates (pruning before the exact re-score)
    nh::DevBuf w_cinit; // [n_pad] accumulator seeds of a search under row masks (dead rows at -inf)
    nh::DevBuf w_qerr;  // [q_pad] |query operand - its bf16 copy|
    nh::DevBuf r_idx, r_q, r_ids, r_dists, r_cnt;  // queries a reduced-precision pass could not decide, re-run in fp32
    nh::DevBuf x_keys, x_u32, x_u64, x_sel, x_mask;
    nh::DevBuf v_probe_ids, v_probe_d, v_probe_n, v_probe32, v_items, v_itemq, v_lbase, v_regroup, v_probe_sel;
    nh::DevBuf w_pqtab; // PQ table scan: interleaved distance tables of a chunk of queries
    nh::DevBuf b_work;  // binary scan: query codes, histograms, tie counts, collected keys
    // Re-entrant searches: every batch search runs on a search slot -- a view of this index (same resident data, its
    // own workspace buffers) bound to a context of its own (own stream, events, metrics).  Slots are created on demand,
    // at most nh::MAX_SEARCH_SLOTS per index; a search that finds them all busy waits for one.
    std::mutex pool_mu;
    std::condition_variable pool_cv;
    std::vector<nh::SearchSlot*> slots;
    bool is_view = false;
    nh::SearchSlot* slot = nullptr;  // a view: the search slot it belongs to
};

namespace nh {
constexpr size_t MAX_SEARCH_SLOTS = 4;
constexpr uint32_t UNSAFE_ROWS_MAX = 192;
// cosine flat index: (re)derives the list of rows with unrepresentable norms and their seeds from d_norms (api.hip)
needle_hip_status refresh_unsafe_rows(needle_hip_index* ix);
struct SearchSlot {
    needle_hip_ctx ctx;       // device and CU count of the parent context; stream, events and metrics of its own
    needle_hip_index ix;      // view of the parent index
    needle_hip_index cent;    // view of the parent's centroid index (IVF)
    // A filtered search of a flat index whose masks admit a small part of the rows scans a gathered copy of the admitted
    // rows: `compact` is a flat index over that copy (its resident arrays live in the z_ buffers, grow-only like any
    // workspace; z_pos maps a row of the copy back to its row of the base).
    needle_hip_index compact;
    nh::DevBuf z_raw, z_op16, z_op32, z_cinit, z_pos, z_cnt;
    bool busy = false;
};
// Takes a free slot of `parent` (creating one, or waiting), refreshes its view of the parent's resident data (building
// the parent's bf16 operand copy first when the search will use it) and gives it back on destruction, folding the
// slot's metrics into the parent context.
class SlotLease {
public:
    SlotLease(needle_hip_index* parent, const needle_search_params* params);
    ~SlotLease();
    SlotLease(const SlotLease&) = delete;
    SlotLease& operator=(const SlotLease&) = delete;
    needle_hip_status status() const { return st_; }
    needle_hip_index* view() const { return slot_ ? &slot_->ix : nullptr; }
private:
    needle_hip_index* parent_;
    SearchSlot* slot_ = nullptr;
    needle_hip_status st_ = NEEDLE_HIP_OK;
};
void free_search_slots(needle_hip_index* ix);
// flat search of one batch (api.hip); used by the IVF probe step as well.  defer_flagged: return with the work enqueued and
// without the host learning which queries the bound check left open: ix->w_flagcnt / w_flagged (device) name them, and
// the caller owes them an answer
needle_hip_status flat_search_batch_device(needle_hip_index* ix, const float* d_queries, uint32_t q, uint32_t k,
                                           int precision, const uint32_t* d_filter, uint64_t* d_out_ids,
                                           float* d_out_dists, uint32_t* d_out_counts, bool set_only = false,
                                           bool defer_flagged = false);
// exact scan of f_total slots; IVF: d_probe/d_probe_n name the lists each query may draw rows from
needle_hip_status run_exact(needle_hip_index* ix, const float* d_queries, const uint32_t* d_qsel, uint32_t f_total,
                            uint32_t k, const uint32_t* d_filter, uint64_t* d_out_ids, float* d_out_dists,
                            uint32_t* d_out_counts, const uint32_t* d_probe = nullptr,
                            const uint32_t* d_probe_n = nullptr, uint32_t nprobe = 0);
uint32_t choose_kprime(uint32_t k, int precision);
// builds ix->d_op_bf16 (and ix->bf16_err) on first use; the caller holds ix->mu
needle_hip_status ensure_bf16_copy(needle_hip_index* ix);
// takes the device address of the static top-k workspace (ops.hip); the context's device must be current
needle_hip_status topk_static_setup(needle_hip_ctx* c);
needle_hip_index* index_new();  // registered as live; null when out of host memory
hipError_t ensure_side_stream(needle_hip_ctx* c);
void destroy_side_stream(needle_hip_ctx* c);
void index_free(needle_hip_index* ix);
needle_hip_status finish_operands(needle_hip_index* ix, bool validate);
needle_hip_status upload_mask(needle_hip_index* ix, const uint32_t* mask_host);
// quantized scans of one batch (api_adc.hip)
// Hamming scan of a binary code matrix (binary.hip)
needle_hip_status binary_search_batch_device(needle_hip_index* ix, const float* d_queries, uint32_t q, uint32_t k,
                                             const uint32_t* d_filter, uint64_t* d_out_ids, float* d_out_dists,
                                             uint32_t* d_out_counts);
needle_hip_status pq_search_batch_device(needle_hip_index* ix, const float* d_queries, uint32_t q, uint32_t k,
                                         const uint32_t* d_filter, uint64_t* d_out_ids, float* d_out_dists,
                                         uint32_t* d_out_counts);
// IVF search of one batch (api_ivf.hip)
needle_hip_status ivf_search_batch_device(needle_hip_index* ix, const float* d_queries, uint32_t q, uint32_t k,
                                          uint32_t nprobe, int precision, const uint32_t* d_filter,
                                          uint64_t* d_out_ids, float* d_out_dists, uint32_t* d_out_counts);
}  // namespace nh
