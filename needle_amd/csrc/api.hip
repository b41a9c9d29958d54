// C ABI of libneedle_hip.so (declarations and reference citations:
// include/needle_hip.h).  Host-side orchestration only; kernels live in
// flat_scan.hip, flat_post.hip and prep.hip.
#include "nh_api.h"

#include <unordered_set>

using namespace nh;

// ---------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------
static thread_local std::string g_last_error;

needle_hip_status nh::fail(needle_hip_status st, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return st;
}

// ---------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------
extern "C" int needle_hip_version(void) { return 100; }

extern "C" const char* needle_hip_last_error(void) { return g_last_error.c_str(); }

extern "C" int needle_hip_metric_from_str(const char* s) {
    if (!s) return -1;
    static const struct { const char* name; int m; } tab[] = {
        {"cosine", NEEDLE_METRIC_COSINE},
        {"cosine_normalized", NEEDLE_METRIC_COSINE_NORMALIZED},
        {"cosinenormalized", NEEDLE_METRIC_COSINE_NORMALIZED},
        {"euclidean", NEEDLE_METRIC_EUCLIDEAN}, {"l2", NEEDLE_METRIC_EUCLIDEAN},
        {"dot", NEEDLE_METRIC_DOT_PRODUCT}, {"dotproduct", NEEDLE_METRIC_DOT_PRODUCT},
        {"dot_product", NEEDLE_METRIC_DOT_PRODUCT}, {"inner_product", NEEDLE_METRIC_DOT_PRODUCT},
        {"manhattan", NEEDLE_METRIC_MANHATTAN}, {"l1", NEEDLE_METRIC_MANHATTAN},
        {"hamming", NEEDLE_METRIC_HAMMING},
        {"chebyshev", NEEDLE_METRIC_CHEBYSHEV}, {"linf", NEEDLE_METRIC_CHEBYSHEV},
        {"l_inf", NEEDLE_METRIC_CHEBYSHEV}, {"l_infinity", NEEDLE_METRIC_CHEBYSHEV},
    };
    for (const auto& t : tab)
        if (strcasecmp(s, t.name) == 0) return t.m;
    return -1;
}

// Live handles.  Destroying an index twice, destroying it after its context was shut down, or shutting
// a context down twice are all harmless: a handle that is not registered any more is left alone, and
// a context takes the indexes still bound to it along.
static std::mutex g_reg_mu;
static std::unordered_set<const void*> g_live_indexes, g_live_contexts;

// Handles are registered by address, so an address must never come back: the host block of a destroyed handle is shrunk
// to the allocator's minimum and kept for the life of the process instead of being returned.  A stale handle handed to
// a destroy call again (a garbage-collected wrapper that outlived its context, say) can then never name a newer object.
template <typename T>
static T* handle_alloc() {
    void* m = malloc(sizeof(T));
    return m ? new (m) T() : nullptr;
}
template <typename T>
static void handle_retire(T* h) {
    h->~T();
    void* t = realloc((void*)h, 1);
    if (t && t != (void*)h) free(t);  // the allocator moved the block (it does not when shrinking): nothing left to hold
}

needle_hip_index* nh::index_new() {
    needle_hip_index* ix = handle_alloc<needle_hip_index>();
    if (ix) {
        std::lock_guard g(g_reg_mu);
        g_live_indexes.insert(ix);
    }
    return ix;
}

static void register_context(needle_hip_ctx* c) {
    std::lock_guard g(g_reg_mu);
    g_live_contexts.insert(c);
}

extern "C" needle_hip_status needle_hip_init(int device_id, needle_hip_ctx** out_ctx) {
    if (!out_ctx) return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "out_ctx is null");
    *out_ctx = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return fail(NEEDLE_HIP_ERR_NO_DEVICE, "no HIP device is visible; this library has no CPU path");
    if (device_id < 0 || device_id >= count)
        return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "device %d out of range (%d visible)", device_id, count);
    HIP_TRY(hipSetDevice(device_id));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device_id));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(NEEDLE_HIP_ERR_NO_DEVICE, "device %d is %s; this library is built for gfx950 only", device_id,
                    prop.gcnArchName);
    needle_hip_ctx* c = handle_alloc<needle_hip_ctx>();
    if (!c) return fail(NEEDLE_HIP_ERR_OUT_OF_MEMORY, "host allocation failed");
    c->device = device_id;
    c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    register_context(c);
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    for (int i = 0; i < 4 && e == hipSuccess; ++i) e = hipEventCreate(&c->ev[i]);
    if (e != hipSuccess) {
        needle_hip_shutdown(c);
        return fail(NEEDLE_HIP_ERR_HIP, "HIP error %d (%s) creating stream/events", (int)e, hipGetErrorString(e));
    }
    const needle_hip_status st = topk_static_setup(c);  // device_id is current (hipSetDevice above)
    if (st != NEEDLE_HIP_OK) { needle_hip_shutdown(c); return st; }
    *out_ctx = c;
    return NEEDLE_HIP_OK;
}

hipError_t nh::ensure_side_stream(needle_hip_ctx* c) {
    if (c->side) return hipSuccess;
    // lowest priority: what runs here (a wide, memory-heavy kernel) should fill the chip beside the short dependent kernels
    // of the main stream, not stand in their way
    int least = 0, greatest = 0;
    hipError_t e = hipDeviceGetStreamPriorityRange(&least, &greatest);
    if (e == hipSuccess) e = hipStreamCreateWithPriority(&c->side, hipStreamNonBlocking, least);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming);
    if (e != hipSuccess) destroy_side_stream(c);
    return e;
}

void nh::destroy_side_stream(needle_hip_ctx* c) {
    if (c->side) (void)hipStreamSynchronize(c->side);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->side) (void)hipStreamDestroy(c->side);
    c->side = nullptr; c->ev_fork = nullptr; c->ev_join = nullptr;
}

extern "C" void needle_hip_shutdown(needle_hip_ctx* c) {
    if (!c) return;
    std::vector<needle_hip_index*> bound;
    {
        std::lock_guard g(g_reg_mu);
        if (!g_live_contexts.erase(c)) return;
        for (const void* p : g_live_indexes)
            if (((const needle_hip_index*)p)->ctx == c) bound.push_back((needle_hip_index*)p);
    }
    for (needle_hip_index* ix : bound) nh::index_free(ix);  // (a nested centroid index may already be gone: no-op)
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    destroy_side_stream(c);
    for (auto& b : c->scratch) b.release();
    for (int i = 0; i < 4; ++i)
        if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    handle_retire(c);
}

extern "C" needle_hip_status needle_hip_set_profiling(needle_hip_ctx* c, int on) {
    if (!c) return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "ctx is null");
    std::lock_guard g(c->mu);
    c->profiling = on != 0;
    return NEEDLE_HIP_OK;
}

extern "C" needle_hip_status needle_hip_get_metrics(needle_hip_ctx* c, needle_hip_metrics* out) {
    if (!c || !out) return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "null argument");
    std::lock_guard g(c->mu);
    *out = c->m;
    return NEEDLE_HIP_OK;
}

extern "C" needle_hip_status needle_hip_set_tuning(needle_hip_ctx* c, const char* key, double value) {
    if (!c || !key) return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "null argument");
    if (!(value >= 0.0) || value > 4.0e9) return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "tuning value out of range");
    std::lock_guard g(c->mu);
    Tuning& t = c->tune;
    const uint32_t u = (uint32_t)value;
    const std::string k = key;
    if (k == "reset") t = Tuning{};
    else if (k == "debug") t.debug = u;
    else if (k == "splits") t.splits = u;
    else if (k == "pre_tiles") t.pre_tiles = (int32_t)u;
    else if (k == "kseed") t.kseed = u;
    else if (k == "band") t.band = (float)value;
    else if (k == "stage0" || k == "stage1" || k == "stage2" || k == "stage3") {
        const uint32_t i = (uint32_t)(k[5] - '0');
        if (u == 0) t.n_stage_pct = std::min(t.n_stage_pct, i);  // stageN = 0 ends the schedule before launch N
        else { t.stage_pct[i] = u; t.n_stage_pct = std::max(t.n_stage_pct, i + 1); }
    } else if (k == "no_compact") t.no_compact = u;
    else if (k == "compact_pct") t.compact_pct = u;
    else if (k == "compact_min_rows") t.compact_min_rows = u;
    else if (k == "no_dense") t.no_dense = u;
    else if (k == "no_set_only") t.no_set_only = u;
    else if (k == "merge_keys") t.merge_keys = u;
    else if (k == "merge_lists") t.merge_lists = u;
    else if (k == "rs_cap") t.rs_cap = u;
    else if (k == "rs_dc") t.rs_dc = u;
    else if (k == "ivf_no_seed") t.ivf_no_seed = u;
    else if (k == "ivf_seed_rows") t.ivf_seed_rows = u;
    else if (k == "ivf_split") t.ivf_split = u;
    else if (k == "binary_two_scans") t.binary_two_scans = u;
    else if (k == "trace") g_trace = u != 0;
    else if (k == "poison") g_poison = u != 0;
    else return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "unknown tuning key '%s'", key);
    return NEEDLE_HIP_OK;
}

extern "C" needle_hip_status needle_hip_reset_metrics(needle_hip_ctx* c) {
    if (!c) return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "ctx is null");
    std::lock_guard g(c->mu);
    c->m = needle_hip_metrics{};
    return NEEDLE_HIP_OK;
}

// ---------------------------------------------------------------------------
// flat index build
// ---------------------------------------------------------------------------
static bool metric_has_first_pass(int metric) { return metric <= NEEDLE_METRIC_DOT_PRODUCT; }
static void release_workspace(needle_hip_index* ix);

void nh::index_free(needle_hip_index* ix) {
    if (!ix) return;
    {
        std::lock_guard g(g_reg_mu);
        if (!g_live_indexes.erase(ix)) return;
    }
    (void)hipSetDevice(ix->ctx->device);
    (void)hipStreamSynchronize(ix->ctx->stream);
    free_search_slots(ix);
    if (ix->centroids) nh::index_free(ix->centroids);
    if (ix->d_list_pos) (void)hipFree(ix->d_list_pos);
    if (ix->d_list_len) (void)hipFree(ix->d_list_len);
    if (ix->d_pos2id) (void)hipFree(ix->d_pos2id);
    if (ix->d_codes) (void)hipFree(ix->d_codes);
    if (ix->d_codes_gm) (void)hipFree(ix->d_codes_gm);
    if (ix->d_codebooks) (void)hipFree(ix->d_codebooks);
    if (ix->d_sq_min) (void)hipFree(ix->d_sq_min);
    if (ix->d_sq_scale) (void)hipFree(ix->d_sq_scale);
    if (ix->d_sq_inv) (void)hipFree(ix->d_sq_inv);
    if (ix->d_op_f32 && !ix->op_f32_alias) (void)hipFree(ix->d_op_f32);
    if (ix->d_raw) (void)hipFree(ix->d_raw);
    if (ix->d_op_bf16) (void)hipFree(ix->d_op_bf16);
    if (ix->d_cinit) (void)hipFree(ix->d_cinit);
    if (ix->d_norms) (void)hipFree(ix->d_norms);
    if (ix->d_valid) (void)hipFree(ix->d_valid);
    if (ix->d_unsafe) (void)hipFree(ix->d_unsafe);
    release_workspace(ix);
    handle_retire(ix);
}

// ---------------------------------------------------------------------------
// search slots (re-entrant searches)
// ---------------------------------------------------------------------------
static void release_workspace(needle_hip_index* ix) {
    ix->w_scores.release();
    for (DevBuf* b : {&ix->w_qraw, &ix->w_qop, &ix->w_qn2, &ix->w_lists, &ix->w_lcnt, &ix->w_cand, &ix->w_candn,
                      &ix->w_tau, &ix->w_flagged, &ix->w_flagcnt, &ix->w_filter, &ix->w_premax, &ix->w_tau0, &ix->x_keys, &ix->x_u32,
                      &ix->x_u64, &ix->x_sel, &ix->x_mask, &ix->v_probe_ids, &ix->v_probe_d, &ix->v_probe_n, &ix->v_probe32,
                      &ix->v_items, &ix->v_itemq, &ix->v_lbase, &ix->w_qerr, &ix->r_idx, &ix->r_q, &ix->r_ids, &ix->r_dists, &ix->r_cnt, &ix->v_regroup, &ix->v_probe_sel, &ix->b_work, &ix->w_cinit, &ix->w_cscore, &ix->w_pqtab, &ix->w_extra})
        b->release();
}

// The view `v` sees the resident data of `p` (pointers and sizes; nothing is owned) through context `vc`.
static void sync_view(needle_hip_index* v, const needle_hip_index* p, needle_hip_ctx* vc, bool first) {
    v->ctx = vc;
    v->is_view = true;
    v->kind = p->kind; v->n = p->n; v->dim = p->dim; v->n_pad = p->n_pad; v->metric = p->metric; v->arith = p->arith;
    v->d_raw = p->d_raw; v->d_op_f32 = p->d_op_f32; v->op_f32_alias = p->op_f32_alias; v->d_op_bf16 = p->d_op_bf16;
    v->dpad_f32 = p->dpad_f32; v->dpad_bf16 = p->dpad_bf16; v->d_cinit = p->d_cinit; v->d_norms = p->d_norms;
    v->d_valid = p->d_valid; v->max_norm = p->max_norm; v->cos_norm_unsafe = p->cos_norm_unsafe;
    v->d_unsafe = p->d_unsafe; v->n_unsafe = p->n_unsafe;
    v->bf16_extra_slack = p->bf16_extra_slack; v->bf16_err = p->bf16_err;
    v->nlist = p->nlist; v->d_list_pos = p->d_list_pos; v->d_list_len = p->d_list_len; v->d_pos2id = p->d_pos2id;
    v->d_codes = p->d_codes; v->d_codes_gm = p->d_codes_gm; v->d_codebooks = p->d_codebooks; v->pq_m = p->pq_m; v->pq_dsub = p->pq_dsub;
    v->bin_words = p->bin_words; v->d_sq_min = p->d_sq_min; v->d_sq_scale = p->d_sq_scale; v->d_sq_inv = p->d_sq_inv;
    v->sq_min_norm = p->sq_min_norm;
    if (first) {  // host-side list geometry: fixed once the index is built
        v->list_pos = p->list_pos; v->list_len = p->list_len; v->list_tiles_desc = p->list_tiles_desc;
    }
}

static SearchSlot* new_search_slot(const needle_hip_index* parent) {
    SearchSlot* s = new (std::nothrow) SearchSlot();
    if (!s) return nullptr;
    s->ctx.device = parent->ctx->device;
    s->ctx.n_cu = parent->ctx->n_cu;
    s->ctx.topk_static = parent->ctx->topk_static;
    s->ctx.topk_static_bytes = parent->ctx->topk_static_bytes;
    hipError_t e = hipStreamCreateWithFlags(&s->ctx.stream, hipStreamNonBlocking);
    for (int i = 0; i < 4 && e == hipSuccess; ++i) e = hipEventCreate(&s->ctx.ev[i]);
    if (e != hipSuccess) {
        for (int i = 0; i < 4; ++i) if (s->ctx.ev[i]) (void)hipEventDestroy(s->ctx.ev[i]);
        if (s->ctx.stream) (void)hipStreamDestroy(s->ctx.stream);
        delete s;
        return nullptr;
    }
    return s;
}

void nh::free_search_slots(needle_hip_index* ix) {
    for (SearchSlot* s : ix->slots) {
        (void)hipStreamSynchronize(s->ctx.stream);
        destroy_side_stream(&s->ctx);
        release_workspace(&s->ix);
        release_workspace(&s->cent);
        release_workspace(&s->compact);
        for (DevBuf* b : {&s->z_raw, &s->z_op16, &s->z_op32, &s->z_cinit, &s->z_pos, &s->z_cnt}) b->release();
        for (auto& b : s->ctx.scratch) b.release();
        for (int i = 0; i < 4; ++i) if (s->ctx.ev[i]) (void)hipEventDestroy(s->ctx.ev[i]);
        (void)hipStreamDestroy(s->ctx.stream);
        delete s;
    }
    ix->slots.clear();
}

static bool search_uses_bf16_copy(const needle_hip_index* ix, const needle_search_params* sp) {
    if (ix->kind == KIND_BINARY || ix->n == 0 || sp->k == 0) return false;
    if (ix->kind == KIND_IVF && ix->pq_m) return false;  // built with the index
    if (sp->precision != NEEDLE_PRECISION_BF16 || flat_scan_bf16_dpad(ix->dim) == 0) return false;
    if (ix->kind == KIND_FLAT && !metric_has_first_pass(ix->metric)) return false;
    const uint32_t kp = choose_kprime((uint32_t)std::min<uint64_t>(sp->k, ix->n), NEEDLE_PRECISION_BF16);
    if (ix->cos_norm_unsafe && (uint64_t)kp + ix->n_unsafe > (uint32_t)KPRIME_MAX) return false;  // served by the exact scan
    return kp <= (uint32_t)KPRIME_MAX;
}

nh::SlotLease::SlotLease(needle_hip_index* parent, const needle_search_params* params) : parent_(parent) {
    if (hipSetDevice(parent->ctx->device) != hipSuccess) { st_ = fail(NEEDLE_HIP_ERR_HIP, "hipSetDevice failed"); return; }
    bool first = false;
    {
        std::unique_lock<std::mutex> lk(parent->pool_mu);
        for (;;) {
            for (SearchSlot* s : parent->slots)
                if (!s->busy) { slot_ = s; break; }
            if (slot_) break;
            if (parent->slots.size() < MAX_SEARCH_SLOTS) {
                slot_ = new_search_slot(parent);
                if (!slot_) { st_ = fail(NEEDLE_HIP_ERR_OUT_OF_MEMORY, "could not create a search slot (stream / events)"); return; }
                parent->slots.push_back(slot_);
                first = true;
                break;
            }
            parent->pool_cv.wait(lk);
        }
        slot_->busy = true;
    }
    // operand copies a view must not build for itself, then the view of the parent's data
    std::lock_guard g(parent->mu);
    if (st_ == NEEDLE_HIP_OK && search_uses_bf16_copy(parent, params)) st_ = ensure_bf16_copy(parent);
    if (st_ == NEEDLE_HIP_OK && parent->kind == KIND_IVF && parent->centroids && parent->centroids->n > 0 &&
        flat_scan_bf16_dpad(parent->dim) != 0)
        st_ = ensure_bf16_copy(parent->centroids);  // the probe step always scores the centroids in bf16
    slot_->ctx.profiling = parent->ctx->profiling;
    slot_->ctx.tune = parent->ctx->tune;
    slot_->ctx.m = needle_hip_metrics{};
    if (g_trace)
        fprintf(stderr, "[needle_hip] index %p: search takes slot %p (%s), stream %p\n", (void*)parent, (void*)slot_,
                first ? "new" : "reused", (void*)slot_->ctx.stream);
    sync_view(&slot_->ix, parent, &slot_->ctx, first);
    slot_->ix.slot = slot_;
    if (parent->centroids) {
        sync_view(&slot_->cent, parent->centroids, &slot_->ctx, first);
        slot_->ix.centroids = &slot_->cent;
    }
}

nh::SlotLease::~SlotLease() {
    if (!slot_) return;
    if (g_trace) fprintf(stderr, "[needle_hip] index %p: slot %p released\n", (void*)parent_, (void*)slot_);
    {
        std::lock_guard g(parent_->mu);  // what the search learned about the index
        parent_->bf16_extra_slack = std::max(parent_->bf16_extra_slack, slot_->ix.bf16_extra_slack);
        if (parent_->centroids) parent_->centroids->bf16_extra_slack = std::max(parent_->centroids->bf16_extra_slack, slot_->cent.bf16_extra_slack);
    }
    {
        needle_hip_ctx* pc = parent_->ctx;
        const needle_hip_metrics& d = slot_->ctx.m;
        std::lock_guard g(pc->mu);
        pc->m.searches += d.searches; pc->m.queries += d.queries; pc->m.fallback_queries += d.fallback_queries;
        pc->m.scan_launches += d.scan_launches; pc->m.scan_ms_total += d.scan_ms_total;
        pc->m.refined_queries += d.refined_queries; pc->m.binary_code_scans += d.binary_code_scans;
        pc->m.compacted_searches += d.compacted_searches; pc->m.compacted_rows += d.compacted_rows;
        pc->m.list_scans += d.list_scans; pc->m.list_scan_ms_total += d.list_scan_ms_total;
        if (d.scan_launches) pc->m.last_scan_ms = d.last_scan_ms;
        if (d.searches && slot_->ctx.profiling) pc->m.last_search_ms = d.last_search_ms;
    }
    {
        std::lock_guard lk(parent_->pool_mu);
        slot_->busy = false;
    }
    parent_->pool_cv.notify_one();
}


needle_hip_status nh::upload_mask(needle_hip_index* ix, const uint32_t* mask_host) {
    // the new mask is uploaded in full before it replaces the old one: a failure leaves the index as it was
    uint32_t* fresh = nullptr;
    if (mask_host && ix->n > 0) {
        const size_t words = (ix->n + 31) / 32;
        HIP_TRY(hipMalloc((void**)&fresh, words * sizeof(uint32_t)));
        hipError_t e = hipMemcpyAsync(fresh, mask_host, words * sizeof(uint32_t), hipMemcpyHostToDevice, ix->ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ix->ctx->stream);
        if (e != hipSuccess) { (void)hipFree(fresh); HIP_TRY(e); }
    }
    if (ix->d_valid) (void)hipFree(ix->d_valid);
    ix->d_valid = fresh;
    return NEEDLE_HIP_OK;
}

// d_raw already holds the rows (IVF: all n_pad list-ordered positions); derive everything else.
needle_hip_status nh::finish_operands(needle_hip_index* ix, bool validate) {
    needle_hip_ctx* c = ix->ctx;
    if (ix->n == 0) return NEEDLE_HIP_OK;
    const uint64_t rows = ix->kind == KIND_IVF ? ix->n_pad : ix->n;
    uint32_t* d_flag = nullptr;
    HIP_TRY(hipMalloc((void**)&d_flag, 2 * sizeof(uint32_t)));
    hipError_t e = hipSuccess;
    if (validate) {
        // validate_insert_input, src/collection/validation.rs:44-52: no NaN / Inf rows
        e = launch_validate_rows(ix->d_raw, rows, ix->dim, d_flag, c->stream);
        uint32_t bad = 0xffffffffu;
        if (e == hipSuccess) e = hipMemcpyAsync(&bad, d_flag, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) { (void)hipFree(d_flag); HIP_TRY(e); }
        if (bad != 0xffffffffu) {
            (void)hipFree(d_flag);
            return fail(NEEDLE_HIP_ERR_INVALID_VECTOR, "Invalid vector: row %u contains NaN or Inf", bad);
        }
    }
    if (!metric_has_first_pass(ix->metric)) { (void)hipFree(d_flag); return NEEDLE_HIP_OK; }

    const bool normalize = ix->metric == NEEDLE_METRIC_COSINE;
    ix->dpad_f32 = round_up(ix->dim, flat_scan_kstep(NEEDLE_PRECISION_FP32));
    ix->op_f32_alias = !normalize && ix->dpad_f32 == ix->dim;
    e = hipSuccess;
    if (ix->op_f32_alias) {
        ix->d_op_f32 = ix->d_raw;
    } else {
        const size_t bytes = (size_t)ix->n_pad * ix->dpad_f32 * sizeof(float);
        e = hipMalloc((void**)&ix->d_op_f32, bytes);
        if (e == hipSuccess) e = hipMemsetAsync(ix->d_op_f32, 0, bytes, c->stream);
    }
    if (e == hipSuccess) e = hipMalloc((void**)&ix->d_norms, (size_t)ix->n_pad * sizeof(float));
    if (e == hipSuccess && ix->metric == NEEDLE_METRIC_EUCLIDEAN) {
        e = hipMalloc((void**)&ix->d_cinit, (size_t)ix->n_pad * sizeof(float));
        if (e == hipSuccess) e = hipMemsetAsync(ix->d_cinit, 0, (size_t)ix->n_pad * sizeof(float), c->stream);
    }
    uint32_t bits[2] = {0u, 0x7f7fffffu};  // running maximum norm, running minimum of the non-zero norms
    if (e == hipSuccess) e = hipMemcpyAsync(d_flag, bits, sizeof(bits), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess)
        e = launch_row_prep(ix->d_raw, rows, ix->dim, ix->dpad_f32, normalize, ix->op_f32_alias ? nullptr : ix->d_op_f32,
                            nullptr, ix->d_cinit, ix->d_norms, nullptr, d_flag, nullptr, nullptr, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(bits, d_flag, sizeof(bits), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d_flag);
    HIP_TRY(e);
    memcpy(&ix->max_norm, &bits[0], sizeof(float));
    float min_norm;
    memcpy(&min_norm, &bits[1], sizeof(float));
    ix->min_norm = min_norm;
    return refresh_unsafe_rows(ix);
}

// Cosine: the first pass scores rows by their true cosine (normalised in double), but the reference's f32 dot(b, b)
// underflows for rows below about 1e-19 and overflows above about 1e19, which moves its distance by tens of percent or
// forces it to 1.0.  No bound on a first-pass score covers that.  Such rows are therefore kept out of every first pass
// (accumulator seed -inf) and listed on the side: a search re-scores them exactly for every query, next to the
// query's own candidates, and the bound check speaks about the other rows only.  An index with more of them than a
// search can carry as extra candidates serves that search from the exact scan.
needle_hip_status nh::refresh_unsafe_rows(needle_hip_index* ix) {
    if (ix->kind != KIND_FLAT || ix->metric != NEEDLE_METRIC_COSINE || ix->n == 0 || !ix->d_norms) {
        ix->n_unsafe = 0;
        ix->cos_norm_unsafe = false;
        return NEEDLE_HIP_OK;
    }
    // the running extrema bound every row's norm (they are not lowered when rows are removed)
    if (!(ix->min_norm < 1e-15f || ix->max_norm > 1e15f) && !ix->d_unsafe) return NEEDLE_HIP_OK;
    needle_hip_ctx* c = ix->ctx;
    const size_t rows_alloc = std::max(ix->n_pad, ix->cap_pad);
    if (!ix->d_cinit) {
        HIP_TRY(hipMalloc((void**)&ix->d_cinit, rows_alloc * sizeof(float)));
        HIP_TRY(hipMemsetAsync(ix->d_cinit, 0, rows_alloc * sizeof(float), c->stream));
    }
    if (!ix->d_unsafe) HIP_TRY(hipMalloc((void**)&ix->d_unsafe, (UNSAFE_ROWS_MAX + 1) * sizeof(uint32_t)));
    uint32_t count = 0;
    HIP_TRY(launch_unsafe_rows(ix->d_norms, ix->n, ix->d_cinit, ix->d_unsafe, UNSAFE_ROWS_MAX, ix->d_unsafe + UNSAFE_ROWS_MAX, c->stream));
    HIP_TRY(hipMemcpyAsync(&count, ix->d_unsafe + UNSAFE_ROWS_MAX, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    ix->n_unsafe = count;
    ix->cos_norm_unsafe = count > 0;
    return NEEDLE_HIP_OK;
}

static needle_hip_status flat_alloc(needle_hip_ctx* c, uint64_t n, uint32_t dim, needle_metric metric,
                                    needle_arith arith, needle_hip_index** out) {
    if (!c || !out) return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    if (dim == 0) return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "dimension must be positive");
    if ((int)metric < 0 || (int)metric > NEEDLE_METRIC_CHEBYSHEV)
        return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "unknown metric %d", (int)metric);
    if (arith != NEEDLE_ARITH_DISTANCE_RS && arith != NEEDLE_ARITH_ACCEL_CHUNK4 && arith != NEEDLE_ARITH_AVX2 &&
        arith != NEEDLE_ARITH_AVX512 && arith != NEEDLE_ARITH_NEON)
        return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "unknown arithmetic %d", (int)arith);
    if (arith == NEEDLE_ARITH_ACCEL_CHUNK4 && metric != NEEDLE_METRIC_COSINE && metric != NEEDLE_METRIC_EUCLIDEAN &&
        metric != NEEDLE_METRIC_DOT_PRODUCT)
        return fail(NEEDLE_HIP_ERR_UNSUPPORTED, "the accelerator arithmetic covers cosine, euclidean and dot product only");
    if (n >= 0xfffffff0ull - TILE_ROWS) return fail(NEEDLE_HIP_ERR_UNSUPPORTED, "at most 2^32-272 rows per index");
    HIP_TRY(hipSetDevice(c->device));
    needle_hip_index* ix = index_new();
    if (!ix) return fail(NEEDLE_HIP_ERR_OUT_OF_MEMORY, "host allocation failed");
    ix->ctx = c;
    ix->n = n;
    ix->dim = dim;
    ix->metric = metric;
    ix->arith = arith;
    ix->n_pad = round_up((uint32_t)n, TILE_ROWS);
    ix->cap_pad = ix->n_pad;
    if (n > 0) {
        const size_t bytes = (size_t)ix->n_pad * dim * sizeof(float);
        hipError_t e = hipMalloc((void**)&ix->d_raw, bytes);
        if (e == hipSuccess && ix->n_pad > n)
            e = hipMemsetAsync(ix->d_raw + (size_t)n * dim, 0, (size_t)(ix->n_pad - n) * dim * sizeof(float), c->stream);
        if (e != hipSuccess) { index_free(ix); HIP_TRY(e); }
    }
    *out = ix;
    return NEEDLE_HIP_OK;
}

extern "C" needle_hip_status needle_hip_flat_build(needle_hip_ctx* c, const float* base, uint64_t n, uint32_t dim,
                                                   needle_metric metric, needle_arith arith,
                                                   const uint32_t* valid_bitmask, needle_hip_index** out) {
    if (n > 0 && !base) return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "base is null");
    needle_hip_index* ix = nullptr;
    ST_TRY(flat_alloc(c, n, dim, metric, arith, &ix));
    needle_hip_status st = NEEDLE_HIP_OK;
    if (n > 0) {
        hipError_t e = hipMemcpyAsync(ix->d_raw, base, (size_t)n * dim * sizeof(float), hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) st = fail(NEEDLE_HIP_ERR_HIP, "HIP error %d (%s) uploading the base", (int)e, hipGetErrorString(e));
    }
    if (st == NEEDLE_HIP_OK) st = finish_operands(ix, true);
    if (st == NEEDLE_HIP_OK) st = upload_mask(ix, valid_bitmask);
    if (st != NEEDLE_HIP_OK) { index_free(ix); *out = nullptr; return st; }
    *out = ix;
    return NEEDLE_HIP_OK;
}

extern "C" needle_hip_status needle_hip_flat_build_half(needle_hip_ctx* c, const uint16_t* rows, uint64_t n, uint32_t dim,
                                                        int half_kind, needle_metric metric, needle_arith arith,
                                                        const uint32_t* valid_bitmask, needle_hip_index** out) {
    if (!c || !out) return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    if (half_kind != NEEDLE_HALF_F16 && half_kind != NEEDLE_HALF_BF16)
        return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "unknown half precision kind %d", half_kind);
    if (n > 0 && !rows) return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "rows is null");
    if (dim == 0) return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "dimension must be positive");
    if (n >= 0xfffffff0ull - TILE_ROWS) return fail(NEEDLE_HIP_ERR_UNSUPPORTED, "at most 2^32-272 rows per index");
    HIP_TRY(hipSetDevice(c->device));
    // upload the 16-bit rows (half the bytes of an f32 base), widen them on the device, then the usual build
    uint16_t* d_half = nullptr;
    float* d_rows = nullptr;
    const uint64_t total = n * dim;
    hipError_t e = hipSuccess;
    if (total > 0) {
        e = hipMalloc((void**)&d_half, total * sizeof(uint16_t));
        if (e == hipSuccess) e = hipMalloc((void**)&d_rows, total * sizeof(float));
        if (e == hipSuccess) e = hipMemcpyAsync(d_half, rows, total * sizeof(uint16_t), hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = launch_half_widen(d_half, total, half_kind == NEEDLE_HALF_BF16, d_rows, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    }
    needle_hip_status st = NEEDLE_HIP_OK;
    if (e != hipSuccess)
        st = fail(e == hipErrorOutOfMemory ? NEEDLE_HIP_ERR_OUT_OF_MEMORY : NEEDLE_HIP_ERR_HIP, "HIP error %d (%s) widening the rows", (int)e,
                  hipGetErrorString(e));
    if (st == NEEDLE_HIP_OK) st = needle_hip_flat_build_device(c, d_rows, n, dim, metric, arith, valid_bitmask, out);
    if (d_half) (void)hipFree(d_half);
    if (d_rows) (void)hipFree(d_rows);
    return st;
}

extern "C" needle_hip_status needle_hip_flat_build_rows(needle_hip_ctx* c, const float* const* rows, uint64_t n,
                                                        uint32_t dim, needle_metric metric, needle_arith arith,
                                                        const uint32_t* valid_bitmask, needle_hip_index** out) {
    if (n > 0 && !rows) return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "rows is null");
    // flatten Vec<Vec<f32>> into one staging block, then the contiguous path
    std::vector<float> flat;
    try {
        flat.resize((size_t)n * dim);
    } catch (...) {
        return fail(NEEDLE_HIP_ERR_OUT_OF_MEMORY, "host staging allocation failed");
    }
    for (uint64_t i = 0; i < n; ++i) {
        if (!rows[i]) return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "row %llu is null", (unsigned long long)i);
        memcpy(flat.data() + (size_t)i * dim, rows[i], (size_t)dim * sizeof(float));
    }
    return needle_hip_flat_build(c, flat.data(), n, dim, metric, arith, valid_bitmask, out);
}

extern "C" needle_hip_status needle_hip_flat_build_device(needle_hip_ctx* c, const float* d_base, uint64_t n,
                                                          uint32_t dim, needle_metric metric, needle_arith arith,
                                                          const uint32_t* valid_bitmask_host, needle_hip_index** out) {
    if (n > 0 && !d_base) return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "d_base is null");
    needle_hip_index* ix = nullptr;
    ST_TRY(flat_alloc(c, n, dim, metric, arith, &ix));
    needle_hip_status st = NEEDLE_HIP_OK;
    if (n > 0) {
        hipError_t e = hipMemcpyAsync(ix->d_raw, d_base, (size_t)n * dim * sizeof(float), hipMemcpyDeviceToDevice, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) st = fail(NEEDLE_HIP_ERR_HIP, "HIP error %d (%s) copying the base", (int)e, hipGetErrorString(e));
    }
    if (st == NEEDLE_HIP_OK) st = finish_operands(ix, true);
    if (st == NEEDLE_HIP_OK) st = upload_mask(ix, valid_bitmask_host);
    if (st != NEEDLE_HIP_OK) { index_free(ix); *out = nullptr; return st; }
    *out = ix;
    return NEEDLE_HIP_OK;
}

extern "C" needle_hip_status needle_hip_index_set_valid_mask(needle_hip_index* ix, const uint32_t* valid_bitmask) {
    if (!ix) return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "index is null");
    std::lock_guard g(ix->mu);
    HIP_TRY(hipSetDevice(ix->ctx->device));
    return upload_mask(ix, valid_bitmask);
}

// ---------------------------------------------------------------------------
// append (GpuResidentIndex::add / add_batch, src/experimental/gpu/mod.rs:1102-1160)
// ---------------------------------------------------------------------------
// A resident array of `rows_old` rows of `row_bytes` bytes moves to a zeroed allocation for `rows_new` rows (device to
// device: nothing is uploaded again).
static hipError_t grow_rows(void** p, size_t row_bytes, uint64_t rows_old, uint64_t rows_new, hipStream_t stream) {
    if (!*p && rows_old > 0) return hipErrorInvalidValue;
    void* fresh = nullptr;
    hipError_t e = hipMalloc(&fresh, (size_t)rows_new * row_bytes);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync((char*)fresh + (size_t)rows_old * row_bytes, 0, (size_t)(rows_new - rows_old) * row_bytes, stream);
    if (e == hipSuccess && rows_old > 0) e = hipMemcpyAsync(fresh, *p, (size_t)rows_old * row_bytes, hipMemcpyDeviceToDevice, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e != hipSuccess) { (void)hipFree(fresh); return e; }
    if (*p) (void)hipFree(*p);
    *p = fresh;
    return hipSuccess;
}

static needle_hip_status flat_append(needle_hip_index* ix, const float* rows, bool rows_on_device, uint64_t n_new,
                                     const uint32_t* new_valid_bitmask) {
    if (!ix) return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "index is null");
    if (ix->kind != KIND_FLAT || ix->is_view)
        return fail(NEEDLE_HIP_ERR_UNSUPPORTED, "rows can be appended to flat indexes only (inverted and quantized indexes are rebuilt)");
    if (n_new == 0) return NEEDLE_HIP_OK;
    if (!rows) return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "rows is null");
    if (ix->n + n_new >= 0xfffffff0ull - TILE_ROWS) return fail(NEEDLE_HIP_ERR_UNSUPPORTED, "at most 2^32-272 rows per index");
    std::lock_guard g(ix->mu);
    needle_hip_ctx* c = ix->ctx;
    HIP_TRY(hipSetDevice(c->device));
    const uint64_t n_old = ix->n, n_all = n_old + n_new;
    const uint32_t pad_old = ix->n_pad, pad_all = round_up((uint32_t)n_all, TILE_ROWS);
    const uint32_t dim = ix->dim;
    const bool first_pass = metric_has_first_pass(ix->metric);
    if (first_pass && ix->dpad_f32 == 0) {  // an index built empty: the operand layout was never fixed
        ix->dpad_f32 = round_up(dim, flat_scan_kstep(NEEDLE_PRECISION_FP32));
        ix->op_f32_alias = ix->metric != NEEDLE_METRIC_COSINE && ix->dpad_f32 == dim;
    }
    // 1. capacity: every resident array moves to an allocation of twice the rows when the new rows do not fit
    if (pad_all > ix->cap_pad) {
        const uint32_t cap = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(2ull * ix->cap_pad, pad_all), 0xffffff80ull);
        const uint64_t have = pad_old;
        HIP_TRY(grow_rows((void**)&ix->d_raw, (size_t)dim * sizeof(float), have, cap, c->stream));
        if (first_pass) {
            if (ix->op_f32_alias) ix->d_op_f32 = ix->d_raw;
            else HIP_TRY(grow_rows((void**)&ix->d_op_f32, (size_t)ix->dpad_f32 * sizeof(float), have, cap, c->stream));
            HIP_TRY(grow_rows((void**)&ix->d_norms, sizeof(float), have, cap, c->stream));
            if (ix->metric == NEEDLE_METRIC_EUCLIDEAN || ix->d_cinit)
                HIP_TRY(grow_rows((void**)&ix->d_cinit, sizeof(float), have, cap, c->stream));
            if (ix->d_op_bf16) HIP_TRY(grow_rows((void**)&ix->d_op_bf16, (size_t)ix->dpad_bf16 * sizeof(uint16_t), have, cap, c->stream));
        }
        ix->cap_pad = cap;
    }
    // 2. the new rows go behind the old ones and are checked there (validate_insert_input, src/collection/validation.rs:44-52);
    //    a rejected batch leaves the index as it was
    float* tail = ix->d_raw + (size_t)n_old * dim;
    HIP_TRY(hipMemcpyAsync(tail, rows, (size_t)n_new * dim * sizeof(float), rows_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                           c->stream));
    uint32_t* d_flag = nullptr;
    HIP_TRY(hipMalloc((void**)&d_flag, 4 * sizeof(uint32_t)));
    uint32_t bad = 0xffffffffu;
    hipError_t e = launch_validate_rows(tail, n_new, dim, d_flag, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(&bad, d_flag, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess && bad != 0xffffffffu) {
        (void)hipMemsetAsync(tail, 0, (size_t)(pad_all - n_old) * dim * sizeof(float), c->stream);  // padding rows are zero again
        (void)hipStreamSynchronize(c->stream);
        (void)hipFree(d_flag);
        return fail(NEEDLE_HIP_ERR_INVALID_VECTOR, "Invalid vector: appended row %u contains NaN or Inf", bad);
    }
    if (e == hipSuccess && pad_all > n_all)  // (an earlier, rejected batch may have left values behind the last row)
        e = hipMemsetAsync(ix->d_raw + (size_t)n_all * dim, 0, (size_t)(pad_all - n_all) * dim * sizeof(float), c->stream);
    // 3. operands, norms, seeds and error bounds of the new rows only; the running extrema continue from the index's
    if (e == hipSuccess && first_pass) {
        uint32_t bits[4];
        memcpy(&bits[0], &ix->max_norm, 4);
        memcpy(&bits[1], &ix->min_norm, 4);
        memcpy(&bits[2], &ix->bf16_err, 4);
        bits[3] = 0;
        const bool normalize = ix->metric == NEEDLE_METRIC_COSINE;
        e = hipMemcpyAsync(d_flag, bits, sizeof(bits), hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess)
            e = launch_row_prep(tail, n_new, dim, ix->dpad_f32, normalize,
                                ix->op_f32_alias ? nullptr : ix->d_op_f32 + (size_t)n_old * ix->dpad_f32, nullptr,
                                ix->metric == NEEDLE_METRIC_EUCLIDEAN ? ix->d_cinit + n_old : nullptr, ix->d_norms + n_old, nullptr, d_flag,
                                nullptr, nullptr, c->stream);
        if (e == hipSuccess && ix->d_op_bf16)
            e = launch_row_prep(tail, n_new, dim, ix->dpad_bf16, normalize, nullptr, ix->d_op_bf16 + (size_t)n_old * ix->dpad_bf16, nullptr,
                                nullptr, nullptr, nullptr, nullptr, d_flag + 2, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(bits, d_flag, sizeof(bits), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e == hipSuccess) {
            memcpy(&ix->max_norm, &bits[0], 4);
            memcpy(&ix->min_norm, &bits[1], 4);
            if (ix->d_op_bf16) memcpy(&ix->bf16_err, &bits[2], 4);
        }
    }
    (void)hipFree(d_flag);
    HIP_TRY(e);
    // 4. the live-row mask: bits of the new rows behind the old ones (all set where no mask is given)
    if (ix->d_valid || new_valid_bitmask) {
        const size_t w_old = (n_old + 31) / 32, w_all = (n_all + 31) / 32;
        std::vector<uint32_t> m;
        try {
            m.assign(w_all, 0u);
        } catch (...) {
            return fail(NEEDLE_HIP_ERR_OUT_OF_MEMORY, "host allocation failed");
        }
        if (ix->d_valid) {
            HIP_TRY(hipMemcpyAsync(m.data(), ix->d_valid, w_old * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
        } else {
            for (uint64_t i = 0; i < n_old; ++i) m[i >> 5] |= 1u << (i & 31);
        }
        for (uint64_t j = 0; j < n_new; ++j) {
            const uint64_t i = n_old + j;
            const bool live = !new_valid_bitmask || ((new_valid_bitmask[j >> 5] >> (j & 31)) & 1u);
            if (live) m[i >> 5] |= 1u << (i & 31); else m[i >> 5] &= ~(1u << (i & 31));
        }
        ix->n = n_all;  // upload_mask sizes the copy from the row count
        const needle_hip_status st = upload_mask(ix, m.data());
        if (st != NEEDLE_HIP_OK) { ix->n = n_old; return st; }
    }
    ix->n = n_all;
    ix->n_pad = pad_all;
    return refresh_unsafe_rows(ix);  // (cosine: the new rows may hold norms the first pass cannot bound)
}

// GpuResidentIndex::remove, gpu/mod.rs:1198-1207 (Vec::swap_remove): the last row takes the place of row `pos`, the
// index shrinks by one.  Row copies stay on the device; the norm and error extrema remain valid as bounds.
extern "C" needle_hip_status needle_hip_index_swap_remove(needle_hip_index* ix, uint64_t pos) {
    if (!ix) return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "index is null");
    if (ix->kind != KIND_FLAT || ix->is_view) return fail(NEEDLE_HIP_ERR_UNSUPPORTED, "rows can be removed from flat indexes only");
    if (pos >= ix->n) return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "row %llu out of range (%llu rows)", (unsigned long long)pos,
                                  (unsigned long long)ix->n);
    std::lock_guard g(ix->mu);
    needle_hip_ctx* c = ix->ctx;
    HIP_TRY(hipSetDevice(c->device));
    const uint64_t last = ix->n - 1;
    auto move_row = [&](void* arr, size_t row_bytes) -> hipError_t {  // arr[pos] = arr[last]; arr[last] = 0 (padding rows are zero)
        if (!arr) return hipSuccess;
        char* a = (char*)arr;
        hipError_t e = hipSuccess;
        if (pos != last) e = hipMemcpyAsync(a + pos * row_bytes, a + last * row_bytes, row_bytes, hipMemcpyDeviceToDevice, c->stream);
        if (e == hipSuccess) e = hipMemsetAsync(a + last * row_bytes, 0, row_bytes, c->stream);
        return e;
    };
    HIP_TRY(move_row(ix->d_raw, (size_t)ix->dim * sizeof(float)));
    if (!ix->op_f32_alias) HIP_TRY(move_row(ix->d_op_f32, (size_t)ix->dpad_f32 * sizeof(float)));
    HIP_TRY(move_row(ix->d_op_bf16, (size_t)ix->dpad_bf16 * sizeof(uint16_t)));
    HIP_TRY(move_row(ix->d_cinit, sizeof(float)));
    HIP_TRY(move_row(ix->d_norms, sizeof(float)));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (ix->d_valid) {  // the moved row keeps its bit
        const size_t words = (ix->n + 31) / 32;
        std::vector<uint32_t> m(words);
        HIP_TRY(hipMemcpy(m.data(), ix->d_valid, words * sizeof(uint32_t), hipMemcpyDeviceToHost));
        const bool live = (m[last >> 5] >> (last & 31)) & 1u;
        if (live) m[pos >> 5] |= 1u << (pos & 31); else m[pos >> 5] &= ~(1u << (pos & 31));
        m[last >> 5] &= ~(1u << (last & 31));
        ix->n = last;
        if (last == 0) { (void)hipFree(ix->d_valid); ix->d_valid = nullptr; }
        else ST_TRY(upload_mask(ix, m.data()));
    }
    ix->n = last;
    ix->n_pad = round_up((uint32_t)last, TILE_ROWS);
    return refresh_unsafe_rows(ix);  // (the side list names rows by id)
}

extern "C" needle_hip_status needle_hip_index_append(needle_hip_index* ix, const float* rows, uint64_t n_new,
                                                     const uint32_t* new_valid_bitmask) {
    return flat_append(ix, rows, false, n_new, new_valid_bitmask);
}

extern "C" needle_hip_status needle_hip_index_append_device(needle_hip_index* ix, const float* d_rows, uint64_t n_new,
                                                            const uint32_t* new_valid_bitmask_host) {
    return flat_append(ix, d_rows, true, n_new, new_valid_bitmask_host);
}

extern "C" void needle_hip_index_destroy(needle_hip_index* ix) { nh::index_free(ix); }
extern "C" uint64_t needle_hip_index_len(const needle_hip_index* ix) { return ix ? ix->n : 0; }
extern "C" uint32_t needle_hip_index_dimensions(const needle_hip_index* ix) { return ix ? ix->dim : 0; }

// ---------------------------------------------------------------------------
// search
// ---------------------------------------------------------------------------
uint32_t nh::choose_kprime(uint32_t k, int precision) {
    // candidates kept per query by the first pass: k plus slack for the rows
    // whose first-pass order differs from the reference order
    const uint32_t slack = precision == NEEDLE_PRECISION_BF16 ? 54 : 22;
    return round_up(k + slack, 16);
}

struct SplitPlan { uint32_t n_splits, tiles_per_split, xcd_map; };

// Slices of the base per query block.  Workgroups = q_tiles * n_splits should
// fill the chip in whole rounds; a multiple of 8 slices lets each XCD keep its
// own slices in its L2 while its 32 CUs sweep them with different query blocks.
static SplitPlan plan_splits(uint32_t q_tiles, uint32_t n_tiles, uint32_t slots, uint32_t max_splits) {
    SplitPlan best{1, n_tiles, 0};
    double best_score = -1.0;
    const uint32_t s_max = std::min<uint32_t>(n_tiles, std::min<uint32_t>(256, max_splits));  // few query blocks: more slices
    for (uint32_t s = 1; s <= s_max; ++s) {
        const uint32_t tps = (n_tiles + s - 1) / s;
        const uint32_t s_eff = (n_tiles + tps - 1) / tps;
        if (s_eff != s) continue;
        const uint64_t blocks = (uint64_t)s_eff * q_tiles;
        const uint64_t rounds = (blocks + slots - 1) / slots;
        double score = (double)blocks / (double)(rounds * slots);
        if (tps < 8 && s > 1) score -= 0.05 * (8 - tps);  // list warm-up dominates tiny slices
        score -= 0.0003 * s;
        const bool xcd = (s_eff % 8 == 0);
        if (xcd) score += 0.03;
        if (score > best_score) { best_score = score; best = SplitPlan{s_eff, tps, xcd ? 1u : 0u}; }
    }
    return best;
}

needle_hip_status nh::run_exact(needle_hip_index* ix, const float* d_queries, const uint32_t* d_qsel, uint32_t f_total,
                                uint32_t k, const uint32_t* d_filter, uint64_t* d_out_ids, float* d_out_dists,
                                uint32_t* d_out_counts, const uint32_t* d_probe, const uint32_t* d_probe_n,
                                uint32_t nprobe) {
    needle_hip_ctx* c = ix->ctx;
    if (f_total == 0) return NEEDLE_HIP_OK;
    const bool ivf = ix->kind == KIND_IVF;
    const uint32_t n = ivf ? ix->n_pad : (uint32_t)ix->n;  // IVF: rows are list-ordered positions
    const uint32_t mask_words = (n + 31) / 32;
    // slots per round: keep the key workspace near 512 MiB
    uint64_t fmax = (512ull << 20) / ((uint64_t)std::max<uint32_t>(n, 1) * 8ull);
    fmax = std::max<uint64_t>(1, std::min<uint64_t>(fmax, 4096));
    const uint32_t F = (uint32_t)std::min<uint64_t>(fmax, f_total);
    HIP_TRY(ix->x_keys.ensure((size_t)F * n * sizeof(uint64_t)));
    HIP_TRY(ix->x_u32.ensure((size_t)F * (4 + 256) * sizeof(uint32_t)));
    HIP_TRY(ix->x_u64.ensure((size_t)F * sizeof(uint64_t)));
    HIP_TRY(ix->x_sel.ensure((size_t)F * k * sizeof(uint64_t)));
    if (ivf) HIP_TRY(ix->x_mask.ensure((size_t)F * mask_words * sizeof(uint32_t)));
    for (uint32_t f0 = 0; f0 < f_total; f0 += F) {
        const uint32_t f = std::min(F, f_total - f0);
        ExactParams p{};
        p.base_raw = ix->d_raw;
        p.valid = ix->d_valid;
        p.filter = d_filter;
        p.keys = ix->x_keys.as<uint64_t>();
        uint32_t* u = ix->x_u32.as<uint32_t>();
        p.live = u; p.k_eff = u + F; p.kth_rem = u + 2 * F; p.sel_n = u + 3 * F; p.hist = u + 4 * F;
        p.prefix = ix->x_u64.as<uint64_t>();
        p.sel = ix->x_sel.as<uint64_t>();
        p.f = f; p.n = n; p.dim = ix->dim; p.k = k; p.metric = ix->metric; p.arith = ix->arith;
        if ((ix->kind == KIND_IVF && ix->pq_m) || ix->kind == KIND_PQ) {
            p.pq_codes = ix->d_codes; p.pq_books = ix->d_codebooks; p.pq_m = ix->pq_m; p.pq_dsub = ix->pq_dsub;
            p.pq_asym = ix->kind == KIND_IVF;
        }
        if (ix->kind == KIND_SQ8) {
            p.sq_codes = ix->d_codes; p.sq_min = ix->d_sq_min; p.sq_scale = ix->d_sq_scale; p.sq_stride = ix->dpad_f32;
        }
        if (ivf) {
            // rows a query may return = the rows of the lists it probes
            const uint32_t* sel = d_qsel ? d_qsel + f0 : nullptr;
            HIP_TRY(launch_ivf_mark_lists(sel, f, d_qsel ? d_probe : d_probe + (size_t)f0 * nprobe,
                                          d_qsel ? d_probe_n : d_probe_n + f0, nprobe, ix->d_list_pos, ix->d_list_len,
                                          ix->x_mask.as<uint32_t>(), mask_words, c->stream));
            p.pos2id = ix->d_pos2id;
            p.pos_filter = ix->x_mask.as<uint32_t>();
            p.pos_filter_words = mask_words;
        }
        if (d_qsel) {  // slots name arbitrary queries of the batch
            p.qsel = d_qsel + f0;
            p.query_raw = d_queries;
            p.out_ids = d_out_ids; p.out_dists = d_out_dists; p.out_counts = d_out_counts;
        } else {       // slot s is query f0 + s
            p.qsel = nullptr;
            p.query_raw = d_queries + (size_t)f0 * ix->dim;
            p.out_ids = d_out_ids + (size_t)f0 * k; p.out_dists = d_out_dists + (size_t)f0 * k;
            p.out_counts = d_out_counts + f0;
        }
        HIP_TRY(launch_exact_scan(p, c->stream));
    }
    return NEEDLE_HIP_OK;
}

// The bf16 operand copy is built on the first bf16 search of an index (callers hold ix->mu).
needle_hip_status nh::ensure_bf16_copy(needle_hip_index* ix) {
    if (ix->d_op_bf16) return NEEDLE_HIP_OK;
    needle_hip_ctx* c = ix->ctx;
    ix->dpad_bf16 = flat_scan_bf16_dpad(ix->dim);
    // (a flat index that has grown by appends keeps every resident array at its capacity, this copy included)
    const size_t bytes = (size_t)std::max(ix->n_pad, ix->cap_pad) * ix->dpad_bf16 * sizeof(uint16_t);
    uint16_t* d = nullptr;
    uint32_t* d_bits = nullptr;
    HIP_TRY(hipMalloc((void**)&d, bytes));
    hipError_t e = hipMalloc((void**)&d_bits, 2 * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemsetAsync(d_bits, 0, 2 * sizeof(uint32_t), c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(d, 0, bytes, c->stream);
    const bool sq8 = ix->kind == KIND_SQ8, pq = ix->pq_m != 0;  // PQ index, or IVF lists holding PQ codes
    if (e == hipSuccess && pq) {
        // the decoded rows are the operand: score = q.row - |row|^2/2, as a Euclidean flat index over them
        if (!ix->d_cinit) {
            e = hipMalloc((void**)&ix->d_cinit, (size_t)ix->n_pad * sizeof(float));
            if (e == hipSuccess) e = hipMemsetAsync(ix->d_cinit, 0, (size_t)ix->n_pad * sizeof(float), c->stream);
        }
        if (e == hipSuccess)
            e = launch_pq_decode_bf16(ix->d_codes, ix->d_codebooks, ix->kind == KIND_IVF ? ix->n_pad : ix->n, ix->pq_m, ix->pq_dsub, ix->dpad_bf16, d, ix->d_cinit,
                                      d_bits, c->stream);
        uint32_t two[2] = {0, 0};
        if (e == hipSuccess) e = hipMemcpyAsync(two, d_bits, sizeof(two), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (d_bits) (void)hipFree(d_bits);
        if (e != hipSuccess) { (void)hipFree(d); HIP_TRY(e); }
        memcpy(&ix->max_norm, &two[0], sizeof(float));
        memcpy(&ix->bf16_err, &two[1], sizeof(float));
        ix->d_op_bf16 = d;
        return NEEDLE_HIP_OK;
    }
    if (e == hipSuccess && sq8)  // codes are exact in bf16; ix->bf16_err already holds max |code vector|
        e = launch_sq8_codes_bf16(ix->d_codes, ix->n, ix->dim, ix->dpad_f32, ix->dpad_bf16, d, c->stream);
    else if (e == hipSuccess)
        e = launch_row_prep(ix->d_raw, ix->kind == KIND_IVF ? ix->n_pad : ix->n, ix->dim, ix->dpad_bf16,
                            ix->metric == NEEDLE_METRIC_COSINE, nullptr, d,
                            nullptr, nullptr, nullptr, nullptr, nullptr, d_bits, c->stream);
    uint32_t bits = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&bits, d_bits, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (d_bits) (void)hipFree(d_bits);
    if (e != hipSuccess) { (void)hipFree(d); HIP_TRY(e); }
    if (!sq8) memcpy(&ix->bf16_err, &bits, sizeof(float));
    ix->d_op_bf16 = d;
    return NEEDLE_HIP_OK;
}

// Device-resident search of one batch of at most QB queries.
needle_hip_status nh::flat_search_batch_device(needle_hip_index* ix, const float* d_queries, uint32_t q, uint32_t k,
                                               int precision, const uint32_t* d_filter, uint64_t* d_out_ids,
                                               float* d_out_dists, uint32_t* d_out_counts, bool set_only, bool defer_flagged) {
    needle_hip_ctx* c = ix->ctx;
    const uint32_t n = (uint32_t)ix->n;
    const bool sq8 = ix->kind == KIND_SQ8;
    // bf16 is a hint: code matrices and dimensions beyond the register-resident query block use fp32
    if (precision == NEEDLE_PRECISION_BF16 && flat_scan_bf16_dpad(ix->dim) == 0) precision = NEEDLE_PRECISION_FP32;
    if (precision == NEEDLE_PRECISION_BF16 && choose_kprime(std::min(k, n), precision) > KPRIME_MAX)
        precision = NEEDLE_PRECISION_FP32;
    uint32_t kprime = choose_kprime(std::min(k, n), precision);
    // bf16: an index whose earlier batches left many queries undecided keeps more candidates per query (see below)
    if (precision == NEEDLE_PRECISION_BF16) kprime = std::min<uint32_t>(round_up(kprime + ix->bf16_extra_slack, 16), KPRIME_MAX);
    // cosine rows with norms no first pass can bound ride along as extra candidates of every query, when they fit
    const bool unsafe_extra = ix->cos_norm_unsafe && ix->d_unsafe && ix->n_unsafe <= UNSAFE_ROWS_MAX &&
                              (uint64_t)kprime + ix->n_unsafe <= (uint32_t)KPRIME_MAX;
    const bool exact_only = precision == NEEDLE_PRECISION_EXACT_SCAN || (!sq8 && !metric_has_first_pass(ix->metric)) ||
                            kprime > KPRIME_MAX || (ix->cos_norm_unsafe && !unsafe_extra);
    if (exact_only) {
        if (defer_flagged) {  // nothing is left open: the caller will read a zero
            HIP_TRY(ix->w_flagcnt.ensure(sizeof(uint32_t)));
            HIP_TRY(hipMemsetAsync(ix->w_flagcnt.p, 0, sizeof(uint32_t), c->stream));
        }
        ST_TRY(run_exact(ix, d_queries, nullptr, q, k, d_filter, d_out_ids, d_out_dists, d_out_counts));
        HIP_TRY(hipStreamSynchronize(c->stream));
        std::lock_guard g(c->mu);
        c->m.fallback_queries += q;
        return NEEDLE_HIP_OK;
    }
    const bool bf16 = precision == NEEDLE_PRECISION_BF16;
    if (bf16) ST_TRY(ensure_bf16_copy(ix));

    const uint32_t dpad = bf16 ? ix->dpad_bf16 : ix->dpad_f32;
    const uint32_t q_block = flat_scan_queries_per_block(precision, dpad);
    const uint32_t q_pad = round_up(q, q_block), q_tiles = q_pad / q_block;
    const uint32_t n_tiles = ix->n_pad / TILE_ROWS;
    const uint32_t cap = list_cap(kprime);
    const uint32_t lists_per_split = bf16 ? flat_scan_bf16_lists_per_slice() : 1;  // bf16: one list per lane group
    // the merge gathers a query's lists in LDS: its one-shot path takes up to 1024 lists (what they actually hold is a
    // few hundred entries in all); beyond that it falls back to gathering in groups
    const uint32_t max_splits = std::max<uint32_t>(1, 1024u / lists_per_split);
    SplitPlan plan = plan_splits(q_tiles, n_tiles, (uint32_t)c->n_cu * (uint32_t)flat_scan_blocks_per_cu(precision), max_splits);
    const Tuning& tune = c->tune;
    if (tune.splits) {  // tuning override, applied before any sizing
        const uint32_t s = tune.splits;
        if (s >= 1 && s <= n_tiles && s <= 256 && s <= max_splits) {
            plan.tiles_per_split = (n_tiles + s - 1) / s;
            plan.n_splits = (n_tiles + plan.tiles_per_split - 1) / plan.tiles_per_split;
            plan.xcd_map = (plan.n_splits % 8 == 0) ? 1u : 0u;
        }
    }
    const size_t elt = bf16 ? 2 : 4;
    // Small bases (the centroid set of an IVF index, k-means assignment): with few rows per query the candidate
    // lists would hold a large part of the base.  The scan then writes the whole score matrix and a selection
    // kernel takes the kprime best of every row.
    const bool dense = bf16 && ix->n_pad <= DENSE_MAX_ROWS && !tune.no_dense;
    const uint32_t n_lists = dense ? 1 : plan.n_splits * lists_per_split;  // per query
    if (dense) HIP_TRY(ix->w_scores.ensure((size_t)q_pad * ix->n_pad * sizeof(float)));

    HIP_TRY(ix->w_qop.ensure((size_t)q_pad * dpad * elt));
    HIP_TRY(ix->w_qn2.ensure((size_t)q_pad * 2 * sizeof(double)));  // second half: SQ8 |q / scale|^2
    if (!dense) {
        HIP_TRY(ix->w_lists.ensure((size_t)q_pad * n_lists * cap * sizeof(uint2)));
        HIP_TRY(ix->w_lcnt.ensure((size_t)q_pad * n_lists * sizeof(uint32_t)));
    }
    HIP_TRY(ix->w_cand.ensure((size_t)q * kprime * sizeof(uint32_t)));
    HIP_TRY(ix->w_candn.ensure((size_t)q * sizeof(uint32_t)));
    HIP_TRY(ix->w_tau.ensure((size_t)q * sizeof(float)));
    HIP_TRY(ix->w_flagged.ensure((size_t)q * sizeof(uint32_t)));
    HIP_TRY(ix->w_flagcnt.ensure(sizeof(uint32_t)));
    if (bf16) HIP_TRY(ix->w_qerr.ensure((size_t)q_pad * sizeof(float)));

    HIP_TRY(hipMemsetAsync(ix->w_qop.p, 0, (size_t)q_pad * dpad * elt, c->stream));
    HIP_TRY(hipMemsetAsync(ix->w_flagcnt.p, 0, sizeof(uint32_t), c->stream));
    const bool normalize = ix->metric == NEEDLE_METRIC_COSINE;
    if (sq8)
        HIP_TRY(launch_sq8_query_prep(d_queries, q, ix->dim, dpad, ix->d_sq_min, ix->d_sq_scale,
                                      bf16 ? nullptr : ix->w_qop.as<float>(), bf16 ? ix->w_qop.as<uint16_t>() : nullptr,
                                      ix->w_qn2.as<double>(), ix->w_qn2.as<double>() + q_pad, c->stream));
    else
        HIP_TRY(launch_row_prep(d_queries, q, ix->dim, dpad, normalize,
                                precision == NEEDLE_PRECISION_BF16 ? nullptr : ix->w_qop.as<float>(),
                                precision == NEEDLE_PRECISION_BF16 ? ix->w_qop.as<uint16_t>() : nullptr, nullptr, nullptr,
                                ix->w_qn2.as<double>(), nullptr, bf16 ? ix->w_qerr.as<float>() : nullptr, nullptr, c->stream));

    ScanParams sp{};
    sp.base_op = precision == NEEDLE_PRECISION_BF16 ? (const void*)ix->d_op_bf16 : (const void*)ix->d_op_f32;
    if (sq8 && !bf16) { sp.base_op = ix->d_codes; sp.base_is_u8 = 1; }
    sp.query_op = ix->w_qop.p;
    sp.cinit = ix->d_cinit;
    sp.valid = ix->d_valid;
    sp.filter = d_filter;
    if (!dense && (ix->d_valid || d_filter)) {
        // Row masks are folded into the accumulator seeds: a dead row starts at -inf and can never pass a threshold, so
        // the scan does no mask lookup per listed row (a dependent load it cannot hide with one wave per SIMD) and its
        // wave-level early-out is not defeated by dead rows scoring above the threshold.
        HIP_TRY(ix->w_cinit.ensure((size_t)ix->n_pad * sizeof(float)));
        HIP_TRY(launch_masked_seed(ix->d_cinit, ix->d_valid, d_filter, n, ix->n_pad, ix->w_cinit.as<float>(), c->stream));
        sp.cinit = ix->w_cinit.as<float>();
        sp.valid = nullptr;
        sp.filter = nullptr;
    }
    sp.lists = ix->w_lists.as<uint2>();
    sp.list_counts = ix->w_lcnt.as<uint32_t>();
    sp.n = n;
    sp.n_tiles = n_tiles;
    sp.tiles_per_split = plan.tiles_per_split;
    sp.n_splits = plan.n_splits;
    sp.q_tiles = q_tiles;
    sp.n_queries = q;
    sp.dpad = dpad;
    sp.kprime = kprime;
    sp.cap = cap;
    sp.xcd_map = plan.xcd_map;
    sp.debug = tune.debug;
    // threshold seeding: worth it once a slice is much longer than the sample
    // The seeded threshold is the kseed-th largest sampled group maximum: at least kseed >= k live rows
    // exceed it, and the higher it ranks the fewer rows the main pass lists (rank k against k + 6: 3 % of the
    // 1M x 384 search).  The bound check decides afterwards whether the candidates it leaves are enough.
    sp.kseed = std::min<uint32_t>(kprime, std::min(k, n));
    // sample: 8 tiles per slice; 16 when many hits are wanted -- the k-th best of a sample sits deeper in it the larger k
    // is, and what the loose threshold lets through is paid again by the merge and the re-score (1M x 768, 10 % filter,
    // k = 100: 2.02 -> 1.86 ms per 1K queries; 24 tiles no better)
    uint32_t pre_tiles = std::min<uint32_t>(sp.kseed > 48 ? 16 : 8, plan.tiles_per_split / 8);
    if (tune.pre_tiles >= 0) pre_tiles = std::min<uint32_t>((uint32_t)tune.pre_tiles, plan.tiles_per_split);
    if (tune.kseed) sp.kseed = std::min<uint32_t>(std::max<uint32_t>(tune.kseed, std::min(k, n)), kprime);
    const bool seeded = pre_tiles >= 1 && !(sp.debug & 2u) && !dense;
    TauBand band{};
    band.qn2 = ix->w_qn2.as<double>();
    band.qs2 = (sq8 && bf16) ? ix->w_qn2.as<double>() + q_pad : nullptr;
    band.qerr = (bf16 && !sq8) ? ix->w_qerr.as<float>() : nullptr;
    band.berr = ix->bf16_err;
    band.bmax = ix->max_norm;
    band.mnorm = ix->sq_min_norm * (1.0 + 1e-6);
    band.dim = ix->dim; band.metric = ix->metric; band.precision = precision; band.sq8 = sq8 ? 1 : 0;
    if (tune.band > 0.0f) band.width = tune.band;
    if (c->profiling) HIP_TRY(hipEventRecord(c->ev[0], c->stream));
    if (seeded) {
        HIP_TRY(ix->w_premax.ensure((size_t)q_pad * 8 * plan.n_splits * sizeof(float)));
        HIP_TRY(ix->w_tau0.ensure((size_t)q_pad * sizeof(float)));
        sp.pre_max = ix->w_premax.as<float>();
        sp.pre_tiles = pre_tiles;
        sp.pre_stride = std::max<uint32_t>(1, plan.tiles_per_split / pre_tiles);
        HIP_TRY(launch_flat_prepass(precision, sp, ix->w_tau0.as<float>(), band, c->stream));
        sp.tau0 = ix->w_tau0.as<float>();
    }
    sp.tile_off = 0;
    sp.tile_len = plan.tiles_per_split;
    sp.resume = 0;
    // Long slices are scanned in three or four launches of growing length: the thresholds of each are refreshed from
    // everything listed before it, so that the rows listed per query stay near kseed * ln(rows / sample).
    constexpr int MAX_STAGES = 5;
    uint32_t stages[MAX_STAGES] = {plan.tiles_per_split, 0, 0, 0, 0};
    if (bf16 && seeded && plan.tiles_per_split >= 96 && !(sp.debug & 4u)) {
        if (plan.tiles_per_split <= 512) {
            stages[0] = std::max<uint32_t>(3 * pre_tiles, plan.tiles_per_split / 10);
            stages[1] = plan.tiles_per_split / 4;
            stages[2] = plan.tiles_per_split - stages[0] - stages[1];
        } else {
            // Long slices (10M rows: 2441 tiles): a launch lists about k * (its rows / the rows seen before it) rows per
            // query, so the first launch must stay small beside the sample (10 % of such a slice is 30 times the sample:
            // 300 rows listed per query before the first refresh); 3 / 7 / 20 / 70 % measured 6 % faster than 10 / 25 / 65.
            stages[0] = std::max<uint32_t>(3 * pre_tiles, plan.tiles_per_split * 3 / 100);
            stages[1] = plan.tiles_per_split * 7 / 100;
            stages[2] = plan.tiles_per_split / 5;
            stages[3] = plan.tiles_per_split - stages[0] - stages[1] - stages[2];
        }
        if (tune.n_stage_pct) {  // tuning: percentages of a slice per launch, the last takes the rest
            uint32_t used = 0;
            int r = 0;
            for (; r < (int)tune.n_stage_pct && r < MAX_STAGES - 1; ++r) {
                stages[r] = std::min<uint32_t>(std::max<uint32_t>(1, plan.tiles_per_split * tune.stage_pct[r] / 100), plan.tiles_per_split - used - 1);
                used += stages[r];
            }
            stages[r++] = plan.tiles_per_split - used;
            for (; r < MAX_STAGES; ++r) stages[r] = 0;
        }
    }
    if (dense) {
        sp.dense_scores = ix->w_scores.as<float>();
        sp.dense_ld = ix->n_pad;
        HIP_TRY(launch_dense_scores_bf16(sp, c->stream));
        stages[0] = 0;
    }
    for (int r = 0; r < MAX_STAGES && stages[r] > 0; ++r) {
        if (r > 0) {
            HIP_TRY(launch_tau_refresh(sp, n_lists, ix->w_tau0.as<float>(), band, c->stream));
            sp.resume = 1;
        }
        sp.tile_len = stages[r];
        HIP_TRY(launch_flat_scan(precision, sp, c->stream));
        sp.tile_off += stages[r];
    }
    if (c->profiling) HIP_TRY(hipEventRecord(c->ev[1], c->stream));
    if (sp.debug & 64u) {  // development, with a BF_STAMP build: in-kernel clock of the last launch
        const uint32_t nb = plan.n_splits * q_tiles;
        std::vector<uint64_t> st((size_t)nb * 2);
        HIP_TRY(hipMemcpyAsync(st.data(), ix->w_premax.p, st.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        std::vector<double> ghz;
        double cyc = 0;
        for (uint32_t b = 0; b < nb; ++b)
            if (st[2 * b + 1]) { ghz.push_back((double)st[2 * b] / (double)st[2 * b + 1] * 0.1); cyc += (double)st[2 * b]; }
        std::sort(ghz.begin(), ghz.end());
        if (!ghz.empty())
            fprintf(stderr, "[needle_hip] in-kernel clock: median %.3f GHz (min %.3f, max %.3f), mean block cycles %.0f\n",
                    ghz[ghz.size() / 2], ghz.front(), ghz.back(), cyc / ghz.size());
    }
    if ((sp.debug & 8u) && !dense) {  // development: how many candidates did the first pass list?
        std::vector<uint32_t> lc((size_t)q_pad * n_lists);
        HIP_TRY(hipMemcpyAsync(lc.data(), sp.list_counts, lc.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        uint64_t tot = 0; uint32_t mx = 0;
        for (size_t i = 0; i < (size_t)q * n_lists; ++i) { tot += lc[i]; mx = std::max(mx, lc[i]); }
        float dbg_ms = -1.0f;
        if (c->profiling) (void)hipEventElapsedTime(&dbg_ms, c->ev[0], c->ev[1]);
        fprintf(stderr, "[needle_hip] q %u: listed candidates %.1f per query, longest list %u, lists per query %u, kprime %u, kseed %u, "
                "pre_tiles %u, slices %u x %u tiles, query blocks %u, first pass %s %.3f ms\n",
                q, (double)tot / std::max<uint32_t>(q, 1), mx, n_lists, kprime, sp.kseed, seeded ? pre_tiles : 0, plan.n_splits,
                plan.tiles_per_split, q_tiles, bf16 ? "bf16" : "fp32", dbg_ms);
    }

    PostParams pp{};
    pp.base_raw = ix->d_raw;
    pp.query_raw = d_queries;
    pp.qnorm2 = ix->w_qn2.as<double>();
    pp.lists = sp.lists;
    pp.list_counts = sp.list_counts;
    pp.cand_ids = ix->w_cand.as<uint32_t>();
    pp.cand_n = ix->w_candn.as<uint32_t>();
    pp.tau_star = ix->w_tau.as<float>();
    pp.tau0 = sp.tau0;
    pp.flagged = ix->w_flagged.as<uint32_t>();
    pp.flag_count = ix->w_flagcnt.as<uint32_t>();
    pp.out_ids = d_out_ids;
    pp.out_dists = d_out_dists;
    pp.out_counts = d_out_counts;
    pp.q = q; pp.n = n; pp.dim = ix->dim; pp.k = k; pp.kprime = kprime; pp.cap = cap; pp.n_splits = n_lists;
    pp.metric = ix->metric; pp.arith = ix->arith; pp.precision = precision;
    pp.max_norm = ix->max_norm;
    pp.qerr = (bf16 && !sq8) ? ix->w_qerr.as<float>() : nullptr;
    pp.berr = ix->bf16_err;
    if (unsafe_extra) {
        HIP_TRY(ix->w_extra.ensure((UNSAFE_ROWS_MAX + 1) * sizeof(uint32_t)));
        uint32_t* ex = ix->w_extra.as<uint32_t>();
        HIP_TRY(launch_unsafe_live(ix->d_unsafe, ix->n_unsafe, ix->d_valid, d_filter, ex, ex + UNSAFE_ROWS_MAX, c->stream));
        pp.extra_ids = ex;
        pp.extra_n = ex + UNSAFE_ROWS_MAX;
    }
    pp.tune_merge_keys = tune.merge_keys; pp.tune_merge_lists = tune.merge_lists; pp.tune_rs_cap = tune.rs_cap; pp.tune_rs_dc = tune.rs_dc;
    if (ix->kind == KIND_PQ) {  // candidates are re-scored with the distance-table arithmetic
        pp.pq_codes = ix->d_codes; pp.pq_books = ix->d_codebooks; pp.pq_m = ix->pq_m; pp.pq_dsub = ix->pq_dsub;
    }
    if (sq8) {
        pp.sq_codes = ix->d_codes; pp.sq_min = ix->d_sq_min; pp.sq_scale = ix->d_sq_scale; pp.sq_stride = ix->dpad_f32;
        pp.aux_norm = (float)(ix->sq_min_norm * (1.0 + 1e-6));
        if (bf16) pp.qaux = ix->w_qn2.as<double>() + q_pad;
    }
    if (g_trace)
        fprintf(stderr, "[needle_hip] view %p: q %u kprime %u slices %u x %u tiles, %s, %s; base %p qop %p lists %p counts %p cand %p "
                "queries %p out %p\n", (void*)ix, q, kprime, plan.n_splits, plan.tiles_per_split, seeded ? "seeded" : "unseeded",
                bf16 ? "bf16" : "fp32", sp.base_op, sp.query_op, (void*)sp.lists, (void*)sp.list_counts, ix->w_cand.p,
                (const void*)d_queries, (void*)d_out_ids);
    const bool prune = !(sp.debug & 16u);
    if (prune) {
        HIP_TRY(ix->w_cscore.ensure((size_t)q * kprime * sizeof(float)));
        pp.cand_scores = ix->w_cscore.as<float>();
    }
    if (dense) {
        pp.dense_scores = sp.dense_scores;
        pp.dense_ld = sp.dense_ld;
        pp.valid = ix->d_valid;
        pp.filter = d_filter;
        HIP_TRY(launch_dense_select(pp, c->stream));
    } else {
        HIP_TRY(launch_merge_select(pp, c->stream));
    }
    // set_only (the probe step of an inverted-file search): the caller needs the k nearest rows as a set, so only the
    // candidates whose first-pass scores leave their membership open are re-scored
    const bool set_mode = set_only && dense && prune && bf16 && !sq8 && ix->kind == KIND_FLAT && ix->metric == NEEDLE_METRIC_EUCLIDEAN &&
                          !tune.no_set_only && ix->arith != NEEDLE_ARITH_AVX2 && ix->arith != NEEDLE_ARITH_AVX512 && ix->arith != NEEDLE_ARITH_NEON;
    if (prune && !set_mode) HIP_TRY(launch_prune_candidates(pp, band, c->stream));
    HIP_TRY(launch_rescore_finalize(pp, c->stream, set_mode ? &band : nullptr));
    if (defer_flagged) return NEEDLE_HIP_OK;  // (this pass's device time is then not added to the metrics)

    uint32_t n_flagged = 0;
    HIP_TRY(hipMemcpyAsync(&n_flagged, pp.flag_count, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->profiling) {
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
        std::lock_guard g(c->mu);
        c->m.scan_launches += 1;
        c->m.scan_ms_total += ms;
        c->m.last_scan_ms = ms;
    }
    // The rows a bf16 score cannot tell apart from the k-th hit must all be among the candidates kept.  How many
    // there are depends on the data (wide, unnormalised rows at large k: dozens), so the slack adapts: a batch that
    // left more than 0.5 % of its queries undecided doubles it for the batches that follow.  Only the cost moves; the
    // hits are proven either way.
    if (bf16 && q >= 32 && (uint64_t)n_flagged * 200 > q && kprime < (uint32_t)KPRIME_MAX)
        ix->bf16_extra_slack = std::min<uint32_t>(ix->bf16_extra_slack ? ix->bf16_extra_slack * 2 : 48, KPRIME_MAX);
    if (n_flagged > 0) {
        if (n_flagged > q) n_flagged = q;
        // The exact scan streams the base once per 4 queries: cheaper than another first pass for a handful of
        // queries, or for any number of them over a small base (an IVF centroid set).
        const bool few = n_flagged <= 16 || (uint64_t)n_flagged * ix->n_pad <= (4ull << 20);
        if (ix->kind == KIND_PQ) {
            // product-quantized rows: the undecided queries take the distance-table scan, which is exact by itself
            HIP_TRY(ix->r_idx.ensure((size_t)n_flagged * sizeof(uint32_t)));
            HIP_TRY(ix->r_q.ensure((size_t)n_flagged * ix->dim * sizeof(float)));
            HIP_TRY(ix->r_ids.ensure((size_t)n_flagged * k * sizeof(uint64_t)));
            HIP_TRY(ix->r_dists.ensure((size_t)n_flagged * k * sizeof(float)));
            HIP_TRY(ix->r_cnt.ensure((size_t)n_flagged * sizeof(uint32_t)));
            HIP_TRY(hipMemcpyAsync(ix->r_idx.p, pp.flagged, (size_t)n_flagged * sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
            HIP_TRY(launch_gather_rows(d_queries, ix->r_idx.as<uint32_t>(), n_flagged, ix->dim, ix->r_q.as<float>(), c->stream));
            ST_TRY(pq_search_batch_device(ix, ix->r_q.as<float>(), n_flagged, k, d_filter, ix->r_ids.as<uint64_t>(),
                                          ix->r_dists.as<float>(), ix->r_cnt.as<uint32_t>()));
            HIP_TRY(launch_scatter_results(ix->r_idx.as<uint32_t>(), n_flagged, k, ix->r_ids.as<uint64_t>(), ix->r_dists.as<float>(),
                                           ix->r_cnt.as<uint32_t>(), d_out_ids, d_out_dists, d_out_counts, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            std::lock_guard g(c->mu);
            c->m.fallback_queries += n_flagged;
            return NEEDLE_HIP_OK;
        }
        if (bf16 && !few) {
            // The bf16 scores could not separate these queries' k-th hit from the rows left out: give them
            // the fp32 first pass (whose own leftovers go to the exact scan) and put the answers in place.
            HIP_TRY(ix->r_idx.ensure((size_t)n_flagged * sizeof(uint32_t)));
            HIP_TRY(ix->r_q.ensure((size_t)n_flagged * ix->dim * sizeof(float)));
            HIP_TRY(ix->r_ids.ensure((size_t)n_flagged * k * sizeof(uint64_t)));
            HIP_TRY(ix->r_dists.ensure((size_t)n_flagged * k * sizeof(float)));
            HIP_TRY(ix->r_cnt.ensure((size_t)n_flagged * sizeof(uint32_t)));
            HIP_TRY(hipMemcpyAsync(ix->r_idx.p, pp.flagged, (size_t)n_flagged * sizeof(uint32_t), hipMemcpyDeviceToDevice,
                                   c->stream));
            HIP_TRY(launch_gather_rows(d_queries, ix->r_idx.as<uint32_t>(), n_flagged, ix->dim, ix->r_q.as<float>(), c->stream));
            ST_TRY(flat_search_batch_device(ix, ix->r_q.as<float>(), n_flagged, k, NEEDLE_PRECISION_FP32, d_filter,
                                            ix->r_ids.as<uint64_t>(), ix->r_dists.as<float>(), ix->r_cnt.as<uint32_t>()));
            HIP_TRY(launch_scatter_results(ix->r_idx.as<uint32_t>(), n_flagged, k, ix->r_ids.as<uint64_t>(),
                                           ix->r_dists.as<float>(), ix->r_cnt.as<uint32_t>(), d_out_ids, d_out_dists,
                                           d_out_counts, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            std::lock_guard g(c->mu);
            c->m.refined_queries += n_flagged;
            return NEEDLE_HIP_OK;
        }
        ST_TRY(run_exact(ix, d_queries, pp.flagged, n_flagged, k, d_filter, d_out_ids, d_out_dists, d_out_counts));
        HIP_TRY(hipStreamSynchronize(c->stream));
        std::lock_guard g(c->mu);
        c->m.fallback_queries += n_flagged;
    }
    return NEEDLE_HIP_OK;
}

// A filtered search of a flat index whose masks (deleted rows and the caller's filter together) admit at most half of the
// rows does not scan the base: the admitted rows are counted (one small kernel and a 4-byte read-back), their numbers
// listed in ascending order, and their raw rows, operands and accumulator seeds gathered into the slot's `compact`
// index -- a flat index without masks, whose row r is row z_pos[r] of the base.  Position order is row order, so ties
// still go to the lower row number; the maxima the bounds use (largest norm, largest rounding error) are the base's and
// cover its subset.  *out stays null when the search should run over the base itself.
static needle_hip_status compact_admitted_rows(needle_hip_index* ix, const uint32_t* d_filter, int precision, needle_hip_index** out,
                                               uint32_t* out_live) {
    *out = nullptr;
    needle_hip_ctx* c = ix->ctx;
    SearchSlot* s = ix->slot;
    const Tuning& tune = c->tune;
    const uint32_t n = (uint32_t)ix->n;
    const uint32_t min_rows = tune.compact_min_rows ? tune.compact_min_rows : 65536u;
    if (!s || !d_filter || ix->kind != KIND_FLAT || tune.no_compact || ix->n < min_rows || ix->n > 0xfffffff0ull ||
        ix->cos_norm_unsafe || !metric_has_first_pass(ix->metric) || precision == NEEDLE_PRECISION_EXACT_SCAN)
        return NEEDLE_HIP_OK;
    const bool want_bf16 = precision == NEEDLE_PRECISION_BF16 && flat_scan_bf16_dpad(ix->dim) != 0;
    if (want_bf16 && !ix->d_op_bf16) return NEEDLE_HIP_OK;  // (the search will not use the bf16 route either)
    const uint32_t blocks = mask_blocks(n);
    HIP_TRY(s->z_cnt.ensure(((size_t)blocks + 1) * sizeof(uint32_t)));
    uint32_t* d_cnt = s->z_cnt.as<uint32_t>();  // [blocks] admitted rows per block of 8192, then their total
    HIP_TRY(launch_mask_count(ix->d_valid, d_filter, n, d_cnt, d_cnt + blocks, c->stream));
    uint32_t live = 0;
    HIP_TRY(hipMemcpyAsync(&live, d_cnt + blocks, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    const uint32_t pct = tune.compact_pct ? std::min<uint32_t>(tune.compact_pct, 100u) : 50u;
    if (live == 0 || live > n || (uint64_t)live * 100u > (uint64_t)n * pct) return NEEDLE_HIP_OK;
    const uint32_t n_pad = round_up(live, TILE_ROWS);
    HIP_TRY(s->z_pos.ensure((size_t)n_pad * sizeof(uint32_t)));
    uint32_t* d_pos = s->z_pos.as<uint32_t>();
    HIP_TRY(launch_mask_positions(ix->d_valid, d_filter, n, d_cnt, live, d_pos, c->stream));
    const size_t raw_bytes = (size_t)ix->dim * sizeof(float);
    HIP_TRY(s->z_raw.ensure((size_t)n_pad * raw_bytes));
    HIP_TRY(launch_gather_row_bytes(ix->d_raw, d_pos, live, n_pad, n, raw_bytes, s->z_raw.p, c->stream));
    needle_hip_index& x = s->compact;
    x.ctx = c; x.is_view = true; x.slot = nullptr; x.kind = KIND_FLAT;
    x.n = live; x.n_pad = n_pad; x.cap_pad = n_pad; x.dim = ix->dim; x.metric = ix->metric; x.arith = ix->arith;
    x.d_raw = s->z_raw.as<float>();
    x.dpad_f32 = ix->dpad_f32; x.dpad_bf16 = ix->dpad_bf16; x.op_f32_alias = ix->op_f32_alias;
    if (ix->op_f32_alias) {
        x.d_op_f32 = x.d_raw;
    } else {
        const size_t b = (size_t)ix->dpad_f32 * sizeof(float);
        HIP_TRY(s->z_op32.ensure((size_t)n_pad * b));
        HIP_TRY(launch_gather_row_bytes(ix->d_op_f32, d_pos, live, n_pad, n, b, s->z_op32.p, c->stream));
        x.d_op_f32 = s->z_op32.as<float>();
    }
    x.d_op_bf16 = nullptr;
    if (want_bf16) {
        const size_t b = (size_t)ix->dpad_bf16 * sizeof(uint16_t);
        HIP_TRY(s->z_op16.ensure((size_t)n_pad * b));
        HIP_TRY(launch_gather_row_bytes(ix->d_op_bf16, d_pos, live, n_pad, n, b, s->z_op16.p, c->stream));
        x.d_op_bf16 = s->z_op16.as<uint16_t>();
    }
    x.d_cinit = nullptr;
    if (ix->d_cinit) {
        HIP_TRY(s->z_cinit.ensure((size_t)n_pad * sizeof(float)));
        HIP_TRY(launch_gather_row_bytes(ix->d_cinit, d_pos, live, n_pad, n, sizeof(float), s->z_cinit.p, c->stream));
        x.d_cinit = s->z_cinit.as<float>();
    }
    x.d_norms = nullptr; x.d_valid = nullptr; x.d_unsafe = nullptr; x.n_unsafe = 0; x.cos_norm_unsafe = false;
    x.max_norm = ix->max_norm; x.min_norm = ix->min_norm; x.bf16_err = ix->bf16_err;
    x.bf16_extra_slack = std::max(x.bf16_extra_slack, ix->bf16_extra_slack);
    *out = &x;
    *out_live = live;
    return NEEDLE_HIP_OK;
}

static needle_hip_status search_device_locked(needle_hip_index* ix, const float* d_queries, uint32_t q,
                                              const needle_search_params* params, const uint32_t* d_filter,
                                              uint64_t* d_out_ids, float* d_out_dists, uint32_t* d_out_counts) {
    needle_hip_ctx* c = ix->ctx;
    const uint32_t k = params->k;
    const int precision = params->precision;
    if (precision < NEEDLE_PRECISION_FP32 || precision > NEEDLE_PRECISION_EXACT_SCAN)
        return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "unknown precision %d", precision);
    if (q == 0) return NEEDLE_HIP_OK;
    HIP_TRY(hipSetDevice(c->device));
    // the timing events belong to the context: with profiling on, searches of ALL its indexes run one at a time
    std::unique_lock<std::mutex> profiled(c->prof_mu, std::defer_lock);
    if (c->profiling) profiled.lock();
    if (c->profiling) HIP_TRY(hipEventRecord(c->ev[2], c->stream));
    if (k == 0 || ix->n == 0) {
        // clamp_k == 0 -> vec![Vec::new(); queries.len()], src/collection/batch.rs:40-43
        HIP_TRY(launch_fill_empty(d_out_ids, d_out_dists, d_out_counts, q, k, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    } else {
        needle_hip_index* base_ix = ix;
        needle_hip_index* compact = nullptr;
        uint32_t compact_live = 0;
        ST_TRY(compact_admitted_rows(ix, d_filter, precision, &compact, &compact_live));
        if (compact) {  // the admitted rows as an index of their own: no masks left
            ix = compact;
            d_filter = nullptr;
        }
        const uint32_t QB = 16384;
        for (uint32_t q0 = 0; q0 < q; q0 += QB) {
            const uint32_t qb = std::min(QB, q - q0);
            // product-quantized rows: precision bf16 scores the decoded rows on the matrix cores and re-scores the
            // candidates with the distance-table arithmetic; otherwise the table scan itself
            const bool pq_mfma = ix->kind == KIND_PQ && precision == NEEDLE_PRECISION_BF16 && flat_scan_bf16_dpad(ix->dim) != 0 &&
                                 choose_kprime((uint32_t)std::min<uint64_t>(k, ix->n), NEEDLE_PRECISION_BF16) <= (uint32_t)KPRIME_MAX;
            if (ix->kind == KIND_PQ && !pq_mfma)
                ST_TRY(pq_search_batch_device(ix, d_queries + (size_t)q0 * ix->dim, qb, k, d_filter,
                                              d_out_ids + (size_t)q0 * k, d_out_dists + (size_t)q0 * k,
                                              d_out_counts + q0));
            else if (ix->kind == KIND_BINARY)
                ST_TRY(binary_search_batch_device(ix, d_queries + (size_t)q0 * ix->dim, qb, k, d_filter,
                                                  d_out_ids + (size_t)q0 * k, d_out_dists + (size_t)q0 * k,
                                                  d_out_counts + q0));
            else if (ix->kind == KIND_IVF)
                ST_TRY(ivf_search_batch_device(ix, d_queries + (size_t)q0 * ix->dim, qb, k, params->nprobe, precision,
                                               d_filter, d_out_ids + (size_t)q0 * k, d_out_dists + (size_t)q0 * k,
                                               d_out_counts + q0));
            else
                ST_TRY(flat_search_batch_device(ix, d_queries + (size_t)q0 * ix->dim, qb, k, precision, d_filter,
                                                d_out_ids + (size_t)q0 * k, d_out_dists + (size_t)q0 * k,
                                                d_out_counts + q0));
        }
        if (compact) {
            HIP_TRY(launch_map_ids(d_out_ids, (uint64_t)q * k, base_ix->slot->z_pos.as<uint32_t>(), compact_live, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            std::lock_guard g(c->mu);
            c->m.compacted_searches += 1;
            c->m.compacted_rows += compact_live;
        }
        ix = base_ix;
    }
    if (c->profiling) {
        HIP_TRY(hipEventRecord(c->ev[3], c->stream));
        HIP_TRY(hipEventSynchronize(c->ev[3]));
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, c->ev[2], c->ev[3]));
        std::lock_guard g(c->mu);
        c->m.last_search_ms = ms;
    }
    std::lock_guard g(c->mu);
    c->m.searches += 1;
    c->m.queries += q;
    return NEEDLE_HIP_OK;
}

extern "C" needle_hip_status needle_hip_batch_search_device(needle_hip_index* ix, const float* d_queries, uint32_t q,
                                                            const needle_search_params* params, uint64_t* d_out_ids,
                                                            float* d_out_dists, uint32_t* d_out_counts) {
    if (!ix || !params) return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "null argument");
    if (q > 0 && (!d_queries || !d_out_counts || (params->k > 0 && (!d_out_ids || !d_out_dists))))
        return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "null query or output pointer");
    if (q == 0) return NEEDLE_HIP_OK;
    SlotLease lease(ix, params);  // re-entrant: concurrent searches of one index run on slots of their own
    ST_TRY(lease.status());
    return search_device_locked(lease.view(), d_queries, q, params, params->filter_bitmask, d_out_ids, d_out_dists, d_out_counts);
}

extern "C" needle_hip_status needle_hip_batch_search(needle_hip_index* ix, const float* queries, uint32_t q,
                                                     uint32_t query_dim, const needle_search_params* params,
                                                     uint64_t* out_ids, float* out_dists, uint32_t* out_counts) {
    if (!ix || !params) return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "null argument");
    // validate_query for EVERY query before any work, src/collection/batch.rs:37-39
    if (q > 0 && query_dim != ix->dim)
        return fail(NEEDLE_HIP_ERR_DIMENSION_MISMATCH, "Dimension mismatch: expected %u, got %u", ix->dim, query_dim);
    if (q > 0 && (!queries || !out_counts || (params->k > 0 && (!out_ids || !out_dists))))
        return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "null query or output pointer");
    if (q == 0) return NEEDLE_HIP_OK;
    SlotLease lease(ix, params);
    ST_TRY(lease.status());
    ix = lease.view();
    needle_hip_ctx* c = ix->ctx;
    HIP_TRY(hipSetDevice(c->device));
    const uint32_t k = params->k;
    const size_t kk = k ? k : 1;
    HIP_TRY(ix->w_qraw.ensure((size_t)q * ix->dim * sizeof(float) + (size_t)q * kk * (sizeof(uint64_t) + sizeof(float)) +
                              (size_t)q * sizeof(uint32_t) + 64));
    // carve: queries | ids | dists | counts   (ids first after queries keeps 8-byte alignment)
    unsigned char* base = ix->w_qraw.as<unsigned char>();
    size_t off = ((size_t)q * ix->dim * sizeof(float) + 15) & ~(size_t)15;
    float* d_q = (float*)base;
    uint64_t* d_ids = (uint64_t*)(base + off);
    off += (size_t)q * kk * sizeof(uint64_t);
    float* d_d = (float*)(base + off);
    off += (size_t)q * kk * sizeof(float);
    uint32_t* d_c = (uint32_t*)(base + off);
    HIP_TRY(hipMemcpyAsync(d_q, queries, (size_t)q * ix->dim * sizeof(float), hipMemcpyHostToDevice, c->stream));
    {
        // NaN / Inf in ANY query fails the whole batch before any search.  The uploaded copy is checked on the device (the
        // first offending query); the host then reads only that query to word the error as the reference does.
        HIP_TRY(ix->w_flagcnt.ensure(sizeof(uint32_t)));
        HIP_TRY(launch_validate_rows(d_q, q, ix->dim, ix->w_flagcnt.as<uint32_t>(), c->stream));
        uint32_t bad = 0xffffffffu;
        HIP_TRY(hipMemcpyAsync(&bad, ix->w_flagcnt.p, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (bad != 0xffffffffu && bad < q) {
            const float* v = queries + (size_t)bad * query_dim;
            for (uint32_t j = 0; j < query_dim; ++j) {
                if (std::isnan(v[j]))
                    return fail(NEEDLE_HIP_ERR_INVALID_VECTOR, "Invalid vector: Vector contains NaN at index %u (query %u)", j, bad);
                if (std::isinf(v[j]))
                    return fail(NEEDLE_HIP_ERR_INVALID_VECTOR, "Invalid vector: Vector contains Inf at index %u (query %u)", j, bad);
            }
            return fail(NEEDLE_HIP_ERR_INVALID_VECTOR, "Invalid vector: Vector contains NaN or Inf (query %u)", bad);
        }
    }
    const uint32_t* d_filter = nullptr;
    if (params->filter_bitmask && ix->n > 0) {
        const size_t words = (ix->n + 31) / 32;
        HIP_TRY(ix->w_filter.ensure(words * sizeof(uint32_t)));
        HIP_TRY(hipMemcpyAsync(ix->w_filter.p, params->filter_bitmask, words * sizeof(uint32_t), hipMemcpyHostToDevice,
                               c->stream));
        d_filter = ix->w_filter.as<uint32_t>();
    }
    ST_TRY(search_device_locked(ix, d_q, q, params, d_filter, d_ids, d_d, d_c));
    if (k > 0) {
        HIP_TRY(hipMemcpyAsync(out_ids, d_ids, (size_t)q * k * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(out_dists, d_d, (size_t)q * k * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    }
    HIP_TRY(hipMemcpyAsync(out_counts, d_c, (size_t)q * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return NEEDLE_HIP_OK;
}

// ---------------------------------------------------------------------------
// operator-level calls
// ---------------------------------------------------------------------------
extern "C" needle_hip_status needle_hip_batch_distance(needle_hip_ctx* c, needle_metric metric, needle_arith arith,
                                                       const float* query, const float* base, uint64_t n,
                                                       uint32_t dim, float* out) {
    if (!c) return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "ctx is null");
    if (metric != NEEDLE_METRIC_COSINE && metric != NEEDLE_METRIC_EUCLIDEAN && metric != NEEDLE_METRIC_DOT_PRODUCT)
        return fail(NEEDLE_HIP_ERR_UNSUPPORTED, "batch distance covers cosine, euclidean and dot product");
    if (n == 0) return NEEDLE_HIP_OK;  // vectors.is_empty() -> Ok(vec![]), gpu/mod.rs:315-317
    if (!query || !base || !out || dim == 0) return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "null argument");
    std::lock_guard g(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(c->scratch[0].ensure((size_t)dim * sizeof(float)));
    HIP_TRY(c->scratch[1].ensure((size_t)n * dim * sizeof(float)));
    HIP_TRY(c->scratch[2].ensure((size_t)n * sizeof(float)));
    HIP_TRY(hipMemcpyAsync(c->scratch[0].p, query, (size_t)dim * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->scratch[1].p, base, (size_t)n * dim * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(launch_batch_distance(c->scratch[0].as<float>(), c->scratch[1].as<float>(), n, dim, metric, arith,
                                  c->scratch[2].as<float>(), c->stream));
    HIP_TRY(hipMemcpyAsync(out, c->scratch[2].p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return NEEDLE_HIP_OK;
}

extern "C" needle_hip_status needle_hip_kmeans_assign(needle_hip_ctx* c, needle_arith arith, const float* vectors,
                                                      uint64_t n, const float* centroids, uint32_t n_centroids,
                                                      uint32_t dim, uint32_t* out_assignment) {
    if (!c) return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "ctx is null");
    if (n == 0) return NEEDLE_HIP_OK;
    if (!vectors || !centroids || !out_assignment || n_centroids == 0 || dim == 0)
        return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "null or empty argument");
    if (n > 0xffffffffull) return fail(NEEDLE_HIP_ERR_UNSUPPORTED, "too many vectors");
    // nearest centroid == 1-NN search of the centroid set, ties to the lower index
    needle_hip_index* ix = nullptr;
    ST_TRY(needle_hip_flat_build(c, centroids, n_centroids, dim, NEEDLE_METRIC_EUCLIDEAN, arith, nullptr, &ix));
    needle_search_params sp{};
    sp.k = 1;
    sp.precision = NEEDLE_PRECISION_FP32;
    std::vector<uint64_t> ids;
    std::vector<float> d;
    std::vector<uint32_t> cnt;
    needle_hip_status st = NEEDLE_HIP_OK;
    const uint64_t B = 1u << 18;
    try {
        ids.resize(std::min<uint64_t>(B, n)); d.resize(ids.size()); cnt.resize(ids.size());
    } catch (...) {
        needle_hip_index_destroy(ix);
        return fail(NEEDLE_HIP_ERR_OUT_OF_MEMORY, "host allocation failed");
    }
    for (uint64_t i0 = 0; i0 < n && st == NEEDLE_HIP_OK; i0 += B) {
        const uint32_t b = (uint32_t)std::min<uint64_t>(B, n - i0);
        // rows were validated by the caller's insert path; NaN rows would be rejected here too
        st = needle_hip_batch_search(ix, vectors + (size_t)i0 * dim, b, dim, &sp, ids.data(), d.data(), cnt.data());
        if (st == NEEDLE_HIP_OK)
            for (uint32_t i = 0; i < b; ++i) out_assignment[i0 + i] = cnt[i] ? (uint32_t)ids[i] : 0u;
    }
    needle_hip_index_destroy(ix);
    return st;
}

extern "C" needle_hip_status needle_hip_merge_topk_device(needle_hip_ctx* c, const float* d_dists, const uint64_t* d_ids,
                                                          const uint32_t* d_counts, uint32_t lists, uint32_t q,
                                                          uint32_t per, uint32_t k, uint64_t* d_out_ids,
                                                          float* d_out_dists, uint32_t* d_out_counts) {
    if (!c) return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "ctx is null");
    if (q == 0) return NEEDLE_HIP_OK;
    if (!d_counts || !d_out_counts || lists == 0) return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "null argument");
    if ((uint64_t)lists * per * 16ull > 60000ull)
        return fail(NEEDLE_HIP_ERR_UNSUPPORTED, "merge of %u x %u entries per query exceeds the on-chip budget", lists, per);
    std::lock_guard g(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    if (k == 0 || per == 0) {
        HIP_TRY(launch_fill_empty(d_out_ids, d_out_dists, d_out_counts, q, k, c->stream));
    } else {
        HIP_TRY(launch_merge_topk(d_dists, d_ids, d_counts, lists, q, per, k, d_out_ids, d_out_dists, d_out_counts,
                                  c->stream));
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    return NEEDLE_HIP_OK;
}
