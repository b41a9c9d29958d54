// Quantized (ADC) indexes behind the C ABI (include/needle_hip.h, "quantized (ADC) indexes").
#include "nh_api.h"

using namespace nh;

extern "C" needle_hip_status needle_hip_pq_build(needle_hip_ctx* c, const uint8_t* codes, uint64_t n, uint32_t dim,
                                                 uint32_t m, const float* codebooks, const uint32_t* valid_bitmask,
                                                 needle_hip_index** out) {
    if (!c || !out) return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    if (dim == 0 || m == 0 || dim % m != 0 || !codebooks)
        return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "need codebooks and dim (%u) divisible by the subvector count (%u)", dim, m);
    if (n > 0 && !codes) return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "codes is null");
    if (n >= 0xfffffff0ull) return fail(NEEDLE_HIP_ERR_UNSUPPORTED, "too many rows");
    HIP_TRY(hipSetDevice(c->device));
    needle_hip_index* ix = index_new();
    if (!ix) return fail(NEEDLE_HIP_ERR_OUT_OF_MEMORY, "host allocation failed");
    ix->ctx = c;
    ix->kind = KIND_PQ;
    ix->n = n;
    ix->dim = dim;
    ix->n_pad = round_up((uint32_t)n, TILE_ROWS);  // rows of the decoded bf16 operand, built on the first bf16 search
    ix->metric = NEEDLE_METRIC_EUCLIDEAN;
    ix->arith = NEEDLE_ARITH_DISTANCE_RS;
    ix->pq_m = m;
    ix->pq_dsub = dim / m;
    const size_t cb_bytes = (size_t)m * 256 * ix->pq_dsub * sizeof(float);
    hipError_t e = hipMalloc((void**)&ix->d_codebooks, cb_bytes);
    if (e == hipSuccess) e = hipMemcpyAsync(ix->d_codebooks, codebooks, cb_bytes, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess && n > 0) {
        e = hipMalloc((void**)&ix->d_codes, (size_t)n * m + 16);
        if (e == hipSuccess) e = hipMemcpyAsync(ix->d_codes, codes, (size_t)n * m, hipMemcpyHostToDevice, c->stream);
        // the same codes, 16 subspaces of a row per 16-byte read, for the distance-table scan
        if (e == hipSuccess) e = hipMalloc((void**)&ix->d_codes_gm, (size_t)((m + PQ_SG - 1) / PQ_SG) * n * PQ_SG);
        if (e == hipSuccess) e = launch_pq_group_codes(ix->d_codes, n, m, ix->d_codes_gm, c->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    needle_hip_status st = NEEDLE_HIP_OK;
    if (e != hipSuccess)
        st = fail(e == hipErrorOutOfMemory ? NEEDLE_HIP_ERR_OUT_OF_MEMORY : NEEDLE_HIP_ERR_HIP, "HIP error %d (%s) uploading codes",
                  (int)e, hipGetErrorString(e));
    if (st == NEEDLE_HIP_OK) st = upload_mask(ix, valid_bitmask);
    if (st != NEEDLE_HIP_OK) { index_free(ix); return st; }
    *out = ix;
    return NEEDLE_HIP_OK;
}

needle_hip_status nh::pq_search_batch_device(needle_hip_index* ix, const float* d_queries, uint32_t q, uint32_t k,
                                             const uint32_t* d_filter, uint64_t* d_out_ids, float* d_out_dists,
                                             uint32_t* d_out_counts) {
    needle_hip_ctx* c = ix->ctx;
    const uint32_t n = (uint32_t)ix->n;
    if (k > KPRIME_MAX) {  // beyond the per-list budget of the table scan: every row scored in the table's arithmetic
        ST_TRY(run_exact(ix, d_queries, nullptr, q, k, d_filter, d_out_ids, d_out_dists, d_out_counts));
        HIP_TRY(hipStreamSynchronize(c->stream));
        std::lock_guard g(c->mu);
        c->m.fallback_queries += q;
        return NEEDLE_HIP_OK;
    }
    // Queries go through in chunks whose interleaved distance tables (1 KB per query and subspace, whole slices of 16
    // subspaces) stay within 1 GB.
    const uint32_t m = ix->pq_m;
    const uint32_t q_chunk = std::max<uint32_t>(
        PQ_QB, (uint32_t)std::min<uint64_t>(16384, (1ull << 30) / ((uint64_t)round_up(m, (uint32_t)PQ_SG) * 1024)) / PQ_QB * PQ_QB);
    // Room per list: a compaction (cut back to the k best, the list's own threshold) holds up the whole workgroup at its
    // next barrier, so with seeded thresholds a list should seldom need one before the scan ends.
    const uint32_t cap = std::max<uint32_t>(2 * k + 64, 320);
    for (uint32_t q0 = 0; q0 < q; q0 += q_chunk) {
        const uint32_t qc = std::min(q_chunk, q - q0);
        // Row splits of the code matrix.  Workgroup ids go round the XCDs, and the workgroups an XCD runs at a time (one per
        // CU) are the S splits of 32 / S query groups: S is a power of two up to the CUs of an XCD, bounded by whole blocks
        // of rows per split and by the PQ_WAVES * S lists of k entries per query the merge kernel gathers in its LDS.  Among
        // those the S whose rounds of workgroups cost the least (rounds x rows per split), the smallest on a tie: every
        // list is workspace and one more cold start.
        const uint32_t groups = (qc + PQ_QB - 1) / PQ_QB;
        const uint32_t xcd_slots = std::max<uint32_t>(1, (uint32_t)c->n_cu / 8);
        uint32_t s_max = std::max<uint32_t>(1, n / (4 * pq_scan_block_rows()));
        s_max = std::min<uint32_t>(s_max, std::max<uint32_t>(1, 60000u / ((uint32_t)PQ_WAVES * k * (uint32_t)sizeof(uint64_t))));
        uint32_t S = 1;
        double best_cost = 0.0;
        for (uint32_t s = 1; s <= s_max && xcd_slots % s == 0; s *= 2) {
            const uint32_t per_round = 8 * (xcd_slots / s), rounds = (groups + per_round - 1) / per_round;
            const double cost = (double)rounds / s;
            if (s == 1 || cost < best_cost * 0.999) { best_cost = cost; S = s; }
        }
        if (c->tune.splits && c->tune.splits <= s_max && xcd_slots % c->tune.splits == 0) S = c->tune.splits;
        const uint32_t rows_per_split = round_up((n + S - 1) / S, (uint32_t)PQ_SG);  // see PqScanParams
        // Threshold seeding: every wave keeps a list of its own per query, and a list that starts empty takes (and keeps
        // compacting) the first rows it meets.  A first launch scans a sample -- two blocks of rows at each of 8 places
        // spread over the base -- and the k-th best score it finds per query is where every list of the real scan
        // starts: a row scoring below it has k better rows in the sample alone.
        // (planning override pre_tiles > 0: one block per place, and bases from twice the sample on -- for tests)
        const bool forced = c->tune.pre_tiles > 0;
        const uint32_t pre_S = 8, pre_rows = (forced ? 1 : 2) * pq_scan_block_rows();
        const bool seed = n >= (forced ? 2 : 16) * pre_S * pre_rows && !(c->tune.debug & 2u) && xcd_slots % pre_S == 0;
        const uint32_t lists_per_q = PQ_WAVES * S, pre_lists_per_q = PQ_WAVES * pre_S, max_lists = std::max(lists_per_q, seed ? pre_lists_per_q : 0u);
        const uint32_t pre_cap = 2 * k + 64;  // the sample's many short-lived lists stay small (they may compact as they go)

        HIP_TRY(ix->w_pqtab.ensure(pq_table_bytes(qc, m)));
        HIP_TRY(ix->w_lists.ensure(std::max((size_t)qc * lists_per_q * cap, seed ? (size_t)qc * pre_lists_per_q * pre_cap : (size_t)0) * sizeof(uint2)));
        HIP_TRY(ix->w_lcnt.ensure((size_t)qc * max_lists * sizeof(uint32_t)));
        HIP_TRY(ix->w_cand.ensure((size_t)qc * k * sizeof(uint32_t)));
        HIP_TRY(ix->w_scores.ensure((size_t)qc * k * sizeof(float)));
        HIP_TRY(ix->w_candn.ensure((size_t)qc * sizeof(uint32_t)));
        HIP_TRY(ix->w_tau.ensure((size_t)qc * sizeof(float)));

        PqScanParams sp{};
        sp.codes_gm = ix->d_codes_gm;
        sp.tables = ix->w_pqtab.as<float>();
        sp.valid = ix->d_valid;
        sp.filter = d_filter;
        sp.lists = ix->w_lists.as<uint2>();
        sp.list_counts = ix->w_lcnt.as<uint32_t>();
        sp.n = n; sp.m = m; sp.n_sg = (m + PQ_SG - 1) / PQ_SG; sp.q = qc;
        sp.k = k; sp.cap = cap; sp.xcd_slots = xcd_slots; sp.dev = c->tune.debug & (1u | 256u | 512u);
        PostParams pp{};
        pp.lists = sp.lists;
        pp.list_counts = sp.list_counts;
        pp.cand_ids = ix->w_cand.as<uint32_t>();
        pp.cand_scores = ix->w_scores.as<float>();
        pp.cand_n = ix->w_candn.as<uint32_t>();
        pp.tau_star = ix->w_tau.as<float>();
        pp.out_ids = d_out_ids + (size_t)q0 * k;
        pp.out_dists = d_out_dists + (size_t)q0 * k;
        pp.out_counts = d_out_counts + q0;
        pp.q = qc; pp.n = n; pp.dim = ix->dim; pp.k = k; pp.kprime = k; pp.cap = cap;

        if (c->profiling) HIP_TRY(hipEventRecord(c->ev[0], c->stream));
        HIP_TRY(launch_pq_tables(d_queries + (size_t)q0 * ix->dim, ix->d_codebooks, qc, m, ix->pq_dsub, ix->dim, ix->w_pqtab.as<float>(),
                                 c->stream));
        if (seed) {
            HIP_TRY(hipMemsetAsync(ix->w_lcnt.p, 0, (size_t)qc * pre_lists_per_q * sizeof(uint32_t), c->stream));
            sp.n_splits = pre_S; sp.rows_per_split = pre_rows; sp.split_stride = n / pre_S / PQ_SG * PQ_SG; sp.tau_seed = nullptr;
            sp.cap = pp.cap = pre_cap;
            HIP_TRY(launch_pq_scan(sp, c->stream));
            pp.n_splits = pre_lists_per_q;
            HIP_TRY(launch_merge_select(pp, c->stream));  // tau_star: the k-th best score of the sample (-inf: fewer than k live rows)
            sp.cap = pp.cap = cap;
        }
        HIP_TRY(hipMemsetAsync(ix->w_lcnt.p, 0, (size_t)qc * lists_per_q * sizeof(uint32_t), c->stream));
        sp.n_splits = S; sp.rows_per_split = rows_per_split; sp.split_stride = rows_per_split;
        sp.tau_seed = seed ? ix->w_tau.as<float>() : nullptr;
        HIP_TRY(launch_pq_scan(sp, c->stream));
        if (c->profiling) HIP_TRY(hipEventRecord(c->ev[1], c->stream));

        pp.n_splits = lists_per_q;
        HIP_TRY(launch_merge_select(pp, c->stream));
        HIP_TRY(launch_emit_sorted(pp, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (c->profiling) {
            float ms = 0.0f;
            HIP_TRY(hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
            std::lock_guard g(c->mu);
            c->m.scan_launches += 1;
            c->m.scan_ms_total += ms;
            c->m.last_scan_ms = ms;
        }
    }
    return NEEDLE_HIP_OK;
}

// ---------------------------------------------------------------------------
// scalar-quantized code matrix
// ---------------------------------------------------------------------------
extern "C" needle_hip_status needle_hip_sq8_build(needle_hip_ctx* c, const uint8_t* codes, uint64_t n, uint32_t dim,
                                                  const float* min_vals, const float* scale,
                                                  const uint32_t* valid_bitmask, needle_hip_index** out) {
    if (!c || !out) return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    if (dim == 0 || !min_vals || !scale) return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "need min_vals and scale of a trained quantizer");
    if (n > 0 && !codes) return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "codes is null");
    if (n >= 0xfffffff0ull - TILE_ROWS) return fail(NEEDLE_HIP_ERR_UNSUPPORTED, "too many rows");
    double mn2 = 0.0;
    for (uint32_t j = 0; j < dim; ++j) {
        if (!std::isfinite(min_vals[j]) || !std::isfinite(scale[j]) || scale[j] == 0.0f)
            return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "quantizer parameter %u is not finite / zero scale", j);
        mn2 += (double)min_vals[j] * (double)min_vals[j];
    }
    HIP_TRY(hipSetDevice(c->device));
    needle_hip_index* ix = index_new();
    if (!ix) return fail(NEEDLE_HIP_ERR_OUT_OF_MEMORY, "host allocation failed");
    ix->ctx = c;
    ix->kind = KIND_SQ8;
    ix->n = n;
    ix->dim = dim;
    ix->n_pad = round_up((uint32_t)n, TILE_ROWS);
    ix->metric = NEEDLE_METRIC_EUCLIDEAN;
    ix->arith = NEEDLE_ARITH_DISTANCE_RS;
    ix->dpad_f32 = round_up(dim, flat_scan_kstep(NEEDLE_PRECISION_FP32));
    ix->sq_min_norm = std::sqrt(mn2);
    uint32_t* d_bits = nullptr;
    hipError_t e = hipMalloc((void**)&ix->d_sq_min, sizeof(float) * dim);
    if (e == hipSuccess) e = hipMalloc((void**)&ix->d_sq_scale, sizeof(float) * dim);
    if (e == hipSuccess) e = hipMemcpyAsync(ix->d_sq_min, min_vals, sizeof(float) * dim, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(ix->d_sq_scale, scale, sizeof(float) * dim, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess && n > 0) {
        const size_t bytes = (size_t)ix->n_pad * ix->dpad_f32;
        e = hipMalloc((void**)&ix->d_codes, bytes);
        if (e == hipSuccess) e = hipMemsetAsync(ix->d_codes, 0, bytes, c->stream);
        // One contiguous upload (a strided copy from pageable host memory moves row by row: 56 s for 6.25M x 384
        // codes); rows are then spread to the padded stride on the device when the stride differs.
        uint8_t* d_tmp = nullptr;
        if (e == hipSuccess && ix->dpad_f32 == dim) {
            e = hipMemcpyAsync(ix->d_codes, codes, (size_t)n * dim, hipMemcpyHostToDevice, c->stream);
        } else if (e == hipSuccess) {
            e = hipMalloc((void**)&d_tmp, (size_t)n * dim);
            if (e == hipSuccess) e = hipMemcpyAsync(d_tmp, codes, (size_t)n * dim, hipMemcpyHostToDevice, c->stream);
            if (e == hipSuccess)
                e = hipMemcpy2DAsync(ix->d_codes, ix->dpad_f32, d_tmp, dim, dim, n, hipMemcpyDeviceToDevice, c->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
            if (d_tmp) (void)hipFree(d_tmp);
        }
        if (e == hipSuccess) e = hipMalloc((void**)&ix->d_cinit, sizeof(float) * ix->n_pad);
        if (e == hipSuccess) e = hipMemsetAsync(ix->d_cinit, 0, sizeof(float) * ix->n_pad, c->stream);
        if (e == hipSuccess) e = hipMalloc((void**)&d_bits, 2 * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMemsetAsync(d_bits, 0, 2 * sizeof(uint32_t), c->stream);
        if (e == hipSuccess)
            e = launch_sq8_row_prep(ix->d_codes, n, dim, ix->dpad_f32, ix->d_sq_min, ix->d_sq_scale, ix->d_cinit, d_bits,
                                    c->stream);
        uint32_t bits[2] = {0, 0};
        if (e == hipSuccess) e = hipMemcpyAsync(bits, d_bits, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        memcpy(&ix->max_norm, &bits[0], sizeof(float));
        memcpy(&ix->bf16_err, &bits[1], sizeof(float));  // max |code vector|: scales the bf16 query rounding error
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (d_bits) (void)hipFree(d_bits);
    needle_hip_status st = NEEDLE_HIP_OK;
    if (e != hipSuccess)
        st = fail(e == hipErrorOutOfMemory ? NEEDLE_HIP_ERR_OUT_OF_MEMORY : NEEDLE_HIP_ERR_HIP, "HIP error %d (%s) uploading codes",
                  (int)e, hipGetErrorString(e));
    if (st == NEEDLE_HIP_OK) st = upload_mask(ix, valid_bitmask);
    if (st != NEEDLE_HIP_OK) { index_free(ix); return st; }
    *out = ix;
    return NEEDLE_HIP_OK;
}
