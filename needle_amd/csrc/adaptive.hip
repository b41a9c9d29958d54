// Two-phase search over Matryoshka-style embeddings (C ABI: include/needle_hip.h, "two-phase search").
//
// Reference: adaptive_search, src/indexing/quantization.rs:890-943 -- phase 1 keeps k * overfetch candidates by the
// Euclidean distance over a PREFIX of the dimensions (MatryoshkaTruncation::truncate, :801-804), phase 2 re-ranks them
// by the distance over all dimensions and keeps k.
//
// Phase 1 is the exact flat search of a Euclidean index over the prefix rows (needle_hip_flat_build_prefix builds
// it from the resident full rows), so its candidates are the k * overfetch nearest by (coarse distance, row) -- the
// intended set; the reference's own heap evicts its NEAREST entry once full (:913-918), which none of its tests
// reaches.  Phase 2 keeps the reference's order exactly: its candidates arrive farthest-first
// (BinaryHeap<Reverse<..>>::into_sorted_vec, :922-926) and are stably sorted by full distance (:939), so of two
// candidates at the same full distance the one with the LARGER (coarse distance, row) comes first.
#include "nh_api.h"
#include "nh_exact.h"

using namespace nh;

namespace {

__global__ void ad_prefix_kernel(const float* __restrict__ src, uint64_t n, uint32_t dim, uint32_t dims, float* __restrict__ dst) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * dims) return;
    dst[i] = src[(i / dims) * dim + i % dims];
}

constexpr uint32_t AD_MAX_CAND = K_MAX;

// one workgroup per query: full-dimension distances of its candidates in the reference's scalar order, ranked by
// (full distance, coarse rank descending)
__global__ __launch_bounds__(256) void ad_rerank_kernel(const float* __restrict__ queries, const float* __restrict__ rows,
                                                        uint32_t dim, const uint64_t* __restrict__ cand_ids,
                                                        const float* __restrict__ cand_d, const uint32_t* __restrict__ cand_n,
                                                        uint32_t coarse_k, uint32_t k, uint64_t* __restrict__ out_ids,
                                                        float* __restrict__ out_d, float* __restrict__ out_coarse,
                                                        uint32_t* __restrict__ out_n) {
    __shared__ uint64_t key[AD_MAX_CAND];
    const uint32_t q = blockIdx.x, t = threadIdx.x;
    const uint32_t c = min(cand_n[q], coarse_k);
    const float* qv = queries + (uint64_t)q * dim;
    for (uint32_t j = t; j < c; j += 256) {
        const uint64_t id = cand_ids[(uint64_t)q * coarse_k + j];
        const float d = canon_distance(exact_rs(NEEDLE_METRIC_EUCLIDEAN, qv, rows + id * dim, dim));
        key[j] = ((uint64_t)f2ord(d) << 32) | (uint64_t)(0xffffffffu - j);  // j = rank by (coarse distance, row)
    }
    __syncthreads();
    const uint32_t kk = min(k, c);
    for (uint32_t j = t; j < c; j += 256) {
        const uint64_t mine = key[j];
        uint32_t rank = 0;
        for (uint32_t i = 0; i < c; ++i) rank += key[i] < mine ? 1u : 0u;
        if (rank < kk) {
            out_ids[(uint64_t)q * k + rank] = cand_ids[(uint64_t)q * coarse_k + j];
            out_d[(uint64_t)q * k + rank] = ord2f((uint32_t)(mine >> 32));
            out_coarse[(uint64_t)q * k + rank] = cand_d[(uint64_t)q * coarse_k + j];
        }
    }
    for (uint32_t r = kk + t; r < k; r += 256) {
        out_ids[(uint64_t)q * k + r] = ~0ull;
        out_d[(uint64_t)q * k + r] = INFINITY;
        out_coarse[(uint64_t)q * k + r] = INFINITY;
    }
    if (t == 0) out_n[q] = kk;
}

struct DevMem {
    std::vector<void*> p;
    ~DevMem() { for (void* q : p) if (q) (void)hipFree(q); }
    hipError_t get(void** out, size_t bytes) {
        hipError_t e = hipMalloc(out, bytes ? bytes : 1);
        if (e == hipSuccess) p.push_back(*out);
        return e;
    }
};

}  // namespace

extern "C" needle_hip_status needle_hip_flat_build_prefix(needle_hip_index* full, uint32_t dims, needle_hip_index** out) {
    if (!full || !out) return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    if (full->kind != KIND_FLAT || (full->n > 0 && !full->d_raw))
        return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "the prefix index is built from a flat index holding its rows");
    if (dims == 0 || dims > full->dim) return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "prefix of %u dimensions of %u", dims, full->dim);
    needle_hip_ctx* c = full->ctx;
    std::lock_guard g(full->mu);
    HIP_TRY(hipSetDevice(c->device));
    DevMem mem;
    float* d_tmp = nullptr;
    std::vector<uint32_t> mask;
    try {
        if (full->d_valid) mask.resize((full->n + 31) / 32);
    } catch (...) {
        return fail(NEEDLE_HIP_ERR_OUT_OF_MEMORY, "host allocation failed");
    }
    if (full->n > 0) {
        HIP_TRY(mem.get((void**)&d_tmp, (size_t)full->n * dims * sizeof(float)));
        const uint64_t total = full->n * dims;
        hipLaunchKernelGGL(ad_prefix_kernel, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, c->stream, full->d_raw, full->n,
                           full->dim, dims, d_tmp);
        HIP_TRY(hipGetLastError());
        if (full->d_valid)
            HIP_TRY(hipMemcpyAsync(mask.data(), full->d_valid, mask.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return needle_hip_flat_build_device(c, d_tmp, full->n, dims, NEEDLE_METRIC_EUCLIDEAN, NEEDLE_ARITH_DISTANCE_RS,
                                        full->d_valid ? mask.data() : nullptr, out);
}

extern "C" needle_hip_status needle_hip_adaptive_search(needle_hip_index* full, needle_hip_index* coarse, const float* queries,
                                                        uint32_t q, uint32_t query_dim, uint32_t k, uint32_t overfetch,
                                                        uint64_t* out_ids, float* out_dists, float* out_coarse_dists,
                                                        uint32_t* out_counts) {
    if (!full || !coarse) return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "null argument");
    if (full->kind != KIND_FLAT || coarse->kind != KIND_FLAT || coarse->metric != NEEDLE_METRIC_EUCLIDEAN ||
        coarse->n != full->n || coarse->dim > full->dim || coarse->ctx != full->ctx)
        return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "the coarse index must be the Euclidean prefix index of the full one");
    if (q > 0 && query_dim != full->dim)
        return fail(NEEDLE_HIP_ERR_DIMENSION_MISMATCH, "Dimension mismatch: expected %u, got %u", full->dim, query_dim);
    if (q == 0) return NEEDLE_HIP_OK;
    if (!queries || !out_counts || (k > 0 && (!out_ids || !out_dists || !out_coarse_dists)))
        return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "null query or output pointer");
    for (uint64_t i = 0; i < (uint64_t)q * query_dim; ++i)
        if (!std::isfinite(queries[i]))
            return fail(NEEDLE_HIP_ERR_INVALID_VECTOR, "Invalid vector: Vector contains %s at index %u (query %u)",
                        std::isnan(queries[i]) ? "NaN" : "Inf", (unsigned)(i % query_dim), (unsigned)(i / query_dim));
    if (k == 0 || full->n == 0) {  // no hits: counts zero, every output row padded as the ABI promises
        memset(out_counts, 0, sizeof(uint32_t) * q);
        for (uint64_t i = 0; i < (uint64_t)q * k; ++i) {
            out_ids[i] = UINT64_MAX;
            out_dists[i] = INFINITY;
            out_coarse_dists[i] = INFINITY;
        }
        return NEEDLE_HIP_OK;
    }
    const uint64_t want = (uint64_t)k * std::max<uint32_t>(overfetch, 1);  // quantization.rs:903
    const uint32_t coarse_k = (uint32_t)std::min<uint64_t>(want, full->n);
    if (coarse_k > AD_MAX_CAND)
        return fail(NEEDLE_HIP_ERR_UNSUPPORTED, "k * overfetch = %u candidates exceeds %u", coarse_k, AD_MAX_CAND);
    needle_hip_ctx* c = full->ctx;
    HIP_TRY(hipSetDevice(c->device));
    DevMem mem;
    float *d_q = nullptr, *d_qt = nullptr, *d_cd = nullptr, *d_od = nullptr, *d_oc = nullptr;
    uint64_t *d_cids = nullptr, *d_oids = nullptr;
    uint32_t *d_cn = nullptr, *d_on = nullptr;
    const uint32_t cd = coarse->dim;
    HIP_TRY(mem.get((void**)&d_q, (size_t)q * query_dim * sizeof(float)));
    HIP_TRY(mem.get((void**)&d_qt, (size_t)q * cd * sizeof(float)));
    HIP_TRY(mem.get((void**)&d_cids, (size_t)q * coarse_k * sizeof(uint64_t)));
    HIP_TRY(mem.get((void**)&d_cd, (size_t)q * coarse_k * sizeof(float)));
    HIP_TRY(mem.get((void**)&d_cn, (size_t)q * sizeof(uint32_t)));
    HIP_TRY(mem.get((void**)&d_oids, (size_t)q * k * sizeof(uint64_t)));
    HIP_TRY(mem.get((void**)&d_od, (size_t)q * k * sizeof(float)));
    HIP_TRY(mem.get((void**)&d_oc, (size_t)q * k * sizeof(float)));
    HIP_TRY(mem.get((void**)&d_on, (size_t)q * sizeof(uint32_t)));
    HIP_TRY(hipMemcpyAsync(d_q, queries, (size_t)q * query_dim * sizeof(float), hipMemcpyHostToDevice, c->stream));
    const uint64_t total = (uint64_t)q * cd;
    hipLaunchKernelGGL(ad_prefix_kernel, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, c->stream, d_q, (uint64_t)q, query_dim, cd,
                       d_qt);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    // phase 1: the coarse_k nearest over the prefix, ascending by (coarse distance, row)
    needle_search_params sp{};
    sp.k = coarse_k;
    sp.precision = NEEDLE_PRECISION_BF16;
    ST_TRY(needle_hip_batch_search_device(coarse, d_qt, q, &sp, d_cids, d_cd, d_cn));
    // phase 2: re-rank over all dimensions
    {
        std::lock_guard g(full->mu);
        hipLaunchKernelGGL(ad_rerank_kernel, dim3(q), dim3(256), 0, c->stream, d_q, full->d_raw, full->dim, d_cids, d_cd, d_cn, coarse_k,
                           k, d_oids, d_od, d_oc, d_on);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(out_ids, d_oids, (size_t)q * k * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(out_dists, d_od, (size_t)q * k * sizeof(float), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(out_coarse_dists, d_oc, (size_t)q * k * sizeof(float), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(out_counts, d_on, (size_t)q * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return NEEDLE_HIP_OK;
}
