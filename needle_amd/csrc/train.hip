// IVF coarse-quantizer training on the device (C ABI: include/needle_hip.h, "training").
//
// Reference: IvfIndex::{kmeans, kmeans_pp_init}, src/indexing/ivf.rs:330-450 -- k-means++ seeding (weights = the
// Euclidean distance to the nearest chosen centroid, squared), Lloyd iterations until no assignment changes, the
// largest centroid shift falls below `tolerance`, or max_iterations; an empty cluster is re-seeded from a random
// row.  The reference draws from thread_rng; here the draws come from splitmix64(seed), everything else keeps the
// reference's arithmetic: distances in src/distance.rs order (sequential f32 sums), cluster sums in row order,
// one divide by the count.  Given the same draws the centroids are bit-equal to a scalar CPU run.
//
// Where the work goes: the n x k x dim distance work of seeding and of every assignment runs on the GPU (seeding:
// one exact distance column per new centroid over a dimension-major copy of the rows, coalesced; assignment: the
// flat search with k = 1 over the centroid set, matrix cores + exact re-score); the strictly sequential pieces
// (the running f32 sum of the weights and its threshold walk, n adds per new centroid) stay on the host.
#include "nh_api.h"

#include <cstring>
#include <new>

#pragma clang fp contract(off)

using namespace nh;

namespace {

__global__ __launch_bounds__(256) void tr_transpose_kernel(const float* __restrict__ v, uint32_t n, uint32_t dim,
                                                           float* __restrict__ vt) {
    __shared__ float tile[32][33];
    const uint32_t r0 = blockIdx.x * 32, c0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (uint32_t y = ty; y < 32; y += 8)
        tile[y][tx] = (r0 + y < n && c0 + tx < dim) ? v[(uint64_t)(r0 + y) * dim + c0 + tx] : 0.0f;
    __syncthreads();
    for (uint32_t y = ty; y < 32; y += 8)
        if (c0 + y < dim && r0 + tx < n) vt[(uint64_t)(c0 + y) * n + r0 + tx] = tile[tx][y];
}

// dist[i] = min(dist[i], euclidean(row i, centroid)) with the reference's sequential arithmetic (distance.rs:249-257)
__global__ __launch_bounds__(256) void tr_seed_distance_kernel(const float* __restrict__ vt, uint32_t n, uint32_t dim,
                                                               const float* __restrict__ centroid, float* __restrict__ dist) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float s = -0.0f;
    for (uint32_t j = 0; j < dim; ++j) {
        const float d = vt[(uint64_t)j * n + i] - centroid[j];
        const float p = d * d;
        s = s + p;
    }
    dist[i] = fminf(dist[i], sqrtf(s));
}

__global__ void tr_fill_kernel(float* p, uint32_t n, float v) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

__global__ void tr_narrow_kernel(const uint64_t* __restrict__ ids, const uint32_t* __restrict__ cnt, uint32_t n,
                                 uint32_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = cnt[i] ? (uint32_t)ids[i] : 0u;
}

// new centroid of cluster c, dimension j: the sum of its rows IN ROW ORDER from 0.0, divided by the count
// (ivf.rs:372-390).  One workgroup per cluster, threads over the dimensions (coalesced row reads).
__global__ __launch_bounds__(256) void tr_centroid_kernel(const float* __restrict__ v, uint32_t dim,
                                                          const uint32_t* __restrict__ order,
                                                          const uint32_t* __restrict__ offsets, float* __restrict__ out) {
    const uint32_t c = blockIdx.x, b = offsets[c], e = offsets[c + 1];
    if (e == b) return;  // empty: the host re-seeds it
    const float cnt = (float)(e - b);
    for (uint32_t j = threadIdx.x; j < dim; j += 256) {
        float s = 0.0f;
        for (uint32_t r = b; r < e; ++r) s = s + v[(uint64_t)order[r] * dim + j];
        out[(uint64_t)c * dim + j] = s / cnt;
    }
}

inline uint64_t splitmix64(uint64_t& s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
inline float rng_unit(uint64_t& s) { return (float)(splitmix64(s) >> 40) * (1.0f / 16777216.0f); }  // [0, 1), 24 bits

float host_euclidean(const float* a, const float* b, uint32_t dim) {  // distance.rs:249-257
    float s = -0.0f;
    for (uint32_t j = 0; j < dim; ++j) {
        const float d = a[j] - b[j];
        const float p = d * d;
        s = s + p;
    }
    return sqrtf(s);
}

struct DevFree {
    std::vector<void*> p;
    ~DevFree() { for (void* q : p) if (q) (void)hipFree(q); }
};

}  // namespace

static needle_hip_status ivf_train_impl(needle_hip_ctx* c, const float* vectors, uint64_t n, uint32_t dim, uint32_t n_clusters,
                                        uint32_t max_iterations, float tolerance, uint64_t seed, float* out_centroids,
                                        uint32_t* out_k);

extern "C" needle_hip_status needle_hip_ivf_train(needle_hip_ctx* c, const float* vectors, uint64_t n, uint32_t dim,
                                                  uint32_t n_clusters, uint32_t max_iterations, float tolerance,
                                                  uint64_t seed, float* out_centroids, uint32_t* out_k) {
    if (!c || !out_centroids || !out_k) return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "null argument");
    *out_k = 0;
    if (n == 0 || !vectors) return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "Empty training data");  // IvfError::EmptyTrainingData
    if (dim == 0 || n_clusters == 0) return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "dimension and cluster count must be positive");
    if (n > 0x7fffffffull) return fail(NEEDLE_HIP_ERR_UNSUPPORTED, "too many training rows");
    try {
        return ivf_train_impl(c, vectors, n, dim, n_clusters, max_iterations, tolerance, seed, out_centroids, out_k);
    } catch (const std::bad_alloc&) {
        return fail(NEEDLE_HIP_ERR_OUT_OF_MEMORY, "host allocation failed");
    }
}

static needle_hip_status ivf_train_impl(needle_hip_ctx* c, const float* vectors, uint64_t n, uint32_t dim, uint32_t n_clusters,
                                        uint32_t max_iterations, float tolerance, uint64_t seed, float* out_centroids,
                                        uint32_t* out_k) {
    HIP_TRY(hipSetDevice(c->device));
    const uint32_t nn = (uint32_t)n;
    uint32_t k = std::min<uint32_t>(n_clusters, nn);  // ivf.rs: n_clusters capped at the number of rows
    uint64_t rs = seed;

    DevFree dev;
    float *d_v = nullptr, *d_vt = nullptr, *d_dist = nullptr, *d_cent = nullptr, *d_newc = nullptr, *d_d = nullptr;
    uint32_t *d_assign = nullptr, *d_order = nullptr, *d_off = nullptr, *d_cnt = nullptr;
    uint64_t* d_ids = nullptr;
    auto dmalloc = [&](void** p, size_t bytes) {
        hipError_t e = hipMalloc(p, bytes ? bytes : 1);
        if (e == hipSuccess) dev.p.push_back(*p);
        return e;
    };
    HIP_TRY(dmalloc((void**)&d_v, (size_t)n * dim * sizeof(float)));
    HIP_TRY(dmalloc((void**)&d_vt, (size_t)n * dim * sizeof(float)));
    HIP_TRY(dmalloc((void**)&d_dist, (size_t)n * sizeof(float)));
    HIP_TRY(dmalloc((void**)&d_cent, (size_t)dim * sizeof(float)));
    HIP_TRY(hipMemcpyAsync(d_v, vectors, (size_t)n * dim * sizeof(float), hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(tr_transpose_kernel, dim3((nn + 31) / 32, (dim + 31) / 32), dim3(256), 0, c->stream, d_v, nn, dim, d_vt);
    hipLaunchKernelGGL(tr_fill_kernel, dim3((nn + 255) / 256), dim3(256), 0, c->stream, d_dist, nn, 3.40282347e+38f);
    HIP_TRY(hipGetLastError());

    // ---- k-means++ seeding, ivf.rs:396-450 ----
    std::vector<float> cent((size_t)k * dim), dist(n);
    memcpy(cent.data(), vectors + (splitmix64(rs) % n) * dim, sizeof(float) * dim);
    uint32_t nc = 1;
    for (uint32_t ci = 1; ci < k; ++ci) {
        HIP_TRY(hipMemcpyAsync(d_cent, cent.data() + (size_t)(nc - 1) * dim, sizeof(float) * dim, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(tr_seed_distance_kernel, dim3((nn + 255) / 256), dim3(256), 0, c->stream, d_vt, nn, dim, d_cent, d_dist);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(dist.data(), d_dist, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        float total = -0.0f;
        for (uint64_t i = 0; i < n; ++i) { const float w = dist[i] * dist[i]; total = total + w; }
        if (total == 0.0f) {  // every row coincides with a chosen centroid: take a row that does not, if any
            std::vector<uint64_t> rem;
            for (uint64_t i = 0; i < n; ++i) {
                bool at = false;
                for (uint32_t j = 0; j < nc && !at; ++j) at = host_euclidean(vectors + i * dim, cent.data() + (size_t)j * dim, dim) < 1e-10f;
                if (!at) rem.push_back(i);
            }
            if (rem.empty()) break;
            memcpy(cent.data() + (size_t)nc * dim, vectors + rem[splitmix64(rs) % rem.size()] * dim, sizeof(float) * dim);
            ++nc;
        } else {
            const float thr = rng_unit(rs) * total;
            float cum = 0.0f;
            for (uint64_t i = 0; i < n; ++i) {
                const float w = dist[i] * dist[i];
                cum = cum + w;
                if (cum >= thr) {
                    memcpy(cent.data() + (size_t)nc * dim, vectors + i * dim, sizeof(float) * dim);
                    ++nc;
                    break;
                }
            }
        }
    }
    k = nc;

    // ---- Lloyd iterations, ivf.rs:340-394 ----
    HIP_TRY(dmalloc((void**)&d_assign, (size_t)n * sizeof(uint32_t)));
    HIP_TRY(dmalloc((void**)&d_order, (size_t)n * sizeof(uint32_t)));
    HIP_TRY(dmalloc((void**)&d_off, (size_t)(k + 1) * sizeof(uint32_t)));
    HIP_TRY(dmalloc((void**)&d_newc, (size_t)k * dim * sizeof(float)));
    const uint32_t B = 1u << 18;
    HIP_TRY(dmalloc((void**)&d_ids, (size_t)std::min(B, nn) * sizeof(uint64_t)));
    HIP_TRY(dmalloc((void**)&d_d, (size_t)std::min(B, nn) * sizeof(float)));
    HIP_TRY(dmalloc((void**)&d_cnt, (size_t)std::min(B, nn) * sizeof(uint32_t)));
    std::vector<uint32_t> assign(n, 0u), fresh(n), order(n), offsets(k + 1);
    std::vector<float> newc((size_t)k * dim);
    needle_search_params sp{};
    sp.k = 1;
    sp.precision = NEEDLE_PRECISION_BF16;  // a hint: the hits are the exact nearest centroids either way
    for (uint32_t it = 0; it < max_iterations; ++it) {
        // nearest centroid of every row == 1-NN search of the centroid set, ties to the lower index (ivf.rs:453-459)
        needle_hip_index* ix = nullptr;
        ST_TRY(needle_hip_flat_build(c, cent.data(), k, dim, NEEDLE_METRIC_EUCLIDEAN, NEEDLE_ARITH_DISTANCE_RS, nullptr, &ix));
        needle_hip_status st = NEEDLE_HIP_OK;
        for (uint32_t i0 = 0; i0 < nn && st == NEEDLE_HIP_OK; i0 += B) {
            const uint32_t b = std::min(B, nn - i0);
            st = needle_hip_batch_search_device(ix, d_v + (size_t)i0 * dim, b, &sp, d_ids, d_d, d_cnt);
            if (st == NEEDLE_HIP_OK) {
                hipLaunchKernelGGL(tr_narrow_kernel, dim3((b + 255) / 256), dim3(256), 0, c->stream, d_ids, d_cnt, b, d_assign + i0);
                if (hipGetLastError() != hipSuccess) st = fail(NEEDLE_HIP_ERR_HIP, "assignment kernel launch failed");
            }
        }
        needle_hip_index_destroy(ix);
        ST_TRY(st);
        HIP_TRY(hipMemcpyAsync(fresh.data(), d_assign, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (fresh == assign) break;  // no assignment changed (the first pass compares with all-zero, as the reference)
        assign = fresh;
        // rows grouped by cluster, row order kept inside a cluster
        std::fill(offsets.begin(), offsets.end(), 0u);
        for (uint64_t i = 0; i < n; ++i) offsets[assign[i] + 1] += 1;
        for (uint32_t j = 0; j < k; ++j) offsets[j + 1] += offsets[j];
        {
            std::vector<uint32_t> at(offsets.begin(), offsets.end() - 1);
            for (uint64_t i = 0; i < n; ++i) order[at[assign[i]]++] = (uint32_t)i;
        }
        HIP_TRY(hipMemcpyAsync(d_order, order.data(), (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(d_off, offsets.data(), (size_t)(k + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(tr_centroid_kernel, dim3(k), dim3(256), 0, c->stream, d_v, dim, d_order, d_off, d_newc);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(newc.data(), d_newc, (size_t)k * dim * sizeof(float), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        for (uint32_t j = 0; j < k; ++j)
            if (offsets[j + 1] == offsets[j])  // empty cluster: a random row (ivf.rs:383-388)
                memcpy(newc.data() + (size_t)j * dim, vectors + (splitmix64(rs) % n) * dim, sizeof(float) * dim);
        float max_shift = 0.0f;
        for (uint32_t j = 0; j < k; ++j)
            max_shift = fmaxf(max_shift, host_euclidean(cent.data() + (size_t)j * dim, newc.data() + (size_t)j * dim, dim));
        memcpy(cent.data(), newc.data(), (size_t)k * dim * sizeof(float));
        if (max_shift < tolerance) break;
    }
    memcpy(out_centroids, cent.data(), (size_t)k * dim * sizeof(float));
    *out_k = k;
    return NEEDLE_HIP_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Product quantizer: encoding and codebook training.
//
// Reference: ProductQuantizer::{train, encode, find_nearest_centroid}, src/indexing/quantization.rs:330-446, and its
// kmeans, :597-711: per subspace 256 centroids, seeding weighted by the squared distance to the nearest chosen
// centroid (normalised by the total, threshold walk), 20 Lloyd iterations with strict '<' from f32::MAX (first
// minimum), empty clusters re-seeded from a random row.  Squared distances are the sequential sum of (a - b)^2.
// ---------------------------------------------------------------------------------------------------------------
namespace {

constexpr uint32_t PQ_LDS_DSUB = 64;  // widest sub-vector whose 256-entry codebook is staged in LDS (64 KB)

__device__ __forceinline__ float pq_l2sq(const float* __restrict__ a, const float* __restrict__ b, uint32_t n) {
    float s = -0.0f;
    for (uint32_t i = 0; i < n; ++i) { const float d = a[i] - b[i]; const float p = d * d; s = s + p; }
    return s;
}

// nearest centroid of sub-vector (row, subspace blockIdx.y): codebook of the subspace in LDS, one thread per row
template <int DSUB>  // 0: any width (the sub-vector is re-read through the cache)
__global__ __launch_bounds__(256) void pq_encode_kernel(const float* __restrict__ v, uint32_t n, uint32_t dim, uint32_t dsub_rt,
                                                        const float* __restrict__ codebooks, const uint32_t* __restrict__ ncent,
                                                        uint32_t m, uint8_t* __restrict__ out) {
    extern __shared__ float cb_lds[];  // [256][dsub] when it fits (sub-vectors of up to PQ_LDS_DSUB dimensions)
    const uint32_t s = blockIdx.y, dsub = DSUB ? DSUB : dsub_rt;
    const uint32_t k = ncent ? min(ncent[s], 256u) : 256u;
    const float* cb = codebooks + (uint64_t)s * 256 * dsub;  // wider codebooks are read through the cache
    if (dsub <= PQ_LDS_DSUB) {  // block-uniform
        for (uint32_t i = threadIdx.x; i < 256 * dsub; i += 256) cb_lds[i] = cb[i];
        __syncthreads();
        cb = cb_lds;
    }
    const uint32_t row = blockIdx.x * 256 + threadIdx.x;
    if (row >= n) return;
    const float* sub = v + (uint64_t)row * dim + s * dsub;
    float x[DSUB ? DSUB : 1];
    if (DSUB) {
#pragma unroll
        for (int j = 0; j < DSUB; ++j) x[j] = sub[j];
    }
    uint32_t best = 0;
    float bd = 3.40282347e+38f;
    for (uint32_t c = 0; c < k; ++c) {
        float d;
        if (DSUB) {
            d = -0.0f;
#pragma unroll
            for (int j = 0; j < DSUB; ++j) { const float t = x[j] - cb[c * DSUB + j]; const float p = t * t; d = d + p; }
        } else {
            d = pq_l2sq(sub, cb + c * dsub, dsub);
        }
        if (d < bd) { bd = d; best = c; }
    }
    out[(uint64_t)row * m + s] = (uint8_t)best;
}

// seeding: dist[i] = min(dist[i], |sub-vector i - centroid|^2)
__global__ __launch_bounds__(256) void pq_seed_distance_kernel(const float* __restrict__ v, uint32_t n, uint32_t dim, uint32_t off,
                                                               uint32_t dsub, const float* __restrict__ centroid,
                                                               float* __restrict__ dist) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    dist[i] = fminf(dist[i], pq_l2sq(v + (uint64_t)i * dim + off, centroid, dsub));
}

// Lloyd assignment: strict '<' from f32::MAX -> first minimum (quantization.rs:640-655)
__global__ __launch_bounds__(256) void pq_assign_kernel(const float* __restrict__ v, uint32_t n, uint32_t dim, uint32_t off,
                                                        uint32_t dsub, const float* __restrict__ cent, uint32_t k,
                                                        uint32_t* __restrict__ assign) {
    extern __shared__ float cb_lds[];
    const float* cb = cent;
    if (dsub <= PQ_LDS_DSUB) {  // block-uniform
        for (uint32_t i = threadIdx.x; i < k * dsub; i += 256) cb_lds[i] = cent[i];
        __syncthreads();
        cb = cb_lds;
    }
    const uint32_t row = blockIdx.x * 256 + threadIdx.x;
    if (row >= n) return;
    const float* sub = v + (uint64_t)row * dim + off;
    uint32_t best = 0;
    float bd = 3.40282347e+38f;
    for (uint32_t c = 0; c < k; ++c) {
        const float d = pq_l2sq(sub, cb + c * dsub, dsub);
        if (d < bd) { bd = d; best = c; }
    }
    assign[row] = best;
}

// cluster means of sub-vectors: row order, from 0.0, one divide (quantization.rs:660-690)
__global__ __launch_bounds__(64) void pq_centroid_kernel(const float* __restrict__ v, uint32_t dim, uint32_t off, uint32_t dsub,
                                                         const uint32_t* __restrict__ order, const uint32_t* __restrict__ offsets,
                                                         float* __restrict__ out) {
    const uint32_t c = blockIdx.x, b = offsets[c], e = offsets[c + 1];
    if (e == b) return;
    const float cnt = (float)(e - b);
    for (uint32_t j = threadIdx.x; j < dsub; j += 64) {
        float s = 0.0f;
        for (uint32_t r = b; r < e; ++r) s = s + v[(uint64_t)order[r] * dim + off + j];
        out[(uint64_t)c * dsub + j] = s / cnt;
    }
}

needle_hip_status pq_train_impl(needle_hip_ctx* c, const float* vectors, uint64_t n, uint32_t dim, uint32_t m, uint32_t iterations,
                                uint64_t seed, float* out_books, uint32_t* out_ncent) {
    HIP_TRY(hipSetDevice(c->device));
    const uint32_t nn = (uint32_t)n, dsub = dim / m;
    DevFree dev;
    auto dmalloc = [&](void** p, size_t bytes) {
        hipError_t e = hipMalloc(p, bytes ? bytes : 1);
        if (e == hipSuccess) dev.p.push_back(*p);
        return e;
    };
    float *d_v = nullptr, *d_dist = nullptr, *d_cent = nullptr, *d_newc = nullptr;
    uint32_t *d_assign = nullptr, *d_order = nullptr, *d_off = nullptr;
    HIP_TRY(dmalloc((void**)&d_v, (size_t)n * dim * sizeof(float)));
    HIP_TRY(dmalloc((void**)&d_dist, (size_t)n * sizeof(float)));
    HIP_TRY(dmalloc((void**)&d_cent, (size_t)256 * dsub * sizeof(float)));
    HIP_TRY(dmalloc((void**)&d_newc, (size_t)256 * dsub * sizeof(float)));
    HIP_TRY(dmalloc((void**)&d_assign, (size_t)n * sizeof(uint32_t)));
    HIP_TRY(dmalloc((void**)&d_order, (size_t)n * sizeof(uint32_t)));
    HIP_TRY(dmalloc((void**)&d_off, 257 * sizeof(uint32_t)));
    HIP_TRY(hipMemcpyAsync(d_v, vectors, (size_t)n * dim * sizeof(float), hipMemcpyHostToDevice, c->stream));
    memset(out_books, 0, (size_t)m * 256 * dsub * sizeof(float));
    uint64_t rs = seed;
    const uint32_t kmax = std::min<uint32_t>(256, nn);
    std::vector<float> cent((size_t)256 * dsub), newc((size_t)256 * dsub), dist(n);
    std::vector<uint32_t> assign(n), fresh(n), order(n), offsets(257);
    auto sub = [&](uint64_t i, uint32_t s) { return vectors + i * dim + (size_t)s * dsub; };
    for (uint32_t s = 0; s < m; ++s) {
        const uint32_t off = s * dsub;
        // ---- seeding, quantization.rs:603-637 ----
        memcpy(cent.data(), sub(splitmix64(rs) % n, s), sizeof(float) * dsub);
        uint32_t nc = 1;
        hipLaunchKernelGGL(tr_fill_kernel, dim3((nn + 255) / 256), dim3(256), 0, c->stream, d_dist, nn, 3.40282347e+38f);
        HIP_TRY(hipGetLastError());
        uint32_t scored = 0;  // centroids already folded into d_dist
        while (nc < kmax) {
            for (; scored < nc; ++scored) {
                HIP_TRY(hipMemcpyAsync(d_cent, cent.data() + (size_t)scored * dsub, sizeof(float) * dsub, hipMemcpyHostToDevice, c->stream));
                hipLaunchKernelGGL(pq_seed_distance_kernel, dim3((nn + 255) / 256), dim3(256), 0, c->stream, d_v, nn, dim, off, dsub,
                                   d_cent, d_dist);
                HIP_TRY(hipGetLastError());
                HIP_TRY(hipStreamSynchronize(c->stream));  // d_cent is reused by the next copy
            }
            HIP_TRY(hipMemcpyAsync(dist.data(), d_dist, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            float total = -0.0f;
            for (uint64_t i = 0; i < n; ++i) total = total + dist[i];
            if (total <= 0.0f) {  // every row equals a chosen centroid: take one that does not, if any
                std::vector<uint64_t> rem;
                for (uint64_t i = 0; i < n; ++i) {
                    bool at = false;
                    for (uint32_t j = 0; j < nc && !at; ++j) at = memcmp(sub(i, s), cent.data() + (size_t)j * dsub, sizeof(float) * dsub) == 0;
                    if (!at) rem.push_back(i);
                }
                if (rem.empty()) break;
                memcpy(cent.data() + (size_t)nc * dsub, sub(rem[splitmix64(rs) % rem.size()], s), sizeof(float) * dsub);
                ++nc;
                continue;
            }
            const float r = rng_unit(rs);
            float cum = 0.0f;
            for (uint64_t i = 0; i < n; ++i) {
                const float w = dist[i] / total;
                cum = cum + w;
                if (cum >= r) {
                    memcpy(cent.data() + (size_t)nc * dsub, sub(i, s), sizeof(float) * dsub);
                    ++nc;
                    break;
                }
            }
            // rounding can leave cum < r after the walk: draw again, as the reference's loop does
        }
        const uint32_t k = nc;
        // ---- Lloyd iterations, quantization.rs:639-708 ----
        std::fill(assign.begin(), assign.end(), 0u);
        for (uint32_t it = 0; it < iterations; ++it) {
            HIP_TRY(hipMemcpyAsync(d_cent, cent.data(), (size_t)k * dsub * sizeof(float), hipMemcpyHostToDevice, c->stream));
            hipLaunchKernelGGL(pq_assign_kernel, dim3((nn + 255) / 256), dim3(256), dsub <= PQ_LDS_DSUB ? (size_t)k * dsub * sizeof(float) : 0, c->stream, d_v, nn,
                               dim, off, dsub, d_cent, k, d_assign);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpyAsync(fresh.data(), d_assign, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            if (fresh == assign) break;
            assign = fresh;
            std::fill(offsets.begin(), offsets.end(), 0u);
            for (uint64_t i = 0; i < n; ++i) offsets[assign[i] + 1] += 1;
            for (uint32_t j = 0; j < 256; ++j) offsets[j + 1] += offsets[j];
            {
                std::vector<uint32_t> at(offsets.begin(), offsets.end() - 1);
                for (uint64_t i = 0; i < n; ++i) order[at[assign[i]]++] = (uint32_t)i;
            }
            HIP_TRY(hipMemcpyAsync(d_order, order.data(), (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
            HIP_TRY(hipMemcpyAsync(d_off, offsets.data(), 257 * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
            hipLaunchKernelGGL(pq_centroid_kernel, dim3(k), dim3(64), 0, c->stream, d_v, dim, off, dsub, d_order, d_off, d_newc);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpyAsync(newc.data(), d_newc, (size_t)k * dsub * sizeof(float), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            for (uint32_t j = 0; j < k; ++j)
                if (offsets[j + 1] == offsets[j]) memcpy(newc.data() + (size_t)j * dsub, sub(splitmix64(rs) % n, s), sizeof(float) * dsub);
            memcpy(cent.data(), newc.data(), (size_t)k * dsub * sizeof(float));
        }
        memcpy(out_books + (size_t)s * 256 * dsub, cent.data(), (size_t)k * dsub * sizeof(float));
        if (out_ncent) out_ncent[s] = k;
    }
    return NEEDLE_HIP_OK;
}

}  // namespace

hipError_t nh::launch_pq_encode(const float* d_v, uint32_t n, uint32_t dim, uint32_t m, const float* d_books, const uint32_t* d_ncent,
                            uint8_t* d_out, hipStream_t stream) {
    const uint32_t dsub = dim / m;
    const dim3 grid((n + 255) / 256, m), block(256);
    const size_t lds = dsub <= PQ_LDS_DSUB ? (size_t)256 * dsub * sizeof(float) : 0;
    switch (dsub) {
    case 4: hipLaunchKernelGGL(pq_encode_kernel<4>, grid, block, lds, stream, d_v, n, dim, dsub, d_books, d_ncent, m, d_out); break;
    case 8: hipLaunchKernelGGL(pq_encode_kernel<8>, grid, block, lds, stream, d_v, n, dim, dsub, d_books, d_ncent, m, d_out); break;
    case 16: hipLaunchKernelGGL(pq_encode_kernel<16>, grid, block, lds, stream, d_v, n, dim, dsub, d_books, d_ncent, m, d_out); break;
    default: hipLaunchKernelGGL(pq_encode_kernel<0>, grid, block, lds, stream, d_v, n, dim, dsub, d_books, d_ncent, m, d_out); break;
    }
    return hipGetLastError();
}

extern "C" needle_hip_status needle_hip_pq_train(needle_hip_ctx* c, const float* vectors, uint64_t n, uint32_t dim, uint32_t m,
                                                 uint32_t iterations, uint64_t seed, float* out_codebooks, uint32_t* out_ncent) {
    if (!c || !out_codebooks) return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "null argument");
    if (n == 0 || !vectors) return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "Empty training data");
    if (dim == 0 || m == 0 || dim % m != 0)
        return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "dim (%u) must be divisible by the subvector count (%u)", dim, m);
    if (n > 0x7fffffffull) return fail(NEEDLE_HIP_ERR_UNSUPPORTED, "too many training rows");
    try {
        return pq_train_impl(c, vectors, n, dim, m, iterations, seed, out_codebooks, out_ncent);
    } catch (const std::bad_alloc&) {
        return fail(NEEDLE_HIP_ERR_OUT_OF_MEMORY, "host allocation failed");
    }
}

extern "C" needle_hip_status needle_hip_pq_encode(needle_hip_ctx* c, const float* codebooks, uint32_t dim, uint32_t m,
                                                  const uint32_t* ncent, const float* vectors, uint64_t n, uint8_t* out_codes) {
    if (!c) return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "null argument");
    if (n == 0) return NEEDLE_HIP_OK;
    if (!codebooks || !vectors || !out_codes) return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "null argument");
    if (dim == 0 || m == 0 || dim % m != 0)
        return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "dim (%u) must be divisible by the subvector count (%u)", dim, m);
    std::lock_guard g(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    const uint64_t B = 1u << 20;  // rows per upload
    const uint32_t dsub = dim / m;
    HIP_TRY(c->scratch[0].ensure((size_t)std::min<uint64_t>(B, n) * dim * sizeof(float)));
    HIP_TRY(c->scratch[1].ensure((size_t)m * 256 * dsub * sizeof(float)));
    HIP_TRY(c->scratch[2].ensure((size_t)std::min<uint64_t>(B, n) * m));
    HIP_TRY(c->scratch[3].ensure((size_t)m * sizeof(uint32_t)));
    HIP_TRY(hipMemcpyAsync(c->scratch[1].p, codebooks, (size_t)m * 256 * dsub * sizeof(float), hipMemcpyHostToDevice, c->stream));
    if (ncent) HIP_TRY(hipMemcpyAsync(c->scratch[3].p, ncent, (size_t)m * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    for (uint64_t i0 = 0; i0 < n; i0 += B) {
        const uint32_t b = (uint32_t)std::min<uint64_t>(B, n - i0);
        HIP_TRY(hipMemcpyAsync(c->scratch[0].p, vectors + i0 * dim, (size_t)b * dim * sizeof(float), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(launch_pq_encode(c->scratch[0].as<float>(), b, dim, m, c->scratch[1].as<float>(),
                                 ncent ? c->scratch[3].as<uint32_t>() : nullptr, c->scratch[2].as<uint8_t>(), c->stream));
        HIP_TRY(hipMemcpyAsync(out_codes + i0 * m, c->scratch[2].p, (size_t)b * m, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return NEEDLE_HIP_OK;
}
