// Operator-level calls with no index behind them: normalisation, top-k of score vectors, host-pointer shard merge
// (declarations and reference citations: include/needle_hip.h).
#include "nh_api.h"
#include "nh_exact.h"
#include "nh_topk.h"

#pragma clang fp contract(off)

using namespace nh;

// ---------------------------------------------------------------------------
// normalize
// ---------------------------------------------------------------------------
// The norm is a sequential f32 sum in the reference's order, so one thread owns a row's additions; the rows themselves
// are staged through LDS in pieces of 32 dimensions with coalesced loads (256 rows per workgroup).
//   DISTANCE_RS : normalize, src/distance.rs:426-434 + 464-469 -- dot folded from -0.0, inv = 1 / sqrt, x * inv
//   ACCEL_CHUNK4: normalize_vector_simd, src/experimental/gpu/common.rs:628-635 -- groups of four, x / norm
//   AVX2 / AVX512: the same normalize in a build with AVX2 / AVX-512 enabled -- the norm from dot_product_avx2 / _avx512
//                 (src/distance.rs:505-526, :636-655), inv = 1 / sqrt, x * inv (scale_avx2 / _avx512 multiply per element too)
constexpr int NZ_ROWS = 256, NZ_DC = 32, NZ_LD = NZ_DC + 1;
__global__ __launch_bounds__(NZ_ROWS) void normalize_rows_kernel(float* __restrict__ v, uint64_t n, uint32_t dim, int arith) {
    __shared__ float tile[NZ_ROWS][NZ_LD];
    __shared__ float s_norm[NZ_ROWS];
    const uint64_t row0 = (uint64_t)blockIdx.x * NZ_ROWS;
    const uint32_t t = threadIdx.x;
    const uint32_t rows = (uint32_t)(n - row0 < (uint64_t)NZ_ROWS ? n - row0 : (uint64_t)NZ_ROWS);
    const bool chunk4 = arith == NEEDLE_ARITH_ACCEL_CHUNK4, avx = arith == NEEDLE_ARITH_AVX2, avx16 = arith == NEEDLE_ARITH_AVX512,
               neon = arith == NEEDLE_ARITH_NEON;
    const uint32_t dim4 = dim & ~3u;
    float sum = chunk4 ? 0.0f : -0.0f;
    AvxAcc vsum;
    AvxAccT<16> vsum16;
    AvxAccT<4> vsum4;
    avx_init(vsum);
    avx_init(vsum16);
    avx_init(vsum4);
    for (uint32_t c0 = 0; c0 < dim; c0 += NZ_DC) {
        const uint32_t w = dim - c0 < (uint32_t)NZ_DC ? dim - c0 : (uint32_t)NZ_DC;
        for (uint32_t e = t; e < (uint32_t)NZ_ROWS * NZ_DC; e += NZ_ROWS) {
            const uint32_t r = e / NZ_DC, c = e % NZ_DC;
            tile[r][c] = (r < rows && c < w) ? v[(row0 + r) * dim + c0 + c] : 0.0f;
        }
        __syncthreads();
        if (t < rows) {
            if (avx) {
                avx_accumulate<AVX_DOT>(vsum, tile[t], tile[t], c0, c0 + w, dim);
            } else if (avx16) {
                avx_accumulate<AVX_DOT>(vsum16, tile[t], tile[t], c0, c0 + w, dim);
            } else if (neon) {
                avx_accumulate<AVX_DOT>(vsum4, tile[t], tile[t], c0, c0 + w, dim);
            } else if (chunk4) {
                uint32_t c = 0;
                for (; c + 4 <= w && c0 + c + 4 <= dim4; c += 4) {
                    const float p0 = tile[t][c] * tile[t][c], p1 = tile[t][c + 1] * tile[t][c + 1];
                    const float p2 = tile[t][c + 2] * tile[t][c + 2], p3 = tile[t][c + 3] * tile[t][c + 3];
                    const float g = ((p0 + p1) + p2) + p3;
                    sum = sum + g;
                }
                for (; c < w; ++c) { const float p = tile[t][c] * tile[t][c]; sum = sum + p; }  // the dim % 4 tail
            } else {
                for (uint32_t c = 0; c < w; ++c) { const float p = tile[t][c] * tile[t][c]; sum = sum + p; }
            }
        }
        __syncthreads();
    }
    s_norm[t] = sqrtf(avx ? avx_finish(vsum, dim) : (avx16 ? avx_finish(vsum16, dim) : (neon ? avx_finish(vsum4, dim) : sum)));
    __syncthreads();
    for (uint64_t e = t; e < (uint64_t)rows * dim; e += NZ_ROWS) {
        const uint32_t r = (uint32_t)(e / dim);
        const float norm = s_norm[r];
        if (norm > 0.0f) {
            float* x = v + row0 * dim + e;
            if (chunk4) {
                *x = *x / norm;
            } else {
                const float inv = 1.0f / norm;
                *x = *x * inv;
            }
        }
    }
}

static needle_hip_status check_arith(needle_arith arith) {
    if (arith != NEEDLE_ARITH_DISTANCE_RS && arith != NEEDLE_ARITH_ACCEL_CHUNK4 && arith != NEEDLE_ARITH_AVX2 &&
        arith != NEEDLE_ARITH_AVX512 && arith != NEEDLE_ARITH_NEON)
        return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "unknown arithmetic %d", (int)arith);
    return NEEDLE_HIP_OK;
}

extern "C" needle_hip_status needle_hip_normalize_device(needle_hip_ctx* c, needle_arith arith, float* d_vectors, uint64_t n,
                                                         uint32_t dim) {
    if (!c) return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "ctx is null");
    ST_TRY(check_arith(arith));
    if (n == 0 || dim == 0) return NEEDLE_HIP_OK;
    if (!d_vectors) return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "vectors is null");
    if ((n + NZ_ROWS - 1) / NZ_ROWS > 0x7fffffffull) return fail(NEEDLE_HIP_ERR_UNSUPPORTED, "too many vectors");
    std::lock_guard g(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    hipLaunchKernelGGL(normalize_rows_kernel, dim3((uint32_t)((n + NZ_ROWS - 1) / NZ_ROWS)), dim3(NZ_ROWS), 0, c->stream, d_vectors,
                       n, dim, (int)arith);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    return NEEDLE_HIP_OK;
}

extern "C" needle_hip_status needle_hip_normalize(needle_hip_ctx* c, needle_arith arith, float* vectors, uint64_t n,
                                                  uint32_t dim) {
    if (!c) return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "ctx is null");
    ST_TRY(check_arith(arith));
    if (n == 0 || dim == 0) return NEEDLE_HIP_OK;
    if (!vectors) return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "vectors is null");
    // one lock for staging, the device form and the copy back: the staging buffer is shared by the context's callers
    std::lock_guard whole(c->mu);
    float* d = nullptr;
    {
        std::lock_guard g(c->mu);
        HIP_TRY(hipSetDevice(c->device));
        HIP_TRY(c->scratch[1].ensure((size_t)n * dim * sizeof(float)));
        d = c->scratch[1].as<float>();
        HIP_TRY(hipMemcpyAsync(d, vectors, (size_t)n * dim * sizeof(float), hipMemcpyHostToDevice, c->stream));
    }
    ST_TRY(needle_hip_normalize_device(c, arith, d, n, dim));
    std::lock_guard g(c->mu);
    HIP_TRY(hipMemcpyAsync(vectors, d, (size_t)n * dim * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return NEEDLE_HIP_OK;
}

// ---------------------------------------------------------------------------
// batch_top_k
// ---------------------------------------------------------------------------
// The k smallest of every score row by (score, index): an 8-bit-per-pass radix descent over the order-preserving
// image of the scores finds the k-th smallest value T and how many of the rows AT T belong to the answer; a gather
// takes everything below T plus the lowest-index ties; the k hits are then ranked.  Workspace (TopkWork) per row:
// a 256-bin histogram, the descent state, one tie count per block and the k selected keys.
namespace nh {

__device__ inline uint32_t topk_ord(float s) { return f2ord(canon_distance(s)); }

// chunk of a score row this block owns: [lo, hi)
__device__ inline void topk_chunk(const TopkWork& w, uint64_t& lo, uint64_t& hi) {
    lo = (uint64_t)blockIdx.x * w.chunk;
    hi = lo + w.chunk < w.n ? lo + w.chunk : w.n;
}

__global__ __launch_bounds__(256) void topk_hist_kernel(TopkWork w, const float* __restrict__ scores, int shift) {
    __shared__ uint32_t h[256];
    const uint32_t r = blockIdx.y;
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t* st = w.state + (uint64_t)r * TOPK_STATE_WORDS;
    const uint32_t prefix = shift == 24 ? 0u : st[0];
    const float* row = scores + (uint64_t)r * w.n;
    uint64_t lo, hi;
    topk_chunk(w, lo, hi);
    for (uint64_t i = lo + threadIdx.x; i < hi; i += 256) {
        const uint32_t o = topk_ord(row[i]);
        const bool match = shift == 24 || (o >> (shift + 8)) == (prefix >> (shift + 8));
        if (match) atomicAdd(&h[(o >> shift) & 255u], 1u);
    }
    __syncthreads();
    const uint32_t v = h[threadIdx.x];
    if (v) atomicAdd(&w.hist[(uint64_t)r * 256 + threadIdx.x], v);
}

// state: [0] prefix, [1] ranks still to descend (after the last pass: ties to take), [2] k_eff, [3] rows below T taken so far
__global__ void topk_pick_kernel(TopkWork w, int shift) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= w.rows) return;
    uint32_t* st = w.state + (uint64_t)r * TOPK_STATE_WORDS;
    uint32_t* h = w.hist + (uint64_t)r * 256;
    if (shift == 24) {
        const uint32_t ke = (uint64_t)w.k < w.n ? w.k : (uint32_t)w.n;
        st[0] = 0; st[1] = ke; st[2] = ke; st[3] = 0;
    }
    if (st[2] != 0) {
        uint32_t rem = st[1], cum = 0;
        for (uint32_t b = 0; b < 256; ++b) {
            const uint32_t c = h[b];
            if (cum + c >= rem) { st[0] |= b << shift; st[1] = rem - cum; break; }
            cum += c;
        }
    }
    for (uint32_t b = 0; b < 256; ++b) h[b] = 0;
}

// rows AT the threshold per block, so that the gather can hand out tie ranks in index order
__global__ __launch_bounds__(256) void topk_tie_count_kernel(TopkWork w, const float* __restrict__ scores) {
    __shared__ uint32_t s_cnt;
    const uint32_t r = blockIdx.y;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    const uint32_t T = w.state[(uint64_t)r * TOPK_STATE_WORDS];
    const float* row = scores + (uint64_t)r * w.n;
    uint64_t lo, hi;
    topk_chunk(w, lo, hi);
    uint32_t c = 0;
    for (uint64_t i = lo + threadIdx.x; i < hi; i += 256) c += topk_ord(row[i]) == T ? 1u : 0u;
    if (c) atomicAdd(&s_cnt, c);
    __syncthreads();
    if (threadIdx.x == 0) w.tie[(uint64_t)r * w.nblk + blockIdx.x] = s_cnt;
}

// exclusive prefix sums of the per-block tie counts of every row (nblk <= TOPK_MAX_BLOCKS)
__global__ __launch_bounds__(256) void topk_tie_scan_kernel(TopkWork w) {
    __shared__ uint32_t s[TOPK_MAX_BLOCKS];
    const uint32_t r = blockIdx.x;
    uint32_t* t = w.tie + (uint64_t)r * w.nblk;
    for (uint32_t i = threadIdx.x; i < w.nblk; i += 256) s[i] = t[i];
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (uint32_t i = 0; i < w.nblk; ++i) { const uint32_t c = s[i]; s[i] = run; run += c; }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < w.nblk; i += 256) t[i] = s[i];
}

__global__ __launch_bounds__(256) void topk_gather_kernel(TopkWork w, const float* __restrict__ scores) {
    __shared__ uint32_t s_w[4];
    const uint32_t r = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t* st = w.state + (uint64_t)r * TOPK_STATE_WORDS;
    const uint32_t T = st[0], take = st[1], ke = st[2];
    if (ke == 0) return;
    const uint32_t below = ke - take;  // rows strictly below T (all taken): slots [0, below); ties: [below, ke)
    uint64_t* sel = w.sel + (uint64_t)r * w.k;
    const float* row = scores + (uint64_t)r * w.n;
    uint64_t lo, hi;
    topk_chunk(w, lo, hi);
    uint32_t tie_run = w.tie[(uint64_t)r * w.nblk + blockIdx.x];
    for (uint64_t i0 = lo; i0 < hi; i0 += 256) {  // block-uniform trip count
        const uint64_t i = i0 + threadIdx.x;
        const bool in = i < hi;
        const uint32_t o = in ? topk_ord(row[i]) : 0xffffffffu;
        const bool is_tie = in && o == T;
        if (in && o < T) {
            const uint32_t slot = atomicAdd(&st[3], 1u);
            if (slot < below) sel[slot] = ((uint64_t)o << 32) | (uint64_t)(uint32_t)i;
        }
        const uint64_t m = __builtin_amdgcn_ballot_w64(is_tie);
        if (lane == 0) s_w[wave] = (uint32_t)__builtin_popcountll(m);
        __syncthreads();
        uint32_t before = 0, all = 0;
        for (uint32_t x = 0; x < 4; ++x) { const uint32_t c = s_w[x]; all += c; if (x < wave) before += c; }
        if (is_tie) {
            const uint32_t rank = tie_run + before + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull));
            if (rank < take) sel[below + rank] = ((uint64_t)o << 32) | (uint64_t)(uint32_t)i;
        }
        tie_run += all;
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void topk_emit_kernel(TopkWork w, const float* __restrict__ scores, uint64_t* __restrict__ out_idx,
                                                        float* __restrict__ out_scores, uint32_t* __restrict__ out_counts) {
    __shared__ uint64_t s[K_MAX];
    const uint32_t r = blockIdx.x;
    const uint32_t ke = w.state[(uint64_t)r * TOPK_STATE_WORDS + 2];
    for (uint32_t i = threadIdx.x; i < ke; i += 256) s[i] = w.sel[(uint64_t)r * w.k + i];
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < ke; i += 256) {
        const uint64_t key = s[i];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < ke; ++j) rank += (s[j] < key) ? 1u : 0u;
        out_idx[(uint64_t)r * w.k + rank] = key & 0xffffffffull;
        out_scores[(uint64_t)r * w.k + rank] = scores[(uint64_t)r * w.n + (key & 0xffffffffull)];  // as stored (-0.0 stays -0.0)
    }
    for (uint32_t j = ke + threadIdx.x; j < w.k; j += 256) {
        out_idx[(uint64_t)r * w.k + j] = ~0ull;
        out_scores[(uint64_t)r * w.k + j] = INFINITY;
    }
    if (threadIdx.x == 0) out_counts[r] = ke;
}

size_t topk_work_bytes(uint32_t rows, uint64_t n, uint32_t k, TopkWork* w) {
    const uint64_t blocks = (n + 4095) / 4096;  // at least 4096 scores per block
    w->rows = rows; w->n = n; w->k = k;
    w->nblk = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(blocks, TOPK_MAX_BLOCKS));
    const uint64_t chunk = (n + w->nblk - 1) / w->nblk;
    w->chunk = (chunk + 255) / 256 * 256;
    w->nblk = (uint32_t)std::max<uint64_t>(1, (n + w->chunk - 1) / std::max<uint64_t>(w->chunk, 1));
    // carve: sel (8-byte aligned) | hist | state | tie
    w->off_sel = 0;
    w->off_hist = (size_t)rows * k * sizeof(uint64_t);
    w->off_state = w->off_hist + (size_t)rows * 256 * sizeof(uint32_t);
    w->off_tie = w->off_state + (size_t)rows * TOPK_STATE_WORDS * sizeof(uint32_t);
    return w->off_tie + (size_t)rows * w->nblk * sizeof(uint32_t);
}

hipError_t launch_batch_top_k(TopkWork w, void* d_work, const float* d_scores, uint64_t* d_out_idx, float* d_out_scores,
                              uint32_t* d_out_counts, hipStream_t stream) {
    unsigned char* base = (unsigned char*)d_work;
    w.sel = (uint64_t*)(base + w.off_sel);
    w.hist = (uint32_t*)(base + w.off_hist);
    w.state = (uint32_t*)(base + w.off_state);
    w.tie = (uint32_t*)(base + w.off_tie);
    // only the histograms must start at zero: the state is initialised by the first pick, tie counts and selected
    // keys are written before they are read
    hipError_t e = hipMemsetAsync(w.hist, 0, (size_t)w.rows * 256 * sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
    const dim3 grid(w.nblk, w.rows), rb((w.rows + 63) / 64);
    for (int shift = 24; shift >= 0; shift -= 8) {
        hipLaunchKernelGGL(topk_hist_kernel, grid, dim3(256), 0, stream, w, d_scores, shift);
        hipLaunchKernelGGL(topk_pick_kernel, rb, dim3(64), 0, stream, w, shift);
    }
    hipLaunchKernelGGL(topk_tie_count_kernel, grid, dim3(256), 0, stream, w, d_scores);
    hipLaunchKernelGGL(topk_tie_scan_kernel, dim3(w.rows), dim3(256), 0, stream, w);
    hipLaunchKernelGGL(topk_gather_kernel, grid, dim3(256), 0, stream, w, d_scores);
    hipLaunchKernelGGL(topk_emit_kernel, dim3(w.rows), dim3(256), 0, stream, w, d_scores, d_out_idx, d_out_scores, d_out_counts);
    return hipGetLastError();
}

needle_hip_status topk_check_args(needle_hip_ctx* c, const float* d_scores, uint32_t rows, uint64_t n, uint32_t k,
                                  const uint64_t* d_out_idx, const float* d_out_scores, const uint32_t* d_out_counts) {
    if (!c) return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "ctx is null");
    if (rows == 0) return NEEDLE_HIP_OK;
    if (!d_out_counts || (k > 0 && (!d_out_idx || !d_out_scores)) || (n > 0 && !d_scores))
        return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "null score or output pointer");
    if (k > (uint32_t)K_MAX) return fail(NEEDLE_HIP_ERR_UNSUPPORTED, "k = %u exceeds %d", k, K_MAX);
    if (n > 0xffffffffull) return fail(NEEDLE_HIP_ERR_UNSUPPORTED, "at most 2^32-1 scores per row");
    if (rows > 65535u) return fail(NEEDLE_HIP_ERR_UNSUPPORTED, "at most 65535 score rows per call");
    return NEEDLE_HIP_OK;
}

}  // namespace nh

extern "C" needle_hip_status needle_hip_batch_top_k_device(needle_hip_ctx* c, const float* d_scores, uint32_t rows, uint64_t n,
                                                           uint32_t k, uint64_t* d_out_idx, float* d_out_scores,
                                                           uint32_t* d_out_counts) {
    ST_TRY(topk_check_args(c, d_scores, rows, n, k, d_out_idx, d_out_scores, d_out_counts));
    if (rows == 0) return NEEDLE_HIP_OK;
    std::lock_guard g(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    if (k == 0 || n == 0) {
        HIP_TRY(launch_fill_empty(d_out_idx, d_out_scores, d_out_counts, rows, k, c->stream));
    } else {
        TopkWork w{};
        const size_t bytes = topk_work_bytes(rows, n, k, &w);
        HIP_TRY(c->scratch[3].ensure(bytes));
        HIP_TRY(launch_batch_top_k(w, c->scratch[3].p, d_scores, d_out_idx, d_out_scores, d_out_counts, c->stream));
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    return NEEDLE_HIP_OK;
}

// The same operator for callers that cannot allocate (tight serving loops, harnesses that forbid hipMalloc; the
// call synchronises its stream, so it is not for use under graph capture): its workspace is this file-scope array, one copy per GPU, held until the process ends.
// The name below is a host-side stand-in; the device address is taken once per context by
// topk_static_setup (called from needle_hip_init with the context's device current) and only that pointer is
// ever handed to kernels and memsets.
constexpr size_t TOPK_STATIC_BYTES = 1u << 20;
static __device__ __align__(256) unsigned char scratch[TOPK_STATIC_BYTES];

needle_hip_status nh::topk_static_setup(needle_hip_ctx* c) {
    void* ptr = nullptr;
    HIP_TRY(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
    if (!ptr) return fail(NEEDLE_HIP_ERR_HIP, "the static top-k workspace has no device address");
    c->topk_static = ptr;
    c->topk_static_bytes = TOPK_STATIC_BYTES;
    return NEEDLE_HIP_OK;
}

extern "C" needle_hip_status needle_hip_batch_top_k_device_noalloc(needle_hip_ctx* c, const float* d_scores, uint32_t rows,
                                                                   uint64_t n, uint32_t k, uint64_t* d_out_idx,
                                                                   float* d_out_scores, uint32_t* d_out_counts) {
    ST_TRY(topk_check_args(c, d_scores, rows, n, k, d_out_idx, d_out_scores, d_out_counts));
    if (rows == 0) return NEEDLE_HIP_OK;
    TopkWork w{};
    const size_t bytes = (k == 0 || n == 0) ? 0 : topk_work_bytes(rows, n, k, &w);
    if (bytes > c->topk_static_bytes)
        return fail(NEEDLE_HIP_ERR_UNSUPPORTED,
                    "top-k of %u rows x %llu scores at k = %u needs %zu bytes of workspace; the allocation-free entry point "
                    "holds %zu (use needle_hip_batch_top_k_device, or fewer rows per call)",
                    rows, (unsigned long long)n, k, bytes, c->topk_static_bytes);
    std::lock_guard g(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    if (bytes == 0) {
        HIP_TRY(launch_fill_empty(d_out_idx, d_out_scores, d_out_counts, rows, k, c->stream));
    } else {
        if (!c->topk_static) return fail(NEEDLE_HIP_ERR_HIP, "the static top-k workspace was not set up");
        // the array keeps the previous call's contents: launch_batch_top_k clears the histograms, everything else
        // is written before it is read
        HIP_TRY(launch_batch_top_k(w, c->topk_static, d_scores, d_out_idx, d_out_scores, d_out_counts, c->stream));
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    return NEEDLE_HIP_OK;
}

extern "C" needle_hip_status needle_hip_batch_top_k(needle_hip_ctx* c, const float* scores, uint32_t rows, uint64_t n, uint32_t k,
                                                    uint64_t* out_idx, float* out_scores, uint32_t* out_counts) {
    ST_TRY(topk_check_args(c, scores, rows, n, k, out_idx, out_scores, out_counts));
    if (rows == 0) return NEEDLE_HIP_OK;
    const size_t kk = k ? k : 1;
    const size_t in_bytes = ((size_t)rows * n * sizeof(float) + 15) & ~(size_t)15;
    // one lock for staging, the device form and the copy back: the staging buffer is shared by the context's callers
    std::lock_guard whole(c->mu);
    unsigned char* d = nullptr;
    {
        std::lock_guard g(c->mu);
        HIP_TRY(hipSetDevice(c->device));
        HIP_TRY(c->scratch[2].ensure(in_bytes + (size_t)rows * kk * 12 + (size_t)rows * 4 + 64));
        d = c->scratch[2].as<unsigned char>();
        if (n > 0) HIP_TRY(hipMemcpyAsync(d, scores, (size_t)rows * n * sizeof(float), hipMemcpyHostToDevice, c->stream));
    }
    uint64_t* d_idx = (uint64_t*)(d + in_bytes);
    float* d_sc = (float*)(d + in_bytes + (size_t)rows * kk * 8);
    uint32_t* d_cnt = (uint32_t*)(d + in_bytes + (size_t)rows * kk * 12);
    ST_TRY(needle_hip_batch_top_k_device(c, (const float*)d, rows, n, k, d_idx, d_sc, d_cnt));
    std::lock_guard g(c->mu);
    if (k > 0) {
        HIP_TRY(hipMemcpyAsync(out_idx, d_idx, (size_t)rows * k * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(out_scores, d_sc, (size_t)rows * k * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    }
    HIP_TRY(hipMemcpyAsync(out_counts, d_cnt, (size_t)rows * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return NEEDLE_HIP_OK;
}

// ---------------------------------------------------------------------------
// distance matrix and multi-query search (src/services/compute/gpu_kernels.rs:223-300)
// ---------------------------------------------------------------------------
static needle_hip_status matrix_check_args(needle_hip_ctx* c, needle_metric metric, needle_arith arith, uint32_t q, uint64_t n,
                                           uint32_t dim) {
    if (!c) return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "ctx is null");
    if (arith != NEEDLE_ARITH_DISTANCE_RS && arith != NEEDLE_ARITH_ACCEL_CHUNK4 && arith != NEEDLE_ARITH_SERVICE &&
        arith != NEEDLE_ARITH_AVX2 && arith != NEEDLE_ARITH_AVX512 && arith != NEEDLE_ARITH_NEON)
        return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "unknown arithmetic %d", (int)arith);
    if (metric < NEEDLE_METRIC_COSINE || metric > NEEDLE_METRIC_CHEBYSHEV)
        return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "unknown metric %d", (int)metric);
    if (arith != NEEDLE_ARITH_DISTANCE_RS && arith != NEEDLE_ARITH_AVX2 && arith != NEEDLE_ARITH_AVX512 && arith != NEEDLE_ARITH_NEON &&
        metric != NEEDLE_METRIC_COSINE &&
        metric != NEEDLE_METRIC_EUCLIDEAN && metric != NEEDLE_METRIC_DOT_PRODUCT)
        return fail(NEEDLE_HIP_ERR_UNSUPPORTED, "this arithmetic covers cosine, euclidean and dot product only");
    if (q > 0 && n > 0 && dim == 0) return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "dim is 0");
    if (q > 65535u * 16u) return fail(NEEDLE_HIP_ERR_UNSUPPORTED, "at most 1048560 queries per call");
    if ((n + 63) / 64 > 0x7fffffffull) return fail(NEEDLE_HIP_ERR_UNSUPPORTED, "too many vectors");
    return NEEDLE_HIP_OK;
}

extern "C" needle_hip_status needle_hip_distance_matrix_device(needle_hip_ctx* c, needle_metric metric, needle_arith arith,
                                                               const float* d_queries, uint32_t q, const float* d_base,
                                                               uint64_t n, uint32_t dim, float* d_out) {
    ST_TRY(matrix_check_args(c, metric, arith, q, n, dim));
    if (q == 0 || n == 0) return NEEDLE_HIP_OK;  // an empty matrix
    if (!d_queries || !d_base || !d_out) return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "null argument");
    std::lock_guard g(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(launch_distance_matrix(d_queries, q, d_base, n, dim, (int)metric, (int)arith, d_out, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return NEEDLE_HIP_OK;
}

extern "C" needle_hip_status needle_hip_distance_matrix(needle_hip_ctx* c, needle_metric metric, needle_arith arith,
                                                        const float* queries, uint32_t q, const float* base, uint64_t n,
                                                        uint32_t dim, float* out) {
    ST_TRY(matrix_check_args(c, metric, arith, q, n, dim));
    if (q == 0 || n == 0) return NEEDLE_HIP_OK;
    if (!queries || !base || !out) return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "null argument");
    const size_t qf = (size_t)q * dim, bf = (size_t)n * dim, of = (size_t)q * n;
    // one lock for staging, the device form and the copy back: the staging buffer is shared by the context's callers
    std::lock_guard whole(c->mu);
    float* d = nullptr;
    {
        std::lock_guard g(c->mu);
        HIP_TRY(hipSetDevice(c->device));
        HIP_TRY(c->scratch[4].ensure((qf + bf + of) * sizeof(float)));
        d = c->scratch[4].as<float>();
        HIP_TRY(hipMemcpyAsync(d, queries, qf * sizeof(float), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(d + qf, base, bf * sizeof(float), hipMemcpyHostToDevice, c->stream));
    }
    ST_TRY(needle_hip_distance_matrix_device(c, metric, arith, d, q, d + qf, n, dim, d + qf + bf));
    std::lock_guard g(c->mu);
    HIP_TRY(hipMemcpyAsync(out, d + qf + bf, of * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return NEEDLE_HIP_OK;
}

extern "C" needle_hip_status needle_hip_multi_query_search_device(needle_hip_ctx* c, needle_metric metric, needle_arith arith,
                                                                  const float* d_queries, uint32_t q, const float* d_base,
                                                                  uint64_t n, uint32_t dim, uint32_t k, uint64_t* d_out_idx,
                                                                  float* d_out_dists, uint32_t* d_out_counts) {
    ST_TRY(matrix_check_args(c, metric, arith, q, n, dim));
    if (q == 0) return NEEDLE_HIP_OK;
    if (!d_out_counts || (k > 0 && (!d_out_idx || !d_out_dists)) || (n > 0 && (!d_queries || !d_base)))
        return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "null argument");
    if (k > (uint32_t)K_MAX) return fail(NEEDLE_HIP_ERR_UNSUPPORTED, "k = %u exceeds %d", k, K_MAX);
    if (n > 0xffffffffull) return fail(NEEDLE_HIP_ERR_UNSUPPORTED, "at most 2^32-1 vectors");
    std::lock_guard g(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    if (k == 0 || n == 0) {
        HIP_TRY(launch_fill_empty(d_out_idx, d_out_dists, d_out_counts, q, k, c->stream));
    } else {
        // a block of queries at a time: its distance rows stay within 1 GiB (and within the selection's 65535 rows)
        uint64_t blk = (1ull << 30) / (n * sizeof(float));
        blk = std::max<uint64_t>(16, std::min<uint64_t>(blk / 16 * 16, 65520));
        const uint32_t rows_max = (uint32_t)std::min<uint64_t>(blk, q);
        HIP_TRY(c->scratch[5].ensure((size_t)rows_max * n * sizeof(float)));
        TopkWork w{};
        HIP_TRY(c->scratch[3].ensure(topk_work_bytes(rows_max, n, k, &w)));
        for (uint32_t q0 = 0; q0 < q; q0 += rows_max) {
            const uint32_t rows = std::min(rows_max, q - q0);
            HIP_TRY(launch_distance_matrix(d_queries + (size_t)q0 * dim, rows, d_base, n, dim, (int)metric, (int)arith,
                                           c->scratch[5].as<float>(), c->stream));
            (void)topk_work_bytes(rows, n, k, &w);
            HIP_TRY(launch_batch_top_k(w, c->scratch[3].p, c->scratch[5].as<float>(), d_out_idx + (size_t)q0 * k,
                                       d_out_dists + (size_t)q0 * k, d_out_counts + q0, c->stream));
        }
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    return NEEDLE_HIP_OK;
}

extern "C" needle_hip_status needle_hip_multi_query_search(needle_hip_ctx* c, needle_metric metric, needle_arith arith,
                                                           const float* queries, uint32_t q, const float* base, uint64_t n,
                                                           uint32_t dim, uint32_t k, uint64_t* out_idx, float* out_dists,
                                                           uint32_t* out_counts) {
    ST_TRY(matrix_check_args(c, metric, arith, q, n, dim));
    if (q == 0) return NEEDLE_HIP_OK;
    if (!out_counts || (k > 0 && (!out_idx || !out_dists)) || (n > 0 && (!queries || !base)))
        return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "null argument");
    const size_t kk = k ? k : 1;
    const size_t qb = ((size_t)q * dim * sizeof(float) + 15) & ~(size_t)15, bb = ((size_t)n * dim * sizeof(float) + 15) & ~(size_t)15;
    // one lock for staging, the device form and the copy back: the staging buffer is shared by the context's callers
    std::lock_guard whole(c->mu);
    unsigned char* d = nullptr;
    {
        std::lock_guard g(c->mu);
        HIP_TRY(hipSetDevice(c->device));
        HIP_TRY(c->scratch[4].ensure(qb + bb + (size_t)q * kk * 12 + (size_t)q * 4 + 64));
        d = c->scratch[4].as<unsigned char>();
        if (n > 0) {
            HIP_TRY(hipMemcpyAsync(d, queries, (size_t)q * dim * sizeof(float), hipMemcpyHostToDevice, c->stream));
            HIP_TRY(hipMemcpyAsync(d + qb, base, (size_t)n * dim * sizeof(float), hipMemcpyHostToDevice, c->stream));
        }
    }
    uint64_t* d_idx = (uint64_t*)(d + qb + bb);
    float* d_ds = (float*)(d + qb + bb + (size_t)q * kk * 8);
    uint32_t* d_cnt = (uint32_t*)(d + qb + bb + (size_t)q * kk * 12);
    ST_TRY(needle_hip_multi_query_search_device(c, metric, arith, (const float*)d, q, (const float*)(d + qb), n, dim, k, d_idx, d_ds,
                                                d_cnt));
    std::lock_guard g(c->mu);
    if (k > 0) {
        HIP_TRY(hipMemcpyAsync(out_idx, d_idx, (size_t)q * k * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(out_dists, d_ds, (size_t)q * k * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    }
    HIP_TRY(hipMemcpyAsync(out_counts, d_cnt, (size_t)q * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return NEEDLE_HIP_OK;
}

// ---------------------------------------------------------------------------
// merge_topk, host pointers
// ---------------------------------------------------------------------------
extern "C" needle_hip_status needle_hip_merge_topk(needle_hip_ctx* c, const float* dists, const uint64_t* ids,
                                                   const uint32_t* counts, uint32_t lists, uint32_t q, uint32_t per, uint32_t k,
                                                   uint64_t* out_ids, float* out_dists, uint32_t* out_counts) {
    if (!c) return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "ctx is null");
    if (q == 0) return NEEDLE_HIP_OK;
    if (!counts || !out_counts || lists == 0 || (per > 0 && (!dists || !ids)) || (k > 0 && (!out_ids || !out_dists)))
        return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "null argument");
    const size_t kk = k ? k : 1, pp = per ? per : 1;
    const size_t n_in = (size_t)lists * q * pp;
    // one lock for staging, the device form and the copy back: the staging buffer is shared by the context's callers
    std::lock_guard whole(c->mu);
    unsigned char* d = nullptr;
    {
        std::lock_guard g(c->mu);
        HIP_TRY(hipSetDevice(c->device));
        HIP_TRY(c->scratch[2].ensure(n_in * 12 + (size_t)lists * q * 4 + (size_t)q * kk * 12 + (size_t)q * 4 + 64));
        d = c->scratch[2].as<unsigned char>();
    }
    uint64_t* d_ids = (uint64_t*)d;
    uint64_t* d_oi = d_ids + n_in;
    float* d_d = (float*)(d_oi + (size_t)q * kk);
    float* d_od = d_d + n_in;
    uint32_t* d_c = (uint32_t*)(d_od + (size_t)q * kk);
    uint32_t* d_oc = d_c + (size_t)lists * q;
    {
        std::lock_guard g(c->mu);
        if (per > 0) {
            HIP_TRY(hipMemcpyAsync(d_ids, ids, n_in * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
            HIP_TRY(hipMemcpyAsync(d_d, dists, n_in * sizeof(float), hipMemcpyHostToDevice, c->stream));
        }
        HIP_TRY(hipMemcpyAsync(d_c, counts, (size_t)lists * q * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    }
    ST_TRY(needle_hip_merge_topk_device(c, d_d, d_ids, d_c, lists, q, per, k, d_oi, d_od, d_oc));
    std::lock_guard g(c->mu);
    if (k > 0) {
        HIP_TRY(hipMemcpyAsync(out_ids, d_oi, (size_t)q * k * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(out_dists, d_od, (size_t)q * k * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    }
    HIP_TRY(hipMemcpyAsync(out_counts, d_oc, (size_t)q * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return NEEDLE_HIP_OK;
}
