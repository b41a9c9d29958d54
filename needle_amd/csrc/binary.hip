// Binary-quantized index: 1 bit per dimension, Hamming scan (C ABI: include/needle_hip.h, "binary codes").
//
// Reference: BinaryQuantizer, src/indexing/quantization.rs:536-594 (quantize: bit i of byte i/8 set iff
// v[i] > threshold[i]; hamming_distance: sum of popcounts of the XORed bytes) and its search call site,
// src/indexing/compression_advisor.rs:658-671: all distances, STABLE sort by distance, first k  ==  order
// (distance, row index).
//
// The distance is a small integer (0 .. dim), so the k nearest are found exactly with a histogram instead of
// candidate lists:
//   A  scan: per query a histogram of the distances of all live rows
//   B  pick: T = the distance at which the cumulative count reaches k; every row below T is a hit, of the rows
//      AT T the `need` lowest row ids are
//   C  scan: rows at T per (query, run of consecutive rows), then exclusive sums over the runs
//   D  scan: collect rows below T, and rows at T whose rank among the ties (run prefix + position) < need
//   E  emit: the k collected (distance, id) keys of a query, sorted
// When no query has more than BIN_TIE_CAP rows at its T (the histogram says so after B: the usual case for k << n),
// C is skipped: one scan collects the rows below T and ALL rows at T, and E keeps the `need` lowest ids of those.
// Codes are stored word-major ([word][row], 64-bit words), so a wave reads 64 consecutive rows' words with one
// coalesced load; a lane keeps QT distance counters and every loaded word is XORed with the QT query words of
// the workgroup's query tile (LDS broadcast reads).  Integer work throughout.
#include "nh_api.h"

using namespace nh;

namespace {

struct BinParams {
    const uint64_t* codes;   // [W][ld]
    const uint64_t* qcodes;  // [q][W]
    const uint32_t *valid, *filter;
    uint32_t n, ld, W, q, k, hb;  // hb = dim + 1 histogram bins
    uint32_t rows_per_wave, units;
    uint32_t* hist;    // [q][hb]
    uint32_t* thr;     // [q]
    uint32_t* need;    // [q]
    uint32_t* tiecnt;  // [q][units]; exclusive sums after pass C
    uint32_t* outn;    // [q]
    uint64_t* keys;    // [q][k]: distance << 32 | row
    uint32_t* tien;    // [q] rows at T collected by the single-scan path
    uint32_t* tierow;  // [q][BIN_TIE_CAP] their row ids, unordered
    uint32_t* max_ties;  // [1] most rows at T of any query of the batch
};

constexpr uint32_t BIN_TIE_CAP = 1024;

enum { PASS_HIST = 0, PASS_TIES = 1, PASS_COLLECT = 2, PASS_COLLECT_ALL = 3 };

template <int PASS, int QT>
__global__ __launch_bounds__(256) void bin_scan_kernel(BinParams p) {
    extern __shared__ __attribute__((aligned(16))) uint64_t bin_lds[];
    uint64_t* qc = bin_lds;                             // [W][QT]
    uint32_t* hist = (uint32_t*)(bin_lds + p.W * QT);   // PASS_HIST: [QT][hb]
    __shared__ uint32_t thr_s[QT], need_s[QT];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t q0 = blockIdx.y * QT, nq = min((uint32_t)QT, p.q - q0);
    for (uint32_t i = tid; i < p.W * QT; i += 256) {
        const uint32_t w = i / QT, qi = i % QT;
        qc[i] = qi < nq ? p.qcodes[(uint64_t)(q0 + qi) * p.W + w] : 0ull;
    }
    if (PASS == PASS_HIST)
        for (uint32_t i = tid; i < QT * p.hb; i += 256) hist[i] = 0u;
    if (PASS != PASS_HIST && tid < QT) {
        thr_s[tid] = tid < nq ? p.thr[q0 + tid] : 0u;
        need_s[tid] = tid < nq ? p.need[q0 + tid] : 0u;
    }
    __syncthreads();

    const uint32_t unit = blockIdx.x * 4 + wave;  // a run of rows_per_wave consecutive rows
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    uint32_t ties[QT];  // PASS_TIES: ties counted; PASS_COLLECT: rank of the run's next tie (wave-uniform)
#pragma unroll
    for (int qi = 0; qi < QT; ++qi)
        ties[qi] = (PASS == PASS_COLLECT && unit < p.units && (uint32_t)qi < nq) ? p.tiecnt[(uint64_t)(q0 + qi) * p.units + unit] : 0u;
    if (unit < p.units) {  // wave-uniform
        const uint32_t r_begin = unit * p.rows_per_wave, r_end = min(p.n, r_begin + p.rows_per_wave);
        for (uint32_t r0 = r_begin; r0 < r_end; r0 += 64) {
            const uint32_t row = r0 + lane;
            bool live = row < r_end;
            if (live && p.valid) live = bit_set(p.valid, row);
            if (live && p.filter) live = bit_set(p.filter, row);
            uint32_t d[QT];
#pragma unroll
            for (int qi = 0; qi < QT; ++qi) d[qi] = 0u;
            for (uint32_t w = 0; w < p.W; ++w) {
                const uint64_t rw = p.codes[(uint64_t)w * p.ld + min(row, p.ld - 1)];
                const uint64_t* qw = qc + w * QT;
#pragma unroll
                for (int qi = 0; qi < QT; ++qi) d[qi] += (uint32_t)__builtin_popcountll(rw ^ qw[qi]);
            }
            if (PASS == PASS_HIST) {
                if (live) {
#pragma unroll
                    for (int qi = 0; qi < QT; ++qi)
                        if ((uint32_t)qi < nq) atomicAdd(&hist[qi * p.hb + d[qi]], 1u);
                }
            } else {
#pragma unroll
                for (int qi = 0; qi < QT; ++qi) {
                    const uint32_t T = thr_s[qi];
                    const bool tie = live && (uint32_t)qi < nq && d[qi] == T;
                    if (PASS == PASS_COLLECT_ALL) {  // no ranking here: every row at T is kept for the emit step
                        if (live && (uint32_t)qi < nq && d[qi] < T) {
                            const uint32_t at = atomicAdd(&p.outn[q0 + qi], 1u);
                            if (at < p.k) p.keys[(uint64_t)(q0 + qi) * p.k + at] = ((uint64_t)d[qi] << 32) | row;
                        } else if (tie) {
                            const uint32_t at = atomicAdd(&p.tien[q0 + qi], 1u);
                            if (at < BIN_TIE_CAP) p.tierow[(uint64_t)(q0 + qi) * BIN_TIE_CAP + at] = row;
                        }
                        continue;
                    }
                    const uint64_t tm = __builtin_amdgcn_ballot_w64(tie);
                    if (PASS == PASS_COLLECT) {
                        const bool take = live && (uint32_t)qi < nq && (d[qi] < T || (tie && ties[qi] + (uint32_t)__builtin_popcountll(tm & lt_mask) < need_s[qi]));
                        if (take) {
                            const uint32_t at = atomicAdd(&p.outn[q0 + qi], 1u);
                            if (at < p.k) p.keys[(uint64_t)(q0 + qi) * p.k + at] = ((uint64_t)d[qi] << 32) | row;
                        }
                    }
                    ties[qi] += (uint32_t)__builtin_popcountll(tm);
                }
            }
        }
    }
    if (PASS == PASS_TIES && unit < p.units && lane == 0) {
#pragma unroll
        for (int qi = 0; qi < QT; ++qi)
            if ((uint32_t)qi < nq) p.tiecnt[(uint64_t)(q0 + qi) * p.units + unit] = ties[qi];
    }
    if (PASS == PASS_HIST) {
        __syncthreads();
        for (uint32_t i = tid; i < nq * p.hb; i += 256) {
            const uint32_t v = hist[i];
            if (v) atomicAdd(&p.hist[(uint64_t)(q0 + i / p.hb) * p.hb + i % p.hb], v);
        }
    }
}

// pass B: one workgroup per query
__global__ __launch_bounds__(256) void bin_pick_kernel(BinParams p, uint32_t* out_counts) {
    __shared__ uint32_t part[256];
    const uint32_t q = blockIdx.x, t = threadIdx.x;
    const uint32_t* h = p.hist + (uint64_t)q * p.hb;
    const uint32_t per = (p.hb + 255) / 256, b = min(p.hb, t * per), e = min(p.hb, b + per);
    uint32_t s = 0;
    for (uint32_t i = b; i < e; ++i) s += h[i];
    part[t] = s;
    __syncthreads();
    if (t != 0) return;
    uint32_t total = 0;
    for (uint32_t i = 0; i < 256; ++i) total += part[i];
    uint32_t T = p.hb, need = 0;  // fewer than k live rows: every one of them is a hit (d < hb always)
    if (total > p.k) {
        uint32_t cum = 0, seg = 0;
        while (cum + part[seg] < p.k) cum += part[seg++];
        uint32_t i = min(p.hb, seg * per);
        while (cum + h[i] < p.k) cum += h[i++];
        T = i;
        need = p.k - cum;
    }
    p.thr[q] = T;
    p.need[q] = need;
    out_counts[q] = min(total, p.k);
    if (need) atomicMax(p.max_ties, h[T]);
}

// exclusive sums of a query's tie counts over the runs, in place; one workgroup per query
__global__ __launch_bounds__(256) void bin_tie_prefix_kernel(BinParams p) {
    __shared__ uint32_t part[256];
    const uint32_t t = threadIdx.x;
    uint32_t* c = p.tiecnt + (uint64_t)blockIdx.x * p.units;
    const uint32_t per = (p.units + 255) / 256, b = min(p.units, t * per), e = min(p.units, b + per);
    uint32_t s = 0;
    for (uint32_t i = b; i < e; ++i) s += c[i];
    part[t] = s;
    __syncthreads();
    for (uint32_t off = 1; off < 256; off <<= 1) {
        const uint32_t v = t >= off ? part[t - off] : 0u;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    uint32_t acc = t ? part[t - 1] : 0u;
    for (uint32_t i = b; i < e; ++i) {
        const uint32_t v = c[i];
        c[i] = acc;
        acc += v;
    }
}

// pass E: rank the collected keys of a query (k <= BIN_K_MAX) and write the hits
constexpr uint32_t BIN_K_MAX = 1024;

__global__ __launch_bounds__(256) void bin_emit_kernel(BinParams p, const uint32_t* counts, uint64_t* out_ids, float* out_dists,
                                                       uint32_t out_ld, int ties_unranked) {  // out_ld >= p.k: length of an output row
    __shared__ uint64_t sk[BIN_K_MAX];
    const uint32_t q = blockIdx.x, t = threadIdx.x, c = min(counts[q], p.k);
    if (ties_unranked) {
        // single-scan path: keys holds the rows below T; of the rows at T the `need` lowest ids complete the k
        __shared__ uint32_t tr[BIN_TIE_CAP];
        const uint32_t below = min(p.outn[q], p.k), tn = min(p.tien[q], BIN_TIE_CAP), need = p.need[q], T = p.thr[q];
        for (uint32_t i = t; i < below; i += 256) sk[i] = p.keys[(uint64_t)q * p.k + i];
        for (uint32_t i = t; i < tn; i += 256) tr[i] = p.tierow[(uint64_t)q * BIN_TIE_CAP + i];
        __syncthreads();
        for (uint32_t i = t; i < tn; i += 256) {
            const uint32_t row = tr[i];
            uint32_t rank = 0;
            for (uint32_t j = 0; j < tn; ++j) rank += tr[j] < row ? 1u : 0u;
            if (rank < need && below + rank < p.k) sk[below + rank] = ((uint64_t)T << 32) | row;
        }
    } else {
        for (uint32_t i = t; i < c; i += 256) sk[i] = p.keys[(uint64_t)q * p.k + i];
    }
    __syncthreads();
    for (uint32_t i = t; i < c; i += 256) {
        const uint64_t key = sk[i];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < c; ++j) rank += sk[j] < key ? 1u : 0u;
        out_ids[(uint64_t)q * out_ld + rank] = key & 0xffffffffull;
        out_dists[(uint64_t)q * out_ld + rank] = (float)(uint32_t)(key >> 32);  // <= dim < 2^24: exact
    }
    for (uint32_t i = c + t; i < out_ld; i += 256) {
        out_ids[(uint64_t)q * out_ld + i] = ~0ull;
        out_dists[(uint64_t)q * out_ld + i] = INFINITY;
    }
}

// quantize: one wave per (vector, 64-bit word); lane = bit.  Bit i of byte i/8 (quantization.rs:578-581) is bit
// i % 64 of little-endian word i / 64.
__global__ __launch_bounds__(64) void bin_quantize_kernel(const float* __restrict__ vecs, const float* __restrict__ thr, uint32_t dim,
                                                          uint32_t W, uint64_t* __restrict__ out, uint64_t row_stride, uint64_t word_stride) {
    const uint32_t v = blockIdx.x, w = blockIdx.y, i = w * 64 + threadIdx.x;
    const bool bit = i < dim && vecs[(uint64_t)v * dim + i] > thr[i];
    const uint64_t m = __builtin_amdgcn_ballot_w64(bit);
    if (threadIdx.x == 0) out[(uint64_t)v * row_stride + (uint64_t)w * word_stride] = m;
}

// row-major bytes [n][nbytes] -> word-major words [W][ld]
__global__ __launch_bounds__(256) void bin_transpose_kernel(const uint8_t* __restrict__ src, uint32_t n, uint32_t nbytes, uint32_t W,
                                                            uint32_t ld, uint64_t* __restrict__ dst) {
    const uint32_t row = blockIdx.x * 256 + threadIdx.x, w = blockIdx.y;
    if (row >= ld) return;
    uint64_t v = 0;
    if (row < n)
        for (uint32_t b = 0; b < 8; ++b) {
            const uint32_t at = w * 8 + b;
            if (at < nbytes) v |= (uint64_t)src[(uint64_t)row * nbytes + at] << (8 * b);
        }
    dst[(uint64_t)w * ld + row] = v;
}

template <int PASS>
hipError_t launch_scan(const BinParams& p, uint32_t qt, hipStream_t stream) {
    const dim3 grid((p.units + 3) / 4, (p.q + qt - 1) / qt);
    const size_t lds = (size_t)p.W * qt * sizeof(uint64_t) + (PASS == PASS_HIST ? (size_t)qt * p.hb * sizeof(uint32_t) : 0);
    switch (qt) {
    case 16: hipLaunchKernelGGL((bin_scan_kernel<PASS, 16>), grid, dim3(256), lds, stream, p); break;
    case 4: hipLaunchKernelGGL((bin_scan_kernel<PASS, 4>), grid, dim3(256), lds, stream, p); break;
    default: hipLaunchKernelGGL((bin_scan_kernel<PASS, 1>), grid, dim3(256), lds, stream, p); break;
    }
    return hipGetLastError();
}

}  // namespace

extern "C" needle_hip_status needle_hip_binary_build(needle_hip_ctx* c, const uint8_t* codes, uint64_t n, uint32_t dim,
                                                     const float* thresholds, const uint32_t* valid_bitmask,
                                                     needle_hip_index** out) {
    if (!c || !out) return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    if (dim == 0 || !thresholds) return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "need the thresholds of a trained quantizer");
    if (dim > 12000) return fail(NEEDLE_HIP_ERR_UNSUPPORTED, "%u dimensions: the distance histogram does not fit on chip", dim);
    if (n > 0 && !codes) return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "codes is null");
    if (n >= 0xfffffff0ull - 64) return fail(NEEDLE_HIP_ERR_UNSUPPORTED, "too many rows");
    HIP_TRY(hipSetDevice(c->device));
    needle_hip_index* ix = index_new();
    if (!ix) return fail(NEEDLE_HIP_ERR_OUT_OF_MEMORY, "host allocation failed");
    ix->ctx = c;
    ix->kind = KIND_BINARY;
    ix->n = n;
    ix->dim = dim;
    ix->n_pad = round_up((uint32_t)n, 64);
    ix->metric = NEEDLE_METRIC_HAMMING;
    ix->bin_words = (dim + 63) / 64;
    const uint32_t nbytes = (dim + 7) / 8;  // dimensions.div_ceil(8), quantization.rs:575
    uint8_t* d_src = nullptr;
    hipError_t e = hipMalloc((void**)&ix->d_sq_min, (size_t)dim * sizeof(float));  // the thresholds
    if (e == hipSuccess) e = hipMemcpyAsync(ix->d_sq_min, thresholds, (size_t)dim * sizeof(float), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess && n > 0) {
        e = hipMalloc((void**)&ix->d_codes, (size_t)ix->bin_words * ix->n_pad * sizeof(uint64_t));
        if (e == hipSuccess) e = hipMalloc((void**)&d_src, (size_t)n * nbytes);
        if (e == hipSuccess) e = hipMemcpyAsync(d_src, codes, (size_t)n * nbytes, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(bin_transpose_kernel, dim3((ix->n_pad + 255) / 256, ix->bin_words), dim3(256), 0, c->stream, d_src,
                               (uint32_t)n, nbytes, ix->bin_words, ix->n_pad, (uint64_t*)ix->d_codes);
            e = hipGetLastError();
        }
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (d_src) (void)hipFree(d_src);
    needle_hip_status st = NEEDLE_HIP_OK;
    if (e != hipSuccess)
        st = fail(e == hipErrorOutOfMemory ? NEEDLE_HIP_ERR_OUT_OF_MEMORY : NEEDLE_HIP_ERR_HIP, "HIP error %d (%s) uploading codes",
                  (int)e, hipGetErrorString(e));
    if (st == NEEDLE_HIP_OK) st = upload_mask(ix, valid_bitmask);
    if (st != NEEDLE_HIP_OK) { index_free(ix); return st; }
    *out = ix;
    return NEEDLE_HIP_OK;
}

extern "C" needle_hip_status needle_hip_binary_quantize(needle_hip_ctx* c, const float* thresholds, const float* vectors,
                                                        uint64_t n, uint32_t dim, uint8_t* out_codes) {
    if (!c) return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "null argument");
    if (n == 0) return NEEDLE_HIP_OK;
    if (dim == 0 || !thresholds || !vectors || !out_codes) return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "null argument");
    if (n > 0x7fffffffull) return fail(NEEDLE_HIP_ERR_UNSUPPORTED, "too many rows in one call");
    std::lock_guard g(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    const uint32_t W = (dim + 63) / 64, nbytes = (dim + 7) / 8;
    HIP_TRY(c->scratch[0].ensure((size_t)n * dim * sizeof(float)));
    HIP_TRY(c->scratch[1].ensure((size_t)dim * sizeof(float)));
    HIP_TRY(c->scratch[2].ensure((size_t)n * W * sizeof(uint64_t)));
    HIP_TRY(hipMemcpyAsync(c->scratch[0].p, vectors, (size_t)n * dim * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->scratch[1].p, thresholds, (size_t)dim * sizeof(float), hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(bin_quantize_kernel, dim3((uint32_t)n, W), dim3(64), 0, c->stream, c->scratch[0].as<float>(),
                       c->scratch[1].as<float>(), dim, W, c->scratch[2].as<uint64_t>(), (uint64_t)W, 1ull);
    HIP_TRY(hipGetLastError());
    std::vector<uint64_t> words((size_t)n * W);
    HIP_TRY(hipMemcpyAsync(words.data(), c->scratch[2].p, words.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (uint64_t i = 0; i < n; ++i)  // little-endian words -> the reference's byte vector
        memcpy(out_codes + i * nbytes, (const uint8_t*)(words.data() + i * W), nbytes);
    return NEEDLE_HIP_OK;
}

needle_hip_status nh::binary_search_batch_device(needle_hip_index* ix, const float* d_queries, uint32_t q, uint32_t k,
                                                 const uint32_t* d_filter, uint64_t* d_out_ids, float* d_out_dists,
                                                 uint32_t* d_out_counts) {
    needle_hip_ctx* c = ix->ctx;
    const uint32_t n = (uint32_t)ix->n;
    const uint32_t k_out = k;            // output row length
    const uint32_t k_eff = std::min(k, n);  // hits wanted
    if (k_eff > BIN_K_MAX) return fail(NEEDLE_HIP_ERR_UNSUPPORTED, "k = %u exceeds %u for the binary scan", k_eff, BIN_K_MAX);
    BinParams p{};
    p.W = ix->bin_words;
    p.hb = ix->dim + 1;
    const uint32_t qt = p.hb <= 900 ? 16u : (p.hb <= 3600 ? 4u : 1u);
    const uint32_t qtiles = (q + qt - 1) / qt;
    // a wave scans a run of consecutive rows: long enough that the per-workgroup histogram flush stays small,
    // short enough that the grid fills the chip
    uint64_t rpw = ((uint64_t)n * qtiles) / (4ull * 2048ull);
    rpw = std::min<uint64_t>(std::max<uint64_t>(rpw, 64), 16384);
    p.rows_per_wave = round_up((uint32_t)rpw, 64);
    p.units = (n + p.rows_per_wave - 1) / p.rows_per_wave;
    p.codes = (const uint64_t*)ix->d_codes;
    p.valid = ix->d_valid;
    p.filter = d_filter;
    p.n = n; p.ld = ix->n_pad; p.q = q; p.k = k_eff;

    const size_t words = (size_t)q * p.W * 2 + (size_t)q * p.hb + 5 * (size_t)q + (size_t)q * p.units + (size_t)q * k_eff * 2 +
                         (size_t)q * BIN_TIE_CAP + 64;
    HIP_TRY(ix->b_work.ensure(words * sizeof(uint32_t)));
    uint64_t* w64 = ix->b_work.as<uint64_t>();
    uint64_t* qcodes = w64;                          // [q][W]
    p.keys = qcodes + (size_t)q * p.W;               // [q][k_eff]
    uint32_t* w32 = (uint32_t*)(p.keys + (size_t)q * k_eff);
    p.hist = w32;                                    // [q][hb]
    p.thr = p.hist + (size_t)q * p.hb;
    p.need = p.thr + q;
    p.outn = p.need + q;
    p.tien = p.outn + q;
    p.max_ties = p.tien + q;                         // one word, zeroed with the counters (a whole [q] row is reserved)
    p.tiecnt = p.max_ties + q;                       // [q][units]
    p.tierow = p.tiecnt + (size_t)q * p.units;       // [q][BIN_TIE_CAP]
    p.qcodes = qcodes;
    HIP_TRY(hipMemsetAsync(p.hist, 0, ((size_t)q * p.hb + 5 * (size_t)q) * sizeof(uint32_t), c->stream));
    hipLaunchKernelGGL(bin_quantize_kernel, dim3(q, p.W), dim3(64), 0, c->stream, d_queries, ix->d_sq_min, ix->dim, p.W, qcodes,
                       (uint64_t)p.W, 1ull);
    HIP_TRY(hipGetLastError());
    if (c->profiling) HIP_TRY(hipEventRecord(c->ev[0], c->stream));
    HIP_TRY(launch_scan<PASS_HIST>(p, qt, c->stream));
    hipLaunchKernelGGL(bin_pick_kernel, dim3(q), dim3(256), 0, c->stream, p, d_out_counts);
    HIP_TRY(hipGetLastError());
    // few rows at the threshold distance for every query: one scan collects them all and the emit step ranks them
    uint32_t max_ties = 0;
    HIP_TRY(hipMemcpyAsync(&max_ties, p.max_ties, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    const bool single_scan = max_ties <= BIN_TIE_CAP && !c->tune.binary_two_scans;
    if (single_scan) {
        HIP_TRY(launch_scan<PASS_COLLECT_ALL>(p, qt, c->stream));
    } else {
        HIP_TRY(launch_scan<PASS_TIES>(p, qt, c->stream));
        hipLaunchKernelGGL(bin_tie_prefix_kernel, dim3(q), dim3(256), 0, c->stream, p);
        HIP_TRY(hipGetLastError());
        HIP_TRY(launch_scan<PASS_COLLECT>(p, qt, c->stream));
    }
    if (c->profiling) HIP_TRY(hipEventRecord(c->ev[1], c->stream));
    hipLaunchKernelGGL(bin_emit_kernel, dim3(q), dim3(256), 0, c->stream, p, d_out_counts, d_out_ids, d_out_dists, k_out, single_scan ? 1 : 0);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->profiling) {
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
        std::lock_guard g(c->mu);
        c->m.scan_launches += 1;
        c->m.scan_ms_total += ms;
        c->m.last_scan_ms = ms;
    }
    {
        std::lock_guard g(c->mu);
        c->m.binary_code_scans += single_scan ? 2 : 3;
    }
    return NEEDLE_HIP_OK;
}
