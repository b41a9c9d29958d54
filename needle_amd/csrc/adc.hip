// Quantized (ADC) scans.
//
// Product quantizer: the reference's table scan
// (ProductQuantizer::{compute_distance_table, distance_with_table},
// src/indexing/quantization.rs:468-503) evaluated directly in its own order, so
// the scan's distances ARE the reported distances: no re-scoring, no bound
// check.
//
// Q * N * m table lookups are the whole cost, and a lookup is an LDS gather (lane = row, the
// row's code picks the table entry).  What the kernel is built around:
//   * the tables of a group of 8 queries are interleaved, [subspace][half][code][4 queries]: one
//     16-byte ds_read_b128 fetches the entries of 4 queries for one code, its 16-lane phases spread
//     over the 16 four-bank slots by the code's low bits (random codes: ~3 LDS cycles per phase
//     instead of one, the price of a gather);
//   * a workgroup keeps the tables of 16 subspaces resident (128 KB) and swaps the slice while its
//     lanes hold the partial sums of 8 rows x 8 queries in registers, so the code matrix streams
//     once per 8 queries and a slice is re-read once per 8192 rows (the tables themselves are
//     computed once per batch by pq_tables_kernel and stay in L2 / Infinity Cache);
//   * the codes are stored a second time subspace-group-major ([group][row][16 bytes]), so a lane
//     loads the 16 codes of a slice with one coalesced 16-byte read;
//   * sums run in the reference's subspace order (slice after slice, +0.0 first), the square root is
//     taken only for rows whose sum is not above the list's threshold squared.
// Each wave keeps a candidate list per query (scores = -distance) with the flat scan's compaction.
#include "nh_common.h"
#include "nh_kernels.h"
#include "nh_select.h"

#pragma clang fp contract(off)

namespace nh {

constexpr int PQ_RPL = 8;                                  // rows per lane and block
constexpr uint32_t PQ_THREADS = 64 * PQ_WAVES;
constexpr uint32_t PQ_BLOCK_ROWS = PQ_THREADS * PQ_RPL;    // rows a workgroup carries through the slices at a time
constexpr uint32_t PQ_SLICE_F4 = 2 * 256;                  // float4 entries of one subspace: [half][code]

uint32_t pq_scan_block_rows() { return PQ_BLOCK_ROWS; }

// compute_distance_table, quantization.rs:468-494: per entry a sum over the subspace folded from
// -0.0 in dimension order.  tables[(((G * m + s) * 2 + h) * 256 + c) * 4 + l] belongs to query 8 G + 4 h + l.
__global__ __launch_bounds__(256) void pq_tables_kernel(const float* __restrict__ queries, const float* __restrict__ codebooks,
                                                        uint32_t q, uint32_t m, uint32_t dsub, uint32_t dim,
                                                        float* __restrict__ tables, uint64_t total) {
    const uint64_t idx = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const uint32_t l = (uint32_t)idx & 3u, c = (uint32_t)(idx >> 2) & 255u, h = (uint32_t)(idx >> 10) & 1u;
    const uint64_t gs = idx >> 11;
    const uint32_t s = (uint32_t)(gs % m), query = (uint32_t)(gs / m) * PQ_QB + 4 * h + l;
    float sum = -0.0f;
    if (query < q) {
        const float* qs = queries + (uint64_t)query * dim + s * dsub;
        const float* cb = codebooks + ((uint64_t)s * 256 + c) * dsub;
        for (uint32_t j = 0; j < dsub; ++j) {
            const float d = qs[j] - cb[j];
            const float sq = d * d;
            sum = sum + sq;
        }
    }
    tables[idx] = sum;
}

// codes [n][m] -> [ceil(m / 16)][n][16], missing subspaces of the last group zero
__global__ __launch_bounds__(256) void pq_group_codes_kernel(const uint8_t* __restrict__ codes, uint64_t n, uint32_t m, uint32_t n_sg,
                                                             uint8_t* __restrict__ out) {
    const uint64_t idx = (uint64_t)blockIdx.x * 256 + threadIdx.x;  // one output byte
    if (idx >= (uint64_t)n_sg * n * PQ_SG) return;
    const uint32_t j = (uint32_t)idx & (PQ_SG - 1);
    const uint64_t gr = idx / PQ_SG, row = gr % n;
    const uint32_t s = (uint32_t)(gr / n) * PQ_SG + j;
    out[idx] = s < m ? codes[row * m + s] : (uint8_t)0;
}

hipError_t launch_pq_group_codes(const uint8_t* codes, uint64_t n, uint32_t m, uint8_t* out, hipStream_t stream) {
    const uint32_t n_sg = (m + PQ_SG - 1) / PQ_SG;
    const uint64_t total = (uint64_t)n_sg * n * PQ_SG;
    if (total == 0) return hipSuccess;
    hipLaunchKernelGGL(pq_group_codes_kernel, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, stream, codes, n, m, n_sg, out);
    return hipGetLastError();
}

// every distance not above this has a square root that may pass `tau` (score = -distance): the sum of squares is
// compared first, the correctly rounded root taken only for rows that survive
__device__ inline float pq_sum_bound(float tau) {
    const float td = -tau;
    return fmaxf(td * td * 1.000001f, 1e-30f);
}

template <bool FULL>
__device__ inline void pq_accumulate(const float4* __restrict__ lut4, const uint4& cw, uint32_t s_g, float (&dist)[PQ_QB]) {
    const uint32_t w[4] = {cw.x, cw.y, cw.z, cw.w};
#pragma unroll
    for (int sl = 0; sl < PQ_SG; ++sl) {
        if (!FULL && (uint32_t)sl >= s_g) break;  // block-uniform
        const uint32_t code = (w[sl >> 2] >> (8 * (sl & 3))) & 255u;
        // distance_with_table, quantization.rs:497-503: dist += table[s][code_s], s ascending
        const float4 a = lut4[(sl * 2 + 0) * 256 + code];
        const float4 b = lut4[(sl * 2 + 1) * 256 + code];
        dist[0] = dist[0] + a.x; dist[1] = dist[1] + a.y; dist[2] = dist[2] + a.z; dist[3] = dist[3] + a.w;
        dist[4] = dist[4] + b.x; dist[5] = dist[5] + b.y; dist[6] = dist[6] + b.z; dist[7] = dist[7] + b.w;
    }
}

__global__ __launch_bounds__(PQ_THREADS) void pq_scan_kernel(PqScanParams p) {
    extern __shared__ __attribute__((aligned(16))) float4 lut4[];  // [subspace of the slice][half][code] x 4 queries
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t split = blockIdx.x, grp = blockIdx.y, g0 = grp * PQ_QB;
    const uint32_t r0 = split * p.rows_per_split;
    const uint32_t r1 = min(r0 + p.rows_per_split, p.n);
    const uint32_t lists_per_q = p.n_splits * PQ_WAVES;
    float tau[PQ_QB], bound[PQ_QB];  // wave-uniform: score of the k-th best entry after the last compaction, and its sum bound
    uint32_t cnt[PQ_QB];             // wave-uniform list lengths
#pragma unroll
    for (int g = 0; g < PQ_QB; ++g) { tau[g] = -INFINITY; bound[g] = INFINITY; cnt[g] = 0; }
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    // pairs of padding queries (beyond the batch) are never looked at
    const uint64_t qmask = 0x0101010101010101ull * (uint64_t)((g0 + PQ_QB <= p.q) ? 0xffu : ((1u << (p.q - g0)) - 1u));
    const float4* tab = (const float4*)p.tables + (uint64_t)grp * p.m * PQ_SLICE_F4;
    const uint4* codes = (const uint4*)p.codes_gm;

    for (uint32_t blk = r0; blk < r1; blk += PQ_BLOCK_ROWS) {  // block-uniform trip count
        float dist[PQ_RPL][PQ_QB];
#pragma unroll
        for (int r = 0; r < PQ_RPL; ++r)
#pragma unroll
            for (int g = 0; g < PQ_QB; ++g) dist[r][g] = 0.0f;  // distance_with_table starts at +0.0
        for (uint32_t sg = 0; sg < p.n_sg; ++sg) {
            const uint32_t s_g = min((uint32_t)PQ_SG, p.m - sg * PQ_SG);
            __syncthreads();  // the previous slice is no longer read
            const float4* src = tab + (uint64_t)sg * PQ_SG * PQ_SLICE_F4;
            for (uint32_t i = tid; i < s_g * PQ_SLICE_F4; i += PQ_THREADS) lut4[i] = src[i];
            __syncthreads();
            const uint4* cg = codes + (uint64_t)sg * p.n;
#pragma unroll
            for (int rh = 0; rh < PQ_RPL; rh += 4) {
                uint4 cw[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) cw[r] = cg[min(blk + (uint32_t)(rh + r) * PQ_THREADS + tid, p.n - 1)];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (s_g == PQ_SG) pq_accumulate<true>(lut4, cw[r], s_g, dist[rh + r]);
                    else pq_accumulate<false>(lut4, cw[r], s_g, dist[rh + r]);
                }
            }
        }
        // Which (row slot, query) pairs have a lane whose sum is not above the list's bound?  After the first blocks
        // almost none: one compare and a few scalar operations per pair, no branch.
        uint32_t liveb = 0;  // bit r: this lane's row of slot r may be listed
#pragma unroll
        for (int r = 0; r < PQ_RPL; ++r) {
            const uint32_t i = blk + (uint32_t)r * PQ_THREADS + tid;
            bool live = i < r1;
            if (live && p.valid) live = bit_set(p.valid, i);
            if (live && p.filter) live = bit_set(p.filter, i);
            liveb |= live ? (1u << r) : 0u;
        }
        uint64_t todo = 0;  // wave-uniform, bit 8 r + g
#pragma unroll
        for (int r = 0; r < PQ_RPL; ++r)
#pragma unroll
            for (int g = 0; g < PQ_QB; ++g) {
                const bool close_by = ((liveb >> r) & 1u) && dist[r][g] <= bound[g];
                todo |= __builtin_amdgcn_ballot_w64(close_by) != 0 ? (1ull << (8 * r + g)) : 0ull;
            }
        todo &= qmask;
        // The pairs that do, in ascending (row slot, query) order: per list the rows still arrive in ascending id, so a
        // later row tying the k-th kept score loses the id tie-break and strict '>' discards exactly the rows that
        // cannot enter.
        while (todo) {
            const uint32_t idx = (uint32_t)__builtin_ctzll(todo);
            todo &= todo - 1;
            const uint32_t r = idx >> 3, g = idx & 7u;
            float d = 0.0f;
#pragma unroll
            for (int j = 0; j < PQ_RPL * PQ_QB; ++j)
                if (idx == (uint32_t)j) d = dist[j >> 3][j & 7];
            float tau_g = tau[0];
            uint32_t cnt_g = cnt[0];
#pragma unroll
            for (int j = 1; j < PQ_QB; ++j)
                if (g == (uint32_t)j) { tau_g = tau[j]; cnt_g = cnt[j]; }
            const uint32_t i = blk + r * PQ_THREADS + tid;
            const float score = -canon_distance(sqrtf(d));
            const bool pass = ((liveb >> r) & 1u) && (score > tau_g);
            const uint64_t pm = __builtin_amdgcn_ballot_w64(pass);
            if (pm == 0) continue;
            const uint32_t li = (g0 + g) * lists_per_q + split * PQ_WAVES + wave;
            uint2* list = p.lists + (uint64_t)li * p.cap;
            if (pass) list[cnt_g + __builtin_popcountll(pm & lt_mask)] = make_uint2(__float_as_uint(score), i);
            cnt_g += __builtin_popcountll(pm);
            if (cnt_g > p.cap - 64) {
                tau_g = wave_compact_list(list, cnt_g, p.k, lane);
                cnt_g = p.k;
            }
            const float bound_g = pq_sum_bound(tau_g);
#pragma unroll
            for (int j = 0; j < PQ_QB; ++j)
                if (g == (uint32_t)j) { tau[j] = tau_g; cnt[j] = cnt_g; bound[j] = bound_g; }
        }
    }
#pragma unroll
    for (int g = 0; g < PQ_QB; ++g) {
        if (g0 + g >= p.q) continue;
        const uint32_t li = (g0 + g) * lists_per_q + split * PQ_WAVES + wave;
        if (cnt[g] > p.k) {
            wave_compact_list(p.lists + (uint64_t)li * p.cap, cnt[g], p.k, lane);
            cnt[g] = p.k;
        }
        if (lane == 0) p.list_counts[li] = cnt[g];
    }
}

hipError_t launch_pq_tables(const float* d_queries, const float* d_codebooks, uint32_t q, uint32_t m, uint32_t dsub, uint32_t dim,
                            float* d_tables, hipStream_t stream) {
    const uint64_t total = (uint64_t)((q + PQ_QB - 1) / PQ_QB) * m * PQ_SLICE_F4 * 4;
    if (total == 0) return hipSuccess;
    hipLaunchKernelGGL(pq_tables_kernel, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, stream, d_queries, d_codebooks, q, m, dsub,
                       dim, d_tables, total);
    return hipGetLastError();
}

hipError_t launch_pq_scan(const PqScanParams& p, hipStream_t stream) {
    if (p.q == 0 || p.n == 0 || p.m == 0) return hipSuccess;
    if (p.n_sg != (p.m + PQ_SG - 1) / PQ_SG || !p.tables || !p.codes_gm) return hipErrorInvalidValue;
    const size_t lds = (size_t)std::min<uint32_t>(p.m, PQ_SG) * PQ_SLICE_F4 * sizeof(float4);
    hipError_t e = hipFuncSetAttribute((const void*)pq_scan_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const dim3 grid(p.n_splits, (p.q + PQ_QB - 1) / PQ_QB);
    hipLaunchKernelGGL(pq_scan_kernel, grid, dim3(PQ_THREADS), lds, stream, p);
    return hipGetLastError();
}

}  // namespace nh
