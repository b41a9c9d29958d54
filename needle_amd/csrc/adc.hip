// Quantized (ADC) scans.
//
// Product quantizer: the reference's table scan
// (ProductQuantizer::{compute_distance_table, distance_with_table},
// src/indexing/quantization.rs:468-503) evaluated directly in its own order, so
// the scan's distances ARE the reported distances: no re-scoring, no bound
// check.  A workgroup builds the distance tables of `qg` queries in LDS
// ([m][256] f32 each, 48 KB at m = 48), then streams its slice of the code
// matrix: lane = row, the row's m code bytes are read once (4 bytes at a time)
// and feed the qg table sums; each wave keeps its own per-query candidate list
// (scores = -distance) with the same compaction as the flat scan.
#include "nh_common.h"
#include "nh_kernels.h"
#include "nh_select.h"

#pragma clang fp contract(off)

namespace nh {

constexpr int PQ_QG_MAX = 3;                  // tables resident per workgroup
constexpr uint32_t PQ_LDS_BUDGET = 150 * 1024;

uint32_t pq_scan_queries_per_block(uint32_t m) {
    const uint32_t per = m * 256u * sizeof(float);
    if (per == 0 || per > PQ_LDS_BUDGET) return 0;
    return std::min<uint32_t>(PQ_QG_MAX, PQ_LDS_BUDGET / per);
}

__global__ __launch_bounds__(64 * PQ_WAVES) void pq_scan_kernel(PqScanParams p) {
    extern __shared__ __attribute__((aligned(16))) float lut[];  // [qg][m][256]
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t split = blockIdx.x, g0 = blockIdx.y * p.qg;
    const uint32_t tbl = p.m * 256;

    // compute_distance_table, quantization.rs:468-494: per entry a sum over the
    // subspace folded from -0.0 in dimension order
    for (uint32_t idx = tid; idx < p.qg * tbl; idx += 64 * PQ_WAVES) {
        const uint32_t g = idx / tbl, rem = idx - g * tbl, s = rem >> 8, c = rem & 255;
        float sum = -0.0f;
        if (g0 + g < p.q) {
            const float* qs = p.queries + (uint64_t)(g0 + g) * p.dim + s * p.dsub;
            const float* cb = p.codebooks + ((uint64_t)s * 256 + c) * p.dsub;
            for (uint32_t j = 0; j < p.dsub; ++j) {
                const float d = qs[j] - cb[j];
                const float sq = d * d;
                sum = sum + sq;
            }
        }
        lut[idx] = sum;
    }
    __syncthreads();

    const uint32_t r0 = split * p.rows_per_split;
    const uint32_t r1 = min(r0 + p.rows_per_split, p.n);
    const uint32_t lists_per_q = p.n_splits * PQ_WAVES;
    float tau[PQ_QG_MAX];      // wave-uniform: score of the k-th best entry after the last compaction
    uint32_t cnt[PQ_QG_MAX];   // wave-uniform list lengths
#pragma unroll
    for (int g = 0; g < PQ_QG_MAX; ++g) { tau[g] = -INFINITY; cnt[g] = 0; }
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    const uint32_t words = p.m >> 2, tail = p.m & 3;

    for (uint32_t base = r0 + wave * 64; base < r1; base += 64 * PQ_WAVES) {
        const uint32_t i = base + lane;
        bool live = i < r1;
        if (live && p.valid) live = bit_set(p.valid, i);
        if (live && p.filter) live = bit_set(p.filter, i);
        const uint8_t* row = p.codes + (uint64_t)(i < p.n ? i : 0) * p.m;
        float dist[PQ_QG_MAX];
#pragma unroll
        for (int g = 0; g < PQ_QG_MAX; ++g) dist[g] = 0.0f;  // distance_with_table starts at +0.0
        // distance_with_table, quantization.rs:497-503: dist += table[i][code_i], i ascending
        for (uint32_t w = 0; w < words; ++w) {
            uint32_t cw;
            if ((p.m & 3) == 0) cw = *(const uint32_t*)(row + 4 * w);
            else cw = (uint32_t)row[4 * w] | ((uint32_t)row[4 * w + 1] << 8) | ((uint32_t)row[4 * w + 2] << 16) | ((uint32_t)row[4 * w + 3] << 24);
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const uint32_t code = (cw >> (8 * b)) & 255u;
                const uint32_t off = (4 * w + b) * 256 + code;
#pragma unroll
                for (int g = 0; g < PQ_QG_MAX; ++g)
                    if ((uint32_t)g < p.qg) dist[g] = dist[g] + lut[g * tbl + off];
            }
        }
        for (uint32_t b = 0; b < tail; ++b) {
            const uint32_t s = 4 * words + b, off = s * 256 + row[s];
#pragma unroll
            for (int g = 0; g < PQ_QG_MAX; ++g)
                if ((uint32_t)g < p.qg) dist[g] = dist[g] + lut[g * tbl + off];
        }
#pragma unroll
        for (int g = 0; g < PQ_QG_MAX; ++g) {
            if ((uint32_t)g >= p.qg || g0 + g >= p.q) continue;  // wave-uniform
            const float score = -canon_distance(sqrtf(dist[g]));
            // rows arrive in ascending id per list, so a later row tying the k-th kept score
            // loses the id tie-break: strict '>' discards exactly the rows that cannot enter
            const bool pass = live && (score > tau[g]);
            const uint64_t pm = __builtin_amdgcn_ballot_w64(pass);
            if (pm == 0) continue;
            const uint32_t li = (g0 + g) * lists_per_q + split * PQ_WAVES + wave;
            uint2* list = p.lists + (uint64_t)li * p.cap;
            if (pass) list[cnt[g] + __builtin_popcountll(pm & lt_mask)] = make_uint2(__float_as_uint(score), i);
            cnt[g] += __builtin_popcountll(pm);
            if (cnt[g] > p.cap - 64) {
                tau[g] = wave_compact_list(list, cnt[g], p.k, lane);
                cnt[g] = p.k;
            }
        }
    }
#pragma unroll
    for (int g = 0; g < PQ_QG_MAX; ++g) {
        if ((uint32_t)g >= p.qg || g0 + g >= p.q) continue;
        const uint32_t li = (g0 + g) * lists_per_q + split * PQ_WAVES + wave;
        if (cnt[g] > p.k) {
            wave_compact_list(p.lists + (uint64_t)li * p.cap, cnt[g], p.k, lane);
            cnt[g] = p.k;
        }
        if (lane == 0) p.list_counts[li] = cnt[g];
    }
}

hipError_t launch_pq_scan(const PqScanParams& p, hipStream_t stream) {
    if (p.q == 0 || p.n == 0 || p.qg == 0) return hipSuccess;
    const size_t lds = (size_t)p.qg * p.m * 256 * sizeof(float);
    hipError_t e = hipFuncSetAttribute((const void*)pq_scan_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const dim3 grid(p.n_splits, (p.q + p.qg - 1) / p.qg);
    hipLaunchKernelGGL(pq_scan_kernel, grid, dim3(64 * PQ_WAVES), lds, stream, p);
    return hipGetLastError();
}

}  // namespace nh
