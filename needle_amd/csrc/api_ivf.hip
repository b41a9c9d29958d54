// IVF-Flat on the device (C ABI: include/needle_hip.h, "IVF index").
//
// Layout: rows are stored list by list (insertion order inside a list, as
// IvfCluster::{ids,vectors}, src/indexing/ivf.rs:181-196), every list padded to
// a whole number of 128-row tiles so that a tile never straddles two lists.
// pos2id maps a position of that copy back to the row id (~0 for padding).
//
// Search = IvfIndex::search, ivf.rs:509-567:
//   1. coarse: exact flat search of the centroid set with k = nprobe (Euclidean,
//      ties to the lower list id == the stable sort of ivf.rs:528)
//   2. list scan, LIST-MAJOR: the (query, probed list) pairs are regrouped by
//      list; one workgroup multiplies one list's tiles with up to 128 of the
//      queries probing it, so a list is read once per 128 probing queries rather
//      than once per query
//   3. merge of the query's nprobe candidate lists, exact re-scoring, bound
//      check, exact scan of the probed lists for rejected queries (flat_post.hip)
#include "nh_api.h"

#include <algorithm>
#include <chrono>
#include <functional>

using namespace nh;

__global__ __launch_bounds__(256) void ivf_gather_rows_kernel(const float* __restrict__ src,
                                                              const uint32_t* __restrict__ pos2id, uint32_t n_pos,
                                                              uint32_t dim, float* __restrict__ dst) {
    const uint32_t pos = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (pos >= n_pos) return;
    const uint32_t id = pos2id[pos];
    float* d = dst + (uint64_t)pos * dim;
    if (id == 0xffffffffu) {
        for (uint32_t i = lane; i < dim; i += 64) d[i] = 0.0f;
    } else {
        const float* s = src + (uint64_t)id * dim;
        for (uint32_t i = lane; i < dim; i += 64) d[i] = s[i];
    }
}

// Accumulator seeds of padding positions: -inf, so that a padding row scores -inf in every first pass and the list
// scan needs no position -> id lookup to reject it (row_live<MODE_IVF>).
__global__ void ivf_pad_seed_kernel(const uint32_t* __restrict__ pos2id, uint32_t n_pos, float* __restrict__ cinit) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_pos && pos2id[i] == 0xffffffffu) cinit[i] = -INFINITY;
}

// Per-search accumulator seeds under row masks: positions whose row is deleted or filtered out get -inf as well, so
// the list scan runs without any position -> id -> mask lookups.
__global__ void ivf_masked_seed_kernel(const float* __restrict__ cinit, const uint32_t* __restrict__ pos2id, uint32_t n_pos,
                                       const uint32_t* __restrict__ valid, const uint32_t* __restrict__ filter,
                                       float* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pos) return;
    const uint32_t id = pos2id[i];
    bool live = id != 0xffffffffu;
    if (live && valid) live = nh::bit_set(valid, id);
    if (live && filter) live = nh::bit_set(filter, id);
    out[i] = live ? cinit[i] : -INFINITY;
}

__global__ void ivf_narrow_probe_kernel(const uint64_t* __restrict__ ids, uint32_t total, uint32_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) out[i] = (uint32_t)ids[i];  // padding UINT64_MAX -> UINT32_MAX
}

// Starting thresholds of the list scan.  One workgroup (four waves) per query scores (up to) the first IVF_SEED_ROWS rows of the
// query's nearest list with plain vector arithmetic, in the first pass's score units (q.b - |b|^2/2), and takes
// the kseed-th largest score of live rows minus twice the score uncertainty: kseed live rows reach it, so the
// scan may start from it (a lower rank or a shorter sample only makes the threshold weaker, never wrong).
constexpr uint32_t IVF_SEED_ROWS = 256;  // slots of the sample; the launch passes how many rows it scores

__device__ __forceinline__ float seed_elt(const float* r, uint64_t i) { return r[i]; }
__device__ __forceinline__ float seed_elt(const uint16_t* r, uint64_t i) { return __uint_as_float((uint32_t)r[i] << 16); }

// eight consecutive elements of a row as f32; p is 16-byte aligned
__device__ __forceinline__ void seed_load8(const float* p, float (&v)[8]) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void seed_load8(const uint16_t* p, float (&v)[8]) {
    const uint4 u = *reinterpret_cast<const uint4*>(p);
    v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xffff0000u);
    v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xffff0000u);
    v[4] = __uint_as_float(u.z << 16); v[5] = __uint_as_float(u.z & 0xffff0000u);
    v[6] = __uint_as_float(u.w << 16); v[7] = __uint_as_float(u.w & 0xffff0000u);
}

template <typename T>  // rows: the f32 rows, or their bf16 copy (row stride ld elements)
__global__ __launch_bounds__(256) void ivf_seed_tau_kernel(const float* __restrict__ queries, const T* __restrict__ rows, uint32_t ld,
                                                           uint32_t sample_rows, const float* __restrict__ cinit, const uint32_t* __restrict__ probe,
                                                           const uint32_t* __restrict__ probe_n, uint32_t np, uint32_t nlist,
                                                           const uint32_t* __restrict__ list_pos,
                                                           const uint32_t* __restrict__ list_len,
                                                           const uint32_t* __restrict__ pos2id, const uint32_t* __restrict__ valid,
                                                           const uint32_t* __restrict__ filter, uint32_t dim, uint32_t kseed,
                                                           nh::TauBand band, float* __restrict__ tau) {
    __shared__ uint32_t ord[IVF_SEED_ROWS];  // order-preserving score bits; 0 (below every score) for dead or absent rows
    __shared__ float sums[IVF_SEED_ROWS];    // q . row of sample row r
    __shared__ uint32_t n_live;
    const uint32_t q = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (threadIdx.x == 0) n_live = 0;
    ord[threadIdx.x] = 0u;
    __syncthreads();
    if (probe_n[q] > 0) {  // block-uniform
        const uint32_t l = min(probe[(uint64_t)q * np], nlist - 1);
        const uint32_t p0 = list_pos[l], len = min(list_len[l], min(sample_rows, IVF_SEED_ROWS));
        const float* qv = queries + (uint64_t)q * dim;
        // every wave scores a quarter of the sample (the steps of the scoring loop are serial: an even split keeps the
        // loop short however many rows are sampled); lane L owns row wave * per + L: its liveness and seed term are
        // looked up once, ahead of the scoring loop, which then holds no dependent loads
        const uint32_t per = (len + 3) / 4;  // <= 64
        const uint32_t rbase = wave * per, rend = min(len, rbase + per), rmine = rbase + lane;
        bool ok = false;
        float ci = 0.0f, mine = 0.0f;
        if (rmine < rend) {
            const uint32_t id = pos2id[p0 + rmine];
            ok = id != 0xffffffffu;
            if (ok && valid) ok = nh::bit_set(valid, id);
            if (ok && filter) ok = nh::bit_set(filter, id);
            ci = cinit[p0 + rmine];
        }
        // SEED_STEP rows per step: their loads are all in flight before the first reduction.  Rows whose stride is a
        // multiple of 8 elements are read 8 elements (16 or 32 bytes) per lane against query values held in registers.
        constexpr uint32_t SEED_PASSES = 4;  // 64 lanes x 8 elements per pass
        constexpr uint32_t SEED_STEP = 8;
        const bool wide = ld % 8 == 0 && ld <= SEED_PASSES * 512;
        float qreg[SEED_PASSES][8];
        if (wide) {
#pragma unroll
            for (uint32_t ps = 0; ps < SEED_PASSES; ++ps)
#pragma unroll
                for (uint32_t t = 0; t < 8; ++t) {
                    const uint32_t e = ps * 512 + lane * 8 + t;
                    qreg[ps][t] = e < dim ? qv[e] : 0.0f;  // the stride's padding beyond dim holds zeros or is not read
                }
        }
        for (uint32_t r0 = rbase; r0 < rend; r0 += SEED_STEP) {
            float s[SEED_STEP];
#pragma unroll
            for (uint32_t j = 0; j < SEED_STEP; ++j) s[j] = 0.0f;
            if (wide) {
#pragma unroll
                for (uint32_t ps = 0; ps < SEED_PASSES; ++ps) {
                    const uint32_t e = ps * 512 + lane * 8;
                    if (e < ld) {
                        float v[SEED_STEP][8];
#pragma unroll
                        for (uint32_t j = 0; j < SEED_STEP; ++j)
                            seed_load8(rows + (uint64_t)(p0 + min(r0 + j, len - 1)) * ld + e, v[j]);
#pragma unroll
                        for (uint32_t j = 0; j < SEED_STEP; ++j)
#pragma unroll
                            for (uint32_t t = 0; t < 8; ++t) s[j] += qreg[ps][t] * v[j][t];
                    }
                }
            } else {
                for (uint32_t i = lane; i < dim; i += 64) {
                    const float x = qv[i];
#pragma unroll
                    for (uint32_t j = 0; j < SEED_STEP; ++j) s[j] += x * seed_elt(rows, (uint64_t)(p0 + min(r0 + j, len - 1)) * ld + i);
                }
            }
            // Sum over the lanes, the 8 rows together: at the first three butterfly levels a lane keeps half of the rows it
            // still holds and hands the other half to its partner (4 + 2 + 1 exchanges), then three levels on the one row
            // left: 10 exchanges for 8 rows instead of 48, the same additions in the same order.  Lane L then holds the
            // sum of row r0 + (L >> 3).
            const bool b5 = lane & 32, b4 = lane & 16, b3 = lane & 8;
            float t4[4], t2[2];
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j) t4[j] = (b5 ? s[j + 4] : s[j]) + __shfl_xor(b5 ? s[j] : s[j + 4], 32);
#pragma unroll
            for (uint32_t j = 0; j < 2; ++j) t2[j] = (b4 ? t4[j + 2] : t4[j]) + __shfl_xor(b4 ? t4[j] : t4[j + 2], 16);
            float t1 = (b3 ? t2[1] : t2[0]) + __shfl_xor(b3 ? t2[0] : t2[1], 8);
            t1 += __shfl_xor(t1, 4);
            t1 += __shfl_xor(t1, 2);
            t1 += __shfl_xor(t1, 1);
            if ((lane & 7) == 0 && r0 + (lane >> 3) < rend) sums[r0 + (lane >> 3)] = t1;  // r0 + 7 < rbase + 64 <= 256
        }
        __syncthreads();  // (the branch is uniform over the workgroup)
        if (rmine < rend) mine = sums[rmine];
        if (ok) ord[rmine] = nh::f2ord(mine + ci);
        const uint32_t live = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(ok));
        if (lane == 0 && live) atomicAdd(&n_live, live);
    }
    __syncthreads();
    if (wave != 0) return;
    float t = -INFINITY;
    if (n_live >= kseed) {
        uint32_t prefix = 0;
        const uint32_t o0 = ord[lane], o1 = ord[64 + lane], o2 = ord[128 + lane], o3 = ord[192 + lane];
        for (int bit = 31; bit >= 0; --bit) {
            const uint32_t trial = prefix | (1u << bit);  // > 0: never counts an empty slot
            const uint32_t cnt = __builtin_popcountll(__builtin_amdgcn_ballot_w64(o0 >= trial)) +
                                 __builtin_popcountll(__builtin_amdgcn_ballot_w64(o1 >= trial)) +
                                 __builtin_popcountll(__builtin_amdgcn_ballot_w64(o2 >= trial)) +
                                 __builtin_popcountll(__builtin_amdgcn_ballot_w64(o3 >= trial));
            if (cnt >= kseed) prefix = trial;
        }
        const double qn = sqrt(band.qn2[q]);
        const double lowp = nh::bf16_score_err(band.metric, qn, band.bmax, band.qerr ? (double)band.qerr[q] : -1.0, band.berr);
        const float tb = (float)((double)nh::ord2f(prefix) - 2.0 * nh::score_eps(band.metric, lowp, qn, band.bmax, band.dim));
        if (tb == tb) t = tb;
    }
    if (lane == 0) tau[q] = t;
}

// ---- regrouping of the (query, probed list) pairs by list, on the device ----------------------------------
// Stage r of the list scan covers probe slots [bounds[r], bounds[r+1]); per (stage, list) the pairs probing the list
// are collected (counting sort: count, exclusive scan, scatter) and cut into work items of <= item_queries pairs per
// run of `run` row tiles.  The order of the pairs inside a list follows the atomics and is not deterministic; the
// answer does not depend on it (every pair owns its candidate lists, and the bound check proves the result).
struct RegroupBufs {
    const uint32_t *probe, *probe_n, *list_len, *list_pos;
    uint32_t *cnt, *fill, *pair_off, *item_off;  // [3 * nlist] (+1 for the offsets)
    uint32_t *qlists, *lbase, *lbase_dev;        // candidate lists of every query; exclusive sums ([q + 1]; x lpe)
    uint32_t* slot_rel;                          // [q][np] first candidate list of (query, slot), relative to the query
    uint32_t* totals;                            // [0] candidate lists, [1..3] end of stage 0..2 in items, [4] most lists of one query,
                                                 // [5..8] item bounds of the three launches
    uint2* itemq;
    nh::IvfItem* items;
    uint32_t q, np, nlist, run, item_queries, lpe, b1, b2;
};

__device__ __forceinline__ uint32_t rg_runs(const RegroupBufs& g, uint32_t l) {
    const uint32_t tiles = (g.list_len[l] + nh::TILE_ROWS - 1) / nh::TILE_ROWS;
    return max(1u, (tiles + g.run - 1) / g.run);
}
__device__ __forceinline__ uint32_t rg_stage(const RegroupBufs& g, uint32_t slot) { return slot < g.b1 ? 0u : slot < g.b2 ? 1u : 2u; }

__global__ __launch_bounds__(256) void ivf_regroup_count_kernel(RegroupBufs g) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= g.q) return;
    uint32_t at = 0;
    const uint32_t n = min(g.probe_n[i], g.np);
    // eight probe slots per step: their list ids, then their list lengths, are fetched together (a slot at a time the
    // loop is a chain of two dependent loads per slot)
    for (uint32_t s0 = 0; s0 < n; s0 += 8) {
        uint32_t l[8], r[8];
#pragma unroll
        for (uint32_t u = 0; u < 8; ++u) l[u] = g.probe[(uint64_t)i * g.np + min(s0 + u, n - 1)];
#pragma unroll
        for (uint32_t u = 0; u < 8; ++u) r[u] = l[u] < g.nlist ? rg_runs(g, l[u]) : 0u;
#pragma unroll
        for (uint32_t u = 0; u < 8; ++u) {
            const uint32_t sl = s0 + u;
            if (sl < n && l[u] < g.nlist) {  // (a list id out of range is ignored, here and in the scatter)
                g.slot_rel[(uint64_t)i * g.np + sl] = at;
                at += r[u];
                atomicAdd(&g.cnt[rg_stage(g, sl) * g.nlist + l[u]], 1u);
            }
        }
    }
    g.qlists[i] = at;
}

// three exclusive sums at once by one workgroup of 1024 threads (one set of barriers for all three):
// value_j(0 .. n_j - 1) into out_j[0 .. n_j]; part holds 3 x 1024 words
template <typename F0, typename F1, typename F2>
__device__ void rg_block_scan3(uint32_t n0, F0 v0, uint32_t* out0, uint32_t n1, F1 v1, uint32_t* out1, uint32_t n2, F2 v2,
                               uint32_t* out2, uint32_t (*part)[1024], uint32_t (&total)[3], uint32_t* out0_scaled,
                               uint32_t scale0, uint32_t& max0) {
    // out0_scaled[i] = out0[i] * scale0 rides along with the first sum, max0 returns the thread's largest value_0: what
    // the caller would otherwise re-read from memory behind the sums (a single workgroup is a chain of memory round trips)
    const uint32_t t = threadIdx.x;
    const uint32_t n[3] = {n0, n1, n2};
    uint32_t b[3], e[3], s[3] = {0u, 0u, 0u};
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const uint32_t per = (n[j] + 1023) / 1024;
        b[j] = min(n[j], t * per);
        e[j] = min(n[j], b[j] + per);
    }
    // a thread's values are fetched once, all loads in flight together (up to RG_CACHE per sum: 16K queries, 5461 lists;
    // beyond that the plain loops run, one dependent load after the other)
    constexpr uint32_t RG_CACHE = 16;
    uint32_t c0[RG_CACHE], c1[RG_CACHE], c2[RG_CACHE];
    const bool cached = e[0] - b[0] <= RG_CACHE && e[1] - b[1] <= RG_CACHE && e[2] - b[2] <= RG_CACHE;  // block-uniform
    if (cached) {
#pragma unroll
        for (uint32_t u = 0; u < RG_CACHE; ++u) {
            c0[u] = b[0] + u < e[0] ? v0(b[0] + u) : 0u;
            c1[u] = b[1] + u < e[1] ? v1(b[1] + u) : 0u;
            c2[u] = b[2] + u < e[2] ? v2(b[2] + u) : 0u;
        }
#pragma unroll
        for (uint32_t u = 0; u < RG_CACHE; ++u) { s[0] += c0[u]; s[1] += c1[u]; s[2] += c2[u]; max0 = max(max0, c0[u]); }
    } else {
        for (uint32_t i = b[0]; i < e[0]; ++i) { const uint32_t v = v0(i); s[0] += v; max0 = max(max0, v); }
        for (uint32_t i = b[1]; i < e[1]; ++i) s[1] += v1(i);
        for (uint32_t i = b[2]; i < e[2]; ++i) s[2] += v2(i);
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) part[j][t] = s[j];
    __syncthreads();
    for (uint32_t off = 1; off < 1024; off <<= 1) {
        uint32_t v[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) v[j] = t >= off ? part[j][t - off] : 0u;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 3; ++j) part[j][t] += v[j];
        __syncthreads();
    }
    uint32_t acc[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        acc[j] = t ? part[j][t - 1] : 0u;
        total[j] = part[j][1023];
    }
    if (cached) {
#pragma unroll
        for (uint32_t u = 0; u < RG_CACHE; ++u) {
            if (b[0] + u < e[0]) { out0[b[0] + u] = acc[0]; out0_scaled[b[0] + u] = acc[0] * scale0; acc[0] += c0[u]; }
            if (b[1] + u < e[1]) { out1[b[1] + u] = acc[1]; acc[1] += c1[u]; }
            if (b[2] + u < e[2]) { out2[b[2] + u] = acc[2]; acc[2] += c2[u]; }
        }
    } else {
        for (uint32_t i = b[0]; i < e[0]; ++i) { out0[i] = acc[0]; out0_scaled[i] = acc[0] * scale0; acc[0] += v0(i); }
        for (uint32_t i = b[1]; i < e[1]; ++i) { out1[i] = acc[1]; acc[1] += v1(i); }
        for (uint32_t i = b[2]; i < e[2]; ++i) { out2[i] = acc[2]; acc[2] += v2(i); }
    }
    if (t == 0) { out0[n0] = total[0]; out0_scaled[n0] = total[0] * scale0; out1[n1] = total[1]; out2[n2] = total[2]; }
    __syncthreads();
}

__global__ __launch_bounds__(1024) void ivf_regroup_scan_kernel(RegroupBufs g) {
    __shared__ uint32_t part[3][1024];
    __shared__ uint32_t most;
    if (threadIdx.x == 0) most = 0;
    __syncthreads();
    const uint32_t SL = 3 * g.nlist;
    uint32_t total[3];
    uint32_t m = 0;
    rg_block_scan3(
        g.q, [&](uint32_t i) { return g.qlists[i]; }, g.lbase,
        SL, [&](uint32_t i) { return g.cnt[i]; }, g.pair_off,
        SL, [&](uint32_t i) {
            const uint32_t l = i % g.nlist, c = g.cnt[i];
            return (c == 0 || g.list_len[l] == 0) ? 0u : rg_runs(g, l) * ((c + g.item_queries - 1) / g.item_queries);
        }, g.item_off, part, total, g.lbase_dev, g.lpe, m);
    atomicMax(&most, m);
    __syncthreads();
    if (threadIdx.x == 0) {
        g.totals[0] = total[0];
        g.totals[1] = g.item_off[g.nlist];
        g.totals[2] = g.item_off[2 * g.nlist];
        g.totals[3] = g.item_off[SL];
        g.totals[4] = most;
        // items of stage r: [totals[5 + r], totals[6 + r]) -- what a list-scan launch reads as its bounds
        g.totals[5] = 0;
        g.totals[6] = g.item_off[g.nlist];
        g.totals[7] = g.item_off[2 * g.nlist];
        g.totals[8] = g.item_off[SL];
    }
}

__global__ __launch_bounds__(256) void ivf_regroup_scatter_kernel(RegroupBufs g) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= g.q * g.np) return;
    const uint32_t i = t / g.np, sl = t % g.np;
    if (sl >= min(g.probe_n[i], g.np)) return;
    const uint32_t l = g.probe[t];
    if (l >= g.nlist) return;
    const uint32_t e = rg_stage(g, sl) * g.nlist + l;
    const uint32_t at = g.pair_off[e] + atomicAdd(&g.fill[e], 1u);
    g.itemq[at] = make_uint2(i, g.lbase[i] + g.slot_rel[t]);
}

__global__ __launch_bounds__(256) void ivf_regroup_items_kernel(RegroupBufs g) {
    const uint32_t e = blockIdx.x * 256 + threadIdx.x;
    if (e >= 3 * g.nlist) return;
    const uint32_t l = e % g.nlist, c = g.cnt[e], len = g.list_len[l];
    if (c == 0 || len == 0) return;
    const uint32_t t0 = g.list_pos[l] / nh::TILE_ROWS, nt = (len + nh::TILE_ROWS - 1) / nh::TILE_ROWS;
    uint32_t at = g.item_off[e];
    for (uint32_t ch = 0; ch * g.run < nt; ++ch)
        for (uint32_t o = 0; o < c; o += g.item_queries)
            g.items[at++] = nh::IvfItem{t0 + ch * g.run, t0 + min(nt, (ch + 1) * g.run), g.pair_off[e] + o, min(g.item_queries, c - o), ch};
}

static needle_hip_status ivf_build_impl(needle_hip_ctx* c, const float* base, uint64_t n, uint32_t dim, const float* centroids,
                                        uint32_t nlist, const uint32_t* assignment, const uint32_t* valid_bitmask, uint32_t pq_m,
                                        const float* pq_codebooks, needle_hip_index** out);

extern "C" needle_hip_status needle_hip_ivf_build(needle_hip_ctx* c, const float* base, uint64_t n, uint32_t dim,
                                                  const float* centroids, uint32_t nlist, const uint32_t* assignment,
                                                  const uint32_t* valid_bitmask, needle_hip_index** out) {
    return ivf_build_impl(c, base, n, dim, centroids, nlist, assignment, valid_bitmask, 0, nullptr, out);
}

extern "C" needle_hip_status needle_hip_ivf_pq_build(needle_hip_ctx* c, const float* base, uint64_t n, uint32_t dim,
                                                     const float* centroids, uint32_t nlist, const uint32_t* assignment,
                                                     const uint32_t* valid_bitmask, uint32_t m, const float* codebooks,
                                                     needle_hip_index** out) {
    if (m == 0 || !codebooks || dim % m != 0)
        return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "need codebooks and dim (%u) divisible by the subvector count (%u)", dim, m);
    if (flat_scan_bf16_dpad(dim) == 0)
        return fail(NEEDLE_HIP_ERR_UNSUPPORTED, "IVF-PQ lists are scored through a bf16 copy of the decoded rows: at most 1536 dimensions");
    return ivf_build_impl(c, base, n, dim, centroids, nlist, assignment, valid_bitmask, m, codebooks, out);
}

static needle_hip_status ivf_build_impl(needle_hip_ctx* c, const float* base, uint64_t n, uint32_t dim, const float* centroids,
                                        uint32_t nlist, const uint32_t* assignment, const uint32_t* valid_bitmask, uint32_t pq_m,
                                        const float* pq_codebooks, needle_hip_index** out) {
    if (!c || !out) return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    if (dim == 0) return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "dimension must be positive");
    if (nlist == 0 || !centroids) return fail(NEEDLE_HIP_ERR_NOT_TRAINED, "Index not trained");  // IvfError::NotTrained
    if (n > 0 && !base) return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "base is null");
    if (n >= 0xf0000000ull) return fail(NEEDLE_HIP_ERR_UNSUPPORTED, "too many rows for one IVF index");
    HIP_TRY(hipSetDevice(c->device));

    // insert(): nearest centroid of every row, ivf.rs:474-482
    std::vector<uint32_t> assign;
    try {
        assign.resize(n);
    } catch (...) {
        return fail(NEEDLE_HIP_ERR_OUT_OF_MEMORY, "host allocation failed");
    }
    if (assignment) {
        for (uint64_t i = 0; i < n; ++i) {
            if (assignment[i] >= nlist) return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "assignment[%llu] = %u >= nlist", (unsigned long long)i, assignment[i]);
            assign[i] = assignment[i];
        }
    } else if (n > 0) {
        ST_TRY(needle_hip_kmeans_assign(c, NEEDLE_ARITH_DISTANCE_RS, base, n, centroids, nlist, dim, assign.data()));
    }

    needle_hip_index* ix = index_new();
    if (!ix) return fail(NEEDLE_HIP_ERR_OUT_OF_MEMORY, "host allocation failed");
    ix->ctx = c;
    ix->kind = KIND_IVF;
    ix->n = n;
    ix->dim = dim;
    ix->metric = NEEDLE_METRIC_EUCLIDEAN;  // ivf.rs:67: always euclidean_distance
    ix->arith = NEEDLE_ARITH_DISTANCE_RS;
    ix->nlist = nlist;
    ix->list_len.assign(nlist, 0);
    ix->list_pos.assign(nlist, 0);
    for (uint64_t i = 0; i < n; ++i) ix->list_len[assign[i]] += 1;
    uint64_t pos = 0;
    for (uint32_t l = 0; l < nlist; ++l) {
        ix->list_pos[l] = (uint32_t)pos;
        pos += round_up(ix->list_len[l], TILE_ROWS);
    }
    ix->list_tiles_desc.resize(nlist);
    for (uint32_t l = 0; l < nlist; ++l) ix->list_tiles_desc[l] = round_up(ix->list_len[l], TILE_ROWS) / TILE_ROWS;
    std::sort(ix->list_tiles_desc.begin(), ix->list_tiles_desc.end(), std::greater<uint32_t>());
    if (pos >= 0xfffffff0ull) { index_free(ix); return fail(NEEDLE_HIP_ERR_UNSUPPORTED, "padded IVF layout exceeds 2^32 positions"); }
    ix->n_pad = (uint32_t)pos;
    std::vector<uint32_t> pos2id((size_t)ix->n_pad, 0xffffffffu), fill(nlist, 0);
    for (uint64_t i = 0; i < n; ++i) {  // insertion order inside a list == row id order
        const uint32_t l = assign[i];
        pos2id[(size_t)ix->list_pos[l] + fill[l]++] = (uint32_t)i;
    }

    needle_hip_status st = NEEDLE_HIP_OK;
    float* d_src = nullptr;
    auto hip_fail = [&](hipError_t e, const char* what) {
        return fail(e == hipErrorOutOfMemory ? NEEDLE_HIP_ERR_OUT_OF_MEMORY : NEEDLE_HIP_ERR_HIP, "HIP error %d (%s) %s",
                    (int)e, hipGetErrorString(e), what);
    };
    hipError_t e = hipMalloc((void**)&ix->d_list_pos, sizeof(uint32_t) * nlist);
    if (e == hipSuccess) e = hipMalloc((void**)&ix->d_list_len, sizeof(uint32_t) * nlist);
    if (e == hipSuccess) e = hipMemcpyAsync(ix->d_list_pos, ix->list_pos.data(), sizeof(uint32_t) * nlist, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(ix->d_list_len, ix->list_len.data(), sizeof(uint32_t) * nlist, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess && ix->n_pad > 0) {
        e = hipMalloc((void**)&ix->d_pos2id, sizeof(uint32_t) * ix->n_pad);
        if (e == hipSuccess) e = hipMemcpyAsync(ix->d_pos2id, pos2id.data(), sizeof(uint32_t) * ix->n_pad, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipMalloc((void**)&ix->d_raw, (size_t)ix->n_pad * dim * sizeof(float));
        if (e == hipSuccess) e = hipMalloc((void**)&d_src, (size_t)n * dim * sizeof(float));
        if (e == hipSuccess) e = hipMemcpyAsync(d_src, base, (size_t)n * dim * sizeof(float), hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) {
            // validate the rows as inserted, then lay them out list by list
            uint32_t* d_bad = nullptr;
            e = hipMalloc((void**)&d_bad, sizeof(uint32_t));
            uint32_t bad = 0xffffffffu;
            if (e == hipSuccess) e = launch_validate_rows(d_src, n, dim, d_bad, c->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(&bad, d_bad, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
            if (d_bad) (void)hipFree(d_bad);
            if (e == hipSuccess && bad != 0xffffffffu)
                st = fail(NEEDLE_HIP_ERR_INVALID_VECTOR, "Invalid vector: row %u contains NaN or Inf", bad);
        }
        if (e == hipSuccess && st == NEEDLE_HIP_OK) {
            hipLaunchKernelGGL(ivf_gather_rows_kernel, dim3((ix->n_pad + 3) / 4), dim3(256), 0, c->stream, d_src,
                               ix->d_pos2id, ix->n_pad, dim, ix->d_raw);
            e = hipGetLastError();
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        }
    }
    if (d_src) (void)hipFree(d_src);
    if (e != hipSuccess) st = hip_fail(e, "building the IVF layout");
    if (st == NEEDLE_HIP_OK && pq_m) {
        // IVF-PQ (ivf.rs:488-491): a list keeps pq.encode(vector), not the vector.  Encode the list-ordered rows,
        // drop them, and build the operand of the list scan from the decoded codes.
        ix->pq_m = pq_m;
        ix->pq_dsub = dim / pq_m;
        const size_t cb_bytes = (size_t)pq_m * 256 * ix->pq_dsub * sizeof(float);
        e = hipMalloc((void**)&ix->d_codebooks, cb_bytes);
        if (e == hipSuccess) e = hipMemcpyAsync(ix->d_codebooks, pq_codebooks, cb_bytes, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess && ix->n_pad > 0) {
            e = hipMalloc((void**)&ix->d_codes, (size_t)ix->n_pad * pq_m + 16);
            if (e == hipSuccess) e = launch_pq_encode(ix->d_raw, ix->n_pad, dim, pq_m, ix->d_codebooks, nullptr, ix->d_codes, c->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
            (void)hipFree(ix->d_raw);
            ix->d_raw = nullptr;
        }
        if (e != hipSuccess) st = hip_fail(e, "encoding the lists");
        if (st == NEEDLE_HIP_OK && ix->n_pad > 0) st = ensure_bf16_copy(ix);  // decoded rows, -|row|^2/2, max norm, rounding error
    } else if (st == NEEDLE_HIP_OK) st = finish_operands(ix, false);
    if (st == NEEDLE_HIP_OK && ix->n_pad > 0 && ix->d_cinit) {
        hipLaunchKernelGGL(ivf_pad_seed_kernel, dim3((ix->n_pad + 255) / 256), dim3(256), 0, c->stream, ix->d_pos2id, ix->n_pad,
                           ix->d_cinit);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) st = hip_fail(e, "seeding the padding positions");
    }
    if (st == NEEDLE_HIP_OK) st = upload_mask(ix, valid_bitmask);
    if (st == NEEDLE_HIP_OK)
        st = needle_hip_flat_build(c, centroids, nlist, dim, NEEDLE_METRIC_EUCLIDEAN, NEEDLE_ARITH_DISTANCE_RS, nullptr,
                                   &ix->centroids);
    if (st != NEEDLE_HIP_OK) { index_free(ix); return st; }
    *out = ix;
    return NEEDLE_HIP_OK;
}

extern "C" needle_hip_status needle_hip_ivf_list_sizes(const needle_hip_index* ix, uint32_t* out_sizes) {
    if (!ix || !out_sizes || ix->kind != KIND_IVF) return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "not an IVF index");
    memcpy(out_sizes, ix->list_len.data(), sizeof(uint32_t) * ix->nlist);
    return NEEDLE_HIP_OK;
}

// Coarse step on device-resident queries: v_probe32 [q][np] list ids (u32), v_probe_n [q].
// set_only: the lists in any order (the search regroups them by list); otherwise ascending by (distance, list).
// defer: the search goes on without waiting to learn whether a query's probe set is still open (the caller reads
// ix->centroids->w_flagcnt at its own final synchronisation and re-runs those queries).
static needle_hip_status ivf_probe_device(needle_hip_index* ix, const float* d_queries, uint32_t q, uint32_t np, bool set_only,
                                          bool defer = false) {
    needle_hip_ctx* c = ix->ctx;
    HIP_TRY(ix->v_probe_ids.ensure((size_t)q * np * sizeof(uint64_t)));
    HIP_TRY(ix->v_probe_d.ensure((size_t)q * np * sizeof(float)));
    HIP_TRY(ix->v_probe_n.ensure((size_t)q * sizeof(uint32_t)));
    HIP_TRY(ix->v_probe32.ensure((size_t)q * np * sizeof(uint32_t)));
    {
        std::lock_guard g(ix->centroids->mu);
        // bf16 first pass over the centroids: the probe lists it returns are the exact ones all the same
        // (an fp32 first pass over a few thousand centroids measured slower: 5.3 against 3.9 ms per 10K-query batch)
        ST_TRY(flat_search_batch_device(ix->centroids, d_queries, q, np, NEEDLE_PRECISION_BF16, nullptr,
                                        ix->v_probe_ids.as<uint64_t>(), ix->v_probe_d.as<float>(),
                                        ix->v_probe_n.as<uint32_t>(), set_only, defer));
    }
    const uint32_t total = q * np;
    hipLaunchKernelGGL(ivf_narrow_probe_kernel, dim3((total + 255) / 256), dim3(256), 0, c->stream,
                       ix->v_probe_ids.as<uint64_t>(), total, ix->v_probe32.as<uint32_t>());
    HIP_TRY(hipGetLastError());
    return NEEDLE_HIP_OK;
}

extern "C" needle_hip_status needle_hip_ivf_probe(needle_hip_index* ix, const float* queries, uint32_t q,
                                                  uint32_t query_dim, uint32_t nprobe, uint32_t* out_list_ids,
                                                  uint32_t* out_counts) {
    if (!ix || ix->kind != KIND_IVF) return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "not an IVF index");
    if (q > 0 && query_dim != ix->dim)
        return fail(NEEDLE_HIP_ERR_DIMENSION_MISMATCH, "Dimension mismatch: expected %u, got %u", ix->dim, query_dim);
    if (q == 0) return NEEDLE_HIP_OK;
    if (!queries || !out_list_ids || !out_counts) return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "null argument");
    if (nprobe == 0) { memset(out_counts, 0, sizeof(uint32_t) * q); return NEEDLE_HIP_OK; }
    std::lock_guard g(ix->mu);
    needle_hip_ctx* c = ix->ctx;
    HIP_TRY(hipSetDevice(c->device));
    const uint32_t np = std::min(nprobe, ix->nlist);
    if (np > K_MAX) return fail(NEEDLE_HIP_ERR_UNSUPPORTED, "nprobe = %u exceeds %d", np, K_MAX);
    HIP_TRY(ix->w_qraw.ensure((size_t)q * ix->dim * sizeof(float)));
    HIP_TRY(hipMemcpyAsync(ix->w_qraw.p, queries, (size_t)q * ix->dim * sizeof(float), hipMemcpyHostToDevice, c->stream));
    ST_TRY(ivf_probe_device(ix, ix->w_qraw.as<float>(), q, np, false));
    std::vector<uint32_t> tmp((size_t)q * np);
    HIP_TRY(hipMemcpyAsync(tmp.data(), ix->v_probe32.p, tmp.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(out_counts, ix->v_probe_n.p, (size_t)q * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (uint32_t i = 0; i < q; ++i)
        for (uint32_t j = 0; j < nprobe; ++j)
            out_list_ids[(size_t)i * nprobe + j] = j < np ? tmp[(size_t)i * np + j] : 0xffffffffu;
    return NEEDLE_HIP_OK;
}

// Gathers the probe lists of the queries named by sel into a compact [n_sel][np] block (re-run of queries whose probe
// set the deferred coarse step left open).
__global__ __launch_bounds__(256) void ivf_gather_probe_kernel(const uint32_t* __restrict__ sel, uint32_t n_sel, uint32_t np,
                                                               const uint32_t* __restrict__ probe, const uint32_t* __restrict__ probe_n,
                                                               uint32_t* __restrict__ out_probe, uint32_t* __restrict__ out_n) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n_sel * np) return;
    const uint32_t i = t / np, j = t % np, src = sel[i];
    out_probe[t] = probe[(uint64_t)src * np + j];
    if (j == 0) out_n[i] = probe_n[src];
}

// Everything behind the coarse step: regrouping, list scan, merge, exact re-score, bound check.  Nothing in here waits
// for the device; h_flagged[0] (host) receives the number of queries the bound check left open once the caller has
// synchronised the stream (their indices: ix->w_flagged).  shared_* : the bf16 query operand, |q|^2 and rounding errors
// the coarse step prepared for the same queries (null: prepared here).
static needle_hip_status ivf_scan_probed(needle_hip_index* ix, const float* d_queries, uint32_t q, uint32_t k, uint32_t np, bool bf16,
                                         const uint32_t* d_filter, const uint32_t* d_probe, const uint32_t* d_probe_n,
                                         uint64_t* d_out_ids, float* d_out_dists, uint32_t* d_out_counts, const void* shared_qop,
                                         const double* shared_qn2, const float* shared_qerr, uint32_t* h_flagged) {
    needle_hip_ctx* c = ix->ctx;
    const bool pq = ix->pq_m != 0;
    const int scan_precision = bf16 ? NEEDLE_PRECISION_BF16 : NEEDLE_PRECISION_FP32;
    const uint32_t lpe = bf16 ? flat_scan_bf16_lists_per_slice() : 1;  // candidate lists per (query, list run): the bf16 kernel keeps one per lane group
    const uint32_t kprime = choose_kprime((uint32_t)std::min<uint64_t>(k, ix->n), scan_precision);

    // Regroup the (query, probed list) pairs by list ON THE DEVICE (ivf_regroup_* kernels above).  The host reads
    // nothing back: every size below is an upper bound it can derive from the list lengths, and the list-scan launches
    // take their real item ranges from device memory (workgroups beyond them leave at once).
    // Work items.  A list is cut into runs of at most `run` row tiles, so that a long list (hub centroid) becomes
    // many workgroups instead of one long one; every (query, probed list, run) owns a candidate list of its own,
    // numbered consecutively per query: query i owns lists list_base[i] .. list_base[i+1]-1.  `run` doubles until the
    // lists of any query (bounded by the nprobe longest lists of the index) fit the merge kernel's bookkeeping.
    // Thresholds start from a sample of every query's nearest list (ivf_seed_tau_kernel).
    const bool dbg_t = (c->tune.debug & 8u) != 0;
    const uint32_t max_lists_per_query = std::max<uint32_t>(np, 2048);  // bounds the list bookkeeping, not correctness
    const uint32_t item_queries = bf16 ? flat_scan_bf16_ivf_queries_per_block(flat_scan_bf16_dpad(ix->dim)) : (uint32_t)TILE_QUERIES;
    uint32_t run = 8;
    uint64_t worst = 0;  // most candidate lists one query can own: the runs of the np longest lists
    for (; run < (1u << 20); run *= 2) {
        worst = 0;
        for (uint32_t i = 0; i < np; ++i) worst += std::max<uint32_t>(1, (ix->list_tiles_desc[i] + run - 1) / run);
        if (worst <= max_lists_per_query) break;
    }
    uint64_t runs_all = 0;  // runs of all lists together, and the most runs of one list
    const uint32_t max_runs = std::max<uint32_t>(1, (ix->list_tiles_desc.empty() ? 0u : ix->list_tiles_desc[0] + run - 1) / run);
    for (uint32_t t : ix->list_tiles_desc) runs_all += std::max<uint32_t>(1, (t + run - 1) / run);
    // One launch covers all probe slots (measured on 1.25M x 384, nlist 4096, nprobe 32, 10K queries: 1.35 ms per batch
    // against 1.51 with slots 0-3 first and a threshold refresh behind them; every launch streams nearly all lists, and the
    // refresh saves fewer listed rows than the second pass over the lists costs).  ivf_split = s < nprobe: slots [0, s) first.
    uint32_t bounds[4] = {0, 0, std::min<uint32_t>(c->tune.ivf_split ? c->tune.ivf_split : np, np), np};
    // items of stage r: per (list, run) one item per item_queries pairs, rounded up once per (list, run) holding pairs
    uint64_t item_cap[3], items_cap = 0;
    for (int r = 0; r < 3; ++r) {
        const uint64_t pairs = (uint64_t)q * (bounds[r + 1] - bounds[r]);
        item_cap[r] = pairs == 0 ? 0 : std::min<uint64_t>(runs_all, pairs * max_runs) + (pairs / item_queries) * max_runs;
        items_cap += item_cap[r];
    }
    const uint64_t lists_cap = (uint64_t)q * worst * lpe;
    if (items_cap >= 0x7fffffffull || lists_cap >= 0x7fffffffull)
        return fail(NEEDLE_HIP_ERR_UNSUPPORTED, "IVF batch of %u queries x %u probes over these lists exceeds the work-item bookkeeping; "
                    "search fewer queries per call", q, np);
    RegroupBufs rg{};
    {
        const uint32_t SL = 3 * ix->nlist;
        const size_t words = (size_t)4 * SL + 2 + (size_t)q + (size_t)q + 1 + (size_t)q * np + 16;
        HIP_TRY(ix->v_regroup.ensure(words * sizeof(uint32_t)));
        uint32_t* w = ix->v_regroup.as<uint32_t>();
        rg.cnt = w; rg.fill = w + SL; rg.pair_off = w + 2 * (size_t)SL; rg.item_off = rg.pair_off + SL + 1;
        rg.qlists = rg.item_off + SL + 1; rg.lbase = rg.qlists + q; rg.slot_rel = rg.lbase + q + 1;
        rg.totals = rg.slot_rel + (size_t)q * np;
        HIP_TRY(hipMemsetAsync(rg.cnt, 0, (size_t)2 * SL * sizeof(uint32_t), c->stream));
    }
    HIP_TRY(ix->v_lbase.ensure((size_t)(q + 1) * sizeof(uint32_t)));
    HIP_TRY(ix->v_itemq.ensure(std::max<size_t>((size_t)q * np, 1) * sizeof(uint2)));
    HIP_TRY(ix->v_items.ensure(std::max<size_t>(items_cap, 1) * sizeof(IvfItem)));
    rg.lbase_dev = ix->v_lbase.as<uint32_t>();
    rg.probe = d_probe; rg.probe_n = d_probe_n; rg.list_len = ix->d_list_len; rg.list_pos = ix->d_list_pos;
    rg.q = q; rg.np = np; rg.nlist = ix->nlist; rg.run = run; rg.item_queries = item_queries; rg.lpe = lpe;
    rg.b1 = bounds[1]; rg.b2 = bounds[2];
    rg.itemq = ix->v_itemq.as<uint2>();
    rg.items = ix->v_items.as<IvfItem>();
    // Fork: the regrouping (four short kernels, one of them a single workgroup) stays on the search's stream; the query
    // operand, the masked seeds and the threshold sample need only the probe lists and run beside it on the side stream.
    HIP_TRY(ensure_side_stream(c));
    HIP_TRY(hipEventRecord(c->ev_fork, c->stream));
    HIP_TRY(hipStreamWaitEvent(c->side, c->ev_fork, 0));
    if (c->profiling) HIP_TRY(hipEventRecord(c->ev[0], c->stream));
    hipLaunchKernelGGL(ivf_regroup_count_kernel, dim3((q + 255) / 256), dim3(256), 0, c->stream, rg);
    hipLaunchKernelGGL(ivf_regroup_scan_kernel, dim3(1), dim3(1024), 0, c->stream, rg);
    hipLaunchKernelGGL(ivf_regroup_scatter_kernel, dim3((q * np + 255) / 256), dim3(256), 0, c->stream, rg);
    hipLaunchKernelGGL(ivf_regroup_items_kernel, dim3((3 * ix->nlist + 255) / 256), dim3(256), 0, c->stream, rg);
    HIP_TRY(hipGetLastError());
    const uint32_t total_lists = (uint32_t)lists_cap;  // device numbering: entry y owns lists lpe*y .. lpe*y + lpe-1

    if (bf16) ST_TRY(ensure_bf16_copy(ix));
    const uint32_t dpad = bf16 ? ix->dpad_bf16 : ix->dpad_f32, cap = list_cap(kprime);
    const size_t elt = bf16 ? 2 : 4;
    const bool shared = bf16 && shared_qop && shared_qn2 && shared_qerr;
    if (!shared) {
        HIP_TRY(ix->w_qop.ensure((size_t)q * dpad * elt));
        HIP_TRY(ix->w_qn2.ensure((size_t)q * sizeof(double)));
        if (bf16) HIP_TRY(ix->w_qerr.ensure((size_t)q * sizeof(float)));
    }
    HIP_TRY(ix->w_lists.ensure((size_t)std::max<uint32_t>(total_lists, 1) * cap * sizeof(uint2)));
    HIP_TRY(ix->w_lcnt.ensure((size_t)std::max<uint32_t>(total_lists, 1) * sizeof(uint32_t)));
    HIP_TRY(ix->w_cand.ensure((size_t)q * kprime * sizeof(uint32_t)));
    HIP_TRY(ix->w_candn.ensure((size_t)q * sizeof(uint32_t)));
    HIP_TRY(ix->w_tau.ensure((size_t)q * sizeof(float)));
    HIP_TRY(ix->w_flagged.ensure((size_t)q * sizeof(uint32_t)));
    HIP_TRY(ix->w_flagcnt.ensure(sizeof(uint32_t)));
    HIP_TRY(ix->w_tau0.ensure((size_t)q * sizeof(float)));
    if (ix->d_valid || d_filter) HIP_TRY(ix->w_cinit.ensure((size_t)ix->n_pad * sizeof(float)));
    HIP_TRY(hipMemsetAsync(ix->w_lcnt.p, 0, (size_t)std::max<uint32_t>(total_lists, 1) * sizeof(uint32_t), c->stream));
    HIP_TRY(hipMemsetAsync(ix->w_flagcnt.p, 0, sizeof(uint32_t), c->stream));
    // ---- side stream ----
    if (!shared) {
        HIP_TRY(hipMemsetAsync(ix->w_qop.p, 0, (size_t)q * dpad * elt, c->side));
        HIP_TRY(launch_row_prep(d_queries, q, ix->dim, dpad, false, bf16 ? nullptr : ix->w_qop.as<float>(),
                                bf16 ? ix->w_qop.as<uint16_t>() : nullptr, nullptr, nullptr, ix->w_qn2.as<double>(), nullptr,
                                bf16 ? ix->w_qerr.as<float>() : nullptr, nullptr, c->side));
    }
    const double* d_qn2 = shared ? shared_qn2 : ix->w_qn2.as<double>();
    const float* d_qerr = bf16 ? (shared ? shared_qerr : ix->w_qerr.as<float>()) : nullptr;

    ScanParams sp{};
    sp.base_op = bf16 ? (const void*)ix->d_op_bf16 : (const void*)ix->d_op_f32;
    sp.query_op = shared ? shared_qop : ix->w_qop.p;
    sp.cinit = ix->d_cinit;
    if (ix->d_valid || d_filter) {
        hipLaunchKernelGGL(ivf_masked_seed_kernel, dim3((ix->n_pad + 255) / 256), dim3(256), 0, c->side, ix->d_cinit, ix->d_pos2id,
                           ix->n_pad, ix->d_valid, d_filter, ix->w_cinit.as<float>());
        HIP_TRY(hipGetLastError());
        sp.cinit = ix->w_cinit.as<float>();
    }
    sp.valid = nullptr;  // dead rows and padding positions carry -inf seeds: the scan needs no mask
    sp.filter = nullptr;
    sp.lists = ix->w_lists.as<uint2>();
    sp.list_counts = ix->w_lcnt.as<uint32_t>();
    sp.n = ix->n_pad;
    sp.n_tiles = ix->n_pad / TILE_ROWS;
    sp.dpad = dpad;
    sp.kprime = kprime;
    sp.cap = cap;
    sp.items = ix->v_items.as<IvfItem>();
    sp.item_queries = ix->v_itemq.as<uint2>();
    sp.pos2id = ix->d_pos2id;
    sp.n_queries = q;
    sp.kseed = std::min<uint32_t>(kprime, (uint32_t)std::min<uint64_t>(k, ix->n));  // rank k: the tightest threshold k sampled rows still reach
    sp.tau0 = ix->w_tau0.as<float>();
    TauBand band{};
    band.qn2 = d_qn2;
    band.bmax = ix->max_norm;
    band.qerr = d_qerr;
    band.berr = ix->bf16_err;
    band.dim = ix->dim; band.metric = ix->metric; band.precision = scan_precision;
    const bool no_seed = c->tune.ivf_no_seed != 0;  // tuning: thresholds start at -inf
    if (no_seed) HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)ix->w_tau0.p, 0xff800000u, q, c->side));
    if (!no_seed) {
        // rows sampled per query: fewer rows shorten this kernel (its scoring steps are serial) and weaken the thresholds
        // (more rows listed by the scans); measured on 1.25M x 384, nlist 4096, nprobe 32: 256 rows 2.34 ms per batch,
        // 128 rows 2.20, 64 rows 2.18, 32 rows 2.25
        uint32_t sample_rows = 128;
        if (c->tune.ivf_seed_rows) sample_rows = std::min<uint32_t>(c->tune.ivf_seed_rows, IVF_SEED_ROWS);
        if (bf16)
            hipLaunchKernelGGL(ivf_seed_tau_kernel<uint16_t>, dim3(q), dim3(256), 0, c->side, d_queries, (const uint16_t*)ix->d_op_bf16,
                               dpad, sample_rows, ix->d_cinit, d_probe, d_probe_n, np, ix->nlist, ix->d_list_pos, ix->d_list_len, ix->d_pos2id,
                               ix->d_valid, d_filter, ix->dim, sp.kseed, band, ix->w_tau0.as<float>());
        else
            hipLaunchKernelGGL(ivf_seed_tau_kernel<float>, dim3(q), dim3(256), 0, c->side, d_queries, (const float*)ix->d_raw, ix->dim,
                               sample_rows, ix->d_cinit, d_probe, d_probe_n, np, ix->nlist, ix->d_list_pos, ix->d_list_len, ix->d_pos2id,
                               ix->d_valid, d_filter, ix->dim, sp.kseed, band, ix->w_tau0.as<float>());
    }
    HIP_TRY(hipGetLastError());
    // ---- join ----
    HIP_TRY(hipEventRecord(c->ev_join, c->side));
    HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_join, 0));
    bool launched = false;
    for (int r = 0; r < 3; ++r) {
        if (item_cap[r] == 0) continue;
        if (launched)
            HIP_TRY(launch_tau_refresh(sp, (uint32_t)worst * lpe, ix->w_tau0.as<float>(), band, c->stream, ix->v_lbase.as<uint32_t>()));
        ScanParams spr = sp;
        spr.item_bounds = rg.totals + 5 + r;
        HIP_TRY(launch_ivf_scan(scan_precision, spr, (uint32_t)item_cap[r], c->stream));
        launched = true;
    }
    if (c->profiling) HIP_TRY(hipEventRecord(c->ev[1], c->stream));

    PostParams pp{};
    pp.base_raw = ix->d_raw;
    pp.query_raw = d_queries;
    pp.qnorm2 = d_qn2;
    pp.lists = sp.lists;
    pp.list_counts = sp.list_counts;
    pp.list_base = ix->v_lbase.as<uint32_t>();
    pp.cand_ids = ix->w_cand.as<uint32_t>();
    pp.cand_n = ix->w_candn.as<uint32_t>();
    pp.tau_star = ix->w_tau.as<float>();
    pp.tau0 = sp.tau0;  // rows a launch did not list scored at most the threshold it started from
    pp.pos2id = ix->d_pos2id;
    pp.flagged = ix->w_flagged.as<uint32_t>();
    pp.flag_count = ix->w_flagcnt.as<uint32_t>();
    pp.out_ids = d_out_ids;
    pp.out_dists = d_out_dists;
    pp.out_counts = d_out_counts;
    pp.q = q; pp.n = ix->n_pad; pp.dim = ix->dim; pp.k = k; pp.kprime = kprime; pp.cap = cap; pp.n_splits = (uint32_t)worst * lpe;
    pp.metric = ix->metric; pp.arith = ix->arith; pp.precision = scan_precision;
    pp.max_norm = ix->max_norm;
    pp.qerr = d_qerr;
    pp.berr = ix->bf16_err;
    if (pq) {  // candidates are re-scored with ProductQuantizer::asymmetric_distance on the codes (ivf.rs:540)
        pp.pq_codes = ix->d_codes; pp.pq_books = ix->d_codebooks; pp.pq_m = ix->pq_m; pp.pq_dsub = ix->pq_dsub; pp.pq_asym = 1;
    }
    HIP_TRY(ix->w_cscore.ensure((size_t)q * kprime * sizeof(float)));
    pp.cand_scores = ix->w_cscore.as<float>();
    HIP_TRY(launch_merge_select(pp, c->stream));
    HIP_TRY(launch_prune_candidates(pp, band, c->stream));
    HIP_TRY(launch_rescore_finalize(pp, c->stream));
    HIP_TRY(hipMemcpyAsync(h_flagged, pp.flag_count, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    if (dbg_t) {
        uint32_t totals[9] = {0};
        HIP_TRY(hipMemcpyAsync(totals, rg.totals, sizeof(totals), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        std::vector<uint32_t> lc((size_t)totals[0] * lpe);
        HIP_TRY(hipMemcpy(lc.data(), ix->w_lcnt.p, lc.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
        uint64_t tot = 0;
        for (uint32_t v : lc) tot += std::min(v, kprime);
        fprintf(stderr, "[needle_hip] ivf: %u queries, %u items (grids sized for %llu), %u candidate lists (sized for %u; busiest query %u, "
                "run %u tiles), undecided %u, listed %.1f per query\n", q, totals[3], (unsigned long long)items_cap, totals[0] * lpe,
                total_lists, totals[4], run, *h_flagged, (double)tot / q);
    }
    return NEEDLE_HIP_OK;
}

needle_hip_status nh::ivf_search_batch_device(needle_hip_index* ix, const float* d_queries, uint32_t q, uint32_t k,
                                              uint32_t nprobe, int precision, const uint32_t* d_filter,
                                              uint64_t* d_out_ids, float* d_out_dists, uint32_t* d_out_counts) {
    needle_hip_ctx* c = ix->ctx;
    const uint32_t np = std::min(nprobe, ix->nlist);
    if (np == 0) {  // no list probed: no candidates
        HIP_TRY(launch_fill_empty(d_out_ids, d_out_dists, d_out_counts, q, k, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        return NEEDLE_HIP_OK;
    }
    if (np > K_MAX) return fail(NEEDLE_HIP_ERR_UNSUPPORTED, "nprobe = %u exceeds %d", np, K_MAX);
    // bf16 is a hint, as in the flat search: dimensions beyond the register-resident query block use fp32
    const bool pq = ix->pq_m != 0;  // lists of PQ codes: only the bf16 copy of the decoded rows exists, so bf16 it is
    bool bf16 = (precision == NEEDLE_PRECISION_BF16 || pq) && flat_scan_bf16_dpad(ix->dim) != 0;
    if (bf16 && choose_kprime((uint32_t)std::min<uint64_t>(k, ix->n), NEEDLE_PRECISION_BF16) > KPRIME_MAX) bf16 = false;
    const uint32_t kprime = choose_kprime((uint32_t)std::min<uint64_t>(k, ix->n), bf16 ? NEEDLE_PRECISION_BF16 : NEEDLE_PRECISION_FP32);
    const bool exact_only = precision == NEEDLE_PRECISION_EXACT_SCAN || kprime > KPRIME_MAX || (pq && !bf16);
    if (exact_only) {
        ST_TRY(ivf_probe_device(ix, d_queries, q, np, true));
        ST_TRY(run_exact(ix, d_queries, nullptr, q, k, d_filter, d_out_ids, d_out_dists, d_out_counts, ix->v_probe32.as<uint32_t>(),
                         ix->v_probe_n.as<uint32_t>(), np));
        HIP_TRY(hipStreamSynchronize(c->stream));
        std::lock_guard g(c->mu);
        c->m.fallback_queries += q;
        return NEEDLE_HIP_OK;
    }
    // The coarse step is enqueued and the search goes on behind it at once: whether the first-pass scores left some
    // query's probe set open is read at the ONE synchronisation at the end (no device -> host copy between the probe and
    // the merge); such queries -- none on typical data -- are then re-run with an exact coarse step.
    ST_TRY(ivf_probe_device(ix, d_queries, q, np, true, /*defer=*/true));
    needle_hip_index* cent = ix->centroids;
    // the coarse step's query operand (bf16, unnormalised: both steps rank by Euclidean distance) serves the list scan too
    const bool probe_bf16 = flat_scan_bf16_dpad(ix->dim) != 0 &&
                            choose_kprime(std::min(np, ix->nlist), NEEDLE_PRECISION_BF16) <= (uint32_t)KPRIME_MAX;
    const bool share = bf16 && probe_bf16 && cent->w_qop.p && cent->w_qn2.p && cent->w_qerr.p &&
                       cent->dpad_bf16 == flat_scan_bf16_dpad(ix->dim);
    uint32_t n_flagged = 0, n_probe_open = 0;
    ST_TRY(ivf_scan_probed(ix, d_queries, q, k, np, bf16, d_filter, ix->v_probe32.as<uint32_t>(), ix->v_probe_n.as<uint32_t>(), d_out_ids,
                           d_out_dists, d_out_counts, share ? cent->w_qop.p : nullptr, share ? cent->w_qn2.as<double>() : nullptr,
                           share ? cent->w_qerr.as<float>() : nullptr, &n_flagged));
    HIP_TRY(hipMemcpyAsync(&n_probe_open, cent->w_flagcnt.p, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->profiling) {
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
        std::lock_guard g(c->mu);
        c->m.list_scans += 1;  // (the probe's own first pass over the centroids is not timed separately)
        c->m.list_scan_ms_total += ms;
    }
    if (n_flagged > 0) {
        if (n_flagged > q) n_flagged = q;
        ST_TRY(run_exact(ix, d_queries, ix->w_flagged.as<uint32_t>(), n_flagged, k, d_filter, d_out_ids, d_out_dists, d_out_counts,
                         ix->v_probe32.as<uint32_t>(), ix->v_probe_n.as<uint32_t>(), np));
        HIP_TRY(hipStreamSynchronize(c->stream));
        std::lock_guard g(c->mu);
        c->m.fallback_queries += n_flagged;
    }
    if (n_probe_open > 0) {
        // These queries were searched with a probe set the coarse step had not settled: settle it with the exact scan of the
        // centroids, then search them again (gathered into a batch of their own) and put their answers in place.
        if (n_probe_open > q) n_probe_open = q;
        const uint32_t f = n_probe_open;
        const uint32_t* d_sel = cent->w_flagged.as<uint32_t>();
        ST_TRY(run_exact(cent, d_queries, d_sel, f, np, nullptr, ix->v_probe_ids.as<uint64_t>(), ix->v_probe_d.as<float>(),
                         ix->v_probe_n.as<uint32_t>()));
        const uint32_t total = q * np;
        hipLaunchKernelGGL(ivf_narrow_probe_kernel, dim3((total + 255) / 256), dim3(256), 0, c->stream, ix->v_probe_ids.as<uint64_t>(), total,
                           ix->v_probe32.as<uint32_t>());
        HIP_TRY(hipGetLastError());
        HIP_TRY(ix->r_idx.ensure((size_t)f * sizeof(uint32_t)));
        HIP_TRY(ix->r_q.ensure((size_t)f * ix->dim * sizeof(float)));
        HIP_TRY(ix->r_ids.ensure((size_t)f * k * sizeof(uint64_t)));
        HIP_TRY(ix->r_dists.ensure((size_t)f * k * sizeof(float)));
        HIP_TRY(ix->r_cnt.ensure((size_t)f * sizeof(uint32_t)));
        HIP_TRY(ix->v_probe_sel.ensure(((size_t)f * np + f) * sizeof(uint32_t)));
        uint32_t* sel_probe = ix->v_probe_sel.as<uint32_t>();
        uint32_t* sel_n = sel_probe + (size_t)f * np;
        HIP_TRY(hipMemcpyAsync(ix->r_idx.p, d_sel, (size_t)f * sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
        HIP_TRY(launch_gather_rows(d_queries, ix->r_idx.as<uint32_t>(), f, ix->dim, ix->r_q.as<float>(), c->stream));
        hipLaunchKernelGGL(ivf_gather_probe_kernel, dim3((f * np + 255) / 256), dim3(256), 0, c->stream, ix->r_idx.as<uint32_t>(), f, np,
                           ix->v_probe32.as<uint32_t>(), ix->v_probe_n.as<uint32_t>(), sel_probe, sel_n);
        HIP_TRY(hipGetLastError());
        uint32_t again = 0;
        ST_TRY(ivf_scan_probed(ix, ix->r_q.as<float>(), f, k, np, bf16, d_filter, sel_probe, sel_n, ix->r_ids.as<uint64_t>(),
                               ix->r_dists.as<float>(), ix->r_cnt.as<uint32_t>(), nullptr, nullptr, nullptr, &again));
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (again > 0) {
            if (again > f) again = f;
            ST_TRY(run_exact(ix, ix->r_q.as<float>(), ix->w_flagged.as<uint32_t>(), again, k, d_filter, ix->r_ids.as<uint64_t>(),
                             ix->r_dists.as<float>(), ix->r_cnt.as<uint32_t>(), sel_probe, sel_n, np));
        }
        HIP_TRY(launch_scatter_results(ix->r_idx.as<uint32_t>(), f, k, ix->r_ids.as<uint64_t>(), ix->r_dists.as<float>(),
                                       ix->r_cnt.as<uint32_t>(), d_out_ids, d_out_dists, d_out_counts, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        std::lock_guard g(c->mu);
        c->m.refined_queries += f;
    }
    return NEEDLE_HIP_OK;
}
