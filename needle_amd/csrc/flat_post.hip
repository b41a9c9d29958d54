// Everything behind the first pass: merging the per-slice candidate lists,
// re-scoring candidates in the reference's scalar order, the error-bound check
// that proves the candidate set sufficient, the exact scan that serves every
// query the check rejects, the one-query distance operator and the multi-shard
// merge.
#include "nh_common.h"
#include "nh_exact.h"
#include "nh_kernels.h"
#include "nh_scan.h"
#include "nh_select.h"

#include <cstring>

#include <rocprim/rocprim.hpp>

namespace nh {

// ---------------------------------------------------------------------------
// merge_select: one wave per query.  Gathers the query's candidate lists into LDS, keeps the kprime best by
// first-pass score and records the score of the last one kept (tau*): every row that is NOT a candidate scored
// <= tau*.  Fewer than kprime entries in total means no list discarded anything: every listed row is a
// candidate, tau* = the threshold the scan started from (or -inf).
// The LDS holds MERGE_KEYS keys whatever the number of lists: the lists are taken in groups, and when the next
// group would not fit, what has been gathered so far is first cut down to its kprime best (the cut only ever
// removes keys below the final kprime-th one).
// ---------------------------------------------------------------------------
// Sizes of a launch (dynamic LDS: 8 bytes per key, 4 per list offset): the smaller they are, the more queries a CU
// works on at once -- the kernel is a chain of short dependent steps per query, so it lives on occupancy.
// Measured (10K queries, 64 candidates kept of ~130 listed in 128 lists): 2304 keys + 1024 offsets 0.18 ms,
// 768 + 256 0.07 ms.  A query that lists more than the buffer holds takes the gather in groups.
constexpr uint32_t MERGE_KEYS_MAX = 2048 + KPRIME_MAX;

// kprime-th largest of keys[0 .. total) (total >= kprime), found bit by bit; all keys are distinct
// per-lane counts over pairs of keys (a count does not care which lane sees which key), one cross-lane sum per bit
template <int PER>
__device__ inline uint64_t kth_largest_lds_regs(const uint64_t* keys, uint32_t total, uint32_t kprime, uint32_t lane) {
    uint64_t r[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) r[u] = ((uint32_t)u * 64u < total && (uint32_t)u * 64u + lane < total) ? keys[u * 64 + lane] : 0ull;
    return wave_kth_largest_regs64<PER>(r, total, kprime);
}
// the keys go to registers once: PER per lane (12 cover the usual gather of 768 keys, 36 the largest buffer)
template <int PER>
__device__ inline uint64_t wave_kth_largest_lds(const uint64_t* keys, uint32_t total, uint32_t kprime, uint32_t lane) {
    return kth_largest_lds_regs<PER>(keys, total, kprime, lane);
}

template <int PER>
__global__ __launch_bounds__(64) void merge_select_kernel(PostParams p, uint32_t MERGE_KEYS, uint32_t MERGE_FAST_LISTS) {
    extern __shared__ __attribute__((aligned(16))) uint64_t mkeys[];  // [MERGE_KEYS], then loff[MERGE_FAST_LISTS + 1]
    const uint32_t q = blockIdx.x, lane = threadIdx.x;
    const uint64_t first = p.list_base ? p.list_base[q] : (uint64_t)q * p.n_splits;
    const uint32_t n_lists = p.list_base ? p.list_base[q + 1] - p.list_base[q] : p.n_splits;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    uint32_t* loff = reinterpret_cast<uint32_t*>(mkeys + MERGE_KEYS);
    uint32_t total = 0;
    bool done = false;
    if (n_lists <= MERGE_FAST_LISTS) {
        // common case: all counts at once (independent loads), then every lane copies entries lane, lane+64, ... of
        // the concatenation, finding the list of an entry by bisection of the offsets
        for (uint32_t l0 = 0; l0 < n_lists; l0 += 64) {
            const uint32_t li = l0 + lane;
            const uint32_t c = li < n_lists ? min(p.list_counts[first + li], p.kprime) : 0u;
            uint32_t incl = c;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t t = (uint32_t)__shfl_up((int)incl, off);
                if (lane >= (uint32_t)off) incl += t;
            }
            if (li < n_lists) loff[li] = total + incl - c;
            total += (uint32_t)__shfl((int)incl, 63);
        }
        if (lane == 0) loff[n_lists] = total;
        __syncthreads();
        // The concatenation is taken MERGE_KEYS entries at a time (usually all of it at once); between two pieces the
        // buffer is cut down to its kprime best.  Four entries per lane and step: their lists are found first, then
        // their loads are in flight together.
        const uint32_t all = total;
        uint32_t kept = 0;
        for (uint32_t e0 = 0; e0 < all;) {
            const uint32_t chunk = min(all - e0, MERGE_KEYS - kept);
            for (uint32_t e = lane; e < chunk; e += 256) {
                uint2 v[4];
#pragma unroll
                for (uint32_t u = 0; u < 4; ++u) {
                    const uint32_t g = e0 + e + 64u * u;
                    v[u] = make_uint2(0u, 0u);
                    if (e + 64u * u < chunk) {
                        uint32_t lo = 0, hi = n_lists;  // largest list index with loff <= g
                        while (hi - lo > 1) {
                            const uint32_t mid = (lo + hi) >> 1;
                            if (loff[mid] <= g) lo = mid; else hi = mid;
                        }
                        v[u] = p.lists[(first + lo) * p.cap + (g - loff[lo])];
                    }
                }
#pragma unroll
                for (uint32_t u = 0; u < 4; ++u)
                    if (e + 64u * u < chunk) mkeys[kept + e + 64u * u] = cand_key(v[u].x, v[u].y);
            }
            kept += chunk;
            e0 += chunk;
            if (e0 < all) {  // wave-uniform; the buffer is full (kept == MERGE_KEYS > kprime)
                __syncthreads();
                const uint64_t kth = wave_kth_largest_lds<PER>(mkeys, kept, p.kprime, lane);
                uint32_t base = 0;
                for (uint32_t i = 0; i < kept; i += 64) {  // in-place: a chunk is read whole before any of its slots is written
                    const uint64_t key = (i + lane < kept) ? mkeys[i + lane] : 0ull;
                    const bool keep = (i + lane < kept) && key >= kth;
                    const uint64_t m = __builtin_amdgcn_ballot_w64(keep);
                    __syncthreads();
                    if (keep) mkeys[base + __builtin_popcountll(m & lt_mask)] = key;
                    base += __builtin_popcountll(m);
                    __syncthreads();
                }
                kept = base;  // == kprime: the keys are distinct
            }
        }
        total = kept;
        done = true;
    }
    // lists per group: their entries (<= kprime each) fit beside kprime kept keys
    const uint32_t group = min(64u, max(1u, (MERGE_KEYS - p.kprime) / p.kprime));
    for (uint32_t l0 = 0; !done && l0 < n_lists; l0 += group) {
        // lane l copies list l0 + l: the lists are short, so the copies run side by side
        const uint32_t li = l0 + lane;
        const uint32_t c = (lane < group && li < n_lists) ? min(p.list_counts[first + li], p.kprime) : 0u;
        uint32_t incl = c;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t t = (uint32_t)__shfl_up((int)incl, off);
            if (lane >= (uint32_t)off) incl += t;
        }
        const uint32_t gtotal = (uint32_t)__shfl((int)incl, 63);
        if (total + gtotal > MERGE_KEYS) {  // wave-uniform; then total > kprime, because gtotal <= MERGE_KEYS - kprime
            __syncthreads();
            const uint64_t kth = wave_kth_largest_lds<PER>(mkeys, total, p.kprime, lane);
            uint32_t base = 0;
            for (uint32_t i = 0; i < total; i += 64) {  // in-place: a chunk is read whole before any of its slots is written
                const uint64_t key = (i + lane < total) ? mkeys[i + lane] : 0ull;
                const bool keep = (i + lane < total) && key >= kth;
                const uint64_t m = __builtin_amdgcn_ballot_w64(keep);
                __syncthreads();
                if (keep) mkeys[base + __builtin_popcountll(m & lt_mask)] = key;
                base += __builtin_popcountll(m);
                __syncthreads();
            }
            total = base;  // == kprime
        }
        const uint32_t dst = total + incl - c;
        const uint2* l = p.lists + (first + li) * p.cap;
        for (uint32_t i = 0; i < c; ++i) {
            const uint2 v = l[i];
            mkeys[dst + i] = cand_key(v.x, v.y);
        }
        total += gtotal;
    }
    __syncthreads();
    uint64_t prefix = 0;
    float tau = -INFINITY;
    if (total >= p.kprime) {
        prefix = wave_kth_largest_lds<PER>(mkeys, total, p.kprime, lane);
        tau = ord2f((uint32_t)(prefix >> 32));
    }
    uint32_t base = 0;
    for (uint32_t i = 0; i < total; i += 64) {
        const bool keep = (i + lane < total) && (mkeys[i + lane] >= prefix);
        const uint64_t m = __builtin_amdgcn_ballot_w64(keep);
        if (keep) {
            const uint64_t slot = (uint64_t)q * p.kprime + base + __builtin_popcountll(m & lt_mask);
            p.cand_ids[slot] = ~(uint32_t)mkeys[i + lane];
            if (p.cand_scores) p.cand_scores[slot] = ord2f((uint32_t)(mkeys[i + lane] >> 32));
        }
        base += __builtin_popcountll(m);
    }
    if (lane == 0) {
        // rows the main pass never listed scored <= its starting threshold
        if (p.tau0) tau = fmaxf(tau, p.tau0[q]);
        p.cand_n[q] = base;
        p.tau_star[q] = tau;
    }
}

hipError_t launch_merge_select(const PostParams& p, hipStream_t stream) {
    if (p.q == 0) return hipSuccess;
    if (p.kprime == 0 || p.kprime > (uint32_t)KPRIME_MAX) return hipErrorInvalidValue;
    // what a query lists grows with k (k = 100: about a thousand entries); room for them in one gather when it is cheap
    uint32_t keys = std::min<uint32_t>(MERGE_KEYS_MAX, std::max<uint32_t>(768u, 10u * p.kprime + 128u));
    // lists per query the one-shot gather handles: all of them for a flat scan (their number is known here)
    uint32_t fast = p.list_base ? 256u : std::min<uint32_t>(1024u, std::max<uint32_t>(64u, p.n_splits));
    if (p.tune_merge_keys && p.tune_merge_lists) {  // planning override
        const uint32_t a = p.tune_merge_keys, b = p.tune_merge_lists;
        if (a >= 2u * KPRIME_MAX && a <= 4096 && b >= 1 && b <= 2048) { keys = a / 2 * 2; fast = b; }
    }
    const size_t lds = (size_t)keys * sizeof(uint64_t) + ((size_t)fast + 1) * sizeof(uint32_t);
    static_assert(MERGE_KEYS_MAX <= 64 * 36, "register image of the merge buffer");
    if (keys <= 64u * 12u) hipLaunchKernelGGL(merge_select_kernel<12>, dim3(p.q), dim3(64), lds, stream, p, keys, fast);
    else hipLaunchKernelGGL(merge_select_kernel<36>, dim3(p.q), dim3(64), lds, stream, p, keys, fast);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// dense_select: one wave per query over a row of the dense score matrix: the kprime best live rows by (score, lower
// id); tau* = the kprime-th score, -inf when every live row is a candidate.
// The scores are narrowed before the exact selection: every lane keeps the T = ceil(kprime / 64) largest of its own
// scores; the kprime-th largest of those 64 T values (the pivot) is a score at least kprime rows reach, so the rows at
// or above it contain the answer and are usually few more than kprime.  A second sweep over the row (it comes from L2)
// compacts them into LDS and the exact bit-by-bit selection runs over them.  The row itself is never held in LDS -- a
// workgroup needs 8 KB, so a CU keeps a wave per query in flight on every slot it has.  A row that leaves more than
// DS_SURV rows above the pivot (equal scores), or whose live rows sit in too few lanes for a pivot, takes the selection
// over the whole row, re-read from memory for every bit.
// ---------------------------------------------------------------------------
constexpr uint32_t DS_SURV = 1024;  // survivors the narrowed selection holds
constexpr int DS_T = (KPRIME_MAX + 63) / 64;
constexpr int DS_HELD = 16;         // steps of 256 scores a wave keeps in registers between its two sweeps

// kprime-th largest of ord[0 .. total) bit by bit (total >= kprime); zeros sort below every live score
__device__ inline uint32_t wave_kth_largest_u32(const uint32_t* ord, uint32_t total, uint32_t kprime, uint32_t lane) {
    const uint4* o4 = reinterpret_cast<const uint4*>(ord);
    const uint32_t t4 = (total + 3u) / 4u;  // the caller pads the array with zeros up to a multiple of 256 entries
    if (t4 <= 64u * 4u) {  // up to 1024 values: four 16-byte reads per lane, then everything from registers
        uint32_t r[16];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint4 v = ((uint32_t)u * 64u + lane < t4) ? o4[u * 64 + lane] : uint4{0u, 0u, 0u, 0u};
            r[4 * u] = v.x; r[4 * u + 1] = v.y; r[4 * u + 2] = v.z; r[4 * u + 3] = v.w;
        }
        // (register u covers values of the u/4-th block of 256: its ballots are skipped once the block is past the end)
        uint32_t kth = 0;
        for (int bit = 31; bit >= 0; --bit) {
            const uint32_t trial = kth | (1u << bit);
            uint32_t cnt = 0;
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if ((uint32_t)(u / 4) * 256u < total) cnt += __builtin_popcountll(__builtin_amdgcn_ballot_w64(r[u] >= trial));
            if (cnt >= kprime) kth = trial;
        }
        return kth;
    }
    uint32_t kth = 0;
    for (int bit = 31; bit >= 0; --bit) {
        const uint32_t trial = kth | (1u << bit);
        uint32_t cnt = 0;  // wave-uniform
        for (uint32_t i0 = 0; i0 < t4; i0 += 64) {
            const uint4 v = (i0 + lane < t4) ? o4[i0 + lane] : uint4{0u, 0u, 0u, 0u};
            cnt += __builtin_popcountll(__builtin_amdgcn_ballot_w64(v.x >= trial)) + __builtin_popcountll(__builtin_amdgcn_ballot_w64(v.y >= trial)) +
                   __builtin_popcountll(__builtin_amdgcn_ballot_w64(v.z >= trial)) + __builtin_popcountll(__builtin_amdgcn_ballot_w64(v.w >= trial));
        }
        if (cnt >= kprime) kth = trial;
    }
    return kth;
}

// order-preserving image of four consecutive scores (rows i .. i+3, i a multiple of 4 below the padded row length);
// 0 -- below every real score, f2ord(-inf) = 0x007fffff -- for rows that are masked out or beyond n
__device__ inline void dense_ord4(const PostParams& p, const float* __restrict__ sc, uint32_t i, bool masked, uint32_t (&o4)[4]) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(sc + i);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t r = i + j;
        bool ok = r < p.n;
        if (masked && ok && p.valid) ok = bit_set(p.valid, r);
        if (masked && ok && p.filter) ok = bit_set(p.filter, r);
        if (ok && p.extra_n && v[j] == -INFINITY) ok = false;  // a row every query re-scores anyway (seeded -inf): not a candidate twice
        o4[j] = ok ? f2ord(v[j]) : 0u;
    }
}

__global__ __launch_bounds__(64) void dense_select_kernel(PostParams p) {
    __shared__ __attribute__((aligned(16))) uint32_t s_ord[DS_SURV];
    __shared__ uint32_t s_row[DS_SURV];
    const uint32_t q = blockIdx.x, lane = threadIdx.x;
    const float* sc = p.dense_scores + (uint64_t)q * p.dense_ld;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    const uint32_t T = (p.kprime + 63u) / 64u;
    uint32_t live = 0;
    uint32_t top[DS_T];  // this lane's T largest scores, descending
#pragma unroll
    for (int t = 0; t < DS_T; ++t) top[t] = 0u;
    // four scores per lane and step, several steps in flight (a step per dependent global load would leave the wave
    // waiting on memory 64 times over)
    const uint32_t nq4 = (p.n + 3u) & ~3u;  // <= dense_ld: the matrix rows are padded to whole 128-row tiles
    const bool masked = p.valid != nullptr || p.filter != nullptr;
    // A row of up to DS_HELD * 256 scores is read from memory ONCE: all its loads are issued together and the lane's
    // 4 * DS_HELD values stay in registers for the second sweep (a wave is otherwise a chain of memory latencies: one
    // per step of either sweep).
    uint32_t held[DS_HELD][4];
    const bool in_regs = nq4 <= (uint32_t)DS_HELD * 256u;  // wave-uniform
    auto offer = [&](uint32_t o) {
        live += o != 0u ? 1u : 0u;
#pragma unroll
        for (int t = 0; t < DS_T; ++t) {  // insertion: o sinks to its place, the smallest falls out
            const uint32_t hi = max(top[t], o);
            o = min(top[t], o);
            top[t] = hi;
        }
    };
    if (in_regs) {
#pragma unroll
        for (int u = 0; u < DS_HELD; ++u) {
            const uint32_t i = (uint32_t)u * 256u + lane * 4u;
#pragma unroll
            for (int j = 0; j < 4; ++j) held[u][j] = 0u;
            if (i < nq4) dense_ord4(p, sc, i, masked, held[u]);
        }
#pragma unroll
        for (int u = 0; u < DS_HELD; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) offer(held[u][j]);
    } else {
#pragma unroll 4
        for (uint32_t i = lane * 4; i < nq4; i += 256) {
            uint32_t o4[4];
            dense_ord4(p, sc, i, masked, o4);
#pragma unroll
            for (int j = 0; j < 4; ++j) offer(o4[j]);
        }
    }
#pragma unroll
    for (int off = 32; off; off >>= 1) live += (uint32_t)__shfl_xor((int)live, off);
    // pivot: the kprime-th largest of the lanes' top-T scores (64 T >= kprime of them); 0 when the live rows sit in
    // too few lanes, or when every live row is a candidate anyway
    uint32_t pivot = 0;
    if (live > p.kprime) {
        for (int bit = 31; bit >= 0; --bit) {
            const uint32_t trial = pivot | (1u << bit);
            uint32_t cnt = 0;
#pragma unroll
            for (int t = 0; t < DS_T; ++t)
                cnt += __builtin_popcountll(__builtin_amdgcn_ballot_w64((uint32_t)t < T && top[t] >= trial));
            if (cnt >= p.kprime) pivot = trial;
        }
    }
    // second sweep: the rows at or above the pivot (all live rows when there is none), compacted in sweep order
    const uint32_t floor_ord = max(pivot, 1u);
    uint32_t surv = 0;
    auto sweep4 = [&](const uint32_t (&o4)[4], uint32_t i) {  // every lane takes every step: the ballots count whole waves
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool keep = o4[j] >= floor_ord;
            const uint64_t m = __builtin_amdgcn_ballot_w64(keep);
            const uint32_t pos = surv + __builtin_popcountll(m & lt_mask);
            if (keep && pos < DS_SURV) { s_ord[pos] = o4[j]; s_row[pos] = i + j; }
            surv += __builtin_popcountll(m);
        }
    };
    if (in_regs) {
#pragma unroll
        for (int u = 0; u < DS_HELD; ++u)
            if ((uint32_t)u * 256u < nq4) sweep4(held[u], (uint32_t)u * 256u + lane * 4u);
    } else {
        for (uint32_t i0 = 0; i0 < nq4; i0 += 256) {
            const uint32_t i = i0 + lane * 4;
            uint32_t o4[4] = {0u, 0u, 0u, 0u};
            if (i < nq4) dense_ord4(p, sc, i, masked, o4);
            sweep4(o4, i);
        }
    }
    if (live <= p.kprime && surv <= DS_SURV) {  // every live row is a candidate (kprime <= KPRIME_MAX < DS_SURV)
        __syncthreads();
        for (uint32_t e = lane; e < surv; e += 64) {
            const uint64_t slot = (uint64_t)q * p.kprime + e;
            p.cand_ids[slot] = s_row[e];
            if (p.cand_scores) p.cand_scores[slot] = ord2f(s_ord[e]);
        }
        if (lane == 0) { p.cand_n[q] = surv; p.tau_star[q] = -INFINITY; }
        return;
    }
    if (surv <= DS_SURV) {
        for (uint32_t e = surv + lane; e < ((surv + 255u) & ~255u) && e < DS_SURV; e += 64) s_ord[e] = 0u;
        __syncthreads();
        const uint32_t kth = wave_kth_largest_u32(s_ord, surv, p.kprime, lane);
        uint32_t gt = 0;
        for (uint32_t i = 0; i < surv; i += 64) {
            const bool g = (i + lane < surv) && (s_ord[i + lane] > kth);
            gt += __builtin_popcountll(__builtin_amdgcn_ballot_w64(g));
        }
        const uint32_t need_ties = p.kprime - gt;  // rows scoring exactly the kprime-th score: the lowest ids win
        uint32_t base = 0;
        for (uint32_t i = 0; i < surv; i += 64) {
            const uint32_t e = i + lane;
            const uint32_t o = e < surv ? s_ord[e] : 0u;
            const uint32_t row = e < surv ? s_row[e] : 0u;
            // survivors are not in row order: a tie's rank is the number of ties with a lower row id
            uint32_t rank = 0;
            const bool tie = e < surv && o == kth;
            if (tie)
                for (uint32_t j = 0; j < surv; ++j) rank += (s_ord[j] == kth && s_row[j] < row) ? 1u : 0u;
            const bool keep = o > kth || (tie && rank < need_ties);
            const uint64_t m = __builtin_amdgcn_ballot_w64(keep);
            if (keep) {
                const uint64_t slot = (uint64_t)q * p.kprime + base + __builtin_popcountll(m & lt_mask);
                p.cand_ids[slot] = row;
                if (p.cand_scores) p.cand_scores[slot] = ord2f(o);
            }
            base += __builtin_popcountll(m);
        }
        if (lane == 0) { p.cand_n[q] = base; p.tau_star[q] = ord2f(kth); }
        return;
    }
    // The wide case: more rows at or above the pivot than the LDS holds.  The kprime-th largest score over the whole
    // row, every bit from a fresh sweep of the scores; then one more sweep in row order, ties to the lowest ids.
    uint32_t kth = 0;
    for (int bit = 31; bit >= 0; --bit) {
        const uint32_t trial = kth | (1u << bit);
        uint32_t cnt = 0;
        for (uint32_t i = lane * 4; i < nq4; i += 256) {
            uint32_t o4[4];
            dense_ord4(p, sc, i, masked, o4);
#pragma unroll
            for (int j = 0; j < 4; ++j) cnt += o4[j] >= trial ? 1u : 0u;
        }
#pragma unroll
        for (int off = 32; off; off >>= 1) cnt += (uint32_t)__shfl_xor((int)cnt, off);
        if (cnt >= p.kprime) kth = trial;
    }
    uint32_t gt = 0;
    for (uint32_t i = lane * 4; i < nq4; i += 256) {
        uint32_t o4[4];
        dense_ord4(p, sc, i, masked, o4);
#pragma unroll
        for (int j = 0; j < 4; ++j) gt += o4[j] > kth ? 1u : 0u;
    }
#pragma unroll
    for (int off = 32; off; off >>= 1) gt += (uint32_t)__shfl_xor((int)gt, off);
    const uint32_t need_ties = p.kprime - gt;
    uint32_t base = 0, ties = 0;
    for (uint32_t i = 0; i < p.n; i += 64) {  // row order: lane = row within the chunk
        const uint32_t r = i + lane;
        uint32_t o = 0u;
        if (r < p.n) {
            bool ok = true;
            if (masked && p.valid) ok = bit_set(p.valid, r);
            if (masked && ok && p.filter) ok = bit_set(p.filter, r);
            o = ok ? f2ord(sc[r]) : 0u;
        }
        const bool tie = o == kth && o != 0u;
        const uint64_t tm = __builtin_amdgcn_ballot_w64(tie);
        const bool keep = o > kth || (tie && ties + __builtin_popcountll(tm & lt_mask) < need_ties);
        const uint64_t m = __builtin_amdgcn_ballot_w64(keep);
        if (keep) {
            const uint64_t slot = (uint64_t)q * p.kprime + base + __builtin_popcountll(m & lt_mask);
            p.cand_ids[slot] = r;
            if (p.cand_scores) p.cand_scores[slot] = ord2f(o);
        }
        base += __builtin_popcountll(m);
        ties += __builtin_popcountll(tm);
    }
    if (lane == 0) { p.cand_n[q] = base; p.tau_star[q] = ord2f(kth); }
}

hipError_t launch_dense_select(const PostParams& p, hipStream_t stream) {
    if (p.q == 0) return hipSuccess;
    if (p.n > DENSE_MAX_ROWS || !p.dense_scores || p.kprime == 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(dense_select_kernel, dim3(p.q), dim3(64), 0, stream, p);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// prune_candidates: one wave per query, between selection and exact re-scoring.
// With s_k the k-th largest first-pass score among the candidates and eps the score uncertainty, a candidate scoring
// below s_k - 2 eps is worse than k other rows whatever the rounding did, so it needs no exact distance.  Dropped
// candidates become "rows not re-scored": tau_star rises to the best dropped score and the bound check covers them
// like any other non-candidate.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void prune_candidates_kernel(PostParams p, TauBand band) {
    const uint32_t q = blockIdx.x, lane = threadIdx.x;
    const uint32_t nc = min(p.cand_n[q], p.kprime);
    if (nc <= p.k) return;  // every candidate may be a hit
    constexpr int PER = (KPRIME_MAX + 63) / 64;
    uint32_t ord[PER], id[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const uint32_t e = i * 64 + lane;
        ord[i] = e < nc ? f2ord(p.cand_scores[(uint64_t)q * p.kprime + e]) : 0u;
        id[i] = e < nc ? p.cand_ids[(uint64_t)q * p.kprime + e] : 0u;
    }
    uint32_t kth = 0;  // k-th largest score, bit by bit
    for (int bit = 31; bit >= 0; --bit) {
        const uint32_t trial = kth | (1u << bit);
        uint32_t cnt = 0;
#pragma unroll
        for (int i = 0; i < PER; ++i) cnt += __builtin_popcountll(__builtin_amdgcn_ballot_w64(i * 64 + lane < nc && ord[i] >= trial));
        if (cnt >= p.k) kth = trial;
    }
    const float cut = (float)((double)ord2f(kth) - 2.0 * band_eps(band, q));
    if (!(cut == cut)) return;
    const uint32_t cut_ord = f2ord(cut);
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    uint32_t base = 0, best_dropped = 0;
    bool any_dropped = false;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const bool in = i * 64 + lane < nc;
        const bool keep = in && ord[i] >= cut_ord;
        const uint64_t m = __builtin_amdgcn_ballot_w64(keep);
        if (keep) p.cand_ids[(uint64_t)q * p.kprime + base + __builtin_popcountll(m & lt_mask)] = id[i];
        base += __builtin_popcountll(m);
        if (in && !keep) { best_dropped = max(best_dropped, ord[i]); any_dropped = true; }
    }
    if (__builtin_amdgcn_ballot_w64(any_dropped) == 0) return;
    for (int off = 32; off; off >>= 1) best_dropped = max(best_dropped, (uint32_t)__shfl_xor((int)best_dropped, off));
    if (lane == 0) {
        p.cand_n[q] = base;
        p.tau_star[q] = fmaxf(p.tau_star[q], ord2f(best_dropped));
    }
}

hipError_t launch_prune_candidates(const PostParams& p, const TauBand& band, hipStream_t stream) {
    if (p.q == 0 || !p.cand_scores) return hipSuccess;
    hipLaunchKernelGGL(prune_candidates_kernel, dim3(p.q), dim3(64), 0, stream, p, band);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// The bound check.  tau is the first-pass score threshold (larger = closer),
// dk the exact distance of the k-th hit.  Returns true when no non-candidate
// row can beat or tie the k-th hit:
//     (lower bound on any non-candidate's reference distance)  >  dk.
// u = 2^-24; a length-D f32 sum of products errs by at most D*u*sum|x_i y_i| in
// any order (FMA chain, sequential, groups of four); gam doubles that and adds
// the few extra roundings of normalisation, sqrt and divide.  A bf16 operand
// pair adds 2^-8 relative per product.
// ---------------------------------------------------------------------------
__device__ inline bool bound_check(int metric, double lowp_abs, float tau, float dk, double qn2,
                                   double bmax, uint32_t dim) {
    if (tau == -INFINITY) return true;  // every live row was re-scored
    if (!(dk == dk) || !(tau == tau)) return false;
    const double u = 5.9604644775390625e-08;
    const double gam = 2.0 * ((double)dim + 16.0) * u;
    const double lowp = lowp_abs;  // absolute first-pass error from operand rounding (bf16_score_err)
    const double qn = sqrt(qn2);
    const double t = (double)tau, d = (double)dk;
    switch (metric) {
    case NEEDLE_METRIC_COSINE: {
        // degenerate magnitudes: squares may underflow in the reference's f32 norms
        if (!(qn > 1e-15) || !(qn < 1e15)) return false;
        const double eps = lowp + 2.0 * gam /*first pass*/ + 2.0 * gam /*reference value*/;
        return (1.0 - t - eps) > d;
    }
    case NEEDLE_METRIC_COSINE_NORMALIZED: {
        const double scale = qn * bmax;
        if (!(scale < 1e17)) return false;
        return (1.0 - t - lowp - 2.0 * gam * scale) > d;
    }
    case NEEDLE_METRIC_DOT_PRODUCT: {
        const double scale = qn * bmax;
        if (!(scale < 1e17)) return false;
        return (-t - lowp - 2.0 * gam * scale) > d;
    }
    case NEEDLE_METRIC_EUCLIDEAN: {
        // score = q.b - |b|^2/2  ->  d^2 = |q|^2 - 2*score
        if (!(qn < 1e17) || !(bmax < 1e17)) return false;
        const double eps_s = lowp + gam * qn * bmax + gam * 0.5 * bmax * bmax;
        const double d2 = qn2 - 2.0 * t - 2.0 * eps_s;
        if (!(d2 > 0.0)) return false;
        return sqrt(d2) * (1.0 - gam) > d;
    }
    default: return false;
    }
}

// Scalar-quantized rows: score = sum_j (q_j/scale_j) c_j - |v|^2/2 with v = decoded row, so the
// squared distance is qconst - 2*score, qconst = sum_j (q_j^2 - 2 q_j min_j).  vmax = max |v|,
// mnorm = |min_vals|.  Both the first-pass score and the reference's own value (decode, subtract,
// square, sequential sum) err by at most gam * (|q| + |v| + |min|)^2-sized terms.
// lowp_e: absolute error the operand rounding adds to a first-pass score (bf16 query operand against exact
// codes: 2^-9 |q/scale| |c| with |c| <= 255 sqrt(dim)); 0 for f32 operands.
__device__ inline bool bound_check_sq8(float tau, float dk, double qconst, double vmax, double mnorm, uint32_t dim,
                                       double lowp_e) {
    if (tau == -INFINITY) return true;
    if (!(dk == dk) || !(tau == tau)) return false;
    const double u = 5.9604644775390625e-08;
    const double gam = 2.0 * ((double)dim + 16.0) * u;
    // |q| <= sqrt(|qconst| + 2|q||min|): bound |q| through qconst = |q|^2 - 2 q.min  ->  |q| <= |min| + sqrt(|min|^2 + |qconst|)
    const double qn = mnorm + sqrt(mnorm * mnorm + fabs(qconst));
    if (!(qn < 1e17) || !(vmax < 1e17)) return false;
    const double eps_s = gam * (qn * (vmax + mnorm) + 0.5 * vmax * vmax) + lowp_e;
    const double lb2 = qconst - 2.0 * (double)tau - 2.0 * eps_s;
    if (!(lb2 > 0.0)) return false;
    const double lb = sqrt(lb2);
    const double e_ref = gam * (lb + 2.0 * vmax + mnorm) * (lb + 2.0 * vmax + mnorm);
    const double lb2r = lb2 - e_ref;
    if (!(lb2r > 0.0)) return false;
    return sqrt(lb2r) * (1.0 - gam) > (double)dk;
}

// ---------------------------------------------------------------------------
// rescore_finalize: one workgroup per query, one thread per candidate.
// Reference-order distance of every candidate, rank by (distance, id), emit the
// k best, run the bound check on the k-th.
// ---------------------------------------------------------------------------
// The candidate rows are staged through LDS a block of dimensions at a time (coalesced reads of
// whole row segments), then thread j walks candidate j's staged segment in dimension order, so
// the arithmetic is the scalar sequence of nh_exact.h continued across the blocks.
struct RefAcc { float a, b, c; };  // running sums: a = dot | squared distance, b = |q|^2, c = |row|^2

enum RefKind { RK_RS_COS, RK_RS_DOT, RK_RS_L2, RK_C4_COS, RK_C4_DOT, RK_C4_L2, RK_L1, RK_HAMMING, RK_CHEB, RK_PQ_TABLE, RK_SV_COS };

__device__ inline int ref_kind(int metric, int arith) {
    // the compute service's kernels (src/services/compute/gpu_kernels.rs:345-358): its cosine is one loop over three
    // accumulators from +0.0 with a denominator test; its euclidean and dot product are distance.rs's sums
    if (arith == NEEDLE_ARITH_SERVICE && metric == NEEDLE_METRIC_COSINE) return RK_SV_COS;
    if (arith == NEEDLE_ARITH_ACCEL_CHUNK4)
        return metric == NEEDLE_METRIC_COSINE ? RK_C4_COS : (metric == NEEDLE_METRIC_EUCLIDEAN ? RK_C4_L2 : RK_C4_DOT);
    switch (metric) {
    case NEEDLE_METRIC_COSINE: return RK_RS_COS;
    case NEEDLE_METRIC_EUCLIDEAN: return RK_RS_L2;
    case NEEDLE_METRIC_MANHATTAN: return RK_L1;
    case NEEDLE_METRIC_HAMMING: return RK_HAMMING;
    case NEEDLE_METRIC_CHEBYSHEV: return RK_CHEB;
    default: return RK_RS_DOT;  // CosineNormalized, DotProduct
    }
}

// dims [i0, i1) of the vectors, staged at qs[0 .. i1-i0) and bs[0 .. i1-i0); i0 is a multiple of 4.
// RK_PQ_TABLE: `dim` carries the subspace width instead; s.b is the running sum of the current subspace (from -0.0),
// folded into s.a (from +0.0) at each subspace boundary -- compute_distance_table + distance_with_table,
// src/indexing/quantization.rs:468-503.
template <int KIND>
__device__ inline void ref_accumulate(RefAcc& s, const float* qs, const float* bs, uint32_t i0, uint32_t i1, uint32_t dim) {
    if (KIND == RK_PQ_TABLE) {
        for (uint32_t i = 0; i < i1 - i0; ++i) {
            const float d = qs[i] - bs[i]; const float pd = d * d; s.b = s.b + pd;
            if ((i0 + i + 1) % dim == 0) { s.a = s.a + s.b; s.b = -0.0f; }
        }
    } else if (KIND == RK_RS_COS || KIND == RK_SV_COS) {  // distance.rs:153-164; gpu_kernels.rs:345-350 (same loop body)
        for (uint32_t i = 0; i < i1 - i0; ++i) {
            const float x = qs[i], y = bs[i];
            const float pxy = x * y; s.a = s.a + pxy;
            const float pxx = x * x; s.b = s.b + pxx;
            const float pyy = y * y; s.c = s.c + pyy;
        }
    } else if (KIND == RK_RS_DOT) {  // distance.rs:193-196, 264-266
        for (uint32_t i = 0; i < i1 - i0; ++i) { const float pxy = qs[i] * bs[i]; s.a = s.a + pxy; }
    } else if (KIND == RK_RS_L2) {  // distance.rs:249-257
        for (uint32_t i = 0; i < i1 - i0; ++i) { const float d = qs[i] - bs[i]; const float pd = d * d; s.a = s.a + pd; }
    } else if (KIND == RK_L1) {  // distance.rs:360-362
        for (uint32_t i = 0; i < i1 - i0; ++i) s.a = s.a + fabsf(qs[i] - bs[i]);
    } else if (KIND == RK_HAMMING) {  // distance.rs:387-395 (a count: exact in f32 below 2^24 dimensions)
        for (uint32_t i = 0; i < i1 - i0; ++i) s.a = s.a + (((qs[i] != 0.0f) != (bs[i] != 0.0f)) ? 1.0f : 0.0f);
    } else if (KIND == RK_CHEB) {  // distance.rs:415-423: fold(0.0, f32::max), max drops NaN
        for (uint32_t i = 0; i < i1 - i0; ++i) {
            const float d = fabsf(qs[i] - bs[i]);
            s.a = (d != d) ? s.a : ((s.a != s.a) ? d : (d > s.a ? d : s.a));
        }
    } else {  // gpu/common.rs:515-581: groups of four, then the remainder one by one
        const uint32_t grouped = (dim / 4) * 4;
        uint32_t i = i0;
        for (; i + 4 <= min(i1, grouped); i += 4) {
            const float* x = qs + (i - i0);
            const float* y = bs + (i - i0);
            if (KIND == RK_C4_L2) {
                const float d0 = x[0] - y[0], d1 = x[1] - y[1], d2 = x[2] - y[2], d3 = x[3] - y[3];
                const float s0 = d0 * d0, s1 = d1 * d1, s2 = d2 * d2, s3 = d3 * d3;
                const float g = ((s0 + s1) + s2) + s3;
                s.a = s.a + g;
            } else {
                const float p0 = x[0] * y[0], p1 = x[1] * y[1], p2 = x[2] * y[2], p3 = x[3] * y[3];
                const float g = ((p0 + p1) + p2) + p3;
                s.a = s.a + g;
                if (KIND == RK_C4_COS) {
                    const float a0 = x[0] * x[0], a1 = x[1] * x[1], a2 = x[2] * x[2], a3 = x[3] * x[3];
                    const float ga = ((a0 + a1) + a2) + a3;
                    s.b = s.b + ga;
                    const float b0 = y[0] * y[0], b1 = y[1] * y[1], b2 = y[2] * y[2], b3 = y[3] * y[3];
                    const float gb = ((b0 + b1) + b2) + b3;
                    s.c = s.c + gb;
                }
            }
        }
        for (; i < i1; ++i) {
            const float x = qs[i - i0], y = bs[i - i0];
            if (KIND == RK_C4_L2) {
                const float d = x - y; const float pd = d * d; s.a = s.a + pd;
            } else {
                const float pxy = x * y; s.a = s.a + pxy;
                if (KIND == RK_C4_COS) {
                    const float pxx = x * x; s.b = s.b + pxx;
                    const float pyy = y * y; s.c = s.c + pyy;
                }
            }
        }
    }
}

// The walk over a staged segment, 16 dimensions at a time: the values are fetched with 16-byte LDS reads into registers
// first (eight reads in flight), then folded in order -- a loop that reads LDS element by element waits out one LDS
// latency per dimension.  qs and bs are 16-byte aligned, i0 a multiple of 4.
template <int KIND>
__device__ inline void ref_accumulate_staged(RefAcc& s, const float* qs, const float* bs, uint32_t i0, uint32_t i1, uint32_t dim) {
    const uint32_t len = i1 - i0;
    uint32_t i = 0;
    for (; i + 16 <= len; i += 16) {
        float qv[16], bv[16];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const f32x4 a = *(const f32x4*)(qs + i + 4 * t);
            const f32x4 b = *(const f32x4*)(bs + i + 4 * t);
#pragma unroll
            for (int e = 0; e < 4; ++e) { qv[4 * t + e] = a[e]; bv[4 * t + e] = b[e]; }
        }
        ref_accumulate<KIND>(s, qv, bv, i0 + i, i0 + i + 16, dim);
    }
    if (i < len) ref_accumulate<KIND>(s, qs + i, bs + i, i0 + i, i1, dim);
}

__device__ inline float ref_finish(int kind, const RefAcc& s) {
    switch (kind) {
    case RK_SV_COS: {  // gpu_kernels.rs:348-349
        const float den = sqrtf(s.b) * sqrtf(s.c);
        if (den < 1.1920929e-07f) return 1.0f;
        return 1.0f - (s.a / den);
    }
    case RK_RS_COS:
    case RK_C4_COS: {
        const float na = sqrtf(s.b);
        if (kind == RK_C4_COS && na == 0.0f) return 1.0f;  // gpu/mod.rs:717-762: query norm first
        const float nb = sqrtf(s.c);
        if (na == 0.0f || nb == 0.0f) return 1.0f;
        const float den = na * nb;
        return 1.0f - (s.a / den);
    }
    case RK_RS_L2:
    case RK_PQ_TABLE:
    case RK_C4_L2: return sqrtf(s.a);
    default: return s.a;  // caller applies 1 - dot or -dot
    }
}

constexpr int RS_THREADS = 256;
#ifndef RS_UNROLL
#define RS_UNROLL 8  // candidate rows a wave has in flight while staging (4: 2.5 % slower on the filtered 768-d search, 16: slower everywhere)
#endif

// SET_ONLY (Euclidean rows, candidates with their first-pass scores): the caller wants the SET of the k nearest rows,
// not their order or distances (the probe step of an inverted-file search).  With s_k, s_k+1 the k-th and (k+1)-th
// largest candidate scores and m = set_margin (score uncertainty on both sides + the reference's own rounding),
//   * a row scoring below s_k - m has a larger reference distance than each of the k rows scoring >= s_k: it is out
//     (candidates, and -- through tau* < s_k - m, else the query is flagged -- every row that is no candidate);
//   * a candidate scoring above s_k+1 + m has a smaller reference distance than every row scoring <= s_k+1, so
//     fewer than k rows precede it: it is in, with no exact distance computed;
//   * the candidates in between are re-scored exactly and the best (k - certain) of them by (distance, id) join.
// Output: the ids in out_ids (certain ones by descending score, then the chosen ones by distance), distances unset.
__device__ inline double set_margin(const TauBand& band, uint32_t q) {
    const double u = 5.9604644775390625e-08;
    const double gam = 2.0 * ((double)band.dim + 16.0) * u;
    const double qn = sqrt(band.qn2[q]), s = qn + band.bmax;
    // |score - (q.b - |b|^2/2)| <= eps for both rows; d_ref within (1 +- gam) of the true distance, d^2 <= (|q| + |b|)^2
    return 2.0 * band_eps(band, q) + 2.05 * gam * s * s;
}

// AVX = 8 / 16 / 4: the distances follow the order of a reference build with AVX2 / AVX-512 / NEON (exact_avx: that many
// lane sums per accumulator, kept across the dimension blocks; dc is then a multiple of AVX).
template <bool SET_ONLY, int AVX = 0>
__global__ __launch_bounds__(RS_THREADS) void rescore_finalize_kernel(PostParams p, uint32_t dc, uint32_t rcap, TauBand band) {
    extern __shared__ __attribute__((aligned(16))) float stage[];  // [dc] query block, then [rcap][dc + 4] row blocks
    __shared__ uint64_t skey[KPRIME_MAX];
    __shared__ uint32_t spos[KPRIME_MAX];
    __shared__ uint32_t s_ord[SET_ONLY ? KPRIME_MAX : 1];
    __shared__ uint32_t s_misc[4];  // SET_ONLY: ordinals of s_k and s_k+1, number of certain / in-between candidates
    const uint32_t q = blockIdx.x, j = threadIdx.x;
    const uint32_t n_own = min(p.cand_n[q], p.kprime);
    uint32_t nc = n_own;
    if (!SET_ONLY && p.extra_n) nc = min(n_own + *p.extra_n, (uint32_t)KPRIME_MAX);  // (the host keeps kprime + extras within KPRIME_MAX)
    const uint32_t k_eff = min(p.k, nc);
    uint32_t emit = k_eff, emit_base = 0;  // hits still to be chosen by exact distance, and their first output slot
    uint64_t key = ~0ull;
    float d = 0.0f;
    uint32_t id = 0, pos = 0;
    if (j < nc) {
        pos = j < n_own ? p.cand_ids[(uint64_t)q * p.kprime + j] : p.extra_ids[j - n_own];
        id = p.pos2id ? p.pos2id[pos] : pos;
    }
    if (j < KPRIME_MAX) spos[j] = pos;
    if (SET_ONLY) {
        const uint32_t ord = j < nc ? f2ord(p.cand_scores[(uint64_t)q * p.kprime + j]) : 0u;
        if (j < KPRIME_MAX) s_ord[j] = ord;
        if (j < 4) s_misc[j] = 0u;
        __syncthreads();
        uint32_t srank = 0;  // rank by (score descending, position ascending)
        if (j < nc)
            // eight entries per step, their LDS reads in flight together (entries past nc hold ordinal 0: never counted)
            for (uint32_t i0 = 0; i0 < nc; i0 += 8) {
                uint32_t o8[8], p8[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { o8[u] = s_ord[i0 + u]; p8[u] = spos[i0 + u]; }
#pragma unroll
                for (int u = 0; u < 8; ++u) srank += (o8[u] > ord || (o8[u] == ord && p8[u] < pos)) ? 1u : 0u;
            }
        if (j < nc && nc >= p.k && srank + 1 == p.k) s_misc[0] = ord;
        if (j < nc && srank == p.k) s_misc[1] = ord;
        __syncthreads();
        const double m = set_margin(band, q);
        const float tau = p.tau_star[q];
        bool ok = (m == m) && m < 1e300 && (tau == tau);
        int cls = 0;  // 2: certainly a hit, 1: decided by its exact distance, 0: out
        if (nc < p.k) {  // fewer candidates than hits wanted: they are all hits, provided every live row is a candidate
            ok = ok && tau == -INFINITY;
            cls = j < nc ? 1 : 0;
        } else {
            const double lo = (double)ord2f(s_misc[0]) - m;
            ok = ok && (tau == -INFINITY || (double)tau < lo);
            if (j < nc) {
                const double sc = (double)ord2f(ord);
                cls = sc < lo ? 0 : 1;
                if (nc > p.k && sc > (double)ord2f(s_misc[1]) + m) cls = 2;
            }
        }
        if (cls == 2) atomicAdd(&s_misc[2], 1u);
        if (cls == 1) atomicAdd(&s_misc[3], 1u);
        __syncthreads();
        if (!ok) {  // block-uniform: the exact scan decides this query
            if (j == 0) {
                p.flagged[atomicAdd(p.flag_count, 1u)] = q;
                // until then the query has no hits: a caller that goes on before the exact scan has run (the inverted-file
                // search does not wait for its coarse step) must not take the unwritten slots for row ids
                p.out_counts[q] = 0;
            }
            return;
        }
        const uint32_t n_certain = s_misc[2], n_open = s_misc[3];
        emit = k_eff - n_certain;  // certain candidates hold the first n_certain score ranks, so n_certain <= k
        emit_base = n_certain;
        if (cls == 2) p.out_ids[(uint64_t)q * p.k + srank] = id;
        if (emit == n_open || emit == 0) {
            // the in-between candidates all join (they follow the certain ones in score rank), or none is needed
            if (cls == 1 && emit != 0) p.out_ids[(uint64_t)q * p.k + srank] = id;
            for (uint32_t r = k_eff + j; r < p.k; r += blockDim.x) {
                p.out_ids[(uint64_t)q * p.k + r] = ~0ull;
                p.out_dists[(uint64_t)q * p.k + r] = INFINITY;
            }
            if (j == 0) p.out_counts[q] = k_eff;
            return;
        }
        // the in-between candidates, compacted in score order, are the rows the exact pass walks
        if (cls == 1) { skey[srank - n_certain] = ((uint64_t)id << 32) | pos; }
        __syncthreads();
        nc = n_open;
        pos = 0; id = 0;
        if (j < nc) { pos = (uint32_t)skey[j]; id = (uint32_t)(skey[j] >> 32); }
        __syncthreads();
        if (j < KPRIME_MAX) spos[j] = pos;
    }
    if (p.pq_codes) {
        if (j < nc)
            d = canon_distance(p.pq_asym ? exact_pq_asym(p.query_raw + (uint64_t)q * p.dim, p.pq_codes + (uint64_t)pos * p.pq_m,
                                                         p.pq_books, p.pq_m, p.pq_dsub)
                                         : exact_pq(p.query_raw + (uint64_t)q * p.dim, p.pq_codes + (uint64_t)pos * p.pq_m, p.pq_books,
                                                    p.pq_m, p.pq_dsub));
    } else if (p.sq_codes) {
        if (j < nc)
            d = canon_distance(exact_sq8(p.query_raw + (uint64_t)q * p.dim, p.sq_codes + (uint64_t)pos * p.sq_stride,
                                         p.sq_min, p.sq_scale, p.dim));
    } else {
        const int kind = ref_kind(p.metric, p.arith);
        RefAcc acc;
        const float zero = p.arith == NEEDLE_ARITH_ACCEL_CHUNK4 ? 0.0f : -0.0f;  // Rust's f32 Sum folds from -0.0
        acc.a = zero; acc.b = zero; acc.c = zero;
        AvxAccT<(AVX ? AVX : 8)> va, vq, vb;  // AVX: q.b (or the differences), q.q, b.b
        if (AVX) { avx_init(va); avx_init(vq); avx_init(vb); }
        float* qs = stage;
        float* rows = stage + dc;
        const uint32_t ld = dc + 4;  // rows 16-byte aligned, 4 banks apart: the 16-byte reads of thread j's walk over row j are conflict free
        const uint32_t wave = j >> 6, lane = j & 63;
        const float* qsrc = p.query_raw + (uint64_t)q * p.dim;
        // rcap candidate rows are staged at a time (the LDS image is sized for rcap rows): thread j's candidate belongs to
        // exactly one such group, so its running sums live through that group's pass over the dimensions only
        for (uint32_t g0 = 0; g0 < nc; g0 += rcap) {
        const uint32_t gn = min(rcap, nc - g0);
        for (uint32_t c0 = 0; c0 < p.dim; c0 += dc) {
            const uint32_t len = min(dc, p.dim - c0);
            __syncthreads();  // the previous block has been consumed (and spos is visible)
            for (uint32_t i = j; i < len; i += RS_THREADS) qs[i] = qsrc[c0 + i];
            // a wave stages rows wave, wave + 4, ...: RS_UNROLL rows (two loads each per lane, dc <= 128) are in flight before
            // the first LDS store, so the gather is bound by bandwidth rather than by one row's latency at a time
            constexpr uint32_t RS_WAVES = RS_THREADS / 64;
            for (uint32_t r0 = wave; r0 < gn; r0 += RS_WAVES * RS_UNROLL) {
                float v[RS_UNROLL][2];
#pragma unroll
                for (uint32_t u = 0; u < RS_UNROLL; ++u) {
                    const uint32_t r = min(r0 + RS_WAVES * u, gn - 1);
                    const float* src = p.base_raw + (uint64_t)spos[g0 + r] * p.dim + c0;
                    v[u][0] = lane < len ? src[lane] : 0.0f;
                    v[u][1] = lane + 64 < len ? src[lane + 64] : 0.0f;
                }
#pragma unroll
                for (uint32_t u = 0; u < RS_UNROLL; ++u) {
                    const uint32_t r = r0 + RS_WAVES * u;
                    if (r < gn) {
                        if (lane < len) rows[r * ld + lane] = v[u][0];
                        if (lane + 64 < len) rows[r * ld + lane + 64] = v[u][1];
                    }
                }
            }
            __syncthreads();
            if (AVX) {
                if (j >= g0 && j < g0 + gn) {
                    const float* bs = rows + (j - g0) * ld;
                    if (p.metric == NEEDLE_METRIC_EUCLIDEAN) avx_accumulate<AVX_L2>(va, qs, bs, c0, c0 + len, p.dim);
                    else if (p.metric == NEEDLE_METRIC_MANHATTAN) avx_accumulate<AVX_L1>(va, qs, bs, c0, c0 + len, p.dim);
                    else avx_accumulate<AVX_DOT>(va, qs, bs, c0, c0 + len, p.dim);
                    if (p.metric == NEEDLE_METRIC_COSINE) {
                        avx_accumulate<AVX_DOT>(vq, qs, qs, c0, c0 + len, p.dim);
                        avx_accumulate<AVX_DOT>(vb, bs, bs, c0, c0 + len, p.dim);
                    }
                }
            } else if (j >= g0 && j < g0 + gn) {
                const float* bs = rows + (j - g0) * ld;
                switch (kind) {
                case RK_RS_COS: ref_accumulate_staged<RK_RS_COS>(acc, qs, bs, c0, c0 + len, p.dim); break;
                case RK_RS_DOT: ref_accumulate_staged<RK_RS_DOT>(acc, qs, bs, c0, c0 + len, p.dim); break;
                case RK_RS_L2: ref_accumulate_staged<RK_RS_L2>(acc, qs, bs, c0, c0 + len, p.dim); break;
                case RK_C4_COS: ref_accumulate<RK_C4_COS>(acc, qs, bs, c0, c0 + len, p.dim); break;
                case RK_C4_DOT: ref_accumulate<RK_C4_DOT>(acc, qs, bs, c0, c0 + len, p.dim); break;
                default: ref_accumulate<RK_C4_L2>(acc, qs, bs, c0, c0 + len, p.dim); break;
                }
            }
        }
        }
        if (j < nc) {
            float v = ref_finish(kind, acc);
            if (kind == RK_RS_DOT || kind == RK_C4_DOT)
                v = p.metric == NEEDLE_METRIC_COSINE_NORMALIZED ? 1.0f - v : -v;
            if (AVX) v = avx2_distance(p.metric, avx_finish(va, p.dim), avx_finish(vq, p.dim), avx_finish(vb, p.dim));
            d = canon_distance(v);
        }
    }
    if (j < nc) key = hit_key(d, id);
    if (j < KPRIME_MAX) skey[j] = key;
    __syncthreads();
    if (j < nc) {
        uint32_t rank = 0;
        for (uint32_t i0 = 0; i0 < nc; i0 += 8) {  // entries past nc hold the largest key
            uint64_t k8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) k8[u] = skey[i0 + u];
#pragma unroll
            for (int u = 0; u < 8; ++u) rank += (k8[u] < key) ? 1u : 0u;
        }
        if (rank < emit) {
            p.out_ids[(uint64_t)q * p.k + emit_base + rank] = id;
            p.out_dists[(uint64_t)q * p.k + emit_base + rank] = d;
        }
        if (!SET_ONLY && rank + 1 == k_eff) {
            const bool ok = p.sq_codes
                                ? bound_check_sq8(p.tau_star[q], d, p.qnorm2[q], (double)p.max_norm, (double)p.aux_norm, p.dim,
                                                  p.qaux ? 1.000001 * sqrt(p.qaux[q]) * p.berr : 0.0)
                                : bound_check(p.metric,
                                              bf16_score_err(p.metric, sqrt(p.qnorm2[q]), (double)p.max_norm,
                                                             p.qerr ? (double)p.qerr[q] : -1.0, p.berr),
                                              p.tau_star[q], d, p.qnorm2[q], (double)p.max_norm, p.dim);
            if (!ok) p.flagged[atomicAdd(p.flag_count, 1u)] = q;
        }
    }
    for (uint32_t r = k_eff + j; r < p.k; r += blockDim.x) {
        p.out_ids[(uint64_t)q * p.k + r] = ~0ull;
        p.out_dists[(uint64_t)q * p.k + r] = INFINITY;
    }
    if (j == 0) p.out_counts[q] = k_eff;
}

hipError_t launch_rescore_finalize(const PostParams& p, hipStream_t stream, const TauBand* set_only_band) {
    if (p.q == 0) return hipSuccess;
    if (set_only_band && (p.metric != NEEDLE_METRIC_EUCLIDEAN || !p.cand_scores || p.sq_codes || p.pq_codes || p.pos2id ||
                          avx_lanes(p.arith) != 0))
        return hipErrorInvalidValue;
    // Rows staged at a time (rcap) and block of dimensions per stage (dc, a multiple of 4: groups of four never straddle
    // blocks).  A workgroup is a chain of dependent gathers, so what counts is how many of them a CU holds: the image is
    // kept near 16 KB (64 rows x 64 dimensions measured best of 32 .. 128 dimensions with every candidate staged at once).
    // After pruning a query rarely keeps more than 64 candidates; the set-only form re-scores a dozen.
    uint32_t rcap = std::min<uint32_t>(p.kprime, set_only_band ? 32u : 64u);
    uint32_t dc = set_only_band ? 128u : 64u;
    // a batch small enough to be resident all at once even with the larger image is bound by its chain of staging rounds,
    // not by how many workgroups a CU holds: half as many rounds (1M x 768, 10 % filter, k = 100, 1K queries: 0.81 -> 0.76 ms)
    if (!set_only_band && p.q <= 1024) dc = 128u;
    // few hits wanted: after pruning a query keeps some two dozen candidates, so 32 rows x 128 dimensions -- the same image --
    // serves most queries in half the rounds (10K queries, k = 10: flat search 6.12 -> 6.05 ms, IVF 1.26 -> 1.24 ms)
    if (!set_only_band && p.kprime <= 96) { rcap = std::min<uint32_t>(rcap, 32u); dc = 128u; }
    if (p.tune_rs_cap >= 4 && p.tune_rs_cap <= p.kprime) rcap = p.tune_rs_cap;  // planning overrides
    if (p.tune_rs_dc / 4 * 4 >= 4 && p.tune_rs_dc <= 128) dc = p.tune_rs_dc / 4 * 4;
    const int avx = (!p.sq_codes && !p.pq_codes && avx2_metric(p.metric)) ? avx_lanes(p.arith) : 0;
    if (avx) dc = std::max<uint32_t>(16u, dc / 16 * 16);  // whole groups of lanes per block
    const size_t lds = ((size_t)dc + (size_t)rcap * (dc + 4)) * sizeof(float);
    if (avx == 8 && !set_only_band)
        hipLaunchKernelGGL((rescore_finalize_kernel<false, 8>), dim3(p.q), dim3(RS_THREADS), lds, stream, p, dc, rcap, TauBand{});
    else if (avx == 16 && !set_only_band)
        hipLaunchKernelGGL((rescore_finalize_kernel<false, 16>), dim3(p.q), dim3(RS_THREADS), lds, stream, p, dc, rcap, TauBand{});
    else if (avx == 4 && !set_only_band)
        hipLaunchKernelGGL((rescore_finalize_kernel<false, 4>), dim3(p.q), dim3(RS_THREADS), lds, stream, p, dc, rcap, TauBand{});
    else if (set_only_band)
        hipLaunchKernelGGL(rescore_finalize_kernel<true>, dim3(p.q), dim3(RS_THREADS), lds, stream, p, dc, rcap, *set_only_band);
    else
        hipLaunchKernelGGL(rescore_finalize_kernel<false>, dim3(p.q), dim3(RS_THREADS), lds, stream, p, dc, rcap, TauBand{});
    return hipGetLastError();
}

// emit_sorted: the merged candidates already carry reference distances (score = -distance).
__global__ __launch_bounds__(256) void emit_sorted_kernel(PostParams p) {
    __shared__ uint64_t skey[KPRIME_MAX];
    const uint32_t q = blockIdx.x, j = threadIdx.x;
    const uint32_t nc = min(p.cand_n[q], p.kprime);
    const uint32_t k_eff = min(p.k, nc);
    uint64_t key = ~0ull;
    float d = 0.0f;
    uint32_t id = 0;
    if (j < nc) {
        id = p.cand_ids[(uint64_t)q * p.kprime + j];
        d = canon_distance(-p.cand_scores[(uint64_t)q * p.kprime + j]);
        key = hit_key(d, id);
    }
    if (j < KPRIME_MAX) skey[j] = key;
    __syncthreads();
    if (j < nc) {
        uint32_t rank = 0;
        for (uint32_t i = 0; i < nc; ++i) rank += (skey[i] < key) ? 1u : 0u;
        if (rank < k_eff) {
            p.out_ids[(uint64_t)q * p.k + rank] = id;
            p.out_dists[(uint64_t)q * p.k + rank] = d;
        }
    }
    for (uint32_t r = k_eff + j; r < p.k; r += blockDim.x) {
        p.out_ids[(uint64_t)q * p.k + r] = ~0ull;
        p.out_dists[(uint64_t)q * p.k + r] = INFINITY;
    }
    if (j == 0) p.out_counts[q] = k_eff;
}

hipError_t launch_emit_sorted(const PostParams& p, hipStream_t stream) {
    if (p.q == 0) return hipSuccess;
    const uint32_t threads = ((p.kprime + 63) / 64) * 64;
    hipLaunchKernelGGL(emit_sorted_kernel, dim3(p.q), dim3(threads), 0, stream, p);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Exact scan.  keys[f][i] = (f2ord(distance) << 32 | row) for live rows, all
// ones otherwise; the k-th smallest key is found by an 8-bit-per-pass radix
// descent over histograms, the keys <= it are gathered and ranked.
// ---------------------------------------------------------------------------
// One workgroup = 256 rows x up to EX_SLOTS slots (queries).  The rows are staged through LDS a block of
// dimensions at a time with coalesced reads (thread t then walks row t of the staged block), so the base is
// read once per EX_SLOTS slots at streaming speed; the arithmetic per (row, slot) is the scalar sequence of
// nh_exact.h continued across the blocks.
constexpr int EX_ROWS = 256, EX_DC = 32, EX_LD = EX_DC + 1, EX_SLOTS = 4;

template <int KIND>
__device__ inline void exact_accumulate_slots(RefAcc (&acc)[EX_SLOTS], const float (*qs)[EX_DC], const float* bs, uint32_t i0,
                                              uint32_t i1, uint32_t dim, uint32_t slots) {
#pragma unroll
    for (int s = 0; s < EX_SLOTS; ++s)
        if ((uint32_t)s < slots) ref_accumulate<KIND>(acc[s], qs[s], bs, i0, i1, dim);
}

__global__ __launch_bounds__(EX_ROWS) void exact_keys_kernel(ExactParams p) {
    __shared__ float rows[EX_ROWS * EX_LD];
    __shared__ float qs[EX_SLOTS][EX_DC];
    const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const uint32_t f0 = blockIdx.y * EX_SLOTS;
    const uint32_t slots = min((uint32_t)EX_SLOTS, p.f - f0);
    const uint32_t row0 = blockIdx.x * EX_ROWS;
    const uint32_t i = row0 + t;
    uint32_t qidx[EX_SLOTS];
    bool live[EX_SLOTS];
    uint32_t id = i;
    bool row_ok = i < p.n;
    if (row_ok && p.pos2id) { id = p.pos2id[i]; row_ok = id != 0xffffffffu; }
    if (row_ok && p.valid) row_ok = bit_set(p.valid, id);
    if (row_ok && p.filter) row_ok = bit_set(p.filter, id);
#pragma unroll
    for (int s = 0; s < EX_SLOTS; ++s) {
        const uint32_t f = f0 + min((uint32_t)s, slots - 1);
        qidx[s] = p.qsel ? p.qsel[f] : f;
        live[s] = row_ok && (uint32_t)s < slots;
        if (live[s] && p.pos_filter) live[s] = bit_set(p.pos_filter + (uint64_t)f * p.pos_filter_words, i);
    }
    const bool sq8 = p.sq_codes != nullptr, pq = p.pq_codes != nullptr;
    const int kind = (pq && !p.pq_asym) ? (int)RK_PQ_TABLE : ((sq8 || pq) ? (int)RK_RS_L2 : ref_kind(p.metric, p.arith));
    RefAcc acc[EX_SLOTS];
    // Rust's f32 Sum folds from -0.0; the accelerator's sums, the SQ8 and PQ asymmetric sums, the Hamming count and the
    // Chebyshev fold start from +0.0
    const float zero = (sq8 || pq || p.arith == NEEDLE_ARITH_ACCEL_CHUNK4 || kind == RK_HAMMING || kind == RK_CHEB) ? 0.0f : -0.0f;
#pragma unroll
    for (int s = 0; s < EX_SLOTS; ++s) { acc[s].a = zero; acc[s].b = kind == RK_PQ_TABLE ? -0.0f : zero; acc[s].c = zero; }

    const uint32_t rows_here = min((uint32_t)EX_ROWS, p.n > row0 ? p.n - row0 : 0u);
    // the SIMD-build orders (8 or 16 lane sums per accumulator) are not staged: every thread walks its own row
    const bool avx = avx_lanes(p.arith) && !sq8 && !pq && avx2_metric(p.metric);  // uniform over the grid
    float avx_d[EX_SLOTS];
    if (avx) {
#pragma unroll
        for (int s = 0; s < EX_SLOTS; ++s)
            avx_d[s] = live[s] ? exact_simd_build(p.arith, p.metric, p.query_raw + (uint64_t)qidx[s] * p.dim, p.base_raw + (uint64_t)i * p.dim, p.dim)
                               : 0.0f;
    }
    for (uint32_t c0 = 0; !avx && c0 < p.dim; c0 += EX_DC) {
        const uint32_t len = min((uint32_t)EX_DC, p.dim - c0);
        __syncthreads();
        if (t < EX_SLOTS * EX_DC) {
            const uint32_t s = t / EX_DC, j = t % EX_DC;
            qs[s][j] = j < len ? p.query_raw[(uint64_t)qidx[s] * p.dim + c0 + j] : 0.0f;
        }
        // wave w stages rows w*64 .. w*64+63, two rows per step (32 lanes each)
        for (uint32_t r = wave * 64 + (lane >> 5); r < wave * 64 + 64; r += 2) {
            const uint32_t j = lane & 31;
            float v = 0.0f;
            if (r < rows_here && j < len) {
                if (sq8) {  // decoded exactly as ScalarQuantizer::dequantize / asymmetric_distance do
                    const float cv = (float)p.sq_codes[(uint64_t)(row0 + r) * p.sq_stride + c0 + j];
                    v = (cv / p.sq_scale[c0 + j]) + p.sq_min[c0 + j];
                } else if (pq) {  // the centroid value asymmetric_distance subtracts at this dimension
                    const uint32_t dd = c0 + j, sub = dd / p.pq_dsub;
                    v = p.pq_books[((uint64_t)sub * 256 + p.pq_codes[(uint64_t)(row0 + r) * p.pq_m + sub]) * p.pq_dsub + dd % p.pq_dsub];
                } else {
                    v = p.base_raw[(uint64_t)(row0 + r) * p.dim + c0 + j];
                }
            }
            rows[r * EX_LD + j] = v;
        }
        __syncthreads();
        const float* bs = rows + t * EX_LD;
        switch (kind) {
        case RK_RS_COS: exact_accumulate_slots<RK_RS_COS>(acc, qs, bs, c0, c0 + len, p.dim, slots); break;
        case RK_RS_DOT: exact_accumulate_slots<RK_RS_DOT>(acc, qs, bs, c0, c0 + len, p.dim, slots); break;
        case RK_RS_L2: exact_accumulate_slots<RK_RS_L2>(acc, qs, bs, c0, c0 + len, p.dim, slots); break;
        case RK_C4_COS: exact_accumulate_slots<RK_C4_COS>(acc, qs, bs, c0, c0 + len, p.dim, slots); break;
        case RK_C4_DOT: exact_accumulate_slots<RK_C4_DOT>(acc, qs, bs, c0, c0 + len, p.dim, slots); break;
        case RK_C4_L2: exact_accumulate_slots<RK_C4_L2>(acc, qs, bs, c0, c0 + len, p.dim, slots); break;
        case RK_L1: exact_accumulate_slots<RK_L1>(acc, qs, bs, c0, c0 + len, p.dim, slots); break;
        case RK_HAMMING: exact_accumulate_slots<RK_HAMMING>(acc, qs, bs, c0, c0 + len, p.dim, slots); break;
        case RK_PQ_TABLE: exact_accumulate_slots<RK_PQ_TABLE>(acc, qs, bs, c0, c0 + len, p.pq_dsub, slots); break;
        default: exact_accumulate_slots<RK_CHEB>(acc, qs, bs, c0, c0 + len, p.dim, slots); break;
        }
    }
#pragma unroll
    for (int s = 0; s < EX_SLOTS; ++s) {
        if ((uint32_t)s >= slots) continue;  // block-uniform
        const uint32_t f = f0 + s;
        uint64_t key = ~0ull;
        if (live[s]) {
            float v = ref_finish(kind, acc[s]);
            if (kind == RK_RS_DOT || kind == RK_C4_DOT) v = p.metric == NEEDLE_METRIC_COSINE_NORMALIZED ? 1.0f - v : -v;
            if (avx) v = avx_d[s];
            key = hit_key(canon_distance(v), id);
        }
        if (i < p.n) p.keys[(uint64_t)f * p.n + i] = key;
        const uint64_t m = __builtin_amdgcn_ballot_w64(live[s]);
        if (lane == 0 && m) atomicAdd(&p.live[f], (uint32_t)__builtin_popcountll(m));
    }
}

__global__ void exact_init_kernel(ExactParams p) {
    const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= p.f) return;
    const uint32_t ke = min(p.k, p.live[f]);
    p.k_eff[f] = ke;
    p.kth_rem[f] = ke;
    p.prefix[f] = 0;
    p.sel_n[f] = 0;
}

__global__ __launch_bounds__(256) void exact_hist_kernel(ExactParams p, int shift) {
    __shared__ uint32_t h[256];
    const uint32_t f = blockIdx.y;
    h[threadIdx.x] = 0;
    __syncthreads();
    if (p.k_eff[f] != 0) {
        const uint64_t prefix = p.prefix[f];
        const uint64_t* keys = p.keys + (uint64_t)f * p.n;
        for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < p.n; i += gridDim.x * 256) {
            const uint64_t key = keys[i];
            const bool match = (shift == 56) || ((key >> (shift + 8)) == (prefix >> (shift + 8)));
            if (match) atomicAdd(&h[(uint32_t)(key >> shift) & 255u], 1u);
        }
    }
    __syncthreads();
    const uint32_t v = h[threadIdx.x];
    if (v) atomicAdd(&p.hist[(uint64_t)f * 256 + threadIdx.x], v);
}

__global__ void exact_pick_kernel(ExactParams p, int shift) {
    const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= p.f) return;
    uint32_t* h = p.hist + (uint64_t)f * 256;
    if (p.k_eff[f] != 0) {
        uint32_t rem = p.kth_rem[f], cum = 0;
        for (uint32_t b = 0; b < 256; ++b) {
            const uint32_t c = h[b];
            if (cum + c >= rem) {
                p.prefix[f] |= (uint64_t)b << shift;
                p.kth_rem[f] = rem - cum;
                break;
            }
            cum += c;
        }
    }
    for (uint32_t b = 0; b < 256; ++b) h[b] = 0;
}

__global__ __launch_bounds__(256) void exact_gather_kernel(ExactParams p) {
    const uint32_t f = blockIdx.y;
    if (p.k_eff[f] == 0) return;
    const uint64_t kth = p.prefix[f];
    const uint64_t* keys = p.keys + (uint64_t)f * p.n;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < p.n; i += gridDim.x * 256) {
        const uint64_t key = keys[i];
        if (key <= kth && key != ~0ull) {
            const uint32_t slot = atomicAdd(&p.sel_n[f], 1u);
            if (slot < p.k) p.sel[(uint64_t)f * p.k + slot] = key;
        }
    }
}

__global__ __launch_bounds__(256) void exact_emit_kernel(ExactParams p) {
    __shared__ uint64_t s[K_MAX];
    const uint32_t f = blockIdx.x;
    const uint32_t q = p.qsel ? p.qsel[f] : f;
    const uint32_t ke = p.k_eff[f];
    for (uint32_t i = threadIdx.x; i < ke; i += 256) s[i] = p.sel[(uint64_t)f * p.k + i];
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < ke; i += 256) {
        const uint64_t key = s[i];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < ke; ++j) rank += (s[j] < key) ? 1u : 0u;
        p.out_ids[(uint64_t)q * p.k + rank] = key & 0xffffffffull;
        p.out_dists[(uint64_t)q * p.k + rank] = ord2f((uint32_t)(key >> 32));
    }
    for (uint32_t r = ke + threadIdx.x; r < p.k; r += 256) {
        p.out_ids[(uint64_t)q * p.k + r] = ~0ull;
        p.out_dists[(uint64_t)q * p.k + r] = INFINITY;
    }
    if (threadIdx.x == 0) p.out_counts[q] = ke;
}

// k beyond the in-LDS ranking of exact_emit_kernel: the gathered keys of every slot are sorted in
// global memory (segmented radix sort, one segment per slot) and emitted in order.
__global__ void exact_offsets_kernel(uint32_t* __restrict__ off, uint32_t f, uint32_t k) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= f) off[i] = i * k;
}

__global__ __launch_bounds__(256) void exact_emit_presorted_kernel(ExactParams p, const uint64_t* __restrict__ sorted) {
    const uint32_t f = blockIdx.y;
    const uint32_t q = p.qsel ? p.qsel[f] : f;
    const uint32_t ke = p.k_eff[f];
    for (uint32_t r = blockIdx.x * 256 + threadIdx.x; r < p.k; r += gridDim.x * 256) {
        const uint64_t key = sorted[(uint64_t)f * p.k + r];
        p.out_ids[(uint64_t)q * p.k + r] = r < ke ? (key & 0xffffffffull) : ~0ull;
        p.out_dists[(uint64_t)q * p.k + r] = r < ke ? ord2f((uint32_t)(key >> 32)) : INFINITY;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) p.out_counts[q] = ke;
}

static hipError_t exact_emit_large_k(const ExactParams& p, hipStream_t stream) {
    uint64_t* sorted = nullptr;
    uint32_t* off = nullptr;
    void* tmp = nullptr;
    size_t tmp_bytes = 0;
    const size_t total = (size_t)p.f * p.k;
    hipError_t e = hipMalloc((void**)&sorted, total * sizeof(uint64_t));
    if (e == hipSuccess) e = hipMalloc((void**)&off, ((size_t)p.f + 1) * sizeof(uint32_t));
    if (e == hipSuccess) {
        hipLaunchKernelGGL(exact_offsets_kernel, dim3((p.f + 256) / 256), dim3(256), 0, stream, off, p.f, p.k);
        e = rocprim::segmented_radix_sort_keys(nullptr, tmp_bytes, p.sel, sorted, (unsigned int)total, p.f, off, off + 1, 0, 64,
                                               stream);
    }
    if (e == hipSuccess) e = hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 16);
    if (e == hipSuccess)
        e = rocprim::segmented_radix_sort_keys(tmp, tmp_bytes, p.sel, sorted, (unsigned int)total, p.f, off, off + 1, 0, 64,
                                               stream);
    if (e == hipSuccess) {
        const uint32_t kb = (p.k + 255) / 256 < 1024 ? (p.k + 255) / 256 : 1024;
        hipLaunchKernelGGL(exact_emit_presorted_kernel, dim3(kb, p.f), dim3(256), 0, stream, p, sorted);
        e = hipGetLastError();
    }
    const hipError_t es = hipStreamSynchronize(stream);  // the temporaries are freed below
    if (e == hipSuccess) e = es;
    if (tmp) (void)hipFree(tmp);
    if (off) (void)hipFree(off);
    if (sorted) (void)hipFree(sorted);
    return e;
}

hipError_t launch_exact_scan(const ExactParams& p, hipStream_t stream) {
    if (p.f == 0) return hipSuccess;
    hipError_t e;
    if (p.k > (uint32_t)K_MAX) {  // unused selection slots must sort behind every real key
        if ((e = hipMemsetAsync(p.sel, 0xff, sizeof(uint64_t) * (size_t)p.f * p.k, stream)) != hipSuccess) return e;
    }
    if ((e = hipMemsetAsync(p.live, 0, sizeof(uint32_t) * p.f, stream)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(p.hist, 0, sizeof(uint32_t) * 256 * p.f, stream)) != hipSuccess) return e;
    const uint32_t nb = (p.n + 255) / 256;
    const uint32_t fb = (p.f + 63) / 64;
    if (nb) hipLaunchKernelGGL(exact_keys_kernel, dim3(nb, (p.f + EX_SLOTS - 1) / EX_SLOTS), dim3(EX_ROWS), 0, stream, p);
    hipLaunchKernelGGL(exact_init_kernel, dim3(fb), dim3(64), 0, stream, p);
    const uint32_t hb = nb < 1024 ? (nb ? nb : 1) : 1024;
    for (int shift = 56; shift >= 0; shift -= 8) {
        hipLaunchKernelGGL(exact_hist_kernel, dim3(hb, p.f), dim3(256), 0, stream, p, shift);
        hipLaunchKernelGGL(exact_pick_kernel, dim3(fb), dim3(64), 0, stream, p, shift);
    }
    hipLaunchKernelGGL(exact_gather_kernel, dim3(hb, p.f), dim3(256), 0, stream, p);
    if (p.k > (uint32_t)K_MAX) return exact_emit_large_k(p, stream);
    hipLaunchKernelGGL(exact_emit_kernel, dim3(p.f), dim3(256), 0, stream, p);
    return hipGetLastError();
}

// Position bitmask of the rows in the lists a query probes (one slot per query
// being re-run on the exact scan).  Lists start on 128-row boundaries, so no two
// lists share a mask word.
__global__ __launch_bounds__(256) void ivf_mark_lists_kernel(const uint32_t* __restrict__ qsel,
                                                             const uint32_t* __restrict__ probe,
                                                             const uint32_t* __restrict__ probe_n, uint32_t nprobe,
                                                             const uint32_t* __restrict__ list_pos,
                                                             const uint32_t* __restrict__ list_len,
                                                             uint32_t* __restrict__ mask, uint32_t words) {
    const uint32_t f = blockIdx.y, s = blockIdx.x;
    const uint32_t q = qsel ? qsel[f] : f;
    if (s >= probe_n[q]) return;
    const uint32_t l = probe[(uint64_t)q * nprobe + s];
    const uint32_t p0 = list_pos[l], len = list_len[l];
    uint32_t* m = mask + (uint64_t)f * words + (p0 >> 5);
    for (uint32_t w = threadIdx.x; w * 32 < len; w += 256) {
        const uint32_t rem = len - w * 32;
        m[w] = rem >= 32 ? 0xffffffffu : ((1u << rem) - 1u);
    }
}

hipError_t launch_ivf_mark_lists(const uint32_t* d_qsel, uint32_t f, const uint32_t* d_probe, const uint32_t* d_probe_n,
                                 uint32_t nprobe, const uint32_t* d_list_pos, const uint32_t* d_list_len,
                                 uint32_t* d_mask, uint32_t words, hipStream_t stream) {
    if (f == 0 || nprobe == 0) return hipSuccess;
    hipError_t e = hipMemsetAsync(d_mask, 0, (size_t)f * words * sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(ivf_mark_lists_kernel, dim3(nprobe, f), dim3(256), 0, stream, d_qsel, d_probe, d_probe_n, nprobe,
                       d_list_pos, d_list_len, d_mask, words);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// batch_{cosine,euclidean}_distance / batch_dot_product: raw distances of one
// query against n rows.  The dot product is reported positive.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void batch_distance_kernel(const float* __restrict__ query,
                                                             const float* __restrict__ base, uint64_t n,
                                                             uint32_t dim, int metric, int arith,
                                                             float* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float d = exact_distance(metric, arith, query, base + i * dim, dim);
    if (metric == NEEDLE_METRIC_DOT_PRODUCT) d = -d;
    out[i] = d;
}

hipError_t launch_batch_distance(const float* d_query, const float* d_base, uint64_t n, uint32_t dim, int metric,
                                 int arith, float* d_out, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(batch_distance_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, stream, d_query,
                       d_base, n, dim, metric, arith, d_out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// distance_matrix: every query against every row, each value the metric's distance in the chosen arithmetic
// (dot product negated), row-major [queries][rows].  A workgroup owns 64 rows x 16 queries; the rows and the queries
// pass through LDS 32 dimensions at a time (coalesced loads), thread (row, query group) then continues its four
// sums over the staged block in dimension order, so every value is the scalar sequence of the reference.
// ---------------------------------------------------------------------------
constexpr uint32_t DM_ROWS = 64, DM_Q = 16, DM_DC = 32;

__global__ __launch_bounds__(256) void distance_matrix_kernel(const float* __restrict__ queries, uint32_t nq,
                                                              const float* __restrict__ base, uint64_t n, uint32_t dim,
                                                              int metric, int arith, float* __restrict__ out) {
    __shared__ float sb[DM_ROWS][DM_DC + 1];  // odd stride: lane = row walks its row without bank conflicts
    __shared__ float sq[DM_Q][DM_DC];         // a wave reads one query element at a time: broadcast
    const uint32_t t = threadIdx.x, r = t & 63, g = t >> 6;
    const uint64_t row0 = (uint64_t)blockIdx.x * DM_ROWS;
    const uint32_t q0 = blockIdx.y * DM_Q;
    if (avx_lanes(arith) && avx2_metric(metric)) {
        // the SIMD-build orders (8 or 16 lane sums per accumulator) are not staged: every thread walks its own row and queries
        if (row0 + r < n)
            for (int j = 0; j < 4; ++j) {
                const uint32_t qi = q0 + g * 4 + j;
                if (qi < nq)
                    out[(uint64_t)qi * n + row0 + r] = exact_simd_build(arith, metric, queries + (uint64_t)qi * dim, base + (row0 + r) * dim, dim);
            }
        return;  // uniform over the workgroup
    }
    const int kind = ref_kind(metric, arith);
    // Rust's f32 Sum folds from -0.0; the chunked sums, the service's cosine and Chebyshev's fold(0.0, max) start at +0.0
    const float zero = (kind == RK_SV_COS || kind == RK_CHEB || arith == NEEDLE_ARITH_ACCEL_CHUNK4) ? 0.0f : -0.0f;
    RefAcc acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { acc[j].a = zero; acc[j].b = zero; acc[j].c = zero; }
    for (uint32_t c0 = 0; c0 < dim; c0 += DM_DC) {
        const uint32_t len = min(DM_DC, dim - c0);
        __syncthreads();
        for (uint32_t i = t; i < DM_ROWS * DM_DC; i += 256) {
            const uint32_t rr = i / DM_DC, cc = i % DM_DC;
            sb[rr][cc] = (row0 + rr < n && cc < len) ? base[(row0 + rr) * dim + c0 + cc] : 0.0f;
        }
        for (uint32_t i = t; i < DM_Q * DM_DC; i += 256) {
            const uint32_t qq = i / DM_DC, cc = i % DM_DC;
            sq[qq][cc] = (q0 + qq < nq && cc < len) ? queries[(uint64_t)(q0 + qq) * dim + c0 + cc] : 0.0f;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float* qs = sq[g * 4 + j];
            switch (kind) {
            case RK_RS_COS: ref_accumulate<RK_RS_COS>(acc[j], qs, sb[r], c0, c0 + len, dim); break;
            case RK_SV_COS: ref_accumulate<RK_SV_COS>(acc[j], qs, sb[r], c0, c0 + len, dim); break;
            case RK_RS_DOT: ref_accumulate<RK_RS_DOT>(acc[j], qs, sb[r], c0, c0 + len, dim); break;
            case RK_RS_L2: ref_accumulate<RK_RS_L2>(acc[j], qs, sb[r], c0, c0 + len, dim); break;
            case RK_C4_COS: ref_accumulate<RK_C4_COS>(acc[j], qs, sb[r], c0, c0 + len, dim); break;
            case RK_C4_DOT: ref_accumulate<RK_C4_DOT>(acc[j], qs, sb[r], c0, c0 + len, dim); break;
            case RK_C4_L2: ref_accumulate<RK_C4_L2>(acc[j], qs, sb[r], c0, c0 + len, dim); break;
            case RK_L1: ref_accumulate<RK_L1>(acc[j], qs, sb[r], c0, c0 + len, dim); break;
            case RK_HAMMING: ref_accumulate<RK_HAMMING>(acc[j], qs, sb[r], c0, c0 + len, dim); break;
            default: ref_accumulate<RK_CHEB>(acc[j], qs, sb[r], c0, c0 + len, dim); break;
            }
        }
    }
    if (row0 + r >= n) return;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t qi = q0 + g * 4 + j;
        if (qi >= nq) continue;
        float v = ref_finish(kind, acc[j]);
        if (kind == RK_RS_DOT || kind == RK_C4_DOT) v = metric == NEEDLE_METRIC_COSINE_NORMALIZED ? 1.0f - v : -v;
        out[(uint64_t)qi * n + row0 + r] = v;
    }
}

hipError_t launch_distance_matrix(const float* d_queries, uint32_t q, const float* d_base, uint64_t n, uint32_t dim, int metric,
                                  int arith, float* d_out, hipStream_t stream) {
    if (q == 0 || n == 0) return hipSuccess;
    const uint64_t gx = (n + DM_ROWS - 1) / DM_ROWS, gy = (q + DM_Q - 1) / DM_Q;
    if (gx > 0x7fffffffull || gy > 65535ull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(distance_matrix_kernel, dim3((uint32_t)gx, (uint32_t)gy), dim3(256), 0, stream, d_queries, q, d_base, n, dim,
                       metric, arith, d_out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// merge_topk: per query, `lists` sorted runs of (distance, id) -> the k best by
// (distance, id), repeated ids dropped (the better copy stays).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void merge_topk_kernel(const float* __restrict__ dists,
                                                         const uint64_t* __restrict__ ids,
                                                         const uint32_t* __restrict__ counts, uint32_t lists,
                                                         uint32_t nq, uint32_t per, uint32_t k,
                                                         uint64_t* __restrict__ out_ids,
                                                         float* __restrict__ out_dists,
                                                         uint32_t* __restrict__ out_counts) {
    extern __shared__ __attribute__((aligned(16))) unsigned char mraw[];
    const uint32_t cap = lists * per;
    uint64_t* sid = (uint64_t*)mraw;
    uint32_t* sord = (uint32_t*)(sid + cap);
    uint32_t* sdup = sord + cap;
    __shared__ uint32_t s_total, s_kept;
    const uint32_t q = blockIdx.x;
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (uint32_t l = 0; l < lists; ++l) {
            const uint32_t c = min(counts[(uint64_t)l * nq + q], per);
            for (uint32_t j = 0; j < c; ++j) {
                sid[t] = ids[((uint64_t)l * nq + q) * per + j];
                sord[t] = f2ord(canon_distance(dists[((uint64_t)l * nq + q) * per + j]));
                ++t;
            }
        }
        s_total = t;
        s_kept = 0;
    }
    __syncthreads();
    const uint32_t total = s_total;
    // entry i precedes entry j: (distance, id, position) lexicographic
    auto before = [&](uint32_t i, uint32_t j) {
        if (sord[i] != sord[j]) return sord[i] < sord[j];
        if (sid[i] != sid[j]) return sid[i] < sid[j];
        return i < j;
    };
    for (uint32_t j = threadIdx.x; j < total; j += 256) {
        uint32_t dup = 0;
        for (uint32_t i = 0; i < total; ++i) dup |= (sid[i] == sid[j] && i != j && before(i, j)) ? 1u : 0u;
        sdup[j] = dup;
    }
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < total; j += 256) {
        if (sdup[j]) continue;
        uint32_t rank = 0;
        for (uint32_t i = 0; i < total; ++i) rank += (!sdup[i] && i != j && before(i, j)) ? 1u : 0u;
        atomicAdd(&s_kept, 1u);
        if (rank < k) {
            out_ids[(uint64_t)q * k + rank] = sid[j];
            out_dists[(uint64_t)q * k + rank] = ord2f(sord[j]);
        }
    }
    __syncthreads();
    const uint32_t kept = min(s_kept, k);
    for (uint32_t r = kept + threadIdx.x; r < k; r += 256) {
        out_ids[(uint64_t)q * k + r] = ~0ull;
        out_dists[(uint64_t)q * k + r] = INFINITY;
    }
    if (threadIdx.x == 0) out_counts[q] = kept;
}

hipError_t launch_merge_topk(const float* d_dists, const uint64_t* d_ids, const uint32_t* d_counts, uint32_t lists,
                             uint32_t q, uint32_t per, uint32_t k, uint64_t* d_out_ids, float* d_out_dists,
                             uint32_t* d_out_counts, hipStream_t stream) {
    if (q == 0) return hipSuccess;
    const size_t lds = (size_t)lists * per * (sizeof(uint64_t) + 2 * sizeof(uint32_t));
    hipLaunchKernelGGL(merge_topk_kernel, dim3(q), dim3(256), lds, stream, d_dists, d_ids, d_counts, lists, q, per,
                       k, d_out_ids, d_out_dists, d_out_counts);
    return hipGetLastError();
}

__global__ void fill_empty_kernel(uint64_t* out_ids, float* out_dists, uint32_t* out_counts, uint32_t q,
                                  uint32_t k) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (uint64_t)q * k) { out_ids[i] = ~0ull; out_dists[i] = INFINITY; }
    if (i < q) out_counts[i] = 0;
}

hipError_t launch_fill_empty(uint64_t* d_out_ids, float* d_out_dists, uint32_t* d_out_counts, uint32_t q,
                             uint32_t k, hipStream_t stream) {
    const uint64_t total = (uint64_t)q * (k ? k : 1);
    if (q == 0) return hipSuccess;
    hipLaunchKernelGGL(fill_empty_kernel, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, stream, d_out_ids,
                       d_out_dists, d_out_counts, q, k);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ src, const uint32_t* __restrict__ idx,
                                                          uint32_t dim, float* __restrict__ dst) {
    const float* s = src + (uint64_t)idx[blockIdx.x] * dim;
    float* d = dst + (uint64_t)blockIdx.x * dim;
    for (uint32_t i = threadIdx.x; i < dim; i += blockDim.x) d[i] = s[i];
}

hipError_t launch_gather_rows(const float* d_src, const uint32_t* d_idx, uint32_t n, uint32_t dim, float* d_dst,
                              hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(n), dim3(256), 0, stream, d_src, d_idx, dim, d_dst);
    return hipGetLastError();
}

__global__ __launch_bounds__(64) void scatter_results_kernel(const uint32_t* __restrict__ idx, uint32_t k,
                                                             const uint64_t* __restrict__ ids, const float* __restrict__ dists,
                                                             const uint32_t* __restrict__ counts, uint64_t* __restrict__ out_ids,
                                                             float* __restrict__ out_dists, uint32_t* __restrict__ out_counts) {
    const uint32_t s = blockIdx.x, q = idx[s];
    for (uint32_t i = threadIdx.x; i < k; i += blockDim.x) {
        out_ids[(uint64_t)q * k + i] = ids[(uint64_t)s * k + i];
        out_dists[(uint64_t)q * k + i] = dists[(uint64_t)s * k + i];
    }
    if (threadIdx.x == 0) out_counts[q] = counts[s];
}

hipError_t launch_scatter_results(const uint32_t* d_idx, uint32_t n, uint32_t k, const uint64_t* d_ids, const float* d_dists,
                                  const uint32_t* d_counts, uint64_t* d_out_ids, float* d_out_dists, uint32_t* d_out_counts,
                                  hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(scatter_results_kernel, dim3(n), dim3(64), 0, stream, d_idx, k, d_ids, d_dists, d_counts, d_out_ids,
                       d_out_dists, d_out_counts);
    return hipGetLastError();
}

}  // namespace nh
