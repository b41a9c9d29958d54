// Wave-level (64-lane) exact selection over small candidate lists.
//
// A list entry is (score bits, row id); larger score = better candidate, ties
// go to the lower row id.  Entries are compared through a 64-bit key
// (f2ord(score) << 32 | ~id), all keys of one list are distinct, and the
// kprime-th largest key is found by building it bit by bit from the top: a bit
// stays set iff at least kprime keys are >= the trial value.  Counting is a
// v_cmp + s_bcnt1 per 64 entries, no shuffles and no LDS.
#pragma once
#include "nh_common.h"

namespace nh {

__device__ inline uint64_t cand_key(uint32_t score_bits, uint32_t id) {
    return ((uint64_t)f2ord(__uint_as_float(score_bits)) << 32) | (uint64_t)(~id);
}

// Reduce the list at `list` (n_e <= 64 * LIST_E_MAX entries, written earlier by
// this same wave) to its kprime best entries, stored at positions
// [0, kprime).  Requires n_e > kprime.  Returns the score of the kprime-th best
// entry: every discarded entry has a score <= it.  All 64 lanes must call.
__device__ inline float wave_compact_list(uint2* __restrict__ list, uint32_t n_e, uint32_t kprime,
                                          uint32_t lane) {
    // the pushes being read back were stored by other lanes of this wave
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    uint64_t keys[LIST_E_MAX];
    const uint32_t e_cnt = (n_e + 63) >> 6;
#pragma unroll
    for (int e = 0; e < LIST_E_MAX; ++e) {
        keys[e] = 0;
        if ((uint32_t)e < e_cnt) {
            uint32_t idx = e * 64 + lane;
            if (idx < n_e) {
                uint2 v = list[idx];
                keys[e] = cand_key(v.x, v.y);
            }
        }
    }
    uint64_t prefix = 0;
    for (int bit = 63; bit >= 0; --bit) {
        const uint64_t trial = prefix | (1ull << bit);
        uint32_t total = 0;
#pragma unroll
        for (int e = 0; e < LIST_E_MAX; ++e) {
            if ((uint32_t)e < e_cnt) total += __builtin_popcountll(__builtin_amdgcn_ballot_w64(keys[e] >= trial));
        }
        if (total >= kprime) prefix = trial;
    }
    // prefix == the kprime-th largest key; exactly kprime keys are >= it
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    uint32_t base = 0;
#pragma unroll
    for (int e = 0; e < LIST_E_MAX; ++e) {
        if ((uint32_t)e < e_cnt) {
            const bool keep = keys[e] >= prefix;
            const uint64_t m = __builtin_amdgcn_ballot_w64(keep);
            if (keep) {
                uint32_t pos = base + __builtin_popcountll(m & lt_mask);
                list[pos] = make_uint2(__float_as_uint(ord2f((uint32_t)(keys[e] >> 32))), ~(uint32_t)keys[e]);
            }
            base += __builtin_popcountll(m);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    return ord2f((uint32_t)(prefix >> 32));
}

// The k-th largest of up to 64 * PER 32-bit values held in registers (element u * 64 + lane in v[u]; missing elements
// are 0 and never counted; total >= k live values, all non-zero).  Every bit costs a compare and a ballot count per 64
// values: no LDS round trip, no shuffle.
template <int PER>
__device__ inline uint32_t wave_kth_largest_regs(const uint32_t (&v)[PER], uint32_t total, uint32_t k) {
    uint32_t prefix = 0;
    for (int bit = 31; bit >= 0; --bit) {
        const uint32_t trial = prefix | (1u << bit);
        uint32_t cnt = 0;
#pragma unroll
        for (int u = 0; u < PER; ++u)
            if ((uint32_t)u * 64u < total) cnt += __builtin_popcountll(__builtin_amdgcn_ballot_w64(v[u] >= trial));
        if (cnt >= k) prefix = trial;
    }
    return prefix;
}

// The same for distinct 64-bit keys (high word: score ordinal, never 0): the high word of the k-th largest key first,
// then -- only when several keys share that high word -- its low word among those keys.
template <int PER>
__device__ inline uint64_t wave_kth_largest_regs64(const uint64_t (&v)[PER], uint32_t total, uint32_t k) {
    uint32_t hp = 0;
    for (int bit = 31; bit >= 0; --bit) {
        const uint32_t trial = hp | (1u << bit);
        uint32_t cnt = 0;
#pragma unroll
        for (int u = 0; u < PER; ++u)
            if ((uint32_t)u * 64u < total) cnt += __builtin_popcountll(__builtin_amdgcn_ballot_w64((uint32_t)(v[u] >> 32) >= trial));
        if (cnt >= k) hp = trial;
    }
    uint32_t gt = 0, eq = 0;
#pragma unroll
    for (int u = 0; u < PER; ++u)
        if ((uint32_t)u * 64u < total) {
            gt += __builtin_popcountll(__builtin_amdgcn_ballot_w64((uint32_t)(v[u] >> 32) > hp));
            eq += __builtin_popcountll(__builtin_amdgcn_ballot_w64((uint32_t)(v[u] >> 32) == hp));
        }
    const uint32_t need = k - gt;  // 1 <= need <= eq
    uint32_t lp = 0;
    if (eq == 1) {
#pragma unroll
        for (int u = 0; u < PER; ++u)
            if ((uint32_t)u * 64u < total) {
                const uint64_t m = __builtin_amdgcn_ballot_w64((uint32_t)(v[u] >> 32) == hp);
                if (m) lp = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v[u], (int)__builtin_ctzll(m));
            }
    } else {
        for (int bit = 31; bit >= 0; --bit) {
            const uint32_t trial = lp | (1u << bit);
            uint32_t cnt = 0;
#pragma unroll
            for (int u = 0; u < PER; ++u)
                if ((uint32_t)u * 64u < total)
                    cnt += __builtin_popcountll(__builtin_amdgcn_ballot_w64((uint32_t)(v[u] >> 32) == hp && (uint32_t)v[u] >= trial));
            if (cnt >= need) lp = trial;
        }
    }
    return ((uint64_t)hp << 32) | lp;
}

}  // namespace nh
