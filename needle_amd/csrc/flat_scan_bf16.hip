// bf16 first pass of the exact flat search on the matrix cores, f32 accumulate.  Two kernels share one design:
// flat_scan_bf16x_kernel (v_mfma_f32_16x16x32_bf16; the one that ships, eight waves of 32 queries per workgroup up to
// 384 dimensions, four waves beyond) and flat_scan_bf16_kernel (v_mfma_f32_32x32x16_bf16, four waves, one per SIMD;
// kept as the BF_MF16 = 0 build for comparison).
//
// Structure:
//   * the queries are stationary IN REGISTERS: a wave holds the B fragments of its own queries for the full
//     dimension, loaded once;
//   * the rows stream through a ring of LDS slots, one slot = 64 rows x 128 dims (16 KB), filled
//     by direct-to-LDS loads three stages ahead of the MFMAs.  A wave-instruction of such a load
//     writes 1 KB linearly (4 rows x 256 B), so the bank-conflict swizzle (16-byte chunk c of row r
//     lives at chunk c ^ (r & 15)) is applied to the per-lane SOURCE address;
//   * every wave multiplies ALL 64 staged rows with ITS queries, so one 16-byte LDS fragment feeds one MFMA per
//     query tile of the wave and the rows are fetched from L2/HBM once per workgroup (256 or 128 queries).  A 128-row
//     tile is walked as two half tiles, which keeps the accumulators small beside the query fragments.
// One barrier per stage; the loads in flight are retired by counted vmcnt waits, never vmcnt(0).
//
// Selection: each lane compares its results per tile with its query's threshold; every LANE GROUP sharing a query's
// column (lane halves in the 32x32 form, lane quarters in the 16x16 form) owns its candidate list (query, slice,
// group), so a push is a predicated store and a counter increment with no cross-lane step.
#include "nh_common.h"
#include "nh_kernels.h"
#include "nh_scan.h"
#include "nh_select.h"

#include <type_traits>
#include <utility>

namespace nh {

// compile-time loop: f(std::integral_constant<int, 0>) ... f(std::integral_constant<int, N-1>)
template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

#ifndef BF_FILL_KK
#define BF_FILL_KK 4      // k-step of a stage at which the fills of the stage three ahead start
#endif
#ifndef BF_FILL_SPREAD
#define BF_FILL_SPREAD 1  // 1: one fill piece per k-step from BF_FILL_KK on; 0: all four at BF_FILL_KK
#endif
#ifndef BF_STAGGER
#define BF_STAGGER 0      // 1: eight-wave workgroups hold waves 4-7 half a stage behind waves 0-3 (measured 1-2 % slower, and 2-3 %
                          // slower when the lagging waves also fill half a stage later than the leading ones)
#endif
#ifndef BF_FLAT_SKIP_DEAD
#define BF_FLAT_SKIP_DEAD 0  // 1: waves of the flat scan whose query slots are all padding skip the matrix work too
#endif
#define BF_SKIPS_DEAD(MODE) ((MODE) == MODE_IVF || (BF_FLAT_SKIP_DEAD && (MODE) == MODE_FLAT))
#ifndef BF_READ_AHEAD
#define BF_READ_AHEAD 2   // fragment reads run this many k-steps ahead of the MFMAs (2 or 3)
#endif

constexpr int BF_ROWS = 64;                               // rows per stage: half a 128-row tile
constexpr int BF_BK = 128;                                // dims per stage
constexpr int BF_STAGE_BYTES = BF_ROWS * BF_BK * 2;       // 16 KB
#ifndef BF_RING_SLOTS
#define BF_RING_SLOTS 4
#endif
#ifndef BF_RING_AHEAD
#define BF_RING_AHEAD 3
#endif
constexpr int BF_NBUF = BF_RING_SLOTS;                    // ring slots
constexpr int BF_PD = BF_RING_AHEAD;                      // stages in flight ahead of the one being multiplied (< BF_NBUF)
constexpr int BF_CSLOT = 1024;                            // per slot: the half tile's accumulator seeds, once per wave
constexpr int BF_LDS_BYTES = BF_NBUF * (BF_STAGE_BYTES + BF_CSLOT);

// byte offset of 16-byte chunk c (0..15) of row r (0..63) inside a stage: rows are 256 B = one sweep of
// the LDS banks, so the 16 rows one ds_read_b128 phase touches are spread over the 16 chunk positions
__device__ inline uint32_t bf_lds_off(uint32_t r, uint32_t c) { return r * 256u + ((c ^ (r & 15u)) << 4); }

template <int MODE>
__device__ inline void push_lane(const f32x16& acc, uint32_t row_base, uint32_t h, uint32_t lane, const ScanParams& p,
                                 uint32_t list_idx, float& tau, uint32_t& cnt) {
    const float g0 = fmaxf(fmaxf(acc[0], acc[1]), fmaxf(acc[2], acc[3]));
    const float g1 = fmaxf(fmaxf(acc[4], acc[5]), fmaxf(acc[6], acc[7]));
    const float g2 = fmaxf(fmaxf(acc[8], acc[9]), fmaxf(acc[10], acc[11]));
    const float g3 = fmaxf(fmaxf(acc[12], acc[13]), fmaxf(acc[14], acc[15]));
    const float m = fmaxf(fmaxf(g0, g1), fmaxf(g2, g3));
    if (__builtin_amdgcn_ballot_w64(m > tau) == 0) return;  // wave-uniform
    uint2* mylist = p.lists + (uint64_t)list_idx * p.cap;
    const float gm[4] = {g0, g1, g2, g3};
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        if (gm[g] > tau) {  // rows 8g + 4h .. + 3 of the tile
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float key = acc[4 * g + j];
                if (key > tau) {
                    const uint32_t row = row_base + row_of_reg(4 * g + j, h);
                    if (row_live<MODE>(p, row)) {
                        mylist[cnt] = make_uint2(__float_as_uint(key), row);
                        cnt += 1;
                    }
                }
            }
        }
    }
    uint64_t need = __builtin_amdgcn_ballot_w64(cnt > p.cap - 16);
    while (need) {
        const uint32_t b = (uint32_t)__builtin_ctzll(need);
        need &= need - 1;
        const uint32_t n_e = (uint32_t)__builtin_amdgcn_readlane((int)cnt, (int)b);
        const uint32_t li = (uint32_t)__builtin_amdgcn_readlane((int)list_idx, (int)b);
        const float t = wave_compact_list(p.lists + (uint64_t)li * p.cap, n_e, p.kprime, lane);
        if (lane == b) { tau = t; cnt = p.kprime; }
    }
}

// direct-to-LDS loads a wave issues for stage u of a tile (u = half * STAGES + s, taken modulo the tile)
template <int STAGES, bool CINIT, int PIECES = 4>
constexpr int bf_stage_loads(int u) { return PIECES + ((CINIT && (u % STAGES) == 0) ? 1 : 0); }
// loads issued behind stage u that may still be in flight when stage u is awaited
// loads that may stay in flight while stage u + 1 is awaited at the fill point of stage u: those of stages
// u + 2 .. u + BF_PD - 1 (stage u + BF_PD is issued right behind that wait)
template <int STAGES, bool CINIT, int PIECES = 4>
constexpr int bf_steady_inflight(int u) {
    int n = 0;
    for (int j = 2; j < BF_PD; ++j) n += bf_stage_loads<STAGES, CINIT, PIECES>(u + j);
    return n;
}
template <int STAGES, bool CINIT, int PIECES = 4>
constexpr int bf_inflight_behind(int u) {
    int n = 0;
    for (int j = 1; j < BF_PD; ++j) n += bf_stage_loads<STAGES, CINIT, PIECES>(u + j);
    return n;
}

// NT query tiles per wave, KS k-steps of 16 dims (dpad = 16 * KS).  Queries per workgroup = 128 * NT.
template <int MODE, int NT, int KS, bool CINIT>
__global__ __launch_bounds__(256, 1) void flat_scan_bf16_kernel(ScanParams p) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[BF_LDS_BYTES];
    constexpr int STAGES = KS / 8;       // stages per half tile
    constexpr int UT = 2 * STAGES;       // stages per tile
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const uint32_t l31 = lane & 31, h = lane >> 5;
    constexpr uint32_t QB = 128 * NT;

    // FLAT / PRE: workgroup = (query block, slice); IVF: workgroup = one work item = a run of consecutive tiles of
    // one inverted list x up to 128*NT of the (query, candidate list) pairs probing it.
    uint32_t split = 0, q0 = 0, tile_first, tile_mul, cnt_tiles;
    uint32_t my_query[NT];   // row of query_op each of this lane's columns multiplies
    bool q_live[NT];
    float tau[NT];
    uint32_t cnt[NT], list_idx[NT];
    float gmax[NT][4];
    if (MODE == MODE_IVF) {
        const IvfItem it = p.items[blockIdx.x];
        tile_first = it.tile0;
        tile_mul = 1;
        cnt_tiles = it.tile1 - it.tile0;
        const uint2* iq = p.item_queries + it.qoff;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const uint32_t jq = wave * (32 * NT) + nt * 32 + l31;
            q_live[nt] = jq < it.nq;
            const uint2 e = q_live[nt] ? iq[jq] : make_uint2(0u, 0u);
            my_query[nt] = e.x;
            list_idx[nt] = (e.y + it.list_add) * 2 + h;  // every lane half owns a list
        }
    } else {
        uint32_t q_tile;
        if (p.xcd_map) {
            const uint32_t xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
            split = xcd + 8 * (j / p.q_tiles);
            q_tile = j % p.q_tiles;
        } else {
            split = blockIdx.x / p.q_tiles;
            q_tile = blockIdx.x % p.q_tiles;
        }
        q0 = q_tile * QB + wave * (32 * NT);  // first query of this wave
        // the block's run: tiles split + (jb + jj * jstep) * n_splits, jj = 0 .. cnt-1
        const uint32_t own = split < p.n_tiles ? (p.n_tiles - split + p.n_splits - 1) / p.n_splits : 0u;  // tiles of this slice
        uint32_t jb = 0, jstep = 1;
        if (MODE == MODE_PRE) {
            jstep = max(p.pre_stride, 1u);
            cnt_tiles = min(p.pre_tiles, (own + jstep - 1) / jstep);
        } else {
            jb = min(p.tile_off, own);
            cnt_tiles = min(p.tile_len, own - jb);
        }
        tile_mul = jstep * p.n_splits;
        tile_first = split + jb * p.n_splits;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            my_query[nt] = q0 + nt * 32 + l31;
            q_live[nt] = my_query[nt] < p.n_queries;
            list_idx[nt] = (my_query[nt] * p.n_splits + split) * 2 + h;
        }
    }
    const uint32_t t0 = 0, t1 = cnt_tiles;  // positions jj of the run
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        tau[nt] = -INFINITY;
        if (MODE != MODE_PRE && p.tau0 && q_live[nt]) tau[nt] = p.tau0[my_query[nt]];
        if ((p.debug & 1u) || !q_live[nt]) tau[nt] = INFINITY;
        cnt[nt] = (MODE == MODE_FLAT && p.resume) ? p.list_counts[list_idx[nt]] : 0u;
#pragma unroll
        for (int g = 0; g < 4; ++g) gmax[nt][g] = -INFINITY;
    }

#ifdef BF_STAMP
    // diagnostic build only: shader-clock and 100 MHz real-time stamps around the block's run
    const uint64_t st_c0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    if (t1 > t0) {  // block-uniform
        // stationary query fragments: lane (l31, h) holds dims 16*ks + 8*h .. +7 of query q0 + 32*nt + l31
        bf16x8 bq[NT][KS];
        const uint16_t* __restrict__ gB = (const uint16_t*)p.query_op;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const uint16_t* qrow = gB + (uint64_t)my_query[nt] * p.dpad + 8 * h;  // (a dead IVF slot reads query 0: never lists)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) bq[nt][ks] = __builtin_bit_cast(bf16x8, *(const uint4*)(qrow + ks * 16));
        }
        // keep the fragments in the accumulation half of the register file: the MFMA reads its B operand from
        // there directly, and the other half stays free for the row fragments (pinned after ALL loads were
        // issued, so that the loads overlap)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+a"(bq[nt][ks]));

        const unsigned char* __restrict__ gA = (const unsigned char*)p.base_op;
        const uint32_t row_bytes = p.dpad * 2;
        // piece i (0..3) of this wave = rows (wave*4+i)*4 .. +3 of the stage; lane -> (row, physical chunk)
        uint32_t src_off[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t r = (wave * 4 + i) * 4 + (lane >> 4), cphys = lane & 15;
            src_off[i] = r * row_bytes + ((cphys ^ (r & 15u)) << 4);
        }
        // fill slot `slot` with stage u (0 .. UT-1) of tile `tile` (tile clamped by the caller); piece i = 0..3, or
        // all four.  The accumulator seeds of a half tile travel with piece 3 of its first stage.
        auto issue_piece = [&](uint32_t jj, int u, uint32_t slot, int i) {
            const uint32_t tile = tile_first + jj * tile_mul;
            const uint32_t half = (uint32_t)(u / STAGES), s2 = (uint32_t)(u % STAGES);
            const unsigned char* src = gA + (uint64_t)(tile * TILE_ROWS + half * BF_ROWS) * row_bytes + s2 * 256u;
            unsigned char* dst = smem + slot * BF_STAGE_BYTES + wave * 4096u;
            __builtin_amdgcn_global_load_lds((glb_void*)(src + src_off[i]), (lds_void*)(dst + i * 1024), 16, 0, 0);
            if (CINIT && s2 == 0 && i == 3)
                __builtin_amdgcn_global_load_lds((glb_void*)(p.cinit + (uint64_t)tile * TILE_ROWS + half * BF_ROWS + lane),
                                                 (lds_void*)(smem + BF_NBUF * BF_STAGE_BYTES + slot * BF_CSLOT + wave * 256u), 4,
                                                 0, 0);
        };
        auto issue_stage = [&](uint32_t tile, int u, uint32_t slot) {
#pragma unroll
            for (int i = 0; i < 4; ++i) issue_piece(tile, u, slot, i);
        };

        // LDS address of this lane's fragment of k-step 0 of row tile 0 in slot 0: row l31, chunk h ^ (l31 & 15).
        // K-step kk lives at that address ^ (kk << 5) (the chunk index is XOR-swizzled), row tile 1 at +8192.
        const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void*)smem;
        const uint32_t frag0 = lds0 + l31 * 256u + (((h ^ l31) & 15u) << 4);
        // The fragment reads are issued and awaited by hand: with the direct-to-LDS fills always in flight the
        // compiler would wait for ALL outstanding LDS traffic before every MFMA group.  Fragment sets
        // rotate: the reads of k-step kk+2 are issued before the MFMAs of k-step kk, and the wait before those
        // MFMAs leaves the four reads of k-steps kk+1 and kk+2 in flight.  (Four sets, not three: a stage has
        // 8 k-steps, so the set of k-step kk is kk % 4 in every stage, also across half tile boundaries.)
        bf16x8 a[4][2];
        auto read_frags = [&](bf16x8 (&dst)[2], uint32_t slot_, int kk_) {
            const uint32_t addr = (frag0 + slot_ * BF_STAGE_BYTES) ^ ((uint32_t)kk_ << 5);
            asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:8192"
                         : "=&v"(dst[0]), "=&v"(dst[1])
                         : "v"(addr)
                         : "memory");
        };

        // prologue: the first PD stages of the block's run (clamped to its last tile: surplus fills are harmless)
#pragma unroll
        for (int j = 0; j < BF_PD; ++j) issue_stage(min(t0 + (uint32_t)(j / UT), t1 - 1), j % UT, (uint32_t)j);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(bf_inflight_behind<STAGES, CINIT>(0)) : "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int j = 0; j < BF_READ_AHEAD; ++j) read_frags(a[j], 0, j);

        f32x16 acc[2][NT];
        uint32_t slot = 0;  // slot of the stage being multiplied
        for (uint32_t jj = t0; jj < t1; ++jj) {
            const uint32_t tile = tile_first + jj * tile_mul;
            static_for<2>([&](auto half_c) {
                constexpr int half = decltype(half_c)::value;
                static_for<STAGES>([&](auto s_c) {
                    constexpr int s = decltype(s_c)::value;
                    constexpr int u = half * STAGES + s;
                    if (s == 0) {
                        if (CINIT) {
                            const float* cs = (const float*)(smem + BF_NBUF * BF_STAGE_BYTES + slot * BF_CSLOT + wave * 256u);
#pragma unroll
                            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                                for (int j = 0; j < 4; ++j) {
                                    const f32x4 v = *(const f32x4*)(cs + mt * 32 + 8 * j + 4 * h);
#pragma unroll
                                    for (int nt = 0; nt < NT; ++nt) {
                                        acc[mt][nt][4 * j + 0] = v[0];
                                        acc[mt][nt][4 * j + 1] = v[1];
                                        acc[mt][nt][4 * j + 2] = v[2];
                                        acc[mt][nt][4 * j + 3] = v[3];
                                    }
                                }
                        } else {
#pragma unroll
                            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                                    for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.0f;
                        }
                    }
                    static_for<8>([&](auto kk_c) {
                        constexpr int kk = decltype(kk_c)::value;
                        constexpr int g = s * 8 + kk;  // k-step within the dimension
                        if (kk == BF_FILL_KK) {
                            // The NEXT stage: this wave's pieces have landed (the stage after it may be in flight);
                            // behind the barrier everybody's have, and everybody has left the previous stage, whose
                            // slot takes the stage three ahead, one piece per k-step from here on.
                            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(bf_steady_inflight<STAGES, CINIT>(u)) : "memory");
                            __builtin_amdgcn_s_barrier();
                        }
                        if (BF_FILL_SPREAD) {
                            if (kk >= BF_FILL_KK && kk < BF_FILL_KK + 4)
                                issue_piece(min(jj + (uint32_t)((u + BF_PD) / UT), t1 - 1), (u + BF_PD) % UT,
                                            (slot + BF_PD) % BF_NBUF, kk - BF_FILL_KK);
                        } else if (kk == BF_FILL_KK) {
                            issue_stage(min(jj + (uint32_t)((u + BF_PD) / UT), t1 - 1), (u + BF_PD) % UT,
                                        (slot + BF_PD) % BF_NBUF);
                        }
                        if (kk + BF_READ_AHEAD < 8) read_frags(a[(kk + BF_READ_AHEAD) % 4], slot, kk + BF_READ_AHEAD);
                        else  // next stage (junk behind the last)
                            read_frags(a[(kk + BF_READ_AHEAD) % 4], (slot + 1) % BF_NBUF, kk + BF_READ_AHEAD - 8);
                        asm volatile("s_waitcnt lgkmcnt(%2)"
                                     : "+v"(a[kk % 4][0]), "+v"(a[kk % 4][1])
                                     : "n"(2 * BF_READ_AHEAD)
                                     : "memory");
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt)
                                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[kk % 4][mt], bq[nt][g], acc[mt][nt], 0, 0, 0);
                    });
                    slot = (slot + 1) % BF_NBUF;
                });
                // selection behind the half tile
                const uint32_t row0 = tile * TILE_ROWS + half * BF_ROWS;
                if (MODE == MODE_PRE) {
                    const bool plain = !p.valid && !p.filter && (tile + 1) * TILE_ROWS <= p.n;
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                bool ok = true;
                                if (!plain) ok = row_live<MODE_FLAT>(p, row0 + mt * 32 + row_of_reg(r, h));
                                if (ok) gmax[nt][half * 2 + mt] = fmaxf(gmax[nt][half * 2 + mt], acc[mt][nt][r]);
                            }
                } else if (MODE == MODE_DENSE) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        float* dst = p.dense_scores + (uint64_t)my_query[nt] * p.dense_ld + row0 + 4 * h;
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                            for (int j = 0; j < 4; ++j)  // registers 4j .. 4j+3 = rows mt*32 + 8j + 4h .. + 3
                                *(f32x4*)(dst + mt * 32 + 8 * j) = f32x4{acc[mt][nt][4 * j], acc[mt][nt][4 * j + 1],
                                                                          acc[mt][nt][4 * j + 2], acc[mt][nt][4 * j + 3]};
                    }
                } else {
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            push_lane<MODE>(acc[mt][nt], row0 + mt * 32, h, lane, p, list_idx[nt], tau[nt], cnt[nt]);
                }
            });
        }
        // the surplus fills issued behind the last stage must land before the LDS is given back
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
#ifdef BF_STAMP
    if (MODE == MODE_FLAT && tid == 0 && p.tile_off != 0 && p.tile_off + p.tile_len >= p.tiles_per_split) {
        const uint64_t st_c1 = __builtin_amdgcn_s_memtime(), st_r1 = __builtin_amdgcn_s_memrealtime();
        uint64_t* out = (uint64_t*)p.pre_max + (uint64_t)blockIdx.x * 2;
        out[0] = st_c1 - st_c0;
        out[1] = st_r1 - st_r0;
    }
#endif

    if (MODE == MODE_DENSE) return;
    if (MODE == MODE_PRE) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            float* dst = p.pre_max + (uint64_t)(q0 + nt * 32 + l31) * (8 * p.n_splits) + (split * 2 + h) * 4;
#pragma unroll
            for (int g = 0; g < 4; ++g) dst[g] = gmax[nt][g];
        }
        return;
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        uint64_t need = __builtin_amdgcn_ballot_w64(cnt[nt] > p.kprime);
        while (need) {
            const uint32_t b = (uint32_t)__builtin_ctzll(need);
            need &= need - 1;
            const uint32_t n_e = (uint32_t)__builtin_amdgcn_readlane((int)cnt[nt], (int)b);
            const uint32_t li = (uint32_t)__builtin_amdgcn_readlane((int)list_idx[nt], (int)b);
            wave_compact_list(p.lists + (uint64_t)li * p.cap, n_e, p.kprime, lane);
            if (lane == b) cnt[nt] = p.kprime;
        }
        if (MODE != MODE_IVF || q_live[nt]) p.list_counts[list_idx[nt]] = cnt[nt];
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The same scan on v_mfma_f32_16x16x32_bf16.  Cycles per flop, LDS traffic and register budget equal the 32x32x16
// form's (per 32 dimensions: 4 fragment reads feed 4 x NTQ MFMAs of 16 cycles), but on random operands the chip holds
// a higher clock under this shape.
//   * a wave multiplies the 64 staged rows (4 row tiles of 16) with its 16 * NTQ queries (NTQ query tiles of 16);
//   * lane (l15 = lane & 15, quad = lane >> 4) holds, of query tile t, dims 32 ks + 8 quad .. +7 of query
//     q0 + 16 t + l15 (B operand) and reads dims 8 quad .. +7 of k-step ks of row 16 mt + l15 (A operand):
//     logical chunk 4 ks + quad of the row, at physical chunk (4 ks + quad) ^ l15 -- the 16 lanes of every
//     ds_read_b128 phase land on 16 different chunk positions, i.e. all 64 banks;
//   * result register r of tile (mt, t) is row 16 mt + 4 quad + r against the lane's query of tile t: column = lane,
//     as before, so thresholds stay per-lane registers; the four quads of a query keep a candidate list each.
// ---------------------------------------------------------------------------------------------------------------
template <int MODE>
__device__ inline void push_lane16(const f32x4 (&acc)[4], uint32_t row_base, uint32_t quad, uint32_t lane, const ScanParams& p,
                                   uint32_t list_idx, float& tau, uint32_t& cnt) {
    float gm[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) gm[mt] = fmaxf(fmaxf(acc[mt][0], acc[mt][1]), fmaxf(acc[mt][2], acc[mt][3]));
    const float m = fmaxf(fmaxf(gm[0], gm[1]), fmaxf(gm[2], gm[3]));
    if (__builtin_amdgcn_ballot_w64(m > tau) == 0) return;  // wave-uniform
    uint2* mylist = p.lists + (uint64_t)list_idx * p.cap;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        if (gm[mt] > tau) {  // rows 16 mt + 4 quad .. + 3 of the half tile
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float key = acc[mt][j];
                if (key > tau) {
                    const uint32_t row = row_base + 16 * mt + 4 * quad + j;
                    if (row_live<MODE>(p, row)) {
                        mylist[cnt] = make_uint2(__float_as_uint(key), row);
                        cnt += 1;
                    }
                }
            }
        }
    }
    uint64_t need = __builtin_amdgcn_ballot_w64(cnt > p.cap - 16);
    while (need) {
        const uint32_t b = (uint32_t)__builtin_ctzll(need);
        need &= need - 1;
        const uint32_t n_e = (uint32_t)__builtin_amdgcn_readlane((int)cnt, (int)b);
        const uint32_t li = (uint32_t)__builtin_amdgcn_readlane((int)list_idx, (int)b);
        const float t = wave_compact_list(p.lists + (uint64_t)li * p.cap, n_e, p.kprime, lane);
        if (lane == b) { tau = t; cnt = p.kprime; }
    }
}

// NTQ query tiles of 16 per wave, KS32 k-steps of 32 dims (dpad = 32 * KS32), WAVES waves per workgroup (4: one per
// SIMD with the whole register file; 8: two per SIMD, so that one wave's selection and list pushes run beside the
// other's MFMAs).  Queries per workgroup = 16 * NTQ * WAVES.
#ifndef BF_FLAT_WG4X2
#define BF_FLAT_WG4X2 0  // 1: dimensions up to 384 run two workgroups of four waves (32 queries each) per CU instead of one of eight
#endif
template <int MODE, int NTQ, int KS32, bool CINIT, int WAVES>
__global__ __launch_bounds__(64 * WAVES, (WAVES == 8 || (BF_FLAT_WG4X2 && KS32 <= 12)) ? 2 : 1) void flat_scan_bf16x_kernel(ScanParams p) {
    constexpr int CSLOT = 256 * WAVES;   // per slot: the half tile's accumulator seeds, once per wave
    constexpr int PIECES = 16 / WAVES;   // fill pieces (4 rows x 256 B) a wave issues per stage
    __shared__ __attribute__((aligned(1024))) unsigned char smem[BF_NBUF * (BF_STAGE_BYTES + CSLOT)];
    constexpr int STAGES = KS32 / 4;     // stages (128 dims) per half tile
    constexpr int UT = 2 * STAGES;       // stages per tile
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const uint32_t l15 = lane & 15, quad = lane >> 4;
    constexpr uint32_t QW = 16 * NTQ;    // queries per wave
    constexpr uint32_t QB = WAVES * QW;

    uint32_t split = 0, q0 = 0, tile_first, tile_mul, cnt_tiles;
    uint32_t my_query[NTQ];
    bool q_live[NTQ];
    float tau[NTQ];
    uint32_t cnt[NTQ], list_idx[NTQ];
    float gmax[NTQ][2];
    // A list is probed by far fewer queries than a workgroup has slots for: a wave whose slots are all empty keeps
    // staging rows and meeting the barriers, but leaves the matrix pipe to the wave it shares a SIMD with.
    bool wave_live = true;  // wave-uniform
    if (MODE == MODE_IVF) {
        const IvfItem it = p.items[blockIdx.x];
        wave_live = wave * QW < it.nq;
        tile_first = it.tile0;
        tile_mul = 1;
        cnt_tiles = it.tile1 - it.tile0;
        const uint2* iq = p.item_queries + it.qoff;
#pragma unroll
        for (int t = 0; t < NTQ; ++t) {
            const uint32_t jq = wave * QW + t * 16 + l15;
            q_live[t] = jq < it.nq;
            const uint2 e = q_live[t] ? iq[jq] : make_uint2(0u, 0u);
            my_query[t] = e.x;
            list_idx[t] = (e.y + it.list_add) * 4 + quad;  // every quad owns a list
        }
    } else {
        uint32_t q_tile;
        if (p.xcd_map) {
            const uint32_t xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
            split = xcd + 8 * (j / p.q_tiles);
            q_tile = j % p.q_tiles;
        } else {
            split = blockIdx.x / p.q_tiles;
            q_tile = blockIdx.x % p.q_tiles;
        }
        q0 = q_tile * QB + wave * QW;
        const uint32_t own = split < p.n_tiles ? (p.n_tiles - split + p.n_splits - 1) / p.n_splits : 0u;
        uint32_t jb = 0, jstep = 1;
        if (MODE == MODE_PRE) {
            jstep = max(p.pre_stride, 1u);
            cnt_tiles = min(p.pre_tiles, (own + jstep - 1) / jstep);
        } else {
            jb = min(p.tile_off, own);
            cnt_tiles = min(p.tile_len, own - jb);
        }
        tile_mul = jstep * p.n_splits;
        tile_first = split + jb * p.n_splits;
#pragma unroll
        for (int t = 0; t < NTQ; ++t) {
            my_query[t] = q0 + t * 16 + l15;
            q_live[t] = my_query[t] < p.n_queries;
            list_idx[t] = (my_query[t] * p.n_splits + split) * 4 + quad;
        }
        if (BF_SKIPS_DEAD(MODE)) wave_live = q0 < p.n_queries;  // the padding slots of the last query block
    }
    const uint32_t t0 = 0, t1 = cnt_tiles;
#pragma unroll
    for (int t = 0; t < NTQ; ++t) {
        tau[t] = -INFINITY;
        if (MODE != MODE_PRE && p.tau0 && q_live[t]) tau[t] = p.tau0[my_query[t]];
        if ((p.debug & 1u) || !q_live[t]) tau[t] = INFINITY;
        cnt[t] = (MODE == MODE_FLAT && p.resume) ? p.list_counts[list_idx[t]] : 0u;
        gmax[t][0] = gmax[t][1] = -INFINITY;
    }

#ifdef BF_STAMP
    const uint64_t st_c0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    if (t1 > t0) {  // block-uniform
        // stationary query fragments, pinned in the accumulation half of the register file
        bf16x8 bq[NTQ][KS32];
        const uint16_t* __restrict__ gB = (const uint16_t*)p.query_op;
#pragma unroll
        for (int t = 0; t < NTQ; ++t) {
            const uint16_t* qrow = gB + (uint64_t)my_query[t] * p.dpad + 8 * quad;
#pragma unroll
            for (int ks = 0; ks < KS32; ++ks)
                bq[t][ks] = (!BF_SKIPS_DEAD(MODE) || wave_live) ? __builtin_bit_cast(bf16x8, *(const uint4*)(qrow + ks * 32)) : bf16x8{};
        }
#pragma unroll
        for (int t = 0; t < NTQ; ++t)
#pragma unroll
            for (int ks = 0; ks < KS32; ++ks) asm volatile("" : "+a"(bq[t][ks]));

        const unsigned char* __restrict__ gA = (const unsigned char*)p.base_op;
        const uint32_t row_bytes = p.dpad * 2;
        uint32_t src_off[PIECES];
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            const uint32_t r = (wave * PIECES + i) * 4 + (lane >> 4), cphys = lane & 15;
            src_off[i] = r * row_bytes + ((cphys ^ (r & 15u)) << 4);
        }
        auto issue_piece = [&](uint32_t jj, int u, uint32_t slot, int i) {
#ifdef BF_DEV_NOFILL  // development, timing only: after the prologue nothing is staged any more
            if (jj != t0) return;
#endif
            const uint32_t tile = tile_first + jj * tile_mul;
            const uint32_t half = (uint32_t)(u / STAGES), s2 = (uint32_t)(u % STAGES);
            const unsigned char* src = gA + (uint64_t)(tile * TILE_ROWS + half * BF_ROWS) * row_bytes + s2 * 256u;
            unsigned char* dst = smem + slot * BF_STAGE_BYTES + wave * (PIECES * 1024u);
            __builtin_amdgcn_global_load_lds((glb_void*)(src + src_off[i]), (lds_void*)(dst + i * 1024), 16, 0, 0);
            if (CINIT && s2 == 0 && i == PIECES - 1)
                __builtin_amdgcn_global_load_lds((glb_void*)(p.cinit + (uint64_t)tile * TILE_ROWS + half * BF_ROWS + lane),
                                                 (lds_void*)(smem + BF_NBUF * BF_STAGE_BYTES + slot * CSLOT + wave * 256u), 4,
                                                 0, 0);
        };
        auto issue_stage = [&](uint32_t tile, int u, uint32_t slot) {
#pragma unroll
            for (int i = 0; i < PIECES; ++i) issue_piece(tile, u, slot, i);
        };

        // this lane's fragment of k-step 0 of row tile 0 in slot 0: row l15, physical chunk quad ^ l15; k-step kk of the
        // stage at that address ^ (kk << 6), row tile mt at + mt * 4096
        const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void*)smem;
        const uint32_t frag0 = lds0 + l15 * 256u + (((quad ^ l15) & 15u) << 4);
        bf16x8 a[2][4];  // two fragment sets: the reads of k-step kk + 1 are issued before the MFMAs of k-step kk
        auto read_frags = [&](bf16x8 (&dst)[4], uint32_t slot_, int kk_) {
            const uint32_t addr = (frag0 + slot_ * BF_STAGE_BYTES) ^ ((uint32_t)kk_ << 6);
            asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:4096\n\tds_read_b128 %2, %4 offset:8192\n\t"
                         "ds_read_b128 %3, %4 offset:12288"
                         : "=&v"(dst[0]), "=&v"(dst[1]), "=&v"(dst[2]), "=&v"(dst[3])
                         : "v"(addr)
                         : "memory");
        };

#pragma unroll
        for (int j = 0; j < BF_PD; ++j) issue_stage(min(t0 + (uint32_t)(j / UT), t1 - 1), j % UT, (uint32_t)j);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(bf_inflight_behind<STAGES, CINIT, PIECES>(0)) : "memory");
        __builtin_amdgcn_s_barrier();
        uint32_t slot = 0;
        if (BF_SKIPS_DEAD(MODE) && !wave_live) {
            // no live query in this wave: it keeps its share of the staging and meets every barrier, nothing else
            for (uint32_t jj = t0; jj < t1; ++jj) {
                static_for<UT>([&](auto u_c) {
                    constexpr int u = decltype(u_c)::value;
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(bf_steady_inflight<STAGES, CINIT, PIECES>(u)) : "memory");
                    __builtin_amdgcn_s_barrier();
#pragma unroll
                    for (int i = 0; i < PIECES; ++i)
                        issue_piece(min(jj + (uint32_t)((u + BF_PD) / UT), t1 - 1), (u + BF_PD) % UT, (slot + BF_PD) % BF_NBUF, i);
                    slot = (slot + 1) % BF_NBUF;
                });
            }
        } else {
        read_frags(a[0], 0, 0);

        f32x4 acc[4][NTQ];
        const bool lag = BF_STAGGER && WAVES == 8 && wave >= 4;  // wave-uniform
        for (uint32_t jj = t0; jj < t1; ++jj) {
            const uint32_t tile = tile_first + jj * tile_mul;
            static_for<2>([&](auto half_c) {
                constexpr int half = decltype(half_c)::value;
                static_for<STAGES>([&](auto s_c) {
                    constexpr int s = decltype(s_c)::value;
                    constexpr int u = half * STAGES + s;
                    if (s == 0) {
                        if (CINIT) {
                            const float* cs = (const float*)(smem + BF_NBUF * BF_STAGE_BYTES + slot * CSLOT + wave * 256u);
#pragma unroll
                            for (int mt = 0; mt < 4; ++mt) {
                                const f32x4 v = *(const f32x4*)(cs + 16 * mt + 4 * quad);
#pragma unroll
                                for (int t = 0; t < NTQ; ++t) acc[mt][t] = v;
                            }
                        } else {
#pragma unroll
                            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                                for (int t = 0; t < NTQ; ++t) acc[mt][t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                        }
                    }
                    static_for<4>([&](auto kk_c) {
                        constexpr int kk = decltype(kk_c)::value;
                        constexpr int g = s * 4 + kk;  // k-step within the dimension
                        // Once per stage: the NEXT stage has landed for this wave; behind the barrier for everybody, and
                        // everybody has left the previous stage, whose slot takes the stage BF_PD ahead (half of the wave's
                        // pieces per k-step).  Waves 0-3 do this at k-step 2 of the stage, waves 4-7 at k-step 0: the
                        // barrier then holds the second wave of every SIMD half a stage behind the first, so that one's
                        // selection phase runs beside the other's MFMAs instead of both stalling together.
                        if ((kk < 2) == lag) {
                            if (kk == 0 || kk == 2) {
                                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(bf_steady_inflight<STAGES, CINIT, PIECES>(u)) : "memory");
                                __builtin_amdgcn_s_barrier();
                            }
#pragma unroll
                            for (int i = 0; i < PIECES / 2; ++i)
                                issue_piece(min(jj + (uint32_t)((u + BF_PD) / UT), t1 - 1), (u + BF_PD) % UT,
                                            (slot + BF_PD) % BF_NBUF, (PIECES / 2) * (kk & 1) + i);
                        }
                        if (kk + 1 < 4) read_frags(a[(kk + 1) % 2], slot, kk + 1);
                        else read_frags(a[(kk + 1) % 2], (slot + 1) % BF_NBUF, 0);  // next stage (junk behind the last)
                        asm volatile("s_waitcnt lgkmcnt(4)"
                                     : "+v"(a[kk % 2][0]), "+v"(a[kk % 2][1]), "+v"(a[kk % 2][2]), "+v"(a[kk % 2][3])
                                     :
                                     : "memory");
#pragma unroll
                        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                            for (int t = 0; t < NTQ; ++t)
                                acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kk % 2][mt], bq[t][g], acc[mt][t], 0, 0, 0);
                    });
                    slot = (slot + 1) % BF_NBUF;
                });
                // selection behind the half tile
                const uint32_t row0 = tile * TILE_ROWS + half * BF_ROWS;
                if (MODE == MODE_PRE) {
                    const bool plain = !p.valid && !p.filter && (tile + 1) * TILE_ROWS <= p.n;
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                        for (int t = 0; t < NTQ; ++t)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                bool ok = true;
                                if (!plain) ok = row_live<MODE_FLAT>(p, row0 + 16 * mt + 4 * quad + r);
                                if (ok) gmax[t][half] = fmaxf(gmax[t][half], acc[mt][t][r]);
                            }
                } else if (MODE == MODE_DENSE) {
#pragma unroll
                    for (int t = 0; t < NTQ; ++t) {
                        float* dst = p.dense_scores + (uint64_t)my_query[t] * p.dense_ld + row0 + 4 * quad;
#pragma unroll
                        for (int mt = 0; mt < 4; ++mt) *(f32x4*)(dst + 16 * mt) = acc[mt][t];
                    }
                } else {
#ifdef BF_DEV_NOSEL  // development, timing only: the accumulators are kept alive, nothing is selected
#pragma unroll
                    for (int t = 0; t < NTQ; ++t)
#pragma unroll
                        for (int mt = 0; mt < 4; ++mt) asm volatile("" ::"v"(acc[mt][t]));
#else
#pragma unroll
                    for (int t = 0; t < NTQ; ++t) {
                        const f32x4 col[4] = {acc[0][t], acc[1][t], acc[2][t], acc[3][t]};
                        push_lane16<MODE>(col, row0, quad, lane, p, list_idx[t], tau[t], cnt[t]);
                    }
#endif
                }
            });
        }
        }  // live wave
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
#ifdef BF_STAMP
    if (MODE == MODE_FLAT && tid == 0 && p.tile_off != 0 && p.tile_off + p.tile_len >= p.tiles_per_split) {
        const uint64_t st_c1 = __builtin_amdgcn_s_memtime(), st_r1 = __builtin_amdgcn_s_memrealtime();
        uint64_t* out = (uint64_t*)p.pre_max + (uint64_t)blockIdx.x * 2;
        out[0] = st_c1 - st_c0;
        out[1] = st_r1 - st_r0;
    }
#endif

    if (MODE == MODE_DENSE) return;
    if (MODE == MODE_PRE) {
        // 8 group maxima per (query, slice): (quad, half), 16 rows of every sampled tile each
#pragma unroll
        for (int t = 0; t < NTQ; ++t) {
            float* dst = p.pre_max + (uint64_t)(q0 + t * 16 + l15) * (8 * p.n_splits) + (split * 4 + quad) * 2;
            dst[0] = gmax[t][0];
            dst[1] = gmax[t][1];
        }
        return;
    }
#pragma unroll
    for (int t = 0; t < NTQ; ++t) {
        uint64_t need = __builtin_amdgcn_ballot_w64(cnt[t] > p.kprime);
        while (need) {
            const uint32_t b = (uint32_t)__builtin_ctzll(need);
            need &= need - 1;
            const uint32_t n_e = (uint32_t)__builtin_amdgcn_readlane((int)cnt[t], (int)b);
            const uint32_t li = (uint32_t)__builtin_amdgcn_readlane((int)list_idx[t], (int)b);
            wave_compact_list(p.lists + (uint64_t)li * p.cap, n_e, p.kprime, lane);
            if (lane == b) cnt[t] = p.kprime;
        }
        if (MODE != MODE_IVF || q_live[t]) p.list_counts[list_idx[t]] = cnt[t];
    }
}

#ifndef BF_MF16
#define BF_MF16 1  // 1: the 16x16x32 form (four candidate lists per query and slice); 0: the 32x32x16 form (two)
#endif
#ifndef BF_WAVES8
#define BF_WAVES8 1  // 1: dimensions up to 384 run eight waves of 32 queries per workgroup; 0: four waves of 64
#endif
uint32_t flat_scan_bf16_lists_per_slice() { return BF_MF16 ? 4 : 2; }
constexpr int BF_NTQ = (BF_WAVES8 || BF_FLAT_WG4X2) ? 2 : 4, BF_NW = (BF_WAVES8 && !BF_FLAT_WG4X2) ? 8 : 4;

// dpad (dims padded to a supported k-step count) and queries per workgroup for a dimension; 0 = unsupported
uint32_t flat_scan_bf16_dpad(uint32_t dim) {
    const uint32_t ks = (dim + 15) / 16;
    if (ks <= 8) return 128;
    if (ks <= 16) return 256;
    if (ks <= 24) return 384;
    if (ks <= 32) return 512;
    if (ks <= 48) return 768;
    return 0;
}
uint32_t flat_scan_bf16_queries_per_block(uint32_t dpad) { return (dpad <= 384 && BF_MF16) ? 16u * BF_NTQ * BF_NW : (dpad <= 384 ? 256u : 128u); }
#ifndef BF_IVF_WAVES
#define BF_IVF_WAVES 4  // waves per workgroup of the inverted-list scan up to 384 dimensions: 128 query slots per work item, two
                        // items per CU (measured 2.5 % faster than 8 waves / 256 slots at nlist 4096, nprobe 32, 10K queries)
#endif
// query slots of an inverted-list work item
uint32_t flat_scan_bf16_ivf_queries_per_block(uint32_t dpad) { return (BF_MF16 && dpad <= 384) ? 32u * BF_IVF_WAVES : flat_scan_bf16_queries_per_block(dpad); }

template <int MODE, bool CINIT>
static hipError_t launch_bf16_mode(const ScanParams& p, hipStream_t stream) {
    const dim3 grid(p.n_splits * p.q_tiles);
    dim3 block(256);
#if BF_MF16
    if (p.dpad <= 384) block = dim3(64 * BF_NW);
    switch (p.dpad) {
    case 128: hipLaunchKernelGGL((flat_scan_bf16x_kernel<MODE, BF_NTQ, 4, CINIT, BF_NW>), grid, block, 0, stream, p); break;
    case 256: hipLaunchKernelGGL((flat_scan_bf16x_kernel<MODE, BF_NTQ, 8, CINIT, BF_NW>), grid, block, 0, stream, p); break;
    case 384: hipLaunchKernelGGL((flat_scan_bf16x_kernel<MODE, BF_NTQ, 12, CINIT, BF_NW>), grid, block, 0, stream, p); break;
    case 512: hipLaunchKernelGGL((flat_scan_bf16x_kernel<MODE, 2, 16, CINIT, 4>), grid, block, 0, stream, p); break;
    case 768: hipLaunchKernelGGL((flat_scan_bf16x_kernel<MODE, 2, 24, CINIT, 4>), grid, block, 0, stream, p); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
#endif
    switch (p.dpad) {
    case 128: hipLaunchKernelGGL((flat_scan_bf16_kernel<MODE, 2, 8, CINIT>), grid, block, 0, stream, p); break;
    case 256: hipLaunchKernelGGL((flat_scan_bf16_kernel<MODE, 2, 16, CINIT>), grid, block, 0, stream, p); break;
    case 384: hipLaunchKernelGGL((flat_scan_bf16_kernel<MODE, 2, 24, CINIT>), grid, block, 0, stream, p); break;
    case 512: hipLaunchKernelGGL((flat_scan_bf16_kernel<MODE, 1, 32, CINIT>), grid, block, 0, stream, p); break;
    case 768: hipLaunchKernelGGL((flat_scan_bf16_kernel<MODE, 1, 48, CINIT>), grid, block, 0, stream, p); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// inverted-list scan (Euclidean: the accumulators are always seeded): one workgroup per p.items[i]
hipError_t launch_ivf_scan_bf16(const ScanParams& p, uint32_t n_items, hipStream_t stream) {
    if (n_items == 0) return hipSuccess;
    if (!p.cinit) return hipErrorInvalidValue;
    const dim3 grid(n_items);
    dim3 block(256);
#if BF_MF16
    if (p.dpad <= 384) block = dim3(64 * BF_IVF_WAVES);
    switch (p.dpad) {
    case 128: hipLaunchKernelGGL((flat_scan_bf16x_kernel<MODE_IVF, 2, 4, true, BF_IVF_WAVES>), grid, block, 0, stream, p); break;
    case 256: hipLaunchKernelGGL((flat_scan_bf16x_kernel<MODE_IVF, 2, 8, true, BF_IVF_WAVES>), grid, block, 0, stream, p); break;
    case 384: hipLaunchKernelGGL((flat_scan_bf16x_kernel<MODE_IVF, 2, 12, true, BF_IVF_WAVES>), grid, block, 0, stream, p); break;
    case 512: hipLaunchKernelGGL((flat_scan_bf16x_kernel<MODE_IVF, 2, 16, true, 4>), grid, block, 0, stream, p); break;
    case 768: hipLaunchKernelGGL((flat_scan_bf16x_kernel<MODE_IVF, 2, 24, true, 4>), grid, block, 0, stream, p); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
#endif
    switch (p.dpad) {
    case 128: hipLaunchKernelGGL((flat_scan_bf16_kernel<MODE_IVF, 2, 8, true>), grid, block, 0, stream, p); break;
    case 256: hipLaunchKernelGGL((flat_scan_bf16_kernel<MODE_IVF, 2, 16, true>), grid, block, 0, stream, p); break;
    case 384: hipLaunchKernelGGL((flat_scan_bf16_kernel<MODE_IVF, 2, 24, true>), grid, block, 0, stream, p); break;
    case 512: hipLaunchKernelGGL((flat_scan_bf16_kernel<MODE_IVF, 1, 32, true>), grid, block, 0, stream, p); break;
    case 768: hipLaunchKernelGGL((flat_scan_bf16_kernel<MODE_IVF, 1, 48, true>), grid, block, 0, stream, p); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

template <bool CINIT>
static hipError_t launch_bf16_dense(const ScanParams& p, hipStream_t stream) {
    const dim3 grid(p.n_splits * p.q_tiles);
    dim3 block(256);
#if BF_MF16
    if (p.dpad <= 384) block = dim3(64 * BF_NW);
    switch (p.dpad) {
    case 128: hipLaunchKernelGGL((flat_scan_bf16x_kernel<MODE_DENSE, BF_NTQ, 4, CINIT, BF_NW>), grid, block, 0, stream, p); break;
    case 256: hipLaunchKernelGGL((flat_scan_bf16x_kernel<MODE_DENSE, BF_NTQ, 8, CINIT, BF_NW>), grid, block, 0, stream, p); break;
    case 384: hipLaunchKernelGGL((flat_scan_bf16x_kernel<MODE_DENSE, BF_NTQ, 12, CINIT, BF_NW>), grid, block, 0, stream, p); break;
    case 512: hipLaunchKernelGGL((flat_scan_bf16x_kernel<MODE_DENSE, 2, 16, CINIT, 4>), grid, block, 0, stream, p); break;
    case 768: hipLaunchKernelGGL((flat_scan_bf16x_kernel<MODE_DENSE, 2, 24, CINIT, 4>), grid, block, 0, stream, p); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
#endif
    switch (p.dpad) {
    case 128: hipLaunchKernelGGL((flat_scan_bf16_kernel<MODE_DENSE, 2, 8, CINIT>), grid, block, 0, stream, p); break;
    case 256: hipLaunchKernelGGL((flat_scan_bf16_kernel<MODE_DENSE, 2, 16, CINIT>), grid, block, 0, stream, p); break;
    case 384: hipLaunchKernelGGL((flat_scan_bf16_kernel<MODE_DENSE, 2, 24, CINIT>), grid, block, 0, stream, p); break;
    case 512: hipLaunchKernelGGL((flat_scan_bf16_kernel<MODE_DENSE, 1, 32, CINIT>), grid, block, 0, stream, p); break;
    case 768: hipLaunchKernelGGL((flat_scan_bf16_kernel<MODE_DENSE, 1, 48, CINIT>), grid, block, 0, stream, p); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// every score of [q_pad] x [n_pad] written to p.dense_scores (row stride p.dense_ld >= n_pad)
hipError_t launch_dense_scores_bf16(const ScanParams& p0, hipStream_t stream) {
    if (p0.n_splits * p0.q_tiles == 0) return hipSuccess;
    if (!p0.dense_scores) return hipErrorInvalidValue;
    ScanParams p = p0;
    p.tile_off = 0;
    p.tile_len = p.tiles_per_split;
    return p.cinit ? launch_bf16_dense<true>(p, stream) : launch_bf16_dense<false>(p, stream);
}

hipError_t launch_flat_scan_bf16(const ScanParams& p0, bool prepass, hipStream_t stream) {
    if (p0.n_splits * p0.q_tiles == 0) return hipSuccess;
    ScanParams p = p0;
    if (p.tile_len == 0) { p.tile_off = 0; p.tile_len = p.tiles_per_split; }  // unset: the whole slice
    if (p.cinit)
        return prepass ? launch_bf16_mode<MODE_PRE, true>(p, stream) : launch_bf16_mode<MODE_FLAT, true>(p, stream);
    return prepass ? launch_bf16_mode<MODE_PRE, false>(p, stream) : launch_bf16_mode<MODE_FLAT, false>(p, stream);
}

}  // namespace nh
