// bf16 first pass of the exact flat search on the matrix cores, f32 accumulate: flat_scan_bf16x_kernel
// (v_mfma_f32_16x16x32_bf16; eight waves of 32 queries per workgroup up to 384 dimensions, four waves beyond).
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
// Selection: each lane compares its results per tile with its query's threshold; every lane quarter sharing a query's
// column owns its candidate list (query, slice, quarter), so a push is a predicated store and a counter increment
// with no cross-lane step.
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
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

#ifndef BF_STAGGER
#define BF_STAGGER 0      // 1: eight-wave workgroups hold waves 4-7 half a stage behind waves 0-3 (measured 1-2 % slower, and 2-3 %
                          // slower when the lagging waves also fill half a stage later than the leading ones)
#endif
#ifndef BF_FLAT_SKIP_DEAD
#define BF_FLAT_SKIP_DEAD 0  // 1: waves of the flat scan whose query slots are all padding skip the matrix work too
#endif
#define BF_SKIPS_DEAD(MODE) ((MODE) == MODE_IVF || (BF_FLAT_SKIP_DEAD && (MODE) == MODE_FLAT))
constexpr int BF_ROWS = 64;                               // rows per stage: half a 128-row tile
constexpr int BF_BK = 128;                                // dims per stage
constexpr int BF_STAGE_BYTES = BF_ROWS * BF_BK * 2;       // 16 KB
// The ring of staged half-tile slices.  A workgroup meets at a barrier every `be` stages; between two barriers its waves
// may drift up to `be - 1` stages apart, so a wave held up by a list push does not hold up the other seven at once
// (two stages per barrier against one: 1.5 % on 1M and 10M rows, 3-5 % on shards of 125K-250K rows, SQ8 1.4 %; three
// stages no better).  Fills run `pd` stages ahead of the stage being multiplied and the ring has pd + be slots: the
// barrier of stage u proves stages u + 1 .. u + be landed and every wave out of stages <= u - 1, whose slots take the
// fills issued during stages u .. u + be - 1.  Inverted-list scans keep one stage per barrier: their two workgroups per
// CU leave room for four slots each.
#ifndef BF_BARRIER_EVERY
#define BF_BARRIER_EVERY 2
#endif
#ifndef BF_RING_AHEAD
#define BF_RING_AHEAD 4
#endif
#if defined(BF_FLAT_WG4X2) && BF_FLAT_WG4X2 && BF_BARRIER_EVERY != 1
#error "two four-wave workgroups per CU leave room for four ring slots each: build with -DBF_BARRIER_EVERY=1 -DBF_RING_AHEAD=3"
#endif
template <int MODE> constexpr int bf_be() { return MODE == MODE_IVF ? 1 : BF_BARRIER_EVERY; }
template <int MODE> constexpr int bf_pd() { return MODE == MODE_IVF ? 3 : BF_RING_AHEAD; }
template <int MODE> constexpr int bf_nbuf() { return bf_pd<MODE>() + bf_be<MODE>(); }

// byte offset of 16-byte chunk c (0..15) of row r (0..63) inside a stage: rows are 256 B = one sweep of
// the LDS banks, so the 16 rows one ds_read_b128 phase touches are spread over the 16 chunk positions
__device__ inline uint32_t bf_lds_off(uint32_t r, uint32_t c) { return r * 256u + ((c ^ (r & 15u)) << 4); }

// direct-to-LDS loads a wave issues for stage u of a tile (u = half * STAGES + s, taken modulo the tile)
template <int STAGES, bool CINIT, int PIECES = 4>
constexpr int bf_stage_loads(int u) { return PIECES + ((CINIT && (u % STAGES) == 0) ? 1 : 0); }
// loads issued behind stage u that may still be in flight when stage u is awaited
// loads that may stay in flight while stage u + 1 is awaited at the fill point of stage u: those of stages
// u + 2 .. u + PD - 1 (stage u + PD is issued right behind that wait)
// (a barrier every second stage awaits stages u + 1 AND u + 2: those of stages u + 3 .. u + PD - 1 may stay in flight)
template <int STAGES, bool CINIT, int PIECES, int PD, int BE>
constexpr int bf_steady_inflight(int u) {
    int n = 0;
    for (int j = 1 + BE; j < PD; ++j) n += bf_stage_loads<STAGES, CINIT, PIECES>(u + j);
    return n;
}
template <int STAGES, bool CINIT, int PIECES, int PD>
constexpr int bf_inflight_behind(int u) {
    int n = 0;
    for (int j = 1; j < PD; ++j) n += bf_stage_loads<STAGES, CINIT, PIECES>(u + j);
    return n;
}

// ---------------------------------------------------------------------------------------------------------------
// The tile instruction is v_mfma_f32_16x16x32_bf16 (per 32 dimensions: 4 fragment reads feed 4 x NTQ MFMAs of 16
// cycles): cycles per flop, LDS traffic and register budget equal the 32x32x16 form's, but on random operands the
// chip holds a higher clock under this shape.
//   * a wave multiplies the 64 staged rows (4 row tiles of 16) with its 16 * NTQ queries (NTQ query tiles of 16);
//   * lane (l15 = lane & 15, quad = lane >> 4) holds, of query tile t, dims 32 ks + 8 quad .. +7 of query
//     q0 + 16 t + l15 (B operand) and reads dims 8 quad .. +7 of k-step ks of row 16 mt + l15 (A operand):
//     logical chunk 4 ks + quad of the row, at physical chunk (4 ks + quad) ^ l15 -- the 16 lanes of every
//     ds_read_b128 phase land on 16 different chunk positions, i.e. all 64 banks;
//   * result register r of tile (mt, t) is row 16 mt + 4 quad + r against the lane's query of tile t: column = lane,
//     so thresholds stay per-lane registers; the four quads of a query keep a candidate list each.
// ---------------------------------------------------------------------------------------------------------------
// `pad_tile` (wave-uniform): the tile holds rows beyond n.  The flat scan receives row masks folded into the accumulator
// seeds (a dead row starts at -inf), so a listed row needs no mask lookup: the check left is row < n, in the last tile only.
template <int MODE>
__device__ inline void push_lane16(const f32x4 (&acc)[4], uint32_t row_base, uint32_t quad, uint32_t lane, const ScanParams& p,
                                   uint32_t list_idx, float& tau, uint32_t& cnt, bool pad_tile) {
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
                    if (MODE == MODE_FLAT ? (!pad_tile || row < p.n) : row_live<MODE>(p, row)) {
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
    constexpr int BF_BE = bf_be<MODE>(), BF_PD = bf_pd<MODE>(), BF_NBUF = bf_nbuf<MODE>();  // stages per barrier, stages ahead, ring slots
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
        // the grid is an upper bound sized on the host; the items this launch really has are counted on the device
        uint32_t item_idx = blockIdx.x;
        if (p.item_bounds) {
            item_idx += p.item_bounds[0];
            if (item_idx >= p.item_bounds[1]) return;  // workgroup-uniform, before any barrier
        }
        const IvfItem it = p.items[item_idx];
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
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(bf_inflight_behind<STAGES, CINIT, PIECES, BF_PD>(0)) : "memory");
        __builtin_amdgcn_s_barrier();
        uint32_t slot = 0;
        if (BF_SKIPS_DEAD(MODE) && !wave_live) {
            // no live query in this wave: it keeps its share of the staging and meets every barrier, nothing else
            for (uint32_t jj = t0; jj < t1; ++jj) {
                static_for<UT>([&](auto u_c) {
                    constexpr int u = decltype(u_c)::value;
                    if (u % BF_BE == 0) {
                        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(bf_steady_inflight<STAGES, CINIT, PIECES, BF_PD, BF_BE>(u)) : "memory");
                        __builtin_amdgcn_s_barrier();
                    }
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
                            // (with a barrier every second stage the waves of a workgroup may drift a stage apart between
                            // two of them: a wave held up by a list push no longer holds up the other seven at once)
                            if ((kk == 0 || kk == 2) && u % BF_BE == 0) {
                                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(bf_steady_inflight<STAGES, CINIT, PIECES, BF_PD, BF_BE>(u)) : "memory");
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
#ifdef BF_DEV_I8SIM  // development, timing only (results are garbage): the same loop issuing v_mfma_i32_16x16x64_i8 on the same registers
                                acc[mt][t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_mfma_i32_16x16x64_i8(
                                    __builtin_bit_cast(i32x4, a[kk % 2][mt]), __builtin_bit_cast(i32x4, bq[t][g]),
                                    __builtin_bit_cast(i32x4, acc[mt][t]), 0, 0, 0));
#else
                                acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kk % 2][mt], bq[t][g], acc[mt][t], 0, 0, 0);
#endif
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
                        push_lane16<MODE>(col, row0, quad, lane, p, list_idx[t], tau[t], cnt[t], (tile + 1) * TILE_ROWS > p.n);
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

#ifndef BF_WAVES8
#define BF_WAVES8 1  // 1: dimensions up to 384 run eight waves of 32 queries per workgroup; 0: four waves of 64
#endif
uint32_t flat_scan_bf16_lists_per_slice() { return 4; }  // one per lane quarter
constexpr int BF_NTQ = (BF_WAVES8 || BF_FLAT_WG4X2) ? 2 : 4, BF_NW = (BF_WAVES8 && !BF_FLAT_WG4X2) ? 8 : 4;

// dpad (dims padded to a supported k-step count) and queries per workgroup for a dimension; 0 = unsupported
uint32_t flat_scan_bf16_dpad(uint32_t dim) {
    const uint32_t ks = (dim + 15) / 16;
    if (ks <= 8) return 128;
    if (ks <= 16) return 256;
    if (ks <= 24) return 384;
    if (ks <= 32) return 512;
    if (ks <= 48) return 768;
    if (ks <= 64) return 1024;
    if (ks <= 96) return 1536;
    return 0;
}
// (beyond 1024 dimensions a wave keeps one query tile: its stationary fragments alone fill 192 registers)
uint32_t flat_scan_bf16_queries_per_block(uint32_t dpad) { return dpad <= 384 ? 16u * BF_NTQ * BF_NW : (dpad <= 1024 ? 128u : 64u); }
#ifndef BF_IVF_WAVES
#define BF_IVF_WAVES 4  // waves per workgroup of the inverted-list scan up to 384 dimensions: 128 query slots per work item, two
                        // items per CU (measured 2.5 % faster than 8 waves / 256 slots at nlist 4096, nprobe 32, 10K queries)
#endif
// query slots of an inverted-list work item
uint32_t flat_scan_bf16_ivf_queries_per_block(uint32_t dpad) { return dpad <= 384 ? 32u * BF_IVF_WAVES : flat_scan_bf16_queries_per_block(dpad); }

template <int MODE, bool CINIT>
static hipError_t launch_bf16_mode(const ScanParams& p, hipStream_t stream) {
    const dim3 grid(p.n_splits * p.q_tiles);
    dim3 block(256);
    if (p.dpad <= 384) block = dim3(64 * BF_NW);
    switch (p.dpad) {
    case 128: hipLaunchKernelGGL((flat_scan_bf16x_kernel<MODE, BF_NTQ, 4, CINIT, BF_NW>), grid, block, 0, stream, p); break;
    case 256: hipLaunchKernelGGL((flat_scan_bf16x_kernel<MODE, BF_NTQ, 8, CINIT, BF_NW>), grid, block, 0, stream, p); break;
    case 384: hipLaunchKernelGGL((flat_scan_bf16x_kernel<MODE, BF_NTQ, 12, CINIT, BF_NW>), grid, block, 0, stream, p); break;
    case 512: hipLaunchKernelGGL((flat_scan_bf16x_kernel<MODE, 2, 16, CINIT, 4>), grid, block, 0, stream, p); break;
    case 768: hipLaunchKernelGGL((flat_scan_bf16x_kernel<MODE, 2, 24, CINIT, 4>), grid, block, 0, stream, p); break;
    case 1024: hipLaunchKernelGGL((flat_scan_bf16x_kernel<MODE, 2, 32, CINIT, 4>), grid, block, 0, stream, p); break;
    case 1536: hipLaunchKernelGGL((flat_scan_bf16x_kernel<MODE, 1, 48, CINIT, 4>), grid, block, 0, stream, p); break;
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
    if (p.dpad <= 384) block = dim3(64 * BF_IVF_WAVES);
    switch (p.dpad) {
    case 128: hipLaunchKernelGGL((flat_scan_bf16x_kernel<MODE_IVF, 2, 4, true, BF_IVF_WAVES>), grid, block, 0, stream, p); break;
    case 256: hipLaunchKernelGGL((flat_scan_bf16x_kernel<MODE_IVF, 2, 8, true, BF_IVF_WAVES>), grid, block, 0, stream, p); break;
    case 384: hipLaunchKernelGGL((flat_scan_bf16x_kernel<MODE_IVF, 2, 12, true, BF_IVF_WAVES>), grid, block, 0, stream, p); break;
    case 512: hipLaunchKernelGGL((flat_scan_bf16x_kernel<MODE_IVF, 2, 16, true, 4>), grid, block, 0, stream, p); break;
    case 768: hipLaunchKernelGGL((flat_scan_bf16x_kernel<MODE_IVF, 2, 24, true, 4>), grid, block, 0, stream, p); break;
    case 1024: hipLaunchKernelGGL((flat_scan_bf16x_kernel<MODE_IVF, 2, 32, true, 4>), grid, block, 0, stream, p); break;
    case 1536: hipLaunchKernelGGL((flat_scan_bf16x_kernel<MODE_IVF, 1, 48, true, 4>), grid, block, 0, stream, p); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

template <bool CINIT>
static hipError_t launch_bf16_dense(const ScanParams& p, hipStream_t stream) {
    const dim3 grid(p.n_splits * p.q_tiles);
    dim3 block(256);
    if (p.dpad <= 384) block = dim3(64 * BF_NW);
    switch (p.dpad) {
    case 128: hipLaunchKernelGGL((flat_scan_bf16x_kernel<MODE_DENSE, BF_NTQ, 4, CINIT, BF_NW>), grid, block, 0, stream, p); break;
    case 256: hipLaunchKernelGGL((flat_scan_bf16x_kernel<MODE_DENSE, BF_NTQ, 8, CINIT, BF_NW>), grid, block, 0, stream, p); break;
    case 384: hipLaunchKernelGGL((flat_scan_bf16x_kernel<MODE_DENSE, BF_NTQ, 12, CINIT, BF_NW>), grid, block, 0, stream, p); break;
    case 512: hipLaunchKernelGGL((flat_scan_bf16x_kernel<MODE_DENSE, 2, 16, CINIT, 4>), grid, block, 0, stream, p); break;
    case 768: hipLaunchKernelGGL((flat_scan_bf16x_kernel<MODE_DENSE, 2, 24, CINIT, 4>), grid, block, 0, stream, p); break;
    case 1024: hipLaunchKernelGGL((flat_scan_bf16x_kernel<MODE_DENSE, 2, 32, CINIT, 4>), grid, block, 0, stream, p); break;
    case 1536: hipLaunchKernelGGL((flat_scan_bf16x_kernel<MODE_DENSE, 1, 48, CINIT, 4>), grid, block, 0, stream, p); break;
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
    if (!prepass && (p0.valid || p0.filter)) return hipErrorInvalidValue;  // the listing pass takes row masks as accumulator seeds
    ScanParams p = p0;
    if (p.tile_len == 0) { p.tile_off = 0; p.tile_len = p.tiles_per_split; }  // unset: the whole slice
    if (p.cinit)
        return prepass ? launch_bf16_mode<MODE_PRE, true>(p, stream) : launch_bf16_mode<MODE_FLAT, true>(p, stream);
    return prepass ? launch_bf16_mode<MODE_PRE, false>(p, stream) : launch_bf16_mode<MODE_FLAT, false>(p, stream);
}

}  // namespace nh
