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
//   * the tables of a group of 8 queries are interleaved: one 16-byte ds_read_b128 fetches the
//     entries of 4 queries for one code;
//   * the gather is free of bank conflicts BY CONSTRUCTION.  A resident slice (16 subspaces, 128 KB)
//     is laid out [code][half][subspace][4 queries]: the 16-byte slot of the 64 banks an entry lives
//     in is its SUBSPACE, whatever the code.  The 16 lanes a ds_read_b128 serves per LDS cycle
//     ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ...) have 16 different values of lane & 15, and lane
//     l runs l & 15 steps behind lane 0: at step t of a slice it looks up subspace (t - l) & 15, so
//     the lanes of a group always read 16 different slots.  With codes as the slot (random codes)
//     the same reads took about three LDS cycles per group;
//   * the skew costs each step two accumulation targets: lanes that have already wrapped add into
//     row slot j, the others still into row slot j - 1.  Both are packed adds (v_pk_add_f32, two
//     queries each) under constant EXEC masks; 143 steps carry the 128 lookups of 8 rows;
//   * the codes are stored a second time subspace-group-major ([group][row][16 bytes]) with the
//     bytes of row r rotated by r & 15 places, so that byte t of the 16-byte word a lane loads is the
//     code it needs at step t;
//   * a workgroup keeps a slice resident and swaps it while its lanes hold the partial sums of
//     8 rows x 8 queries in registers, so the code matrix streams once per 8 queries and a slice is
//     re-read once per block of rows (the tables themselves are computed once per batch by
//     pq_tables_kernel, already in the LDS image's order, and stay in L2 / Infinity Cache);
//   * sums run in the reference's subspace order (slice after slice, +0.0 first; subspaces a last
//     slice does not have hold +0.0, which leaves a sum of squares unchanged), the square root is
//     taken only for rows whose sum is not above the list's threshold squared.
// Each wave keeps a candidate list per query (scores = -distance) with the flat scan's compaction.
#include "nh_common.h"
#include "nh_kernels.h"
#include "nh_select.h"

#pragma clang fp contract(off)

namespace nh {

#ifndef PQ_RPL_N
#define PQ_RPL_N 6
#endif
constexpr int PQ_RPL = PQ_RPL_N;                           // rows per lane and block
constexpr uint32_t PQ_THREADS = 64 * PQ_WAVES;
constexpr uint32_t PQ_BLOCK_ROWS = PQ_THREADS * PQ_RPL;    // rows a workgroup carries through the slices at a time
constexpr uint32_t PQ_IMG_F4 = 256 * 2 * PQ_SG;            // float4 entries of a resident slice: [code][half][subspace]
constexpr int PQ_IMG_PIECES = (int)(PQ_IMG_F4 * sizeof(float4) / 1024);              // 1 KB wave-loads of an image
constexpr int PQ_FILL_PIECES = (PQ_IMG_PIECES + PQ_WAVES - 1) / PQ_WAVES;             // per wave (wave w: pieces w, w + PQ_WAVES, ...)
static_assert(PQ_RPL % 2 == 0, "codes are fetched half of the row slots at a time");
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;
constexpr uint32_t PQ_XCDS = 8;                            // consecutive workgroup ids land on consecutive XCDs
constexpr int PQ_STEPS = (PQ_RPL + 1) * PQ_SG - 1;         // of a slice: 16 per row slot + 15 for the lanes behind to finish
#ifndef PQ_AHEAD_N
#define PQ_AHEAD_N 2
#endif
constexpr int PQ_AHEAD = PQ_AHEAD_N;                       // table lookups in flight per lane, in steps
static_assert(PQ_SG == 16 && PQ_QB == 8, "the lane skew and the packed adds are written for 16 subspaces x 8 queries");

uint32_t pq_scan_block_rows() { return PQ_BLOCK_ROWS; }
size_t pq_table_bytes(uint32_t q, uint32_t m) {
    return (size_t)((q + PQ_QB - 1) / PQ_QB) * ((m + PQ_SG - 1) / PQ_SG) * PQ_IMG_F4 * sizeof(float4);
}

// compute_distance_table, quantization.rs:468-494: per entry a sum over the subspace folded from
// -0.0 in dimension order.  Stored as the LDS images of the scan, one per (group of 8 queries, slice of 16 subspaces):
// tables[((((G * n_sg + sg) * 256 + c) * 2 + h) * 16 + j) * 4 + l] belongs to query 8 G + 4 h + l, subspace 16 sg + j,
// code c; subspaces beyond m hold +0.0.
__global__ __launch_bounds__(256) void pq_tables_kernel(const float* __restrict__ queries, const float* __restrict__ codebooks,
                                                        uint32_t q, uint32_t m, uint32_t n_sg, uint32_t dsub, uint32_t dim,
                                                        float* __restrict__ tables, uint64_t total) {
    const uint64_t idx = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const uint32_t l = (uint32_t)idx & 3u, j = (uint32_t)(idx >> 2) & 15u, h = (uint32_t)(idx >> 6) & 1u, c = (uint32_t)(idx >> 7) & 255u;
    const uint64_t gsg = idx >> 15;
    const uint32_t s = (uint32_t)(gsg % n_sg) * PQ_SG + j, query = (uint32_t)(gsg / n_sg) * PQ_QB + 4 * h + l;
    float sum = 0.0f;
    if (query < q && s < m) {
        sum = -0.0f;
        const float* qs = queries + (uint64_t)query * dim + s * dsub;
        const float* cb = codebooks + ((uint64_t)s * 256 + c) * dsub;
        for (uint32_t jj = 0; jj < dsub; ++jj) {
            const float d = qs[jj] - cb[jj];
            const float sq = d * d;
            sum = sum + sq;
        }
    }
    tables[idx] = sum;
}

// codes [n][m] -> [ceil(m / 16)][n][16]; byte (j + row) & 15 of a row's word holds subspace 16 g + j (the rotation the
// scan's lane skew wants), missing subspaces of the last group zero
__global__ __launch_bounds__(256) void pq_group_codes_kernel(const uint8_t* __restrict__ codes, uint64_t n, uint32_t m, uint32_t n_sg,
                                                             uint8_t* __restrict__ out) {
    const uint64_t idx = (uint64_t)blockIdx.x * 256 + threadIdx.x;  // one output byte
    if (idx >= (uint64_t)n_sg * n * PQ_SG) return;
    const uint64_t gr = idx / PQ_SG, row = gr % n;
    const uint32_t j = ((uint32_t)idx - (uint32_t)row) & (PQ_SG - 1);
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

typedef float f2v __attribute__((ext_vector_type(2)));

// Lanes whose skew (lane & 15) is at most i: the low i + 1 bits of every 16-lane group.
__host__ __device__ constexpr uint64_t pq_skew_mask(int i) { return (uint64_t)((1u << (i + 1)) - 1u) * 0x0001000100010001ull; }

// distance_with_table, quantization.rs:497-503: dist += table[s][code_s], s ascending.  `a` / `b` are the entries of
// queries 0-3 / 4-7 for this lane's code; lanes in `m` add them to the sums X, the others to the sums Y (the row slot
// before).  EXEC is all ones on entry (the caller's loops are workgroup-uniform) and on exit.
__device__ inline void pq_add_split(f2v (&X)[4], f2v (&Y)[4], const float4& a, const float4& b, uint64_t m) {
    const f2v x0 = {a.x, a.y}, x1 = {a.z, a.w}, x2 = {b.x, b.y}, x3 = {b.z, b.w};
    asm volatile(
        "s_mov_b64 exec, %[m]\n\t"
        "v_pk_add_f32 %[a0], %[a0], %[x0]\n\t"
        "v_pk_add_f32 %[a1], %[a1], %[x1]\n\t"
        "v_pk_add_f32 %[a2], %[a2], %[x2]\n\t"
        "v_pk_add_f32 %[a3], %[a3], %[x3]\n\t"
        "s_mov_b64 exec, %[nm]\n\t"
        "v_pk_add_f32 %[b0], %[b0], %[x0]\n\t"
        "v_pk_add_f32 %[b1], %[b1], %[x1]\n\t"
        "v_pk_add_f32 %[b2], %[b2], %[x2]\n\t"
        "v_pk_add_f32 %[b3], %[b3], %[x3]\n\t"
        "s_mov_b64 exec, -1"
        : [a0] "+v"(X[0]), [a1] "+v"(X[1]), [a2] "+v"(X[2]), [a3] "+v"(X[3]), [b0] "+v"(Y[0]), [b1] "+v"(Y[1]), [b2] "+v"(Y[2]),
          [b3] "+v"(Y[3])
        : [x0] "v"(x0), [x1] "v"(x1), [x2] "v"(x2), [x3] "v"(x3), [m] "s"(m), [nm] "s"(~m));
}
// the same for steps with one target only (the first and the last window of a slice): lanes in `m` add, the others idle
__device__ inline void pq_add_masked(f2v (&X)[4], const float4& a, const float4& b, uint64_t m) {
    const f2v x0 = {a.x, a.y}, x1 = {a.z, a.w}, x2 = {b.x, b.y}, x3 = {b.z, b.w};
    asm volatile(
        "s_mov_b64 exec, %[m]\n\t"
        "v_pk_add_f32 %[a0], %[a0], %[x0]\n\t"
        "v_pk_add_f32 %[a1], %[a1], %[x1]\n\t"
        "v_pk_add_f32 %[a2], %[a2], %[x2]\n\t"
        "v_pk_add_f32 %[a3], %[a3], %[x3]\n\t"
        "s_mov_b64 exec, -1"
        : [a0] "+v"(X[0]), [a1] "+v"(X[1]), [a2] "+v"(X[2]), [a3] "+v"(X[3])
        : [x0] "v"(x0), [x1] "v"(x1), [x2] "v"(x2), [x3] "v"(x3), [m] "s"(m));
}
__device__ inline void pq_add_all(f2v (&X)[4], const float4& a, const float4& b) {
    X[0] = X[0] + f2v{a.x, a.y}; X[1] = X[1] + f2v{a.z, a.w}; X[2] = X[2] + f2v{b.x, b.y}; X[3] = X[3] + f2v{b.z, b.w};
}

// the adds of step u = 16 j + i of a slice: window j, lanes at or past their skew (skew <= i) add into row slot j, the
// others into row slot j - 1; the first window has no slot before it, the last one none after
template <int AHEAD>
__device__ inline void pq_consume(f2v (&acc)[PQ_RPL][4], const float4 (&va)[AHEAD], const float4 (&vb)[AHEAD], int u) {
    const int j = u / PQ_SG, i = u % PQ_SG, sl = u % AHEAD;
    if (j == 0) {
        if (i == PQ_SG - 1) pq_add_all(acc[0], va[sl], vb[sl]);
        else pq_add_masked(acc[0], va[sl], vb[sl], pq_skew_mask(i));
    } else if (j >= PQ_RPL) {
        pq_add_masked(acc[PQ_RPL - 1], va[sl], vb[sl], ~pq_skew_mask(i));
    } else {
        if (i == PQ_SG - 1) pq_add_all(acc[j], va[sl], vb[sl]);
        else pq_add_split(acc[j], acc[j - 1], va[sl], vb[sl], pq_skew_mask(i));
    }
}

__global__ __launch_bounds__(PQ_THREADS) void pq_scan_kernel(PqScanParams p) {
    extern __shared__ __attribute__((aligned(16))) float4 lut4[];  // [code][half][subspace of the slice] x 4 queries
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // Workgroup ids go round the 8 XCDs; the `xcd_slots` workgroups an XCD runs at a time (one per CU: the image fills a
    // CU's LDS) are the row splits of xcd_slots / n_splits query groups, so the few images they keep re-reading stay in
    // that XCD's L2.
    const uint32_t xcd = blockIdx.x % PQ_XCDS, w = blockIdx.x / PQ_XCDS, round = w / p.xcd_slots, slot = w % p.xcd_slots;
    const uint32_t split = slot % p.n_splits, grp = (round * (p.xcd_slots / p.n_splits) + slot / p.n_splits) * PQ_XCDS + xcd;
    if (grp * PQ_QB >= p.q) return;  // workgroup-uniform
    const uint32_t g0 = grp * PQ_QB;
    const uint32_t r0 = split * p.split_stride;  // a multiple of 16: row & 15 == lane & 15 for every row of this lane
    const uint32_t r1 = min(r0 + p.rows_per_split, p.n);
    const uint32_t lists_per_q = p.n_splits * PQ_WAVES;
    float tau[PQ_QB], bound[PQ_QB];  // wave-uniform: score of the k-th best entry after the last compaction, and its sum bound
    uint32_t cnt[PQ_QB];             // wave-uniform list lengths
#pragma unroll
    for (int g = 0; g < PQ_QB; ++g) {
        tau[g] = -INFINITY;
        if (p.tau_seed && g0 + g < p.q) {
            // rows that TIE the seed may still belong to the k nearest: the list takes scores strictly above the next
            // float below it (scores are -distance: negative, or -0.0)
            const uint32_t b = __float_as_uint(p.tau_seed[g0 + g]);
            if (b >= 0x80000000u && b < 0xff800000u) tau[g] = __uint_as_float(b + 1u);
        }
        bound[g] = pq_sum_bound(tau[g]);
        cnt[g] = 0;
    }
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    // pairs of padding queries (beyond the batch) are never looked at
    const uint64_t qmask = 0x0101010101010101ull * (uint64_t)((g0 + PQ_QB <= p.q) ? 0xffu : ((1u << (p.q - g0)) - 1u));
    const float4* tab = (const float4*)p.tables + (uint64_t)grp * p.n_sg * PQ_IMG_F4;
    const uint4* codes = (const uint4*)p.codes_gm;
    const char* lds = (const char*)lut4;

    // Per-lane constants of the skew: at step i of a window the lane looks up subspace (i - skew) & 15, whose entries sit
    // 16 * that many bytes into the 512 bytes of a code; bytes below `skew` of a window's code word come from the row
    // slot before.
    const uint32_t skew = tid & 15u;
    uint32_t soff[PQ_SG], prev_bytes[4];
#pragma unroll
    for (int i = 0; i < PQ_SG; ++i) soff[i] = (((uint32_t)i - skew) & 15u) << 4;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const uint32_t t = skew > 4u * w ? min(skew - 4u * w, 4u) : 0u;
        prev_bytes[w] = t >= 4u ? 0xffffffffu : ((1u << (8u * t)) - 1u);
    }

    // The codes of a slice are fetched while the slice before it is still being summed: row slots 0-3 once the fifth
    // window has left them behind, row slots 4-7 behind the last window.
    uint32_t rot[PQ_RPL][4];  // the slice's codes of the lane's rows, byte t = the code wanted at step t of the row's windows
    auto fetch_codes = [&](uint32_t blk, uint32_t sg, int r_first) {
        const uint4* cg = codes + (uint64_t)sg * p.n;
#pragma unroll
        for (int r = r_first; r < r_first + PQ_RPL / 2; ++r) {
            const uint4 w = cg[min(blk + (uint32_t)r * PQ_THREADS + tid, p.n - 1)];
            rot[r][0] = w.x; rot[r][1] = w.y; rot[r][2] = w.z; rot[r][3] = w.w;
        }
    };
    fetch_codes(r0, 0, 0);
    fetch_codes(r0, 0, PQ_RPL / 2);

    for (uint32_t blk = r0; blk < r1; blk += PQ_BLOCK_ROWS) {  // block-uniform trip count
        f2v acc[PQ_RPL][4];  // [row slot][pair of queries]
#pragma unroll
        for (int r = 0; r < PQ_RPL; ++r)
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[r][g] = f2v{0.0f, 0.0f};  // distance_with_table starts at +0.0
        for (uint32_t sg = 0; sg < p.n_sg; ++sg) {
            const uint32_t next_sg = sg + 1 < p.n_sg ? sg + 1 : 0, next_blk = sg + 1 < p.n_sg ? blk : blk + PQ_BLOCK_ROWS;
            __syncthreads();  // the previous slice is no longer read
            // the image goes from memory straight into LDS, 1 KB per wave and instruction, nothing staged in registers
            if (!(p.dev & 256u) || (blk == r0 && sg == 0)) {
                const unsigned char* src = (const unsigned char*)(tab + (uint64_t)sg * PQ_IMG_F4) + wave * 1024u + lane * 16u;
                unsigned char* dst = (unsigned char*)lut4 + wave * 1024u;
#pragma unroll
                for (int i = 0; i < PQ_FILL_PIECES; ++i)
                    if (PQ_IMG_PIECES % PQ_WAVES == 0 || wave + (uint32_t)i * PQ_WAVES < (uint32_t)PQ_IMG_PIECES)  // wave-uniform
                        __builtin_amdgcn_global_load_lds((glb_void*)(src + i * (PQ_WAVES * 1024)), (lds_void*)(dst + i * (PQ_WAVES * 1024)), 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (p.dev & 512u) {
#pragma unroll
                for (int r = 0; r < PQ_RPL; ++r) acc[r][0].x += __uint_as_float(rot[r][0] ^ rot[r][1] ^ rot[r][2] ^ rot[r][3]);
                fetch_codes(next_blk, next_sg, 0);
                fetch_codes(next_blk, next_sg, PQ_RPL / 2);
                continue;
            }
            // Step t = 16 j + i: window j, lanes at or past their skew (skew <= i) work on row slot j, the others still on
            // row slot j - 1.  The lookups run PQ_AHEAD steps ahead of the adds that consume them (the asm blocks keep
            // program order, so the order written here is the order issued).
            float4 va[PQ_AHEAD], vb[PQ_AHEAD];
            uint32_t cw[4] = {rot[0][0], rot[0][1], rot[0][2], rot[0][3]};
#pragma unroll
            for (int jj = 0; jj <= PQ_RPL + 1; ++jj) {
#pragma unroll
                for (int ii = 0; ii < PQ_SG; ++ii) {
                    const int t = jj * PQ_SG + ii;
                    if (t >= PQ_AHEAD && t < PQ_STEPS + PQ_AHEAD) pq_consume<PQ_AHEAD>(acc, va, vb, t - PQ_AHEAD);
                    if (t == (PQ_RPL / 2 + 1) * PQ_SG + 1) fetch_codes(next_blk, next_sg, 0);
                    if (t < PQ_STEPS) {
                        const int j = jj, i = ii, sl = t % PQ_AHEAD;
                        if (i == 0 && j >= 1) {
#pragma unroll
                            for (int w = 0; w < 4; ++w) {
                                const uint32_t o = rot[j - 1][w], c = rot[j < PQ_RPL ? j : PQ_RPL - 1][w];
                                cw[w] = j == PQ_RPL ? o : ((o & prev_bytes[w]) | (c & ~prev_bytes[w]));
                            }
                        }
                        const uint32_t code = (cw[i >> 2] >> (8 * (i & 3))) & 255u;
                        const char* e = lds + ((code << 9) | soff[i]);
                        va[sl] = *(const float4*)e;
                        vb[sl] = *(const float4*)(e + 256);
                    }
                }
            }
            fetch_codes(next_blk, next_sg, PQ_RPL / 2);
        }
        float dist[PQ_RPL][PQ_QB];
#pragma unroll
        for (int r = 0; r < PQ_RPL; ++r)
#pragma unroll
            for (int g = 0; g < PQ_QB; ++g) dist[r][g] = acc[r][g >> 1][g & 1];
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
        static_assert(PQ_RPL * PQ_QB <= 64, "one bit per (row slot, query)");
        uint64_t todo = 0;  // wave-uniform, bit 8 r + g
#pragma unroll
        for (int r = 0; r < PQ_RPL; ++r)
#pragma unroll
            for (int g = 0; g < PQ_QB; ++g) {
                const bool close_by = ((liveb >> r) & 1u) && dist[r][g] <= bound[g];
                todo |= __builtin_amdgcn_ballot_w64(close_by) != 0 ? (1ull << (8 * r + g)) : 0ull;
            }
        todo &= qmask;
        if (p.dev & 1u) todo = 0;
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
    const uint32_t n_sg = (m + PQ_SG - 1) / PQ_SG;
    const uint64_t total = pq_table_bytes(q, m) / sizeof(float);
    if (total == 0) return hipSuccess;
    hipLaunchKernelGGL(pq_tables_kernel, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, stream, d_queries, d_codebooks, q, m, n_sg,
                       dsub, dim, d_tables, total);
    return hipGetLastError();
}

hipError_t launch_pq_scan(const PqScanParams& p, hipStream_t stream) {
    if (p.q == 0 || p.n == 0 || p.m == 0) return hipSuccess;
    if (p.n_sg != (p.m + PQ_SG - 1) / PQ_SG || !p.tables || !p.codes_gm) return hipErrorInvalidValue;
    if (p.split_stride % PQ_SG != 0) return hipErrorInvalidValue;  // a lane's rows must all have row & 15 == lane & 15
    const size_t lds = (size_t)PQ_IMG_F4 * sizeof(float4);
    hipError_t e = hipFuncSetAttribute((const void*)pq_scan_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    if (p.n_splits == 0 || p.xcd_slots % p.n_splits != 0) return hipErrorInvalidValue;
    const uint32_t groups = (p.q + PQ_QB - 1) / PQ_QB, per_round = PQ_XCDS * (p.xcd_slots / p.n_splits);
    const uint32_t rounds = (groups + per_round - 1) / per_round;
    hipLaunchKernelGGL(pq_scan_kernel, dim3(rounds * PQ_XCDS * p.xcd_slots), dim3(PQ_THREADS), lds, stream, p);
    return hipGetLastError();
}

}  // namespace nh
