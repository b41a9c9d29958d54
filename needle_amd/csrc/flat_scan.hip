// First pass of the exact searches: scores = rows x queries^T on the matrix
// cores, with the per-query candidate selection fused behind each tile.
//
// One workgroup = 4 waves = a block of up to 128 queries walking a run of row
// tiles (128 rows each).  The rows are the MFMA A operand (rows of the 32x32
// result land in the accumulator registers), the queries are the B operand (one
// query per lane), so each lane compares its 16 results per tile against ITS
// query's running threshold with no cross-lane traffic.  A result above the
// threshold is appended to that query's candidate list; a list that grows past
// 2*kprime is cut back to its kprime best entries by the owning wave
// (nh_select.h), which also raises the threshold.  Waves own disjoint queries,
// so the selection needs no LDS, no atomics and no workgroup barrier.
//
// Two work decompositions share the kernel:
//   FLAT: workgroup = (query block, slice of the base); queries q0..q0+127.
//   IVF : workgroup = one work item: the tiles of ONE inverted list x up to 128
//         of the queries probing it (gathered by index); rows are positions in
//         the list-ordered copy of the base.
//
// The scores are only a filter: the kept candidates are re-scored in the
// reference's scalar order afterwards (flat_post.hip).
#include "nh_common.h"
#include "nh_kernels.h"
#include "nh_scan.h"
#include "nh_select.h"

namespace nh {

// Append the passing results of one 32x32 tile (16 accumulator registers per
// lane) and compact the lists that filled up.  `acc` rows are rows
// row_base + row_of_reg(r, h); the lane's query owns candidate list `list_idx`.
template <int MODE>
__device__ inline void push_tile(const f32x16& acc, uint32_t row_base, uint32_t h, uint32_t l31,
                                 uint32_t lane, const ScanParams& p, uint32_t list_idx, float& tau,
                                 uint32_t& cnt) {
    float m = acc[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) m = fmaxf(m, acc[r]);
    if (__builtin_amdgcn_ballot_w64(m > tau) == 0) return;  // wave-uniform
    uint2* mylist = p.lists + (uint64_t)list_idx * p.cap;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float key = acc[r];
        const uint32_t row = row_base + row_of_reg(r, h);
        bool pass = key > tau;
        if (pass) pass = row_live<MODE>(p, row);
        const uint64_t pm = __builtin_amdgcn_ballot_w64(pass);
        if (pm != 0) {
            // lanes l31 and l31+32 serve the same query: the low half takes the
            // first slot, the high half the next one
            const uint32_t lo = (uint32_t)(pm >> l31) & 1u, hi = (uint32_t)(pm >> (l31 + 32)) & 1u;
            if (pass) mylist[cnt + (h ? lo : 0u)] = make_uint2(__float_as_uint(key), row);
            cnt += lo + hi;
        }
    }
    uint64_t need = __builtin_amdgcn_ballot_w64(cnt > p.cap - 32) & 0xffffffffull;
    while (need) {
        const uint32_t b = (uint32_t)__builtin_ctzll(need);
        need &= need - 1;
        const uint32_t n_e = (uint32_t)__builtin_amdgcn_readlane((int)cnt, (int)b);
        const uint32_t li = (uint32_t)__builtin_amdgcn_readlane((int)list_idx, (int)b);
        const float t = wave_compact_list(p.lists + (uint64_t)li * p.cap, n_e, p.kprime, lane);
        if (l31 == b) { tau = t; cnt = p.kprime; }
    }
}

// ---------------------------------------------------------------------------
// fp32 operands: v_mfma_f32_32x32x2_f32, exact f32 products and accumulation
// ---------------------------------------------------------------------------
constexpr int F32_BK = 32;  // k-step staged per LDS buffer
constexpr int F32_LD = 36;  // padded LDS row stride (floats): conflict-free ds_read_b128

// A_U8: the row operand is a u8 code matrix ([n_pad][dpad] bytes), widened to f32 while staging.
template <int MODE, bool A_U8 = false>
__global__ __launch_bounds__(256, 2) void flat_scan_f32_kernel(ScanParams p) {
    // [A buf0][A buf1][B buf0][B buf1], each 128 rows x 36 floats
    __shared__ __attribute__((aligned(16))) float smem[4 * TILE_ROWS * F32_LD];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t l31 = lane & 31, h = lane >> 5;
    const uint32_t jq = wave * 32 + l31;  // this lane's query slot within the block

    uint32_t t0, t1, split = 0, q0 = 0;
    uint32_t tile_first = 0, tile_mul = 1;  // FLAT / PRE: position jj of the block's run is tile tile_first + jj * tile_mul
    uint32_t list_idx = 0;       // candidate list of this lane's query
    uint32_t my_query = 0;       // row of query_op this lane's column multiplies
    bool q_live = true;          // false: padding slot of an IVF item
    uint32_t qrow[4];            // query_op rows this thread stages (r = (tid + 256 i) >> 3)
    if (MODE == MODE_IVF) {
        // the grid is an upper bound sized on the host; the items this launch really has are counted on the device
        uint32_t item_idx = blockIdx.x;
        if (p.item_bounds) {
            item_idx += p.item_bounds[0];
            if (item_idx >= p.item_bounds[1]) return;  // workgroup-uniform, before any barrier
        }
        const IvfItem it = p.items[item_idx];
        t0 = it.tile0;
        t1 = it.tile1;
        const uint2* iq = p.item_queries + it.qoff;
        q_live = jq < it.nq;
        if (q_live) {
            const uint2 e = iq[jq];
            my_query = e.x;
            list_idx = e.y + it.list_add;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t r = (tid + i * 256) >> 3;
            qrow[i] = r < it.nq ? iq[r].x : 0xffffffffu;
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
        q0 = q_tile * TILE_QUERIES;
        // slice `split` owns tiles split, split + n_splits, ...; the pre-pass samples every pre_stride-th of them
        const uint32_t own = split < p.n_tiles ? (p.n_tiles - split + p.n_splits - 1) / p.n_splits : 0u;
        const uint32_t jstep = MODE == MODE_PRE ? max(p.pre_stride, 1u) : 1u;
        t0 = 0;
        t1 = MODE == MODE_PRE ? min(p.pre_tiles, (own + jstep - 1) / jstep) : own;
        tile_first = split;
        tile_mul = jstep * p.n_splits;
        my_query = q0 + jq;
        list_idx = my_query * p.n_splits + split;
#pragma unroll
        for (int i = 0; i < 4; ++i) qrow[i] = q0 + ((tid + i * 256) >> 3);
    }

    float tau = -INFINITY;
    if (MODE != MODE_PRE && p.tau0 && q_live) tau = p.tau0[my_query];
    if (!q_live || (p.debug & 1u) || (MODE != MODE_IVF && my_query >= p.n_queries)) tau = INFINITY;
    uint32_t cnt = 0;
    float gmax[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};  // MODE_PRE: this lane's group maxima

    const uint32_t KC = p.dpad / F32_BK;
    const uint32_t total = (t1 > t0 ? (t1 - t0) : 0) * KC;
    const float* __restrict__ gA = (const float*)p.base_op;
    const uint8_t* __restrict__ gA8 = (const uint8_t*)p.base_op;
    const float* __restrict__ gB = (const float*)p.query_op;

    f32x4 ra[4], rb[4];
    auto load_chunk = [&](uint32_t c) {
        const uint32_t tile = tile_first + (t0 + c / KC) * tile_mul, kc = c % KC;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t idx = tid + i * 256, r = idx >> 3, c4 = idx & 7;
            if (A_U8) {
                const uint32_t w = *(const uint32_t*)(gA8 + (uint64_t)(tile * TILE_ROWS + r) * p.dpad + kc * F32_BK + c4 * 4);
                ra[i] = f32x4{(float)(w & 255u), (float)((w >> 8) & 255u), (float)((w >> 16) & 255u), (float)(w >> 24)};
            } else {
                ra[i] = *(const f32x4*)(gA + (uint64_t)(tile * TILE_ROWS + r) * p.dpad + kc * F32_BK + c4 * 4);
            }
            if (MODE == MODE_IVF && qrow[i] == 0xffffffffu) {
                rb[i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            } else {
                rb[i] = *(const f32x4*)(gB + (uint64_t)qrow[i] * p.dpad + kc * F32_BK + c4 * 4);
            }
        }
    };
    auto store_chunk = [&](uint32_t buf) {
        float* As = smem + buf * (TILE_ROWS * F32_LD);
        float* Bs = smem + (2 + buf) * (TILE_ROWS * F32_LD);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t idx = tid + i * 256, r = idx >> 3, c4 = idx & 7;
            *(f32x4*)(As + r * F32_LD + c4 * 4) = ra[i];
            *(f32x4*)(Bs + r * F32_LD + c4 * 4) = rb[i];
        }
    };

    f32x16 acc[4];
    if (total > 0) {
        load_chunk(0);
        store_chunk(0);
    }
    __syncthreads();
    for (uint32_t c = 0; c < total; ++c) {
        const uint32_t buf = c & 1, kc = c % KC, tile = tile_first + (t0 + c / KC) * tile_mul;
        if (c + 1 < total) load_chunk(c + 1);  // in flight behind this chunk's MFMAs
        if (kc == 0) {
            if (p.cinit) {
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        acc[mt][r] = p.cinit[tile * TILE_ROWS + mt * 32 + row_of_reg(r, h)];
            } else {
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mt][r] = 0.0f;
            }
        }
        const float* As = smem + buf * (TILE_ROWS * F32_LD);
        const float* Bs = smem + (2 + buf) * (TILE_ROWS * F32_LD);
        // lane half h takes k = 8g + 4h + j for its j-th MFMA of group g, on both
        // operands alike, so each operand fragment is one 16-byte LDS read
#pragma unroll
        for (int g = 0; g < F32_BK / 8; ++g) {
            const f32x4 b = *(const f32x4*)(Bs + (wave * 32 + l31) * F32_LD + g * 8 + h * 4);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const f32x4 a = *(const f32x4*)(As + (mt * 32 + l31) * F32_LD + g * 8 + h * 4);
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b[0], acc[mt], 0, 0, 0);
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], b[1], acc[mt], 0, 0, 0);
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], b[2], acc[mt], 0, 0, 0);
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], b[3], acc[mt], 0, 0, 0);
            }
        }
        if (c + 1 < total) store_chunk(buf ^ 1);
        __syncthreads();
        if (kc == KC - 1) {
            if (MODE == MODE_PRE) {
                const bool plain = !p.valid && !p.filter && (tile + 1) * TILE_ROWS <= p.n;
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        bool ok = true;
                        if (!plain) ok = row_live<MODE_FLAT>(p, tile * TILE_ROWS + mt * 32 + row_of_reg(r, h));
                        if (ok) gmax[mt] = fmaxf(gmax[mt], acc[mt][r]);
                    }
                }
            } else {
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
                    push_tile<MODE>(acc[mt], tile * TILE_ROWS + mt * 32, h, l31, lane, p, list_idx, tau, cnt);
            }
        }
    }
    if (MODE == MODE_PRE) {
        float* dst = p.pre_max + (uint64_t)my_query * (8 * p.n_splits) + (split * 2 + h) * 4;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) dst[mt] = gmax[mt];
        return;
    }

    uint64_t need = __builtin_amdgcn_ballot_w64(cnt > p.kprime) & 0xffffffffull;
    while (need) {
        const uint32_t b = (uint32_t)__builtin_ctzll(need);
        need &= need - 1;
        const uint32_t n_e = (uint32_t)__builtin_amdgcn_readlane((int)cnt, (int)b);
        const uint32_t li = (uint32_t)__builtin_amdgcn_readlane((int)list_idx, (int)b);
        wave_compact_list(p.lists + (uint64_t)li * p.cap, n_e, p.kprime, lane);
        if (l31 == b) cnt = p.kprime;
    }
    if (h == 0 && q_live) p.list_counts[list_idx] = cnt;
}

// One wave per query, between two launches that share candidate lists: with at least kseed candidates listed,
// the kseed-th largest listed score is a threshold that kseed live rows reach; the next launch starts from it.
__global__ __launch_bounds__(64) void tau_refresh_kernel(const uint2* __restrict__ lists, const uint32_t* __restrict__ counts,
                                                         uint32_t n_lists, uint32_t cap, uint32_t kprime, uint32_t kseed,
                                                         float* __restrict__ tau, TauBand band,
                                                         const uint32_t* __restrict__ list_base) {
    constexpr uint32_t MAXE = 1024;  // more listed than this: keep the threshold (the lists are long anyway)
    __shared__ uint32_t ord[MAXE];
    const uint32_t q = blockIdx.x, lane = threadIdx.x;
    const uint64_t first = list_base ? list_base[q] : (uint64_t)q * n_lists;
    if (list_base) n_lists = list_base[q + 1] - list_base[q];
    uint32_t total = 0;
    for (uint32_t l0 = 0; l0 < n_lists; l0 += 64) {
        const uint32_t li = l0 + lane;
        const uint32_t c = li < n_lists ? min(counts[first + li], kprime) : 0u;
        uint32_t incl = c;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t t = (uint32_t)__shfl_up((int)incl, off);
            if (lane >= (uint32_t)off) incl += t;
        }
        const uint32_t dst = total + incl - c;
        const uint2* l = lists + (first + li) * cap;
        for (uint32_t i = 0; i < c; ++i)
            if (dst + i < MAXE) ord[dst + i] = f2ord(__uint_as_float(l[i].x));
        total += (uint32_t)__shfl((int)incl, 63);
    }
    __syncthreads();
    if (total < kseed || total > MAXE) return;
    // the listed scores go to registers once (f2ord of a score is never 0, the filler); the bit passes then cost a
    // compare and a ballot count per 64 of them
    uint32_t r[MAXE / 64];
#pragma unroll
    for (uint32_t u = 0; u < MAXE / 64; ++u) r[u] = (u * 64 < total && u * 64 + lane < total) ? ord[u * 64 + lane] : 0u;
    const uint32_t prefix = wave_kth_largest_regs<MAXE / 64>(r, total, kseed);
    if (lane == 0) {
        // Stay below the kseed-th listed score by a few times the score uncertainty: the rows inside that band
        // are the ones the bound check needs to see listed before it can certify the k-th hit.
        const double eps = band_eps(band, q);
        const float t = (float)((double)ord2f(prefix) - (double)band.width * eps);
        if (t == t) tau[q] = fmaxf(tau[q], t);
    }
}

hipError_t launch_tau_refresh(const ScanParams& p, uint32_t n_lists, float* d_tau, const TauBand& band, hipStream_t stream,
                              const uint32_t* d_list_base) {
    if (p.n_queries == 0) return hipSuccess;
    hipLaunchKernelGGL(tau_refresh_kernel, dim3(p.n_queries), dim3(64), 0, stream, p.lists, p.list_counts, n_lists, p.cap,
                       p.kprime, p.kseed, d_tau, band, d_list_base);
    return hipGetLastError();
}

// One wave per query: tau0 = (kprime-th largest of the query's group maxima) - 2 ulps.
__global__ __launch_bounds__(64) void tau0_kernel(const float* __restrict__ pre_max, uint32_t groups,
                                                  uint32_t kprime, float* __restrict__ tau0, TauBand band, uint32_t n_queries) {
    const uint32_t q = blockIdx.x, lane = threadIdx.x;
    const float* v = pre_max + (uint64_t)q * groups;
    uint32_t prefix = 0;
    constexpr uint32_t T0_PER = 16;  // up to 1024 group maxima are read once and selected from registers
    if (groups >= kprime && groups <= 64 * T0_PER) {
        uint32_t r[T0_PER];
#pragma unroll
        for (uint32_t u = 0; u < T0_PER; ++u) r[u] = (u * 64 < groups && u * 64 + lane < groups) ? f2ord(v[u * 64 + lane]) : 0u;
        // f2ord(-inf) = 0x007fffff > 0: a group with no live row still counts as a value, below every score
        prefix = wave_kth_largest_regs<T0_PER>(r, groups, kprime);
    } else if (groups >= kprime) {
        for (int bit = 31; bit >= 0; --bit) {
            const uint32_t trial = prefix | (1u << bit);
            uint32_t cnt = 0;
            for (uint32_t i = 0; i < groups; i += 64) {
                const bool ge = (i + lane < groups) && (f2ord(v[i + lane]) >= trial);
                cnt += __builtin_popcountll(__builtin_amdgcn_ballot_w64(ge));
            }
            if (cnt >= kprime) prefix = trial;
        }
    }
    if (lane == 0) {
        float t = -INFINITY;
        // f2ord(-inf) = 0x007fffff; anything at or below it means "no threshold"
        if (groups >= kprime && prefix > 0x00800001u) {
            t = ord2f(prefix - 2);
            // as in the refresh between launches: leave the band of scores the bound check must see listed
            if (q < n_queries) {
                const float tb = (float)((double)t - (double)band.width * band_eps(band, q));
                if (tb == tb) t = fminf(t, tb);
            }
        }
        tau0[q] = t;
    }
}

hipError_t launch_flat_scan(int precision, const ScanParams& p, hipStream_t stream) {
    const uint32_t grid = p.n_splits * p.q_tiles;
    if (grid == 0) return hipSuccess;
    if (precision == NEEDLE_PRECISION_FP32) {
        if (p.base_is_u8) hipLaunchKernelGGL((flat_scan_f32_kernel<MODE_FLAT, true>), dim3(grid), dim3(256), 0, stream, p);
        else hipLaunchKernelGGL(flat_scan_f32_kernel<MODE_FLAT>, dim3(grid), dim3(256), 0, stream, p);
    } else {
        return launch_flat_scan_bf16(p, false, stream);
    }
    return hipGetLastError();
}

hipError_t launch_flat_prepass(int precision, const ScanParams& p, float* d_tau0, const TauBand& band, hipStream_t stream) {
    const uint32_t grid = p.n_splits * p.q_tiles;
    if (grid == 0) return hipSuccess;
    if (precision == NEEDLE_PRECISION_FP32) {
        if (p.base_is_u8) hipLaunchKernelGGL((flat_scan_f32_kernel<MODE_PRE, true>), dim3(grid), dim3(256), 0, stream, p);
        else hipLaunchKernelGGL(flat_scan_f32_kernel<MODE_PRE>, dim3(grid), dim3(256), 0, stream, p);
    } else {
        const hipError_t e = launch_flat_scan_bf16(p, true, stream);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(tau0_kernel, dim3(p.q_tiles * flat_scan_queries_per_block(precision, p.dpad)), dim3(64), 0, stream,
                       p.pre_max, 8 * p.n_splits, p.kseed, d_tau0, band, p.n_queries);
    return hipGetLastError();
}

hipError_t launch_ivf_scan(int precision, const ScanParams& p, uint32_t n_items, hipStream_t stream) {
    if (n_items == 0) return hipSuccess;
    if (precision == NEEDLE_PRECISION_BF16) return launch_ivf_scan_bf16(p, n_items, stream);
    if (precision != NEEDLE_PRECISION_FP32) return hipErrorInvalidValue;
    hipLaunchKernelGGL(flat_scan_f32_kernel<MODE_IVF>, dim3(n_items), dim3(256), 0, stream, p);
    return hipGetLastError();
}

int flat_scan_blocks_per_cu(int precision) {
    int nb = 0;
    hipError_t e = hipErrorInvalidValue;
    if (precision == NEEDLE_PRECISION_FP32)
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, flat_scan_f32_kernel<MODE_FLAT>, 256, 0);
    if (e != hipSuccess || nb < 1) nb = 1;  // bf16: one workgroup per CU by design (one wave per SIMD)
    return nb;
}

uint32_t flat_scan_queries_per_block(int precision, uint32_t dpad) {
    return precision == NEEDLE_PRECISION_FP32 ? (uint32_t)TILE_QUERIES : flat_scan_bf16_queries_per_block(dpad);
}

uint32_t flat_scan_kstep(int precision) { return precision == NEEDLE_PRECISION_FP32 ? F32_BK : 64; }

}  // namespace nh
