// Row-sharded search across the GPUs of one node, one process per GPU: every rank searches the same query batch on
// its own block of rows, the per-rank top-k lists travel in ONE RCCL all-gather over xGMI, and every rank merges them
// by (distance, global id).  Declarations and reference citations: include/needle_hip.h.
//
// The local search runs on one of the index's search slots (its own stream) and returns when its results are in this
// rank's slot of the gather buffer; id globalisation, all-gather and merge are then enqueued on the context's stream.
// A rank whose local search fails still takes part in the collective (its slot carries a failure mark), so that no
// other rank waits for it for ever; every rank then reports the failure.
#include "nh_api.h"

#include <rccl/rccl.h>

using namespace nh;

static_assert(NEEDLE_HIP_UNIQUE_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");

struct needle_hip_comm {
    needle_hip_ctx* ctx = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    nh::DevBuf gather;  // [world] slots of (ids | dists | counts)
    nh::DevBuf failed;  // one word: bit min(rank, 31) set when that rank's slot carries the failure mark
    std::mutex mu;
};

#define RCCL_TRY(expr)                                                                                         \
    do {                                                                                                       \
        ncclResult_t _r = (expr);                                                                              \
        if (_r != ncclSuccess)                                                                                 \
            return nh::fail(NEEDLE_HIP_ERR_RCCL, "RCCL error %d (%s) at %s:%d: %s", (int)_r, ncclGetErrorString(_r), \
                            __FILE__, __LINE__, #expr);                                                        \
    } while (0)

namespace nh {

// bytes of one rank's slot: ids [q][k] u64 | dists [q][k] f32 | counts [q] u32, rounded up to 16
__host__ __device__ inline size_t shard_slot_bytes(uint32_t q, uint32_t k) {
    return (((size_t)q * k * 12 + (size_t)q * 4) + 15) & ~(size_t)15;
}

// local row ids -> global ids (padding entries stay UINT64_MAX)
__global__ __launch_bounds__(256) void shard_globalize_kernel(uint64_t* __restrict__ ids, uint64_t total, uint64_t row_offset) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < total && ids[i] != ~0ull) ids[i] += row_offset;
}

constexpr uint32_t SHARD_SLOT_FAILED = 0xffffffffu;  // counts[0] of a slot whose rank could not search

// TPQ threads per query (one wave while lists * per <= 64, a whole workgroup beyond): the `lists` runs of <= per hits
// (one run per rank, each ascending by (distance, id)) are read from the packed slots, ranked by (distance, id, rank)
// and the k best emitted; repeated ids keep their best copy (merge_shard_results, src/persistence/shard.rs:622-643).
// A slot whose first count is SHARD_SLOT_FAILED contributes nothing and is reported through `failed`.
template <uint32_t TPQ>
__global__ __launch_bounds__(256) void shard_merge_kernel(const unsigned char* __restrict__ slots, size_t slot_bytes, uint32_t lists,
                                                          uint32_t nq, uint32_t per, uint32_t k, uint64_t* __restrict__ out_ids,
                                                          float* __restrict__ out_dists, uint32_t* __restrict__ out_counts,
                                                          uint32_t* __restrict__ failed) {
    extern __shared__ __attribute__((aligned(16))) unsigned char mraw[];
    constexpr uint32_t QPB = 256 / TPQ;  // queries per workgroup
    const uint32_t cap = lists * per;
    const uint32_t sub = threadIdx.x / TPQ, tid = threadIdx.x % TPQ;
    const uint32_t q = blockIdx.x * QPB + sub;
    const bool q_ok = q < nq;  // (no early exit: every thread meets the barriers)
    unsigned char* mine = mraw + (size_t)sub * cap * 20;
    uint64_t* sid = (uint64_t*)mine;
    uint32_t* sord = (uint32_t*)(sid + cap);
    uint32_t* sdead = sord + cap;  // 1: empty slot or a worse copy of a repeated id
    uint32_t* sdup = sdead + cap;
    __shared__ uint32_t s_kept[QPB];
    if (tid == 0) s_kept[sub] = 0;
    if (failed && blockIdx.x == 0 && threadIdx.x < lists) {
        const uint32_t* counts = (const uint32_t*)(slots + (size_t)threadIdx.x * slot_bytes + (size_t)nq * per * 12);
        if (counts[0] == SHARD_SLOT_FAILED) atomicOr(failed, 1u << min(threadIdx.x, 31u));
    }
    for (uint32_t t = tid; t < cap; t += TPQ) {
        const uint32_t l = t / per, j = t % per;
        const unsigned char* slot = slots + (size_t)l * slot_bytes;
        const uint64_t* ids = (const uint64_t*)slot;
        const float* dists = (const float*)(slot + (size_t)nq * per * 8);
        const uint32_t* counts = (const uint32_t*)(slot + (size_t)nq * per * 12);
        const bool live = q_ok && counts[0] != SHARD_SLOT_FAILED && j < min(counts[q], per);
        sid[t] = live ? ids[(size_t)q * per + j] : ~0ull;
        sord[t] = live ? f2ord(canon_distance(dists[(size_t)q * per + j])) : 0xffffffffu;
        sdead[t] = live ? 0u : 1u;
    }
    __syncthreads();
    auto before = [&](uint32_t i, uint32_t j) {  // entry i precedes entry j: (distance, id, position)
        if (sord[i] != sord[j]) return sord[i] < sord[j];
        if (sid[i] != sid[j]) return sid[i] < sid[j];
        return i < j;
    };
    // (both passes below read four entries per step, so that their LDS reads are in flight together: a loop that reads
    // one entry per iteration waits out an LDS latency per entry)
    for (uint32_t j = tid; j < cap; j += TPQ) {
        uint32_t dup = 0;
        if (!sdead[j]) {
            uint32_t i = 0;
            for (; i + 4 <= cap; i += 4) {
#pragma unroll
                for (uint32_t u = 0; u < 4; ++u) dup |= (!sdead[i + u] && i + u != j && sid[i + u] == sid[j] && before(i + u, j)) ? 1u : 0u;
            }
            for (; i < cap; ++i) dup |= (!sdead[i] && i != j && sid[i] == sid[j] && before(i, j)) ? 1u : 0u;
        }
        sdup[j] = dup;
    }
    __syncthreads();
    for (uint32_t j = tid; j < cap; j += TPQ)
        if (sdup[j]) sdead[j] = 1u;
    __syncthreads();
    for (uint32_t j = tid; j < cap; j += TPQ) {
        if (sdead[j]) continue;
        uint32_t rank = 0, i = 0;
        for (; i + 4 <= cap; i += 4) {
#pragma unroll
            for (uint32_t u = 0; u < 4; ++u) rank += (!sdead[i + u] && i + u != j && before(i + u, j)) ? 1u : 0u;
        }
        for (; i < cap; ++i) rank += (!sdead[i] && i != j && before(i, j)) ? 1u : 0u;
        atomicAdd(&s_kept[sub], 1u);
        if (rank < k) {
            out_ids[(size_t)q * k + rank] = sid[j];
            out_dists[(size_t)q * k + rank] = ord2f(sord[j]);
        }
    }
    __syncthreads();
    if (!q_ok) return;
    const uint32_t kept = min(s_kept[sub], k);
    for (uint32_t r = kept + tid; r < k; r += TPQ) {
        out_ids[(size_t)q * k + r] = ~0ull;
        out_dists[(size_t)q * k + r] = INFINITY;
    }
    if (tid == 0) out_counts[q] = kept;
}

// one wave per query while its entries fit one pass of a wave, a workgroup per query beyond
static hipError_t launch_shard_merge(const unsigned char* slots, size_t slot_bytes, uint32_t lists, uint32_t q, uint32_t k,
                                     uint64_t* out_ids, float* out_dists, uint32_t* out_counts, uint32_t* failed, hipStream_t stream) {
    const uint32_t cap = lists * k;
    if (cap <= 128) {
        hipLaunchKernelGGL(shard_merge_kernel<64>, dim3((q + 3) / 4), dim3(256), (size_t)4 * cap * 20, stream, slots, slot_bytes, lists, q,
                           k, k, out_ids, out_dists, out_counts, failed);
    } else {
        hipLaunchKernelGGL(shard_merge_kernel<256>, dim3(q), dim3(256), (size_t)cap * 20, stream, slots, slot_bytes, lists, q, k, k,
                           out_ids, out_dists, out_counts, failed);
    }
    return hipGetLastError();
}

constexpr uint32_t SHARD_MERGE_MAX_ENTRIES = 2048;  // lists * per of one query (40 KB of LDS)

}  // namespace nh

extern "C" needle_hip_status needle_hip_comm_unique_id(uint8_t* out_id) {
    if (!out_id) return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "out_id is null");
    ncclUniqueId id;
    RCCL_TRY(ncclGetUniqueId(&id));
    memcpy(out_id, id.internal, NCCL_UNIQUE_ID_BYTES);
    return NEEDLE_HIP_OK;
}

extern "C" needle_hip_status needle_hip_comm_init(needle_hip_ctx* c, const uint8_t* unique_id, int rank, int world,
                                                  needle_hip_comm** out) {
    if (!c || !unique_id || !out) return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world)
        return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "rank %d out of range for %d ranks", rank, world);
    HIP_TRY(hipSetDevice(c->device));
    needle_hip_comm* m = new (std::nothrow) needle_hip_comm();
    if (!m) return fail(NEEDLE_HIP_ERR_OUT_OF_MEMORY, "host allocation failed");
    m->ctx = c;
    m->rank = rank;
    m->world = world;
    ncclUniqueId id;
    memcpy(id.internal, unique_id, NCCL_UNIQUE_ID_BYTES);
    const ncclResult_t r = ncclCommInitRank(&m->comm, world, id, rank);
    if (r != ncclSuccess) {
        delete m;
        return fail(NEEDLE_HIP_ERR_RCCL, "RCCL error %d (%s) in ncclCommInitRank(rank %d of %d)", (int)r, ncclGetErrorString(r), rank,
                    world);
    }
    *out = m;
    return NEEDLE_HIP_OK;
}

extern "C" void needle_hip_comm_destroy(needle_hip_comm* m) {
    if (!m) return;
    (void)hipSetDevice(m->ctx->device);
    (void)hipStreamSynchronize(m->ctx->stream);
    if (m->comm) (void)ncclCommDestroy(m->comm);
    m->gather.release();
    m->failed.release();
    delete m;
}

extern "C" int needle_hip_comm_rank(const needle_hip_comm* m) { return m ? m->rank : -1; }
extern "C" int needle_hip_comm_world(const needle_hip_comm* m) { return m ? m->world : 0; }

extern "C" needle_hip_status needle_hip_sharded_batch_search_device(needle_hip_index* ix, needle_hip_comm* m, uint64_t row_offset,
                                                                    const float* d_queries, uint32_t q,
                                                                    const needle_search_params* params, uint64_t* d_out_ids,
                                                                    float* d_out_dists, uint32_t* d_out_counts) {
    if (!ix || !m || !params) return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "null argument");
    if (ix->ctx != m->ctx) return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "the index and the communicator belong to different contexts");
    if (q == 0) return NEEDLE_HIP_OK;
    if (!d_queries || !d_out_counts || (params->k > 0 && (!d_out_ids || !d_out_dists)))
        return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "null query or output pointer");
    needle_hip_ctx* c = m->ctx;
    const uint32_t k = params->k;
    std::lock_guard g(m->mu);
    HIP_TRY(hipSetDevice(c->device));
    if (k == 0) {  // nothing to exchange
        HIP_TRY(launch_fill_empty(d_out_ids, d_out_dists, d_out_counts, q, 0, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        return NEEDLE_HIP_OK;
    }
    if ((uint64_t)m->world * k > SHARD_MERGE_MAX_ENTRIES)
        return fail(NEEDLE_HIP_ERR_UNSUPPORTED, "merge of %d shards x k = %u exceeds %u entries per query", m->world, k,
                    SHARD_MERGE_MAX_ENTRIES);
    const size_t slot = shard_slot_bytes(q, k);
    HIP_TRY(m->gather.ensure(slot * (size_t)m->world));
    unsigned char* mine = m->gather.as<unsigned char>() + slot * (size_t)m->rank;
    uint64_t* l_ids = (uint64_t*)mine;
    float* l_dists = (float*)(mine + (size_t)q * k * 8);
    uint32_t* l_counts = (uint32_t*)(mine + (size_t)q * k * 12);
    // local top-k of this rank's row block, written straight into its slot of the gather buffer
    const needle_hip_status local = needle_hip_batch_search_device(ix, d_queries, q, params, l_ids, l_dists, l_counts);
    std::string local_msg;
    const uint64_t total = (uint64_t)q * k;
    if (local != NEEDLE_HIP_OK) {
        // the other ranks are on their way into the collective: join it with a marked slot, report afterwards
        local_msg = needle_hip_last_error();
        HIP_TRY(hipMemsetAsync(l_counts, 0xff, sizeof(uint32_t), c->stream));
    } else if (row_offset != 0) {
        hipLaunchKernelGGL(shard_globalize_kernel, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, c->stream, l_ids, total,
                           row_offset);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(m->failed.ensure(sizeof(uint32_t)));
    HIP_TRY(hipMemsetAsync(m->failed.p, 0, sizeof(uint32_t), c->stream));
    // one collective: every rank's slot to every rank (in place: the send buffer is this rank's slot of the receive buffer)
    RCCL_TRY(ncclAllGather(mine, m->gather.p, slot, ncclChar, m->comm, c->stream));
    HIP_TRY(launch_shard_merge(m->gather.as<unsigned char>(), slot, (uint32_t)m->world, q, k, d_out_ids, d_out_dists, d_out_counts,
                               m->failed.as<uint32_t>(), c->stream));
    uint32_t failed_ranks = 0;
    HIP_TRY(hipMemcpyAsync(&failed_ranks, m->failed.p, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (local != NEEDLE_HIP_OK) return fail(local, "%s", local_msg.c_str());
    if (failed_ranks)
        return fail(NEEDLE_HIP_ERR_RCCL, "the local search of another rank failed (rank mask 0x%x); the merged result is incomplete",
                    failed_ranks);
    return NEEDLE_HIP_OK;
}

// The merge step alone, on slots some other transport filled (tests; hosts that gather over their own channel):
// d_slots holds `lists` slots of shard_slot_bytes(q, k) bytes each, laid out ids | dists | counts.
extern "C" needle_hip_status needle_hip_merge_shard_slots_device(needle_hip_ctx* c, const void* d_slots, uint32_t lists, uint32_t q,
                                                                 uint32_t k, uint64_t* d_out_ids, float* d_out_dists,
                                                                 uint32_t* d_out_counts) {
    if (!c) return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "ctx is null");
    if (q == 0) return NEEDLE_HIP_OK;
    if (!d_slots || !d_out_counts || lists == 0 || (k > 0 && (!d_out_ids || !d_out_dists)))
        return fail(NEEDLE_HIP_ERR_INVALID_ARGUMENT, "null argument");
    if ((uint64_t)lists * k > SHARD_MERGE_MAX_ENTRIES)
        return fail(NEEDLE_HIP_ERR_UNSUPPORTED, "merge of %u shards x k = %u exceeds %u entries per query", lists, k,
                    SHARD_MERGE_MAX_ENTRIES);
    std::lock_guard g(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    if (k == 0) {
        HIP_TRY(launch_fill_empty(d_out_ids, d_out_dists, d_out_counts, q, 0, c->stream));
    } else {
        HIP_TRY(launch_shard_merge((const unsigned char*)d_slots, shard_slot_bytes(q, k), lists, q, k, d_out_ids, d_out_dists,
                                   d_out_counts, nullptr, c->stream));
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    return NEEDLE_HIP_OK;
}

extern "C" uint64_t needle_hip_shard_slot_bytes(uint32_t q, uint32_t k) { return shard_slot_bytes(q, k); }
