// Workspace layout of the batch_top_k operator (ops.hip).
#pragma once
#include "nh_common.h"

namespace nh {

constexpr uint32_t TOPK_MAX_BLOCKS = 1024;  // workgroups per score row
constexpr uint32_t TOPK_STATE_WORDS = 4;    // per row: threshold, ranks left / ties to take, k_eff, rows taken below the threshold

struct TopkWork {
    uint64_t n;        // scores per row
    uint64_t chunk;    // scores per workgroup (multiple of 256)
    uint32_t rows, k, nblk;
    size_t off_sel, off_hist, off_state, off_tie;  // byte offsets of the four parts inside the workspace
    uint64_t* sel;     // [rows][k] selected keys (order image of the score << 32 | index)
    uint32_t* hist;    // [rows][256]
    uint32_t* state;   // [rows][TOPK_STATE_WORDS]
    uint32_t* tie;     // [rows][nblk] rows at the threshold per workgroup, then their exclusive prefix sums
};

// Fills the geometry and offsets of `w` for (rows, n, k); returns the workspace bytes needed.
size_t topk_work_bytes(uint32_t rows, uint64_t n, uint32_t k, TopkWork* w);
// The whole operator on `stream`, with the workspace at d_work (at least topk_work_bytes, 16-byte aligned).
hipError_t launch_batch_top_k(TopkWork w, void* d_work, const float* d_scores, uint64_t* d_out_idx, float* d_out_scores,
                              uint32_t* d_out_counts, hipStream_t stream);
needle_hip_status topk_check_args(needle_hip_ctx* c, const float* d_scores, uint32_t rows, uint64_t n, uint32_t k,
                                  const uint64_t* d_out_idx, const float* d_out_scores, const uint32_t* d_out_counts);

}  // namespace nh
