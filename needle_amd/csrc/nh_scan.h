// Pieces shared by the first-pass scan kernels (fp32 and bf16 operands).
#pragma once
#include "nh_common.h"
#include "nh_select.h"

namespace nh {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

enum ScanMode { MODE_FLAT = 0, MODE_PRE = 1, MODE_IVF = 2, MODE_DENSE = 3 };  // DENSE: every score is written out

// accumulator register r of lane half h holds result row (r&3) + 8*(r>>2) + 4*h
__device__ inline uint32_t row_of_reg(int r, uint32_t h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// May row `row` be returned?  FLAT: row < n and mask bits of the row; IVF: the
// position maps to a real row whose mask bits (by original id) are set.
template <int MODE>
__device__ inline bool row_live(const ScanParams& p, uint32_t row) {
    if (MODE == MODE_IVF) {
        // padding positions of an inverted list carry an accumulator seed of -inf (set when the index is built), so
        // they never score above a threshold: without masks no lookup is needed
        if (!p.valid && !p.filter) return true;
        const uint32_t id = p.pos2id[row];
        if (id == 0xffffffffu) return false;
        if (p.valid && !bit_set(p.valid, id)) return false;
        if (p.filter && !bit_set(p.filter, id)) return false;
        return true;
    }
    if (row >= p.n) return false;
    if (p.valid && !bit_set(p.valid, row)) return false;
    if (p.filter && !bit_set(p.filter, row)) return false;
    return true;
}

}  // namespace nh
