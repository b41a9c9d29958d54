// Index-build / query-upload kernels: norms and the operand copies the first
// pass multiplies.  None of these values reaches a reported distance: reported
// distances are recomputed from the raw rows in the reference's order.
#include "nh_common.h"
#include "nh_kernels.h"

namespace nh {

__device__ inline uint16_t bf16_rne(float v) {
    uint32_t u = __float_as_uint(v);
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);  // inputs are finite
}

// one wave per row, coalesced
__global__ __launch_bounds__(256) void row_prep_kernel(const float* __restrict__ rows, uint64_t n,
                                                       uint32_t dim, uint32_t dpad, int normalize,
                                                       float* __restrict__ op_f32,
                                                       uint16_t* __restrict__ op_bf16,
                                                       float* __restrict__ cinit,
                                                       float* __restrict__ norms,
                                                       double* __restrict__ norm2,
                                                       uint32_t* __restrict__ max_norm_bits,
                                                       float* __restrict__ bf16_err,
                                                       uint32_t* __restrict__ max_bf16_err_bits) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t row = (uint64_t)blockIdx.x * 4 + wave;
    if (row >= n) return;
    const float* src = rows + row * dim;
    double s = 0.0;
    for (uint32_t i = lane; i < dim; i += 64) { double x = (double)src[i]; s += x * x; }
    for (int off = 32; off; off >>= 1) s += __shfl_xor(s, off);
    const float norm = (float)sqrt(s);
    float inv = 1.0f;
    if (normalize) inv = (s > 0.0) ? (float)(1.0 / sqrt(s)) : 0.0f;
    double err = 0.0;  // |operand - bf16(operand)|^2
    for (uint32_t i = lane; i < dim; i += 64) {
        const float v = src[i] * inv;
        if (op_f32) op_f32[row * dpad + i] = v;
        if (op_bf16) {
            const uint16_t b = bf16_rne(v);
            op_bf16[row * dpad + i] = b;
            const double e = (double)v - (double)__uint_as_float((uint32_t)b << 16);
            err += e * e;
        }
    }
    if (op_bf16 && (bf16_err || max_bf16_err_bits)) {
        for (int off = 32; off; off >>= 1) err += __shfl_xor(err, off);
        // rounded up, so that the stored value is never below the true one
        const float ef = __uint_as_float(__float_as_uint((float)sqrt(err)) + 2u);
        if (lane == 0 && bf16_err) bf16_err[row] = ef;
        if (lane == 0 && max_bf16_err_bits) atomicMax(max_bf16_err_bits, __float_as_uint(ef));
    }
    if (lane == 0) {
        if (norms) norms[row] = norm;
        if (norm2) norm2[row] = s;
        if (cinit) cinit[row] = -(float)(0.5 * s);
        if (max_norm_bits) {
            // round up so the stored maximum is never below a true norm; word [1] keeps the smallest non-zero norm
            // (the caller starts it at the bits of FLT_MAX)
            const float up = __uint_as_float(__float_as_uint(norm) + 1u);
            atomicMax(max_norm_bits, __float_as_uint(up));
            if (s > 0.0) atomicMin(max_norm_bits + 1, __float_as_uint(norm));
        }
    }
}

hipError_t launch_row_prep(const float* d_rows, uint64_t n, uint32_t dim, uint32_t dpad, bool normalize,
                           float* d_op_f32, uint16_t* d_op_bf16, float* d_cinit, float* d_norms,
                           double* d_norm2, uint32_t* d_max_norm_bits, float* d_bf16_err, uint32_t* d_max_bf16_err_bits,
                           hipStream_t stream) {
    if (n == 0) return hipSuccess;
    const uint32_t grid = (uint32_t)((n + 3) / 4);
    hipLaunchKernelGGL(row_prep_kernel, dim3(grid), dim3(256), 0, stream, d_rows, n, dim, dpad,
                       normalize ? 1 : 0, d_op_f32, d_op_bf16, d_cinit, d_norms, d_norm2, d_max_norm_bits, d_bf16_err,
                       d_max_bf16_err_bits);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void validate_rows_kernel(const float* __restrict__ rows, uint64_t total,
                                                            uint32_t dim, uint32_t* __restrict__ bad) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const float v = rows[i];
        if (!(fabsf(v) <= 3.40282347e+38f)) atomicMin(bad, (uint32_t)(i / dim));
    }
}

// ---- scalar-quantized operands --------------------------------------------
// v = (c / scale) + min in f32 exactly as ScalarQuantizer::dequantize, src/indexing/quantization.rs:254-260
__global__ __launch_bounds__(256) void sq8_row_prep_kernel(const uint8_t* __restrict__ codes, uint64_t n, uint32_t dim,
                                                           uint32_t stride, const float* __restrict__ mins,
                                                           const float* __restrict__ scale, float* __restrict__ cinit,
                                                           uint32_t* __restrict__ max_norm_bits) {  // [0] max |v|, [1] max |codes|
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t row = (uint64_t)blockIdx.x * 4 + wave;
    if (row >= n) return;
    const uint8_t* c = codes + row * stride;
    double s = 0.0, cc = 0.0;
    for (uint32_t i = lane; i < dim; i += 64) {
        const float v = ((float)c[i] / scale[i]) + mins[i];
        s += (double)v * (double)v;
        cc += (double)c[i] * (double)c[i];
    }
    for (int off = 32; off; off >>= 1) { s += __shfl_xor(s, off); cc += __shfl_xor(cc, off); }
    if (lane == 0) {
        cinit[row] = -(float)(0.5 * s);
        const float norm = (float)sqrt(s);
        atomicMax(max_norm_bits, __float_as_uint(norm) + 1u);
        atomicMax(max_norm_bits + 1, __float_as_uint((float)sqrt(cc)) + 1u);
    }
}

hipError_t launch_sq8_row_prep(const uint8_t* d_codes, uint64_t n, uint32_t dim, uint32_t stride, const float* d_min,
                               const float* d_scale, float* d_cinit, uint32_t* d_max_norm_bits, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(sq8_row_prep_kernel, dim3((uint32_t)((n + 3) / 4)), dim3(256), 0, stream, d_codes, n, dim, stride,
                       d_min, d_scale, d_cinit, d_max_norm_bits);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void sq8_query_prep_kernel(const float* __restrict__ queries, uint32_t q, uint32_t dim,
                                                             uint32_t dpad, const float* __restrict__ mins,
                                                             const float* __restrict__ scale, float* __restrict__ qop,
                                                             uint16_t* __restrict__ qop_bf16, double* __restrict__ qconst,
                                                             double* __restrict__ qscaled_err2) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t row = blockIdx.x * 4 + wave;
    if (row >= q) return;
    const float* src = queries + (uint64_t)row * dim;
    double s = 0.0, t = 0.0;
    for (uint32_t i = lane; i < dim; i += 64) {
        const float x = src[i];
        const float xs = x / scale[i];
        if (qop) qop[(uint64_t)row * dpad + i] = xs;
        if (qop_bf16) {
            const uint16_t b = bf16_rne(xs);
            qop_bf16[(uint64_t)row * dpad + i] = b;
            const double e = (double)xs - (double)__uint_as_float((uint32_t)b << 16);
            t += e * e;  // |q/scale - bf16(q/scale)|^2
        }
        s += (double)x * (double)x - 2.0 * (double)x * (double)mins[i];
    }
    for (int off = 32; off; off >>= 1) { s += __shfl_xor(s, off); t += __shfl_xor(t, off); }
    if (lane == 0) {
        qconst[row] = s;
        if (qscaled_err2) qscaled_err2[row] = t;
    }
}

hipError_t launch_sq8_query_prep(const float* d_queries, uint32_t q, uint32_t dim, uint32_t dpad, const float* d_min,
                                 const float* d_scale, float* d_qop, uint16_t* d_qop_bf16, double* d_qconst,
                                 double* d_qscaled_err2, hipStream_t stream) {
    if (q == 0) return hipSuccess;
    hipLaunchKernelGGL(sq8_query_prep_kernel, dim3((q + 3) / 4), dim3(256), 0, stream, d_queries, q, dim, dpad, d_min,
                       d_scale, d_qop, d_qop_bf16, d_qconst, d_qscaled_err2);
    return hipGetLastError();
}

// bf16 operand copy of a code matrix: every u8 value is exact in bf16
__global__ __launch_bounds__(256) void sq8_codes_bf16_kernel(const uint8_t* __restrict__ codes, uint64_t n, uint32_t dim,
                                                             uint32_t stride, uint32_t dpad, uint16_t* __restrict__ out) {
    const uint64_t row = blockIdx.x;
    for (uint32_t i = threadIdx.x; i < dim; i += 256)
        out[row * dpad + i] = (uint16_t)(__float_as_uint((float)codes[row * stride + i]) >> 16);
}

hipError_t launch_sq8_codes_bf16(const uint8_t* d_codes, uint64_t n, uint32_t dim, uint32_t stride, uint32_t dpad,
                                 uint16_t* d_out, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(sq8_codes_bf16_kernel, dim3((uint32_t)n), dim3(256), 0, stream, d_codes, n, dim, stride, dpad, d_out);
    return hipGetLastError();
}

hipError_t launch_validate_rows(const float* d_rows, uint64_t n, uint32_t dim, uint32_t* d_bad_row,
                                hipStream_t stream) {
    hipError_t e = hipMemsetAsync(d_bad_row, 0xff, sizeof(uint32_t), stream);
    if (e != hipSuccess || n == 0) return e;
    const uint64_t total = n * dim;
    const uint32_t grid = (uint32_t)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(validate_rows_kernel, dim3(grid), dim3(256), 0, stream, d_rows, total, dim, d_bad_row);
    return hipGetLastError();
}

// ---- product-quantized rows as a matrix-core operand ------------------------------------------------------------
// The decoded row (concatenation of the centroids its codes select) rounded to bf16; cinit = -|row|^2 / 2;
// max_bits[0] = max |row| and max_bits[1] = max |row - bf16(row)|, both rounded up.  One wave per row.
__global__ __launch_bounds__(256) void pq_decode_bf16_kernel(const uint8_t* __restrict__ codes, const float* __restrict__ books,
                                                             uint64_t n, uint32_t m, uint32_t dsub, uint32_t dpad,
                                                             uint16_t* __restrict__ op, float* __restrict__ cinit,
                                                             uint32_t* __restrict__ max_bits) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t row = (uint64_t)blockIdx.x * 4 + wave;
    if (row >= n) return;
    const uint32_t dim = m * dsub;
    double s = 0.0, err = 0.0;
    for (uint32_t i = lane; i < dim; i += 64) {
        const uint32_t sub = i / dsub, j = i % dsub;
        const float v = books[((uint64_t)sub * 256 + codes[row * m + sub]) * dsub + j];
        const uint16_t b = bf16_rne(v);
        op[row * dpad + i] = b;
        const double e = (double)v - (double)__uint_as_float((uint32_t)b << 16);
        s += (double)v * (double)v;
        err += e * e;
    }
    for (int off = 32; off; off >>= 1) { s += __shfl_xor(s, off); err += __shfl_xor(err, off); }
    if (lane == 0) {
        cinit[row] = -(float)(0.5 * s);
        atomicMax(max_bits, __float_as_uint((float)sqrt(s)) + 1u);
        atomicMax(max_bits + 1, __float_as_uint((float)sqrt(err)) + 2u);
    }
}

hipError_t launch_pq_decode_bf16(const uint8_t* d_codes, const float* d_books, uint64_t n, uint32_t m, uint32_t dsub, uint32_t dpad,
                                 uint16_t* d_op_bf16, float* d_cinit, uint32_t* d_max_bits, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(pq_decode_bf16_kernel, dim3((uint32_t)((n + 3) / 4)), dim3(256), 0, stream, d_codes, d_books, n, m, dsub, dpad,
                       d_op_bf16, d_cinit, d_max_bits);
    return hipGetLastError();
}

// ---- half-precision rows -------------------------------------------------------------------------------------
// F16::to_f32 / Bf16::to_f32, src/indexing/float16.rs:101-136, 223-225: exact widening; an f16 NaN becomes the
// canonical quiet NaN (which the build then rejects like any NaN row).
__device__ inline float half_to_f32(uint32_t bits, int bf16) {
    if (bf16) return __uint_as_float(bits << 16);
    const uint32_t sign = (bits >> 15) & 1u, exp = (bits >> 10) & 0x1fu, mant = bits & 0x3ffu;
    if (exp == 0) {
        if (mant == 0) return __uint_as_float(sign << 31);
        const int sh = __builtin_clz(mant) - 21;  // shifts that bring the leading one to bit 10
        const uint32_t m = (mant << sh) & 0x3ffu;
        return __uint_as_float((sign << 31) | ((uint32_t)(127 - 15 + 1 - sh) << 23) | (m << 13));
    }
    if (exp == 31) return __uint_as_float((sign << 31) | (mant == 0 ? 0x7f800000u : 0x7fc00000u));
    return __uint_as_float((sign << 31) | ((exp - 15 + 127) << 23) | (mant << 13));
}

__global__ void half_widen_kernel(const uint16_t* __restrict__ in, uint64_t total, int bf16, float* __restrict__ out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < total; i += stride) out[i] = half_to_f32(in[i], bf16);
}

hipError_t launch_half_widen(const uint16_t* d_in, uint64_t total, bool bf16, float* d_out, hipStream_t stream) {
    if (total == 0) return hipSuccess;
    const uint32_t grid = (uint32_t)std::min<uint64_t>((total + 255) / 256, 1u << 20);
    hipLaunchKernelGGL(half_widen_kernel, dim3(grid), dim3(256), 0, stream, d_in, total, bf16 ? 1 : 0, d_out);
    return hipGetLastError();
}

// Accumulator seeds of one search under row masks: a row that is deleted, filtered out or beyond n gets -inf, so
// its first-pass score is -inf and it never passes a threshold -- the scan itself then runs without mask lookups.
__global__ __launch_bounds__(256) void masked_seed_kernel(const float* __restrict__ seed, const uint32_t* __restrict__ valid,
                                                          const uint32_t* __restrict__ filter, uint32_t n, uint32_t n_pad,
                                                          float* __restrict__ out) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_pad) return;
    bool live = i < n;
    if (live && valid) live = bit_set(valid, i);
    if (live && filter) live = bit_set(filter, i);
    out[i] = live ? (seed ? seed[i] : 0.0f) : -INFINITY;
}

// Cosine rows whose norm the reference's f32 dot(b, b) cannot represent (it underflows below about 1e-19 and overflows
// above about 1e19; the cut is taken well inside): they are listed (at most `cap` of them; the count keeps running) and
// their accumulator seeds set to -inf, so that no first pass ever lists them -- the search re-scores them for every query.
__global__ __launch_bounds__(256) void unsafe_rows_kernel(const float* __restrict__ norms, uint64_t n, float* __restrict__ seeds,
                                                          uint32_t* __restrict__ list, uint32_t cap, uint32_t* __restrict__ count) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float v = norms[i];
    const bool bad = v != 0.0f && !(v >= 1e-15f && v <= 1e15f);
    seeds[i] = bad ? -INFINITY : 0.0f;
    if (bad) {
        const uint32_t at = atomicAdd(count, 1u);
        if (at < cap) list[at] = (uint32_t)i;
    }
}

hipError_t launch_unsafe_rows(const float* d_norms, uint64_t n, float* d_seeds, uint32_t* d_list, uint32_t cap, uint32_t* d_count,
                              hipStream_t stream) {
    hipError_t e = hipMemsetAsync(d_count, 0, sizeof(uint32_t), stream);
    if (e != hipSuccess || n == 0) return e;
    hipLaunchKernelGGL(unsafe_rows_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, stream, d_norms, n, d_seeds, d_list, cap, d_count);
    return hipGetLastError();
}

// the listed rows a search may return (live and admitted by the filter), in any order
__global__ __launch_bounds__(256) void unsafe_live_kernel(const uint32_t* __restrict__ list, uint32_t n_list,
                                                          const uint32_t* __restrict__ valid, const uint32_t* __restrict__ filter,
                                                          uint32_t* __restrict__ out, uint32_t* __restrict__ out_n) {
    for (uint32_t i = threadIdx.x; i < n_list; i += 256) {
        const uint32_t row = list[i];
        if ((!valid || bit_set(valid, row)) && (!filter || bit_set(filter, row))) out[atomicAdd(out_n, 1u)] = row;
    }
}

hipError_t launch_unsafe_live(const uint32_t* d_list, uint32_t n_list, const uint32_t* d_valid, const uint32_t* d_filter, uint32_t* d_out,
                              uint32_t* d_out_n, hipStream_t stream) {
    hipError_t e = hipMemsetAsync(d_out_n, 0, sizeof(uint32_t), stream);
    if (e != hipSuccess || n_list == 0) return e;
    hipLaunchKernelGGL(unsafe_live_kernel, dim3(1), dim3(256), 0, stream, d_list, n_list, d_valid, d_filter, d_out, d_out_n);
    return hipGetLastError();
}

hipError_t launch_masked_seed(const float* d_seed, const uint32_t* d_valid, const uint32_t* d_filter, uint32_t n, uint32_t n_pad,
                              float* d_out, hipStream_t stream) {
    if (n_pad == 0) return hipSuccess;
    hipLaunchKernelGGL(masked_seed_kernel, dim3((n_pad + 255) / 256), dim3(256), 0, stream, d_seed, d_valid, d_filter, n, n_pad, d_out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// Admitted rows of a filtered search, compacted: a search whose masks admit a small part of the base scans a gathered
// copy of the admitted rows instead of the whole base (work proportional to the rows that can be returned).
//   mask_count      per block of 256 mask words (8192 rows) the number of admitted rows; the total
//   mask_offsets    exclusive sums of the block counts (one workgroup)
//   mask_positions  the admitted row numbers in ascending order
//   gather_words    rows of 4-byte words gathered through the positions; rows at or beyond `live` are zero
//   map_ids         hits of the compacted search renamed to row numbers of the base
// ---------------------------------------------------------------------------------------------------------------
constexpr uint32_t MASK_BLOCK_WORDS = 256;

__device__ inline uint32_t admitted_word(const uint32_t* __restrict__ valid, const uint32_t* __restrict__ filter, uint32_t w,
                                         uint32_t n) {
    const uint32_t words = (n + 31u) / 32u;
    if (w >= words) return 0u;
    uint32_t m = 0xffffffffu;
    if (valid) m &= valid[w];
    if (filter) m &= filter[w];
    if (w == words - 1 && (n & 31u)) m &= (1u << (n & 31u)) - 1u;  // bits beyond the last row name no row
    return m;
}

__global__ __launch_bounds__(256) void mask_count_kernel(const uint32_t* __restrict__ valid, const uint32_t* __restrict__ filter,
                                                         uint32_t n, uint32_t* __restrict__ block_counts,
                                                         uint32_t* __restrict__ total) {
    __shared__ uint32_t part[4];
    const uint32_t w = blockIdx.x * MASK_BLOCK_WORDS + threadIdx.x;
    uint32_t c = (uint32_t)__builtin_popcount(admitted_word(valid, filter, w, n));
#pragma unroll
    for (int off = 32; off; off >>= 1) c += (uint32_t)__shfl_xor((int)c, off);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t s = part[0] + part[1] + part[2] + part[3];
        block_counts[blockIdx.x] = s;
        if (s) atomicAdd(total, s);
    }
}

__global__ __launch_bounds__(1024) void mask_offsets_kernel(uint32_t* __restrict__ block_counts, uint32_t n_blocks) {
    // in place: counts -> exclusive sums; 1024 threads walk the blocks in pieces of 1024 with a running carry
    __shared__ uint32_t wave_sum[16];
    __shared__ uint32_t carry;
    const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t == 0) carry = 0;
    __syncthreads();
    for (uint32_t b0 = 0; b0 < n_blocks; b0 += 1024) {
        const uint32_t b = b0 + t;
        const uint32_t c = b < n_blocks ? block_counts[b] : 0u;
        uint32_t incl = c;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t v = (uint32_t)__shfl_up((int)incl, off);
            if (lane >= (uint32_t)off) incl += v;
        }
        if (lane == 63) wave_sum[wave] = incl;
        __syncthreads();
        uint32_t before = carry;
        for (uint32_t w = 0; w < wave; ++w) before += wave_sum[w];
        if (b < n_blocks) block_counts[b] = before + incl - c;
        __syncthreads();
        if (t == 1023) carry = before + incl;
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void mask_positions_kernel(const uint32_t* __restrict__ valid, const uint32_t* __restrict__ filter,
                                                             uint32_t n, const uint32_t* __restrict__ block_offsets,
                                                             uint32_t live, uint32_t* __restrict__ pos) {
    __shared__ uint32_t wave_sum[4];
    const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const uint32_t w = blockIdx.x * MASK_BLOCK_WORDS + t;
    uint32_t m = admitted_word(valid, filter, w, n);
    const uint32_t c = (uint32_t)__builtin_popcount(m);
    uint32_t incl = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t v = (uint32_t)__shfl_up((int)incl, off);
        if (lane >= (uint32_t)off) incl += v;
    }
    if (lane == 63) wave_sum[wave] = incl;
    __syncthreads();
    uint32_t at = block_offsets[blockIdx.x] + incl - c;
    for (uint32_t v = 0; v < wave; ++v) at += wave_sum[v];
    while (m) {
        const uint32_t bit = (uint32_t)__builtin_ctz(m);
        m &= m - 1;
        if (at < live) pos[at] = w * 32u + bit;  // (at < live always: the counts come from the same masks)
        ++at;
    }
}

// dst[r][0 .. words) = src[pos[r]][0 .. words) for r < live, zeros for live <= r < rows; VEC 4-byte words per access
template <typename V>
__global__ __launch_bounds__(256) void gather_words_kernel(const V* __restrict__ src, const uint32_t* __restrict__ pos,
                                                           uint32_t live, uint32_t rows, uint32_t n_src, uint32_t vecs_per_row,
                                                           V* __restrict__ dst) {
    const uint64_t total = (uint64_t)rows * vecs_per_row;
    for (uint64_t e = (uint64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (uint64_t)gridDim.x * 256) {
        const uint32_t r = (uint32_t)(e / vecs_per_row), v = (uint32_t)(e % vecs_per_row);
        V x{};
        if (r < live) {
            const uint32_t from = min(pos[r], n_src - 1);  // (a position is a row number of the base by construction)
            x = src[(uint64_t)from * vecs_per_row + v];
        }
        dst[e] = x;
    }
}

__global__ __launch_bounds__(256) void map_ids_kernel(uint64_t* __restrict__ ids, uint64_t total, const uint32_t* __restrict__ pos,
                                                      uint32_t live) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const uint64_t id = ids[i];
    if (id < (uint64_t)live) ids[i] = pos[id];  // padding entries (all ones) stay
}

hipError_t launch_mask_count(const uint32_t* d_valid, const uint32_t* d_filter, uint32_t n, uint32_t* d_block_counts, uint32_t* d_total,
                             hipStream_t stream) {
    hipError_t e = hipMemsetAsync(d_total, 0, sizeof(uint32_t), stream);
    if (e != hipSuccess || n == 0) return e;
    const uint32_t words = (n + 31u) / 32u, blocks = (words + MASK_BLOCK_WORDS - 1) / MASK_BLOCK_WORDS;
    hipLaunchKernelGGL(mask_count_kernel, dim3(blocks), dim3(256), 0, stream, d_valid, d_filter, n, d_block_counts, d_total);
    return hipGetLastError();
}

uint32_t mask_blocks(uint32_t n) { return ((n + 31u) / 32u + MASK_BLOCK_WORDS - 1) / MASK_BLOCK_WORDS; }

hipError_t launch_mask_positions(const uint32_t* d_valid, const uint32_t* d_filter, uint32_t n, uint32_t* d_block_counts, uint32_t live,
                                 uint32_t* d_pos, hipStream_t stream) {
    if (n == 0 || live == 0) return hipSuccess;
    const uint32_t blocks = mask_blocks(n);
    hipLaunchKernelGGL(mask_offsets_kernel, dim3(1), dim3(1024), 0, stream, d_block_counts, blocks);
    hipLaunchKernelGGL(mask_positions_kernel, dim3(blocks), dim3(256), 0, stream, d_valid, d_filter, n, d_block_counts, live, d_pos);
    return hipGetLastError();
}

hipError_t launch_gather_row_bytes(const void* d_src, const uint32_t* d_pos, uint32_t live, uint32_t rows, uint32_t n_src,
                                   size_t row_bytes, void* d_dst, hipStream_t stream) {
    if (rows == 0 || row_bytes == 0) return hipSuccess;
    if (n_src == 0 || row_bytes % 4 != 0) return hipErrorInvalidValue;
    const bool wide = row_bytes % 16 == 0 && ((uintptr_t)d_src % 16 == 0) && ((uintptr_t)d_dst % 16 == 0);
    const uint32_t vecs = (uint32_t)(row_bytes / (wide ? 16 : 4));
    const uint64_t total = (uint64_t)rows * vecs;
    const uint32_t grid = (uint32_t)std::min<uint64_t>((total + 255) / 256, 1u << 20);
    if (wide)
        hipLaunchKernelGGL(gather_words_kernel<uint4>, dim3(grid), dim3(256), 0, stream, (const uint4*)d_src, d_pos, live, rows, n_src,
                           vecs, (uint4*)d_dst);
    else
        hipLaunchKernelGGL(gather_words_kernel<uint32_t>, dim3(grid), dim3(256), 0, stream, (const uint32_t*)d_src, d_pos, live, rows,
                           n_src, vecs, (uint32_t*)d_dst);
    return hipGetLastError();
}

hipError_t launch_map_ids(uint64_t* d_ids, uint64_t total, const uint32_t* d_pos, uint32_t live, hipStream_t stream) {
    if (total == 0) return hipSuccess;
    hipLaunchKernelGGL(map_ids_kernel, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, stream, d_ids, total, d_pos, live);
    return hipGetLastError();
}

}  // namespace nh
