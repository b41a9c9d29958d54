// Exact (reference-order) scalar distances on the device.
//
// These reproduce the reference's default-build arithmetic bit for bit: one
// rounding per multiply and per add (the translation units are compiled with
// -ffp-contract=off), IEEE-correct sqrt and divide
// (-fhip-fp32-correctly-rounded-divide-sqrt), subnormals kept.  The one fused
// operation is the explicit fma of the SIMD-build orders (exact_avx).
#pragma once
#include "nh_common.h"

#pragma clang fp contract(off)

namespace nh {

// ---- src/distance.rs:153-316 (sequential sums folded from -0.0) ----------
__device__ inline float exact_rs(int metric, const float* __restrict__ q,
                                 const float* __restrict__ b, uint32_t dim) {
    switch (metric) {
    case NEEDLE_METRIC_COSINE: {  // distance.rs:153-164
        float dot = -0.0f, qa = -0.0f, bb = -0.0f;
        for (uint32_t i = 0; i < dim; ++i) {
            float x = q[i], y = b[i];
            float p = x * y; dot = dot + p;
            float pq = x * x; qa = qa + pq;
            float pb = y * y; bb = bb + pb;
        }
        float na = sqrtf(qa), nb = sqrtf(bb);
        if (na == 0.0f || nb == 0.0f) return 1.0f;
        float den = na * nb;
        return 1.0f - (dot / den);
    }
    case NEEDLE_METRIC_COSINE_NORMALIZED: {  // distance.rs:193-196
        float dot = -0.0f;
        for (uint32_t i = 0; i < dim; ++i) { float p = q[i] * b[i]; dot = dot + p; }
        return 1.0f - dot;
    }
    case NEEDLE_METRIC_EUCLIDEAN: {  // distance.rs:203-205, 249-257
        float s = -0.0f;
        for (uint32_t i = 0; i < dim; ++i) { float d = q[i] - b[i]; float p = d * d; s = s + p; }
        return sqrtf(s);
    }
    case NEEDLE_METRIC_DOT_PRODUCT: {  // distance.rs:264-266
        float dot = -0.0f;
        for (uint32_t i = 0; i < dim; ++i) { float p = q[i] * b[i]; dot = dot + p; }
        return -dot;
    }
    case NEEDLE_METRIC_MANHATTAN: {  // distance.rs:360-362
        float s = -0.0f;
        for (uint32_t i = 0; i < dim; ++i) s = s + fabsf(q[i] - b[i]);
        return s;
    }
    case NEEDLE_METRIC_HAMMING: {  // distance.rs:387-395
        uint32_t c = 0;
        for (uint32_t i = 0; i < dim; ++i) c += ((q[i] != 0.0f) != (b[i] != 0.0f)) ? 1u : 0u;
        return (float)c;
    }
    default: {  // Chebyshev, distance.rs:415-423: fold(0.0, f32::max), max drops NaN
        float m = 0.0f;
        for (uint32_t i = 0; i < dim; ++i) {
            float d = fabsf(q[i] - b[i]);
            m = (d != d) ? m : ((m != m) ? d : (d > m ? d : m));
        }
        return m;
    }
    }
}

// ---- src/experimental/gpu/common.rs:515-581 (groups of four) -------------
__device__ inline float dot_chunk4(const float* __restrict__ a, const float* __restrict__ b,
                                   uint32_t dim) {
    uint32_t chunks = dim / 4;
    float sum = 0.0f;
    for (uint32_t c = 0; c < chunks; ++c) {
        uint32_t o = c * 4;
        float p0 = a[o] * b[o], p1 = a[o + 1] * b[o + 1];
        float p2 = a[o + 2] * b[o + 2], p3 = a[o + 3] * b[o + 3];
        float g = ((p0 + p1) + p2) + p3;
        sum = sum + g;
    }
    for (uint32_t i = chunks * 4; i < dim; ++i) { float p = a[i] * b[i]; sum = sum + p; }
    return sum;
}

__device__ inline float euclidean_chunk4(const float* __restrict__ a, const float* __restrict__ b,
                                         uint32_t dim) {
    uint32_t chunks = dim / 4;
    float sum = 0.0f;
    for (uint32_t c = 0; c < chunks; ++c) {
        uint32_t o = c * 4;
        float d0 = a[o] - b[o], d1 = a[o + 1] - b[o + 1];
        float d2 = a[o + 2] - b[o + 2], d3 = a[o + 3] - b[o + 3];
        float s0 = d0 * d0, s1 = d1 * d1, s2 = d2 * d2, s3 = d3 * d3;
        float g = ((s0 + s1) + s2) + s3;
        sum = sum + g;
    }
    for (uint32_t i = chunks * 4; i < dim; ++i) { float d = a[i] - b[i]; float p = d * d; sum = sum + p; }
    return sqrtf(sum);
}

// fused_search's distances, gpu/mod.rs:717-762: cosine with the query norm
// precomputed (zero query norm -> 1.0 for every row), Euclidean, negated dot.
__device__ inline float exact_chunk4(int metric, const float* __restrict__ q,
                                     const float* __restrict__ b, uint32_t dim) {
    switch (metric) {
    case NEEDLE_METRIC_COSINE: {
        float qn = sqrtf(dot_chunk4(q, q, dim));
        if (qn == 0.0f) return 1.0f;
        float dot = dot_chunk4(q, b, dim);
        float nb = sqrtf(dot_chunk4(b, b, dim));
        if (nb == 0.0f) return 1.0f;
        float den = qn * nb;
        return 1.0f - (dot / den);
    }
    case NEEDLE_METRIC_EUCLIDEAN: return euclidean_chunk4(q, b, dim);
    default: return -dot_chunk4(q, b, dim);  // DotProduct
    }
}

// ScalarQuantizer::asymmetric_distance, src/indexing/quantization.rs:277-285:
// decoded = (c as f32 / scale[i]) + min[i]; sum (from +0.0) of (q - decoded)^2; sqrt.
__device__ inline float exact_sq8(const float* __restrict__ q, const uint8_t* __restrict__ codes,
                                  const float* __restrict__ mins, const float* __restrict__ scale, uint32_t dim) {
    float sum = 0.0f;
    for (uint32_t i = 0; i < dim; ++i) {
        const float dec = ((float)codes[i] / scale[i]) + mins[i];
        const float diff = q[i] - dec;
        const float sq = diff * diff;
        sum = sum + sq;
    }
    return sqrtf(sum);
}

// ProductQuantizer::{compute_distance_table, distance_with_table}, src/indexing/quantization.rs:468-503: per subspace
// the sum (from -0.0) of (q - centroid)^2 over its dimensions -- the table entry of the row's code --, the entries
// summed from +0.0 in subspace order; sqrt.
__device__ inline float exact_pq(const float* __restrict__ q, const uint8_t* __restrict__ codes,
                                 const float* __restrict__ codebooks, uint32_t m, uint32_t dsub) {
    float dist = 0.0f;
    for (uint32_t s = 0; s < m; ++s) {
        const float* c = codebooks + ((uint64_t)s * 256 + codes[s]) * dsub;
        const float* x = q + s * dsub;
        float t = -0.0f;
        for (uint32_t j = 0; j < dsub; ++j) { const float d = x[j] - c[j]; const float sq = d * d; t = t + sq; }
        dist = dist + t;
    }
    return sqrtf(dist);
}

// ProductQuantizer::asymmetric_distance, src/indexing/quantization.rs:449-465 (the distance of an IVF-PQ list scan,
// src/indexing/ivf.rs:538-546): ONE accumulator from +0.0 across all subspaces and dimensions, then sqrt.
__device__ inline float exact_pq_asym(const float* __restrict__ q, const uint8_t* __restrict__ codes,
                                      const float* __restrict__ codebooks, uint32_t m, uint32_t dsub) {
    float dist = 0.0f;
    for (uint32_t s = 0; s < m; ++s) {
        const float* c = codebooks + ((uint64_t)s * 256 + codes[s]) * dsub;
        const float* x = q + s * dsub;
        for (uint32_t j = 0; j < dsub; ++j) { const float d = x[j] - c[j]; const float sq = d * d; dist = dist + sq; }
    }
    return sqrtf(dist);
}

// ---- src/services/compute/gpu_kernels.rs:345-358 (the compute service's own kernels) ----
// cosine: one loop, three accumulators from +0.0, denominator below f32::EPSILON -> 1.0; euclidean and dot
// product are distance.rs's sums (f32::sum from -0.0).
__device__ inline float exact_service(int metric, const float* __restrict__ q, const float* __restrict__ b, uint32_t dim) {
    if (metric != NEEDLE_METRIC_COSINE) return exact_rs(metric, q, b, dim);
    float dot = 0.0f, na = 0.0f, nb = 0.0f;
    for (uint32_t i = 0; i < dim; ++i) {
        const float x = q[i], y = b[i];
        const float p = x * y; dot = dot + p;
        const float pa = x * x; na = na + pa;
        const float pb = y * y; nb = nb + pb;
    }
    const float den = sqrtf(na) * sqrtf(nb);
    if (den < 1.1920929e-07f) return 1.0f;
    return 1.0f - (dot / den);
}

// ---- src/distance.rs:481-738: the arithmetic of a reference build with AVX2 or AVX-512 enabled ----
// L lane sums from +0.0 (L = 8: AVX2, :505-591; L = 16: AVX-512, :636-715; L = 4: the aarch64 build's NEON, :741-834); lane l folds elements l, l + L, l + 2L, ...
// with ONE rounding per element (a fused multiply-add; Manhattan: a plain add of |a - b|); the lanes are folded as
// hsum_avx does (:489-496), ((l0 + l4) + (l2 + l6)) + ((l1 + l5) + (l3 + l7)) -- sixteen lanes first as l_i + l_(i+8)
// (hsum_avx512, :619-627); the dim % L tail is then added to that result one product at a time (multiply and add
// rounded separately).
enum AvxOp { AVX_DOT = 0, AVX_L2 = 1, AVX_L1 = 2 };
template <int L>
struct AvxAccT { float l[L]; float r; };  // lane sums; r: the folded sum, once the tail has begun
using AvxAcc = AvxAccT<8>;

template <int L>
__device__ inline void avx_init(AvxAccT<L>& s) {
#pragma unroll
    for (int j = 0; j < L; ++j) s.l[j] = 0.0f;
    s.r = 0.0f;
}
__device__ inline float avx_hsum(const float (&l)[8]) {
    const float s0 = l[0] + l[4], s1 = l[1] + l[5], s2 = l[2] + l[6], s3 = l[3] + l[7];
    const float t0 = s0 + s2, t1 = s1 + s3;
    return t0 + t1;
}
// vaddvq_f32 (NEON, four lanes): the pairwise reduction the intrinsic compiles to, FADDP twice: (l0 + l1) + (l2 + l3)
__device__ inline float avx_hsum(const float (&l)[4]) {
    const float s0 = l[0] + l[1], s1 = l[2] + l[3];
    return s0 + s1;
}
__device__ inline float avx_hsum(const float (&l)[16]) {
    float h[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) h[j] = l[j] + l[j + 8];
    return avx_hsum(h);
}
// elements [i0, i1) of the two vectors, at x[0 ..] and y[0 ..]; i0 is a multiple of L, i1 a multiple of L or dim
template <int OP, int L>
__device__ inline void avx_accumulate(AvxAccT<L>& s, const float* x, const float* y, uint32_t i0, uint32_t i1, uint32_t dim) {
    const uint32_t vec_end = dim / L * L;
    uint32_t i = i0;
    for (; i + L <= i1 && i + L <= vec_end; i += L) {
#pragma unroll
        for (int j = 0; j < L; ++j) {
            const float a = x[i - i0 + j], b = y[i - i0 + j];
            if (OP == AVX_DOT) {
                s.l[j] = __builtin_fmaf(a, b, s.l[j]);
            } else if (OP == AVX_L2) {
                const float d = a - b;
                s.l[j] = __builtin_fmaf(d, d, s.l[j]);
            } else {
                const float d = a - b;
                s.l[j] = s.l[j] + fabsf(d);
            }
        }
    }
    for (; i < i1; ++i) {  // the tail: i >= vec_end
        if (i == vec_end) s.r = avx_hsum(s.l);
        const float a = x[i - i0], b = y[i - i0];
        if (OP == AVX_DOT) {
            const float pr = a * b;
            s.r = s.r + pr;
        } else if (OP == AVX_L2) {
            const float d = a - b;
            const float pr = d * d;
            s.r = s.r + pr;
        } else {
            s.r = s.r + fabsf(a - b);
        }
    }
}
template <int L>
__device__ inline float avx_finish(const AvxAccT<L>& s, uint32_t dim) { return (dim % L) ? s.r : avx_hsum(s.l); }

// lanes of the SIMD build an arithmetic names (0: not one of them)
__device__ __host__ inline int avx_lanes(int arith) {
    return arith == NEEDLE_ARITH_AVX2 ? 8 : (arith == NEEDLE_ARITH_AVX512 ? 16 : (arith == NEEDLE_ARITH_NEON ? 4 : 0));
}
// does a SIMD build compute this metric differently from the scalar build?  (Hamming and Chebyshev have no SIMD form)
__device__ __host__ inline bool avx2_metric(int metric) {
    return metric == NEEDLE_METRIC_COSINE || metric == NEEDLE_METRIC_COSINE_NORMALIZED || metric == NEEDLE_METRIC_EUCLIDEAN ||
           metric == NEEDLE_METRIC_DOT_PRODUCT || metric == NEEDLE_METRIC_MANHATTAN;
}
// the metric's distance from the finished sums: a = q.b (or the squared / absolute differences), b = q.q, c = b.b
__device__ inline float avx2_distance(int metric, float a, float qq, float bb) {
    switch (metric) {
    case NEEDLE_METRIC_COSINE: {  // distance.rs:153-164
        const float na = sqrtf(qq), nb = sqrtf(bb);
        if (na == 0.0f || nb == 0.0f) return 1.0f;
        const float den = na * nb;
        return 1.0f - (a / den);
    }
    case NEEDLE_METRIC_COSINE_NORMALIZED: return 1.0f - a;  // :193-196
    case NEEDLE_METRIC_EUCLIDEAN: return sqrtf(a);          // :203-228
    case NEEDLE_METRIC_DOT_PRODUCT: return -a;              // :264-266
    default: return a;                                      // Manhattan, :323-339
    }
}
template <int L>
__device__ inline float exact_avx(int metric, const float* __restrict__ q, const float* __restrict__ b, uint32_t dim) {
    if (!avx2_metric(metric)) return exact_rs(metric, q, b, dim);
    AvxAccT<L> s, sq, sb;
    avx_init(s); avx_init(sq); avx_init(sb);
    if (metric == NEEDLE_METRIC_EUCLIDEAN) avx_accumulate<AVX_L2>(s, q, b, 0, dim, dim);
    else if (metric == NEEDLE_METRIC_MANHATTAN) avx_accumulate<AVX_L1>(s, q, b, 0, dim, dim);
    else avx_accumulate<AVX_DOT>(s, q, b, 0, dim, dim);
    if (metric == NEEDLE_METRIC_COSINE) {
        avx_accumulate<AVX_DOT>(sq, q, q, 0, dim, dim);
        avx_accumulate<AVX_DOT>(sb, b, b, 0, dim, dim);
    }
    return avx2_distance(metric, avx_finish(s, dim), avx_finish(sq, dim), avx_finish(sb, dim));
}
__device__ inline float exact_avx2(int metric, const float* __restrict__ q, const float* __restrict__ b, uint32_t dim) {
    return exact_avx<8>(metric, q, b, dim);
}
// arith: NEEDLE_ARITH_AVX2, NEEDLE_ARITH_AVX512 or NEEDLE_ARITH_NEON (src/distance.rs:741-834: four lanes of vfmaq_f32,
// folded by vaddvq_f32, the len % 4 tail after the fold)
__device__ inline float exact_simd_build(int arith, int metric, const float* __restrict__ q, const float* __restrict__ b, uint32_t dim) {
    if (arith == NEEDLE_ARITH_NEON) return exact_avx<4>(metric, q, b, dim);
    return arith == NEEDLE_ARITH_AVX512 ? exact_avx<16>(metric, q, b, dim) : exact_avx<8>(metric, q, b, dim);
}

__device__ inline float exact_distance(int metric, int arith, const float* __restrict__ q,
                                       const float* __restrict__ b, uint32_t dim) {
    if (arith == NEEDLE_ARITH_SERVICE) return exact_service(metric, q, b, dim);
    if (avx_lanes(arith)) return exact_simd_build(arith, metric, q, b, dim);
    return arith == NEEDLE_ARITH_ACCEL_CHUNK4 ? exact_chunk4(metric, q, b, dim)
                                              : exact_rs(metric, q, b, dim);
}

}  // namespace nh
