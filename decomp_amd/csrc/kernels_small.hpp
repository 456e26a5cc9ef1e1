// Small HBM-bound kernels around the GEMM cores: split-K slab reduction, row
// normalisation (+ max|dD|), elementwise products / quotients, column sums,
// sign scans.  All are deterministic (no float atomics).
#pragma once
#include <hip/hip_runtime.h>
#include "scalar.hpp"

namespace dcp {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
template <class R>
__device__ __forceinline__ R wave_max(R v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        R w = __shfl_down(v, o, 64);
        v = (w > v || w != w) ? w : v;  // NaN propagates, as in np.max
    }
    return v;
}

__device__ __forceinline__ void atomic_max_nonneg(float* p, float v) {
    atomicMax(reinterpret_cast<unsigned int*>(p), __float_as_uint(v));
}
__device__ __forceinline__ void atomic_max_nonneg(double* p, double v) {
    atomicMax(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v));
}
__device__ __forceinline__ float atomic_max_nonneg_ret(float* p, float v) {
    return __uint_as_float(atomicMax(reinterpret_cast<unsigned int*>(p), __float_as_uint(v)));
}
__device__ __forceinline__ double atomic_max_nonneg_ret(double* p, double v) {
    return __longlong_as_double((long long)atomicMax(reinterpret_cast<unsigned long long*>(p),
                                                     (unsigned long long)__double_as_longlong(v)));
}
// the value as the memory-side atomic unit holds it (a plain load may hit a line of this XCD's L2)
__device__ __forceinline__ float atomic_read_nonneg(float* p) {
    return __uint_as_float(atomicMax(reinterpret_cast<unsigned int*>(p), 0u));
}
__device__ __forceinline__ double atomic_read_nonneg(double* p) {
    return __longlong_as_double((long long)atomicMax(reinterpret_cast<unsigned long long*>(p), 0ull));
}

// Block-wide reductions for 256-thread blocks; result valid in thread 0.
template <class R>
__device__ __forceinline__ R block_sum_256(R v, R* sh /* >= 4 */) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) sh[w] = v;
    __syncthreads();
    R r = 0;
    if (threadIdx.x == 0) r = sh[0] + sh[1] + sh[2] + sh[3];
    __syncthreads();
    return r;
}
template <class R>
__device__ __forceinline__ R block_max_256(R v, R* sh) {
    v = wave_max(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) sh[w] = v;
    __syncthreads();
    R r = 0;
    if (threadIdx.x == 0) {
        r = sh[0];
        for (int i = 1; i < 4; ++i) r = (sh[i] > r || sh[i] != sh[i]) ? sh[i] : r;
    }
    __syncthreads();
    return r;
}

// out[i] = sum_s slabs[s*stride + i], s in fixed order 0..S-1 (bitwise reproducible).
template <class T>
__global__ void __launch_bounds__(256) reduce_slabs_kernel(const T* __restrict__ slabs, long stride,
                                                           int S, long count, T* __restrict__ out) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < count; i += (long)gridDim.x * 256L) {
        T acc = slabs[i];
        int s = 1;
        // same left-to-right order, sixteen (then eight) loads in flight (one at a time the loop is a chain
        // of L2 round trips: 16 us for 64 slabs of a 64 x 64 Gram matrix; eight: 5.9 us)
        for (; s + 15 < S; s += 16) {
            T v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = slabs[(long)(s + u) * stride + i];
#pragma unroll
            for (int u = 0; u < 16; ++u) acc = add(acc, v[u]);
        }
        for (; s + 7 < S; s += 8) {
            T v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = slabs[(long)(s + u) * stride + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = add(acc, v[u]);
        }
        for (; s < S; ++s) acc = add(acc, slabs[(long)s * stride + i]);
        out[i] = acc;
    }
}

// The same sum on 16-byte vectors (float x 4 / double x 2 per thread and slab): same order per element, so the same
// bits; a quarter of the threads and memory instructions on what is a latency-bound stream (the [K, F+K] statistics of
// an NMF iteration: 13 -> ~7 us).  count, stride multiples of VEC; slabs, out 16-byte aligned.
template <class T, int VEC>
__global__ void __launch_bounds__(256) reduce_slabs_vec_kernel(const T* __restrict__ slabs, long stride, int S,
                                                               long count, T* __restrict__ out) {
    typedef T vec_t __attribute__((ext_vector_type(VEC)));
    const long nvec = count / VEC, svec = stride / VEC;
    const vec_t* __restrict__ in = reinterpret_cast<const vec_t*>(slabs);
    vec_t* __restrict__ o = reinterpret_cast<vec_t*>(out);
    for (long i = blockIdx.x * 256L + threadIdx.x; i < nvec; i += (long)gridDim.x * 256L) {
        vec_t acc = in[i];
        int s = 1;
        for (; s + 7 < S; s += 8) {
            vec_t v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = in[(long)(s + u) * svec + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
        }
        for (; s + 1 < S; s += 2) {
            const vec_t v0 = in[(long)s * svec + i], v1 = in[(long)(s + 1) * svec + i];
            acc += v0;
            acc += v1;
        }
        for (; s < S; ++s) acc += in[(long)s * svec + i];
        o[i] = acc;
    }
}

// Ordered slab sum, vectorised where the layout allows (real types).
template <class T>
inline void launch_reduce_slabs(hipStream_t st, const T* slabs, long stride, int S, long count, T* out) {
    if constexpr (std::is_same<T, float>::value || std::is_same<T, double>::value) {
        constexpr int VEC = 16 / sizeof(T);
        if (count % VEC == 0 && stride % VEC == 0 && (reinterpret_cast<uintptr_t>(slabs) & 15) == 0 &&
            (reinterpret_cast<uintptr_t>(out) & 15) == 0 && count >= 64 * 1024) {
            long g = (count / VEC + 255) / 256;
            if (g > 2048) g = 2048;
            hipLaunchKernelGGL((reduce_slabs_vec_kernel<T, VEC>), dim3((unsigned)g), dim3(256), 0, st, slabs, stride, S,
                               count, out);
            return;
        }
    }
    long g = (count + 255) / 256;
    if (g < 1) g = 1;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL((reduce_slabs_kernel<T>), dim3((unsigned)g), dim3(256), 0, st, slabs, stride, S, count, out);
}

// Same, for a [rows, cols] matrix written into a wider destination (leading dim ld_out).
template <class T>
__global__ void __launch_bounds__(256) reduce_slabs_rows_kernel(const T* __restrict__ slabs,
                                                                long stride, int S, long rows,
                                                                long cols, T* __restrict__ out,
                                                                long ld_out) {
    const long count = rows * cols;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < count; i += (long)gridDim.x * 256L) {
        T acc = slabs[i];
        for (int s = 1; s < S; ++s) acc = add(acc, slabs[(long)s * stride + i]);
        const long r = i / cols, c = i - r * cols;
        out[r * ld_out + c] = acc;
    }
}

// out[r, c] = v[r] for c in [0, cols)  (a per-row value broadcast along the row)
template <class T>
__global__ void __launch_bounds__(256) bcast_rows_kernel(const T* __restrict__ v, long rows,
                                                         long cols, T* __restrict__ out,
                                                         long ld_out) {
    const long count = rows * cols;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < count; i += (long)gridDim.x * 256L) {
        const long r = i / cols, c = i - r * cols;
        out[r * ld_out + c] = v[r];
    }
}

// out[r, c] = v[c] for r in [0, rows)  (a per-column value repeated on every row)
template <class T>
__global__ void __launch_bounds__(256) bcast_cols_kernel(const T* __restrict__ v, long rows, long cols,
                                                         T* __restrict__ out) {
    const long count = rows * cols;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < count; i += (long)gridDim.x * 256L)
        out[i] = v[i % cols];
}

// Gaussian.logp (grads.py:127-135), summand (-0.5 ((y - f) / scale)^2 - log(scale) - pi / 2) [* mask] with
// d = y - f given; double accumulation -> partial[block].
template <class T>
__global__ void __launch_bounds__(256) gauss_logp_partial_kernel(const T* __restrict__ d,
                                                                 const T* __restrict__ mask, long n,
                                                                 double inv_scale, double cst,
                                                                 double* __restrict__ partial) {
    __shared__ double sh[4];
    double c = 0;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L) {
        const double z = (double)d[i] * inv_scale;
        double t = -0.5 * z * z - cst;
        if (mask != nullptr) t *= (double)mask[i];
        c += t;
    }
    double t = block_sum_256(c, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

// out = a o b (elementwise; b real "mask" of the same shape, or broadcast along rows when
// b_row_stride == 0).
template <class T>
__global__ void __launch_bounds__(256) mul_mask_kernel(const T* __restrict__ a,
                                                       const real_t<T>* __restrict__ m, long rows,
                                                       long cols, long m_row_stride,
                                                       T* __restrict__ out) {
    const long n = rows * cols;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L) {
        const long r = i / cols, c = i - r * cols;
        out[i] = scale(a[i], m[r * m_row_stride + c]);
    }
}

// Row-bit image of a mask (EpiMulMaskBits): word (g, c) holds (mask[32 g + b, c] != 0) in bit b.
// *not_binary is set when some entry is neither 0 nor 1 (the bit image then cannot stand in for the
// mask).  One thread per (row group, column): the 32 row reads of a wave are 256-byte coalesced rows.
template <class R>
__global__ void __launch_bounds__(256) mask_rowbits_kernel(const R* __restrict__ mask, long rows, long cols,
                                                           uint32_t* __restrict__ bits,
                                                           int* __restrict__ not_binary) {
    const long groups = (rows + 31) / 32;
    const long n = groups * cols;
    bool bad = false;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L) {
        const long g = i / cols, c = i - g * cols;
        uint32_t w = 0;
        // eight rows in flight per round: unconditional loads from a clamped row, selected afterwards
        // (a load under `if (r < rows)` is a branch and a full wait of its own)
        for (int b0 = 0; b0 < 32; b0 += 8) {
            R m[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const long r = g * 32 + b0 + u;
                m[u] = mask[(r < rows ? r : rows - 1) * cols + c];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const bool in = (g * 32 + b0 + u) < rows;
                if (in && m[u] != R(0)) w |= (1u << (b0 + u));
                if (in && m[u] != R(0) && m[u] != R(1)) bad = true;
            }
        }
        bits[i] = w;
    }
    if (bad) atomicOr(not_binary, 1);
}

// out = cur * max(num, 0) / max(den, eps)   (grads.py:84,93), arbitrary leading dims.
// den_bcast: 0 = full [rows, cols]; 1 = one value per row (den[row]); 2 = one per column.
template <class T>
__global__ void __launch_bounds__(256) mu_quotient_kernel(const T* __restrict__ cur, long ld_cur,
                                                          const T* __restrict__ num, long ld_num,
                                                          const T* __restrict__ den, long ld_den,
                                                          int den_bcast, long rows, long cols,
                                                          T* __restrict__ out, long ld_out) {
    const long n = rows * cols;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L) {
        const long r = i / cols, c = i - r * cols;
        const T d = den_bcast == 0 ? den[r * ld_den + c] : (den_bcast == 1 ? den[r] : den[c]);
        const T nu = num[r * ld_num + c];
        out[r * ld_out + c] =
            cur[r * ld_cur + c] * max_np(nu, T(0)) / max_np(d, T(1.0e-15));
    }
}

// out = cur * max(sum_s slabs[s], 0) / max(den, eps): the MU quotient for a split-K numerator.
// den: [rows, cols] (ld_den = cols) or one value per column (ld_den = 0).
template <class T>
__global__ void __launch_bounds__(256) mu_quotient_slabs_kernel(const T* __restrict__ cur,
                                                                const T* __restrict__ slabs,
                                                                long stride, int S,
                                                                const T* __restrict__ den, long ld_den,
                                                                long rows, long cols,
                                                                T* __restrict__ out) {
    const long n = rows * cols;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L) {
        T nu = slabs[i];
        for (int s = 1; s < S; ++s) nu = nu + slabs[(long)s * stride + i];
        const long r = i / cols, c = i - r * cols;
        const T d = den[r * ld_den + c];
        out[i] = cur[i] * max_np(nu, T(0)) / max_np(d, T(1.0e-15));
    }
}

// The MU quotient for the STACKED split-K product [den ; num] (rows [0, count) of every slab hold the
// negative part, rows [count, 2 count) the positive part): out = cur * max(sum num, 0) / max(sum den, eps).
template <class T>
__global__ void __launch_bounds__(256) mu_quotient_stacked_kernel(const T* __restrict__ cur,
                                                                  const T* __restrict__ slabs, long stride,
                                                                  int S, long count, T* __restrict__ out) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < count; i += (long)gridDim.x * 256L) {
        T de = slabs[i], nu = slabs[count + i];
        for (int s = 1; s < S; ++s) {
            de = de + slabs[(long)s * stride + i];
            nu = nu + slabs[(long)s * stride + count + i];
        }
        out[i] = cur[i] * max_np(nu, T(0)) / max_np(de, T(1.0e-15));
    }
}

// One workgroup per row of U[K, F]:  out = U / sqrt(sum |U|^2)  (strict) or
// U / sqrt(max(sum |U|^2, 1)).  Optionally block-max of |ref - out| into rowmax[row].
template <class T>
__global__ void __launch_bounds__(256) row_normalize_kernel(const T* __restrict__ U, long ld_u,
                                                            long F, int strict,
                                                            const T* __restrict__ ref, long ld_ref,
                                                            T* __restrict__ out, long ld_out,
                                                            real_t<T>* __restrict__ rowmax,
                                                            real_t<T>* __restrict__ norm_out = nullptr,
                                                            real_t<T>* __restrict__ gmax = nullptr,
                                                            real_t<T>* __restrict__ gmax_zero = nullptr,
                                                            unsigned int* __restrict__ ticket = nullptr,
                                                            real_t<T>* __restrict__ host_out = nullptr) {
    typedef real_t<T> R;
    __shared__ R sh[4];
    __shared__ R s_inv;
    const long row = blockIdx.x;
    const T* u = U + row * ld_u;
    // rows of up to 4096 elements stay in registers between the two passes (one HBM/L2 round
    // trip instead of two: this kernel sits on the replicated, latency-bound tail of a step)
    constexpr int CACHE = 16;
    const bool cached = F <= 256L * CACHE;
    T rc[CACHE];
    R acc = 0;
    if (cached) {
#pragma unroll
        for (int q = 0; q < CACHE; ++q) {
            const long j = threadIdx.x + 256L * q;
            rc[q] = (j < F) ? u[j] : zero_of<T>();
            acc += abs2(rc[q]);
        }
    } else {
        for (long j = threadIdx.x; j < F; j += 256) acc += abs2(u[j]);
    }
    R tot = block_sum_256(acc, sh);
    if (threadIdx.x == 0) {
        if (!strict) tot = tot > R(1) ? tot : R(1);
        s_inv = sqrt(tot);
        if (norm_out != nullptr) norm_out[row] = s_inv;
    }
    __syncthreads();
    const R nrm = s_inv;
    R md = 0;
    auto emit = [&](long j, T v) {
        T o;  // a true division, as the reference's U / sqrt(.)
        if constexpr (scalar_traits<T>::is_complex) {
            o.re = v.re / nrm;
            o.im = v.im / nrm;
        } else {
            o = v / nrm;
        }
        if (ref != nullptr) {
            const R d = absval(sub(ref[row * ld_ref + j], o));
            md = (d > md || d != d) ? d : md;
        }
        out[row * ld_out + j] = o;
    };
    if (cached) {
#pragma unroll
        for (int q = 0; q < CACHE; ++q) {
            const long j = threadIdx.x + 256L * q;
            if (j < F) emit(j, rc[q]);
        }
    } else {
        for (long j = threadIdx.x; j < F; j += 256) emit(j, u[j]);
    }
    if (rowmax != nullptr || gmax != nullptr) {
        R m = block_max_256(md, sh);
        if (threadIdx.x == 0) {
            if (rowmax != nullptr) rowmax[row] = m;
            // global max without a second launch: |.| >= 0, so the IEEE bit pattern is
            // monotone in the value and a NaN (0x7fc..) wins, as np.max would have it.
            // *gmax must be zero on entry; the other slot is cleared for the next iteration.
            // With a ticket the max is a RETURNING atomic and the ticket's increment is made to depend on the
            // returned value: the max has been performed at the memory side before the arrival is counted, without
            // a __threadfence (an L2 write-back on this part: ~2 us of a 12 us kernel).
            unsigned int inc = 1u;
            if (gmax != nullptr) {
                if (ticket != nullptr) {
                    R old = atomic_max_nonneg_ret(gmax, m);
                    asm volatile("; the arrival is counted behind the max" : "+v"(inc) : "v"(old));
                } else {
                    atomic_max_nonneg(gmax, m);
                }
            }
            if (gmax_zero != nullptr && row == 0) *gmax_zero = R(0);
            // The workgroup that arrives LAST (a ticket per row, no waiting) publishes the finished maximum to
            // device-mapped pinned host memory, where the host polls for it: no copy kernel and no event in between
            // (4.2 us + a launch boundary + ~6 us of barrier packet per MU iteration).  The store needs no system
            // fence -- it is written through to the fabric, and the kernel's end releases it at the latest.
            // *ticket must be zero on entry and is again on exit.
            if (ticket != nullptr && gmax != nullptr) {
                if (atomicAdd(ticket, inc) == gridDim.x - 1u) {
                    *host_out = atomic_read_nonneg(gmax);
                    atomicExch(ticket, 0u);
                }
            }
        }
    }
}

// Single block: out[0] = max_i v[i] (v >= 0); NaN propagates (np.max semantics).
template <class R>
__global__ void __launch_bounds__(256) final_max_kernel(const R* __restrict__ v, long n,
                                                        R* __restrict__ out) {
    __shared__ R sh[4];
    R m = 0;
    for (long i = threadIdx.x; i < n; i += 256) {
        const R x = v[i];
        m = (x > m || x != x) ? x : m;
    }
    R r = block_max_256(m, sh);
    if (threadIdx.x == 0) out[0] = r;
}

// count of elements failing `x >= 0` (so NaN counts, exactly as assertion.py:99-100 fails on
// it); two stage, deterministic: partial[block], then summed by the caller.
template <class T>
__global__ void __launch_bounds__(256) count_negative_kernel(const T* __restrict__ x, long n,
                                                             unsigned long long* __restrict__ partial) {
    __shared__ double sh[4];
    double c = 0;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L)
        c += (x[i] >= T(0)) ? 0.0 : 1.0;
    double t = block_sum_256(c, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = (unsigned long long)t;
}

// Column sums of a[rows, cols] over rows: stage 1 partial[b, c] over a row stripe,
// stage 2 = reduce_slabs_kernel.  (KL denominators: grads.py:146,155.)
template <class T>
__global__ void __launch_bounds__(256) colsum_partial_kernel(const T* __restrict__ a, long ld,
                                                             long rows, long cols, long rows_per_blk,
                                                             T* __restrict__ partial) {
    const long r0 = blockIdx.y * rows_per_blk;
    const long r1 = min(rows, r0 + rows_per_blk);
    for (long c = blockIdx.x * 256L + threadIdx.x; c < cols; c += (long)gridDim.x * 256L) {
        // four independent running sums: the loop is load-latency bound with one
        T a0 = zero_of<T>(), a1 = zero_of<T>(), a2 = zero_of<T>(), a3 = zero_of<T>();
        long r = r0;
        for (; r + 3 < r1; r += 4) {
            a0 = add(a0, a[r * ld + c]);
            a1 = add(a1, a[(r + 1) * ld + c]);
            a2 = add(a2, a[(r + 2) * ld + c]);
            a3 = add(a3, a[(r + 3) * ld + c]);
        }
        for (; r < r1; ++r) a0 = add(a0, a[r * ld + c]);
        partial[blockIdx.y * cols + c] = add(add(a0, a1), add(a2, a3));
    }
}

// Row sums of a[rows, cols]: one block per row.
template <class T>
__global__ void __launch_bounds__(256) rowsum_kernel(const T* __restrict__ a, long ld, long cols,
                                                     T* __restrict__ out) {
    __shared__ T sh[4];
    const long row = blockIdx.x;
    T acc = 0;
    for (long j = threadIdx.x; j < cols; j += 256) acc += a[row * ld + j];
    T t = block_sum_256(acc, sh);
    if (threadIdx.x == 0) out[row] = t;
}

// sum of squares in double -> partial[block] (residual norm).
template <class T>
__global__ void __launch_bounds__(256) sumsq_partial_kernel(const T* __restrict__ a, long n,
                                                            double* __restrict__ partial) {
    __shared__ double sh[4];
    double c = 0;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L)
        c += (double)abs2(a[i]);
    double t = block_sum_256(c, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

inline int grid_for(long n, int cap = 2048) {
    long g = (n + 255) / 256;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

}  // namespace dcp
