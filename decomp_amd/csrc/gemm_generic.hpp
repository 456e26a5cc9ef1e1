// Generic LDS-tiled GEMM: the fallback of the front end (gemm.hpp) for products the MFMA cores do
// not take -- float64 / complex128 outputs thinner than 128 x 128, complex products without an
// extended-operand scratch -- and the cross-check of the MFMA cores in the tests (tile code 2).
//
//   C(m, n) = sum_k opA(A(m, k)) * opB(B(k, n)),  op = identity or conj
//
// Arbitrary element strides: A(m, k) at A[m*sAm + k*sAk], B(k, n) at
// B[k*sBk + n*sBn].  64x64 tile, BK = 16, 256 threads, 4x4 outputs per thread,
// fully bounds-checked (no alignment or divisibility requirement), split-K as in
// the MFMA kernel.  VALU only.
#pragma once
#include <hip/hip_runtime.h>
#include "scalar.hpp"

namespace dcp {

template <class T>
struct GenericProblem {
    const T* A;
    long sAm, sAk;
    const T* B;
    long sBk, sBn;
    const T* B2;  // optional second column segment (columns >= n_b1)
    long sB2k, sB2n;
    int n_b1;
    int M, N, K;
    int ksplits, klen;
    int tiles_m, tiles_n;
    int tiles_n1;  // n tiles of the first B segment (tiles never straddle the seam)
};

template <class T, bool CONJA, bool CONJB, class Epi>
__global__ void __launch_bounds__(256) gemm_generic_kernel(GenericProblem<T> p, Epi epi) {
    constexpr int BM = 64, BN = 64, BK = 16;
    __shared__ T sA[BK][BM + 1];
    __shared__ T sB[BK][BN + 1];
    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;
    const int tiles = p.tiles_m * p.tiles_n;
    const int split = blockIdx.x / tiles;
    const int t = blockIdx.x - split * tiles;
    const int mt = t % p.tiles_m, nt = t / p.tiles_m;
    const int m0 = mt * BM;
    const int kbeg = split * p.klen;
    const int kend = min(p.K, kbeg + p.klen);

    const T* Bp = p.B;
    long sBk = p.sBk, sBn = p.sBn;
    int nB0 = nt * BN, nBlim = p.n_b1, n0 = nB0, ncol_end = p.n_b1;
    if (nt >= p.tiles_n1) {
        Bp = p.B2;
        sBk = p.sB2k;
        sBn = p.sB2n;
        nB0 = (nt - p.tiles_n1) * BN;
        nBlim = p.N - p.n_b1;
        n0 = p.n_b1 + nB0;
        ncol_end = p.N;
    }

    T acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = zero_of<T>();

    const bool a_kfast = (p.sAk == 1);
    const bool b_kfast = (sBk == 1);
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + i * 256;
            int m, k;
            if (a_kfast) { k = idx & 15; m = idx >> 4; } else { m = idx & 63; k = idx >> 6; }
            T v = zero_of<T>();
            if ((m0 + m) < p.M && (k0 + k) < kend)
                v = p.A[(long)(m0 + m) * p.sAm + (long)(k0 + k) * p.sAk];
            sA[k][m] = CONJA ? conj_of(v) : v;
            int n;
            if (b_kfast) { k = idx & 15; n = idx >> 4; } else { n = idx & 63; k = idx >> 6; }
            T w = zero_of<T>();
            if ((nB0 + n) < nBlim && (k0 + k) < kend)
                w = Bp[(long)(k0 + k) * sBk + (long)(nB0 + n) * sBn];
            sB[k][n] = CONJB ? conj_of(w) : w;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < BK; ++k) {
            T a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = sA[k][ty + 16 * i];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = sB[k][tx + 16 * j];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = madd(acc[i][j], a[i], b[j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = m0 + ty + 16 * i;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = n0 + tx + 16 * j;
            if (row < p.M && col < ncol_end) epi(row, col, acc[i][j], split);
        }
    }
}

template <class T, class Epi>
inline hipError_t launch_gemm_generic(hipStream_t stream, GenericProblem<T> p, bool conjA,
                                      bool conjB, const Epi& epi) {
    if (p.B2 == nullptr) p.n_b1 = p.N;
    p.tiles_m = (p.M + 63) / 64;
    p.tiles_n1 = (p.n_b1 + 63) / 64;
    p.tiles_n = p.tiles_n1 + (p.N - p.n_b1 + 63) / 64;
    if (p.ksplits < 1) p.ksplits = 1;
    if (p.ksplits == 1) p.klen = p.K > 0 ? p.K : 1;
    const int grid = p.tiles_m * p.tiles_n * p.ksplits;
    if (grid <= 0) return hipSuccess;
    if (conjA && conjB)
        hipLaunchKernelGGL((gemm_generic_kernel<T, true, true, Epi>), dim3(grid), dim3(256), 0,
                           stream, p, epi);
    else if (conjA)
        hipLaunchKernelGGL((gemm_generic_kernel<T, true, false, Epi>), dim3(grid), dim3(256), 0,
                           stream, p, epi);
    else if (conjB)
        hipLaunchKernelGGL((gemm_generic_kernel<T, false, true, Epi>), dim3(grid), dim3(256), 0,
                           stream, p, epi);
    else
        hipLaunchKernelGGL((gemm_generic_kernel<T, false, false, Epi>), dim3(grid), dim3(256), 0,
                           stream, p, epi);
    return hipGetLastError();
}

}  // namespace dcp
