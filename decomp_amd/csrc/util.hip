// C ABI: handle lifetime, row normalisation, sign scan, and the GEMM test hook.
#include "comm.hpp"
#include "gemm.hpp"
#include "handle.hpp"
#include "kernels_small.hpp"
#include <new>

#define DCP_STR2(x) #x
#define DCP_STR(x) DCP_STR2(x)

using namespace dcp;

namespace {

template <class T>
int l2_normalize_api(dcp_handle* h, T* U, int64_t K, int64_t F, int strict) {
    if (!h) return DCP_ERR_INVALID;
    if (!U) return fail(h, DCP_ERR_INVALID, "null array pointer");
    if (K < 0 || F < 0) return fail(h, DCP_ERR_INVALID, "negative size");
    if (K == 0 || F == 0) return DCP_OK;
    DCP_HIP_OK(h, hipSetDevice(h->device));
    hipLaunchKernelGGL((row_normalize_kernel<T>), dim3((unsigned)K), dim3(256), 0, h->stream,
                       (const T*)U, (long)F, (long)F, strict, (const T*)nullptr, 0L, U, (long)F,
                       (real_t<T>*)nullptr);
    DCP_HIP_OK(h, hipGetLastError());
    return DCP_OK;
}

template <class T>
int count_negative_api(dcp_handle* h, const T* x, int64_t n, int64_t* count) {
    if (!h) return DCP_ERR_INVALID;
    if (!count || (!x && n > 0)) return fail(h, DCP_ERR_INVALID, "null pointer");
    if (n < 0) return fail(h, DCP_ERR_INVALID, "negative size");
    *count = 0;
    if (n == 0) return DCP_OK;
    DCP_HIP_OK(h, hipSetDevice(h->device));
    const int blocks = grid_for(n, 1024);
    WsPlan plan;
    plan.add<unsigned long long>(blocks);
    DCP_TRY(ws_reserve(h, plan.total));
    ws_reset(h);
    unsigned long long* part = ws_alloc<unsigned long long>(h, blocks);
    if (!part) return fail(h, DCP_ERR_INTERNAL, "workspace plan mismatch");
    void* hostv = nullptr;
    DCP_TRY(host_scratch(h, sizeof(unsigned long long) * blocks, &hostv));
    hipLaunchKernelGGL((count_negative_kernel<T>), dim3(blocks), dim3(256), 0, h->stream, x,
                       (long)n, part);
    DCP_HIP_OK(h, hipGetLastError());
    DCP_HIP_OK(h, hipMemcpyAsync(hostv, part, sizeof(unsigned long long) * blocks,
                                 hipMemcpyDeviceToHost, h->stream));
    DCP_HIP_OK(h, hipStreamSynchronize(h->stream));
    unsigned long long tot = 0;
    for (int i = 0; i < blocks; ++i) tot += reinterpret_cast<unsigned long long*>(hostv)[i];
    *count = (int64_t)tot;
    return DCP_OK;
}

// extra tile shapes, reachable through the test hook for tuning sweeps (tile codes 3..9)
typedef TileCfg<128, 256, 16, 64, 128, 2> CfgWideNT;     // 3 (NT, NN)
typedef TileCfg<256, 128, 16, 128, 64, 2> CfgWideTN;     // 3 (TN)
typedef TileCfg<128, 128, 32, 64, 64, 2> CfgDeep;        // 4: BK = 32
typedef TileCfg<128, 128, 16, 64, 32, 2> Cfg8w;          // 5: 8 waves, 64x32 per wave
typedef TileCfg<256, 128, 16, 64, 64, 2> Cfg8wBig;       // 6: 8 waves, 256x128
typedef TileCfg<256, 256, 16, 64, 64, 1> Cfg16w;         // 7: 16 waves, 256x256
typedef TileCfg<128, 128, 16, 32, 64, 2> Cfg8wT;         // 8: 8 waves, 32x64 per wave
typedef TileCfg<256, 128, 32, 64, 64, 1> Cfg8wBigDeep;   // 9: 8 waves, 256x128, BK = 32
typedef TileCfg<256, 256, 32, 64, 64, 1> Cfg16wDeep;     // 10: 16 waves, 256x256, BK = 32 (128 KiB dynamic LDS)
typedef TileCfg<256, 256, 32, 128, 64, 1> Cfg8wHuge;      // 11: 8 waves of 128x64, 256x256, BK = 32
typedef TileCfg<256, 256, 16, 64, 64, 1, 1> Cfg16wPipe;    // 12: 16 waves, 256x256, barrier between MFMA groups
typedef TileCfg<128, 128, 16, 64, 64, 2, 1> CfgLargePipe;  // 13: 4 waves, 128x128, same
typedef TileCfg<256, 256, 32, 64, 64, 1, 1> Cfg16wDeepPipe; // 14: 16 waves, BK = 32, same

// small tiles with deep K blocks: shard-size products (few rows per GPU) and the K = 256 D-side products
typedef TileCfg<64, 128, 64, 32, 32, 2> CfgRow64x128d64;   // 15: 8 waves of 32x32, BK = 64 (96 KiB LDS)
typedef TileCfg<64, 128, 32, 32, 32, 2> CfgRow64x128d32;   // 16: same, BK = 32 (48 KiB)
typedef TileCfg<64, 64, 64, 32, 32, 1> CfgSmallD64;        // 17: 4 waves of 32x32, BK = 64 (64 KiB)
typedef TileCfg<64, 64, 32, 32, 32, 2> CfgSmallD32;        // 18: 4 waves, BK = 32 (32 KiB)
typedef TileCfg<128, 64, 64, 32, 32, 2> CfgRow128x64d64;   // 19: 8 waves of 32x32 (4 x 2), BK = 64
typedef TileCfg<128, 128, 32, 32, 64, 2> CfgLarge8wD32;    // 20: 8 waves of 32x64, BK = 32 (64 KiB)
typedef TileCfg<128, 128, 64, 32, 64, 2> CfgLarge8wD64;    // 21: 8 waves of 32x64, BK = 64 (128 KiB)
typedef TileCfg<64, 256, 32, 32, 64, 2> CfgRow64x256d32;   // 22: 8 waves of 32x64 (2 x 4), BK = 32 (80 KiB)
// measured and NOT adopted (kept reachable for A/B runs, tools/mf16_ab.sh, tools/ahead_ab.sh):
typedef TileCfg<256, 256, 16, 64, 64, 1, 0, 16> Cfg16wMf16;    // 23: 256x256 / 16 waves on v_mfma_f32_16x16x4_f32
typedef TileCfg<128, 128, 16, 64, 64, 2, 0, 16> CfgLargeMf16;  // 24: 128x128 / 4 waves on 16x16x4
typedef TileCfg<256, 256, 16, 64, 64, 1, 2> Cfg16wAhead;       // 25: all fragment reads of a K block up front
typedef TileCfg<128, 128, 16, 64, 64, 2, 2> CfgLargeAhead;     // 26: same, 128x128 / 4 waves

// narrow outputs (<= 64 atoms): 64-wide tiles (NT / NN) and 64-tall tiles (TN)
typedef TileCfg<128, 64, 16, 32, 64, 2> CfgTall128;     // 27: 4 waves of 32x64
typedef TileCfg<256, 64, 16, 64, 64, 2> CfgTall256;     // 28: 4 waves of 64x64
typedef TileCfg<64, 128, 16, 64, 32, 2> CfgFlat128;     // 29: 4 waves of 64x32 (TN: 64 output rows)
typedef TileCfg<64, 256, 16, 64, 64, 2> CfgFlat256;     // 30: 4 waves of 64x64

typedef TileCfg<256, 32, 32, 64, 32, 2> CfgTall32;      // 31: 4 waves of 64x32, BK = 32 (<= 32 atoms)
typedef TileCfg<128, 32, 32, 32, 32, 2> CfgTall32s;     // 32: 4 waves of 32x32, BK = 32
typedef TileCfg<32, 256, 32, 32, 64, 2> CfgFlat32;      // 33: 4 waves of 32x64, BK = 32 (TN, <= 32 atoms)
typedef TileCfg<32, 128, 32, 32, 32, 2> CfgFlat32s;     // 34: 4 waves of 32x32, BK = 32

template <class Cfg, int FORM, class Epi>
hipError_t hook_cfg(hipStream_t st, const GemmArgs<float>& a, const Epi& epi) {
    GemmProblem p;
    p.A = a.A; p.lda = a.lda; p.B = a.B; p.ldb = a.ldb;
    p.B2 = nullptr; p.ldb2 = 0; p.n_b1 = a.N;
    p.M = a.M; p.N = a.N; p.K = a.K; p.ksplits = a.ksplits; p.klen = a.klen;
    p.tiles_m = p.tiles_n = 0;
    p.mt_fast = (FORM == FORM_NT) ? 0 : 1;
    constexpr int AL = (FORM == FORM_TN) ? XMAJOR : KMAJOR;
    constexpr int BL = (FORM == FORM_NT) ? KMAJOR : XMAJOR;
    return launch_gemm_mfma<Cfg, AL, BL, Epi>(st, p, epi);
}

// fp64 tile experiments (test hook only): codes 3..6
typedef F64Cfg<256, 128, 32, 64, 1> F64Big;      // 3: 16 waves, 256 x 128
typedef F64Cfg<128, 128, 64, 64, 1> F64Fat;      // 4: 4 waves of 64 x 64 (128 accumulator registers)
typedef F64Cfg<128, 256, 32, 64, 1> F64Wide;     // 5: 16 waves, 128 x 256
typedef F64Cfg<128, 128, 32, 64, 2> F64Thin;     // 6: 8 waves of 32 x 64, two workgroups / CU

template <int FORM, class Epi>
hipError_t gemm_hook_launch_f64(hipStream_t st, const GemmArgs<double>& a, int tile, const Epi& epi) {
    if (tile >= 3 && tile <= 6) {
        GemmProblemD p;
        p.A = a.A; p.lda = a.lda; p.B = a.B; p.ldb = a.ldb;
        p.B2 = nullptr; p.ldb2 = 0; p.n_b1 = a.N;
        p.M = a.M; p.N = a.N; p.K = a.K;
        p.ksplits = a.ksplits; p.klen = a.klen;
        p.tiles_m = p.tiles_n = 0;
        p.mt_fast = (FORM == FORM_NT) ? 0 : 1;
        constexpr int AL = (FORM == FORM_TN) ? XMAJOR : KMAJOR;
        constexpr int BL = (FORM == FORM_NT) ? KMAJOR : XMAJOR;
        if (tile == 3) return launch_gemm_mfma_f64_cfg<F64Big, AL, BL, Epi>(st, p, epi);
        if (tile == 4) return launch_gemm_mfma_f64_cfg<F64Fat, AL, BL, Epi>(st, p, epi);
        if (tile == 5) return launch_gemm_mfma_f64_cfg<F64Wide, AL, BL, Epi>(st, p, epi);
        return launch_gemm_mfma_f64_cfg<F64Thin, AL, BL, Epi>(st, p, epi);
    }
    return gemm<FORM>(st, a, epi);
}

template <int FORM, class T, class Epi>
hipError_t gemm_hook_launch(hipStream_t st, const GemmArgs<T>& a, int tile, const Epi& epi) {
    if constexpr (std::is_same<T, double>::value) return gemm_hook_launch_f64<FORM>(st, a, tile, epi);
    if constexpr (std::is_same<T, float>::value) {
        switch (tile) {
            case 3:
                if (FORM == FORM_TN) return hook_cfg<CfgWideTN, FORM>(st, a, epi);
                return hook_cfg<CfgWideNT, FORM>(st, a, epi);
            case 4: return hook_cfg<CfgDeep, FORM>(st, a, epi);
            case 5: return hook_cfg<Cfg8w, FORM>(st, a, epi);
            case 6: return hook_cfg<Cfg8wBig, FORM>(st, a, epi);
            case 7: return hook_cfg<Cfg16w, FORM>(st, a, epi);
            case 8: return hook_cfg<Cfg8wT, FORM>(st, a, epi);
            case 9: return hook_cfg<Cfg8wBigDeep, FORM>(st, a, epi);
            case 10: return hook_cfg<Cfg16wDeep, FORM>(st, a, epi);
            case 11: return hook_cfg<Cfg8wHuge, FORM>(st, a, epi);
            case 12: return hook_cfg<Cfg16wPipe, FORM>(st, a, epi);
            case 13: return hook_cfg<CfgLargePipe, FORM>(st, a, epi);
            case 14: return hook_cfg<Cfg16wDeepPipe, FORM>(st, a, epi);
            case 15: return hook_cfg<CfgRow64x128d64, FORM>(st, a, epi);
            case 16: return hook_cfg<CfgRow64x128d32, FORM>(st, a, epi);
            case 17: return hook_cfg<CfgSmallD64, FORM>(st, a, epi);
            case 18: return hook_cfg<CfgSmallD32, FORM>(st, a, epi);
            case 19: return hook_cfg<CfgRow128x64d64, FORM>(st, a, epi);
            case 20: return hook_cfg<CfgLarge8wD32, FORM>(st, a, epi);
            case 21: return hook_cfg<CfgLarge8wD64, FORM>(st, a, epi);
            case 22: return hook_cfg<CfgRow64x256d32, FORM>(st, a, epi);
            case 23: return hook_cfg<Cfg16wMf16, FORM>(st, a, epi);
            case 24: return hook_cfg<CfgLargeMf16, FORM>(st, a, epi);
            case 25: return hook_cfg<Cfg16wAhead, FORM>(st, a, epi);
            case 26: return hook_cfg<CfgLargeAhead, FORM>(st, a, epi);
            case 27: return hook_cfg<CfgTall128, FORM>(st, a, epi);
            case 28: return hook_cfg<CfgTall256, FORM>(st, a, epi);
            case 29: return hook_cfg<CfgFlat128, FORM>(st, a, epi);
            case 30: return hook_cfg<CfgFlat256, FORM>(st, a, epi);
            case 31: return hook_cfg<CfgTall32, FORM>(st, a, epi);
            case 32: return hook_cfg<CfgTall32s, FORM>(st, a, epi);
            case 33: return hook_cfg<CfgFlat32, FORM>(st, a, epi);
            case 34: return hook_cfg<CfgFlat32s, FORM>(st, a, epi);
            default: break;
        }
    }
    return gemm<FORM>(st, a, epi);
}

template <class T>
int gemm_api(dcp_handle* h, int form, const T* A, const T* B, T* C, int64_t M, int64_t N,
             int64_t K, int ksplits, int tile) {
    if (!h) return DCP_ERR_INVALID;
    if (!A || !B || !C) return fail(h, DCP_ERR_INVALID, "null array pointer");
    if (M <= 0 || N <= 0 || K <= 0) return fail(h, DCP_ERR_INVALID, "sizes must be positive");
    if (form < 0 || form > 2) return fail(h, DCP_ERR_INVALID, "bad form");
    DCP_HIP_OK(h, hipSetDevice(h->device));
    GemmArgs<T> a;
    a.A = A; a.B = B; a.M = (int)M; a.N = (int)N; a.K = (int)K;
    a.lda = (form == FORM_TN) ? M : K;
    a.ldb = (form == FORM_NT) ? K : N;
    if constexpr (scalar_traits<T>::is_complex) {   // the hook computes A B^H, A B, A^H B
        a.conjB = (form == FORM_NT);
        a.conjA = (form == FORM_TN);
    }
    typedef real_t<T> RT;
    RT* ext = nullptr;
    const size_t ext_floats = scalar_traits<T>::is_complex ? (size_t)4 * N * K : 0;
    a.tile = (tile >= 3) ? TILE_LARGE : tile;   // hook-only shapes plan splits as 128x128
    hipError_t e = hipSuccess;
    if (ksplits > 1) {
        const long kblocks = (K + 15) / 16;
        long s = ksplits > kblocks ? kblocks : ksplits;
        a.klen = (int)(((kblocks + s - 1) / s) * 16);
        a.ksplits = (int)((K + a.klen - 1) / a.klen);
        // slabs for the larger of the two plans (complex products may re-plan over the real-extended 2K below)
        const size_t max_slabs = (size_t)(ksplits > a.ksplits ? ksplits : a.ksplits) + 1;
        WsPlan plan;
        plan.add<T>(max_slabs * M * N);
        plan.add<RT>(ext_floats + 4);
        DCP_TRY(ws_reserve(h, plan.total));
        ws_reset(h);
        T* slabs = ws_alloc<T>(h, max_slabs * M * N);
        ext = ws_alloc<RT>(h, ext_floats + 4);
        if (!slabs || !ext) return fail(h, DCP_ERR_INTERNAL, "workspace plan mismatch");
        if (ext_floats) {
            a.ext_ws = ext;
            bool planar = form == FORM_NN && cplx_planar_a<FORM_NN>(a.M, a.N, a.conjA, a.conjB, a.ext_ws);
            if (std::is_same<T, c128>::value) planar = planar && f64_tier(2 * a.M, 2 * a.N, a.tile) != F64_GENERIC;
            if (form != FORM_TN && !planar) {   // the MFMA core splits the real-extended reduction (2K)
                const long kb2 = (2 * K + 15) / 16;
                long s2 = ksplits > kb2 ? kb2 : ksplits;
                a.klen = (int)(((kb2 + s2 - 1) / s2) * 16);
                a.ksplits = (int)((2 * K + a.klen - 1) / a.klen);
                if ((size_t)a.ksplits > max_slabs) return fail(h, DCP_ERR_INTERNAL, "hook slab plan");
            }
        }
        EpiSlab<T> epi{slabs, (long)N, (long)M * N};
        if (form == FORM_NT) e = gemm_hook_launch<FORM_NT>(h->stream, a, tile, epi);
        else if (form == FORM_NN) e = gemm_hook_launch<FORM_NN>(h->stream, a, tile, epi);
        else e = gemm_hook_launch<FORM_TN>(h->stream, a, tile, epi);
        if (e != hipSuccess) return fail(h, DCP_ERR_HIP, hipGetErrorString(e));
        hipLaunchKernelGGL((reduce_slabs_kernel<T>), dim3(grid_for(M * N)), dim3(256), 0, h->stream,
                           slabs, (long)(M * N), a.ksplits, (long)(M * N), C);
        DCP_HIP_OK(h, hipGetLastError());
    } else {
        if (ext_floats) {
            WsPlan plan;
            plan.add<RT>(ext_floats + 4);
            DCP_TRY(ws_reserve(h, plan.total));
            ws_reset(h);
            ext = ws_alloc<RT>(h, ext_floats + 4);
            if (!ext) return fail(h, DCP_ERR_INTERNAL, "workspace plan mismatch");
            a.ext_ws = ext;
        }
        EpiStore<T> epi{C, (long)N};
        if (form == FORM_NT) e = gemm_hook_launch<FORM_NT>(h->stream, a, tile, epi);
        else if (form == FORM_NN) e = gemm_hook_launch<FORM_NN>(h->stream, a, tile, epi);
        else e = gemm_hook_launch<FORM_TN>(h->stream, a, tile, epi);
        if (e != hipSuccess) return fail(h, DCP_ERR_HIP, hipGetErrorString(e));
    }
    return DCP_OK;
}

// PMC calibration aid (MI355X_MICROARCH.md, HBM section: "calibrate on a known byte count in your
// own access pattern"): reads a [rows, cols] float matrix ONCE with exactly the global-load shape
// of the KMAJOR panel loader (pattern 0: each wave instruction = 16 rows x 64 B, 16 floats of K per
// step) or of the XMAJOR loader (pattern 1: 512-B contiguous row segments), summing into `out`.
__global__ void __launch_bounds__(256) calib_read_kernel(const float* __restrict__ p, long rows,
                                                         long cols, int pattern,
                                                         float* __restrict__ out) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int tid = threadIdx.x;
    if (pattern == 0) {
        const long r0 = (long)blockIdx.x * 128;           // a 128-row panel per workgroup
        for (long k0 = 0; k0 + 16 <= cols; k0 += 16)
            for (int i = 0; i < 2; ++i) {
                const int idx = tid + i * 256;
                const long row = r0 + idx / 4;
                if (row < rows)
                    acc += *reinterpret_cast<const f32x4*>(p + row * cols + k0 + (idx % 4) * 4);
            }
    } else {
        const long c0 = (long)blockIdx.x * 128;           // a 128-column stripe per workgroup
        for (long r = tid / 32; r < rows; r += 8)
            if (c0 + (tid % 32) * 4 + 4 <= cols)
                acc += *reinterpret_cast<const f32x4*>(p + r * cols + c0 + (tid % 32) * 4);
    }
    const float s = acc[0] + acc[1] + acc[2] + acc[3];
    if (s == 12345.678f) out[0] = s;   // keeps the loads alive, (almost) never stores
}

}  // namespace

// Row mover for the minibatch containers (decomp/utils/data.py:124-156, 214-313): dtype-agnostic,
//   out[(out_index ? out_index[i] : i), :] = in[(in_index ? in_index[i] : i), :],  i < rows.
// Either side may be PINNED HOST memory (the out-of-core container gathers a shuffled minibatch
// straight over PCIe, 16 bytes per lane, no host-side shuffle pass).  One workgroup per
// (row, 16 KiB chunk); rows whose byte length or base is not 16-byte aligned take the byte loop.
template <int GROUPS>
__global__ void __launch_bounds__(256 * GROUPS) move_rows_kernel(const unsigned char* __restrict__ in,
                                                                 const long long* __restrict__ in_index,
                                                                 unsigned char* __restrict__ out,
                                                                 const long long* __restrict__ out_index,
                                                                 long rows, long row_bytes, int chunks,
                                                                 int vec_ok) {
    // GROUPS > 1 (PCIe mover): the launch asks for the CU's whole LDS so that no compute
    // workgroup shares the CU -- see move_rows() below.  The array is never touched.
    extern __shared__ unsigned char move_rows_lds[];
    const int tid = threadIdx.x & 255;
    const long group = (long)blockIdx.x * GROUPS + (threadIdx.x >> 8);
    const long ngroups = (long)gridDim.x * GROUPS;
    if (vec_ok) {
        // the matrix as a stream of 16-byte pieces; a unit = 1024 consecutive pieces (16 KiB,
        // possibly several short rows), 4 pieces per lane, all loads issued before the stores
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        const long P = row_bytes >> 4;              // pieces per row
        const long total = rows * P;
        const long units = (total + 1023) >> 10;
        for (long unit = group; unit < units; unit += ngroups) {
            u32x4 v[4];
            long dst_off[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long g = (unit << 10) + tid + u * 256;
                dst_off[u] = -1;
                if (g < total) {
                    const long i = g / P;
                    const long o = (g - i * P) << 4;
                    const long src_row = in_index ? (long)in_index[i] : i;
                    const long dst_row = out_index ? (long)out_index[i] : i;
                    v[u] = *reinterpret_cast<const u32x4*>(in + src_row * row_bytes + o);
                    dst_off[u] = dst_row * row_bytes + o;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (dst_off[u] >= 0) *reinterpret_cast<u32x4*>(out + dst_off[u]) = v[u];
        }
    } else {
        for (long wg = group; wg < rows * chunks; wg += ngroups) {
            const long i = wg / chunks;
            const int ch = (int)(wg - i * chunks);
            const long src_row = in_index ? (long)in_index[i] : i;
            const long dst_row = out_index ? (long)out_index[i] : i;
            const unsigned char* s = in + src_row * row_bytes;
            unsigned char* d = out + dst_row * row_bytes;
            const long b0 = (long)ch * 16384;
            const long b1 = min(row_bytes, b0 + 16384);
            for (long o = b0 + tid; o < b1; o += 256) d[o] = s[o];
        }
    }
}

// Is p ordinary device memory?
static bool is_device_memory(const void* p) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();
        return true;
    }
    return a.type == hipMemoryTypeDevice;
}

namespace dcp {
int move_rows_on(dcp_handle* h, hipStream_t stream, const void* in, const int64_t* in_index, void* out,
                 const int64_t* out_index, int64_t rows, int64_t row_bytes) {
    if (!h) return DCP_ERR_INVALID;
    if (!in || !out) return dcp::fail(h, DCP_ERR_INVALID, "null pointer");
    if (rows < 0 || row_bytes < 0) return dcp::fail(h, DCP_ERR_INVALID, "negative size");
    if (rows == 0 || row_bytes == 0) return DCP_OK;
    const long chunks = (row_bytes + 16383) / 16384;
    if (rows * chunks > 0x7fffffffL) return dcp::fail(h, DCP_ERR_INVALID, "too many row chunks");
    const int vec_ok = (row_bytes % 16 == 0) && ((reinterpret_cast<uintptr_t>(in) & 15) == 0) &&
                       ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
    DCP_HIP_OK(h, hipSetDevice(h->device));
    const bool on_device = is_device_memory(in) && is_device_memory(out);
    const long want = vec_ok ? (rows * (row_bytes >> 4) + 1023) / 1024 : rows * chunks;
    const unsigned char* inb = reinterpret_cast<const unsigned char*>(in);
    unsigned char* outb = reinterpret_cast<unsigned char*>(out);
    const long long* ii = reinterpret_cast<const long long*>(in_index);
    const long long* oi = reinterpret_cast<const long long*>(out_index);
    if (on_device) {
        const long cap = 8192;
        hipLaunchKernelGGL(move_rows_kernel<1>, dim3((unsigned)(want < cap ? want : cap)), dim3(256), 0,
                           stream, inb, ii, outb, oi, (long)rows, (long)row_bytes, (int)chunks, vec_ok);
    } else {
        // One side is pinned host memory, reached over PCIe.  Waves that sit on PCIe latency clog
        // the memory pipeline of their CU: a compute workgroup sharing that CU runs several times
        // slower, and a one-round GEMM is as slow as its slowest workgroup (seen: 350 us -> 2 ms).
        // So the mover takes a dozen CUs for itself -- 1024 threads and the CU's whole LDS per
        // workgroup keep every other workgroup off them -- which is plenty to saturate the link
        // (12 x 64 KiB in flight) and costs the overlapped compute < 5 % of the chip.
        static DynLdsRaised raised_flag;
        std::atomic<bool>& raised = raised_flag.on_current_device();
        const int lds_bytes = 160 * 1024;
        if (!raised) {
            DCP_HIP_OK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&move_rows_kernel<4>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
            raised = true;
        }
        const long groups = (want + 3) / 4;
        hipLaunchKernelGGL(move_rows_kernel<4>, dim3((unsigned)(groups < 12 ? groups : 12)), dim3(1024),
                           lds_bytes, stream, inb, ii, outb, oi, (long)rows, (long)row_bytes,
                           (int)chunks, vec_ok);
    }
    DCP_HIP_OK(h, hipGetLastError());
    return DCP_OK;
}
}  // namespace dcp

static int move_rows(dcp_handle* h, const void* in, const int64_t* in_index, void* out,
                     const int64_t* out_index, int64_t rows, int64_t row_bytes) {
    if (!h) return DCP_ERR_INVALID;
    return dcp::move_rows_on(h, h->stream, in, in_index, out, out_index, rows, row_bytes);
}

extern "C" {

int dcp_debug_tn_plain(int on) {
    const int prev = dcp::tn_plain_schedule() ? 1 : 0;
    if (on >= 0) dcp::tn_plain_flag().store(on ? 1 : 0, std::memory_order_relaxed);
    return prev;
}

int dcp_calib_read_f32(dcp_handle* h, const float* p, int64_t rows, int64_t cols, int pattern,
                       float* out) {
    if (!h) return DCP_ERR_INVALID;
    if (!p || !out || rows <= 0 || cols <= 0 || cols % 128 != 0)
        return fail(h, DCP_ERR_INVALID, "bad calibration arguments");
    DCP_HIP_OK(h, hipSetDevice(h->device));
    const unsigned grid = pattern == 0 ? (unsigned)((rows + 127) / 128) : (unsigned)(cols / 128);
    hipLaunchKernelGGL(calib_read_kernel, dim3(grid), dim3(256), 0, h->stream, p, (long)rows, (long)cols,
                       pattern, out);
    DCP_HIP_OK(h, hipGetLastError());
    return DCP_OK;
}

int dcp_create(dcp_handle** out, int device) {
    if (!out) return DCP_ERR_INVALID;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return DCP_ERR_HIP;
    if (hipSetDevice(device) != hipSuccess) return DCP_ERR_HIP;
    dcp_handle* h = new (std::nothrow) dcp_handle();
    if (!h) return DCP_ERR_NOMEM;
    h->device = device;
    *out = h;
    return DCP_OK;
}

int dcp_destroy(dcp_handle* h) {
    if (!h) return DCP_OK;
    (void)hipSetDevice(h->device);
    comm_release(h);
    if (h->arena) {
        (void)hipStreamSynchronize(h->stream);
        (void)hipFree(h->arena);
    }
    if (h->host_pinned) (void)hipHostFree(h->host_pinned);
    for (auto& r : h->prof_recs) {
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    for (auto e : h->prof_pool) (void)hipEventDestroy(e);
    if (h->stream_switch) (void)hipEventDestroy(h->stream_switch);
    if (h->side) {
        (void)hipStreamSynchronize(h->side);
        (void)hipStreamDestroy(h->side);
    }
    if (h->ev_main) (void)hipEventDestroy(h->ev_main);
    if (h->ev_side) (void)hipEventDestroy(h->ev_side);
    delete h;
    return DCP_OK;
}

int dcp_set_stream(dcp_handle* h, void* hip_stream) {
    if (!h) return DCP_ERR_INVALID;
    // Only the handle's current stream changes here.  Ordering of the shared workspace arena between
    // the old and the new stream is established lazily by the next call that actually uses the arena
    // (ws_reserve -> ws_order_streams); calls that use none (dcp_gather_rows_bytes / dcp_scatter_rows_bytes
    // on a copy stream) therefore overlap freely with compute enqueued on the other stream.
    h->stream = reinterpret_cast<hipStream_t>(hip_stream);
    return DCP_OK;
}

int dcp_profile_enable(dcp_handle* h, int on) {
    if (!h) return DCP_ERR_INVALID;
    h->prof_on = on != 0;
    return DCP_OK;
}

int dcp_profile_select(dcp_handle* h, unsigned label_mask) {
    if (!h) return DCP_ERR_INVALID;
    h->prof_mask = label_mask;
    return DCP_OK;
}

static int prof_fold(dcp_handle* h) {
    if (h->prof_recs.empty()) return DCP_OK;
    DCP_HIP_OK(h, hipSetDevice(h->device));
    DCP_HIP_OK(h, hipStreamSynchronize(h->stream));
    for (auto& r : h->prof_recs) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess && r.label >= 0 &&
            r.label < DCP_PROF_NLABELS) {
            h->prof_ms[r.label] += ms;
            h->prof_cnt[r.label] += 1;
        }
        h->prof_pool.push_back(r.a);
        h->prof_pool.push_back(r.b);
    }
    h->prof_recs.clear();
    return DCP_OK;
}

int dcp_profile_reset(dcp_handle* h) {
    if (!h) return DCP_ERR_INVALID;
    DCP_TRY(prof_fold(h));
    for (int i = 0; i < DCP_PROF_NLABELS; ++i) {
        h->prof_ms[i] = 0.0;
        h->prof_cnt[i] = 0;
    }
    return DCP_OK;
}

int dcp_profile_read(dcp_handle* h, int label, double* total_ms, int64_t* count) {
    if (!h) return DCP_ERR_INVALID;
    if (label < 0 || label >= DCP_PROF_NLABELS || !total_ms || !count)
        return fail(h, DCP_ERR_INVALID, "bad profile label or null output");
    DCP_TRY(prof_fold(h));
    *total_ms = h->prof_ms[label];
    *count = h->prof_cnt[label];
    return DCP_OK;
}

const char* dcp_profile_label_name(int label) {
    static const char* names[DCP_PROF_NLABELS] = {"gram",  "x_neg",     "x_update", "forward", "stats",
                                                  "stats_sum", "d_update", "d_norm",  "misc", "exchange"};
    return (label >= 0 && label < DCP_PROF_NLABELS) ? names[label] : "?";
}

const char* dcp_last_error_string(dcp_handle* h) { return h ? h->err.c_str() : "null handle"; }

const char* dcp_build_info(void) {
    return "libdecomp_hip: gfx950, fp32 MFMA 32x32x2 + fp64 MFMA 16x16x4 GEMM cores, HIP " DCP_STR(HIP_VERSION_MAJOR) "." DCP_STR(HIP_VERSION_MINOR);
}

int dcp_l2_normalize_f32(dcp_handle* h, float* U, int64_t K, int64_t F, int strict) {
    return l2_normalize_api<float>(h, U, K, F, strict);
}
int dcp_l2_normalize_f64(dcp_handle* h, double* U, int64_t K, int64_t F, int strict) {
    return l2_normalize_api<double>(h, U, K, F, strict);
}
int dcp_l2_normalize_c64(dcp_handle* h, void* U, int64_t K, int64_t F, int strict) {
    return l2_normalize_api<c64>(h, reinterpret_cast<c64*>(U), K, F, strict);
}
int dcp_l2_normalize_c128(dcp_handle* h, void* U, int64_t K, int64_t F, int strict) {
    return l2_normalize_api<c128>(h, reinterpret_cast<c128*>(U), K, F, strict);
}
int dcp_count_negative_f32(dcp_handle* h, const float* x, int64_t n, int64_t* count) {
    return count_negative_api<float>(h, x, n, count);
}
int dcp_count_negative_f64(dcp_handle* h, const double* x, int64_t n, int64_t* count) {
    return count_negative_api<double>(h, x, n, count);
}
int dcp_gemm_f32(dcp_handle* h, int form, const float* A, const float* B, float* C, int64_t M,
                 int64_t N, int64_t K, int ksplits, int tile) {
    return gemm_api<float>(h, form, A, B, C, M, N, K, ksplits, tile);
}
int dcp_gemm_f64(dcp_handle* h, int form, const double* A, const double* B, double* C, int64_t M,
                 int64_t N, int64_t K, int ksplits, int tile) {
    return gemm_api<double>(h, form, A, B, C, M, N, K, ksplits, tile);
}
int dcp_gemm_c64(dcp_handle* h, int form, const void* A, const void* B, void* C, int64_t M, int64_t N,
                 int64_t K, int ksplits, int tile) {
    return gemm_api<c64>(h, form, reinterpret_cast<const c64*>(A), reinterpret_cast<const c64*>(B),
                         reinterpret_cast<c64*>(C), M, N, K, ksplits, tile);
}
int dcp_gemm_c128(dcp_handle* h, int form, const void* A, const void* B, void* C, int64_t M, int64_t N,
                  int64_t K, int ksplits, int tile) {
    return gemm_api<c128>(h, form, reinterpret_cast<const c128*>(A), reinterpret_cast<const c128*>(B),
                          reinterpret_cast<c128*>(C), M, N, K, ksplits, tile);
}

int dcp_dict_prefetch_rows_bytes(dcp_handle* h, const void* in, const int64_t* index, int64_t rows,
                                 int64_t row_bytes, void* out) {
    if (!h) return DCP_ERR_INVALID;
    if (rows < 0 || row_bytes < 0) return dcp::fail(h, DCP_ERR_INVALID, "negative size");
    if (rows > 0 && row_bytes > 0 && (!in || !index || !out)) return dcp::fail(h, DCP_ERR_INVALID, "null pointer");
    h->pf_in = in;
    h->pf_index = index;
    h->pf_out = out;
    h->pf_rows = rows;
    h->pf_row_bytes = row_bytes;
    return DCP_OK;
}

int dcp_gather_rows_bytes(dcp_handle* h, const void* in, const int64_t* index, int64_t rows,
                          int64_t row_bytes, void* out) {
    if (h && !index) return dcp::fail(h, DCP_ERR_INVALID, "null index");
    return move_rows(h, in, index, out, nullptr, rows, row_bytes);
}

int dcp_scatter_rows_bytes(dcp_handle* h, const void* in, const int64_t* index, int64_t rows,
                           int64_t row_bytes, void* out) {
    if (h && !index) return dcp::fail(h, DCP_ERR_INVALID, "null index");
    return move_rows(h, in, nullptr, out, index, rows, row_bytes);
}

int dcp_dict_set_pcd_order(dcp_handle* h, const int32_t* order, int64_t rows, int64_t K) {
    if (!h) return DCP_ERR_INVALID;
    if (order != nullptr && (rows <= 0 || K <= 0)) return dcp::fail(h, DCP_ERR_INVALID, "bad table shape");
    h->pcd_order = order;
    h->pcd_rows = order ? rows : 0;
    h->pcd_K = order ? K : 0;
    return DCP_OK;
}

}  // extern "C"
