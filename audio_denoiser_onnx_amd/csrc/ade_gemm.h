// ade_gemm.h — fp32 GEMM on the MI355X matrix cores with functor operands (shared by the STFT operator and DFSMN).
//
//   C(m, n) = sum_k A(m, k) * B(k, n)        exact fp32 (v_mfma_f32_16x16x4_f32 = a k-ordered fmaf chain)
//
// Operands and the result are FUNCTORS, so framing a waveform, reflecting its ends, masking a spectrum, squaring a
// spectrum into a power spectrum, adding a bias or applying an activation all happen in the loaders / the store instead of
// in separate passes over HBM:
//   struct ALoad { static constexpr bool kAlongK = ...; __device__ float operator()(int m, int k) const; };
//        kAlongK = true : consecutive k are contiguous in memory (row-major weight matrices)
//        kAlongK = false: consecutive m are contiguous (an activation read "transposed")
//   struct BLoad { static constexpr bool kAlongN = ...; __device__ float operator()(int k, int n) const; };
//        kAlongN = true : consecutive n contiguous (row-major activations / tables);  false: consecutive k contiguous
//   struct Store { __device__ void operator()(int m, int n, float v) const; };
// The flags only choose which lanes fetch which elements of a slab (so that the fetches coalesce); out-of-range
// elements are never requested.
// One 256-thread workgroup computes a 128 x 128 tile; each of its 4 wavefronts owns a 64 x 64 quadrant as 4 x 4 MFMA tiles
// (64 accumulator VGPRs).  Slabs of 16 k are staged k-major in LDS with a row stride of 144 floats (144 mod 64 = 16):
// the per-lane operand reads -- 16 consecutive rows x 4 consecutive k -- then cover all 64 banks exactly once.
#pragma once
#include "ade_device.h"

namespace ade {
namespace gemm {

using namespace dev;

constexpr int kTM = 128, kTN = 128, kTK = 16;
constexpr int kLds = 144;

// one 128 x 128 tile of C at (m_blk, n_blk); As / Bs: the workgroup's two kTK x kLds LDS slabs
template <class AL, class BL, class ST>
__device__ __forceinline__ void gemm_tile(const AL& a_of, const BL& b_of, const ST& store, int M, int N, int K, int m_blk, int n_blk,
                                          float* As, float* Bs) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int j16 = lane & 15, g = lane >> 4;
    v4f acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = v4f{0.0f, 0.0f, 0.0f, 0.0f};

    // Software pipeline: the operand elements of slab k+1 are requested (into registers) before the 64 MFMAs of slab k
    // run and are written to LDS after them, so the HBM / L2 latency of a slab hides under the matrix work of the
    // previous one.  Which lane fetches which element depends on the operand's contiguous direction (see the header).
    float ra[8], rb[8];
    auto fetch = [&](int k0) {
        if (AL::kAlongK) {          // thread = (row, half slab): 8 consecutive k of one row
            const int r = tid >> 1, kh = (tid & 1) * 8, m = m_blk + r;
#pragma unroll
            for (int u = 0; u < 8; ++u) ra[u] = (m < M && k0 + kh + u < K) ? a_of(m, k0 + kh + u) : 0.0f;
        } else {                    // thread = (row, half slab): consecutive lanes = consecutive rows for each k
            const int r = tid & 127, kh = (tid >> 7) * 8, m = m_blk + r;
#pragma unroll
            for (int u = 0; u < 8; ++u) ra[u] = (m < M && k0 + kh + u < K) ? a_of(m, k0 + kh + u) : 0.0f;
        }
        if (BL::kAlongN) {          // thread = (column, half slab)
            const int c = tid & 127, kh = (tid >> 7) * 8, n = n_blk + c;
#pragma unroll
            for (int u = 0; u < 8; ++u) rb[u] = (n < N && k0 + kh + u < K) ? b_of(k0 + kh + u, n) : 0.0f;
        } else {                    // thread = (k, 16 column groups): consecutive lanes = consecutive k of one column
            const int kk = tid & 15, cg = tid >> 4;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int n = n_blk + cg + 16 * u;
                rb[u] = (n < N && k0 + kk < K) ? b_of(k0 + kk, n) : 0.0f;
            }
        }
    };
    auto stash = [&]() {            // registers -> k-major LDS slabs (same lane maps as fetch)
        if (AL::kAlongK) {
            const int r = tid >> 1, kh = (tid & 1) * 8;
#pragma unroll
            for (int u = 0; u < 8; ++u) As[(kh + u) * kLds + r] = ra[u];
        } else {
            const int r = tid & 127, kh = (tid >> 7) * 8;
#pragma unroll
            for (int u = 0; u < 8; ++u) As[(kh + u) * kLds + r] = ra[u];
        }
        if (BL::kAlongN) {
            const int c = tid & 127, kh = (tid >> 7) * 8;
#pragma unroll
            for (int u = 0; u < 8; ++u) Bs[(kh + u) * kLds + c] = rb[u];
        } else {
            const int kk = tid & 15, cg = tid >> 4;
#pragma unroll
            for (int u = 0; u < 8; ++u) Bs[kk * kLds + cg + 16 * u] = rb[u];
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < K; k0 += kTK) {
        stash();
        __syncthreads();
        if (k0 + kTK < K) fetch(k0 + kTK);
#pragma unroll
        for (int ks = 0; ks < kTK; ks += 4) {       // lane (g, j16) supplies A[row 16 i + j16][k + g] and B[k + g][col 16 j + j16]
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = As[(ks + g) * kLds + wm + 16 * i + j16];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = Bs[(ks + g) * kLds + wn + 16 * j + j16];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mfma16x16x4(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }
    // lane (g, j16), register r of tile (i, j) is C[wm + 16 i + 4 g + r][wn + 16 j + j16]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m_blk + wm + 16 * i + 4 * g + r, n = n_blk + wn + 16 * j + j16;
                if (m < M && n < N) store(m, n, acc[i][j][r]);
            }
}

template <class AL, class BL, class ST>
__global__ __launch_bounds__(256) void k_gemm128(AL a_of, BL b_of, ST store, int M, int N, int K) {
    __shared__ float As[kTK * kLds];
    __shared__ float Bs[kTK * kLds];
    gemm_tile(a_of, b_of, store, M, N, K, (int)blockIdx.y * kTM, (int)blockIdx.x * kTN, As, Bs);
}

template <class AL, class BL, class ST>
inline void launch(hipStream_t s, const AL& a, const BL& b, const ST& st, int M, int N, int K) {
    const dim3 grid((unsigned)((N + kTN - 1) / kTN), (unsigned)((M + kTM - 1) / kTM));
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gemm128<AL, BL, ST>), grid, dim3(256), 0, s, a, b, st, M, N, K);
}

// Batched form: blockIdx.z selects a problem.  `prob(z)` (evaluated once per workgroup, so its table reads are scalar loads)
// returns a struct with members a, b, st (functors as above) and M, N, K; problems may differ in every one of them -- tiles
// outside a problem's own M x N exit at once, the grid is sized for the largest.
template <class P>
__global__ __launch_bounds__(256) void k_gemm128_batched(P prob) {
    __shared__ float As[kTK * kLds];
    __shared__ float Bs[kTK * kLds];
    const auto q = prob((int)blockIdx.z);
    const int m_blk = (int)blockIdx.y * kTM, n_blk = (int)blockIdx.x * kTN;
    if (m_blk >= q.M || n_blk >= q.N) return;
    gemm_tile(q.a, q.b, q.st, q.M, q.N, q.K, m_blk, n_blk, As, Bs);
}

template <class P>
inline void launch_batched(hipStream_t s, const P& prob, int batch, int max_M, int max_N) {
    const dim3 grid((unsigned)((max_N + kTN - 1) / kTN), (unsigned)((max_M + kTM - 1) / kTM), (unsigned)batch);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gemm128_batched<P>), grid, dim3(256), 0, s, prob);
}

// ---- common functors ------------------------------------------------------------------------------------------
struct RowMajorA {      // A(m, k) = p[m * ld + k]
    static constexpr bool kAlongK = true;
    const float* p;
    int ld;
    __device__ float operator()(int m, int k) const { return p[(size_t)m * ld + k]; }
};
struct WeightNK {       // B(k, n) = p[n * ld + k]: a torch Linear weight (out_features, in_features) used as x @ W^T
    static constexpr bool kAlongN = false;
    const float* p;
    int ld;
    __device__ float operator()(int k, int n) const { return p[(size_t)n * ld + k]; }
};
struct RowMajorB {      // B(k, n) = p[k * ld + n]
    static constexpr bool kAlongN = true;
    const float* p;
    int ld;
    __device__ float operator()(int k, int n) const { return p[(size_t)k * ld + n]; }
};
enum { kActNone = 0, kActRelu = 1, kActSigmoid = 2, kActLogFloor = 3 };
template <int ACT>
struct BiasActStore {   // C(m, n) -> p[m * ld + n] = act(v + bias[m])   (bias may be null)
    float* p;
    int ld;
    const float* bias;
    float floor_;       // kActLogFloor: log(max(v, floor_))
    __device__ void operator()(int m, int n, float v) const {
        if (bias) v += bias[m];
        if (ACT == kActRelu) v = v > 0.0f ? v : 0.0f;
        if (ACT == kActSigmoid) v = 1.0f / (1.0f + expf(-v));
        if (ACT == kActLogFloor) v = logf(v > floor_ ? v : floor_);
        p[(size_t)m * ld + n] = v;
    }
};

}  // namespace gemm
}  // namespace ade
