// ade_gemm64.h — fp32 matrix-core GEMM for TALL-SKINNY problems: millions of rows (tokens), 48..336 output columns, K = 48..1536.
//
//   C(m, n) = sum_k A(m, k) * B(k, n)        exact fp32 (v_mfma_f32_16x16x4_f32)
//
// ZipEnhancer's channel width is 64, so every one of its products -- the causal dense convolutions as implicit GEMMs, the
// Zipformer projections -- has N that is 64 or a small multiple of 16.  The 128 x 128 tile of ade_gemm.h would spend half of
// its matrix-core work on padding there; this variant computes a 256 x 64 tile per 256-thread workgroup (each wavefront a
// 64 x 64 block as 4 x 4 MFMA tiles, 64 accumulator VGPRs) and leaves everything else as in ade_gemm.h: slabs of 16 k staged
// row-major in LDS at 20 floats per row (one ds_read_b128 per operand tile, all 64 banks covered once), slab k + 1 requested
// into registers before the 64 MFMAs of slab k.
// Operands are FUNCTORS, both contiguous along k and fetched as float4 (whole 64-byte lines per 4 lanes):
//   struct ALoad { struct Row {...}; __device__ Row row(int m) const;                    // per-row state, computed ONCE per tile
//                  __device__ float4 vec4(const Row&, int k) const; };                   // k % 4 == 0, k + 3 < K
//   struct BLoad { __device__ float4 vec4(int n, int k) const; };                        // weights stored (N, K) row-major
//   struct Store { __device__ void operator()(int m, int n, float v) const; };
// The Row hook is what lets an implicit-convolution loader do its (batch, frame, bin) index split once instead of per fetch.
// K must be a multiple of 4.
#pragma once
#include "ade_device.h"
#include "ade_gemm.h"

namespace ade {
namespace gemm64 {

using namespace dev;

constexpr int kTM = 256, kTN = 64, kTK = 16, kRow = gemm::kRow;

// waves per SIMD the register allocator aims for: 4 (128 VGPRs) unless the A loader says otherwise -- the implicit-convolution loaders keep per-row state in
// registers and spill at 128 (measured: the ZipEnhancer step 170 -> 188 ms with spills), so they declare kWavesPerSimd = 3
template <class T, class = void>
struct WavesPerSimd { static constexpr int value = 4; };
template <class T>
struct WavesPerSimd<T, std::void_t<decltype(T::kWavesPerSimd)>> { static constexpr int value = T::kWavesPerSimd; };

template <class T, class = void>
struct HasFinish : std::false_type {};
template <class T>
struct HasFinish<T, std::void_t<decltype(&T::finish)>> : std::true_type {};
template <class AL, class BL, class ST>
__global__ __launch_bounds__(256, WavesPerSimd<AL>::value) void k_gemm256x64(AL a_of, BL b_of, ST store, int M, int N, int K) {
    __shared__ __attribute__((aligned(16))) float As[kTM * kRow];
    __shared__ __attribute__((aligned(16))) float Bs[kTN * kRow];
    const int gx = (int)gridDim.x, id = gemm::xcd_contiguous_id((int)blockIdx.x + gx * (int)blockIdx.y, gx * (int)gridDim.y);
    const int m_blk = (id / gx) * kTM, n_blk = (id % gx) * kTN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave * 64, j16 = lane & 15, g = lane >> 4;
    v4f acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = v4f{0.0f, 0.0f, 0.0f, 0.0f};

    const int r = tid >> 2, kq = 4 * (tid & 3);          // this lane fetches k quarter kq of rows r, r + 64, r + 128, r + 192 and of column r
    typename AL::Row rows[4];
    bool rok[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        const int m = m_blk + r + 64 * h;
        rok[h] = m < M;
        rows[h] = a_of.row(rok[h] ? m : 0);
    }
    const int nb = n_blk + r;
    const bool nok = nb < N;
    float4 ra[4], rb;
    // A fetch is LOADS ONLY, from addresses that are always in range: rows beyond M / N re-read row 0 (their products are never stored); the k >= K part of the last slab is
    // zeroed when the slab goes to LDS, behind the wait that write needs anyway.  (select(ok, load, 0) at the load made the compiler wait for the slab right after requesting
    // it, ahead of the previous slab's MFMAs: the prefetch hid nothing.)
    auto fetch = [&](int k0) {
        const int kc = k0 + kq < K ? k0 + kq : 0;
#pragma unroll
        for (int h = 0; h < 4; ++h) ra[h] = a_of.vec4(rows[h], kc);
        rb = b_of.vec4(nok ? nb : 0, kc);
    };
    fetch(0);
    for (int k0 = 0; k0 < K; k0 += kTK) {
        if constexpr (HasFinish<AL>::value) {            // an operand that is a function of what was loaded (an activation): applied here, not at the load
#pragma unroll
            for (int h = 0; h < 4; ++h) ra[h] = AL::finish(ra[h]);
        }
        if (k0 + kTK > K) {                              // wave-uniform: the last slab of a K that is not a multiple of 16
            const bool in = k0 + kq < K;
#pragma unroll
            for (int h = 0; h < 4; ++h) ra[h] = keep4(in, ra[h]);
            rb = keep4(in, rb);
        }
#pragma unroll
        for (int h = 0; h < 4; ++h) *reinterpret_cast<float4*>(As + (r + 64 * h) * kRow + kq) = ra[h];
        *reinterpret_cast<float4*>(Bs + r * kRow + kq) = rb;
        __syncthreads();
        if (k0 + kTK < K) fetch(k0 + kTK);
        float4 a4[4], b4[4];                // lane (g, j16): A[row 16 i + j16][k = 4 g + s], B[k = 4 g + s][col 16 j + j16]
#pragma unroll
        for (int i = 0; i < 4; ++i) a4[i] = *reinterpret_cast<const float4*>(As + (wm + 16 * i + j16) * kRow + 4 * g);
#pragma unroll
        for (int j = 0; j < 4; ++j) b4[j] = *reinterpret_cast<const float4*>(Bs + (16 * j + j16) * kRow + 4 * g);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[i][j] = mfma16x16x4(a4[i].x, b4[j].x, acc[i][j]);
                acc[i][j] = mfma16x16x4(a4[i].y, b4[j].y, acc[i][j]);
                acc[i][j] = mfma16x16x4(a4[i].z, b4[j].z, acc[i][j]);
                acc[i][j] = mfma16x16x4(a4[i].w, b4[j].w, acc[i][j]);
            }
        __syncthreads();
    }
    // lane (g, j16), register q of tile (i, j) is C[wm + 16 i + 4 g + q][16 j + j16]
    if constexpr (gemm::HasCtx<ST>::value) {                 // context form (ade_gemm.h): batched reads ahead of each band's 16 writes, interior tiles unguarded
        auto emit = [&](auto guard_c) {
            constexpr bool G = decltype(guard_c)::value;
            decltype(store.col(0)) cc[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = n_blk + 16 * j + j16;
                cc[j] = store.col(G && n >= N ? N - 1 : n);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                decltype(store.row(0)) rc[4];
                decltype(store.pre(0, 0, store.row(0))) pc[4][4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int m = m_blk + wm + 16 * i + 4 * g + q, mc = G && m >= M ? M - 1 : m;
                    rc[q] = store.row(mc);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int n = n_blk + 16 * j + j16;
                        pc[q][j] = store.pre(mc, G && n >= N ? N - 1 : n, rc[q]);
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int m = m_blk + wm + 16 * i + 4 * g + q, n = n_blk + 16 * j + j16;
                        if (!G || (m < M && n < N)) store(m, n, acc[i][j][q], rc[q], cc[j], pc[q][j]);
                    }
            }
        };
        if (m_blk + kTM <= M && n_blk + kTN <= N) emit(std::false_type{});
        else emit(std::true_type{});
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int m = m_blk + wm + 16 * i + 4 * g + q;
                if (m >= M) continue;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int n = n_blk + 16 * j + j16;
                    if (n < N) store(m, n, acc[i][j][q]);
                }
            }
    }
}

template <class AL, class BL, class ST>
inline void launch(hipStream_t s, const AL& a, const BL& b, const ST& st, int M, int N, int K) {
    if (M <= 0 || N <= 0) return;
    const dim3 grid((unsigned)((N + kTN - 1) / kTN), (unsigned)((M + kTM - 1) / kTM));
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gemm256x64<AL, BL, ST>), grid, dim3(256), 0, s, a, b, st, M, N, K);
}

// ---- K = 64 projections: out(m, off + n) = sum_{k < 64} x(m, k) w(n, k) + bias(n) ------------------------------------------------------------------------
// ZipEnhancer's eight in-projections per layer all contract over the 64 channels.  With K that short a slab pipeline is mostly prologue, so this kernel drops it:
//   * a workgroup owns 64 RT rows and ALL output columns; each wavefront's 16 RT x 64 activation block is fetched ONCE, straight from global memory into the registers the
//     matrix cores read (lane (g, j16): row 16 i + j16, k = 16 ks + 4 g .. + 3 as one float4 -- no LDS round trip for the activations);
//   * the weights stream through LDS 64 columns at a time, double-buffered (one barrier per 64 columns);
//   * the product is formed TRANSPOSED (weights as the A operand, activations as B), so a lane's four accumulator registers are four CONSECUTIVE output columns of one
//     row: one 16-byte store per tile instead of four 4-byte ones.
// ldx, ldo, off: multiples of 4; w: (N, 64) row-major; N a multiple of 16.
template <int RT>
__global__ __launch_bounds__(256) void k_proj64(const float* __restrict__ x, int ldx, const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ out,
                                                int ldo, int off, int M, int N) {
    __shared__ __attribute__((aligned(16))) float Ws[2][4 * 64 * kRow];     // [buffer][ks][column][16 k + pad]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j16 = lane & 15, g = lane >> 4;
    const int m0 = (int)blockIdx.x * (64 * RT) + wave * (16 * RT);
    float4 xa[RT][4];                        // [row tile i][ks]
#pragma unroll
    for (int i = 0; i < RT; ++i) {
        const int m = m0 + 16 * i + j16;
        const float* row = x + (size_t)(m < M ? m : 0) * ldx + 4 * g;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            xa[i][ks] = keep4(m < M, *reinterpret_cast<const float4*>(row + 16 * ks));           // clamped row, no branch around the load
        }
    }
    auto stage = [&](int n0, int buf) {       // 64 columns x 64 k: 1024 float4, four per lane
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = tid + 256 * u, c = idx >> 4, k4 = (idx & 15) * 4, n = n0 + c;
            const float4 v = keep4(n < N, *reinterpret_cast<const float4*>(w + (size_t)(n < N ? n : 0) * 64 + k4));
            *reinterpret_cast<float4*>(Ws[buf] + ((k4 >> 4) * 64 + c) * kRow + (k4 & 15)) = v;
        }
    };
    stage(0, 0);
    int buf = 0;
    for (int n0 = 0; n0 < N; n0 += 64, buf ^= 1) {
        __syncthreads();                     // Ws[buf] is complete; every wave has finished reading Ws[buf ^ 1] (previous iteration)
        if (n0 + 64 < N) stage(n0 + 64, buf ^ 1);
        float4 bj[4];                         // the column block's biases, requested before the products (clamped: no branch around the loads)
#pragma unroll
        for (int jn = 0; jn < 4; ++jn) { const int n = n0 + 16 * jn + 4 * g; bj[jn] = *reinterpret_cast<const float4*>(bias + (n < N ? n : 0)); }
        v4f acc[4][RT];                       // [column tile jn][row tile i]: lane (g, j16) holds out[row 16 i + j16][columns 16 jn + 4 g .. + 3]
#pragma unroll
        for (int jn = 0; jn < 4; ++jn)
#pragma unroll
            for (int i = 0; i < RT; ++i) acc[jn][i] = v4f{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            {
                float4 wv[4];
#pragma unroll
                for (int jn = 0; jn < 4; ++jn) wv[jn] = *reinterpret_cast<const float4*>(Ws[buf] + (ks * 64 + 16 * jn + j16) * kRow + 4 * g);
                // k-step outermost: sixteen independent accumulators between two steps of the same one (no back-to-back dependent MFMAs)
#pragma unroll
                for (int jn = 0; jn < 4; ++jn)
#pragma unroll
                    for (int i = 0; i < RT; ++i) acc[jn][i] = mfma16x16x4(wv[jn].x, xa[i][ks].x, acc[jn][i]);
#pragma unroll
                for (int jn = 0; jn < 4; ++jn)
#pragma unroll
                    for (int i = 0; i < RT; ++i) acc[jn][i] = mfma16x16x4(wv[jn].y, xa[i][ks].y, acc[jn][i]);
#pragma unroll
                for (int jn = 0; jn < 4; ++jn)
#pragma unroll
                    for (int i = 0; i < RT; ++i) acc[jn][i] = mfma16x16x4(wv[jn].z, xa[i][ks].z, acc[jn][i]);
#pragma unroll
                for (int jn = 0; jn < 4; ++jn)
#pragma unroll
                    for (int i = 0; i < RT; ++i) acc[jn][i] = mfma16x16x4(wv[jn].w, xa[i][ks].w, acc[jn][i]);
            }
        }
#pragma unroll
        for (int jn = 0; jn < 4; ++jn) {
            const int n = n0 + 16 * jn + 4 * g;
            if (n >= N) continue;
            const float4 b = bj[jn];
#pragma unroll
            for (int i = 0; i < RT; ++i) {
                const int m = m0 + 16 * i + j16;
                if (m < M)
                    *reinterpret_cast<float4*>(out + (size_t)m * ldo + off + n) =
                        make_float4(acc[jn][i][0] + b.x, acc[jn][i][1] + b.y, acc[jn][i][2] + b.z, acc[jn][i][3] + b.w);
            }
        }
    }
}
inline void launch_proj64(hipStream_t s, const float* x, int ldx, const float* w, const float* bias, float* out, int ldo, int off, int M, int N) {
    if (M <= 0 || N <= 0) return;
    constexpr int RT = 2;                        // 16-row tiles per wavefront: 2 -> 128 rows per workgroup, ~120 VGPRs, four wavefronts per SIMD (4 -> 232 VGPRs, two)
    const dim3 grid((unsigned)((M + 64 * RT - 1) / (64 * RT)));
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_proj64<RT>), grid, dim3(256), 0, s, x, ldx, w, bias, out, ldo, off, M, N);
}

// ---- common operands ---------------------------------------------------------------------------------------------------
struct RowsA {           // A(m, k) = p[m * ld + k]
    const float* p;
    int ld;
    struct Row { const float* q; };
    __device__ Row row(int m) const { return Row{p + (size_t)m * ld}; }
    __device__ float4 vec4(const Row& r, int k) const { return *reinterpret_cast<const float4*>(r.q + k); }
};
struct WeightB {         // B(k, n) = p[n * ld + k]: a (N, K) row-major weight used as x @ W^T
    const float* p;
    int ld;
    __device__ float4 vec4(int n, int k) const { return *reinterpret_cast<const float4*>(p + (size_t)n * ld + k); }
};

}  // namespace gemm64
}  // namespace ade
