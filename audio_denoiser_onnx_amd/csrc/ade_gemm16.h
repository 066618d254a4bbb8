// ade_gemm16.h — bf16 GEMM on the gfx950 matrix cores with bf16 operands STORED in HBM (the real bf16 path of BASELINE.json configs[2] / [3]).
//
//   C(m, n) = sum_k A[m][k] * B[n][k]        A, B: bf16, row-major with k contiguous (activations (rows, features); torch Linear weights (out, in));
//                                            products exact, accumulation fp32 (v_mfma_f32_32x32x16_bf16)
//
// The operands are half the bytes everywhere -- HBM, L2, LDS -- and the instruction is gfx950's full-rate 32x32x16 form (16 x the f32 matrix rate).  At these shapes
// (M ~ 1.5 M rows, N and K 384 .. 1544) the loop is NOT matrix-bound: the counters of the bare loop (tools/pmc_gemm16_unit.sh, K = 1536) show the matrix cores 41 % and the LDS
// pipe 44 % busy with the wavefronts waiting to ISSUE 60 % of their cycles -- a 128 x 128 x 64 slab step costs the CU's one LDS pipe ~768 cycles (32 ds_write_b128 at 16,
// 64 ds_read_b128 at 4) against 512 matrix cycles per SIMD, and every VALU instruction of a store takes an issue slot the MFMAs want.  So the kernel is built around
// full-line transfers, a prefetch that really stays in flight, and stores with as few instructions as they can have:
//   * one 256-thread workgroup = one 128 x 128 tile of C, four wavefronts in 2 x 2, each a 64 x 64 quadrant as 2 x 2 MFMA tiles (64 accumulator registers);
//   * k runs in slabs of 64: a slab of an operand is 128 rows x 128 bytes, fetched as whole 128-byte lines (8 lanes x 16 bytes per row) into registers one slab
//     ahead of the MFMAs and written to LDS with a 144-byte row pitch (36 words: the 16 lanes of every ds_read_b128 service group land on 16 distinct 4-bank sets);
//   * a lane's operand of one MFMA (8 consecutive k of its row) is ONE ds_read_b128;
//   * the tile is computed TRANSPOSED (MFMA A operand = rows of B, MFMA B operand = rows of A), so a lane's accumulator registers are runs of four consecutive n of
//     one row m; the epilogue passes the tile through LDS (fp32, 64 rows at a time) and hands the store functor whole float4s (m, n .. n + 3) with consecutive lanes on
//     consecutive n: every read the store makes (bias, rotary table, the old value of a residual) and every write it makes is a full-line access, whatever the output type.
// Store functor:  ColT col(int n, int cnt) const;                                      what it reads per column group (bias ...), fetched once per thread
//                 __device__ void operator()(int m, int n, float4 v, int cnt, const ColT&) const;     n % 4 == 0, cnt = min(4, N - n) valid columns
// K % 8 == 0, lda % 8 == 0, ldb % 8 == 0 and 16-byte aligned bases are required (checked by the launchers' callers); M, N are free.
#pragma once
#include "ade_device.h"

#include <cstdint>

namespace ade {
namespace gemm16 {

using namespace dev;

typedef unsigned short bf16_t;                       // storage type (bit pattern)
#if defined(__clang__)
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef __bf16 v2bf __attribute__((ext_vector_type(2)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v2f32 __attribute__((ext_vector_type(2)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef _Float16 v2h __attribute__((ext_vector_type(2)));
#else                                                // g++ spelling for the host-side simulator build under tests/hipsim
typedef short v8bf __attribute__((vector_size(16)));
typedef short v8h __attribute__((vector_size(16)));
typedef float v16f __attribute__((vector_size(64)));
#endif

constexpr int kTM = 128, kTN = 128, kTK = 64;
constexpr int kPitch = 144;                          // bytes per staged row (128 + 16)
constexpr int kSlabBytes = kTM * kPitch;             // one operand's slab
constexpr int kEpiPitch = 132;                       // floats per row of the epilogue tile (128 + 4: a wave's float4 writes of 8 consecutive rows cover all banks once)
constexpr int kEpiBytes = 64 * kEpiPitch * 4;
constexpr int kMainLds = 2 * kSlabBytes > kEpiBytes ? 2 * kSlabBytes : kEpiBytes;       // 36 864: the two operand slabs, later the epilogue tile
constexpr int kLdsBytes = kMainLds + 2 * kTM * 4;    // + the tile's row scales and row contexts (four workgroups per CU: 151.5 KB)

// fp32 -> bf16, round to nearest even: v_cvt_pk_bf16_f32 on gfx950
__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {
#if defined(__clang__)
    v2f32 v = {a, b};
    v2bf r = __builtin_convertvector(v, v2bf);
    unsigned w;
    __builtin_memcpy(&w, &r, 4);
    return w;
#else
    auto one = [](float x) { unsigned u = __float_as_uint(x); u += 0x7fffu + ((u >> 16) & 1u); return u >> 16; };
    return one(a) | (one(b) << 16);
#endif
}
__device__ __forceinline__ uint2 pack_bf16x4(const float4& v) { return make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w)); }
__device__ __forceinline__ float bf16_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16_hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }

__device__ __forceinline__ v8bf as_v8bf(const uint4& u) { v8bf r; __builtin_memcpy(&r, &u, 16); return r; }      // (a register rename on the GPU)
__device__ __forceinline__ v16f mfma32x32x16(const uint4& a, const uint4& b, v16f c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_v8bf(a), as_v8bf(b), c, 0, 0, 0); }

// IEEE half (fp16) operands of the same matrix instruction family (v_mfma_f32_32x32x16_f16: the bf16 rate): three more mantissa bits than bf16 for tensors whose range is
// known -- ZipEnhancer's dense blocks, InstanceNorm'd activations of order one (csrc/ade_zip16.h).  fp32 -> fp16 rounds to nearest even (v_cvt_f16_f32 / v_cvt_pk_f16_f32),
// saturating at the largest finite half instead of overflowing to infinity.
__host__ __device__ inline unsigned short f16_bits(float x) {            // portable form: the host's weight conversion and the host simulator
    unsigned u;
    __builtin_memcpy(&u, &x, 4);
    const unsigned sign = (u >> 16) & 0x8000u;
    u &= 0x7fffffffu;
    if (u > 0x7f800000u) return (unsigned short)(sign | 0x7e00u);        // NaN
    if (u >= 0x477ff000u) return (unsigned short)(sign | 0x7bffu);       // >= 65520 rounds beyond the largest finite half: saturate
    if (u < 0x33000001u) return (unsigned short)sign;                    // <= 2^-25: rounds to zero
    if (u < 0x38800000u) {                                               // subnormal half: value = m * 2^-24
        const int e = (int)(u >> 23);                                    // biased fp32 exponent, 102 .. 112
        const unsigned m = (u & 0x7fffffu) | 0x800000u;
        const int sh = 126 - e;                                          // 14 .. 24
        unsigned r = m >> sh;
        const unsigned rem = m & ((1u << sh) - 1u), half = 1u << (sh - 1);
        if (rem > half || (rem == half && (r & 1u))) ++r;
        return (unsigned short)(sign | r);
    }
    unsigned r = u - 0x38000000u;                                        // re-bias the exponent: 127 -> 15
    const unsigned rem = r & 0x1fffu;
    r >>= 13;
    if (rem > 0x1000u || (rem == 0x1000u && (r & 1u))) ++r;
    return (unsigned short)(sign | r);
}
__device__ __forceinline__ unsigned pack_f16x2(float a, float b) {
#if defined(__clang__) && defined(__AMDGCN__)
    a = __builtin_amdgcn_fmed3f(a, -65504.0f, 65504.0f);
    b = __builtin_amdgcn_fmed3f(b, -65504.0f, 65504.0f);
    v2f32 v = {a, b};
    v2h r = __builtin_convertvector(v, v2h);
    unsigned w;
    __builtin_memcpy(&w, &r, 4);
    return w;
#else
    return (unsigned)f16_bits(a) | ((unsigned)f16_bits(b) << 16);
#endif
}
__device__ __forceinline__ uint2 pack_f16x4(const float4& v) { return make_uint2(pack_f16x2(v.x, v.y), pack_f16x2(v.z, v.w)); }
__host__ __device__ inline float f16_value(unsigned short b) {           // portable form (the host simulator)
    const unsigned sign = (unsigned)(b & 0x8000u) << 16, e = (b >> 10) & 31u, m = b & 0x3ffu;
    unsigned u;
    if (e == 0) {
        if (m == 0) u = sign;
        else { int k = 0; unsigned mm = m; while (!(mm & 0x400u)) { mm <<= 1; ++k; } u = sign | ((unsigned)(113 - k) << 23) | ((mm & 0x3ffu) << 13); }
    } else if (e == 31) u = sign | 0x7f800000u | (m << 13);
    else u = sign | ((e + 112u) << 23) | (m << 13);
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
}
__device__ __forceinline__ void unpack_f16x2(unsigned w, float& lo, float& hi) {
#if defined(__clang__) && defined(__AMDGCN__)
    v2h h;
    __builtin_memcpy(&h, &w, 4);
    lo = (float)h[0]; hi = (float)h[1];
#else
    lo = f16_value((unsigned short)(w & 0xffffu)); hi = f16_value((unsigned short)(w >> 16));
#endif
}
__device__ __forceinline__ v8h as_v8h(const uint4& u) { v8h r; __builtin_memcpy(&r, &u, 16); return r; }
__device__ __forceinline__ v16f mfma32x32x16_f16(const uint4& a, const uint4& b, v16f c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(as_v8h(a), as_v8h(b), c, 0, 0, 0); }
// one spelling for kernels templated on the 16-bit operand type of a tensor family: HALF = IEEE half, otherwise bf16
template <bool HALF> __device__ __forceinline__ v16f mfma32x32x16_t(const uint4& a, const uint4& b, v16f c) { return HALF ? mfma32x32x16_f16(a, b, c) : mfma32x32x16(a, b, c); }
template <bool HALF> __device__ __forceinline__ uint2 pack16x4_t(const float4& v) { return HALF ? pack_f16x4(v) : pack_bf16x4(v); }

__device__ __forceinline__ uint4 zero_unless(bool ok, const uint4& v) { return make_uint4(ok ? v.x : 0u, ok ? v.y : 0u, ok ? v.z : 0u, ok ? v.w : 0u); }
// eight bf16 at p if ok, else zeros, selected by ADDRESS (ade_device.h, ld4_or_zero): a prefetch written with it stays in flight across the matrix work that follows
__device__ __forceinline__ uint4 ld8_or_zero(bool ok, const bf16_t* p) { return *reinterpret_cast<const uint4*>(ok ? reinterpret_cast<const void*>(p) : reinterpret_cast<const void*>(g_zero4)); }

// one 128 x 128 tile of C at (m_blk, n_blk); lds: kLdsBytes, 16-byte aligned
// A store may provide `int row_ctx(int m) const` (something it needs per ROW that costs integer divisions to derive: Mel-Band's rotary position): the tile computes it once
// per row into LDS when it starts and hands it to `operator()(m, n, v, cnt, colctx, rowctx)`.
template <class T, class = void>
struct HasRowCtx : std::false_type {};
template <class T>
struct HasRowCtx<T, std::void_t<decltype(&T::row_ctx)>> : std::true_type {};
// row_scale (may be null): C(m, n) is multiplied by row_scale[m] before the store sees it (the norm of a normalised operand: ade_melband.hip).  The tile's 128 scales are
// fetched into LDS when the tile starts, so the epilogue -- which runs with nothing in flight to hide a global load behind -- reads them from LDS.
template <class ST>
__device__ __forceinline__ void gemm_tile(const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ B, int ldb, const ST& store, int M, int N, int K, int m_blk, int n_blk,
                                          unsigned char* lds, const float* __restrict__ row_scale = nullptr) {
    float* rsc = reinterpret_cast<float*>(lds + kMainLds);
    if (row_scale && threadIdx.x < kTM) rsc[threadIdx.x] = row_scale[m_blk + (int)threadIdx.x < M ? m_blk + (int)threadIdx.x : M - 1];
    int* rcx = reinterpret_cast<int*>(lds + kMainLds + kTM * 4);
    if constexpr (HasRowCtx<ST>::value) {
        if (threadIdx.x >= 128 && threadIdx.x < 128 + kTM) { const int rr = (int)threadIdx.x - 128; rcx[rr] = store.row_ctx(m_blk + rr < M ? m_blk + rr : M - 1); }
    }
    unsigned char* As = lds;
    unsigned char* Bs = lds + kSlabBytes;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int l31 = lane & 31, h = lane >> 5;
    v16f acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // staging map: lane = (row, 16-byte piece): 8 lanes read one row's 128-byte line; rows r, r + 32, r + 64, r + 96
    const int sr = tid >> 3, sc = tid & 7;
    // (addresses = a wave-uniform tile base + a 32-bit lane offset: eight 64-bit lane pointers would cost 16 registers at the 128-register ceiling)
    const bf16_t* At = A + (size_t)m_blk * lda;
    const bf16_t* Bt = B + (size_t)n_blk * ldb;
    int ao[4], bo[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int r = sr + 32 * u;
        ao[u] = (m_blk + r < M ? r : M - 1 - m_blk) * lda;
        bo[u] = (n_blk + r < N ? r : N - 1 - n_blk) * ldb;
    }
    // (Two register sets -- slab k + 2 requested while slab k is multiplied -- were measured SLOWER on the first form of this loop: 150+ registers cost the fourth resident
    //  workgroup, 268.9 -> 279.0 ms per Mel-Band step.)
    // A fetch is LOADS ONLY, from addresses that are always in range: rows beyond M / N re-read the last row (their products are never stored) and the pieces of a slab beyond
    // K re-read k = 0 and are zeroed in stash(), behind the wait the LDS write needs anyway.  (Zeroing at the load -- `select(ok, load, 0)` -- made the compiler wait for the
    // slab right after requesting it, ahead of the previous slab's MFMAs: the prefetch hid nothing.)
    uint4 ra[4], rb[4];
    bool kok = true, ktail = false;
    auto fetch = [&](int k0) {
        ktail = k0 + kTK > K;                                            // wave-uniform
        kok = k0 + 8 * sc < K;                                           // K % 8 == 0: a piece is all in or all out
        const int ko = kok ? k0 + 8 * sc : 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) ra[u] = *reinterpret_cast<const uint4*>(At + (ao[u] + ko));
#pragma unroll
        for (int u = 0; u < 4; ++u) rb[u] = *reinterpret_cast<const uint4*>(Bt + (bo[u] + ko));
    };
    auto stash = [&]() {
        if (ktail) {
#pragma unroll
            for (int u = 0; u < 4; ++u) { ra[u] = zero_unless(kok, ra[u]); rb[u] = zero_unless(kok, rb[u]); }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            *reinterpret_cast<uint4*>(As + (sr + 32 * u) * kPitch + 16 * sc) = ra[u];
            *reinterpret_cast<uint4*>(Bs + (sr + 32 * u) * kPitch + 16 * sc) = rb[u];
        }
    };
    // MFMA operand layout (32x32x16): lane l supplies row (l & 31), k = 8 (l >> 5) .. + 7 of the 16-deep step.  MFMA-A = rows of B (n), MFMA-B = rows of A (m):
    // D[i = n][j = m], lane (j = l & 31, h): register r holds n = (r & 3) + 8 (r >> 2) + 4 h of row m = j.
    auto compute = [&]() {
#pragma unroll
        for (int ks = 0; ks < kTK / 16; ++ks) {
            uint4 fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const uint4*>(As + (wm + 32 * i + l31) * kPitch + 32 * ks + 16 * h);
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[j] = *reinterpret_cast<const uint4*>(Bs + (wn + 32 * j + l31) * kPitch + 32 * ks + 16 * h);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mfma32x32x16(fb[j], fa[i], acc[i][j]);
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < K; k0 += kTK) {
        stash();
        __syncthreads();
        if (k0 + kTK < K) fetch(k0 + kTK);
        compute();
        __syncthreads();
    }
    // epilogue: 64 rows at a time through LDS
    float* E = reinterpret_cast<float*>(lds);
    // whatever the store reads per column (bias, gains): a thread's column is the same in both halves, and the load is issued here so that it is in flight across the
    // accumulators' trip to LDS and the barrier instead of stalling each half's stores
    const int c4 = tid & 31, n = n_blk + 4 * c4, cnt = N - n < 4 ? N - n : 4;
    const bool full = m_blk + kTM <= M && n_blk + kTN <= N;      // (wave-uniform) an interior tile: no row or column of it needs a bounds check, every store is a whole float4
    const auto cc = full ? store.col(n, 4) : store.col(n < N ? n : 0, n < N ? cnt : 1);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if ((wave >> 1) == half) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        *reinterpret_cast<float4*>(E + (32 * i + l31) * kEpiPitch + wn + 32 * j + 8 * q + 4 * h) =
                            make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
        }
        __syncthreads();
        {   // lane = (row tid >> 5 (+ 8 u), float4 column tid & 31): the column, and whatever the store reads per column (bias), is the same for a thread's 8 rows
            if (full) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int row = (tid >> 5) + 8 * u;
                    float4 v = *reinterpret_cast<const float4*>(E + row * kEpiPitch + 4 * c4);
                    if (row_scale) {
                        const float rs = rsc[64 * half + row];
                        v = make_float4(v.x * rs, v.y * rs, v.z * rs, v.w * rs);
                    }
                    if constexpr (HasRowCtx<ST>::value) store(m_blk + 64 * half + row, n, v, 4, cc, rcx[64 * half + row]);
                    else store(m_blk + 64 * half + row, n, v, 4, cc);
                }
            } else if (n < N) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int row = (tid >> 5) + 8 * u, m = m_blk + 64 * half + row;
                    if (m < M) {
                        float4 v = *reinterpret_cast<const float4*>(E + row * kEpiPitch + 4 * c4);
                        if (row_scale) {
                            const float rs = rsc[64 * half + row];
                            v = make_float4(v.x * rs, v.y * rs, v.z * rs, v.w * rs);
                        }
                        if constexpr (HasRowCtx<ST>::value) store(m, n, v, cnt, cc, rcx[64 * half + row]);
                        else store(m, n, v, cnt, cc);
                    }
                }
            }
        }
        __syncthreads();
    }
}

// (Measured and not kept: the same tile with both operands DMA'd straight into LDS -- global_load_lds_dwordx4, two unpadded 32 KB stages with an XOR swizzle (piece p of
//  row r in slot p ^ ((r >> 1) & 7), conflict-free for ds_read_b128), one barrier per slab.  Bit-identical results, but 64 KB of LDS leave two workgroups per CU instead of
//  four: 230 -> 250 ms per Mel-Band step, same box, alternating runs.  Two register sets (slab k + 2 in flight) lost the fourth workgroup to registers: 269 -> 279 ms.
//  Eight wavefronts per workgroup (2 x 4 quadrants of 64 x 32, 70 registers, 24 - 32 wavefronts per CU instead of 16): within 3 % of this form on every Mel-Band shape
//  (tests/unit/gemm16_unit -t: 357 / 419 / 694 / 502 against 371 / 417 / 681 / 461 TFLOP/s at N x K = 1544 x 384, 1536 x 384, 384 x 1536, 384 x 512) -- the loop is not
//  short of wavefronts to cover its loads with; at 0.016 byte per flop it asks the L2s for ~11 TB/s at 700 TFLOP/s, and a larger tile is what would lower that.)
// Consecutive logical tile ids share an XCD (ade_gemm.h): all n-tiles of an m-strip re-read that strip of A from one L2.
__device__ __forceinline__ int xcd_contiguous_id(int w, int total) {
    const int per = total >> 3, rem = total & 7, xcd = w & 7, idx = w >> 3;
    return (xcd < rem ? xcd * (per + 1) : rem * (per + 1) + (xcd - rem) * per) + idx;
}

template <class ST>
__global__ __launch_bounds__(256, 4) void k_gemm16(const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ B, int ldb, ST store, int M, int N, int K,
                                                   const float* __restrict__ row_scale) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[kLdsBytes];
    const int gx = (int)gridDim.x, id = xcd_contiguous_id((int)blockIdx.x + gx * (int)blockIdx.y, gx * (int)gridDim.y);
    gemm_tile(A, lda, B, ldb, store, M, N, K, (id / gx) * kTM, (id % gx) * kTN, lds, row_scale);
}

template <class ST>
inline void launch(hipStream_t s, const bf16_t* A, int lda, const bf16_t* B, int ldb, const ST& st, int M, int N, int K, const float* row_scale = nullptr) {
    const dim3 grid((unsigned)((N + kTN - 1) / kTN), (unsigned)((M + kTM - 1) / kTM));
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gemm16<ST>), grid, dim3(256), 0, s, A, lda, B, ldb, st, M, N, K, row_scale);
}


// Batched form: blockIdx.z selects a problem; prob(z) returns {A, lda, B, ldb, st, M, N, K} (evaluated once per workgroup); tiles outside a problem's own M x N exit at once.
template <class ST>
struct Prob { const bf16_t* A; int lda; const bf16_t* B; int ldb; ST st; int M, N, K; const float* row_scale; };
template <class P>
__global__ __launch_bounds__(256, 4) void k_gemm16_batched(P prob) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[kLdsBytes];
    const int gx = (int)gridDim.x, gy = (int)gridDim.y;
    const int id = xcd_contiguous_id((int)blockIdx.x + gx * ((int)blockIdx.y + gy * (int)blockIdx.z), gx * gy * (int)gridDim.z);
    const int z = id / (gx * gy), in_z = id - z * gx * gy;
    const auto q = prob(z);
    const int m_blk = (in_z / gx) * kTM, n_blk = (in_z % gx) * kTN;
    if (m_blk >= q.M || n_blk >= q.N) return;
    gemm_tile(q.A, q.lda, q.B, q.ldb, q.st, q.M, q.N, q.K, m_blk, n_blk, lds, q.row_scale);
}
template <class P>
inline void launch_batched(hipStream_t s, const P& prob, int batch, int max_M, int max_N) {
    const dim3 grid((unsigned)((max_N + kTN - 1) / kTN), (unsigned)((max_M + kTM - 1) / kTM), (unsigned)batch);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gemm16_batched<P>), grid, dim3(256), 0, s, prob);
}

// ---- common stores ------------------------------------------------------------------------------------------------------
struct NoCol {};
// n / d for 0 <= n < 2^31 and a divisor fixed per launch: one v_mul_hi_u32 and a correction instead of the ~40-instruction division sequence.
// mul = floor(2^32 / d) + 1 over-estimates 2^32 / d by less than 1 / d relative, so the estimate is floor(n / d) or one more (n < 2^32).
struct FastDiv {
    unsigned d, mul;
    __device__ int div(int n) const {
        if (d == 1u) return n;
        unsigned q = __umulhi((unsigned)n, mul);
        if (q * d > (unsigned)n) --q;
        return (int)q;
    }
};
inline FastDiv make_fastdiv(int d) { return FastDiv{(unsigned)d, d > 1 ? (unsigned)(0x100000000ull / (unsigned)d) + 1u : 0u}; }
// erf to ~1.5e-7 absolute (Abramowitz & Stegun 7.1.26) on the hardware exp and reciprocal: the bf16 path rounds what follows to 8 mantissa bits, erff()'s ~40 instructions
// per element are what bound a K = 384 product's epilogue
__device__ __forceinline__ float erf_fast(float x) {
    const float a = fabsf(x), t = fast_rcp(fmaf(0.3275911f, a, 1.0f));
    const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
    const float r = 1.0f - poly * __expf(-a * a);
    return copysignf(r, x);
}

// GELU of two values on the packed fp32 pipe: x (1/2 + x_c P(x_c^2)), x_c = x clamped to +-4, P a degree-7 fit of (Phi(x) - 1/2) / x on [-4, 4] (|error| 2.1e-5, Phi(4) = 1 - 3e-5:
// |gelu - exact| < 1e-4 inside, < 5e-5 |x| outside; the result is rounded to bf16's 8 bits).  14 instructions per PAIR (2 v_med3, 10 v_pk_fma / v_pk_mul) against erf_fast's ~22 per
// element: a K = 384 product's store is bound by its VALU work (64 outputs per thread and tile at 4 cycles per wave64 instruction outweigh the tile's 96 MFMAs)
__device__ __forceinline__ v2f gelu_pk(v2f x) {
    const v2f xc = mk2(__builtin_amdgcn_fmed3f(x[0], -4.0f, 4.0f), __builtin_amdgcn_fmed3f(x[1], -4.0f, 4.0f));
    const v2f t = xc * xc;
    v2f p = mk2(-1.580784725e-09f, -1.580784725e-09f);
    p = p * t + 1.217110110e-07f;
    p = p * t + -4.100864317e-06f;
    p = p * t + 8.066737064e-05f;
    p = p * t + -1.048204256e-03f;
    p = p * t + 9.664873593e-03f;
    p = p * t + -6.617537886e-02f;
    p = p * t + 3.988475204e-01f;
    return x * (xc * p) + x * 0.5f;
}

__device__ __forceinline__ void store_bf16x4(bf16_t* p, const float4& v, int cnt) {       // p 8-byte aligned when cnt == 4
    if (cnt == 4) { *reinterpret_cast<uint2*>(p) = pack_bf16x4(v); return; }
    const float t[4] = {v.x, v.y, v.z, v.w};
    for (int i = 0; i < cnt; ++i) p[i] = (bf16_t)(pack_bf16x2(t[i], 0.0f) & 0xffffu);
}
__device__ __forceinline__ float4 load_f32x4(const float* p, int cnt) {
    if (cnt == 4) return *reinterpret_cast<const float4*>(p);
    float t[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int i = 0; i < cnt; ++i) t[i] = p[i];
    return make_float4(t[0], t[1], t[2], t[3]);
}
__device__ __forceinline__ void store_f32x4(float* p, const float4& v, int cnt) {
    if (cnt == 4) { *reinterpret_cast<float4*>(p) = v; return; }
    const float t[4] = {v.x, v.y, v.z, v.w};
    for (int i = 0; i < cnt; ++i) p[i] = t[i];
}

}  // namespace gemm16
}  // namespace ade
