// ade_gemm.h — fp32 GEMM on the MI355X matrix cores with functor operands (shared by the STFT operator, DFSMN and
// Mel-Band-Roformer).
//
//   C(m, n) = sum_k A(m, k) * B(k, n)        exact fp32 (v_mfma_f32_16x16x4_f32 = an fmaf chain over k)
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
// elements are never requested.  A k-contiguous operand may also offer
//        __device__ bool can_vec4(int K) const;  __device__ float4 vec4(int row, int k) const;     (k % 4 == 0, k + 3 < K)
// and is then fetched as whole 64-byte lines (4 lanes x float4 per row and slab) instead of 16 scalar loads per lane --
// the difference between being bound by the texture-cache line rate and by the matrix cores.
// One 256-thread workgroup computes a 128 x 128 tile; each of its 4 wavefronts owns a 64 x 64 quadrant as 4 x 4 MFMA tiles
// (64 accumulator VGPRs).  Slabs of 16 k are staged ROW-major in LDS, 20 floats per row (16 + 4 padding):
//   * the contraction index of MFMA step s is mapped to k = 4 g + s for lane group g = lane >> 4 (any bijection is legal
//     as long as A and B agree), so the four steps' operands of a lane are ONE ds_read_b128 at row * 20 + 4 g;
//   * 20 * j mod 64 takes 16 distinct multiples of 4 for j = 0..15, so the 16 lanes of a group cover all 64 banks once.
#pragma once
#include "ade_device.h"

#include <cstdint>
#include <type_traits>

namespace ade {
namespace gemm {

using namespace dev;

constexpr int kTM = 128, kTN = 128, kTK = 16;
constexpr int kRow = 20;               // LDS floats per staged row
#ifndef ADE_GEMM_KSUB
#define ADE_GEMM_KSUB 1
#endif
#ifndef ADE_GEMM_DOUBLE
#define ADE_GEMM_DOUBLE false
#endif
constexpr int kSub = ADE_GEMM_KSUB;                // 16-k slabs staged per barrier pair; 2 measured SLOWER (MossFormer in-projection 1384 -> 1616 us: the 40 KB of LDS cost a resident workgroup)
constexpr bool kDouble = ADE_GEMM_DOUBLE;        // two LDS slabs used alternately, ONE barrier per 16-k slab (needs kSub == 1): measured SLOWER on the final round-2 tree (MossFormer 960 -> 1007 ms,
                                       // Mel-Band 981 -> 1042 ms) -- 40 KB of LDS per workgroup costs a resident workgroup, which hides more than the second barrier costs
constexpr int kSlab = (kDouble ? 2 : kSub) * kTM * kRow;   // floats per operand staging area

// A Store may declare `static constexpr bool kCtx = true` and then provides, instead of operator()(m, n, v),
//     RowT row(int m) const;  ColT col(int n) const;  PreT pre(int m, int n, const RowT&) const;      (what it needs to READ: per row, per column, per element; `None` if nothing)
//     void operator()(int m, int n, float v, const RowT&, const ColT&, const PreT&) const;
// The tile epilogue calls the three readers in batches ahead of the writes (arguments always in range).
struct None {};
template <class T, class = void>
struct HasCtx : std::false_type {};
template <class T>
struct HasCtx<T, std::void_t<decltype(T::kCtx)>> : std::true_type {};

// A B operand read along n may declare `bool keep(int k) const`: its operator() is then a LOAD ONLY (from an address that is always in range) and the rows k it wants as
// zeros are zeroed when the slab goes to LDS -- behind the wait that write needs anyway.  (A select on the loaded VALUE inside operator() makes the compiler wait for the
// slab right after requesting it, ahead of the previous slab's MFMAs: the prefetch hides nothing.)
template <class T, class = void>
struct HasKeep : std::false_type {};
template <class T>
struct HasKeep<T, std::void_t<decltype(&T::keep)>> : std::true_type {};

// A context Store may ALSO declare `static constexpr bool kV4 = true` (round 5): the tile is then computed TRANSPOSED (the MFMA operands swapped: the same products, the same
// order of accumulation over k, the same bits), so that a lane's four accumulator registers are four CONSECUTIVE columns of one row, and the epilogue hands the store whole
// float4s -- 16 of them per lane and tile instead of 64 scalars, 4 row contexts instead of 16, nothing exchanged between lanes for a store that pairs neighbouring columns:
//     __host__ __device__ bool can_v4(int N) const;      (N, the leading dimension and the pointers allow 16-byte accesses; launch() falls back to the scalar form if not)
//     Col4T col4(int n) const;  Pre4T pre4(int m, int n, const RowT&) const;      (n % 4 == 0: what the store READS for columns n .. n + 3)
//     void store4(int m, int n, float4 v, const RowT&, const Col4T&, const Pre4T&) const;
// row(m) is shared with the scalar form, which stays for N % 4 != 0 (NoV4<ST> hides the flag).
template <class T, class = void>
struct HasV4 : std::false_type {};
template <class T>
struct HasV4<T, std::void_t<decltype(T::kV4)>> : std::integral_constant<bool, T::kV4> {};
template <class S>
struct NoV4 : S { static constexpr bool kV4 = false; };

template <class T, class = void>
struct HasVec4 : std::false_type {};
template <class T>
struct HasVec4<T, std::void_t<decltype(&T::vec4)>> : std::true_type {};

// one 128 x 128 tile of C at (m_blk, n_blk); As / Bs: the workgroup's two kSlab LDS slabs
template <class AL, class BL, class ST>
__device__ __forceinline__ void gemm_tile(const AL& a_of, const BL& b_of, const ST& store, int M, int N, int K, int m_blk, int n_blk,
                                          float* As_all, float* Bs_all) {
    constexpr int kSlabW = kTM * kRow;                 // words per staged 16-k slab
    constexpr int kS = kSub;
    auto put4 = [&](float* base, int row, int kk, const float4& v) { *reinterpret_cast<float4*>(base + row * kRow + kk) = v; };      // four consecutive k of one row
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int j16 = lane & 15, g = lane >> 4;
    v4f acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = v4f{0.0f, 0.0f, 0.0f, 0.0f};

    constexpr bool kT = HasV4<ST>::value;              // transposed tiles: lane (g, j16), register r of tile (i, j) is C[wm + 16 i + j16][wn + 16 j + 4 g + r]
    constexpr bool kAVec = AL::kAlongK && HasVec4<AL>::value, kBVec = !BL::kAlongN && HasVec4<BL>::value;
    bool a_rt = false, b_rt = false;
    if constexpr (kAVec) a_rt = a_of.can_vec4(K);
    if constexpr (kBVec) b_rt = b_of.can_vec4(K);

    // Software pipeline: the operand elements of slab k+1 are requested (into registers) before the 64 MFMAs of slab k
    // run and are written to LDS after them, so the HBM / L2 latency of a slab hides under the matrix work of the
    // previous one.  Which lane fetches which element depends on the operand's contiguous direction (see the header);
    // every variant leaves each lane with two runs of 4 consecutive k of one row, stored as two ds_write_b128.
    // The vector / scalar choice of each operand is a run-time property (alignment); the k loop is instantiated per combination and entered once, so
    // that each copy keeps only its own addressing live (one loop with both paths hoists the address registers of both).
    auto k_loop = [&](auto av_c, auto bv_c) {
    constexpr bool a_v4 = decltype(av_c)::value, b_v4 = decltype(bv_c)::value;
    float4 ra[kS][2], rb[kS][2];
    auto fetch = [&](int k0, int sub) {
        if (a_v4) {                         // lane = (row, quarter): 4 lanes read one row's 64-byte line; rows r and r + 64
            if constexpr (AL::kAlongK && HasVec4<AL>::value) {
                const int r = tid >> 2, k = k0 + 4 * (tid & 3);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int m = m_blk + r + 64 * h;                       // LOADS ONLY, from addresses that are always in range (see stash)
                    ra[sub][h] = a_of.vec4(m < M ? m : M - 1, k < K ? k : 0);
                }
            }
        } else {                            // lane = (row, half slab): 8 k of one row; consecutive lanes = consecutive k-halves / rows
            const int r = AL::kAlongK ? tid >> 1 : tid & 127, kh = (AL::kAlongK ? tid & 1 : tid >> 7) * 8, m = m_blk + r;
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = a_of(m < M ? m : M - 1, k0 + kh + u < K ? k0 + kh + u : 0);
            ra[sub][0] = make_float4(t[0], t[1], t[2], t[3]);
            ra[sub][1] = make_float4(t[4], t[5], t[6], t[7]);
        }
        if (b_v4) {
            if constexpr (!BL::kAlongN && HasVec4<BL>::value) {
                const int c = tid >> 2, k = k0 + 4 * (tid & 3);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int n = n_blk + c + 64 * h;
                    rb[sub][h] = b_of.vec4(n < N ? n : N - 1, k < K ? k : 0);
                }
            }
        } else if (BL::kAlongN) {           // lane = (column, half slab): consecutive lanes = consecutive columns
            const int c = tid & 127, kh = (tid >> 7) * 8, n = n_blk + c;
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = b_of(k0 + kh + u < K ? k0 + kh + u : 0, n < N ? n : N - 1);
            rb[sub][0] = make_float4(t[0], t[1], t[2], t[3]);
            rb[sub][1] = make_float4(t[4], t[5], t[6], t[7]);
        } else {                            // lane = (k, column group): consecutive lanes = consecutive k of one column; columns cg + 16 u
            const int kk = tid & 15, cg = tid >> 4;
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int n = n_blk + cg + 16 * u;
                t[u] = b_of(k0 + kk < K ? k0 + kk : 0, n < N ? n : N - 1);
            }
            rb[sub][0] = make_float4(t[0], t[1], t[2], t[3]);
            rb[sub][1] = make_float4(t[4], t[5], t[6], t[7]);
        }
    };
    // stash(sub, buf, k0): registers -> row-major LDS slabs (same lane maps as fetch); k0 = the slab's first k.  What a fetch read from a clamped address is dealt with
    // HERE, behind the wait the LDS write needs anyway: rows beyond M / N are left as they are (the last row again: their products are never stored), the k >= K part of
    // the last slab is zeroed in both operands.  (Zeroing at the load -- select(ok, load, 0) -- made the compiler wait for the slab right after requesting it, ahead of the
    // previous slab's MFMAs: the prefetch hid nothing.)
    auto stash = [&](int sub, int buf, int k0) {
        float* As = As_all + buf * kSlabW;
        float* Bs = Bs_all + buf * kSlabW;
        if (k0 + kTK > K) {                                   // wave-uniform: only the last slab of a K that is not a multiple of 16
            auto cut = [&](float4 (&v)[2], int first, int step) {     // v[h] holds k = first + step h .. + 3
                float t[8] = {v[0].x, v[0].y, v[0].z, v[0].w, v[1].x, v[1].y, v[1].z, v[1].w};
#pragma unroll
                for (int u = 0; u < 8; ++u) t[u] = first + (u >> 2) * step + (u & 3) < K ? t[u] : 0.0f;
                v[0] = make_float4(t[0], t[1], t[2], t[3]);
                v[1] = make_float4(t[4], t[5], t[6], t[7]);
            };
            if (a_v4) { const bool in = k0 + 4 * (tid & 3) < K; ra[sub][0] = keep4(in, ra[sub][0]); ra[sub][1] = keep4(in, ra[sub][1]); }
            else cut(ra[sub], k0 + (AL::kAlongK ? tid & 1 : tid >> 7) * 8, 4);
            if (b_v4) { const bool in = k0 + 4 * (tid & 3) < K; rb[sub][0] = keep4(in, rb[sub][0]); rb[sub][1] = keep4(in, rb[sub][1]); }
            else if (BL::kAlongN) cut(rb[sub], k0 + (tid >> 7) * 8, 4);
            else { const bool in = k0 + (tid & 15) < K; rb[sub][0] = keep4(in, rb[sub][0]); rb[sub][1] = keep4(in, rb[sub][1]); }
        }
        if (a_v4) {
            const int r = tid >> 2, kq = 4 * (tid & 3);
            put4(As, r, kq, ra[sub][0]);
            put4(As, r + 64, kq, ra[sub][1]);
        } else {
            const int r = AL::kAlongK ? tid >> 1 : tid & 127, kh = (AL::kAlongK ? tid & 1 : tid >> 7) * 8;
            put4(As, r, kh, ra[sub][0]);
            put4(As, r, kh + 4, ra[sub][1]);
        }
        if (b_v4) {
            const int c = tid >> 2, kq = 4 * (tid & 3);
            put4(Bs, c, kq, rb[sub][0]);
            put4(Bs, c + 64, kq, rb[sub][1]);
        } else if (BL::kAlongN) {
            const int c = tid & 127, kh = (tid >> 7) * 8;
            if constexpr (HasKeep<BL>::value) {
                const int kb = k0 + kh;
                rb[sub][0] = make_float4(b_of.keep(kb) ? rb[sub][0].x : 0.0f, b_of.keep(kb + 1) ? rb[sub][0].y : 0.0f, b_of.keep(kb + 2) ? rb[sub][0].z : 0.0f,
                                         b_of.keep(kb + 3) ? rb[sub][0].w : 0.0f);
                rb[sub][1] = make_float4(b_of.keep(kb + 4) ? rb[sub][1].x : 0.0f, b_of.keep(kb + 5) ? rb[sub][1].y : 0.0f, b_of.keep(kb + 6) ? rb[sub][1].z : 0.0f,
                                         b_of.keep(kb + 7) ? rb[sub][1].w : 0.0f);
            }
            put4(Bs, c, kh, rb[sub][0]);
            put4(Bs, c, kh + 4, rb[sub][1]);
        } else {
            const int kk = tid & 15, cg = tid >> 4;
            const float t[8] = {rb[sub][0].x, rb[sub][0].y, rb[sub][0].z, rb[sub][0].w, rb[sub][1].x, rb[sub][1].y, rb[sub][1].z, rb[sub][1].w};
#pragma unroll
            for (int u = 0; u < 8; ++u) Bs[(cg + 16 * u) * kRow + kk] = t[u];
        }
    };
    auto compute = [&](int buf) {
            const float* As = As_all + buf * kSlabW;
            const float* Bs = Bs_all + buf * kSlabW;
            float4 a4[4];                   // lane (g, j16): A[row 16 i + j16][k = 4 g + s], B[k = 4 g + s][col 16 j + j16], s = 0..3
#pragma unroll
            for (int i = 0; i < 4; ++i) a4[i] = *reinterpret_cast<const float4*>(As + (wm + 16 * i + j16) * kRow + 4 * g);
            // one B operand ahead: the ds_read of column tile j + 1 is issued BEFORE the 16 MFMAs of tile j (a second float4; the scheduling barrier keeps the compiler from
            // sinking it behind them again -- with one register set it placed every read after the products that still used the set and waited for it with an idle pipe)
            float4 bq[2];
            bq[0] = *reinterpret_cast<const float4*>(Bs + (wn + j16) * kRow + 4 * g);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (j < 3) bq[(j + 1) & 1] = *reinterpret_cast<const float4*>(Bs + (wn + 16 * (j + 1) + j16) * kRow + 4 * g);
                __builtin_amdgcn_sched_barrier(0);
                const float4 b4 = bq[j & 1];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if constexpr (kT) {
                        acc[i][j] = mfma16x16x4(b4.x, a4[i].x, acc[i][j]);
                        acc[i][j] = mfma16x16x4(b4.y, a4[i].y, acc[i][j]);
                        acc[i][j] = mfma16x16x4(b4.z, a4[i].z, acc[i][j]);
                        acc[i][j] = mfma16x16x4(b4.w, a4[i].w, acc[i][j]);
                    } else {
                        acc[i][j] = mfma16x16x4(a4[i].x, b4.x, acc[i][j]);
                        acc[i][j] = mfma16x16x4(a4[i].y, b4.y, acc[i][j]);
                        acc[i][j] = mfma16x16x4(a4[i].z, b4.z, acc[i][j]);
                        acc[i][j] = mfma16x16x4(a4[i].w, b4.w, acc[i][j]);
                    }
                }
            }
    };
    if constexpr (kDouble) {
        static_assert(!kDouble || kSub == 1, "the double-buffered loop stages one f32 slab at a time");
        fetch(0, 0);
        stash(0, 0, 0);
        __syncthreads();
        int cur = 0;
        for (int k0 = 0; k0 < K; k0 += kTK) {
            const bool more = k0 + kTK < K;
            if (more) fetch(k0 + kTK, 0);
            compute(cur);
            if (more) stash(0, cur ^ 1, k0 + kTK);   // the other slab: last read in the previous iteration, behind that iteration's barrier
            __syncthreads();
            cur ^= 1;
        }
    } else {
#pragma unroll
        for (int sub = 0; sub < kS; ++sub) fetch(kTK * sub, sub);
        for (int k0 = 0; k0 < K; k0 += kTK * kS) {
#pragma unroll
            for (int sub = 0; sub < kS; ++sub) stash(sub, sub, k0 + kTK * sub);
            __syncthreads();
            if (k0 + kTK * kS < K) {
#pragma unroll
                for (int sub = 0; sub < kS; ++sub) fetch(k0 + kTK * (kS + sub), sub);
            }
#pragma unroll
            for (int sub = 0; sub < kS; ++sub) {
                if (sub > 0 && k0 + kTK * sub >= K) break;                    // the tail of K (uniform)
                compute(sub);
            }
            __syncthreads();
        }
    }
    };
    if constexpr (kAVec && kBVec) {
        if (a_rt && b_rt) k_loop(std::true_type{}, std::true_type{});
        else if (a_rt) k_loop(std::true_type{}, std::false_type{});
        else if (b_rt) k_loop(std::false_type{}, std::true_type{});
        else k_loop(std::false_type{}, std::false_type{});
    } else if constexpr (kAVec) {
        if (a_rt) k_loop(std::true_type{}, std::false_type{});
        else k_loop(std::false_type{}, std::false_type{});
    } else if constexpr (kBVec) {
        if (b_rt) k_loop(std::false_type{}, std::true_type{});
        else k_loop(std::false_type{}, std::false_type{});
    } else {
        k_loop(std::false_type{}, std::false_type{});
    }
    if constexpr (kT) {
        // transposed tiles, float4 stores: per 16-row band ONE row context, then the four float4s' reads in a batch, then the four stores
        auto emit4 = [&](auto guard_c) {
            constexpr bool G = decltype(guard_c)::value;
            decltype(store.col4(0)) cc[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = n_blk + wn + 16 * j + 4 * g;
                cc[j] = store.col4(G && n >= N ? 0 : n);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = m_blk + wm + 16 * i + j16, mc = G && m >= M ? M - 1 : m;
                const auto rc = store.row(mc);
                decltype(store.pre4(0, 0, rc)) pc[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int n = n_blk + wn + 16 * j + 4 * g;
                    pc[j] = store.pre4(mc, G && n >= N ? 0 : n, rc);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int n = n_blk + wn + 16 * j + 4 * g;
                    if (!G || (m < M && n < N)) store.store4(m, n, make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]), rc, cc[j], pc[j]);
                }
            }
        };
        if (m_blk + kTM <= M && n_blk + kTN <= N) emit4(std::false_type{});
        else emit4(std::true_type{});
    } else
    // lane (g, j16), register r of tile (i, j) is C[wm + 16 i + 4 g + r][wn + 16 j + j16]
    if constexpr (HasCtx<ST>::value) {
        // context form (see the header): everything the store READS is requested in batches -- 4 column contexts, then per 16-row band 4 row contexts and 16
        // per-element values -- before the band's 16 writes, and interior tiles run without bounds checks.  With the plain form every element is load, wait, store.
        auto emit = [&](auto guard_c) {
            constexpr bool G = decltype(guard_c)::value;
            decltype(store.col(0)) cc[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = n_blk + wn + 16 * j + j16;
                cc[j] = store.col(G && n >= N ? N - 1 : n);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                decltype(store.row(0)) rc[4];
                decltype(store.pre(0, 0, store.row(0))) pc[4][4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m_blk + wm + 16 * i + 4 * g + r, mc = G && m >= M ? M - 1 : m;
                    rc[r] = store.row(mc);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int n = n_blk + wn + 16 * j + j16;
                        pc[r][j] = store.pre(mc, G && n >= N ? N - 1 : n, rc[r]);
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int m = m_blk + wm + 16 * i + 4 * g + r, n = n_blk + wn + 16 * j + j16;
                        if (!G || (m < M && n < N)) store(m, n, acc[i][j][r], rc[r], cc[j], pc[r][j]);
                    }
            }
        };
        if (m_blk + kTM <= M && n_blk + kTN <= N) emit(std::false_type{});
        else emit(std::true_type{});
    } else {
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
}

// Workgroups are dispatched round-robin over the 8 XCDs (id mod 8), each with its own 4 MB L2.  Renumbering them so that
// CONSECUTIVE logical ids share an XCD keeps the tiles that re-read one strip of A (all n-tiles of an m-tile) and the whole
// B operand of a batch entry inside one L2 instead of fetching them over the fabric from eight.  (Measured neutral on the
// Mel-Band-Roformer shapes -- the 256 MB Infinity Cache already absorbs those re-reads -- kept because it is never worse.)
__device__ __forceinline__ int xcd_contiguous_id(int w, int total) {
    const int per = total >> 3, rem = total & 7, xcd = w & 7, idx = w >> 3;
    return (xcd < rem ? xcd * (per + 1) : rem * (per + 1) + (xcd - rem) * per) + idx;
}

template <class AL, class BL, class ST>
__global__ __launch_bounds__(256, 4) void k_gemm128(AL a_of, BL b_of, ST store, int M, int N, int K) {
    __shared__ __attribute__((aligned(16))) float As[kSlab];
    __shared__ __attribute__((aligned(16))) float Bs[kSlab];
    const int gx = (int)gridDim.x, id = xcd_contiguous_id((int)blockIdx.x + gx * (int)blockIdx.y, gx * (int)gridDim.y);
    gemm_tile(a_of, b_of, store, M, N, K, (id / gx) * kTM, (id % gx) * kTN, As, Bs);
}

template <class AL, class BL, class ST>
inline void launch(hipStream_t s, const AL& a, const BL& b, const ST& st, int M, int N, int K) {
    if constexpr (HasV4<ST>::value) {
        if (!st.can_v4(N)) { launch(s, a, b, NoV4<ST>{st}, M, N, K); return; }
    }
    const dim3 grid((unsigned)((N + kTN - 1) / kTN), (unsigned)((M + kTM - 1) / kTM));
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gemm128<AL, BL, ST>), grid, dim3(256), 0, s, a, b, st, M, N, K);
}

// Batched form: blockIdx.z selects a problem.  `prob(z)` (evaluated once per workgroup, so its table reads are scalar loads)
// returns a struct with members a, b, st (functors as above) and M, N, K; problems may differ in every one of them -- tiles
// outside a problem's own M x N exit at once, the grid is sized for the largest.
template <class P>
__global__ __launch_bounds__(256, 4) void k_gemm128_batched(P prob) {
    __shared__ __attribute__((aligned(16))) float As[kSlab];
    __shared__ __attribute__((aligned(16))) float Bs[kSlab];
    const int gx = (int)gridDim.x, gy = (int)gridDim.y;
    const int id = xcd_contiguous_id((int)blockIdx.x + gx * ((int)blockIdx.y + gy * (int)blockIdx.z), gx * gy * (int)gridDim.z);
    const int z = id / (gx * gy), in_z = id - z * gx * gy;
    const auto q = prob(z);
    const int m_blk = (in_z / gx) * kTM, n_blk = (in_z % gx) * kTN;
    if (m_blk >= q.M || n_blk >= q.N) return;
    gemm_tile(q.a, q.b, q.st, q.M, q.N, q.K, m_blk, n_blk, As, Bs);
}

template <class P>
inline void launch_batched(hipStream_t s, const P& prob, int batch, int max_M, int max_N) {
    const dim3 grid((unsigned)((max_N + kTN - 1) / kTN), (unsigned)((max_M + kTM - 1) / kTM), (unsigned)batch);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gemm128_batched<P>), grid, dim3(256), 0, s, prob);
}

// ---- common functors ------------------------------------------------------------------------------------------
__device__ __forceinline__ bool vec4_ok(const float* p, int ld, int K) {
    return ((K | ld) & 3) == 0 && (reinterpret_cast<uintptr_t>(p) & 15) == 0;
}
struct RowMajorA {      // A(m, k) = p[m * ld + k]
    static constexpr bool kAlongK = true;
    const float* p;
    int ld;
    __device__ float operator()(int m, int k) const { return p[(size_t)m * ld + k]; }
    __device__ bool can_vec4(int K) const { return vec4_ok(p, ld, K); }
    __device__ float4 vec4(int m, int k) const { return *reinterpret_cast<const float4*>(p + (size_t)m * ld + k); }
};
struct WeightNK {       // B(k, n) = p[n * ld + k]: a torch Linear weight (out_features, in_features) used as x @ W^T
    static constexpr bool kAlongN = false;
    const float* p;
    int ld;
    __device__ float operator()(int k, int n) const { return p[(size_t)n * ld + k]; }
    __device__ bool can_vec4(int K) const { return vec4_ok(p, ld, K); }
    __device__ float4 vec4(int n, int k) const { return *reinterpret_cast<const float4*>(p + (size_t)n * ld + k); }
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
    static constexpr bool kCtx = true;
    float* p;
    int ld;
    const float* bias;
    float floor_;       // kActLogFloor: log(max(v, floor_))
    __device__ float row(int m) const { return bias ? bias[m] : 0.0f; }
    __device__ None col(int) const { return None{}; }
    __device__ None pre(int, int, float) const { return None{}; }
    __device__ void operator()(int m, int n, float v, float b, None, None) const {
        v += b;
        if (ACT == kActRelu) v = v > 0.0f ? v : 0.0f;
        if (ACT == kActSigmoid) v = 1.0f / (1.0f + expf(-v));
        if (ACT == kActLogFloor) v = logf(v > floor_ ? v : floor_);
        p[(size_t)m * ld + n] = v;
    }
};

}  // namespace gemm
}  // namespace ade
