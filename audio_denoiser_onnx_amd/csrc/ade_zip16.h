// ade_zip16.h — ZipEnhancer's bf16 path (BASELINE.json configs[2]: "bf16 dual-path transformer ... (MFMA attention)"): bf16 activations and weights STORED in HBM for the
// causal dense blocks, the feed-forward modules, the K = 64 projections and the attention / convolution-module operands; fp32 exactly where the reference's own
// reduced-precision plan keeps it (ZipEnhancer/Optimize_ONNX.py:25-64: the RMS front, the magnitude compression, the InstanceNorm statistics, the PCM tail) plus every
// accumulator, the softmax statistics and the residual stream.  All products run on gfx950's full-rate v_mfma_f32_32x32x16_bf16 (csrc/ade_gemm16.h has the operand maps:
// lane l supplies row (l & 31), k = 8 (l >> 5) .. + 7 of a 16-deep step; D[i][j] sits in lane (j = l & 31, h = l >> 5), register r = row i = (r & 3) + 8 (r >> 2) + 4 h).
//
// Kernels (each cites the reference lines of the f32 kernel it mirrors in ade_zipenhancer.hip):
//   k_zip_dense16     one layer of a causal dense block as a token-tiled implicit GEMM, 256 tokens x 64 channels per workgroup, the three frequency taps of a time tap share
//                     one staged operand; tiles never straddle a window, so the epilogue also emits the layer's InstanceNorm partial sums (the f32 path's separate
//                     statistics pass over the output is gone)
//   k_zip_hist_norm16 raw fp32 layer output -> InstanceNorm + PReLU -> the bf16 dense history
//   k_zip_ff16        a whole feed-forward module (64 -> fd SwooshL -> 64 + residual form): the hidden tile feeds the second product from the accumulator registers
//   k_rows16          the K <= 192 products around the attention / convolution modules: rows straight from global memory into the matrix cores' registers (fp32 rows are
//                     rounded on the way), weights staged once, full-line stores through LDS
#pragma once
#include <type_traits>
#include "ade_gemm16.h"

namespace ade {
namespace zip16 {

using namespace dev;
using gemm16::bf16_t;
using gemm16::ld8_or_zero;
using gemm16::mfma32x32x16;
using gemm16::pack_bf16x2;
using gemm16::v16f;
using gemm16::zero_unless;

// F.softplus (threshold 20) straight on the transcendental units: ln(1 + e^x) = ln 2 * v_log_f32(1 + v_exp_f32(x log2 e)) -- no range fix-ups (1 + e^x lies in [1, 5e8] where it
// is used; e^x flushing to 0 below -87 gives the exact limit).  __expf / __logf wrap the same two instructions in denormal scaling and compares: 28 VALU instructions per
// activation in k_zip_ff16 (4554 per wavefront against 80 matrix instructions, profiles/r05_i_zip_bf16_pmc_summary.txt: the module was VALU-bound), ~9 this way.
__device__ __forceinline__ float softplus16(float x) {
    const float sp = 0.6931471805599453f * __builtin_amdgcn_logf(1.0f + __builtin_amdgcn_exp2f(x * kLog2e));
    return x > 20.0f ? x : sp;
}
__device__ __forceinline__ float swoosh_l16(float x) { return softplus16(x - 4.0f) - 0.08f * x; }       // (Export_ZipEnhancer.py:135-136)
__device__ __forceinline__ float swoosh_r16(float x) { return softplus16(x - 1.0f) - 0.08f * x; }       // (:138)

__device__ __forceinline__ float sigmoid16(float x) { return 1.0f / (1.0f + __expf(-x)); }                // (the convolution module's GLU gate, :322-323: the form its staging used)
__device__ __forceinline__ uint4 pack8(const float4& a, const float4& b) { return make_uint4(pack_bf16x2(a.x, a.y), pack_bf16x2(a.z, a.w), pack_bf16x2(b.x, b.y), pack_bf16x2(b.z, b.w)); }
__device__ __forceinline__ void unpack8(const uint4& u, float* v) {
    v[0] = gemm16::bf16_lo(u.x); v[1] = gemm16::bf16_hi(u.x); v[2] = gemm16::bf16_lo(u.y); v[3] = gemm16::bf16_hi(u.y);
    v[4] = gemm16::bf16_lo(u.z); v[5] = gemm16::bf16_hi(u.z); v[6] = gemm16::bf16_lo(u.w); v[7] = gemm16::bf16_hi(u.w);
}

// ---- one layer of a causal dense block (Export_ZipEnhancer.py:701-757), bf16 operands -------------------------------------------------------------------------------
// A(token, k): k = (kt, channel block, kf, channel); the value is input channel ci of position (t - (1 - kt) dil, f + kf - 1), zero outside the map.  Input channels are
// [this group's newer dense outputs (hist, normalised + PReLU'd bf16) ..., block input (inp, bf16 [tokens][64])].  w: [64 co][6 taps][cin] bf16.
// A stage = (time tap kt, 32 input channels): rows m_blk - 1 .. m_blk + 256 of that channel block are staged ONCE (64 bytes per row at an 80-byte pitch: the 16 lanes of a
// ds_read_b128 service group land on 16 distinct 4-bank sets) and the three kf products read it shifted by one row; neighbours outside the map are zeroed in the operand
// register.  Grid: (blocks per window) x windows through the XCD-contiguous map; a tile's rows all belong to ONE window, so the per-channel sum and sum of squares of
// (product + bias) over the tile's rows are that window's InstanceNorm partial sums: partial[((win * nblk + blk) * 64 + c) * 2 + {sum, sumsq}] (fp64, k_zip_stats_final's
// layout).  raw: [tokens][64] fp32 (product + bias).
constexpr int kDARows = 258;
// CB = input channels per stage: 32 (80-byte rows, 38 KB of LDS: four workgroups per CU) or 64 (144-byte rows, 65 KB: two per CU, but half as many barriers and load round
// trips per product -- the stage's loads are a microsecond or two away and one stage of look-ahead does not cover them)
// HALF: the block's operands (history, input, weights) are IEEE half instead of bf16 -- the engine's default for the dense blocks (ZipEngine::dense_half): their activations are
// InstanceNorm'd + PReLU'd values of order one, and bf16's eight significant bits in these three blocks cost the waveform 10 dB (profiles/r06_e_zip_bf16_budget.txt).
template <int CB, bool HALF>
__global__ __launch_bounds__(256, CB == 32 ? 4 : 2) void k_zip_dense16(const bf16_t* __restrict__ hist, const bf16_t* __restrict__ inp, int hist_ld, int hist_off, int hist_n, int cin, int T,
                                                        int F, int dil, const bf16_t* __restrict__ w, const float* __restrict__ bias, float* __restrict__ raw,
                                                        double* __restrict__ partial, int nblk) {
    constexpr int kDPitch = 2 * CB + 16, kPieces = CB / 8, kU = CB / 32;          // bytes per staged row; 16-byte pieces per row; pieces per lane and row
    __shared__ __attribute__((aligned(16))) unsigned char As[kDARows * kDPitch];
    __shared__ __attribute__((aligned(16))) unsigned char Bs[3 * 64 * kDPitch];
    double (*red)[64][2] = reinterpret_cast<double (*)[64][2]>(Bs);        // [4][64][2]: the weights' buffer, dead once the stage loop is over
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave * 64, l31 = lane & 31, h = lane >> 5;
    const int id = gemm16::xcd_contiguous_id((int)blockIdx.x, (int)gridDim.x), win = id / nblk, blk = id - win * nblk;
    const int TF = T * F, m_blk = blk * 256;
    const size_t wbase = (size_t)win * TF;
    const bf16_t* const hist_w = hist + wbase * hist_ld + hist_off;
    const bf16_t* const inp_w = inp + wbase * 64;
    // staged rows of this lane: r = (tid >> 2) + 64 hh, hh < 4, and for tid < 8 the two halo rows 256, 257; token (inside the window) = m_blk - 1 + r
    const int sr = tid >> 2, pc = tid & 3;
    int stok[5];
    bool sok0[5], sok1[5];
#pragma unroll
    for (int hh = 0; hh < 5; ++hh) {
        const int r = hh < 4 ? sr + 64 * hh : 256 + sr;
        const int m = m_blk - 1 + r;
        const bool in = m >= 0 && m < TF && (hh < 4 || tid < 8);
        const int mc = in ? m : 0, t = mc / F;
        stok[hh] = mc;
        sok1[hh] = in;                               // kt = 1: the row itself
        sok0[hh] = in && t >= dil;                   // kt = 0: the row dil frames earlier, inside the window
    }
    unsigned left = 0, right = 0;                    // bit i: row wm + 32 i + l31 has a left (kf = 0) / right (kf = 2) neighbour
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m_blk + wm + 32 * i + l31, f = m % F;
        if (f != 0) left |= 1u << i;
        if (f != F - 1) right |= 1u << i;
    }
    v16f acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    const int ncb = cin / CB, nstage = 2 * ncb;
    uint4 ra[5][kU], rb0[kU], rb1[kU], rb2[kU];
    auto fetch = [&](int st) __attribute__((always_inline)) {
        const int kt = st / ncb, ci0 = (st - kt * ncb) * CB;
        const bool from_hist = ci0 < hist_n;
        const bf16_t* src = from_hist ? hist_w + ci0 + 8 * kU * pc : inp_w + (ci0 - hist_n) + 8 * kU * pc;
        const int ld = from_hist ? hist_ld : 64, shift = kt ? 0 : dil * F;
#pragma unroll
        for (int hh = 0; hh < 5; ++hh) {
            const bool ok = kt ? sok1[hh] : sok0[hh];
#pragma unroll
            for (int u = 0; u < kU; ++u) ra[hh][u] = ld8_or_zero(ok, src + (size_t)(stok[hh] - (ok ? shift : 0)) * ld + (ok ? 8 * u : 0));       // (zeros by address: the loads stay in flight across this stage's products)
        }
        const bf16_t* wp = w + (size_t)sr * (6 * cin) + (size_t)(kt * 3) * cin + ci0 + 8 * kU * pc;
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            rb0[u] = *reinterpret_cast<const uint4*>(wp + 8 * u);
            rb1[u] = *reinterpret_cast<const uint4*>(wp + cin + 8 * u);
            rb2[u] = *reinterpret_cast<const uint4*>(wp + 2 * cin + 8 * u);
        }
    };
    fetch(0);
    for (int st = 0; st < nstage; ++st) {
#pragma unroll
        for (int hh = 0; hh < 4; ++hh)
#pragma unroll
            for (int u = 0; u < kU; ++u) *reinterpret_cast<uint4*>(As + (sr + 64 * hh) * kDPitch + 16 * (kU * pc + u)) = ra[hh][u];
        if (tid < 8) {
#pragma unroll
            for (int u = 0; u < kU; ++u) *reinterpret_cast<uint4*>(As + (256 + sr) * kDPitch + 16 * (kU * pc + u)) = ra[4][u];
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            *reinterpret_cast<uint4*>(Bs + sr * kDPitch + 16 * (kU * pc + u)) = rb0[u];
            *reinterpret_cast<uint4*>(Bs + (64 + sr) * kDPitch + 16 * (kU * pc + u)) = rb1[u];
            *reinterpret_cast<uint4*>(Bs + (128 + sr) * kDPitch + 16 * (kU * pc + u)) = rb2[u];
        }
        __syncthreads();
        if (st + 1 < nstage) fetch(st + 1);
#pragma unroll
        for (int kf = 0; kf < 3; ++kf) {
            const unsigned mask = kf == 0 ? left : (kf == 2 ? right : 3u);
#pragma unroll
            for (int ks = 0; ks < CB / 16; ++ks) {
                uint4 fa[2], fb[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) fa[i] = zero_unless((mask >> i) & 1u, *reinterpret_cast<const uint4*>(As + (wm + 32 * i + l31 + kf) * kDPitch + 32 * ks + 16 * h));
#pragma unroll
                for (int j = 0; j < 2; ++j) fb[j] = *reinterpret_cast<const uint4*>(Bs + (kf * 64 + 32 * j + l31) * kDPitch + 32 * ks + 16 * h);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = gemm16::mfma32x32x16_t<HALF>(fa[i], fb[j], acc[i][j]);      // D[token][channel]
            }
        }
        __syncthreads();
    }
    // lane (channel 32 j + l31, h): register r of tile (i, j) is token wm + 32 i + (r & 3) + 8 (r >> 2) + 4 h -- a store instruction writes 32 consecutive channels of two tokens
    float* const raw_w = raw + wbase * 64;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int co = 32 * j + l31;
        const float b = bias[co];
        double s = 0.0, ss = 0.0;                    // (fp64 from the first addend: E[x^2] - mean^2 of a channel with a large mean does not survive fp32 sums)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m_blk + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (m < TF) {
                    const float v = acc[i][j][r] + b;
                    raw_w[(size_t)m * 64 + co] = v;
                    s += (double)v;
                    ss = fma((double)v, (double)v, ss);
                }
            }
        s += shfl_xor_f64(s, 32);
        ss += shfl_xor_f64(ss, 32);
        if (h == 0) { red[wave][co][0] = s; red[wave][co][1] = ss; }
    }
    __syncthreads();
    if (tid < 128) {
        const int c = tid >> 1, q = tid & 1;
        const double v = (red[0][c][q] + red[1][c][q]) + (red[2][c][q] + red[3][c][q]);
        partial[(((size_t)win * nblk + blk) * 64 + c) * 2 + q] = v;
    }
}

// raw fp32 layer output -> InstanceNorm + PReLU -> bf16 dense history [tokens][ld] at channels ch0 .. ch0 + 63; optionally the fp32 values too (out32, [tokens][64]: the last
// layer's output while a consumer still reads fp32).  nrm: [(window * nrm_ld + channel) * 2 + {scale, shift}].  thread = (token, channel quad)
template <bool HALF>
__global__ __launch_bounds__(256) void k_zip_hist_norm16(const float* __restrict__ raw, bf16_t* __restrict__ hist, int ld, int ch0, const float* __restrict__ nrm, int nrm_ld,
                                                         const float* __restrict__ slope, int tok_per_win, float* __restrict__ out32, long long total16) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total16) return;
    const long long tok = i >> 4;
    const int c = (int)(i & 15) * 4, ch = ch0 + c, b = (int)(tok / tok_per_win);
    float4 v = *reinterpret_cast<const float4*>(raw + tok * 64 + c);
    const float* k = nrm + ((size_t)b * nrm_ld + ch) * 2;
    const float4 s0 = *reinterpret_cast<const float4*>(k), s1 = *reinterpret_cast<const float4*>(k + 4), sl = *reinterpret_cast<const float4*>(slope + ch);
    v.x = prelu_f(v.x * s0.x + s0.y, sl.x);
    v.y = prelu_f(v.y * s0.z + s0.w, sl.y);
    v.z = prelu_f(v.z * s1.x + s1.y, sl.z);
    v.w = prelu_f(v.w * s1.z + s1.w, sl.w);
    *reinterpret_cast<uint2*>(hist + tok * ld + ch) = gemm16::pack16x4_t<HALF>(v);
    if (out32) *reinterpret_cast<float4*>(out32 + tok * 64 + c) = v;
}

// fp32 [n] -> bf16 [n] (n % 4 == 0): the decoder pair's dense-block input is the last encoder's fp32 output
template <bool HALF>
__global__ __launch_bounds__(256) void k_zip_to_bf16(const float* __restrict__ x, bf16_t* __restrict__ y, long long n4) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    *reinterpret_cast<uint2*>(y + 4 * i) = gemm16::pack16x4_t<HALF>(*reinterpret_cast<const float4*>(x + 4 * i));
}

// ---- fused feed-forward module (:160, :170-174) on bf16 operands: out = epilogue(W2 swooshL(W1 x + b1) + b2) --------------------------------------------------------
// A 256-thread workgroup owns 128 rows (32 per wavefront).  x is read ONCE, fp32, straight into the registers the matrix cores read (rounded to bf16 there); the weights stream
// through LDS 64 hidden units at a time (W1 rows and W2 columns of the chunk, 144-byte pitch), double-buffered with one barrier per chunk.  Both products are formed
// TRANSPOSED, so the hidden tile never changes lanes:
//   H^T (32 hidden x 32 rows) = W1 (32 x 64) . X^T : lane (row j, h) ends with hidden units (r & 3) + 8 (r >> 2) + 4 h of ITS row in the sixteen accumulator registers;
//   Y^T (32 out x 32 rows)   += W2 (32 x 16 hidden) . act(H^T): the B operand of a 16-deep step is the lane's OWN eight registers 8 s .. 8 s + 7 (bias + SwooshL applied, rounded
//   to bf16) -- i.e. step s contracts over hidden 16 s + 8 (e >> 2) + 4 h + (e & 3), e < 8, and W2's columns are stored in that order (w2p: the host permutes every group of
//   16 hidden units: position 8 h + e holds unit 8 (e >> 2) + 4 h + (e & 3)), so its A operand is one ds_read_b128.
// The output tile goes through LDS (the weight buffers, dead by then) so that the residual reads and the stores are whole 256-byte rows.
// MODE 0: out = res + ff   1: out = xin + ff   2: out = res + ((xin + ff) - res) * cmid      (as k_zip_ff)
// MODE 3: the layer's LAST module with the layer's final norm (:175-183) in the same store: y = xin + ff;  out = y / |y - nb|_2 * fs + res * rs  (res = the layer input, out = the
//         layer output; cmid carries nb | fs | rs, 64 floats each) -- the row's sum of squares meets through four shuffles among the 16 lanes that own the row.
constexpr int kF16Pitch = 144, kF16Buf = 2 * 64 * kF16Pitch, kF16EPitch = 68;
constexpr int kF16Lds = 2 * kF16Buf > 4 * 32 * kF16EPitch * 4 ? 2 * kF16Buf : 4 * 32 * kF16EPitch * 4;
template <int MODE>
__global__ __launch_bounds__(256) void k_zip_ff16(const float* xin, const bf16_t* __restrict__ w1, const float* __restrict__ b1, const bf16_t* __restrict__ w2p,
                                                  const float* __restrict__ b2, const float* res, const float* __restrict__ cmid, float* out, int M, int fd) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[kF16Lds];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
    const int row0 = (int)blockIdx.x * 128 + wave * 32, row = row0 + l31;
    uint4 xb0, xb1, xb2, xb3;          // (named, not an array: as arrays these operands and the prefetched weight pieces below were kept in scratch memory by the compiler)
    {
        const float* src = xin + (size_t)(row < M ? row : M - 1) * 64 + 8 * h;
        float4 t[8];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { t[2 * ks] = *reinterpret_cast<const float4*>(src + 16 * ks); t[2 * ks + 1] = *reinterpret_cast<const float4*>(src + 16 * ks + 4); }
        xb0 = pack8(t[0], t[1]); xb1 = pack8(t[2], t[3]); xb2 = pack8(t[4], t[5]); xb3 = pack8(t[6], t[7]);
    }
    v16f acc2[2];
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[jt][r] = 0.0f;
    // staging: chunk cg of W1 = rows 64 cg .. + 63 (128 bytes each), of W2p = columns 64 cg .. + 63 of its 64 rows: 512 16-byte pieces each, two + two per thread
    const int sr = tid >> 3, sp = tid & 7;
    uint4 p1a, p1b, p2a, p2b;
    auto request = [&](int cg) __attribute__((always_inline)) {
        p1a = *reinterpret_cast<const uint4*>(w1 + (size_t)(64 * cg + sr) * 64 + 8 * sp);
        p1b = *reinterpret_cast<const uint4*>(w1 + (size_t)(64 * cg + sr + 32) * 64 + 8 * sp);
        p2a = *reinterpret_cast<const uint4*>(w2p + (size_t)sr * fd + 64 * cg + 8 * sp);
        p2b = *reinterpret_cast<const uint4*>(w2p + (size_t)(sr + 32) * fd + 64 * cg + 8 * sp);
    };
    auto deposit = [&](unsigned char* buf) __attribute__((always_inline)) {
        *reinterpret_cast<uint4*>(buf + sr * kF16Pitch + 16 * sp) = p1a;
        *reinterpret_cast<uint4*>(buf + (sr + 32) * kF16Pitch + 16 * sp) = p1b;
        *reinterpret_cast<uint4*>(buf + (64 + sr) * kF16Pitch + 16 * sp) = p2a;
        *reinterpret_cast<uint4*>(buf + (64 + sr + 32) * kF16Pitch + 16 * sp) = p2b;
    };
    const int ncg = fd / 64;
    request(0);
    deposit(lds);
    __syncthreads();
    for (int cg = 0; cg < ncg; ++cg) {
        const unsigned char* W1s = lds + (cg & 1) * kF16Buf;
        const unsigned char* W2s = W1s + 64 * kF16Pitch;
        if (cg + 1 < ncg) request(cg + 1);
        // the chunk's two 32-unit halves side by side: two independent chains of first products, then their activations, then two chains of second products -- the matrix
        // cores work on one half while the vector pipe finishes the other
        v16f hh[2];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) hh[c][r] = 0.0f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int c = 0; c < 2; ++c) hh[c] = mfma32x32x16(*reinterpret_cast<const uint4*>(W1s + (32 * c + l31) * kF16Pitch + 32 * ks + 16 * h), ks == 0 ? xb0 : (ks == 1 ? xb1 : (ks == 2 ? xb2 : xb3)), hh[c]);
        uint4 hb[2][2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            float4 bb[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) bb[q] = *reinterpret_cast<const float4*>(b1 + 64 * cg + 32 * c + 8 * q + 4 * h);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                float a[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int r = 8 * s + e;
                    const float4 bq = bb[r >> 2];
                    const float bv = (r & 3) == 0 ? bq.x : ((r & 3) == 1 ? bq.y : ((r & 3) == 2 ? bq.z : bq.w));
                    a[e] = swoosh_l16(hh[c][r] + bv);
                }
                hb[c][s] = make_uint4(pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3]), pack_bf16x2(a[4], a[5]), pack_bf16x2(a[6], a[7]));
            }
        }
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int jt = 0; jt < 2; ++jt)
                    acc2[jt] = mfma32x32x16(*reinterpret_cast<const uint4*>(W2s + (32 * jt + l31) * kF16Pitch + 64 * c + 32 * s + 16 * h), hb[c][s], acc2[jt]);
        if (cg + 1 < ncg) deposit(lds + ((cg + 1) & 1) * kF16Buf);          // the other buffer: last read in iteration cg - 1, behind that iteration's barrier
        __syncthreads();
    }
    // lane (row l31, h): register r of tile jt is output channel 32 jt + (r & 3) + 8 (r >> 2) + 4 h -> through the wave's LDS tile -> whole rows
    float* E = reinterpret_cast<float*>(lds) + wave * 32 * kF16EPitch;
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4*>(E + l31 * kF16EPitch + 32 * jt + 8 * q + 4 * h) = make_float4(acc2[jt][4 * q], acc2[jt][4 * q + 1], acc2[jt][4 * q + 2], acc2[jt][4 * q + 3]);
    wave_sync();
    const int c4 = (lane & 15) * 4;
    const float4 bo = *reinterpret_cast<const float4*>(b2 + c4);
    const float4 cv = (MODE == 2 || MODE == 3) ? *reinterpret_cast<const float4*>(cmid + c4) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    const float4 fsv = MODE == 3 ? *reinterpret_cast<const float4*>(cmid + 64 + c4) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    const float4 rsv = MODE == 3 ? *reinterpret_cast<const float4*>(cmid + 128 + c4) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    float4 xi[8], rv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int m = row0 + (lane >> 4) + 4 * u, mc = m < M ? m : M - 1;
        xi[u] = MODE != 0 ? *reinterpret_cast<const float4*>(xin + (size_t)mc * 64 + c4) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        rv[u] = MODE != 1 ? *reinterpret_cast<const float4*>(res + (size_t)mc * 64 + c4) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int rl = (lane >> 4) + 4 * u, m = row0 + rl;
        if (MODE == 3) {                                                    // (every lane takes part in the row sums: rows beyond M compute on the clamped row and store nothing)
            const float4 a = *reinterpret_cast<const float4*>(E + rl * kF16EPitch + c4);
            const float4 y = make_float4(xi[u].x + (a.x + bo.x), xi[u].y + (a.y + bo.y), xi[u].z + (a.z + bo.z), xi[u].w + (a.w + bo.w));
            const float4 d = make_float4(y.x - cv.x, y.y - cv.y, y.z - cv.z, y.w - cv.w);
            float ssq = fmaf(d.x, d.x, fmaf(d.y, d.y, fmaf(d.z, d.z, d.w * d.w)));
            ssq += __shfl_xor(ssq, 1, 64); ssq += __shfl_xor(ssq, 2, 64); ssq += __shfl_xor(ssq, 4, 64); ssq += __shfl_xor(ssq, 8, 64);
            const float nrm = sqrtf(ssq);
            if (m < M) *reinterpret_cast<float4*>(out + (size_t)m * 64 + c4) = make_float4((y.x / nrm) * fsv.x + rv[u].x * rsv.x, (y.y / nrm) * fsv.y + rv[u].y * rsv.y,
                                                                                          (y.z / nrm) * fsv.z + rv[u].z * rsv.z, (y.w / nrm) * fsv.w + rv[u].w * rsv.w);
            continue;
        }
        if (m >= M) continue;
        const float4 a = *reinterpret_cast<const float4*>(E + rl * kF16EPitch + c4);
        auto fin = [](float acc, float b, float x, float r, float c) -> float {
            const float f = acc + b;
            if (MODE == 0) return r + f;
            if (MODE == 1) return x + f;
            const float sum = x + f;
            return r + (sum - r) * c;
        };
        *reinterpret_cast<float4*>(out + (size_t)m * 64 + c4) = make_float4(fin(a.x, bo.x, xi[u].x, rv[u].x, cv.x), fin(a.y, bo.y, xi[u].y, rv[u].y, cv.y),
                                                                             fin(a.z, bo.z, xi[u].z, rv[u].z, cv.z), fin(a.w, bo.w, xi[u].w, rv[u].w, cv.w));
    }
}
template <int MODE>
inline void launch_zip_ff16(hipStream_t s, int M, const float* xin, const bf16_t* w1, const float* b1, const bf16_t* w2p, const float* b2, const float* res, const float* cmid,
                            float* out, int fd) {
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_zip_ff16<MODE>), dim3((unsigned)((M + 127) / 128)), dim3(256), 0, s, xin, w1, b1, w2p, b2, res, cmid, out, M, fd);
}

// ---- element access shared by the kernels that exist for both dtypes (attention core, convolution module): four consecutive elements as a float4 -----------------------
__device__ __forceinline__ float4 ldx4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 ldx4(const bf16_t* p) {
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    return make_float4(gemm16::bf16_lo(u.x), gemm16::bf16_hi(u.x), gemm16::bf16_lo(u.y), gemm16::bf16_hi(u.y));
}
// the same for a bf16_t-typed buffer that holds IEEE half (the convolution module's two private tensors on the bf16 path: neither feeds a matrix instruction directly)
__device__ __forceinline__ float4 ldh4(const bf16_t* p) {
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    float4 v;
    gemm16::unpack_f16x2(u.x, v.x, v.y);
    gemm16::unpack_f16x2(u.y, v.z, v.w);
    return v;
}
__device__ __forceinline__ void sth1(bf16_t* p, float v) { *p = (bf16_t)(gemm16::pack_f16x2(v, 0.0f) & 0xffffu); }
__device__ __forceinline__ float ldx1(const float* p) { return *p; }
__device__ __forceinline__ float ldx1(const bf16_t* p) { return __uint_as_float((unsigned)*p << 16); }
__device__ __forceinline__ void stx4(float* p, const float4& v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void stx4(bf16_t* p, const float4& v) { *reinterpret_cast<uint2*>(p) = gemm16::pack_bf16x4(v); }
__device__ __forceinline__ void stx1(float* p, float v) { *p = v; }
__device__ __forceinline__ void stx1(bf16_t* p, float v) { *p = (bf16_t)(pack_bf16x2(v, 0.0f) & 0xffffu); }

// ---- the small-K products of a Zipformer2 layer and of the (1, 3) convolutions: out(m, n) = sum_k A(m, k) W[n][k], K = 16 KS <= 192, N <= 32 NT ------------------------
// A 256-thread workgroup owns 128 rows, a wavefront 32 of them and ALL output columns.  A row's operand goes straight from global memory into the registers the matrix cores
// read (lane (row l31, h): k = 16 ks + 8 h .. + 7 of its row, one 16-byte load per step for bf16 rows, two for fp32 rows that are rounded here); the weights (N x K bf16) are
// staged once per workgroup (pitch 2 K + 16 bytes: conflict-free ds_read_b128 for K = 48, 64, 192).  The product is formed TRANSPOSED (weights = MFMA-A), so a lane ends with
// runs of four consecutive columns of ITS row; the tile passes through LDS 64 columns at a time and the store functor gets float4s (m, n .. n + 3) with 16 consecutive lanes
// on one row: whole-line reads (bias, residual) and writes.
//   AL: uint4 operator()(int m /* < M, clamped by the caller */, int ks, int h) const
//   ST: void operator()(int m, int n, float4 v) const          n % 4 == 0, n < N
struct F32Rows {               // fp32 [rows][ld]: the residual stream, rounded to bf16 on the way into the matrix cores
    const float* p; int ld;
    __device__ uint4 operator()(int m, int ks, int h) const {
        const float* q = p + (size_t)m * ld + 16 * ks + 8 * h;
        return pack8(*reinterpret_cast<const float4*>(q), *reinterpret_cast<const float4*>(q + 4));
    }
};
template <int ACT, bool HALFIN = false>     // 0: as stored; 2: SwooshR of the stored value (the convolution module's out-projection, :339); HALFIN: the stored values are IEEE half (ACT 2 only)
struct B16Rows {               // bf16 [rows][ld]
    const bf16_t* p; int ld;
    __device__ uint4 operator()(int m, int ks, int h) const {
        const uint4 u = *reinterpret_cast<const uint4*>(p + (size_t)m * ld + 16 * ks + 8 * h);
        static_assert(!HALFIN || ACT == 2, "half rows pass through the activation's unpack / repack");
        if (ACT == 0) return u;
        float v[8];
        if (HALFIN) { gemm16::unpack_f16x2(u.x, v[0], v[1]); gemm16::unpack_f16x2(u.y, v[2], v[3]); gemm16::unpack_f16x2(u.z, v[4], v[5]); gemm16::unpack_f16x2(u.w, v[6], v[7]); }
        else unpack8(u, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = swoosh_r16(v[e]);
        return make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
    }
};
struct RowConv16 {             // (1, 3) convolution along f over 64 channels at ch0 of the bf16 dense history: k = kf * 64 + ci, input column f * stride - 1 + kf (zero outside)
    const bf16_t* hist; int ld, ch0, T, Fin, Fout, stride;
    __device__ uint4 operator()(int m, int ks, int h) const {
        const int tf = T * Fout, b = m / tf, rem = m - b * tf, t = rem / Fout, f = rem - t * Fout;
        const int kf = ks >> 2, f2 = f * stride - 1 + kf;
        const bool ok = f2 >= 0 && f2 < Fin;
        return ld8_or_zero(ok, hist + ((size_t)(b * T + t) * Fin + (ok ? f2 : 0)) * ld + ch0 + 16 * (ks & 3) + 8 * h);
    }
};
struct Bf16BiasStore {         // out[m][off + n] = bf16(v + bias[n])
    bf16_t* out; const float* bias; int ld, off;
    __device__ void operator()(int m, int n, float4 v) const {
        const float4 b = *reinterpret_cast<const float4*>(bias + n);
        *reinterpret_cast<uint2*>(out + (size_t)m * ld + off + n) = gemm16::pack_bf16x4(make_float4(v.x + b.x, v.y + b.y, v.z + b.z, v.w + b.w));
    }
};
struct F32BiasStore {          // out[m][n] = v + bias[n]
    float* out; const float* bias; int ld;
    __device__ void operator()(int m, int n, float4 v) const {
        const float4 b = *reinterpret_cast<const float4*>(bias + n);
        *reinterpret_cast<float4*>(out + (size_t)m * ld + n) = make_float4(v.x + b.x, v.y + b.y, v.z + b.z, v.w + b.w);
    }
};
struct ResidualStore {         // y[m][n] += v + bias[n]      (fp32 residual stream, 64 columns)
    float* y; const float* bias;
    __device__ void operator()(int m, int n, float4 v) const {
        const float4 b = *reinterpret_cast<const float4*>(bias + n);
        float4* q = reinterpret_cast<float4*>(y + (size_t)m * 64 + n);
        const float4 o = *q;
        *q = make_float4(o.x + (v.x + b.x), o.y + (v.y + b.y), o.z + (v.z + b.z), o.w + (v.w + b.w));
    }
};
struct SubPixelStore16 {       // conv channel n = c * r + u of sub-band f -> U[(b, t, f * r + u)][ch0 + c] (+ bias)   (:767-769), fp32
    float* u; const float* bias; int ld, ch0, r;
    __device__ void operator()(int m, int n, float4 v) const {
        const float t[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) { const int c = (n + e) / r, s = (n + e) - c * r; u[((size_t)m * r + s) * ld + ch0 + c] = t[e] + bias[n + e]; }
    }
};
template <int KS, int NT>
constexpr int rows16_lds() { return (32 * NT * (32 * KS + 16)) > 4 * 32 * 68 * 4 ? (32 * NT * (32 * KS + 16)) : 4 * 32 * 68 * 4; }
struct GluStore16 {            // the two-kernel form of k_rows16_chain<.., GLU>: columns (a | gate) of a 2 C-wide product -> out[m][c] = half((a + bias) * sigmoid(gate + bias)), C = 64 columns
    bf16_t* out; const float* bias; int ld;
};
template <class ST> struct IsGluStore : std::false_type {};
template <> struct IsGluStore<GluStore16> : std::true_type {};
// STATS (round 6): the product feeds an InstanceNorm (the (1, 3) convolutions before the encoders and after the decoders, :853, :767-769).  The grid is then (tiles per window) x
// windows -- TFw rows per window, a tile never straddles one -- and the epilogue also emits, per tile and output CHANNEL (= column / cdiv: the sub-pixel convolution folds
// cdiv = 2 columns into a channel), the fp64 sum and sum of squares of (product + bias) over the tile's rows: partial[((window * nblk + tile) * 64 + channel) * 2 + {sum, sumsq}],
// k_zip_stats_final's layout.  The separate pass over the stored tensor (k_zip_stats_partial: 0.2 ms per call at 128 x 1 s) is gone.  ST must expose `bias` indexed by column.
template <int KS, int NT, class AL, class ST, bool HALF = false, bool STATS = false>        // HALF: operand rows and weights are IEEE half (the (1, 3) convolutions over the dense history)
__global__ __launch_bounds__(256) void k_rows16(AL a_of, const bf16_t* __restrict__ w, ST store, int M_all, int N, int TFw = 0, int nblk = 0, double* __restrict__ partial = nullptr,
                                                int cdiv = 1) {
    constexpr int kPitch = 32 * KS + 16;
    HIP_DYNAMIC_SHARED(unsigned char, lds)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
    const int win = STATS ? (int)blockIdx.x / nblk : 0, blk = STATS ? (int)blockIdx.x - win * nblk : (int)blockIdx.x;
    const int M = STATS ? (win + 1) * TFw : M_all;                       // rows at or beyond M are clamped on the way in and dropped on the way out
    const int row0 = (STATS ? win * TFw : 0) + blk * 128 + wave * 32, row = row0 + l31;
    uint4 xa[KS];
    {
        const int mc = row < M ? row : M - 1;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) xa[ks] = a_of(mc, ks, h);
    }
    // weights: N rows of 2 KS 16-byte pieces (rows beyond N re-read row N - 1: their products are never stored)
    for (int i = tid; i < 32 * NT * 2 * KS; i += 256) {
        const int n = i / (2 * KS), pc = i - n * (2 * KS);
        *reinterpret_cast<uint4*>(lds + n * kPitch + 16 * pc) = *reinterpret_cast<const uint4*>(w + (size_t)(n < N ? n : N - 1) * (16 * KS) + 8 * pc);
    }
    __syncthreads();
    v16f acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = gemm16::mfma32x32x16_t<HALF>(*reinterpret_cast<const uint4*>(lds + (32 * t + l31) * kPitch + 32 * ks + 16 * h), xa[ks], acc[t]);
    __syncthreads();                                                    // the weights are dead: the wave's epilogue tile takes their place
    float* E = reinterpret_cast<float*>(lds) + wave * 32 * 68;
    const int c4 = (lane & 15) * 4;
    float4 glu_a[8];                                                    // (GluStore16 only)
    (void)glu_a;
    constexpr int kGroups = (NT + 1) / 2;
    double ssum[STATS ? kGroups : 1][4], ssq[STATS ? kGroups : 1][4];   // (STATS) this lane's columns 64 g + c4 .. + 3 over its eight rows
    if (STATS) {
#pragma unroll
        for (int g = 0; g < kGroups; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) { ssum[g][e] = 0.0; ssq[g][e] = 0.0; }
    }
#pragma unroll
    for (int g = 0; g < kGroups; ++g) {
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) {
            if (2 * g + jt >= NT) continue;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<float4*>(E + l31 * 68 + 32 * jt + 8 * q + 4 * h) =
                    make_float4(acc[2 * g + jt][4 * q], acc[2 * g + jt][4 * q + 1], acc[2 * g + jt][4 * q + 2], acc[2 * g + jt][4 * q + 3]);
        }
        wave_sync();
        const int n = 64 * g + c4;
        if constexpr (IsGluStore<ST>::value) {
            static_assert(!IsGluStore<ST>::value || NT == 4, "GLU: 64 value columns + 64 gate columns");
            const float4 bb = *reinterpret_cast<const float4*>(store.bias + n);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int rl = (lane >> 4) + 4 * u, m = row0 + rl;
                const float4 v = *reinterpret_cast<const float4*>(E + rl * 68 + c4);
                const float4 o = make_float4(v.x + bb.x, v.y + bb.y, v.z + bb.z, v.w + bb.w);
                if (g == 0) { glu_a[u] = o; continue; }
                if (m < M) *reinterpret_cast<uint2*>(store.out + (size_t)m * store.ld + c4) =
                    gemm16::pack_f16x4(make_float4(glu_a[u].x * sigmoid16(o.x), glu_a[u].y * sigmoid16(o.y), glu_a[u].z * sigmoid16(o.z), glu_a[u].w * sigmoid16(o.w)));
            }
        } else if (n < N && (2 * g + 1 < NT || c4 < 32)) {
            float4 bq = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if constexpr (STATS) bq = *reinterpret_cast<const float4*>(store.bias + n);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int rl = (lane >> 4) + 4 * u, m = row0 + rl;
                if (m < M) {
                    const float4 v = *reinterpret_cast<const float4*>(E + rl * 68 + c4);
                    store(m, n, v);
                    if constexpr (STATS) {
                        const float t[4] = {v.x + bq.x, v.y + bq.y, v.z + bq.z, v.w + bq.w};        // (the stored values: the store adds the same bias)
#pragma unroll
                        for (int e = 0; e < 4; ++e) { ssum[g][e] += (double)t[e]; ssq[g][e] = fma((double)t[e], (double)t[e], ssq[g][e]); }
                    }
                }
            }
        }
        wave_sync();
    }
    if constexpr (STATS) {
        // rows: the four 16-lane groups of a wave, then the four waves (through LDS: every wave is past its tile), then the cdiv columns of a channel
#pragma unroll
        for (int g = 0; g < kGroups; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                ssum[g][e] += shfl_xor_f64(ssum[g][e], 16); ssum[g][e] += shfl_xor_f64(ssum[g][e], 32);
                ssq[g][e] += shfl_xor_f64(ssq[g][e], 16); ssq[g][e] += shfl_xor_f64(ssq[g][e], 32);
            }
        __syncthreads();
        double (*red)[32 * NT][2] = reinterpret_cast<double (*)[32 * NT][2]>(lds);          // [4 waves][columns][2]
        if (lane < 16) {
#pragma unroll
            for (int g = 0; g < kGroups; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int n = 64 * g + c4 + e;
                    if (n < 32 * NT) { red[wave][n][0] = n < N ? ssum[g][e] : 0.0; red[wave][n][1] = n < N ? ssq[g][e] : 0.0; }
                }
        }
        __syncthreads();
        const int nch = N / cdiv;
        if (tid < 2 * nch) {
            const int c = tid >> 1, q = tid & 1;
            double v = 0.0;
            for (int u = 0; u < cdiv; ++u) { const int n = c * cdiv + u; v += (red[0][n][q] + red[1][n][q]) + (red[2][n][q] + red[3][n][q]); }
            partial[(((size_t)win * nblk + blk) * 64 + c) * 2 + q] = v;
        }
    }
}
template <int KS, int NT, bool HALF = false, class AL, class ST>
inline void launch_rows16(hipStream_t s, const AL& a, const bf16_t* w, const ST& st, int M, int N) {
    if (M <= 0) return;
    auto kern = k_rows16<KS, NT, AL, ST, HALF>;
    constexpr int bytes = rows16_lds<KS, NT>();
    // (more than 48 KB of dynamic LDS -- K = 192 x N = 128 only -- needs the attribute: raised once per device when the engine is created, raise_rows16_lds below)
    hipLaunchKernelGGL(kern, dim3((unsigned)((M + 127) / 128)), dim3(256), bytes, s, a, w, st, M, N, 0, 0, (double*)nullptr, 1);
}
template <int KS, int NT, bool HALF = false, class AL, class ST>
inline void launch_rows16_stats(hipStream_t s, const AL& a, const bf16_t* w, const ST& st, int N, int TFw, int windows, double* partial, int cdiv) {
    static_assert(4 * 32 * NT * 2 * 8 <= rows16_lds<KS, NT>(), "the per-wave column sums fit the epilogue's LDS");
    const int nblk = (TFw + 127) / 128;
    auto kern = k_rows16<KS, NT, AL, ST, HALF, true>;
    constexpr int bytes = rows16_lds<KS, NT>();
    hipLaunchKernelGGL(kern, dim3((unsigned)(nblk * windows)), dim3(256), bytes, s, a, w, st, windows * TFw, N, TFw, nblk, partial, cdiv);
}
template <int KS, int NT, bool HALF, class AL, class ST, bool STATS = false>
inline hipError_t raise_rows16_lds() {
    constexpr int bytes = rows16_lds<KS, NT>();
    if (bytes <= 48 * 1024) return hipSuccess;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(k_rows16<KS, NT, AL, ST, HALF, STATS>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}


// ---- an out-projection and the NEXT module's in-projection in one launch ----------------------------------------------------------------------------------------------
//   y[m] += A1(m) W1^T + b1        (the fp32 residual stream, 64 columns: exactly k_rows16<KS1, 2, AL, ResidualStore>)
//   out2[m] = bf16(y[m] W2^T + b2)  (exactly k_rows16<4, NT2, F32Rows, Bf16BiasStore> on the updated row)
// The updated row never leaves the wavefront: it goes from the first product's accumulators through the wave's LDS tile (where the stream's old value and the bias are added
// and the row is stored, whole lines) straight into the second product's operand registers -- one read of the 256-byte stream row per pair instead of two, one launch instead
// of two; the values are those of the two-kernel form bit for bit (the second product rounds the fp32 row it would have re-read).  LDS: the epilogue tile (the first weights
// alias it until the first product is done) + the second weights: 44 - 53 KB, three workgroups per CU.
template <int KS1, int NT2>
constexpr int chain16_lds() { return 4 * 32 * 68 * 4 + 32 * NT2 * 144; }
// GLU (round 6): the second product is the convolution module's in-projection (:321), whose 2 C columns are (a | gate): the store writes bf16(a * sigmoid(gate)), C columns at ld2,
// instead of both halves -- the gate was applied by the depthwise kernel's staging, which read 256 bytes per row (and the halo rows again) where it now reads 128.  The C columns
// are stored as IEEE HALF (saturating): the tensor is private to the convolution module (vector-pipe consumer), so the three extra mantissa bits are free.
template <int KS1, int NT2, class AL, bool GLU = false>
__global__ __launch_bounds__(256) void k_rows16_chain(AL a_of, const bf16_t* __restrict__ w1, const float* __restrict__ b1, float* __restrict__ y, const bf16_t* __restrict__ w2,
                                                      const float* __restrict__ b2, bf16_t* __restrict__ out2, int ld2, int M, int N2) {
    constexpr int kP1 = 32 * KS1 + 16, kP2 = 144, kE = 4 * 32 * 68 * 4;
    static_assert(64 * kP1 <= kE, "the first weights alias the epilogue tile");
    HIP_DYNAMIC_SHARED(unsigned char, lds)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
    const int row0 = (int)blockIdx.x * 128 + wave * 32, row = row0 + l31;
    uint4 xa[KS1];
    {
        const int mc = row < M ? row : M - 1;
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) xa[ks] = a_of(mc, ks, h);
    }
    unsigned char* W2s = lds + kE;
    for (int i = tid; i < 64 * 2 * KS1; i += 256) {
        const int n = i / (2 * KS1), pc = i - n * (2 * KS1);
        *reinterpret_cast<uint4*>(lds + n * kP1 + 16 * pc) = *reinterpret_cast<const uint4*>(w1 + (size_t)n * (16 * KS1) + 8 * pc);
    }
    for (int i = tid; i < 32 * NT2 * 8; i += 256) {
        const int n = i >> 3, pc = i & 7;
        *reinterpret_cast<uint4*>(W2s + n * kP2 + 16 * pc) = *reinterpret_cast<const uint4*>(w2 + (size_t)(n < N2 ? n : N2 - 1) * 64 + 8 * pc);
    }
    __syncthreads();
    v16f acc1[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[t][r] = 0.0f;
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks)
#pragma unroll
        for (int t = 0; t < 2; ++t) acc1[t] = mfma32x32x16(*reinterpret_cast<const uint4*>(lds + (32 * t + l31) * kP1 + 32 * ks + 16 * h), xa[ks], acc1[t]);
    __syncthreads();                                                    // the first weights are dead: the wave's tile takes their place
    float* E = reinterpret_cast<float*>(lds) + wave * 32 * 68;
    const int c4 = (lane & 15) * 4;
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4*>(E + l31 * 68 + 32 * jt + 8 * q + 4 * h) = make_float4(acc1[jt][4 * q], acc1[jt][4 * q + 1], acc1[jt][4 * q + 2], acc1[jt][4 * q + 3]);
    wave_sync();
    {   // the stream's update, whole rows; the new row goes back into the tile for the second product
        const float4 bb = *reinterpret_cast<const float4*>(b1 + c4);
        float4 old[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int m = row0 + (lane >> 4) + 4 * u; old[u] = *reinterpret_cast<const float4*>(y + (size_t)(m < M ? m : M - 1) * 64 + c4); }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int rl = (lane >> 4) + 4 * u, m = row0 + rl;
            const float4 v = *reinterpret_cast<const float4*>(E + rl * 68 + c4);
            const float4 o = make_float4(old[u].x + (v.x + bb.x), old[u].y + (v.y + bb.y), old[u].z + (v.z + bb.z), old[u].w + (v.w + bb.w));
            if (m < M) *reinterpret_cast<float4*>(y + (size_t)m * 64 + c4) = o;
            *reinterpret_cast<float4*>(E + rl * 68 + c4) = o;
        }
    }
    wave_sync();
    uint4 xb0, xb1, xb2, xb3;
    {
        const float* er = E + l31 * 68 + 8 * h;
        xb0 = pack8(*reinterpret_cast<const float4*>(er), *reinterpret_cast<const float4*>(er + 4));
        xb1 = pack8(*reinterpret_cast<const float4*>(er + 16), *reinterpret_cast<const float4*>(er + 20));
        xb2 = pack8(*reinterpret_cast<const float4*>(er + 32), *reinterpret_cast<const float4*>(er + 36));
        xb3 = pack8(*reinterpret_cast<const float4*>(er + 48), *reinterpret_cast<const float4*>(er + 52));
    }
    wave_sync();
    v16f acc2[NT2];
#pragma unroll
    for (int t = 0; t < NT2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[t][r] = 0.0f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int t = 0; t < NT2; ++t)
            acc2[t] = mfma32x32x16(*reinterpret_cast<const uint4*>(W2s + (32 * t + l31) * kP2 + 32 * ks + 16 * h), ks == 0 ? xb0 : (ks == 1 ? xb1 : (ks == 2 ? xb2 : xb3)), acc2[t]);
    static_assert(!GLU || NT2 == 4, "GLU: 64 value columns + 64 gate columns");
    float4 glu_a[8];                                                    // GLU: the value half (+ bias) of this lane's eight (row, column quad) slots, kept until the gate half arrives
#pragma unroll
    for (int g = 0; g < NT2 / 2; ++g) {
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<float4*>(E + l31 * 68 + 32 * jt + 8 * q + 4 * h) =
                    make_float4(acc2[2 * g + jt][4 * q], acc2[2 * g + jt][4 * q + 1], acc2[2 * g + jt][4 * q + 2], acc2[2 * g + jt][4 * q + 3]);
        wave_sync();
        const int n = 64 * g + c4;
        if (n < N2) {
            const float4 bb = *reinterpret_cast<const float4*>(b2 + n);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int rl = (lane >> 4) + 4 * u, m = row0 + rl;
                const float4 v = *reinterpret_cast<const float4*>(E + rl * 68 + c4);
                const float4 o = make_float4(v.x + bb.x, v.y + bb.y, v.z + bb.z, v.w + bb.w);
                if (GLU) {
                    if (g == 0) { glu_a[u] = o; continue; }
                    if (m < M) *reinterpret_cast<uint2*>(out2 + (size_t)m * ld2 + c4) =
                        gemm16::pack_f16x4(make_float4(glu_a[u].x * sigmoid16(o.x), glu_a[u].y * sigmoid16(o.y), glu_a[u].z * sigmoid16(o.z), glu_a[u].w * sigmoid16(o.w)));
                    continue;
                }
                if (m >= M) continue;
                *reinterpret_cast<uint2*>(out2 + (size_t)m * ld2 + n) = gemm16::pack_bf16x4(o);
            }
        }
        wave_sync();
    }
}
template <int KS1, int NT2, bool GLU = false, class AL>
inline void launch_rows16_chain(hipStream_t s, const AL& a, const bf16_t* w1, const float* b1, float* y, const bf16_t* w2, const float* b2, bf16_t* out2, int ld2, int M, int N2) {
    static_assert(NT2 % 2 == 0, "whole 64-column groups");
    if (M <= 0) return;
    auto kern = k_rows16_chain<KS1, NT2, AL, GLU>;
    constexpr int bytes = chain16_lds<KS1, NT2>();
    hipLaunchKernelGGL(kern, dim3((unsigned)((M + 127) / 128)), dim3(256), bytes, s, a, w1, b1, y, w2, b2, out2, ld2, M, N2);
}
template <int KS1, int NT2, class AL, bool GLU = false>
inline hipError_t raise_rows16_chain_lds() {
    constexpr int bytes = chain16_lds<KS1, NT2>();
    if (bytes <= 48 * 1024) return hipSuccess;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(k_rows16_chain<KS1, NT2, AL, GLU>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

// ---- a feed-forward module WITH the row-local products around it in one launch (round 6) -------------------------------------------------------------------------------------
// A Zipformer2 layer touches its fp32 stream row ~20 times; between two sequence-mixing cores (attention, depthwise convolution) everything is row-local, and each of these
// launches read and wrote the 256-byte row again: a projection before a feed-forward module, the module, a projection after it.  k_zip_ffx is k_zip_ff16 with
//   PRE = 1   the convolution module's out-projection in front: xin_row := xin_row + (Wpre swooshR(o16 row) + bpre)   (k_rows16<4, 2, B16Rows<2>, ResidualStore>, :339); the
//             updated row feeds the module from the wave's LDS tile and is NOT written back (its only reader is this module);
//   PRE = 2   a side product of the module's input: pout = bf16(Wpre xin_row + bpre), N = pre.n <= 160 columns   (k_rows16<4, 5, F32Rows, Bf16BiasStore>: the layer's attention
//             in-projection, :148-153);
//   POSTNT    a product of the module's OUTPUT row: post.out = bf16(Wpost out_row + bpost), N = post.n <= 32 POSTNT   (k_rows16<4, NT, F32Rows, Bf16BiasStore>: the next
//             module's in-projection) -- the row goes from the epilogue's whole-row pass back through the wave's tile into the operand registers.
// The products are the same matrix instructions on the same operands in the same order as the separate kernels', every rounding happens where it did: the fused layer's
// output equals the unfused one's bit for bit (ADE_ZIP_FUSE=0 keeps that form; tests/test_zipenhancer.py).  The projections' weights take the module's LDS before / after its
// streamed chunks (36.9 KB per workgroup as before); their bf16 outputs leave straight from the accumulators -- a lane owns runs of four consecutive columns of ITS row:
// 8-byte stores -- because the wave's LDS tile is under the weights at that point.
struct FfxPre { const bf16_t* o16; const bf16_t* w; const float* b; bf16_t* pout; int ldp, n; };
struct FfxPost { const bf16_t* w; const float* b; bf16_t* out; int ld, n; };
template <int NT>
__device__ __forceinline__ void ffx_product(const unsigned char* W, const uint4& x0, const uint4& x1, const uint4& x2, const uint4& x3, int l31, int h, v16f* acc) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int t = 0; t < NT; ++t)
            acc[t] = mfma32x32x16(*reinterpret_cast<const uint4*>(W + (32 * t + l31) * kF16Pitch + 32 * ks + 16 * h), ks == 0 ? x0 : (ks == 1 ? x1 : (ks == 2 ? x2 : x3)), acc[t]);
}
// N rows of 64 bf16 (8 pieces of 16 bytes) into LDS at the 144-byte pitch; rows beyond n re-read row n - 1 (their products are never stored)
template <int NT>
__device__ __forceinline__ void ffx_stage(unsigned char* lds, const bf16_t* __restrict__ w, int n, int tid) {
    for (int i = tid; i < 32 * NT * 8; i += 256) {
        const int r = i >> 3, pc = i & 7;
        *reinterpret_cast<uint4*>(lds + r * kF16Pitch + 16 * pc) = *reinterpret_cast<const uint4*>(w + (size_t)(r < n ? r : n - 1) * 64 + 8 * pc);
    }
}
// lane (row, h): register r of tile t is column 32 t + (r & 3) + 8 (r >> 2) + 4 h -> bf16(v + bias), four columns per store
template <int NT>
__device__ __forceinline__ void ffx_store16(const v16f* acc, const float* __restrict__ bias, bf16_t* __restrict__ out, int ld, int n, int row, int M, int h) {
    if (row >= M) return;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int col = 32 * t + 8 * q + 4 * h;
            if (col >= n) continue;
            const float4 b = *reinterpret_cast<const float4*>(bias + col);
            *reinterpret_cast<uint2*>(out + (size_t)row * ld + col) =
                gemm16::pack_bf16x4(make_float4(acc[t][4 * q] + b.x, acc[t][4 * q + 1] + b.y, acc[t][4 * q + 2] + b.z, acc[t][4 * q + 3] + b.w));
        }
}
// The same store as WHOLE ROWS: the wave's tile (+ bias, rounded) goes through the wave's own LDS region at a pitch of 2 n + 16 bytes (16-byte aligned rows; 8-byte writes of
// rows r and r + 16 share banks: two-way at worst) and leaves as 16-byte pieces, consecutive lanes on consecutive pieces of a row: full lines at the memory side (ffx_store16 hands it 16 bytes per row and
// instruction, 20 instructions per 288-byte row: the in-projections' stores were 40 % of k_zip_ffx<0, 2, 5>'s bytes and all of its partial-line writes).  The caller has met
// at a workgroup barrier after the product (the weights under the regions are dead) and meets again before the LDS is re-used.  n % 8 == 0, 32 (2 n + 16) <= kFfxRegion.
constexpr int kFfxRegion = 32 * (2 * 144 + 16);
constexpr int kFfxLds = kF16Lds > 4 * kFfxRegion ? kF16Lds : 4 * kFfxRegion;
template <int NT>
__device__ __forceinline__ void ffx_store_rows(const v16f* acc, const float* __restrict__ bias, bf16_t* __restrict__ out, int ld, int n, int row0, int M, unsigned char* region, int lane) {
    const int l31 = lane & 31, h = lane >> 5, pitch = 2 * n + 16;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int col = 32 * t + 8 * q + 4 * h;
            if (col >= n) continue;
            const float4 b = *reinterpret_cast<const float4*>(bias + col);
            *reinterpret_cast<uint2*>(region + l31 * pitch + 2 * col) =
                gemm16::pack_bf16x4(make_float4(acc[t][4 * q] + b.x, acc[t][4 * q + 1] + b.y, acc[t][4 * q + 2] + b.z, acc[t][4 * q + 3] + b.w));
        }
    wave_sync();
    const int pieces = n >> 3, items = 32 * pieces;
    for (int i = lane; i < items; i += 64) {
        const int r = i / pieces, pc = i - r * pieces;
        if (row0 + r < M) *reinterpret_cast<uint4*>(out + (size_t)(row0 + r) * ld + 8 * pc) = *reinterpret_cast<const uint4*>(region + r * pitch + 16 * pc);
    }
}
// Three wavefronts per SIMD: the PRE = 1 forms keep the updated rows in 32 registers across the module and spill 7 - 15 dwords at that bound; at two per SIMD (198 registers, no
// spill) the step is 1.3 ms slower (profiles/r06_n_zip_ffx_occupancy.txt: 41.9 against 43.2 ms).
// (measurement knobs of profiles/r06_t_zip_ffx_occupancy.txt: tools/build_variant.sh <name> ade_zipenhancer -DADE_FFX_WAVES_PRE2=4 ...)
#ifndef ADE_FFX_WAVES_PRE1
#define ADE_FFX_WAVES_PRE1 3
#endif
#ifndef ADE_FFX_WAVES_PRE2
#define ADE_FFX_WAVES_PRE2 3
#endif
template <int MODE, int PRE, int POSTNT>
__global__ __launch_bounds__(256, (PRE == 1 ? ADE_FFX_WAVES_PRE1 : ADE_FFX_WAVES_PRE2)) void k_zip_ffx(const float* xin, const bf16_t* __restrict__ w1, const float* __restrict__ b1, const bf16_t* __restrict__ w2p,
                                                 const float* __restrict__ b2, const float* res, const float* __restrict__ cmid, float* out, int M, int fd, FfxPre pre, FfxPost post) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[kFfxLds];
    static_assert(5 * 32 * kF16Pitch <= kFfxLds, "a 160-column projection's weights fit the module's LDS");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
    const int row0 = (int)blockIdx.x * 128 + wave * 32, row = row0 + l31, mrow = row < M ? row : M - 1;
    const int c4 = (lane & 15) * 4;
    float* const E = reinterpret_cast<float*>(lds) + wave * 32 * kF16EPitch;
    uint4 xb0, xb1, xb2, xb3;
    float4 xiv[8];                     // PRE == 1: the updated rows in the epilogue's ownership (rows (lane >> 4) + 4 u, columns c4 ..)
    if (PRE == 1) {
        const B16Rows<2, true> ld{pre.o16, 64};                             // (the depthwise kernel's output: IEEE half on this path)
        const uint4 a0 = ld(mrow, 0, h), a1 = ld(mrow, 1, h), a2 = ld(mrow, 2, h), a3 = ld(mrow, 3, h);
        float4 old[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int m = row0 + (lane >> 4) + 4 * u; old[u] = *reinterpret_cast<const float4*>(xin + (size_t)(m < M ? m : M - 1) * 64 + c4); }
        ffx_stage<2>(lds, pre.w, 64, tid);
        __syncthreads();
        v16f acc1[2];
        ffx_product<2>(lds, a0, a1, a2, a3, l31, h, acc1);
        __syncthreads();                                                    // the weights are dead: the waves' tiles take their place
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<float4*>(E + l31 * kF16EPitch + 32 * jt + 8 * q + 4 * h) = make_float4(acc1[jt][4 * q], acc1[jt][4 * q + 1], acc1[jt][4 * q + 2], acc1[jt][4 * q + 3]);
        wave_sync();
        const float4 bb = *reinterpret_cast<const float4*>(pre.b + c4);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int rl = (lane >> 4) + 4 * u;
            const float4 v = *reinterpret_cast<const float4*>(E + rl * kF16EPitch + c4);
            const float4 o = make_float4(old[u].x + (v.x + bb.x), old[u].y + (v.y + bb.y), old[u].z + (v.z + bb.z), old[u].w + (v.w + bb.w));
            xiv[u] = o;
            *reinterpret_cast<float4*>(E + rl * kF16EPitch + c4) = o;
        }
        wave_sync();
        const float* er = E + l31 * kF16EPitch + 8 * h;
        xb0 = pack8(*reinterpret_cast<const float4*>(er), *reinterpret_cast<const float4*>(er + 4));
        xb1 = pack8(*reinterpret_cast<const float4*>(er + 16), *reinterpret_cast<const float4*>(er + 20));
        xb2 = pack8(*reinterpret_cast<const float4*>(er + 32), *reinterpret_cast<const float4*>(er + 36));
        xb3 = pack8(*reinterpret_cast<const float4*>(er + 48), *reinterpret_cast<const float4*>(er + 52));
        __syncthreads();                                                    // every wave holds its operands: the module's weights may take the tiles
    } else {
        const float* src = xin + (size_t)mrow * 64 + 8 * h;
        float4 t[8];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { t[2 * ks] = *reinterpret_cast<const float4*>(src + 16 * ks); t[2 * ks + 1] = *reinterpret_cast<const float4*>(src + 16 * ks + 4); }
        xb0 = pack8(t[0], t[1]); xb1 = pack8(t[2], t[3]); xb2 = pack8(t[4], t[5]); xb3 = pack8(t[6], t[7]);
    }
    if (PRE == 2) {
        ffx_stage<5>(lds, pre.w, pre.n, tid);
        __syncthreads();
        v16f accs[5];
        ffx_product<5>(lds, xb0, xb1, xb2, xb3, l31, h, accs);
        if (32 * (2 * pre.n + 16) <= kFfxRegion) {
            __syncthreads();                                                // the projection's weights are dead: the waves' store regions take their place
            ffx_store_rows<5>(accs, pre.b, pre.pout, pre.ldp, pre.n, row0, M, lds + wave * kFfxRegion, lane);
        } else ffx_store16<5>(accs, pre.b, pre.pout, pre.ldp, pre.n, row, M, h);
        __syncthreads();                                                    // ... and then the module's first chunk
    }
    v16f acc2[2];
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[jt][r] = 0.0f;
    const int sr = tid >> 3, sp = tid & 7;
    uint4 p1a, p1b, p2a, p2b;
    auto request = [&](int cg) __attribute__((always_inline)) {
        p1a = *reinterpret_cast<const uint4*>(w1 + (size_t)(64 * cg + sr) * 64 + 8 * sp);
        p1b = *reinterpret_cast<const uint4*>(w1 + (size_t)(64 * cg + sr + 32) * 64 + 8 * sp);
        p2a = *reinterpret_cast<const uint4*>(w2p + (size_t)sr * fd + 64 * cg + 8 * sp);
        p2b = *reinterpret_cast<const uint4*>(w2p + (size_t)(sr + 32) * fd + 64 * cg + 8 * sp);
    };
    auto deposit = [&](unsigned char* buf) __attribute__((always_inline)) {
        *reinterpret_cast<uint4*>(buf + sr * kF16Pitch + 16 * sp) = p1a;
        *reinterpret_cast<uint4*>(buf + (sr + 32) * kF16Pitch + 16 * sp) = p1b;
        *reinterpret_cast<uint4*>(buf + (64 + sr) * kF16Pitch + 16 * sp) = p2a;
        *reinterpret_cast<uint4*>(buf + (64 + sr + 32) * kF16Pitch + 16 * sp) = p2b;
    };
    const int ncg = fd / 64;
    request(0);
    deposit(lds);
    __syncthreads();
    for (int cg = 0; cg < ncg; ++cg) {
        const unsigned char* W1s = lds + (cg & 1) * kF16Buf;
        const unsigned char* W2s = W1s + 64 * kF16Pitch;
        if (cg + 1 < ncg) request(cg + 1);
        v16f hh[2];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) hh[c][r] = 0.0f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int c = 0; c < 2; ++c) hh[c] = mfma32x32x16(*reinterpret_cast<const uint4*>(W1s + (32 * c + l31) * kF16Pitch + 32 * ks + 16 * h), ks == 0 ? xb0 : (ks == 1 ? xb1 : (ks == 2 ? xb2 : xb3)), hh[c]);
        uint4 hb[2][2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            float4 bb[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) bb[q] = *reinterpret_cast<const float4*>(b1 + 64 * cg + 32 * c + 8 * q + 4 * h);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                float a[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int r = 8 * s + e;
                    const float4 bq = bb[r >> 2];
                    const float bv = (r & 3) == 0 ? bq.x : ((r & 3) == 1 ? bq.y : ((r & 3) == 2 ? bq.z : bq.w));
                    a[e] = swoosh_l16(hh[c][r] + bv);
                }
                hb[c][s] = make_uint4(pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3]), pack_bf16x2(a[4], a[5]), pack_bf16x2(a[6], a[7]));
            }
        }
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int jt = 0; jt < 2; ++jt)
                    acc2[jt] = mfma32x32x16(*reinterpret_cast<const uint4*>(W2s + (32 * jt + l31) * kF16Pitch + 64 * c + 32 * s + 16 * h), hb[c][s], acc2[jt]);
        if (cg + 1 < ncg) deposit(lds + ((cg + 1) & 1) * kF16Buf);
        __syncthreads();
    }
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4*>(E + l31 * kF16EPitch + 32 * jt + 8 * q + 4 * h) = make_float4(acc2[jt][4 * q], acc2[jt][4 * q + 1], acc2[jt][4 * q + 2], acc2[jt][4 * q + 3]);
    wave_sync();
    const float4 bo = *reinterpret_cast<const float4*>(b2 + c4);
    const float4 cv = (MODE == 2 || MODE == 3) ? *reinterpret_cast<const float4*>(cmid + c4) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    const float4 fsv = MODE == 3 ? *reinterpret_cast<const float4*>(cmid + 64 + c4) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    const float4 rsv = MODE == 3 ? *reinterpret_cast<const float4*>(cmid + 128 + c4) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    float4 xi[8], rv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int m = row0 + (lane >> 4) + 4 * u, mc = m < M ? m : M - 1;
        xi[u] = PRE == 1 ? xiv[u] : (MODE != 0 ? *reinterpret_cast<const float4*>(xin + (size_t)mc * 64 + c4) : make_float4(0.0f, 0.0f, 0.0f, 0.0f));
        rv[u] = MODE != 1 ? *reinterpret_cast<const float4*>(res + (size_t)mc * 64 + c4) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int rl = (lane >> 4) + 4 * u, m = row0 + rl;
        const float4 a = *reinterpret_cast<const float4*>(E + rl * kF16EPitch + c4);
        float4 ov;
        if (MODE == 3) {                                                    // (every lane takes part in the row sums: rows beyond M compute on the clamped row and store nothing)
            const float4 y = make_float4(xi[u].x + (a.x + bo.x), xi[u].y + (a.y + bo.y), xi[u].z + (a.z + bo.z), xi[u].w + (a.w + bo.w));
            const float4 d = make_float4(y.x - cv.x, y.y - cv.y, y.z - cv.z, y.w - cv.w);
            float ssq = fmaf(d.x, d.x, fmaf(d.y, d.y, fmaf(d.z, d.z, d.w * d.w)));
            ssq += __shfl_xor(ssq, 1, 64); ssq += __shfl_xor(ssq, 2, 64); ssq += __shfl_xor(ssq, 4, 64); ssq += __shfl_xor(ssq, 8, 64);
            const float nrm = sqrtf(ssq);
            ov = make_float4((y.x / nrm) * fsv.x + rv[u].x * rsv.x, (y.y / nrm) * fsv.y + rv[u].y * rsv.y, (y.z / nrm) * fsv.z + rv[u].z * rsv.z, (y.w / nrm) * fsv.w + rv[u].w * rsv.w);
        } else {
            auto fin = [](float acc, float b, float x, float r, float c) -> float {
                const float f = acc + b;
                if (MODE == 0) return r + f;
                if (MODE == 1) return x + f;
                const float sum = x + f;
                return r + (sum - r) * c;
            };
            ov = make_float4(fin(a.x, bo.x, xi[u].x, rv[u].x, cv.x), fin(a.y, bo.y, xi[u].y, rv[u].y, cv.y), fin(a.z, bo.z, xi[u].z, rv[u].z, cv.z), fin(a.w, bo.w, xi[u].w, rv[u].w, cv.w));
        }
        if (m < M) *reinterpret_cast<float4*>(out + (size_t)m * 64 + c4) = ov;
        if (POSTNT > 0) *reinterpret_cast<float4*>(E + rl * kF16EPitch + c4) = ov;          // (this lane's own slot of the tile: read above, nobody else's)
    }
    if (POSTNT > 0) {
        wave_sync();
        const float* er = E + l31 * kF16EPitch + 8 * h;
        const uint4 y0 = pack8(*reinterpret_cast<const float4*>(er), *reinterpret_cast<const float4*>(er + 4));
        const uint4 y1 = pack8(*reinterpret_cast<const float4*>(er + 16), *reinterpret_cast<const float4*>(er + 20));
        const uint4 y2 = pack8(*reinterpret_cast<const float4*>(er + 32), *reinterpret_cast<const float4*>(er + 36));
        const uint4 y3 = pack8(*reinterpret_cast<const float4*>(er + 48), *reinterpret_cast<const float4*>(er + 52));
        __syncthreads();                                                    // every wave holds its operands: the projection's weights take the tiles
        constexpr int NTP = POSTNT > 0 ? POSTNT : 1;
        ffx_stage<NTP>(lds, post.w, post.n, tid);
        __syncthreads();
        v16f acc3[NTP];
        ffx_product<NTP>(lds, y0, y1, y2, y3, l31, h, acc3);
        if (32 * (2 * post.n + 16) <= kFfxRegion) {
            __syncthreads();
            ffx_store_rows<NTP>(acc3, post.b, post.out, post.ld, post.n, row0, M, lds + wave * kFfxRegion, lane);
        } else ffx_store16<NTP>(acc3, post.b, post.out, post.ld, post.n, row, M, h);
    }
}
template <int MODE, int PRE, int POSTNT>
inline void launch_zip_ffx(hipStream_t s, int M, const float* xin, const bf16_t* w1, const float* b1, const bf16_t* w2p, const float* b2, const float* res, const float* cmid,
                           float* out, int fd, const FfxPre& pre, const FfxPost& post) {
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_zip_ffx<MODE, PRE, POSTNT>), dim3((unsigned)((M + 127) / 128)), dim3(256), 0, s, xin, w1, b1, w2p, b2, res, cmid, out, M, fd, pre, post);
}

}  // namespace zip16
}  // namespace ade
