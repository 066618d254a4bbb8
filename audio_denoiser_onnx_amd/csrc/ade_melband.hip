// ade_melband.hip — Mel-Band-Roformer (44.1 kHz stereo source separation / denoising) on the MI355X: SURVEY.md section 8 row a17.
//
// Reference: MelBandRoformer.forward / _core over the FUSED buffers its constructor registers
// (Mel_Band_Roformer/Stereo/Export_MelBandRoformer.py:540-680; buffers :330-538):
//   int16 (B, 2, L) -> * 2^-15 -> STFT(2048, hop 441, periodic hann, reflect) -> channel-interleaved bins (f*2+ch) ->
//   gather the 60 overlapping mel bands -> per band L2-normalise + Linear(d_i, 384) -> depth x [time transformer over the
//   frames of each band, frequency transformer over the bands of each frame] -> per band tanh MLP 384-1536-1536-2 d_i + GLU
//   -> scatter-add back to bins (the averaging is folded into the GLU value rows) -> complex mask x spectrum -> ISTFT
//   -> * 32767, clamp, truncate -> int16.
//   transformer(x) = { x += W_o [softmax(rot(q) rot(k)^T) v * sigmoid(gates)];  x += W_2 gelu(W_1 n(x) + b_1) + b_2;  n(x) * g }
//   with n(x) = x / max(|x|_2, 1e-12) and (q | k | v | gates) = W_in n(x) + b_in   (q pre-scaled by dim_head^-1/2).
// Everything but the attention core and three row-wise kernels is a matrix product, so the model runs on the generic
// matrix-core GEMM of csrc/ade_gemm.h (exact fp32, v_mfma_f32_16x16x4_f32) with functor operands and stores:
//   * the token matrix X is (row, 384) row-major with row = (band * B + b) * T + t.  BOTH attention axes read it in place:
//     a time sequence is T consecutive rows, a frequency sequence is 60 rows a stride of B*T apart -- the reference's two
//     permutes per layer pair (:611-614) never materialise.
//   * every n(x) feeding a Linear is folded into that GEMM's store (v * inv_norm[row] + bias): X is read once by the GEMM
//     and once by a one-wave-per-row norm kernel, never rewritten.
//   * residual adds, biases, GELU, tanh are GEMM stores; the band gather is the A-operand loader of the band-split GEMM and
//     the complex mask the A-operand of the synthesis GEMM.
//   * the per-band problems (band split, mask estimator; K or N differ per band) run as ONE batched launch each
//     (blockIdx.z = band, gemm::launch_batched).
// DFT tables: by default the REFERENCE's (cos / sin of fp32 angles up to 2 pi * 1024, i.e. up to ~1e-4 off).  Unlike
// the other models this one is sensitive to them: each band is L2-normalised before its Linear (:576), so in a band that is
// nearly silent the leakage those table errors cause IS the normalised input.  Matching the reference there means matching its
// tables; manifest key ade_dft_tables = "exact" selects exactly reduced angles instead (closer to torch.stft, which the
// checkpoints were trained with).
#include "ade_gemm.h"
#include "ade_gemm16.h"
#include "ade_internal.h"
#include "../../include/ade.h"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace ade {

namespace {

using namespace dev;

constexpr int kNfftM = 2048, kHopM = 441, kBinsM = kNfftM / 2 + 1;   // Export_MelBandRoformer.py:42-45
constexpr int kChan = 2, kFc = kBinsM * kChan;                        // channel-interleaved bins
constexpr int kDh = 64;                                               // dim_head
constexpr int kMaxBands = 1024, kMaxFrames = 8192;                    // sanity bounds only: attention streams K / V, any length works

// ---- operand / store functors ----------------------------------------------------------------------------------------
// PCM of one call is [channel][window][L] (batch-fold, :644-647; n_win = 1 without it), calls follow each other; the model's
// rows are window-major / channel-minor: row r = (call * n_win + window) * 2 + channel.
__device__ __forceinline__ size_t pcm_row_offset(int r, int n_win, int L) {
    const int b = r >> 1, ch = r & 1, call = b / n_win, w = b - call * n_win;
    return ((size_t)(call * kChan + ch) * n_win + w) * L;
}
template <typename SAMPLE>      // int16_t: the caller's PCM; float: the same samples in PCM units after the engine's resampling edge (audio.float() + F.interpolate, :630-644)
struct StereoFrameB {          // B(k, j) = reflect-padded sample k of frame j = (row r, t), * 2^-15 (:326-327, :649)
    static constexpr bool kAlongN = false;
    const SAMPLE* pcm;
    int L, T, n_win;
    __device__ float operator()(int k, int j) const {
        const int r = j / T, t = j - r * T;
        int idx = t * kHopM + k - kNfftM / 2;
        if (idx < 0) idx = -idx;
        else if (idx >= L) idx = 2 * (L - 1) - idx;
        return (float)pcm[pcm_row_offset(r, n_win, L) + idx] * (1.0f / 32768.0f);
    }
};
struct BinStore {              // C(c*1025 + f, (b, ch, t)) -> S[(f*2 + ch)*2 + c][b*T + t]   (:596)
    float* S;
    int T, BT;
    __device__ void operator()(int m, int n, float v) const {
        const int c = m >= kBinsM ? 1 : 0, f = m - c * kBinsM;
        const int r = n / T, t = n - r * T, b = r >> 1, ch = r & 1;
        S[(size_t)((f * 2 + ch) * 2 + c) * BT + b * T + t] = v;
    }
};
struct BandGatherA {           // A(bt, k) = S[gcol[off + k]][bt]: column k of band's slice of the gathered (t, 2S) matrix (:597-598)
    static constexpr bool kAlongK = false;
    const float* S;
    const int* gcol;           // (fi[col >> 1] * 2 + (col & 1)) per gathered column
    int BT;
    __device__ float operator()(int m, int k) const { return S[(size_t)gcol[k] * BT + m]; }
};
struct ScaleBiasStore {        // out[(row0 + m) * ld + n] = v * scale[m] + bias[n]
    static constexpr bool kCtx = true;
    float* out;
    const float* scale;
    const float* bias;
    int ld;
    __device__ float row(int m) const { return scale[m]; }
    __device__ float col(int n) const { return bias[n]; }
    __device__ gemm::None pre(int, int, float) const { return gemm::None{}; }
    __device__ void operator()(int m, int n, float v, float sc, float b, gemm::None) const { out[(size_t)m * ld + n] = v * sc + b; }
};
// The attention in-projection's store: columns n < rot_cols (the q and k blocks) also get the rotary embedding (:552, :438-453) here, so the attention core
// reads them ready-made instead of rotating every key once per query block.  rotate_half is a swap inside (even, odd) column pairs with the sign folded into
// rsin; the partner column lives in the neighbouring lane of the accumulator tile (same row), and whole 128-column tiles are either rotated or not.
struct RotaryQkStore {
    static constexpr bool kCtx = true;
    float* out;
    const float* scale;
    const float* bias;
    const float *rcos, *rsin;      // [position][kDh]
    int ld, rot_cols, pos_stride, n_pos;
    struct RowC { float sc; int tab; };                              // row scale, offset of the row's position in the rotary tables
    __device__ RowC row(int m) const { return RowC{scale[m], ((m / pos_stride) % n_pos) * kDh}; }
    __device__ float col(int n) const { return bias[n]; }
    __device__ float2 pre(int, int n, const RowC& r) const {         // (cos, sin) of (position, n mod 64); the v / gate columns are not rotated
        const int at = r.tab + (n & (kDh - 1));
        return n < rot_cols ? make_float2(rcos[at], rsin[at]) : make_float2(1.0f, 0.0f);
    }
    __device__ void operator()(int m, int n, float v, const RowC& r, float b, float2 cs) const {
        const float u = v * r.sc + b;
        const float partner = __shfl_xor(u, 1, 64);                  // same row, column n ^ 1
        // explicit rounding order: the guarded and the interior copy of the epilogue must not contract this sum of two products differently
        out[(size_t)m * ld + n] = n < rot_cols ? __fmaf_rn(u, cs.x, __fmul_rn(partner, cs.y)) : u;
    }
    // float4 form (gemm::HasV4: transposed tiles, a lane holds columns n .. n + 3 of a row -- both members of a rotary pair, no lane exchange; the same products and sums)
    static constexpr bool kV4 = true;
    __host__ __device__ bool can_v4(int N) const {
        return !((N | ld | rot_cols) & 3) && !(((size_t)out | (size_t)bias | (size_t)rcos | (size_t)rsin) & 15);
    }
    struct Pre4 { float4 c, s; };
    __device__ float4 col4(int n) const { return *reinterpret_cast<const float4*>(bias + n); }
    __device__ Pre4 pre4(int, int n, const RowC& r) const {
        const int at = r.tab + (n & (kDh - 1));
        const float* pc = n < rot_cols ? rcos + at : rcos;                   // (loads through a selected pointer; the v / gate columns ignore what they read)
        const float* ps = n < rot_cols ? rsin + at : rsin;
        return Pre4{*reinterpret_cast<const float4*>(pc), *reinterpret_cast<const float4*>(ps)};
    }
    __device__ void store4(int m, int n, float4 v, const RowC& r, const float4& b, const Pre4& p) const {
        const float ux = v.x * r.sc + b.x, uy = v.y * r.sc + b.y, uz = v.z * r.sc + b.z, uw = v.w * r.sc + b.w;
        float4 o = make_float4(ux, uy, uz, uw);
        if (n < rot_cols)
            o = make_float4(__fmaf_rn(ux, p.c.x, __fmul_rn(uy, p.s.x)), __fmaf_rn(uy, p.c.y, __fmul_rn(ux, p.s.y)), __fmaf_rn(uz, p.c.z, __fmul_rn(uw, p.s.z)),
                            __fmaf_rn(uw, p.c.w, __fmul_rn(uz, p.s.w)));
        *reinterpret_cast<float4*>(out + (size_t)m * ld + n) = o;
    }
};
struct ScaleBiasGeluStore {    // gelu(v * scale[m] + bias[n]), erf form (:564)
    static constexpr bool kCtx = true;
    float* out;
    const float* scale;
    const float* bias;
    int ld;
    __device__ float row(int m) const { return scale[m]; }
    __device__ float col(int n) const { return bias[n]; }
    __device__ gemm::None pre(int, int, float) const { return gemm::None{}; }
    __device__ void operator()(int m, int n, float v, float sc, float b, gemm::None) const {
        const float x = v * sc + b;
        out[(size_t)m * ld + n] = gelu1(x);
    }
    static __device__ __forceinline__ float gelu1(float x) { return 0.5f * x * (1.0f + gemm16::erf_fast(x * 0.70710678118654752440f)); }
    static constexpr bool kV4 = true;
    __host__ __device__ bool can_v4(int N) const { return !((N | ld) & 3) && !(((size_t)out | (size_t)bias) & 15); }
    __device__ float4 col4(int n) const { return *reinterpret_cast<const float4*>(bias + n); }
    __device__ gemm::None pre4(int, int, float) const { return gemm::None{}; }
    __device__ void store4(int m, int n, float4 v, float sc, const float4& b, gemm::None) const {
        *reinterpret_cast<float4*>(out + (size_t)m * ld + n) = make_float4(gelu1(v.x * sc + b.x), gelu1(v.y * sc + b.y), gelu1(v.z * sc + b.z), gelu1(v.w * sc + b.w));
        // (round 5) erf to 1.5e-7 on the hardware exp / rcp: erff() is ~40 instructions per element of the FFN's hidden tensor
    }
};
struct ResidualStore {         // x[m][n] += v (+ bias[n])    (:569-570)
    static constexpr bool kCtx = true;
    float* x;
    const float* bias;         // may be null
    int ld;
    struct ColC { float b; bool has; };
    __device__ gemm::None row(int) const { return gemm::None{}; }
    __device__ ColC col(int n) const { return ColC{bias ? bias[n] : 0.0f, bias != nullptr}; }
    __device__ float pre(int m, int n, gemm::None) const { return x[(size_t)m * ld + n]; }
    __device__ void operator()(int m, int n, float v, gemm::None, const ColC& c, float old) const { x[(size_t)m * ld + n] = old + (c.has ? v + c.b : v); }
    static constexpr bool kV4 = true;
    __host__ __device__ bool can_v4(int N) const { return !((N | ld) & 3) && !(((size_t)x | (size_t)bias) & 15); }
    struct ColC4 { float4 b; bool has; };
    __device__ ColC4 col4(int n) const { return ColC4{bias ? *reinterpret_cast<const float4*>(bias + n) : make_float4(0.0f, 0.0f, 0.0f, 0.0f), bias != nullptr}; }
    __device__ float4 pre4(int m, int n, gemm::None) const { return *reinterpret_cast<const float4*>(x + (size_t)m * ld + n); }
    __device__ void store4(int m, int n, float4 v, gemm::None, const ColC4& c, const float4& old) const {
        *reinterpret_cast<float4*>(x + (size_t)m * ld + n) = make_float4(old.x + (c.has ? v.x + c.b.x : v.x), old.y + (c.has ? v.y + c.b.y : v.y), old.z + (c.has ? v.z + c.b.z : v.z),
                                                                         old.w + (c.has ? v.w + c.b.w : v.w));
    }
};
struct BiasTanhStore {         // tanh(v + bias[n])   (:581-582)
    static constexpr bool kCtx = true;
    float* out;
    const float* bias;
    int ld;
    __device__ gemm::None row(int) const { return gemm::None{}; }
    __device__ float col(int n) const { return bias[n]; }
    __device__ gemm::None pre(int, int, gemm::None) const { return gemm::None{}; }
    __device__ void operator()(int m, int n, float v, gemm::None, float b, gemm::None) const { out[(size_t)m * ld + n] = tanhf(v + b); }
};
struct BiasTransposedStore {   // YT[(col0 + n) * BT + m] = v + bias[n]: the raw last Linear of the mask estimator, column-major
    static constexpr bool kCtx = true;
    float* yt;
    const float* bias;
    int BT;
    __device__ gemm::None row(int) const { return gemm::None{}; }
    __device__ float col(int n) const { return bias[n]; }
    __device__ gemm::None pre(int, int, gemm::None) const { return gemm::None{}; }
    __device__ void operator()(int m, int n, float v, gemm::None, float b, gemm::None) const { yt[(size_t)n * BT + m] = v + b; }
};
struct PlanarSpecA {           // A(j, k) = MS[k][j]: masked spectrum, k = c*1025 + f, j = (b, ch, t)
    static constexpr bool kAlongK = false;
    const float* ms;
    int J;
    __device__ float operator()(int j, int k) const { return ms[(size_t)k * J + j]; }
};

struct BandTable {             // per band: first gathered column, width d_i, weight offsets into the arena (floats)
    const int* off;            // [nb + 1] prefix sums of dim_inputs
    const long long* bs_w;     // [nb]
    const long long* bs_b;
    const long long* w3;
    const long long* b3;
};
template <class A, class B, class S>
struct Prob { A a; B b; S st; int M, N, K; };

struct BandSplitProb {         // z = band: X[band rows] = (gathered slice) x bs_w_z^T * inv_norm + bs_b_z      (:574-577)
    const float* S;
    const int* gcol;
    const float* arena;
    BandTable bt;
    const float* invn;         // [nb][BT]
    float* X;
    int BT, dim;
    __device__ Prob<BandGatherA, gemm::WeightNK, ScaleBiasStore> operator()(int z) const {
        const int off = bt.off[z], d = bt.off[z + 1] - off;
        return {BandGatherA{S, gcol + off, BT}, gemm::WeightNK{arena + bt.bs_w[z], d},
                ScaleBiasStore{X + (size_t)z * BT * dim, invn + (size_t)z * BT, arena + bt.bs_b[z], dim}, BT, dim, d};
    }
};
struct MeHiddenProb {          // z = band: out_z = tanh(in_z x Wt_z + b_z), Wt (K, N) row-major (:581-582)
    const float* in;
    const float* wt;
    const float* bias;
    float* out;
    int BT, K, N;
    __device__ Prob<gemm::RowMajorA, gemm::RowMajorB, BiasTanhStore> operator()(int z) const {
        return {gemm::RowMajorA{in + (size_t)z * BT * K, K}, gemm::RowMajorB{wt + (size_t)z * K * N, N},
                BiasTanhStore{out + (size_t)z * BT * N, bias + (size_t)z * N, N}, BT, N, K};
    }
};
struct MeOutProb {             // z = band: YT[2 off_z ..][bt] = h_z x me_w3_z^T + me_b3_z     (:583)
    const float* in;
    const float* arena;
    BandTable bt;
    float* yt;
    int BT, K;
    __device__ Prob<gemm::RowMajorA, gemm::WeightNK, BiasTransposedStore> operator()(int z) const {
        const int off = bt.off[z], d = bt.off[z + 1] - off;
        return {gemm::RowMajorA{in + (size_t)z * BT * K, K}, gemm::WeightNK{arena + bt.w3[z], K},
                BiasTransposedStore{yt + (size_t)2 * off * BT, arena + bt.b3[z], BT}, BT, 2 * d, K};
    }
};

// ---- row-wise kernels (one wavefront per row) ----------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// inv[row] = 1 / max(|x_row|_2, 1e-12)   (:533-538)
__global__ __launch_bounds__(256) void k_row_invnorm(const float* __restrict__ x, float* __restrict__ inv, int rows, int dim) {
    const int row = (int)blockIdx.x * 4 + ((int)threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    float s = 0.0f;
    for (int k = lane; k < dim; k += 64) { const float v = x[(size_t)row * dim + k]; s += v * v; }
    s = wave_sum(s);
    if (lane == 0) inv[row] = 1.0f / fmaxf(sqrtf(s), 1e-12f);
}

// x_row = x_row / max(|x_row|, eps) * g (:571), and inv[row] = 1 / max(|new x_row|, eps) for the next consumer
__global__ __launch_bounds__(256) void k_row_normalize_gain(float* __restrict__ x, const float* __restrict__ g, float* __restrict__ inv, int rows,
                                                            int dim) {
    const int row = (int)blockIdx.x * 4 + ((int)threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    float s = 0.0f;
    for (int k = lane; k < dim; k += 64) { const float v = x[(size_t)row * dim + k]; s += v * v; }
    const float r = 1.0f / fmaxf(sqrtf(wave_sum(s)), 1e-12f);
    float s2 = 0.0f;
    for (int k = lane; k < dim; k += 64) {
        const float v = x[(size_t)row * dim + k] * r * g[k];
        x[(size_t)row * dim + k] = v;
        s2 += v * v;
    }
    s2 = wave_sum(s2);
    if (lane == 0) inv[row] = 1.0f / fmaxf(sqrtf(s2), 1e-12f);
}

// band-input norms: inv[band][bt] = 1 / max(|gathered slice|, eps); thread = bt (coalesced over bt), loop over the band's columns
__global__ __launch_bounds__(256) void k_band_invnorm(const float* __restrict__ S, const int* __restrict__ gcol, const int* __restrict__ off,
                                                      float* __restrict__ inv, int BT) {
    const int band = blockIdx.y, bt = (int)blockIdx.x * 256 + threadIdx.x;
    if (bt >= BT) return;
    const int lo = off[band], hi = off[band + 1];
    float s = 0.0f;
    for (int c = lo; c < hi; ++c) { const float v = S[(size_t)gcol[c] * BT + bt]; s += v * v; }
    inv[(size_t)band * BT + bt] = 1.0f / fmaxf(sqrtf(s), 1e-12f);
}

// ---- attention core (:546-561) on the matrix cores, flash style ------------------------------------------------------------
// grid = (sequence, head, block of 64 queries); wave w of the workgroup owns queries 16 w .. 16 w + 15 of the block and keeps
// their rotated Q, running max / sum and the 16 x 64 output tile in registers, while the workgroup streams K (rotated) and V
// through LDS 64 keys at a time -- any sequence length, 35 KB of LDS, 4 workgroups per CU.
// Both products are computed TRANSPOSED so that no operand ever changes lanes (v_mfma_f32_16x16x4_f32: lane l supplies
// A[l & 15][l >> 4] and B[l >> 4][l & 15], and holds D[4 (l >> 4) + r][l & 15]):
//   S^T tile  = K_tile (16 keys x 64) . Q^T (64 x 16 queries): lane (g, j) holds S[key 4 g + r][query j]   -> softmax statistics are per LANE COLUMN j
//   O^T tile += V^T (16 dims x keys) . P^T (keys x 16 queries), contraction step s taking keys {4 k + s}: the B operand of lane (k, j) is its own
//               register p[s], the A operand V[key 4 k + s][dim]; the accumulator again has the query in the lane column, so the online-softmax
//               rescale is lane-local.
// row(seq, p) = seq * seq_stride + p * pos_stride.
constexpr int kKc = 64;            // keys per LDS chunk
constexpr int kKvStride = 68;      // floats per staged row: 16 rows x one float4 per lane group cover all 64 banks once (conflict-free ds_read_b128)

// One chunk = 64 keys.  Contraction indices are mapped k = 16 ks + 4 g + s (ks = 0..3 slabs, g = lane >> 4, s = the four steps of a slab) on BOTH operands, so a
// lane's operands of four MFMA steps are ONE ds_read_b128: K rows are staged row-major (key, dim), V TRANSPOSED (dim, key).  The online-softmax rescale runs once per
// chunk (after all four score tiles), not once per 16 keys.  A wave owns QT tiles of 16 queries: every K / V operand read from LDS feeds QT MFMA chains, so with
// QT = 2 (sequences longer than 64) a chunk is 128 + 128 MFMAs per wave against 32 ds_read_b128 -- at QT = 1 the sixteen waves of a CU ask LDS for its full
// 128 bytes per cycle and the matrix cores wait for it.
template <int QT>
__global__ __launch_bounds__(256, 2) void k_attention(const float* __restrict__ qkvg, float* __restrict__ ao, int n, long long seq_stride, long long pos_stride, int ldq,
                                                      int di) {
    constexpr int kPitch = kKvStride;
    __shared__ __attribute__((aligned(16))) float Ks[kKc * kPitch];            // [key][dim]
    __shared__ __attribute__((aligned(16))) float Vt[kDh * kPitch];            // [dim][key]
    constexpr int kQw = 16 * QT, kQb = 4 * kQw;                                 // queries per wave / per workgroup
    // Block order: the query blocks of one (sequence, head) are CONSECUTIVE logical ids on ONE XCD (gemm::xcd_contiguous_id), so its K / V rows come from HBM once and the
    // other query blocks find them in that XCD's L2.  With (sequence, head, block) as the grid's x / y / z the blocks of a sequence were nseq * heads ids apart and every one
    // of them fetched K / V from HBM again (counters: 11.4 GB per launch against 1.2 GB of q | k | v at 8 clips -- the kernel ran at the HBM rate, not the matrix rate).
    const int nqb = (int)gridDim.z, nhead = (int)gridDim.y;
    const int id = gemm::xcd_contiguous_id((int)blockIdx.x + (int)gridDim.x * ((int)blockIdx.y + nhead * (int)blockIdx.z), (int)gridDim.x * nhead * nqb);
    const int qblk = id % nqb, head = (id / nqb) % nhead, seq = id / (nqb * nhead);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j16 = lane & 15, g = lane >> 4;
    const long long row0 = (long long)seq * seq_stride;
    int qi[QT];
    bool q_ok[QT];
    size_t qrow[QT];
    float4 qreg[QT][4];                                                         // Q[query j16 of tile t][d = 16 ks + 4 g + s] (rotary applied by the in-projection's store)
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        qi[t] = qblk * kQb + wave * kQw + 16 * t + j16;                         // this lane's query of tile t
        q_ok[t] = qi[t] < n;
        qrow[t] = (size_t)(row0 + (long long)(q_ok[t] ? qi[t] : 0) * pos_stride);
        const float* src = qkvg + qrow[t] * ldq + head * kDh + 4 * g;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            qreg[t][ks] = keep4(q_ok[t], *reinterpret_cast<const float4*>(src + 16 * ks));
        }
    }
    float m[QT], l[QT];
    v4f acc[QT][4];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        m[t] = -INFINITY;
        l[t] = 0.0f;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) acc[t][dt] = v4f{0.0f, 0.0f, 0.0f, 0.0f};
    }
    const bool wave_live = qblk * kQb + wave * kQw < n;                        // a wave whose queries are all padding only helps loading

    // K / V staging, software-pipelined: the 64 keys of chunk c + 1 are requested into registers (four 16-byte K and V pieces per lane: key p = i >> 4, dims 4 (i & 15) ..)
    // right after chunk c has been written to LDS, and land under chunk c's MFMAs.  Padded keys are zero rows: their p is 0 and 0 * 0 stays 0.
    float4 pk[4], pv[4];
    auto request = [&](int c0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = tid + 256 * u, p = i >> 4, d = (i & 15) * 4, key = c0 + p;
            const bool ok = key < n;
            const float* src = qkvg + (size_t)(row0 + (long long)(ok ? key : 0) * pos_stride) * ldq + head * kDh + d;
            pk[u] = ld4_or_zero(ok, src + di);            // (zeros by address, not by a select on the loaded value: the requests stay in flight across the chunk's products)
            pv[u] = ld4_or_zero(ok, src + 2 * di);
        }
    };
    request(0);
    for (int c0 = 0; c0 < n; c0 += kKc) {
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = tid + 256 * u, p = i >> 4, d = (i & 15) * 4;
            *reinterpret_cast<float4*>(Ks + p * kPitch + d) = pk[u];
            Vt[d * kPitch + p] = pv[u].x; Vt[(d + 1) * kPitch + p] = pv[u].y; Vt[(d + 2) * kPitch + p] = pv[u].z; Vt[(d + 3) * kPitch + p] = pv[u].w;
        }
        __syncthreads();
        if (c0 + kKc < n) request(c0 + kKc);
        if (!wave_live) continue;
        v4f st[QT][4];
        // The K operand of step e + 1 (e = 4 kt + ks) is requested BEFORE the MFMAs of step e (a second float4; the scheduling barrier keeps the compiler from folding the
        // two back into one register set).  With one set every ds_read_b128 was issued behind the products that still used it and waited for with an empty pipe -- at two
        // wavefronts per SIMD nothing else covered it: the kernel ran at 67 % matrix-busy (round 5; the same fault as in ade_gemm.h's tile).
        {
            float4 kq[2];
            kq[0] = *reinterpret_cast<const float4*>(Ks + j16 * kPitch + 4 * g);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int kt = e >> 2, ks = e & 3;
                if (e + 1 < 16) kq[(e + 1) & 1] = *reinterpret_cast<const float4*>(Ks + (16 * ((e + 1) >> 2) + j16) * kPitch + 16 * ((e + 1) & 3) + 4 * g);
                __builtin_amdgcn_sched_barrier(0);
                const float4 kv = kq[e & 1];
#pragma unroll
                for (int t = 0; t < QT; ++t) {
                    if (ks == 0) st[t][kt] = v4f{0.0f, 0.0f, 0.0f, 0.0f};
                    st[t][kt] = mfma16x16x4(kv.x, qreg[t][ks].x, st[t][kt]);
                    st[t][kt] = mfma16x16x4(kv.y, qreg[t][ks].y, st[t][kt]);
                    st[t][kt] = mfma16x16x4(kv.z, qreg[t][ks].z, st[t][kt]);
                    st[t][kt] = mfma16x16x4(kv.w, qreg[t][ks].w, st[t][kt]);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                const int key0 = c0 + 16 * kt + 4 * g;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (key0 + r >= n) st[t][kt][r] = -INFINITY;
                    mx = fmaxf(mx, st[t][kt][r]);
                }
            }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m[t], mx);                                // finite: every chunk starts with a real key
            const float alpha = __expf(m[t] - m_new);
            m[t] = m_new;
            float psum = 0.0f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) { st[t][kt][r] = __expf(st[t][kt][r] - m_new); psum += st[t][kt][r]; }
            l[t] = l[t] * alpha + psum;                                         // this lane's share of the row sum; the four g-lanes are added at the end
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) acc[t][dt] = acc[t][dt] * alpha;
        }
        {   // (the V^T operand one step ahead, as above; e = 4 kt + dt)
            float4 vq[2];
            vq[0] = *reinterpret_cast<const float4*>(Vt + j16 * kPitch + 4 * g);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int kt = e >> 2, dt = e & 3;
                if (e + 1 < 16) vq[(e + 1) & 1] = *reinterpret_cast<const float4*>(Vt + (16 * ((e + 1) & 3) + j16) * kPitch + 16 * ((e + 1) >> 2) + 4 * g);
                __builtin_amdgcn_sched_barrier(0);
                const float4 vv = vq[e & 1];
#pragma unroll
                for (int t = 0; t < QT; ++t) {
                    acc[t][dt] = mfma16x16x4(vv.x, st[t][kt][0], acc[t][dt]);
                    acc[t][dt] = mfma16x16x4(vv.y, st[t][kt][1], acc[t][dt]);
                    acc[t][dt] = mfma16x16x4(vv.z, st[t][kt][2], acc[t][dt]);
                    acc[t][dt] = mfma16x16x4(vv.w, st[t][kt][3], acc[t][dt]);
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        if (!q_ok[t]) continue;
        float lt = l[t];
        lt += __shfl_xor(lt, 16, 64);
        lt += __shfl_xor(lt, 32, 64);
        const float gate = 1.0f / (1.0f + expf(-qkvg[qrow[t] * ldq + 3 * di + head]));    // sigmoid(gates) (:559)
        const float sc = gate / lt;
        float* dst = ao + qrow[t] * di + head * kDh + 4 * g;                             // lane (g, j): dims 16 dt + 4 g + r of query j
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
            *reinterpret_cast<float4*>(dst + 16 * dt) = make_float4(acc[t][dt][0] * sc, acc[t][dt][1] * sc, acc[t][dt][2] * sc, acc[t][dt][3] * sc);
    }
}

// ---- bf16 path (ade_gemm_dtype = "bf16"): bf16 activations and weights STORED in HBM, v_mfma_f32_32x32x16_bf16 / 16x16x32 products, fp32 islands ------------
// What stays fp32: the STFT / ISTFT GEMMs and the band-split GEMM (the front and the PCM tail), the residual stream X and the norms over it, the softmax
// statistics and all accumulators, biases, rotary tables, the raw mask-estimator output YT and the GLU / scatter / complex mask.  What is bf16: the operand copies
// of the residual stream that feed the in-projection, the first FFN Linear and the mask estimator, q | k | v | gates, the attention output, the FFN and
// mask-estimator hidden activations, and every Linear weight.  Stores of csrc/ade_gemm16.h: (m, n .. n + 3) float4s, consecutive lanes on consecutive n.
using gemm16::bf16_t;

// The norms of the reference's transformer (:533-538, :571) never run as passes of their own on this path.  A transformer maps the stream x to
//     x1 = x + W_o attn(n(x)),   x2 = x1 + FF(n(x1)),   y = n(x2) * g          with n(v) = v / max(|v|_2, 1e-12) per row,
// and every n() is a ROW SCALAR: the product with a normalised operand is the product with the raw operand times that scalar, applied to the accumulator by the store.
// So the residual stream X holds the UN-normalised x2 of the last transformer, and what a Linear reads is a bf16 copy of its raw operand plus per-row partial sums of
// squares, one per 128-column tile of the GEMM that produced the operand (kept apart and added in tile order by the consumer: no atomics, the same bits every run):
//     FFN-out store     x2 = X + v + b         -> X ; Xg = bf16(x2 * g) ; sq2[m][tile] = sum x2^2 ; sqg[m][tile] = sum (x2 g)^2
//     in-projection     (q | k | v | gates) = (Xg W_in^T) / |x2 g| + b_in                       (n(y) = x2 g / |x2 g|: the 1 / |x2| inside y cancels)
//     out-proj store    x1 = X * g / |x2| + v  -> X ; X1 = bf16(x1) ; sq1[m][tile] = sum x1^2   (y itself is formed here, g and 1 / |x2| from the PREVIOUS transformer)
//     FFN-in store      gelu((X1 W_1^T) / |x1| + b_1)
//     mask estimator    tanh((Xg W^T) / |x2| + b)   (its input is y = n(x2) g)
// The first transformer reads the band split's output through k_row_prep16 (g = 1, sq = |x|^2); the tap "tokens" applies g / |x2| to X when asked.
constexpr int kSqTiles = 4;        // partial sums per row: dim <= 512 in 128-column tiles
// inv[m] = 1 / max(sqrt(sum of the row's partial sums), 1e-12): run once per produced operand, so that a consuming store reads ONE float per row (a consumer that
// added the partials and took the root itself, per float4, cost a K = 384 product 30 % of its time)
__global__ __launch_bounds__(256) void k_rows_inv_norm(const float* __restrict__ sq, int tiles, float* __restrict__ inv, const float* __restrict__ sq_b, float* __restrict__ inv_b, int rows) {
    const int m = (int)blockIdx.x * 256 + threadIdx.x;
    if (m >= rows) return;
    const float4 a = *reinterpret_cast<const float4*>(sq + (size_t)m * kSqTiles);
    inv[m] = 1.0f / fmaxf(sqrtf(tiles > 3 ? ((a.x + a.y) + a.z) + a.w : (tiles > 2 ? (a.x + a.y) + a.z : (tiles > 1 ? a.x + a.y : a.x))), 1e-12f);
    if (sq_b) {
        const float4 b = *reinterpret_cast<const float4*>(sq_b + (size_t)m * kSqTiles);
        inv_b[m] = 1.0f / fmaxf(sqrtf(tiles > 3 ? ((b.x + b.y) + b.z) + b.w : (tiles > 2 ? (b.x + b.y) + b.z : (tiles > 1 ? b.x + b.y : b.x))), 1e-12f);
    }
}
// sum over the 32 lanes that hold one row's 128 columns of a tile (lanes (tid & 31) of the epilogue of csrc/ade_gemm16.h): all 32 must be active
__device__ __forceinline__ float row32_sum(float v) {
    v += quad_rot<1>(v);
    v += quad_rot<2>(v);
    v += row_ror<4>(v);
    v += row_ror<8>(v);
    const Swapped t = swap16(v, v);          // a = [r0, r0, r2, r2], b = [r1, r1, r3, r3] of the four 16-lane rows: one v_permlane16_swap instead of a trip through the LDS crossbar
    return t.a + t.b;
}
struct RotaryQkStore16 {       // q | k | v | gates = (Xg W_in^T) / |x2 g| + b_in, rotary on the q and k blocks (:547-548, :552), -> bf16
    bf16_t* out;
    const float* bias;
    const float *rcos, *rsin;  // [position][kDh], rotate_half's sign folded into rsin
    int ld, rot_cols;
    gemm16::FastDiv pos_stride, n_pos;      // position of row m = (m / pos_stride) % n_pos
    __device__ float4 col(int n, int cnt) const { return gemm16::load_f32x4(bias + n, cnt); }
    __device__ int row_ctx(int m) const { const int q = pos_stride.div(m); return (q - n_pos.div(q) * (int)n_pos.d) * kDh; }      // the row's offset into the rotary tables (once per tile row: gemm16)
    __device__ void operator()(int m, int n, float4 v, int cnt, const float4& b, int row_at) const {
        float4 u = make_float4(v.x + b.x, v.y + b.y, v.z + b.z, v.w + b.w);      // (v arrives scaled by 1 / |operand row|: gemm16's row_scale)
        if (n < rot_cols) {                                          // rot_cols % 4 == 0: a float4 is rotated whole or not at all
            const int at = row_at + (n & (kDh - 1));
            const float4 c = *reinterpret_cast<const float4*>(rcos + at), sn = *reinterpret_cast<const float4*>(rsin + at);
            u = make_float4(__fmaf_rn(u.x, c.x, __fmul_rn(u.y, sn.x)), __fmaf_rn(u.y, c.y, __fmul_rn(u.x, sn.y)),
                            __fmaf_rn(u.z, c.z, __fmul_rn(u.w, sn.z)), __fmaf_rn(u.w, c.w, __fmul_rn(u.z, sn.w)));
        }
        gemm16::store_bf16x4(out + (size_t)m * ld + n, u, cnt);
    }
};
template <int ACT>             // 0: gelu (erf form, :564); 1: tanh (:581-582)
struct BiasActStore16 {        // act(v + bias[n]) -> bf16   (v arrives scaled by 1 / |operand row| where the operand is a normalised one: gemm16's row_scale)
    bf16_t* out;
    const float* bias;
    int ld;
    __device__ float4 col(int n, int cnt) const { return gemm16::load_f32x4(bias + n, cnt); }
    __device__ void operator()(int m, int n, float4 v, int cnt, const float4& b) const {
        if (ACT == 0) {            // GELU, two values per instruction (gemm16::gelu_pk)
            const v2f lo = gemm16::gelu_pk(mk2(v.x, v.y) + mk2(b.x, b.y)), hi = gemm16::gelu_pk(mk2(v.z, v.w) + mk2(b.z, b.w));
            gemm16::store_bf16x4(out + (size_t)m * ld + n, make_float4(lo[0], lo[1], hi[0], hi[1]), cnt);
        } else {
            gemm16::store_bf16x4(out + (size_t)m * ld + n, make_float4(tanh_f(v.x + b.x), tanh_f(v.y + b.y), tanh_f(v.z + b.z), tanh_f(v.w + b.w)), cnt);
        }
    }
};
struct OutProjStore16 {        // x1 = X * g / |x2| + v -> X (fp32) ; X1 = bf16(x1) ; sq1[m][tile] = sum over the tile's columns of x1^2      (N % 128 == 0)
    float* x;
    bf16_t* x1b;
    const float* g_prev;       // the previous transformer's gain, or null: X already holds the stream value (first transformer)
    const float* inv2;         // 1 / |x2 row| of the previous transformer (with g_prev)
    float* sq1;
    int ld;
    __device__ float4 col(int n, int) const { return g_prev ? *reinterpret_cast<const float4*>(g_prev + n) : make_float4(1.0f, 1.0f, 1.0f, 1.0f); }
    __device__ void operator()(int m, int n, float4 v, int, const float4& g) const {
        float* p = x + (size_t)m * ld + n;
        const float4 old = *reinterpret_cast<const float4*>(p);
        const float r = g_prev ? inv2[m] : 1.0f;
        const float4 y = make_float4(fmaf(old.x * r, g.x, v.x), fmaf(old.y * r, g.y, v.y), fmaf(old.z * r, g.z, v.z), fmaf(old.w * r, g.w, v.w));
        *reinterpret_cast<float4*>(p) = y;
        *reinterpret_cast<uint2*>(x1b + (size_t)m * ld + n) = gemm16::pack_bf16x4(y);
        const float s = row32_sum(y.x * y.x + y.y * y.y + y.z * y.z + y.w * y.w);
        if ((threadIdx.x & 31) == 0) sq1[(size_t)m * kSqTiles + (n >> 7)] = s;
    }
};
struct FfOutStore16 {          // x2 = X + v + b -> X (fp32) ; Xg = bf16(x2 * g) ; sq2 / sqg[m][tile] = sums of x2^2 / (x2 g)^2 over the tile's columns   (N % 128 == 0)
    float* x;
    bf16_t* xg;
    const float *bias, *g;
    float *sq2, *sqg;
    int ld;
    struct ColC { float4 b, g; };
    __device__ ColC col(int n, int) const { return ColC{*reinterpret_cast<const float4*>(bias + n), *reinterpret_cast<const float4*>(g + n)}; }
    __device__ void operator()(int m, int n, float4 v, int, const ColC& c) const {
        float* p = x + (size_t)m * ld + n;
        const float4 old = *reinterpret_cast<const float4*>(p);
        const float4 y = make_float4(old.x + (v.x + c.b.x), old.y + (v.y + c.b.y), old.z + (v.z + c.b.z), old.w + (v.w + c.b.w));
        const float4 yg = make_float4(y.x * c.g.x, y.y * c.g.y, y.z * c.g.z, y.w * c.g.w);
        *reinterpret_cast<float4*>(p) = y;
        *reinterpret_cast<uint2*>(xg + (size_t)m * ld + n) = gemm16::pack_bf16x4(yg);
        const float s2 = row32_sum(y.x * y.x + y.y * y.y + y.z * y.z + y.w * y.w);
        const float sg = row32_sum(yg.x * yg.x + yg.y * yg.y + yg.z * yg.z + yg.w * yg.w);
        if ((threadIdx.x & 31) == 0) { sq2[(size_t)m * kSqTiles + (n >> 7)] = s2; sqg[(size_t)m * kSqTiles + (n >> 7)] = sg; }
    }
};
struct RowBiasStoreF32 {       // yt[m][n] = v + bias[m], fp32 row-major with any ld (the mask estimator's last Linear computed transposed: rows = output columns, n = bt)
    float* yt;
    const float* bias;
    int ld;
    __device__ gemm16::NoCol col(int, int) const { return gemm16::NoCol{}; }
    __device__ void operator()(int m, int n, float4 v, int cnt, gemm16::NoCol) const {
        const float b = bias[m];
        float* p = yt + (size_t)m * ld + n;
        const float4 r = make_float4(v.x + b, v.y + b, v.z + b, v.w + b);
        if (cnt == 4 && (ld & 3) == 0) { *reinterpret_cast<float4*>(p) = r; return; }
        const float t[4] = {r.x, r.y, r.z, r.w};
        for (int i = 0; i < cnt; ++i) p[i] = t[i];
    }
};
struct MeHiddenProb16 {        // z = band: out_z = tanh((in_z W_z^T) [/ |x2 row|] + b_z), W_z (N, K) bf16
    const bf16_t* in;
    const bf16_t* w;
    const float* bias;
    const float* inv;          // first layer: 1 / |x2 row| (null for the second)
    bf16_t* out;
    int BT, K, N;
    __device__ gemm16::Prob<BiasActStore16<1>> operator()(int z) const {
        return {in + (size_t)z * BT * K, K, w + (size_t)z * N * K, K,
                BiasActStore16<1>{out + (size_t)z * BT * N, bias + (size_t)z * N, N}, BT, N, K, inv ? inv + (size_t)z * BT : nullptr};
    }
};
struct MeOutProb16 {           // z = band: YT[2 off_z + c][bt] = sum_k w3_z[c][k] h_z[bt][k] + b3_z[c]   (:583), the product taken transposed so that YT is its row-major result
    const bf16_t* in;
    const bf16_t* arena16;     // bf16 copies of the arena's me_w3_z at the same offsets
    const float* arena;
    BandTable bt;
    float* yt;
    int BT, K;
    __device__ gemm16::Prob<RowBiasStoreF32> operator()(int z) const {
        const int off = bt.off[z], d = bt.off[z + 1] - off;
        return {arena16 + bt.w3[z], K, in + (size_t)z * BT * K, K, RowBiasStoreF32{yt + (size_t)2 * off * BT, arena + bt.b3[z], BT}, 2 * d, BT, K, nullptr};
    }
};

// the band split's output as the first transformer's operand: Xg = bf16(x), invg[m] = 1 / |x|; one wavefront per row
__global__ __launch_bounds__(256) void k_row_prep16(const float* __restrict__ x, bf16_t* __restrict__ xg, float* __restrict__ invg, int rows, int dim) {
    const int row = (int)blockIdx.x * 4 + ((int)threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* xr = x + (size_t)row * dim;
    float s = 0.0f;
    for (int k = 4 * lane; k < dim; k += 256) {
        const float4 v = *reinterpret_cast<const float4*>(xr + k);
        s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        *reinterpret_cast<uint2*>(xg + (size_t)row * dim + k) = gemm16::pack_bf16x4(v);
    }
    s = wave_sum(s);
    if (lane == 0) invg[row] = 1.0f / fmaxf(sqrtf(s), 1e-12f);
}
// y = x2 * g / |x2| written out (the tap "tokens"): out of place, X keeps x2
__global__ __launch_bounds__(256) void k_row_apply_gain(const float* __restrict__ x, const float* __restrict__ g, const float* __restrict__ inv2, float* __restrict__ y, int rows, int dim) {
    const int row = (int)blockIdx.x * 4 + ((int)threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float r = inv2[row];
    for (int k = lane; k < dim; k += 64) y[(size_t)row * dim + k] = x[(size_t)row * dim + k] * r * g[k];
}

// Attention core on bf16 q | k | v (rotary applied by the in-projection's store), v_mfma_f32_16x16x32_bf16, flash style; same decomposition as k_attention above:
// grid = (sequence, head, block of 64 QT queries), wave w owns QT tiles of 16 queries, the workgroup streams K and V through LDS 64 keys at a time.
//   S^T tile (16 keys x 16 queries) = K_tile . Q^T over the 64 dims in two 32-deep steps: lane (g, j) supplies K[key j][dims 32 ks + 8 g .. + 7] (ONE ds_read_b128)
//     and Q[query j][the same dims] (registers, loaded once), and holds S[key 4 g + r][query j];
//   O^T tile (16 dims x 16 queries) += V^T . P^T over 32 keys per step: the B operand of lane (g, j) is its own eight probabilities of score tiles 2 kp and 2 kp + 1
//     (keys 16 (2 kp) + 4 g + r and 16 (2 kp + 1) + 4 g + r: contraction slot 8 g + e <-> key 16 (2 kp + (e >> 2)) + 4 g + (e & 3)), the A operand V^T[dim j][the same
//     keys]: two ds_read_b64 of the TRANSPOSED V chunk.  Scores, softmax statistics and the output accumulate in fp32; the output is written as bf16.
constexpr int kKPitch16 = 160;     // bytes per staged K row (64 dims): the 16 lanes of a ds_read_b128 service group (rows j, pieces g) cover all 64 banks once
constexpr int kVPitch16 = 136;     // bytes per staged V^T row (64 keys + 4)
__device__ __forceinline__ v4f mfma16x16x32(const uint4& a, const uint4& b, v4f c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(gemm16::as_v8bf(a), gemm16::as_v8bf(b), c, 0, 0, 0);
}
template <int QT>
__global__ __launch_bounds__(256, 3) void k_attention16(const bf16_t* __restrict__ qkvg, bf16_t* __restrict__ ao, int n, long long seq_stride, long long pos_stride, int ldq, int di) {
    __shared__ __attribute__((aligned(16))) unsigned char Ks[kKc * kKPitch16];           // [key][dim]
    __shared__ __attribute__((aligned(16))) unsigned char Vt[kDh * kVPitch16];           // [dim][key]
    constexpr int kQw = 16 * QT, kQb = 4 * kQw;
    const int nqb = (int)gridDim.z, nhead = (int)gridDim.y;                              // block order: see k_attention (the query blocks of a (sequence, head) share an XCD's L2)
    const int id = gemm::xcd_contiguous_id((int)blockIdx.x + (int)gridDim.x * ((int)blockIdx.y + nhead * (int)blockIdx.z), (int)gridDim.x * nhead * nqb);
    const int qblk = id % nqb, head = (id / nqb) % nhead, seq = id / (nqb * nhead);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j16 = lane & 15, g = lane >> 4;
    const long long row0 = (long long)seq * seq_stride;
    bool q_ok[QT];
    size_t qrow[QT];
    uint4 qb[QT][2];                                                                     // Q[query j16 of tile t][dims 32 ks + 8 g .. + 7]
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const int qi = qblk * kQb + wave * kQw + 16 * t + j16;
        q_ok[t] = qi < n;
        qrow[t] = (size_t)(row0 + (long long)(q_ok[t] ? qi : 0) * pos_stride);
        const bf16_t* src = qkvg + qrow[t] * ldq + head * kDh + 8 * g;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) qb[t][ks] = gemm16::zero_unless(q_ok[t], *reinterpret_cast<const uint4*>(src + 32 * ks));
    }
    float m[QT], l[QT];
    v4f acc[QT][4];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        m[t] = -INFINITY;
        l[t] = 0.0f;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) acc[t][dt] = v4f{0.0f, 0.0f, 0.0f, 0.0f};
    }
    const bool wave_live = qblk * kQb + wave * kQw < n;
    // staging: lane i = tid, tid + 256 = (key pair i >> 4, 16-byte piece (i >> 1) & 7, key of the pair i & 1): 16 lanes read the 128-byte K (and V) lines of two
    // consecutive keys; padded keys are zero rows.  V goes to LDS TRANSPOSED ([dim][key]): the two lanes of a key pair trade halves of their eight dims
    // (quad_perm DPP), so that each holds four dims of BOTH keys and writes four 32-bit words (dim, keys 2 kp | 2 kp + 1) -- 16-bit stores of single elements
    // were 4-way bank conflicts on a third of the LDS cycles.
    uint4 pk[2], pv[2];
    const bf16_t* kp[2];                                                                 // this lane's K pieces of the next chunk (advanced by a chunk per request: no per-chunk 64-bit address arithmetic)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int i = tid + 256 * u, p = 2 * (i >> 4) + (i & 1);
        kp[u] = qkvg + (size_t)(row0 + (long long)p * pos_stride) * ldq + head * kDh + 8 * ((i >> 1) & 7) + di;
    }
    const long long kstep = (long long)kKc * pos_stride * ldq;
    auto request = [&](int c0) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = tid + 256 * u, p = 2 * (i >> 4) + (i & 1);
            const bool ok = c0 + p < n;                                                  // (a padded key reads zeros by address: its pointer is never dereferenced)
            pk[u] = gemm16::ld8_or_zero(ok, kp[u]);
            pv[u] = gemm16::ld8_or_zero(ok, kp[u] + di);
            kp[u] += kstep;
        }
    };
    // m / l are kept in the exponent's units: scores are multiplied by log2(e) inside the fused multiply-add that subtracts the running maximum, exp is v_exp_f32 alone
    request(0);
    for (int c0 = 0; c0 < n; c0 += kKc) {
        const bool last = c0 + kKc >= n;                                                 // (uniform) only the last chunk has padded keys to mask
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = tid + 256 * u, kp2 = i >> 4, odd = i & 1, d0 = 8 * ((i >> 1) & 7);
            *reinterpret_cast<uint4*>(Ks + (2 * kp2 + odd) * kKPitch16 + 2 * d0) = pk[u];
            const unsigned w0 = pv[u].x, w1 = pv[u].y, w2 = pv[u].z, w3 = pv[u].w;         // dims d0 + (0,1), (2,3), (4,5), (6,7) of this lane's key
            const unsigned r0 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(odd ? w0 : w2), 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]: the partner's word
            const unsigned r1 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(odd ? w1 : w3), 0xB1, 0xF, 0xF, true);
            // even lane (key 2 kp): dims d0 .. d0 + 3, own = low halves; odd lane (key 2 kp + 1): dims d0 + 4 .. d0 + 7, own = high halves
            const unsigned lo0 = odd ? r0 : w0, hi0 = odd ? w2 : r0, lo1 = odd ? r1 : w1, hi1 = odd ? w3 : r1;
            unsigned char* vrow = Vt + (d0 + 4 * odd) * kVPitch16 + 4 * kp2;
            *reinterpret_cast<unsigned*>(vrow) = (lo0 & 0xffffu) | (hi0 << 16);
            *reinterpret_cast<unsigned*>(vrow + kVPitch16) = (lo0 >> 16) | (hi0 & 0xffff0000u);
            *reinterpret_cast<unsigned*>(vrow + 2 * kVPitch16) = (lo1 & 0xffffu) | (hi1 << 16);
            *reinterpret_cast<unsigned*>(vrow + 3 * kVPitch16) = (lo1 >> 16) | (hi1 & 0xffff0000u);
        }
        __syncthreads();
        if (!last) request(c0 + kKc);
        if (!wave_live) continue;
        v4f st[QT][4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
            for (int t = 0; t < QT; ++t) st[t][kt] = v4f{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const uint4 kb = *reinterpret_cast<const uint4*>(Ks + (16 * kt + j16) * kKPitch16 + 64 * ks + 16 * g);
#pragma unroll
                for (int t = 0; t < QT; ++t) st[t][kt] = mfma16x16x32(kb, qb[t][ks], st[t][kt]);
            }
        }
        if (last) {                       // (wave-uniform branch: the other chunks carry no compare / select per score)
#pragma unroll
            for (int t = 0; t < QT; ++t)
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) {
                    const int key0 = c0 + 16 * kt + 4 * g;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (key0 + r >= n) st[t][kt][r] = -INFINITY;
                }
        }
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, st[t][kt][r]);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m[t], mx * kLog2e);                       // finite: every chunk starts with a real key
            const float alpha = __builtin_amdgcn_exp2f(m[t] - m_new);
            m[t] = m_new;
            float psum = 0.0f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) { st[t][kt][r] = __builtin_amdgcn_exp2f(fmaf(st[t][kt][r], kLog2e, -m_new)); psum += st[t][kt][r]; }
            l[t] = l[t] * alpha + psum;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) acc[t][dt] = acc[t][dt] * alpha;
        }
#pragma unroll
        for (int kp = 0; kp < 2; ++kp) {
            uint4 pb[QT];
#pragma unroll
            for (int t = 0; t < QT; ++t)
                pb[t] = make_uint4(gemm16::pack_bf16x2(st[t][2 * kp][0], st[t][2 * kp][1]), gemm16::pack_bf16x2(st[t][2 * kp][2], st[t][2 * kp][3]),
                                   gemm16::pack_bf16x2(st[t][2 * kp + 1][0], st[t][2 * kp + 1][1]), gemm16::pack_bf16x2(st[t][2 * kp + 1][2], st[t][2 * kp + 1][3]));
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const unsigned char* vr = Vt + (16 * dt + j16) * kVPitch16 + 2 * (32 * kp + 4 * g);
                const uint2 lo = *reinterpret_cast<const uint2*>(vr), hi = *reinterpret_cast<const uint2*>(vr + 32);
                const uint4 vb = make_uint4(lo.x, lo.y, hi.x, hi.y);
#pragma unroll
                for (int t = 0; t < QT; ++t) acc[t][dt] = mfma16x16x32(vb, pb[t], acc[t][dt]);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        if (!q_ok[t]) continue;
        float lt = l[t];
        lt += __shfl_xor(lt, 16, 64);
        lt += __shfl_xor(lt, 32, 64);
        const float gv = gemm16::bf16_lo((unsigned)qkvg[qrow[t] * ldq + 3 * di + head]);
        const float sc = (1.0f / (1.0f + expf(-gv))) / lt;                                  // sigmoid(gates) (:559) / softmax denominator
        bf16_t* dst = ao + qrow[t] * di + head * kDh + 4 * g;                                // lane (g, j): dims 16 dt + 4 g + r of query j
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
            *reinterpret_cast<uint2*>(dst + 16 * dt) = gemm16::pack_bf16x4(make_float4(acc[t][dt][0] * sc, acc[t][dt][1] * sc, acc[t][dt][2] * sc, acc[t][dt][3] * sc));
    }
}

// GLU + scatter-add + complex mask (:583, :616-624).  thread = (bt, fc); the bands that own bin fc are listed in CSR order (ascending
// gathered index = the reference's scatter order).  Writes the masked spectrum planar for the synthesis GEMM: MS[c*1025 + f][(b, ch, t)].
__global__ __launch_bounds__(256) void k_mask_apply(const float* __restrict__ S, const float* __restrict__ yt, const int* __restrict__ csr_start,
                                                    const int* __restrict__ csr_col, const int* __restrict__ csr_d, float* __restrict__ ms,
                                                    float* __restrict__ mask_tap, int B, int T) {
    const int BT = B * T, bt = (int)blockIdx.x * 256 + threadIdx.x, fc = blockIdx.y;
    if (bt >= BT) return;
    float mr = 0.0f, mi = 0.0f;
    for (int e = csr_start[fc]; e < csr_start[fc + 1]; ++e) {
        const int col = csr_col[e], d = csr_d[e];
        const float ar = yt[(size_t)col * BT + bt], gr = yt[(size_t)(col + d) * BT + bt];
        const float ai = yt[(size_t)(col + 1) * BT + bt], gi = yt[(size_t)(col + 1 + d) * BT + bt];
        mr += ar * (1.0f / (1.0f + expf(-gr)));
        mi += ai * (1.0f / (1.0f + expf(-gi)));
    }
    const float re = S[(size_t)(fc * 2) * BT + bt], im = S[(size_t)(fc * 2 + 1) * BT + bt];
    const int f = fc >> 1, ch = fc & 1, b = bt / T, t = bt - b * T;
    const size_t J = (size_t)BT * kChan, j = (size_t)(b * kChan + ch) * T + t;
    ms[(size_t)f * J + j] = re * mr - im * mi;
    ms[(size_t)(kBinsM + f) * J + j] = re * mi + im * mr;
    if (mask_tap) { mask_tap[(size_t)(fc * 2) * BT + bt] = mr; mask_tap[(size_t)(fc * 2 + 1) * BT + bt] = mi; }
}

// overlap-add gather, / static COLA sum, then the PCM tail: * 32767, clamp, truncate (:667, :676)
__global__ __launch_bounds__(256) void k_melband_ola_pcm(const float* __restrict__ frames, const float* __restrict__ wsum, int16_t* __restrict__ pcm,
                                                         float* __restrict__ f32, int T, int L, int n_win, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int r = (int)(i / L), m = (int)(i - (long long)r * L) + kNfftM / 2;
    const size_t o = pcm_row_offset(r, n_win, L) + (size_t)(m - kNfftM / 2);      // stitch each channel's windows back (:663-664)
    int t_hi = m / kHopM;
    if (t_hi > T - 1) t_hi = T - 1;
    const int t_lo = m - kNfftM + 1 <= 0 ? 0 : (m - kNfftM + kHopM) / kHopM;
    float s = 0.0f;
    for (int t = t_lo; t <= t_hi; ++t) s += frames[((size_t)r * T + t) * kNfftM + (m - t * kHopM)];
    const float y = s / wsum[m - kNfftM / 2];
    if (f32) f32[o] = y;
    if (pcm) pcm[o] = (short)(int)fminf(fmaxf(y * 32767.0f, -32768.0f), 32767.0f);
}

int mfail(std::string& err, int st, const std::string& msg) { err = msg; return st; }
#define MB_HIP(expr)                                                                                  \
    do {                                                                                              \
        hipError_t _e = (expr);                                                                       \
        if (_e != hipSuccess) return mfail(err, ADE_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

struct TfW {
    const float *in_w, *in_b, *out_w, *ff1_w, *ff1_b, *ff2_w, *ff2_b, *out_g;
    const gemm16::bf16_t *in_w16, *out_w16, *ff1_w16, *ff2_w16;      // bf16 copies of the four Linear weights (bf16 path only)
};

}  // namespace

struct MelbandEngine : SubEngine {
    int device = 0, L = 0 /* one window */, n_win = 1, T = 0, depth = 0, nb = 0, dim = 0, di = 0, heads = 0, ffd = 0, med = 0, S2 = 0 /* gathered columns */, max_d = 0 /* widest band */;
    float* d_w = nullptr;          // arena: DFT tables, rotary tables, COLA sum, every fused buffer
    int* d_i = nullptr;            // arena of int tables
    long long* d_ll = nullptr;     // per-band weight offsets
    const float *k_fwd = nullptr, *k_inv = nullptr, *wsum = nullptr, *tcos = nullptr, *tsin = nullptr, *fcos = nullptr, *fsin = nullptr;
    const float *me_w1t = nullptr, *me_b1 = nullptr, *me_w2t = nullptr, *me_b2 = nullptr;
    std::vector<TfW> time_tf, freq_tf;
    const int *gcol = nullptr, *off = nullptr, *csr_start = nullptr, *csr_col = nullptr, *csr_d = nullptr;
    BandTable bt{};
    bool bf16 = false;             // ade_gemm_dtype = "bf16": the transformer stack and the mask estimator on bf16 activations / weights stored in HBM (csrc/ade_gemm16.h, fp32 accumulation);
                                   // STFT, band split, residual stream, norms, softmax statistics, GLU / mask, ISTFT stay fp32
    gemm16::bf16_t* d_w16 = nullptr;   // bf16 path: arena of bf16 weights (the fp32 arena's Linear weights at the same offsets; me_w1t / me_w2t transposed to (out, in))
    const gemm16::bf16_t *me_w1_16 = nullptr, *me_w2_16 = nullptr;
    gemm16::bf16_t *Xg = nullptr, *X1 = nullptr, *A16 = nullptr, *B16 = nullptr, *AO16 = nullptr;      // bf16 operand copies of the stream (see the stores), q | k | v | gates, hidden activations, attention output
    float *sq1 = nullptr, *sq2 = nullptr, *sqg = nullptr;      // per-row partial sums of squares [R][kSqTiles]
    float *inv1 = nullptr, *inv2 = nullptr, *invg = nullptr;   // 1 / max(|row|, 1e-12) of x1, x2, x2 * g
    const float* g_last = nullptr;                            // gain of the last transformer that ran (the stream is x2 * g / |x2|)
    int capacity = 0;
    float* ws = nullptr;
    float *Sp = nullptr, *X = nullptr, *invn = nullptr, *bufA = nullptr, *bufB = nullptr, *AO = nullptr, *YT = nullptr, *MS = nullptr,
          *frames_buf = nullptr, *mask_tap = nullptr;

    ~MelbandEngine() override {
        (void)hipSetDevice(device);
        if (d_w) (void)hipFree(d_w);
        if (d_i) (void)hipFree(d_i);
        if (d_ll) (void)hipFree(d_ll);
        if (d_w16) (void)hipFree(d_w16);
        if (ws) (void)hipFree(ws);
    }
    int frames() const override { return T; }
    int Lo = 0;                // output samples per window: L for a static export; 441 (T - 1) + 1024 with the dynamic-length ISTFT trim (Stereo/STFT_Process.py:296-306)
    int in_len() const override { return L * n_win; }
    int out_len() const override { return Lo * n_win; }
    bool accepts_float_input() const override { return true; }
    int channels() const override { return kChan; }
    int reserve(int batch, std::string& err) override;
    int run(hipStream_t s, const int16_t* d_in, int batch, int16_t* d_out, float* d_f32, std::string& err) override;
    int tap(hipStream_t s, const char* name, int batch, float* out, size_t count, size_t* written, std::string& err) override;
    void transformer(hipStream_t s, const TfW& w, int R, int n, int nseq, long long seq_stride, long long pos_stride, const float* rc, const float* rs);
};

int melband_create(const std::map<std::string, Tensor>& tensors, int in_len, int n_win, bool exact_dft, bool bf16, bool dynamic, int device, SubEngine** out, std::string& err) {
    *out = nullptr;
    if (n_win < 1) return mfail(err, ADE_ERR_BAD_VALUE, "melband: n_win must be >= 1");
    if (dynamic && n_win != 1) return mfail(err, ADE_ERR_BAD_VALUE, "melband: batch folding requires a static shape (Export_MelBandRoformer.py:315-316)");
    if (in_len < kNfftM || (!dynamic && in_len % kHopM != 0))
        return mfail(err, ADE_ERR_SHAPE_MISMATCH, "melband: input_audio_length must be a multiple of the 441-sample hop and at least 2048 (any length >= 2048 with dynamic_axes=1)");
    const int T = in_len / kHopM + 1;
    // static export: the trim [n_fft / 2 : raw - n_fft / 2] = 441 (T - 1) samples (= in_len); dynamic: [n_fft / 2 : out_end(max_frames)] = everything after the first half window
    const int out_one = dynamic ? kHopM * (T - 1) + kNfftM / 2 : kHopM * (T - 1);
    auto find = [&](const std::string& name) -> const Tensor* {
        auto it = tensors.find(name);
        if (it == tensors.end()) { err = "weights: tensor missing: " + name; return nullptr; }
        return &it->second;
    };
    auto shaped = [&](const std::string& name, std::vector<int> dims) -> const Tensor* {
        const Tensor* t = find(name);
        if (t && t->dims != dims) { err = "weights: tensor has the wrong shape: " + name; return nullptr; }
        return t;
    };
    auto status = [&]() { return err.find("missing") != std::string::npos ? ADE_ERR_MISSING_KEY : ADE_ERR_SHAPE_MISMATCH; };
    const Tensor *t_fi = find("freq_indices"), *t_di = find("dim_inputs");
    if (!t_fi || !t_di) return status();
    const int nb = (int)t_di->count, nsel = (int)t_fi->count;
    std::vector<int> off((size_t)nb + 1, 0);
    for (int i = 0; i < nb; ++i) {
        const int d = (int)t_di->data[i];
        if (d < 2 || (d & 1)) return mfail(err, ADE_ERR_BAD_VALUE, "melband: dim_inputs entries must be positive and even");
        off[i + 1] = off[i] + d;
    }
    if (off[nb] != 2 * nsel) return mfail(err, ADE_ERR_SHAPE_MISMATCH, "melband: sum(dim_inputs) must equal 2 * len(freq_indices)");
    if (nb < 1 || nb > kMaxBands) return mfail(err, ADE_ERR_UNSUPPORTED, "melband: 1..1024 bands supported");
    if (T > kMaxFrames) return mfail(err, ADE_ERR_UNSUPPORTED, "melband: at most 8192 frames per clip; fold longer audio into windows");
    const Tensor* t_bs0 = find("bs_w_0");
    if (!t_bs0 || t_bs0->dims.size() != 2) return status();
    const int dim = t_bs0->dims[0];
    int depth = 0;
    while (tensors.count("time" + std::to_string(depth) + "_in_w")) ++depth;
    if (depth < 1) return mfail(err, ADE_ERR_MISSING_KEY, "weights: tensor missing: time0_in_w");
    const Tensor* t_ow = find("time0_out_w");
    const Tensor* t_f1 = find("time0_ff1_w");
    const Tensor* t_m1 = find("me_w1t");
    if (!t_ow || !t_f1 || !t_m1 || t_ow->dims.size() != 2 || t_f1->dims.size() != 2 || t_m1->dims.size() != 3) return status();
    const int di = t_ow->dims[1], ffd = t_f1->dims[0], med = t_m1->dims[2];
    if (di % kDh) return mfail(err, ADE_ERR_UNSUPPORTED, "melband: dim_inner must be a multiple of dim_head = 64");
    const int heads = di / kDh, ldq = 3 * di + heads;
    if (ldq % 4) return mfail(err, ADE_ERR_UNSUPPORTED, "melband: the attention kernel stages K / V with 16-byte loads: heads must be a multiple of 4");

    MelbandEngine* e = new MelbandEngine();
    e->bf16 = bf16;
    e->device = device; e->L = in_len; e->Lo = out_one; e->n_win = n_win; e->T = T; e->depth = depth; e->nb = nb; e->dim = dim; e->di = di; e->heads = heads; e->ffd = ffd; e->med = med;
    e->S2 = off[nb];
    for (int i = 0; i < nb; ++i) e->max_d = std::max(e->max_d, off[i + 1] - off[i]);
    auto bail = [&](int st) { delete e; return st; };

    // ---- collect every tensor with its place in the arena
    struct Item { const float* src; size_t n; size_t at; };
    std::vector<Item> items;
    size_t arena = 0;
    auto place = [&](const float* src, size_t n) { const size_t at = arena; arena += (n + 63) & ~(size_t)63; items.push_back({src, n, at}); return at; };
    std::vector<long long> ll((size_t)4 * nb);
    for (int i = 0; i < nb; ++i) {
        const int d = off[i + 1] - off[i];
        const std::string s = std::to_string(i);
        const Tensor *w = shaped("bs_w_" + s, {dim, d}), *b = shaped("bs_b_" + s, {dim}), *w3 = shaped("me_w3_" + s, {2 * d, med}), *b3 = shaped("me_b3_" + s, {2 * d});
        if (!w || !b || !w3 || !b3) return bail(status());
        ll[i] = (long long)place(w->data, w->count); ll[nb + i] = (long long)place(b->data, b->count);
        ll[2 * nb + i] = (long long)place(w3->data, w3->count); ll[3 * nb + i] = (long long)place(b3->data, b3->count);
    }
    struct TfAt { size_t v[8]; };
    std::vector<TfAt> tf_at;
    for (int i = 0; i < depth; ++i)
        for (const char* axis : {"time", "freq"}) {
            const std::string p = std::string(axis) + std::to_string(i) + "_";
            const Tensor* ts[8] = {shaped(p + "in_w", {ldq, dim}), shaped(p + "in_b", {ldq}), shaped(p + "out_w", {dim, di}), shaped(p + "ff1_w", {ffd, dim}),
                                   shaped(p + "ff1_b", {ffd}), shaped(p + "ff2_w", {dim, ffd}), shaped(p + "ff2_b", {dim}), shaped(p + "out_g", {dim})};
            TfAt at{};
            for (int k = 0; k < 8; ++k) {
                if (!ts[k]) return bail(status());
                at.v[k] = place(ts[k]->data, ts[k]->count);
            }
            tf_at.push_back(at);
        }
    const Tensor *m1 = shaped("me_w1t", {nb, dim, med}), *mb1 = shaped("me_b1", {nb, 1, med}), *m2 = shaped("me_w2t", {nb, med, med}), *mb2 = shaped("me_b2", {nb, 1, med});
    if (!m1 || !mb1 || !m2 || !mb2) return bail(status());
    const size_t a_m1 = place(m1->data, m1->count), a_mb1 = place(mb1->data, mb1->count), a_m2 = place(m2->data, m2->count), a_mb2 = place(mb2->data, mb2->count);

    // ---- host-built tables: DFT matrices (exact angles), COLA sum, rotary tables (:395-401, :438-449)
    std::vector<float> fwd((size_t)2 * kBinsM * kNfftM), inv((size_t)2 * kBinsM * kNfftM), win((size_t)kNfftM), wsum((size_t)out_one, 0.0f);
    {
        const float step = (float)(2.0 * M_PI / (double)kNfftM);
        for (int n = 0; n < kNfftM; ++n) win[n] = cosf((float)n * step) * (-0.5f) + 0.5f;       // torch.hann_window(periodic=True) in fp32
        for (int f = 0; f < kBinsM; ++f) {
            const float scale = (f == 0 || f == kBinsM - 1) ? 1.0f : 2.0f;
            for (int n = 0; n < kNfftM; ++n) {
                float c, s;
                if (exact_dft) {
                    const double a = 2.0 * M_PI * (double)(((long long)f * n) % kNfftM) / kNfftM;
                    c = (float)cos(a); s = (float)sin(a);
                } else {        // the reference's tables: cos / sin of the fp32 product (2 pi / N) * f * n (Stereo/STFT_Process.py:205-243)
                    const float omega = (step * (float)f) * (float)n;
                    c = cosf(omega); s = sinf(omega);
                }
                fwd[(size_t)f * kNfftM + n] = c * win[n];
                fwd[(size_t)(kBinsM + f) * kNfftM + n] = -s * win[n];
                inv[(size_t)f * kNfftM + n] = ((scale * c) * (float)(1.0 / kNfftM)) * win[n];
                inv[(size_t)(kBinsM + f) * kNfftM + n] = ((scale * -s) * (float)(1.0 / kNfftM)) * win[n];
            }
        }
        std::vector<float> raw((size_t)kNfftM + (size_t)kHopM * (T - 1), 0.0f);
        for (int t = 0; t < T; ++t)
            for (int n = 0; n < kNfftM; ++n) raw[(size_t)t * kHopM + n] += win[n] * win[n];
        for (int m = 0; m < out_one; ++m) wsum[m] = raw[(size_t)m + kNfftM / 2];      // (the dynamic tail has only the last frame under it: w^2 down to ~0 at the end)
    }
    auto half_round = [](float v) {      // float -> IEEE binary16 -> float, round to nearest even (|v| <= 1 here: cos / sin)
        uint32_t u;
        memcpy(&u, &v, 4);
        const uint32_t sign = u & 0x80000000u;
        uint32_t a = u & 0x7fffffffu;
        if (a >= 0x38800000u) {          // a normal half: keep 10 mantissa bits
            a += 0xfffu + ((a >> 13) & 1u);
            a &= ~0x1fffu;
        } else {                         // a subnormal half: multiples of 2^-24
            float f;
            memcpy(&f, &a, 4);
            f = rintf(f * 16777216.0f) * (1.0f / 16777216.0f);
            memcpy(&a, &f, 4);
        }
        u = sign | a;
        memcpy(&v, &u, 4);
        return v;
    };
    std::vector<float> tcos((size_t)T * kDh), tsin((size_t)T * kDh), fcos((size_t)nb * kDh), fsin((size_t)nb * kDh);
    {
        float inv_freq[kDh / 2];
        for (int i = 0; i < kDh / 2; ++i) inv_freq[i] = (float)pow(10000.0, -(double)((float)(2 * i) / (float)kDh));
        const int npos = T > nb ? T : nb;
        for (int p = 0; p < npos; ++p)
            for (int d = 0; d < kDh; ++d) {
                const float ang = (float)p * inv_freq[d >> 1];
                const float c = cosf(ang), s = sinf(ang), sign = (d & 1) ? 1.0f : -1.0f;
                if (p < T) { tcos[(size_t)p * kDh + d] = half_round(c); tsin[(size_t)p * kDh + d] = half_round(s) * sign; }   // time tables pass through fp16 (:413-414)
                if (p < nb) { fcos[(size_t)p * kDh + d] = c; fsin[(size_t)p * kDh + d] = s * sign; }
            }
    }
    const size_t a_fwd = place(fwd.data(), fwd.size()), a_inv = place(inv.data(), inv.size()), a_ws = place(wsum.data(), wsum.size()),
                 a_tc = place(tcos.data(), tcos.size()), a_ts = place(tsin.data(), tsin.size()), a_fc = place(fcos.data(), fcos.size()),
                 a_fs = place(fsin.data(), fsin.size());

    // ---- int tables: gather columns, band offsets, bin -> owning gathered entries (CSR, ascending gathered index)
    std::vector<int> gcol((size_t)off[nb]);
    std::vector<int> band_of((size_t)nsel);
    for (int i = 0; i < nb; ++i)
        for (int c = off[i]; c < off[i + 1]; c += 2) band_of[c >> 1] = i;
    std::vector<int> count((size_t)kFc + 1, 0);
    for (int sidx = 0; sidx < nsel; ++sidx) {
        const int fc = (int)t_fi->data[sidx];
        if (fc < 0 || fc >= kFc || (float)fc != t_fi->data[sidx]) return bail(mfail(err, ADE_ERR_BAD_VALUE, "melband: freq_indices entries must be integers in [0, 2050)"));
        gcol[2 * sidx] = fc * 2; gcol[2 * sidx + 1] = fc * 2 + 1;
        ++count[fc + 1];
    }
    for (int fc = 0; fc < kFc; ++fc) count[fc + 1] += count[fc];
    std::vector<int> csr_col((size_t)nsel), csr_d((size_t)nsel), fill(count.begin(), count.end() - 1);
    for (int sidx = 0; sidx < nsel; ++sidx) {
        const int fc = (int)t_fi->data[sidx], band = band_of[sidx], at = fill[fc]++;
        csr_col[at] = off[band] + 2 * sidx;        // raw column of the GLU value (re) in YT: 2*off_band + (2*sidx - off_band)
        csr_d[at] = off[band + 1] - off[band];     // + d: the matching gate column
    }
    std::vector<int> ints;
    auto place_i = [&](const std::vector<int>& v) { const size_t at = ints.size(); ints.insert(ints.end(), v.begin(), v.end()); ints.resize((ints.size() + 15) & ~(size_t)15); return at; };
    const size_t i_gcol = place_i(gcol), i_off = place_i(off), i_cs = place_i(count), i_cc = place_i(csr_col), i_cd = place_i(csr_d);

    if (hipSetDevice(device) != hipSuccess) return bail(mfail(err, ADE_ERR_DEVICE, "hipSetDevice failed"));
    if (hipMalloc((void**)&e->d_w, arena * sizeof(float)) != hipSuccess) return bail(mfail(err, ADE_ERR_DEVICE, "hipMalloc of the Mel-Band-Roformer weights failed"));
    for (const Item& it : items)
        if (hipMemcpy(e->d_w + it.at, it.src, it.n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
            return bail(mfail(err, ADE_ERR_DEVICE, "upload of the Mel-Band-Roformer weights failed"));
    if (hipMalloc((void**)&e->d_i, ints.size() * sizeof(int)) != hipSuccess || hipMalloc((void**)&e->d_ll, ll.size() * sizeof(long long)) != hipSuccess ||
        hipMemcpy(e->d_i, ints.data(), ints.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(e->d_ll, ll.data(), ll.size() * sizeof(long long), hipMemcpyHostToDevice) != hipSuccess)
        return bail(mfail(err, ADE_ERR_DEVICE, "upload of the band tables failed"));
    if (bf16) {
        // bf16 copies (round to nearest even) of every Linear weight at the SAME arena offsets; the mask estimator's first two Linears are stored (in, out) by the
        // reference's fused buffers (me_w1t / me_w2t) and are transposed here to (out, in): csrc/ade_gemm16.h takes both operands with k contiguous.
        if (((dim | di | ffd | med | ldq) & 7) || (dim & 127) || dim > 128 * kSqTiles)
            return bail(mfail(err, ADE_ERR_UNSUPPORTED, "melband: ade_gemm_dtype = bf16 needs dim a multiple of 128 (at most 512) and dim_inner, heads, the FFN and mask-estimator widths multiples of 8"));
        auto to_bf16 = [](float x) { uint32_t u; memcpy(&u, &x, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); };
        std::vector<uint16_t> h16(arena, 0);
        auto conv = [&](const Item& it) { for (size_t i = 0; i < it.n; ++i) h16[it.at + i] = to_bf16(it.src[i]); };
        auto conv_t = [&](size_t at, const float* src, int batch, int K, int N) {      // [batch][K][N] -> [batch][N][K]
            for (int z = 0; z < batch; ++z)
                for (int k = 0; k < K; ++k)
                    for (int n = 0; n < N; ++n) h16[at + ((size_t)z * N + n) * K + k] = to_bf16(src[((size_t)z * K + k) * N + n]);
        };
        for (const Item& it : items)
            if (it.src != m1->data && it.src != m2->data && it.src != fwd.data() && it.src != inv.data()) conv(it);
        conv_t(a_m1, m1->data, nb, dim, med);
        conv_t(a_m2, m2->data, nb, med, med);
        if (hipMalloc((void**)&e->d_w16, arena * sizeof(uint16_t)) != hipSuccess ||
            hipMemcpy(e->d_w16, h16.data(), arena * sizeof(uint16_t), hipMemcpyHostToDevice) != hipSuccess)
            return bail(mfail(err, ADE_ERR_DEVICE, "upload of the bf16 Mel-Band-Roformer weights failed"));
        e->me_w1_16 = e->d_w16 + a_m1; e->me_w2_16 = e->d_w16 + a_m2;
    }
    e->k_fwd = e->d_w + a_fwd; e->k_inv = e->d_w + a_inv; e->wsum = e->d_w + a_ws;
    e->tcos = e->d_w + a_tc; e->tsin = e->d_w + a_ts; e->fcos = e->d_w + a_fc; e->fsin = e->d_w + a_fs;
    e->me_w1t = e->d_w + a_m1; e->me_b1 = e->d_w + a_mb1; e->me_w2t = e->d_w + a_m2; e->me_b2 = e->d_w + a_mb2;
    for (size_t k = 0; k < tf_at.size(); ++k) {
        const TfAt& a = tf_at[k];
        const gemm16::bf16_t* w16 = e->d_w16;     // null on the fp32 path
        TfW w{e->d_w + a.v[0], e->d_w + a.v[1], e->d_w + a.v[2], e->d_w + a.v[3], e->d_w + a.v[4], e->d_w + a.v[5], e->d_w + a.v[6], e->d_w + a.v[7],
              w16 ? w16 + a.v[0] : nullptr, w16 ? w16 + a.v[2] : nullptr, w16 ? w16 + a.v[3] : nullptr, w16 ? w16 + a.v[5] : nullptr};
        (k & 1 ? e->freq_tf : e->time_tf).push_back(w);
    }
    e->gcol = e->d_i + i_gcol; e->off = e->d_i + i_off; e->csr_start = e->d_i + i_cs; e->csr_col = e->d_i + i_cc; e->csr_d = e->d_i + i_cd;
    e->bt = BandTable{e->off, e->d_ll, e->d_ll + nb, e->d_ll + 2 * nb, e->d_ll + 3 * nb};
    *out = e;
    return ADE_OK;
}

int MelbandEngine::reserve(int batch, std::string& err) {
    if (batch <= capacity) return ADE_OK;
    MB_HIP(hipSetDevice(device));
    MB_HIP(hipDeviceSynchronize());
    if (ws) (void)hipFree(ws);
    ws = nullptr;
    capacity = 0;
    const size_t BT = (size_t)batch * n_win * T, R = (size_t)nb * BT;
    const size_t wide = (size_t)(3 * di + heads) > (size_t)med ? (size_t)(3 * di + heads) : (size_t)med;
    const size_t hid = (size_t)ffd > (size_t)med ? (size_t)ffd : (size_t)med;
    size_t sizes[10] = {(size_t)kFc * 2 * BT, R * dim, R, R * wide, R * hid, R * di, (size_t)2 * S2 * BT, (size_t)2 * kBinsM * kChan * BT,
                        (size_t)kChan * BT * kNfftM, (size_t)kFc * 2 * BT};
    if (bf16) {      // the three activation buffers hold bf16 views only (sized for fp32 they were twice what the path touches -- 9.5 GB of bufA alone at 32 x 8 s):
                     // q | k | v | gates (and the mask estimator's second hidden layer) + the stream copy Xg; the hidden activations (and, as fp32, the "tokens" tap: R dim floats);
                     // the attention output + the stream copy X1.  invn's place takes the row sums.
        sizes[2] = (size_t)(3 * kSqTiles + 3) * R;
        sizes[3] = (((R * wide + 63) & ~(size_t)63) + R * dim) / 2 + 64;
        sizes[4] = std::max((R * hid + 1) / 2 + 64, R * dim);
        sizes[5] = (((R * di + 63) & ~(size_t)63) + R * dim) / 2 + 64;
    }
    size_t total = 0;
    for (size_t s : sizes) total += (s + 63) & ~(size_t)63;
    MB_HIP(hipMalloc((void**)&ws, total * sizeof(float)));
    float** ptrs[10] = {&Sp, &X, &invn, &bufA, &bufB, &AO, &YT, &MS, &frames_buf, &mask_tap};
    size_t at = 0;
    for (int i = 0; i < 10; ++i) { *ptrs[i] = ws + at; at += (sizes[i] + 63) & ~(size_t)63; }
    if (bf16) {      // the bf16 activations live in the fp32 path's buffers (each at most half their size): q | k | v | gates in bufA with the stream copy Xg behind it,
                     // the hidden activations in bufB, the attention output in AO with the stream copy X1 behind it; the partial sums of squares in invn's place
        A16 = reinterpret_cast<gemm16::bf16_t*>(bufA); Xg = A16 + ((R * wide + 63) & ~(size_t)63);
        B16 = reinterpret_cast<gemm16::bf16_t*>(bufB);
        AO16 = reinterpret_cast<gemm16::bf16_t*>(AO); X1 = AO16 + ((R * di + 63) & ~(size_t)63);
        sq1 = invn; sq2 = sq1 + R * kSqTiles; sqg = sq2 + R * kSqTiles; inv1 = sqg + R * kSqTiles; inv2 = inv1 + R; invg = inv2 + R;
    }
    capacity = batch;
    return ADE_OK;
}

void MelbandEngine::transformer(hipStream_t s, const TfW& w, int R, int n, int nseq, long long seq_stride, long long pos_stride, const float* rc,
                                const float* rs) {
    using namespace gemm;
    const int ldq = 3 * di + heads;
    if (bf16) {      // on entry: Xg = bf16 operand copy of the stream, sqg its rows' partial sums of squares; X = x2 of the previous transformer (g_last, sq2) or the stream itself
        const int tiles = (dim + 127) / 128;
        const dim3 rows256((unsigned)((R + 255) / 256));
        gemm16::launch(s, Xg, dim, w.in_w16, dim, RotaryQkStore16{A16, w.in_b, rc, rs, ldq, 2 * di, gemm16::make_fastdiv((int)pos_stride), gemm16::make_fastdiv(n)}, R, ldq, dim, invg);   // (:547-548, :552)
        if (n > 64) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_attention16<2>), dim3((unsigned)nseq, (unsigned)heads, (unsigned)((n + 127) / 128)), dim3(256), 0, s,
                                       (const gemm16::bf16_t*)A16, AO16, n, seq_stride, pos_stride, ldq, di);                                            // (:549-560)
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_attention16<1>), dim3((unsigned)nseq, (unsigned)heads, 1), dim3(256), 0, s, (const gemm16::bf16_t*)A16, AO16, n,
                                seq_stride, pos_stride, ldq, di);
        gemm16::launch(s, AO16, di, w.out_w16, di, OutProjStore16{X, X1, g_last, inv2, sq1, dim}, R, dim, di);                                           // (:561, :569; the previous :571)
        hipLaunchKernelGGL(k_rows_inv_norm, rows256, dim3(256), 0, s, (const float*)sq1, tiles, inv1, (const float*)nullptr, (float*)nullptr, R);
        gemm16::launch(s, X1, dim, w.ff1_w16, dim, BiasActStore16<0>{B16, w.ff1_b, ffd}, R, ffd, dim, inv1);                                             // (:564)
        gemm16::launch(s, B16, ffd, w.ff2_w16, ffd, FfOutStore16{X, Xg, w.ff2_b, w.out_g, sq2, sqg, dim}, R, dim, ffd);                                  // (:565, :570; :571 is applied by the consumers)
        hipLaunchKernelGGL(k_rows_inv_norm, rows256, dim3(256), 0, s, (const float*)sq2, tiles, inv2, (const float*)sqg, invg, R);
        g_last = w.out_g;
        return;
    }
    // invn holds 1 / |x_row| on entry (written by whoever produced X)
    launch(s, RowMajorA{X, dim}, WeightNK{w.in_w, dim}, RotaryQkStore{bufA, invn, w.in_b, rc, rs, ldq, 2 * di, (int)pos_stride, n}, R, ldq, dim);   // (:547-548, :552)
    const dim3 g2((unsigned)nseq, (unsigned)heads, (unsigned)((n + 127) / 128)), g1((unsigned)nseq, (unsigned)heads, 1);            // (:549-560)
    if (n > 64) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_attention<2>), g2, dim3(256), 0, s, (const float*)bufA, AO, n, seq_stride, pos_stride, ldq, di);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_attention<1>), g1, dim3(256), 0, s, (const float*)bufA, AO, n, seq_stride, pos_stride, ldq, di);
    launch(s, RowMajorA{AO, di}, WeightNK{w.out_w, di}, ResidualStore{X, nullptr, dim}, R, dim, di);                               // (:561, :569)
    hipLaunchKernelGGL(k_row_invnorm, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, s, (const float*)X, invn, R, dim);
    launch(s, RowMajorA{X, dim}, WeightNK{w.ff1_w, dim}, ScaleBiasGeluStore{bufB, invn, w.ff1_b, ffd}, R, ffd, dim);               // (:564)
    launch(s, RowMajorA{bufB, ffd}, WeightNK{w.ff2_w, ffd}, ResidualStore{X, w.ff2_b, dim}, R, dim, ffd);                          // (:565, :570)
    hipLaunchKernelGGL(k_row_normalize_gain, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, s, X, w.out_g, invn, R, dim);            // (:571)
}

int MelbandEngine::run(hipStream_t s, const int16_t* d_in, int batch, int16_t* d_out, float* d_f32, std::string& err) {
    if (batch == 0) return ADE_OK;
    int st = reserve(batch, err);
    if (st != ADE_OK) return st;
    using namespace gemm;
    const int B = batch * n_win, BT = B * T, R = nb * BT, J = BT * kChan;      // every fold window is an independent stereo clip (:588-594)
    // STFT of every (clip, channel) row into channel-interleaved bins                                                              (:648-651, :596)
    if (float_in) launch(s, RowMajorA{k_fwd, kNfftM}, StereoFrameB<float>{float_in, L, T, n_win}, BinStore{Sp, T, BT}, 2 * kBinsM, J, kNfftM);
    else launch(s, RowMajorA{k_fwd, kNfftM}, StereoFrameB<int16_t>{d_in, L, T, n_win}, BinStore{Sp, T, BT}, 2 * kBinsM, J, kNfftM);
    // band split                                                                                                                   (:597-599)
    hipLaunchKernelGGL(k_band_invnorm, dim3((unsigned)((BT + 255) / 256), (unsigned)nb), dim3(256), 0, s, (const float*)Sp, gcol, off, invn, BT);
    launch_batched(s, BandSplitProb{Sp, gcol, d_w, bt, invn, X, BT, dim}, nb, BT, dim);     // fp32 on both paths (the front)
    if (bf16) { hipLaunchKernelGGL(k_row_prep16, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, s, (const float*)X, Xg, invg, R, dim); g_last = nullptr; }
    else hipLaunchKernelGGL(k_row_invnorm, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, s, (const float*)X, invn, R, dim);
    // axial transformers: time = T consecutive rows per (band, clip); frequency = nb rows B*T apart per (clip, frame)              (:609-614)
    for (int i = 0; i < depth; ++i) {
        transformer(s, time_tf[i], R, T, nb * B, (long long)T, 1LL, tcos, tsin);
        transformer(s, freq_tf[i], R, nb, BT, 1LL, (long long)BT, fcos, fsin);
    }
    // mask estimator: per band 384 -> 1536 -> 1536 (tanh) -> 2 d_i, kept raw and column-major for the GLU / scatter kernel         (:579-585)
    if (bf16) {
        gemm16::launch_batched(s, MeHiddenProb16{Xg, me_w1_16, me_b1, inv2, B16, BT, dim, med}, nb, BT, med);      // its input is the stream n(x2) g = Xg / |x2|
        gemm16::launch_batched(s, MeHiddenProb16{B16, me_w2_16, me_b2, nullptr, A16, BT, med, med}, nb, BT, med);
        gemm16::launch_batched(s, MeOutProb16{A16, d_w16, d_w, bt, YT, BT, med}, nb, 2 * max_d, BT);
    } else {
        launch_batched(s, MeHiddenProb{X, me_w1t, me_b1, bufB, BT, dim, med}, nb, BT, med);
        launch_batched(s, MeHiddenProb{bufB, me_w2t, me_b2, bufA, BT, med, med}, nb, BT, med);
        launch_batched(s, MeOutProb{bufA, d_w, bt, YT, BT, med}, nb, BT, 2 * max_d);
    }
    hipLaunchKernelGGL(k_mask_apply, dim3((unsigned)((BT + 255) / 256), (unsigned)kFc), dim3(256), 0, s, (const float*)Sp, (const float*)YT, csr_start, csr_col,
                       csr_d, MS, mask_tap, B, T);                                                                                  // (:616-624)
    // synthesis GEMM + overlap-add + PCM tail                                                                                      (:661, :667-676)
    launch(s, PlanarSpecA{MS, J}, RowMajorB{k_inv, kNfftM}, BiasActStore<kActNone>{frames_buf, kNfftM, nullptr, 0.0f}, J, kNfftM, 2 * kBinsM);
    const long long total = (long long)B * kChan * Lo;
    hipLaunchKernelGGL(k_melband_ola_pcm, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const float*)frames_buf, wsum, d_out, d_f32, T, Lo, n_win, total);
    MB_HIP(hipGetLastError());
    return ADE_OK;
}

int MelbandEngine::tap(hipStream_t s, const char* name, int batch, float* out, size_t count, size_t* written, std::string& err) {
    const size_t BT = (size_t)batch * n_win * T;
    const float* src = nullptr;
    size_t n = 0;
    if (strcmp(name, "tokens") == 0) {                                                     // transformer output (band, clip, frame, dim)
        src = X; n = (size_t)nb * BT * dim;
        if (bf16 && g_last && batch > 0 && X) {      // the bf16 path keeps x2; the stream value x2 * g / |x2| (see the stores) is formed in the hidden-activation buffer, free after a run
            float* y = bufB;
            hipLaunchKernelGGL(k_row_apply_gain, dim3((unsigned)((n / dim + 3) / 4)), dim3(256), 0, s, (const float*)X, g_last, (const float*)inv2, y, (int)(n / dim), dim);
            src = y;
        }
    }
    else if (strcmp(name, "mask") == 0) { src = mask_tap; n = (size_t)kFc * 2 * BT; }     // averaged complex mask [fc][re|im][clip*T + t]
    else if (strcmp(name, "spec") == 0) { src = Sp; n = (size_t)kFc * 2 * BT; }
    else return mfail(err, ADE_ERR_NOT_FOUND, std::string("unknown tap: ") + name);
    if (!src || batch <= 0) return mfail(err, ADE_ERR_NOT_FOUND, "tap has no data yet");
    if (count < n) return mfail(err, ADE_ERR_SHAPE_MISMATCH, "tap buffer too small");
    MB_HIP(hipStreamSynchronize(s));
    MB_HIP(hipMemcpy(out, src, n * sizeof(float), hipMemcpyDeviceToHost));
    *written = n;
    return ADE_OK;
}

}  // namespace ade
