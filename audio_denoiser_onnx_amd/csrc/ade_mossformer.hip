// ade_mossformer.hip — MossFormer2-SS-16K (two-speaker separation, 16 kHz) on the MI355X: SURVEY.md section 8 row a18.
//
// Reference: MOSSFORMER_SS.forward / _run_mdl / norm_audio / group_norm_static over the FUSED buffers its constructor registers
// (MossFormer2_SS_16K/Export_MossFormer2_SS_16K.py:398-662; buffers :130-395):
//   int16 window -> two-stage RMS normalisation -> Conv1d(1, 512, k 16, s 8) + ReLU -> window norm + 1x1 conv + sinusoid positions
//   -> L x [ FLASH block: token shift, ScaleNorm, fused Linear(512 -> 2048 | 128) + SiLU + depthwise k 17, OffsetScale x 4 + rotary,
//            quadratic ReLU^2 attention inside groups of 256 tokens + global linear attention, u/v gating, ScaleNorm, Linear + SiLU +
//            depthwise, residual ;
//            gated FSMN block: 1x1 + PReLU, LayerNorm, fused u|v Linear + SiLU + depthwise, Linear-ReLU-Linear, two dilated dense
//            memory convolutions (39 taps, dilation 1 / 2) each with InstanceNorm + PReLU, gate, LayerNorm, 1x1, residual ]
//   -> LayerNorm, window norm + affine, skip -> PReLU -> per speaker tanh x sigmoid gate, 1x1 + ReLU, x encoder output
//   -> ConvTranspose1d(512, 1, k 16, s 8) -> per (window, speaker) RMS restore -> int32 truncate, clamp -> int16.
// Tokens are rows: every activation is (row, channels) row-major with row = window * frames + t, so all 1x1 convolutions, Linears
// and both attention products are the functor GEMM of csrc/ade_gemm.h (exact fp32 on the matrix cores), with
//   * ScaleNorm / window-norm folded into GEMM stores (v * inv_norm[row] + bias; v * rstd - rstd * mean * rowsum(W) + bias),
//   * the token shift, PCM framing + normalisation, zero padding of attention groups, PReLU and the speaker gate as operand loaders,
//   * SiLU / ReLU / ReLU^2 / residual adds / the encoder-mask product as stores,
//   * the per-group attention products and the per-window linear attention as batched launches (blockIdx.z); the quadratic and the
//     linear attention outputs come out of ONE contraction over [256 keys of the group | 128 linear-attention channels].
// What is not a matrix product -- depthwise convolutions over time, row norms, per-channel instance norms, the RMS stages --
// are small bandwidth-bound kernels below.
#include "ade_gemm.h"
#include "ade_internal.h"
#include "../../include/ade.h"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace ade {

namespace {

using namespace dev;

constexpr int kDim = 512, kHalf = 256, kVu = 1024, kVu2 = 2048, kQk = 128, kIn = kVu2 + kQk /* 2176 */, kInner = 256;
constexpr int kEncK = 16, kEncS = 8, kDw = 17, kMemK = 39, kSpk = 2;
enum Hyper { hNormFactor, hGroup, hRotDim, hDwPad, hFlNormEps, hFlOutNormEps, hFrontEps, hMmEps, hIntraEps, hFsLnEps, hFsN1Eps, hFsN2Eps,
             hMemDepth, hTailAlpha, hMemNormEps, hLorder, hCount };

// (round 5) on the hardware exp2 / rcp (~1 ulp each): `x / (1 + expf(-x))` is 24 VALU instructions per element (libm range reduction + the IEEE division sequence) against 5, and the
// in-projection's store applies it to 1.1 G elements per layer
__device__ __forceinline__ float sigm(float x) { return dev::fast_rcp(1.0f + __builtin_amdgcn_exp2f(-dev::kLog2e * x)); }
__device__ __forceinline__ float silu(float x) { return x * sigm(x); }
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// block-wide sum of one float per thread (blockDim.x <= 1024); every thread gets the result
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float s = 0.0f;
    for (int i = 0; i < nw; ++i) s += red[i];
    return s;
}

// ---- RMS stages --------------------------------------------------------------------------------------------------------------
// norm_audio (:403-423): one workgroup per window.  gains[b] = {scalar, scalarx}; rms_in[b] = rms * g * 1 / (g + eps) * 32767.
__global__ __launch_bounds__(1024) void k_norm_audio(const int16_t* __restrict__ pcm, const float* __restrict__ fpcm, int W, float norm_factor,
                                                     float2* __restrict__ gains, float* __restrict__ rms_in) {
    __shared__ float red[16];
    const int16_t* x = pcm + (size_t)blockIdx.x * W;
    const float* xf = fpcm ? fpcm + (size_t)blockIdx.x * W : nullptr;       // resampled input (floats in int16 units) replaces pcm when set
    const float eps = 1e-6f;
    float s = 0.0f;
    for (int i = threadIdx.x; i < W; i += blockDim.x) { const float v = (xf ? xf[i] : (float)x[i]) * (1.0f / 32768.0f); s += v * v; }
    const float avg = block_sum(s, red) / (float)W;
    float hs = 0.0f, hc = 0.0f;
    for (int i = threadIdx.x; i < W; i += blockDim.x) {
        const float v = (xf ? xf[i] : (float)x[i]) * (1.0f / 32768.0f), p = v * v;
        if (p > avg) { hs += p; hc += 1.0f; }
    }
    hs = block_sum(hs, red);
    hc = block_sum(hc, red);
    if (threadIdx.x == 0) {
        const float rms = sqrtf(avg), scalar = norm_factor / (rms + eps);
        const float high = sqrtf(hs / fmaxf(hc, 1.0f)), scalarx = norm_factor / (high * scalar + eps);
        const float gp = scalar * scalarx, undo = 1.0f / (gp + eps);
        gains[blockIdx.x] = make_float2(scalar, scalarx);
        rms_in[blockIdx.x] = rms * gp * undo * 32767.0f;
    }
}

// mean / rstd of one window's (frames x 512) block (group_norm_static :398-401): kWsSlices workgroups per window each reduce a slice
// to (count, mean, M2) in two passes; k_window_stats_final merges the slices in a fixed order (Chan's update): deterministic, and a
// single window still fills 32 CUs instead of one.
constexpr int kWsSlices = 32;
// split-K factor of the linear-attention key / value product: FIXED, so that a window's result does not depend on how many other windows
// share the call (16 output tiles per window x 8 = 128 workgroups for a single window; at 32 windows the partial sums cost ~1 %)
constexpr int kLkvSplits = 8;
__global__ __launch_bounds__(1024) void k_window_stats_partial(const float* __restrict__ x, long long count, float4* __restrict__ part) {
    __shared__ float red[16];
    const long long per = (count + kWsSlices - 1) / kWsSlices, lo = (long long)blockIdx.y * per, hi = lo + per < count ? lo + per : count;
    const float* p = x + (size_t)blockIdx.x * count;
    float s = 0.0f;
    for (long long i = lo + threadIdx.x; i < hi; i += blockDim.x) s += p[i];
    const float nn = (float)(hi > lo ? hi - lo : 0);
    const float mean = block_sum(s, red) / fmaxf(nn, 1.0f);
    float q = 0.0f;
    for (long long i = lo + threadIdx.x; i < hi; i += blockDim.x) { const float d = p[i] - mean; q += d * d; }
    q = block_sum(q, red);
    if (threadIdx.x == 0) part[(size_t)blockIdx.x * kWsSlices + blockIdx.y] = make_float4(nn, mean, q, 0.0f);
}
__global__ __launch_bounds__(64) void k_window_stats_final(const float4* __restrict__ part, float eps, float2* __restrict__ stats, int B) {
    const int b = (int)blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    float na = 0.0f, mean = 0.0f, M2 = 0.0f;
    for (int i = 0; i < kWsSlices; ++i) {
        const float4 p = part[(size_t)b * kWsSlices + i];
        if (p.x <= 0.0f) continue;
        const float nb = p.x, delta = p.y - mean, nn = na + nb;
        mean += delta * (nb / nn);
        M2 += p.z + delta * delta * (na * nb / nn);
        na = nn;
    }
    stats[b] = make_float2(mean, 1.0f / sqrtf(M2 / fmaxf(na, 1.0f) + eps));
}

// ---- operand / store functors ------------------------------------------------------------------------------------------------
struct EncFrameA {             // A((b, t), k) = normalised sample 8 t + k of window b (:579-582); the two gains apply in the reference's order
    static constexpr bool kAlongK = true;
    const int16_t* pcm;
    const float* fpcm;
    const float2* gains;
    int W, n;
    __device__ float operator()(int m, int k) const {
        const int b = m / n, t = m - b * n;
        const float2 g = gains[b];
        const size_t at = (size_t)b * W + kEncS * t + k;
        return (((fpcm ? fpcm[at] : (float)pcm[at]) * (1.0f / 32768.0f)) * g.x) * g.y;
    }
};
template <int ACT>             // 0 none, 1 relu, 2 silu, 3 leaky(alpha)
struct BiasActRowStore {       // out[m * ld + n] = act(v + bias[n])
    static constexpr bool kCtx = true;
    float* out;
    const float* bias;         // may be null
    int ld;
    float alpha;
    __device__ gemm::None row(int) const { return gemm::None{}; }
    __device__ float col(int n) const { return bias ? bias[n] : 0.0f; }
    __device__ gemm::None pre(int, int, gemm::None) const { return gemm::None{}; }
    __device__ void operator()(int m, int n, float v, gemm::None, float b, gemm::None) const {
        v += b;
        if (ACT == 1) v = fmaxf(v, 0.0f);
        if (ACT == 2) v = silu(v);
        if (ACT == 3) v = v >= 0.0f ? v : v * alpha;
        out[(size_t)m * ld + n] = v;
    }
    // float4 form (gemm::HasV4: transposed tiles, 16 float4 stores per lane and tile instead of 64 scalar ones; the same arithmetic per element)
    static constexpr bool kV4 = true;
    __host__ __device__ bool can_v4(int N) const { return !((N | ld) & 3) && !(((size_t)out | (size_t)bias) & 15); }
    __device__ float4 col4(int n) const { return bias ? *reinterpret_cast<const float4*>(bias + n) : make_float4(0.0f, 0.0f, 0.0f, 0.0f); }
    __device__ gemm::None pre4(int, int, gemm::None) const { return gemm::None{}; }
    __device__ float act1(float v) const {
        if (ACT == 1) v = fmaxf(v, 0.0f);
        if (ACT == 2) v = silu(v);
        if (ACT == 3) v = v >= 0.0f ? v : v * alpha;
        return v;
    }
    __device__ void store4(int m, int n, float4 v, gemm::None, const float4& b, gemm::None) const {
        *reinterpret_cast<float4*>(out + (size_t)m * ld + n) = make_float4(act1(v.x + b.x), act1(v.y + b.y), act1(v.z + b.z), act1(v.w + b.w));
    }
};
struct FrontStore {            // window norm folded: H = MI = v * rstd - rstd * mean * rowsum(W)[n] + b[n] + emb_pos[n][t]   (:586-591)
    float *h, *mi;
    const float2* stats;
    const float *wsum, *bias, *emb;     // emb: (512, n_frames)
    int n;
    __device__ void operator()(int m, int c, float v) const {
        const int b = m / n, t = m - b * n;
        const float2 st = stats[b];
        const float y = (v * st.y - st.y * st.x * wsum[c] + bias[c]) + emb[(size_t)c * n + t];
        h[(size_t)m * kDim + c] = y;
        mi[(size_t)m * kDim + c] = y;
    }
};
struct ShiftA {                // token shift (:454-456): first half of the channels comes from the previous frame of the same window
    static constexpr bool kAlongK = true;
    const float* h;
    int n;
    __device__ float operator()(int m, int k) const {
        if (k >= kHalf) return h[(size_t)m * kDim + k];
        return (m % n) ? h[(size_t)(m - 1) * kDim + k] : 0.0f;
    }
    __device__ bool can_vec4(int) const { return true; }
    __device__ float4 vec4(int m, int k) const {          // branch-free: one load from a valid row, zeroed for the first frame's shifted half
        const bool shifted = k < kHalf, first = (m % n) == 0;
        return ld4_or_zero(!(shifted && first), h + (size_t)(shifted && !first ? m - 1 : m) * kDim + k);      // (zeros by address: a prefetch stays in flight)
    }
};
struct ScaleSiluStore {        // silu(v * inv[m] + bias[n])   (ScaleNorm folded, :458-459, :502-503)
    static constexpr bool kCtx = true;
    float* out;
    const float* inv;
    const float* bias;
    int ld;
    __device__ float row(int m) const { return inv[m]; }
    __device__ float col(int n) const { return bias[n]; }
    __device__ gemm::None pre(int, int, float) const { return gemm::None{}; }
    __device__ void operator()(int m, int n, float v, float sc, float b, gemm::None) const { out[(size_t)m * ld + n] = silu(v * sc + b); }
    static constexpr bool kV4 = true;
    __host__ __device__ bool can_v4(int N) const { return !((N | ld) & 3) && !(((size_t)out | (size_t)bias) & 15); }
    __device__ float4 col4(int n) const { return *reinterpret_cast<const float4*>(bias + n); }
    __device__ gemm::None pre4(int, int, float) const { return gemm::None{}; }
    __device__ void store4(int m, int n, float4 v, float sc, const float4& b, gemm::None) const {
        *reinterpret_cast<float4*>(out + (size_t)m * ld + n) = make_float4(silu(v.x * sc + b.x), silu(v.y * sc + b.y), silu(v.z * sc + b.z), silu(v.w * sc + b.w));
    }
};
struct Relu2Store {            // attn = relu(q k^T)^2   (:487-488)
    float* out;
    int ld;
    __device__ void operator()(int m, int n, float v) const { const float r = fmaxf(v, 0.0f); out[(size_t)m * ld + n] = r * r; }
};
struct PaddedRowsB {           // B(k, n) = rows[(row0 + k) * ld + n] for k < valid, else 0: the zero-padded tail of the last attention group (:480-484)
    static constexpr bool kAlongN = true;
    const float* rows;
    int ld, valid;
    __device__ float operator()(int k, int n) const { return k < valid ? rows[(size_t)k * ld + n] : 0.0f; }
};
struct ColMajorA {             // A(m, k) = p[k * ld + m]: lin_k^T (:490), consecutive m contiguous
    static constexpr bool kAlongK = false;
    const float* p;
    int ld;
    __device__ float operator()(int m, int k) const { return p[(size_t)k * ld + m]; }
};
struct GuardedStore {          // out[m * ld + n] (=|+=) v for m < valid
    static constexpr bool kCtx = true;
    float* out;
    int ld, valid, accumulate;
    __device__ gemm::None row(int) const { return gemm::None{}; }
    __device__ gemm::None col(int) const { return gemm::None{}; }
    __device__ float pre(int m, int n, gemm::None) const { return accumulate ? out[(size_t)(m < valid ? m : 0) * ld + n] : 0.0f; }
    __device__ void operator()(int m, int n, float v, gemm::None, gemm::None, float old) const {
        if (m < valid) out[(size_t)m * ld + n] = old + v;
    }
};
struct ResidualBiasStore {     // x[m][n] += v + bias[n]   (:541)
    static constexpr bool kCtx = true;
    float* x;
    const float* bias;
    int ld;
    __device__ gemm::None row(int) const { return gemm::None{}; }
    __device__ float col(int n) const { return bias[n]; }
    __device__ float pre(int m, int n, gemm::None) const { return x[(size_t)m * ld + n]; }
    __device__ void operator()(int m, int n, float v, gemm::None, float b, float old) const { x[(size_t)m * ld + n] = old + (v + b); }
    static constexpr bool kV4 = true;
    __host__ __device__ bool can_v4(int N) const { return !((N | ld) & 3) && !(((size_t)x | (size_t)bias) & 15); }
    __device__ float4 col4(int n) const { return *reinterpret_cast<const float4*>(bias + n); }
    __device__ float4 pre4(int m, int n, gemm::None) const { return *reinterpret_cast<const float4*>(x + (size_t)m * ld + n); }
    __device__ void store4(int m, int n, float4 v, gemm::None, const float4& b, const float4& old) const {
        *reinterpret_cast<float4*>(x + (size_t)m * ld + n) = make_float4(old.x + (v.x + b.x), old.y + (v.y + b.y), old.z + (v.z + b.z), old.w + (v.w + b.w));
    }
};
struct LeakyA {                // A(m, k) = leaky_relu(x[m][k], alpha)   (:599)
    static constexpr bool kAlongK = true;
    const float* x;
    int ld;
    float alpha;
    __device__ float operator()(int m, int k) const { const float v = x[(size_t)m * ld + k]; return v >= 0.0f ? v : v * alpha; }
    __device__ bool can_vec4(int) const { return true; }
    __device__ float4 vec4(int m, int k) const {
        float4 v = *reinterpret_cast<const float4*>(x + (size_t)m * ld + k);
        v.x = v.x >= 0.0f ? v.x : v.x * alpha; v.y = v.y >= 0.0f ? v.y : v.y * alpha;
        v.z = v.z >= 0.0f ? v.z : v.z * alpha; v.w = v.w >= 0.0f ? v.w : v.w * alpha;
        return v;
    }
};
struct SpeakerGateA {          // A((spk, row), k) = tanh(gp[row][spk*1024 + k]) * sigmoid(gp[row][spk*1024 + 512 + k])   (:601-605)
    static constexpr bool kAlongK = true;
    const float* gp;
    int R;
    __device__ float operator()(int m, int k) const {
        const int spk = m / R, row = m - spk * R;
        const float* p = gp + (size_t)row * (kSpk * 2 * kDim) + spk * 2 * kDim;
        return tanhf(p[k]) * sigm(p[kDim + k]);
    }
};
struct MaskEncStore {          // sep[(spk, row)][c] = relu(v) * x_enc[row][c]   (:606-611)
    float* sep;
    const float* xe;
    int R;
    __device__ void operator()(int m, int c, float v) const {
        const int row = m % R;
        sep[(size_t)m * kDim + c] = fmaxf(v, 0.0f) * xe[(size_t)row * kDim + c];
    }
};

template <class A, class B, class S>
struct Prob { A a; B b; S st; int M, N, K; };
struct QuadScoreProb {         // z = (window, group): ATT_z = relu(quad_q quad_k^T)^2, 256 x 256 x 128
    const float *qq, *qk;
    float* att;
    int g;
    __device__ Prob<gemm::RowMajorA, gemm::WeightNK, Relu2Store> operator()(int z) const {
        const size_t o = (size_t)z * g * kQk;
        return {gemm::RowMajorA{qq + o, kQk}, gemm::WeightNK{qk + o, kQk}, Relu2Store{att + (size_t)z * g * g, g}, g, g, kQk};
    }
};
struct AttLinA {               // A(m, k) = k < g ? ATT_z[m][k] : lin_q[row m][k - g]: the quadratic and the linear attention share one contraction
    static constexpr bool kAlongK = true;
    const float *att, *lq;
    int g;
    __device__ float operator()(int m, int k) const { return k < g ? att[(size_t)m * g + k] : lq[(size_t)m * kQk + (k - g)]; }
    __device__ bool can_vec4(int) const { return (g & 3) == 0; }
    __device__ float4 vec4(int m, int k) const {          // one load through a selected pointer (no branch around the load)
        const float* p = k < g ? att + (size_t)m * g + k : lq + (size_t)m * kQk + (k - g);
        return *reinterpret_cast<const float4*>(p);
    }
};
struct VuLkvB {                // B(k, n) = k < g ? value row k of the group (zero beyond the window's frames, :480-484) : LKV[k - g][n]
    static constexpr bool kAlongN = true;
    const float *rows, *lkv;
    int ld, valid, g;
    __device__ float operator()(int k, int n) const {     // one LOAD through a selected pointer (no branch around it, no select on its value: the padded rows are zeroed by keep())
        const bool lin = k >= g;
        const float* p = lin ? lkv + (size_t)(k - g) * kVu2 + n : rows + (size_t)(k < valid ? k : 0) * ld + n;
        return *p;
    }
    __device__ bool keep(int k) const { return k >= g || k < valid; }      // gemm::HasKeep: applied when the slab goes to LDS
};
struct LinKvProb {             // z = (window, split): partial LKV (128 x 2048) = lin_k^T x value rows over the split's frames   (:490-492; padded keys are
    const float *lk, *vu;      // zero rows, so K = frames).  A single window has only 16 output tiles: splitting its long contraction over
    float* lkv;                // `splits` workgroups per tile keeps the chip busy; k_lkv_reduce adds the partial sums in split order.
    int n, padded, splits;
    __device__ Prob<ColMajorA, gemm::RowMajorB, GuardedStore> operator()(int z) const {
        const int b = z / splits, sp = z - b * splits, per = ((n + splits - 1) / splits + 15) & ~15, k0 = sp * per, kn = max(0, min(per, n - k0));
        return {ColMajorA{lk + ((size_t)b * padded + k0) * kQk, kQk}, gemm::RowMajorB{vu + ((size_t)b * n + k0) * kIn, kIn},
                GuardedStore{lkv + (size_t)z * kQk * kVu2, kVu2, kQk, 0}, kQk, kVu2, kn};
    }
};
struct AttOutProb {            // z = (window, group): AO rows of the group = ATT_z x value rows + lin_q x LKV_window   (:488, :495-497), K = g + 128
    const float *att, *vu, *lq, *lkv;
    float* ao;
    int g, groups, n, padded;
    __device__ Prob<AttLinA, VuLkvB, GuardedStore> operator()(int z) const {
        const int b = z / groups, gi = z - b * groups, valid = min(g, n - gi * g);
        const size_t row0 = (size_t)b * n + (size_t)gi * g;
        return {AttLinA{att + (size_t)z * g * g, lq + ((size_t)b * padded + (size_t)gi * g) * kQk, g},
                VuLkvB{vu + row0 * kIn, lkv + (size_t)b * kQk * kVu2, kIn, valid, g}, GuardedStore{ao + row0 * kVu2, kVu2, valid, 0}, g, kVu2, g + kQk};
    }
};

// LKV[b] = sum over splits of the partial products, in split order (deterministic)
__global__ __launch_bounds__(256) void k_lkv_reduce(const float* __restrict__ part, float* __restrict__ lkv, int splits, long long per_window, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const long long b = i / per_window, e = i - b * per_window;
    float s = 0.0f;
    for (int sp = 0; sp < splits; ++sp) s += part[((size_t)b * splits + sp) * per_window + e];
    lkv[i] = s;
}

// DYNAMIC_AXES export (:183, :430, :500-501): 1 / frames is not folded into the linear-key OffsetScale row; the reduced 128 x 2048 product is scaled once at run time
__global__ __launch_bounds__(256) void k_lkv_scale(float* __restrict__ lkv, float scale, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < total) lkv[i] = lkv[i] * scale;
}

// ---- row-wise and time-wise kernels ------------------------------------------------------------------------------------------
// inv[m] = 1 / max(|token-shifted row m|, eps)   (:457)
__global__ __launch_bounds__(256) void k_shift_invnorm(const float* __restrict__ h, float* __restrict__ inv, int rows, int n, float eps) {
    const int m = (int)blockIdx.x * 4 + ((int)threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (m >= rows) return;
    const bool first = (m % n) == 0;
    float s = 0.0f;
    for (int k = lane; k < kDim; k += 64) {
        const float v = k >= kHalf ? h[(size_t)m * kDim + k] : (first ? 0.0f : h[(size_t)(m - 1) * kDim + k]);
        s += v * v;
    }
    s = wave_sum(s);
    if (lane == 0) inv[m] = 1.0f / fmaxf(sqrtf(s), eps);
}

// Depthwise / grouped convolution over time, register form (:460, :504-505, :516 with TAPS = 17; the dilated dense memory convolutions :521-527 with
// TAPS = 39, CIN = j + 1 input channels per group).
//   out = (res ? res : 0) + (CENTER ? x : 0) + sum_{c < CIN} sum_k w[g][c][k] * in[g * CIN + c][t + k dil - pad]
// The input channels of group g are g * CIN + c of the concatenation [src0 (split channels) | src1] (:534-535).
// A lane owns one output channel (64 consecutive channels per wavefront: 256-byte rows) and one run of `kTconvRun` outputs of one dilation phase
// (t = phase + dil (j0 + i)); it streams the run's input rows from global memory once and keeps TAPS output accumulators in a register ring
// (acc[(row - k) mod TAPS] += w[k] x[row]; after its last tap an accumulator is stored and recycled), so the unrolled body has static register
// indices, no LDS and no barrier, and HBM / L2 sees each input (1 + (TAPS - 1) / run) times instead of TAPS times.  grid = (channels / 64,
// ceil(runs dil / 4), windows).
// With `partial` set every lane also writes (count, mean, M2) of its run for the instance norm that follows (:528-531); k_chan_finalize merges the runs
// in a fixed order (Chan's update), so the statistics are deterministic and independent of the batch size.
constexpr int kTconvRun = 128;
static_assert(kIn % 64 == 0 && kDim % 64 == 0 && kInner % 128 == 0, "k_tconv: a wavefront's 64 * CIN input channels must lie in one source tensor");
struct TconvRun {              // per-lane state of one run
    float* y;
    size_t base;
    int ldy, n, dil, pad, phase, j0, count, total, c;
    float shift, s1, s2;
};
// rows row0 + U .. row0 + TAPS - 1 of the ring (compile-time recursion instead of a 39 x 39 unrolled loop nest: every accumulator index is a constant);
// xs / rs are the rows' inputs and residuals, loaded beforehand in one batch so that TAPS * CIN loads are in flight per lane
template <int TAPS, int CIN, int U>
__device__ __forceinline__ void tconv_rows(float (&acc)[TAPS], const float (&wreg)[CIN][TAPS], const float (&xs)[TAPS][CIN], const float (&rs)[TAPS], TconvRun& q, int row0) {
#pragma unroll
    for (int k = 0; k < TAPS; ++k)
#pragma unroll
        for (int i = 0; i < CIN; ++i) acc[(U - k + TAPS) % TAPS] += wreg[i][k] * xs[U][i];
    const int done = row0 + U - (TAPS - 1);                                  // this output has now seen all its taps
    if (done >= 0 && done < q.count) {
        const float v = acc[(U + 1) % TAPS];
        q.y[(q.base + q.phase + (size_t)q.dil * (q.j0 + done)) * q.ldy + q.c] = rs[U] + v;
        if (done == 0) q.shift = v;
        const float d = v - q.shift;
        q.s1 += d;
        q.s2 += d * d;
    }
    acc[(U + 1) % TAPS] = 0.0f;
    if constexpr (U + 1 < TAPS) tconv_rows<TAPS, CIN, U + 1>(acc, wreg, xs, rs, q, row0);
}

template <int TAPS, int CIN, bool CENTER, bool RES>
__global__ __launch_bounds__(256) void k_tconv(const float* __restrict__ src0, const float* __restrict__ src1, int ld0, int ld1, int split,
                                               const float* __restrict__ w, const float* __restrict__ res, float* __restrict__ y, int ldy, int n, int dil,
                                               int pad, int runs, float4* __restrict__ partial) {
    const int c = (int)blockIdx.x * 64 + (threadIdx.x & 63), b = blockIdx.z, C = (int)gridDim.x * 64;
    const int slot = (int)blockIdx.y * 4 + (threadIdx.x >> 6), tiles = runs * dil;
    if (slot >= tiles) return;
    const int r = slot / dil, phase = slot - r * dil, j0 = r * kTconvRun;
    const int nsub = phase < n ? (n - phase + dil - 1) / dil : 0;           // outputs of this dilation phase
    const int count = nsub - j0 < kTconvRun ? nsub - j0 : kTconvRun;
    if (count <= 0) {
        if (partial) partial[((size_t)b * tiles + slot) * C + c] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        return;
    }
    float wreg[CIN][TAPS];
#pragma unroll
    for (int i = 0; i < CIN; ++i)
#pragma unroll
        for (int k = 0; k < TAPS; ++k) wreg[i][k] = w[((size_t)c * CIN + i) * TAPS + k];
    // the wavefront's 64 * CIN input channels lie in one source (split is a multiple of 64 * CIN): uniform row pointers + one lane offset
    const int ch0 = (int)blockIdx.x * 64 * CIN, lane_off = (threadIdx.x & 63) * CIN;
    const float* in = ch0 < split ? src0 + ch0 : src1 + (ch0 - split);
    const int ld = ch0 < split ? ld0 : ld1;
    if (CENTER) wreg[0][(TAPS - 1) / 2] += 1.0f;                             // + x[t]: the centre tap (dil = 1, pad = (TAPS - 1) / 2)
    float acc[TAPS];
#pragma unroll
    for (int k = 0; k < TAPS; ++k) acc[k] = 0.0f;
    TconvRun q{y, (size_t)b * n, ldy, n, dil, pad, phase, j0, count, count + TAPS - 1, c, 0.0f, 0.0f, 0.0f};
    for (int row0 = 0; row0 < q.total; row0 += TAPS) {
        float xs[TAPS][CIN], rs[TAPS];
#pragma unroll
        for (int u = 0; u < TAPS; ++u) {                                     // unconditional loads from clamped addresses (no branches between them), masked afterwards
            const int row = row0 + u, t_in = phase - pad + dil * (j0 + row);
            const bool ok = row < q.total && t_in >= 0 && t_in < n;
            const float* rowp = in + (q.base + (size_t)(ok ? t_in : 0)) * ld;
#pragma unroll
            for (int i = 0; i < CIN; ++i) {
                const float v = rowp[lane_off + i];
                xs[u][i] = ok ? v : 0.0f;
            }
            if (RES) {
                const int done = row - (TAPS - 1);
                const int dc = done < 0 ? 0 : (done < count ? done : count - 1);
                rs[u] = (res + (q.base + phase + (size_t)dil * (j0 + dc)) * ldy)[c];
            } else {
                rs[u] = 0.0f;
            }
        }
        tconv_rows<TAPS, CIN, 0>(acc, wreg, xs, rs, q, row0);
    }
    if (partial) {                                                           // shifted sums: mean = shift + s1 / count, M2 = s2 - s1^2 / count
        const float cn = (float)count, m = q.s1 / cn;
        partial[((size_t)b * tiles + slot) * C + c] = make_float4(cn, q.shift + m, fmaxf(q.s2 - q.s1 * m, 0.0f), 0.0f);
    }
}

// merge the per-tile (count, mean, M2) of one (window, channel) in tile order -> (mean, rstd) with the biased variance
__global__ __launch_bounds__(256) void k_chan_finalize(const float4* __restrict__ partial, int tiles, int C, float eps, float2* __restrict__ stats, int total) {
    const int i = (int)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int b = i / C, c = i - b * C;
    float na = 0.0f, mean = 0.0f, M2 = 0.0f;
    for (int t0 = 0; t0 < tiles; t0 += 8) {                  // the 8 loads of a group are independent: one memory latency per group, not per tile
        float4 p8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) p8[u] = t0 + u < tiles ? partial[((size_t)b * tiles + t0 + u) * C + c] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float4 p = p8[u];
            if (p.x <= 0.0f) continue;
            const float nb = p.x, delta = p.y - mean, nn = na + nb;
            mean += delta * (nb / nn);
            M2 += p.z + delta * delta * (na * nb / nn);
            na = nn;
        }
    }
    stats[i] = make_float2(mean, 1.0f / sqrtf(M2 / fmaxf(na, 1.0f) + eps));
}

// OffsetScale x 4 + rotary on the first rot_dim channels (:466-474): heads[h][(b * padded + t)][128]; rows t >= n are zero
__global__ __launch_bounds__(128) void k_offset_rotary(const float* __restrict__ proj, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ rcos, const float* __restrict__ rsin, float* __restrict__ heads, int n,
                                                       int padded, int rot_dim, long long head_stride) {
    const int d = threadIdx.x, tp = blockIdx.x, b = blockIdx.y;
    const size_t o = ((size_t)b * padded + tp) * kQk + d;
    if (tp >= n) {
#pragma unroll
        for (int hd = 0; hd < 4; ++hd) heads[hd * head_stride + o] = 0.0f;
        return;
    }
    const float* q = proj + ((size_t)b * n + tp) * kIn + kVu2;
    const float v = q[d], vp = q[d ^ 1];
#pragma unroll
    for (int hd = 0; hd < 4; ++hd) {
        float s = v * gamma[hd * kQk + d] + beta[hd * kQk + d];
        if (d < rot_dim) {
            const float sp = vp * gamma[hd * kQk + (d ^ 1)] + beta[hd * kQk + (d ^ 1)];
            s = s * rcos[tp * rot_dim + d] + sp * rsin[tp * rot_dim + d];          // pair swap, sign folded into rsin (:196-205)
        }
        heads[hd * head_stride + o] = s;
    }
}

// gate (:498-499) + ScaleNorm statistics (:502): g[m][j] = (att_u * v) * sigmoid(att_v * u); inv[m] = 1 / max(|g_m|, eps); one wave per row
__global__ __launch_bounds__(256) void k_gate_invnorm(const float* __restrict__ ao, const float* __restrict__ proj, float* __restrict__ g,
                                                      float* __restrict__ inv, int rows, float eps) {
    const int m = (int)blockIdx.x * 4 + ((int)threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (m >= rows) return;
    const float* a = ao + (size_t)m * kVu2;
    const float* p = proj + (size_t)m * kIn;
    float s = 0.0f;
    for (int j = lane; j < kVu; j += 64) {
        const float o = (a[kVu + j] * p[j]) * sigm(a[j] * p[kVu + j]);
        g[(size_t)m * kVu + j] = o;
        s += o * o;
    }
    s = wave_sum(s);
    if (lane == 0) inv[m] = 1.0f / fmaxf(sqrtf(s), eps);
}

// gf = LayerNorm(c1; w, b, eps1) and xn = LayerNorm(gf; no affine, eps2) over 256 channels (:510-512); one wave per row
__global__ __launch_bounds__(256) void k_ln_pair(const float* __restrict__ c1, const float* __restrict__ w, const float* __restrict__ b,
                                                 float* __restrict__ gf, float* __restrict__ xn, int rows, float eps1, float eps2) {
    const int m = (int)blockIdx.x * 4 + ((int)threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (m >= rows) return;
    float v[4];
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] = c1[(size_t)m * kInner + lane + 64 * i]; s += v[i]; }
    const float mu = wave_sum(s) * (1.0f / kInner);
    float q = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] -= mu; q += v[i] * v[i]; }
    const float r = 1.0f / sqrtf(wave_sum(q) * (1.0f / kInner) + eps1);
    s = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[i] = v[i] * r * w[lane + 64 * i] + b[lane + 64 * i];
        gf[(size_t)m * kInner + lane + 64 * i] = v[i];
        s += v[i];
    }
    const float mu2 = wave_sum(s) * (1.0f / kInner);
    q = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] -= mu2; q += v[i] * v[i]; }
    const float r2 = 1.0f / sqrtf(wave_sum(q) * (1.0f / kInner) + eps2);
#pragma unroll
    for (int i = 0; i < 4; ++i) xn[(size_t)m * kInner + lane + 64 * i] = v[i] * r2;
}

// instance-norm affine + per-channel PReLU (:528-533), in place
__global__ __launch_bounds__(256) void k_mem_norm_prelu(float* __restrict__ x, const float2* __restrict__ stats, const float* __restrict__ w,
                                                        const float* __restrict__ b, const float* __restrict__ slope, int n, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % kInner);
    const long long win = i / kInner / n;
    const float2 st = stats[(size_t)win * kInner + c];
    const float v = (x[i] - st.x) * st.y * w[c] + b[c];
    x[i] = v >= 0.0f ? v : v * slope[c];
}

// xu += memory; y = xv * xu + gf; n2 = LayerNorm(y; w, b, eps)   (:536-540); one wave per row
__global__ __launch_bounds__(256) void k_fsmn_combine(const float* __restrict__ uv, const float* __restrict__ mem, const float* __restrict__ gf,
                                                      const float* __restrict__ w, const float* __restrict__ b, float* __restrict__ n2, int rows, float eps) {
    const int m = (int)blockIdx.x * 4 + ((int)threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (m >= rows) return;
    float v[4];
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane + 64 * i;
        const float xu = uv[(size_t)m * kDim + c] + mem[(size_t)m * kInner + c];
        v[i] = uv[(size_t)m * kDim + kInner + c] * xu + gf[(size_t)m * kInner + c];
        s += v[i];
    }
    const float mu = wave_sum(s) * (1.0f / kInner);
    float q = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] -= mu; q += v[i] * v[i]; }
    const float r = 1.0f / sqrtf(wave_sum(q) * (1.0f / kInner) + eps);
#pragma unroll
    for (int i = 0; i < 4; ++i) n2[(size_t)m * kInner + lane + 64 * i] = v[i] * r * w[lane + 64 * i] + b[lane + 64 * i];
}

// LayerNorm over 512 channels with affine (:544), out of place; one wave per row
__global__ __launch_bounds__(256) void k_ln512(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b, float* __restrict__ y,
                                               int rows, float eps) {
    const int m = (int)blockIdx.x * 4 + ((int)threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (m >= rows) return;
    float v[8];
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; ++i) { v[i] = x[(size_t)m * kDim + lane + 64 * i]; s += v[i]; }
    const float mu = wave_sum(s) * (1.0f / kDim);
    float q = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; ++i) { v[i] -= mu; q += v[i] * v[i]; }
    const float r = 1.0f / sqrtf(wave_sum(q) * (1.0f / kDim) + eps);
#pragma unroll
    for (int i = 0; i < 8; ++i) y[(size_t)m * kDim + lane + 64 * i] = v[i] * r * w[lane + 64 * i] + b[lane + 64 * i];
}

// window norm + per-channel affine + skip (:549-551): out = (x - mean_b) * rstd_b * w[c] + b[c] + mi
__global__ __launch_bounds__(256) void k_window_affine_skip(const float* __restrict__ x, const float2* __restrict__ stats, const float* __restrict__ w,
                                                            const float* __restrict__ b, const float* __restrict__ mi, float* __restrict__ out, int n,
                                                            long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % kDim);
    const float2 st = stats[i / kDim / n];
    out[i] = ((x[i] - st.x) * st.y * w[c] + b[c]) + mi[i];
}

// ConvTranspose1d overlap-add (every sample has <= 2 frames), per (window, speaker) RMS restore, int32 truncate + clamp (:612-645).
// grid = windows * 2 (window-major, speaker-minor); PCM out is [call][speaker][n_win * W] (:655-656).
__global__ __launch_bounds__(1024) void k_decode_restore(const float* __restrict__ fr, const float* __restrict__ rms_in, int16_t* __restrict__ pcm,
                                                         float* __restrict__ f32, float* __restrict__ wav_ws, int W, int n, int R, int n_win) {
    __shared__ float red[16];
    const int b = blockIdx.x / kSpk, spk = blockIdx.x % kSpk;
    const float* rows = fr + ((size_t)spk * R + (size_t)b * n) * kEncK;
    float* wav = wav_ws + (size_t)blockIdx.x * W;
    float s = 0.0f;
    for (int i = threadIdx.x; i < W; i += blockDim.x) {
        const int t1 = min(i / kEncS, n - 1), t0 = t1 - 1;
        float v = 0.0f;
        if (t0 >= 0 && i - kEncS * t0 < kEncK) v += rows[(size_t)t0 * kEncK + (i - kEncS * t0)];
        if (i - kEncS * t1 < kEncK) v += rows[(size_t)t1 * kEncK + (i - kEncS * t1)];
        wav[i] = v;
        s += v * v;
    }
    const float rms_out = sqrtf(block_sum(s, red) / (float)W);
    const float gain = rms_out > 0.0f ? rms_in[b] / rms_out : 0.0f;
    const int call = b / n_win, win = b - call * n_win;
    const size_t o = ((size_t)(call * kSpk + spk) * n_win + win) * W;
    for (int i = threadIdx.x; i < W; i += blockDim.x) {
        const float y = wav[i] * gain;
        if (f32) f32[o + i] = y;
        if (pcm) pcm[o + i] = (int16_t)(int)fminf(fmaxf(truncf(y), -32768.0f), 32767.0f);
    }
}

int xfail(std::string& err, int st, const std::string& msg) { err = msg; return st; }
#define MF_HIP(expr)                                                                                  \
    do {                                                                                              \
        hipError_t _e = (expr);                                                                       \
        if (_e != hipSuccess) return xfail(err, ADE_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

struct LayerW {
    const float *in_w, *in_b, *in_c, *gamma, *beta, *out_w, *out_b, *out_c;
    const float *front_w, *front_b, *n1_w, *n1_b, *uv_w, *uv_b, *uv_c, *ml_w, *ml_b, *mp_w, *mem_w[4], *mn_w[4], *mn_b[4], *mprelu[4], *n2_w, *n2_b, *back_w,
        *back_b;
    float front_alpha;
};

}  // namespace

struct MossformerEngine : SubEngine {
    int device = 0, W = 0, n_win = 1, n = 0, padded = 0, groups = 0, layers = 0;
    float hyper[hCount] = {};
    float* d_w = nullptr;
    const float *encoder_w = nullptr, *front_w = nullptr, *front_b = nullptr, *front_wsum = nullptr, *emb_pos = nullptr, *rot_cos = nullptr, *rot_sin = nullptr,
                *mm_w = nullptr, *mm_b = nullptr, *intra_w = nullptr, *intra_b = nullptr, *tail_w = nullptr, *tail_b = nullptr, *maskdec_w = nullptr,
                *decoder_w = nullptr;
    std::vector<LayerW> L;
    int capacity = 0;
    float* ws = nullptr;
    float2 *gains = nullptr, *wstats = nullptr, *cstats = nullptr;
    float4* partial = nullptr;     // per (window, time tile, channel) partial statistics of the memory convolutions
    float4* wpart = nullptr;       // per (window, slice) partial statistics of the two window norms
    float* lkv_part = nullptr;     // split-K partial sums of the linear-attention key / value product
    float lin_scale = 0.0f;        // dynamic-axes export: 1 / frames applied to the reduced product at run time (0: folded into the weights)
    int lkv_splits = 1;
    float *rms_in = nullptr, *XE = nullptr, *MI = nullptr, *H = nullptr, *inv = nullptr, *P = nullptr, *P2 = nullptr, *heads = nullptr, *ATT = nullptr, *AO = nullptr,
          *LKV = nullptr, *G = nullptr, *Y = nullptr, *C1 = nullptr, *GF = nullptr, *XN = nullptr, *UV = nullptr, *UV2 = nullptr, *F1 = nullptr, *XP = nullptr,
          *M0 = nullptr, *M1 = nullptr, *N2 = nullptr, *HL = nullptr, *MO = nullptr, *GP = nullptr, *SEP = nullptr, *FR = nullptr, *WAV = nullptr;

    ~MossformerEngine() override {
        (void)hipSetDevice(device);
        if (d_w) (void)hipFree(d_w);
        if (ws) (void)hipFree(ws);
    }
    int frames() const override { return n; }
    int in_len() const override { return W * n_win; }
    int out_len() const override { return W * n_win; }
    bool accepts_float_input() const override { return true; }
    int n_outputs() const override { return kSpk; }          // separated_0 / separated_1 (:689-690)
    int reserve(int batch, std::string& err) override;
    int run(hipStream_t s, const int16_t* d_in, int batch, int16_t* d_out, float* d_f32, std::string& err) override;
    int tap(hipStream_t s, const char* name, int batch, float* out, size_t count, size_t* written, std::string& err) override;
};

int mossformer_create(const std::map<std::string, Tensor>& tensors, int window_len, int n_win, bool dynamic, int device, SubEngine** out, std::string& err) {
    *out = nullptr;
    if (n_win < 1) return xfail(err, ADE_ERR_BAD_VALUE, "mossformer: n_win must be >= 1");
    if (window_len < kEncK || (window_len - kEncK) % kEncS != 0)
        return xfail(err, ADE_ERR_SHAPE_MISMATCH, "mossformer: the window length must be 16 + a multiple of the encoder stride 8 (ConvTranspose reconstructs exactly)");
    const int n = (window_len - kEncK) / kEncS + 1;
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
    const Tensor* hy = shaped("hyper", {hCount});
    if (!hy) return status();
    int layers = 0;
    while (tensors.count("fl_in_w_" + std::to_string(layers))) ++layers;
    if (layers < 1) return xfail(err, ADE_ERR_MISSING_KEY, "weights: tensor missing: fl_in_w_0");
    const Tensor* alpha = shaped("fs_front_alpha", {layers});
    if (!alpha) return status();
    MossformerEngine* e = new MossformerEngine();
    auto bail = [&](int st) { delete e; return st; };
    memcpy(e->hyper, hy->data, sizeof(e->hyper));
    const int group = (int)e->hyper[hGroup], rot = (int)e->hyper[hRotDim], depth = (int)e->hyper[hMemDepth], lorder = (int)e->hyper[hLorder];
    if (group < 16 || group > 4096 || rot < 2 || rot > kQk || (rot & 1) || (int)e->hyper[hDwPad] != (kDw - 1) / 2 || depth < 1 || depth > 2 || 2 * lorder - 1 != kMemK)
        return bail(xfail(err, ADE_ERR_UNSUPPORTED, "mossformer: unsupported geometry (group 16..4096, even rotary <= 128, depthwise k 17, memory order 20, depth 1..2)"));
    e->lin_scale = dynamic ? (float)(1.0 / (double)n) : 0.0f;       // torch: tensor * (1.0 / n) -- the Python float becomes one fp32 scalar
    e->device = device; e->W = window_len; e->n_win = n_win; e->n = n; e->layers = layers;
    e->padded = (n + group - 1) / group * group;
    e->groups = e->padded / group;

    struct Item { const float* src; size_t count; size_t at; };
    std::vector<Item> items;
    size_t arena = 0;
    auto place = [&](const Tensor* t) { const size_t at = arena; arena += (t->count + 63) & ~(size_t)63; items.push_back({t->data, t->count, at}); return at; };
    std::vector<std::pair<const float**, size_t>> fix;
    auto want = [&](const float** dst, const std::string& name, std::vector<int> dims) -> bool {
        const Tensor* t = shaped(name, dims);
        if (!t) return false;
        fix.push_back({dst, place(t)});
        return true;
    };
    bool ok = want(&e->encoder_w, "encoder_w", {kDim, 1, kEncK}) && want(&e->front_w, "front_w", {kDim, kDim, 1}) && want(&e->front_b, "front_b", {kDim}) &&
              want(&e->emb_pos, "emb_pos", {1, kDim, n}) && want(&e->rot_cos, "rot_cos", {1, n, 1, rot}) && want(&e->rot_sin, "rot_signed_sin", {1, n, 1, rot}) &&
              want(&e->mm_w, "mm_norm_w", {kDim}) && want(&e->mm_b, "mm_norm_b", {kDim}) && want(&e->intra_w, "intra_norm_w", {kDim}) &&
              want(&e->intra_b, "intra_norm_b", {kDim}) && want(&e->tail_w, "tail_gate_w", {kSpk * 2 * kDim, kDim, 1}) &&
              want(&e->tail_b, "tail_gate_b", {kSpk * 2 * kDim}) && want(&e->maskdec_w, "mask_decoder_w", {kDim, kDim, 1}) &&
              want(&e->decoder_w, "decoder_w", {kDim, 1, kEncK});
    e->L.resize((size_t)layers);
    for (int i = 0; ok && i < layers; ++i) {
        const std::string s = "_" + std::to_string(i);
        LayerW& l = e->L[i];
        l.front_alpha = alpha->data[i];
        ok = want(&l.in_w, "fl_in_w" + s, {kIn, kDim}) && want(&l.in_b, "fl_in_b" + s, {kIn}) && want(&l.in_c, "fl_in_c" + s, {kIn, 1, kDw}) &&
             want(&l.gamma, "qkos_gamma" + s, {4, kQk}) && want(&l.beta, "qkos_beta" + s, {4, kQk}) && want(&l.out_w, "fl_out_w" + s, {kDim, kVu}) &&
             want(&l.out_b, "fl_out_b" + s, {kDim}) && want(&l.out_c, "fl_out_c" + s, {kDim, 1, kDw}) && want(&l.front_w, "fs_front_w" + s, {kInner, kDim}) &&
             want(&l.front_b, "fs_front_b" + s, {kInner}) && want(&l.n1_w, "fs_n1_w" + s, {kInner}) && want(&l.n1_b, "fs_n1_b" + s, {kInner}) &&
             want(&l.uv_w, "fs_uv_w" + s, {2 * kInner, kInner}) && want(&l.uv_b, "fs_uv_b" + s, {2 * kInner}) && want(&l.uv_c, "fs_uv_c" + s, {2 * kInner, 1, kDw}) &&
             want(&l.ml_w, "fs_mem_linear_w" + s, {kInner, kInner}) && want(&l.ml_b, "fs_mem_linear_b" + s, {kInner}) &&
             want(&l.mp_w, "fs_mem_project_w" + s, {kInner, kInner}) && want(&l.n2_w, "fs_n2_w" + s, {kInner}) && want(&l.n2_b, "fs_n2_b" + s, {kInner}) &&
             want(&l.back_w, "fs_back_w" + s, {kDim, kInner}) && want(&l.back_b, "fs_back_b" + s, {kDim});
        for (int j = 0; ok && j < depth; ++j) {
            const std::string sj = s + "_" + std::to_string(j);
            ok = want(&l.mem_w[j], "fs_mem_w" + sj, {kInner, j + 1, kMemK}) && want(&l.mn_w[j], "fs_mem_norm_w" + sj, {kInner}) &&
                 want(&l.mn_b[j], "fs_mem_norm_b" + sj, {kInner}) && want(&l.mprelu[j], "fs_mem_prelu" + sj, {kInner});
        }
    }
    if (!ok) return bail(status());
    // row sums of the front 1x1 weight: the window mean folds into the GEMM store
    std::vector<float> wsum((size_t)kDim);
    {
        const Tensor* fw = find("front_w");
        for (int o = 0; o < kDim; ++o) {
            double a = 0.0;
            for (int c = 0; c < kDim; ++c) a += fw->data[(size_t)o * kDim + c];
            wsum[o] = (float)a;
        }
    }
    const size_t a_wsum = arena;
    arena += kDim;
    if (hipSetDevice(device) != hipSuccess) return bail(xfail(err, ADE_ERR_DEVICE, "hipSetDevice failed"));
    if (hipMalloc((void**)&e->d_w, arena * sizeof(float)) != hipSuccess) return bail(xfail(err, ADE_ERR_DEVICE, "hipMalloc of the MossFormer2 weights failed"));
    for (const Item& it : items)
        if (hipMemcpy(e->d_w + it.at, it.src, it.count * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
            return bail(xfail(err, ADE_ERR_DEVICE, "upload of the MossFormer2 weights failed"));
    if (hipMemcpy(e->d_w + a_wsum, wsum.data(), kDim * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
        return bail(xfail(err, ADE_ERR_DEVICE, "upload of the MossFormer2 weights failed"));
    for (auto& f : fix) *f.first = e->d_w + f.second;
    e->front_wsum = e->d_w + a_wsum;
    *out = e;
    return ADE_OK;
}

int MossformerEngine::reserve(int batch, std::string& err) {
    if (batch <= capacity) return ADE_OK;
    MF_HIP(hipSetDevice(device));
    MF_HIP(hipDeviceSynchronize());
    if (ws) (void)hipFree(ws);
    ws = nullptr;
    capacity = 0;
    const size_t B = (size_t)batch * n_win, R = B * n, RP = B * padded, g = (size_t)hyper[hGroup];
    struct Carve { float** p; size_t count; };
    float *f_gains = nullptr, *f_wstats = nullptr, *f_cstats = nullptr, *f_partial = nullptr, *f_wpart = nullptr;
    std::vector<Carve> cs = {{&f_gains, 2 * B}, {&f_wstats, 2 * B}, {&f_cstats, 2 * B * kInner}, {&f_partial, 4 * B * (size_t)((n + 31) / 32 + 2) * kInner}, {&f_wpart, 4 * B * kWsSlices}, {&lkv_part, (size_t)kLkvSplits * B * kQk * kVu2}, {&rms_in, B}, {&XE, R * kDim}, {&MI, R * kDim}, {&H, R * kDim},
                             {&inv, R}, {&P, R * kIn}, {&P2, R * kIn}, {&heads, 4 * RP * kQk}, {&ATT, B * groups * g * g}, {&AO, R * kVu2},
                             {&LKV, B * kQk * kVu2}, {&G, R * kVu}, {&Y, R * kDim}, {&C1, R * kInner}, {&GF, R * kInner}, {&XN, R * kInner}, {&UV, R * kDim},
                             {&UV2, R * kDim}, {&F1, R * kInner}, {&XP, R * kInner}, {&M0, R * kInner}, {&M1, R * kInner}, {&N2, R * kInner}, {&HL, R * kDim},
                             {&MO, R * kDim}, {&GP, R * kSpk * 2 * kDim}, {&SEP, kSpk * R * kDim}, {&FR, kSpk * R * kEncK}, {&WAV, kSpk * B * (size_t)W}};
    size_t total = 0;
    for (auto& c : cs) total += (c.count + 63) & ~(size_t)63;
    MF_HIP(hipMalloc((void**)&ws, total * sizeof(float)));
    size_t at = 0;
    for (auto& c : cs) { *c.p = ws + at; at += (c.count + 63) & ~(size_t)63; }
    gains = reinterpret_cast<float2*>(f_gains);
    wstats = reinterpret_cast<float2*>(f_wstats);
    cstats = reinterpret_cast<float2*>(f_cstats);
    partial = reinterpret_cast<float4*>(f_partial);
    wpart = reinterpret_cast<float4*>(f_wpart);
    capacity = batch;
    return ADE_OK;
}

int MossformerEngine::run(hipStream_t s, const int16_t* d_in, int batch, int16_t* d_out, float* d_f32, std::string& err) {
    if (batch == 0) return ADE_OK;
    int st = reserve(batch, err);
    if (st != ADE_OK) return st;
    using namespace gemm;
    const int B = batch * n_win, R = B * n, g = (int)hyper[hGroup], rot = (int)hyper[hRotDim], depth = (int)hyper[hMemDepth], lorder = (int)hyper[hLorder];
    const long long head_stride = (long long)B * padded * kQk;
    lkv_splits = kLkvSplits;
    auto rows4 = [&](int rows) { return dim3((unsigned)((rows + 3) / 4)); };
    auto flat = [&](long long total) { return dim3((unsigned)((total + 255) / 256)); };
    auto window_stats = [&](hipStream_t st_, const float* x, float eps) {
        hipLaunchKernelGGL(k_window_stats_partial, dim3((unsigned)B, kWsSlices), dim3(1024), 0, st_, x, (long long)n * kDim, wpart);
        hipLaunchKernelGGL(k_window_stats_final, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, st_, (const float4*)wpart, eps, wstats, B);
    };
    auto dwconv = [&](hipStream_t st_, const float* x, const float* w, const float* res, float* y, int C, int nb) {   // y = res + x + depthwise_17(x)
        const int runs = (n + kTconvRun - 1) / kTconvRun;
        const dim3 grid((unsigned)(C / 64), (unsigned)((runs + 3) / 4), (unsigned)nb);
        if (res)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_tconv<kDw, 1, true, true>), grid, dim3(256), 0, st_, x, x, C, C, C, w, res, y, C, n, 1, (kDw - 1) / 2, runs, (float4*)nullptr);
        else
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_tconv<kDw, 1, true, false>), grid, dim3(256), 0, st_, x, x, C, C, C, w, res, y, C, n, 1, (kDw - 1) / 2, runs, (float4*)nullptr);
    };

    // front end: RMS stages, encoder, window norm folded into the 1x1 conv, positions                         (:571-591)
    hipLaunchKernelGGL(k_norm_audio, dim3((unsigned)B), dim3(1024), 0, s, d_in, float_in, W, hyper[hNormFactor], gains, rms_in);
    launch(s, EncFrameA{d_in, float_in, gains, W, n}, WeightNK{encoder_w, kEncK}, BiasActRowStore<1>{XE, nullptr, kDim, 0.0f}, R, kDim, kEncK);
    window_stats(s, XE, hyper[hFrontEps]);
    launch(s, RowMajorA{XE, kDim}, WeightNK{front_w, kDim}, FrontStore{H, MI, wstats, front_wsum, front_b, emb_pos, n}, R, kDim, kDim);

    for (int i = 0; i < layers; ++i) {
        const LayerW& l = L[i];
        // ---- FLASH block (:451-505)
        hipLaunchKernelGGL(k_shift_invnorm, rows4(R), dim3(256), 0, s, (const float*)H, inv, R, n, hyper[hFlNormEps]);
        launch(s, ShiftA{H, n}, WeightNK{l.in_w, kDim}, ScaleSiluStore{P, inv, l.in_b, kIn}, R, kIn, kDim);
        dwconv(s, P, l.in_c, nullptr, P2, kIn, B);
        hipLaunchKernelGGL(k_offset_rotary, dim3((unsigned)padded, (unsigned)B), dim3(kQk), 0, s, (const float*)P2, l.gamma, l.beta, rot_cos, rot_sin, heads, n,
                           padded, rot, head_stride);
        const float *quad_q = heads, *lin_q = heads + head_stride, *quad_k = heads + 2 * head_stride, *lin_k = heads + 3 * head_stride;
        launch_batched(s, QuadScoreProb{quad_q, quad_k, ATT, g}, B * groups, g, g);
        if (lkv_splits == 1) launch_batched(s, LinKvProb{lin_k, P2, LKV, n, padded, 1}, B, kQk, kVu2);
        else {
            launch_batched(s, LinKvProb{lin_k, P2, lkv_part, n, padded, lkv_splits}, B * lkv_splits, kQk, kVu2);
            hipLaunchKernelGGL(k_lkv_reduce, flat((long long)B * kQk * kVu2), dim3(256), 0, s, (const float*)lkv_part, LKV, lkv_splits, (long long)kQk * kVu2,
                               (long long)B * kQk * kVu2);
        }
        if (lin_scale != 0.0f) hipLaunchKernelGGL(k_lkv_scale, flat((long long)B * kQk * kVu2), dim3(256), 0, s, LKV, lin_scale, (long long)B * kQk * kVu2);
        launch_batched(s, AttOutProb{ATT, P2, lin_q, LKV, AO, g, groups, n, padded}, B * groups, g, kVu2);
        hipLaunchKernelGGL(k_gate_invnorm, rows4(R), dim3(256), 0, s, (const float*)AO, (const float*)P2, G, inv, R, hyper[hFlOutNormEps]);
        launch(s, RowMajorA{G, kVu}, WeightNK{l.out_w, kVu}, ScaleSiluStore{Y, inv, l.out_b, kDim}, R, kDim, kVu);
        dwconv(s, Y, l.out_c, H, H, kDim, B);
        // ---- gated FSMN block (:507-541)
        launch(s, RowMajorA{H, kDim}, WeightNK{l.front_w, kDim}, BiasActRowStore<3>{C1, l.front_b, kInner, l.front_alpha}, R, kInner, kDim);
        hipLaunchKernelGGL(k_ln_pair, rows4(R), dim3(256), 0, s, (const float*)C1, l.n1_w, l.n1_b, GF, XN, R, hyper[hFsN1Eps], hyper[hFsLnEps]);
        launch(s, RowMajorA{XN, kInner}, WeightNK{l.uv_w, kInner}, BiasActRowStore<2>{UV, l.uv_b, kDim, 0.0f}, R, kDim, kInner);
        dwconv(s, UV, l.uv_c, nullptr, UV2, kDim, B);
        launch(s, RowMajorA{UV2, kDim}, WeightNK{l.ml_w, kInner}, BiasActRowStore<1>{F1, l.ml_b, kInner, 0.0f}, R, kInner, kInner);
        launch(s, RowMajorA{F1, kInner}, WeightNK{l.mp_w, kInner}, BiasActRowStore<0>{XP, nullptr, kInner, 0.0f}, R, kInner, kInner);
        float* mem_prev = nullptr;
        for (int j = 0; j < depth; ++j) {
            float* dst = (j & 1) ? M1 : M0;
            const int dil = 1 << j, pad = lorder + (dil - 1) * (lorder - 1) - 1;
            // dense input of conv j = [out_{j-1}, xp] (depth <= 2): group g reads concatenated channels g (j + 1) .. g (j + 1) + j
            const int runs = ((n + dil - 1) / dil + kTconvRun - 1) / kTconvRun, tiles = runs * dil;
            const dim3 grid(kInner / 64, (unsigned)((tiles + 3) / 4), (unsigned)B);
            if (j == 0)
                hipLaunchKernelGGL(HIP_KERNEL_NAME(k_tconv<kMemK, 1, false, false>), grid, dim3(256), 0, s, (const float*)XP, (const float*)XP, kInner, kInner, kInner,
                                   l.mem_w[0], (const float*)nullptr, dst, kInner, n, dil, pad, runs, partial);
            else
                hipLaunchKernelGGL(HIP_KERNEL_NAME(k_tconv<kMemK, 2, false, false>), grid, dim3(256), 0, s, (const float*)mem_prev, (const float*)XP, kInner, kInner,
                                   kInner, l.mem_w[1], (const float*)nullptr, dst, kInner, n, dil, pad, runs, partial);
            hipLaunchKernelGGL(k_chan_finalize, flat((long long)B * kInner), dim3(256), 0, s, (const float4*)partial, tiles, kInner, hyper[hMemNormEps], cstats,
                               B * kInner);
            hipLaunchKernelGGL(k_mem_norm_prelu, flat((long long)R * kInner), dim3(256), 0, s, dst, (const float2*)cstats, l.mn_w[j], l.mn_b[j], l.mprelu[j], n,
                               (long long)R * kInner);
            mem_prev = dst;
        }
        hipLaunchKernelGGL(k_fsmn_combine, rows4(R), dim3(256), 0, s, (const float*)UV2, (const float*)mem_prev, (const float*)GF, l.n2_w, l.n2_b, N2, R,
                           hyper[hFsN2Eps]);
        launch(s, RowMajorA{N2, kInner}, WeightNK{l.back_w, kInner}, ResidualBiasStore{H, l.back_b, kDim}, R, kDim, kInner);
    }
    // final norms + skip (:544-551)
    hipLaunchKernelGGL(k_ln512, rows4(R), dim3(256), 0, s, (const float*)H, mm_w, mm_b, HL, R, hyper[hMmEps]);
    window_stats(s, HL, hyper[hIntraEps]);
    hipLaunchKernelGGL(k_window_affine_skip, flat((long long)R * kDim), dim3(256), 0, s, (const float*)HL, (const float2*)wstats, intra_w, intra_b,
                       (const float*)MI, MO, n, (long long)R * kDim);
    // speaker tail, decoder, RMS restore (:599-645)
    launch(s, LeakyA{MO, kDim, hyper[hTailAlpha]}, WeightNK{tail_w, kDim}, BiasActRowStore<0>{GP, tail_b, kSpk * 2 * kDim, 0.0f}, R, kSpk * 2 * kDim, kDim);
    launch(s, SpeakerGateA{GP, R}, WeightNK{maskdec_w, kDim}, MaskEncStore{SEP, XE, R}, kSpk * R, kDim, kDim);
    launch(s, RowMajorA{SEP, kDim}, RowMajorB{decoder_w, kEncK}, BiasActRowStore<0>{FR, nullptr, kEncK, 0.0f}, kSpk * R, kEncK, kDim);
    hipLaunchKernelGGL(k_decode_restore, dim3((unsigned)(B * kSpk)), dim3(1024), 0, s, (const float*)FR, (const float*)rms_in, d_out, d_f32, WAV, W, n, R, n_win);
    MF_HIP(hipGetLastError());
    return ADE_OK;
}

int MossformerEngine::tap(hipStream_t s, const char* name, int batch, float* out, size_t count, size_t* written, std::string& err) {
    const size_t R = (size_t)batch * n_win * n;
    const float* src = nullptr;
    size_t cnt = 0;
    if (strcmp(name, "mdl_in") == 0) { src = MI; cnt = R * kDim; }            // (window, frame, 512)
    else if (strcmp(name, "mdl_out") == 0) { src = MO; cnt = R * kDim; }
    else if (strcmp(name, "x_enc") == 0) { src = XE; cnt = R * kDim; }
    else return xfail(err, ADE_ERR_NOT_FOUND, std::string("unknown tap: ") + name);
    if (!src || batch <= 0) return xfail(err, ADE_ERR_NOT_FOUND, "tap has no data yet");
    if (count < cnt) return xfail(err, ADE_ERR_SHAPE_MISMATCH, "tap buffer too small");
    MF_HIP(hipStreamSynchronize(s));
    MF_HIP(hipMemcpy(out, src, cnt * sizeof(float), hipMemcpyDeviceToHost));
    *written = cnt;
    return ADE_OK;
}

}  // namespace ade
