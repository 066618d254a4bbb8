// ade_zipenhancer.hip — ZipEnhancer (16 kHz speech enhancement, dual-path Zipformer2) on the MI355X: SURVEY.md section 8 rows a15 / a16.
//
// Reference: ZipEnhancer.forward over the FUSED tensors its constructor registers (ZipEnhancer/Export_ZipEnhancer.py:818-927, :437-664)
// with the ten forward overrides it installs on the ModelScope layers (:118-355):
//   int16 (B, 1, n_win W) -> float, fold into windows (:837) -> / sqrt(mean x^2 + 1e-6) per window (:839) -> STFT(400, hop 100, hann,
//   reflect) -> (|X|^2 + 1e-9)^0.15 and atan2(im, re + 1e-5) (:843-844) -> DenseEncoder: 1x1 conv, causal dilated 2 x 3 dense block of
//   depth 4, (1, 3) stride-2 conv, each + InstanceNorm + PReLU (:847-853, :701-723) -> 4 dual-path Zipformer2 encoders, the middle
//   two on a 2 x 2 down-sampled grid (:859-863, :782-816) -> the mask | phase decoder pair: dense block, sub-pixel (1, 3) up-sampling,
//   (1, 2) heads (:864-877, :725-780) -> relu(mask)^(1/0.3) x unit phase vector (:882-892) -> ISTFT (x 1 / sum w^2) -> x the window's
//   norm factor (:900) -> NaN -> 0, clamp, truncate to int16 (:917-918).
// One Zipformer2 layer on S sequences of n tokens (:143-187), batch-major:
//   [attn | ff1] = x W^T + b;  A = softmax(q k^T + skew(p P_h))  (:232-289);  x += W swooshL(ff1);  x += W ((A_0 (tanh(s) u)) y)  (:304-317);
//   x += W (A v) (:292-301);  x += W swooshR(dw15(a sigmoid(g)))  (:320-339);  x += ff2;  x = x0 + (x - x0) c_mid;  x += W (A v');
//   x += conv';  x += ff3;  x = x / |x - nb|_2 * fs + x0 * rs  (:175-183).
//
// MI355X mapping.  Activations are token-major, channels-last fp32: X[(b, t, f)][64].  Both attention axes read X in place (a frequency
// sequence is F consecutive rows, a time sequence F rows apart): the reference's four permutes per encoder never materialise.
//   * every Linear / 1x1 conv AND every dense 2 x 3 / (1, 3) convolution is a tall-skinny fp32 matrix-core GEMM (csrc/ade_gemm64.h,
//     256 x 64 tiles, v_mfma_f32_16x16x4_f32): the convolutions are implicit GEMMs whose A-operand loader shifts (frame, bin), zero-pads
//     and applies the PRODUCER's InstanceNorm + PReLU on the fly -- the dense block's skip tensors are stored once, raw, and never
//     rewritten (the reference concatenates and normalises them layer by layer);
//   * InstanceNorm statistics: one deterministic two-stage fp64 reduction per produced tensor; for the first (1x1) layer they follow
//     analytically from five moments of (mag, phase), so that layer costs no reduction over its 64-channel output at all;
//   * Swoosh activations, biases, residual adds, the mid bypass and the sub-pixel shuffle are GEMM loaders / stores;
//   * attention weights are materialised once per layer (they are used three times: head 0 by NonlinAttention, all heads by both
//     SelfAttention modules) with rows padded to 4 floats; the three weighted sums run on the matrix cores with the weights as the
//     A operand straight from HBM (float4 per lane) and the values staged in LDS;
//   * STFT / ISTFT are the reference's dense windowed DFT as MFMA GEMMs with ITS fp32-angle tables by default (manifest
//     ade_dft_tables = "exact" for exactly reduced angles): the phase feature of a near-silent bin is ill-conditioned, so the tables'
//     own error is part of what the network sees (as for Mel-Band-Roformer).
// Geometry (channels, heads, kernel sizes) comes from the blob's `zip_config` (audio_denoiser_onnx_amd/zipenhancer.py).
#include "ade_gemm64.h"
#include "ade_zip16.h"
#include "ade_internal.h"
#include "../../include/ade.h"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace ade {

namespace {

using namespace dev;
using zip16::ldx1;
using zip16::ldx4;
using zip16::ldh4;
using zip16::stx1;
using zip16::stx4;

constexpr int kZN = 400, kZHop = 100, kZF = kZN / 2 + 1, kZC2 = 2 * kZF;   // Export_ZipEnhancer.py:47-49
constexpr int kChunkTok = 2048;                                             // tokens per partial-statistics block

// ---- small device helpers -------------------------------------------------------------------------------------------------
// F.softplus (threshold 20) on the hardware exp2 / log2 units (v_exp_f32, v_log_f32, ~1 ulp each): log(1 + e^x) loses RELATIVE accuracy where e^x < 1e-7,
// i.e. where the result is below 1e-7 ABSOLUTE next to the -0.08 x term of the Swoosh it feeds -- far inside the parity tolerance; libm's expf + log1pf cost
// ~50 VALU instructions per element and made the Swoosh-loading GEMMs VALU-bound (19 % matrix-core busy, profiles/r02_zipenhancer_mfma_busy.txt).
// (round 5: the two instructions themselves -- __builtin_amdgcn_exp2f / __builtin_amdgcn_logf -- instead of __expf / __logf, which wrap them in denormal scaling and compares
//  that 1 + e^x in [1, 5e8] never needs: ~9 VALU instructions per activation instead of ~28; the Swoosh-loading kernels were VALU-bound on it)
__device__ __forceinline__ float softplus_f(float x) { return x > 20.0f ? x : 0.6931471805599453f * __builtin_amdgcn_logf(1.0f + __builtin_amdgcn_exp2f(x * kLog2e)); }
__device__ __forceinline__ float swoosh_l(float x) { return softplus_f(x - 4.0f) - 0.08f * x; }        // (:135-136), offset folded into the bias
__device__ __forceinline__ float swoosh_r(float x) { return softplus_f(x - 1.0f) - 0.08f * x; }        // (:138)
__device__ __forceinline__ float sigmoid_p(float x) { return 1.0f / (1.0f + __expf(-x)); }

// rows of a sequence: row(seq, p) = (seq / sdiv) * sa + (seq % sdiv) * sb + p * ps
struct SeqGeo {
    int nseq, n, sdiv;
    long long sa, sb, ps;
    __device__ __forceinline__ long long row0(int seq) const { return (long long)(seq / sdiv) * sa + (long long)(seq % sdiv) * sb; }
};

// ---- front: window norm, STFT operands, features ------------------------------------------------------------------------------
// norm[r] = sqrt(mean(x^2) + 1e-6) over the window's L samples (int16 amplitude, :819, :839)
__global__ __launch_bounds__(256) void k_zip_window_norm(const int16_t* __restrict__ pcm, const float* __restrict__ fin, float* __restrict__ norm, int L) {
    __shared__ double red[256];
    const int r = blockIdx.x, tid = threadIdx.x;
    double s = 0.0;
    for (int i = tid; i < L; i += 256) {
        const float v = fin ? fin[(size_t)r * L + i] : (float)pcm[(size_t)r * L + i];
        s += (double)(v * v);
    }
    red[tid] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) red[tid] += red[tid + o];
        __syncthreads();
    }
    if (tid == 0) norm[r] = sqrtf((float)(red[0] / (double)L) + 1e-6f);
}

struct ZFrameB {               // B(k, j) = reflect-padded sample k of frame j = (window r, t), divided by the window's norm (:840, STFT_Process.py:268-281)
    static constexpr bool kAlongN = false;
    const int16_t* pcm;
    const float* fin;
    const float* norm;
    int L, T;
    __device__ float operator()(int k, int j) const {
        const int r = j / T, t = j - r * T;
        int idx = t * kZHop + k - kZN / 2;
        if (idx < 0) idx = -idx;
        else if (idx >= L) idx = 2 * (L - 1) - idx;
        const float v = fin ? fin[(size_t)r * L + idx] : (float)pcm[(size_t)r * L + idx];
        return v / norm[r];
    }
};
struct ZSpecStore {            // C(c, j) -> spec[c][j]   (re rows 0..200, im rows 201..401)
    float* spec;
    int J;
    __device__ void operator()(int m, int n, float v) const { spec[(size_t)m * J + n] = v; }
};

// feat[(j, f)] = (mag, pha) and the per-block partial moments of (mag, pha) for the analytic InstanceNorm of the 1x1 layer.
// grid (chunks over the T * 201 positions of one window, windows); partial[(r * nchunk + chunk) * 5 + q]
__global__ __launch_bounds__(256) void k_zip_features(const float* __restrict__ spec, float2* __restrict__ feat, double* __restrict__ partial, int T, int J,
                                                      int per_chunk) {
    __shared__ double red[5][256];
    const int r = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x, TF = T * kZF;
    const int lo = chunk * per_chunk, hi = min(TF, lo + per_chunk);
    double acc[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    for (int i = lo + tid; i < hi; i += 256) {
        const int t = i / kZF, f = i - t * kZF, j = r * T + t;
        const float re = spec[(size_t)f * J + j], im = spec[(size_t)(kZF + f) * J + j];
        const float mag = powf(re * re + im * im + 1e-9f, 0.15f);                      // compress_factor / 2 (:365, :843)
        const float pha = atan2f(im, re + 1e-5f);                                     // (:844)
        feat[(size_t)r * TF + i] = make_float2(mag, pha);
        acc[0] += mag; acc[1] += pha; acc[2] += (double)mag * mag; acc[3] += (double)pha * pha; acc[4] += (double)mag * pha;
    }
#pragma unroll
    for (int q = 0; q < 5; ++q) red[q][tid] = acc[q];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o)
#pragma unroll
            for (int q = 0; q < 5; ++q) red[q][tid] += red[q][tid + o];
        __syncthreads();
    }
    if (tid < 5) partial[((size_t)r * gridDim.x + chunk) * 5 + tid] = red[tid][0];
}

// coef[r][c] = (a, b, d): the 1x1 conv (2 -> C) followed by its InstanceNorm collapses to a * mag + b * pha + d per (window, channel)  (:851)
__global__ void k_zip_conv1_coef(const double* __restrict__ partial, int nchunk, double count, const float* __restrict__ w, const float* __restrict__ bias,
                                 const float* __restrict__ gamma, const float* __restrict__ beta, float4* __restrict__ coef, int C) {
    const int r = blockIdx.x, c = threadIdx.x;
    if (c >= C) return;
    double m[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    for (int k = 0; k < nchunk; ++k)
        for (int q = 0; q < 5; ++q) m[q] += partial[((size_t)r * nchunk + k) * 5 + q];
    const double em = m[0] / count, ep = m[1] / count;
    const double vm = m[2] / count - em * em, vp = m[3] / count - ep * ep, cov = m[4] / count - em * ep;
    const double w0 = w[c * 2], w1 = w[c * 2 + 1];
    const double mean = w0 * em + w1 * ep + bias[c];
    double var = w0 * w0 * vm + w1 * w1 * vp + 2.0 * w0 * w1 * cov;
    if (var < 0.0) var = 0.0;
    const double sc = (double)gamma[c] / sqrt(var + 1e-5);
    coef[(size_t)r * C + c] = make_float4((float)(w0 * sc), (float)(w1 * sc), (float)(((double)bias[c] - mean) * sc + beta[c]), 0.0f);
}

// E0[(r, t, f)][c] = prelu(a mag + b pha + d): the dense block's input, final values
__global__ __launch_bounds__(256) void k_zip_conv1_apply(const float2* __restrict__ feat, const float4* __restrict__ coef, const float* __restrict__ slope,
                                                         float* __restrict__ e0, int TF, int C, long long total4) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;             // (token, channel quad)
    if (i >= total4) return;
    const int cq = C >> 2;
    const long long tok = i / cq;
    const int c = (int)(i - tok * cq) * 4, r = (int)(tok / TF);
    const float2 mp = feat[tok];
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const float4 k = coef[(size_t)r * C + c + u];
        v[u] = prelu_f(k.x * mp.x + k.y * mp.y + k.z, slope[c + u]);
    }
    st4(e0 + tok * C + c, v);
}

// the same values rounded to bf16 / IEEE half (the bf16 path's dense-block input)
template <bool HALF>
__global__ __launch_bounds__(256) void k_zip_conv1_apply16(const float2* __restrict__ feat, const float4* __restrict__ coef, const float* __restrict__ slope,
                                                           gemm16::bf16_t* __restrict__ e0, int TF, int C, long long total4) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total4) return;
    const int cq = C >> 2;
    const long long tok = i / cq;
    const int c = (int)(i - tok * cq) * 4, r = (int)(tok / TF);
    const float2 mp = feat[tok];
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const float4 k = coef[(size_t)r * C + c + u];
        v[u] = prelu_f(k.x * mp.x + k.y * mp.y + k.z, slope[c + u]);
    }
    *reinterpret_cast<uint2*>(e0 + tok * C + c) = gemm16::pack16x4_t<HALF>(make_float4(v[0], v[1], v[2], v[3]));
}

// ---- InstanceNorm statistics of a raw tensor: 64 channels at column ch0 of a [tokens][ld] matrix, per window ------------------------
// stage 1: grid (chunks, windows), thread = (channel quad, token lane); partial[((r * nchunk + chunk) * 64 + c) * 2 + {sum, sumsq}]
__global__ __launch_bounds__(256) void k_zip_stats_partial(const float* __restrict__ x, int ld, int ch0, int tok_per_win, double* __restrict__ partial) {
    __shared__ double red[16][16][8];
    const int r = blockIdx.y, chunk = blockIdx.x, q = threadIdx.x & 15, tl = threadIdx.x >> 4;
    const int lo = chunk * kChunkTok, hi = min(tok_per_win, lo + kChunkTok);
    double s[4] = {0.0, 0.0, 0.0, 0.0}, ss[4] = {0.0, 0.0, 0.0, 0.0};
    const float* base = x + (size_t)r * tok_per_win * ld + ch0 + 4 * q;
    // (the pass is latency-bound: four rows in flight per lane; the fp64 sums are order-insensitive at the precision the statistics are used at)
    for (int i0 = lo + tl; i0 < hi; i0 += 64) {
        float v[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = i0 + 16 * r;
            if (i < hi) ld4(base + (size_t)i * ld, v[r]);
            else v[r][0] = v[r][1] = v[r][2] = v[r][3] = 0.0f;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int u = 0; u < 4; ++u) { s[u] += v[r][u]; ss[u] += (double)v[r][u] * v[r][u]; }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) { red[tl][q][u] = s[u]; red[tl][q][4 + u] = ss[u]; }
    __syncthreads();
    if (tl == 0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            double a = 0.0, b = 0.0;
            for (int k = 0; k < 16; ++k) { a += red[k][q][u]; b += red[k][q][4 + u]; }
            double* dst = partial + (((size_t)r * gridDim.x + chunk) * 64 + 4 * q + u) * 2;
            dst[0] = a; dst[1] = b;
        }
    }
}
// stage 2: nrm[(r * nrm_ld + nrm_ch0 + c) * 2 + {scale, shift}] = (gamma / sqrt(var + eps), beta - mean * scale)     (eps 1e-5, biased variance)
__global__ __launch_bounds__(256) void k_zip_stats_final(const double* __restrict__ partial, int nchunk, double count, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         float* __restrict__ nrm, int nrm_ld, int nrm_ch0) {
    __shared__ double red[4][64][2];
    const int r = blockIdx.x, c = threadIdx.x & 63, part = threadIdx.x >> 6;       // 64 threads per window walked up to 64 partials one after the other: 30 us of latency per call, 15 calls per step
    double a = 0.0, b = 0.0;
    for (int k = part; k < nchunk; k += 4) {
        const double* src = partial + (((size_t)r * nchunk + k) * 64 + c) * 2;
        a += src[0]; b += src[1];
    }
    red[part][c][0] = a; red[part][c][1] = b;
    __syncthreads();
    if (part != 0) return;
    a = (red[0][c][0] + red[1][c][0]) + (red[2][c][0] + red[3][c][0]);
    b = (red[0][c][1] + red[1][c][1]) + (red[2][c][1] + red[3][c][1]);
    const double mean = a / count;
    double var = b / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const double sc = (double)gamma[c] / sqrt(var + 1e-5);
    float* dst = nrm + ((size_t)r * nrm_ld + nrm_ch0 + c) * 2;
    dst[0] = (float)sc;
    dst[1] = (float)((double)beta[c] - mean * sc);
}
// in-place normalise + PReLU of a [tokens][C] tensor (the encoder input: it has many consumers)
__global__ __launch_bounds__(256) void k_zip_norm_apply(float* __restrict__ x, const float* __restrict__ nrm, const float* __restrict__ slope, int tok_per_win, int C,
                                                        long long total4) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total4) return;
    const int cq = C >> 2;
    const long long tok = i / cq;
    const int c = (int)(i - tok * cq) * 4, r = (int)(tok / tok_per_win);
    float v[4];
    ld4(x + tok * C + c, v);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const float* k = nrm + ((size_t)r * C + c + u) * 2;
        v[u] = prelu_f(v[u] * k[0] + k[1], slope[c + u]);
    }
    st4(x + tok * C + c, v);
}

// ---- one layer of a causal dense block (:701-757) as an implicit GEMM ----------------------------------------------------------------
// A(token, k): k = tap * cin + ci, tap = kt * 3 + kf; the value is input channel ci of position (t - (1 - kt) dil, f + kf - 1), zero outside the
// map.  Input channels are [this group's newer dense outputs ..., block input]: the first hist_n live in `hist` (normalised + PReLU'd in place by
// k_zip_hist_norm once their producer's statistics were known), the last C in `inp`.
__device__ __forceinline__ float4 norm_prelu4(float4 v, const float* nrm2, const float* slope) {
    const float4 s0 = *reinterpret_cast<const float4*>(nrm2), s1 = *reinterpret_cast<const float4*>(nrm2 + 4), sl = *reinterpret_cast<const float4*>(slope);
    v.x = prelu_f(v.x * s0.x + s0.y, sl.x);
    v.y = prelu_f(v.y * s0.z + s0.w, sl.y);
    v.z = prelu_f(v.z * s1.x + s1.y, sl.z);
    v.w = prelu_f(v.w * s1.z + s1.w, sl.w);
    return v;
}
// in-place InstanceNorm + PReLU of one 64-channel block of the dense history ([tokens][ld], channels ch0 .. ch0 + 63) once its statistics are known: the block has up
// to three consumers with six taps each, and a loader that normalises on the fly has to wait for its loads BEFORE the slab's MFMAs (the arithmetic depends on them),
// which forfeits the fetch / MFMA overlap of the tile pipeline.  One extra 2 x 256 B per token here buys plain loads there.
__global__ __launch_bounds__(256) void k_zip_hist_norm(float* __restrict__ hist, int ld, int ch0, const float* __restrict__ nrm, const float* __restrict__ slope, int tok_per_win,
                                                       long long total16) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total16) return;
    const long long tok = i >> 4;
    const int ch = ch0 + (int)(i & 15) * 4, b = (int)(tok / tok_per_win);
    float4* p = reinterpret_cast<float4*>(hist + tok * ld + ch);
    *p = norm_prelu4(*p, nrm + ((size_t)b * ld + ch) * 2, slope + ch);
}
// The token-tiled kernel: the three frequency taps of a time tap share ONE staged operand.
// A generic implicit GEMM (one fetch per (token, tap, channel): the first round-2 form, 84 TFLOP/s) reads every (token, channel) six times.  Taps (kt, kf = 0..2) of one kt read the SAME 16-channel slab of the
// same rows shifted by -1 / 0 / +1 token, so this kernel stages rows m_blk - 1 .. m_blk + 256 of a (kt, channel block) once and runs the three kf products
// against it (the A operand of tap kf is LDS row r + kf; positions whose neighbour lies outside the map -- f = 0 for kf = 0, f = F - 1 for kf = 2 -- are zeroed
// in the operand register): a third of the global / L2 operand traffic, a third of the staging stores and barriers per MFMA.  256 tokens x 64 output
// channels per workgroup, wavefront tile 64 x 64, pipeline and LDS layout as in ade_gemm64.h.  k runs (kt, channel block, kf, channel) instead of
// (tap, channel): the same sum in another order.
// (round 5) Tiles never straddle a window -- grid = (blocks per window) x windows -- so the epilogue's per-channel sums of (product + bias) and of its square over the tile's rows
// ARE the layer's InstanceNorm partial sums: partial[((win * nblk + blk) * 64 + c) * 2 + {sum, sumsq}] (k_zip_stats_final's layout); the separate statistics pass over the
// output (k_zip_stats_partial: 12 launches, 2.6 ms per 128 x 1 s) is gone, as on the bf16 path.
__global__ __launch_bounds__(256, 3) void k_zip_dense(const float* hist, const float* __restrict__ inp, int hist_ld, int hist_off, int hist_n, int cin, int T, int F,
                                                      int dil, const float* __restrict__ w, const float* __restrict__ bias, float* out, int out_ld, int out_off,
                                                      double* __restrict__ partial, int nblk) {
    constexpr int kRowW = gemm::kRow, kARows = 264;
    __shared__ __attribute__((aligned(16))) float As[kARows * kRowW];
    __shared__ __attribute__((aligned(16))) float Bs[3 * 64 * kRowW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave * 64, j16 = lane & 15, g = lane >> 4;
    const int id = gemm::xcd_contiguous_id((int)blockIdx.x, (int)gridDim.x), win = id / nblk, blk = id - win * nblk;
    const int TF = T * F;
    const int wlo = win * TF, M = wlo + TF, m_blk = wlo + blk * 256;           // this tile's rows lie in [wlo, M): one window
    // staged rows of this lane: r = (tid >> 2) + 64 h, h < 4, and for tid < 8 the two halo rows 256, 257; token = m_blk - 1 + r
    const int sr = tid >> 2, kq = 4 * (tid & 3);
    long long stok[5];
    bool sok0[5], sok1[5];
#pragma unroll
    for (int h = 0; h < 5; ++h) {
        const int r = h < 4 ? sr + 64 * h : 256 + sr;
        const int m = m_blk - 1 + r;
        const bool in = m >= wlo && m < M && (h < 4 || tid < 8);
        const int mc = in ? m : wlo, t = (mc - wlo) / F;
        stok[h] = mc;
        sok1[h] = in;                               // kt = 1: the row itself
        sok0[h] = in && t >= dil;                   // kt = 0: the row dil frames earlier, inside the window
    }
    // operand masks of the rows this lane feeds to the matrix cores: bit i = row wm + 16 i + j16 has a left (kf = 0) / right (kf = 2) neighbour
    unsigned left = 0, right = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m_blk + wm + 16 * i + j16, f = m % F;
        if (f != 0) left |= 1u << i;
        if (f != F - 1) right |= 1u << i;
    }
    v4f acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = v4f{0.0f, 0.0f, 0.0f, 0.0f};
    const int ncb = cin / 16, nstage = 2 * ncb;
    float4 ra[5], rb0, rb1, rb2;
    auto fetch = [&](int st) {
        const int kt = st / ncb, ci0 = (st - kt * ncb) * 16;
        const bool from_hist = ci0 < hist_n;
        const float* src = from_hist ? hist + hist_off + ci0 + kq : inp + (ci0 - hist_n) + kq;
        const long long ld = from_hist ? hist_ld : 64, shift = kt ? 0 : (long long)dil * F;
#pragma unroll
        for (int h = 0; h < 5; ++h) {
            const bool ok = kt ? sok1[h] : sok0[h];
            ra[h] = ld4_or_zero(ok, src + (stok[h] - (ok ? shift : 0)) * ld);       // (zeros by address: the loads stay in flight across this stage's products)
        }
        const float* wp = w + (size_t)sr * (6 * cin) + (size_t)(kt * 3) * cin + ci0 + kq;
        rb0 = *reinterpret_cast<const float4*>(wp);
        rb1 = *reinterpret_cast<const float4*>(wp + cin);
        rb2 = *reinterpret_cast<const float4*>(wp + 2 * cin);
    };
    auto put4 = [&](float* base, int row, const float4& v) { *reinterpret_cast<float4*>(base + row * kRowW + kq) = v; };
    fetch(0);
    for (int st = 0; st < nstage; ++st) {
#pragma unroll
        for (int h = 0; h < 4; ++h) put4(As, sr + 64 * h, ra[h]);
        if (tid < 8) put4(As, 256 + sr, ra[4]);
        put4(Bs, sr, rb0);
        put4(Bs + 64 * kRowW, sr, rb1);
        put4(Bs + 2 * 64 * kRowW, sr, rb2);
        __syncthreads();
        if (st + 1 < nstage) fetch(st + 1);
#pragma unroll
        for (int kf = 0; kf < 3; ++kf) {
            const unsigned mask = kf == 0 ? left : (kf == 2 ? right : 0xfu);
            {
                float4 a4[4], b4[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) a4[i] = keep4((mask >> i) & 1u, *reinterpret_cast<const float4*>(As + (wm + 16 * i + j16 + kf) * kRowW + 4 * g));
#pragma unroll
                for (int j = 0; j < 4; ++j) b4[j] = *reinterpret_cast<const float4*>(Bs + (kf * 64 + 16 * j + j16) * kRowW + 4 * g);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc[i][j] = mfma16x16x4(a4[i].x, b4[j].x, acc[i][j]);
                        acc[i][j] = mfma16x16x4(a4[i].y, b4[j].y, acc[i][j]);
                        acc[i][j] = mfma16x16x4(a4[i].z, b4[j].z, acc[i][j]);
                        acc[i][j] = mfma16x16x4(a4[i].w, b4[j].w, acc[i][j]);
                    }
            }
        }
        __syncthreads();
    }
    // lane (g, j16), register q of tile (i, j) is C[wm + 16 i + 4 g + q][16 j + j16]
    // (the sums are fp64 from the first addend: a channel whose mean is large against its spread loses E[x^2] - mean^2 to fp32 round-off -- 1.5e-3 on the encoder tap, measured)
    float bj[4];
    double sj[4] = {0.0, 0.0, 0.0, 0.0}, qj[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int j = 0; j < 4; ++j) bj[j] = bias[16 * j + j16];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int m = m_blk + wm + 16 * i + 4 * g + q;
            if (m >= M) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float v = acc[i][j][q] + bj[j];
                out[(size_t)m * out_ld + out_off + 16 * j + j16] = v;
                sj[j] += (double)v;
                qj[j] = fma((double)v, (double)v, qj[j]);
            }
        }
    double (*red)[64][2] = reinterpret_cast<double (*)[64][2]>(Bs);        // [4][64][2]: the weights' buffer is dead (the stage loop ended on a barrier)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        double a = sj[j], b = qj[j];
        a += shfl_xor_f64(a, 16); a += shfl_xor_f64(a, 32);
        b += shfl_xor_f64(b, 16); b += shfl_xor_f64(b, 32);
        if (g == 0) { red[wave][16 * j + j16][0] = a; red[wave][16 * j + j16][1] = b; }
    }
    __syncthreads();
    if (tid < 128) {
        const int c = tid >> 1, q = tid & 1;
        partial[(((size_t)win * nblk + blk) * 64 + c) * 2 + q] = (red[0][c][q] + red[1][c][q]) + (red[2][c][q] + red[3][c][q]);
    }
}

struct BiasColStore {          // out[m * ld + off + n] = v + bias[n]
    static constexpr bool kCtx = true;
    float* out;
    const float* bias;
    int ld, off;
    __device__ gemm::None row(int) const { return gemm::None{}; }
    __device__ float col(int n) const { return bias[n]; }
    __device__ gemm::None pre(int, int, gemm::None) const { return gemm::None{}; }
    __device__ void operator()(int m, int n, float v, gemm::None, float b, gemm::None) const { out[(size_t)m * ld + off + n] = v + b; }
};
// (1, K3) convolution along f of a dense output (channel block `ch0` of hist, normalised in place): stride 2 / pad 1 for dense_conv_2 (:853), stride 1 /
// pad 1 for the sub-pixel up-sampler (:761-766).  A(token_out, k): k = kf * C + ci.
struct RowConvA {
    static constexpr int kWavesPerSimd = 3;
    const float* hist;
    int hist_ld, ch0, C, T, Fin, Fout, stride;
    struct Row { long long base; int f0; };
    __device__ Row row(int m) const {
        const int tf = T * Fout, b = m / tf, rem = m - b * tf, t = rem / Fout, f = rem - t * Fout;
        return Row{((long long)b * T + t) * Fin, f * stride - 1};
    }
    __device__ float4 vec4(const Row& r, int k) const {                  // branch-free: the padding positions read a block of zeros (no select on the loaded value)
        const int kf = k / C, ci = k - kf * C, f2 = r.f0 + kf;
        const bool ok = f2 >= 0 && f2 < Fin;
        const int ch = ch0 + ci;
        return ld4_or_zero(ok, hist + (size_t)(r.base + (ok ? f2 : 0)) * hist_ld + ch);
    }
};
struct SubPixelStore {         // conv channel n = c * r + u of sub-band f -> U[(b, t, f * r + u)][ch0 + c] (+ bias)   (:767-769)
    static constexpr bool kCtx = true;
    float* u;
    const float* bias;
    int ld, ch0, r;
    struct ColC { float b; int c, s; };
    __device__ gemm::None row(int) const { return gemm::None{}; }
    __device__ ColC col(int n) const { const int c = n / r; return ColC{bias[n], c, n - c * r}; }
    __device__ gemm::None pre(int, int, gemm::None) const { return gemm::None{}; }
    __device__ void operator()(int m, int, float v, gemm::None, const ColC& k, gemm::None) const { u[((size_t)m * r + k.s) * ld + ch0 + k.c] = v + k.b; }
};

// ---- Zipformer2 layer pieces ----------------------------------------------------------------------------------------------------
template <int KIND>            // activation applied while loading a row-major operand: 0 none, 1 SwooshL, 2 SwooshR
struct ActRowsA {
    const float* p;
    int ld;
    struct Row { const float* q; };
    __device__ Row row(int m) const { return Row{p + (size_t)m * ld}; }
    __device__ float4 vec4(const Row& r, int k) const { return *reinterpret_cast<const float4*>(r.q + k); }      // the load alone: it stays in flight across a slab's products
    __device__ static float4 finish(float4 v) {           // applied when the slab goes to LDS (k_gemm256x64)
        if (KIND == 1) { v.x = swoosh_l(v.x); v.y = swoosh_l(v.y); v.z = swoosh_l(v.z); v.w = swoosh_l(v.w); }
        if (KIND == 2) { v.x = swoosh_r(v.x); v.y = swoosh_r(v.y); v.z = swoosh_r(v.z); v.w = swoosh_r(v.w); }
        return v;
    }
};
// ---- fused feed-forward module (:160, :170-174): out = epilogue(W2 swooshL(W1 x + b1) + b2), the hidden activations never leave the CU ---------------------
// Both products are formed TRANSPOSED on v_mfma_f32_16x16x4_f32 (lane l supplies A[l & 15][l >> 4], B[l >> 4][l & 15] and holds D[4 (l >> 4) + r][l & 15]):
//   H^T tile (16 hidden x 16 rows) = W1 (16 x 64) . X^T :  lane (g, j) ends with hidden units 4 g + r of row j in its four accumulator registers;
//   Y^T tile (16 out x 16 rows)   += W2 (16 x 16 hidden) . act(H^T): contraction step s takes hidden 4 g' + s, so the B operand of lane (g, j) is its own register s --
// the hidden tile feeds the second product straight from the accumulators (bias + SwooshL applied in place): no LDS round trip, no M x fd tensor in HBM (the unfused
// form wrote and re-read 2 x 4 fd bytes per token; the module moves 512).  A 512-thread workgroup owns 256 rows (wavefront: two 16-row tiles, X in registers for the
// whole kernel); the weights stream through LDS 64 hidden units at a time (W1 rows and W2 columns of the chunk, 68-float pitch: conflict-free ds_read_b128),
// double-buffered with one barrier per chunk: 256 MFMAs per wavefront and chunk against 32 ds_read_b128.
// MODE 0: out = res + ff (feed_forward1: res = the layer input)   1: out = in + ff (in place)   2: out = res + ((in + ff) - res) * cmid (feed_forward2 + bypass_mid)
// MODE 3 (round 5): the layer's last module with its final norm (:175-183) in the same store: y = in + ff; out = y / |y - nb|_2 * fs + res * rs (res = the layer input, out = the
//         layer output, cmid = nb | fs | rs, 64 floats each); a row's 64 columns sit in the four lanes (g) that share its j16: the sum of squares meets through two shuffles.
constexpr int kFfChunk = 64, kFfPitch = 68, kFfRows = 256;
constexpr size_t kFfLds = (size_t)2 * 2 * kFfChunk * kFfPitch * sizeof(float);
template <int MODE>
__global__ __launch_bounds__(512) void k_zip_ff(const float* xin, const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ w2,
                                                const float* __restrict__ b2, const float* res, const float* __restrict__ cmid, float* out, int M, int fd) {
    HIP_DYNAMIC_SHARED(float, lds)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j16 = lane & 15, g = lane >> 4;
    const int row_base = (int)blockIdx.x * kFfRows + wave * 32;
    float4 xr[2][4];                                                        // X[row j16 of tile t][16 ks + 4 g ..]
    int row[2];
    bool rok[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        row[t] = row_base + 16 * t + j16;
        rok[t] = row[t] < M;
        const float* src = xin + (size_t)(rok[t] ? row[t] : 0) * 64 + 4 * g;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) xr[t][ks] = *reinterpret_cast<const float4*>(src + 16 * ks);
    }
    constexpr int kPitch = kFfPitch;                                        // 32-bit words per staged row
    v4f acc2[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) acc2[t][jt] = v4f{0.0f, 0.0f, 0.0f, 0.0f};
    // staging: chunk cg of W1 = rows 64 cg .. + 63 (64 k each), of W2 = columns 64 cg .. + 63 of its 64 rows: 1024 float4 each, two per thread
    const int sr = tid >> 4, sk = (tid & 15) * 4;                           // staged row (0..31, + 32), first of its four columns
    float4 p1a, p1b, p2a, p2b;            // (named, not arrays: as `float4 p1[2], p2[2]` written by one lambda and read by another they lived in scratch memory -- 80 bytes per lane, a
                                          //  scratch store + load per chunk on the kernel's critical path; found in round 5 with -Rpass-analysis=kernel-resource-usage)
    auto request = [&](int cg) {
        p1a = *reinterpret_cast<const float4*>(w1 + (size_t)(64 * cg + sr) * 64 + sk);
        p1b = *reinterpret_cast<const float4*>(w1 + (size_t)(64 * cg + sr + 32) * 64 + sk);
        p2a = *reinterpret_cast<const float4*>(w2 + (size_t)sr * fd + 64 * cg + sk);
        p2b = *reinterpret_cast<const float4*>(w2 + (size_t)(sr + 32) * fd + 64 * cg + sk);
    };
    auto deposit = [&](float* buf) {
        *reinterpret_cast<float4*>(buf + sr * kPitch + sk) = p1a;
        *reinterpret_cast<float4*>(buf + (sr + 32) * kPitch + sk) = p1b;
        *reinterpret_cast<float4*>(buf + (kFfChunk + sr) * kPitch + sk) = p2a;
        *reinterpret_cast<float4*>(buf + (kFfChunk + sr + 32) * kPitch + sk) = p2b;
    };
    const int ncg = fd / kFfChunk;
    request(0);
    deposit(lds);
    __syncthreads();
    for (int cg = 0; cg < ncg; ++cg) {
        const float* W1s = lds + (cg & 1) * (2 * kFfChunk * kPitch);
        const float* W2s = W1s + kFfChunk * kPitch;
        if (cg + 1 < ncg) request(cg + 1);
#pragma unroll
        for (int c = 0; c < 4; ++c) {                                       // 16 hidden units at a time
            v4f h[2] = {v4f{0.0f, 0.0f, 0.0f, 0.0f}, v4f{0.0f, 0.0f, 0.0f, 0.0f}};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                {
                    const float4 a = *reinterpret_cast<const float4*>(W1s + (16 * c + j16) * kPitch + 16 * ks + 4 * g);
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        h[t] = mfma16x16x4(a.x, xr[t][ks].x, h[t]);
                        h[t] = mfma16x16x4(a.y, xr[t][ks].y, h[t]);
                        h[t] = mfma16x16x4(a.z, xr[t][ks].z, h[t]);
                        h[t] = mfma16x16x4(a.w, xr[t][ks].w, h[t]);
                    }
                }
            }
            const float4 bb = *reinterpret_cast<const float4*>(b1 + 64 * cg + 16 * c + 4 * g);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                h[t][0] = swoosh_l(h[t][0] + bb.x); h[t][1] = swoosh_l(h[t][1] + bb.y);
                h[t][2] = swoosh_l(h[t][2] + bb.z); h[t][3] = swoosh_l(h[t][3] + bb.w);
            }
            {
#pragma unroll
                for (int jt = 0; jt < 4; ++jt) {
                    const float4 a = *reinterpret_cast<const float4*>(W2s + (16 * jt + j16) * kPitch + 16 * c + 4 * g);
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        acc2[t][jt] = mfma16x16x4(a.x, h[t][0], acc2[t][jt]);
                        acc2[t][jt] = mfma16x16x4(a.y, h[t][1], acc2[t][jt]);
                        acc2[t][jt] = mfma16x16x4(a.z, h[t][2], acc2[t][jt]);
                        acc2[t][jt] = mfma16x16x4(a.w, h[t][3], acc2[t][jt]);
                    }
                }
            }
        }
        if (cg + 1 < ncg) deposit(lds + ((cg + 1) & 1) * (2 * kFfChunk * kPitch));      // the other buffer: last read in iteration cg - 1, behind that iteration's barrier
        __syncthreads();
    }
    // lane (g, j16): output columns 16 jt + 4 g .. + 3 of row j16 of tile t
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        if (MODE == 3) {                                                    // (every lane takes part in the row sums: a row beyond M works on row 0 and stores nothing)
            const size_t at = (size_t)(rok[t] ? row[t] : 0) * 64 + 4 * g;
            float4 y[4], r4[4];
            float ssq = 0.0f;
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) {
                r4[jt] = *reinterpret_cast<const float4*>(res + at + 16 * jt);
                const float4 bo = *reinterpret_cast<const float4*>(b2 + 16 * jt + 4 * g), nb = *reinterpret_cast<const float4*>(cmid + 16 * jt + 4 * g);
                y[jt] = make_float4(xr[t][jt].x + (acc2[t][jt][0] + bo.x), xr[t][jt].y + (acc2[t][jt][1] + bo.y), xr[t][jt].z + (acc2[t][jt][2] + bo.z), xr[t][jt].w + (acc2[t][jt][3] + bo.w));
                const float dx = y[jt].x - nb.x, dy = y[jt].y - nb.y, dz = y[jt].z - nb.z, dw = y[jt].w - nb.w;
                ssq = fmaf(dx, dx, fmaf(dy, dy, fmaf(dz, dz, fmaf(dw, dw, ssq))));
            }
            ssq += __shfl_xor(ssq, 16, 64);
            ssq += __shfl_xor(ssq, 32, 64);
            const float nrm = sqrtf(ssq);
            if (rok[t]) {
#pragma unroll
                for (int jt = 0; jt < 4; ++jt) {
                    const float4 fs = *reinterpret_cast<const float4*>(cmid + 64 + 16 * jt + 4 * g), rs = *reinterpret_cast<const float4*>(cmid + 128 + 16 * jt + 4 * g);
                    *reinterpret_cast<float4*>(out + at + 16 * jt) = make_float4((y[jt].x / nrm) * fs.x + r4[jt].x * rs.x, (y[jt].y / nrm) * fs.y + r4[jt].y * rs.y,
                                                                                (y[jt].z / nrm) * fs.z + r4[jt].z * rs.z, (y[jt].w / nrm) * fs.w + r4[jt].w * rs.w);
                }
            }
            continue;
        }
        if (!rok[t]) continue;
        const size_t at = (size_t)row[t] * 64 + 4 * g;
        float4 rv[4], cv[4];
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
            rv[jt] = MODE != 1 ? *reinterpret_cast<const float4*>(res + at + 16 * jt) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            cv[jt] = MODE == 2 ? *reinterpret_cast<const float4*>(cmid + 16 * jt + 4 * g) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
            const float4 bo = *reinterpret_cast<const float4*>(b2 + 16 * jt + 4 * g);
            // this lane's slice of the input row, columns 16 jt + 4 g .. + 3, is xr[t][jt] (the same 16 ks + 4 g mapping)
            auto fin = [](float acc, float b, float xi, float r, float c) -> float {
                const float f = acc + b;
                if (MODE == 0) return r + f;
                if (MODE == 1) return xi + f;
                const float sum = xi + f;
                return r + (sum - r) * c;
            };
            const float4 o = make_float4(fin(acc2[t][jt][0], bo.x, xr[t][jt].x, rv[jt].x, cv[jt].x), fin(acc2[t][jt][1], bo.y, xr[t][jt].y, rv[jt].y, cv[jt].y),
                                         fin(acc2[t][jt][2], bo.z, xr[t][jt].z, rv[jt].z, cv[jt].z), fin(acc2[t][jt][3], bo.w, xr[t][jt].w, rv[jt].w, cv[jt].w));
            *reinterpret_cast<float4*>(out + at + 16 * jt) = o;
        }
    }
}

template <int MODE>
inline void launch_zip_ff(hipStream_t s, int M, const float* xin, const float* w1, const float* b1, const float* w2, const float* b2, const float* res, const float* cmid,
                          float* out, int fd) {
    const dim3 grid((unsigned)((M + kFfRows - 1) / kFfRows));
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_zip_ff<MODE>), grid, dim3(512), kFfLds, s, xin, w1, b1, w2, b2, res, cmid, out, M, fd);
}

struct AddFromStore {          // y[m][n] = x[m][n] + v + bias[n]     (the layer's first residual: x stays the layer input, :146, :160)
    static constexpr bool kCtx = true;
    const float* x;
    float* y;
    const float* bias;
    int ld;
    __device__ gemm::None row(int) const { return gemm::None{}; }
    __device__ float col(int n) const { return bias[n]; }
    __device__ float pre(int m, int n, gemm::None) const { return x[(size_t)m * ld + n]; }
    __device__ void operator()(int m, int n, float v, gemm::None, float b, float old) const { y[(size_t)m * ld + n] = old + (v + b); }
};
struct ResidualBiasStore {     // y[m][n] += v + bias[n]
    static constexpr bool kCtx = true;
    float* y;
    const float* bias;
    int ld;
    __device__ gemm::None row(int) const { return gemm::None{}; }
    __device__ float col(int n) const { return bias[n]; }
    __device__ float pre(int m, int n, gemm::None) const { return y[(size_t)m * ld + n]; }
    __device__ void operator()(int m, int n, float v, gemm::None, float b, float old) const { y[(size_t)m * ld + n] = old + (v + b); }
};
struct BypassMidStore {        // y = x0 + ((y + v + bias) - x0) * c     (feed_forward2's residual then bypass_mid, :170-171, :190-191)
    static constexpr bool kCtx = true;
    const float* x0;
    float* y;
    const float* bias;
    const float* c;
    int ld;
    __device__ gemm::None row(int) const { return gemm::None{}; }
    __device__ float2 col(int n) const { return make_float2(bias[n], c[n]); }
    __device__ float2 pre(int m, int n, gemm::None) const { const size_t i = (size_t)m * ld + n; return make_float2(y[i], x0[i]); }
    __device__ void operator()(int m, int n, float v, gemm::None, float2 bc, float2 yo) const {
        const float s = yo.x + (v + bc.x);
        y[(size_t)m * ld + n] = yo.y + (s - yo.y) * bc.y;
    }
};

// Relative-position attention (:232-301, :304-317), fused: scores, softmax and the weighted sum of the values in ONE kernel -- the (n, n) weights
// never leave the CU.  The layer uses the same weights three times (head 0 in NonlinAttention, all heads in both SelfAttention modules) on three
// DIFFERENT value tensors with residual updates in between, so each use recomputes them: 6 MFMAs per 16 x 16 score tile against streaming 0.4 MB
// of weights per sequence and use through HBM.
//   grid (sequence, head); 4 wavefronts; wave w takes the query tiles w, w + 4, ...; per query tile ALL key tiles' scores stay in registers.
// Everything is computed TRANSPOSED so that no operand changes lanes (v_mfma_f32_16x16x4_f32: lane l supplies A[l & 15][l >> 4], B[l >> 4][l & 15] and
// holds D[4 (l >> 4) + r][l & 15]; with one float4 per lane the four k-steps of a slab take k = 4 g + s on both operands):
//   S^T tile   = K_tile (16 keys x 16 dims) . Q^T (16 dims x 16 queries)            -> lane (g, j) holds S[query j][key 4 g + r]
//   pos term   : U^T (32 offsets x 16 queries) = P_h^T[c0 .. c0 + 31] (x 4 dims) . p^T (4 dims x 16 queries): TWO one-step MFMAs; the element that
//                belongs to (query j, key 4 g + r) is U^T[15 - j + 4 g + r][j] (out[i][j] = (p_i . P_h)[n - 1 - i + j], the reference's skew :270-284)
//                -- same lane column, other rows -- fetched through a per-wave LDS scratch
//   softmax    : per lane column (= query): max / sum over the lane's registers, then over the four lane groups (two shuffles)
//   O^T tile  += V^T (16 dims x 16 keys) . P^T (16 keys x 16 queries): the B operand is the lane's own exp() registers, V^T rows come from LDS as float4
// MODE 0 (NonlinAttention): value = tanh(s) * u of the (s | u | y) projection, head 0, DT = 3 tiles of 16 dims; result x y.
// MODE 1 (SelfAttention): value = head h's dv <= 16 columns of the value projection; DT = 1.
// NT = key tiles held in registers (n <= 16 NT).
constexpr int kQKs = 20;                    // floats per staged Q / K row (16 + 4: conflict-free ds_read_b128, as in ade_gemm.h)
// TI: element type of proj / src / out -- float (the parity path) or bf16 (ade_gemm_dtype = bf16: the operands are stored in HBM as bf16 and widened on their way into LDS /
// the registers; scores, softmax and accumulation are fp32 either way)
template <int MODE, int NT, int DT, class TI>
__global__ __launch_bounds__(256, (NT <= 11 ? 4 : (NT <= 20 ? 2 : 1))) void k_zip_attn(const TI* __restrict__ proj, int ldp, const float* __restrict__ pos, const TI* __restrict__ src, int lds_,
                                                  TI* __restrict__ out, int ldo, SeqGeo geo, int qd_, int pd_, int dv, int heads) {
    HIP_DYNAMIC_SHARED(float, lds)
    // workgroup -> (sequence, head): the heads of one sequence back to back on ONE XCD (consecutive workgroups go round-robin to the 8 XCDs; see k_zip_attn16)
    const int wg = (int)blockIdx.x, xcd = wg & 7, slot = wg >> 3, h = slot % heads, seq = (slot / heads) * 8 + xcd;
    if (seq >= geo.nseq) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = geo.n;
    const int j16 = lane & 15, g = lane >> 4;
    const int np16 = NT * 16, n2 = 2 * n - 1, vst = np16 + 4;
    float* Ks = lds;                               // [np16][kQKs]   (queries and their position projections are read straight from global memory by the wave that owns them:
    constexpr int kPtFront = 16, kPtRows = 2 * 16 * NT + 2 * kPtFront;          // table rows in LDS: offsets -16 .. 32 NT + 15, zeros outside [0, n2) (padded queries / keys index there)
    float* Pt = Ks + np16 * kQKs;                  // [kPtRows][4]: the four dims of an offset side by side, row kPtFront = offset 0  (keeping Q out of LDS is what lets a third /
    float* Vt = Pt + 4 * kPtRows;                  // [DT * 16][vst]   values, transposed                                              fourth workgroup share the CU)
    // (Round 5: the position term U[query][key] = sum_d p[query][d] table[key - query + n - 1][d] is four multiply-adds per score on the vector pipe -- a lane's four scores
    //  of a tile are four consecutive table rows, 64 contiguous bytes.  Before, U^T = table x p^T was two matrix instructions per tile, un-skewed through a per-wave LDS
    //  scratch: 8 writes + 4 reads per tile and 10 KB of LDS per workgroup.)
    (void)n2;
    const long long r0 = geo.row0(seq);
    // Staging.  Every loop below issues its global loads as one batch from in-range addresses (clamped position, zeroed afterwards) before it touches LDS: with a
    // branch around each load the compiler fences every single one (s_waitcnt vmcnt(0)) and the ~40 loads per lane cost ~40 memory latencies -- more than the 30 score
    // tiles per wavefront that follow.
    constexpr int hd = 36;                         // 2 * 16 + 4 (checked by the host: query_head_dim 16, pos_head_dim 4): a position's projection row is (16 q | 16 k | 4 p)
    {
        constexpr int kIt = (NT * 16 * 4 + 255) / 256;
        float4 t4[kIt];
#pragma unroll
        for (int u = 0; u < kIt; ++u) {
            const int i = tid + 256 * u, p = i >> 2, q = i & 3;
            const bool ok = p < n;                                                       // (p >= np16 only in the last, partial round: those lanes skip the store)
            t4[u] = keep4(ok, ldx4(proj + (size_t)(r0 + (long long)(ok ? p : 0) * geo.ps) * ldp + h * hd + 16 + 4 * q));
        }
#pragma unroll
        for (int u = 0; u < kIt; ++u) {
            const int i = tid + 256 * u, p = i >> 2, q = i & 3;
            if (p < np16) *reinterpret_cast<float4*>(Ks + p * kQKs + 4 * q) = t4[u];
        }
    }
    {   // position table (head, 4 dims, n2) -> [offset][4], zeros in front of offset 0 and behind offset n2 - 1
        constexpr int kIt = (kPtRows + 255) / 256;
        float t[kIt][4];
#pragma unroll
        for (int u = 0; u < kIt; ++u) {
            const int c = tid + 256 * u - kPtFront, cc = c >= 0 && c < n2 ? c : 0;
#pragma unroll
            for (int d = 0; d < 4; ++d) t[u][d] = pos[(size_t)(h * 4 + d) * n2 + cc];
        }
#pragma unroll
        for (int u = 0; u < kIt; ++u) {
            const int row = tid + 256 * u, c = row - kPtFront;
            if (row < kPtRows) *reinterpret_cast<float4*>(Pt + 4 * row) = c >= 0 && c < n2 ? make_float4(t[u][0], t[u][1], t[u][2], t[u][3]) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
    }
    {   // values, transposed into Vt[dim][key]: four dims per lane and load
        constexpr int kQuads = DT * 4, kIt = (NT * 16 * kQuads + 255) / 256, kBatch = 4;
#pragma unroll
        for (int u0 = 0; u0 < kIt; u0 += kBatch) {
            float4 va[kBatch], vb[kBatch];
#pragma unroll
            for (int u = 0; u < kBatch; ++u) {
                const int i = tid + 256 * (u0 + u), key = i / kQuads, d = (i - key * kQuads) * 4;
                const bool ok = key < n && d < dv;
                const TI* q = src + (size_t)(r0 + (long long)(key < n ? key : 0) * geo.ps) * lds_ + (ok ? d : 0);
                va[u] = keep4(ok, ldx4(MODE == 0 ? q : q + h * dv));
                if (MODE == 0) vb[u] = keep4(ok, ldx4(q + dv));
            }
#pragma unroll
            for (int u = 0; u < kBatch; ++u) {
                const int i = tid + 256 * (u0 + u), key = i / kQuads, d = (i - key * kQuads) * 4;
                if (u0 + u >= kIt || key >= np16) continue;
                float4 v = va[u];
                if (MODE == 0) v = make_float4(tanh_f(v.x) * vb[u].x, tanh_f(v.y) * vb[u].y, tanh_f(v.z) * vb[u].z, tanh_f(v.w) * vb[u].w);
                Vt[d * vst + key] = v.x; Vt[(d + 1) * vst + key] = v.y; Vt[(d + 2) * vst + key] = v.z; Vt[(d + 3) * vst + key] = v.w;
            }
        }
    }
    __syncthreads();
    for (int qt = wave; qt * 16 < n; qt += 4) {
        const int q0 = qt * 16, qi = q0 + j16;
        const TI* qrow = proj + (size_t)(r0 + (long long)(qi < n ? qi : 0) * geo.ps) * ldp + h * hd;
        const float4 qv = keep4(qi < n, ldx4(qrow + 4 * g));  // B operand of the score product: Q[query j16][dims 4 g ..]
        const float4 pq = ldx4(qrow + 32);                                               // p[query j16][dims 0 .. 3]
        const float* prow = Pt + 4 * (kPtFront + n - 1 - q0 - j16 + 4 * g);              // the lane's first table row of tile 0 (offset of (query q0 + j16, key 4 g)); a tile further on is 16 rows further on
        float st[NT][4];
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            const int k0 = kt * 16;
            const float4 kv = *reinterpret_cast<const float4*>(Ks + (k0 + j16) * kQKs + 4 * g);
            v4f sc = v4f{0.0f, 0.0f, 0.0f, 0.0f};
            sc = mfma16x16x4(kv.x, qv.x, sc);
            sc = mfma16x16x4(kv.y, qv.y, sc);
            sc = mfma16x16x4(kv.z, qv.z, sc);
            sc = mfma16x16x4(kv.w, qv.w, sc);
            float4 tr[4];                                      // table rows of keys k0 + 4 g + r, r = 0 .. 3
#pragma unroll
            for (int r = 0; r < 4; ++r) tr[r] = *reinterpret_cast<const float4*>(prow + 4 * (k0 + r));
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float u = fmaf(pq.w, tr[r].w, fmaf(pq.z, tr[r].z, fmaf(pq.y, tr[r].y, pq.x * tr[r].x)));     // the matrix instruction's order: dims 0 .. 3, one rounding per step
                st[kt][r] = sc[r] + u;
            }
            if (k0 + 16 > n) {              // (wave-uniform: only the tiles that hold padded keys pay the compare / select; the empty asm keeps it a branch)
                asm volatile("" ::: "memory");
#pragma unroll
                for (int r = 0; r < 4; ++r) st[kt][r] = (k0 + 4 * g + r < n) ? st[kt][r] : -INFINITY;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) mx = fmaxf(mx, st[kt][r]);
            ADE_OPAQUE_V(st[kt][0]); ADE_OPAQUE_V(st[kt][1]); ADE_OPAQUE_V(st[kt][2]); ADE_OPAQUE_V(st[kt][3]);      // (see k_zip_attn16: keeps a tile's work with the tile)
            if (kt & 1) __builtin_amdgcn_sched_barrier(0);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.0f;
        const float mb = -mx * kLog2e;
        v4f acc[DT];
#pragma unroll
        for (int d = 0; d < DT; ++d) acc[d] = v4f{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            float pr[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { pr[r] = __builtin_amdgcn_exp2f(fmaf(st[kt][r], kLog2e, mb)); sum += pr[r]; }          // hardware exp2 (~1 ulp), the subtraction inside the multiply-add; exp2(-inf) = 0 for the padded keys
#pragma unroll
            for (int d = 0; d < DT; ++d) {
                const float4 vv = *reinterpret_cast<const float4*>(Vt + (16 * d + j16) * vst + kt * 16 + 4 * g);   // V^T[dim 16 d + j16][keys k0 + 4 g ..]
                acc[d] = mfma16x16x4(vv.x, pr[0], acc[d]);
                acc[d] = mfma16x16x4(vv.y, pr[1], acc[d]);
                acc[d] = mfma16x16x4(vv.z, pr[2], acc[d]);
                acc[d] = mfma16x16x4(vv.w, pr[3], acc[d]);
            }
        }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        if (qi < n) {
            // lane (g, j16) holds O[query qi][dims 16 d + 4 g + r]
            const size_t row = (size_t)(r0 + (long long)qi * geo.ps);
#pragma unroll
            for (int d = 0; d < DT; ++d) {
                const int col = 16 * d + 4 * g;
                if (col >= dv) continue;
                float o[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = acc[d][r] / sum;
                if (MODE == 0) {
                    const float4 y = ldx4(src + row * lds_ + 2 * dv + col);
                    o[0] *= y.x; o[1] *= y.y; o[2] *= y.z; o[3] *= y.w;
                }
                stx4(out + row * ldo + (MODE == 0 ? 0 : h * dv) + col, make_float4(o[0], o[1], o[2], o[3]));
            }
        }
    }
}
template <int MODE, int NT, int DT>
inline size_t zip_attn_lds(int n) {
    const int np16 = NT * 16, n2 = 2 * n - 1;
    (void)n2;
    return ((size_t)np16 * kQKs + 4 * (size_t)(2 * 16 * NT + 32) + (size_t)DT * 16 * (np16 + 4)) * sizeof(float);
}

// ConvolutionModule core (:325-336): GLU then the depthwise Conv1d(k, padding k / 2) along the sequence.  grid (sequence, 64-position blocks);
// the block's gated rows (+ k / 2 halo on both sides) are staged in LDS once.  g: [rows][2 C] = (value | gate); out: [rows][C].
// PREGLU (the bf16 path, round 6): g holds C columns of IEEE half, the GLU already applied by the in-projection's store (zip16::k_rows16_chain<.., GLU> / GluStore16), and out
// is written as IEEE half too (its one reader, the out-projection's loader zip16::B16Rows<2, true>, unpacks it for SwooshR anyway)
template <int CC, int KK, class TI, bool PREGLU = false>       // CC, KK > 0: channel count / kernel size known at compile time (the published geometry: 64, 15); 0: run-time values; TI: float | bf16 storage of g / out
__global__ __launch_bounds__(256) void k_zip_dwconv(const TI* __restrict__ g, const float* __restrict__ w, const float* __restrict__ bias, TI* __restrict__ out,
                                                    SeqGeo geo, int C_, int K_) {
    HIP_DYNAMIC_SHARED(float, lds)
    const int C = CC > 0 ? CC : C_, K = KK > 0 ? KK : K_;
    const int seq = blockIdx.x, p0 = (int)blockIdx.y * 64, tid = threadIdx.x, n = geo.n, half = K / 2, rowsN = 64 + K - 1;
    const long long r0 = geo.row0(seq);
    if constexpr (CC == 64 && KK > 0) {
        // the published geometry: 16 lanes per staged row, four channels each; all of a lane's loads are issued as one batch from clamped rows (no branch around a
        // load: see k_zip_attn), the lane's 15 taps live in registers, and a lane owns ONE channel for its sixteen outputs
        constexpr int kRows = 64 + KK - 1, kIt = (kRows * 16 + 255) / 256;
        float4 va[kIt], vg[kIt];
#pragma unroll
        for (int u = 0; u < kIt; ++u) {
            const int i = tid + 256 * u, p = i >> 4, c = (i & 15) * 4, pos = p0 - half + p;
            const bool ok = p < kRows && pos >= 0 && pos < n;
            const TI* q = g + (size_t)(r0 + (long long)(ok ? pos : 0) * geo.ps) * (PREGLU ? CC : 2 * CC) + c;
            if constexpr (PREGLU) va[u] = keep4(ok, ldh4(q));
            else va[u] = keep4(ok, ldx4(q));
            if (!PREGLU) vg[u] = ldx4(q + CC);
        }
#pragma unroll
        for (int u = 0; u < kIt; ++u) {
            const int i = tid + 256 * u, p = i >> 4, c = (i & 15) * 4;
            if (p < kRows)
                *reinterpret_cast<float4*>(lds + p * CC + c) = PREGLU ? va[u] : make_float4(va[u].x * sigmoid_p(vg[u].x), va[u].y * sigmoid_p(vg[u].y), va[u].z * sigmoid_p(vg[u].z),
                                                                                             va[u].w * sigmoid_p(vg[u].w));
        }
        const int c = tid & 63;
        float wk[KK];
#pragma unroll
        for (int k = 0; k < KK; ++k) wk[k] = w[c * KK + k];
        const float bc = bias[c];
        __syncthreads();
#pragma unroll 4
        for (int it = 0; it < 16; ++it) {
            const int p = (tid >> 6) + 4 * it, pos = p0 + p;
            float a = 0.0f;
#pragma unroll
            for (int k = 0; k < KK; ++k) a = fmaf(wk[k], lds[(p + k) * CC + c], a);
            if (pos < n) {
                if constexpr (PREGLU) zip16::sth1(out + (size_t)(r0 + (long long)pos * geo.ps) * CC + c, a + bc);        // (PREGLU = the bf16 path: input and output are IEEE half)
                else stx1(out + (size_t)(r0 + (long long)pos * geo.ps) * CC + c, a + bc);
            }
        }
        return;
    }
    static_assert(!PREGLU || (CC == 64 && KK > 0), "PREGLU: the published geometry only");
    for (int i = tid; i < rowsN * C; i += 256) {
        const int p = i / C, c = i - p * C, pos = p0 - half + p;
        float v = 0.0f;
        if (pos >= 0 && pos < n) {
            const TI* q = g + (size_t)(r0 + (long long)pos * geo.ps) * (2 * C);
            v = ldx1(q + c) * sigmoid_p(ldx1(q + C + c));
        }
        lds[p * C + c] = v;
    }
    __syncthreads();
    for (int i = tid; i < 64 * C; i += 256) {
        const int p = i / C, c = i - p * C, pos = p0 + p;
        if (pos >= n) continue;
        float a = 0.0f;
        for (int k = 0; k < K; ++k) a = fmaf(w[c * K + k], lds[(p + k) * C + c], a);
        stx1(out + (size_t)(r0 + (long long)pos * geo.ps) * C + c, a + bias[c]);
    }
}

// x = y / |y - nb|_2 * fs + x * rs  (:175-183): 16 lanes per row (float4 each, C = 64) or a generic loop
__global__ __launch_bounds__(256) void k_zip_final_norm(float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ nb, const float* __restrict__ fs,
                                                        const float* __restrict__ rs, long long rows, int C) {
    const long long row = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int l = threadIdx.x & 15;
    float s = 0.0f;
    if (row < rows)
        for (int c = l; c < C; c += 16) { const float d = y[row * C + c] - nb[c]; s = fmaf(d, d, s); }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (row >= rows) return;
    const float nrm = sqrtf(s);
    for (int c = l; c < C; c += 16) x[row * C + c] = (y[row * C + c] / nrm) * fs[c] + x[row * C + c] * rs[c];
}

// SimpleDownsample over time then sub-bands (:194-218, :799-801): the tail group repeats the last frame / sub-band.  thread = (output token, channel quad): 16-byte accesses
// (one element per thread moved 4 bytes per lane and ran at 2.5 TB/s; the sums per element are unchanged)
__global__ __launch_bounds__(256) void k_zip_downsample(const float* __restrict__ x, float* __restrict__ y, const float* __restrict__ wt, const float* __restrict__ wf,
                                                        int T, int F, int dT, int dF, int dst, int dsf, int C, long long total4) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total4) return;
    const int cq = C >> 2, c = (int)(i % cq) * 4;
    long long tok = i / cq;
    const int f = (int)(tok % dF);
    tok /= dF;
    const int t = (int)(tok % dT), b = (int)(tok / dT);
    float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    for (int v = 0; v < dsf; ++v) {
        const int f2 = min(f * dsf + v, F - 1);
        float4 at = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        for (int u = 0; u < dst; ++u) {
            const int t2 = min(t * dst + u, T - 1);
            const float4 xv = *reinterpret_cast<const float4*>(x + (((size_t)b * T + t2) * F + f2) * C + c);
            const float w = wt[u];
            at.x += xv.x * w; at.y += xv.y * w; at.z += xv.z * w; at.w += xv.w * w;       // time first (:799) ...
        }
        const float w = wf[v];
        acc.x += at.x * w; acc.y += at.y * w; acc.z += at.z * w; acc.w += at.w * w;       // ... then sub-bands (:801)
    }
    *reinterpret_cast<float4*>(y + 4 * i) = acc;
}
// x = x * rs + up(y) * os  (:812-816); thread = (token, channel quad)
__global__ __launch_bounds__(256) void k_zip_upsample_combine(float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ os, const float* __restrict__ rs,
                                                              int T, int F, int dT, int dF, int dst, int dsf, int C, long long total4) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total4) return;
    const int cq = C >> 2, c = (int)(i % cq) * 4;
    long long tok = i / cq;
    const int f = (int)(tok % F);
    tok /= F;
    const int t = (int)(tok % T), b = (int)(tok / T);
    const float4 yv = *reinterpret_cast<const float4*>(y + (((size_t)b * dT + t / dst) * dF + f / dsf) * C + c);
    const float4 o = *reinterpret_cast<const float4*>(os + c), r = *reinterpret_cast<const float4*>(rs + c);
    float4* xp = reinterpret_cast<float4*>(x + 4 * i);
    const float4 xv = *xp;
    // Both products rounded, then the sum: the reference's three operations.  Under -ffp-contract=fast the back end fuses one product into the add whatever the source says
    // (__fmul_rn / __fadd_rn and `#pragma clang fp contract(off)` do not stop it); the opaque uses do.  Where the two terms cancel that is not a last-bit matter: with the fused
    // form the bf16 path's distance from the f32 path on the reference's speech clip drops from 40.2 to 37.9 dB (measured: the bisection is in DESIGN.md section 9).
    float4 a = make_float4(xv.x * r.x, xv.y * r.y, xv.z * r.z, xv.w * r.w), bq = make_float4(yv.x * o.x, yv.y * o.y, yv.z * o.z, yv.w * o.w);
    ADE_OPAQUE_V(a.x); ADE_OPAQUE_V(a.y); ADE_OPAQUE_V(a.z); ADE_OPAQUE_V(a.w);
    ADE_OPAQUE_V(bq.x); ADE_OPAQUE_V(bq.y); ADE_OPAQUE_V(bq.z); ADE_OPAQUE_V(bq.w);
    *xp = make_float4(a.x + bq.x, a.y + bq.y, a.z + bq.z, a.w + bq.w);
}

// decoder heads (:868, :874-893): (1, 2) convolutions over the normalised up-sampled maps, then relu(mask)^(1/0.3) x unit phase vector, stored
// planar for the synthesis GEMM: packed[c][j], c = re bin | 201 + im bin.  thread = (frame j, bin f).
__global__ __launch_bounds__(256) void k_zip_heads(const float* __restrict__ u, const float* __restrict__ nrm, const float* __restrict__ slope, const float* __restrict__ mw,
                                                   const float* __restrict__ mb, const float* __restrict__ pw, const float* __restrict__ pb, float* __restrict__ packed,
                                                   float* __restrict__ mask_tap, int T, int F2, int C, int J) {
    // 16 lanes per output (j, f): lane l holds channels 4 l .. 4 l + 3 of the mask half and of the phase half of rows f and f + 1 (512-byte rows read as whole
    // lines), the three dot products meet through four shuffle steps; a workgroup is 16 consecutive frames j of one bin f, so the (bin, frame) stores are 64-byte runs.
    // (One output per lane read a different row in every lane and stored 4 bytes a row pitch apart: 2.2 ms per 128 x 1 s against ~0.5 for this form.)  C = 64.
    const int f = blockIdx.y, j = (int)blockIdx.x * 16 + (threadIdx.x >> 4), l = threadIdx.x & 15, c = 4 * l;
    const bool live = j < J;
    const int jc = live ? j : J - 1, r = jc / T;
    float m = 0.0f, pr = 0.0f, pi = 0.0f;
    float4 ra[2], rb[2];
#pragma unroll
    for (int kf = 0; kf < 2; ++kf) {
        const float* row = u + ((size_t)jc * F2 + f + kf) * (2 * C);
        ra[kf] = *reinterpret_cast<const float4*>(row + c);
        rb[kf] = *reinterpret_cast<const float4*>(row + C + c);
    }
#pragma unroll
    for (int kf = 0; kf < 2; ++kf) {
        const float4 a = norm_prelu4(ra[kf], nrm + ((size_t)r * 2 * C + c) * 2, slope + c);
        const float4 b = norm_prelu4(rb[kf], nrm + ((size_t)r * 2 * C + C + c) * 2, slope + C + c);
        const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            m = fmaf(mw[(c + q) * 2 + kf], av[q], m);
            pr = fmaf(pw[(c + q) * 2 + kf], bv[q], pr);
            pi = fmaf(pw[(C + c + q) * 2 + kf], bv[q], pi);
        }
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
        m += __shfl_xor(m, o, 64);
        pr += __shfl_xor(pr, o, 64);
        pi += __shfl_xor(pi, o, 64);
    }
    if (l != 0 || !live) return;
    m += mb[0];
    pr += pb[0];
    pi += pb[1];
    if (mask_tap) mask_tap[(size_t)j * kZF + f] = m;
    const float mag = powf(fmaxf(m, 0.0f), 1.0f / 0.3f);                        // (:882-883)
    float pn = sqrtf(pr * pr + pi * pi);                                        // (:885)
    if (!(pn > 0.0f)) { pr = 1.0f; pi = 0.0f; pn = 1.0f; }                      // zero phase vector -> (1, 0) (:886-888)
    const float gain = mag / pn;                                                // (:891)
    packed[(size_t)f * J + j] = pr * gain;
    packed[(size_t)(kZF + f) * J + j] = pi * gain;
}
struct PlanarA {               // synthesis A operand: A(j, c) = packed[c][j]
    static constexpr bool kAlongK = false;
    const float* p;
    int J;
    __device__ float operator()(int j, int c) const { return p[(size_t)c * J + j]; }
};
// overlap-add gather, x 1 / sum w^2 (the static export's precomputed reciprocal, STFT_Process.py:245-249, :294-295) or / sum w^2 (divide = 1: the dynamic export builds
// the denominator from the frame count and divides, :297-299), x the window's norm factor, then the PCM tail (:893-918)
__global__ __launch_bounds__(256) void k_zip_ola_pcm(const float* __restrict__ frames, const float* __restrict__ inv_wsum, const float* __restrict__ norm,
                                                     int16_t* __restrict__ pcm, float* __restrict__ f32, int T, int L, int divide, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int r = (int)(i / L), m = (int)(i - (long long)r * L) + kZN / 2;
    int t_hi = m / kZHop;
    if (t_hi > T - 1) t_hi = T - 1;
    const int t_lo = m - kZN + 1 <= 0 ? 0 : (m - kZN + kZHop) / kZHop;
    float s = 0.0f;
    for (int t = t_lo; t <= t_hi; ++t) s += frames[((size_t)r * T + t) * kZN + (m - t * kZHop)];
    float y = (divide ? s / inv_wsum[m - kZN / 2] : s * inv_wsum[m - kZN / 2]) * norm[r];
    if (f32) f32[i] = y;
    if (pcm) {
        if (y != y) y = 0.0f;                                                   // NaN -> 0 (:917)
        pcm[i] = (short)(int)fminf(fmaxf(y, -32768.0f), 32767.0f);              // clamp, truncate (:918)
    }
}

// the fused attention kernel for this sequence length: NT = key tiles kept in registers (16 NT >= n)
template <int MODE, int NT, int DT, class TI>
bool launch_attn_nt(hipStream_t s, int heads, const TI* proj, int ldp, const float* pos, const TI* src, int lds_, TI* out, int ldo, SeqGeo geo, int dv) {
    if (geo.n > 16 * NT) return false;
    const size_t bytes = zip_attn_lds<MODE, NT, DT>(geo.n);
    auto kern = k_zip_attn<MODE, NT, DT, TI>;
    hipLaunchKernelGGL(kern, dim3((unsigned)(((geo.nseq + 7) / 8) * 8 * heads)), dim3(256), bytes, s, proj, ldp, pos, src, lds_, out, ldo, geo, 16, 4, dv, heads);
    return true;
}
template <int MODE, int DT, class TI>
void launch_attn(hipStream_t s, int heads, const TI* proj, int ldp, const float* pos, const TI* src, int lds_, TI* out, int ldo, SeqGeo geo, int dv) {
    (void)(launch_attn_nt<MODE, 4, DT, TI>(s, heads, proj, ldp, pos, src, lds_, out, ldo, geo, dv) || launch_attn_nt<MODE, 6, DT, TI>(s, heads, proj, ldp, pos, src, lds_, out, ldo, geo, dv) ||
           launch_attn_nt<MODE, 7, DT, TI>(s, heads, proj, ldp, pos, src, lds_, out, ldo, geo, dv) || launch_attn_nt<MODE, 11, DT, TI>(s, heads, proj, ldp, pos, src, lds_, out, ldo, geo, dv) ||
           launch_attn_nt<MODE, 16, DT, TI>(s, heads, proj, ldp, pos, src, lds_, out, ldo, geo, dv) || launch_attn_nt<MODE, 20, DT, TI>(s, heads, proj, ldp, pos, src, lds_, out, ldo, geo, dv) ||
           launch_attn_nt<MODE, 31, DT, TI>(s, heads, proj, ldp, pos, src, lds_, out, ldo, geo, dv));      // 496 keys: the reference's longest un-folded window (3 s = 481 frames) at 124 score registers, 157 KB of LDS
}

// ---- the same attention core on the bf16 matrix instruction (ade_gemm_dtype = bf16; v_mfma_f32_16x16x32_bf16: lane l supplies row (l & 15), k = 8 (l >> 4) .. + 7 of a 32-deep
// step; D as in the f32 form: lane (g, j16) holds D[4 g + r][j16]) -------------------------------------------------------------------------------------------------------
// Same decomposition, same skew, fp32 scores / softmax / accumulation.  The kernel is bound by the vector pipe (the first bf16 form spent 17 vector instructions per score
// against 1.5 matrix instructions per 64 scores, `profiles/r06_o_zip_bf16_kernel_stats.csv`), so everything that can ride in an instruction that is issued anyway does:
//   S^T tile   : ONE instruction (K rows from LDS as 16-byte pieces, Q^T from global).  The 16 head dims fill k = 0 .. 15; k = 16 carries the KEY MASK: a staged key row holds
//                0 (a key of the sequence) or -2^97 (padding) there and every query supplies 1, so padded keys leave the instruction at -1.6e29 and need no compare / select;
//   pos term   : a lane's four scores of a tile are four CONSECUTIVE offsets of the table (offset = key - query + n - 1): it reads its 32 bytes of the bf16 table
//                [offset][4 dims] and adds the two 2-term dot products with its own query's p (packed bf16 pairs as the projection stored them) onto the score with
//                v_dot2c_f32_bf16: two instructions per score (were four conversions + four multiply-adds + an add);
//   softmax    : exp2(s log2e - max log2e): one multiply-add + v_exp_f32; the row SUM comes out of the O^T product (SUMROW: a V^T row of ones in a spare value dim, i.e. the sum
//                of the bf16-rounded probabilities the product actually uses) instead of one add per score; one v_rcp_f32 per query instead of IEEE divisions;
//   O^T tile   : the probabilities of TWO key tiles are one B operand (the lane's own 2 x 4 registers, rounded to bf16), V^T comes from LDS as two 8-byte pieces (keys 4 g ..
//                of the even tile, 16 + 4 g .. of the odd one): one instruction per 32 keys and 16 value dims.  V^T is staged two keys at a time (4-byte LDS writes).
// Key tiles in steps of one (NT = 7 for the 101 sub-bands, 11 for the 161 frames: the last pair's second tile is a dummy of -inf scores).
__device__ __forceinline__ v4f zmfma16x16x32(const uint4& a, const uint4& b, v4f c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(gemm16::as_v8bf(a), gemm16::as_v8bf(b), c, 0, 0, 0); }
// c + a.lo * b.lo + a.hi * b.hi on packed bf16 pairs
__device__ __forceinline__ float dot2_bf16(unsigned a, unsigned b, float c) {
#if defined(__AMDGCN__)
    typedef __bf16 v2bf_t __attribute__((ext_vector_type(2)));
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(v2bf_t, a), __builtin_bit_cast(v2bf_t, b), c, false);
#else
    return fmaf(gemm16::bf16_hi(a), gemm16::bf16_hi(b), fmaf(gemm16::bf16_lo(a), gemm16::bf16_lo(b), c));        // (tests/hipsim)
#endif
}
__device__ __forceinline__ float exp2_raw(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float rcp_raw(float x) { return __builtin_amdgcn_rcpf(x); }
constexpr int kK16Pitch = 48;               // bytes per staged key row: 16 dims bf16 + the mask block (8 bf16: mask, 0 ...); the 16 lanes of a read land on 16 distinct 4-bank sets
constexpr unsigned kKeyMaskBf16 = 0xF000u;  // -2^97
// (the second launch bound keeps the register budget at <= 256 per lane, where the compiler writes the matrix instructions' results to ordinary registers: with the default
//  budget of 512 it parks every score tile in the accumulation file and copies it out, one v_accvgpr_read per score)
template <int MODE, int NT, int DT, bool SUMROW>
__global__ __launch_bounds__(256, (NT <= 12 ? 4 : 2)) void k_zip_attn16(const gemm16::bf16_t* __restrict__ proj, int ldp, const float* __restrict__ pos, const gemm16::bf16_t* __restrict__ src, int lds_,
                                                    gemm16::bf16_t* __restrict__ out, int ldo, SeqGeo geo, int dv, int geo_heads) {
    HIP_DYNAMIC_SHARED(unsigned char, lds8)
    typedef gemm16::bf16_t bf;
    constexpr int NP = (NT + 1) / 2, np32 = NP * 32, hd = 36;
    // workgroup -> (sequence, head): the hardware deals consecutive workgroups round-robin to the 8 XCDs; the heads of one sequence run back to back on ONE XCD (they read the
    // same projection rows: one trip to HBM for the four of them), sequences 8 apart follow each other there
    const int heads = geo_heads, wg = (int)blockIdx.x, xcd = wg & 7, slot = wg >> 3, h = slot % heads, seq = (slot / heads) * 8 + xcd;
    if (seq >= geo.nseq) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = geo.n;
    const int j16 = lane & 15, g = lane >> 4, n2 = 2 * n - 1;
    constexpr int vp = np32 * 2 + 16;                                                    // bytes per V^T row
    constexpr int kPtFront = 16, kPtRows = 2 * 16 * NT + 2 * kPtFront;                   // table rows in LDS: offsets -16 .. 32 NT + 15, zeros outside [0, n2) (padded queries / keys index there)
    unsigned char* Zs = lds8;                                                             // 16 bytes of zeros
    unsigned char* Ks = lds8 + 16;                                                        // [np32][kK16Pitch]
    unsigned char* Pt = Ks + np32 * kK16Pitch;                                           // [kPtRows][8]: bf16 x 4 dims per offset, row kPtFront = offset 0
    unsigned char* Vt = Pt + kPtRows * 8;                                                // [DT * 16][vp]
    const long long r0 = geo.row0(seq);
    if (tid < 4) reinterpret_cast<unsigned*>(Zs)[tid] = 0u;
    // staging: EVERY global load of the workgroup's operands is requested first (a workgroup lives for a few microseconds: three load -> LDS phases one after the other were
    // three round trips to HBM of its life), then the registers go to LDS as they arrive
    constexpr int kItK = (np32 * 2 + 255) / 256, kItT = (kPtRows + 255) / 256;
    constexpr int kQuads = DT * 4, kItems = (np32 / 2) * kQuads, kItV = (kItems + 255) / 256;
    uint4 t4[kItK];
    float tb[kItT][4];
    uint2 va[kItV][2], vg[kItV][2];
#pragma unroll
    for (int u = 0; u < kItK; ++u) {        // keys: two 16-byte pieces per key (rows beyond n are zeros)
        const int i = tid + 256 * u, p = i >> 1, q = i & 1;
        const bool ok = p < n;
        t4[u] = gemm16::ld8_or_zero(ok, proj + (size_t)(r0 + (long long)(ok ? p : 0) * geo.ps) * ldp + h * hd + 16 + 8 * q);
    }
#pragma unroll
    for (int u = 0; u < kItV; ++u) {        // values: an item = four dims of TWO consecutive keys
        const int i = tid + 256 * u, kp = i / kQuads, d = (i - kp * kQuads) * 4;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int key = 2 * kp + e;
            const bool ok = key < n && d < dv;
            const bf* q = src + (size_t)(r0 + (long long)(key < n ? key : 0) * geo.ps) * lds_ + (ok ? d : 0) + (MODE == 0 ? 0 : h * dv);
            va[u][e] = *reinterpret_cast<const uint2*>(ok ? reinterpret_cast<const void*>(q) : reinterpret_cast<const void*>(g_zero4));
            if (MODE == 0) vg[u][e] = *reinterpret_cast<const uint2*>(ok ? reinterpret_cast<const void*>(q + dv) : reinterpret_cast<const void*>(g_zero4));
        }
    }
#pragma unroll
    for (int u = 0; u < kItT; ++u) {        // position table (head, 4 dims, n2) fp32
        const int c = tid + 256 * u - kPtFront, cc = c >= 0 && c < n2 ? c : 0;
#pragma unroll
        for (int d = 0; d < 4; ++d) tb[u][d] = pos[(size_t)(h * 4 + d) * n2 + cc];
    }
    // a query tile's operands: B of the score product = Q[query j16][dims 8 g ..] (group 2: the mask's multiplier (1, 0 ...); group 3: zeros) and p[query j16][dims 0, 1 | 2, 3]
    uint4 qv_n;
    uint2 pq_n;
    auto load_q = [&](int qt) __attribute__((always_inline)) {
        const int qi = qt * 16 + j16;
        const bf* qrow = proj + (size_t)(r0 + (long long)(qi < n ? qi : 0) * geo.ps) * ldp + h * hd;
        qv_n = gemm16::ld8_or_zero(qi < n && g < 2, qrow + 8 * (g & 1));
        if (g == 2) qv_n.x = 0x3F80u;
        pq_n = *reinterpret_cast<const uint2*>(qrow + 32);
    };
    load_q(wave);
#pragma unroll
    for (int u = 0; u < kItK; ++u) {
        const int i = tid + 256 * u, p = i >> 1, q = i & 1;
        if (p < np32) *reinterpret_cast<uint4*>(Ks + p * kK16Pitch + 16 * q) = t4[u];
    }
    for (int p = tid; p < np32; p += 256) *reinterpret_cast<uint4*>(Ks + p * kK16Pitch + 32) = make_uint4(p < n ? 0u : kKeyMaskBf16, 0u, 0u, 0u);          // the mask block
#pragma unroll
    for (int u = 0; u < kItV; ++u) {        // -> Vt[dim][key] (bf16), four 4-byte writes (key pair) per item; keys beyond n are zeros; SUMROW: dim dv is all ones
        const int i = tid + 256 * u, kp = i / kQuads, d = (i - kp * kQuads) * 4;
        unsigned w[4];
        if (MODE == 0) {                    // (:310-316) tanh(s) * x
            float a0[4], a1[4], g0[4], g1[4];
            a0[0] = gemm16::bf16_lo(va[u][0].x); a0[1] = gemm16::bf16_hi(va[u][0].x); a0[2] = gemm16::bf16_lo(va[u][0].y); a0[3] = gemm16::bf16_hi(va[u][0].y);
            a1[0] = gemm16::bf16_lo(va[u][1].x); a1[1] = gemm16::bf16_hi(va[u][1].x); a1[2] = gemm16::bf16_lo(va[u][1].y); a1[3] = gemm16::bf16_hi(va[u][1].y);
            g0[0] = gemm16::bf16_lo(vg[u][0].x); g0[1] = gemm16::bf16_hi(vg[u][0].x); g0[2] = gemm16::bf16_lo(vg[u][0].y); g0[3] = gemm16::bf16_hi(vg[u][0].y);
            g1[0] = gemm16::bf16_lo(vg[u][1].x); g1[1] = gemm16::bf16_hi(vg[u][1].x); g1[2] = gemm16::bf16_lo(vg[u][1].y); g1[3] = gemm16::bf16_hi(vg[u][1].y);
#pragma unroll
            for (int e = 0; e < 4; ++e) w[e] = gemm16::pack_bf16x2(tanh_f(a0[e]) * g0[e], tanh_f(a1[e]) * g1[e]);
        } else {                            // values as stored: a shuffle of 16-bit halves
            w[0] = (va[u][0].x & 0xffffu) | (va[u][1].x << 16);
            w[1] = (va[u][0].x >> 16) | (va[u][1].x & 0xffff0000u);
            w[2] = (va[u][0].y & 0xffffu) | (va[u][1].y << 16);
            w[3] = (va[u][0].y >> 16) | (va[u][1].y & 0xffff0000u);
            if (SUMROW && d == dv) w[0] = 0x3F803F80u;
        }
        if (i < kItems) {
            unsigned char* col = Vt + (size_t)d * vp + kp * 4;
            *reinterpret_cast<unsigned*>(col) = w[0];
            *reinterpret_cast<unsigned*>(col + vp) = w[1];
            *reinterpret_cast<unsigned*>(col + 2 * vp) = w[2];
            *reinterpret_cast<unsigned*>(col + 3 * vp) = w[3];
        }
    }
#pragma unroll
    for (int u = 0; u < kItT; ++u) {        // -> [offset][4] bf16, zeros in front of offset 0 and behind offset n2 - 1
        const int row = tid + 256 * u, c = row - kPtFront;
        if (row < kPtRows)
            *reinterpret_cast<uint2*>(Pt + row * 8) = c >= 0 && c < n2 ? make_uint2(gemm16::pack_bf16x2(tb[u][0], tb[u][1]), gemm16::pack_bf16x2(tb[u][2], tb[u][3])) : make_uint2(0u, 0u);
    }
    __syncthreads();
    const int sum_lane = 16 * ((dv & 15) >> 2) + j16, sum_reg = dv & 3;                         // SUMROW: value dim dv of the last 16-dim tile = D row dv & 15
    for (int qt = wave; qt * 16 < n; qt += 4) {
        const int q0 = qt * 16, qi = q0 + j16;
        const uint4 qv = qv_n;
        const uint2 pq = pq_n;
        if ((qt + 4) * 16 < n) load_q(qt + 4);                                                     // the next tile's query rows travel during this tile
        // the lane's first table row of tile 0: offset of (query q0 + j16, key 4 g) = n - 1 - (q0 + j16) + 4 g; a tile further on is 16 rows further on
        const unsigned char* prow = Pt + (kPtFront + n - 1 - q0 - j16 + 4 * g) * 8;
        const unsigned char* krow = g < 3 ? Ks + j16 * kK16Pitch + 16 * g : Zs;
        const int kstep = g < 3 ? 16 * kK16Pitch : 0;
        float st[2 * NP][4];
        float mx = -INFINITY;
        v4f sc_n = zmfma16x16x32(*reinterpret_cast<const uint4*>(krow), qv, v4f{0.0f, 0.0f, 0.0f, 0.0f});
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            const int k0 = kt * 16;
            const v4f sc = sc_n;                               // (the next tile's score product is issued before this tile's vector work: its latency passes there)
            if (kt + 1 < NT) sc_n = zmfma16x16x32(*reinterpret_cast<const uint4*>(krow + (kt + 1) * kstep), qv, v4f{0.0f, 0.0f, 0.0f, 0.0f});
            uint2 tr[4];                                       // table rows of keys k0 + 4 g + r, r = 0 .. 3: 32 consecutive bytes
#pragma unroll
            for (int r = 0; r < 4; ++r) tr[r] = *reinterpret_cast<const uint2*>(prow + (k0 + r) * 8);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                st[kt][r] = dot2_bf16(tr[r].y, pq.y, dot2_bf16(tr[r].x, pq.x, sc[r]));
                mx = fmaxf(mx, st[kt][r]);
            }
            // (the tiles are independent: left alone, the compiler requests every tile's table rows up front and sinks the dot products to the softmax below -- 256 registers at
            //  NT = 12, one wavefront per SIMD.  The opaque uses pin a tile's four scores where they are written; the fence keeps the next tiles' loads behind them.)
            ADE_OPAQUE_V(st[kt][0]); ADE_OPAQUE_V(st[kt][1]); ADE_OPAQUE_V(st[kt][2]); ADE_OPAQUE_V(st[kt][3]);
            if (kt & 1) __builtin_amdgcn_sched_barrier(0);
        }
        if (NT & 1) { st[NT][0] = st[NT][1] = st[NT][2] = st[NT][3] = -INFINITY; }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mb = -mx * kLog2e;
        float sum = 0.0f;
        v4f acc[DT];
#pragma unroll
        for (int d = 0; d < DT; ++d) acc[d] = v4f{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            float pr[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) { pr[r] = exp2_raw(fmaf(st[2 * p][r], kLog2e, mb)); pr[4 + r] = exp2_raw(fmaf(st[2 * p + 1][r], kLog2e, mb)); }
            if (!SUMROW) {
#pragma unroll
                for (int r = 0; r < 8; ++r) sum += pr[r];
            }
            const uint4 pb = make_uint4(gemm16::pack_bf16x2(pr[0], pr[1]), gemm16::pack_bf16x2(pr[2], pr[3]), gemm16::pack_bf16x2(pr[4], pr[5]), gemm16::pack_bf16x2(pr[6], pr[7]));
#pragma unroll
            for (int d = 0; d < DT; ++d) {
                const unsigned char* vr = Vt + (size_t)(16 * d + j16) * vp + (32 * p + 4 * g) * 2;             // V^T[dim 16 d + j16][keys 32 p + 4 g .. | 32 p + 16 + 4 g ..]
                const uint2 va = *reinterpret_cast<const uint2*>(vr), vb = *reinterpret_cast<const uint2*>(vr + 32);
                acc[d] = zmfma16x16x32(make_uint4(va.x, va.y, vb.x, vb.y), pb, acc[d]);
            }
        }
        if (SUMROW) {
            const v4f a = acc[DT - 1];
            const float sv = sum_reg == 0 ? a[0] : (sum_reg == 1 ? a[1] : (sum_reg == 2 ? a[2] : a[3]));
            sum = __shfl(sv, sum_lane, 64);
        } else {
            sum += __shfl_xor(sum, 16, 64);
            sum += __shfl_xor(sum, 32, 64);
        }
        const float inv = rcp_raw(sum);
        if (qi < n) {
            const size_t row = (size_t)(r0 + (long long)qi * geo.ps);
#pragma unroll
            for (int d = 0; d < DT; ++d) {
                const int col = 16 * d + 4 * g;
                if (col >= dv) continue;
                float o[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = acc[d][r] * inv;
                if (MODE == 0) {
                    const float4 y = ldx4(src + row * lds_ + 2 * dv + col);
                    o[0] *= y.x; o[1] *= y.y; o[2] *= y.z; o[3] *= y.w;
                }
                stx4(out + row * ldo + (MODE == 0 ? 0 : h * dv) + col, make_float4(o[0], o[1], o[2], o[3]));
            }
        }
    }
}
template <int MODE, int NT, int DT>
inline size_t zip_attn16_lds(int n) {
    constexpr int NP = (NT + 1) / 2, np32 = NP * 32;
    (void)n;
    return 16 + (size_t)np32 * kK16Pitch + (size_t)(2 * 16 * NT + 32) * 8 + (size_t)DT * 16 * (np32 * 2 + 16);
}
template <int MODE, int NT, int DT>
bool launch_attn16_nt(hipStream_t s, int heads, const gemm16::bf16_t* proj, int ldp, const float* pos, const gemm16::bf16_t* src, int lds_, gemm16::bf16_t* out, int ldo, SeqGeo geo, int dv) {
    if (geo.n > 16 * NT) return false;
    const size_t bytes = zip_attn16_lds<MODE, NT, DT>(geo.n);
    const dim3 grid((unsigned)(((geo.nseq + 7) / 8) * 8 * heads));
    // a spare value dim in the last 16-dim tile carries the row sums (dv % 4 == 0 keeps the dim's quad apart from the values')
    if (dv < 16 * DT && dv > 16 * (DT - 1) && (dv & 3) == 0)
        hipLaunchKernelGGL((k_zip_attn16<MODE, NT, DT, true>), grid, dim3(256), bytes, s, proj, ldp, pos, src, lds_, out, ldo, geo, dv, heads);
    else
        hipLaunchKernelGGL((k_zip_attn16<MODE, NT, DT, false>), grid, dim3(256), bytes, s, proj, ldp, pos, src, lds_, out, ldo, geo, dv, heads);
    return true;
}
// windows of up to 256 frames / sub-bands take the bf16-instruction kernel (at most 64 KB of LDS, 64 score registers); longer ones the fp32-instruction kernel on bf16 storage
template <int MODE, int DT>
void launch_attn16(hipStream_t s, int heads, const gemm16::bf16_t* proj, int ldp, const float* pos, const gemm16::bf16_t* src, int lds_, gemm16::bf16_t* out, int ldo, SeqGeo geo, int dv) {
    if (launch_attn16_nt<MODE, 4, DT>(s, heads, proj, ldp, pos, src, lds_, out, ldo, geo, dv) || launch_attn16_nt<MODE, 6, DT>(s, heads, proj, ldp, pos, src, lds_, out, ldo, geo, dv) ||
        launch_attn16_nt<MODE, 7, DT>(s, heads, proj, ldp, pos, src, lds_, out, ldo, geo, dv) || launch_attn16_nt<MODE, 8, DT>(s, heads, proj, ldp, pos, src, lds_, out, ldo, geo, dv) ||
        launch_attn16_nt<MODE, 11, DT>(s, heads, proj, ldp, pos, src, lds_, out, ldo, geo, dv) || launch_attn16_nt<MODE, 12, DT>(s, heads, proj, ldp, pos, src, lds_, out, ldo, geo, dv) ||
        launch_attn16_nt<MODE, 16, DT>(s, heads, proj, ldp, pos, src, lds_, out, ldo, geo, dv)) return;
    launch_attn<MODE, DT, gemm16::bf16_t>(s, heads, proj, ldp, pos, src, lds_, out, ldo, geo, dv);
}
template <int MODE, int DT>
hipError_t raise_attn16_lds() {
    hipError_t e = hipSuccess;
    auto one = [&](auto kern) { if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); };
    one(k_zip_attn16<MODE, 4, DT, true>); one(k_zip_attn16<MODE, 6, DT, true>); one(k_zip_attn16<MODE, 7, DT, true>); one(k_zip_attn16<MODE, 8, DT, true>); one(k_zip_attn16<MODE, 11, DT, true>);
    one(k_zip_attn16<MODE, 12, DT, true>); one(k_zip_attn16<MODE, 16, DT, true>);
    one(k_zip_attn16<MODE, 4, DT, false>); one(k_zip_attn16<MODE, 6, DT, false>); one(k_zip_attn16<MODE, 7, DT, false>); one(k_zip_attn16<MODE, 8, DT, false>); one(k_zip_attn16<MODE, 11, DT, false>);
    one(k_zip_attn16<MODE, 12, DT, false>); one(k_zip_attn16<MODE, 16, DT, false>);
    return e;
}
template <int MODE, int NT, int DT, class TI>
hipError_t raise_one() {
    auto kern = k_zip_attn<MODE, NT, DT, TI>;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}
template <int MODE, int DT, class TI>
hipError_t raise_attn_lds() {         // the long-window instantiations need more than the default 64 KB of dynamic LDS
    hipError_t e = raise_one<MODE, 11, DT, TI>();
    if (e == hipSuccess) e = raise_one<MODE, 16, DT, TI>();
    if (e == hipSuccess) e = raise_one<MODE, 20, DT, TI>();
    if (e == hipSuccess) e = raise_one<MODE, 31, DT, TI>();
    return e;
}

int zfail(std::string& err, int st, const std::string& msg) { err = msg; return st; }
#define ZP_HIP(expr)                                                                                  \
    do {                                                                                              \
        hipError_t _e = (expr);                                                                       \
        if (_e != hipSuccess) return zfail(err, ADE_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

struct ZLayer {
    const float *attn_ff1_w, *attn_ff1_b, *pos, *ff1_out_w, *ff1_out_b, *nonlin_in_w, *nonlin_in_b, *nonlin_out_w, *nonlin_out_b;
    const float *sa_in_w[2], *sa_in_b[2], *sa_out_w[2], *sa_out_b[2];
    const float *cv_in_w[2], *cv_in_b[2], *cv_dw_w[2], *cv_dw_b[2], *cv_out_w[2], *cv_out_b[2];
    const float *ff_in_w[2], *ff_in_b[2], *ff_out_w[2], *ff_out_b[2];      // feed_forward2, feed_forward3
    const float *bypass_mid, *norm_bias, *fnorm, *fres;
};
struct ZLayer16 {              // bf16 copies of a layer's matrix weights (ade_gemm_dtype = bf16; csrc/ade_zip16.h)
    const gemm16::bf16_t *ff1_w1, *ff1_w2p;                // the ff1 rows of attn_ff1_w, ff1_out_w with its hidden units in k_zip_ff16's order
    const gemm16::bf16_t *ff_w1[2], *ff_w2p[2];            // feed_forward2, feed_forward3
    const gemm16::bf16_t *attn_w, *nonlin_in_w, *nonlin_out_w, *sa_in_w[2], *sa_out_w[2], *cv_in_w[2], *cv_out_w[2];      // (N, K) row-major as the fp32 tensors
};
struct ZDense {                // one causal dense block: per layer the repacked weights [co][tap][ci] (per group), bias, norm affine; slopes per hist channel
    const gemm16::bf16_t* w16[2][8];                       // the same repacked weights in bf16 (bf16 path)
    const float* w[2][8];
    const float* b[2][8];
    const float* gamma[2][8];
    const float* beta[2][8];
    const float* slope;        // [groups * depth * C] in hist-channel order
};

}  // namespace

struct ZipEngine : SubEngine {
    int device = 0, L = 0 /* samples per window in */, Lo = 0 /* out: whole hops, hop * (T - 1) */, n_win = 1, T = 0, F = 0;
    int C = 64, H = 4, qd = 16, pd = 4, vd = 12, pos_dim = 48, ffd = 256, K = 15, dst = 2, dsf = 2, up = 2, depth = 4;
    int hid = 48, ff1 = 192, ff3 = 320, attn_dim = 144, dT = 0, dF = 0;
    bool exact = false;
    float* d_w = nullptr;
    const float *k_fwd = nullptr, *k_inv = nullptr, *inv_wsum = nullptr;
    bool dynamic_norm = false;         // a DYNAMIC_AXES export: inv_wsum holds sum w^2 itself and the overlap-add divides
    const float *c1_w = nullptr, *c1_b = nullptr, *c1_g = nullptr, *c1_beta = nullptr, *c1_slope = nullptr;
    const float *c2_w = nullptr, *c2_b = nullptr, *c2_g = nullptr, *c2_beta = nullptr, *c2_slope = nullptr;
    ZDense enc_dense{}, dec_dense{};
    ZLayer layers[4][2]{};
    // ade_gemm_dtype = bf16 (BASELINE.json configs[2]'s dtype): bf16 weights / activations stored in HBM for the dense blocks and the feed-forward modules (csrc/ade_zip16.h)
    bool bf16 = false;
    gemm16::bf16_t* d_w16 = nullptr;
    ZLayer16 layers16[4][2]{};
    gemm16::bf16_t* ws16 = nullptr;
    gemm16::bf16_t *Dh16 = nullptr, *E016 = nullptr, *X16 = nullptr;     // dense history, dense-encoder input, the decoder pair's dense-block input
    gemm16::bf16_t *P16 = nullptr, *S16 = nullptr, *O16 = nullptr;       // a layer's attention projection (q | k | p per head), module projection, module core output
    const gemm16::bf16_t *c2_w16 = nullptr, *up_w16[2] = {};
    float* raw = nullptr;                                                 // a dense layer's raw fp32 output (+ bias)
    bool zip_chain = !(getenv("ADE_ZIP_CHAIN") && atoi(getenv("ADE_ZIP_CHAIN")) == 0);
    bool zip_fuse = !(getenv("ADE_ZIP_FUSE") && atoi(getenv("ADE_ZIP_FUSE")) == 0);      // bf16 path: a feed-forward module and the row-local projections around it in one launch (k_zip_ffx); 0: separate launches, same bits
    int dense_cb = getenv("ADE_ZIP_DENSE_CB") ? atoi(getenv("ADE_ZIP_DENSE_CB")) : 32;      // input channels per stage of k_zip_dense16 (measurement knob)
    // Error-budget knob of the bf16 path (tools/zip_bf16_budget.py): which parts of a bf16 handle run on bf16 operands -- bit 0 the dense encoder block, bit 1 the eight
    // Zipformer layers, bit 2 the decoder pair's dense block + sub-pixel convolution.  7 (default) = the bf16 path; a cleared bit runs that part's f32 kernels instead.
    int parts16 = getenv("ADE_ZIP16_PARTS") ? (atoi(getenv("ADE_ZIP16_PARTS")) & 7) : 7;
    // The 16-bit type of the three causal dense blocks (history, block input, weights, the (1, 3) convolutions that read the history): IEEE half by default -- the same
    // matrix rate and bytes as bf16, three more mantissa bits, and these tensors are InstanceNorm'd activations of order one (the reference's own reduced-precision plan is
    // fp16: Optimize_ONNX.py:25-64).  BASELINE configs[2]'s "bf16 dual-path transformer" stays bf16.  ADE_ZIP_DENSE_F16=0 keeps bf16 there too (the budget's comparison leg).
    bool dense_half = !(getenv("ADE_ZIP_DENSE_F16") && atoi(getenv("ADE_ZIP_DENSE_F16")) == 0);
    // ADE_ZIP_LAYER_TAPS=1: the residual stream after every sub-module of the FIRST layer (encoder 0, frequency path) is kept for taps "l0_0" .. "l0_7"
    // (ff1, nonlin-attention, self-attention 1, convolution 1, ff2 + bypass, self-attention 2, convolution 2, ff3 + final norm); calls of at most 8 windows
    bool want_layer_taps = getenv("ADE_ZIP_LAYER_TAPS") && atoi(getenv("ADE_ZIP_LAYER_TAPS")) == 1;
    float* lt = nullptr;
    bool lt_armed = false;
    size_t lt_stride = 0;
    void layer_tap(hipStream_t s, int k, const float* src, long long R) {
        if (lt && lt_armed && (size_t)R * C <= lt_stride) (void)hipMemcpyAsync(lt + (size_t)k * lt_stride, src, (size_t)R * C * sizeof(float), hipMemcpyDeviceToDevice, s);
    }
    const float *down_t[4] = {}, *down_f[4] = {}, *out_scale[4] = {}, *res_scale[4] = {};
    const float *up_w[2] = {}, *up_b[2] = {}, *up_g = nullptr, *up_beta = nullptr, *up_slope = nullptr;
    const float *mask_w = nullptr, *mask_b = nullptr, *phase_w = nullptr, *phase_b = nullptr;
    int capacity = 0;
    float* ws = nullptr;
    double* partial = nullptr;
    float *norm = nullptr, *spec = nullptr, *feat = nullptr, *coef = nullptr, *E0 = nullptr, *Dh = nullptr, *nrm = nullptr, *nrm2 = nullptr, *X = nullptr, *Y = nullptr, *X2 = nullptr,
          *P = nullptr, *S1 = nullptr, *O = nullptr, *U = nullptr, *packed = nullptr, *frames_buf = nullptr, *mask_tap = nullptr, *enc_tap[5] = {};
    bool keep_taps = false;

    ~ZipEngine() override {
        (void)hipSetDevice(device);
        if (d_w) (void)hipFree(d_w);
        if (d_w16) (void)hipFree(d_w16);
        if (ws16) (void)hipFree(ws16);
        if (ws) (void)hipFree(ws);
        if (partial) (void)hipFree(partial);
    }
    int frames() const override { return T; }
    int in_len() const override { return L * n_win; }
    int out_len() const override { return Lo * n_win; }
    bool accepts_float_input() const override { return true; }
    int reserve(int batch, std::string& err) override;
    int run(hipStream_t s, const int16_t* d_in, int batch, int16_t* d_out, float* d_f32, std::string& err) override;
    int tap(hipStream_t s, const char* name, int batch, float* out, size_t count, size_t* written, std::string& err) override;

    void stats(hipStream_t s, const float* x, int ld, int ch0, int tok_per_win, int windows, const float* gamma, const float* beta, float* nrm_, int nrm_ld, int nrm_ch0);
    void dense_block(hipStream_t s, const ZDense& d, int groups, const float* inp, int windows, int Fd);
    void dense_block16(hipStream_t s, const ZDense& d, int groups, const gemm16::bf16_t* inp, int windows, int Fd);
    void layer(hipStream_t s, const ZLayer& w, const ZLayer16& w16, float* x, long long R, SeqGeo geo);
    void layer16(hipStream_t s, const ZLayer& w, const ZLayer16& w16, float* x, long long R, SeqGeo geo);
    void attention(hipStream_t s, int mode, const float* pos, const float* src, int lds_, float* out, int ldo, SeqGeo geo, int dv);
    void dualpath(hipStream_t s, int e, float* x, int B, int Tt, int Ff);
};

int zipenhancer_create(const std::map<std::string, Tensor>& tensors, int window_len, int n_win, bool exact_dft, bool bf16, bool dynamic, int device, SubEngine** out, std::string& err) {
    *out = nullptr;
    if (n_win < 1) return zfail(err, ADE_ERR_BAD_VALUE, "zipenhancer: n_win must be >= 1");
    if (window_len < kZN || (n_win > 1 && window_len % kZHop))
        return zfail(err, ADE_ERR_SHAPE_MISMATCH, "zipenhancer: the window length must be at least 400 samples (a fold window a multiple of the 100-sample hop)");
    auto find = [&](const std::string& name) -> const Tensor* {
        auto it = tensors.find(name);
        if (it == tensors.end()) { if (err.empty()) err = "weights: tensor missing: " + name; return nullptr; }
        return &it->second;
    };
    const Tensor* cfg = find("zip_config");
    if (!cfg) return ADE_ERR_MISSING_KEY;
    if (cfg->count < 14) return zfail(err, ADE_ERR_SHAPE_MISMATCH, "weights: tensor has the wrong shape: zip_config");
    ZipEngine* e = new ZipEngine();
    auto bail = [&](int st) { delete e; return st; };
    auto ci = [&](int i) { return (int)lrintf(cfg->data[i]); };
    e->device = device; e->L = window_len; e->n_win = n_win; e->exact = exact_dft; e->bf16 = bf16;
    e->C = ci(0); e->H = ci(1); e->qd = ci(2); e->pd = ci(3); e->vd = ci(4); e->pos_dim = ci(5); e->ffd = ci(6); e->K = ci(7);
    e->dst = ci(10); e->dsf = ci(11); e->up = ci(12); e->depth = ci(13);
    const int C = e->C;
    if (ci(8) != 1 || ci(9) != 1) return bail(zfail(err, ADE_ERR_UNSUPPORTED, "zipenhancer: encoders 0 and 3 must not be down-sampled"));
    if (C != 64) return bail(zfail(err, ADE_ERR_UNSUPPORTED, "zipenhancer: this build's statistics / norm kernels are specialised for 64 channels"));
    if (e->depth < 1 || e->depth > 4 || e->up < 1 || e->up > 4 || e->dst < 1 || e->dsf < 1 || e->dst > 8 || e->dsf > 8 || !(e->K & 1) || e->K > 63)
        return bail(zfail(err, ADE_ERR_UNSUPPORTED, "zipenhancer: unsupported geometry (dense depth <= 4, up-scale <= 4, down-sampling <= 8, odd depthwise kernel <= 63)"));
    e->hid = C * 3 / 4; e->ff1 = e->ffd * 3 / 4; e->ff3 = e->ffd * 5 / 4; e->attn_dim = e->H * (2 * e->qd + e->pd);
    if ((e->hid % 4) || (e->ff1 % 4) || (e->ff3 % 4) || (e->ffd % 4) || (e->attn_dim % 4) || ((e->H * e->vd) % 4) || (e->vd % 4) || e->hid > 64 || e->vd > 16 || e->H < 1)
        return bail(zfail(err, ADE_ERR_UNSUPPORTED, "zipenhancer: projection widths and value_head_dim must be multiples of 4, hidden_channels <= 64, value_head_dim <= 16"));
    if (e->qd != 16 || e->pd != 4)
        return bail(zfail(err, ADE_ERR_UNSUPPORTED, "zipenhancer: the attention kernel is built for query_head_dim 16 and pos_head_dim 4"));
    e->T = window_len / kZHop + 1;
    e->dynamic_norm = dynamic;
    e->Lo = kZHop * (e->T - 1);                       // STFT (centre pad) -> ISTFT reconstructs whole hops (STFT_Process.py:168-172)
    e->F = (kZF + 2 - 3) / 2 + 1;
    e->dT = (e->T + e->dst - 1) / e->dst;
    e->dF = (e->F + e->dsf - 1) / e->dsf;
    if (std::max(e->T, e->F) > 496)     // the un-folded export reaches INPUT_AUDIO_LENGTH = 48000 -> 481 frames (ZipEnhancer/Export_ZipEnhancer.py:44, :57)
        return bail(zfail(err, ADE_ERR_UNSUPPORTED, "zipenhancer: at most 496 frames per window; fold longer audio into windows (use_batch_fold)"));

    // ---- arena: blob tensors (some repacked), DFT tables, position tables
    std::vector<float> arena;
    auto place = [&](const float* src, size_t n) { const size_t at = arena.size(); arena.insert(arena.end(), src, src + n); arena.resize((arena.size() + 63) & ~(size_t)63); return at; };
    auto need = [&](const std::string& name, std::vector<int> dims) -> const Tensor* {
        const Tensor* t = find(name);
        if (t && t->dims != dims) { if (err.empty() || err.find("missing") != std::string::npos) err = "weights: tensor has the wrong shape: " + name; return nullptr; }
        return t;
    };
    bool ok = true;
    auto put = [&](const std::string& name, std::vector<int> dims) -> size_t {
        const Tensor* t = need(name, dims);
        if (!t) { ok = false; return 0; }
        return place(t->data, t->count);
    };
    auto put_conv = [&](const std::string& name, int co0, int co_n, int cin, int kh, int kw, int co_total) -> size_t {   // [co][ci][kh][kw] rows co0.. -> [co][tap][ci]
        const Tensor* t = need(name, {co_total, cin, kh, kw});
        if (!t) { ok = false; return 0; }
        std::vector<float> r((size_t)co_n * kh * kw * cin);
        for (int co = 0; co < co_n; ++co)
            for (int c = 0; c < cin; ++c)
                for (int tap = 0; tap < kh * kw; ++tap) r[((size_t)co * kh * kw + tap) * cin + c] = t->data[((size_t)(co0 + co) * cin + c) * kh * kw + tap];
        return place(r.data(), r.size());
    };
    struct Off { size_t v; };
    std::vector<std::pair<const float**, size_t>> fix;      // pointer slots to patch once the arena is on the device
    auto bind = [&](const float** slot, size_t at) { fix.push_back({slot, at}); };
    // bf16 copies (round to nearest even, as torch's .to(bfloat16)) of the matrix weights the bf16 path reads: a second arena, patched like the first
    std::vector<uint16_t> arena16;
    std::vector<std::pair<const gemm16::bf16_t**, size_t>> fix16;
    auto to_bf16 = [](float x) -> uint16_t { uint32_t u; memcpy(&u, &x, 4); if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); };
    auto place16 = [&](const float* src, size_t n) { const size_t at = arena16.size(); arena16.resize(at + n); for (size_t i = 0; i < n; ++i) arena16[at + i] = to_bf16(src[i]); arena16.resize((arena16.size() + 63) & ~(size_t)63); return at; };
    auto bind16 = [&](const gemm16::bf16_t** slot, size_t at) { fix16.push_back({slot, at}); };
    const bool dense_half = e->dense_half;
    auto place16d = [&](const float* src, size_t n) {          // the dense blocks' weights: IEEE half unless ADE_ZIP_DENSE_F16=0
        if (!dense_half) return place16(src, n);
        const size_t at = arena16.size(); arena16.resize(at + n); for (size_t i = 0; i < n; ++i) arena16[at + i] = gemm16::f16_bits(src[i]); arena16.resize((arena16.size() + 63) & ~(size_t)63); return at;
    };
    // W2 (C, fd) with every group of 16 hidden units in k_zip_ff16's contraction order: position 8 h + e holds unit 8 (e >> 2) + 4 h + (e & 3)
    auto place16_ffout = [&](const float* src, int fd) {
        std::vector<float> r((size_t)C * fd);
        for (int co = 0; co < C; ++co)
            for (int g0 = 0; g0 < fd; g0 += 16)
                for (int hh = 0; hh < 2; ++hh)
                    for (int e2 = 0; e2 < 8; ++e2) r[(size_t)co * fd + g0 + 8 * hh + e2] = src[(size_t)co * fd + g0 + 8 * (e2 >> 2) + 4 * hh + (e2 & 3)];
        return place16(r.data(), r.size());
    };
    auto slice = [&](const float** slot, const std::string& name, std::vector<int> dims, size_t off = 0) { const size_t at = put(name, dims); bind(slot, at + off); };

    slice(&e->c1_w, "enc_conv1_w", {C, 2}); slice(&e->c1_b, "enc_conv1_b", {C}); slice(&e->c1_g, "enc_norm1_w", {C}); slice(&e->c1_beta, "enc_norm1_b", {C});
    slice(&e->c1_slope, "enc_prelu1", {C});
    {
        const size_t at = put_conv("enc_conv2_w", 0, C, C, 1, 3, C);
        bind(&e->c2_w, at);
        if (bf16 && ok) bind16(&e->c2_w16, place16d(arena.data() + at, (size_t)C * 3 * C));
    } slice(&e->c2_b, "enc_conv2_b", {C}); slice(&e->c2_g, "enc_norm2_w", {C}); slice(&e->c2_beta, "enc_norm2_b", {C});
    slice(&e->c2_slope, "enc_prelu2", {C});
    auto dense = [&](ZDense& d, const std::string& pre, int groups) {
        std::vector<float> slopes((size_t)groups * 4 * C, 1.0f);
        for (int i = 0; i < e->depth; ++i) {
            const std::string p = pre + std::to_string(i);
            const Tensor* pr = need(p + "_pr", {groups * C});
            if (!pr) { ok = false; return; }
            const size_t a_b = put(p + "_b", {groups * C}), a_g = put(p + "_nw", {groups * C}), a_be = put(p + "_nb", {groups * C});
            for (int g = 0; g < groups; ++g) {
                {
                    const size_t at = put_conv(p + "_w", g * C, C, C * (i + 1), 2, 3, groups * C);
                    bind(&d.w[g][i], at);
                    if (bf16 && ok) bind16(&d.w16[g][i], place16d(arena.data() + at, (size_t)C * 6 * C * (i + 1)));
                }
                bind(&d.b[g][i], a_b + (size_t)g * C); bind(&d.gamma[g][i], a_g + (size_t)g * C); bind(&d.beta[g][i], a_be + (size_t)g * C);
                for (int c = 0; c < C; ++c) slopes[((size_t)g * 4 + (3 - i)) * C + c] = pr->data[g * C + c];      // layer i's output lives in slot 3 - i
            }
        }
        bind(&d.slope, place(slopes.data(), slopes.size()));
    };
    dense(e->enc_dense, "enc_dense", 1);
    dense(e->dec_dense, "dec_dense", 2);
    const int ad = e->attn_dim, vdim = e->H * e->vd;
    size_t pos_w_at[4][2];
    for (int en = 0; en < 4; ++en) {
        for (int p = 0; p < 2; ++p) {
            const std::string pre = "enc" + std::to_string(en) + (p ? "_t_" : "_f_");
            ZLayer& w = e->layers[en][p];
            slice(&w.attn_ff1_w, pre + "attn_ff1_w", {ad + e->ff1, C}); slice(&w.attn_ff1_b, pre + "attn_ff1_b", {ad + e->ff1});
            pos_w_at[en][p] = put(pre + "pos_w", {e->H * e->pd, e->pos_dim});
            slice(&w.ff1_out_w, pre + "ff1_out_w", {C, e->ff1}); slice(&w.ff1_out_b, pre + "ff1_out_b", {C});
            slice(&w.nonlin_in_w, pre + "nonlin_in_w", {3 * e->hid, C}); slice(&w.nonlin_in_b, pre + "nonlin_in_b", {3 * e->hid});
            slice(&w.nonlin_out_w, pre + "nonlin_out_w", {C, e->hid}); slice(&w.nonlin_out_b, pre + "nonlin_out_b", {C});
            for (int i = 0; i < 2; ++i) {
                const std::string a = pre + "sa" + std::to_string(i + 1), cv = pre + "conv" + std::to_string(i + 1), ff = pre + "ff" + std::to_string(i + 2);
                const int fd = i ? e->ff3 : e->ffd;
                slice(&w.sa_in_w[i], a + "_in_w", {vdim, C}); slice(&w.sa_in_b[i], a + "_in_b", {vdim}); slice(&w.sa_out_w[i], a + "_out_w", {C, vdim});
                slice(&w.sa_out_b[i], a + "_out_b", {C});
                slice(&w.cv_in_w[i], cv + "_in_w", {2 * C, C}); slice(&w.cv_in_b[i], cv + "_in_b", {2 * C}); slice(&w.cv_dw_w[i], cv + "_dw_w", {C, e->K});
                slice(&w.cv_dw_b[i], cv + "_dw_b", {C}); slice(&w.cv_out_w[i], cv + "_out_w", {C, C}); slice(&w.cv_out_b[i], cv + "_out_b", {C});
                slice(&w.ff_in_w[i], ff + "_in_w", {fd, C}); slice(&w.ff_in_b[i], ff + "_in_b", {fd}); slice(&w.ff_out_w[i], ff + "_out_w", {C, fd});
                slice(&w.ff_out_b[i], ff + "_out_b", {C});
            }
            if (bf16 && ok) {
                ZLayer16& w16 = e->layers16[en][p];
                const Tensor* t = find(pre + "attn_ff1_w");
                bind16(&w16.ff1_w1, place16(t->data + (size_t)ad * C, (size_t)e->ff1 * C));
                bind16(&w16.ff1_w2p, place16_ffout(find(pre + "ff1_out_w")->data, e->ff1));
                bind16(&w16.attn_w, place16(t->data, (size_t)ad * C));
                bind16(&w16.nonlin_in_w, place16(find(pre + "nonlin_in_w")->data, (size_t)3 * e->hid * C));
                bind16(&w16.nonlin_out_w, place16(find(pre + "nonlin_out_w")->data, (size_t)C * e->hid));
                for (int i = 0; i < 2; ++i) {
                    const std::string a = pre + "sa" + std::to_string(i + 1), cv = pre + "conv" + std::to_string(i + 1);
                    bind16(&w16.sa_in_w[i], place16(find(a + "_in_w")->data, (size_t)vdim * C));
                    bind16(&w16.sa_out_w[i], place16(find(a + "_out_w")->data, (size_t)C * vdim));
                    bind16(&w16.cv_in_w[i], place16(find(cv + "_in_w")->data, (size_t)2 * C * C));
                    bind16(&w16.cv_out_w[i], place16(find(cv + "_out_w")->data, (size_t)C * C));
                }
                for (int i = 0; i < 2; ++i) {
                    const std::string ff = pre + "ff" + std::to_string(i + 2);
                    const int fd = i ? e->ff3 : e->ffd;
                    bind16(&w16.ff_w1[i], place16(find(ff + "_in_w")->data, (size_t)fd * C));
                    bind16(&w16.ff_w2p[i], place16_ffout(find(ff + "_out_w")->data, fd));
                }
            }
            slice(&w.bypass_mid, pre + "bypass_mid", {C}); slice(&w.norm_bias, pre + "norm_bias", {C}); slice(&w.fnorm, pre + "final_norm_scale", {C});
            slice(&w.fres, pre + "final_residual_scale", {C});
        }
        if (en == 1 || en == 2) {
            const std::string pre = "enc" + std::to_string(en);
            slice(&e->down_t[en], pre + "_down_t_w", {e->dst}); slice(&e->down_f[en], pre + "_down_f_w", {e->dsf});
            slice(&e->out_scale[en], pre + "_out_scale", {C}); slice(&e->res_scale[en], pre + "_res_scale", {C});
        }
    }
    for (int g = 0; g < 2; ++g) {
        const size_t at = put_conv("dec_up_w", g * C * e->up, C * e->up, C, 1, 3, 2 * C * e->up);
        bind(&e->up_w[g], at);
        if (bf16 && ok) bind16(&e->up_w16[g], place16d(arena.data() + at, (size_t)C * e->up * 3 * C));
    }
    {
        const size_t a = put("dec_up_b", {2 * C * e->up});
        bind(&e->up_b[0], a); bind(&e->up_b[1], a + (size_t)C * e->up);
    }
    slice(&e->up_g, "dec_up_nw", {2 * C}); slice(&e->up_beta, "dec_up_nb", {2 * C}); slice(&e->up_slope, "dec_up_pr", {2 * C});
    slice(&e->mask_w, "mask_out_w", {1, C, 1, 2}); slice(&e->mask_b, "mask_out_b", {1}); slice(&e->phase_w, "phase_out_w", {2, C, 1, 2}); slice(&e->phase_b, "phase_out_b", {2});
    if (!ok) return bail(err.find("missing") != std::string::npos ? ADE_ERR_MISSING_KEY : ADE_ERR_SHAPE_MISMATCH);

    // ---- DFT tables (ZipEnhancer/STFT_Process.py:205-249): torch.hann_window(periodic=True) in fp32; angles fp32(2 pi / N) * f * n in fp32 unless "exact"
    {
        const int N = kZN;
        std::vector<float> fwd((size_t)kZC2 * N), inv((size_t)kZC2 * N), win((size_t)N), iws((size_t)e->Lo);
        const float step = (float)(2.0 * M_PI / (double)N);
        for (int n = 0; n < N; ++n) win[n] = cosf((float)n * step) * (-0.5f) + 0.5f;
        for (int f = 0; f < kZF; ++f) {
            const float scale = (f == 0 || f == kZF - 1) ? 1.0f : 2.0f;
            for (int n = 0; n < N; ++n) {
                float c, s;
                if (exact_dft) {
                    const double a = 2.0 * M_PI * (double)(((long long)f * n) % N) / N;
                    c = (float)cos(a); s = (float)sin(a);
                } else {
                    const float omega = (step * (float)f) * (float)n;
                    c = cosf(omega); s = sinf(omega);
                }
                fwd[(size_t)f * N + n] = c * win[n];
                fwd[(size_t)(kZF + f) * N + n] = -s * win[n];
                inv[(size_t)f * N + n] = ((scale * c) * (float)(1.0 / N)) * win[n];
                inv[(size_t)(kZF + f) * N + n] = ((scale * -s) * (float)(1.0 / N)) * win[n];
            }
        }
        std::vector<float> raw((size_t)N + (size_t)kZHop * (e->T - 1), 0.0f);
        for (int t = 0; t < e->T; ++t)
            for (int n = 0; n < N; ++n) raw[(size_t)t * kZHop + n] += win[n] * win[n];
        for (int m = 0; m < e->Lo; ++m) iws[m] = dynamic ? raw[(size_t)m + N / 2] : 1.0f / raw[(size_t)m + N / 2];          // inv_win_sum (static_norm, :245-249); the sum itself for a dynamic export
        bind(&e->k_fwd, place(fwd.data(), fwd.size())); bind(&e->k_inv, place(inv.data(), inv.size())); bind(&e->inv_wsum, place(iws.data(), iws.size()));
    }
    // ---- projected position tables (:597-604): rows x = -(n - 1) .. n - 1 of CompactRelPositionalEncoding (the published Zipformer2 table, fp32 like
    //      torch), times linear_pos, stored (head, pos_head_dim, 2 n - 1)
    for (int en = 0; en < 4; ++en)
        for (int p = 0; p < 2; ++p) {
            const bool down = en == 1 || en == 2;
            const int n = p ? (down ? e->dT : e->T) : (down ? e->dF : e->F), n2 = 2 * n - 1, D = e->pos_dim, HP = e->H * e->pd;
            std::vector<float> pe((size_t)n2 * D), tab((size_t)HP * n2);
            const float cl = sqrtf((float)D), lcl = (float)log(sqrt((double)D)), ls = (float)((double)D / (2.0 * M_PI));
            for (int r = 0; r < n2; ++r) {
                const float x = (float)(r - (n - 1)), sg = x > 0.0f ? 1.0f : x < 0.0f ? -1.0f : 0.0f;
                const float xc = cl * sg * (logf(fabsf(x) + cl) - lcl), xa = atanf(xc / ls);
                for (int k = 0; k < D / 2; ++k) { pe[(size_t)r * D + 2 * k] = cosf(xa * (float)(1 + k)); pe[(size_t)r * D + 2 * k + 1] = sinf(xa * (float)(1 + k)); }
                pe[(size_t)r * D + D - 1] = 1.0f;
            }
            const float* W = arena.data() + pos_w_at[en][p];
            for (int hp = 0; hp < HP; ++hp)
                for (int r = 0; r < n2; ++r) {
                    float a = 0.0f;
                    for (int k = 0; k < D; ++k) a += pe[(size_t)r * D + k] * W[(size_t)hp * D + k];
                    tab[(size_t)hp * n2 + r] = a;
                }
            bind(&e->layers[en][p].pos, place(tab.data(), tab.size()));
        }

    if (hipSetDevice(device) != hipSuccess) return bail(zfail(err, ADE_ERR_DEVICE, "hipSetDevice failed"));
    if (hipMalloc((void**)&e->d_w, arena.size() * sizeof(float)) != hipSuccess) return bail(zfail(err, ADE_ERR_DEVICE, "hipMalloc of the ZipEnhancer weights failed"));
    if (hipMemcpy(e->d_w, arena.data(), arena.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
        return bail(zfail(err, ADE_ERR_DEVICE, "upload of the ZipEnhancer weights failed"));
    for (auto& f : fix) *f.first = e->d_w + f.second;
    if (bf16) {
        if ((e->ff1 % 64) || (e->ffd % 64) || (e->ff3 % 64) || e->depth != 4 || e->hid != 48 || e->H * e->vd != 48 || e->attn_dim != 144 || e->up != 2 || e->K != 15)
            return bail(zfail(err, ADE_ERR_UNSUPPORTED, "zipenhancer: ade_gemm_dtype = bf16 is built for the published geometry (feed-forward widths multiples of 64, dense depth 4, 48 hidden / value "
                                                        "channels, 4 heads of 16 + 16 + 4, kernel 15, up-scale 2); other geometries run f32"));
        if (hipMalloc((void**)&e->d_w16, arena16.size() * sizeof(uint16_t)) != hipSuccess ||
            hipMemcpy(e->d_w16, arena16.data(), arena16.size() * sizeof(uint16_t), hipMemcpyHostToDevice) != hipSuccess)
            return bail(zfail(err, ADE_ERR_DEVICE, "upload of the ZipEnhancer bf16 weights failed"));
        for (auto& f : fix16) *f.first = e->d_w16 + f.second;
    }
    if (raise_attn_lds<0, 3, float>() != hipSuccess || raise_attn_lds<0, 4, float>() != hipSuccess || raise_attn_lds<1, 1, float>() != hipSuccess ||
        (bf16 && (raise_attn_lds<0, 3, gemm16::bf16_t>() != hipSuccess || raise_attn_lds<1, 1, gemm16::bf16_t>() != hipSuccess || raise_attn16_lds<0, 3>() != hipSuccess ||
                  raise_attn16_lds<1, 1>() != hipSuccess)))
        return bail(zfail(err, ADE_ERR_DEVICE, "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed for the attention kernel"));
    // the bf16 path's row kernels that need more than 48 KB of dynamic LDS: raised here, once per created engine on ITS device, and checked (they used to be raised behind a
    // per-process flag at the first launch, result ignored)
    if (bf16 && (zip16::raise_rows16_lds<12, 4, true, zip16::RowConv16, zip16::SubPixelStore16>() != hipSuccess ||
                 zip16::raise_rows16_lds<12, 4, false, zip16::RowConv16, zip16::SubPixelStore16>() != hipSuccess ||
                 zip16::raise_rows16_lds<12, 4, true, zip16::RowConv16, zip16::SubPixelStore16, true>() != hipSuccess ||
                 zip16::raise_rows16_lds<12, 4, false, zip16::RowConv16, zip16::SubPixelStore16, true>() != hipSuccess ||
                 zip16::raise_rows16_chain_lds<3, 4, zip16::B16Rows<0>, true>() != hipSuccess || zip16::raise_rows16_chain_lds<3, 2, zip16::B16Rows<0>>() != hipSuccess))
        return bail(zfail(err, ADE_ERR_DEVICE, "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed for the bf16 row kernels"));
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_zip_ff<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFfLds) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_zip_ff<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFfLds) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_zip_ff<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFfLds) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_zip_ff<3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFfLds) != hipSuccess)
        return bail(zfail(err, ADE_ERR_DEVICE, "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed for the feed-forward kernel"));
    *out = e;
    return ADE_OK;
}

int ZipEngine::reserve(int batch, std::string& err) {
    if (batch <= capacity) return ADE_OK;
    ZP_HIP(hipSetDevice(device));
    ZP_HIP(hipDeviceSynchronize());
    if (ws) (void)hipFree(ws);
    if (partial) (void)hipFree(partial);
    ws = nullptr; partial = nullptr; capacity = 0;
    const size_t B = (size_t)batch * n_win, J = B * T, tok0 = J * kZF, R = J * F, Rd = B * dT * dF, F2 = (size_t)F * up;
    const size_t wide = (size_t)std::max({3 * hid, H * vd, 2 * C, ffd, ff3});
    const size_t Rt = std::min<size_t>(B, 8) * T * F;     // encoder snapshots: always carved, for calls of at most 8 windows whatever capacity was reserved
    const size_t dh = std::max(tok0 * 4 * C, R * 8 * C);
    lt_stride = want_layer_taps ? ((Rt * C + 63) & ~(size_t)63) : 0;
    const size_t sizes[] = {B, (size_t)kZC2 * J, tok0 * 2, B * C * 4, tok0 * C, (bf16 && parts16 == 7) ? (size_t)64 : dh, B * 8 * C * 2 + B * 2 * C * 2, R * C, R * C, Rd * C, R * (size_t)(attn_dim + ff1), R * wide, R * C,
                            J * F2 * 2 * C, (size_t)kZC2 * J, J * kZN, J * kZF, Rt * C, Rt * C, Rt * C, Rt * C, Rt * C, 8 * lt_stride};
    float** ptrs[] = {&norm, &spec, &feat, &coef, &E0, &Dh, &nrm, &X, &Y, &X2, &P, &S1, &O, &U, &packed, &frames_buf, &mask_tap, &enc_tap[0], &enc_tap[1], &enc_tap[2],
                      &enc_tap[3], &enc_tap[4], &lt};
    const int nbuf = 23;
    size_t total = 0;
    for (int i = 0; i < nbuf; ++i) total += (sizes[i] + 63) & ~(size_t)63;
    ZP_HIP(hipMalloc((void**)&ws, total * sizeof(float)));
    size_t at = 0;
    for (int i = 0; i < nbuf; ++i) { *ptrs[i] = ws + at; at += (sizes[i] + 63) & ~(size_t)63; }
    if (!want_layer_taps) lt = nullptr;
    nrm2 = nrm + B * 8 * C * 2;                    // statistics of the tensors normalised outside the dense blocks (dense_conv_2, the up-sampler)
    if (bf16) {
        if (ws16) (void)hipFree(ws16);
        ws16 = nullptr;
        const size_t n_dh = (dh + 63) & ~(size_t)63, n_e0 = (tok0 * C + 63) & ~(size_t)63, n_x = (R * C + 63) & ~(size_t)63, n_p = (R * (size_t)attn_dim + 63) & ~(size_t)63;
        ZP_HIP(hipMalloc((void**)&ws16, (n_dh + n_e0 + 2 * n_x + 2 * n_p) * sizeof(gemm16::bf16_t)));
        Dh16 = ws16; E016 = Dh16 + n_dh; X16 = E016 + n_e0; O16 = X16 + n_x; P16 = O16 + n_x; S16 = P16 + n_p;
        raw = E0;                                  // the fp32 E0 buffer is free on this path (E0 itself is bf16): a dense layer's raw output, [tokens][64]
    }
    const size_t nchunk0 = ((size_t)T * kZF + kChunkTok - 1) / kChunkTok, nchunk2 = ((size_t)T * F2 + kChunkTok - 1) / kChunkTok;
    const size_t nblk0 = ((size_t)T * kZF + 255) / 256;         // the dense kernels emit one partial per 256-token tile
    const size_t nblk_rows = ((size_t)T * F + 127) / 128;                  // (zip16::k_rows16<.., STATS>: tiles of 128 rows per window)
    ZP_HIP(hipMalloc((void**)&partial, B * std::max({nchunk0, nchunk2, nblk0, nblk_rows}) * 64 * 2 * sizeof(double)));
    capacity = batch;
    return ADE_OK;
}

void ZipEngine::stats(hipStream_t s, const float* x, int ld, int ch0, int tok_per_win, int windows, const float* gamma, const float* beta, float* nrm_, int nrm_ld, int nrm_ch0) {
    const int nchunk = (tok_per_win + kChunkTok - 1) / kChunkTok;
    hipLaunchKernelGGL(k_zip_stats_partial, dim3((unsigned)nchunk, (unsigned)windows), dim3(256), 0, s, x, ld, ch0, tok_per_win, partial);
    hipLaunchKernelGGL(k_zip_stats_final, dim3((unsigned)windows), dim3(256), 0, s, (const double*)partial, nchunk, (double)tok_per_win, gamma, beta, nrm_, nrm_ld, nrm_ch0);
}

// DenseBlockV2 (:701-757): layer i of group g writes its raw output (+ bias) to hist channels [g 4 C + (3 - i) C, + C) and its statistics to nrm
void ZipEngine::dense_block(hipStream_t s, const ZDense& d, int groups, const float* inp, int windows, int Fd) {
    const int ld = groups * 4 * C, M = windows * T * Fd, TF = T * Fd, nblk = (TF + 255) / 256;
    for (int i = 0; i < depth; ++i)
        for (int g = 0; g < groups; ++g) {
            const int cin = (i + 1) * C, off_out = g * 4 * C + (3 - i) * C;
            const dim3 grid((unsigned)(nblk * windows));              // (C == 64: checked at create) tiles per window x windows: a tile's rows share their InstanceNorm statistics
            hipLaunchKernelGGL(k_zip_dense, grid, dim3(256), 0, s, (const float*)Dh, inp, ld, g * 4 * C + (4 - i) * C, i * C, cin, T, Fd, 1 << i, d.w[g][i],
                                    d.b[g][i], Dh, ld, off_out, partial, nblk);
            hipLaunchKernelGGL(k_zip_stats_final, dim3((unsigned)windows), dim3(256), 0, s, (const double*)partial, nblk, (double)TF, d.gamma[g][i], d.beta[g][i], nrm, ld, off_out);
            const long long total16 = (long long)M * 16;
            hipLaunchKernelGGL(k_zip_hist_norm, dim3((unsigned)((total16 + 255) / 256)), dim3(256), 0, s, Dh, ld, off_out, (const float*)nrm, d.slope, T * Fd, total16);
        }
}

// The same block on bf16 operands (csrc/ade_zip16.h): the history is bf16 [tokens][groups 4 C], layer i's raw fp32 output goes to `raw`, its InstanceNorm partial sums come
// out of the product's own epilogue, and k_zip_hist_norm16 writes the normalised + PReLU'd bf16 values into the history (the last layer's also in fp32 for its consumers).
void ZipEngine::dense_block16(hipStream_t s, const ZDense& d, int groups, const gemm16::bf16_t* inp, int windows, int Fd) {
    const int ld = groups * 4 * C, TF = T * Fd, nblk = (TF + 255) / 256;
    const long long M = (long long)windows * TF;
    for (int i = 0; i < depth; ++i)
        for (int g = 0; g < groups; ++g) {
            const int cin = (i + 1) * C, off_out = g * 4 * C + (3 - i) * C;
#define ADE_DENSE16(CB, HALF) hipLaunchKernelGGL((zip16::k_zip_dense16<CB, HALF>), dim3((unsigned)(nblk * windows)), dim3(256), 0, s, (const gemm16::bf16_t*)Dh16, inp, ld, g * 4 * C + (4 - i) * C, \
                                                 i * C, cin, T, Fd, 1 << i, d.w16[g][i], d.b[g][i], raw, partial, nblk)
            if (dense_cb == 64) { if (dense_half) ADE_DENSE16(64, true); else ADE_DENSE16(64, false); }
            else { if (dense_half) ADE_DENSE16(32, true); else ADE_DENSE16(32, false); }
#undef ADE_DENSE16
            hipLaunchKernelGGL(k_zip_stats_final, dim3((unsigned)windows), dim3(256), 0, s, (const double*)partial, nblk, (double)TF, d.gamma[g][i], d.beta[g][i], nrm, ld, off_out);
            const long long total16 = M * 16;
            if (dense_half) hipLaunchKernelGGL(zip16::k_zip_hist_norm16<true>, dim3((unsigned)((total16 + 255) / 256)), dim3(256), 0, s, (const float*)raw, Dh16, ld, off_out, (const float*)nrm, ld, d.slope,
                                               TF, (float*)nullptr, total16);
            else hipLaunchKernelGGL(zip16::k_zip_hist_norm16<false>, dim3((unsigned)((total16 + 255) / 256)), dim3(256), 0, s, (const float*)raw, Dh16, ld, off_out, (const float*)nrm, ld, d.slope,
                                    TF, (float*)nullptr, total16);
        }
}

void ZipEngine::attention(hipStream_t s, int mode, const float* pos, const float* src, int lds_, float* out, int ldo, SeqGeo geo, int dv) {
    const int ldp = attn_dim + ff1;
    if (mode == 0) {
        if (dv <= 48) launch_attn<0, 3, float>(s, 1, P, ldp, pos, src, lds_, out, ldo, geo, dv);
        else launch_attn<0, 4, float>(s, 1, P, ldp, pos, src, lds_, out, ldo, geo, dv);
    } else launch_attn<1, 1, float>(s, H, P, ldp, pos, src, lds_, out, ldo, geo, dv);
}

// one fused Zipformer2 encoder layer in place on x (R rows), sequences described by geo (:143-187)
void ZipEngine::layer(hipStream_t s, const ZLayer& w, const ZLayer16& w16, float* x, long long R, SeqGeo geo) {
    if (bf16 && (parts16 & 2)) return layer16(s, w, w16, x, R, geo);
    using namespace gemm64;
    const int M = (int)R, ldp = attn_dim + ff1, n = geo.n, vdim = H * vd;
    // feed-forward modules run fused (k_zip_ff) when their width is a multiple of the 64-unit weight chunk
    auto fused_ff = [&](int fd) { return C == 64 && fd % kFfChunk == 0 && attn_dim % 4 == 0; };
    const bool fnorm_fused = fused_ff(ff3) && w.fnorm == w.norm_bias + 64 && w.fres == w.norm_bias + 128;      // nb | fs | rs side by side in the arena: k_zip_ff<3> applies the final norm
    if (fused_ff(ff1)) {
        launch_proj64(s, x, C, w.attn_ff1_w, w.attn_ff1_b, P, ldp, 0, M, attn_dim);                               // (:148-153) attention part of the joint projection
        launch_zip_ff<0>(s, M, x, w.attn_ff1_w + (size_t)attn_dim * C, w.attn_ff1_b + attn_dim, w.ff1_out_w, w.ff1_out_b, x, nullptr, Y, ff1);        // (:160)
    } else {
        launch_proj64(s, x, C, w.attn_ff1_w, w.attn_ff1_b, P, ldp, 0, M, ldp);                                    // (:148-153)
        launch(s, ActRowsA<1>{P + attn_dim, ldp}, WeightB{w.ff1_out_w, ff1}, AddFromStore{x, Y, w.ff1_out_b, C}, M, C, ff1);                  // (:160)
    }
    layer_tap(s, 0, Y, R);
    launch_proj64(s, Y, C, w.nonlin_in_w, w.nonlin_in_b, S1, 3 * hid, 0, M, 3 * hid);                           // (:305)
    attention(s, 0, w.pos, S1, 3 * hid, O, hid, geo, hid);                                                                                    // (:154-159, :310-316) head 0
    launch(s, RowsA{O, hid}, WeightB{w.nonlin_out_w, hid}, ResidualBiasStore{Y, w.nonlin_out_b, C}, M, C, hid);                              // (:317, :167)
    layer_tap(s, 1, Y, R);
    for (int i = 0; i < 2; ++i) {
        launch_proj64(s, Y, C, w.sa_in_w[i], w.sa_in_b[i], S1, vdim, 0, M, vdim);                               // (:296)
        attention(s, 1, w.pos, S1, vdim, O, vdim, geo, vd);                                                                                   // (:297-300) all heads
        launch(s, RowsA{O, vdim}, WeightB{w.sa_out_w[i], vdim}, ResidualBiasStore{Y, w.sa_out_b[i], C}, M, C, vdim);                         // (:301, :168 / :172)
        layer_tap(s, 2 + 3 * i, Y, R);
        launch_proj64(s, Y, C, w.cv_in_w[i], w.cv_in_b[i], S1, 2 * C, 0, M, 2 * C);                             // (:321)
        const dim3 cg((unsigned)geo.nseq, (unsigned)((n + 63) / 64));
        const size_t cl = (size_t)(64 + K - 1) * C * sizeof(float);
        if (C == 64 && K == 15) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_zip_dwconv<64, 15, float>), cg, dim3(256), cl, s, (const float*)S1, w.cv_dw_w[i], w.cv_dw_b[i], O, geo, C, K);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_zip_dwconv<0, 0, float>), cg, dim3(256), cl, s, (const float*)S1, w.cv_dw_w[i], w.cv_dw_b[i], O, geo, C, K);   // (:325-336)
        launch(s, ActRowsA<2>{O, C}, WeightB{w.cv_out_w[i], C}, ResidualBiasStore{Y, w.cv_out_b[i], C}, M, C, C);                            // (:339, :169 / :173)
        layer_tap(s, 3 + 3 * i, Y, R);
        const int fd = i ? ff3 : ffd;
        if (fused_ff(fd)) {
            if (i == 0) launch_zip_ff<2>(s, M, Y, w.ff_in_w[i], w.ff_in_b[i], w.ff_out_w[i], w.ff_out_b[i], x, w.bypass_mid, Y, fd);                  // (:170-171)
            else if (fnorm_fused) launch_zip_ff<3>(s, M, Y, w.ff_in_w[i], w.ff_in_b[i], w.ff_out_w[i], w.ff_out_b[i], x, w.norm_bias, x, fd);         // (:174-183) the final norm rides in the store
            else launch_zip_ff<1>(s, M, Y, w.ff_in_w[i], w.ff_in_b[i], w.ff_out_w[i], w.ff_out_b[i], nullptr, nullptr, Y, fd);                        // (:174)
            if (i == 0) layer_tap(s, 4, Y, R);
            else if (fnorm_fused) layer_tap(s, 7, x, R);
            continue;
        }
        launch_proj64(s, Y, C, w.ff_in_w[i], w.ff_in_b[i], S1, fd, 0, M, fd);
        if (i == 0) { launch(s, ActRowsA<1>{S1, fd}, WeightB{w.ff_out_w[i], fd}, BypassMidStore{x, Y, w.ff_out_b[i], w.bypass_mid, C}, M, C, fd); layer_tap(s, 4, Y, R); }   // (:170-171)
        else launch(s, ActRowsA<1>{S1, fd}, WeightB{w.ff_out_w[i], fd}, ResidualBiasStore{Y, w.ff_out_b[i], C}, M, C, fd);                   // (:174)
    }
    if (!fnorm_fused) { hipLaunchKernelGGL(k_zip_final_norm, dim3((unsigned)((R + 15) / 16)), dim3(256), 0, s, x, (const float*)Y, w.norm_bias, w.fnorm, w.fres, R, C); layer_tap(s, 7, x, R); }   // (:175-183)
}

// The same layer on the bf16 path (csrc/ade_zip16.h): the residual stream x / Y stays fp32; every projection reads it through a loader that rounds to bf16 and stores bf16
// (P16: q | k | p per head; S16: a module's projection; O16: a module's core output); the attention core and the convolution module read / write bf16 and compute in fp32;
// the out-projections add into the fp32 stream; the feed-forward modules are k_zip_ff16.
void ZipEngine::layer16(hipStream_t s, const ZLayer& w, const ZLayer16& w16, float* x, long long R, SeqGeo geo) {
    using namespace zip16;
    const int M = (int)R, n = geo.n, vdim = H * vd;
    const bool chain = zip_chain;                    // out-projection + next in-projection pairs in one launch (k_rows16_chain; ADE_ZIP_CHAIN=0: the two-kernel form, same bits)
    // Row-local runs of the layer in one launch each (k_zip_ffx): [attention in-projection | feed-forward 1 | NonlinAttention in-projection], [convolution 1 out-projection |
    // feed-forward 2 + bypass | self-attention 2 in-projection], [convolution 2 out-projection | feed-forward 3 + final norm].  Not while the per-module taps of the first layer
    // are armed: the stream after a convolution module is not written back in the fused form.
    const bool fuse = zip_fuse && !(lt && lt_armed) && attn_dim <= 160 && 3 * hid <= 160 && vdim <= 64;
    if (fuse) {
        launch_zip_ffx<0, 2, 5>(s, M, x, w16.ff1_w1, w.attn_ff1_b + attn_dim, w16.ff1_w2p, w.ff1_out_b, x, nullptr, Y, ff1,
                                FfxPre{nullptr, w16.attn_w, w.attn_ff1_b, P16, attn_dim, attn_dim}, FfxPost{w16.nonlin_in_w, w.nonlin_in_b, S16, 3 * hid, 3 * hid});       // (:148-153, :160, :305)
    } else {
    launch_rows16<4, 5>(s, F32Rows{x, C}, w16.attn_w, Bf16BiasStore{P16, w.attn_ff1_b, attn_dim, 0}, M, attn_dim);                                 // (:148-153)
    launch_zip_ff16<0>(s, M, x, w16.ff1_w1, w.attn_ff1_b + attn_dim, w16.ff1_w2p, w.ff1_out_b, x, nullptr, Y, ff1);                                  // (:160)
    layer_tap(s, 0, Y, R);
    launch_rows16<4, 5>(s, F32Rows{Y, C}, w16.nonlin_in_w, Bf16BiasStore{S16, w.nonlin_in_b, 3 * hid, 0}, M, 3 * hid);                              // (:305)
    }
    launch_attn16<0, 3>(s, 1, (const gemm16::bf16_t*)P16, attn_dim, w.pos, (const gemm16::bf16_t*)S16, 3 * hid, O16, hid, geo, hid);   // (:154-159, :310-316)
    if (chain) launch_rows16_chain<3, 2>(s, B16Rows<0>{O16, hid}, w16.nonlin_out_w, w.nonlin_out_b, Y, w16.sa_in_w[0], w.sa_in_b[0], S16, vdim, M, vdim);   // (:317, :167) + (:296)
    else launch_rows16<3, 2>(s, B16Rows<0>{O16, hid}, w16.nonlin_out_w, ResidualStore{Y, w.nonlin_out_b}, M, C);                                    // (:317, :167)
    layer_tap(s, 1, Y, R);
    for (int i = 0; i < 2; ++i) {
        if (!(chain && i == 0) && !(fuse && i == 1)) launch_rows16<4, 2>(s, F32Rows{Y, C}, w16.sa_in_w[i], Bf16BiasStore{S16, w.sa_in_b[i], vdim, 0}, M, vdim);          // (:296)
        launch_attn16<1, 1>(s, H, (const gemm16::bf16_t*)P16, attn_dim, w.pos, (const gemm16::bf16_t*)S16, vdim, O16, vdim, geo, vd);  // (:297-300)
        // the in-projection's store applies the GLU (:322-323): S16 holds C columns, a * sigmoid(gate)
        if (chain) launch_rows16_chain<3, 4, true>(s, B16Rows<0>{O16, vdim}, w16.sa_out_w[i], w.sa_out_b[i], Y, w16.cv_in_w[i], w.cv_in_b[i], S16, C, M, 2 * C);   // (:301) + (:321-323)
        else {
            launch_rows16<3, 2>(s, B16Rows<0>{O16, vdim}, w16.sa_out_w[i], ResidualStore{Y, w.sa_out_b[i]}, M, C);                                  // (:301)
            launch_rows16<4, 4>(s, F32Rows{Y, C}, w16.cv_in_w[i], GluStore16{S16, w.cv_in_b[i], C}, M, 2 * C);                                      // (:321-323)
        }
        layer_tap(s, 2 + 3 * i, Y, R);
        const dim3 cg((unsigned)geo.nseq, (unsigned)((n + 63) / 64));
        const size_t cl = (size_t)(64 + K - 1) * C * sizeof(float);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_zip_dwconv<64, 15, gemm16::bf16_t, true>), cg, dim3(256), cl, s, (const gemm16::bf16_t*)S16, w.cv_dw_w[i], w.cv_dw_b[i], O16, geo, C, K);   // (:325-336)
        const int fd = i ? ff3 : ffd;
        if (fuse && (i == 0 || (w.fnorm == w.norm_bias + 64 && w.fres == w.norm_bias + 128))) {
            const FfxPre pre{O16, w16.cv_out_w[i], w.cv_out_b[i], nullptr, 0, C};                                                                    // (:339)
            if (i == 0) launch_zip_ffx<2, 1, 2>(s, M, Y, w16.ff_w1[i], w.ff_in_b[i], w16.ff_w2p[i], w.ff_out_b[i], x, w.bypass_mid, Y, fd, pre,
                                                FfxPost{w16.sa_in_w[1], w.sa_in_b[1], S16, vdim, vdim});                                             // (:170-171) + (:296)
            else launch_zip_ffx<3, 1, 0>(s, M, Y, w16.ff_w1[i], w.ff_in_b[i], w16.ff_w2p[i], w.ff_out_b[i], x, w.norm_bias, x, fd, pre, FfxPost{nullptr, nullptr, nullptr, 0, 0});   // (:174-183)
            continue;
        }
        launch_rows16<4, 2>(s, B16Rows<2, true>{O16, C}, w16.cv_out_w[i], ResidualStore{Y, w.cv_out_b[i]}, M, C);                                         // (:339)
        layer_tap(s, 3 + 3 * i, Y, R);
        if (i == 0) { launch_zip_ff16<2>(s, M, Y, w16.ff_w1[i], w.ff_in_b[i], w16.ff_w2p[i], w.ff_out_b[i], x, w.bypass_mid, Y, fd); layer_tap(s, 4, Y, R); }   // (:170-171)
        else if (w.fnorm == w.norm_bias + 64 && w.fres == w.norm_bias + 128) {        // (nb | fs | rs sit side by side in the arena: the final norm rides in the module's store)
            launch_zip_ff16<3>(s, M, Y, w16.ff_w1[i], w.ff_in_b[i], w16.ff_w2p[i], w.ff_out_b[i], x, w.norm_bias, x, fd);                            // (:174-183)
            layer_tap(s, 7, x, R);
        } else {
            launch_zip_ff16<1>(s, M, Y, w16.ff_w1[i], w.ff_in_b[i], w16.ff_w2p[i], w.ff_out_b[i], nullptr, nullptr, Y, fd);                        // (:174)
            hipLaunchKernelGGL(k_zip_final_norm, dim3((unsigned)((R + 15) / 16)), dim3(256), 0, s, x, (const float*)Y, w.norm_bias, w.fnorm, w.fres, R, C);   // (:175-183)
            layer_tap(s, 7, x, R);
        }
    }
}

void ZipEngine::dualpath(hipStream_t s, int e, float* x, int B, int Tt, int Ff) {        // (:782-792)
    const long long R = (long long)B * Tt * Ff;
    lt_armed = e == 0 && keep_taps;
    layer(s, layers[e][0], layers16[e][0], x, R, SeqGeo{B * Tt, Ff, 1, (long long)Ff, 0, 1});                               // frequency path: (b, t) sequences of Ff consecutive rows
    lt_armed = false;
    layer(s, layers[e][1], layers16[e][1], x, R, SeqGeo{B * Ff, Tt, Ff, (long long)Tt * Ff, 1, (long long)Ff});             // time path: (b, f) sequences, rows Ff apart
}

int ZipEngine::run(hipStream_t s, const int16_t* d_in, int batch, int16_t* d_out, float* d_f32, std::string& err) {
    if (batch == 0) return ADE_OK;
    int st = reserve(batch, err);
    if (st != ADE_OK) return st;
    const int B = batch * n_win, J = B * T, TF0 = T * kZF, F2 = F * up;
    const long long tok0 = (long long)J * kZF, R = (long long)J * F;
    auto flat = [](long long n) { return dim3((unsigned)((n + 255) / 256)); };
    keep_taps = B <= 8;                            // encoder snapshots (5 device copies per call) only for test-sized CALLS
    auto snap = [&](int i) { if (keep_taps) (void)hipMemcpyAsync(enc_tap[i], X, (size_t)R * C * sizeof(float), hipMemcpyDeviceToDevice, s); };
    // ---- front (:819-844)
    hipLaunchKernelGGL(k_zip_window_norm, dim3((unsigned)B), dim3(256), 0, s, d_in, float_in, norm, L);
    gemm::launch(s, gemm::RowMajorA{k_fwd, kZN}, ZFrameB{d_in, float_in, norm, L, T}, ZSpecStore{spec, J}, kZC2, J, kZN);
    const int nchunk0 = (TF0 + kChunkTok - 1) / kChunkTok;
    hipLaunchKernelGGL(k_zip_features, dim3((unsigned)nchunk0, (unsigned)B), dim3(256), 0, s, (const float*)spec, (float2*)feat, partial, T, J, kChunkTok);
    hipLaunchKernelGGL(k_zip_conv1_coef, dim3((unsigned)B), dim3(64), 0, s, (const double*)partial, nchunk0, (double)TF0, c1_w, c1_b, c1_g, c1_beta, (float4*)coef, C);
    const bool enc16 = bf16 && (parts16 & 1), dec16 = bf16 && (parts16 & 4);
    if (enc16) {
        if (dense_half) hipLaunchKernelGGL(k_zip_conv1_apply16<true>, flat(tok0 * (C / 4)), dim3(256), 0, s, (const float2*)feat, (const float4*)coef, c1_slope, E016, TF0, C, tok0 * (C / 4));
        else hipLaunchKernelGGL(k_zip_conv1_apply16<false>, flat(tok0 * (C / 4)), dim3(256), 0, s, (const float2*)feat, (const float4*)coef, c1_slope, E016, TF0, C, tok0 * (C / 4));
        dense_block16(s, enc_dense, 1, E016, B, kZF);
        // (:853) the convolution's epilogue also emits its output's InstanceNorm partial sums (zip16::k_rows16<.., STATS>): no pass over X for them
        const int nblk_x = (T * F + 127) / 128;
        if (dense_half) zip16::launch_rows16_stats<12, 2, true>(s, zip16::RowConv16{Dh16, 4 * C, 0, T, kZF, F, 2}, c2_w16, zip16::F32BiasStore{X, c2_b, C}, C, T * F, B, partial, 1);
        else zip16::launch_rows16_stats<12, 2>(s, zip16::RowConv16{Dh16, 4 * C, 0, T, kZF, F, 2}, c2_w16, zip16::F32BiasStore{X, c2_b, C}, C, T * F, B, partial, 1);
        hipLaunchKernelGGL(k_zip_stats_final, dim3((unsigned)B), dim3(256), 0, s, (const double*)partial, nblk_x, (double)(T * F), c2_g, c2_beta, nrm2, C, 0);
    } else {
    hipLaunchKernelGGL(k_zip_conv1_apply, flat(tok0 * (C / 4)), dim3(256), 0, s, (const float2*)feat, (const float4*)coef, c1_slope, E0, TF0, C, tok0 * (C / 4));   // (:851)
    // ---- DenseEncoder (:852-853)
    dense_block(s, enc_dense, 1, E0, B, kZF);
    gemm64::launch(s, RowConvA{Dh, 4 * C, (4 - depth) * C, C, T, kZF, F, 2}, gemm64::WeightB{c2_w, 3 * C}, BiasColStore{X, c2_b, C, 0}, (int)R, C, 3 * C);
    stats(s, X, C, 0, T * F, B, c2_g, c2_beta, nrm2, C, 0);
    }
    hipLaunchKernelGGL(k_zip_norm_apply, flat(R * (C / 4)), dim3(256), 0, s, X, (const float*)nrm2, c2_slope, T * F, C, R * (C / 4));
    snap(0);
    // ---- the four dual-path encoders (:859-863)
    for (int e = 0; e < 4; ++e) {
        if (e == 1 || e == 2) {                                                                                     // (:794-816)
            const long long Rd = (long long)B * dT * dF;
            hipLaunchKernelGGL(k_zip_downsample, flat(Rd * (C / 4)), dim3(256), 0, s, (const float*)X, X2, down_t[e], down_f[e], T, F, dT, dF, dst, dsf, C, Rd * (C / 4));
            dualpath(s, e, X2, B, dT, dF);
            hipLaunchKernelGGL(k_zip_upsample_combine, flat(R * (C / 4)), dim3(256), 0, s, X, (const float*)X2, out_scale[e], res_scale[e], T, F, dT, dF, dst, dsf, C, R * (C / 4));
        } else dualpath(s, e, X, B, T, F);
        snap(e + 1);
    }
    // ---- mask | phase decoder pair (:864-893)
    if (dec16) {
        if (dense_half) hipLaunchKernelGGL(zip16::k_zip_to_bf16<true>, flat(R * (C / 4)), dim3(256), 0, s, (const float*)X, X16, R * (C / 4));
        else hipLaunchKernelGGL(zip16::k_zip_to_bf16<false>, flat(R * (C / 4)), dim3(256), 0, s, (const float*)X, X16, R * (C / 4));
        dense_block16(s, dec_dense, 2, X16, B, F);
    } else dense_block(s, dec_dense, 2, X, B, F);
    for (int g = 0; g < 2; ++g) {
        if (dec16 && up == 2 && C * up <= 128) {       // statistics of the up-sampled map out of the sub-pixel convolution's own epilogue (a channel = `up` adjacent columns)
            const int nblk_u = (T * F + 127) / 128;
            if (dense_half) zip16::launch_rows16_stats<12, 4, true>(s, zip16::RowConv16{Dh16, 8 * C, g * 4 * C, T, F, F, 1}, up_w16[g], zip16::SubPixelStore16{U, up_b[g], 2 * C, g * C, up}, C * up, T * F, B, partial, up);
            else zip16::launch_rows16_stats<12, 4>(s, zip16::RowConv16{Dh16, 8 * C, g * 4 * C, T, F, F, 1}, up_w16[g], zip16::SubPixelStore16{U, up_b[g], 2 * C, g * C, up}, C * up, T * F, B, partial, up);
            hipLaunchKernelGGL(k_zip_stats_final, dim3((unsigned)B), dim3(256), 0, s, (const double*)partial, nblk_u, (double)(T * F2), up_g + g * C, up_beta + g * C, nrm2, 2 * C, g * C);
            continue;
        }
        if (dec16 && dense_half) zip16::launch_rows16<12, 4, true>(s, zip16::RowConv16{Dh16, 8 * C, g * 4 * C, T, F, F, 1}, up_w16[g], zip16::SubPixelStore16{U, up_b[g], 2 * C, g * C, up}, (int)R, C * up);
        else if (dec16) zip16::launch_rows16<12, 4>(s, zip16::RowConv16{Dh16, 8 * C, g * 4 * C, T, F, F, 1}, up_w16[g], zip16::SubPixelStore16{U, up_b[g], 2 * C, g * C, up}, (int)R, C * up);
        else gemm64::launch(s, RowConvA{Dh, 8 * C, g * 4 * C + (4 - depth) * C, C, T, F, F, 1}, gemm64::WeightB{up_w[g], 3 * C},
                            SubPixelStore{U, up_b[g], 2 * C, g * C, up}, (int)R, C * up, 3 * C);
        stats(s, U, 2 * C, g * C, T * F2, B, up_g + g * C, up_beta + g * C, nrm2, 2 * C, g * C);
    }
    hipLaunchKernelGGL(k_zip_heads, dim3((unsigned)((J + 15) / 16), (unsigned)kZF), dim3(256), 0, s, (const float*)U, (const float*)nrm2, up_slope, mask_w, mask_b,
                       phase_w, phase_b, packed, mask_tap, T, F2, C, J);
    gemm::launch(s, PlanarA{packed, J}, gemm::RowMajorB{k_inv, kZN}, gemm::BiasActStore<gemm::kActNone>{frames_buf, kZN, nullptr, 0.0f}, J, kZN, kZC2);
    const long long total = (long long)B * Lo;
    hipLaunchKernelGGL(k_zip_ola_pcm, flat(total), dim3(256), 0, s, (const float*)frames_buf, inv_wsum, (const float*)norm, d_out, d_f32, T, Lo, dynamic_norm ? 1 : 0, total);
    ZP_HIP(hipGetLastError());
    return ADE_OK;
}

// taps: "enc_in", "enc0" .. "enc3": (windows, T, F, C) after the dense encoder / each dual-path encoder (kept for calls of at most 8 windows); "mask": (windows, T, 201) raw mask head; "packed": (402, windows * T) synthesis input.
int ZipEngine::tap(hipStream_t s, const char* name, int batch, float* out, size_t count, size_t* written, std::string& err) {
    const size_t B = (size_t)batch * n_win, J = B * T;
    const float* src = nullptr;
    size_t n = 0;
    if (strcmp(name, "enc_in") == 0) { src = enc_tap[0]; n = J * F * C; }
    else if (strncmp(name, "enc", 3) == 0 && name[3] >= '0' && name[3] <= '3' && !name[4]) { src = enc_tap[1 + name[3] - '0']; n = J * F * C; }
    else if (strcmp(name, "mask") == 0) { src = mask_tap; n = J * kZF; }
    else if (strcmp(name, "packed") == 0) { src = packed; n = (size_t)kZC2 * J; }
    else if (strcmp(name, "spec") == 0) { src = spec; n = (size_t)kZC2 * J; }                     // (402, windows * T)
    else if (strcmp(name, "feat") == 0) { src = feat; n = J * kZF * 2; }                          // (windows, T, 201, {mag, pha})
    else if (strcmp(name, "e0") == 0) { src = E0; n = J * kZF * C; }                              // (windows, T, 201, C): dense-encoder input
    else if (strcmp(name, "dense") == 0) { src = Dh; n = J * F * 8 * C; }                         // the decoder pair's dense outputs, normalised + PReLU (windows, T, F, 8 C)
    else if (strcmp(name, "nrm") == 0) { src = nrm; n = B * 8 * C * 2; }
    else if (strncmp(name, "l0_", 3) == 0 && name[3] >= '0' && name[3] <= '7' && !name[4]) { src = (lt && B <= 8) ? lt + (size_t)(name[3] - '0') * lt_stride : nullptr; n = J * F * C; }   // ADE_ZIP_LAYER_TAPS=1
    else return zfail(err, ADE_ERR_NOT_FOUND, std::string("unknown tap: ") + name);
    // a bf16 handle keeps the dense layers' history and raw buffers in bf16 (Dh16 / E016; Dh is 64 floats and E0 the last layer's raw output there): these two fp32 taps do not exist on it
    if (bf16 && parts16 == 7 && (strcmp(name, "e0") == 0 || strcmp(name, "dense") == 0))
        return zfail(err, ADE_ERR_UNSUPPORTED, std::string("tap \"") + name + "\" exists on f32 handles only (ade_gemm_dtype = bf16 keeps this tensor in bf16)");
    if (strncmp(name, "enc", 3) == 0 && B > 8) src = nullptr;
    if (!src || batch <= 0)
        return zfail(err, ADE_ERR_NOT_FOUND, "tap has no data (the encoder snapshots are taken by calls of at most 8 windows, whatever capacity was reserved)");
    if (count < n) return zfail(err, ADE_ERR_SHAPE_MISMATCH, "tap buffer too small");
    ZP_HIP(hipStreamSynchronize(s));
    ZP_HIP(hipMemcpy(out, src, n * sizeof(float), hipMemcpyDeviceToHost));
    *written = n;
    return ADE_OK;
}

}  // namespace ade
