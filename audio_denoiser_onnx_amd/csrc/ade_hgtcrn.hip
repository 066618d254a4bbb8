// ade_hgtcrn.hip — H-GTCRN (two-microphone hybrid front-end + GTCRN) for gfx950, as a sub-engine of libade.
//
// Mirrors H_GTCRN_CUSTOM.forward (H-GTCRN/Export_H_GTCRN.py:941-1063):
//   int16 (2, L) -> /32768, minus the mean of the whole call -> [fold into windows] -> STFT 512 / 256 periodic hann, reflect (generic plan,
//   ade_stft.hip) -> WPE dereverberation: per (window, bin) a 36 x 36 complex normal-equation system built over the frames and solved with
//   6 conjugate-gradient steps (:581-757, :499-555) -> AuxIVA: 10 iterations, each a cross-bin source-activity reduction and a per-bin
//   2 x 2 update with Cramer solves (:760-900, :557-598), projection back to microphone 0 -> six features (two spectra, the two separated
//   log-magnitudes ordered by energy) -> GTCRN_IVA network (:428-494): ERB merge + SFE + Conv(18 -> 16) here, every later block on GTCRN's
//   own kernels (ade_kernels.hip) -> ERB split + complex ratio mask on microphone 0 -> ISTFT -> x32767, NaN -> 0, clamp, int16.
//
// One workgroup per (window, bin) for WPE (the system lives in LDS), one wavefront per (window, bin) for the AuxIVA updates; the only
// cross-bin step (the source activity r) is its own small kernel, so an AuxIVA iteration is two launches.  All sums run in a fixed order:
// results do not depend on the batch size.  NOTE: the fp32 conjugate-gradient solve is ill-conditioned for some bins -- the
// reference's own fp32 and fp64 runs differ there by O(1) -- so parity of the WPE stage is defined on the well-conditioned bins, and
// everything downstream is pinned by continuing the oracle from this engine's WPE output (tests/test_hgtcrn.py).
#include "ade_internal.h"
#include "ade_gtcrn_pack.h"
#include "ade_device.h"

#include "../../include/ade.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace ade {
namespace {

using dev::mk2;
using dev::v2f;

constexpr int kHBins = 257, kHNfft = 512, kHHop = 256;
constexpr int kLg = 18, kDelay = 2, kTaps = 2 * kLg;      // int(0.3 * 16000 / 256) taps per microphone, prediction delay (:50-52, :610-614)
constexpr int kCgIter = 6, kIvaIter = 10;                // (:53-54)
constexpr int kMaxFrames = 1024;

__global__ __launch_bounds__(256) void k_hg_pcm2f(const int16_t* __restrict__ pcm, const float* __restrict__ mean, float* __restrict__ x, int W, int n_win,
                                                  long long total) {
    // in: (call, 2, n_win * W) ; out row (call * n_win + win) * 2 + ch (:972-981)
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int n = (int)(i % W);
    const long long row = i / W;
    const int ch = (int)(row & 1);
    const long long wn = row >> 1;
    const long long call = wn / n_win;
    const int win = (int)(wn - call * n_win);
    const float v = (float)pcm[(call * 2 + ch) * (long long)n_win * W + (long long)win * W + n] * (1.0f / 32768.0f);
    x[i] = v - mean[call];
}

// the resampled entry (:953-970): x = float_in / 32768 - mean.  The mean is that of the tensor the reference centres: the interpolated one when
// the input rate is above the model rate (mean_src = that float tensor), the caller-rate PCM otherwise (interpolation commutes with the shift).
__global__ __launch_bounds__(256) void k_hg_mean_f32(const float* __restrict__ x, long long n, float* __restrict__ mean, float gain = 1.0f) {
    __shared__ double part[256];
    const float* base = x + (size_t)blockIdx.x * n;
    double s = 0.0;
    for (long long i = threadIdx.x; i < n; i += 256) s += (double)base[i];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = 0.0;
        for (int i = 0; i < 256; ++i) tot += part[i];
        mean[blockIdx.x] = (float)(tot * (double)gain / ((double)n * 32768.0));
    }
}
__global__ __launch_bounds__(256) void k_hg_f2f(const float* __restrict__ in, const float* __restrict__ mean, float* __restrict__ x, int W, int n_win, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int n = (int)(i % W);                                                // the same gather as k_hg_pcm2f (a float input tensor on a batch-fold export)
    const long long row = i / W;
    const int ch = (int)(row & 1);
    const long long wn = row >> 1;
    const long long call = wn / n_win;
    const int win = (int)(wn - call * n_win);
    x[i] = in[(call * 2 + ch) * (long long)n_win * W + (long long)win * W + n] * (1.0f / 32768.0f) - mean[call];
}

// eps[b] = 1e-3 * mean_f( max_{m,t} |X|^2 )  (:690-691)
__global__ __launch_bounds__(256) void k_hg_wpe_eps(const float* __restrict__ spec, float* __restrict__ eps, int T) {
    __shared__ float part[256];
    const int b = blockIdx.x;
    float acc = 0.0f;
    for (int f = threadIdx.x; f < kHBins; f += 256) {
        float mx = 0.0f;
        for (int m = 0; m < 2; ++m) {
            const float* re = spec + ((size_t)(b * 2 + m) * 2 * kHBins + f) * T;
            const float* im = re + (size_t)kHBins * T;
            for (int t = 0; t < T; ++t) mx = fmaxf(mx, re[t] * re[t] + im[t] * im[t]);
        }
        acc += mx;
    }
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) part[threadIdx.x] += part[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) eps[b] = 1e-3f * (part[0] / (float)kHBins);
}

// A butterfly leaves every lane with the sum of the same 64 numbers but in a lane-dependent association, i.e. with lane-dependent
// rounding; the demixing algebra below must be identical in all lanes (each lane applies W to its own frames), so lane 0's sum is broadcast.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return __shfl(v, 0, 64);
}

// WPE, one iteration (:686-757).  grid (257, B), 256 threads, dynamic LDS: X (4 T) | 1 / lambda (T) | R (2 x 36 x 36) | P, x, p (2 x 72 each).  Delay-bank row k = l * 2 + m is microphone m delayed kDelay + l frames (:627-684).
__global__ __launch_bounds__(256, 8) void k_hg_wpe(const float* __restrict__ spec, const float* __restrict__ eps_b, float* __restrict__ out, int T) {
    HIP_DYNAMIC_SHARED(float, lds)
    float* Xr = lds;                 // [2][T]
    float* Xi = Xr + 2 * T;
    float* il = Xi + 2 * T;          // [T]
    float* Rr = il + T;              // [36][36]
    float* Ri = Rr + kTaps * kTaps;
    float* Pr = Ri + kTaps * kTaps;  // [36][2]
    float* Pi = Pr + 72;
    float* xr = Pi + 72;
    float* xi = xr + 72;
    float* pr = xi + 72;
    float* pi = pr + 72;
    const int f = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const float eps = eps_b[b];
    for (int i = tid; i < 2 * T; i += 256) {
        const int m = i / T, t = i - m * T;
        const size_t base = ((size_t)(b * 2 + m) * 2 * kHBins + f) * T + t;
        Xr[i] = spec[base];
        Xi[i] = spec[base + (size_t)kHBins * T];
    }
    __syncthreads();
    for (int t = tid; t < T; t += 256) {
        const float p0 = Xr[t] * Xr[t] + Xi[t] * Xi[t], p1 = Xr[T + t] * Xr[T + t] + Xi[T + t] * Xi[T + t];
        il[t] = 1.0f / fmaxf((p0 + p1) * 0.5f, eps);                         // lambda = clamp(mean_m |Y|^2, eps), Y = X on the only iteration
    }
    __syncthreads();
    // R = (D / lambda) D^H + eps I ; P = (D / lambda) X^H     (:706-718).  One thread per 2 x 2 block (the two microphones of tap l against the
    // two of tap l' >= l, or against the undelayed pair for P): nine LDS reads feed 32 multiply-adds.  R is Hermitian; the lower blocks are
    // the conjugate transposes of the upper ones.
    constexpr int kBlocks = kLg * (kLg + 1) / 2;
    for (int e = tid; e < kBlocks + kLg; e += 256) {
        int l = 0, l2 = 0;
        const bool isR = e < kBlocks;
        if (isR) {
            int rem = e;
            while (rem >= kLg - l) { rem -= kLg - l; ++l; }
            l2 = l + rem;
        } else l = e - kBlocks;
        const int si = kDelay + l, sj = isR ? kDelay + l2 : 0;
        // (round 5) the 32 multiply-adds of a frame as 16 packed ones: (a_rr, a_ii) += (dr, di) * (er, ei) and (a_ir, a_ri) += (di, dr) * (er, ei) -- the same products, the same
        // running sums, half the instructions (the kernel is VALU-issue-bound: six wavefronts per SIMD at 189 busy lanes of 256; 2.10 ms per 256 windows before)
        v2f a_d[4], a_x[4];             // a_d = (a_rr, a_ii), a_x = (a_ir, a_ri)
#pragma unroll
        for (int q = 0; q < 4; ++q) { a_d[q] = mk2(0.0f, 0.0f); a_x[q] = mk2(0.0f, 0.0f); }
        for (int t = si > sj ? si : sj; t < T; ++t) {
            const float w = il[t];
            const v2f d[2] = {mk2(Xr[t - si] * w, Xi[t - si] * w), mk2(Xr[T + t - si] * w, Xi[T + t - si] * w)};         // (dr, di) per microphone
            const v2f e[2] = {mk2(Xr[t - sj], Xi[t - sj]), mk2(Xr[T + t - sj], Xi[T + t - sj])};                         // (er, ei)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                const v2f ds = mk2(d[mi][1], d[mi][0]);                                                                   // (di, dr)
#pragma unroll
                for (int mj = 0; mj < 2; ++mj) {
                    a_d[mi * 2 + mj] += d[mi] * e[mj];
                    a_x[mi * 2 + mj] += ds * e[mj];
                }
            }
        }
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int mj = 0; mj < 2; ++mj) {
                const float re = a_d[mi * 2 + mj][0] + a_d[mi * 2 + mj][1], im = a_x[mi * 2 + mj][0] - a_x[mi * 2 + mj][1];
                const int i = 2 * l + mi;
                if (isR) {
                    const int j = 2 * l2 + mj;
                    Rr[i * kTaps + j] = re + (i == j ? eps : 0.0f);
                    Ri[i * kTaps + j] = im;
                    if (l2 != l) { Rr[j * kTaps + i] = re; Ri[j * kTaps + i] = -im; }
                } else { Pr[i * 2 + mj] = re; Pi[i * 2 + mj] = im; }
            }
    }
    __syncthreads();
    // conjugate gradient (:499-555): wavefront c solves right-hand side c; lane i owns entry i of x, r, p (registers), p is mirrored in LDS for
    // the matrix-vector product (row i of R read as column i of the Hermitian R: consecutive lanes, consecutive words), the dot products are
    // wavefront reductions; one barrier per step.
    {
        const int c = (tid >> 6) & 1, lane = tid & 63;
        const bool act = tid < 128 && lane < kTaps;
        const int i = lane < kTaps ? lane : kTaps - 1;
        float x_r = 0.0f, x_i = 0.0f;
        float r_r = act ? Pr[i * 2 + c] : 0.0f, r_i = act ? Pi[i * 2 + c] : 0.0f;
        float p_r = r_r, p_i = r_i;
        if (act) { pr[i * 2 + c] = p_r; pi[i * 2 + c] = p_i; }
        float rr = wave_sum(r_r * r_r + r_i * r_i) + 1e-12f;
        __syncthreads();
        for (int it = 0; it < kCgIter; ++it) {
            float s1 = 0.0f, s2 = 0.0f, s3 = 0.0f, s4 = 0.0f;
            if (act) {
                for (int k = 0; k < kTaps; ++k) {
                    const float a = Rr[k * kTaps + i], bb = -Ri[k * kTaps + i], u = pr[k * 2 + c], v = pi[k * 2 + c];
                    s1 += a * u; s2 += bb * v; s3 += a * v; s4 += bb * u;
                }
            }
            const float a_r = s1 - s2, a_i = s3 + s4;
            const float pAp = wave_sum(p_r * a_r + p_i * a_i) + 1e-12f;
            const float alpha = rr / pAp;
            x_r += alpha * p_r; x_i += alpha * p_i;
            r_r -= alpha * a_r; r_i -= alpha * a_i;
            const float rr_new = wave_sum(r_r * r_r + r_i * r_i) + 1e-12f;
            const float beta = rr_new / rr;
            p_r = r_r + beta * p_r; p_i = r_i + beta * p_i;
            rr = rr_new;
            __syncthreads();                                                   // every lane has read the old p
            if (act) { pr[i * 2 + c] = p_r; pi[i * 2 + c] = p_i; }
            __syncthreads();
        }
        if (act) { xr[i * 2 + c] = x_r; xi[i * 2 + c] = x_i; }
    }
    __syncthreads();
    // Y = X - conj(G)^T D    (:741-750)
    for (int i = tid; i < 2 * T; i += 256) {
        const int m = i / T, t = i - m * T;
        float s1 = 0.0f, s2 = 0.0f, s3 = 0.0f, s4 = 0.0f;
        for (int k = 0; k < kTaps; ++k) {
            const int sh = kDelay + (k >> 1), mk = k & 1;
            if (t < sh) break;                                                 // rows are ordered by delay: every later row is zero here too
            const float gr = xr[k * 2 + m], gi = -xi[k * 2 + m];
            const float dr = Xr[mk * T + t - sh], di = Xi[mk * T + t - sh];
            s1 += gr * dr; s2 += gi * di; s3 += gi * dr; s4 += gr * di;
        }
        const size_t base = ((size_t)(b * 2 + m) * 2 * kHBins + f) * T + t;
        out[base] = Xr[i] - (s1 - s2);
        out[base + (size_t)kHBins * T] = Xi[i] - (s3 + s4);
    }
}

// AuxIVA source activity (:814-818): rinv[b][m][t] = 1 / (2 sqrt(sum_f |Y|^2 + 1e-10)).  grid (ceil(T / 32), B * 2); 8 groups of bins x 32 frames
// per workgroup, the eight partial sums added in a fixed order.
__global__ __launch_bounds__(256) void k_hg_iva_r(const float* __restrict__ Y, float* __restrict__ rinv, int T) {
    __shared__ float part[8][32];
    const int tt = threadIdx.x & 31, g = threadIdx.x >> 5, t = blockIdx.x * 32 + tt, row = blockIdx.y;      // row = b * 2 + m
    float s = 0.0f;
    if (t < T) {
        const float* re = Y + (size_t)row * 2 * kHBins * T + t;
        const float* im = re + (size_t)kHBins * T;
        for (int f = g; f < kHBins; f += 8) s += re[(size_t)f * T] * re[(size_t)f * T] + im[(size_t)f * T] * im[(size_t)f * T];
    }
    part[g][tt] = s;
    __syncthreads();
    if (g == 0 && t < T) {
        float tot = part[0][tt];
#pragma unroll
        for (int k = 1; k < 8; ++k) tot += part[k][tt];
        rinv[(size_t)row * T + t] = 1.0f / (2.0f * sqrtf(tot + 1e-10f));
    }
}

// One AuxIVA iteration for one (window, bin): both source updates (:820-878) and Y = W X (:880-884).  One wavefront per bin, lanes over frames.
// Wst: [b][f][8] = W real (2 x 2) | W imaginary (2 x 2).
__global__ __launch_bounds__(256) void k_hg_iva_step(const float* __restrict__ X, const float* __restrict__ rinv, float* __restrict__ Wst, float* __restrict__ Y, int T,
                                                     int first) {
    const int lane = threadIdx.x & 63, f = blockIdx.x * 4 + (threadIdx.x >> 6), b = blockIdx.y;
    if (f >= kHBins) return;
    const float* x0r = X + ((size_t)(b * 2) * 2 * kHBins + f) * T;
    const float* x0i = x0r + (size_t)kHBins * T;
    const float* x1r = x0r + (size_t)2 * kHBins * T;
    const float* x1i = x1r + (size_t)kHBins * T;
    const float* r0 = rinv + (size_t)(b * 2) * T;
    const float* r1 = r0 + T;
    // the demixing matrix of the previous iteration, read by every lane before the reductions below (lane 0 overwrites it at the end)
    float Wr[4], Wi[4];
    float* wst = Wst + ((size_t)b * kHBins + f) * 8;
    if (first) { Wr[0] = 1.0f; Wr[1] = 0.0f; Wr[2] = 0.0f; Wr[3] = 1.0f; Wi[0] = Wi[1] = Wi[2] = Wi[3] = 0.0f; }
    else {
#pragma unroll
        for (int e = 0; e < 4; ++e) { Wr[e] = wst[e]; Wi[e] = wst[4 + e]; }
    }
    // V_s = (w_s X) X^H / T, all four entries of the real and imaginary parts as the reference's two matmuls form them (:829-831)
    float vr[2][4] = {}, vi[2][4] = {};
    for (int t = lane; t < T; t += 64) {
        const float xr[2] = {x0r[t], x1r[t]}, xi[2] = {x0i[t], x1i[t]};
        const float w[2] = {r0[t], r1[t]};
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float ar = xr[i] * w[s], ai = xi[i] * w[s];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    vr[s][i * 2 + j] += ar * xr[j] + ai * xi[j];
                    vi[s][i * 2 + j] += ai * xr[j] - ar * xi[j];
                }
            }
    }
    const float inv_t = 1.0f / (float)T;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int e = 0; e < 4; ++e) { vr[s][e] = wave_sum(vr[s][e]) * inv_t; vi[s][e] = wave_sum(vi[s][e]) * inv_t; }
    const float eps = 1e-10f;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const float* Vr = vr[s];
        const float* Vi = vi[s];
        float ar[4], ai[4];                                                     // WV = W V (+ eps on the real diagonal)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                ar[i * 2 + j] = (Wr[i * 2] * Vr[j] + Wr[i * 2 + 1] * Vr[2 + j]) - (Wi[i * 2] * Vi[j] + Wi[i * 2 + 1] * Vi[2 + j]);
                ai[i * 2 + j] = (Wr[i * 2] * Vi[j] + Wr[i * 2 + 1] * Vi[2 + j]) + (Wi[i * 2] * Vr[j] + Wi[i * 2 + 1] * Vr[2 + j]);
            }
        ar[0] += eps; ar[3] += eps;
        // Cramer (:557-598): a = [0], b = [1], c = [2], d = [3]; right-hand side e_s
        const float det_r = (ar[0] * ar[3] - ai[0] * ai[3]) - (ar[1] * ar[2] - ai[1] * ai[2]);
        const float det_i = (ar[0] * ai[3] + ai[0] * ar[3]) - (ar[1] * ai[2] + ai[1] * ar[2]);
        const float q = 1.0f / ((det_r * det_r + det_i * det_i) + 1e-12f);
        const float ir = det_r * q, ii = -det_i * q;
        float n0r, n0i, n1r, n1i;
        if (s == 0) { n0r = ar[3]; n0i = ai[3]; n1r = -ar[2]; n1i = -ai[2]; }
        else { n0r = -ar[1]; n0i = -ai[1]; n1r = ar[0]; n1i = ai[0]; }
        const float wr0 = n0r * ir - n0i * ii, wi0 = n0r * ii + n0i * ir;
        const float wr1 = n1r * ir - n1i * ii, wi1 = n1r * ii + n1i * ir;
        // denom = conj(w)^T V w (real part) (:862-866)
        const float vw0r = (Vr[0] * wr0 + Vr[1] * wr1) - (Vi[0] * wi0 + Vi[1] * wi1), vw0i = (Vr[0] * wi0 + Vr[1] * wi1) + (Vi[0] * wr0 + Vi[1] * wr1);
        const float vw1r = (Vr[2] * wr0 + Vr[3] * wr1) - (Vi[2] * wi0 + Vi[3] * wi1), vw1i = (Vr[2] * wi0 + Vr[3] * wi1) + (Vi[2] * wr0 + Vi[3] * wr1);
        const float den = (wr0 * vw0r + wi0 * vw0i) + (wr1 * vw1r + wi1 * vw1i);
        const float scl = 1.0f / sqrtf(fmaxf(den, 0.0f) + eps);
        Wr[s * 2] = wr0 * scl; Wr[s * 2 + 1] = wr1 * scl;
        Wi[s * 2] = -wi0 * scl; Wi[s * 2 + 1] = -wi1 * scl;
    }
    if (lane == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { wst[e] = Wr[e]; wst[4 + e] = Wi[e]; }
    }
    float* y0r = Y + ((size_t)(b * 2) * 2 * kHBins + f) * T;
    float* y0i = y0r + (size_t)kHBins * T;
    float* y1r = y0r + (size_t)2 * kHBins * T;
    float* y1i = y1r + (size_t)kHBins * T;
    for (int t = lane; t < T; t += 64) {
        const float ar0 = x0r[t], ai0 = x0i[t], ar1 = x1r[t], ai1 = x1i[t];
        y0r[t] = (Wr[0] * ar0 + Wr[1] * ar1) - (Wi[0] * ai0 + Wi[1] * ai1);
        y0i[t] = (Wr[0] * ai0 + Wr[1] * ai1) + (Wi[0] * ar0 + Wi[1] * ar1);
        y1r[t] = (Wr[2] * ar0 + Wr[3] * ar1) - (Wi[2] * ai0 + Wi[3] * ai1);
        y1i[t] = (Wr[2] * ai0 + Wr[3] * ai1) + (Wi[2] * ar0 + Wi[3] * ar1);
    }
}

// Projection back onto microphone 0 (:886-898), in place; epart[b][m][f] = sum_t |Y_out|^2 for the energy ordering (:1000-1003).
__global__ __launch_bounds__(256) void k_hg_iva_project(const float* __restrict__ X, float* __restrict__ Y, float* __restrict__ epart, int T) {
    const int lane = threadIdx.x & 63, f = blockIdx.x * 4 + (threadIdx.x >> 6), b = blockIdx.y;
    if (f >= kHBins) return;
    const float* xr = X + ((size_t)(b * 2) * 2 * kHBins + f) * T;
    const float* xi = xr + (size_t)kHBins * T;
    for (int m = 0; m < 2; ++m) {
        float* yr = Y + ((size_t)(b * 2 + m) * 2 * kHBins + f) * T;
        float* yi = yr + (size_t)kHBins * T;
        float nr = 0.0f, ni = 0.0f, dn = 0.0f;
        for (int t = lane; t < T; t += 64) {
            nr += xr[t] * yr[t] + xi[t] * yi[t];
            ni += xr[t] * yi[t] - xi[t] * yr[t];
            dn += yr[t] * yr[t] + yi[t] * yi[t];
        }
        nr = wave_sum(nr); ni = wave_sum(ni); dn = wave_sum(dn);
        const bool valid = dn > 0.0f;
        const float safe = 1.0f / (valid ? dn : 1.0f);
        const float cr = valid ? nr * safe : 1.0f, ci = valid ? ni * safe : 0.0f;
        float e = 0.0f;
        for (int t = lane; t < T; t += 64) {
            const float a = yr[t], c = yi[t];
            const float orr = cr * a + ci * c, oi = cr * c - ci * a;
            yr[t] = orr; yi[t] = oi;
            e += orr * orr + oi * oi;
        }
        e = wave_sum(e);
        if (lane == 0) epart[((size_t)b * 2 + m) * kHBins + f] = e;
    }
}

__global__ __launch_bounds__(64) void k_hg_pred(const float* __restrict__ epart, int* __restrict__ pred, int B) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    float e0 = 0.0f, e1 = 0.0f;
    for (int f = 0; f < kHBins; ++f) { e0 += epart[((size_t)b * 2) * kHBins + f]; e1 += epart[((size_t)b * 2 + 1) * kHBins + f]; }
    pred[b] = e0 < e1 ? 1 : 0;
}

// Six features, ERB-merged (:1005-1024, ERB.bm :125-128): feat[frame][c][132]; c = re0, im0, re1, im1, selected log-magnitude, the other one.
__global__ __launch_bounds__(256) void k_hg_feat(const float* __restrict__ spec, const float* __restrict__ iva, const int* __restrict__ pred, BandTab bm,
                                                 float* __restrict__ feat, int T, long long total) {
    // frames fastest: the spectra are read along their contiguous axis (the 132-float feature rows are written strided instead)
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int t = (int)(i % T);
    long long q = i / T;
    const int e = (int)(q % kErb);
    q /= kErb;
    const int c = (int)(q % 6);
    const long long b = q / 6;
    const long long frame = b * T + t;
    const float* src;
    bool logm = false;
    if (c < 4) src = spec + ((size_t)(b * 2 + (c >> 1)) * 2 + (c & 1)) * kHBins * T + t;
    else {
        const int p = pred[b];
        const int m = (c == 4) ? (p ? 0 : 1) : (p ? 1 : 0);           // where(pred, log_0, log_1) / where(pred, log_1, log_0)
        src = iva + ((size_t)(b * 2 + m) * 2) * kHBins * T + t;
        logm = true;
    }
    auto val = [&](int f) {
        if (!logm) return src[(size_t)f * T];
        const float re = src[(size_t)f * T], im = src[((size_t)kHBins + f) * T];
        return 0.5f * log10f(fmaxf(re * re + im * im, 1e-24f));
    };
    float v;
    if (e < kErbLow) v = val(e);
    else {
        const int o = e - kErbLow, st = bm.start[o];
        v = 0.0f;
        for (int n = 0; n < bm.count; ++n) {
            const float w = bm.w[(size_t)n * bm.n_out + o];
            if (w != 0.0f) v += val(kErbLow + st + n) * w;
        }
    }
    feat[(size_t)frame * 6 * kErbPad + c * kErbPad + e] = v;
}

// SFE(3) + Conv2d(18 -> 16, (1,5), stride (1,2), pad (0,2)) + folded BN + PReLU (:388-389); one lane per (frame, fo).  w: [k][ci = c*3+o][co].
__global__ __launch_bounds__(256) void k_hg_conv0(const float* __restrict__ feat, const float* __restrict__ w, const float* __restrict__ bias, float slope,
                                                  float* __restrict__ e0, int nframes) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= nframes * kF1) return;
    const int frame = idx / kF1, fo = idx - frame * kF1;
    const float* fr = feat + (size_t)frame * 6 * kErbPad;
    float acc[16];
#pragma unroll
    for (int co = 0; co < 16; ++co) acc[co] = bias[co];
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        float v[7];
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const int q = 2 * fo - 3 + j;
            v[j] = (q >= 0 && q < kErb) ? fr[c * kErbPad + q] : 0.0f;
        }
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int p = 2 * fo - 2 + k;
            const bool pv = p >= 0 && p < kErb;
#pragma unroll
            for (int o = 0; o < 3; ++o) {
                const float x = pv ? v[k + o] : 0.0f;
#pragma unroll
                for (int co = 0; co < 16; ++co) acc[co] += w[(k * 18 + c * 3 + o) * 16 + co] * x;
            }
        }
    }
#pragma unroll
    for (int co = 0; co < 16; ++co) acc[co] = acc[co] >= 0.0f ? acc[co] : acc[co] * slope;
    float4* dst = reinterpret_cast<float4*>(e0 + (size_t)idx * kCh);
#pragma unroll
    for (int i = 0; i < 4; ++i) dst[i] = make_float4(acc[4 * i], acc[4 * i + 1], acc[4 * i + 2], acc[4 * i + 3]);
}

// ERB split + complex ratio mask on microphone 0 (ERB.bs :130-133, :486-493): out (B, 514, T) for the synthesis plan.
__global__ __launch_bounds__(256) void k_hg_mask(const float* __restrict__ mask, const float* __restrict__ spec, BandTab bs, float* __restrict__ out, int T, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int t = (int)(i % T);
    const long long q = i / T;
    const int f = (int)(q % kHBins);
    const long long b = q / kHBins;
    const float* mr = mask + ((size_t)b * T + t) * 2 * kErbPad;
    float m[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        if (f < kErbLow) m[c] = mr[c * kErbPad + f];
        else {
            const int o = f - kErbLow, st = bs.start[o];
            float v = 0.0f;
            for (int n = 0; n < bs.count; ++n) {
                const float w = bs.w[(size_t)n * bs.n_out + o];
                if (w != 0.0f) v += mr[c * kErbPad + kErbLow + st + n] * w;
            }
            m[c] = v;
        }
    }
    const float re = spec[((size_t)(b * 2) * 2 * kHBins + f) * T + t], im = spec[((size_t)(b * 2) * 2 * kHBins + kHBins + f) * T + t];
    out[((size_t)b * 2 * kHBins + f) * T + t] = re * m[0] - im * m[1];
    out[((size_t)b * 2 * kHBins + kHBins + f) * T + t] = im * m[0] + re * m[1];
}

// x32767, NaN -> 0, clamp, truncate (:1045-1058)
__global__ __launch_bounds__(256) void k_hg_f2pcm(const float* __restrict__ y, int16_t* __restrict__ pcm, float* __restrict__ f32, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const float w = y[i] == y[i] ? y[i] : 0.0f;                      // the NaN guard sits before every output dtype's return (:1054-1060)
    if (f32) f32[i] = w;
    float v = y[i] * 32767.0f;
    if (v != v) v = 0.0f;
    if (pcm) pcm[i] = (int16_t)(int)fminf(fmaxf(v, -32768.0f), 32767.0f);
}

int hfail(std::string& err, int st, const std::string& msg) { err = msg; return st; }
#define HG_HIP(expr)                                                                                  \
    do {                                                                                              \
        hipError_t _e = (expr);                                                                       \
        if (_e != hipSuccess) return hfail(err, ADE_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

struct MapLoader {
    const std::map<std::string, Tensor>& tensors;
    std::string& err;
    int st = ADE_OK;
    const float* get(const std::string& name, std::initializer_list<int> dims) {
        auto it = tensors.find(name);
        if (it == tensors.end()) {
            if (st == ADE_OK) { st = ADE_ERR_MISSING_KEY; err = "weights: tensor missing: " + name; }
            return nullptr;
        }
        if (it->second.dims != std::vector<int>(dims)) {
            if (st == ADE_OK) { st = ADE_ERR_SHAPE_MISMATCH; err = "weights: tensor has the wrong shape: " + name; }
            return nullptr;
        }
        return it->second.data;
    }
};

}  // namespace

// channels-last (B, P, 16) <-> quad-planar (B, 4, P, 4): thread = one float4 (window b, channel quad q, position p); to_planar = 1: cl -> pl, 0: pl -> cl
__global__ __launch_bounds__(256) void k_hg_relayout(const float* __restrict__ src, float* __restrict__ dst, int P, int to_planar, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const long long b = i / (4LL * P);
    const int rem = (int)(i - b * 4LL * P), q = rem / P, p = rem - q * P;
    const size_t pl = ((size_t)(b * 4 + q) * P + p) * 4, cl = ((size_t)b * P + p) * 16 + 4 * q;
    if (to_planar) *reinterpret_cast<float4*>(dst + pl) = *reinterpret_cast<const float4*>(src + cl);
    else *reinterpret_cast<float4*>(dst + cl) = *reinterpret_cast<const float4*>(src + pl);
}

struct HgtcrnEngine : SubEngine {
    int device = 0, W = 0 /* one window */, n_win = 1, T = 0, out_len_ = 0;
    ade_stft_handle plan = nullptr;
    float* d_w = nullptr;
    int* d_ints = nullptr;
    BandTab erb_bm{}, erb_bs{};
    ConvW en0{}, en1{}, de3{}, de4{};
    GtConvW en_gt[3]{}, de_gt[3]{};
    DpW dp[2]{};
    int capacity = 0;
    float* ws = nullptr;
    float *mean = nullptr, *xf = nullptr, *spec = nullptr, *drb = nullptr, *iva = nullptr, *eps = nullptr, *rinv = nullptr, *wst = nullptr, *epart = nullptr, *feat = nullptr,
          *e0 = nullptr, *e1 = nullptr, *h = nullptr, *zt = nullptr, *xe[3] = {}, *ate[3] = {}, *xd[3] = {}, *atd[3] = {}, *rnn = nullptr, *dpm = nullptr, *dpo[2] = {},
          *d3 = nullptr, *mask = nullptr, *sout = nullptr, *yf = nullptr;
    int* pred = nullptr;
    // (round 5) the network's middle -- three GTConvBlocks, two DPGRNNs, three GTConvBlocks -- on GTCRN's FUSED per-stage kernels (csrc/ade_fused.hip: LDS-resident segments of 16
    // frames that hand their recurrent states on through the exchange area) instead of the 26-launch multi-kernel sequence; ADE_HG_FUSED=0 keeps that sequence
    bool fused_net = !(getenv("ADE_HG_FUSED") && atoi(getenv("ADE_HG_FUSED")) == 0);
    int fgeo = -1, fseg = 0;              // workgroup geometry / segments per window of the fused stages (-1: the frame count does not fit: multi-kernel)
    float* d_xchg = nullptr;
    unsigned* d_xflags = nullptr;
    int* d_xerr = nullptr;               // page-locked: a bounded inter-workgroup wait that gave up leaves its code here (read by the engine after its synchronise: exchange_error_and_reset)
    int xwait_ticks = 20000000;          // bound of one inter-workgroup wait, 10 ns ticks (the engine's option "xwait_ms")

    ~HgtcrnEngine() override {
        (void)hipSetDevice(device);
        if (d_xchg) (void)hipFree(d_xchg);
        if (d_xflags) (void)hipFree(d_xflags);
        if (d_xerr) (void)hipHostFree(d_xerr);
        if (plan) ade_stft_destroy(plan);
        if (d_w) (void)hipFree(d_w);
        if (d_ints) (void)hipFree(d_ints);
        if (ws) (void)hipFree(ws);
    }
    int exchange_error_and_reset() override {
        if (!d_xerr) return 0;
        const int code = *(volatile int*)d_xerr;
        if (!code) return 0;
        (void)hipSetDevice(device);
        (void)hipDeviceSynchronize();                 // (the workgroups of the failed launch run on for up to their own bounds)
        d_xerr[0] = 0;
        if (d_xflags && capacity > 0 && fseg > 0) (void)hipMemset(d_xflags, 0, (size_t)capacity * n_win * fseg * kXFlags * sizeof(unsigned));
        return code;
    }
    void set_exchange_wait_ticks(int t) override { if (t > 0) xwait_ticks = t; }
    int frames() const override { return T; }
    int in_len() const override { return W * n_win; }
    int out_len() const override { return out_len_ * n_win; }
    int channels() const override { return 2; }
    int out_channels() const override { return 1; }
    bool accepts_float_input() const override { return true; }
    int reserve(int batch, std::string& err) override;
    int run(hipStream_t s, const int16_t* d_in, int batch, int16_t* d_out, float* d_f32, std::string& err) override;
    int tap(hipStream_t s, const char* name, int batch, float* out, size_t count, size_t* written, std::string& err) override;
};

int hgtcrn_create(const std::map<std::string, Tensor>& tensors, int window_len, int n_win, bool dynamic, int device, SubEngine** out, std::string& err) {
    *out = nullptr;
    if (window_len < kHNfft || window_len % kHHop) return hfail(err, ADE_ERR_SHAPE_MISMATCH, "h_gtcrn: the window must be whole 256-sample hops and at least one 512-sample frame");
    if (n_win < 1 || (dynamic && n_win != 1)) return hfail(err, ADE_ERR_BAD_VALUE, "h_gtcrn: bad fold (batch folding requires a static shape)");
    const int T = window_len / kHHop + 1;
    if (T > kMaxFrames) return hfail(err, ADE_ERR_SHAPE_MISMATCH, "h_gtcrn: more than 1024 frames per window");
    MapLoader L{tensors, err};
    Arena A;
    const float* erb_t = L.get("erb.erb_weight_t", {kErbHigh, kErbBands});
    const float* ierb_t = L.get("erb.ierb_weight_t", {kErbBands, kErbHigh});
    const float* w0 = L.get("encoder.en_convs.0.conv.weight", {16, 18, 1, 5});
    const float* b0 = L.get("encoder.en_convs.0.conv.bias", {16});
    const float* a0 = L.get("encoder.en_convs.0.act.weight", {1});
    const float* w1 = L.get("encoder.en_convs.1.conv.weight", {16, 8, 1, 5});
    const float* b1 = L.get("encoder.en_convs.1.conv.bias", {16});
    const float* a1 = L.get("encoder.en_convs.1.act.weight", {1});
    const float* w3 = L.get("decoder.de_convs.3.conv.weight", {16, 8, 1, 5});
    const float* b3 = L.get("decoder.de_convs.3.conv.bias", {16});
    const float* a3 = L.get("decoder.de_convs.3.act.weight", {1});
    const float* w4 = L.get("decoder.de_convs.4.conv.weight", {16, 2, 1, 5});
    const float* b4 = L.get("decoder.de_convs.4.conv.bias", {2});
    if (L.st != ADE_OK) return L.st;
    std::vector<int> bm_start, bs_start;
    std::vector<float> bm_w, bs_w;
    int bm_count = 0, bs_count = 0;
    band_table(erb_t, kErbHigh, kErbBands, bm_start, bm_w, bm_count);
    band_table(ierb_t, kErbBands, kErbHigh, bs_start, bs_w, bs_count);
    const size_t o_bm = A.alloc(bm_w.size()), o_bs = A.alloc(bs_w.size());
    memcpy(&A.f[o_bm], bm_w.data(), bm_w.size() * 4);
    memcpy(&A.f[o_bs], bs_w.data(), bs_w.size() * 4);
    const size_t o_w0 = A.alloc(5 * 18 * 16), o_b0 = A.alloc(16), o_w1 = A.alloc(5 * 2 * 8 * 8), o_b1 = A.alloc(16);
    const size_t o_w3 = A.alloc(5 * 2 * 8 * 8), o_b3 = A.alloc(16), o_w4 = A.alloc(5 * 16 * 2), o_b4 = A.alloc(2);
    for (int co = 0; co < 16; ++co)
        for (int ci = 0; ci < 18; ++ci)
            for (int k = 0; k < 5; ++k) A.f[o_w0 + (k * 18 + ci) * 16 + co] = w0[(co * 18 + ci) * 5 + k];
    for (int g = 0; g < 2; ++g)
        for (int co = 0; co < 8; ++co)
            for (int ci = 0; ci < 8; ++ci)
                for (int k = 0; k < 5; ++k) {
                    A.f[o_w1 + ((k * 2 + g) * 8 + ci) * 8 + co] = w1[((g * 8 + co) * 8 + ci) * 5 + k];   // Conv2d (Cout, Cin/g, 1, 5)
                    A.f[o_w3 + ((k * 2 + g) * 8 + ci) * 8 + co] = w3[((g * 8 + ci) * 8 + co) * 5 + k];   // ConvT  (Cin, Cout/g, 1, 5)
                }
    for (int ci = 0; ci < 16; ++ci)
        for (int co = 0; co < 2; ++co)
            for (int k = 0; k < 5; ++k) A.f[o_w4 + (k * 16 + ci) * 2 + co] = w4[(ci * 2 + co) * 5 + k];
    memcpy(&A.f[o_b0], b0, 64);
    memcpy(&A.f[o_b1], b1, 64);
    memcpy(&A.f[o_b3], b3, 64);
    memcpy(&A.f[o_b4], b4, 8);
    GtOff gte[3], gtd[3];
    DpOff dpo_[2];
    static const int en_dil[3] = {1, 2, 5}, de_dil[3] = {5, 2, 1};          // (:390-392, :405-407)
    for (int i = 0; i < 3; ++i) {
        // the decoder's GTConvBlocks are ordinary Conv2d here (use_deconv is never passed, :405-407)
        if (!load_gt(L, A, "encoder.en_convs." + std::to_string(i + 2) + ".", false, gte[i])) return L.st;
        if (!load_gt(L, A, "decoder.de_convs." + std::to_string(i) + ".", false, gtd[i])) return L.st;
    }
    if (!load_dp(L, A, "dpgrnn1.", dpo_[0]) || !load_dp(L, A, "dpgrnn2.", dpo_[1])) return L.st;

    HgtcrnEngine* e = new HgtcrnEngine();
    auto bail = [&](int st) { delete e; return st; };
    e->device = device; e->W = window_len; e->n_win = n_win; e->T = T;
    // DYNAMIC_AXES export (Export_H_GTCRN.py:27, :1097): the ISTFT keeps everything after the leading half window -- half a window more than the static trim --
    // normalised by the window-square sum of the actual frames (STFT_Process.py:318-327); everything else is the static arithmetic at the call's frame count.
    e->out_len_ = kHHop * (T - 1) + (dynamic ? kHNfft / 2 : 0);
    if (hipSetDevice(device) != hipSuccess) return bail(hfail(err, ADE_ERR_DEVICE, "hipSetDevice failed"));
    if (hipMalloc((void**)&e->d_w, A.f.size() * sizeof(float)) != hipSuccess || hipMemcpy(e->d_w, A.f.data(), A.f.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
        return bail(hfail(err, ADE_ERR_DEVICE, "upload of the H-GTCRN weights failed"));
    std::vector<int> ints(bm_start);
    ints.insert(ints.end(), bs_start.begin(), bs_start.end());
    if (hipMalloc((void**)&e->d_ints, ints.size() * sizeof(int)) != hipSuccess || hipMemcpy(e->d_ints, ints.data(), ints.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess)
        return bail(hfail(err, ADE_ERR_DEVICE, "upload of the ERB tables failed"));
    const float* Wd = e->d_w;
    e->erb_bm = BandTab{e->d_ints, Wd + o_bm, bm_count, kErbBands};
    e->erb_bs = BandTab{e->d_ints + kErbBands, Wd + o_bs, bs_count, kErbHigh};
    e->en0 = ConvW{Wd + o_w0, Wd + o_b0, a0[0]};
    e->en1 = ConvW{Wd + o_w1, Wd + o_b1, a1[0]};
    e->de3 = ConvW{Wd + o_w3, Wd + o_b3, a3[0]};
    e->de4 = ConvW{Wd + o_w4, Wd + o_b4, 0.0f};
    for (int i = 0; i < 3; ++i) {
        const GtOff* src[2] = {&gte[i], &gtd[i]};
        GtConvW* dst[2] = {&e->en_gt[i], &e->de_gt[i]};
        for (int k = 0; k < 2; ++k) {
            const GtOff& o = *src[k];
            *dst[k] = GtConvW{Wd + o.pw1, Wd + o.pw1_b, Wd + o.dw, Wd + o.dw_b, Wd + o.pw2, Wd + o.pw2_b, Wd + o.gru, Wd + o.fc, o.s1, o.s2, k == 0 ? en_dil[i] : de_dil[i], Wd + o.tra_rot};
        }
    }
    for (int i = 0; i < 2; ++i) {
        const DpOff& o = dpo_[i];
        e->dp[i] = DpW{Wd + o.intra_gru, Wd + o.inter_gru, Wd + o.fc[0], Wd + o.fc_b[0], Wd + o.ln_w[0], Wd + o.ln_b[0], Wd + o.fc[1], Wd + o.fc_b[1], Wd + o.ln_w[1], Wd + o.ln_b[1], Wd + o.inter_rot};
    }
    const size_t wpe_lds = (size_t)(5 * T + 2 * kTaps * kTaps + 6 * 72) * sizeof(float);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_hg_wpe), hipFuncAttributeMaxDynamicSharedMemorySize, (int)wpe_lds);
    ade_stft_config cfg{kHNfft, kHNfft, kHHop, "hann", nullptr, 1, "reflect"};          // H-GTCRN/Export_H_GTCRN.py:36-40, 1076-1097
    if (ade_stft_create(&cfg, device, &e->plan) != ADE_OK) return bail(hfail(err, ADE_ERR_DEVICE, std::string("h_gtcrn: STFT plan: ") + ade_stft_last_error(nullptr)));
    if (dynamic) (void)ade_stft_keep_tail(e->plan, 1);
    *out = e;
    return ADE_OK;
}

int HgtcrnEngine::reserve(int calls, std::string& err) {
    if (calls <= capacity) return ADE_OK;
    const int batch = calls * n_win;
    HG_HIP(hipSetDevice(device));
    HG_HIP(hipDeviceSynchronize());
    if (ws) (void)hipFree(ws);
    ws = nullptr;
    capacity = 0;
    const size_t B = batch, nfr = B * T, sp = B * 2 * 2 * kHBins * T, act = nfr * kFw * kCh;
    struct Carve { float** p; size_t n; };
    std::vector<Carve> cs = {{&mean, (size_t)calls}, {&xf, B * 2 * W}, {&spec, sp}, {&drb, sp}, {&iva, sp}, {&eps, B}, {&rinv, B * 2 * T}, {&wst, B * kHBins * 8}, {&epart, B * 2 * kHBins},
                             {&feat, nfr * 6 * kErbPad}, {&e0, nfr * kF1 * kCh}, {&e1, act}, {&h, act}, {&zt, nfr * 8}, {&rnn, act}, {&dpm, act}, {&dpo[0], act}, {&dpo[1], act},
                             {&d3, nfr * kF1 * kCh}, {&mask, nfr * 2 * kErbPad}, {&sout, B * 2 * kHBins * T}, {&yf, B * out_len_}, {(float**)&pred, B}};
    for (int i = 0; i < 3; ++i) { cs.push_back({&xe[i], act}); cs.push_back({&ate[i], nfr * 8}); cs.push_back({&xd[i], act}); cs.push_back({&atd[i], nfr * 8}); }
    size_t total = 0;
    for (auto& c : cs) total += (c.n + 63) & ~(size_t)63;
    HG_HIP(hipMalloc((void**)&ws, total * sizeof(float)));
    HG_HIP(hipMemset(ws, 0, total * sizeof(float)));                       // the pad columns of feat / mask stay zero
    size_t at_ = 0;
    for (auto& c : cs) { *c.p = ws + at_; at_ += (c.n + 63) & ~(size_t)63; }
    fgeo = -1; fseg = 0;
    if (d_xchg) { (void)hipFree(d_xchg); d_xchg = nullptr; }
    if (d_xflags) { (void)hipFree(d_xflags); d_xflags = nullptr; }
    if (fused_net && fused_init() != hipSuccess) { (void)hipGetLastError(); fused_net = false; }      // (the stage kernels' dynamic-LDS limit; refused: multi-kernel sequence)
    if (fused_net) {
        for (int g = 2; g >= 0 && fgeo < 0; --g)
            if (fused_supported(T, g)) fgeo = g;
        if (fgeo >= 0) {
            fseg = fused_segments(T, fgeo);
            if (!d_xerr) { HG_HIP(hipHostMalloc((void**)&d_xerr, 4 * sizeof(int), hipHostMallocDefault)); d_xerr[0] = 0; }
            HG_HIP(hipMalloc((void**)&d_xchg, B * fseg * (size_t)kXFloats * sizeof(float)));
            HG_HIP(hipMalloc((void**)&d_xflags, B * fseg * (size_t)kXFlags * sizeof(unsigned)));
            HG_HIP(hipMemset(d_xflags, 0, B * fseg * (size_t)kXFlags * sizeof(unsigned)));
        }
    }
    // let the STFT plan size its buffers now (it allocates lazily), so that run() never allocates
    if (ade_stft_analyze(plan, xf, batch * 2, W, spec, nullptr) != ADE_OK) return hfail(err, ADE_ERR_DEVICE, std::string("h_gtcrn: ") + ade_stft_last_error(plan));
    if (ade_stft_synthesize(plan, sout, batch, T, yf, nullptr) != ADE_OK) return hfail(err, ADE_ERR_DEVICE, std::string("h_gtcrn: ") + ade_stft_last_error(plan));
    HG_HIP(hipDeviceSynchronize());
    capacity = calls;
    return ADE_OK;
}

int HgtcrnEngine::run(hipStream_t s, const int16_t* d_in, int batch, int16_t* d_out, float* d_f32, std::string& err) {
    if (batch == 0) return ADE_OK;
    int st = reserve(batch, err);
    if (st != ADE_OK) return st;
    const int B = batch * n_win;
    const int nfr = B * T;
    auto flat = [&](long long total) { return dim3((unsigned)((total + 255) / 256)); };
    // (a fused stage that gave up a bounded hand-off wait is reported by the ENGINE: exchange_error_and_reset() after its stream synchronise and at the entry of
    //  the next call, outside the captured graph that replays this function's launches)
    if (float_in) {
        if (float_src_len > 0 && float_src_len < W) {      // upsampled: centred before the interpolation, i.e. with the mean of the CALLER-rate samples
            if (float_src) hipLaunchKernelGGL(k_hg_mean_f32, dim3((unsigned)batch), dim3(256), 0, s, float_src, (long long)2 * float_src_len, mean, float_src_gain);   // a float tensor came in: d_in is not PCM
            else launch_pcm_mean(s, d_in, batch, 2 * float_src_len, mean, 1);
        }
        else hipLaunchKernelGGL(k_hg_mean_f32, dim3((unsigned)batch), dim3(256), 0, s, float_in, (long long)2 * n_win * W, mean, 1.0f);      // one mean per call (:963-964)
        hipLaunchKernelGGL(k_hg_f2f, flat((long long)B * 2 * W), dim3(256), 0, s, float_in, (const float*)mean, xf, W, n_win, (long long)B * 2 * W);
    } else {
        launch_pcm_mean(s, d_in, batch, 2 * n_win * W, mean, 1);
        hipLaunchKernelGGL(k_hg_pcm2f, flat((long long)B * 2 * W), dim3(256), 0, s, d_in, (const float*)mean, xf, W, n_win, (long long)B * 2 * W);
    }
    if (ade_stft_analyze(plan, xf, B * 2, W, spec, (void*)s) != ADE_OK) return hfail(err, ADE_ERR_DEVICE, std::string("h_gtcrn: ") + ade_stft_last_error(plan));
    // WPE
    hipLaunchKernelGGL(k_hg_wpe_eps, dim3((unsigned)B), dim3(256), 0, s, (const float*)spec, eps, T);
    const size_t wpe_lds = (size_t)(5 * T + 2 * kTaps * kTaps + 6 * 72) * sizeof(float);
    hipLaunchKernelGGL(k_hg_wpe, dim3(kHBins, (unsigned)B), dim3(256), wpe_lds, s, (const float*)spec, (const float*)eps, drb, T);
    // AuxIVA: Y starts as the dereverberated spectrum (W = I)
    const dim3 bins((kHBins + 3) / 4, (unsigned)B);
    for (int it = 0; it < kIvaIter; ++it) {
        hipLaunchKernelGGL(k_hg_iva_r, dim3((unsigned)((T + 31) / 32), (unsigned)(B * 2)), dim3(256), 0, s, (const float*)(it == 0 ? drb : iva), rinv, T);
        hipLaunchKernelGGL(k_hg_iva_step, bins, dim3(256), 0, s, (const float*)drb, (const float*)rinv, wst, iva, T, it == 0 ? 1 : 0);
    }
    hipLaunchKernelGGL(k_hg_iva_project, bins, dim3(256), 0, s, (const float*)drb, iva, epart, T);
    hipLaunchKernelGGL(k_hg_pred, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, s, (const float*)epart, pred, B);
    hipLaunchKernelGGL(k_hg_feat, flat((long long)nfr * 6 * kErb), dim3(256), 0, s, (const float*)spec, (const float*)iva, (const int*)pred, erb_bm, feat, T, (long long)nfr * 6 * kErb);
    // network: GTCRN's kernels from the second convolution on
    hipLaunchKernelGGL(k_hg_conv0, flat((long long)nfr * kF1), dim3(256), 0, s, (const float*)feat, en0.w, en0.b, en0.slope, e0, nfr);
    launch_conv1(s, e0, en1, e1, nfr);
    if (fgeo >= 0) {
        // quad-planar tensors (X[b][q][p] = channels 4 q .. 4 q + 3 of position p) in the buffers the multi-kernel sequence uses channels-last: h = e1, xe / dpo / xd as named
        SegPlan plan{};
        plan.wait_ticks = xwait_ticks; plan.nseg = fseg; plan.xchg = d_xchg; plan.flags = d_xflags; plan.err = d_xerr;
        const int P = T * kFw;
        const long long quads = (long long)B * 4 * P;
        hipLaunchKernelGGL(k_hg_relayout, flat(quads), dim3(256), 0, s, (const float*)e1, h, P, 1, quads);
        const float* xp = h;
        for (int i = 0; i < 3; ++i) { launch_gtblock(s, fgeo, plan, i, xp, nullptr, en_gt[i], xe[i], B, T, nullptr); xp = xe[i]; }
        for (int i = 0; i < 2; ++i) { launch_dpgrnn(s, fgeo, plan, i, xp, dp[i], dpo[i], B, T, nullptr); xp = dpo[i]; }
        for (int i = 0; i < 3; ++i) { launch_gtblock(s, fgeo, plan, 3 + i, xp, xe[2 - i], de_gt[i], xd[i], B, T, nullptr); xp = xd[i]; }
        hipLaunchKernelGGL(k_hg_relayout, flat(quads), dim3(256), 0, s, xp, rnn, P, 0, quads);                       // back to channels-last for the (1, 3) transposed convolutions
        launch_deconv3(s, View{rnn, nullptr}, View{e1, nullptr}, de3, d3, nfr);
    } else {
    const View none{nullptr, nullptr};
    View x{e1, nullptr};
    for (int i = 0; i < 3; ++i) {
        launch_gt_pw1(s, x, none, en_gt[i], h, nfr);
        launch_gt_dw_pw2(s, h, x, none, en_gt[i], xe[i], zt, B, T);
        launch_tra(s, zt, en_gt[i], ate[i], B, T);
        x = View{xe[i], ate[i]};
    }
    for (int i = 0; i < 2; ++i) {
        launch_intra_gru(s, x, dp[i].intra_gru, rnn, nfr);
        launch_fc_ln_res(s, rnn, x, dp[i].intra_fc, dp[i].intra_fc_b, dp[i].intra_ln_w, dp[i].intra_ln_b, dpm, B, T);
        launch_inter_gru(s, dpm, dp[i].inter_gru, rnn, B, T, nullptr, dp[i].inter_rot);
        launch_fc_ln_res(s, rnn, View{dpm, nullptr}, dp[i].inter_fc, dp[i].inter_fc_b, dp[i].inter_ln_w, dp[i].inter_ln_b, dpo[i], B, T);
        x = View{dpo[i], nullptr};
    }
    for (int i = 0; i < 3; ++i) {
        const View skip{xe[2 - i], ate[2 - i]};
        launch_gt_pw1(s, x, skip, de_gt[i], h, nfr);
        launch_gt_dw_pw2(s, h, x, skip, de_gt[i], xd[i], zt, B, T);
        launch_tra(s, zt, de_gt[i], atd[i], B, T);
        x = View{xd[i], atd[i]};
    }
    launch_deconv3(s, x, View{e1, nullptr}, de3, d3, nfr);
    }
    launch_deconv4(s, d3, e0, de4, mask, nfr);
    hipLaunchKernelGGL(k_hg_mask, flat((long long)B * kHBins * T), dim3(256), 0, s, (const float*)mask, (const float*)spec, erb_bs, sout, T, (long long)B * kHBins * T);
    if (ade_stft_synthesize(plan, sout, B, T, yf, (void*)s) != ADE_OK) return hfail(err, ADE_ERR_DEVICE, std::string("h_gtcrn: ") + ade_stft_last_error(plan));
    hipLaunchKernelGGL(k_hg_f2pcm, flat((long long)B * out_len_), dim3(256), 0, s, (const float*)yf, d_out, d_f32, (long long)B * out_len_);
    HG_HIP(hipGetLastError());
    return ADE_OK;
}

// taps: "stft", "wpe", "iva": (B, 2 microphones / sources, [re 257 | im 257], T); "mask": (B, T, 2, 132)
int HgtcrnEngine::tap(hipStream_t s, const char* name, int batch, float* out, size_t count, size_t* written, std::string& err) {
    const size_t B = (size_t)batch * n_win;
    const float* src = nullptr;
    size_t n = B * 2 * 2 * kHBins * T;
    if (strcmp(name, "stft") == 0) src = spec;
    else if (strcmp(name, "wpe") == 0) src = drb;
    else if (strcmp(name, "iva") == 0) src = iva;
    else if (strcmp(name, "mask") == 0) { src = mask; n = B * T * 2 * kErbPad; }
    else return hfail(err, ADE_ERR_NOT_FOUND, std::string("unknown tap: ") + name);
    if (!src || batch <= 0) return hfail(err, ADE_ERR_NOT_FOUND, "tap has no data yet");
    if (count < n) return hfail(err, ADE_ERR_SHAPE_MISMATCH, "tap buffer too small");
    HG_HIP(hipStreamSynchronize(s));
    HG_HIP(hipMemcpy(out, src, n * sizeof(float), hipMemcpyDeviceToHost));
    *written = n;
    return ADE_OK;
}

}  // namespace ade
