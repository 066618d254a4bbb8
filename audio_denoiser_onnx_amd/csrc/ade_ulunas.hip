// ade_ulunas.hip — UL-UNAS (ultra-lightweight 16 kHz speech enhancement) on the MI355X: SURVEY.md section 8 row f2.
//
// Reference: ULUNAS_CUSTOM.forward (UL-UNAS/Export_UL_UNAS.py:848-913) around ULUNAS.forward and its blocks (:111-739), over the
// tensors audio_denoiser_onnx_amd/ulunas.py::fold_state_dict produces (BatchNorm folded into the convolutions, AffinePReLU as
// positive / negative slope tables per (channel, bin), the two half-width GRUs of every grouped GRU):
//   int16 -> * 2^-15 -> STFT(512, hop 256, periodic hann, reflect) -> log(max(|X|^2, 1e-24)) -> ERB merge (65 + 64 bands) ->
//   5 encoder blocks (XConvBlock / XMBBlocks / XDWSBlock: grouped, depthwise and pointwise convolutions causal in time, stride on
//   frequency, AffinePReLU, channel shuffle, each ending in a causal time-frequency attention cTFA = time GRU gate x frequency
//   bi-GRU gate) -> 2 x DPGRNN (the same grouped dual-path GRU as GTCRN's) -> 5 decoder blocks on x + skip (transposed
//   convolutions) -> sigmoid -> ERB split -> real mask x spectrum -> ISTFT -> * 32767, clamp, truncate -> int16.
// A first, correctness-first implementation: activations are channels-last (batch, frame, bin, channel) tensors in HBM and every
// operator is its own bandwidth-bound kernel (one generic convolution kernel covers all 23 convolutions through a descriptor).
// The DPGRNN blocks ARE GTCRN's (same widths: 33 bins x 16 channels) and reuse its multi-kernel path (ade_kernels.hip); the STFT
// pair is the generic operator of ade_stft.hip (dense windowed DFT on the matrix cores, exact-angle tables).
#include "ade_device.h"
#include "ade_internal.h"
#include "../../include/ade.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <type_traits>

namespace ade {

namespace {

using dev::mk2;
using dev::v2f;

constexpr int kUNfft = 512, kUHop = 256, kUBins = 257, kULow = 65, kUBands = 64, kUHigh = 192, kUErb = kULow + kUBands;   // 129

__device__ __forceinline__ float usig(float x) { return ade::dev::sigmoid_f(x); }      // (round 5: hardware exp2 / rcp; libm expf + the IEEE division were ~27 instructions per gate of the cTFA GRUs)

struct ConvDesc {              // one (de)convolution + bias [+ AffinePReLU] [+ channel shuffle] over (B, T, F, C) tensors
    const float *w, *b, *pos, *neg, *abias;     // w in torch layout: Conv2d (Cout, Cin/g, kt, kf); ConvTranspose2d (Cin, Cout/g, kt, kf)
    int Cin, Cout, Fi, Fo, kt, kf, stride, groups, deconv, shuffle;
};

__device__ __forceinline__ int shuffled(int j, int C) { return (j & 1) ? (j >> 1) + (C >> 1) : (j >> 1); }   // Shuffle.indices (:200-203)

// One output of one (de)convolution of a frame: position o = fo * Cout + co of the output frame, from the kt staged input frames xin[a][Fi * Cin] (tap a = frame
// t - (KT - 1) + a, or t - a for a transposed convolution) and the weights wl in their torch layout, both in LDS.  Bias, AffinePReLU and the channel shuffle included.
template <int KT, int KF, int STRIDE, bool DECONV>      // the five kernel shapes of ULUNAS() (:667) as compile-time constants: the tap loops unroll
__device__ __forceinline__ float conv_out(const float* __restrict__ xin, int row, const float* __restrict__ wl, const ConvDesc& d, int o, int cog, int cig) {
    constexpr int pf = KF / 2;
    const int fo = o / d.Cout, co = o - fo * d.Cout;
    const int cc = d.shuffle ? shuffled(co, d.Cout) : co;       // output position co holds convolution channel cc
    const int g = cc / cog;
    float acc = d.b[cc];
    if (KT == 1 && KF == 1 && !DECONV && STRIDE == 1 && !(cig & 3) && !(d.Cin & 3)) {
        // pointwise convolution (twelve per call, 28 % of the step as launches of their own): the channel run of the position and the weight row are both contiguous -- four
        // channels per ds_read_b128 pair instead of two scalar LDS reads per multiply-add; four partial sums, added in a fixed order
        const float4* xr4 = reinterpret_cast<const float4*>(xin + fo * d.Cin + g * cig);
        const float4* wr4 = reinterpret_cast<const float4*>(wl + cc * cig);
        float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
        for (int q = 0; q < (cig >> 2); ++q) {
            const float4 xv = xr4[q], wv = wr4[q];
            a0 = fmaf(xv.x, wv.x, a0); a1 = fmaf(xv.y, wv.y, a1); a2 = fmaf(xv.z, wv.z, a2); a3 = fmaf(xv.w, wv.w, a3);
        }
        acc += (a0 + a1) + (a2 + a3);
    } else
#pragma unroll
    for (int a = 0; a < KT; ++a)
#pragma unroll
        for (int bb = 0; bb < KF; ++bb) {
            int fi;
            if (DECONV) {
                const int num = fo + pf - bb;
                if (num < 0 || num % STRIDE) continue;
                fi = num / STRIDE;
            } else {
                fi = fo * STRIDE - pf + bb;
            }
            if (fi < 0 || fi >= d.Fi) continue;
            const float* xr = xin + a * row + fi * d.Cin + g * cig;
            for (int ci = 0; ci < cig; ++ci) {
                const float wv = DECONV ? wl[(((g * cig + ci) * cog + (cc - g * cog)) * KT + a) * KF + bb] : wl[((cc * cig + ci) * KT + a) * KF + bb];
                acc += xr[ci] * wv;
            }
        }
    if (d.pos) acc = (acc > 0.0f ? d.pos[cc * d.Fo + fo] : d.neg[cc * d.Fo + fo]) * acc + d.abias[cc * d.Fo + fo];   // AffinePReLU (:128-130)
    return acc;
}

// A whole output frame of one (de)convolution by the workgroup: emit(o, value) for every position o = fo * Cout + co.  (Round 5.)  conv_out() per output spends its
// time on index arithmetic -- two integer divisions, the shuffle, the group, three table addresses -- not on multiply-adds: a launch ran at ~13 % of the vector rate.
// Here a thread KEEPS its output channel (co = tid % Cout, one division per thread and frame) and walks the positions fo0, fo0 + 256 / Cout, ...: channel, group, bias,
// AffinePReLU rows and -- for the pointwise and the one-input-channel shapes, all but the last deconvolution -- its weight row in registers are per-thread constants.
template <int CIG>
__device__ __forceinline__ float pw_dot(const float* __restrict__ xr, const float (&w)[CIG]) {
    if constexpr (CIG % 4 == 0) {      // four partial sums, added in a fixed order (the order of the float4 path of conv_out)
        float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
#pragma unroll
        for (int q = 0; q < CIG / 4; ++q) {
            const float4 xv = reinterpret_cast<const float4*>(xr)[q];
            a0 = fmaf(xv.x, w[4 * q], a0); a1 = fmaf(xv.y, w[4 * q + 1], a1); a2 = fmaf(xv.z, w[4 * q + 2], a2); a3 = fmaf(xv.w, w[4 * q + 3], a3);
        }
        return (a0 + a1) + (a2 + a3);
    } else {
        float a = 0.0f;
#pragma unroll
        for (int q = 0; q < CIG / 2; ++q) {
            const float2 xv = reinterpret_cast<const float2*>(xr)[q];
            a += xv.x * w[2 * q]; a += xv.y * w[2 * q + 1];
        }
        return a;
    }
}
template <int KT, int KF, int STRIDE, bool DECONV, class Emit>
__device__ __forceinline__ void conv_frame(const float* __restrict__ xin, int row, const float* __restrict__ wl, const ConvDesc& d, Emit&& emit) {
    const int Cout = d.Cout, cog = Cout / d.groups, cig = d.Cin / d.groups, Fo = d.Fo;
    constexpr bool kPw = KT == 1 && KF == 1 && !DECONV && STRIDE == 1;
    const bool fast = kPw ? (cig == 6 && !(d.Cin & 1)) || ((cig == 8 || cig == 12 || cig == 16) && !(d.Cin & 3)) : cig == 1;      // (vector LDS reads of a position's channel run)
    if (!fast || Cout > 256) {         // (the last deconvolution: 12 -> 1 channels, 3 x 3 taps)
        for (int o = threadIdx.x; o < Fo * Cout; o += 256) emit(o, conv_out<KT, KF, STRIDE, DECONV>(xin, row, wl, d, o, cog, cig));
        return;
    }
    const int nper = 256 / Cout, fo0 = (int)threadIdx.x / Cout, co = (int)threadIdx.x - fo0 * Cout;
    if (fo0 >= nper) return;
    const int cc = d.shuffle ? shuffled(co, Cout) : co, g = cc / cog;
    const float bias = d.b[cc];
    const float *pos = d.pos ? d.pos + cc * Fo : nullptr, *neg = d.pos ? d.neg + cc * Fo : nullptr, *ab = d.pos ? d.abias + cc * Fo : nullptr;
    auto act = [&](float acc, int fo) { return pos ? (acc > 0.0f ? pos[fo] : neg[fo]) * acc + ab[fo] : acc; };      // AffinePReLU (:128-130)
    if constexpr (kPw) {
        const float* xg = xin + g * cig;
        auto run = [&](auto cig_c) {
            constexpr int CIG = decltype(cig_c)::value;
            float w[CIG];
#pragma unroll
            for (int q = 0; q < CIG; ++q) w[q] = wl[cc * CIG + q];
            for (int fo = fo0; fo < Fo; fo += nper) emit(fo * Cout + co, act(bias + pw_dot<CIG>(xg + fo * d.Cin, w), fo));
        };
        if (cig == 6) run(std::integral_constant<int, 6>{});
        else if (cig == 8) run(std::integral_constant<int, 8>{});
        else if (cig == 12) run(std::integral_constant<int, 12>{});
        else run(std::integral_constant<int, 16>{});
    } else {                           // one input channel per output channel (depthwise, and the first block's 1 -> 12): KT x KF weights
        constexpr int pf = KF / 2;
        float w[KT][KF];
#pragma unroll
        for (int a = 0; a < KT; ++a)
#pragma unroll
            for (int bb = 0; bb < KF; ++bb) w[a][bb] = wl[(cc * KT + a) * KF + bb];      // (both layouts: Conv2d (Cout, 1, kt, kf); ConvTranspose2d (Cin, cog, kt, kf) at cig == 1)
        const float* xg = xin + g;
        for (int fo = fo0; fo < Fo; fo += nper) {
            float acc = bias;
#pragma unroll
            for (int a = 0; a < KT; ++a)
#pragma unroll
                for (int bb = 0; bb < KF; ++bb) {
                    int fi;
                    if (DECONV) {
                        const int num = fo + pf - bb;
                        if (num < 0 || num % STRIDE) continue;
                        fi = num / STRIDE;
                    } else {
                        fi = fo * STRIDE - pf + bb;
                    }
                    if (fi < 0 || fi >= d.Fi) continue;
                    acc += xg[a * row + fi * d.Cin] * w[a][bb];
                }
            emit(fo * Cout + co, act(acc, fo));
        }
    }
}

// The cTFA statistics of a block's last convolution (:181-182, :150): zt[c] = mean_f y^2, pfreq[f] = mean_c y^2 of the frame this workgroup has just written to orow.
// sq: Fo * Cout floats of LDS nobody reads any more (the caller has synchronised).
__device__ __forceinline__ void frame_stats(const float* __restrict__ orow, float* __restrict__ sq, int Fo, int Cout, long long frame, float* __restrict__ zt,
                                            float* __restrict__ pfreq) {
    for (int o = threadIdx.x; o < Fo * Cout; o += 256) { const float v = orow[o]; sq[o] = v * v; }   // own writes: visible to this thread
    __syncthreads();
    for (int c = threadIdx.x; c < Cout; c += 256) {
        float a = 0.0f;
        for (int f = 0; f < Fo; ++f) a += sq[f * Cout + c];
        zt[frame * Cout + c] = a / (float)Fo;
    }
    for (int f = threadIdx.x; f < Fo; f += 256) {
        float a = 0.0f;
        for (int c = 0; c < Cout; ++c) a += sq[f * Cout + c];
        pfreq[frame * Fo + f] = a / (float)Cout;
    }
}

// out[b][t][fo][co] = act(bias + sum over taps / group channels), causal in t (:222-238, :264-267); optional second input added first (:648).
// One workgroup per frame: the kt input frames it needs (already summed with the skip tensor) and the whole weight tensor are staged in LDS
// once, so HBM / L2 sees every input element kt times instead of once per tap, channel and output.
template <int KT, int KF, int STRIDE, bool DECONV>
__global__ __launch_bounds__(256) void k_ulu_conv(const float* __restrict__ x, const float* __restrict__ x2, ConvDesc d, float* __restrict__ out, int T, int wsize,
                                                  float* __restrict__ zt, float* __restrict__ pfreq) {
    HIP_DYNAMIC_SHARED(float, lds)
    const long long frame = blockIdx.x;
    const long long b = frame / T;
    const int t = (int)(frame - b * T), row = d.Fi * d.Cin;
    float* xin = lds;                      // [kt][Fi * Cin]
    float* wl = lds + KT * row;          // the weights in their torch layout
    for (int a = 0; a < KT; ++a) {
        const int tt = DECONV ? t - a : t - (KT - 1) + a;
        const size_t at = ((size_t)b * T + (tt < 0 ? 0 : tt)) * row;
        for (int i = threadIdx.x; i < row; i += 256) xin[a * row + i] = tt < 0 ? 0.0f : (x2 ? x[at + i] + x2[at + i] : x[at + i]);
    }
    for (int i = threadIdx.x; i < wsize; i += 256) wl[i] = d.w[i];
    __syncthreads();
    float* orow = out + (size_t)frame * d.Fo * d.Cout;
    conv_frame<KT, KF, STRIDE, DECONV>(xin, row, wl, d, [&](int o, float v) { orow[o] = v; });
    if (!zt) return;
    __syncthreads();                       // every thread is done with the staged inputs: the region is reused for the squares (Fo * Cout <= the staged size is NOT guaranteed -> sized by the launcher)
    frame_stats(orow, lds, d.Fo, d.Cout, frame, zt, pfreq);
}

// time attention (:183-186): GRU(C -> 2C) over frames, Linear(2C -> C), sigmoid.  The recurrence is a chain of T dependent steps, so
// everything it touches lives in LDS (weights once per workgroup, the clip's inputs 64 frames at a time) and one step is spread over four
// wavefronts: wavefront p accumulates a quarter of the C + 2C contraction for every hidden unit (lane = unit), wavefront 0 adds the
// four partial sums, applies the gates and the output Linear.
// wih_t [C][3][2C], whh_t [2C][3][2C] (transposed so that lanes read consecutive floats), fc_t [2C][C].
template <int C>               // compile-time channel count: the k loops unroll and their LDS reads pipeline
__global__ __launch_bounds__(256) void k_ulu_ta(const float* __restrict__ zt, const float* __restrict__ wih_t, const float* __restrict__ whh_t,
                                                const float* __restrict__ bih, const float* __restrict__ bhh, const float* __restrict__ fc_t,
                                                const float* __restrict__ fc_b, float* __restrict__ at, int T) {
    HIP_DYNAMIC_SHARED(float, lds)
    __shared__ float hs[64];
    __shared__ float part[4][6][64];
    constexpr int H = 2 * C, KI = (C + 3) / 4, KH = (H + 3) / 4;       // per-wavefront slices of the two contractions
    const int b = blockIdx.x, tid = threadIdx.x, j = tid & 63, p = tid >> 6;
    float* s_wih = lds;                       // C * 3 * H
    float* s_whh = s_wih + C * 3 * H;         // H * 3 * H
    float* s_fc = s_whh + H * 3 * H;          // H * C
    float* s_z = s_fc + H * C;                // 64 frames of the clip's zt rows
    for (int i = tid; i < C * 3 * H; i += 256) s_wih[i] = wih_t[i];
    for (int i = tid; i < H * 3 * H; i += 256) s_whh[i] = whh_t[i];
    for (int i = tid; i < H * C; i += 256) s_fc[i] = fc_t[i];
    const bool unit = j < H;
    float bi[3] = {0.0f, 0.0f, 0.0f}, bh[3] = {0.0f, 0.0f, 0.0f};
    if (unit && p == 0)
        for (int g = 0; g < 3; ++g) { bi[g] = bih[g * H + j]; bh[g] = bhh[g * H + j]; }
    const float fb = (p == 0 && j < C) ? fc_b[j] : 0.0f;
    float h = 0.0f;
    if (tid < 64) hs[tid] = 0.0f;
    __syncthreads();
    for (int t = 0; t < T; ++t) {
        if ((t & 63) == 0) {
            __syncthreads();
            const int n = min(64, T - t) * C;
            for (int i = tid; i < n; i += 256) s_z[i] = zt[((size_t)b * T + t) * C + i];
            __syncthreads();
        }
        const float* z = s_z + (t & 63) * C;
        float gi[3] = {0.0f, 0.0f, 0.0f}, gh[3] = {0.0f, 0.0f, 0.0f};
        if (unit) {
#pragma unroll
            for (int kk = 0; kk < KI; ++kk) {
                const int k = p * KI + kk;
                if (k < C) {
                    const float zk = z[k];
#pragma unroll
                    for (int g = 0; g < 3; ++g) gi[g] += s_wih[(k * 3 + g) * H + j] * zk;
                }
            }
#pragma unroll
            for (int kk = 0; kk < KH; ++kk) {
                const int k = p * KH + kk;
                if (k < H) {
                    const float hk = hs[k];
#pragma unroll
                    for (int g = 0; g < 3; ++g) gh[g] += s_whh[(k * 3 + g) * H + j] * hk;
                }
            }
#pragma unroll
            for (int g = 0; g < 3; ++g) { part[p][g][j] = gi[g]; part[p][3 + g][j] = gh[g]; }
        }
        __syncthreads();
        if (p == 0 && unit) {
            float si[3], sh[3];
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                si[g] = bi[g] + ((part[0][g][j] + part[1][g][j]) + (part[2][g][j] + part[3][g][j]));
                sh[g] = bh[g] + ((part[0][3 + g][j] + part[1][3 + g][j]) + (part[2][3 + g][j] + part[3][3 + g][j]));
            }
            const float r = usig(si[0] + sh[0]), zg = usig(si[1] + sh[1]), n = ade::dev::tanh_f(si[2] + r * sh[2]);
            h = (1.0f - zg) * n + zg * h;
            hs[j] = h;
        }
        __syncthreads();
        if (p == 0 && j < C) {
            float a = fb;
#pragma unroll
            for (int k = 0; k < H; ++k) a += s_fc[k * C + j] * hs[k];
            at[((size_t)b * T + t) * C + j] = usig(a);
        }
    }
}

// The same time attention with the recurrence on ONE wavefront and nothing but the recurrence inside it (round 5).  k_ulu_ta spreads every step's contraction over four
// wavefronts and pays three workgroup barriers per step: 2.5 us per step, 105 - 160 us per launch, a quarter of the model's time.  Here a workgroup (one clip, 64 frames at
// a time) runs three phases: (1) all four wavefronts form the input projections b_ih + W_ih z_t of the whole chunk (no dependence between frames) into LDS; (2) wavefront 0
// alone walks the chunk: lane j = hidden unit j, h_{t-1} broadcast lane by lane into scalar registers (v_readlane), the unit's 3 x H recurrent weights held in REGISTERS (a
// 256-thread workgroup that fills a CU's LDS is one wavefront per SIMD: 512 registers each; reading them from LDS -- 144 to 192 ds_read_b32 per step -- made the step as slow
// as the four-wavefront form, 1.6 us), r | z gates as one packed accumulator, h_t into LDS -- no barrier, no other wavefront involved; (3) all four wavefronts apply the output Linear + sigmoid to the chunk's hidden
// states.  Sums run k ascending (k_ulu_ta adds four partial sums: the results differ by fp32 round-off, 1e-7).
template <int C>
__global__ __launch_bounds__(256) void k_ulu_ta2(const float* __restrict__ zt, const float* __restrict__ wih_t, const float* __restrict__ whh_t,
                                                 const float* __restrict__ bih, const float* __restrict__ bhh, const float* __restrict__ fc_t,
                                                 const float* __restrict__ fc_b, float* __restrict__ at, int T) {
    HIP_DYNAMIC_SHARED(float, lds)
    constexpr int H = 2 * C, G = 3 * H;
    const int b = blockIdx.x, tid = threadIdx.x, j = tid & 63;
    constexpr bool kNLds = C >= 32;           // 3 x 64 weights + the loop's working set pass the 256 architected registers: the n-gate row stays in LDS there (measured: 185 us with
                                              // all three rows in registers, 141 us with all three in LDS)
    float* s_gi = lds;                        // [64 frames][3][H]
    float* s_h = s_gi + 64 * G;               // [64 frames][H]
    float* s_z = s_h + 64 * H;                // [64 frames][C]
    float* s_wn = s_z + 64 * C;               // [H][H]: W_hn^T (kNLds only)
    const int u = j < H ? j : H - 1;          // (lanes beyond H compute unit H - 1 again and store nothing)
    v2f wrz[H];                               // this unit's recurrent rows: (W_hr, W_hz)[u][k] and W_hn[u][k]
    float wn[kNLds ? 1 : H];
    if (kNLds)
        for (int i = tid; i < H * H; i += 256) s_wn[i] = whh_t[((i / H) * 3 + 2) * H + (i % H)];
    float bh[3] = {0.0f, 0.0f, 0.0f};
    if (tid < 64) {
#pragma unroll
        for (int k = 0; k < H; ++k) { wrz[k] = mk2(whh_t[(k * 3 + 0) * H + u], whh_t[(k * 3 + 1) * H + u]); if (!kNLds) wn[k] = whh_t[(k * 3 + 2) * H + u]; }
#pragma unroll
        for (int g = 0; g < 3; ++g) bh[g] = bhh[g * H + u];
    } else {
#pragma unroll
        for (int k = 0; k < H; ++k) { wrz[k] = mk2(0.0f, 0.0f); if (!kNLds) wn[k] = 0.0f; }
    }
    float h = 0.0f;
    for (int t0 = 0; t0 < T; t0 += 64) {
        const int nt = min(64, T - t0);
        __syncthreads();                      // (the previous chunk's phase 3 is done with s_h; first pass: the weights are staged)
        for (int i = tid; i < nt * C; i += 256) s_z[i] = zt[((size_t)b * T + t0) * C + i];
        __syncthreads();
        // (1) input projections of the chunk: output (t, g, u) = b_ih[g][u] + sum_k W_ih^T[k][g][u] z_t[k]; consecutive threads = consecutive u (coalesced weight rows)
        for (int o = tid; o < nt * G; o += 256) {
            const int t = o / G, gu = o - t * G;
            float a = bih[gu];
#pragma unroll 4
            for (int k = 0; k < C; ++k) a = fmaf(wih_t[k * G + gu], s_z[t * C + k], a);
            s_gi[o] = a;
        }
        __syncthreads();
        // (2) the recurrence: wavefront 0, lane j = hidden unit
        if (tid < 64) {
            for (int t = 0; t < nt; ++t) {
                v2f grz = mk2(bh[0], bh[1]);
                float gn = bh[2];
#pragma unroll
                for (int k = 0; k < H; ++k) {
                    const float hk = __shfl(h, k, 64);               // constant lane: v_readlane into a scalar register
                    grz += wrz[k] * hk;
                    gn = fmaf(kNLds ? s_wn[k * H + u] : wn[k], hk, gn);
                }
                const float* gi = s_gi + t * G;
                const float r = usig(gi[u] + grz[0]), zg = usig(gi[H + u] + grz[1]), n = ade::dev::tanh_f(gi[2 * H + u] + r * gn);
                h = (1.0f - zg) * n + zg * h;
                if (j < H) s_h[t * H + j] = h;
            }
        }
        __syncthreads();
        // (3) at[t][c] = sigmoid(fc_b[c] + sum_k fc^T[k][c] h_t[k])
        for (int o = tid; o < nt * C; o += 256) {
            const int t = o / C, c = o - t * C;
            float a = fc_b[c];
#pragma unroll 4
            for (int k = 0; k < H; ++k) a = fmaf(fc_t[k * C + c], s_h[t * H + k], a);
            at[((size_t)b * T + t0) * C + o] = usig(a);
        }
    }
}

// frequency attention GRUs (:151-153): per frame, the zero-padded bin powers in groups of 4 are a sequence of H steps through a
// bidirectional GRU(4 -> 4).  One thread per (frame, direction); weights [12][4] | [12][4] | [12] | [12] per direction.
__global__ __launch_bounds__(256) void k_ulu_fa_gru(const float* __restrict__ pfreq, const float* __restrict__ wf, const float* __restrict__ wb,
                                                    float* __restrict__ fah, int F, int H, long long frames) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= frames * 2) return;
    const long long frame = i >> 1;
    const int dir = (int)(i & 1);
    const float* w = dir ? wb : wf;
    float wih[12][4], whh[12][4], bi[12], bh[12];
    for (int r = 0; r < 12; ++r) {
        for (int k = 0; k < 4; ++k) { wih[r][k] = w[r * 4 + k]; whh[r][k] = w[48 + r * 4 + k]; }
        bi[r] = w[96 + r];
        bh[r] = w[108 + r];
    }
    float h[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int s = 0; s < H; ++s) {
        const int step = dir ? H - 1 - s : s;
        float x[4];
        for (int k = 0; k < 4; ++k) { const int f = 4 * step + k; x[k] = f < F ? pfreq[frame * F + f] : 0.0f; }
        float hn[4];
        for (int u = 0; u < 4; ++u) {
            float gi[3], gh[3];
            for (int g = 0; g < 3; ++g) {
                gi[g] = bi[g * 4 + u];
                gh[g] = bh[g * 4 + u];
                for (int k = 0; k < 4; ++k) { gi[g] += wih[g * 4 + u][k] * x[k]; gh[g] += whh[g * 4 + u][k] * h[k]; }
            }
            const float r = usig(gi[0] + gh[0]), z = usig(gi[1] + gh[1]), n = ade::dev::tanh_f(gi[2] + r * gh[2]);
            hn[u] = (1.0f - z) * n + z * h[u];
        }
        for (int u = 0; u < 4; ++u) { h[u] = hn[u]; fah[((size_t)frame * H + step) * 8 + dir * 4 + u] = hn[u]; }
    }
}

// cTFA output (:154, :188-194) with the block's tail: out[.., co] = at[c] * x[c] * sigmoid(fc(fa)[f]) (+ residual[c]) with c = shuffle(co)
__global__ __launch_bounds__(256) void k_ulu_apply(const float* __restrict__ x, const float* __restrict__ at, const float* __restrict__ fah,
                                                   const float* __restrict__ fa_fc_w, const float* __restrict__ fa_fc_b, const float* __restrict__ res,
                                                   float* __restrict__ out, int F, int C, int H, int shuffle, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int co = (int)(i % C);
    const long long r = i / C;
    const int f = (int)(r % F);
    const long long frame = r / F;
    const int c = shuffle ? shuffled(co, C) : co, u = f & 3;
    const float* hrow = fah + ((size_t)frame * H + (f >> 2)) * 8;
    float a = fa_fc_b[u];
    for (int k = 0; k < 8; ++k) a += fa_fc_w[u * 8 + k] * hrow[k];
    const size_t src = ((size_t)frame * F + f) * C + c;
    float v = (at[(size_t)frame * C + c] * x[src]) * usig(a);
    if (res) v += res[src];
    out[i] = v;
}

__global__ __launch_bounds__(256) void k_ulu_pcm2f(const int16_t* __restrict__ pcm, float* __restrict__ x, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < total) x[i] = (float)pcm[i] * (1.0f / 32768.0f);
}
// the resampled entry (:851-867): fp32 samples in PCM units (audio.float() interpolated by the engine); the * INV_INT16 before or after the interpolation is the
// same number either way (a power of two commutes with every rounding of the interpolation)
__global__ __launch_bounds__(256) void k_ulu_f2f(const float* __restrict__ in, float* __restrict__ x, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < total) x[i] = in[i] * (1.0f / 32768.0f);
}
// log-power + ERB merge (:725-727, :97-100): feat[frame][j], j < 129; spec is [b][re 257 | im 257][T].  band_lo / band_hi: the non-zero
// run of every ERB filter row (the filters are triangles: contiguous support).
__global__ __launch_bounds__(256) void k_ulu_feat(const float* __restrict__ spec, const float* __restrict__ erb, const int* __restrict__ band_lo,
                                                  const int* __restrict__ band_hi, float* __restrict__ feat, int T, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int j = (int)(i % kUErb);
    const long long frame = i / kUErb;
    const long long b = frame / T;
    const int t = (int)(frame - b * T);
    const float* s = spec + (size_t)b * 2 * kUBins * T + t;
    auto logp = [&](int f) { const float re = s[(size_t)f * T], im = s[(size_t)(kUBins + f) * T]; return logf(fmaxf(re * re + im * im, 1e-24f)); };
    float v;
    if (j < kULow) v = logp(j);
    else {
        v = 0.0f;
        const float* row = erb + (size_t)(j - kULow) * kUHigh;
        for (int k = band_lo[j - kULow]; k < band_hi[j - kULow]; ++k) v += row[k] * logp(kULow + k);     // ascending k: the dense matmul's order, zeros dropped
    }
    feat[i] = v;
}
// sigmoid + ERB split (:649, :102-105) + real mask on both spectrum halves (:880), in place on spec.  bin_lo / bin_hi: the bands whose
// filter covers high bin k (contiguous, at most a few).
__global__ __launch_bounds__(256) void k_ulu_mask(const float* __restrict__ m129, const float* __restrict__ erb, const int* __restrict__ bin_lo,
                                                  const int* __restrict__ bin_hi, float* __restrict__ spec, float* __restrict__ mask_tap, int T, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int f = (int)(i % kUBins);
    const long long frame = i / kUBins;
    const long long b = frame / T;
    const int t = (int)(frame - b * T);
    const float* m = m129 + (size_t)frame * kUErb;
    float v;
    if (f < kULow) v = usig(m[f]);
    else {
        v = 0.0f;
        for (int e = bin_lo[f - kULow]; e < bin_hi[f - kULow]; ++e) v += usig(m[kULow + e]) * erb[(size_t)e * kUHigh + (f - kULow)];
    }
    float* s = spec + (size_t)b * 2 * kUBins * T + t;
    s[(size_t)f * T] *= v;
    s[(size_t)(kUBins + f) * T] *= v;
    if (mask_tap) mask_tap[i] = v;
}
// rows of `stride` synthesised samples -> the first `keep` of each (the static trim: keep == stride; a dynamic export slices audio[..., :audio_len], :888-889)
__global__ __launch_bounds__(256) void k_ulu_f2pcm(const float* __restrict__ y, int16_t* __restrict__ pcm, float* __restrict__ f32, int stride, int keep, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const long long r = i / keep;
    const float v = y[r * stride + (i - r * keep)];
    if (f32) f32[i] = v;
    if (pcm) pcm[i] = (int16_t)(int)fminf(fmaxf(v * 32767.0f, -32768.0f), 32767.0f);
}

int ufail(std::string& err, int st, const std::string& msg) { err = msg; return st; }
#define UL_HIP(expr)                                                                                  \
    do {                                                                                              \
        hipError_t _e = (expr);                                                                       \
        if (_e != hipSuccess) return ufail(err, ADE_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

struct CtfaW { const float *ta_wih_t, *ta_whh_t, *ta_bih, *ta_bhh, *ta_fc_t, *ta_fc_b, *fa_f, *fa_b, *fa_fc_w, *fa_fc_b; };
struct Block {
    int type, cin, cout, width, in_width, kt, kf, stride, groups, deconv, last;
    ConvDesc conv[3];          // type 0: conv ; type 1: pconv, dconv ; type 2: pconv1, dconv, pconv2
    CtfaW ctfa;
};
struct DpPacked { const float *intra_gru, *inter_gru, *fc[2], *fc_b[2], *ln_w[2], *ln_b[2]; };

// PyTorch GRU rows of hidden unit j -> [3x8 ih | 3xH hh | 3 b_ih | 3 b_hh]: the lane format of k_intra_gru / k_inter_gru (ade_kernels.hip)
void pack_gru_lane(float* dst, const float* wih, const float* whh, const float* bih, const float* bhh, int H, int j) {
    for (int g = 0; g < 3; ++g)
        for (int k = 0; k < 8; ++k) dst[g * 8 + k] = wih[(g * H + j) * 8 + k];
    for (int g = 0; g < 3; ++g)
        for (int k = 0; k < H; ++k) dst[24 + g * H + k] = whh[(g * H + j) * H + k];
    for (int g = 0; g < 3; ++g) { dst[24 + 3 * H + g] = bih[g * H + j]; dst[24 + 3 * H + 3 + g] = bhh[g * H + j]; }
}

}  // namespace

struct UlunasEngine : SubEngine {
    int device = 0, L = 0 /* one window */, n_win = 1, T = 0, out_len_ = 0;
    int syn_len = 0;                   // samples per row the ISTFT writes: out_len_ for a static export, 256 T (the kept tail) for a dynamic one
    bool ta2 = !(getenv("ADE_ULU_TA2") && atoi(getenv("ADE_ULU_TA2")) == 0);     // the single-wavefront time-attention kernel (k_ulu_ta2; ADE_ULU_TA2=0: the four-wavefront form)
    ade_stft_handle plan = nullptr;
    float* d_w = nullptr;
    const float* erb = nullptr;
    int* d_tab = nullptr;              // band_lo[64] | band_hi[64] | bin_lo[192] | bin_hi[192]
    Block blocks[10];
    DpPacked dp[2];
    int capacity = 0;
    float* ws = nullptr;
    float *xf = nullptr, *spec = nullptr, *yf = nullptr, *bufA = nullptr, *bufB = nullptr, *bufC = nullptr, *bufD = nullptr, *dpa = nullptr, *dpb = nullptr, *skip[5] = {}, *zt = nullptr, *pfreq = nullptr, *at = nullptr,
          *fah = nullptr, *rnn = nullptr, *dpm = nullptr, *mask_tap = nullptr;

    // Row groups on side streams (round 5): the network between the two STFTs runs per group of rows, group k > 0 on its own stream between a fork and a join event, so that
    // one group's latency-bound launches (the time-attention recurrences: one workgroup per clip, 63 dependent steps, 20 % of the call with the chip nearly idle) run under
    // another group's bandwidth-bound ones.  Same kernels on the same rows: the same bits.  ADE_ULU_GROUPS = 1 .. 4 (1: one stream).
    int groups_ = getenv("ADE_ULU_GROUPS") ? std::min(4, std::max(1, atoi(getenv("ADE_ULU_GROUPS")))) : 2;
    hipStream_t side[3] = {};
    hipEvent_t ev_fork = nullptr, ev_join[3] = {};
    void shift_rows(long long rows);
    void run_net(hipStream_t s, int B);

    ~UlunasEngine() override {
        (void)hipSetDevice(device);
        for (int k = 0; k < 3; ++k) { if (side[k]) (void)hipStreamDestroy(side[k]); if (ev_join[k]) (void)hipEventDestroy(ev_join[k]); }
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        if (plan) ade_stft_destroy(plan);
        if (d_w) (void)hipFree(d_w);
        if (d_tab) (void)hipFree(d_tab);
        if (ws) (void)hipFree(ws);
    }
    int frames() const override { return T; }
    // batch-fold (:866-871, :886-887): a call is n_win whole windows back to back; each is an independent clip and the outputs are
    // stitched in place, so folding is only a reinterpretation of the rows (W is a multiple of the hop: every window returns W samples)
    int in_len() const override { return L * n_win; }
    int out_len() const override { return out_len_ * n_win; }
    bool accepts_float_input() const override { return true; }
    int reserve(int batch, std::string& err) override;
    int run(hipStream_t s, const int16_t* d_in, int batch, int16_t* d_out, float* d_f32, std::string& err) override;
    int tap(hipStream_t s, const char* name, int batch, float* out, size_t count, size_t* written, std::string& err) override;
    void ctfa(hipStream_t s, const Block& bk, const float* x, const float* res, float* out, int B, int shuffle);
    float* run_block(hipStream_t s, const Block& bk, const float* x, const float* x2, float* t0, float* t1, float* dst, int B);
};

int ulunas_create(const std::map<std::string, Tensor>& tensors, int in_len, int n_win, int dynamic_keep, int device, SubEngine** out, std::string& err) {
    *out = nullptr;
    if (dynamic_keep > 0 && n_win != 1) return ufail(err, ADE_ERR_BAD_VALUE, "ul_unas: batch folding requires a static shape (Export_UL_UNAS.py:39)");
    if (in_len < kUNfft) return ufail(err, ADE_ERR_SHAPE_MISMATCH, "ul_unas: input_audio_length shorter than one 512-sample frame");
    if (n_win < 1 || (n_win > 1 && in_len % kUHop)) return ufail(err, ADE_ERR_BAD_VALUE, "ul_unas: fold windows must be whole hops");
    // ULUNAS() defaults (:655-668)
    static const int types[5] = {0, 2, 1, 2, 1}, strides[5] = {2, 2, 1, 1, 1}, groups[5] = {1, 2, 2, 2, 2}, channels[5] = {12, 24, 24, 32, 16},
                     kts[5] = {3, 2, 2, 1, 1}, kfs[5] = {3, 3, 3, 5, 5}, widths[5] = {65, 33, 33, 33, 33};
    std::vector<float> arena;
    auto put = [&](const float* src, size_t n) { const size_t at = arena.size(); arena.resize(at + ((n + 63) & ~(size_t)63), 0.0f); if (src) memcpy(&arena[at], src, n * sizeof(float)); return at; };
    bool ok = true;
    auto get = [&](const std::string& name, std::vector<int> dims) -> const float* {
        auto it = tensors.find(name);
        if (it == tensors.end()) { if (ok) err = "weights: tensor missing: " + name; ok = false; return nullptr; }
        if (it->second.dims != dims) { if (ok) err = "weights: tensor has the wrong shape: " + name; ok = false; return nullptr; }
        return it->second.data;
    };
    struct Fix { const float** dst; size_t at; };
    std::vector<Fix> fix;
    auto want = [&](const float** dst, const std::string& name, std::vector<int> dims) {
        const float* p = get(name, dims);
        size_t n = 1;
        for (int d : dims) n *= (size_t)d;
        fix.push_back({dst, put(p, p ? n : 0)});
    };
    UlunasEngine* e = new UlunasEngine();
    auto bail = [&](int st) { delete e; return st; };
    e->device = device; e->L = in_len; e->n_win = n_win; e->T = in_len / kUHop + 1;
    // static: the trim [256 : raw - 256] = 256 (T - 1) samples; dynamic (:43, :888-889): the ISTFT keeps everything after the first half window (256 T samples) and the
    // wrapper slices audio[..., :audio_len] with audio_len = the CALLER-rate input length
    e->syn_len = dynamic_keep > 0 ? kUHop * e->T : kUHop * (e->T - 1);
    e->out_len_ = dynamic_keep > 0 ? std::min(e->syn_len, dynamic_keep) : e->syn_len;
    auto conv = [&](ConvDesc& d, const std::string& wname, const std::string& aname, int cin, int cout, int fi, int fo, int kt, int kf, int stride, int g, int deconv,
                    bool act, int shuffle) {
        d = ConvDesc{nullptr, nullptr, nullptr, nullptr, nullptr, cin, cout, fi, fo, kt, kf, stride, g, deconv, shuffle};
        if (deconv) want(&d.w, wname + "_w", {cin, cout / g, kt, kf});
        else want(&d.w, wname + "_w", {cout, cin / g, kt, kf});
        want(&d.b, wname + "_b", {cout});
        if (act) {
            want(&d.pos, aname + "pos", {cout, fo});
            want(&d.neg, aname + "neg", {cout, fo});
            want(&d.abias, aname + "bias", {cout, fo});
        }
    };
    // transposed copies for the time-attention GRU (lanes read consecutive floats)
    std::vector<std::vector<float>> keep;
    auto ctfa = [&](CtfaW& c, const std::string& p, int C) {
        const int H = 2 * C;
        const float* wih = get(p + "ta_weight_ih_l0", {3 * H, C});
        const float* whh = get(p + "ta_weight_hh_l0", {3 * H, H});
        const float* fcw = get(p + "ta_fc_w", {C, H});
        if (!ok) return;
        std::vector<float> a((size_t)C * 3 * H), b((size_t)H * 3 * H), f((size_t)H * C);
        for (int g = 0; g < 3; ++g)
            for (int j = 0; j < H; ++j) {
                for (int k = 0; k < C; ++k) a[((size_t)k * 3 + g) * H + j] = wih[((size_t)g * H + j) * C + k];
                for (int k = 0; k < H; ++k) b[((size_t)k * 3 + g) * H + j] = whh[((size_t)g * H + j) * H + k];
            }
        for (int cc = 0; cc < C; ++cc)
            for (int k = 0; k < H; ++k) f[(size_t)k * C + cc] = fcw[(size_t)cc * H + k];
        fix.push_back({&c.ta_wih_t, put(a.data(), a.size())});
        fix.push_back({&c.ta_whh_t, put(b.data(), b.size())});
        fix.push_back({&c.ta_fc_t, put(f.data(), f.size())});
        want(&c.ta_bih, p + "ta_bias_ih_l0", {3 * H});
        want(&c.ta_bhh, p + "ta_bias_hh_l0", {3 * H});
        want(&c.ta_fc_b, p + "ta_fc_b", {C});
        for (int dir = 0; dir < 2; ++dir) {
            const std::string sfx = dir ? "_reverse" : "";
            const float *wi = get(p + "fa_weight_ih_l0" + sfx, {12, 4}), *wh = get(p + "fa_weight_hh_l0" + sfx, {12, 4}), *bi = get(p + "fa_bias_ih_l0" + sfx, {12}),
                        *bh = get(p + "fa_bias_hh_l0" + sfx, {12});
            if (!ok) return;
            std::vector<float> pk(120);
            memcpy(&pk[0], wi, 48 * 4); memcpy(&pk[48], wh, 48 * 4); memcpy(&pk[96], bi, 12 * 4); memcpy(&pk[108], bh, 12 * 4);
            fix.push_back({dir ? &c.fa_b : &c.fa_f, put(pk.data(), pk.size())});
        }
        want(&c.fa_fc_w, p + "fa_fc_w", {4, 8});
        want(&c.fa_fc_b, p + "fa_fc_b", {4});
    };
    auto make_block = [&](Block& bk, const std::string& p, int type, int cin, int cout, int width, int kt, int kf, int stride, int g, int deconv, int last) {
        const int in_width = stride == 2 ? (deconv ? width / 2 + 1 : width * 2 - 1) : width;
        bk = Block{};
        bk.type = type; bk.cin = cin; bk.cout = cout; bk.width = width; bk.in_width = in_width; bk.kt = kt; bk.kf = kf; bk.stride = stride; bk.groups = g; bk.deconv = deconv;
        bk.last = last;
        if (type == 0) conv(bk.conv[0], p + "conv", p + "act_", cin, cout, in_width, width, kt, kf, stride, g, deconv, !last, 0);
        else if (type == 1) {
            conv(bk.conv[0], p + "pconv", p + "pconv_act_", cin, cout, in_width, in_width, 1, 1, 1, g, 0, true, g == 2);
            conv(bk.conv[1], p + "dconv", p + "dconv_act_", cout, cout, in_width, width, kt, kf, stride, cout, deconv, !last, 0);
        } else {
            conv(bk.conv[0], p + "pconv1", p + "pconv1_act_", cin, cout, in_width, in_width, 1, 1, 1, g, 0, true, g == 2);
            conv(bk.conv[1], p + "dconv", p + "dconv_act_", cout, cout, in_width, width, kt, kf, stride, cout, deconv, true, 0);
            conv(bk.conv[2], p + "pconv2", "", cout, cout, width, width, 1, 1, 1, g, 0, false, 0);
        }
        ctfa(bk.ctfa, p + "ctfa_", cout);
    };
    int cin = 1;
    for (int i = 0; i < 5; ++i) {
        make_block(e->blocks[i], "encoder.en_convs." + std::to_string(i) + ".", types[i], cin, channels[i], widths[i], kts[i], kfs[i], strides[i], groups[i], 0, 0);
        cin = channels[i];
    }
    int j = 0;
    for (int i = 4; i >= 1; --i, ++j) {
        make_block(e->blocks[5 + j], "decoder.de_convs." + std::to_string(j) + ".", types[i], cin, channels[i - 1], widths[i - 1], kts[i], kfs[i], strides[i], groups[i], 1, 0);
        cin = channels[i - 1];
    }
    for (int i = 5; i < 9; ++i)
        if (e->blocks[i].type == 2 && e->blocks[i].cin == e->blocks[i].cout && e->blocks[i].stride == 1) ok = false;   // residual on a skip-summed input: not in this architecture
    make_block(e->blocks[9], "decoder.de_convs.4.", types[0], cin, 1, kUErb, kts[0], kfs[0], strides[0], groups[0], 1, 1);
    want(&e->erb, "erb_filters", {kUBands, kUHigh});
    // DPGRNN x 2 in GTCRN's lane formats
    for (int i = 0; i < 2 && ok; ++i) {
        const std::string p = "dpgrnn." + std::to_string(i) + ".";
        std::vector<float> ig(16 * 42), og(16 * 54);
        for (int grp = 0; grp < 2 && ok; ++grp) {
            const std::string r = p + "intra_rnn.rnn" + std::to_string(grp + 1) + ".";
            for (int dir = 0; dir < 2; ++dir) {
                const std::string sfx = dir ? "_reverse" : "";
                const float *wih = get(r + "weight_ih_l0" + sfx, {12, 8}), *whh = get(r + "weight_hh_l0" + sfx, {12, 4}), *bih = get(r + "bias_ih_l0" + sfx, {12}),
                            *bhh = get(r + "bias_hh_l0" + sfx, {12});
                if (!ok) break;
                for (int u = 0; u < 4; ++u) pack_gru_lane(&ig[(grp * 8 + dir * 4 + u) * 42], wih, whh, bih, bhh, 4, u);
            }
            const std::string q = p + "inter_rnn.rnn" + std::to_string(grp + 1) + ".";
            const float *wih = get(q + "weight_ih_l0", {24, 8}), *whh = get(q + "weight_hh_l0", {24, 8}), *bih = get(q + "bias_ih_l0", {24}), *bhh = get(q + "bias_hh_l0", {24});
            if (!ok) break;
            for (int u = 0; u < 8; ++u) pack_gru_lane(&og[(grp * 8 + u) * 54], wih, whh, bih, bhh, 8, u);
        }
        if (!ok) break;
        fix.push_back({&e->dp[i].intra_gru, put(ig.data(), ig.size())});
        fix.push_back({&e->dp[i].inter_gru, put(og.data(), og.size())});
        const char* part[2] = {"intra", "inter"};
        for (int k = 0; k < 2; ++k) {
            const float* fw = get(p + part[k] + "_fc.weight", {16, 16});
            if (!ok) break;
            std::vector<float> ft(256);
            for (int a = 0; a < 16; ++a)
                for (int co = 0; co < 16; ++co) ft[a * 16 + co] = fw[co * 16 + a];
            fix.push_back({&e->dp[i].fc[k], put(ft.data(), 256)});
            want(&e->dp[i].fc_b[k], p + part[k] + "_fc.bias", {16});
            want(&e->dp[i].ln_w[k], p + part[k] + "_ln.weight", {kFw, 16});
            want(&e->dp[i].ln_b[k], p + part[k] + "_ln.bias", {kFw, 16});
        }
    }
    if (!ok) return bail(err.find("missing") != std::string::npos ? ADE_ERR_MISSING_KEY : ADE_ERR_SHAPE_MISMATCH);
    if (hipSetDevice(device) != hipSuccess) return bail(ufail(err, ADE_ERR_DEVICE, "hipSetDevice failed"));
    if (hipMalloc((void**)&e->d_w, arena.size() * sizeof(float)) != hipSuccess || hipMemcpy(e->d_w, arena.data(), arena.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
        return bail(ufail(err, ADE_ERR_DEVICE, "upload of the UL-UNAS weights failed"));
    for (auto& f : fix) *f.dst = e->d_w + f.at;
    {   // supports of the ERB triangles, both ways
        const float* ef = tensors.find("erb_filters")->second.data;
        std::vector<int> tab(2 * kUBands + 2 * kUHigh, 0);
        for (int b2 = 0; b2 < kUBands; ++b2) {
            int lo = kUHigh, hi = 0;
            for (int k = 0; k < kUHigh; ++k)
                if (ef[(size_t)b2 * kUHigh + k] != 0.0f) { lo = std::min(lo, k); hi = std::max(hi, k + 1); }
            tab[b2] = lo < hi ? lo : 0;
            tab[kUBands + b2] = lo < hi ? hi : 0;
        }
        for (int k = 0; k < kUHigh; ++k) {
            int lo = kUBands, hi = 0;
            for (int b2 = 0; b2 < kUBands; ++b2)
                if (ef[(size_t)b2 * kUHigh + k] != 0.0f) { lo = std::min(lo, b2); hi = std::max(hi, b2 + 1); }
            tab[2 * kUBands + k] = lo < hi ? lo : 0;
            tab[2 * kUBands + kUHigh + k] = lo < hi ? hi : 0;
        }
        if (hipMalloc((void**)&e->d_tab, tab.size() * sizeof(int)) != hipSuccess || hipMemcpy(e->d_tab, tab.data(), tab.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess)
            return bail(ufail(err, ADE_ERR_DEVICE, "upload of the ERB tables failed"));
    }
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_ulu_ta<32>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);   // 88 KB at 32 channels
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_ulu_ta<24>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_ulu_ta2<32>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);  // 49 KB of recurrent weights + 48 KB of projections + 24 KB
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_ulu_ta2<24>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_ulu_ta2<16>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_ulu_ta2<12>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    ade_stft_config cfg{kUNfft, kUNfft, kUHop, "hann", nullptr, 1, "reflect"};          // UL-UNAS/Export_UL_UNAS.py:33-37, 936-957
    if (ade_stft_create(&cfg, device, &e->plan) != ADE_OK) return bail(ufail(err, ADE_ERR_DEVICE, std::string("ul_unas: STFT plan: ") + ade_stft_last_error(nullptr)));
    if (dynamic_keep > 0) (void)ade_stft_keep_tail(e->plan, 1);
    *out = e;
    return ADE_OK;
}

int UlunasEngine::reserve(int calls, std::string& err) {
    if (calls <= capacity) return ADE_OK;
    const int batch = calls * n_win;
    UL_HIP(hipSetDevice(device));
    UL_HIP(hipDeviceSynchronize());
    if (ws) (void)hipFree(ws);
    ws = nullptr;
    capacity = 0;
    const size_t B = batch, nfr = B * T, act = nfr * 1600;             // widest activation: 65 bins x 24 channels per frame
    struct Carve { float** p; size_t n; };
    std::vector<Carve> cs = {{&xf, B * L}, {&spec, B * 2 * kUBins * T}, {&yf, B * syn_len}, {&bufA, act}, {&bufB, act}, {&bufC, act}, {&bufD, act}, {&dpa, nfr * kFw * kCh}, {&dpb, nfr * kFw * kCh}, {&zt, nfr * 32},
                             {&pfreq, nfr * 132}, {&at, nfr * 32}, {&fah, nfr * 33 * 8}, {&rnn, nfr * kFw * kCh}, {&dpm, nfr * kFw * kCh}, {&mask_tap, nfr * kUBins}};
    for (int i = 0; i < 5; ++i) cs.push_back({&skip[i], nfr * (size_t)blocks[i].width * blocks[i].cout});
    size_t total = 0;
    for (auto& c : cs) total += (c.n + 63) & ~(size_t)63;
    UL_HIP(hipMalloc((void**)&ws, total * sizeof(float)));
    size_t at_ = 0;
    for (auto& c : cs) { *c.p = ws + at_; at_ += (c.n + 63) & ~(size_t)63; }
    for (int k = 0; k + 1 < groups_; ++k) {
        if (!side[k]) UL_HIP(hipStreamCreateWithFlags(&side[k], hipStreamNonBlocking));
        if (!ev_join[k]) UL_HIP(hipEventCreateWithFlags(&ev_join[k], hipEventDisableTiming));
    }
    if (!ev_fork) UL_HIP(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
    // let the STFT plan size its frame buffer now (it allocates lazily), so that run() never allocates
    UL_HIP(hipMemset(spec, 0, B * 2 * kUBins * T * sizeof(float)));
    if (ade_stft_synthesize(plan, spec, batch, T, yf, nullptr) != ADE_OK) return ufail(err, ADE_ERR_DEVICE, std::string("ul_unas: ") + ade_stft_last_error(plan));
    capacity = calls;
    return ADE_OK;
}

void UlunasEngine::ctfa(hipStream_t s, const Block& bk, const float* x, const float* res, float* out, int B, int shuffle) {
    const int F = bk.width, C = bk.cout, H = (F + 3) / 4;
    const long long nfr = (long long)B * T;
    const size_t ta_lds = ta2 ? (size_t)(64 * 3 * 2 * C + 64 * 2 * C + 64 * C + (C >= 32 ? 4 * C * C : 0)) * sizeof(float)
                              : (size_t)(C * 3 * 2 * C + 2 * C * 3 * 2 * C + 2 * C * C + 64 * C) * sizeof(float);
#define ADE_ULU_TA(CC)                                                                                                                                  \
    if (ta2) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_ulu_ta2<CC>), dim3((unsigned)B), dim3(256), ta_lds, s, (const float*)zt, bk.ctfa.ta_wih_t, bk.ctfa.ta_whh_t, \
                                bk.ctfa.ta_bih, bk.ctfa.ta_bhh, bk.ctfa.ta_fc_t, bk.ctfa.ta_fc_b, at, T);                                                \
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_ulu_ta<CC>), dim3((unsigned)B), dim3(256), ta_lds, s, (const float*)zt, bk.ctfa.ta_wih_t, bk.ctfa.ta_whh_t, \
                            bk.ctfa.ta_bih, bk.ctfa.ta_bhh, bk.ctfa.ta_fc_t, bk.ctfa.ta_fc_b, at, T)
    switch (C) {               // the channel counts of ULUNAS() (:665); create() rejects anything else
        case 1: ADE_ULU_TA(1); break;
        case 12: ADE_ULU_TA(12); break;
        case 16: ADE_ULU_TA(16); break;
        case 24: ADE_ULU_TA(24); break;
        default: ADE_ULU_TA(32); break;
    }
#undef ADE_ULU_TA
    hipLaunchKernelGGL(k_ulu_fa_gru, dim3((unsigned)((nfr * 2 + 255) / 256)), dim3(256), 0, s, (const float*)pfreq, bk.ctfa.fa_f, bk.ctfa.fa_b, fah, F, H, nfr);
    const long long total = nfr * F * C;
    hipLaunchKernelGGL(k_ulu_apply, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, (const float*)at, (const float*)fah, bk.ctfa.fa_fc_w, bk.ctfa.fa_fc_b, res,
                       out, F, C, H, shuffle, total);
}

// one encoder / decoder block (:264-273, :342-357, :433-453); x2 (decoder skip) is added to the input; result in dst
float* UlunasEngine::run_block(hipStream_t s, const Block& bk, const float* x, const float* x2, float* t0, float* t1, float* dst, int B) {
    auto conv = [&](const ConvDesc& d, const float* in, const float* in2, float* o, bool stats) {
        const int wsize = d.Cout * (d.Cin / d.groups) * d.kt * d.kf;      // same element count for Conv2d and ConvTranspose2d layouts
        const size_t stage = (size_t)d.kt * d.Fi * d.Cin + wsize, sq = stats ? (size_t)d.Fo * d.Cout : 0;
        const dim3 grid((unsigned)((long long)B * T));
        const size_t lds = std::max(stage, sq) * sizeof(float);
        float *z = stats ? zt : nullptr, *pq = stats ? pfreq : nullptr;
#define ADE_ULU_CONV(KT, KF, S, DC) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_ulu_conv<KT, KF, S, DC>), grid, dim3(256), lds, s, in, in2, d, o, T, wsize, z, pq)
        const int key = d.kt * 1000 + d.kf * 100 + d.stride * 10 + d.deconv;       // create() admits exactly these shapes
        switch (key) {
            case 1110: ADE_ULU_CONV(1, 1, 1, false); break;
            case 3320: ADE_ULU_CONV(3, 3, 2, false); break;
            case 3321: ADE_ULU_CONV(3, 3, 2, true); break;
            case 2320: ADE_ULU_CONV(2, 3, 2, false); break;
            case 2321: ADE_ULU_CONV(2, 3, 2, true); break;
            case 2310: ADE_ULU_CONV(2, 3, 1, false); break;
            case 2311: ADE_ULU_CONV(2, 3, 1, true); break;
            case 1510: ADE_ULU_CONV(1, 5, 1, false); break;
            default: ADE_ULU_CONV(1, 5, 1, true); break;     // 1511
        }
#undef ADE_ULU_CONV
    };
    const int tail_shuffle = (!bk.last && bk.groups == 2) ? 1 : 0;
    if (bk.type == 0) {
        conv(bk.conv[0], x, x2, t0, true);
        ctfa(s, bk, t0, nullptr, dst, B, tail_shuffle);
    } else if (bk.type == 1) {
        conv(bk.conv[0], x, x2, t0, false);
        conv(bk.conv[1], t0, nullptr, t1, true);
        ctfa(s, bk, t1, nullptr, dst, B, 0);
    } else {
        conv(bk.conv[0], x, x2, t0, false);
        conv(bk.conv[1], t0, nullptr, t1, false);
        conv(bk.conv[2], t1, nullptr, t0, true);
        const float* res = (bk.cin == bk.cout && bk.stride == 1) ? x : nullptr;      // use_residual (:397, :448-449); never together with a skip input (checked at create)
        ctfa(s, bk, t0, res, dst, B, tail_shuffle);
    }
    return dst;
}

int UlunasEngine::run(hipStream_t s, const int16_t* d_in, int batch, int16_t* d_out, float* d_f32, std::string& err) {
    if (batch == 0) return ADE_OK;
    int st = reserve(batch, err);
    if (st != ADE_OK) return st;
    const int B = batch * n_win;
    auto flat = [&](long long total) { return dim3((unsigned)((total + 255) / 256)); };
    if (float_in) hipLaunchKernelGGL(k_ulu_f2f, flat((long long)B * L), dim3(256), 0, s, float_in, xf, (long long)B * L);
    else hipLaunchKernelGGL(k_ulu_pcm2f, flat((long long)B * L), dim3(256), 0, s, d_in, xf, (long long)B * L);
    if (ade_stft_analyze(plan, xf, B, L, spec, (void*)s) != ADE_OK) return ufail(err, ADE_ERR_DEVICE, std::string("ul_unas: ") + ade_stft_last_error(plan));
    // the network per row group (see groups_): every buffer between the two STFTs is row-major in the clip, so a group is the same launch sequence on shifted pointers
    const int floor_rows = getenv("ADE_ULU_GROUPS") ? 1 : 16;          // (an explicit request groups any batch: the tests run three rows in three groups)
    const int G = B >= floor_rows * groups_ ? groups_ : 1, per = (B + G - 1) / G;
    if (G > 1) {
        UL_HIP(hipEventRecord(ev_fork, s));
        for (int k = 1; k < G; ++k) {
            const int r0 = k * per, nr = std::min(per, B - r0);
            if (nr <= 0) break;
            UL_HIP(hipStreamWaitEvent(side[k - 1], ev_fork, 0));
            shift_rows(r0);
            run_net(side[k - 1], nr);
            shift_rows(-r0);
            UL_HIP(hipEventRecord(ev_join[k - 1], side[k - 1]));
        }
        run_net(s, per);
        for (int k = 1; k < G && k * per < B; ++k) UL_HIP(hipStreamWaitEvent(s, ev_join[k - 1], 0));
    } else run_net(s, B);
    if (ade_stft_synthesize(plan, spec, B, T, yf, (void*)s) != ADE_OK) return ufail(err, ADE_ERR_DEVICE, std::string("ul_unas: ") + ade_stft_last_error(plan));
    hipLaunchKernelGGL(k_ulu_f2pcm, flat((long long)B * out_len_), dim3(256), 0, s, (const float*)yf, d_out, d_f32, syn_len, out_len_, (long long)B * out_len_);
    UL_HIP(hipGetLastError());
    return ADE_OK;
}

void UlunasEngine::shift_rows(long long rows) {
    const long long fr = rows * T;
    spec += rows * 2 * kUBins * T; mask_tap += fr * kUBins;
    bufA += fr * 1600; bufB += fr * 1600; bufC += fr * 1600; bufD += fr * 1600;
    dpa += fr * kFw * kCh; dpb += fr * kFw * kCh; rnn += fr * kFw * kCh; dpm += fr * kFw * kCh;
    zt += fr * 32; pfreq += fr * 132; at += fr * 32; fah += fr * 33 * 8;
    for (int i = 0; i < 5; ++i) skip[i] += fr * blocks[i].width * blocks[i].cout;
}

// feature extraction .. mask on the spectrum for B consecutive clips starting at the buffers' current row
void UlunasEngine::run_net(hipStream_t s, int B) {
    const long long nfr = (long long)B * T;
    auto flat = [&](long long total) { return dim3((unsigned)((total + 255) / 256)); };
    hipLaunchKernelGGL(k_ulu_feat, flat(nfr * kUErb), dim3(256), 0, s, (const float*)spec, erb, (const int*)d_tab, (const int*)(d_tab + kUBands), bufC, T, nfr * kUErb);
    const float* x = bufC;
    for (int i = 0; i < 5; ++i) x = run_block(s, blocks[i], x, nullptr, bufA, bufB, skip[i], B);
    // DPGRNN x 2 on (B, T, 33, 16): GTCRN's kernels (Export_UL_UNAS.py:561-574 == GTCRN's DPGRNN)
    const float* in = x;
    for (int i = 0; i < 2; ++i) {
        float* o = i == 0 ? dpa : dpb;
        launch_intra_gru(s, View{in, nullptr}, dp[i].intra_gru, rnn, (int)nfr);
        launch_fc_ln_res(s, rnn, View{in, nullptr}, dp[i].fc[0], dp[i].fc_b[0], dp[i].ln_w[0], dp[i].ln_b[0], dpm, B, T);
        launch_inter_gru(s, dpm, dp[i].inter_gru, rnn, B, T);
        launch_fc_ln_res(s, rnn, View{dpm, nullptr}, dp[i].fc[1], dp[i].fc_b[1], dp[i].ln_w[1], dp[i].ln_b[1], o, B, T);
        in = o;
    }
    // decoder: block i runs on x + en_outs[4 - i] (:647-648)
    const float* dx = in;
    for (int i = 0; i < 5; ++i) {
        float* dst = (i & 1) ? bufD : bufC;
        run_block(s, blocks[5 + i], dx, skip[4 - i], bufA, bufB, dst, B);
        dx = dst;
    }
    // sigmoid, ERB split, real mask on the spectrum (:649, :734-736, :880); ISTFT; PCM tail (:955, :908)
    hipLaunchKernelGGL(k_ulu_mask, flat(nfr * kUBins), dim3(256), 0, s, dx, erb, (const int*)(d_tab + 2 * kUBands), (const int*)(d_tab + 2 * kUBands + kUHigh), spec, mask_tap, T,
                       nfr * kUBins);
}

int UlunasEngine::tap(hipStream_t s, const char* name, int batch, float* out, size_t count, size_t* written, std::string& err) {
    const size_t n = (size_t)batch * n_win * T * kUBins;
    if (strcmp(name, "mask") != 0) return ufail(err, ADE_ERR_NOT_FOUND, std::string("unknown tap: ") + name);
    if (!mask_tap || batch <= 0) return ufail(err, ADE_ERR_NOT_FOUND, "tap has no data yet");
    if (count < n) return ufail(err, ADE_ERR_SHAPE_MISMATCH, "tap buffer too small");
    UL_HIP(hipStreamSynchronize(s));
    UL_HIP(hipMemcpy(out, mask_tap, n * sizeof(float), hipMemcpyDeviceToHost));
    *written = n;
    return ADE_OK;
}

}  // namespace ade
