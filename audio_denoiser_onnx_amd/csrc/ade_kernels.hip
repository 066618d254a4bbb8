// ade_kernels.hip — hand-written gfx950 (CDNA4, wave64) kernels for the GTCRN chunk path.
//
// Conventions
//  * activations live in HBM channels-last (B,T,F,16) fp32, so one (b,t,f) position is one 64-byte line;
//  * "position" kernels map ONE LANE to ONE (b,t,f) position and keep the tiny (<=24x16) weight matrices
//    wave-uniform: they are read through the scalar cache (s_load) and fed to v_fmac as SGPR operands, so the
//    VALU does nothing but FMAs and no LDS/cross-lane traffic is needed;
//  * recurrent kernels map ONE LANE to ONE hidden unit, hold that unit's weight rows in VGPRs and exchange the
//    hidden vector with wave shuffles — thousands of independent sequences fill the chip instead of one
//    sequence serialising a workgroup;
//  * STFT/ISTFT are 512-point real FFTs done as a packed 256-point radix-4 Stockham FFT, one wavefront per
//    frame, twiddles/window from small L2-resident tables, butterflies exchanged through LDS.
// Reference arithmetic being reproduced is cited per kernel (paths relative to the reference repo).
#include "ade_device.h"

namespace ade {

using namespace dev;

namespace {

// ---------------------------------------------------------------------------------------------------------
// F1 (part): per-chunk DC mean.  Export_GTCRN.py:645-647 — mean over THIS call's samples after the 2^-15 scale.
// Integer sum is exact; one rounding at the end.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pcm_mean(const int16_t* __restrict__ pcm, int L, int rows, float* __restrict__ mean) {
    // One workgroup per reference CALL: `rows` consecutive windows of L samples share one DC mean (rows = 1 unless the
    // model was exported with USE_BATCH_FOLD, where torch.mean runs over the whole input before the fold,
    // Export_GTCRN.py:647-660).  The mean is written once per window so that downstream kernels index it by row.
    __shared__ long long part[256];
    const int16_t* base = pcm + (size_t)blockIdx.x * rows * L;
    const long long n = (long long)rows * L;
    long long s = 0;
    if ((reinterpret_cast<uintptr_t>(base) & 15) == 0) {                 // eight samples per 16-byte load; integer sums are order-free
        const long long n8 = n >> 3;
        const int4* v = reinterpret_cast<const int4*>(base);
        for (long long i = threadIdx.x; i < n8; i += 256) {
            const int4 q = v[i];
            const int w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) s += (long long)(short)(w[k] & 0xffff) + (long long)(w[k] >> 16);
        }
        for (long long i = (n8 << 3) + threadIdx.x; i < n; i += 256) s += (long long)base[i];
    } else {
        for (long long i = threadIdx.x; i < n; i += 256) s += (long long)base[i];
    }
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        long long tot = 0;
        for (int i = 0; i < 256; ++i) tot += part[i];
        const float m = (float)((double)tot / ((double)n * 32768.0));
        for (int r = 0; r < rows; ++r) mean[(size_t)blockIdx.x * rows + r] = m;
    }
}

// ---------------------------------------------------------------------------------------------------------
// 256-point complex forward FFT of one wavefront's 256 points: lane holds z[lane + 64 r], r < 4, on entry and
// Z[lane + 64 r] on exit (natural order).  Radix-4 Stockham autosort, 4 passes; buf = this wave's two 256-point
// LDS planes.  Every thread of the workgroup must call it (contains __syncthreads()).
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

__device__ __forceinline__ void radix4(float2* v) {
    const float2 a0 = make_float2(v[0].x + v[2].x, v[0].y + v[2].y);
    const float2 a1 = make_float2(v[0].x - v[2].x, v[0].y - v[2].y);
    const float2 a2 = make_float2(v[1].x + v[3].x, v[1].y + v[3].y);
    const float2 d = make_float2(v[1].x - v[3].x, v[1].y - v[3].y);
    const float2 a3 = make_float2(d.y, -d.x);   // -i * d
    v[0] = make_float2(a0.x + a2.x, a0.y + a2.y);
    v[1] = make_float2(a1.x + a3.x, a1.y + a3.y);
    v[2] = make_float2(a0.x - a2.x, a0.y - a2.y);
    v[3] = make_float2(a1.x - a3.x, a1.y - a3.y);
}

__device__ __forceinline__ void fft256_wave(float2* v, float2 (*buf)[256], int lane, const float2* __restrict__ tw256) {
    // pass 0: Ns = 1 (no twiddle)
    radix4(v);
#pragma unroll
    for (int r = 0; r < 4; ++r) buf[0][4 * lane + r] = v[r];
    __syncthreads();
    int cur = 0;
#pragma unroll
    for (int pass = 1; pass < 4; ++pass) {
        const int Ns = 1 << (2 * pass);          // 4, 16, 64
        const int k = lane & (Ns - 1);
        const int tstride = 64 / Ns;             // angle -2 pi r k / (4 Ns)  ->  tw256[r * k * 64 / Ns]
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = buf[cur][lane + 64 * r];
#pragma unroll
        for (int r = 1; r < 4; ++r) v[r] = cmul(v[r], tw256[r * k * tstride]);
        radix4(v);
        if (pass < 3) {
            const int j0 = ((lane - k) << 2) + k;
#pragma unroll
            for (int r = 0; r < 4; ++r) buf[cur ^ 1][j0 + r * Ns] = v[r];
            __syncthreads();
            cur ^= 1;
        }
    }
    // after the Ns = 64 pass the output index is lane + 64 r: already in registers
}

// ---------------------------------------------------------------------------------------------------------
// F1-F5 fused, one wavefront per frame: int16 -> *2^-15 - mean -> reflect pad -> window -> 512-pt rFFT ->
// spectrum (B,T,2,260) ; mag = sqrt(re^2+im^2+1e-12) ; [mag,re,im] -> ERB merge -> feat (B,T,3,132).
// Reference: Export_GTCRN.py:637-647 (scale, DC), STFT_Process.py:303-316 (reflect pad + windowed DFT conv),
// Export_GTCRN.py:594-595,567 (magnitude, stack), :99-102 (ERB.bm).  The reference's DFT table is only ~4e-5
// relative-accurate (fp32 angles); this exact FFT differs from it by that much at the spectrum.
// PCM_IN=false is the STFT_Process operator form: float input, reference (B,2F,T) layout out, nothing else.
// ---------------------------------------------------------------------------------------------------------
// `fin` (PCM_IN only): the call's waveform as FINAL fp32 samples at the model rate -- already scaled and centred by the input sandwich (k_gt_in_*) --
// read instead of the int16 PCM, with no scale and no DC term.
template <bool PCM_IN>
__global__ __launch_bounds__(256) void k_stft(const void* __restrict__ in, const float* __restrict__ mean, int L, int T,
                                              int nframes, FftTabs tabs, BandTab erb, float* __restrict__ spec,
                                              float* __restrict__ feat, float* __restrict__ ref_spec, int center, const float* __restrict__ fin) {
    __shared__ float2 zbuf[4][2][256];
    __shared__ float hi[4][3][kErbHigh];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int frame = blockIdx.x * 4 + wave;
    const bool live = frame < nframes;
    const int b = live ? frame / T : 0;
    const int t = live ? frame - b * T : 0;
    float2 v[4];
    {
        const float dc = (PCM_IN && live && !fin) ? mean[b] : 0.0f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = lane + 64 * r;
            float s[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                int j = kHop * t + 2 * n + q;                      // streaming rows carry their own 256 samples of history: plain framing
                if (center) {
                    j -= kNfft / 2;                                // index into the un-padded chunk
                    j = j < 0 ? -j : (j >= L ? 2 * (L - 1) - j : j);   // reflect (STFT_Process.py:306-309)
                }
                float x = 0.0f;
                if (live) {
                    if (PCM_IN && fin) x = fin[(size_t)b * L + j];
                    else if (PCM_IN) x = (float)(reinterpret_cast<const int16_t*>(in)[(size_t)b * L + j]) * (1.0f / 32768.0f) - dc;
                    else x = reinterpret_cast<const float*>(in)[(size_t)b * L + j];
                }
                s[q] = x * tabs.win[2 * n + q];
            }
            v[r] = make_float2(s[0], s[1]);
        }
    }
    fft256_wave(v, zbuf[wave], lane, tabs.tw256);
    // real-FFT recombination needs Z[256-k]: publish Z through LDS
#pragma unroll
    for (int r = 0; r < 4; ++r) zbuf[wave][1][lane + 64 * r] = v[r];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 5; ++r) {
        // r < 4: bin k = lane + 64 r ; r == 4: Nyquist bin 256 (lane 0 only)
        if (r == 4 && lane != 0) break;
        const int k = r < 4 ? lane + 64 * r : 256;
        const float2 zk = zbuf[wave][1][k & 255];
        const float2 zc0 = zbuf[wave][1][(256 - k) & 255];
        const float2 zc = make_float2(zc0.x, -zc0.y);
        const float2 e = make_float2(0.5f * (zk.x + zc.x), 0.5f * (zk.y + zc.y));
        const float2 d = make_float2(0.5f * (zk.x - zc.x), 0.5f * (zk.y - zc.y));
        const float2 o = make_float2(d.y, -d.x);                 // -i d
        const float2 x = make_float2(e.x + (tabs.tw512[k].x * o.x - tabs.tw512[k].y * o.y),
                                     e.y + (tabs.tw512[k].x * o.y + tabs.tw512[k].y * o.x));
        if (live) {
            if (PCM_IN) {
                spec[((size_t)frame * 2 + 0) * kBinsPad + k] = x.x;
                spec[((size_t)frame * 2 + 1) * kBinsPad + k] = x.y;
                const float mag = sqrtf((x.x * x.x + x.y * x.y) + 1e-12f);
                if (k < kErbLow) {
                    float* fr = feat + (size_t)frame * 3 * kErbPad;
                    fr[k] = mag; fr[kErbPad + k] = x.x; fr[2 * kErbPad + k] = x.y;
                } else {
                    hi[wave][0][k - kErbLow] = mag; hi[wave][1][k - kErbLow] = x.x; hi[wave][2][k - kErbLow] = x.y;
                }
            } else {
                ref_spec[((size_t)b * 2 * kBins + k) * T + t] = x.x;
                ref_spec[((size_t)b * 2 * kBins + kBins + k) * T + t] = x.y;
            }
        }
    }
    if (PCM_IN) {
        __syncthreads();
        // ERB merge: one lane per band, banded sum == dense 192x64 matmul term for term (zeros dropped)
        const int s0 = erb.start[lane];
        float acc[3] = {0.0f, 0.0f, 0.0f};
        for (int n = 0; n < erb.count; ++n) {
            const float w = erb.w[n * kErbBands + lane];
            const int kk = min(s0 + n, kErbHigh - 1);
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[c] += hi[wave][c][kk] * w;
        }
        if (live) {
            float* fr = feat + (size_t)frame * 3 * kErbPad + kErbLow + lane;
            fr[0] = acc[0]; fr[kErbPad] = acc[1]; fr[2 * kErbPad] = acc[2];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// F6-F7a: SFE(3) + Conv2d(9->16, (1,5), stride (1,2), pad (0,2)) + folded BN + PReLU.  One lane per (frame, fo).
// Export_GTCRN.py:117-141 (SFE = zero-padded 3-tap unfold on F), :159-197,:488 (ConvBlock).  w: [k][ci][co].
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_conv0(const float* __restrict__ feat, const float* __restrict__ w,
                                               const float* __restrict__ bias, float slope, float* __restrict__ e0, int nframes) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= nframes * kF1) return;
    const int frame = idx / kF1, fo = idx - frame * kF1;
    const float* fr = feat + (size_t)frame * 3 * kErbPad;
    float v[3][7];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const int q = 2 * fo - 3 + j;
            v[c][j] = (q >= 0 && q < kErb) ? fr[c * kErbPad + q] : 0.0f;
        }
    float acc[16];
#pragma unroll
    for (int co = 0; co < 16; ++co) acc[co] = bias[co];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const int p = 2 * fo - 2 + k;                 // position in the SFE output; conv zero-pads outside [0,129)
        const bool pv = p >= 0 && p < kErb;
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int o = 0; o < 3; ++o) {
                const float x = pv ? v[c][k + o] : 0.0f;   // SFE channel c*3+o at p = feat[c][p-1+o]
#pragma unroll
                for (int co = 0; co < 16; ++co) acc[co] += w[(k * 9 + c * 3 + o) * 16 + co] * x;
            }
    }
#pragma unroll
    for (int co = 0; co < 16; ++co) acc[co] = prelu_f(acc[co], slope);
    st16(e0 + (size_t)idx * kCh, acc);
}

// F7b: Conv2d(16->16, (1,5), stride 2, groups 2) + BN + PReLU (Export_GTCRN.py:489).  w: [k][g][ci][co].
__global__ __launch_bounds__(256) void k_conv1(const float* __restrict__ e0, const float* __restrict__ w,
                                               const float* __restrict__ bias, float slope, float* __restrict__ e1, int nframes) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= nframes * kFw) return;
    const int frame = idx / kFw, fo = idx - frame * kFw;
    float acc[16];
#pragma unroll
    for (int co = 0; co < 16; ++co) acc[co] = bias[co];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const int fi = 2 * fo - 2 + k;
        float x[16];
        if (fi >= 0 && fi < kF1) ld16(e0 + ((size_t)frame * kF1 + fi) * kCh, x);
        else {
#pragma unroll
            for (int i = 0; i < 16; ++i) x[i] = 0.0f;
        }
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int ci = 0; ci < 8; ++ci)
#pragma unroll
                for (int co = 0; co < 8; ++co) acc[g * 8 + co] += w[((k * 2 + g) * 8 + ci) * 8 + co] * x[g * 8 + ci];
    }
#pragma unroll
    for (int co = 0; co < 16; ++co) acc[co] = prelu_f(acc[co], slope);
    st16(e1 + (size_t)idx * kCh, acc);
}

// ---------------------------------------------------------------------------------------------------------
// GTConvBlock part 1: (x [+ skip])[:, :8] -> SFE(3) -> 1x1 (24->16) + BN + PReLU -> h (B,T,33,16).
// Export_GTCRN.py:305-310 ; decoder input add :524-526.  pw1: [ci = c*3+o][co].
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gt_pw1(View a, View skip, const float* __restrict__ pw1, const float* __restrict__ bias,
                                                float slope, float* __restrict__ h, int nframes) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= nframes * kFw) return;
    const int frame = idx / kFw, f = idx - frame * kFw;
    float acc[16];
#pragma unroll
    for (int co = 0; co < 16; ++co) acc[co] = bias[co];
#pragma unroll
    for (int o = 0; o < 3; ++o) {
        const int ff = f - 1 + o;
        float x[8];
        if (ff >= 0 && ff < kFw) {
            view_ld8_lo(a, (size_t)idx - f + ff, frame, x);
            if (skip.x) {
                float y[8];
                view_ld8_lo(skip, (size_t)idx - f + ff, frame, y);
#pragma unroll
                for (int i = 0; i < 8; ++i) x[i] += y[i];
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = 0.0f;
        }
#pragma unroll
        for (int c = 0; c < 8; ++c)
#pragma unroll
            for (int co = 0; co < 16; ++co) acc[co] += pw1[(c * 3 + o) * 16 + co] * x[c];
    }
#pragma unroll
    for (int co = 0; co < 16; ++co) acc[co] = prelu_f(acc[co], slope);
    st16(h + (size_t)idx * kCh, acc);
}

// ---------------------------------------------------------------------------------------------------------
// GTConvBlock part 2: causal dilated depthwise 3x3 + BN + PReLU -> 1x1 (16->8) + BN -> h1 ; write the block's
// (still un-gated) interleaved output xn[2i] = h1[i], xn[2i+1] = bypass[i] (Export_GTCRN.py:311-324) and the TRA
// energy zt[b,t,c] = mean_f h1^2 (:154).  Workgroup = 7 frames x 33 bins of one chunk.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gt_dw_pw2(const float* __restrict__ h, View a, View skip, GtConvW w,
                                                   float* __restrict__ xn, float* __restrict__ zt, int T, int tiles,
                                                   const float* __restrict__ hist) {
    __shared__ float esq[kTileThreads][9];
    const int b = blockIdx.x / tiles, tile = blockIdx.x - b * tiles;
    const int tl = threadIdx.x / kFw, f = threadIdx.x - tl * kFw;
    const int t = tile * kTileFrames + tl;
    const bool live = threadIdx.x < kTileThreads && t < T;
    if (live) {
        const size_t frame = (size_t)b * T + t;
        const size_t pos = frame * kFw + f;
        float acc[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) acc[c] = w.dw_b[c];
#pragma unroll
        for (int kt = 0; kt < 3; ++kt) {
            const int tt = t - (2 - kt) * w.dilation;
            if (tt < 0 && !hist) continue;                            // causal zero pad (:234-241,314-318)
#pragma unroll
            for (int kf = 0; kf < 3; ++kf) {
                const int ff = f - 1 + kf;
                if (ff < 0 || ff >= kFw) continue;
                float x[16];
                if (tt >= 0) ld16(h + (((size_t)b * T + tt) * kFw + ff) * kCh, x);
                else ld16(hist + (((size_t)b * 2 * w.dilation + (2 * w.dilation + tt)) * kFw + ff) * kCh, x);   // streaming: the previous pushes' last 2 d frames
#pragma unroll
                for (int c = 0; c < 16; ++c) acc[c] += w.dw[(kt * 3 + kf) * 16 + c] * x[c];
            }
        }
#pragma unroll
        for (int c = 0; c < 16; ++c) acc[c] = prelu_f(acc[c], w.dw_slope);
        float h1[8];
#pragma unroll
        for (int co = 0; co < 8; ++co) h1[co] = w.pw2_b[co];
#pragma unroll
        for (int ci = 0; ci < 16; ++ci)
#pragma unroll
            for (int co = 0; co < 8; ++co) h1[co] += w.pw2[ci * 8 + co] * acc[ci];
        float by[8];
        view_ld8_hi(a, pos, frame, by);
        if (skip.x) {
            float y[8];
            view_ld8_hi(skip, pos, frame, y);
#pragma unroll
            for (int i = 0; i < 8; ++i) by[i] += y[i];
        }
        float o[16];
#pragma unroll
        for (int i = 0; i < 8; ++i) { o[2 * i] = h1[i]; o[2 * i + 1] = by[i]; }
        st16(xn + pos * kCh, o);
#pragma unroll
        for (int c = 0; c < 8; ++c) esq[threadIdx.x][c] = h1[c] * h1[c];
    }
    __syncthreads();
    if (threadIdx.x < kTileFrames * 8) {
        const int fl = threadIdx.x >> 3, c = threadIdx.x & 7;
        const int tt = tile * kTileFrames + fl;
        if (tt < T) {
            float s = 0.0f;
            for (int ff = 0; ff < kFw; ++ff) s += esq[fl * kFw + ff][c];
            zt[((size_t)b * T + tt) * 8 + c] = s / (float)kFw;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// One GRU step for the lane that owns hidden unit `j` of an H-wide GRU whose H lanes are contiguous in the wave.
// PyTorch gate order r,z,n ; n = tanh(W_in x + b_in + r (W_hn h + b_hn)) ; h' = (1-z) n + z h
// (nn.GRU as used at Export_GTCRN.py:149,337-338 ; ONNX linear_before_reset=1).  pk = this lane's packed rows:
// wi[3][8] | wh[3][H] | b_ih[3] | b_hh[3].
// ---------------------------------------------------------------------------------------------------------
template <int H>
struct GruLane {
    float wi[3][8], wh[3][H], bi[3], bh[3];
    __device__ __forceinline__ void load(const float* __restrict__ pk) {
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int k = 0; k < 8; ++k) wi[g][k] = pk[g * 8 + k];
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int k = 0; k < H; ++k) wh[g][k] = pk[24 + g * H + k];
#pragma unroll
        for (int g = 0; g < 3; ++g) { bi[g] = pk[24 + 3 * H + g]; bh[g] = pk[24 + 3 * H + 3 + g]; }
    }
    __device__ __forceinline__ float step(const float* x, float h) const {
        float gi[3], gh[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) { gi[g] = bi[g]; gh[g] = bh[g]; }
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int g = 0; g < 3; ++g) gi[g] += wi[g][k] * x[k];
#pragma unroll
        for (int k = 0; k < H; ++k) {
            const float hk = __shfl(h, k, H);
#pragma unroll
            for (int g = 0; g < 3; ++g) gh[g] += wh[g][k] * hk;
        }
        const float r = sigmoid_f(gi[0] + gh[0]);
        const float z = sigmoid_f(gi[1] + gh[1]);
        const float n = tanh_f(gi[2] + r * gh[2]);
        return (1.0f - z) * n + z * h;
    }
};

// TRA (Export_GTCRN.py:144-156): zt (B,T,8) -> GRU(8->16) over T -> Linear(16->8) -> sigmoid -> at (B,T,8).
// 16 lanes per chunk (one per hidden unit), 16 chunks per workgroup.
__global__ __launch_bounds__(256) void k_tra(const float* __restrict__ zt, const float* __restrict__ gru,
                                             const float* __restrict__ fc, const float* __restrict__ tra_rot, float* __restrict__ at, int B, int T,
                                             float* __restrict__ state) {
    const int site = blockIdx.x * 16 + (threadIdx.x >> 4);
    const int j = threadIdx.x & 15;
    const bool live = site < B;
    const int b = live ? site : B - 1;
    GruLane<16> g;
    g.load(gru + j * 78);
    // The 16 hidden values of a site live in one 16-lane DPP row: a step gathers them with 15 row rotations (VALU) instead of 16
    // ds_bpermutes; the recurrent rows and the Linear row come pre-ordered to the rotations' arrival order (rot: [3][16] | [16], packed by the
    // host from the probed rotation direction, ade_gtcrn_pack.h).
    const float* rot = tra_rot + j * 64;
    float whr[3][16];
#pragma unroll
    for (int gg = 0; gg < 3; ++gg)
#pragma unroll
        for (int sft = 0; sft < 16; ++sft) whr[gg][sft] = rot[gg * 16 + sft];
    float hs[16];                                                     // hs[s] = hidden value delivered by rotation s (hs[0] = this lane's own)
    auto rotate = [&](float hcur) {
        hs[0] = hcur;
#define ADE_ROT(S) hs[S] = row_ror<S>(hcur);
        ADE_ROT(1) ADE_ROT(2) ADE_ROT(3) ADE_ROT(4) ADE_ROT(5) ADE_ROT(6) ADE_ROT(7) ADE_ROT(8) ADE_ROT(9) ADE_ROT(10) ADE_ROT(11) ADE_ROT(12) ADE_ROT(13) ADE_ROT(14) ADE_ROT(15)
#undef ADE_ROT
    };
    auto gru_step = [&](const float* x) {                             // reads hs (rotations of the current h), returns the next h
        float gi[3], gh[3];
#pragma unroll
        for (int gg = 0; gg < 3; ++gg) { gi[gg] = g.bi[gg]; gh[gg] = g.bh[gg]; }
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int gg = 0; gg < 3; ++gg) gi[gg] += g.wi[gg][k] * x[k];
#pragma unroll
        for (int sft = 0; sft < 16; ++sft)
#pragma unroll
            for (int gg = 0; gg < 3; ++gg) gh[gg] += whr[gg][sft] * hs[sft];
        const float r = sigmoid_f(gi[0] + gh[0]);
        const float z = sigmoid_f(gi[1] + gh[1]);
        const float n = tanh_f(gi[2] + r * gh[2]);
        return (1.0f - z) * n + z * hs[0];
    };
    float fw[16];                                                     // the Linear row of output j & 7, in rotation order as well
#pragma unroll
    for (int k = 0; k < 16; ++k) fw[k] = rot[48 + k];
    const float fb = fc[(j & 7) * 17 + 16];
    float h = state ? state[(size_t)b * 16 + j] : 0.0f;              // streaming: the hidden state carried from the previous push
    // the inputs do not depend on h: a 4-slot register ring keeps three frames of loads in flight, so a step costs the recurrence, not a
    // trip to L2 (the loop was latency-bound on that load: 0.85 us per frame)
    float xq[4][8];
    rotate(h);
#pragma unroll
    for (int d = 0; d < 3; ++d)
        if (d < T) ld8(zt + ((size_t)b * T + d) * 8, xq[d]);
    for (int t0 = 0; t0 < T; t0 += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int t = t0 + u;
            if (t < T) {
                if (t + 3 < T) ld8(zt + ((size_t)b * T + t + 3) * 8, xq[(u + 3) & 3]);
                h = gru_step(xq[u]);
                rotate(h);                                            // feeds this frame's Linear and the next frame's recurrence
                float a = fb;
#pragma unroll
                for (int k = 0; k < 16; ++k) a += fw[k] * hs[k];
                if (live && j < 8) at[((size_t)b * T + t) * 8 + j] = sigmoid_f(a);
            }
        }
    }
    if (state && live) state[(size_t)b * 16 + j] = h;
}

// DPGRNN intra GRNN (Export_GTCRN.py:409-428,441-446,472-473): for every (b,t) frame, 2 groups x 2 directions of
// GRU(8->4) along F.  16 lanes per frame: lane = group*8 + dir*4 + unit ; output channel = the same number.
__global__ __launch_bounds__(256) void k_intra_gru(View x, const float* __restrict__ gru, float* __restrict__ rnn, int nframes) {
    const int site = blockIdx.x * 16 + (threadIdx.x >> 4);
    const int q = threadIdx.x & 15;
    const int grp = q >> 3, dir = (q >> 2) & 1;
    const bool live = site < nframes;
    const size_t frame = live ? site : nframes - 1;
    GruLane<4> g;
    g.load(gru + q * 42);
    float gate[4] = {1.0f, 1.0f, 1.0f, 1.0f};
    if (x.at) ld4(x.at + frame * 8 + grp * 4, gate);
    float h = 0.0f;
    for (int s = 0; s < kFw; ++s) {
        const int f = dir ? kFw - 1 - s : s;
        float xv[8];
        ld8(x.x + (frame * kFw + f) * kCh + grp * 8, xv);
#pragma unroll
        for (int i = 0; i < 4; ++i) xv[2 * i] *= gate[i];
        h = g.step(xv, h);
        if (live) rnn[(frame * kFw + f) * kCh + q] = h;
    }
}

// DPGRNN inter GRNN (Export_GTCRN.py:450-455,478-479): for every (b,f) column, 2 groups of GRU(8->8) along T.
// 16 lanes per column: lane = group*8 + unit.
__global__ __launch_bounds__(256) void k_inter_gru(const float* __restrict__ x, const float* __restrict__ gru,
                                                   float* __restrict__ rnn, int B, int T, float* __restrict__ state) {
    const int site = blockIdx.x * 16 + (threadIdx.x >> 4);
    const int q = threadIdx.x & 15;
    const int grp = q >> 3;
    const bool live = site < B * kFw;
    const int sc = live ? site : B * kFw - 1;
    const int b = sc / kFw, f = sc - b * kFw;
    GruLane<8> g;
    g.load(gru + q * 54);
    float h = state ? state[(size_t)sc * kCh + q] : 0.0f;             // streaming: carried per (stream, bin) column
    float xq[4][8];                                                   // input ring as in k_tra: three frames of loads in flight
    const float* xb = x + ((size_t)b * T * kFw + f) * kCh + grp * 8;  // frame t at xb + t * kFw * kCh
#pragma unroll
    for (int d = 0; d < 3; ++d)
        if (d < T) ld8(xb + (size_t)d * kFw * kCh, xq[d]);
    for (int t0 = 0; t0 < T; t0 += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int t = t0 + u;
            if (t < T) {
                if (t + 3 < T) ld8(xb + (size_t)(t + 3) * kFw * kCh, xq[(u + 3) & 3]);
                h = g.step(xq[u], h);
                if (live) rnn[(((size_t)b * T + t) * kFw + f) * kCh + q] = h;
            }
        }
    }
    if (state && live) state[(size_t)sc * kCh + q] = h;
}

// The same recurrence with the hidden state gathered by DPP row rotations (callers that pack `rot`, DpW::inter_rot): lane = 2 * unit + group
// inside the site's 16-lane row, so a rotation by 2 s stays inside the lane's own group and the 8 hidden values arrive by 7 rotations (VALU)
// instead of 8 ds_bpermutes; rot holds each lane's 3 x 8 recurrent rows in the rotations' arrival order.
__global__ __launch_bounds__(256) void k_inter_gru_rot(const float* __restrict__ x, const float* __restrict__ gru, const float* __restrict__ rot,
                                                       float* __restrict__ rnn, int B, int T, float* __restrict__ state) {
    const int site = blockIdx.x * 16 + (threadIdx.x >> 4);
    const int lane = threadIdx.x & 15;
    const int grp = lane & 1, unit = lane >> 1, q = grp * 8 + unit;   // q = output channel = packed row of the input weights / biases
    const bool live = site < B * kFw;
    const int sc = live ? site : B * kFw - 1;
    const int b = sc / kFw, f = sc - b * kFw;
    const float* pk = gru + q * 54;
    float wi[3][8], whr[3][8], bi[3], bh[3];
#pragma unroll
    for (int gg = 0; gg < 3; ++gg) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { wi[gg][k] = pk[gg * 8 + k]; whr[gg][k] = rot[lane * 24 + gg * 8 + k]; }
        bi[gg] = pk[48 + gg];
        bh[gg] = pk[51 + gg];
    }
    float h = state ? state[(size_t)sc * kCh + q] : 0.0f;
    float xq[4][8];
    const float* xb = x + ((size_t)b * T * kFw + f) * kCh + grp * 8;
#pragma unroll
    for (int d = 0; d < 3; ++d)
        if (d < T) ld8(xb + (size_t)d * kFw * kCh, xq[d]);
    for (int t0 = 0; t0 < T; t0 += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int t = t0 + u;
            if (t < T) {
                if (t + 3 < T) ld8(xb + (size_t)(t + 3) * kFw * kCh, xq[(u + 3) & 3]);
                float hs[8];
                hs[0] = h;
#define ADE_ROT(S) hs[S] = row_ror<2 * S>(h);
                ADE_ROT(1) ADE_ROT(2) ADE_ROT(3) ADE_ROT(4) ADE_ROT(5) ADE_ROT(6) ADE_ROT(7)
#undef ADE_ROT
                float gi[3], gh[3];
#pragma unroll
                for (int gg = 0; gg < 3; ++gg) { gi[gg] = bi[gg]; gh[gg] = bh[gg]; }
#pragma unroll
                for (int k = 0; k < 8; ++k)
#pragma unroll
                    for (int gg = 0; gg < 3; ++gg) gi[gg] += wi[gg][k] * xq[u][k];
#pragma unroll
                for (int sft = 0; sft < 8; ++sft)
#pragma unroll
                    for (int gg = 0; gg < 3; ++gg) gh[gg] += whr[gg][sft] * hs[sft];
                const float r = sigmoid_f(gi[0] + gh[0]);
                const float z = sigmoid_f(gi[1] + gh[1]);
                const float n = tanh_f(gi[2] + r * gh[2]);
                h = (1.0f - z) * n + z * h;
                if (live) rnn[(((size_t)b * T + t) * kFw + f) * kCh + q] = h;
            }
        }
    }
    if (state && live) state[(size_t)sc * kCh + q] = h;
}

// Linear(16,16) + LayerNorm((33,16), eps 1e-8, affine) + residual (Export_GTCRN.py:447-448,473-475,479-481).
// Workgroup = 7 frames x 33 bins; two-pass moments (mean, then centred sum of squares) through LDS.
__global__ __launch_bounds__(256) void k_fc_ln_res(const float* __restrict__ rnn, View res, const float* __restrict__ fc,
                                                   const float* __restrict__ fc_b, const float* __restrict__ ln_w,
                                                   const float* __restrict__ ln_b, float* __restrict__ out, int T, int tiles) {
    __shared__ float red[kTileThreads];
    __shared__ float stat[kTileFrames];
    const int b = blockIdx.x / tiles, tile = blockIdx.x - b * tiles;
    const int tl = threadIdx.x / kFw, f = threadIdx.x - tl * kFw;
    const int t = tile * kTileFrames + tl;
    const bool live = threadIdx.x < kTileThreads && t < T;
    const size_t frame = (size_t)b * T + (live ? t : 0);
    const size_t pos = frame * kFw + (live ? f : 0);
    float v[16];
    if (live) {
        float r[16];
        ld16(rnn + pos * kCh, r);
#pragma unroll
        for (int co = 0; co < 16; ++co) v[co] = fc_b[co];
#pragma unroll
        for (int k = 0; k < 16; ++k)
#pragma unroll
            for (int co = 0; co < 16; ++co) v[co] += fc[k * 16 + co] * r[k];
        float s = 0.0f;
#pragma unroll
        for (int co = 0; co < 16; ++co) s += v[co];
        red[threadIdx.x] = s;
    }
    __syncthreads();
    if (threadIdx.x < kTileFrames) {
        float s = 0.0f;
        for (int ff = 0; ff < kFw; ++ff) s += red[threadIdx.x * kFw + ff];
        stat[threadIdx.x] = s / (float)(kFw * kCh);
    }
    __syncthreads();
    float mean = 0.0f;
    if (live) {
        mean = stat[tl];
        float s = 0.0f;
#pragma unroll
        for (int co = 0; co < 16; ++co) { const float d = v[co] - mean; s += d * d; }
        red[threadIdx.x] = s;
    }
    __syncthreads();
    if (threadIdx.x < kTileFrames) {
        float s = 0.0f;
        for (int ff = 0; ff < kFw; ++ff) s += red[threadIdx.x * kFw + ff];
        stat[threadIdx.x] = 1.0f / sqrtf(s / (float)(kFw * kCh) + 1e-8f);
    }
    __syncthreads();
    if (live) {
        const float rstd = stat[tl];
        float x[16], gw[16], gb[16];
        view_ld16(res, pos, frame, x);
        ld16(ln_w + f * kCh, gw);
        ld16(ln_b + f * kCh, gb);
#pragma unroll
        for (int co = 0; co < 16; ++co) x[co] += (v[co] - mean) * rstd * gw[co] + gb[co];
        st16(out + pos * kCh, x);
    }
}

// ---------------------------------------------------------------------------------------------------------
// Decoder tail.  ConvTranspose2d((1,5), stride (1,2), pad (0,2)): fo = 2 fi - 2 + k (Export_GTCRN.py:515-516).
// One lane per INPUT column m produces the output pair fo = 2m (taps k=0,2,4 <- fi=m+1,m,m-1) and 2m+1
// (taps k=1,3 <- fi=m+1,m), which keeps the weights wave-uniform.
// de3: (x + e1) 16->16 groups 2, PReLU.  w: [k][g][ci][co] ; de4: (d3 + e0) 16->2, Tanh.  w: [k][ci][co].
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_deconv3(View a, View skip, const float* __restrict__ w, const float* __restrict__ bias,
                                                 float slope, float* __restrict__ d3, int nframes) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= nframes * kFw) return;
    const int frame = idx / kFw, m = idx - frame * kFw;
    float ev[16], od[16];
#pragma unroll
    for (int co = 0; co < 16; ++co) { ev[co] = bias[co]; od[co] = bias[co]; }
#pragma unroll
    for (int dlt = -1; dlt <= 1; ++dlt) {
        const int fi = m + dlt;
        if (fi < 0 || fi >= kFw) continue;
        float x[16], y[16];
        view_ld16(a, (size_t)idx + dlt, frame, x);
        view_ld16(skip, (size_t)idx + dlt, frame, y);
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] += y[i];
        const int ke = 2 - 2 * dlt;       // even output tap: fi=m+1 -> k=0, m -> 2, m-1 -> 4
        const int ko = 3 - 2 * dlt;       // odd  output tap: fi=m+1 -> k=1, m -> 3
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int ci = 0; ci < 8; ++ci)
#pragma unroll
                for (int co = 0; co < 8; ++co) {
                    ev[g * 8 + co] += w[((ke * 2 + g) * 8 + ci) * 8 + co] * x[g * 8 + ci];
                    if (dlt >= 0) od[g * 8 + co] += w[((ko * 2 + g) * 8 + ci) * 8 + co] * x[g * 8 + ci];
                }
    }
#pragma unroll
    for (int co = 0; co < 16; ++co) { ev[co] = prelu_f(ev[co], slope); od[co] = prelu_f(od[co], slope); }
    float* o = d3 + ((size_t)frame * kF1 + 2 * m) * kCh;
    st16(o, ev);
    if (2 * m + 1 < kF1) st16(o + kCh, od);
}

__global__ __launch_bounds__(256) void k_deconv4(const float* __restrict__ d3, const float* __restrict__ e0,
                                                 const float* __restrict__ w, const float* __restrict__ bias,
                                                 float* __restrict__ mask, int nframes) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= nframes * kF1) return;
    const int frame = idx / kF1, m = idx - frame * kF1;
    float ev[2] = {bias[0], bias[1]}, od[2] = {bias[0], bias[1]};
#pragma unroll
    for (int dlt = -1; dlt <= 1; ++dlt) {
        const int fi = m + dlt;
        if (fi < 0 || fi >= kF1) continue;
        float x[16], y[16];
        ld16(d3 + ((size_t)idx + dlt) * kCh, x);
        ld16(e0 + ((size_t)idx + dlt) * kCh, y);
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] += y[i];
        const int ke = 2 - 2 * dlt, ko = 3 - 2 * dlt;
#pragma unroll
        for (int ci = 0; ci < 16; ++ci)
#pragma unroll
            for (int co = 0; co < 2; ++co) {
                ev[co] += w[(ke * 16 + ci) * 2 + co] * x[ci];
                if (dlt >= 0) od[co] += w[(ko * 16 + ci) * 2 + co] * x[ci];
            }
    }
    float* mr = mask + (size_t)frame * 2 * kErbPad;
#pragma unroll
    for (int co = 0; co < 2; ++co) {
        mr[co * kErbPad + 2 * m] = tanhf(ev[co]);
        if (2 * m + 1 < kErb) mr[co * kErbPad + 2 * m + 1] = tanhf(od[co]);
    }
}

// ---------------------------------------------------------------------------------------------------------
// F12-F13a, one wavefront per frame: ERB split of the 2x129 mask (banded == dense 64x192 matmul), complex ratio
// mask Y = X * M (Export_GTCRN.py:104-107,583-590), then irFFT-512 as a packed 256-point FFT and the synthesis
// window (== the reference's ConvTranspose1d kernel scale*cos*w/N, -scale*sin*w/N of STFT_Process.py:239-251,
// which ignores Im of the DC and Nyquist bins).  Output: windowed frames (B,T,512) for the overlap-add kernel.
// MASKED=false is the STFT_Process istft_B operator form: reference (B,2F,T) spectrum in, no mask.
// ---------------------------------------------------------------------------------------------------------
template <bool MASKED>
__global__ __launch_bounds__(256) void k_istft(const float* __restrict__ spec, const float* __restrict__ mask, BandTab bs,
                                               FftTabs tabs, float* __restrict__ frames, int T, int nframes) {
    __shared__ float2 ybuf[4][kBins + 3];
    __shared__ float2 zbuf[4][2][256];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int frame = blockIdx.x * 4 + wave;
    const bool live = frame < nframes;
    const int fr = live ? frame : nframes - 1;
#pragma unroll
    for (int r = 0; r < 5; ++r) {
        if (r == 4 && lane != 0) break;
        const int k = r < 4 ? lane + 64 * r : 256;
        float2 y;
        if (MASKED) {
            const float xr = spec[((size_t)fr * 2 + 0) * kBinsPad + k], xi = spec[((size_t)fr * 2 + 1) * kBinsPad + k];
            const float* mr = mask + (size_t)fr * 2 * kErbPad;
            float m0, m1;
            if (k < kErbLow) {
                m0 = mr[k]; m1 = mr[kErbPad + k];
            } else {
                const int o = k - kErbLow;
                const int s0 = bs.start[o];
                m0 = 0.0f; m1 = 0.0f;
                for (int n = 0; n < bs.count; ++n) {
                    const float wv = bs.w[n * kErbHigh + o];
                    const int jj = min(s0 + n, kErbBands - 1);
                    m0 += mr[kErbLow + jj] * wv;
                    m1 += mr[kErbPad + kErbLow + jj] * wv;
                }
            }
            y = make_float2(xr * m0 - xi * m1, xi * m0 + xr * m1);
        } else {
            const int b = fr / T, t = fr - b * T;
            y = make_float2(spec[((size_t)b * 2 * kBins + k) * T + t], spec[((size_t)b * 2 * kBins + kBins + k) * T + t]);
        }
        if (k == 0 || k == 256) y.y = 0.0f;
        ybuf[wave][k] = y;
    }
    __syncthreads();
    float2 v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int k = lane + 64 * r;
        const float2 yk = ybuf[wave][k];
        const float2 yc0 = ybuf[wave][256 - k];
        const float2 yc = make_float2(yc0.x, -yc0.y);
        const float2 e = make_float2(0.5f * (yk.x + yc.x), 0.5f * (yk.y + yc.y));
        const float2 d = make_float2(0.5f * (yk.x - yc.x), 0.5f * (yk.y - yc.y));
        const float2 wc = make_float2(tabs.tw512[k].x, -tabs.tw512[k].y);   // e^{+2 pi i k/512}
        const float2 o = cmul(d, wc);
        // Z = E + i O ; inverse FFT via conj(FFT(conj(Z)))
        v[r] = make_float2(e.x - o.y, -(e.y + o.x));
    }
    fft256_wave(v, zbuf[wave], lane, tabs.tw256);
    if (live) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = lane + 64 * r;
            const float x0 = v[r].x * (1.0f / 256.0f) * tabs.win[2 * n];
            const float x1 = -v[r].y * (1.0f / 256.0f) * tabs.win[2 * n + 1];
            *reinterpret_cast<float2*>(frames + (size_t)frame * kNfft + 2 * n) = make_float2(x0, x1);
        }
    }
}

// F13b-F14: overlap-add of the two frames covering each hop, / sum(w^2) (STFT_Process.py:330-333), then
// * 32767, clamp, truncating cast (Export_GTCRN.py:681,690).  4 samples per lane.
__global__ __launch_bounds__(256) void k_ola_pcm(const float* __restrict__ frames, const float* __restrict__ win_sum, int T,
                                                 int B, int16_t* __restrict__ pcm, float* __restrict__ f32) {
    const int per = (T - 1) * (kHop / 4);
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)B * per) return;
    const int b = (int)(idx / per);
    const int n = (int)(idx - (long long)b * per) * 4;
    const int j = n >> 8, r = n & 255;
    float a[4], c[4], ws[4], v[4];
    ld4(frames + ((size_t)b * T + j) * kNfft + kHop + r, a);
    ld4(frames + ((size_t)b * T + j + 1) * kNfft + r, c);
    ld4(win_sum + r, ws);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = (a[i] + c[i]) / ws[i];
    const size_t o = (size_t)b * (size_t)(T - 1) * kHop + n;
    if (f32) st4(f32 + o, v);
    if (pcm) {
        short q[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float s = fminf(fmaxf(v[i] * 32767.0f, -32768.0f), 32767.0f);
            q[i] = (short)(int)s;
        }
        *reinterpret_cast<short4*>(pcm + o) = make_short4(q[0], q[1], q[2], q[3]);
    }
}

// ---- streaming: state carried across pushes of T frames per stream ---------------------------------
// history of a GTConvBlock's depthwise input: out = last `depth` frames of [hist_in | h]  (depth = 2 * dilation)
__global__ __launch_bounds__(256) void k_hist_shift(const float* __restrict__ hist_in, const float* __restrict__ h, float* __restrict__ hist_out,
                                                    int T, int depth, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int per = kFw * kCh, e = (int)(i % per);
    const long long fr = i / per;
    const int b = (int)(fr / depth), k = (int)(fr - (long long)b * depth);
    const int src = k + T;                                   // index into the concatenation [hist (depth) | h (T)]
    hist_out[i] = src < depth ? hist_in[((size_t)b * depth + src) * per + e] : h[((size_t)b * T + (src - depth)) * per + e];
}
// concat[b] = [256 samples of history | the push]; a fresh stream's history is the reflection x[256], ..., x[1] (STFT_Process.py:306-309)
__global__ __launch_bounds__(256) void k_stream_concat(const int16_t* __restrict__ hist, const int16_t* __restrict__ in, int16_t* __restrict__ concat,
                                                       int P, int first, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int row = P + kHop, b = (int)(i / row), j = (int)(i - (long long)b * row);
    concat[i] = j >= kHop ? in[(size_t)b * P + (j - kHop)] : (first ? in[(size_t)b * P + (kHop - j)] : hist[(size_t)b * kHop + j]);
}
__global__ __launch_bounds__(256) void k_stream_keep(const int16_t* __restrict__ concat, int16_t* __restrict__ hist, int16_t* __restrict__ prev, int P,
                                                     long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int b = (int)(i / kHop), j = (int)(i - (long long)b * kHop);
    hist[i] = concat[(size_t)b * (P + kHop) + P + j];
    if (j == 0) prev[b] = concat[(size_t)b * (P + kHop) + P - 1];      // the sample before the history: the end reflection needs 257 samples
}
// end of stream: the one frame the one-shot graph computes past the signal, [last 256 samples | reflection x[L-2], ..., x[L-257]] (STFT_Process.py:306-309)
__global__ __launch_bounds__(256) void k_stream_concat_flush(const int16_t* __restrict__ hist, const int16_t* __restrict__ prev, int16_t* __restrict__ concat,
                                                             long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int b = (int)(i / (2 * kHop)), j = (int)(i - (long long)b * 2 * kHop);
    if (j < kHop) concat[i] = hist[(size_t)b * kHop + j];
    else { const int r = j - kHop; concat[i] = r < kHop - 1 ? hist[(size_t)b * kHop + (kHop - 2 - r)] : prev[b]; }
}
// overlap-add with a carried half frame: output hop k of the push = second half of frame k - 1 (the carry for k = 0) + first half of
// frame k, / sum(w^2); one hop behind the input (a frame is complete one hop after its centre).  The very first hop of a stream has no
// predecessor and is written as zeros.  PCM tail as k_ola_pcm.
__global__ __launch_bounds__(256) void k_ola_pcm_stream(const float* __restrict__ frames, const float* __restrict__ carry, const float* __restrict__ win_sum,
                                                        int T, int B, int first, int16_t* __restrict__ pcm, float* __restrict__ f32) {
    const int per = T * (kHop / 4);
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)B * per) return;
    const int b = (int)(idx / per);
    const int n = (int)(idx - (long long)b * per) * 4;
    const int j = n >> 8, r = n & 255;
    float a[4], c[4], ws[4], v[4];
    if (j > 0) ld4(frames + ((size_t)b * T + j - 1) * kNfft + kHop + r, a);
    else ld4(carry + (size_t)b * kHop + r, a);
    ld4(frames + ((size_t)b * T + j) * kNfft + r, c);
    ld4(win_sum + r, ws);
    const bool blank = first && j == 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = blank ? 0.0f : (a[i] + c[i]) / ws[i];
    const size_t o = (size_t)b * (size_t)T * kHop + n;
    if (f32) st4(f32 + o, v);
    if (pcm) {
        short q[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) q[i] = (short)(int)fminf(fmaxf(v[i] * 32767.0f, -32768.0f), 32767.0f);
        *reinterpret_cast<short4*>(pcm + o) = make_short4(q[0], q[1], q[2], q[3]);
    }
}
__global__ __launch_bounds__(256) void k_carry_keep(const float* __restrict__ frames, float* __restrict__ carry, int T, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int b = (int)(i / kHop), r = (int)(i - (long long)b * kHop);
    carry[i] = frames[((size_t)b * T + T - 1) * kNfft + kHop + r];
}

// ---- driver-edge resampling: torch.nn.functional.interpolate(mode='linear', align_corners=False) over the last axis (fp32) -----
__device__ __forceinline__ void lerp_coords(int i, int Lin, float scale, int& i0, int& i1, float& l1) {
    float src = scale * ((float)i + 0.5f) - 0.5f;
    src = src < 0.0f ? 0.0f : src;
    i0 = (int)src;
    if (i0 > Lin - 1) i0 = Lin - 1;
    i1 = i0 + (i0 < Lin - 1 ? 1 : 0);
    l1 = src - (float)i0;
}
// mode 1: torch.nan_to_num(nan=0, posinf=32767, neginf=-32768); mode 2: torch.where(isnan, 0, x) -- infinities pass (H-GTCRN, Export_H_GTCRN.py:1056).
// Every finite value passes in both.
__device__ __forceinline__ float nan_to_num_pcm(float v, int mode) {
    if (v != v) return 0.0f;
    if (mode == 1 && __builtin_isinf(v)) return v > 0.0f ? 32767.0f : -32768.0f;
    return v;
}
// float audio tensors (input_audio_dtype F32 / F16): the same edge on normalised floats, times `gain` = what lifts them to the PCM units the sub-engines read (a power of
// two: it commutes with every rounding of the interpolation, so the reference's scale-then-interpolate and interpolate-only forms are both this)
__global__ __launch_bounds__(256) void k_resample_in_f32(const float* __restrict__ in, float* __restrict__ out, int Lin, int Lout, float scale, float gain, long long total) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const long long r = idx / Lout;
    const int i = (int)(idx - r * Lout);
    int i0, i1;
    float l1;
    lerp_coords(i, Lin, scale, i0, i1, l1);
    const float* row = in + r * Lin;
    out[idx] = ((1.0f - l1) * row[i0] + l1 * row[i1]) * gain;
}
__global__ __launch_bounds__(256) void k_resample_in(const int16_t* __restrict__ in, float* __restrict__ out, int Lin, int Lout, float scale, long long total) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const long long r = idx / Lout;
    const int i = (int)(idx - r * Lout);
    int i0, i1;
    float l1;
    lerp_coords(i, Lin, scale, i0, i1, l1);
    const int16_t* row = in + r * Lin;
    out[idx] = (1.0f - l1) * (float)row[i0] + l1 * (float)row[i1];
}
// output edge: interpolate the model-rate float waveform, scale to PCM, clamp, cast (truncate_i32: the int32 cast of MossFormer2, which
// truncates before clamping; otherwise the float clamp then truncating cast of the STFT models)
// f32_scale: what turns the interpolated waveform into the export's F32 / F16 output (1, or 2^-15 for the families whose waveform is in PCM units); nan_to_num: the
// reference's torch.nan_to_num(nan=0, posinf=32767, neginf=-32768) / where(isnan, 0) before the tail (ZipEnhancer always, UL-UNAS for float input)
__global__ __launch_bounds__(256) void k_resample_out(const float* __restrict__ in, int16_t* __restrict__ pcm, float* __restrict__ f32, int Lin, int Lout,
                                                      float scale, float pcm_scale, int truncate_i32, float f32_scale, int nan_to_num, long long total) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const long long r = idx / Lout;
    const int i = (int)(idx - r * Lout);
    int i0, i1;
    float l1;
    lerp_coords(i, Lin, scale, i0, i1, l1);
    const float* row = in + r * Lin;
    float y = (1.0f - l1) * row[i0] + l1 * row[i1];
    if (nan_to_num) y = nan_to_num_pcm(y, nan_to_num);
    if (f32) f32[idx] = y * f32_scale;
    if (pcm) {
        const float v = y * pcm_scale;
        pcm[idx] = (int16_t)(int)fminf(fmaxf(truncate_i32 ? truncf(v) : v, -32768.0f), 32767.0f);
    }
}

// ---- GTCRN_CUSTOM's input / output sandwich for float audio, other sample rates and dynamic-length exports (Export_GTCRN.py:636-693) -------------
// input: audio.float(); [interpolate first when the caller rate is ABOVE the model rate]; * 2^-15 for int16 input; - mean; [interpolate afterwards when it is BELOW].
// Stage 1 -> tmp (rows x L1): the (optionally interpolated) samples times `gain`; stage 2: one mean per row (double accumulation, fixed order);
// stage 3 -> the model-rate waveform: tmp - mean, interpolated afterwards when lerp2 > 0.  lerp = source step of F.interpolate(scale_factor = 1 / lerp).
__global__ __launch_bounds__(256) void k_gt_in_stage1(const int16_t* __restrict__ pcm, const float* __restrict__ fin, float* __restrict__ tmp, int Lin, int L1, float lerp,
                                                      float gain, long long total) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const long long r = idx / L1;
    const int i = (int)(idx - r * L1);
    auto at = [&](int j) { return fin ? fin[r * Lin + j] : (float)pcm[r * Lin + j]; };
    float v;
    if (lerp > 0.0f) {
        int i0, i1;
        float l1;
        lerp_coords(i, Lin, lerp, i0, i1, l1);
        v = (1.0f - l1) * at(i0) + l1 * at(i1);
    } else {
        v = at(i);
    }
    tmp[idx] = v * gain;
}
__global__ __launch_bounds__(256) void k_row_mean_f32(const float* __restrict__ x, int L, float* __restrict__ mean) {
    __shared__ double part[256];
    const float* row = x + (size_t)blockIdx.x * L;
    double s = 0.0;
    for (int i = threadIdx.x; i < L; i += 256) s += (double)row[i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) mean[blockIdx.x] = (float)(part[0] / (double)L);
}
__global__ __launch_bounds__(256) void k_gt_in_stage3(const float* __restrict__ tmp, const float* __restrict__ mean, float* __restrict__ out, int L1, int Lm, float lerp,
                                                      long long total) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const long long r = idx / Lm;
    const int i = (int)(idx - r * Lm);
    const float* row = tmp + r * L1;
    const float m = mean[r];
    if (lerp > 0.0f) {
        int i0, i1;
        float l1;
        lerp_coords(i, L1, lerp, i0, i1, l1);
        out[idx] = (1.0f - l1) * (row[i0] - m) + l1 * (row[i1] - m);
    } else {
        out[idx] = row[i] - m;
    }
}
// overlap-add for the sandwich path: `keep` model-rate samples per row.  Static exports keep 256 (T - 1) (STFT_Process.py:169-176, 330-333); dynamic-length exports
// slice [n_fft / 2 : out_end(max_frames)] of the raw overlap-add, i.e. 256 T samples, and divide by the conv_transpose of the squared window over the frames that
// exist (:337-341): the last hop has only frame T - 1 under it.
__global__ __launch_bounds__(256) void k_ola_keep(const float* __restrict__ frames, const float* __restrict__ win_sum, const float* __restrict__ win, int T, int keep,
                                                  float* __restrict__ out, long long total) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const long long b = idx / keep;
    const int n = (int)(idx - b * keep), j = n >> 8, r = n & 255;
    const float a = frames[((size_t)b * T + j) * kNfft + kHop + r];
    if (j + 1 < T) out[idx] = (a + frames[((size_t)b * T + j + 1) * kNfft + r]) / win_sum[r];
    else { const float w = win[kHop + r]; out[idx] = a / (w * w); }
}
// output: [interpolate first when the caller rate is BELOW the model rate]; * 32767 for int16 output; [interpolate afterwards when it is ABOVE]; clamp + truncating
// cast for int16 (Export_GTCRN.py:673-693).  f32 receives the waveform at the output rate without the PCM scale (the F32 / F16 output of the export).
__global__ __launch_bounds__(256) void k_gt_out(const float* __restrict__ wave, int16_t* __restrict__ pcm, float* __restrict__ f32, int Lw, int Lout, float lerp,
                                                int scale_first, int nan_to_num, long long total) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const long long r = idx / Lout;
    const int i = (int)(idx - r * Lout);
    const float* row = wave + r * Lw;
    float y, q;
    if (lerp > 0.0f) {
        int i0, i1;
        float l1;
        lerp_coords(i, Lw, lerp, i0, i1, l1);
        y = (1.0f - l1) * row[i0] + l1 * row[i1];
        q = scale_first ? (1.0f - l1) * (row[i0] * 32767.0f) + l1 * (row[i1] * 32767.0f) : y * 32767.0f;
    } else {
        y = row[i];
        q = y * 32767.0f;
    }
    if (nan_to_num) {      // torch.nan_to_num(nan=0, posinf=32767, neginf=-32768) before the cast (Export_UL_UNAS.py:906-907, float input only)
        y = nan_to_num_pcm(y, nan_to_num);
        q = q != q ? 0.0f : q;       // (the clamp below maps the infinities)
    }
    if (f32) f32[idx] = y;
    if (pcm) pcm[idx] = (int16_t)(int)fminf(fmaxf(q, -32768.0f), 32767.0f);
}

__global__ void k_probe_row_ror(int* out) {
    const int lane = threadIdx.x & 15;
    out[threadIdx.x] = (int)row_ror<1>((float)lane);
}

}  // namespace

int dpp_row_ror_direction() {
    static int cached = -2;
    if (cached != -2) return cached;
    int* d = nullptr;
    int h[64] = {};
    cached = 0;
    if (hipMalloc((void**)&d, sizeof(h)) != hipSuccess) return cached;
    hipLaunchKernelGGL(k_probe_row_ror, dim3(1), dim3(64), 0, nullptr, d);
    const bool ok = hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess;
    (void)hipFree(d);
    if (!ok) return cached;
    bool up = true, down = true;
    for (int i = 0; i < 64; ++i) { up = up && h[i] == (((i & 15) + 1) & 15); down = down && h[i] == (((i & 15) + 15) & 15); }
    cached = up ? 1 : down ? -1 : 0;
    return cached;
}

// ---- launchers -------------------------------------------------------------------------------------------
void launch_gt_sandwich_in(hipStream_t s, const int16_t* pcm, const float* fin, int rows, int Lin, int L1, int Lm, float lerp1, float lerp2, float gain, float* tmp,
                           float* mean, float* out) {
    hipLaunchKernelGGL(k_gt_in_stage1, grid1((long long)rows * L1, 256), dim3(256), 0, s, pcm, fin, tmp, Lin, L1, lerp1, gain, (long long)rows * L1);
    hipLaunchKernelGGL(k_row_mean_f32, dim3((unsigned)rows), dim3(256), 0, s, (const float*)tmp, L1, mean);
    hipLaunchKernelGGL(k_gt_in_stage3, grid1((long long)rows * Lm, 256), dim3(256), 0, s, (const float*)tmp, (const float*)mean, out, L1, Lm, lerp2, (long long)rows * Lm);
}
// ---- IEEE half tensors at the ABI (ade_process_f16): bit-level conversions, so that the same source also runs under the host simulator
__device__ __forceinline__ float half_bits_to_float(unsigned h) {
    const unsigned sign = (h & 0x8000u) << 16, ex = (h >> 10) & 31u, man = h & 1023u;
    unsigned u;
    if (ex == 0) {
        if (man == 0) u = sign;
        else {                                           // subnormal: value = man * 2^-24, exact in fp32
            const float v = (float)man * 5.9604644775390625e-08f;
            u = sign | (unsigned)__float_as_int(v);
        }
    } else if (ex == 31) u = sign | 0x7f800000u | (man << 13);
    else u = sign | ((ex + 112u) << 23) | (man << 13);
    return __int_as_float((int)u);
}
__device__ __forceinline__ unsigned float_to_half_bits(float f) {      // round to nearest, ties to even; overflow -> inf; NaN stays NaN
    const unsigned u = (unsigned)__float_as_int(f);
    const unsigned sign = (u >> 16) & 0x8000u, a = u & 0x7fffffffu;
    if (a >= 0x7f800000u) return sign | 0x7c00u | (a > 0x7f800000u ? 0x200u : 0u);
    if (a >= 0x477ff000u) return sign | 0x7c00u;                        // >= 65520 rounds to infinity
    if (a < 0x33000001u) return sign;                                   // <= 2^-25 rounds to zero
    if (a < 0x38800000u) {                                              // subnormal half: man = round(|f| * 2^24)
        const float v = __int_as_float((int)a) * 16777216.0f;
        const float r = v + 12582912.0f;                                // 1.5 * 2^23: rounds v to an integer, ties to even
        return sign | ((unsigned)__float_as_int(r) & 0x7ffu);
    }
    unsigned r = a + 0xfffu + ((a >> 13) & 1u);                         // round the 13 dropped mantissa bits, ties to even
    return sign | (((r >> 13) - (112u << 10)) & 0x7fffu);
}
__global__ __launch_bounds__(256) void k_half_to_float(const uint16_t* __restrict__ in, float* __restrict__ out, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = half_bits_to_float(in[i]);
}
__global__ __launch_bounds__(256) void k_float_to_half(const float* __restrict__ in, uint16_t* __restrict__ out, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = (uint16_t)float_to_half_bits(in[i]);
}
void launch_half_to_float(hipStream_t s, const uint16_t* in, float* out, long long n) {
    if (n > 0) hipLaunchKernelGGL(k_half_to_float, grid1(n, 256), dim3(256), 0, s, in, out, n);
}
void launch_float_to_half(hipStream_t s, const float* in, uint16_t* out, long long n) {
    if (n > 0) hipLaunchKernelGGL(k_float_to_half, grid1(n, 256), dim3(256), 0, s, in, out, n);
}

void launch_gt_out(hipStream_t s, const float* wave, int16_t* pcm, float* f32, long long rows, int Lw, int Lout, float lerp, bool scale_first, int nan_to_num) {
    hipLaunchKernelGGL(k_gt_out, grid1(rows * Lout, 256), dim3(256), 0, s, wave, pcm, f32, Lw, Lout, lerp, scale_first ? 1 : 0, nan_to_num, rows * Lout);
}
void launch_gt_sandwich_out(hipStream_t s, const float* frames, FftTabs tabs, int rows, int T, int keep, float* wave, int16_t* pcm, float* f32, int Lout, float lerp,
                            bool scale_first) {
    hipLaunchKernelGGL(k_ola_keep, grid1((long long)rows * keep, 256), dim3(256), 0, s, frames, tabs.win_sum, tabs.win, T, keep, wave, (long long)rows * keep);
    hipLaunchKernelGGL(k_gt_out, grid1((long long)rows * Lout, 256), dim3(256), 0, s, (const float*)wave, pcm, f32, keep, Lout, lerp, scale_first ? 1 : 0, 0, (long long)rows * Lout);
}
void launch_pcm_mean(hipStream_t s, const int16_t* pcm, int B, int L, float* mean, int rows_per_call) {
    hipLaunchKernelGGL(k_pcm_mean, dim3(B / rows_per_call), dim3(256), 0, s, pcm, L, rows_per_call, mean);
}
void launch_stft_pcm(hipStream_t s, const int16_t* pcm, const float* mean, int B, int L, int T, FftTabs tabs, BandTab erb_bm,
                     float* spec, float* feat, bool center, const float* final_f32) {
    const int nframes = B * T;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_stft<true>), grid1(nframes, 4), dim3(256), 0, s, (const void*)pcm, mean, L, T, nframes,
                       tabs, erb_bm, spec, feat, (float*)nullptr, center ? 1 : 0, final_f32);
}
void launch_stft_ref(hipStream_t s, const float* x, int B, int L, int T, FftTabs tabs, float* ref_spec) {
    const int nframes = B * T;
    BandTab none = {nullptr, nullptr, 0, 0};
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_stft<false>), grid1(nframes, 4), dim3(256), 0, s, (const void*)x, (const float*)nullptr, L,
                       T, nframes, tabs, none, (float*)nullptr, (float*)nullptr, ref_spec, 1, (const float*)nullptr);
}
void launch_conv0(hipStream_t s, const float* feat, ConvW w, float* e0, int nframes) {
    hipLaunchKernelGGL(k_conv0, grid1((long long)nframes * kF1, 256), dim3(256), 0, s, feat, w.w, w.b, w.slope, e0, nframes);
}
void launch_conv1(hipStream_t s, const float* e0, ConvW w, float* e1, int nframes) {
    hipLaunchKernelGGL(k_conv1, grid1((long long)nframes * kFw, 256), dim3(256), 0, s, e0, w.w, w.b, w.slope, e1, nframes);
}
void launch_gt_pw1(hipStream_t s, View a, View skip, GtConvW w, float* h, int nframes) {
    hipLaunchKernelGGL(k_gt_pw1, grid1((long long)nframes * kFw, 256), dim3(256), 0, s, a, skip, w.pw1, w.pw1_b, w.pw1_slope, h,
                       nframes);
}
void launch_gt_dw_pw2(hipStream_t s, const float* h, View a, View skip, GtConvW w, float* xn, float* zt, int B, int T, const float* hist) {
    const int tiles = (T + kTileFrames - 1) / kTileFrames;
    hipLaunchKernelGGL(k_gt_dw_pw2, dim3(B * tiles), dim3(256), 0, s, h, a, skip, w, xn, zt, T, tiles, hist);
}
void launch_hist_shift(hipStream_t s, const float* hist_in, const float* h, float* hist_out, int B, int T, int depth) {
    const long long n = (long long)B * depth * kFw * kCh;
    hipLaunchKernelGGL(k_hist_shift, grid1(n, 256), dim3(256), 0, s, hist_in, h, hist_out, T, depth, n);
}
void launch_stream_concat(hipStream_t s, const int16_t* hist, const int16_t* in, int16_t* concat, int B, int P, bool first) {
    hipLaunchKernelGGL(k_stream_concat, grid1((long long)B * (P + kHop), 256), dim3(256), 0, s, hist, in, concat, P, first ? 1 : 0, (long long)B * (P + kHop));
}
void launch_stream_keep(hipStream_t s, const int16_t* concat, int16_t* hist, int16_t* prev, int B, int P) {
    hipLaunchKernelGGL(k_stream_keep, grid1((long long)B * kHop, 256), dim3(256), 0, s, concat, hist, prev, P, (long long)B * kHop);
}
void launch_stream_concat_flush(hipStream_t s, const int16_t* hist, const int16_t* prev, int16_t* concat, int B) {
    hipLaunchKernelGGL(k_stream_concat_flush, grid1((long long)B * 2 * kHop, 256), dim3(256), 0, s, hist, prev, concat, (long long)B * 2 * kHop);
}
void launch_resample_in_f32(hipStream_t s, const float* in, float* out, long long rows, int Lin, int Lout, float scale, float gain) {
    hipLaunchKernelGGL(k_resample_in_f32, grid1(rows * Lout, 256), dim3(256), 0, s, in, out, Lin, Lout, scale, gain, rows * Lout);
}
void launch_resample_in(hipStream_t s, const int16_t* in, float* out, long long rows, int Lin, int Lout, float scale) {
    hipLaunchKernelGGL(k_resample_in, grid1(rows * Lout, 256), dim3(256), 0, s, in, out, Lin, Lout, scale, rows * Lout);
}
void launch_resample_out(hipStream_t s, const float* in, int16_t* pcm, float* f32, long long rows, int Lin, int Lout, float scale, float pcm_scale, bool truncate_i32,
                         float f32_scale, int nan_to_num) {
    hipLaunchKernelGGL(k_resample_out, grid1(rows * Lout, 256), dim3(256), 0, s, in, pcm, f32, Lin, Lout, scale, pcm_scale, truncate_i32 ? 1 : 0, f32_scale, nan_to_num, rows * Lout);
}
void launch_ola_pcm_stream(hipStream_t s, const float* frames, float* carry, FftTabs tabs, int B, int T, bool first, int16_t* pcm, float* f32) {
    const long long n = (long long)B * T * (kHop / 4);
    hipLaunchKernelGGL(k_ola_pcm_stream, grid1(n, 256), dim3(256), 0, s, frames, (const float*)carry, tabs.win_sum, T, B, first ? 1 : 0, pcm, f32);
    hipLaunchKernelGGL(k_carry_keep, grid1((long long)B * kHop, 256), dim3(256), 0, s, frames, carry, T, (long long)B * kHop);
}
void launch_tra(hipStream_t s, const float* zt, GtConvW w, float* at, int B, int T, float* state) {
    hipLaunchKernelGGL(k_tra, grid1(B, 16), dim3(256), 0, s, zt, w.gru, w.fc, w.tra_rot, at, B, T, state);
}
void launch_intra_gru(hipStream_t s, View x, const float* gru, float* rnn, int nframes) {
    hipLaunchKernelGGL(k_intra_gru, grid1(nframes, 16), dim3(256), 0, s, x, gru, rnn, nframes);
}
void launch_inter_gru(hipStream_t s, const float* x, const float* gru, float* rnn, int B, int T, float* state, const float* rot) {
    if (rot) hipLaunchKernelGGL(k_inter_gru_rot, grid1((long long)B * kFw, 16), dim3(256), 0, s, x, gru, rot, rnn, B, T, state);
    else hipLaunchKernelGGL(k_inter_gru, grid1((long long)B * kFw, 16), dim3(256), 0, s, x, gru, rnn, B, T, state);
}
void launch_fc_ln_res(hipStream_t s, const float* rnn, View res, const float* fc, const float* fc_b, const float* ln_w,
                      const float* ln_b, float* out, int B, int T) {
    const int tiles = (T + kTileFrames - 1) / kTileFrames;
    hipLaunchKernelGGL(k_fc_ln_res, dim3(B * tiles), dim3(256), 0, s, rnn, res, fc, fc_b, ln_w, ln_b, out, T, tiles);
}
void launch_deconv3(hipStream_t s, View a, View skip, ConvW w, float* d3, int nframes) {
    hipLaunchKernelGGL(k_deconv3, grid1((long long)nframes * kFw, 256), dim3(256), 0, s, a, skip, w.w, w.b, w.slope, d3, nframes);
}
void launch_deconv4(hipStream_t s, const float* d3, const float* e0, ConvW w, float* mask, int nframes) {
    hipLaunchKernelGGL(k_deconv4, grid1((long long)nframes * kF1, 256), dim3(256), 0, s, d3, e0, w.w, w.b, mask, nframes);
}
void launch_istft_masked(hipStream_t s, const float* spec, const float* mask, BandTab erb_bs, FftTabs tabs, float* frames,
                         int nframes) {
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_istft<true>), grid1(nframes, 4), dim3(256), 0, s, spec, mask, erb_bs, tabs, frames, 0,
                       nframes);
}
void launch_istft_ref(hipStream_t s, const float* ref_spec, int B, int T, FftTabs tabs, float* frames) {
    BandTab none = {nullptr, nullptr, 0, 0};
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_istft<false>), grid1((long long)B * T, 4), dim3(256), 0, s, ref_spec, (const float*)nullptr,
                       none, tabs, frames, T, B * T);
}
void launch_ola_pcm(hipStream_t s, const float* frames, FftTabs tabs, int B, int T, int16_t* pcm, float* f32) {
    const long long n = (long long)B * (T - 1) * (kHop / 4);
    hipLaunchKernelGGL(k_ola_pcm, grid1(n, 256), dim3(256), 0, s, frames, tabs.win_sum, T, B, pcm, f32);
}

}  // namespace ade
