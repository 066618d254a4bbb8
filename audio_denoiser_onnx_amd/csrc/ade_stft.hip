// ade_stft.hip — generic STFT_Process operator.
//
// Two formulations of the same transform.  When n_fft factors into 2, 3 and 5 (every starred model: 512, 400, 2048, 1920) the frames go through the
// mixed-radix LDS FFT of csrc/ade_fft.h: two real frames per complex transform, one to eight frame pairs per workgroup (k_stft_fft_analyze /
// k_stft_fft_synth) -- 2048-point analysis is ~0.1 MFLOP per frame pair against 8.4 MFLOP per frame as the dense product.  Any other size keeps the
// reference's own formulation, below, on the matrix cores.
//
// The reference implements STFT / ISTFT for every model as a dense windowed-DFT convolution (STFT_Process.py:213-251,
// 303-336 in each model folder): a (2F, 1, n_fft) Conv1d kernel [cos*w ; -sin*w] with stride = hop, and the transposed
// convolution with [scale*cos*w/N ; -scale*sin*w/N] followed by a trim and a division by sum(w^2).  The transform sizes
// of the starred models are 512, 400, 2048 and 1920 -- only one of them a power of two -- so this operator keeps the
// reference's formulation and runs it as what it is, a GEMM, on v_mfma_f32_16x16x4_f32 (exact fp32, the f32 vector rate
// without occupying the VALU):
//   analysis   spec[b][c][t]  = sum_n K[c][n] * xpad[b][t*hop + n]          M = 2F, N = B*T, K = n_fft
//              (framing, centre padding and reflection are index arithmetic in the B-operand loader: no im2col buffer)
//   synthesis  frame[b,t][n]  = sum_c spec[b][c][t] * Kinv[c][n]            M = B*T, N = n_fft, K = 2F
//              y[b][m]        = sum_t frame[b,t][m + start - t*hop] / sum_t w^2[m + start - t*hop]   (gather: deterministic)
// The GEMM itself is csrc/ade_gemm.h (128 x 128 workgroup tiles, functor operands).
// Tables use exact angles (reduced f*n mod N, evaluated in double); the reference evaluates cos/sin of fp32 angles up
// to 2*pi*N/2, which costs it up to 1e-4 relative -- the parity tests price that difference explicitly.
#include "ade_fft.h"
#include "ade_gemm.h"
#include "ade_internal.h"
#include "../../include/ade.h"

#include <cmath>
#include <cstring>
#include <string>
#include <vector>

namespace ade {
namespace {

using namespace dev;

struct StftDims {
    int n_fft, hop, F2;        // F2 = 2 * (n_fft/2 + 1)
    int pad;                   // n_fft/2 when centre-padded, else 0
    int reflect;               // 1: reflect, 0: zeros (only meaningful with pad > 0)
};

// analysis B operand: B(k, j) = padded sample k of frame j = (b, t); consecutive k are consecutive samples
struct FrameB {
    static constexpr bool kAlongN = false;
    const float* x;
    StftDims d;
    int L, T;
    __device__ float operator()(int k, int j) const {
        const int b = j / T, t = j - b * T;
        int idx = t * d.hop + k - d.pad;
        if (idx < 0) { if (!d.reflect) return 0.0f; idx = -idx; }
        else if (idx >= L) { if (!d.reflect) return 0.0f; idx = 2 * (L - 1) - idx; }
        return x[(size_t)b * L + idx];
    }
};
struct SpecStore {             // C(c, j) -> spec[b][c][t]
    float* spec;
    int F2, T;
    __device__ void operator()(int c, int j, float v) const {
        const int b = j / T, t = j - b * T;
        spec[((size_t)b * F2 + c) * T + t] = v;
    }
};
struct SpecA {                 // synthesis A operand: A(j, c) = spec[b][c][t]; consecutive j are consecutive t
    static constexpr bool kAlongK = false;
    const float* spec;
    int F2, T;
    __device__ float operator()(int j, int c) const {
        const int b = j / T, t = j - b * T;
        return spec[((size_t)b * F2 + c) * T + t];
    }
};

struct PolarSpecA {            // the same operand from polar inputs (istft_A, STFT_Process.py:343-347): real = mag cos(phase), imag = mag sin(phase)
    static constexpr bool kAlongK = false;
    const float *mag, *phase;
    int F, T;
    __device__ float operator()(int j, int c) const {
        const int b = j / T, t = j - b * T;
        const int f = c < F ? c : c - F;
        const size_t at = ((size_t)b * F + f) * T + t;
        const float m = mag[at], ph = phase[at];
        return c < F ? m * cosf(ph) : m * sinf(ph);
    }
};

// overlap-add as a gather (every output sample sums its <= ceil(n_fft/hop) contributing frames in a fixed order), trim,
// divide by the matching sum of squared window samples (static_norm, STFT_Process.py:253-273,326-336)
__global__ __launch_bounds__(256) void k_stft_ola(const float* __restrict__ frames, const float* __restrict__ wsq, float* __restrict__ y,
                                                  int n_fft, int hop, int T, int out_start, int out_len, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int b = (int)(i / out_len), m = (int)(i - (long long)b * out_len) + out_start;
    int t_hi = m / hop;
    if (t_hi > T - 1) t_hi = T - 1;
    const int t_lo = m - n_fft + 1 <= 0 ? 0 : (m - n_fft + hop) / hop;      // smallest t with m - t*hop <= n_fft - 1
    float s = 0.0f, w = 0.0f;
    for (int t = t_lo; t <= t_hi; ++t) {
        const int n = m - t * hop;
        s += frames[((size_t)b * T + t) * n_fft + n];
        w += wsq[n];
    }
    y[i] = s / w;
}


// ---- FFT formulation ----
// Frame pairs: z[n] = w[n] (x_t[n] + i x_{t+1}[n]); Z = FFT(z); X_t[f] = (Z[f] + conj Z[N - f]) / 2, X_{t+1}[f] = (Z[f] - conj Z[N - f]) / (2 i).
// A workgroup of 256 threads holds G pairs of one batch row (256 / G threads each, every group running the same passes so the barriers line up).
__device__ __forceinline__ float padded_sample(const float* __restrict__ row, const StftDims& d, int L, int at) {
    int idx = at - d.pad;
    if (idx < 0) { if (!d.reflect) return 0.0f; idx = -idx; }
    else if (idx >= L) { if (!d.reflect) return 0.0f; idx = 2 * (L - 1) - idx; }
    return row[idx];
}

__global__ __launch_bounds__(256) void k_stft_fft_analyze(const float* __restrict__ x, const float* __restrict__ win, fft::Plan plan, const float2* __restrict__ tw, StftDims d,
                                                          int L, int T, int G, float* __restrict__ spec) {
    HIP_DYNAMIC_SHARED(float2, lds)
    const int N = d.n_fft, F = d.F2 / 2, per = 256 / G, g = threadIdx.x / per, lt = threadIdx.x - g * per;
    const int ppr = (T + 1) / 2, bpr = (ppr + G - 1) / G;
    const int b = blockIdx.x / bpr, pair = (blockIdx.x - b * bpr) * G + g, t0 = 2 * pair;
    const bool live = pair < ppr, two = t0 + 1 < T;
    float2 *A = lds + (size_t)g * 2 * N, *B = A + N;
    const float* row = x + (size_t)b * L;
    for (int n = lt; n < N; n += per) {
        float2 v = make_float2(0.0f, 0.0f);
        if (live) {
            const float w = win[n];
            v.x = padded_sample(row, d, L, t0 * d.hop + n) * w;
            if (two) v.y = padded_sample(row, d, L, (t0 + 1) * d.hop + n) * w;
        }
        A[n] = v;
    }
    const float2* r = fft::forward(A, B, plan, tw, lt, per);
    // the spectrum is (bin, frame) with the frame fastest: the whole workgroup writes its 2 G consecutive frames of one bin together (one 8 G-byte run per bin and part
    // instead of 4-byte stores a row pitch apart); every group's result sits at the same offset of its own pair of buffers
    const size_t roff = (size_t)(r - A);
    const int nt = 2 * G, tb = 2 * ((int)blockIdx.x - b * bpr) * G;
    float* re = spec + (size_t)b * d.F2 * T;
    float* im = re + (size_t)F * T;
    for (int idx = threadIdx.x; idx < F * nt; idx += 256) {
        const int tl = idx & (nt - 1), f = idx / nt, t = tb + tl;
        if (t >= T) continue;
        const float2* rg = lds + (size_t)(tl >> 1) * 2 * N + roff;
        const float2 z = rg[f], zc = rg[f == 0 ? 0 : N - f];
        const bool odd = tl & 1;
        re[(size_t)f * T + t] = odd ? 0.5f * (z.y + zc.y) : 0.5f * (z.x + zc.x);
        im[(size_t)f * T + t] = odd ? 0.5f * (zc.x - z.x) : 0.5f * (z.y - zc.y);
    }
}

// Synthesis: W = H(Z_t) + i H(Z_{t+1}) with H the Hermitian extension of a half spectrum (the imaginary parts of the DC and Nyquist bins do not
// contribute: their sine rows are zero in the reference's inverse table); x_t + i x_{t+1} = conj(FFT(conj W)) / N; frames = x * synthesis window.
template <bool POLAR>
__global__ __launch_bounds__(256) void k_stft_fft_synth(const float* __restrict__ p0, const float* __restrict__ p1, const float* __restrict__ win, fft::Plan plan,
                                                        const float2* __restrict__ tw, StftDims d, int T, int G, float* __restrict__ frames) {
    HIP_DYNAMIC_SHARED(float2, lds)
    const int N = d.n_fft, F = d.F2 / 2, per = 256 / G, g = threadIdx.x / per, lt = threadIdx.x - g * per;
    const int ppr = (T + 1) / 2, bpr = (ppr + G - 1) / G;
    const int b = blockIdx.x / bpr, pair = (blockIdx.x - b * bpr) * G + g, t0 = 2 * pair;
    const bool live = pair < ppr, two = t0 + 1 < T;
    float2 *A = lds + (size_t)g * 2 * N, *B = A + N;
    // load phase by the whole workgroup, pair index fastest: for one bin the G pairs' 2 G frames are adjacent in memory
    const int tb = 2 * ((int)blockIdx.x - b * bpr) * G;
    auto bin = [&](int f, int t) -> float2 {
        float2 v;
        if (POLAR) {                                  // istft_A: real = mag cos(phase), imag = mag sin(phase)   (STFT_Process.py:343-347)
            const size_t at = ((size_t)b * F + f) * T + t;
            const float m = p0[at], ph = p1[at];
            v = make_float2(m * cosf(ph), m * sinf(ph));
        } else {
            v = make_float2(p0[((size_t)b * d.F2 + f) * T + t], p0[((size_t)b * d.F2 + F + f) * T + t]);
        }
        if (f == 0 || (N % 2 == 0 && f == F - 1)) v.y = 0.0f;
        return v;
    };
    for (int idx = threadIdx.x; idx < F * G; idx += 256) {
        const int pl = idx & (G - 1), f = idx / G, t = tb + 2 * pl;
        float2 z0 = make_float2(0.0f, 0.0f), z1 = z0;
        if (t < T) z0 = bin(f, t);
        if (t + 1 < T) z1 = bin(f, t + 1);
        float2* Ag = lds + (size_t)pl * 2 * N;
        // W[f] = z0 + i z1 = (z0.x - z1.y, z0.y + z1.x); W[N - f] = conj z0 + i conj z1 = (z0.x + z1.y, z1.x - z0.y); both stored conjugated
        Ag[f] = make_float2(z0.x - z1.y, -(z0.y + z1.x));
        if (f > 0 && f < N - f) Ag[N - f] = make_float2(z0.x + z1.y, z0.y - z1.x);
    }
    const float2* r = fft::forward(A, B, plan, tw, lt, per);
    if (!live) return;
    const float inv = 1.0f / (float)N;
    float* out = frames + ((size_t)b * T + t0) * N;
    for (int n = lt; n < N; n += per) {
        const float w = win[n];
        out[n] = (r[n].x * inv) * w;
        if (two) out[N + n] = (-r[n].y * inv) * w;
    }
}

// ---- FFT formulation, RUN form (round 6): the sizes ade_fft.h knows at compile time ---------------------------------------------------------------------------------------
// The spectrum's layout is the reference's (B, 2F, T) with the FRAME fastest, and a transform produces every bin of one frame: whoever writes (or reads) it frame pair by frame
// pair moves 8-byte pieces a row pitch apart (the form above: 8 - 32 bytes per bin and workgroup; 6 - 12 % of the HBM rate, profiles/r06_a_bench.json).  Here a 512-thread
// workgroup owns a RUN of RP frame pairs of one batch row and keeps ALL their transforms resident in LDS (RP slots of N + 1 complex values -- the odd pitch spreads the pairs of
// one bin over the banks -- plus one ping-pong buffer per thread group): groups of 128 / 512 threads transform one pair each, pass by pass in step, and the spectrum is then
// written (analysis) or was read (synthesis) bin by bin with the run's 2 RP frames adjacent: 64 - 256 contiguous bytes per bin, the whole (2F, T) block of a row in one piece
// when the run is the row (GTCRN's 63 frames).  The synthesis also keeps the overlap-add in the workgroup: it transforms `halo` = ceil(N / hop) - 1 frames before its own
// (recomputed, not exchanged), windows and sums each output sample's frames out of the resident slots in ascending frame order -- the sums, the order and therefore the bits of
// the two-kernel form above (frames through HBM, then k_stft_ola), without the B x T x N float round trip.
constexpr int kRunThreads = 512;
// threads that share one transform: ONE wavefront up to 512 points (round 6: the transform then runs in place in its slot with no workgroup barrier and no ping-pong buffer,
// fft::forward_wave -- eight pairs in flight per workgroup, each at its own pace), otherwise a quarter of the points
constexpr int run_gt(int n) { return n <= 512 ? 64 : (n / 4 <= 256 ? 256 : 512); }
template <int N> constexpr int run_group_threads() { return run_gt(N); }

template <int N>
__global__ __launch_bounds__(kRunThreads) void k_stft_run_analyze(const float* __restrict__ x, const float* __restrict__ win, const float2* __restrict__ tw, StftDims d, int L,
                                                                   int T, int RP, int lg_nt, float* __restrict__ spec) {
    HIP_DYNAMIC_SHARED(float2, lds)
    constexpr int GT = run_group_threads<N>(), G = kRunThreads / GT, SP = N + 1, F = N / 2 + 1;
    const int tid = threadIdx.x, grp = tid / GT, lt = tid - grp * GT;
    const int ppr = (T + 1) / 2, wpr = (ppr + RP - 1) / RP;
    const int b = (int)blockIdx.x / wpr, p0 = ((int)blockIdx.x - b * wpr) * RP;
    constexpr bool kWave = GT == 64;                                            // one wavefront per transform: in place, no ping-pong buffers
    float2* const scratch = lds + (size_t)RP * SP + (size_t)grp * N;
    float2* const twl = lds + (size_t)RP * SP + (kWave ? 0 : (size_t)G * N);  // the twiddle table in LDS (three scattered 8-byte reads per butterfly and pass: through the L1 they cost more than the butterflies)
    for (int n = tid; n < N; n += kRunThreads) twl[n] = tw[n];              // (visible behind the barrier that precedes the first pass; pass 0 reads no twiddles)
    if (kWave) __syncthreads();
    const float* const row = x + (size_t)b * L;
    for (int pl = grp; pl < RP; pl += G) {
        const int pair = p0 + pl, t0 = 2 * pair;
        const bool live = pair < ppr, two = t0 + 1 < T;
        float2* const slot = lds + (size_t)pl * SP;
        float2* const A = kWave || fft::result_in_first<N>() ? slot : scratch;          // the transform's result lands in the slot
        float2* const Bf = fft::result_in_first<N>() ? scratch : slot;
        for (int n = lt; n < N; n += GT) {
            float2 v = make_float2(0.0f, 0.0f);
            if (live) {
                const float w = win[n];
                v.x = padded_sample(row, d, L, t0 * d.hop + n) * w;
                if (two) v.y = padded_sample(row, d, L, (t0 + 1) * d.hop + n) * w;
            }
            A[n] = v;
        }
        if constexpr (kWave) {
            wave_sync();
            fft::forward_wave<N>(slot, twl, lt);
        } else {
            fft::forward_static<N>(A, Bf, twl, lt, GT);
            __syncthreads();                           // (the next pair's samples go into a buffer this pair's last pass may still be reading)
        }
    }
    if (kWave) __syncthreads();
    const int nt = 2 * RP, tb = 2 * p0;
    float* const re = spec + (size_t)b * (2 * F) * T;
    float* const im = re + (size_t)F * T;
#pragma unroll 4
    for (int idx = tid; idx < F * nt; idx += kRunThreads) {
        const int tl = idx & (nt - 1), f = idx >> lg_nt, t = tb + tl;
        if (t >= T) continue;
        const float2* rg = lds + (size_t)(tl >> 1) * SP;
        const float2 z = rg[f], zc = rg[f == 0 ? 0 : N - f];
        const bool odd = tl & 1;
        re[(size_t)f * T + t] = odd ? 0.5f * (z.y + zc.y) : 0.5f * (z.x + zc.x);
        im[(size_t)f * T + t] = odd ? 0.5f * (zc.x - z.x) : 0.5f * (z.y - zc.y);
    }
}

// first owned frame of workgroup w = w * RO, RO = 2 RP - halo; the frames ts = w RO - halo .. ts + 2 RP - 1 are transformed (those outside [0, T) as zeros, never read back)
template <int N, bool POLAR, bool TWL>      // TWL: the twiddle table is copied into LDS (when the slots leave room for it)
__global__ __launch_bounds__(kRunThreads) void k_stft_run_synth(const float* __restrict__ p0, const float* __restrict__ p1, const float* __restrict__ win,
                                                                 const float* __restrict__ wsq, const float2* __restrict__ tw, StftDims d, int T, int RP, int lg_rp, int halo,
                                                                 int out_start, int out_len, float* __restrict__ y) {
    HIP_DYNAMIC_SHARED(float2, lds)
    constexpr int GT = run_group_threads<N>(), G = kRunThreads / GT, SP = N + 1, F = N / 2 + 1;
    const int tid = threadIdx.x, grp = tid / GT, lt = tid - grp * GT;
    const int RT = 2 * RP, RO = RT - halo, wpr = (T + RO - 1) / RO;
    const int b = (int)blockIdx.x / wpr, wi = (int)blockIdx.x - b * wpr, to = wi * RO, ts = to - halo;
    constexpr bool kWave = GT == 64;
    float2* const scratch = lds + (size_t)RP * SP + (size_t)grp * N;
    const float2* twl = tw;
    if (TWL) {
        float2* const t2 = lds + (size_t)RP * SP + (kWave ? 0 : (size_t)G * N);
        for (int n = tid; n < N; n += kRunThreads) t2[n] = tw[n];
        twl = t2;
    }
    auto bin = [&](int f, int t) -> float2 {
        float2 v = make_float2(0.0f, 0.0f);
        if (t < 0 || t >= T) return v;
        if (POLAR) {                                  // istft_A: real = mag cos(phase), imag = mag sin(phase)   (STFT_Process.py:343-347)
            const size_t at = ((size_t)b * F + f) * T + t;
            const float m = p0[at], ph = p1[at];
            v = make_float2(m * cosf(ph), m * sinf(ph));
        } else {
            v = make_float2(p0[((size_t)b * d.F2 + f) * T + t], p0[((size_t)b * d.F2 + F + f) * T + t]);
        }
        if (f == 0 || (N % 2 == 0 && f == F - 1)) v.y = 0.0f;
        return v;
    };
#pragma unroll 2
    for (int idx = tid; idx < F * RP; idx += kRunThreads) {
        const int pl = idx & (RP - 1), f = idx >> lg_rp, t = ts + 2 * pl;
        const float2 z0 = bin(f, t), z1 = bin(f, t + 1);
        float2* Ag = lds + (size_t)pl * SP;
        // W[f] = z0 + i z1 = (z0.x - z1.y, z0.y + z1.x); W[N - f] = conj z0 + i conj z1 = (z0.x + z1.y, z1.x - z0.y); both stored conjugated
        Ag[f] = make_float2(z0.x - z1.y, -(z0.y + z1.x));
        if (f > 0 && f < N - f) Ag[N - f] = make_float2(z0.x + z1.y, z0.y - z1.x);
    }
    if (kWave) __syncthreads();                        // (the slots are filled by the whole workgroup, the twiddles copied)
    for (int pl = grp; pl < RP; pl += G) {
        float2* const slot = lds + (size_t)pl * SP;
        if constexpr (kWave) {
            fft::forward_wave<N>(slot, twl, lt);
        } else {
            fft::forward_static<N>(slot, scratch, twl, lt, GT);
            if (!fft::result_in_first<N>()) {          // an odd number of passes leaves the result in the ping-pong buffer: back into the slot
                __syncthreads();
                for (int n = lt; n < N; n += GT) slot[n] = scratch[n];
            }
            __syncthreads();
        }
    }
    if (kWave) __syncthreads();
    // overlap-add out of the slots: the samples whose LAST contributing frame is one of this workgroup's own (raw index m = t hop + n before the trim); the last workgroup
    // of a row also takes the tail
    const int raw_len = N + d.hop * (T - 1);
    const int m0 = to * d.hop, m1 = wi + 1 == wpr ? raw_len : (to + RO) * d.hop;
    const float inv = 1.0f / (float)N;
    float* const yr = y + (size_t)b * out_len;
    for (int m = m0 + tid; m < m1; m += kRunThreads) {
        if (m < out_start || m >= out_start + out_len) continue;
        int t_hi = m / d.hop;
        if (t_hi > T - 1) t_hi = T - 1;
        const int t_lo = m - N + 1 <= 0 ? 0 : (m - N + d.hop) / d.hop;      // smallest t with m - t*hop <= n_fft - 1
        float sacc = 0.0f, wacc = 0.0f;
        for (int t = t_lo; t <= t_hi; ++t) {
            const int n = m - t * d.hop, tl = t - ts;
            const float2 r = lds[(size_t)(tl >> 1) * SP + n];
            const float v = (tl & 1) ? -r.y : r.x;
            sacc += __fmul_rn(v * inv, win[n]);               // (the windowed sample is rounded before it is added, as the frame buffer of the two-kernel form held it)
            wacc += wsq[n];
        }
        yr[m - out_start] = sacc / wacc;
    }
}

// torch.{hann,hamming}_window in fp32 (STFT_Process.py:88-113 registries), centre pad / crop to n_fft
bool make_window(const std::string& name_in, int win_length, int n_fft, std::vector<float>& w, std::string& err) {
    std::string name = name_in;
    bool sym = false;
    if (name.size() > 4 && name.compare(name.size() - 4, 4, "_sym") == 0) { sym = true; name.resize(name.size() - 4); }
    if (name == "hamming_periodic") name = "hamming";
    float alpha, beta;
    bool root = false;
    if (name == "hann") { alpha = 0.5f; beta = 0.5f; }
    else if (name == "hann_sqrt") { alpha = 0.5f; beta = 0.5f; root = true; }
    else if (name == "hamming") { alpha = 0.54f; beta = 0.46f; }
    else { err = "unsupported window type: " + name_in; return false; }
    std::vector<float> raw((size_t)win_length);
    const float step = (float)(2.0 * M_PI / (double)(sym ? win_length - 1 : win_length));
    for (int n = 0; n < win_length; ++n) {
        const float v = cosf((float)n * step) * (-beta) + alpha;
        raw[n] = root ? sqrtf(v) : v;
    }
    w.assign((size_t)n_fft, 0.0f);
    if (win_length <= n_fft) {
        const int left = (n_fft - win_length) / 2;
        for (int n = 0; n < win_length; ++n) w[left + n] = raw[n];
    } else {
        const int start = (win_length - n_fft) / 2;
        for (int n = 0; n < n_fft; ++n) w[n] = raw[start + n];
    }
    return true;
}

}  // namespace
}  // namespace ade

namespace ade {
namespace {
#define ADE_RUN_SIZES(M) M(512) M(400) M(2048) M(1920) M(64) M(60)
bool run_raise_lds(int n) {
    const int bytes = 160 * 1024;
    bool ok = true;
#define ADE_RAISE(NN)                                                                                                                                          \
    if (n == NN) ok = hipFuncSetAttribute(reinterpret_cast<const void*>(k_stft_run_analyze<NN>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess &&   \
                      hipFuncSetAttribute(reinterpret_cast<const void*>(k_stft_run_synth<NN, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess && \
                      hipFuncSetAttribute(reinterpret_cast<const void*>(k_stft_run_synth<NN, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess && \
                      hipFuncSetAttribute(reinterpret_cast<const void*>(k_stft_run_synth<NN, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess && \
                      hipFuncSetAttribute(reinterpret_cast<const void*>(k_stft_run_synth<NN, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess;
    ADE_RUN_SIZES(ADE_RAISE)
#undef ADE_RAISE
    return ok;
}
inline int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }
// pairs per workgroup for a call with `pairs` frame pairs per row: the largest power of two that the LDS holds and the row can use (never below the thread groups' count)
inline int run_pairs_for(int n, int rp_max, int pairs) {
    const int gt = run_gt(n), groups = kRunThreads / gt;
    int rp = rp_max;
    while (rp / 2 >= pairs && rp / 2 >= groups) rp /= 2;
    return rp;
}
inline size_t run_lds_bytes(int n, int rp, bool twl = true) {
    const int gt = run_gt(n), groups = kRunThreads / gt;
    return ((size_t)rp * (n + 1) + (gt == 64 ? 0 : (size_t)groups * n) + (twl ? (size_t)n : 0)) * sizeof(float2);        // slots + ping-pong buffers (none for one-wavefront transforms) + the twiddle table
}
constexpr size_t kRunLdsBudget = 150 * 1024;
void run_analyze(hipStream_t s, int n, const float* x, const float* win, const float2* tw, StftDims d, int L, int T, int rp, int batch, float* spec) {
    const int ppr = (T + 1) / 2, wpr = (ppr + rp - 1) / rp;
#define ADE_LAUNCH(NN) if (n == NN) hipLaunchKernelGGL(k_stft_run_analyze<NN>, dim3((unsigned)(batch * wpr)), dim3(kRunThreads), run_lds_bytes(n, rp), s, x, win, tw, d, L, T, rp, ilog2(2 * rp), spec);
    ADE_RUN_SIZES(ADE_LAUNCH)
#undef ADE_LAUNCH
}
template <bool POLAR>
void run_synth(hipStream_t s, int n, const float* p0, const float* p1, const float* win, const float* wsq, const float2* tw, StftDims d, int T, int rp, int batch, int out_start,
               int out_len, float* y) {
    const int halo = (n + d.hop - 1) / d.hop - 1, ro = 2 * rp - halo, wpr = (T + ro - 1) / ro;
    const bool twl = run_lds_bytes(n, rp, true) <= kRunLdsBudget;
#define ADE_LAUNCH(NN)                                                                                                                                                       \
    if (n == NN) {                                                                                                                                                           \
        if (twl) hipLaunchKernelGGL((k_stft_run_synth<NN, POLAR, true>), dim3((unsigned)(batch * wpr)), dim3(kRunThreads), run_lds_bytes(n, rp, true), s, p0, p1, win, wsq, tw, d, T, rp, ilog2(rp), halo, out_start, out_len, y); \
        else hipLaunchKernelGGL((k_stft_run_synth<NN, POLAR, false>), dim3((unsigned)(batch * wpr)), dim3(kRunThreads), run_lds_bytes(n, rp, false), s, p0, p1, win, wsq, tw, d, T, rp, ilog2(rp), halo, out_start, out_len, y); \
    }
    ADE_RUN_SIZES(ADE_LAUNCH)
#undef ADE_LAUNCH
}
#undef ADE_RUN_SIZES
}  // namespace
}  // namespace ade

struct ade_stft_plan {
    int device = 0;
    ade::StftDims d{};
    int center = 1;
    int keep_tail = 0;                            // the dynamic-length trim: everything after the leading half window (ade_stft_keep_tail)
    float *d_fwd = nullptr, *d_inv = nullptr, *d_wsq = nullptr, *d_frames = nullptr;      // dense tables: only when n_fft has a prime factor above 5
    bool use_fft = false;
    ade::fft::Plan fft_plan{};
    float2* d_tw = nullptr;                       // exp(-2 pi i m / n_fft), m < n_fft
    float *d_wa = nullptr, *d_ws = nullptr;       // analysis / synthesis windows
    int pairs_per_group = 1;                      // frame pairs per workgroup
    bool run_form = false;                        // n_fft is one of ade_fft.h's compile-time sizes: the run kernels (k_stft_run_*)
    int run_pairs_analyze = 1, run_pairs_synth = 1;   // frame pairs per workgroup of the run kernels (powers of two)
    size_t frames_cap = 0;
    hipStream_t stream = nullptr;
    std::string last_error;
};

namespace {
thread_local std::string g_stft_create_error;
ade_status sfail(ade_stft_plan* p, ade_status st, const std::string& msg) {
    if (p) p->last_error = msg; else g_stft_create_error = msg;
    return st;
}
#define STFT_HIP(p, expr)                                                                                      \
    do {                                                                                                       \
        hipError_t _e = (expr);                                                                                \
        if (_e != hipSuccess) return sfail((p), ADE_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)
}  // namespace

extern "C" {

ade_status ade_stft_create(const ade_stft_config* cfg, int device, ade_stft_handle* out) {
    if (!out) return sfail(nullptr, ADE_ERR_BAD_VALUE, "ade_stft_create: out is NULL");
    *out = nullptr;
    if (!cfg || !cfg->window) return sfail(nullptr, ADE_ERR_BAD_VALUE, "ade_stft_create: config / window is NULL");
    if (cfg->n_fft < 4 || cfg->n_fft > 8192 || cfg->hop < 1 || cfg->hop > cfg->n_fft || cfg->win_length < 1)
        return sfail(nullptr, ADE_ERR_BAD_VALUE, "ade_stft_create: need 4 <= n_fft <= 8192, 1 <= hop <= n_fft, win_length >= 1");
    const std::string pad = cfg->pad_mode ? cfg->pad_mode : "reflect";
    if (pad != "reflect" && pad != "constant") return sfail(nullptr, ADE_ERR_UNSUPPORTED, "pad_mode must be 'reflect' or 'constant'");
    std::vector<float> wa, ws;
    std::string err;
    if (!ade::make_window(cfg->window, cfg->win_length, cfg->n_fft, wa, err)) return sfail(nullptr, ADE_ERR_UNSUPPORTED, err);
    if (!ade::make_window(cfg->synthesis_window ? cfg->synthesis_window : cfg->window, cfg->win_length, cfg->n_fft, ws, err))
        return sfail(nullptr, ADE_ERR_UNSUPPORTED, err);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        (void)hipGetLastError();
        return sfail(nullptr, ADE_ERR_DEVICE, "no HIP device visible: libade has no CPU execution mode");
    }
    if (device < 0 || device >= ndev) return sfail(nullptr, ADE_ERR_DEVICE, "device ordinal out of range");
    ade_stft_plan* p = new ade_stft_plan();
    p->device = device;
    const int N = cfg->n_fft, F = N / 2 + 1;
    p->d = ade::StftDims{N, cfg->hop, 2 * F, cfg->center_pad ? N / 2 : 0, pad == "reflect" ? 1 : 0};
    p->center = cfg->center_pad ? 1 : 0;
    std::vector<float> wsq((size_t)N);
    for (int n = 0; n < N; ++n) wsq[n] = ws[n] * ws[n];
    auto bail = [&](ade_status st) { g_stft_create_error = p->last_error; ade_stft_destroy(p); return st; };
    if (hipSetDevice(device) != hipSuccess) return bail(sfail(p, ADE_ERR_DEVICE, "hipSetDevice failed"));
    if (hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking) != hipSuccess) return bail(sfail(p, ADE_ERR_DEVICE, "hipStreamCreate failed"));
    if (hipMalloc((void**)&p->d_wsq, wsq.size() * sizeof(float)) != hipSuccess ||
        hipMemcpy(p->d_wsq, wsq.data(), wsq.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
        return bail(sfail(p, ADE_ERR_DEVICE, "upload of the window table failed"));
    p->use_fft = ade::fft::make_plan(N, &p->fft_plan);
    if (p->use_fft) {
        // twiddles with exact angles (the same values the dense tables below hold); 16 N bytes of LDS per frame pair
        std::vector<float2> tw((size_t)N);
        for (int m = 0; m < N; ++m) {
            const double a = 2.0 * M_PI * (double)m / (double)N;
            tw[m] = make_float2((float)cos(a), (float)-sin(a));
        }
        p->pairs_per_group = 1;
        while (p->pairs_per_group < 8 && (size_t)(2 * p->pairs_per_group) * N <= 4096) p->pairs_per_group *= 2;
        // run form (k_stft_run_*): the compile-time sizes, unless ADE_STFT_RUN=0; RP slots of N + 1 complex values + one ping-pong buffer per thread group in <= 150 KB of LDS
        p->run_form = ade::fft::static_size(N) && !(getenv("ADE_STFT_RUN") && atoi(getenv("ADE_STFT_RUN")) == 0);
        if (p->run_form) {
            // Run lengths (powers of two, at least one pair per thread group).  Measured on one MI355X (profiles/r06_k_stft_run_pairs_kernel_us.txt): the kernels are bound by
            // how many workgroups a CU holds, not by the length of the contiguous pieces -- 512-point analysis 48 us with the whole row (32 pairs, 148 KB of LDS, one workgroup
            // per CU) and 32 us with 4 pairs (33 KB, four per CU).  So: the analysis takes the shortest run; the synthesis the shortest one whose halo (frames transformed for
            // the overlap-add and not owned) stays below a third of it.
            const int gt = ade::run_gt(N), groups = ade::kRunThreads / gt;
            const size_t budget = ade::kRunLdsBudget, fixed = gt == 64 ? 0 : (size_t)groups * N * 8;        // (the synthesis leaves its twiddles in global memory when slots + table do not fit)
            const int halo = (N + cfg->hop - 1) / cfg->hop - 1;
            int rpa = groups > 2 ? groups : 2, rps = groups;
            while (2 * rps < 4 * halo) rps *= 2;
            const int cap = getenv("ADE_STFT_RUN_PAIRS") ? atoi(getenv("ADE_STFT_RUN_PAIRS")) : 0;           // measurement knob: this many pairs per workgroup in both directions
            if (cap > 0) { rpa = cap > groups ? cap : groups; rps = rpa; }
            if ((size_t)rpa * (N + 1) * 8 + fixed + (size_t)N * 8 > budget || (size_t)rps * (N + 1) * 8 + fixed > budget || 2 * rps <= halo) p->run_form = false;      // (hops too small for a run)
            p->run_pairs_analyze = rpa;
            p->run_pairs_synth = rps;
        }
        if (p->run_form && !ade::run_raise_lds(N)) return bail(sfail(p, ADE_ERR_DEVICE, "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed for the run-form FFT kernels"));
        const int lds_max = 16 * 8192;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(ade::k_stft_fft_analyze), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(ade::k_stft_fft_synth<false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(ade::k_stft_fft_synth<true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max) != hipSuccess)
            return bail(sfail(p, ADE_ERR_DEVICE, "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed for the FFT kernels"));
        if (hipMalloc((void**)&p->d_tw, tw.size() * sizeof(float2)) != hipSuccess || hipMalloc((void**)&p->d_wa, (size_t)N * sizeof(float)) != hipSuccess ||
            hipMalloc((void**)&p->d_ws, (size_t)N * sizeof(float)) != hipSuccess)
            return bail(sfail(p, ADE_ERR_DEVICE, "hipMalloc of the FFT tables failed"));
        if (hipMemcpy(p->d_tw, tw.data(), tw.size() * sizeof(float2), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(p->d_wa, wa.data(), (size_t)N * sizeof(float), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(p->d_ws, ws.data(), (size_t)N * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
            return bail(sfail(p, ADE_ERR_DEVICE, "upload of the FFT tables failed"));
        *out = p;
        return ADE_OK;
    }
    // dense tables (STFT_Process.py:213-251), exact angles
    std::vector<float> fwd((size_t)2 * F * N), inv((size_t)2 * F * N);
    for (int f = 0; f < F; ++f) {
        const double scale = (f == 0 || (N % 2 == 0 && f == F - 1)) ? 1.0 : 2.0;
        for (int n = 0; n < N; ++n) {
            const double a = 2.0 * M_PI * (double)(((long long)f * n) % N) / (double)N;
            const float c = (float)cos(a), s = (float)sin(a);
            fwd[(size_t)f * N + n] = c * wa[n];
            fwd[(size_t)(F + f) * N + n] = -s * wa[n];
            inv[(size_t)f * N + n] = (((float)scale * c) * (float)(1.0 / N)) * ws[n];
            inv[(size_t)(F + f) * N + n] = (((float)scale * -s) * (float)(1.0 / N)) * ws[n];
        }
    }
    const size_t tb = fwd.size() * sizeof(float);
    if (hipMalloc((void**)&p->d_fwd, tb) != hipSuccess || hipMalloc((void**)&p->d_inv, tb) != hipSuccess)
        return bail(sfail(p, ADE_ERR_DEVICE, "hipMalloc of the DFT tables failed"));
    if (hipMemcpy(p->d_fwd, fwd.data(), tb, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(p->d_inv, inv.data(), tb, hipMemcpyHostToDevice) != hipSuccess)
        return bail(sfail(p, ADE_ERR_DEVICE, "upload of the DFT tables failed"));
    *out = p;
    return ADE_OK;
}

ade_status ade_stft_frames(ade_stft_handle p, int length, int* frames) {
    if (!p || !frames) return ADE_ERR_BAD_VALUE;
    const int Lp = length + 2 * p->d.pad;
    if (length < 1 || Lp < p->d.n_fft || (p->d.pad && p->d.reflect && length <= p->d.pad))
        return sfail(p, ADE_ERR_SHAPE_MISMATCH, "input shorter than one frame (or than the reflection pad)");
    *frames = (Lp - p->d.n_fft) / p->d.hop + 1;
    return ADE_OK;
}

ade_status ade_stft_output_length(ade_stft_handle p, int frames, int* out_len) {
    if (!p || !out_len || frames < 1) return ADE_ERR_BAD_VALUE;
    const int raw = p->d.n_fft + p->d.hop * (frames - 1);
    *out_len = p->center ? raw - (p->keep_tail ? p->d.n_fft / 2 : p->d.n_fft) : raw;
    return ADE_OK;
}

ade_status ade_stft_keep_tail(ade_stft_handle p, int keep_tail) {
    if (!p) return ADE_ERR_BAD_VALUE;
    p->keep_tail = keep_tail ? 1 : 0;
    return ADE_OK;
}

ade_status ade_stft_analyze(ade_stft_handle p, const float* d_x, int batch, int length, float* d_spec, void* hip_stream) {
    if (!p || batch < 0 || (batch > 0 && (!d_x || !d_spec))) return sfail(p, ADE_ERR_BAD_VALUE, "ade_stft_analyze: bad arguments");
    if (batch == 0) return ADE_OK;
    int T = 0;
    ade_status st = ade_stft_frames(p, length, &T);
    if (st != ADE_OK) return st;
    STFT_HIP(p, hipSetDevice(p->device));
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : p->stream;
    if (p->use_fft && p->run_form) {
        ade::run_analyze(s, p->d.n_fft, d_x, p->d_wa, p->d_tw, p->d, length, T, ade::run_pairs_for(p->d.n_fft, p->run_pairs_analyze, (T + 1) / 2), batch, d_spec);
    } else if (p->use_fft) {
        const int G = p->pairs_per_group, groups = ((T + 1) / 2 + G - 1) / G;
        hipLaunchKernelGGL(ade::k_stft_fft_analyze, dim3((unsigned)(batch * groups)), dim3(256), (size_t)G * 2 * p->d.n_fft * sizeof(float2), s, d_x, (const float*)p->d_wa,
                           p->fft_plan, (const float2*)p->d_tw, p->d, length, T, G, d_spec);
    } else {
        ade::gemm::launch(s, ade::gemm::RowMajorA{p->d_fwd, p->d.n_fft}, ade::FrameB{d_x, p->d, length, T}, ade::SpecStore{d_spec, p->d.F2, T},
                          p->d.F2, batch * T, p->d.n_fft);
    }
    STFT_HIP(p, hipGetLastError());
    if (!hip_stream) STFT_HIP(p, hipStreamSynchronize(s));
    return ADE_OK;
}

}  // extern "C"

namespace {
template <bool POLAR, class ALoader>
ade_status synthesize_from(ade_stft_handle p, ALoader a, const float* p0, const float* p1, int batch, int frames, float* d_y, void* hip_stream) {
    STFT_HIP(p, hipSetDevice(p->device));
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : p->stream;
    if (p->use_fft && p->run_form) {                 // transforms + overlap-add in one launch: no frame buffer
        int out_len = 0;
        (void)ade_stft_output_length(p, frames, &out_len);
        const int halo = (p->d.n_fft + p->d.hop - 1) / p->d.hop - 1;
        const int rp = ade::run_pairs_for(p->d.n_fft, p->run_pairs_synth, (frames + halo + 1) / 2);
        ade::run_synth<POLAR>(s, p->d.n_fft, p0, p1, p->d_ws, p->d_wsq, p->d_tw, p->d, frames, rp, batch, p->center ? p->d.n_fft / 2 : 0, out_len, d_y);
        STFT_HIP(p, hipGetLastError());
        if (!hip_stream) STFT_HIP(p, hipStreamSynchronize(s));
        return ADE_OK;
    }
    const size_t need = (size_t)batch * frames * p->d.n_fft;
    if (need > p->frames_cap) {
        STFT_HIP(p, hipStreamSynchronize(s));
        if (p->d_frames) (void)hipFree(p->d_frames);
        p->d_frames = nullptr;
        p->frames_cap = 0;
        STFT_HIP(p, hipMalloc((void**)&p->d_frames, need * sizeof(float)));
        p->frames_cap = need;
    }
    if (p->use_fft) {
        const int G = p->pairs_per_group, groups = ((frames + 1) / 2 + G - 1) / G;
        hipLaunchKernelGGL(ade::k_stft_fft_synth<POLAR>, dim3((unsigned)(batch * groups)), dim3(256), (size_t)G * 2 * p->d.n_fft * sizeof(float2), s, p0, p1,
                           (const float*)p->d_ws, p->fft_plan, (const float2*)p->d_tw, p->d, frames, G, p->d_frames);
    } else {
        ade::gemm::launch(s, a, ade::gemm::RowMajorB{p->d_inv, p->d.n_fft}, ade::gemm::BiasActStore<ade::gemm::kActNone>{p->d_frames, p->d.n_fft, nullptr, 0.0f},
                          batch * frames, p->d.n_fft, p->d.F2);
    }
    int out_len = 0;
    (void)ade_stft_output_length(p, frames, &out_len);
    const long long total = (long long)batch * out_len;
    hipLaunchKernelGGL(ade::k_stft_ola, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const float*)p->d_frames, (const float*)p->d_wsq,
                       d_y, p->d.n_fft, p->d.hop, frames, p->center ? p->d.n_fft / 2 : 0, out_len, total);
    STFT_HIP(p, hipGetLastError());
    if (!hip_stream) STFT_HIP(p, hipStreamSynchronize(s));
    return ADE_OK;
}
}  // namespace

extern "C" {

ade_status ade_stft_synthesize(ade_stft_handle p, const float* d_spec, int batch, int frames, float* d_y, void* hip_stream) {
    if (!p || batch < 0 || frames < 1 || (batch > 0 && (!d_spec || !d_y))) return sfail(p, ADE_ERR_BAD_VALUE, "ade_stft_synthesize: bad arguments");
    if (batch == 0) return ADE_OK;
    return synthesize_from<false>(p, ade::SpecA{d_spec, p->d.F2, frames}, d_spec, nullptr, batch, frames, d_y, hip_stream);
}

ade_status ade_stft_synthesize_polar(ade_stft_handle p, const float* d_mag, const float* d_phase, int batch, int frames, float* d_y, void* hip_stream) {
    if (!p || batch < 0 || frames < 1 || (batch > 0 && (!d_mag || !d_phase || !d_y))) return sfail(p, ADE_ERR_BAD_VALUE, "ade_stft_synthesize_polar: bad arguments");
    if (batch == 0) return ADE_OK;
    return synthesize_from<true>(p, ade::PolarSpecA{d_mag, d_phase, p->d.F2 / 2, frames}, d_mag, d_phase, batch, frames, d_y, hip_stream);
}

const char* ade_stft_last_error(ade_stft_handle p) { return p ? p->last_error.c_str() : g_stft_create_error.c_str(); }

void ade_stft_destroy(ade_stft_handle p) {
    if (!p) return;
    (void)hipSetDevice(p->device);
    if (p->d_fwd) (void)hipFree(p->d_fwd);
    if (p->d_inv) (void)hipFree(p->d_inv);
    if (p->d_wsq) (void)hipFree(p->d_wsq);
    if (p->d_frames) (void)hipFree(p->d_frames);
    if (p->d_tw) (void)hipFree(p->d_tw);
    if (p->d_wa) (void)hipFree(p->d_wa);
    if (p->d_ws) (void)hipFree(p->d_ws);
    if (p->stream) (void)hipStreamDestroy(p->stream);
    delete p;
}

}  // extern "C"
