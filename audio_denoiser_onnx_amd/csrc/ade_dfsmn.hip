// ade_dfsmn.hip — DFSMN (48 kHz acoustic noise suppression) on the MI355X: SURVEY.md section 8 row a19.
//
// Reference: DFSMN.forward, DFSMN/Export_DFSMN.py:180-246, and the buffers its constructor builds (:86-178):
//   int16 -> * 2^-15 -> ONE strided analysis convolution [Kaldi fbank (DC removal + 0.97 pre-emphasis + symmetric hamming
//   + 2048-pt DFT, folded into a (2050, 1920) matrix) | 1920-pt mask STFT] -> power -> mel(120) -> log ->
//   Linear(120,256)+ReLU -> 9 x [Linear+ReLU, Linear (no bias), causal depthwise memory (lorder 20) with the inner
//   residual folded into the current tap, outer residual] -> Linear(256,961)+Sigmoid -> mask * spectrum ->
//   ISTFT (periodic hamming, no centre pad, static COLA) -> * 32768, clamp, truncate -> int16.
// The two analysis transforms and the synthesis are FFTs (csrc/ade_fft.h: one workgroup per frame, mixed-radix Stockham passes in LDS -- 2048 = 4^5 x 2 for the
// Kaldi filter bank, 1920 = 4^3 x 2 x 3 x 5 for the mask STFT and its inverse): 0.3 MFLOP per frame instead of the 22.9 MFLOP of the reference's dense
// (3972 + 1922) x 1920 windowed-DFT products, which used to be 67 % of this model's step as matrix-core GEMMs.  The steps the reference folds into its
// analysis matrix -- DC removal, 0.97 pre-emphasis, symmetric hamming (:97-120) -- are applied to the frame explicitly before the 2048-point transform.
// The mask network is matrix products and stays on the generic matrix-core GEMM of csrc/ade_gemm.h (activations channels-first (C, N), N = batch * frames):
//   analysis   P(N, 1025) = |FFT_2048(hamming * preemph(frame - mean))|^2 * 2^30 ;  S(N, 2 x 961) = FFT_1920(hamming * frame)         k_dfsmn_analysis
//   log-mel    F(120, N)  = mel(120, 1025) x P^T                         store = log(max(., eps))
//   layers     X(256, N)  = relu(W x + b) ...                            store = bias + ReLU / none ; the last one bias + sigmoid, transposed to (N, 961)
//   synthesis  frames(N, 1920) = hamming_periodic * Re IFFT_1920(hermitian(mask * S))                                                   k_dfsmn_synthesis
// followed by a gather overlap-add with the PCM tail fused.  Twiddles are exact (double-precision angles), like the dense tables they replace.
#include "ade_fft.h"
#include "ade_gemm.h"
#include "ade_internal.h"
#include "../../include/ade.h"

#include <cmath>
#include <cstring>

namespace ade {

namespace {

using namespace dev;

constexpr int kKaldiNfft = 2048, kFrame = 1920, kHopD = 960, kNfftD = 1920;
constexpr int kFbBins = kKaldiNfft / 2 + 1;      // 1025
constexpr int kStBins = kNfftD / 2 + 1;          // 961
constexpr int kMel = 120, kHid = 256;

struct PowerFrameB {           // B(f, j) = power bin f of frame j (frame-major P, written by k_dfsmn_analysis)   (:216)
    static constexpr bool kAlongN = false;
    const float* p;
    __device__ float operator()(int f, int j) const { return p[(size_t)j * kFbBins + f]; }
};
struct SigmoidTStore {         // mask[j][f] = sigmoid(v + bias[f]): the last Linear, stored frame-major for the per-frame synthesis kernel   (:230)
    static constexpr bool kCtx = true;
    float* out;
    const float* bias;
    __device__ float row(int f) const { return bias[f]; }
    __device__ gemm::None col(int) const { return gemm::None{}; }
    __device__ gemm::None pre(int, int, float) const { return gemm::None{}; }
    __device__ void operator()(int f, int j, float v, float b, gemm::None, gemm::None) const { out[(size_t)j * kStBins + f] = 1.0f / (1.0f + expf(-(v + b))); }
};

// One workgroup per PAIR of frames: [Kaldi filter-bank power | mask STFT] of each frame's 1920 samples.  Both transforms take real input, so the two frames ride one
// complex FFT as real and imaginary part: Z = FFT(x0 + i x1), X0[f] = (Z[f] + conj Z[N - f]) / 2, X1[f] = (Z[f] - conj Z[N - f]) / (2 i) -- half the butterflies.
//   x = samples * 2^-15 (:186-190); fbank branch: y = x - mean(x); z[n] = y[n] - 0.97 y[n - 1] (z[0] = 0.03 y[0]); * symmetric hamming; zero-pad to 2048;
//   FFT; power = (re^2 + im^2) * 2^30 -- the steps Export_DFSMN.py:97-120 folds into its analysis matrix, in Kaldi's order.  STFT branch: x * hamming; FFT_1920.
__global__ __launch_bounds__(256) void k_dfsmn_analysis(const int16_t* __restrict__ pcm, const float* __restrict__ fpcm, int L, int T, fft::Plan p2048, fft::Plan p1920,
                                                        const float2* __restrict__ tw2048, const float2* __restrict__ tw1920, const float* __restrict__ win_fb,
                                                        const float* __restrict__ win_st, float* __restrict__ power, float* __restrict__ spec) {
    __shared__ float2 A[kKaldiNfft];
    __shared__ float2 B[kKaldiNfft];
    __shared__ double red[2][256];
    const int tid = threadIdx.x;
    int frame[2];
    size_t at[2];
    bool live[2];
    const int ppr = (T + 1) / 2, b = (int)blockIdx.x / ppr, t0 = 2 * ((int)blockIdx.x - b * ppr);      // pairs never straddle two calls: a row's result must not depend
#pragma unroll                                                                                          // on which row shares its transform (batch rows are independent calls)
    for (int u = 0; u < 2; ++u) {
        live[u] = t0 + u < T;
        const int t = live[u] ? t0 + u : t0;
        frame[u] = b * T + t;
        at[u] = (size_t)b * L + (size_t)t * kHopD;
    }
    auto sample = [&](int u, int n) { return (fpcm ? fpcm[at[u] + n] : (float)pcm[at[u] + n]) * (1.0f / 32768.0f); };
    double part[2] = {0.0, 0.0};
    for (int n = tid; n < kFrame; n += 256) { part[0] += (double)sample(0, n); part[1] += (double)sample(1, n); }
    red[0][tid] = part[0];
    red[1][tid] = part[1];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) { red[0][tid] += red[0][tid + o]; red[1][tid] += red[1][tid + o]; }
        __syncthreads();
    }
    const float mean[2] = {(float)(red[0][0] / (double)kFrame), (float)(red[1][0] / (double)kFrame)};
    for (int n = tid; n < kKaldiNfft; n += 256) {
        float v[2] = {0.0f, 0.0f};
        if (n < kFrame) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const float y = sample(u, n) - mean[u], yp = sample(u, n > 0 ? n - 1 : 0) - mean[u];
                v[u] = live[u] ? (y - 0.97f * yp) * win_fb[n] : 0.0f;          // an odd row's last frame is paired with zeros
            }
        }
        A[n] = make_float2(v[0], v[1]);
    }
    float2* r = fft::forward(A, B, p2048, tw2048, tid, 256);
    for (int f = tid; f < kFbBins; f += 256) {
        const float2 z = r[f], zc = r[f == 0 ? 0 : kKaldiNfft - f];
        const float re0 = 0.5f * (z.x + zc.x), im0 = 0.5f * (z.y - zc.y), re1 = 0.5f * (z.y + zc.y), im1 = 0.5f * (zc.x - z.x);
        if (live[0]) power[(size_t)frame[0] * kFbBins + f] = (re0 * re0 + im0 * im0) * (32768.0f * 32768.0f);
        if (live[1]) power[(size_t)frame[1] * kFbBins + f] = (re1 * re1 + im1 * im1) * (32768.0f * 32768.0f);
    }
    __syncthreads();
    for (int n = tid; n < kNfftD; n += 256) { const float w = win_st[n]; A[n] = make_float2(sample(0, n) * w, live[1] ? sample(1, n) * w : 0.0f); }
    r = fft::forward(A, B, p1920, tw1920, tid, 256);
    for (int f = tid; f < kStBins; f += 256) {
        const float2 z = r[f], zc = r[f == 0 ? 0 : kNfftD - f];
        if (live[0]) {
            spec[(size_t)frame[0] * 2 * kStBins + f] = 0.5f * (z.x + zc.x);
            spec[(size_t)frame[0] * 2 * kStBins + kStBins + f] = 0.5f * (z.y - zc.y);
        }
        if (live[1]) {
            spec[(size_t)frame[1] * 2 * kStBins + f] = 0.5f * (z.y + zc.y);
            spec[(size_t)frame[1] * 2 * kStBins + kStBins + f] = 0.5f * (zc.x - z.x);
        }
    }
}

// One workgroup per PAIR of frames: mask * spectrum (:236-237) -> Hermitian extension -> inverse FFT_1920 -> * periodic hamming / N = the reference's inverse table applied
// to the masked half spectrum (its sine rows of the DC and Nyquist bins are zero, so their imaginary parts do not contribute: set to zero here).
// x N = sum_f Z[f] e^{+i theta} = conj(DFT(conj Z)) with Z Hermitian; two frames share one transform as W = H(Z0) + i H(Z1): x0 + i x1 = conj(DFT(conj W)) / N.
__global__ __launch_bounds__(256) void k_dfsmn_synthesis(const float* __restrict__ spec, const float* __restrict__ mask, int T, fft::Plan p1920, const float2* __restrict__ tw1920,
                                                         const float* __restrict__ win_syn, float* __restrict__ frames) {
    __shared__ float2 A[kNfftD];
    __shared__ float2 B[kNfftD];
    const int tid = threadIdx.x, ppr = (T + 1) / 2, b = (int)blockIdx.x / ppr, t0 = 2 * ((int)blockIdx.x - b * ppr);   // pairs inside one call's row, as in the analysis
    const int f0 = b * T + t0, f1 = f0 + 1;
    const bool two = t0 + 1 < T;
    const float *sp0 = spec + (size_t)f0 * 2 * kStBins, *mk0 = mask + (size_t)f0 * kStBins;
    const float *sp1 = spec + (size_t)(two ? f1 : f0) * 2 * kStBins, *mk1 = mask + (size_t)(two ? f1 : f0) * kStBins;
    for (int f = tid; f < kStBins; f += 256) {
        const bool edge = f == 0 || f == kStBins - 1;
        const float m0 = mk0[f], m1 = two ? mk1[f] : 0.0f;
        const float2 z0 = make_float2(sp0[f] * m0, edge ? 0.0f : sp0[kStBins + f] * m0), z1 = make_float2(sp1[f] * m1, edge ? 0.0f : sp1[kStBins + f] * m1);
        // W[f] = z0 + i z1 = (z0.x - z1.y, z0.y + z1.x); W[N - f] = conj z0 + i conj z1 = (z0.x + z1.y, z1.x - z0.y); both stored conjugated
        A[f] = make_float2(z0.x - z1.y, -(z0.y + z1.x));
        if (!edge) A[kNfftD - f] = make_float2(z0.x + z1.y, z0.y - z1.x);
    }
    const float2* r = fft::forward(A, B, p1920, tw1920, tid, 256);
    for (int n = tid; n < kNfftD; n += 256) {
        const float w = win_syn[n];
        frames[(size_t)f0 * kNfftD + n] = (r[n].x * (1.0f / (float)kNfftD)) * w;
        if (two) frames[(size_t)f1 * kNfftD + n] = (-r[n].y * (1.0f / (float)kNfftD)) * w;
    }
}

// causal depthwise memory + outer residual: x[c][j] += sum_k w[c][k] * p1[c][j - (lo-1) + k], zero before the row's first frame (:228-229)
__global__ __launch_bounds__(256) void k_fsmn_memory(const float* __restrict__ p1, const float* __restrict__ w, float* __restrict__ x, int N, int T,
                                                     int lo, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i / N), j = (int)(i - (long long)c * N);
    const int t = j % T;
    float s = 0.0f;
    const float* row = p1 + (size_t)c * N + j;
    for (int k = 0; k < lo; ++k) {                   // unconditional loads from clamped frames, masked afterwards: a branch around each load would serialise the taps
        const int dt = k - (lo - 1);
        const bool ok = t + dt >= 0;
        const float v = row[ok ? dt : 0];
        s += ok ? w[c * lo + k] * v : 0.0f;
    }
    x[i] += s;
}

// conv_transpose overlap-add as a gather, / static COLA sum, then the PCM tail (:238-244)
__global__ __launch_bounds__(256) void k_dfsmn_ola_pcm(const float* __restrict__ frames, const float* __restrict__ wsum, int16_t* __restrict__ pcm,
                                                       float* __restrict__ f32, int T, int out_len, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int b = (int)(i / out_len), m = (int)(i - (long long)b * out_len);
    int t_hi = m / kHopD;
    if (t_hi > T - 1) t_hi = T - 1;
    const int t_lo = m - kNfftD + 1 <= 0 ? 0 : (m - kNfftD + kHopD) / kHopD;
    float s = 0.0f;
    for (int t = t_lo; t <= t_hi; ++t) s += frames[((size_t)b * T + t) * kNfftD + (m - t * kHopD)];
    const float y = s / wsum[m];
    if (f32) f32[i] = y;
    if (pcm) pcm[i] = (short)(int)fminf(fmaxf(y * 32768.0f, -32768.0f), 32767.0f);
}

void hamming(int n, bool periodic, std::vector<double>& w) {   // torch.hamming_window(alpha 0.54, beta 0.46)
    w.resize((size_t)n);
    const double denom = periodic ? n : n - 1;
    for (int k = 0; k < n; ++k) w[k] = 0.54 - 0.46 * cos(2.0 * M_PI * k / denom);
}
void hamming_f32(int n, bool periodic, std::vector<float>& w) {   // the fp32 evaluation STFT_Process uses
    w.resize((size_t)n);
    const float step = (float)(2.0 * M_PI / (double)(periodic ? n : n - 1));
    for (int k = 0; k < n; ++k) w[k] = cosf((float)k * step) * (-0.46f) + 0.54f;
}

}  // namespace

struct DfsmnEngine : SubEngine {
    int device = 0, in_len_ = 0 /* one window */, n_win = 1, T = 0, out_len_ = 0, depth = 0, lorder = 0;
    float* d_w = nullptr;      // one arena: tables + weights
    const float *win_fb = nullptr, *win_st = nullptr, *win_syn = nullptr, *wsum = nullptr, *mel = nullptr, *lin1_w = nullptr, *lin1_b = nullptr, *lin2_w = nullptr,
                *lin2_b = nullptr;
    const float2 *tw2048 = nullptr, *tw1920 = nullptr;
    fft::Plan p2048, p1920;
    std::vector<const float*> uf_lin_w, uf_lin_b, uf_proj_w, uf_conv_w;
    int capacity = 0;
    float* ws = nullptr;
    float *power = nullptr, *spec = nullptr, *feat = nullptr, *x = nullptr, *f1 = nullptr, *p1 = nullptr, *mask = nullptr, *frames_buf = nullptr;

    ~DfsmnEngine() override {
        (void)hipSetDevice(device);
        if (d_w) (void)hipFree(d_w);
        if (ws) (void)hipFree(ws);
    }
    int frames() const override { return T; }
    // batch-fold (:194-198, :231-232): a call is n_win windows back to back, each an independent clip whose raw overlap-add length equals the window
    int in_len() const override { return in_len_ * n_win; }
    int out_len() const override { return out_len_ * n_win; }
    bool accepts_float_input() const override { return true; }
    int reserve(int batch, std::string& err) override;
    int run(hipStream_t s, const int16_t* d_in, int batch, int16_t* d_out, float* d_f32, std::string& err) override;
    int tap(hipStream_t s, const char* name, int batch, float* out, size_t count, size_t* written, std::string& err) override;
};

namespace {
int dfail(std::string& err, int st, const std::string& msg) { err = msg; return st; }
#define DF_HIP(expr)                                                                                  \
    do {                                                                                              \
        hipError_t _e = (expr);                                                                       \
        if (_e != hipSuccess) return dfail(err, ADE_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)
}  // namespace

int dfsmn_create(const std::map<std::string, Tensor>& tensors, int in_len, int n_win, int device, SubEngine** out, std::string& err) {
    *out = nullptr;
    if (in_len < kFrame) return dfail(err, ADE_ERR_SHAPE_MISMATCH, "dfsmn: input_audio_length shorter than one 1920-sample frame");
    auto get = [&](const std::string& name, std::vector<int> dims, const float** p) -> bool {
        auto it = tensors.find(name);
        if (it == tensors.end()) { err = "weights: tensor missing: " + name; return false; }
        if (!dims.empty() && it->second.dims != dims) { err = "weights: tensor has the wrong shape: " + name; return false; }
        *p = it->second.data;
        return true;
    };
    int depth = 0;
    while (tensors.count("uf_lin_w_" + std::to_string(depth))) ++depth;
    if (depth < 1) return dfail(err, ADE_ERR_MISSING_KEY, "weights: tensor missing: uf_lin_w_0");
    const float *mel, *l1w, *l1b, *l2w, *l2b;
    if (!get("mel_banks", {kMel, kFbBins}, &mel) || !get("lin1_w", {kHid, kMel}, &l1w) || !get("lin1_b", {kHid}, &l1b) ||
        !get("lin2_w", {kStBins, kHid}, &l2w) || !get("lin2_b", {kStBins}, &l2b))
        return err.find("missing") != std::string::npos ? ADE_ERR_MISSING_KEY : ADE_ERR_SHAPE_MISMATCH;
    auto itc = tensors.find("uf_conv_w_0");
    if (itc == tensors.end() || itc->second.dims.size() != 2 || itc->second.dims[0] != kHid)
        return dfail(err, ADE_ERR_SHAPE_MISMATCH, "weights: uf_conv_w_0 must be (256, lorder)");
    const int lo = itc->second.dims[1];
    std::vector<const float*> hw(4 * depth);
    for (int i = 0; i < depth; ++i) {
        const std::string s = std::to_string(i);
        if (!get("uf_lin_w_" + s, {kHid, kHid}, &hw[4 * i]) || !get("uf_lin_b_" + s, {kHid}, &hw[4 * i + 1]) ||
            !get("uf_proj_w_" + s, {kHid, kHid}, &hw[4 * i + 2]) || !get("uf_conv_w_" + s, {kHid, lo}, &hw[4 * i + 3]))
            return err.find("missing") != std::string::npos ? ADE_ERR_MISSING_KEY : ADE_ERR_SHAPE_MISMATCH;
    }
    DfsmnEngine* d = new DfsmnEngine();
    d->device = device;
    d->in_len_ = in_len;
    d->n_win = n_win;
    d->T = (in_len - kFrame) / kHopD + 1;                       // STFT_SIGNAL_LENGTH (Export_DFSMN.py:66)
    d->out_len_ = kNfftD + kHopD * (d->T - 1);                   // raw conv_transpose length, no centre trim
    d->depth = depth;
    d->lorder = lo;

    // ---- host tables
    std::vector<float> arena;
    auto push = [&](const float* src, size_t n) { const size_t off = arena.size(); arena.resize(off + ((n + 63) & ~(size_t)63)); if (src) memcpy(&arena[off], src, n * sizeof(float)); return off; };
    // ---- windows and twiddles
    std::vector<float> wa, wsyn, wfb((size_t)kFrame);
    {
        std::vector<double> win;
        hamming(kFrame, false, win);        // the Kaldi filter bank's symmetric hamming, float64 rounded once like the folded matrix it replaces (Export_DFSMN.py:97-120)
        for (int n = 0; n < kFrame; ++n) wfb[n] = (float)win[n];
    }
    hamming_f32(kNfftD, false, wa);     // analysis: symmetric hamming (DFSMN/STFT_Process.py:92)
    hamming_f32(kNfftD, true, wsyn);    // synthesis: periodic hamming (:93)
    const size_t o_wfb = push(wfb.data(), wfb.size()), o_wst = push(wa.data(), wa.size()), o_wsyn = push(wsyn.data(), wsyn.size());
    auto twiddles = [&](int n) {
        std::vector<float> tw((size_t)2 * n);
        for (int m = 0; m < n; ++m) { const double a = -2.0 * M_PI * (double)m / (double)n; tw[2 * m] = (float)cos(a); tw[2 * m + 1] = (float)sin(a); }
        return push(tw.data(), tw.size());
    };
    const size_t o_tw2048 = twiddles(kKaldiNfft), o_tw1920 = twiddles(kNfftD);
    if (!fft::make_plan(kKaldiNfft, &d->p2048) || !fft::make_plan(kNfftD, &d->p1920)) { delete d; return dfail(err, ADE_ERR_UNSUPPORTED, "dfsmn: FFT plan"); }
    const size_t o_ws = push(nullptr, (size_t)d->out_len_);
    for (int t = 0; t < d->T; ++t)
        for (int n = 0; n < kNfftD; ++n) arena[o_ws + (size_t)t * kHopD + n] += wsyn[n] * wsyn[n];
    const size_t o_mel = push(mel, (size_t)kMel * kFbBins), o_l1w = push(l1w, (size_t)kHid * kMel), o_l1b = push(l1b, kHid),
                 o_l2w = push(l2w, (size_t)kStBins * kHid), o_l2b = push(l2b, kStBins);
    std::vector<size_t> o_h(4 * depth);
    for (int i = 0; i < depth; ++i) {
        o_h[4 * i] = push(hw[4 * i], (size_t)kHid * kHid);
        o_h[4 * i + 1] = push(hw[4 * i + 1], kHid);
        o_h[4 * i + 2] = push(hw[4 * i + 2], (size_t)kHid * kHid);
        o_h[4 * i + 3] = push(hw[4 * i + 3], (size_t)kHid * lo);
    }
    auto bail = [&](int st) { delete d; return st; };
    if (hipSetDevice(device) != hipSuccess) return bail(dfail(err, ADE_ERR_DEVICE, "hipSetDevice failed"));
    if (hipMalloc((void**)&d->d_w, arena.size() * sizeof(float)) != hipSuccess) return bail(dfail(err, ADE_ERR_DEVICE, "hipMalloc of the DFSMN weights failed"));
    if (hipMemcpy(d->d_w, arena.data(), arena.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
        return bail(dfail(err, ADE_ERR_DEVICE, "upload of the DFSMN weights failed"));
    d->win_fb = d->d_w + o_wfb; d->win_st = d->d_w + o_wst; d->win_syn = d->d_w + o_wsyn; d->wsum = d->d_w + o_ws; d->mel = d->d_w + o_mel;
    d->tw2048 = reinterpret_cast<const float2*>(d->d_w + o_tw2048); d->tw1920 = reinterpret_cast<const float2*>(d->d_w + o_tw1920);
    d->lin1_w = d->d_w + o_l1w; d->lin1_b = d->d_w + o_l1b; d->lin2_w = d->d_w + o_l2w; d->lin2_b = d->d_w + o_l2b;
    for (int i = 0; i < depth; ++i) {
        d->uf_lin_w.push_back(d->d_w + o_h[4 * i]);
        d->uf_lin_b.push_back(d->d_w + o_h[4 * i + 1]);
        d->uf_proj_w.push_back(d->d_w + o_h[4 * i + 2]);
        d->uf_conv_w.push_back(d->d_w + o_h[4 * i + 3]);
    }
    *out = d;
    return ADE_OK;
}

int DfsmnEngine::reserve(int calls, std::string& err) {
    DfsmnEngine* d = this;
    if (calls <= d->capacity) return ADE_OK;
    const int batch = calls * n_win;
    DF_HIP(hipSetDevice(d->device));
    DF_HIP(hipDeviceSynchronize());
    if (d->ws) (void)hipFree(d->ws);
    d->ws = nullptr;
    d->capacity = 0;
    const size_t N = (size_t)batch * d->T;
    const size_t sizes[8] = {(size_t)kFbBins * N, (size_t)2 * kStBins * N, (size_t)kMel * N, (size_t)kHid * N, (size_t)kHid * N, (size_t)kHid * N, (size_t)kStBins * N,
                             N * kNfftD};
    size_t total = 0;
    for (size_t s : sizes) total += (s + 63) & ~(size_t)63;
    DF_HIP(hipMalloc((void**)&d->ws, total * sizeof(float)));
    float** ptrs[8] = {&d->power, &d->spec, &d->feat, &d->x, &d->f1, &d->p1, &d->mask, &d->frames_buf};
    size_t off = 0;
    for (int i = 0; i < 8; ++i) { *ptrs[i] = d->ws + off; off += (sizes[i] + 63) & ~(size_t)63; }
    d->capacity = calls;
    return ADE_OK;
}

int DfsmnEngine::run(hipStream_t s, const int16_t* d_in, int batch, int16_t* d_out, float* d_f32, std::string& err) {
    DfsmnEngine* d = this;
    if (batch == 0) return ADE_OK;
    int st = reserve(batch, err);
    if (st != ADE_OK) return st;
    batch *= n_win;
    using namespace gemm;
    const int N = batch * d->T;
    // analysis: Kaldi filter-bank power and the mask STFT of every frame as FFTs                              (Export_DFSMN.py:205-209, 216)
    hipLaunchKernelGGL(k_dfsmn_analysis, dim3((unsigned)(batch * ((d->T + 1) / 2))), dim3(256), 0, s, d_in, float_in, d->in_len_, d->T, d->p2048, d->p1920, d->tw2048, d->tw1920, d->win_fb, d->win_st,
                       d->power, d->spec);
    // Kaldi log-mel: mel_banks x power, clamp(eps), log                                                  (:216-217)
    launch(s, RowMajorA{d->mel, kFbBins}, PowerFrameB{d->power}, BiasActStore<kActLogFloor>{d->feat, N, nullptr, 1.1920928955078125e-07f}, kMel, N, kFbBins);
    // mask network                                                                                          (:224-230)
    launch(s, RowMajorA{d->lin1_w, kMel}, RowMajorB{d->feat, N}, BiasActStore<kActRelu>{d->x, N, d->lin1_b, 0.0f}, kHid, N, kMel);
    for (int i = 0; i < d->depth; ++i) {
        launch(s, RowMajorA{d->uf_lin_w[i], kHid}, RowMajorB{d->x, N}, BiasActStore<kActRelu>{d->f1, N, d->uf_lin_b[i], 0.0f}, kHid, N, kHid);
        launch(s, RowMajorA{d->uf_proj_w[i], kHid}, RowMajorB{d->f1, N}, BiasActStore<kActNone>{d->p1, N, nullptr, 0.0f}, kHid, N, kHid);
        const long long total = (long long)kHid * N;
        hipLaunchKernelGGL(k_fsmn_memory, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const float*)d->p1, d->uf_conv_w[i], d->x, N, d->T,
                           d->lorder, total);
    }
    launch(s, RowMajorA{d->lin2_w, kHid}, RowMajorB{d->x, N}, SigmoidTStore{d->mask, d->lin2_b}, kStBins, N, kHid);
    // masked spectrum -> ISTFT frames, then overlap-add + PCM tail                                        (:236-244)
    hipLaunchKernelGGL(k_dfsmn_synthesis, dim3((unsigned)(batch * ((d->T + 1) / 2))), dim3(256), 0, s, (const float*)d->spec, (const float*)d->mask, d->T, d->p1920, d->tw1920, d->win_syn, d->frames_buf);
    const long long total = (long long)batch * d->out_len_;
    hipLaunchKernelGGL(k_dfsmn_ola_pcm, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const float*)d->frames_buf, d->wsum, d_out, d_f32, d->T,
                       d->out_len_, total);
    DF_HIP(hipGetLastError());
    return ADE_OK;
}

int DfsmnEngine::tap(hipStream_t s, const char* name, int batch, float* out, size_t count, size_t* written, std::string& err) {
    DfsmnEngine* d = this;
    const size_t N = (size_t)batch * n_win * d->T;
    const float* src = nullptr;
    size_t n = 0;
    if (strcmp(name, "logmel") == 0) { src = d->feat; n = kMel * N; }
    else if (strcmp(name, "mask") == 0) { src = d->mask; n = kStBins * N; }          // frame-major (frames, 961)
    else return dfail(err, ADE_ERR_NOT_FOUND, std::string("unknown tap: ") + name);
    if (!src || batch <= 0) return dfail(err, ADE_ERR_NOT_FOUND, "tap has no data yet");
    if (count < n) return dfail(err, ADE_ERR_SHAPE_MISMATCH, "tap buffer too small");
    DF_HIP(hipStreamSynchronize(s));
    DF_HIP(hipMemcpy(out, src, n * sizeof(float), hipMemcpyDeviceToHost));
    *written = n;
    return ADE_OK;
}

}  // namespace ade
