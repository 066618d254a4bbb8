// ade_dfsmn.hip — DFSMN (48 kHz acoustic noise suppression) on the MI355X: SURVEY.md section 8 row a19.
//
// Reference: DFSMN.forward, DFSMN/Export_DFSMN.py:180-246, and the buffers its constructor builds (:86-178):
//   int16 -> * 2^-15 -> ONE strided analysis convolution [Kaldi fbank (DC removal + 0.97 pre-emphasis + symmetric hamming
//   + 2048-pt DFT, folded into a (2050, 1920) matrix) | 1920-pt mask STFT] -> power -> mel(120) -> log ->
//   Linear(120,256)+ReLU -> 9 x [Linear+ReLU, Linear (no bias), causal depthwise memory (lorder 20) with the inner
//   residual folded into the current tap, outer residual] -> Linear(256,961)+Sigmoid -> mask * spectrum ->
//   ISTFT (periodic hamming, no centre pad, static COLA) -> * 32768, clamp, truncate -> int16.
// Every step but the depthwise memory is a matrix product, so the whole model is the generic matrix-core GEMM of
// csrc/ade_gemm.h with functor operands / stores (activations are channels-first (C, N) with N = batch * frames):
//   analysis   AN(3972, N)  = K_an(3972, 1920) x frames(1920, N)        B operand = int16 samples * 2^-15, framed by index
//   log-mel    F(120, N)    = mel(120, 1025) x power(1025, N)            B operand = (re^2 + im^2) * 2^30 of AN rows ; store = log(max(., eps))
//   layers     X(256, N)    = relu(W x + b) ...                          store = bias + ReLU / none / bias + sigmoid
//   synthesis  frames(N, 1920) = (mask * spectrum)^T(N, 1922) x K_inv(1922, 1920)   A operand = AN row * mask row
// followed by a gather overlap-add with the PCM tail fused.  DFT tables use exact angles (see ade_stft.hip).
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
constexpr int kAnRows = 2 * kFbBins + 2 * kStBins;   // 3972
constexpr int kMel = 120, kHid = 256;

struct PcmFrameB {             // B(k, j) = sample k of frame j = (b, t), * 2^-15 (Export_DFSMN.py:186-190, conv1d stride 960, no padding)
    static constexpr bool kAlongN = false;
    const int16_t* pcm;
    const float* fpcm;         // resampled input (floats in int16 units, :186-193) replaces pcm when set
    int L, T;
    __device__ float operator()(int k, int j) const {
        const int b = j / T, t = j - b * T;
        const size_t at = (size_t)b * L + t * kHopD + k;
        return (fpcm ? fpcm[at] : (float)pcm[at]) * (1.0f / 32768.0f);
    }
    // four consecutive samples as one 8-byte load (frames start at multiples of 960 samples, so row starts are 8-byte aligned when L % 4 == 0)
    __device__ bool can_vec4(int K) const { return !fpcm && (K & 3) == 0 && (L & 3) == 0 && (reinterpret_cast<uintptr_t>(pcm) & 7) == 0; }
    __device__ float4 vec4(int j, int k) const {
        const int b = j / T, t = j - b * T;
        const short4 v = *reinterpret_cast<const short4*>(pcm + (size_t)b * L + t * kHopD + k);
        return make_float4((float)v.x * (1.0f / 32768.0f), (float)v.y * (1.0f / 32768.0f), (float)v.z * (1.0f / 32768.0f), (float)v.w * (1.0f / 32768.0f));
    }
};
struct PowerB {                // B(f, j) = (re^2 + im^2) * 32768^2 of the fbank half of AN (:216)
    static constexpr bool kAlongN = true;
    const float* an;
    int N;
    __device__ float operator()(int f, int j) const {
        const float re = an[(size_t)f * N + j], im = an[(size_t)(kFbBins + f) * N + j];
        return (re * re + im * im) * (32768.0f * 32768.0f);
    }
};
struct MaskedSpecA {           // A(j, k) = spectrum row k * mask row (k mod 961)  (:236-237); consecutive j contiguous
    static constexpr bool kAlongK = false;
    const float* spec;         // AN + 2050 * N
    const float* mask;
    int N;
    __device__ float operator()(int j, int k) const {
        const int f = k < kStBins ? k : k - kStBins;
        return spec[(size_t)k * N + j] * mask[(size_t)f * N + j];
    }
};

// causal depthwise memory + outer residual: x[c][j] += sum_k w[c][k] * p1[c][j - (lo-1) + k], zero before the row's first frame (:228-229)
__global__ __launch_bounds__(256) void k_fsmn_memory(const float* __restrict__ p1, const float* __restrict__ w, float* __restrict__ x, int N, int T,
                                                     int lo, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i / N), j = (int)(i - (long long)c * N);
    const int t = j % T;
    float s = 0.0f;
    for (int k = 0; k < lo; ++k) {
        const int dt = k - (lo - 1);
        if (t + dt >= 0) s += w[c * lo + k] * p1[(size_t)c * N + j + dt];
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
    const float *k_an = nullptr, *k_inv = nullptr, *wsum = nullptr, *mel = nullptr, *lin1_w = nullptr, *lin1_b = nullptr, *lin2_w = nullptr,
                *lin2_b = nullptr;
    std::vector<const float*> uf_lin_w, uf_lin_b, uf_proj_w, uf_conv_w;
    int capacity = 0;
    float* ws = nullptr;
    float *an = nullptr, *feat = nullptr, *x = nullptr, *f1 = nullptr, *p1 = nullptr, *mask = nullptr, *frames_buf = nullptr;

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
    const size_t o_an = push(nullptr, (size_t)kAnRows * kFrame);
    {   // fbank rows (Export_DFSMN.py:97-120): float64, rounded once
        std::vector<double> win;
        hamming(kFrame, false, win);
        std::vector<double> basis((size_t)kFrame), filt((size_t)kFrame);
        for (int half = 0; half < 2; ++half)
            for (int f = 0; f < kFbBins; ++f) {
                for (int n = 0; n < kFrame; ++n) {
                    const double a = 2.0 * M_PI * (double)(((long long)f * n) % kKaldiNfft) / kKaldiNfft;
                    basis[n] = (half == 0 ? cos(a) : -sin(a)) * win[n];
                }
                filt[0] = (1.0 - 0.97) * basis[0] - 0.97 * basis[1];
                for (int n = 1; n < kFrame - 1; ++n) filt[n] = basis[n] - 0.97 * basis[n + 1];
                filt[kFrame - 1] = basis[kFrame - 1];
                double mean = 0.0;
                for (int n = 0; n < kFrame; ++n) mean += filt[n];
                mean /= kFrame;
                float* row = &arena[o_an + (size_t)(half * kFbBins + f) * kFrame];
                for (int n = 0; n < kFrame; ++n) row[n] = (float)(filt[n] - mean);
            }
    }
    std::vector<float> wa, wsyn;
    hamming_f32(kNfftD, false, wa);     // analysis: symmetric hamming (DFSMN/STFT_Process.py:92)
    hamming_f32(kNfftD, true, wsyn);    // synthesis: periodic hamming (:93)
    const size_t o_inv = push(nullptr, (size_t)2 * kStBins * kNfftD);
    for (int f = 0; f < kStBins; ++f) {
        const float scale = (f == 0 || f == kStBins - 1) ? 1.0f : 2.0f;
        for (int n = 0; n < kNfftD; ++n) {
            const double a = 2.0 * M_PI * (double)(((long long)f * n) % kNfftD) / kNfftD;
            const float c = (float)cos(a), s = (float)sin(a);
            arena[o_an + (size_t)(2 * kFbBins + f) * kFrame + n] = c * wa[n];
            arena[o_an + (size_t)(2 * kFbBins + kStBins + f) * kFrame + n] = -s * wa[n];
            arena[o_inv + (size_t)f * kNfftD + n] = ((scale * c) * (float)(1.0 / kNfftD)) * wsyn[n];
            arena[o_inv + (size_t)(kStBins + f) * kNfftD + n] = ((scale * -s) * (float)(1.0 / kNfftD)) * wsyn[n];
        }
    }
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
    d->k_an = d->d_w + o_an; d->k_inv = d->d_w + o_inv; d->wsum = d->d_w + o_ws; d->mel = d->d_w + o_mel;
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
    const size_t sizes[7] = {(size_t)kAnRows * N, (size_t)kMel * N, (size_t)kHid * N, (size_t)kHid * N, (size_t)kHid * N, (size_t)kStBins * N,
                             N * kNfftD};
    size_t total = 0;
    for (size_t s : sizes) total += (s + 63) & ~(size_t)63;
    DF_HIP(hipMalloc((void**)&d->ws, total * sizeof(float)));
    float** ptrs[7] = {&d->an, &d->feat, &d->x, &d->f1, &d->p1, &d->mask, &d->frames_buf};
    size_t off = 0;
    for (int i = 0; i < 7; ++i) { *ptrs[i] = d->ws + off; off += (sizes[i] + 63) & ~(size_t)63; }
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
    // fused analysis convolution: [fbank re | fbank im | stft re | stft im] x frames                       (Export_DFSMN.py:205-209)
    launch(s, RowMajorA{d->k_an, kFrame}, PcmFrameB{d_in, float_in, d->in_len_, d->T}, BiasActStore<kActNone>{d->an, N, nullptr, 0.0f}, kAnRows, N, kFrame);
    // Kaldi log-mel: mel_banks x power, clamp(eps), log                                                  (:216-217)
    launch(s, RowMajorA{d->mel, kFbBins}, PowerB{d->an, N}, BiasActStore<kActLogFloor>{d->feat, N, nullptr, 1.1920928955078125e-07f}, kMel, N, kFbBins);
    // mask network                                                                                          (:224-230)
    launch(s, RowMajorA{d->lin1_w, kMel}, RowMajorB{d->feat, N}, BiasActStore<kActRelu>{d->x, N, d->lin1_b, 0.0f}, kHid, N, kMel);
    for (int i = 0; i < d->depth; ++i) {
        launch(s, RowMajorA{d->uf_lin_w[i], kHid}, RowMajorB{d->x, N}, BiasActStore<kActRelu>{d->f1, N, d->uf_lin_b[i], 0.0f}, kHid, N, kHid);
        launch(s, RowMajorA{d->uf_proj_w[i], kHid}, RowMajorB{d->f1, N}, BiasActStore<kActNone>{d->p1, N, nullptr, 0.0f}, kHid, N, kHid);
        const long long total = (long long)kHid * N;
        hipLaunchKernelGGL(k_fsmn_memory, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const float*)d->p1, d->uf_conv_w[i], d->x, N, d->T,
                           d->lorder, total);
    }
    launch(s, RowMajorA{d->lin2_w, kHid}, RowMajorB{d->x, N}, BiasActStore<kActSigmoid>{d->mask, N, d->lin2_b, 0.0f}, kStBins, N, kHid);
    // masked spectrum -> ISTFT frames, then overlap-add + PCM tail                                        (:236-244)
    launch(s, MaskedSpecA{d->an + (size_t)2 * kFbBins * N, d->mask, N}, RowMajorB{d->k_inv, kNfftD}, BiasActStore<kActNone>{d->frames_buf, kNfftD, nullptr, 0.0f},
           N, kNfftD, 2 * kStBins);
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
    else if (strcmp(name, "mask") == 0) { src = d->mask; n = kStBins * N; }
    else return dfail(err, ADE_ERR_NOT_FOUND, std::string("unknown tap: ") + name);
    if (!src || batch <= 0) return dfail(err, ADE_ERR_NOT_FOUND, "tap has no data yet");
    if (count < n) return dfail(err, ADE_ERR_SHAPE_MISMATCH, "tap buffer too small");
    DF_HIP(hipStreamSynchronize(s));
    DF_HIP(hipMemcpy(out, src, n * sizeof(float), hipMemcpyDeviceToHost));
    *written = n;
    return ADE_OK;
}

}  // namespace ade
