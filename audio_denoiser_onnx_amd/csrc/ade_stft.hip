// ade_stft.hip — generic STFT_Process operator on the MI355X matrix cores.
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

struct ade_stft_plan {
    int device = 0;
    ade::StftDims d{};
    int center = 1;
    float *d_fwd = nullptr, *d_inv = nullptr, *d_wsq = nullptr, *d_frames = nullptr;
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
    // tables (STFT_Process.py:213-251), exact angles
    std::vector<float> fwd((size_t)2 * F * N), inv((size_t)2 * F * N), wsq((size_t)N);
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
    for (int n = 0; n < N; ++n) wsq[n] = ws[n] * ws[n];
    auto bail = [&](ade_status st) { g_stft_create_error = p->last_error; ade_stft_destroy(p); return st; };
    if (hipSetDevice(device) != hipSuccess) return bail(sfail(p, ADE_ERR_DEVICE, "hipSetDevice failed"));
    if (hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking) != hipSuccess) return bail(sfail(p, ADE_ERR_DEVICE, "hipStreamCreate failed"));
    const size_t tb = fwd.size() * sizeof(float);
    if (hipMalloc((void**)&p->d_fwd, tb) != hipSuccess || hipMalloc((void**)&p->d_inv, tb) != hipSuccess ||
        hipMalloc((void**)&p->d_wsq, wsq.size() * sizeof(float)) != hipSuccess)
        return bail(sfail(p, ADE_ERR_DEVICE, "hipMalloc of the DFT tables failed"));
    if (hipMemcpy(p->d_fwd, fwd.data(), tb, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(p->d_inv, inv.data(), tb, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(p->d_wsq, wsq.data(), wsq.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
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
    *out_len = p->center ? raw - p->d.n_fft : raw;
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
    ade::gemm::launch(s, ade::gemm::RowMajorA{p->d_fwd, p->d.n_fft}, ade::FrameB{d_x, p->d, length, T}, ade::SpecStore{d_spec, p->d.F2, T},
                      p->d.F2, batch * T, p->d.n_fft);
    STFT_HIP(p, hipGetLastError());
    if (!hip_stream) STFT_HIP(p, hipStreamSynchronize(s));
    return ADE_OK;
}

}  // extern "C"

namespace {
template <class ALoader>
ade_status synthesize_from(ade_stft_handle p, ALoader a, int batch, int frames, float* d_y, void* hip_stream) {
    STFT_HIP(p, hipSetDevice(p->device));
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : p->stream;
    const size_t need = (size_t)batch * frames * p->d.n_fft;
    if (need > p->frames_cap) {
        STFT_HIP(p, hipStreamSynchronize(s));
        if (p->d_frames) (void)hipFree(p->d_frames);
        p->d_frames = nullptr;
        p->frames_cap = 0;
        STFT_HIP(p, hipMalloc((void**)&p->d_frames, need * sizeof(float)));
        p->frames_cap = need;
    }
    ade::gemm::launch(s, a, ade::gemm::RowMajorB{p->d_inv, p->d.n_fft}, ade::gemm::BiasActStore<ade::gemm::kActNone>{p->d_frames, p->d.n_fft, nullptr, 0.0f},
                      batch * frames, p->d.n_fft, p->d.F2);
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
    return synthesize_from(p, ade::SpecA{d_spec, p->d.F2, frames}, batch, frames, d_y, hip_stream);
}

ade_status ade_stft_synthesize_polar(ade_stft_handle p, const float* d_mag, const float* d_phase, int batch, int frames, float* d_y, void* hip_stream) {
    if (!p || batch < 0 || frames < 1 || (batch > 0 && (!d_mag || !d_phase || !d_y))) return sfail(p, ADE_ERR_BAD_VALUE, "ade_stft_synthesize_polar: bad arguments");
    if (batch == 0) return ADE_OK;
    return synthesize_from(p, ade::PolarSpecA{d_mag, d_phase, p->d.F2 / 2, frames}, batch, frames, d_y, hip_stream);
}

const char* ade_stft_last_error(ade_stft_handle p) { return p ? p->last_error.c_str() : g_stft_create_error.c_str(); }

void ade_stft_destroy(ade_stft_handle p) {
    if (!p) return;
    (void)hipSetDevice(p->device);
    if (p->d_fwd) (void)hipFree(p->d_fwd);
    if (p->d_inv) (void)hipFree(p->d_inv);
    if (p->d_wsq) (void)hipFree(p->d_wsq);
    if (p->d_frames) (void)hipFree(p->d_frames);
    if (p->stream) (void)hipStreamDestroy(p->stream);
    delete p;
}

}  // extern "C"
