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
// One 256-thread workgroup computes a 128 x 128 tile; each of its 4 wavefronts owns a 64 x 64 quadrant as 4 x 4 MFMA
// tiles (64 accumulator VGPRs); operands are staged k-major in LDS with a row stride of 144 floats, which makes the
// per-lane operand reads (16 consecutive rows x 4 consecutive k) conflict-free.
// Tables use exact angles (reduced f*n mod N, evaluated in double); the reference evaluates cos/sin of fp32 angles up
// to 2*pi*N/2, which costs it up to 1e-4 relative -- the parity tests price that difference explicitly.
#include "ade_device.h"
#include "ade_internal.h"
#include "../../include/ade.h"

#include <cmath>
#include <cstring>
#include <string>
#include <vector>

namespace ade {
namespace {

using namespace dev;

constexpr int kTM = 128, kTN = 128, kTK = 16;      // workgroup tile
constexpr int kLds = 144;                          // k-major LDS row stride (floats): 144 mod 64 = 16

struct StftDims {
    int n_fft, hop, F2;        // F2 = 2 * (n_fft/2 + 1)
    int pad;                   // n_fft/2 when centre-padded, else 0
    int reflect;               // 1: reflect, 0: zeros (only meaningful with pad > 0)
};

// MODE 0: analysis.  A = K table [F2][n_fft] ; B(k, j) = padded sample ; C(m, j) -> spec[b][m][t]
// MODE 1: synthesis frames.  A(j, k) = spec[b][k][t] ; B = Kinv [F2][n_fft] ; C(j, n) -> frames[j][n]
template <int MODE>
__global__ __launch_bounds__(256) void k_stft_gemm(const float* __restrict__ tab, const float* __restrict__ src, float* __restrict__ dst,
                                                   StftDims d, int Bn, int L, int T) {
    __shared__ float As[kTK * kLds];
    __shared__ float Bs[kTK * kLds];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int M = MODE == 0 ? d.F2 : Bn * T;
    const int N = MODE == 0 ? Bn * T : d.n_fft;
    const int K = MODE == 0 ? d.n_fft : d.F2;
    const int m_blk = blockIdx.y * kTM, n_blk = blockIdx.x * kTN;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;       // this wavefront's quadrant
    const int j16 = lane & 15, g = lane >> 4;
    v4f acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = v4f{0.0f, 0.0f, 0.0f, 0.0f};

    for (int k0 = 0; k0 < K; k0 += kTK) {
        // ---- stage the two operand slabs, k-major
        if (MODE == 0) {
            {   // A: table rows m, 8 consecutive k per thread (two 16-byte loads)
                const int r = tid >> 1, kh = (tid & 1) * 8, m = m_blk + r;
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = (m < M && k0 + kh + u < K) ? tab[(size_t)m * d.n_fft + k0 + kh + u] : 0.0f;
#pragma unroll
                for (int u = 0; u < 8; ++u) As[(kh + u) * kLds + r] = v[u];
            }
            {   // B: frames.  thread = (k, 16 column groups): consecutive lanes read consecutive samples of one frame
                const int kk = tid & 15, jg = tid >> 4;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int jl = jg + 16 * u, j = n_blk + jl;
                    float v = 0.0f;
                    if (j < N && k0 + kk < K) {
                        const int b = j / T, t = j - b * T;
                        int idx = t * d.hop + k0 + kk - d.pad;
                        bool ok = true;
                        if (idx < 0) { ok = d.reflect != 0; idx = -idx; }
                        else if (idx >= L) { ok = d.reflect != 0; idx = 2 * (L - 1) - idx; }
                        if (ok) v = src[(size_t)b * L + idx];
                    }
                    Bs[kk * kLds + jl] = v;
                }
            }
        } else {
            {   // A: spectrum, A(j, k) = spec[b][k][t]: for one k consecutive j are consecutive t
                const int jl = tid & 127, kh = (tid >> 7) * 8, j = m_blk + jl;
                const int b = j < M ? j / T : 0, t = j < M ? j - b * T : 0;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int k = k0 + kh + u;
                    As[(kh + u) * kLds + jl] = (j < M && k < K) ? src[((size_t)b * d.F2 + k) * T + t] : 0.0f;
                }
            }
            {   // B: inverse table rows k, columns n
                const int nl = tid & 127, kh = (tid >> 7) * 8, n = n_blk + nl;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int k = k0 + kh + u;
                    Bs[(kh + u) * kLds + nl] = (n < N && k < K) ? tab[(size_t)k * d.n_fft + n] : 0.0f;
                }
            }
        }
        __syncthreads();
        // ---- 4 k-steps of 4: lane (g, j16) supplies A[row 16 i + j16][k + g] and B[k + g][col 16 j + j16]
#pragma unroll
        for (int ks = 0; ks < kTK; ks += 4) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = As[(ks + g) * kLds + wm + 16 * i + j16];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = Bs[(ks + g) * kLds + wn + 16 * j + j16];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mfma16x16x4(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }
    // ---- store: lane (g, j16), register r of tile (i, j) is C[wm + 16 i + 4 g + r][wn + 16 j + j16]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m_blk + wm + 16 * i + 4 * g + r, n = n_blk + wn + 16 * j + j16;
                if (m < M && n < N) {
                    if (MODE == 0) {
                        const int b = n / T, t = n - b * T;
                        dst[((size_t)b * d.F2 + m) * T + t] = acc[i][j][r];
                    } else {
                        dst[(size_t)m * d.n_fft + n] = acc[i][j][r];
                    }
                }
            }
}

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
    const dim3 grid((unsigned)((batch * T + ade::kTN - 1) / ade::kTN), (unsigned)((p->d.F2 + ade::kTM - 1) / ade::kTM));
    hipLaunchKernelGGL(HIP_KERNEL_NAME(ade::k_stft_gemm<0>), grid, dim3(256), 0, s, (const float*)p->d_fwd, d_x, d_spec, p->d, batch, length, T);
    STFT_HIP(p, hipGetLastError());
    if (!hip_stream) STFT_HIP(p, hipStreamSynchronize(s));
    return ADE_OK;
}

ade_status ade_stft_synthesize(ade_stft_handle p, const float* d_spec, int batch, int frames, float* d_y, void* hip_stream) {
    if (!p || batch < 0 || frames < 1 || (batch > 0 && (!d_spec || !d_y))) return sfail(p, ADE_ERR_BAD_VALUE, "ade_stft_synthesize: bad arguments");
    if (batch == 0) return ADE_OK;
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
    const dim3 grid((unsigned)((p->d.n_fft + ade::kTN - 1) / ade::kTN), (unsigned)((batch * frames + ade::kTM - 1) / ade::kTM));
    hipLaunchKernelGGL(HIP_KERNEL_NAME(ade::k_stft_gemm<1>), grid, dim3(256), 0, s, (const float*)p->d_inv, d_spec, p->d_frames, p->d, batch, 0,
                       frames);
    int out_len = 0;
    (void)ade_stft_output_length(p, frames, &out_len);
    const long long total = (long long)batch * out_len;
    hipLaunchKernelGGL(ade::k_stft_ola, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const float*)p->d_frames, (const float*)p->d_wsq,
                       d_y, p->d.n_fft, p->d.hop, frames, p->center ? p->d.n_fft / 2 : 0, out_len, total);
    STFT_HIP(p, hipGetLastError());
    if (!hip_stream) STFT_HIP(p, hipStreamSynchronize(s));
    return ADE_OK;
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
