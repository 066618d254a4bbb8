/*
 * ade_oracle.c — CPU ORACLE (test infrastructure, never shipped in the product path).
 *
 * Plain-C fp32 restatement of the reference's GTCRN chunk path.  Every function cites the
 * reference file:line it follows (paths relative to the reference repo root).  Tensor layouts
 * are the reference's (NCHW / packed (2F,T) spectra / (T,F,C) inside DPGRNN); the STFT and ISTFT
 * are the reference's dense windowed-DFT convolutions with fp32-evaluated angle tables, NOT an FFT.
 *
 * Pin: tests/test_oracle_golden.py compares every tap and the final PCM with fixtures produced by
 * running the reference in the build container (tools/make_golden_gtcrn.py).
 */
#include "ade_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define NFFT 512
#define HOP 256
#define FBINS 257
#define FERB 129
#define ERB_LOW 65
#define ERB_HIGH 192
#define ERB_BANDS 64
#define FW 33 /* width after the two stride-2 convs */
#define CH 16

static char g_err[256];
const char* ade_oracle_last_error(void) { return g_err; }
static int fail(const char* msg, const char* arg) {
    snprintf(g_err, sizeof g_err, "%s%s", msg, arg ? arg : "");
    return -1;
}

/* ------------------------------------------------------------------------------------------- */
/* weight blob (audio_denoiser_onnx_amd/weights.py)                                              */
/* ------------------------------------------------------------------------------------------- */
typedef struct {
    char name[96];
    int ndim;
    int dims[4];
    size_t count;
    const float* data;
} wt_t;

typedef struct {
    int n;
    wt_t* t;
    float* storage;
} blob_t;

static int blob_parse(const void* blob, size_t nbytes, blob_t* out) {
    const unsigned char* p = (const unsigned char*)blob;
    if (nbytes < 12 || memcmp(p, "ADEWGT01", 8) != 0) return fail("not an ADEWGT01 blob", NULL);
    uint32_t n;
    memcpy(&n, p + 8, 4);
    size_t pos = 12;
    out->n = (int)n;
    out->t = (wt_t*)calloc(n, sizeof(wt_t));
    uint64_t* offs = (uint64_t*)calloc(n, sizeof(uint64_t));
    for (uint32_t i = 0; i < n; ++i) {
        uint16_t ln;
        if (pos + 2 > nbytes) goto bad;
        memcpy(&ln, p + pos, 2);
        pos += 2;
        if (ln >= sizeof out->t[i].name || pos + ln + 2 > nbytes) goto bad;
        memcpy(out->t[i].name, p + pos, ln);
        out->t[i].name[ln] = 0;
        pos += ln;
        int dtype = p[pos], ndim = p[pos + 1];
        pos += 2;
        if (dtype != 0 || ndim > 4 || pos + 4u * ndim + 16 > nbytes) goto bad;
        out->t[i].ndim = ndim;
        size_t cnt = 1;
        for (int d = 0; d < ndim; ++d) {
            uint32_t v;
            memcpy(&v, p + pos, 4);
            pos += 4;
            out->t[i].dims[d] = (int)v;
            cnt *= v;
        }
        uint64_t off, nb;
        memcpy(&off, p + pos, 8);
        memcpy(&nb, p + pos + 8, 8);
        pos += 16;
        if (nb != cnt * 4) goto bad;
        out->t[i].count = cnt;
        offs[i] = off;
    }
    {
        size_t data0 = (pos + 63) & ~(size_t)63;
        size_t total = 0;
        for (int i = 0; i < out->n; ++i) total += out->t[i].count;
        out->storage = (float*)malloc(total * sizeof(float) + 4);
        size_t w = 0;
        for (int i = 0; i < out->n; ++i) {
            if (data0 + offs[i] + out->t[i].count * 4 > nbytes) goto bad;
            memcpy(out->storage + w, p + data0 + offs[i], out->t[i].count * 4);
            out->t[i].data = out->storage + w;
            w += out->t[i].count;
        }
    }
    free(offs);
    return 0;
bad:
    free(offs);
    return fail("malformed weight blob", NULL);
}

static const wt_t* blob_find(const blob_t* b, const char* name) {
    for (int i = 0; i < b->n; ++i)
        if (strcmp(b->t[i].name, name) == 0) return &b->t[i];
    return NULL;
}

/* ------------------------------------------------------------------------------------------- */
/* STFT_Process restatement                                                                      */
/* ------------------------------------------------------------------------------------------- */

/* torch.{hann,hamming}_window(L, periodic=True): arange(L+1) * (2*pi/L) -> cos -> * (-beta) + alpha, all fp32
 * (GTCRN/STFT_Process.py:88-97 registry; 'hann_sqrt' = hann.pow(0.5) :93), then centre pad/crop to n_fft (:100-113). */
static int build_window(float* w, int win_length, int n_fft, const char* type) {
    /* Names follow the GTCRN registry (periodic=True); a "_sym" suffix selects torch's periodic=False form, which is what
     * other folders' registries bind to the same names ('hamming' in DFSMN/STFT_Process.py:92, 'hann_sqrt' in
     * ZipEnhancer/STFT_Process.py:94); "hamming_periodic" (DFSMN/STFT_Process.py:93) is an alias of "hamming". */
    float alpha, beta;
    int do_sqrt = 0, sym = 0;
    char base[32];
    size_t tl = strlen(type);
    if (tl >= sizeof base) return fail("unsupported window type: ", type);
    strcpy(base, type);
    if (tl > 4 && strcmp(base + tl - 4, "_sym") == 0) { base[tl - 4] = 0; sym = 1; }
    if (strcmp(base, "hamming_periodic") == 0) strcpy(base, "hamming");
    if (strcmp(base, "hann") == 0) { alpha = 0.5f; beta = 0.5f; }
    else if (strcmp(base, "hann_sqrt") == 0) { alpha = 0.5f; beta = 0.5f; do_sqrt = 1; }
    else if (strcmp(base, "hamming") == 0) { alpha = 0.54f; beta = 0.46f; }
    else return fail("unsupported window type: ", type);
    float* raw = (float*)malloc(sizeof(float) * (size_t)win_length);
    const float step = (float)(2.0 * M_PI / (double)(sym ? win_length - 1 : win_length));
    for (int n = 0; n < win_length; ++n) {
        float v = cosf((float)n * step) * (-beta) + alpha;
        raw[n] = do_sqrt ? sqrtf(v) : v;
    }
    if (win_length == n_fft) {
        memcpy(w, raw, sizeof(float) * (size_t)n_fft);
    } else if (win_length < n_fft) {
        int pad_left = (n_fft - win_length) / 2;
        memset(w, 0, sizeof(float) * (size_t)n_fft);
        memcpy(w + pad_left, raw, sizeof(float) * (size_t)win_length);
    } else {
        int start = (win_length - n_fft) / 2;
        memcpy(w, raw + start, sizeof(float) * (size_t)n_fft);
    }
    free(raw);
    return 0;
}

/* _build_stft_kernels (GTCRN/STFT_Process.py:213-227): kernel[(c), n] with c in [0,F) = cos(omega)*w,
 * c in [F,2F) = -sin(omega)*w; omega = fp32(2*pi/N) * f * t evaluated in fp32 BEFORE cos/sin. */
static void build_stft_kernel(float* k, const float* w, int n_fft, int exact) {
    const int F = n_fft / 2 + 1;
    const float omega_factor = (float)(2.0 * M_PI / (double)n_fft);
    for (int f = 0; f < F; ++f) {
        const float wf = omega_factor * (float)f;
        for (int t = 0; t < n_fft; ++t) {
            const float omega = wf * (float)t;
            float c = cosf(omega), s = sinf(omega);
            if (exact) { /* test knob: exactly-reduced double angles instead of the reference's fp32 ones */
                const double a = 2.0 * M_PI * (double)(((long)f * t) % n_fft) / (double)n_fft;
                c = (float)cos(a);
                s = (float)sin(a);
            }
            k[(size_t)f * n_fft + t] = c * w[t];
            k[(size_t)(F + f) * n_fft + t] = -s * w[t];
        }
    }
}

/* _build_istft_kernels (GTCRN/STFT_Process.py:229-251): ((scale*cos)*inv_n)*w and ((scale*-sin)*inv_n)*w,
 * scale = 1 for DC and (even N) Nyquist, 2 otherwise. */
static void build_istft_kernel(float* k, const float* w, int n_fft, int exact) {
    const int F = n_fft / 2 + 1;
    const float omega_factor = (float)(2.0 * M_PI / (double)n_fft);
    const float inv_n = (float)(1.0 / (double)n_fft);
    for (int f = 0; f < F; ++f) {
        float scale = 2.0f;
        if (f == 0 || (n_fft % 2 == 0 && f == F - 1)) scale = 1.0f;
        const float wf = omega_factor * (float)f;
        for (int n = 0; n < n_fft; ++n) {
            const float omega = wf * (float)n;
            float c = cosf(omega), s = sinf(omega);
            if (exact) {
                const double a = 2.0 * M_PI * (double)(((long)f * n) % n_fft) / (double)n_fft;
                c = (float)cos(a);
                s = (float)sin(a);
            }
            k[(size_t)f * n_fft + n] = ((scale * c) * inv_n) * w[n];
            k[(size_t)(F + f) * n_fft + n] = ((scale * -s) * inv_n) * w[n];
        }
    }
}

static float dot_f32(const float* a, const float* b, int n) {
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int i = 0;
    for (; i + 8 <= n; i += 8)
        for (int j = 0; j < 8; ++j) acc[j] += a[i + j] * b[i + j];
    float s = ((acc[0] + acc[4]) + (acc[2] + acc[6])) + ((acc[1] + acc[5]) + (acc[3] + acc[7]));
    for (; i < n; ++i) s += a[i] * b[i];
    return s;
}

/* _stft_B_packed_forward (GTCRN/STFT_Process.py:303-316): reflect (manual flip :306-309) or zero pad n_fft/2,
 * then one strided conv1d.  x: [L]; out: [2F][T]. */
static void stft_packed_one(const float* x, int L, const float* kernel, int n_fft, int hop, int center, int reflect,
                            float* xp /* scratch L+n_fft */, float* out, int T) {
    const int half = n_fft / 2;
    const int F = half + 1;
    const float* src = x;
    if (center) {
        if (reflect) {
            for (int i = 0; i < half; ++i) xp[i] = x[half - i];             /* x[1:half+1].flip */
            for (int i = 0; i < half; ++i) xp[half + L + i] = x[L - 2 - i]; /* x[-(half+1):-1].flip */
        } else {
            memset(xp, 0, sizeof(float) * (size_t)half);
            memset(xp + half + L, 0, sizeof(float) * (size_t)half);
        }
        memcpy(xp + half, x, sizeof(float) * (size_t)L);
        src = xp;
    }
    for (int c = 0; c < 2 * F; ++c)
        for (int t = 0; t < T; ++t) out[(size_t)c * T + t] = dot_f32(kernel + (size_t)c * n_fft, src + (size_t)t * hop, n_fft);
}

/* static COLA denominator (GTCRN/STFT_Process.py:253-273): conv_transpose1d(ones(T), w^2, stride=hop)[out_start:out_end] */
static void build_win_sum(float* ws, const float* w, int n_fft, int hop, int T, int out_start, int out_len) {
    const int raw_len = n_fft + hop * (T - 1);
    float* raw = (float*)calloc((size_t)raw_len, sizeof(float));
    for (int t = 0; t < T; ++t)
        for (int n = 0; n < n_fft; ++n) raw[t * hop + n] += w[n] * w[n];
    memcpy(ws, raw + out_start, sizeof(float) * (size_t)out_len);
    free(raw);
}

/* _istft_B_packed_forward, static_norm branch (GTCRN/STFT_Process.py:326-336): conv_transpose1d (irDFT+window+OLA),
 * trim, divide by win_sum.  spec: [2F][T]; raw scratch: n_fft+hop*(T-1). */
static void istft_packed_one(const float* spec, int T, const float* kernel, int n_fft, int hop, const float* win_sum,
                             int out_start, int out_len, float* raw, float* out) {
    const int F = n_fft / 2 + 1;
    const int raw_len = n_fft + hop * (T - 1);
    memset(raw, 0, sizeof(float) * (size_t)raw_len);
    for (int t = 0; t < T; ++t) {
        float* dst = raw + (size_t)t * hop;
        for (int c = 0; c < 2 * F; ++c) {
            const float v = spec[(size_t)c * T + t];
            const float* kr = kernel + (size_t)c * n_fft;
            for (int n = 0; n < n_fft; ++n) dst[n] += v * kr[n];
        }
    }
    for (int i = 0; i < out_len; ++i) out[i] = raw[out_start + i] / win_sum[i];
}

/* test knob for the two generic entry points below: exact (double-angle) DFT tables instead of the reference's fp32 angles */
static int g_generic_exact = 0;
void ade_oracle_set_generic_exact_dft(int exact) { g_generic_exact = exact != 0; }

int ade_oracle_stft(const float* x, int B, int L, int n_fft, int win_length, int hop, const char* window, int center_pad,
                    const char* pad_mode, float* out, int* T_out) {
    const int F = n_fft / 2 + 1;
    float* w = (float*)malloc(sizeof(float) * (size_t)n_fft);
    if (build_window(w, win_length, n_fft, window)) { free(w); return -1; }
    float* k = (float*)malloc(sizeof(float) * (size_t)2 * F * n_fft);
    build_stft_kernel(k, w, n_fft, g_generic_exact);
    const int Lp = center_pad ? L + n_fft : L;
    const int T = (Lp - n_fft) / hop + 1;
    float* xp = (float*)malloc(sizeof(float) * (size_t)(L + n_fft));
    const int reflect = strcmp(pad_mode, "reflect") == 0;
    for (int b = 0; b < B; ++b)
        stft_packed_one(x + (size_t)b * L, L, k, n_fft, hop, center_pad, reflect, xp, out + (size_t)b * 2 * F * T, T);
    if (T_out) *T_out = T;
    free(xp); free(k); free(w);
    return 0;
}

int ade_oracle_istft(const float* spec, int B, int T, int n_fft, int win_length, int hop, const char* window,
                     int center_pad, float* out, int* out_len_p) {
    const int F = n_fft / 2 + 1;
    float* w = (float*)malloc(sizeof(float) * (size_t)n_fft);
    if (build_window(w, win_length, n_fft, window)) { free(w); return -1; }
    float* k = (float*)malloc(sizeof(float) * (size_t)2 * F * n_fft);
    build_istft_kernel(k, w, n_fft, g_generic_exact);
    const int raw_len = n_fft + hop * (T - 1);
    const int out_start = center_pad ? n_fft / 2 : 0;
    const int out_len = center_pad ? raw_len - n_fft : raw_len;
    float* ws = (float*)malloc(sizeof(float) * (size_t)out_len);
    build_win_sum(ws, w, n_fft, hop, T, out_start, out_len);
    float* raw = (float*)malloc(sizeof(float) * (size_t)raw_len);
    for (int b = 0; b < B; ++b)
        istft_packed_one(spec + (size_t)b * 2 * F * T, T, k, n_fft, hop, ws, out_start, out_len, raw, out + (size_t)b * out_len);
    if (out_len_p) *out_len_p = out_len;
    free(raw); free(ws); free(k); free(w);
    return 0;
}

/* ------------------------------------------------------------------------------------------- */
/* GTCRN network restatement (GTCRN/Export_GTCRN.py)                                             */
/* ------------------------------------------------------------------------------------------- */
typedef struct {
    const float *w_ih, *w_hh, *b_ih, *b_hh;
    int I, H;
} gru_w;

typedef struct {
    const float *pw1_w, *pw1_b, *pw1_a, *dw_w, *dw_b, *dw_a, *pw2_w, *pw2_b;
    gru_w tra_gru;
    const float *tra_fc_w, *tra_fc_b;
    int dilation, deconv;
} gtconv_w;

typedef struct {
    gru_w intra[2][2]; /* [group rnn1/rnn2][fwd/reverse] */
    gru_w inter[2];
    const float *intra_fc_w, *intra_fc_b, *intra_ln_w, *intra_ln_b;
    const float *inter_fc_w, *inter_fc_b, *inter_ln_w, *inter_ln_b;
} dpgrnn_w;

typedef struct {
    const float *w, *b, *a; /* a = PReLU slope or NULL (Tanh) */
} convblock_w;

#define NTAPS 48
typedef struct {
    char name[32];
    float* data;
    size_t count;
} tap_t;

struct ade_oracle {
    blob_t blob;
    int in_len, T, out_len;
    float window[NFFT];
    float* stft_kernel;  /* [514][512] */
    float* istft_kernel; /* [514][512] */
    float* win_sum;      /* [out_len] */
    const float *erb_w_t, *ierb_w_t;
    convblock_w en0, en1, de3, de4;
    gtconv_w en_gt[3], de_gt[3];
    dpgrnn_w dp[2];
    tap_t taps[NTAPS];
    int ntaps;
};

static inline float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
static inline float prelu_(float x, float a) { return x >= 0.0f ? x : a * x; }

/* nn.GRU step semantics (PyTorch gate order r,z,n; n = tanh(W_in x + b_in + r*(W_hn h + b_hn)); h' = (1-z)*n + z*h),
 * zero initial state (Export_GTCRN.py:339-352,411-422; ONNX linear_before_reset=1, Rewrite_ONNX_GRU_Zero_State.py:97).
 * x: S rows of stride xs; y: S rows of stride ys (H values written at y + s*ys). */
static void gru_run(const gru_w* g, const float* x, int S, int xs, int reverse, float* y, int ys) {
    const int H = g->H, I = g->I;
    float h[16], hn[16];
    for (int j = 0; j < H; ++j) h[j] = 0.0f;
    for (int step = 0; step < S; ++step) {
        const int s = reverse ? S - 1 - step : step;
        const float* xv = x + (size_t)s * xs;
        for (int j = 0; j < H; ++j) {
            float gi[3], gh[3];
            for (int q = 0; q < 3; ++q) {
                const float* wi = g->w_ih + (size_t)(q * H + j) * I;
                const float* wh = g->w_hh + (size_t)(q * H + j) * H;
                float a = g->b_ih[q * H + j];
                for (int k = 0; k < I; ++k) a += wi[k] * xv[k];
                float c = g->b_hh[q * H + j];
                for (int k = 0; k < H; ++k) c += wh[k] * h[k];
                gi[q] = a;
                gh[q] = c;
            }
            const float r = sigmoidf_(gi[0] + gh[0]);
            const float z = sigmoidf_(gi[1] + gh[1]);
            const float n = tanhf(gi[2] + r * gh[2]);
            hn[j] = (1.0f - z) * n + z * h[j];
        }
        for (int j = 0; j < H; ++j) {
            h[j] = hn[j];
            y[(size_t)s * ys + j] = hn[j];
        }
    }
}

/* ConvBlock, Conv2d branch (Export_GTCRN.py:159-197, instances :488-489): kernel (1,5), stride (1,2), pad (0,2),
 * groups g, BN folded (:171-194), PReLU with one slope.  x: (Cin,T,Fi) -> y: (Cout,T,Fo), Fo = (Fi-1)/2+1. */
static void conv_block(const convblock_w* cw, const float* x, int Cin, int Cout, int groups, int T, int Fi, float* y) {
    const int Fo = (Fi + 4 - 5) / 2 + 1;
    const int Ig = Cin / groups, Og = Cout / groups;
    for (int co = 0; co < Cout; ++co) {
        const int g = co / Og;
        for (int t = 0; t < T; ++t)
            for (int fo = 0; fo < Fo; ++fo) {
                float acc = cw->b[co];
                for (int ci = 0; ci < Ig; ++ci) {
                    const float* xr = x + ((size_t)(g * Ig + ci) * T + t) * Fi;
                    const float* wr = cw->w + ((size_t)co * Ig + ci) * 5;
                    for (int k = 0; k < 5; ++k) {
                        const int fi = 2 * fo - 2 + k;
                        if (fi >= 0 && fi < Fi) acc += wr[k] * xr[fi];
                    }
                }
                y[((size_t)co * T + t) * Fo + fo] = prelu_(acc, cw->a[0]);
            }
    }
}

/* ConvBlock, ConvTranspose2d branch (Export_GTCRN.py:165-166,180-185; instances :515-516): weight (Cin, Cout/groups,1,5),
 * stride (1,2), pad (0,2): fo = 2*fi - 2 + k; PReLU or (is_last) Tanh.  x: (Cin,T,Fi) -> y: (Cout,T,Fo=2*Fi-1). */
static void deconv_block(const convblock_w* cw, const float* x, int Cin, int Cout, int groups, int T, int Fi, float* y) {
    const int Fo = (Fi - 1) * 2 - 4 + 5;
    const int Ig = Cin / groups, Og = Cout / groups;
    for (int co = 0; co < Cout; ++co)
        for (int t = 0; t < T; ++t)
            for (int fo = 0; fo < Fo; ++fo) y[((size_t)co * T + t) * Fo + fo] = cw->b[co];
    for (int g = 0; g < groups; ++g)
        for (int ci = 0; ci < Ig; ++ci)
            for (int co = 0; co < Og; ++co) {
                const float* wr = cw->w + ((size_t)(g * Ig + ci) * Og + co) * 5;
                for (int t = 0; t < T; ++t) {
                    const float* xr = x + ((size_t)(g * Ig + ci) * T + t) * Fi;
                    float* yr = y + ((size_t)(g * Og + co) * T + t) * Fo;
                    for (int fi = 0; fi < Fi; ++fi)
                        for (int k = 0; k < 5; ++k) {
                            const int fo = 2 * fi - 2 + k;
                            if (fo >= 0 && fo < Fo) yr[fo] += wr[k] * xr[fi];
                        }
                }
            }
    for (size_t i = 0; i < (size_t)Cout * T * Fo; ++i) y[i] = cw->a ? prelu_(y[i], cw->a[0]) : tanhf(y[i]);
}

/* TRA (Export_GTCRN.py:144-156): zt = mean_F(x^2) -> GRU(8->16) over T -> Linear(16->8) -> sigmoid -> x * at.
 * x: (8,T,F) in place.  Optional taps: gru output (T,16). */
static void tra_apply(const gtconv_w* w, float* x, int T, int F, float* tap_gru) {
    float* zt = (float*)malloc(sizeof(float) * (size_t)T * 8);
    float* hs = (float*)malloc(sizeof(float) * (size_t)T * 16);
    for (int c = 0; c < 8; ++c)
        for (int t = 0; t < T; ++t) {
            const float* xr = x + ((size_t)c * T + t) * F;
            float s = 0.0f;
            for (int f = 0; f < F; ++f) s += xr[f] * xr[f];
            zt[(size_t)t * 8 + c] = s / (float)F;
        }
    gru_run(&w->tra_gru, zt, T, 8, 0, hs, 16);
    if (tap_gru) memcpy(tap_gru, hs, sizeof(float) * (size_t)T * 16);
    for (int t = 0; t < T; ++t)
        for (int c = 0; c < 8; ++c) {
            float a = w->tra_fc_b[c];
            for (int k = 0; k < 16; ++k) a += w->tra_fc_w[c * 16 + k] * hs[(size_t)t * 16 + k];
            const float at = sigmoidf_(a);
            float* xr = x + ((size_t)c * T + t) * F;
            for (int f = 0; f < F; ++f) xr[f] *= at;
        }
    free(zt);
    free(hs);
}

/* GTConvBlock.forward (Export_GTCRN.py:303-324): split 8|8 -> SFE(3) (:117-141) -> 1x1 (24->16)+BN+PReLU ->
 * causal dilated depthwise 3x3 (+BN+PReLU) -> 1x1 (16->8)+BN -> TRA -> interleave with the bypass half.
 * Encoder: Conv2d on a (k-1)*dil zero-padded input (:314-318).  Decoder: ConvTranspose2d, output[:-pad] (:311-312),
 * pointwise ConvTranspose2d weights are (Cin,Cout,1,1).  x,(y): (16,T,F). */
static void gtconv_block(const gtconv_w* w, const float* x, int T, int F, float* y, float* tap_pw1, float* tap_dw,
                         float* tap_pw2, float* tap_gru, float* tap_tra) {
    const size_t plane = (size_t)T * F;
    float* h = (float*)malloc(sizeof(float) * 16 * plane);
    float* hd = (float*)malloc(sizeof(float) * 16 * plane);
    float* h1 = (float*)malloc(sizeof(float) * 8 * plane);
    const int d = w->dilation;
    /* SFE + point_conv1 + PReLU.  SFE channel (c*3+o) at f = x1[c, f-1+o], zero outside. */
    for (int co = 0; co < 16; ++co)
        for (int t = 0; t < T; ++t)
            for (int f = 0; f < F; ++f) {
                float acc = w->pw1_b[co];
                for (int c = 0; c < 8; ++c)
                    for (int o = 0; o < 3; ++o) {
                        const int ff = f - 1 + o;
                        if (ff < 0 || ff >= F) continue;
                        const int ci = c * 3 + o;
                        const float wv = w->deconv ? w->pw1_w[(size_t)ci * 16 + co] : w->pw1_w[(size_t)co * 24 + ci];
                        acc += wv * x[((size_t)c * T + t) * F + ff];
                    }
                h[((size_t)co * T + t) * F + f] = prelu_(acc, w->pw1_a[0]);
            }
    if (tap_pw1) memcpy(tap_pw1, h, sizeof(float) * 16 * plane);
    /* depthwise 3x3, dilation (d,1) */
    for (int c = 0; c < 16; ++c)
        for (int t = 0; t < T; ++t)
            for (int f = 0; f < F; ++f) {
                float acc = w->dw_b[c];
                for (int kt = 0; kt < 3; ++kt)
                    for (int kf = 0; kf < 3; ++kf) {
                        int tt, ff;
                        if (w->deconv) { tt = t - kt * d; ff = f + 1 - kf; }
                        else { tt = t - (2 - kt) * d; ff = f - 1 + kf; }
                        if (tt < 0 || tt >= T || ff < 0 || ff >= F) continue;
                        acc += w->dw_w[(size_t)c * 9 + kt * 3 + kf] * h[((size_t)c * T + tt) * F + ff];
                    }
                hd[((size_t)c * T + t) * F + f] = prelu_(acc, w->dw_a[0]);
            }
    if (tap_dw) memcpy(tap_dw, hd, sizeof(float) * 16 * plane);
    /* point_conv2 (16->8) + BN */
    for (int co = 0; co < 8; ++co)
        for (size_t p = 0; p < plane; ++p) {
            float acc = w->pw2_b[co];
            for (int ci = 0; ci < 16; ++ci) {
                const float wv = w->deconv ? w->pw2_w[(size_t)ci * 8 + co] : w->pw2_w[(size_t)co * 16 + ci];
                acc += wv * hd[(size_t)ci * plane + p];
            }
            h1[(size_t)co * plane + p] = acc;
        }
    if (tap_pw2) memcpy(tap_pw2, h1, sizeof(float) * 8 * plane);
    tra_apply(w, h1, T, F, tap_gru);
    if (tap_tra) memcpy(tap_tra, h1, sizeof(float) * 8 * plane);
    /* shuffle: out[2i] = h1[i], out[2i+1] = x2[i]  (:229-233,324) */
    for (int i = 0; i < 8; ++i) {
        memcpy(y + (size_t)(2 * i) * plane, h1 + (size_t)i * plane, sizeof(float) * plane);
        memcpy(y + (size_t)(2 * i + 1) * plane, x + (size_t)(8 + i) * plane, sizeof(float) * plane);
    }
    free(h); free(hd); free(h1);
}

/* Linear(16,16) + LayerNorm((33,16), eps=1e-8) + residual (Export_GTCRN.py:447-448,473-475,479-481).
 * r: (T,33,16) rnn output, x: residual (T,33,16), y out.  LN moments accumulated in double (stable, like ATen's
 * RowwiseMoments) then used in fp32. */
static void fc_ln_res(const float* fc_w, const float* fc_b, const float* ln_w, const float* ln_b, const float* r,
                      const float* x, int T, float* y, float* tap_ln) {
    float v[FW * CH];
    for (int t = 0; t < T; ++t) {
        const float* rt = r + (size_t)t * FW * CH;
        for (int f = 0; f < FW; ++f)
            for (int co = 0; co < CH; ++co) {
                float a = fc_b[co];
                for (int k = 0; k < CH; ++k) a += fc_w[co * CH + k] * rt[f * CH + k];
                v[f * CH + co] = a;
            }
        double s = 0.0;
        for (int i = 0; i < FW * CH; ++i) s += v[i];
        const double mean = s / (FW * CH);
        double q = 0.0;
        for (int i = 0; i < FW * CH; ++i) q += (v[i] - mean) * (v[i] - mean);
        const float rstd = (float)(1.0 / sqrt(q / (FW * CH) + 1e-8));
        const float meanf = (float)mean;
        for (int i = 0; i < FW * CH; ++i) {
            const float ln = (v[i] - meanf) * rstd * ln_w[i] + ln_b[i];
            if (tap_ln) tap_ln[(size_t)t * FW * CH + i] = ln;
            y[(size_t)t * FW * CH + i] = x[(size_t)t * FW * CH + i] + ln;
        }
    }
}

/* DPGRNN.forward (Export_GTCRN.py:466-481) on (T,33,16); GRNN = two half-width GRUs on channel halves (:409-428). */
static void dpgrnn(const dpgrnn_w* w, const float* x, int T, float* y, float* tap_intra_rnn, float* tap_intra_ln,
                   float* tap_inter_rnn /* (33,T,16) */, float* tap_inter_ln) {
    const size_t n = (size_t)T * FW * CH;
    float* r = (float*)malloc(sizeof(float) * n);
    float* mid = (float*)malloc(sizeof(float) * n);
    /* intra: sequences along F for each t; bidirectional hid 4: output [fwd4|bwd4] per group */
    for (int t = 0; t < T; ++t)
        for (int g = 0; g < 2; ++g) {
            const float* xs = x + (size_t)t * FW * CH + g * 8;
            float* ys = r + (size_t)t * FW * CH + g * 8;
            gru_run(&w->intra[g][0], xs, FW, CH, 0, ys, CH);
            gru_run(&w->intra[g][1], xs, FW, CH, 1, ys + 4, CH);
        }
    if (tap_intra_rnn) memcpy(tap_intra_rnn, r, sizeof(float) * n);
    fc_ln_res(w->intra_fc_w, w->intra_fc_b, w->intra_ln_w, w->intra_ln_b, r, x, T, mid, tap_intra_ln);
    /* inter: sequences along T for each f; uni hid 8 per group */
    for (int f = 0; f < FW; ++f)
        for (int g = 0; g < 2; ++g)
            gru_run(&w->inter[g], mid + (size_t)f * CH + g * 8, T, FW * CH, 0, r + (size_t)f * CH + g * 8, FW * CH);
    if (tap_inter_rnn)
        for (int f = 0; f < FW; ++f)
            for (int t = 0; t < T; ++t)
                memcpy(tap_inter_rnn + ((size_t)f * T + t) * CH, r + ((size_t)t * FW + f) * CH, sizeof(float) * CH);
    fc_ln_res(w->inter_fc_w, w->inter_fc_b, w->inter_ln_w, w->inter_ln_b, r, mid, T, y, tap_inter_ln);
    free(r);
    free(mid);
}

/* ------------------------------------------------------------------------------------------- */
/* engine                                                                                        */
/* ------------------------------------------------------------------------------------------- */
static const float* need(const blob_t* b, const char* prefix, const char* leaf, size_t count, int* bad) {
    char name[160];
    snprintf(name, sizeof name, "%s%s", prefix, leaf);
    const wt_t* t = blob_find(b, name);
    if (!t || t->count != count) {
        if (!*bad) fail("missing or mis-shaped tensor: ", name);
        *bad = 1;
        return NULL;
    }
    return t->data;
}

static void load_gru(const blob_t* b, const char* prefix, const char* sfx, int I, int H, gru_w* g, int* bad) {
    char leaf[64];
    g->I = I;
    g->H = H;
    snprintf(leaf, sizeof leaf, "weight_ih_l0%s", sfx); g->w_ih = need(b, prefix, leaf, (size_t)3 * H * I, bad);
    snprintf(leaf, sizeof leaf, "weight_hh_l0%s", sfx); g->w_hh = need(b, prefix, leaf, (size_t)3 * H * H, bad);
    snprintf(leaf, sizeof leaf, "bias_ih_l0%s", sfx);   g->b_ih = need(b, prefix, leaf, (size_t)3 * H, bad);
    snprintf(leaf, sizeof leaf, "bias_hh_l0%s", sfx);   g->b_hh = need(b, prefix, leaf, (size_t)3 * H, bad);
}

static void load_gtconv(const blob_t* b, const char* prefix, int dilation, int deconv, gtconv_w* w, int* bad) {
    char p[128];
    w->dilation = dilation;
    w->deconv = deconv;
    w->pw1_w = need(b, prefix, "point_conv1.weight", 16 * 24, bad);
    w->pw1_b = need(b, prefix, "point_conv1.bias", 16, bad);
    w->pw1_a = need(b, prefix, "point_act.weight", 1, bad);
    w->dw_w = need(b, prefix, "depth_conv.weight", 16 * 9, bad);
    w->dw_b = need(b, prefix, "depth_conv.bias", 16, bad);
    w->dw_a = need(b, prefix, "depth_act.weight", 1, bad);
    w->pw2_w = need(b, prefix, "point_conv2.weight", 8 * 16, bad);
    w->pw2_b = need(b, prefix, "point_conv2.bias", 8, bad);
    snprintf(p, sizeof p, "%stra.att_gru.", prefix);
    load_gru(b, p, "", 8, 16, &w->tra_gru, bad);
    w->tra_fc_w = need(b, prefix, "tra.att_fc.weight", 8 * 16, bad);
    w->tra_fc_b = need(b, prefix, "tra.att_fc.bias", 8, bad);
}

static void load_dpgrnn(const blob_t* b, const char* prefix, dpgrnn_w* w, int* bad) {
    char p[128];
    for (int g = 0; g < 2; ++g) {
        snprintf(p, sizeof p, "%sintra_rnn.rnn%d.", prefix, g + 1);
        load_gru(b, p, "", 8, 4, &w->intra[g][0], bad);
        load_gru(b, p, "_reverse", 8, 4, &w->intra[g][1], bad);
        snprintf(p, sizeof p, "%sinter_rnn.rnn%d.", prefix, g + 1);
        load_gru(b, p, "", 8, 8, &w->inter[g], bad);
    }
    w->intra_fc_w = need(b, prefix, "intra_fc.weight", 256, bad);
    w->intra_fc_b = need(b, prefix, "intra_fc.bias", 16, bad);
    w->intra_ln_w = need(b, prefix, "intra_ln.weight", FW * CH, bad);
    w->intra_ln_b = need(b, prefix, "intra_ln.bias", FW * CH, bad);
    w->inter_fc_w = need(b, prefix, "inter_fc.weight", 256, bad);
    w->inter_fc_b = need(b, prefix, "inter_fc.bias", 16, bad);
    w->inter_ln_w = need(b, prefix, "inter_ln.weight", FW * CH, bad);
    w->inter_ln_b = need(b, prefix, "inter_ln.bias", FW * CH, bad);
}

static float* tap_alloc(ade_oracle* o, const char* name, size_t count) {
    tap_t* t = &o->taps[o->ntaps++];
    snprintf(t->name, sizeof t->name, "%s", name);
    t->count = count;
    t->data = (float*)calloc(count, sizeof(float));
    return t->data;
}

static float* tap_get(const ade_oracle* o, const char* name) {
    for (int i = 0; i < o->ntaps; ++i)
        if (strcmp(o->taps[i].name, name) == 0) return o->taps[i].data;
    return NULL;
}

int ade_oracle_tap(const ade_oracle* o, const char* name, const float** data, size_t* count) {
    for (int i = 0; i < o->ntaps; ++i)
        if (strcmp(o->taps[i].name, name) == 0) {
            *data = o->taps[i].data;
            *count = o->taps[i].count;
            return 0;
        }
    return fail("unknown tap: ", name);
}

int ade_oracle_create(const void* blob, size_t nbytes, int in_len, ade_oracle** out) {
    if (in_len < NFFT / 2 + 2) return fail("in_len too short for reflect padding", NULL);
    ade_oracle* o = (ade_oracle*)calloc(1, sizeof *o);
    if (blob_parse(blob, nbytes, &o->blob)) { free(o); return -1; }
    /* static shapes: Export_GTCRN.py:45 (STATIC_SIGNAL_LENGTH = L // HOP + 1); STFT_Process.py:169-176 */
    o->in_len = in_len;
    o->T = in_len / HOP + 1;
    o->out_len = HOP * (o->T - 1);
    build_window(o->window, NFFT, NFFT, "hann_sqrt"); /* Export_GTCRN.py:34 */
    o->stft_kernel = (float*)malloc(sizeof(float) * 2 * FBINS * NFFT);
    o->istft_kernel = (float*)malloc(sizeof(float) * 2 * FBINS * NFFT);
    build_stft_kernel(o->stft_kernel, o->window, NFFT, 0);
    build_istft_kernel(o->istft_kernel, o->window, NFFT, 0);
    o->win_sum = (float*)malloc(sizeof(float) * (size_t)o->out_len);
    build_win_sum(o->win_sum, o->window, NFFT, HOP, o->T, NFFT / 2, o->out_len);
    int bad = 0;
    const blob_t* b = &o->blob;
    o->erb_w_t = need(b, "erb.", "erb_weight_t", ERB_HIGH * ERB_BANDS, &bad);
    o->ierb_w_t = need(b, "erb.", "ierb_weight_t", ERB_BANDS * ERB_HIGH, &bad);
    o->en0.w = need(b, "encoder.en_convs.0.", "conv.weight", 16 * 9 * 5, &bad);
    o->en0.b = need(b, "encoder.en_convs.0.", "conv.bias", 16, &bad);
    o->en0.a = need(b, "encoder.en_convs.0.", "act.weight", 1, &bad);
    o->en1.w = need(b, "encoder.en_convs.1.", "conv.weight", 16 * 8 * 5, &bad);
    o->en1.b = need(b, "encoder.en_convs.1.", "conv.bias", 16, &bad);
    o->en1.a = need(b, "encoder.en_convs.1.", "act.weight", 1, &bad);
    static const int en_dil[3] = {1, 2, 5}, de_dil[3] = {5, 2, 1}; /* Export_GTCRN.py:490-492,512-514 */
    char p[96];
    for (int i = 0; i < 3; ++i) {
        snprintf(p, sizeof p, "encoder.en_convs.%d.", i + 2);
        load_gtconv(b, p, en_dil[i], 0, &o->en_gt[i], &bad);
        snprintf(p, sizeof p, "decoder.de_convs.%d.", i);
        load_gtconv(b, p, de_dil[i], 1, &o->de_gt[i], &bad);
    }
    load_dpgrnn(b, "dpgrnn1.", &o->dp[0], &bad);
    load_dpgrnn(b, "dpgrnn2.", &o->dp[1], &bad);
    o->de3.w = need(b, "decoder.de_convs.3.", "conv.weight", 16 * 8 * 5, &bad);
    o->de3.b = need(b, "decoder.de_convs.3.", "conv.bias", 16, &bad);
    o->de3.a = need(b, "decoder.de_convs.3.", "act.weight", 1, &bad);
    o->de4.w = need(b, "decoder.de_convs.4.", "conv.weight", 16 * 2 * 5, &bad);
    o->de4.b = need(b, "decoder.de_convs.4.", "conv.bias", 2, &bad);
    o->de4.a = NULL;
    if (bad) { ade_oracle_destroy(o); return -1; }
    const size_t T = (size_t)o->T;
    tap_alloc(o, "audio_f32", (size_t)in_len);
    tap_alloc(o, "spec", 2 * FBINS * T);
    tap_alloc(o, "feat_erb", 3 * T * FERB);
    tap_alloc(o, "e0", 16 * T * 65);
    tap_alloc(o, "e1", 16 * T * FW);
    tap_alloc(o, "e2", 16 * T * FW);
    tap_alloc(o, "e3", 16 * T * FW);
    tap_alloc(o, "e4", 16 * T * FW);
    tap_alloc(o, "e2.pw1", 16 * T * FW);
    tap_alloc(o, "e2.dw", 16 * T * FW);
    tap_alloc(o, "e2.pw2", 8 * T * FW);
    tap_alloc(o, "e2.tra_gru", T * 16);
    tap_alloc(o, "e2.tra", 8 * T * FW);
    tap_alloc(o, "d0.pw1", 16 * T * FW);
    tap_alloc(o, "d0.dw", 16 * T * FW);
    tap_alloc(o, "d0.pw2", 8 * T * FW);
    static const char* dpn[2] = {"dp1", "dp2"};
    for (int i = 0; i < 2; ++i) {
        char nm[32];
        snprintf(nm, sizeof nm, "%s", dpn[i]); tap_alloc(o, nm, T * FW * CH);
        snprintf(nm, sizeof nm, "%s.intra_rnn", dpn[i]); tap_alloc(o, nm, T * FW * CH);
        snprintf(nm, sizeof nm, "%s.intra_ln", dpn[i]); tap_alloc(o, nm, T * FW * CH);
        snprintf(nm, sizeof nm, "%s.inter_rnn", dpn[i]); tap_alloc(o, nm, T * FW * CH);
        snprintf(nm, sizeof nm, "%s.inter_ln", dpn[i]); tap_alloc(o, nm, T * FW * CH);
    }
    tap_alloc(o, "d0", 16 * T * FW);
    tap_alloc(o, "d1", 16 * T * FW);
    tap_alloc(o, "d2", 16 * T * FW);
    tap_alloc(o, "d3", 16 * T * 65);
    tap_alloc(o, "d4", 2 * T * FERB);
    tap_alloc(o, "spec_enh", 2 * FBINS * T);
    tap_alloc(o, "wave_f32", (size_t)o->out_len);
    *out = o;
    return 0;
}

void ade_oracle_destroy(ade_oracle* o) {
    if (!o) return;
    for (int i = 0; i < o->ntaps; ++i) free(o->taps[i].data);
    free(o->stft_kernel);
    free(o->istft_kernel);
    free(o->win_sum);
    free(o->blob.t);
    free(o->blob.storage);
    free(o);
}

/* TEST KNOB (not reference behaviour): rebuild the DFT tables from exactly-reduced double-precision angles.  The
 * reference evaluates cos/sin of fp32 angles up to ~1600 rad (STFT_Process.py:215-222), which makes its own STFT
 * ~4e-5 relative-inexact; the HIP path uses an exact FFT.  With this knob on, oracle and HIP path agree to fp32
 * round-off at EVERY tap, which separates kernel bugs from that known table error. */
void ade_oracle_set_exact_dft(ade_oracle* o, int exact) {
    build_stft_kernel(o->stft_kernel, o->window, NFFT, exact);
    build_istft_kernel(o->istft_kernel, o->window, NFFT, exact);
}

int ade_oracle_in_len(const ade_oracle* o) { return o->in_len; }
int ade_oracle_out_len(const ade_oracle* o) { return o->out_len; }

#define TAP(o, on, name) ((on) ? tap_get((o), (name)) : NULL)

/* One reference call: GTCRN_CUSTOM.forward (Export_GTCRN.py:636-693) with B = 1. */
/* `fin` (instead of `in`): the call's waveform as final fp32 samples at the model rate -- scaled and centred already (the input sandwich of GTCRN_CUSTOM,
 * Export_GTCRN.py:636-655, restated in numpy by the tests) -- used as it is.  `dyn`: the dynamic-length ISTFT trim, 256 T samples divided by the conv_transpose of the
 * squared window over the frames that exist (STFT_Process.py:337-341); out_f32 then holds 256 T samples and out_pcm must be NULL. */
static void process_one_ex(ade_oracle* o, const int16_t* in, const float* fin, int16_t* out_pcm, float* out_f32, int taps_on, const float* call_mean, int dyn) {
    const int L = o->in_len, T = o->T;
    const size_t TT = (size_t)T;
    float* audio = (float*)malloc(sizeof(float) * (size_t)L);
    float* xp = (float*)malloc(sizeof(float) * (size_t)(L + NFFT));
    float* spec = (float*)malloc(sizeof(float) * 2 * FBINS * TT);
    float* feat = (float*)malloc(sizeof(float) * 3 * TT * FBINS);
    float* feat_e = (float*)malloc(sizeof(float) * 3 * TT * FERB);
    float* feat_s = (float*)malloc(sizeof(float) * 9 * TT * FERB);
    float* e[5];
    e[0] = (float*)malloc(sizeof(float) * 16 * TT * 65);
    for (int i = 1; i < 5; ++i) e[i] = (float*)malloc(sizeof(float) * 16 * TT * FW);
    float* a = (float*)malloc(sizeof(float) * 16 * TT * FW);
    float* bb = (float*)malloc(sizeof(float) * 16 * TT * FW);
    float* d3 = (float*)malloc(sizeof(float) * 16 * TT * 65);
    float* d3in = (float*)malloc(sizeof(float) * 16 * TT * 65);
    float* m = (float*)malloc(sizeof(float) * 2 * TT * FERB);
    float* mfull = (float*)malloc(sizeof(float) * 2 * TT * FBINS);
    float* enh = (float*)malloc(sizeof(float) * 2 * FBINS * TT);
    float* raw = (float*)malloc(sizeof(float) * (size_t)(NFFT + HOP * (T - 1)));
    const int keep = dyn ? HOP * T : o->out_len;
    float* wave = (float*)malloc(sizeof(float) * (size_t)keep);

    /* F1: int16 -> f32, * 1/32768, minus the mean of THIS call (Export_GTCRN.py:637,645-647) */
    if (fin) {
        memcpy(audio, fin, sizeof(float) * (size_t)L);
    } else {
        const float inv = (float)(1.0 / 32768.0);
        double s = 0.0;
        for (int i = 0; i < L; ++i) { audio[i] = (float)in[i] * inv; s += audio[i]; }
        /* batch-fold (Export_GTCRN.py:647,656-660): the mean was taken over the whole call BEFORE the fold into windows */
        const float mean = call_mean ? *call_mean : (float)(s / L);
        for (int i = 0; i < L; ++i) audio[i] -= mean;
    }
    if (taps_on) memcpy(tap_get(o, "audio_f32"), audio, sizeof(float) * (size_t)L);
    /* F2-F3 */
    stft_packed_one(audio, L, o->stft_kernel, NFFT, HOP, 1, 1, xp, spec, T);
    if (taps_on) memcpy(tap_get(o, "spec"), spec, sizeof(float) * 2 * FBINS * TT);
    /* F4: forward_packed (Export_GTCRN.py:592-596): mag = sqrt(re^2+im^2+1e-12); feat = [mag,re,im] -> (3,T,F) (:567) */
    for (int f = 0; f < FBINS; ++f)
        for (int t = 0; t < T; ++t) {
            const float re = spec[(size_t)f * T + t], im = spec[(size_t)(FBINS + f) * T + t];
            const float mag = sqrtf((re * re + im * im) + 1e-12f);
            feat[((size_t)0 * T + t) * FBINS + f] = mag;
            feat[((size_t)1 * T + t) * FBINS + f] = re;
            feat[((size_t)2 * T + t) * FBINS + f] = im;
        }
    /* F5: ERB.bm (Export_GTCRN.py:99-102): bins [0,65) pass, bins [65,257) @ erb_weight_t (192x64) */
    for (int c = 0; c < 3; ++c)
        for (int t = 0; t < T; ++t) {
            const float* fr = feat + ((size_t)c * T + t) * FBINS;
            float* er = feat_e + ((size_t)c * T + t) * FERB;
            memcpy(er, fr, sizeof(float) * ERB_LOW);
            for (int j = 0; j < ERB_BANDS; ++j) {
                float s = 0.0f;
                for (int k = 0; k < ERB_HIGH; ++k) s += fr[ERB_LOW + k] * o->erb_w_t[(size_t)k * ERB_BANDS + j];
                er[ERB_LOW + j] = s;
            }
        }
    if (taps_on) memcpy(tap_get(o, "feat_erb"), feat_e, sizeof(float) * 3 * TT * FERB);
    /* F6: SFE(3) (Export_GTCRN.py:117-141): out[c*3+o, t, f] = in[c, t, f-1+o], zero outside */
    for (int c = 0; c < 3; ++c)
        for (int o3 = 0; o3 < 3; ++o3)
            for (int t = 0; t < T; ++t)
                for (int f = 0; f < FERB; ++f) {
                    const int ff = f - 1 + o3;
                    feat_s[((size_t)(c * 3 + o3) * T + t) * FERB + f] =
                        (ff >= 0 && ff < FERB) ? feat_e[((size_t)c * T + t) * FERB + ff] : 0.0f;
                }
    /* Encoder (Export_GTCRN.py:499-505) */
    conv_block(&o->en0, feat_s, 9, 16, 1, T, FERB, e[0]);
    conv_block(&o->en1, e[0], 16, 16, 2, T, 65, e[1]);
    for (int i = 0; i < 3; ++i) {
        const int tp = taps_on && i == 0;
        gtconv_block(&o->en_gt[i], e[i + 1], T, FW, e[i + 2], TAP(o, tp, "e2.pw1"), TAP(o, tp, "e2.dw"),
                     TAP(o, tp, "e2.pw2"), TAP(o, tp, "e2.tra_gru"), TAP(o, tp, "e2.tra"));
    }
    if (taps_on) {
        memcpy(tap_get(o, "e0"), e[0], sizeof(float) * 16 * TT * 65);
        static const char* en[4] = {"e1", "e2", "e3", "e4"};
        for (int i = 0; i < 4; ++i) memcpy(tap_get(o, en[i]), e[i + 1], sizeof(float) * 16 * TT * FW);
    }
    /* permute (C,T,F) -> (T,F,C) (Export_GTCRN.py:576), 2x DPGRNN, permute back (:579) */
    for (int c = 0; c < 16; ++c)
        for (int t = 0; t < T; ++t)
            for (int f = 0; f < FW; ++f) a[((size_t)t * FW + f) * CH + c] = e[4][((size_t)c * T + t) * FW + f];
    dpgrnn(&o->dp[0], a, T, bb, TAP(o, taps_on, "dp1.intra_rnn"), TAP(o, taps_on, "dp1.intra_ln"),
           TAP(o, taps_on, "dp1.inter_rnn"), TAP(o, taps_on, "dp1.inter_ln"));
    if (taps_on) memcpy(tap_get(o, "dp1"), bb, sizeof(float) * TT * FW * CH);
    dpgrnn(&o->dp[1], bb, T, a, TAP(o, taps_on, "dp2.intra_rnn"), TAP(o, taps_on, "dp2.intra_ln"),
           TAP(o, taps_on, "dp2.inter_rnn"), TAP(o, taps_on, "dp2.inter_ln"));
    if (taps_on) memcpy(tap_get(o, "dp2"), a, sizeof(float) * TT * FW * CH);
    for (int c = 0; c < 16; ++c)
        for (int t = 0; t < T; ++t)
            for (int f = 0; f < FW; ++f) bb[((size_t)c * T + t) * FW + f] = a[((size_t)t * FW + f) * CH + c];
    /* Decoder (Export_GTCRN.py:523-529): x = de[i](x + en_outs[4-i]) */
    float* x = bb;
    float* y = a;
    static const char* dn[3] = {"d0", "d1", "d2"};
    for (int i = 0; i < 3; ++i) {
        const float* sk = e[4 - i];
        for (size_t q = 0; q < 16 * TT * FW; ++q) x[q] += sk[q];
        const int tp = taps_on && i == 0;
        gtconv_block(&o->de_gt[i], x, T, FW, y, TAP(o, tp, "d0.pw1"), TAP(o, tp, "d0.dw"), TAP(o, tp, "d0.pw2"), NULL, NULL);
        if (taps_on) memcpy(tap_get(o, dn[i]), y, sizeof(float) * 16 * TT * FW);
        float* tmp = x; x = y; y = tmp;
    }
    for (size_t q = 0; q < 16 * TT * FW; ++q) x[q] += e[1][q];
    deconv_block(&o->de3, x, 16, 16, 2, T, FW, d3);
    if (taps_on) memcpy(tap_get(o, "d3"), d3, sizeof(float) * 16 * TT * 65);
    for (size_t q = 0; q < 16 * TT * 65; ++q) d3in[q] = d3[q] + e[0][q];
    deconv_block(&o->de4, d3in, 16, 2, 1, T, 65, m);
    if (taps_on) memcpy(tap_get(o, "d4"), m, sizeof(float) * 2 * TT * FERB);
    /* F12: ERB.bs (Export_GTCRN.py:104-107) + transpose + complex ratio mask (:583-590) */
    for (int c = 0; c < 2; ++c)
        for (int t = 0; t < T; ++t) {
            const float* mr = m + ((size_t)c * T + t) * FERB;
            float* fr = mfull + ((size_t)c * T + t) * FBINS;
            memcpy(fr, mr, sizeof(float) * ERB_LOW);
            for (int k = 0; k < ERB_HIGH; ++k) {
                float s = 0.0f;
                for (int j = 0; j < ERB_BANDS; ++j) s += mr[ERB_LOW + j] * o->ierb_w_t[(size_t)j * ERB_HIGH + k];
                fr[ERB_LOW + k] = s;
            }
        }
    for (int f = 0; f < FBINS; ++f)
        for (int t = 0; t < T; ++t) {
            const float re = spec[(size_t)f * T + t], im = spec[(size_t)(FBINS + f) * T + t];
            const float m0 = mfull[((size_t)0 * T + t) * FBINS + f], m1 = mfull[((size_t)1 * T + t) * FBINS + f];
            enh[(size_t)f * T + t] = re * m0 - im * m1;
            enh[(size_t)(FBINS + f) * T + t] = im * m0 + re * m1;
        }
    if (taps_on) memcpy(tap_get(o, "spec_enh"), enh, sizeof(float) * 2 * FBINS * TT);
    /* F13 */
    if (dyn) {
        /* win_sum = conv_transpose1d(ones(T), w^2, stride = hop)[n_fft / 2 : n_fft / 2 + 256 T] */
        float* ws = (float*)calloc((size_t)(NFFT + HOP * (T - 1)), sizeof(float));
        for (int t = 0; t < T; ++t)
            for (int n = 0; n < NFFT; ++n) ws[(size_t)t * HOP + n] += o->window[n] * o->window[n];
        istft_packed_one(enh, T, o->istft_kernel, NFFT, HOP, ws + NFFT / 2, NFFT / 2, keep, raw, wave);
        free(ws);
    } else {
        istft_packed_one(enh, T, o->istft_kernel, NFFT, HOP, o->win_sum, NFFT / 2, o->out_len, raw, wave);
    }
    if (taps_on && !dyn) memcpy(tap_get(o, "wave_f32"), wave, sizeof(float) * (size_t)o->out_len);
    if (out_f32) memcpy(out_f32, wave, sizeof(float) * (size_t)keep);
    /* F14: * 32767, clamp, truncating cast (Export_GTCRN.py:681,690) */
    if (out_pcm)
        for (int i = 0; i < o->out_len; ++i) {
            float v = wave[i] * 32767.0f;
            v = v < -32768.0f ? -32768.0f : (v > 32767.0f ? 32767.0f : v);
            out_pcm[i] = (int16_t)v;
        }
    free(audio); free(xp); free(spec); free(feat); free(feat_e); free(feat_s);
    for (int i = 0; i < 5; ++i) free(e[i]);
    free(a); free(bb); free(d3); free(d3in); free(m); free(mfull); free(enh); free(raw); free(wave);
}

static void process_one(ade_oracle* o, const int16_t* in, int16_t* out_pcm, float* out_f32, int taps_on, const float* call_mean) {
    process_one_ex(o, in, NULL, out_pcm, out_f32, taps_on, call_mean, 0);
}

/* The network between the two halves of GTCRN_CUSTOM's sandwich: rows of in_len final fp32 samples at the model rate in, the normalised waveform out --
 * out_len samples per row (static export) or 256 T (dynamic_tail: the dynamic_axes export's ISTFT trim). */
int ade_oracle_process_model_f32(ade_oracle* o, const float* in, int B, float* out_f32, int dynamic_tail) {
    if (!o || !in || !out_f32 || B < 0) return fail("bad arguments", NULL);
    const int keep = dynamic_tail ? HOP * o->T : o->out_len;
    for (int b = 0; b < B; ++b) process_one_ex(o, NULL, in + (size_t)b * o->in_len, NULL, out_f32 + (size_t)b * keep, 0, NULL, dynamic_tail);
    return 0;
}

int ade_oracle_process(ade_oracle* o, const int16_t* in, int B, int16_t* out_pcm, float* out_f32, int n_threads) {
    if (!o || !in || B < 0) return fail("bad arguments", NULL);
#ifdef _OPENMP
    if (n_threads > 1) {
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads)
        for (int b = 0; b < B; ++b)
            process_one(o, in + (size_t)b * o->in_len, out_pcm ? out_pcm + (size_t)b * o->out_len : NULL,
                        out_f32 ? out_f32 + (size_t)b * o->out_len : NULL, b == 0, NULL);
        return 0;
    }
#endif
    (void)n_threads;
    for (int b = 0; b < B; ++b)
        process_one(o, in + (size_t)b * o->in_len, out_pcm ? out_pcm + (size_t)b * o->out_len : NULL,
                    out_f32 ? out_f32 + (size_t)b * o->out_len : NULL, b == 0, NULL);
    return 0;
}

/* USE_BATCH_FOLD=True exports (Export_GTCRN.py:41-45,656-660,671-672): one call = n_win windows of in_len samples,
 * DC mean over the WHOLE call, windows processed as a batch and stitched back (in_len must be a multiple of the hop,
 * so out_len == in_len per window). */
int ade_oracle_process_fold(ade_oracle* o, const int16_t* in, int n_calls, int n_win, int16_t* out_pcm, float* out_f32, int n_threads) {
    if (!o || !in || n_calls < 0 || n_win < 1) return fail("bad arguments", NULL);
    if (o->out_len != o->in_len) return fail("fold needs a window length that is a multiple of the hop", NULL);
    (void)n_threads;
    for (int c = 0; c < n_calls; ++c) {
        const int16_t* base = in + (size_t)c * n_win * o->in_len;
        const float inv = (float)(1.0 / 32768.0);
        double s = 0.0;
        for (size_t i = 0; i < (size_t)n_win * o->in_len; ++i) s += (float)base[i] * inv;
        const float mean = (float)(s / ((double)n_win * o->in_len));
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads > 1 ? n_threads : 1)
#endif
        for (int w = 0; w < n_win; ++w) {
            const size_t off = ((size_t)c * n_win + w);
            process_one(o, in + off * o->in_len, out_pcm ? out_pcm + off * o->out_len : NULL,
                        out_f32 ? out_f32 + off * o->out_len : NULL, c == 0 && w == 0, &mean);
        }
    }
    return 0;
}
