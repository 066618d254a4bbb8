// Unit check of csrc/ade_gemm16.h (TEST INFRASTRUCTURE): C = A B^T on bf16 operands against a double-precision host product.
// Built two ways by tests/test_gemm16.py: g++ + tests/hipsim (CPU, small shapes) and hipcc --offload-arch=gfx950 (GPU).
//   usage: gemm16_unit M N K [M N K ...]      exit status 0 = every case within tolerance
#include <hip/hip_runtime.h>

#include "../../audio_denoiser_onnx_amd/csrc/ade_gemm16.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

using namespace ade::gemm16;

struct PlainStore {            // C[m][n] = v (fp32, row-major) and Cb[m][n] = bf16(v + bias[n])
    float* c;
    bf16_t* cb;
    const float* bias;
    int ld;
    __device__ float4 col(int n, int cnt) const { return load_f32x4(bias + n, cnt); }
    __device__ void operator()(int m, int n, float4 v, int cnt, const float4& b) const {
        store_f32x4(c + (size_t)m * ld + n, v, cnt);
        store_bf16x4(cb + (size_t)m * ld + n, make_float4(v.x + b.x, v.y + b.y, v.z + b.z, v.w + b.w), cnt);
    }
};

static unsigned short to_bf16(float x) { unsigned u; memcpy(&u, &x, 4); u += 0x7fffu + ((u >> 16) & 1u); return (unsigned short)(u >> 16); }
static float from_bf16(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }

static int run_case(int M, int N, int K) {
    const int lda = K + 8, ldb = K, ldc = ((N + 3) / 4) * 4 + 4;
    std::vector<unsigned short> A((size_t)M * lda), B((size_t)N * ldb);
    std::vector<float> bias(ldc);
    unsigned s = 12345u + M * 7 + N * 3 + K;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
    for (auto& v : A) v = to_bf16(rnd());
    for (auto& v : B) v = to_bf16(rnd() * 0.5f);
    for (auto& v : bias) v = rnd();
    unsigned short *dA, *dB, *dCb;
    float *dC, *dbias;
    hipMalloc((void**)&dA, A.size() * 2); hipMalloc((void**)&dB, B.size() * 2);
    hipMalloc((void**)&dC, (size_t)M * ldc * 4); hipMalloc((void**)&dCb, (size_t)M * ldc * 2); hipMalloc((void**)&dbias, ldc * 4);
    hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dbias, bias.data(), ldc * 4, hipMemcpyHostToDevice);
    hipMemset(dC, 0xff, (size_t)M * ldc * 4); hipMemset(dCb, 0xff, (size_t)M * ldc * 2);
    launch((hipStream_t)0, dA, lda, dB, ldb, PlainStore{dC, dCb, dbias, ldc}, M, N, K);
    hipDeviceSynchronize();
    std::vector<float> C((size_t)M * ldc);
    std::vector<unsigned short> Cb((size_t)M * ldc);
    hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(Cb.data(), dCb, Cb.size() * 2, hipMemcpyDeviceToHost);
    double worst = 0.0, worst_b = 0.0;
    int bad_pad = 0;
    for (int m = 0; m < M; ++m) {
        for (int n = 0; n < ldc; ++n) {
            if (n >= N) {       // padding columns must be untouched
                unsigned u; memcpy(&u, &C[(size_t)m * ldc + n], 4);
                if (u != 0xffffffffu || Cb[(size_t)m * ldc + n] != 0xffff) ++bad_pad;
                continue;
            }
            double ref = 0.0;
            for (int k = 0; k < K; ++k) ref += (double)from_bf16(A[(size_t)m * lda + k]) * (double)from_bf16(B[(size_t)n * ldb + k]);
            worst = fmax(worst, fabs(ref - (double)C[(size_t)m * ldc + n]));
            worst_b = fmax(worst_b, fabs(ref + bias[n] - (double)from_bf16(Cb[(size_t)m * ldc + n])) / (1.0 + fabs(ref + bias[n])));
        }
    }
    // the same product with a scale per row (launch's row_scale: the tile keeps its 128 scales in LDS); C2(m, n) must be C(m, n) * scale[m] to an ulp
    std::vector<float> scale(M);
    for (auto& v : scale) v = 0.25f + fabsf(rnd());
    float* dscale;
    hipMalloc((void**)&dscale, (size_t)M * 4);
    hipMemcpy(dscale, scale.data(), (size_t)M * 4, hipMemcpyHostToDevice);
    launch((hipStream_t)0, dA, lda, dB, ldb, PlainStore{dC, dCb, dbias, ldc}, M, N, K, dscale);
    hipDeviceSynchronize();
    std::vector<float> C2((size_t)M * ldc);
    hipMemcpy(C2.data(), dC, C2.size() * 4, hipMemcpyDeviceToHost);
    int bad_scale = 0;
    for (int m = 0; m < M; ++m)
        for (int n = 0; n < N; ++n) bad_scale += C2[(size_t)m * ldc + n] != C[(size_t)m * ldc + n] * scale[m];
    hipFree(dscale);
    const double tol = 2e-6 * K + 1e-5;
    const bool ok = worst <= tol && worst_b <= 0.0045 && bad_pad == 0 && bad_scale == 0;
    printf("gemm16 M=%d N=%d K=%d: max|d| fp32 %.3e (tol %.1e), bf16 rel %.3e, touched padding %d, row-scale mismatches %d -> %s\n", M, N, K, worst, tol, worst_b, bad_pad, bad_scale,
           ok ? "OK" : "FAIL");
    hipFree(dA); hipFree(dB); hipFree(dC); hipFree(dCb); hipFree(dbias);
    return ok ? 0 : 1;
}

struct BfStore {               // the cheapest useful store: bf16(v)
    bf16_t* cb;
    int ld;
    __device__ NoCol col(int, int) const { return NoCol{}; }
    __device__ void operator()(int m, int n, float4 v, int cnt, NoCol) const { store_bf16x4(cb + (size_t)m * ld + n, v, cnt); }
};
template <int GELU>
struct BfBiasStore {
    bf16_t* cb;
    const float* bias;
    int ld;
    __device__ float4 col(int n, int cnt) const { return load_f32x4(bias + n, cnt); }
    __device__ void operator()(int m, int n, float4 v, int cnt, const float4& b) const {
        v2f lo = mk2(v.x, v.y) + mk2(b.x, b.y), hi = mk2(v.z, v.w) + mk2(b.z, b.w);
        if (GELU) { lo = gelu_pk(lo); hi = gelu_pk(hi); }
        store_bf16x4(cb + (size_t)m * ld + n, make_float4(lo[0], lo[1], hi[0], hi[1]), cnt);
    }
};
static void time_case(int M, int N, int K) {      // "-t M N K": TFLOP/s of the plain-store product (GPU builds)
    unsigned short *dA, *dB, *dCb; float *dC, *dbias;
    hipMalloc((void**)&dA, (size_t)M * K * 2); hipMalloc((void**)&dB, (size_t)N * K * 2); hipMalloc((void**)&dC, (size_t)M * N * 4); hipMalloc((void**)&dCb, (size_t)M * N * 2); hipMalloc((void**)&dbias, (size_t)N * 4);
    hipMemset(dA, 0x3c, (size_t)M * K * 2); hipMemset(dB, 0x3c, (size_t)N * K * 2); hipMemset(dbias, 0, (size_t)N * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 3; ++it) launch((hipStream_t)0, dA, K, dB, K, PlainStore{dC, dCb, dbias, N}, M, N, K);
    hipEventRecord(e0, 0);
    const int reps = getenv("GEMM16_REPS") ? atoi(getenv("GEMM16_REPS")) : 10;
    for (int it = 0; it < reps; ++it) launch((hipStream_t)0, dA, K, dB, K, PlainStore{dC, dCb, dbias, N}, M, N, K);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    printf("gemm16 M=%d N=%d K=%d: fp32 + bf16 store %.3f ms, %.1f TFLOP/s", M, N, K, ms / reps, 2.0 * M * N * K / (ms / reps * 1e-3) / 1e12);
    hipEventRecord(e0, 0);
    for (int it = 0; it < reps; ++it) launch((hipStream_t)0, dA, K, dB, K, BfStore{dCb, N}, M, N, K);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf(" | bf16 store only %.3f ms, %.1f TFLOP/s", ms / reps, 2.0 * M * N * K / (ms / reps * 1e-3) / 1e12);
    // what a model store adds, one piece at a time: a scale per row, a bias per column, the packed GELU
    float* dscale;
    hipMalloc((void**)&dscale, (size_t)M * 4);
    hipMemset(dscale, 0, (size_t)M * 4);
    auto timed = [&](auto st, const float* rs) {
        launch((hipStream_t)0, dA, K, dB, K, st, M, N, K, rs);
        hipEventRecord(e0, 0);
        for (int it = 0; it < reps; ++it) launch((hipStream_t)0, dA, K, dB, K, st, M, N, K, rs);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        return ms / reps;
    };
    const float t_rs = timed(BfStore{dCb, N}, dscale), t_b = timed(BfBiasStore<0>{dCb, dbias, N}, nullptr), t_g = timed(BfBiasStore<1>{dCb, dbias, N}, nullptr),
                t_all = timed(BfBiasStore<1>{dCb, dbias, N}, dscale);
    printf(" | + row scale %.3f | + bias %.3f | + bias + gelu %.3f | all three %.3f ms\n", t_rs, t_b, t_g, t_all);
    hipFree(dscale);
    hipFree(dA); hipFree(dB); hipFree(dC); hipFree(dCb); hipFree(dbias);
}

#if defined(HIPSIM)
// host-simulator build only (device helpers are plain functions there): the launch-constant division against the real one, and the erf used by the bf16 GELU
static int helper_checks() {
    int bad = 0;
    for (int d : {1, 2, 3, 7, 60, 64, 151, 801, 1000, 25632, 100000, 1537920}) {
        const FastDiv f = make_fastdiv(d);
        for (long long n = 0; n < (1ll << 31); n += (n < 70000 ? 1 : 9973)) bad += f.div((int)n) != (int)(n / d);
        for (long long n : {(1ll << 31) - 1, (long long)d * 1000 - 1, (long long)d * 1000, (long long)d * 1000 + 1}) if (n >= 0 && n < (1ll << 31)) bad += f.div((int)n) != (int)(n / d);
    }
    double worst = 0.0;
    for (int i = -80000; i <= 80000; ++i) { const float x = i * 1e-4f; worst = fmax(worst, fabs((double)erf_fast(x) - erf((double)x))); }
    double worst_g = 0.0, worst_in = 0.0;          // the packed GELU of the bf16 stores against the erf form
    for (int i = -100000; i <= 100000; ++i) {
        const float x = i * 1e-4f;
        const v2f g = gelu_pk(mk2(x, -x));
        const double e0 = fabs((double)g[0] - 0.5 * x * (1.0 + erf((double)x * 0.70710678118654752440))), e1 = fabs((double)g[1] - 0.5 * -x * (1.0 + erf((double)-x * 0.70710678118654752440)));
        worst_g = fmax(worst_g, fmax(e0, e1) / fmax(1.0, fabs((double)x)));
        if (fabsf(x) <= 4.0f) worst_in = fmax(worst_in, fmax(e0, e1));
    }
    const bool ok = bad == 0 && worst < 5e-7 && worst_in < 1.2e-4 && worst_g < 1.2e-4;
    printf("FastDiv mismatches %d; erf_fast max |error| %.2e over [-8, 8]; gelu_pk max |error| %.2e on [-4, 4], %.2e / max(1, |x|) on [-10, 10] -> %s\n", bad, worst, worst_in, worst_g,
           ok ? "OK" : "FAIL");
    return ok ? 0 : 1;
}
#endif

int main(int argc, char** argv) {
    int rc = 0;
#if defined(HIPSIM)
    rc |= helper_checks();
#endif
    if (argc >= 5 && argv[1][0] == '-' && argv[1][1] == 't') {
        for (int i = 2; i + 2 < argc; i += 3) time_case(atoi(argv[i]), atoi(argv[i + 1]), atoi(argv[i + 2]));
        return 0;
    }
    if (argc < 4) { rc |= run_case(130, 70, 72); return rc; }
    for (int i = 1; i + 2 < argc; i += 3) rc |= run_case(atoi(argv[i]), atoi(argv[i + 1]), atoi(argv[i + 2]));
    return rc;
}
