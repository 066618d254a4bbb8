// Unit check of csrc/ade_gemm.h (the 128 x 128 fp32 tile on v_mfma_f32_16x16x4_f32): C = A B against a double-precision host product, over the operand fetch paths
// (16-byte loads along k, scalar loads along k / along n / along m), the K tail of the last slab and partial tiles in M and N.  Host simulator and gfx950.
//   gemm32_unit M N K [M N K ...]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../audio_denoiser_onnx_amd/csrc/ade_gemm.h"

using namespace ade::gemm;

struct ColMajorA {      // A(m, k) = p[k * ld + m]: consecutive lanes = consecutive rows (the fetch path that is not along k)
    static constexpr bool kAlongK = false;
    const float* p;
    int ld;
    __device__ float operator()(int m, int k) const { return p[(size_t)k * ld + m]; }
};
struct PlainC {         // C(m, n) -> p[m * ld + n]
    float* p;
    int ld;
    __device__ void operator()(int m, int n, float v) const { p[(size_t)m * ld + n] = v; }
};

struct ScaleC4 {        // the float4 store form (gemm::HasV4: transposed tiles): C(m, n) = v * scale[m] + bias[n] + old -> p[m * ld + n]; scalar form for N % 4 != 0 / odd pitches
    static constexpr bool kCtx = true;
    float* p;
    const float *scale, *bias;
    int ld;
    __device__ float row(int m) const { return scale[m]; }
    __device__ float col(int n) const { return bias[n]; }
    __device__ float pre(int m, int n, float) const { return p[(size_t)m * ld + n]; }
    __device__ void operator()(int m, int n, float v, float sc, float b, float old) const { p[(size_t)m * ld + n] = old + (v * sc + b); }
    static constexpr bool kV4 = true;
    __host__ __device__ bool can_v4(int N) const { return !((N | ld) & 3) && !(((size_t)p | (size_t)bias) & 15); }
    __device__ float4 col4(int n) const { return *reinterpret_cast<const float4*>(bias + n); }
    __device__ float4 pre4(int m, int n, float) const { return *reinterpret_cast<const float4*>(p + (size_t)m * ld + n); }
    __device__ void store4(int m, int n, float4 v, float sc, const float4& b, const float4& old) const {
        *reinterpret_cast<float4*>(p + (size_t)m * ld + n) = make_float4(old.x + (v.x * sc + b.x), old.y + (v.y * sc + b.y), old.z + (v.z * sc + b.z), old.w + (v.w * sc + b.w));
    }
};

static int run_case(int M, int N, int K) {
    // row-major copies with a pitch that allows (K % 4 == 0) or forbids the 16-byte path, plus transposed copies for the other two paths
    const int lda = K + (K % 4 == 0 ? 4 : 1), ldb = K + (K % 4 == 0 ? 8 : 3), ldc = N + 5;
    std::vector<float> A((size_t)M * lda), Bnk((size_t)N * ldb), At((size_t)K * (M + 2)), Bkn((size_t)K * (N + 1));
    unsigned s = 777u + M * 5 + N * 11 + K;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
    for (int m = 0; m < M; ++m) for (int k = 0; k < lda; ++k) A[(size_t)m * lda + k] = rnd();
    for (int n = 0; n < N; ++n) for (int k = 0; k < ldb; ++k) Bnk[(size_t)n * ldb + k] = rnd() * 0.5f;
    for (int m = 0; m < M; ++m) for (int k = 0; k < K; ++k) At[(size_t)k * (M + 2) + m] = A[(size_t)m * lda + k];
    for (int n = 0; n < N; ++n) for (int k = 0; k < K; ++k) Bkn[(size_t)k * (N + 1) + n] = Bnk[(size_t)n * ldb + k];
    std::vector<double> ref((size_t)M * N);
    for (int m = 0; m < M; ++m)
        for (int n = 0; n < N; ++n) {
            double r = 0.0;
            for (int k = 0; k < K; ++k) r += (double)A[(size_t)m * lda + k] * (double)Bnk[(size_t)n * ldb + k];
            ref[(size_t)m * N + n] = r;
        }
    float *dA, *dB, *dAt, *dBkn, *dC;
    hipMalloc((void**)&dA, A.size() * 4); hipMalloc((void**)&dB, Bnk.size() * 4); hipMalloc((void**)&dAt, At.size() * 4); hipMalloc((void**)&dBkn, Bkn.size() * 4);
    hipMalloc((void**)&dC, (size_t)M * ldc * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, Bnk.data(), Bnk.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dAt, At.data(), At.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dBkn, Bkn.data(), Bkn.size() * 4, hipMemcpyHostToDevice);
    std::vector<float> C((size_t)M * ldc);
    const double tol = 3e-6 * K + 1e-5;
    int rc = 0;
    auto check = [&](const char* what) {
        hipDeviceSynchronize();
        hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
        double worst = 0.0;
        int bad_pad = 0;
        for (int m = 0; m < M; ++m)
            for (int n = 0; n < ldc; ++n) {
                if (n >= N) { unsigned u; memcpy(&u, &C[(size_t)m * ldc + n], 4); bad_pad += u != 0xffffffffu; continue; }
                const double d = fabs(ref[(size_t)m * N + n] - (double)C[(size_t)m * ldc + n]);
                worst = d == d ? fmax(worst, d) : 1e30;
            }
        const bool ok = worst <= tol && bad_pad == 0;
        printf("gemm32 M=%d N=%d K=%d %-34s max|d| %.3e (tol %.1e), touched padding %d -> %s\n", M, N, K, what, worst, tol, bad_pad, ok ? "OK" : "FAIL");
        rc |= ok ? 0 : 1;
    };
    hipMemset(dC, 0xff, C.size() * 4);
    launch((hipStream_t)0, RowMajorA{dA, lda}, WeightNK{dB, ldb}, PlainC{dC, ldc}, M, N, K);
    check(K % 4 == 0 ? "A, B along k (16-byte loads)" : "A, B along k (scalar loads)");
    hipMemset(dC, 0xff, C.size() * 4);
    launch((hipStream_t)0, RowMajorA{dA, lda}, RowMajorB{dBkn, N + 1}, PlainC{dC, ldc}, M, N, K);
    check("A along k, B along n");
    hipMemset(dC, 0xff, C.size() * 4);
    launch((hipStream_t)0, ColMajorA{dAt, M + 2}, WeightNK{dB, ldb}, PlainC{dC, ldc}, M, N, K);
    check("A along m, B along k");
    {   // the float4 store form against the scalar form of the SAME functor (gemm::NoV4): bit-identical, on a pitch that allows it (N % 4 == 0) and on one that does not
        const int ld4 = ((N + 3) & ~3) + 4;
        std::vector<float> sc(M), bi(N + 4), base((size_t)M * ld4);
        for (auto& v : sc) v = rnd() + 1.5f;
        for (auto& v : bi) v = rnd();
        for (auto& v : base) v = rnd();
        float *dS, *dBi, *dX, *dY;
        hipMalloc((void**)&dS, M * 4); hipMalloc((void**)&dBi, bi.size() * 4); hipMalloc((void**)&dX, base.size() * 4); hipMalloc((void**)&dY, base.size() * 4);
        hipMemcpy(dS, sc.data(), M * 4, hipMemcpyHostToDevice); hipMemcpy(dBi, bi.data(), bi.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dX, base.data(), base.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dY, base.data(), base.size() * 4, hipMemcpyHostToDevice);
        const ScaleC4 st4{dX, dS, dBi, ld4};
        launch((hipStream_t)0, RowMajorA{dA, lda}, WeightNK{dB, ldb}, st4, M, N, K);                                   // float4 form when N % 4 == 0, else launch()'s own fallback
        launch((hipStream_t)0, RowMajorA{dA, lda}, WeightNK{dB, ldb}, NoV4<ScaleC4>{ScaleC4{dY, dS, dBi, ld4}}, M, N, K);   // scalar form
        hipDeviceSynchronize();
        std::vector<float> X(base.size()), Y(base.size());
        hipMemcpy(X.data(), dX, X.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(Y.data(), dY, Y.size() * 4, hipMemcpyDeviceToHost);
        size_t diff = 0, untouched_bad = 0;
        double worst = 0.0;
        for (int m = 0; m < M; ++m)
            for (int n = 0; n < ld4; ++n) {
                const size_t i = (size_t)m * ld4 + n;
                if (memcmp(&X[i], &Y[i], 4)) ++diff;
                if (n >= N) { untouched_bad += memcmp(&X[i], &base[i], 4) != 0; continue; }
                const double want = (double)base[i] + (ref[(size_t)m * N + n] * (double)sc[m] + (double)bi[n]);
                worst = fmax(worst, fabs(want - (double)X[i]));
            }
        const bool ok = diff == 0 && untouched_bad == 0 && worst <= 3.0 * tol;
        printf("gemm32 M=%d N=%d K=%d %-34s max|d| %.3e, vs scalar form %zu words differ, touched padding %zu -> %s\n", M, N, K,
               st4.can_v4(N) ? "float4 store form (transposed)" : "float4 store: scalar fallback", worst, diff, untouched_bad, ok ? "OK" : "FAIL");
        rc |= ok ? 0 : 1;
        hipFree(dS); hipFree(dBi); hipFree(dX); hipFree(dY);
    }
    hipFree(dA); hipFree(dB); hipFree(dAt); hipFree(dBkn); hipFree(dC);
    return rc;
}

static void time_case(int M, int N, int K) {      // "-t M N K": TFLOP/s of the plain-store product, both operands along k with 16-byte loads (GPU builds)
    float *dA, *dB, *dC;
    hipMalloc((void**)&dA, (size_t)M * K * 4); hipMalloc((void**)&dB, (size_t)N * K * 4); hipMalloc((void**)&dC, (size_t)M * N * 4);
    hipMemset(dA, 0x3c, (size_t)M * K * 4); hipMemset(dB, 0x3c, (size_t)N * K * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 2; ++it) launch((hipStream_t)0, RowMajorA{dA, K}, WeightNK{dB, K}, PlainC{dC, N}, M, N, K);
    const int reps = 5;
    hipEventRecord(e0, 0);
    for (int it = 0; it < reps; ++it) launch((hipStream_t)0, RowMajorA{dA, K}, WeightNK{dB, K}, PlainC{dC, N}, M, N, K);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    printf("gemm32 M=%d N=%d K=%d: plain fp32 store %.3f ms, %.1f TFLOP/s\n", M, N, K, ms / reps, 2.0 * M * N * K / (ms / reps * 1e-3) / 1e12);
    hipFree(dA); hipFree(dB); hipFree(dC);
}

int main(int argc, char** argv) {
    int rc = 0;
    if (argc >= 5 && !strcmp(argv[1], "-t")) {
        for (int i = 2; i + 2 < argc; i += 3) time_case(atoi(argv[i]), atoi(argv[i + 1]), atoi(argv[i + 2]));
        return 0;
    }
    if (argc < 4) return run_case(130, 70, 37);
    for (int i = 1; i + 2 < argc; i += 3) rc |= run_case(atoi(argv[i]), atoi(argv[i + 1]), atoi(argv[i + 2]));
    return rc;
}
