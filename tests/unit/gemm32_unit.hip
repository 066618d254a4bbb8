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
