// Micro-benchmark: what bounds csrc/ade_gemm.h's k_gemm128?  Same kernel with (a) real row-major operands, (b) operands
// that cost no memory traffic (constant functors): the gap is the operand-fetch path, the rest is LDS + MFMA structure.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../audio_denoiser_onnx_amd/csrc/ade_gemm.h"
using namespace ade::gemm;
struct ConstA { static constexpr bool kAlongK = true; float v; __device__ float operator()(int m, int k) const { return v + (float)((m ^ k) & 3); } };
struct ConstB { static constexpr bool kAlongN = true; float v; __device__ float operator()(int k, int n) const { return v - (float)((n + k) & 1); } };
int main() {
    const int M = 4096, N = 24576, K = 2048;
    float *A, *B, *C;
    hipMalloc(&A, (size_t)M * K * 4); hipMalloc(&B, (size_t)K * N * 4); hipMalloc(&C, (size_t)M * N * 4);
    hipMemset(A, 0, (size_t)M * K * 4); hipMemset(B, 0, (size_t)K * N * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode) {
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0, 0);
            if (mode == 0) launch((hipStream_t)0, RowMajorA{A, K}, RowMajorB{B, N}, BiasActStore<kActNone>{C, N, nullptr, 0.f}, M, N, K);
            else launch((hipStream_t)0, ConstA{1.f}, ConstB{2.f}, BiasActStore<kActNone>{C, N, nullptr, 0.f}, M, N, K);
            hipEventRecord(e1, 0); hipDeviceSynchronize(); hipEventElapsedTime(&ms, e0, e1);
        }
        printf("%s: %.3f ms  %.1f TFLOP/s\n", mode == 0 ? "row-major operands from HBM/L2" : "constant operands (no fetch)  ", ms, 2.0 * M * N * K / (ms * 1e-3) / 1e12);
    }
    return 0;
}
