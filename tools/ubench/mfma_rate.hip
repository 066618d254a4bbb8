// Micro-benchmark: issue rate of v_mfma_f32_16x16x4_f32 / v_mfma_f32_32x32x2_f32 per SIMD with 1..4 wavefronts per SIMD,
// as dependent chains of 6 (the shape of the GTConvBlock pointwise tile) or 4 independent accumulators.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
template <int MODE> __global__ void k(float* out, int iters, long long* cyc) {
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
    v4f d0 = {0, 0, 0, 0}, d1 = d0, d2 = d0, d3 = d0;
    v16f e0 = {0};
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {   // one dependent chain of 6
#pragma unroll
            for (int s = 0; s < 6; ++s) d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, d0, 0, 0, 0);
        } else if (MODE == 1) {   // 4 independent accumulators, 6 rounds... 24 instrs
#pragma unroll
            for (int s = 0; s < 6; ++s) {
                d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, d1, 0, 0, 0);
                d2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, d2, 0, 0, 0);
                d3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, d3, 0, 0, 0);
            }
        } else {   // 32x32x2 chain of 6
#pragma unroll
            for (int s = 0; s < 6; ++s) e0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, e0, 0, 0, 0);
        }
    }
    const long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = d0[0] + d1[1] + d2[2] + d3[3] + e0[5];
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 8);
    const int iters = 2000;
    const char* names[3] = {"16x16x4 chain6", "16x16x4 4 acc ", "32x32x2 chain6"};
    const int per_iter[3] = {6, 24, 6};
    for (int mode = 0; mode < 3; ++mode)
        for (int wps = 1; wps <= 4; wps *= 2) {
            const int threads = 256 * wps;   // 4 SIMDs x wps waves
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            float ms = 0.f;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0, 0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(threads), 0, 0, out, iters, cyc);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(threads), 0, 0, out, iters, cyc);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(threads), 0, 0, out, iters, cyc);
                hipEventRecord(e1, 0);
                hipDeviceSynchronize();
                hipEventElapsedTime(&ms, e0, e1);
            }
            long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            const double n_mfma_per_simd = (double)iters * per_iter[mode] * wps;
            const double flop = (mode == 2 ? 2.0 * 32 * 32 * 2 : 2.0 * 16 * 16 * 4) * n_mfma_per_simd * 4 * 256;
            printf("%s  waves/SIMD=%d : %.1f clock64 ticks per MFMA per wave ; kernel %.3f ms -> %.1f ns per MFMA per SIMD, %.1f TFLOP/s\n",
                   names[mode], wps, (double)c / (iters * per_iter[mode]), ms, ms * 1e6 / n_mfma_per_simd, flop / (ms * 1e-3) / 1e12);
        }
    return 0;
}
