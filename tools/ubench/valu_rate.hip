// Micro-benchmark: issue cost of v_fma_f32 / v_pk_fma_f32 (VGPR vs SGPR weight operand) on gfx950, 1..8 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rate.hip -o gpurun_out/valu_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ void k(float* out, const float* w, int iters, long long* cyc) {
    float a[16];
    float2 p[8];
    for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 0.001f + i;
    for (int i = 0; i < 8; ++i) p[i] = make_float2(a[2 * i], a[2 * i + 1]);
    float x = out[threadIdx.x & 63];
    const float s0 = w[0], s1 = w[1];   // uniform -> SGPRs
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(a[(i + 1) & 15]));
        } else if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "s"(s0));
        } else if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(p[(i + 1) & 7]), "v"(p[(i + 2) & 7]));
        } else {
            float2 sw = make_float2(s0, s1);
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(p[i]) : "v"(p[(i + 1) & 7]), "s"(sw));
        }
    }
    long long t1 = clock64();
    float r = 0;
    for (int i = 0; i < 16; ++i) r += a[i];
    for (int i = 0; i < 8; ++i) r += p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
    float *out, *w; long long* cyc;
    hipMalloc(&out, 1 << 22); hipMalloc(&w, 64); hipMalloc(&cyc, 8);
    hipMemset(out, 0, 1 << 22); hipMemset(w, 0, 64);
    const int iters = 2000;
    const char* names[4] = {"v_fma_f32 vgpr   (16/iter)", "v_fma_f32 sgpr   (16/iter)", "v_pk_fma_f32 vgpr( 8/iter)", "v_pk_fma_f32 sgpr( 8/iter)"};
    for (int mode = 0; mode < 4; ++mode)
        for (int waves_per_simd : {1, 2, 4, 8}) {
            dim3 block(64 * 4 * waves_per_simd > 1024 ? 1024 : 64 * 4 * waves_per_simd);
            int blocks_per_cu = (64 * 4 * waves_per_simd) / block.x;
            dim3 grid(256 * blocks_per_cu);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int rep = 0; rep < 2; ++rep) {
                if (rep == 1) hipEventRecord(e0, 0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, grid, block, 0, 0, out, w, iters, cyc);
                if (mode == 1) hipLaunchKernelGGL(k<1>, grid, block, 0, 0, out, w, iters, cyc);
                if (mode == 2) hipLaunchKernelGGL(k<2>, grid, block, 0, 0, out, w, iters, cyc);
                if (mode == 3) hipLaunchKernelGGL(k<3>, grid, block, 0, 0, out, w, iters, cyc);
                if (rep == 1) hipEventRecord(e1, 0);
                hipDeviceSynchronize();
            }
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            int n = (mode < 2 ? 16 : 8);
            double flops = 2.0 * 16 * iters * 64.0 * 4 * waves_per_simd * 256;
            printf("%s waves/SIMD=%d : %.2f clk64-ticks per instr per wave ; kernel %.1f us -> %.1f TFLOP/s\n", names[mode], waves_per_simd,
                   (double)c / iters / n, ms * 1e3, flops / (ms * 1e-3) / 1e12);
        }
    return 0;
}
