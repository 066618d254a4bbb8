// Micro-benchmark: does a SIMD of gfx950 execute matrix instructions of one wavefront UNDER vector instructions of another?
// 512-thread workgroups (wavefront w and w + 4 share a SIMD): waves 0-3 run MFMAs only, waves 4-7 packed FMAs only; each alone, then together.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef short v8s __attribute__((ext_vector_type(8)));
template <int KIND, int VK> __global__ void k(float* out, const float* in, int it_m, int it_v) {
    const int wave = threadIdx.x >> 6;
    float a0 = in[threadIdx.x & 63], a1 = in[64 + (threadIdx.x & 63)];
    v4f d0 = {0, 0, 0, 0}, d1 = d0, d2 = d0, d3 = d0;
    v2f p0 = {a0, a1}, p1 = {a1, a0}, p2 = p0, p3 = p1, p4 = p0, p5 = p1, p6 = p0, p7 = p1;
    v8s ba = {1, 2, 3, 4, 5, 6, 7, 8}, bb = {2, 3, 4, 5, 6, 7, 8, 9};
    if (wave < 4) {
        for (int i = 0; i < it_m; ++i) {
            if (KIND == 0) {
                d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, a1, d0, 0, 0, 0); d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, a1, d1, 0, 0, 0);
                d2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, a1, d2, 0, 0, 0); d3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, a1, d3, 0, 0, 0);
            } else {
                d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ba, bb, d0, 0, 0, 0); d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ba, bb, d1, 0, 0, 0);
                d2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ba, bb, d2, 0, 0, 0); d3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ba, bb, d3, 0, 0, 0);
            }
        }
    } else {
        for (int i = 0; i < it_v; ++i) {
            if (VK == 0) {
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p0) : "v"(p1), "v"(p2));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p1) : "v"(p2), "v"(p3));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p2) : "v"(p3), "v"(p4));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p3) : "v"(p4), "v"(p5));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p4) : "v"(p5), "v"(p6));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p5) : "v"(p6), "v"(p7));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p6) : "v"(p7), "v"(p0));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p7) : "v"(p0), "v"(p1));
            } else if (VK == 1) {
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(p0[0]) : "v"(p1[0]), "v"(p2[0]));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(p1[0]) : "v"(p2[0]), "v"(p3[0]));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(p2[0]) : "v"(p3[0]), "v"(p4[0]));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(p3[0]) : "v"(p4[0]), "v"(p5[0]));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(p4[0]) : "v"(p5[0]), "v"(p6[0]));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(p5[0]) : "v"(p6[0]), "v"(p7[0]));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(p6[0]) : "v"(p7[0]), "v"(p0[0]));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(p7[0]) : "v"(p0[0]), "v"(p1[0]));
            } else {
                asm volatile("v_exp_f32 %0, %0" : "+v"(p0[0]));
                asm volatile("v_exp_f32 %0, %0" : "+v"(p1[0]));
                asm volatile("v_exp_f32 %0, %0" : "+v"(p2[0]));
                asm volatile("v_exp_f32 %0, %0" : "+v"(p3[0]));
                asm volatile("v_rcp_f32 %0, %0" : "+v"(p4[0]));
                asm volatile("v_rcp_f32 %0, %0" : "+v"(p5[0]));
                asm volatile("v_rcp_f32 %0, %0" : "+v"(p6[0]));
                asm volatile("v_rcp_f32 %0, %0" : "+v"(p7[0]));
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = d0[0] + d1[1] + d2[2] + d3[3] + p0[0] + p1[1] + p2[0] + p3[1] + p4[0] + p5[1] + p6[0] + p7[1];
}
template <int KIND, int VK> float run(float* out, const float* in, int it_m, int it_v) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0.f;
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k<KIND, VK>), dim3(256), dim3(512), 0, 0, out, in, it_m, it_v);
        (void)hipEventRecord(e1, 0);
        (void)hipDeviceSynchronize();
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    return ms * 1e3f;
}
int main() {
    float *out, *in;
    (void)hipMalloc(&out, 256 * 1024 * 4); (void)hipMalloc(&in, 1024);
    (void)hipMemset(in, 0, 1024);
    const int M = 4000, V = 18000;
    const char* vn[3] = {"v_pk_fma_f32", "v_fma_f32   ", "v_exp/v_rcp "};
#define ROW(KIND, VK, MM, name) printf("%s + %s : MFMA waves alone %.1f us | VALU waves alone %.1f us | together %.1f us\n", name, vn[VK], run<KIND, VK>(out, in, MM, 0), run<KIND, VK>(out, in, 0, V), run<KIND, VK>(out, in, MM, V));
    ROW(0, 0, M, "f32 16x16x4  ") ROW(0, 1, M, "f32 16x16x4  ") ROW(0, 2, M, "f32 16x16x4  ")
    ROW(1, 0, 2 * M, "bf16 16x16x32") ROW(1, 1, 2 * M, "bf16 16x16x32") ROW(1, 2, 2 * M, "bf16 16x16x32")
    return 0;
}
