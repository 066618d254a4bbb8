// Micro-benchmark: what one step of a batched GRU recurrence costs on gfx950 (v_mfma_f32_16x16x4_f32 + exp2 / rcp gates), alone and with 4 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/gru_step_probe.hip -o tools/ubench/_build/gru_step_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4f mf(float a, float b, v4f c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ float gates(v4f d, float h) {
    const float r = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(d[0]));
    const float z = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(d[1]));
    const float n = 1.0f - 2.0f * __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(d[3] + r * d[2]) + 1.0f);
    return n + z * (h - n);
}
template <int MODE> __global__ void k(float* out, const float* in, int iters) {
    const float a0 = in[threadIdx.x & 63], a1 = in[64 + (threadIdx.x & 63)], a2 = in[128 + (threadIdx.x & 63)], a3 = in[192 + (threadIdx.x & 63)];
    float h0 = 0.1f, h1 = 0.2f, h2 = 0.3f, h3 = 0.4f;
    v4f cb = {0.1f, 0.2f, 0.3f, 0.4f};
    v4f d0 = cb, d1 = cb, d2 = cb, d3 = cb;
    float x = a0;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {            // dependent MFMA chain (4 per iteration)
            d0 = mf(a0, a1, d0); d0 = mf(a0, a1, d0); d0 = mf(a0, a1, d0); d0 = mf(a0, a1, d0);
        } else if (MODE == 1) {     // 4 independent MFMAs
            d0 = mf(a0, a1, d0); d1 = mf(a0, a1, d1); d2 = mf(a0, a1, d2); d3 = mf(a0, a1, d3);
        } else if (MODE == 2) {     // dependent exp2 -> rcp chain (4 transcendentals per iteration)
            x = __builtin_amdgcn_exp2f(x); x = __builtin_amdgcn_rcpf(x); x = __builtin_amdgcn_exp2f(x); x = __builtin_amdgcn_rcpf(x);
        } else if (MODE == 3) {     // 4 independent transcendentals
            h0 = __builtin_amdgcn_exp2f(h0); h1 = __builtin_amdgcn_rcpf(h1); h2 = __builtin_amdgcn_exp2f(h2); h3 = __builtin_amdgcn_rcpf(h3);
        } else if (MODE == 4) {     // intra-frame step: 1 recurrent MFMA + gates; 2 input MFMAs off the chain
            v4f dx = mf(a0, a2, cb); dx = mf(a1, a3, dx);
            d0 = mf(a2, h0, dx);
            h0 = gates(d0, h0);
        } else if (MODE == 5) {     // inter-frame step, one tile: 2 x (2 chained recurrent MFMAs) + 2 x 2 input MFMAs + 2 gates
            v4f dx0 = mf(a0, a2, cb), dx1 = mf(a1, a2, cb); dx0 = mf(a1, a3, dx0); dx1 = mf(a0, a3, dx1);
            d0 = mf(a2, h0, dx0); d1 = mf(a3, h0, dx1); d0 = mf(a3, h1, d0); d1 = mf(a2, h1, d1);
            h0 = gates(d0, h0); h1 = gates(d1, h1);
        } else if (MODE == 6) {     // inter-frame step, two interleaved tiles
            v4f dx0 = mf(a0, a2, cb), dx1 = mf(a1, a2, cb); dx0 = mf(a1, a3, dx0); dx1 = mf(a0, a3, dx1);
            d0 = mf(a2, h0, dx0); d1 = mf(a3, h0, dx1); d0 = mf(a3, h1, d0); d1 = mf(a2, h1, d1);
            v4f ex0 = mf(a0, a3, cb), ex1 = mf(a1, a3, cb); ex0 = mf(a1, a2, ex0); ex1 = mf(a0, a2, ex1);
            d2 = mf(a2, h2, ex0); d3 = mf(a3, h2, ex1); d2 = mf(a3, h3, d2); d3 = mf(a2, h3, d3);
            h0 = gates(d0, h0); h1 = gates(d1, h1);
            h2 = gates(d2, h2); h3 = gates(d3, h3);
        } else if (MODE == 7) {     // the gates alone (dependent on h through a cheap op)
            d0[0] = h0; d0[2] = h0 * 0.5f;
            h0 = gates(d0, h0);
        } else if (MODE == 8) {     // VALU formulation of one inter step's FMAs: 24 packed FMAs + gates (old form, one unit per lane)
            float s0 = h0, s1 = h1;
#pragma unroll
            for (int q = 0; q < 12; ++q) { s0 = fmaf(a0, s0, a1); s1 = fmaf(a2, s1, a3); }
            d0[0] = s0; d0[1] = s1; d0[2] = s0 + s1;
            h0 = gates(d0, h0); h1 = h0 * 0.5f;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = d0[0] + d1[1] + d2[2] + d3[3] + h0 + h1 + h2 + h3 + x;
}
template <int MODE> void run(const char* name, float* out, const float* in, int per_iter) {
    const int iters = 4000;
    for (int wps = 1; wps <= 4; wps *= 2) {
        const int threads = 256 * wps;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        float ms = 0.f;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, in, iters);
            hipEventRecord(e1, 0);
            hipDeviceSynchronize();
            hipEventElapsedTime(&ms, e0, e1);
        }
        printf("%-44s waves/SIMD=%d : %7.1f ns per iteration (%6.0f cycles @2.4GHz), %6.1f cycles per item\n", name, wps, ms * 1e6 / iters, ms * 1e6 / iters * 2.4,
               ms * 1e6 / iters * 2.4 / per_iter);
    }
}
int main() {
    float *out, *in;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&in, 1024);
    float hin[256]; for (int i = 0; i < 256; ++i) hin[i] = 0.001f * (i % 37) - 0.01f;
    hipMemcpy(in, hin, 1024, hipMemcpyHostToDevice);
    run<0>("0 dependent MFMA chain (4)", out, in, 4);
    run<1>("1 independent MFMAs (4)", out, in, 4);
    run<2>("2 dependent exp2/rcp chain (4)", out, in, 4);
    run<3>("3 independent exp2/rcp (4)", out, in, 4);
    run<4>("4 intra step: 2 in + 1 rec MFMA + gates", out, in, 1);
    run<5>("5 inter step, 1 tile: 8 MFMA + 2 gates", out, in, 1);
    run<6>("6 inter step, 2 tiles: 16 MFMA + 4 gates", out, in, 1);
    run<7>("7 gates alone (dependent)", out, in, 1);
    run<8>("8 VALU step: 24 dep fma + gates", out, in, 1);
    return 0;
}
