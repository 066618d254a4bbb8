// Probe (GPU box): does a host -> device hipMemcpyAsync make progress while a kernel that leaves no free wavefront slot (128 VGPRs x 16 waves per CU, 40 KB LDS x 4) spins on every CU?
// (If the runtime copies with a shader "blit" kernel instead of an SDMA engine, the copy needs a CU and waits for the spinning kernel -- which waits for the copy.)
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/copy_under_full_gpu_probe.hip -o /tmp/cfp && /tmp/cfp
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("  %s -> %s\n", #x, hipGetErrorString(e_)); } } while (0)

__global__ __launch_bounds__(256) void k_spin(const unsigned* flag, unsigned want, long long limit, unsigned* timed_out) {
    HIP_DYNAMIC_SHARED(float, lds)
    asm volatile("v_mov_b32 v127, 0" ::: "v127");        // allocate 128 VGPRs: four such waves fill a SIMD's register file
    lds[threadIdx.x] = 0.0f;
    if (threadIdx.x == 0 && blockIdx.x == 0 && timed_out[1] != 0u) __hip_atomic_store(reinterpret_cast<unsigned*>(const_cast<unsigned*>(flag)) + 16, want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (threadIdx.x == 0) {
        const long long t0 = wall_clock64();
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != want) {
            if (wall_clock64() - t0 > limit) { atomicAdd(timed_out, 1u); break; }
            __builtin_amdgcn_s_sleep(16);
        }
    }
    __syncthreads();
}

int main() {
    hipStream_t sk, sc;
    CK(hipStreamCreateWithFlags(&sk, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sc, hipStreamNonBlocking));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_spin), hipFuncAttributeMaxDynamicSharedMemorySize, 40 * 1024));
    unsigned *flag, *d_to;
    CK(hipExtMallocWithFlags((void**)&flag, 128, hipDeviceMallocFinegrained));
    CK(hipMalloc((void**)&d_to, 8)); CK(hipMemset(d_to, 0, 8)); CK(hipMemset(flag, 0, 64 * 0 + 64));
    char *h_src, *d_dst;
    const size_t big = 16u << 20;
    CK(hipHostMalloc((void**)&h_src, big, 0)); CK(hipMalloc((void**)&d_dst, big));
    unsigned epoch = 0;
    for (size_t bytes : {size_t(64) << 10, size_t(512) << 10, size_t(2) << 20, size_t(8) << 20, size_t(16) << 20}) {
        for (int grid : {256, 1024}) {
            ++epoch;
            CK(hipMemset(d_to, 0, 4));
            CK(hipDeviceSynchronize());
            hipLaunchKernelGGL(k_spin, dim3(grid), dim3(256), 40 * 1024, sk, (const unsigned*)flag, epoch, 20000000LL, d_to);      // 0.2 s bound
            std::this_thread::sleep_for(std::chrono::milliseconds(2));
            const auto t0 = std::chrono::steady_clock::now();
            CK(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, sc));
            CK(hipStreamWriteValue32(sc, flag, epoch, 0));
            CK(hipStreamSynchronize(sc));
            const double ms_c = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            CK(hipStreamSynchronize(sk));
            unsigned to = 0;
            CK(hipMemcpy(&to, d_to, 4, hipMemcpyDeviceToHost));
            printf("copy of %6zu KB under %4d spinning workgroups: copy + flag done after %8.3f ms, workgroups that timed out: %u\n", bytes >> 10, grid, ms_c, to);
        }
    }
    unsigned* h_word;
    CK(hipHostMalloc((void**)&h_word, 64, 0));
    // which of the two needs a CU?  (the spinning kernel is left to time out; only the copy stream is timed)
    for (int what = 0; what < 5; ++what) {
        ++epoch;
        CK(hipMemset(d_to, 0, 4));
        CK(hipDeviceSynchronize());
        if (what == 4) { const unsigned one = 1u; CK(hipMemcpy(d_to + 1, &one, 4, hipMemcpyHostToDevice)); }      // block 0 raises flag[16] = epoch at once
        hipLaunchKernelGGL(k_spin, dim3(1024), dim3(256), 40 * 1024, sk, (const unsigned*)flag, epoch, 20000000LL, d_to);
        std::this_thread::sleep_for(std::chrono::milliseconds(2));
        const auto t0 = std::chrono::steady_clock::now();
        if (what == 0) CK(hipMemcpyAsync(d_dst, h_src, size_t(2) << 20, hipMemcpyHostToDevice, sc));
        if (what == 1) CK(hipStreamWriteValue32(sc, flag + 8, epoch, 0));
        if (what == 2) CK(hipMemcpyAsync(h_src, d_dst, size_t(2) << 20, hipMemcpyDeviceToHost, sc));
        if (what == 3) { *h_word = epoch; CK(hipMemcpyAsync(flag + 8, h_word, 4, hipMemcpyHostToDevice, sc)); }
        if (what == 4) { CK(hipStreamWaitValue32(sc, flag + 16, epoch, hipStreamWaitValueEq, 0xffffffffu)); CK(hipMemcpyAsync(h_src, d_dst, size_t(2) << 20, hipMemcpyDeviceToHost, sc)); }
        CK(hipStreamSynchronize(sc));
        const double ms_c = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        CK(hipStreamSynchronize(sk));
        printf("under 1024 spinning workgroups, %s alone: done after %8.3f ms\n", what == 0 ? "a 2 MB host -> device copy" : what == 1 ? "a hipStreamWriteValue32" : what == 2 ? "a 2 MB device -> host copy" : what == 3 ? "a 4-byte host -> device copy (page-locked source)" : "a hipStreamWaitValue32 on a word the kernel has already raised + a 2 MB device -> host copy", ms_c);
    }
    // from which size does a host -> device copy run on a copy engine?
    for (size_t bytes : {size_t(256), size_t(1) << 10, size_t(4) << 10, size_t(16) << 10, size_t(64) << 10, size_t(256) << 10}) {
        ++epoch;
        CK(hipMemset(d_to, 0, 8));
        CK(hipDeviceSynchronize());
        hipLaunchKernelGGL(k_spin, dim3(1024), dim3(256), 40 * 1024, sk, (const unsigned*)flag, epoch, 20000000LL, d_to);
        std::this_thread::sleep_for(std::chrono::milliseconds(2));
        const auto t0 = std::chrono::steady_clock::now();
        CK(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, sc));
        CK(hipStreamSynchronize(sc));
        const double ms_c = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        CK(hipStreamSynchronize(sk));
        printf("under 1024 spinning workgroups, a %zu-byte host -> device copy alone: done after %8.3f ms\n", bytes, ms_c);
    }
    return 0;
}
