// Probe (GPU box): which memory kinds carry hipStreamWriteValue32 -> a RUNNING kernel's poll, and a running kernel's store -> hipStreamWaitValue32.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/stream_value_probe.hip -o /tmp/svp && /tmp/svp
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("  %s -> %s\n", #x, hipGetErrorString(e_)); } } while (0)

__global__ void k_poll(const unsigned* flag, unsigned want, unsigned* seen, const int* data, int* out, long long limit) {
    const long long t0 = wall_clock64();
    unsigned v;
    while ((v = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) != want && wall_clock64() - t0 < limit) __builtin_amdgcn_s_sleep(8);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    *seen = v;
    *out = data[0] + data[1023];
}
__global__ void k_signal(unsigned* done, unsigned v, int* data) {
    data[0] = 7000 + (int)v;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    __hip_atomic_store(done, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

int main() {
    hipStream_t sk, sc;
    CK(hipStreamCreateWithFlags(&sk, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sc, hipStreamNonBlocking));
    int *d_data, *h_src, *d_out;
    unsigned* d_seen;
    CK(hipMalloc((void**)&d_data, 4096)); CK(hipMalloc((void**)&d_out, 4)); CK(hipMalloc((void**)&d_seen, 4));
    CK(hipHostMalloc((void**)&h_src, 4096, 0));
    const char* names[3] = {"hipDeviceMallocFinegrained", "hipMalloc", "hipHostMalloc"};
    for (int kind = 0; kind < 3; ++kind) {
        unsigned* flag = nullptr;
        hipError_t e = kind == 0 ? hipExtMallocWithFlags((void**)&flag, 64, hipDeviceMallocFinegrained) : kind == 1 ? hipMalloc((void**)&flag, 64) : hipHostMalloc((void**)&flag, 64, 0);
        printf("%s: alloc %s\n", names[kind], hipGetErrorString(e));
        if (e != hipSuccess) continue;
        for (unsigned epoch = 1; epoch <= 3; ++epoch) {
            for (int i = 0; i < 1024; ++i) h_src[i] = (int)epoch * 1000 + i;
            if (epoch == 1) { if (kind == 2) *flag = 0; else CK(hipMemset(flag, 0, 64)); }
            CK(hipDeviceSynchronize());
            hipLaunchKernelGGL(k_poll, dim3(1), dim3(1), 0, sk, (const unsigned*)flag, epoch, d_seen, (const int*)d_data, d_out, 50000000LL);     // 0.5 s bound
            std::this_thread::sleep_for(std::chrono::milliseconds(5));
            CK(hipMemcpyAsync(d_data, h_src, 4096, hipMemcpyHostToDevice, sc));
            e = hipStreamWriteValue32(sc, flag, epoch, 0);
            const auto t0 = std::chrono::steady_clock::now();
            CK(hipStreamSynchronize(sk));
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            unsigned seen = 0; int out = 0;
            CK(hipMemcpy(&seen, d_seen, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&out, d_out, 4, hipMemcpyDeviceToHost));
            printf("  epoch %u: write-value %s, kernel saw %u after %.2f ms, data sum %d (want %d)\n", epoch, hipGetErrorString(e), seen, ms, out, (int)epoch * 2000 + 1023);
        }
    }
    // kernel store -> hipStreamWaitValue32
    for (int kind = 0; kind < 2; ++kind) {
        unsigned* done = nullptr;
        hipError_t e = kind == 0 ? hipExtMallocWithFlags((void**)&done, 8, hipMallocSignalMemory) : hipExtMallocWithFlags((void**)&done, 64, hipDeviceMallocFinegrained);
        printf("wait on %s: alloc %s\n", kind == 0 ? "hipMallocSignalMemory" : "hipDeviceMallocFinegrained", hipGetErrorString(e));
        if (e != hipSuccess) continue;
        if (kind == 0) *reinterpret_cast<volatile unsigned long long*>(done) = 0ull; else CK(hipMemset(done, 0, 64));
        CK(hipDeviceSynchronize());
        for (unsigned epoch = 1; epoch <= 3; ++epoch) {
            CK(hipMemset(d_data, 0, 4));
            CK(hipDeviceSynchronize());
            e = hipStreamWaitValue32(sc, done, epoch, hipStreamWaitValueEq, 0xffffffffu);
            int* h_dst = h_src;
            CK(hipMemcpyAsync(h_dst, d_data, 4, hipMemcpyDeviceToHost, sc));
            std::this_thread::sleep_for(std::chrono::milliseconds(5));
            printf("  epoch %u: copy stream %s before the kernel ran\n", epoch, hipStreamQuery(sc) == hipSuccess ? "ALREADY DONE (the wait did not hold)" : "still waiting");
            hipLaunchKernelGGL(k_signal, dim3(1), dim3(1), 0, sk, done, epoch, d_data);
            const auto t0 = std::chrono::steady_clock::now();
            hipError_t es = hipStreamSynchronize(sc);
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            printf("  epoch %u: wait-value %s, copy stream done (%s) after %.2f ms, host got %d (want %d)\n", epoch, hipGetErrorString(e), hipGetErrorString(es), ms, h_dst[0], 7000 + (int)epoch);
            CK(hipDeviceSynchronize());
        }
    }
    return 0;
}
