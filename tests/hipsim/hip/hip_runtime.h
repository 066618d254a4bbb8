// hipsim — a tiny host-side stand-in for <hip/hip_runtime.h>.  TEST INFRASTRUCTURE ONLY.
//
// Purpose: this build container has hipcc but NO GPU, and a round trip to a real MI355X costs minutes.
// Compiling the very same csrc/*.hip sources with g++ against this header runs every kernel on the CPU
// as cooperative fibers (one per HIP thread, blocks executed one after another) so that indexing,
// __syncthreads()/__shfl() data flow and the host-side engine logic can be checked against the oracle
// before GPU time is spent.  It is NOT a product path: the package only ever loads libade.so built by
// hipcc for gfx950 and fails loudly when that library or a GPU is missing; nothing outside tests/ links this.
#pragma once

#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>

#define HIPSIM 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __noinline__ __attribute__((noinline))
#define __shared__ static
#define __constant__ static
#ifndef __restrict__
#define __restrict__ __restrict
#endif
#define HIP_KERNEL_NAME(...) __VA_ARGS__

struct dim3 {
    unsigned x, y, z;
    constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

struct float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct alignas(16) int4 { int x, y, z, w; };
struct short2 { short x, y; };
struct alignas(8) short4 { short x, y, z, w; };
static inline float2 make_float2(float x, float y) { return {x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return {x, y, z, w}; }
static inline int2 make_int2(int x, int y) { return {x, y}; }
static inline short4 make_short4(short x, short y, short z, short w) { return {x, y, z, w}; }
static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }

namespace hipsim {
struct ThreadState {
    dim3 tid;
};
extern ThreadState* cur;   // fiber currently running
extern dim3 cur_block, cur_bdim, cur_gdim;
void run_grid(dim3 grid, dim3 block, const std::function<void()>& body);
void block_barrier();
uint32_t wave_exchange(uint32_t v, int src_lane);   // every live lane of the wave must call
void wave_allgather2(uint32_t a, uint32_t b, uint32_t* out_a, uint32_t* out_b);   // every lane of the wave must call
int lane_id();
}  // namespace hipsim

#define threadIdx (::hipsim::cur->tid)
#define blockIdx (::hipsim::cur_block)
#define blockDim (::hipsim::cur_bdim)
#define gridDim (::hipsim::cur_gdim)
static const int warpSize = 64;

static inline void __syncthreads() { ::hipsim::block_barrier(); }

template <typename T>
static inline T hipsim_shfl_any(T v, int src) {
    static_assert(sizeof(T) == 4, "hipsim shuffles are 32-bit");
    uint32_t u;
    std::memcpy(&u, &v, 4);
    u = ::hipsim::wave_exchange(u, src);
    T r;
    std::memcpy(&r, &u, 4);
    return r;
}
template <typename T>
static inline T __shfl(T v, int src, int width = 64) {
    const int lane = ::hipsim::lane_id();
    const int base = lane & ~(width - 1);
    return hipsim_shfl_any(v, base + (src & (width - 1)));
}
template <typename T>
static inline T __shfl_xor(T v, int mask, int width = 64) {
    const int lane = ::hipsim::lane_id();
    const int base = lane & ~(width - 1);
    const int s = (lane ^ mask);
    return hipsim_shfl_any(v, (s & ~(width - 1)) == base ? s : lane);
}
template <typename T>
static inline T __shfl_down(T v, unsigned d, int width = 64) {
    const int lane = ::hipsim::lane_id();
    const int s = lane + (int)d;
    return hipsim_shfl_any(v, (s & ~(width - 1)) == (lane & ~(width - 1)) ? s : lane);
}
template <typename T>
static inline T __shfl_up(T v, unsigned d, int width = 64) {
    const int lane = ::hipsim::lane_id();
    const int s = lane - (int)d;
    return hipsim_shfl_any(v, (s >= 0 && (s & ~(width - 1)) == (lane & ~(width - 1))) ? s : lane);
}

#define __expf(x) expf(x)   /* glibc declares extern __expf/__logf itself */
#define __logf(x) logf(x)
// DPP (data-parallel primitives) emulation: quad_perm (ctrl 0x00-0xFF) and row_newbcast:n (0x150+n, gfx90a+).
static inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
    (void)old; (void)row_mask; (void)bank_mask; (void)bound_ctrl;
    const int lane = ::hipsim::lane_id();
    int from;
    if (ctrl >= 0 && ctrl <= 0xFF) from = (lane & ~3) | ((ctrl >> (2 * (lane & 3))) & 3);
    else if (ctrl >= 0x150 && ctrl <= 0x15F) from = (lane & ~15) | (ctrl & 15);
    else if (ctrl >= 0x121 && ctrl <= 0x12F) from = (lane & ~15) | ((lane - (ctrl & 15)) & 15);   // row_ror:n
    else { fprintf(stderr, "hipsim: unsupported dpp ctrl 0x%x\n", ctrl); abort(); }
    return (int)::hipsim::wave_exchange((uint32_t)src, from);
}
// gfx950 v_permlane16_swap_b32 / v_permlane32_swap_b32 (semantics probed on hardware, tools/ubench/permlane_probe.hip):
//   p16: vdst' = [vdst.r0, src0.r0, vdst.r2, src0.r2], src0' = [vdst.r1, src0.r1, vdst.r3, src0.r3]  (16-lane rows)
//   p32: vdst' = [vdst.lo, src0.lo],                   src0' = [vdst.hi, src0.hi]                      (32-lane halves)
struct hipsim_u2 { unsigned v[2]; unsigned operator[](int i) const { return v[i]; } };
static inline hipsim_u2 __builtin_amdgcn_permlane16_swap(unsigned vdst, unsigned src0, bool, bool) {
    const int lane = ::hipsim::lane_id();
    const unsigned s_prev = ::hipsim::wave_exchange(src0, (lane - 16) & 63), v_next = ::hipsim::wave_exchange(vdst, (lane + 16) & 63);
    const bool odd = (lane >> 4) & 1;
    return hipsim_u2{{odd ? s_prev : vdst, odd ? src0 : v_next}};
}
static inline hipsim_u2 __builtin_amdgcn_permlane32_swap(unsigned vdst, unsigned src0, bool, bool) {
    const int lane = ::hipsim::lane_id();
    const unsigned s_lo = ::hipsim::wave_exchange(src0, (lane - 32) & 63), v_hi = ::hipsim::wave_exchange(vdst, (lane + 32) & 63);
    const bool hi = lane >= 32;
    return hipsim_u2{{hi ? s_lo : vdst, hi ? src0 : v_hi}};
}
static inline int __float_as_int(float f) { int i; std::memcpy(&i, &f, 4); return i; }
static inline float __int_as_float(int i) { float f; std::memcpy(&f, &i, 4); return f; }
static inline int __builtin_amdgcn_readlane(int x, int lane) { return (int)::hipsim::wave_exchange((uint32_t)x, lane); }
static inline void __builtin_amdgcn_sched_barrier(int) {}
static inline void __builtin_amdgcn_s_sleep(int) {}
static inline void __builtin_amdgcn_s_setprio(int) {}
// v_mfma_f32_16x16x4_f32: D = A(16x4) B(4x16) + C, operand layout as documented in csrc/ade_device.h; k-ordered fmaf chain.
typedef float hipsim_v4f __attribute__((vector_size(16)));
static inline hipsim_v4f __builtin_amdgcn_mfma_f32_16x16x4f32(float a, float b, hipsim_v4f c, int, int, int) {
    const int lane = ::hipsim::lane_id();
    const int j = lane & 15, g = lane >> 4;
    uint32_t ab, bb, all_a[64], all_b[64];
    std::memcpy(&ab, &a, 4);
    std::memcpy(&bb, &b, 4);
    ::hipsim::wave_allgather2(ab, bb, all_a, all_b);
    hipsim_v4f d = c;
    for (int r = 0; r < 4; ++r)
        for (int k = 0; k < 4; ++k) {
            float af, bf;
            std::memcpy(&af, &all_a[(4 * g + r) + 16 * k], 4);     // A[i = 4g + r][k]
            std::memcpy(&bf, &all_b[j + 16 * k], 4);               // B[k][j]
            d[r] = fmaf(af, bf, d[r]);
        }
    return d;
}
// gfx950 v_mfma_f32_32x32x16_bf16 / v_mfma_f32_16x16x32_bf16: a lane supplies 8 consecutive k of its row as eight bf16 (csrc/ade_gemm16.h documents the maps);
// D = A B + C in fp32, accumulated k-ordered here (products of bf16 values are exact in fp32).
typedef short hipsim_v8s __attribute__((vector_size(16)));
typedef float hipsim_v16f __attribute__((vector_size(64)));
static inline float hipsim_bf16_of(const uint32_t (*all)[64], int lane, int e) {
    const uint32_t w = all[e >> 1][lane], u = ((e & 1) ? (w >> 16) : (w & 0xffffu)) << 16;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}
static inline void hipsim_gather8(hipsim_v8s a, hipsim_v8s b, uint32_t (*all_a)[64], uint32_t (*all_b)[64]) {
    uint32_t aw[4], bw[4];
    std::memcpy(aw, &a, 16);
    std::memcpy(bw, &b, 16);
    for (int q = 0; q < 4; ++q) ::hipsim::wave_allgather2(aw[q], bw[q], all_a[q], all_b[q]);
}
static inline hipsim_v16f __builtin_amdgcn_mfma_f32_32x32x16_bf16(hipsim_v8s a, hipsim_v8s b, hipsim_v16f c, int, int, int) {
    const int lane = ::hipsim::lane_id(), j = lane & 31, h = lane >> 5;
    uint32_t all_a[4][64], all_b[4][64];
    hipsim_gather8(a, b, all_a, all_b);
    hipsim_v16f d = c;
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * h;
        for (int k = 0; k < 16; ++k) d[r] = fmaf(hipsim_bf16_of(all_a, i + 32 * (k >> 3), k & 7), hipsim_bf16_of(all_b, j + 32 * (k >> 3), k & 7), d[r]);
    }
    return d;
}
// v_mfma_f32_32x32x16_f16: the same operand maps with IEEE half elements
static inline float hipsim_f16_of(const uint32_t (*all)[64], int lane, int e) {
    const uint32_t w = all[e >> 1][lane], hb = (e & 1) ? (w >> 16) : (w & 0xffffu);
    const uint32_t sign = (hb & 0x8000u) << 16, ex = (hb >> 10) & 0x1fu, m = hb & 0x3ffu;
    float f;
    if (ex == 0) { f = std::ldexp((float)m, -24); if (sign) f = -f; return f; }
    const uint32_t u = ex == 31 ? (sign | 0x7f800000u | (m << 13)) : (sign | ((ex + 112u) << 23) | (m << 13));
    std::memcpy(&f, &u, 4);
    return f;
}
static inline hipsim_v16f __builtin_amdgcn_mfma_f32_32x32x16_f16(hipsim_v8s a, hipsim_v8s b, hipsim_v16f c, int, int, int) {
    const int lane = ::hipsim::lane_id(), j = lane & 31, h = lane >> 5;
    uint32_t all_a[4][64], all_b[4][64];
    hipsim_gather8(a, b, all_a, all_b);
    hipsim_v16f d = c;
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * h;
        for (int k = 0; k < 16; ++k) d[r] = fmaf(hipsim_f16_of(all_a, i + 32 * (k >> 3), k & 7), hipsim_f16_of(all_b, j + 32 * (k >> 3), k & 7), d[r]);
    }
    return d;
}
static inline hipsim_v4f __builtin_amdgcn_mfma_f32_16x16x32_bf16(hipsim_v8s a, hipsim_v8s b, hipsim_v4f c, int, int, int) {
    const int lane = ::hipsim::lane_id(), j = lane & 15, g = lane >> 4;
    uint32_t all_a[4][64], all_b[4][64];
    hipsim_gather8(a, b, all_a, all_b);
    hipsim_v4f d = c;
    for (int r = 0; r < 4; ++r)
        for (int k = 0; k < 32; ++k) d[r] = fmaf(hipsim_bf16_of(all_a, (4 * g + r) + 16 * (k >> 3), k & 7), hipsim_bf16_of(all_b, j + 16 * (k >> 3), k & 7), d[r]);
    return d;
}
struct alignas(16) uint4 { unsigned x, y, z, w; };
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
static inline float __uint_as_float(unsigned u) { float f; std::memcpy(&f, &u, 4); return f; }
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
struct uint2 { unsigned x, y; };
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
static inline unsigned __float_as_uint(float x) { unsigned u; std::memcpy(&u, &x, 4); return u; }
static inline int __builtin_amdgcn_readfirstlane(int x) { return x; }   // only ever applied to wave-uniform values here
static inline float __builtin_amdgcn_rcpf(float a) { return 1.0f / a; }
static inline float __builtin_amdgcn_sqrtf(float a) { return sqrtf(a); }
static inline float __builtin_amdgcn_rsqf(float a) { return 1.0f / sqrtf(a); }
static inline float __builtin_amdgcn_fmed3f(float a, float b, float c) {
    const float lo = a < b ? a : b, hi = a < b ? b : a;
    return c < lo ? lo : (c > hi ? hi : c);
}
static inline long long wall_clock64() { static thread_local long long ticks = 0; return ticks += 1000; }   // monotonic, so timed waits end
static inline float atomicAdd(float* p, float v) { const float o = *p; *p = o + v; return o; }   // fibers never preempt
static inline void __builtin_amdgcn_wave_barrier() { (void)::hipsim::wave_exchange(0u, ::hipsim::lane_id()); }
static inline long long clock64() { return 0; }
static inline float __builtin_amdgcn_exp2f(float a) { return exp2f(a); }
static inline float __builtin_amdgcn_logf(float a) { return log2f(a); }      // v_log_f32 = log2
namespace hipsim { extern unsigned char dyn_smem[]; }
#define HIP_DYNAMIC_SHARED(type, var) type* var = reinterpret_cast<type*>(::hipsim::dyn_smem);
static inline float __fdividef(float a, float b) { return a / b; }
static inline float __frcp_rn(float a) { return 1.0f / a; }
static inline float __fsqrt_rn(float a) { return sqrtf(a); }
static inline float __frsqrt_rn(float a) { return 1.0f / sqrtf(a); }
static inline float rsqrtf(float a) { return 1.0f / sqrtf(a); }
static inline float __fmaf_rn(float a, float b, float c) { return fmaf(a, b, c); }
static inline float __fmul_rn(float a, float b) { volatile float r = a * b; return r; }
static inline float __fadd_rn(float a, float b) { volatile float r = a + b; return r; }
static inline float __saturatef(float a) { return a < 0.f ? 0.f : (a > 1.f ? 1.f : a); }
template <typename T> static inline T __ldg(const T* p) { return *p; }

// ---- host runtime API subset -------------------------------------------------------------------
typedef int hipError_t;
enum {
    hipSuccess = 0,
    hipErrorInvalidValue = 1,
    hipErrorOutOfMemory = 2,
    hipErrorNoDevice = 100,
    hipErrorNotSupported = 801,
    hipErrorStreamCaptureUnsupported = 900,
};
typedef struct hipsimStream* hipStream_t;
typedef struct hipsimEvent* hipEvent_t;
typedef struct hipsimGraph* hipGraph_t;
typedef struct hipsimGraphExec* hipGraphExec_t;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum hipStreamCaptureMode { hipStreamCaptureModeGlobal, hipStreamCaptureModeThreadLocal, hipStreamCaptureModeRelaxed };
enum { hipStreamNonBlocking = 1, hipHostMallocDefault = 0, hipEventDefault = 0, hipEventDisableTiming = 2 };

struct hipDeviceProp_t {
    char name[256];
    char gcnArchName[256];
    int multiProcessorCount;
    size_t totalGlobalMem;
    int clockRate;
};

extern "C" {
hipError_t hipMalloc(void** p, size_t n);
hipError_t hipFree(void* p);
hipError_t hipHostMalloc(void** p, size_t n, unsigned flags);
enum hipMemoryType { hipMemoryTypeUnregistered = 0, hipMemoryTypeHost = 1, hipMemoryTypeDevice = 2 };
struct hipPointerAttribute_t { hipMemoryType type; };
static inline hipError_t hipPointerGetAttributes(hipPointerAttribute_t* a, const void*) { a->type = hipMemoryTypeUnregistered; return hipSuccess; }
hipError_t hipHostFree(void* p);
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind k);
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind k, hipStream_t st);
hipError_t hipMemset(void* d, int v, size_t n);
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t st);
hipError_t hipStreamCreate(hipStream_t* s);
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned flags);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags);
hipError_t hipDeviceSynchronize(void);
hipError_t hipGetLastError(void);
hipError_t hipPeekAtLastError(void);
const char* hipGetErrorString(hipError_t e);
hipError_t hipSetDevice(int d);
hipError_t hipGetDevice(int* d);
hipError_t hipGetDeviceCount(int* n);
hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int d);
hipError_t hipEventCreate(hipEvent_t* e);
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned flags);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b);
hipError_t hipStreamBeginCapture(hipStream_t s, hipStreamCaptureMode m);
hipError_t hipStreamEndCapture(hipStream_t s, hipGraph_t* g);
hipError_t hipGraphInstantiate(hipGraphExec_t* e, hipGraph_t g, void*, void*, size_t);
hipError_t hipGraphLaunch(hipGraphExec_t e, hipStream_t s);
hipError_t hipGraphDestroy(hipGraph_t g);
hipError_t hipGraphExecDestroy(hipGraphExec_t e);
}
// fine-grained device memory: not simulated -- the engine falls back to its sub-batch path when the allocation is refused
enum { hipDeviceMallocFinegrained = 1 };
static inline hipError_t hipExtMallocWithFlags(void**, size_t, unsigned) { return hipErrorNotSupported; }
static inline hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
static inline hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }

template <typename... KArgs, typename... Args>
static inline void hipLaunchKernelGGL(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t /*shmem*/, hipStream_t /*stream*/,
                                      Args&&... args) {
    std::function<void()> body = [&]() { kernel(static_cast<KArgs>(args)...); };
    ::hipsim::run_grid(grid, block, body);
}
