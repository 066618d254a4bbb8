// ade_device.h — small device-side helpers shared by the gfx950 kernel translation units.
#pragma once

#include "ade_internal.h"

namespace ade {
namespace dev {

// sigmoid / tanh on the hardware exp and reciprocal units (v_exp_f32, v_rcp_f32: ~1 ulp each); the network's
// gates only need ~1e-6 absolute accuracy (parity tolerance 1e-4 on the waveform, observed ~1e-6).
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
// v_sqrt_f32 / v_rsq_f32 as they are (~1 ulp): sqrtf() and 1.0f / sqrtf() expand to the IEEE-correct sequences (scaling, two refinement FMAs, class checks: ~20 instructions
// each), and so does a float division (~10).  None of their callers (magnitudes >= 1e-6, LayerNorm variances >= 1e-8, window sums ~1) needs the last bit or the range handling.
__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float fast_rsq(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float sigmoid_f(float x) { return fast_rcp(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanh_f(float x) {
    // 1 - 2/(e^{2x}+1): saturates cleanly to +-1 for large |x|
    return 1.0f - 2.0f * fast_rcp(__expf(2.0f * x) + 1.0f);
}
__device__ __forceinline__ float prelu_f(float x, float a) { return x >= 0.0f ? x : a * x; }
// butterfly exchange of a double as its two 32-bit halves (what __shfl_xor(double) does on the GPU; the host simulator's shuffles are 32-bit)
__device__ __forceinline__ double shfl_xor_f64(double v, int mask) {
    union { double d; int i[2]; } u;
    u.d = v;
    u.i[0] = __shfl_xor(u.i[0], mask, 64);
    u.i[1] = __shfl_xor(u.i[1], mask, 64);
    return u.d;
}
// The same function as ONE v_med3_f32 per element (the product packs into v_pk_mul_f32): for a slope <= 1 PReLU is max(x, a x), for a slope > 1 it is
// min(x, a x); median(x, a x, +inf) is the first and median(x, a x, -inf) the second, so the slope picks a wave-uniform third operand once and the
// body has no branch and no compare + select pair.  Equal to prelu_f for every finite x (a negative slope turns -0 into +0, which compares equal).
__device__ __forceinline__ float prelu_sel(float a) { return a <= 1.0f ? __builtin_inff() : -__builtin_inff(); }
__device__ __forceinline__ float prelu_m(float x, float ax, float sel) { return __builtin_amdgcn_fmed3f(x, ax, sel); }

__device__ __forceinline__ void ld4(const float* p, float* v) {
    const float4 q = *reinterpret_cast<const float4*>(p);
    v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
}
__device__ __forceinline__ void ld8(const float* p, float* v) { ld4(p, v); ld4(p + 4, v + 4); }
__device__ __forceinline__ void ld16(const float* p, float* v) { ld8(p, v); ld8(p + 8, v + 8); }
__device__ __forceinline__ void st4(float* p, const float* v) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void st16(float* p, const float* v) { st4(p, v); st4(p + 4, v + 4); st4(p + 8, v + 8); st4(p + 12, v + 12); }

// channels [0,8) of a (possibly gated) activation at position `pos` of frame `frame`
__device__ __forceinline__ void view_ld8_lo(const View& a, size_t pos, size_t frame, float* v) {
    ld8(a.x + pos * kCh, v);
    if (a.at) {
        float g[4];
        ld4(a.at + frame * 8, g);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[2 * i] *= g[i];
    }
}
// channels [8,16)
__device__ __forceinline__ void view_ld8_hi(const View& a, size_t pos, size_t frame, float* v) {
    ld8(a.x + pos * kCh + 8, v);
    if (a.at) {
        float g[4];
        ld4(a.at + frame * 8 + 4, g);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[2 * i] *= g[i];
    }
}
__device__ __forceinline__ void view_ld16(const View& a, size_t pos, size_t frame, float* v) {
    view_ld8_lo(a, pos, frame, v);
    view_ld8_hi(a, pos, frame, v + 8);
}

// DPP cross-lane moves (one VALU op, no LDS round trip).  CTRL: quad_perm = p0|p1<<2|p2<<4|p3<<6 ; row_newbcast:n = 0x150+n
// (gfx90a+: lane n of each 16-lane row to the whole row).
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
template <int K> __device__ __forceinline__ float row_bcast(float v) { return dpp_mov<0x150 + K>(v); }    // K < 16
template <int K> __device__ __forceinline__ float quad_bcast(float v) { return dpp_mov<K * 0x55>(v); }   // K < 4
template <int N> __device__ __forceinline__ float row_ror(float v) { return dpp_mov<0x120 + N>(v); }     // rotate within the 16-lane row, 1 <= N <= 15
template <int N> __device__ __forceinline__ float quad_rot(float v) {                                      // lane u of a quad reads lane (u+N)&3
    return dpp_mov<((N) & 3) | (((N + 1) & 3) << 2) | (((N + 2) & 3) << 4) | (((N + 3) & 3) << 6)>(v);
}

// Cross-row moves inside a wavefront (gfx950 v_permlane16_swap_b32 / v_permlane32_swap_b32).  With 16-lane rows r0..r3:
//   swap16(v, s): .a = [v.r0, s.r0, v.r2, s.r2]   .b = [v.r1, s.r1, v.r3, s.r3]
//   swap32(v, s): .a = [v.r0, v.r1, s.r0, s.r1]   .b = [v.r2, v.r3, s.r2, s.r3]
struct Swapped { float a, b; };
__device__ __forceinline__ Swapped swap16(float v, float s) {
    const auto r = __builtin_amdgcn_permlane16_swap((unsigned)__float_as_int(v), (unsigned)__float_as_int(s), false, false);
    return Swapped{__int_as_float((int)r[0]), __int_as_float((int)r[1])};
}
__device__ __forceinline__ Swapped swap32(float v, float s) {
    const auto r = __builtin_amdgcn_permlane32_swap((unsigned)__float_as_int(v), (unsigned)__float_as_int(s), false, false);
    return Swapped{__int_as_float((int)r[0]), __int_as_float((int)r[1])};
}

// Value of lane LANE for the whole wavefront (v_readlane_b32: lands in an SGPR, i.e. a scalar operand of later VALU ops).
template <int LANE> __device__ __forceinline__ float read_lane(float v) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), LANE));
}
constexpr float kLog2e = 1.4426950408889634f;

// fp32 matrix-core tile (v_mfma_f32_16x16x4_f32, exact f32 = a k-ordered fmaf chain, 32 cycles):  D(16x16) += A(16x4) B(4x16)
//   A operand: one float per lane, lane l holds A[i = l & 15][k = l >> 4]
//   B operand: one float per lane, lane l holds B[k = l >> 4][j = l & 15]
//   C / D    : four floats per lane, lane l holds D[i = 4 (l >> 4) + r][j = l & 15], r = 0..3
// With i = output channel and j = position, a lane's four results are the channel quad (l >> 4) of position (l & 15):
// exactly one float4 of the channel-quad planar LDS layout.
#if defined(__clang__)
typedef float v4f __attribute__((ext_vector_type(4)));
#else
typedef float v4f __attribute__((vector_size(16)));
#endif
__device__ __forceinline__ v4f mfma16x16x4(float a, float b, v4f c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// Two-wide fp32 vector: arithmetic on it is what becomes v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 (the packed fp32 ops
// the 157 TFLOP/s fp32 peak is quoted on).  The recurrent loops are VALU-issue-bound, so halving the FMA instruction
// count is a direct win.  (g++ spelling for the host-side simulator build under tests/hipsim.)
#if defined(__clang__)
typedef float v2f __attribute__((ext_vector_type(2)));
#else
typedef float v2f __attribute__((vector_size(8)));
#endif
__device__ __forceinline__ v2f mk2(float a, float b) { v2f r; r[0] = a; r[1] = b; return r; }

// Wave-uniform, read-only tables (conv / Linear weights): viewing them through the CONSTANT address space lets the
// compiler fetch them with scalar loads (s_load through the scalar cache -> SGPR operands of v_fma) even when the
// pointer arrived inside a by-value struct, where no __restrict__/noalias information survives and the loads would
// otherwise become per-lane global_loads (and, hoisted out of loops, spill to scratch).  The engine never writes the
// weight arena while a kernel runs, which is what the constant address space asserts.
#if defined(__has_attribute)
#if __has_attribute(address_space)
#define ADE_CONSTANT_AS __attribute__((address_space(4)))
#endif
#endif
#ifndef ADE_CONSTANT_AS
#define ADE_CONSTANT_AS
#endif
// Optimisation barrier on a (wave-uniform) pointer: the compiler must treat it as changed, so loads through it stay
// inside the loop iteration that uses them instead of being hoisted out as 300+ live SGPRs (which then spill to
// VGPR lanes, one v_readlane per use).  Emits no instruction.
#if defined(__AMDGCN__)
#define ADE_KEEP_IN_LOOP(p) asm volatile("" : "+s"(p))
#else
#define ADE_KEEP_IN_LOOP(p) ((void)0)
#endif
// Same, additionally ordered after the instruction that produced `tok` (a VGPR value): used to pace a fully unrolled
// sequence of scalar weight-row fetches one row ahead of the FMAs instead of all up front (where the rows would not fit
// the SGPR file and would be parked in VGPR lanes, one v_readlane per use).  Emits no instruction.
#if defined(__AMDGCN__)
#define ADE_KEEP_AFTER(p, tok) asm volatile("" : "+s"(p), "+v"(tok))
#else
#define ADE_KEEP_AFTER(p, tok) ((void)0)
#endif
// Fresh scalar copy of a 64-bit pointer that arrived inside a wide kernel-argument tuple (one s_load_dwordx16 for a whole
// by-value struct): the register allocator tracks such a tuple as ONE live range, so keeping a member alive keeps --
// and spills / restores -- all sixteen registers.  One s_mov_b64.
#if defined(__AMDGCN__)
#define ADE_SCALAR_COPY(dst, src) asm volatile("s_mov_b64 %0, %1" : "=s"(dst) : "s"(src))
#else
#define ADE_SCALAR_COPY(dst, src) ((dst) = (src))
#endif
// Opaque per-lane value: everything derived from it is private to the code that follows, so the optimiser cannot
// share (and keep live) index arithmetic across the stages inlined into one kernel.  Emits no instruction.
#if defined(__AMDGCN__)
#define ADE_OPAQUE_V(x) asm volatile("" : "+v"(x))
#else
#define ADE_OPAQUE_V(x) ((void)0)
#endif
// A plain-data struct fetched through the constant address space (scalar loads), dword by dword: C++ has no copy constructor across address spaces.
template <class T>
__device__ __forceinline__ T cload(const void ADE_CONSTANT_AS* p) {
    static_assert(sizeof(T) % 4 == 0, "dword-sized structs only");
    union U { T v; unsigned w[sizeof(T) / 4]; __device__ U() {} } u;
    const unsigned ADE_CONSTANT_AS* s = (const unsigned ADE_CONSTANT_AS*)p;
#pragma unroll
    for (size_t i = 0; i < sizeof(T) / 4; ++i) u.w[i] = s[i];
    return u.v;
}
typedef const float ADE_CONSTANT_AS* cfptr;
__device__ __forceinline__ cfptr cptr(const float* p) { return (cfptr)p; }

// ---- 256-point complex FFT of one wavefront, ONE 256-entry LDS buffer (in place) ---------------------------------
// Radix-4 Stockham: lane holds z[lane + 64 r] on entry, Z[lane + 64 r] (natural order) on exit.  Every pass reads its
// four inputs into registers, then (after a barrier) scatters its outputs into the same buffer, so half the LDS of the
// ping-pong form is needed.  The buffer is private to the wavefront, and a wavefront's LDS operations execute in
// program order, so the only ordering needed between a pass's scatter and the next pass's gather is "do not let the
// compiler move them": wave_sync() (no instruction on the GPU; a real wave barrier under the host simulator).
__device__ __forceinline__ void wave_sync() { __builtin_amdgcn_wave_barrier(); }
__device__ __forceinline__ float2 fcmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ void fradix4(float2* v) {
    const float2 a0 = make_float2(v[0].x + v[2].x, v[0].y + v[2].y);
    const float2 a1 = make_float2(v[0].x - v[2].x, v[0].y - v[2].y);
    const float2 a2 = make_float2(v[1].x + v[3].x, v[1].y + v[3].y);
    const float2 d = make_float2(v[1].x - v[3].x, v[1].y - v[3].y);
    const float2 a3 = make_float2(d.y, -d.x);   // -i * d
    v[0] = make_float2(a0.x + a2.x, a0.y + a2.y);
    v[1] = make_float2(a1.x + a3.x, a1.y + a3.y);
    v[2] = make_float2(a0.x - a2.x, a0.y - a2.y);
    v[3] = make_float2(a1.x - a3.x, a1.y - a3.y);
}
__device__ __forceinline__ void fft256_inplace(float2* v, float2* buf, int lane, const float2* __restrict__ tw256) {
    fradix4(v);                                   // pass 0: Ns = 1, no twiddle
#pragma unroll
    for (int r = 0; r < 4; ++r) buf[4 * lane + r] = v[r];
    wave_sync();
#pragma unroll
    for (int pass = 1; pass < 4; ++pass) {
        const int Ns = 1 << (2 * pass);           // 4, 16, 64
        const int k = lane & (Ns - 1);
        const int tstride = 64 / Ns;              // angle -2 pi r k / (4 Ns) -> tw256[r k 64 / Ns]
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = buf[lane + 64 * r];
#pragma unroll
        for (int r = 1; r < 4; ++r) v[r] = fcmul(v[r], tw256[r * k * tstride]);
        fradix4(v);
        if (pass < 3) {
            wave_sync();                          // every lane has read this pass's inputs
            const int j0 = ((lane - k) << 2) + k;
#pragma unroll
            for (int r = 0; r < 4; ++r) buf[j0 + r * Ns] = v[r];
            wave_sync();
        }
    }
}

// keep(ok, v) = ok ? v : 0 per component.  (A ?: between two float4 OBJECTS makes the compiler select between their addresses and park both on the stack.)
__device__ __forceinline__ float4 keep4(bool ok, const float4& v) { return make_float4(ok ? v.x : 0.0f, ok ? v.y : 0.0f, ok ? v.z : 0.0f, ok ? v.w : 0.0f); }
// ld4_or_zero(ok, p): *p if ok, else zeros -- WITHOUT touching the loaded value: the select is made on the ADDRESS (a 16-byte block of zeros stands in), so a prefetch
// written with it can stay in flight across the matrix work that follows.  keep4(ok, *p) selects on the VALUE, which makes the compiler wait for the load right there.
__device__ __attribute__((aligned(16))) static float g_zero4[4] = {0.0f, 0.0f, 0.0f, 0.0f};       // never written
__device__ __forceinline__ float4 ld4_or_zero(bool ok, const float* p) { return *reinterpret_cast<const float4*>(ok ? p : g_zero4); }
__device__ __forceinline__ float4 pick4(bool first, const float4& a, const float4& b) {
    return make_float4(first ? a.x : b.x, first ? a.y : b.y, first ? a.z : b.z, first ? a.w : b.w);
}

// ---- exchange between workgroups of ONE launch (segments of a chunk hand their recurrent states / convolution histories on) -------------
// Per-XCD L2s are not coherent with each other and a CU's vector L1 is never refreshed by another CU's stores, so every shared word moves
// with agent-scope accesses: payload as write-through (sc1) stores, then EVERY storing wave drains its stores (xdrain), a workgroup barrier,
// ONE lane raises the flag; the consumer polls that one word (relaxed, sc1), a workgroup barrier, then sc1 loads (they bypass the L1, so no
// acquire fence is needed).  Placement-independent: nothing here assumes a dispatch order or a workgroup -> XCD map.
#if defined(__AMDGCN__)
typedef unsigned ade_v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 xld4(const float* base, int float_off) {      // 16-B sc1 load of base[float_off .. +3]
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 0x7fffffff, 0x00020000);
    const ade_v4u v = __builtin_amdgcn_raw_buffer_load_b128(r, float_off * 4, 0, 16);
    return make_float4(__int_as_float((int)v.x), __int_as_float((int)v.y), __int_as_float((int)v.z), __int_as_float((int)v.w));
}
__device__ __forceinline__ void xst4(float* base, int float_off, const float4& v) {   // 16-B sc1 (write-through) store
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7fffffff, 0x00020000);
    ade_v4u u;
    u.x = (unsigned)__float_as_int(v.x); u.y = (unsigned)__float_as_int(v.y); u.z = (unsigned)__float_as_int(v.z); u.w = (unsigned)__float_as_int(v.w);
    __builtin_amdgcn_raw_buffer_store_b128(u, r, float_off * 4, 0, 16);
}
__device__ __forceinline__ float xld1(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void xst1(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned xflag_load(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void xflag_store(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void xdrain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#else
__device__ __forceinline__ float4 xld4(const float* base, int float_off) { return *reinterpret_cast<const float4*>(base + float_off); }
__device__ __forceinline__ void xst4(float* base, int float_off, const float4& v) { *reinterpret_cast<float4*>(base + float_off) = v; }
__device__ __forceinline__ float xld1(const float* p) { return *p; }
__device__ __forceinline__ void xst1(float* p, float v) { *p = v; }
__device__ __forceinline__ unsigned xflag_load(const unsigned* p) { return *(const volatile unsigned*)p; }
__device__ __forceinline__ void xflag_store(unsigned* p, unsigned v) { *(volatile unsigned*)p = v; }
__device__ __forceinline__ void xdrain() {}
#endif
// ONE lane polls ONE flag until it is raised, then lowers it again for the next launch (each flag has exactly one consumer; the kernel
// boundary orders that store before the next launch's producer).  Bounded: `limit` 10 ns ticks (option "xwait_ms", default 0.2 s; a kernel argument -- the wait path
// reads nothing but the flag and, once the flag is found down, that argument).  The FIRST wait of a launch that gives up leaves `code` (xcode(): block index and flag index of the waiting segment) in the
// page-locked host word *err and lets the workgroup run on instead of hanging the device; it does NOT lower the flag.  The engine then fails the call with
// ADE_ERR_DEVICE -- no PCM is handed out -- and clears every flag before the next launch (exchange_status() in ade_engine.hip).  A wait that has lasted a sixteenth
// of the bound looks at *err once (a PCIe round trip; never in a healthy launch, whose waits last microseconds): when the launch has already failed it gives up at
// once, so the failure of one hand-off costs its successors a fraction of the bound per wait, not the whole of it.
// The bound of a wait (SegPlan::wait_ticks): the first word of the kernel-argument segment, fetched where it is needed (a scalar load) instead of living in a register.
#if defined(__AMDGCN__)
__device__ __forceinline__ int xlimit() {
    const int ADE_CONSTANT_AS* p = (const int ADE_CONSTANT_AS*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));          // (opaque: the load stays at the wait, it is not hoisted to the top of the stage)
    return *p;
}
#else
__device__ __forceinline__ int xlimit() { return 20000000; }      // (host simulator: its clock advances 1000 ticks per read, so any bound ends within milliseconds)
#endif
__device__ __forceinline__ int xcode(int flag_index) { return (int)((blockIdx.x + 1u) << 4) | flag_index; }
__device__ __forceinline__ void xwait(unsigned* flag, int* err, int code) {
    if (xflag_load(flag) == 0u) {
        const long long t0 = wall_clock64();
        const int limit = xlimit();
        bool looked = false;
        do {
            __builtin_amdgcn_s_sleep(4);
            const long long dt = wall_clock64() - t0;
            if (dt > (limit >> 4) && !looked) {
                looked = true;
                if (*reinterpret_cast<volatile int*>(err) != 0) return;
            }
            if (dt > limit) {
                volatile int* const e = reinterpret_cast<volatile int*>(err);
                if (*e == 0) *e = code;
                return;
            }
        } while (xflag_load(flag) == 0u);
    }
    xflag_store(flag, 0u);
}

// ---- a launch that a host batch streams through (ChunkCall::in_ready / out_done): system-scope words in fine-grained memory, written / polled by the command processor
#if defined(__AMDGCN__)
__device__ __forceinline__ unsigned sys_load(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void sys_store(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void sys_acquire() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, ""); }
__device__ __forceinline__ void sys_release() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, ""); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#else
__device__ __forceinline__ unsigned sys_load(const unsigned* p) { return *(const volatile unsigned*)p; }
__device__ __forceinline__ void sys_store(unsigned* p, unsigned v) { *(volatile unsigned*)p = v; }
__device__ __forceinline__ void sys_acquire() {}
__device__ __forceinline__ void sys_release() {}
#endif
// ONE lane waits until the copy engine has delivered this workgroup's rows (bounded like xwait: a copy that never arrives fails the call, it does not hang the device),
// then drops whatever this CU and its L2 still hold of the previous call's bytes at those addresses; the caller follows with a workgroup barrier.
__device__ __forceinline__ void wait_rows_in(const unsigned* ready, unsigned epoch, int* err, int code) {
    if (sys_load(ready) != epoch) {
        const long long t0 = wall_clock64();
        const int limit = xlimit();
        do {
            __builtin_amdgcn_s_sleep(8);
            if (wall_clock64() - t0 > limit) {
                volatile int* const e = reinterpret_cast<volatile int*>(err);
                if (*e == 0) *e = code;
                break;
            }
        } while (sys_load(ready) != epoch);
    }
    sys_acquire();
}
// Every wave has drained its stores and the workgroup has met at a barrier: ONE lane writes this XCD's dirty lines back (the copy engine reads memory, not an L2),
// counts the workgroup in, and the group's last workgroup tells the copy-out stream.
__device__ __forceinline__ void signal_rows_out(unsigned* count, unsigned* done, unsigned epoch, unsigned wgs) {
    sys_release();
#if defined(__AMDGCN__)
    const unsigned old = __hip_atomic_fetch_add(count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
    const unsigned old = (*count)++;
#endif
    if (old + 1u == wgs) {
        xflag_store(count, 0u);
        sys_store(done, epoch);
    }
}

// s_setprio takes an immediate: a wave-uniform level goes through a scalar branch chain
__device__ __forceinline__ void set_prio(int p) {
    if (p == 0) __builtin_amdgcn_s_setprio(0);
    else if (p == 1) __builtin_amdgcn_s_setprio(1);
    else if (p == 2) __builtin_amdgcn_s_setprio(2);
    else __builtin_amdgcn_s_setprio(3);
}

inline dim3 grid1(long long n, int per) { return dim3((unsigned)((n + per - 1) / per)); }

}  // namespace dev
}  // namespace ade
