// ade_fft.h — mixed-radix Stockham FFT of one frame held in LDS, executed by a whole workgroup (radices 2, 3, 4, 5).
//
// The reference computes every STFT / ISTFT as a dense windowed DFT (a strided Conv1d with a (2F, 1, N) kernel, e.g. DFSMN/STFT_Process.py); for the
// frame sizes its models use -- 1920 = 2^7 x 3 x 5 and 2048 (DFSMN), 400 = 2^4 x 5^2, 512 -- the same transform is a handful of butterfly passes:
// N = 1920 costs ~0.14 MFLOP per frame as an FFT against 7.4 MFLOP as the dense product.  GTCRN's 512-point frames already run as a register /
// shuffle FFT inside its fused kernel (ade_device.h::fft256_inplace); this header is the general form for the other families.
//
// Autosort (Stockham) passes ping-pong between two LDS buffers of N complex values, so no bit / digit reversal is needed.  Pass with radix R over
// sub-transforms of current length Ns (the product of the earlier radices): butterfly j in [0, N / R), k = j mod Ns
//     v[u] = in[j + u N / R] * w_N^(u k N / (Ns R)),  u < R;   v <- DFT_R(v);   out[(j - k) R + k + u Ns] = v[u]
// Twiddles come from one table w_N^m = exp(-2 pi i m / N), m < N (computed in double on the host).  The inverse transform conjugates on the way in and out.
#pragma once
#include "ade_device.h"

namespace ade {
namespace fft {

using namespace dev;

constexpr int kMaxPasses = 8;
struct Plan {                      // passed by value to kernels
    int n = 0, passes = 0;
    int radix[kMaxPasses] = {};
};
// factorisation into radices 4, 2, 3, 5; returns false when n has another prime factor
inline bool make_plan(int n, Plan* p) {
    p->n = n;
    p->passes = 0;
    int m = n;
    while (m % 4 == 0 && p->passes < kMaxPasses) { p->radix[p->passes++] = 4; m /= 4; }
    while (m % 2 == 0 && p->passes < kMaxPasses) { p->radix[p->passes++] = 2; m /= 2; }
    while (m % 3 == 0 && p->passes < kMaxPasses) { p->radix[p->passes++] = 3; m /= 3; }
    while (m % 5 == 0 && p->passes < kMaxPasses) { p->radix[p->passes++] = 5; m /= 5; }
    return m == 1 && n >= 2;
}

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 mul_mi(float2 a) { return make_float2(a.y, -a.x); }          // -i a

__device__ __forceinline__ void dft2(float2* v) {
    const float2 a = v[0];
    v[0] = cadd(a, v[1]);
    v[1] = csub(a, v[1]);
}
__device__ __forceinline__ void dft4(float2* v) {
    const float2 a0 = cadd(v[0], v[2]), a1 = csub(v[0], v[2]), a2 = cadd(v[1], v[3]), a3 = mul_mi(csub(v[1], v[3]));
    v[0] = cadd(a0, a2); v[1] = cadd(a1, a3); v[2] = csub(a0, a2); v[3] = csub(a1, a3);
}
__device__ __forceinline__ void dft3(float2* v) {          // w = exp(-2 pi i / 3) = -1/2 - i sqrt(3)/2
    const float2 s = cadd(v[1], v[2]), d = csub(v[1], v[2]);
    const float2 m = make_float2(v[0].x - 0.5f * s.x, v[0].y - 0.5f * s.y);
    const float2 r = make_float2(0.86602540378443864676f * d.y, -0.86602540378443864676f * d.x);      // -i sqrt(3)/2 d
    v[0] = cadd(v[0], s);
    v[1] = cadd(m, r);
    v[2] = csub(m, r);
}
__device__ __forceinline__ void dft5(float2* v) {          // exp(-2 pi i k / 5): c1 = cos(72), c2 = cos(144), s1 = sin(72), s2 = sin(144)
    constexpr float c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f, s1 = 0.95105651629515357212f, s2 = 0.58778525229247312917f;
    const float2 a1 = cadd(v[1], v[4]), b1 = csub(v[1], v[4]), a2 = cadd(v[2], v[3]), b2 = csub(v[2], v[3]);
    const float2 m1 = make_float2(v[0].x + c1 * a1.x + c2 * a2.x, v[0].y + c1 * a1.y + c2 * a2.y);
    const float2 m2 = make_float2(v[0].x + c2 * a1.x + c1 * a2.x, v[0].y + c2 * a1.y + c1 * a2.y);
    const float2 n1 = mul_mi(make_float2(s1 * b1.x + s2 * b2.x, s1 * b1.y + s2 * b2.y));            // -i (s1 b1 + s2 b2)
    const float2 n2 = mul_mi(make_float2(s2 * b1.x - s1 * b2.x, s2 * b1.y - s1 * b2.y));            // -i (s2 b1 - s1 b2)
    v[0] = cadd(v[0], cadd(a1, a2));
    v[1] = cadd(m1, n1);
    v[4] = csub(m1, n1);
    v[2] = cadd(m2, n2);
    v[3] = csub(m2, n2);
}

template <int R>
__device__ __forceinline__ void pass(const float2* __restrict__ in, float2* __restrict__ out, int n, int ns, const float2* __restrict__ tw, int tid, int nthreads) {
    const int nb = n / R, tstep = n / (ns * R);
    for (int j = tid; j < nb; j += nthreads) {
        const int k = j % ns;
        float2 v[R];
#pragma unroll
        for (int u = 0; u < R; ++u) {
            v[u] = in[j + u * nb];
            if (u > 0 && k > 0) v[u] = cmul(v[u], tw[u * k * tstep]);
        }
        if (R == 2) dft2(v);
        if (R == 3) dft3(v);
        if (R == 4) dft4(v);
        if (R == 5) dft5(v);
        const int j0 = (j - k) * R + k;
#pragma unroll
        for (int u = 0; u < R; ++u) out[j0 + u * ns] = v[u];
    }
}

// Forward DFT of the n complex values in `a` (LDS), in place semantics: returns the buffer (a or b) that holds the result.  All threads of the
// workgroup must call it; a and b must be distinct LDS buffers of n float2.  The caller synchronises before reading the result.
__device__ __forceinline__ float2* forward(float2* a, float2* b, const Plan& p, const float2* __restrict__ tw, int tid, int nthreads) {
    int ns = 1;
    float2 *in = a, *out = b;
    for (int i = 0; i < p.passes; ++i) {
        __syncthreads();
        const int r = p.radix[i];
        if (r == 4) pass<4>(in, out, p.n, ns, tw, tid, nthreads);
        else if (r == 2) pass<2>(in, out, p.n, ns, tw, tid, nthreads);
        else if (r == 3) pass<3>(in, out, p.n, ns, tw, tid, nthreads);
        else pass<5>(in, out, p.n, ns, tw, tid, nthreads);
        ns *= r;
        float2* t = in; in = out; out = t;
    }
    __syncthreads();
    return in;
}

// ---- the same passes with the transform size and its radices known at compile time (the starred models' sizes, and two small ones the host-simulator tests use): trip counts,
// `j % ns` and the twiddle strides fold into constants, the pass loop unrolls.  The factorisation is make_plan's (4s, then 2s, 3s, 5s), so a size's results are bit-identical
// to the run-time plan's.
template <int N> struct Static { static constexpr bool known = false; };
#define ADE_FFT_STATIC(NN, P, R0, R1, R2, R3, R4, R5)                                                                            \
    template <> struct Static<NN> {                                                                                             \
        static constexpr bool known = true;                                                                                     \
        static constexpr int passes = P;                                                                                        \
        static constexpr int radix(int i) { return i == 0 ? R0 : (i == 1 ? R1 : (i == 2 ? R2 : (i == 3 ? R3 : (i == 4 ? R4 : R5)))); }   \
    }
ADE_FFT_STATIC(512, 5, 4, 4, 4, 4, 2, 1);
ADE_FFT_STATIC(400, 4, 4, 4, 5, 5, 1, 1);
ADE_FFT_STATIC(2048, 6, 4, 4, 4, 4, 4, 2);
ADE_FFT_STATIC(1920, 6, 4, 4, 4, 2, 3, 5);
ADE_FFT_STATIC(64, 3, 4, 4, 4, 1, 1, 1);
ADE_FFT_STATIC(60, 3, 4, 3, 5, 1, 1, 1);
#undef ADE_FFT_STATIC
inline bool static_size(int n) { return n == 512 || n == 400 || n == 2048 || n == 1920 || n == 64 || n == 60; }

template <int R, int N, int NS>
__device__ __forceinline__ void pass_static(const float2* __restrict__ in, float2* __restrict__ out, const float2* __restrict__ tw, int tid, int nthreads) {
    constexpr int nb = N / R, tstep = N / (NS * R);
    for (int j = tid; j < nb; j += nthreads) {
        const int k = j % NS;
        float2 v[R];
#pragma unroll
        for (int u = 0; u < R; ++u) {
            v[u] = in[j + u * nb];
            if (u > 0 && NS > 1 && k > 0) v[u] = cmul(v[u], tw[u * k * tstep]);
        }
        if (R == 2) dft2(v);
        if (R == 3) dft3(v);
        if (R == 4) dft4(v);
        if (R == 5) dft5(v);
        const int j0 = (j - k) * R + k;
#pragma unroll
        for (int u = 0; u < R; ++u) out[j0 + u * NS] = v[u];
    }
}
// Forward DFT of the N values in `a`; the result is in `a` when Static<N>::passes is even, in `b` otherwise (result_in_first<N>()).  Every thread of the workgroup calls it
// (a barrier precedes each pass); tid / nthreads = the caller's index inside the thread group that shares this transform.
template <int N> constexpr bool result_in_first() { return Static<N>::passes % 2 == 0; }
template <int N, int I = 0, int NS = 1>
__device__ __forceinline__ void forward_static(float2* a, float2* b, const float2* __restrict__ tw, int tid, int nthreads) {
    if constexpr (I < Static<N>::passes) {
        __syncthreads();
        constexpr int R = Static<N>::radix(I);
        if (I % 2 == 0) pass_static<R, N, NS>(a, b, tw, tid, nthreads);
        else pass_static<R, N, NS>(b, a, tw, tid, nthreads);
        forward_static<N, I + 1, NS * R>(a, b, tw, tid, nthreads);
    }
}

// ---- the same passes executed by ONE wavefront on a buffer only it touches, IN PLACE: every butterfly of a pass is read into registers before any is written back (LDS
// serves a wavefront's accesses in program order; wave_sync() keeps the compiler -- and the host simulator's lanes -- to that order), so neither a ping-pong buffer nor a
// workgroup barrier is needed.  Same butterflies, twiddles and sums as pass_static: bit-identical results.
template <int R, int N, int NS>
__device__ __forceinline__ void pass_wave(float2* __restrict__ buf, const float2* __restrict__ tw, int lane) {
    constexpr int nb = N / R, tstep = N / (NS * R), KB = (nb + 63) / 64;
    float2 v[KB][R];
#pragma unroll
    for (int b = 0; b < KB; ++b) {
        const int j = lane + 64 * b;
        if (nb % 64 == 0 || j < nb) {
            const int k = j % NS;
#pragma unroll
            for (int u = 0; u < R; ++u) {
                v[b][u] = buf[j + u * nb];
                if (u > 0 && NS > 1 && k > 0) v[b][u] = cmul(v[b][u], tw[u * k * tstep]);
            }
        }
    }
    wave_sync();
#pragma unroll
    for (int b = 0; b < KB; ++b) {
        const int j = lane + 64 * b;
        if (nb % 64 == 0 || j < nb) {
            const int k = j % NS;
            if (R == 2) dft2(v[b]);
            if (R == 3) dft3(v[b]);
            if (R == 4) dft4(v[b]);
            if (R == 5) dft5(v[b]);
            const int j0 = (j - k) * R + k;
#pragma unroll
            for (int u = 0; u < R; ++u) buf[j0 + u * NS] = v[b][u];
        }
    }
    wave_sync();
}
template <int N, int I = 0, int NS = 1>
__device__ __forceinline__ void forward_wave(float2* buf, const float2* __restrict__ tw, int lane) {
    if constexpr (I < Static<N>::passes) {
        constexpr int R = Static<N>::radix(I);
        pass_wave<R, N, NS>(buf, tw, lane);
        forward_wave<N, I + 1, NS * R>(buf, tw, lane);
    }
}

}  // namespace fft
}  // namespace ade
