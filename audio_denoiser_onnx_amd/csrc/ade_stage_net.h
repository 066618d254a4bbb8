// ade_stage_net.h — per-chunk, LDS-resident stage kernels (the fast path for 1 s chunks, T <= 64 frames).
//
// Design: ONE workgroup (1024 threads = 16 wavefronts) owns ONE audio chunk for a whole network stage and keeps
// the stage's (T,33,16) fp32 activation (133 KB for T = 63) in the CU's 160 KB LDS, so that a GTConvBlock or a
// DPGRNN block reads its input from HBM once and writes its output once; everything in between (pointwise ->
// dilated depthwise -> pointwise -> energy -> TRA GRU -> gate, or GRU -> Linear -> LayerNorm -> GRU -> Linear ->
// LayerNorm) happens in LDS / registers.  At batch = 256 chunks this is exactly one workgroup per CU of the
// MI355X (256 CUs); chunks are independent, so there is no inter-workgroup traffic at all.
//  * conv phases: one lane per (t,f) position, weights wave-uniform through the scalar cache;
//  * LDS activation layout is channel-quad planar  H[q][p] = float4(channels 4q..4q+3 of position p)  so that
//    consecutive lanes touch consecutive 16-byte slots (conflict-free ds_read_b128 / ds_write_b128);
//  * recurrences: one lane per hidden unit, hidden vector exchanged with DPP (quad_perm / row_newbcast, one
//    VALU op each, no LDS round trip); the input projections and the TRA Linear are hoisted out of the serial loop.
// Arithmetic reproduced: see the per-phase citations (paths relative to the reference repo).
#pragma once
#include "ade_device.h"

namespace ade {
namespace stage {

using namespace dev;


constexpr int kTmaxFused = 64;
constexpr int kPmax = kTmaxFused * kFw;   // 2112 positions
constexpr int kFusedThreads = 1024;
constexpr int kPosPerThread = 3;          // ceil(2112 / 1024)

__device__ __forceinline__ float comp(const float4& v, int c) { return c == 0 ? v.x : (c == 1 ? v.y : (c == 2 ? v.z : v.w)); }

// Inter-stage tensors of the fused path live in HBM in the same channel-quad planar form as in LDS:
//   X[b][q][p] = float4(channels 4q..4q+3 of position p), p < P positions of the chunk
// so a wavefront whose lanes own consecutive positions moves 1 KB contiguous per instruction (16 B per lane,
// fully coalesced) instead of 64 lanes x 16 B at a 64-byte stride.  Xc = chunk base (X + b * 16 * P).
__device__ __forceinline__ void pl_ld8(const float* Xc, int P, int p, int q0, float* v) {
    ld4(Xc + ((size_t)q0 * P + p) * 4, v);
    ld4(Xc + ((size_t)(q0 + 1) * P + p) * 4, v + 4);
}
__device__ __forceinline__ void pl_ld16(const float* Xc, int P, int p, float* v) {
    pl_ld8(Xc, P, p, 0, v);
    pl_ld8(Xc, P, p, 2, v + 8);
}
__device__ __forceinline__ void pl_st16(float* Xc, int P, int p, const float* v) {
#pragma unroll
    for (int q = 0; q < 4; ++q) st4(Xc + ((size_t)q * P + p) * 4, v + 4 * q);
}

// ---------------------------------------------------------------------------------------------------------
// GTConvBlock, whole block for one chunk (Export_GTCRN.py:303-324 + TRA :144-156).
//   in : a (+ skip), quad-planar (B,4,P,4) ;  out: same layout = interleave(h1 * at, bypass)
// LDS: H[4][kPmax] float4 | zt[64][8] | at[64][8] ; GI[64][48] and HS[64][16] alias H planes 2-3 after phase 2.
// ---------------------------------------------------------------------------------------------------------
constexpr size_t kGtSmemBytes = (size_t)4 * kPmax * 16 + 2 * kTmaxFused * 8 * 4;

#define ADE_REP16(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)

// Optional phase clocks: when `clk` is non-null, thread 0 of workgroup 0 stamps wall_clock64() at each phase boundary.
#define ADE_CLK(i) do { if (clk && chunk == 0 && threadIdx.x == 0) clk[i] = wall_clock64(); } while (0)

// x1_in_lds : the previous stage of the same launch already left this block's pointwise input (a+skip)[:, :8] in LDS planes 2-3.
// next_x1   : leave the NEXT GTConvBlock's pointwise input there (out[:, :8] + next_skip[:, :8]); next_skip may be null.
__device__ __forceinline__ void gtblock_stage(float4* smem, int chunk, const float* __restrict__ a, const float* __restrict__ skip,
                                              const GtConvW& w, float* __restrict__ out, int T, long long* __restrict__ clk,
                                              bool x1_in_lds = false, bool next_x1 = false, const float* __restrict__ next_skip = nullptr) {
    float4* H = smem;
    float* zt = reinterpret_cast<float*>(smem + 4 * kPmax);
    float* at = zt + kTmaxFused * 8;
    float* GI = reinterpret_cast<float*>(smem + 2 * kPmax);
    float* HS = GI + kTmaxFused * 48;
    const int P = T * kFw;
    int tid_ = threadIdx.x;
    ADE_OPAQUE_V(tid_);
    const int tid = tid_;
    const float* ac = a + (size_t)chunk * kCh * P;
    const float* sc = skip ? skip + (size_t)chunk * kCh * P : nullptr;
    float* oc = out + (size_t)chunk * kCh * P;
    const cfptr c_pw1_b = cptr(w.pw1_b), c_dw_b = cptr(w.dw_b), c_pw2_b = cptr(w.pw2_b);
    ADE_CLK(0);

    // ---- phase 0: stage the pointwise input x1 = (a + skip)[:, :8] in LDS planes 2-3 (own position, fully coalesced);
    //      skipped when the previous stage of this launch left it there.
    if (!x1_in_lds) {
        for (int p = tid; p < P; p += kFusedThreads) {
            float x[8];
            pl_ld8(ac, P, p, 0, x);
            if (sc) {
                float y[8];
                pl_ld8(sc, P, p, 0, y);
#pragma unroll
                for (int k = 0; k < 8; ++k) x[k] += y[k];
            }
            H[2 * kPmax + p] = make_float4(x[0], x[1], x[2], x[3]);
            H[3 * kPmax + p] = make_float4(x[4], x[5], x[6], x[7]);
        }
        __syncthreads();
    }
    // ---- phase 1: SFE(3) -> 1x1 (24->16) + BN + PReLU, neighbours from LDS.  Output channels 0-7 go straight to planes 0-1;
    //      channels 8-15 wait in registers until every lane has read its x1 neighbours out of planes 2-3.   (:305-310)
    float hi[kPosPerThread][8];
#pragma unroll
    for (int i = 0; i < kPosPerThread; ++i) {
        const int p = tid + i * kFusedThreads;
        if (p < P) {
            const int t = p / kFw, f = p - t * kFw;
            cfptr c_pw1 = cptr(w.pw1);
            ADE_KEEP_IN_LOOP(c_pw1);
            float acc[16];
#pragma unroll
            for (int co = 0; co < 16; ++co) acc[co] = c_pw1_b[co];
#pragma unroll
            for (int o = 0; o < 3; ++o) {
                const int ff = f - 1 + o;
                if (ff < 0 || ff >= kFw) continue;
                const float4 xa = H[2 * kPmax + p - 1 + o], xb = H[3 * kPmax + p - 1 + o];
                const float x[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
#pragma unroll
                for (int c = 0; c < 8; ++c)
#pragma unroll
                    for (int co = 0; co < 16; ++co) acc[co] += c_pw1[(c * 3 + o) * 16 + co] * x[c];
            }
#pragma unroll
            for (int co = 0; co < 16; ++co) acc[co] = prelu_f(acc[co], w.pw1_slope);
            H[p] = make_float4(acc[0], acc[1], acc[2], acc[3]);
            H[kPmax + p] = make_float4(acc[4], acc[5], acc[6], acc[7]);
#pragma unroll
            for (int k = 0; k < 8; ++k) hi[i][k] = acc[8 + k];
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < kPosPerThread; ++i) {
        const int p = tid + i * kFusedThreads;
        if (p < P) {
            H[2 * kPmax + p] = make_float4(hi[i][0], hi[i][1], hi[i][2], hi[i][3]);
            H[3 * kPmax + p] = make_float4(hi[i][4], hi[i][5], hi[i][6], hi[i][7]);
        }
    }
    __syncthreads();
    ADE_CLK(1);

    // ---- phase 2: causal dilated depthwise 3x3 + BN + PReLU -> 1x1 (16->8) + BN -> h1 (registers)   (:311-320)
    float h1r[kPosPerThread][8];
#pragma unroll
    for (int i = 0; i < kPosPerThread; ++i) {
        const int p = tid + i * kFusedThreads;
        if (p < P) {
            const int t = p / kFw, f = p - t * kFw;
            cfptr c_dw = cptr(w.dw), c_pw2 = cptr(w.pw2);   // per-copy opaque pointers: no cross-copy SGPR hoarding
            ADE_KEEP_IN_LOOP(c_dw);
            ADE_KEEP_IN_LOOP(c_pw2);
            float acc[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) acc[c] = c_dw_b[c];
#pragma unroll
            for (int kt = 0; kt < 3; ++kt) {
                const int tt = t - (2 - kt) * w.dilation;
                if (tt < 0) continue;
#pragma unroll
                for (int kf = 0; kf < 3; ++kf) {
                    const int ff = f - 1 + kf;
                    if (ff < 0 || ff >= kFw) continue;
                    const int pp = tt * kFw + ff;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 x = H[q * kPmax + pp];
                        const cfptr wq = c_dw + (kt * 3 + kf) * 16 + 4 * q;
                        acc[4 * q] += wq[0] * x.x; acc[4 * q + 1] += wq[1] * x.y;
                        acc[4 * q + 2] += wq[2] * x.z; acc[4 * q + 3] += wq[3] * x.w;
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < 16; ++c) acc[c] = prelu_f(acc[c], w.dw_slope);
#pragma unroll
            for (int co = 0; co < 8; ++co) h1r[i][co] = c_pw2_b[co];
#pragma unroll
            for (int ci = 0; ci < 16; ++ci)
#pragma unroll
                for (int co = 0; co < 8; ++co) h1r[i][co] += c_pw2[ci * 8 + co] * acc[ci];
        }
    }
    __syncthreads();   // every tap read of H is done: planes may be reused
    ADE_CLK(2);
#pragma unroll
    for (int i = 0; i < kPosPerThread; ++i) {
        const int p = tid + i * kFusedThreads;
        if (p < P) {
            H[p] = make_float4(h1r[i][0], h1r[i][1], h1r[i][2], h1r[i][3]);
            H[kPmax + p] = make_float4(h1r[i][4], h1r[i][5], h1r[i][6], h1r[i][7]);
        }
    }
    __syncthreads();
    ADE_CLK(3);

    // ---- phase 3+4a: TRA energy zt[t][c] = mean_f h1^2 (:154) and the GRU input projections GI[t][g*16+j] = b_ih + W_ih zt[t],
    //      one 16-lane DPP row per frame: lane j sums f in {j, j+16, j+32}, a 4-step row rotation all-reduce (fixed order,
    //      deterministic) gives every lane the 8 channel energies, then lane j produces its three gate rows.
    {
        const int tr = tid >> 4, j = tid & 15;
        const int t = tr < T ? tr : T - 1;       // rows beyond T recompute the last frame (cross-lane ops need the whole wave)
        {
            float e[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int f = j + 16 * u;
                if (f < kFw) {
                    const float4 a0 = H[t * kFw + f], a1 = H[kPmax + t * kFw + f];
                    e[0] += a0.x * a0.x; e[1] += a0.y * a0.y; e[2] += a0.z * a0.z; e[3] += a0.w * a0.w;
                    e[4] += a1.x * a1.x; e[5] += a1.y * a1.y; e[6] += a1.z * a1.z; e[7] += a1.w * a1.w;
                }
            }
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                e[c] += row_ror<8>(e[c]);
                e[c] += row_ror<4>(e[c]);
                e[c] += row_ror<2>(e[c]);
                e[c] += row_ror<1>(e[c]);
                e[c] = e[c] / (float)kFw;
            }
            const float* pk = w.gru + j * 78;
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                float sacc = pk[72 + g];
#pragma unroll
                for (int k = 0; k < 8; ++k) sacc += pk[g * 8 + k] * e[k];
                if (tr < T) GI[t * 48 + g * 16 + j] = sacc;
            }
        }
    }
    __syncthreads();
    ADE_CLK(4);
    ADE_CLK(5);
    // ---- phase 4b: the serial part, GRU(8->16) over T on one 16-lane row (all four rows of wave 0 run it redundantly;
    //      h exchanged with row_newbcast DPP)                                                    (:149,155)
    if (tid >= 64) {
        // waves 1-15, while wave 0 is busy with the serial recurrence: bypass half (a + skip)[:, 8:] -> planes 0-1 (h1 is dead)
        for (int p = tid - 64; p < P; p += kFusedThreads - 64) {
            float by[8];
            pl_ld8(ac, P, p, 2, by);
            if (sc) {
                float y[8];
                pl_ld8(sc, P, p, 2, y);
#pragma unroll
                for (int k = 0; k < 8; ++k) by[k] += y[k];
            }
            H[p] = make_float4(by[0], by[1], by[2], by[3]);
            H[kPmax + p] = make_float4(by[4], by[5], by[6], by[7]);
        }
    } else {
        const int j = tid & 15;
        const float* pk = w.gru + j * 78;
        float wh[3][16], bh[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) {
#pragma unroll
            for (int k = 0; k < 16; ++k) wh[g][k] = pk[24 + g * 16 + k];
            bh[g] = pk[75 + g];
        }
        float h = 0.0f;
        // The step latency IS the kernel's critical path: the three 16-term dot products run as 4 independent
        // partial sums each (12 chains of 4 FMAs instead of 3 chains of 16) and the step's three input-projection
        // values are fetched from LDS one step ahead, so nothing but the h -> gates -> h chain is serial.
        float gi_r = GI[j], gi_z = GI[16 + j], gi_n = GI[32 + j];
        for (int t = 0; t < T; ++t) {
            const int tn = t + 1 < T ? t + 1 : t;
            const float nx_r = GI[tn * 48 + j], nx_z = GI[tn * 48 + 16 + j], nx_n = GI[tn * 48 + 32 + j];
            float ar[4] = {bh[0], 0.0f, 0.0f, 0.0f}, az[4] = {bh[1], 0.0f, 0.0f, 0.0f}, an[4] = {bh[2], 0.0f, 0.0f, 0.0f};
#define ADE_TRA_K(K) { const float hk = row_bcast<K>(h); ar[K & 3] += wh[0][K] * hk; az[K & 3] += wh[1][K] * hk; an[K & 3] += wh[2][K] * hk; }
            ADE_REP16(ADE_TRA_K)
#undef ADE_TRA_K
            const float gr = (ar[0] + ar[1]) + (ar[2] + ar[3]);
            const float gz = (az[0] + az[1]) + (az[2] + az[3]);
            const float gn = (an[0] + an[1]) + (an[2] + an[3]);
            const float r = sigmoid_f(gi_r + gr);
            const float z = sigmoid_f(gi_z + gz);
            const float n = tanh_f(gi_n + r * gn);
            h = (1.0f - z) * n + z * h;
            if (tid < 16) HS[t * 16 + j] = h;
            gi_r = nx_r; gi_z = nx_z; gi_n = nx_n;
        }
    }
    __syncthreads();
    ADE_CLK(6);
    // ---- phase 4c: at[t][c] = sigmoid(Linear(16->8)(h_t))                                      (:155)
    for (int idx = tid; idx < T * 8; idx += kFusedThreads) {
        const int t = idx >> 3, c = idx & 7;
        const float* fr = w.fc + c * 17;
        float s = fr[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) s += fr[k] * HS[t * 16 + k];
        at[idx] = sigmoid_f(s);
    }
    __syncthreads();
    ADE_CLK(7);
    // ---- phase 5: gate, interleave with the bypass half (LDS), store; optionally leave the next block's pointwise input
    //      (out[:, :8] + next_skip[:, :8]) in planes 2-3                                            (:156,324)
    const float* nsc = next_skip ? next_skip + (size_t)chunk * kCh * P : nullptr;
#pragma unroll
    for (int i = 0; i < kPosPerThread; ++i) {
        const int p = tid + i * kFusedThreads;
        if (p < P) {
            const int t = p / kFw;
            const float4 b0 = H[p], b1 = H[kPmax + p];
            const float by[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
            float o[16];
#pragma unroll
            for (int k = 0; k < 8; ++k) { o[2 * k] = h1r[i][k] * at[t * 8 + k]; o[2 * k + 1] = by[k]; }
            pl_st16(oc, P, p, o);
            if (next_x1) {
                float n8[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) n8[k] = o[k];
                if (nsc) {
                    float y[8];
                    pl_ld8(nsc, P, p, 0, y);
#pragma unroll
                    for (int k = 0; k < 8; ++k) n8[k] += y[k];
                }
                H[2 * kPmax + p] = make_float4(n8[0], n8[1], n8[2], n8[3]);
                H[3 * kPmax + p] = make_float4(n8[4], n8[5], n8[6], n8[7]);
            }
        }
    }
    ADE_CLK(8);
}

// ---------------------------------------------------------------------------------------------------------
// DPGRNN, whole block for one chunk (Export_GTCRN.py:466-481; GRNN :409-428).
//   x (B,T,33,16) -> intra biGRU along F -> Linear -> LayerNorm((33,16)) -> +x = mid
//                 -> inter GRU along T   -> Linear -> LayerNorm           -> +mid = out
// LDS: R[4][kPmax] float4 (rnn outputs, then mid, then rnn outputs again) | red[kPmax] | stat[2][64].
// `mid` is parked in the output buffer between the two halves (same thread writes and re-reads it).
// ---------------------------------------------------------------------------------------------------------
constexpr size_t kDpSmemBytes = (size_t)4 * kPmax * 16 + (size_t)kPmax * 4 + 2 * kTmaxFused * 4 + (size_t)4 * kFw * kCh * 4;   // + 4 LayerNorm tables

// Linear(16,16) on the rnn output of each of this thread's positions, two-pass LayerNorm statistics per frame
// through LDS, then  y = res + (v - mean) * rstd * gamma + beta.   v/res/y: [kPosPerThread][16] registers.
__device__ __forceinline__ void fc_ln_phase(const float4* R, float* red, float* stat, const float* __restrict__ fc,
                                            const float* __restrict__ fc_b, const float* __restrict__ ln_w,
                                            const float* __restrict__ ln_b, int T, int P, int tid, float (*v)[16]) {
    const cfptr c_fc_b = cptr(fc_b);
#pragma unroll
    for (int i = 0; i < kPosPerThread; ++i) {
        const int p = tid + i * kFusedThreads;
        if (p < P) {
            float r[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 x = R[q * kPmax + p];
                r[4 * q] = x.x; r[4 * q + 1] = x.y; r[4 * q + 2] = x.z; r[4 * q + 3] = x.w;
            }
            cfptr c_fc = cptr(fc);
            ADE_KEEP_IN_LOOP(c_fc);
#pragma unroll
            for (int co = 0; co < 16; ++co) v[i][co] = c_fc_b[co];
#pragma unroll
            for (int k = 0; k < 16; ++k)
#pragma unroll
                for (int co = 0; co < 16; ++co) v[i][co] += c_fc[k * 16 + co] * r[k];
            float s = 0.0f;
#pragma unroll
            for (int co = 0; co < 16; ++co) s += v[i][co];
            red[p] = s;
        }
    }
    __syncthreads();
    {   // per-frame total: one 16-lane row per frame, 3 partials per lane, row rotation all-reduce (fixed order)
        const int tr = tid >> 4, j = tid & 15, t = tr < T ? tr : T - 1;
        float s = red[t * kFw + j] + red[t * kFw + j + 16] + (j == 0 ? red[t * kFw + 32] : 0.0f);
        s += row_ror<8>(s); s += row_ror<4>(s); s += row_ror<2>(s); s += row_ror<1>(s);
        if (j == 0 && tr < T) stat[t] = s / (float)(kFw * kCh);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < kPosPerThread; ++i) {
        const int p = tid + i * kFusedThreads;
        if (p < P) {
            const float mean = stat[p / kFw];
            float s = 0.0f;
#pragma unroll
            for (int co = 0; co < 16; ++co) { const float d = v[i][co] - mean; s += d * d; }
            red[p] = s;
        }
    }
    __syncthreads();
    {
        const int tr = tid >> 4, j = tid & 15, t = tr < T ? tr : T - 1;
        float s = red[t * kFw + j] + red[t * kFw + j + 16] + (j == 0 ? red[t * kFw + 32] : 0.0f);
        s += row_ror<8>(s); s += row_ror<4>(s); s += row_ror<2>(s); s += row_ror<1>(s);
        if (j == 0 && tr < T) stat[kTmaxFused + t] = 1.0f / sqrtf(s / (float)(kFw * kCh) + 1e-8f);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < kPosPerThread; ++i) {
        const int p = tid + i * kFusedThreads;
        if (p < P) {
            const int t = p / kFw, f = p - t * kFw;
            const float mean = stat[t], rstd = stat[kTmaxFused + t];
            float gw[16], gb[16];
            ld16(ln_w + f * kCh, gw);
            ld16(ln_b + f * kCh, gb);
#pragma unroll
            for (int co = 0; co < 16; ++co) v[i][co] = (v[i][co] - mean) * rstd * gw[co] + gb[co];
        }
    }
}

__device__ __forceinline__ void dpgrnn_stage(float4* smem, int chunk, const float* __restrict__ x, const DpW& w, float* __restrict__ out,
                                             int T, long long* __restrict__ clk, bool next_x1 = false,
                                             const float* __restrict__ next_skip = nullptr) {
    float4* R = smem;
    float* Rf = reinterpret_cast<float*>(smem);
    float* red = reinterpret_cast<float*>(smem + 4 * kPmax);
    float* stat = red + kPmax;
    float* lnt = stat + 2 * kTmaxFused;   // LDS copies of the 4 LayerNorm tables [intra gamma | intra beta | inter gamma | inter beta]
    const int P = T * kFw;
    int tid_ = threadIdx.x;
    ADE_OPAQUE_V(tid_);
    const int tid = tid_;
    for (int i = tid; i < kFw * kCh; i += kFusedThreads) {   // (made visible by the barrier after phase A)
        lnt[i] = w.intra_ln_w[i];
        lnt[kFw * kCh + i] = w.intra_ln_b[i];
        lnt[2 * kFw * kCh + i] = w.inter_ln_w[i];
        lnt[3 * kFw * kCh + i] = w.inter_ln_b[i];
    }
    const float* xc = x + (size_t)chunk * kCh * P;
    float* oc = out + (size_t)chunk * kCh * P;

    ADE_CLK(16);
    // ---- phase A: intra GRNN.  16 lanes per frame: lane = group*8 + dir*4 + unit (== output channel); GRU(8->4)
    //      along F, hidden exchanged inside the quad with quad_perm DPP.                          (:441-446,472-473)
    {
        const int t = tid >> 4, q = tid & 15;
        const int grp = q >> 3, dir = (q >> 2) & 1;
        const bool live = t < T;
        const int tc = live ? t : T - 1;
        const float* pk = w.intra_gru + q * 42;
        float wi[3][8], wh[3][4], bi[3], bh[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) {
#pragma unroll
            for (int k = 0; k < 8; ++k) wi[g][k] = pk[g * 8 + k];
#pragma unroll
            for (int k = 0; k < 4; ++k) wh[g][k] = pk[24 + g * 4 + k];
            bi[g] = pk[36 + g];
            bh[g] = pk[39 + g];
        }
        const int prow = tc * kFw;
        float h = 0.0f;
        // inputs come from HBM/L2 (~1 us away): a 4-slot register ring keeps three steps of loads in flight, and the
        // fully unrolled loop lets the scheduler start the (h-independent) input projections early.
        float xq[4][8];
#pragma unroll
        for (int d = 0; d < 3; ++d) pl_ld8(xc, P, prow + (dir ? kFw - 1 - d : d), grp * 2, xq[d]);
#pragma unroll
        for (int s = 0; s < kFw; ++s) {
            const int f = dir ? kFw - 1 - s : s;
            if (s + 3 < kFw) pl_ld8(xc, P, prow + (dir ? f - 3 : f + 3), grp * 2, xq[(s + 3) & 3]);
            const float* xv = xq[s & 3];
            float gi[3], gh[3];
#pragma unroll
            for (int g = 0; g < 3; ++g) { gi[g] = bi[g]; gh[g] = bh[g]; }
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int g = 0; g < 3; ++g) gi[g] += wi[g][k] * xv[k];
            {
                const float h0 = quad_bcast<0>(h), h1 = quad_bcast<1>(h), h2 = quad_bcast<2>(h), h3 = quad_bcast<3>(h);
#pragma unroll
                for (int g = 0; g < 3; ++g) gh[g] += (wh[g][0] * h0 + wh[g][1] * h1) + (wh[g][2] * h2 + wh[g][3] * h3);
            }
            const float r = sigmoid_f(gi[0] + gh[0]);
            const float z = sigmoid_f(gi[1] + gh[1]);
            const float n = tanh_f(gi[2] + r * gh[2]);
            h = (1.0f - z) * n + z * h;
            if (live) Rf[((size_t)(q >> 2) * kPmax + tc * kFw + f) * 4 + (q & 3)] = h;
        }
    }
    __syncthreads();
    ADE_CLK(17);
    // ---- phase B: intra Linear + LayerNorm + residual -> mid (registers, and back into R for the inter GRU)
    //      (mid is parked in `out` -- L2-resident, re-read by the same thread in phase D -- instead of 48 live VGPRs)
    {
        float mid[kPosPerThread][16];
        fc_ln_phase(R, red, stat, w.intra_fc, w.intra_fc_b, lnt, lnt + kFw * kCh, T, P, tid, mid);
#pragma unroll
        for (int i = 0; i < kPosPerThread; ++i) {
            const int p = tid + i * kFusedThreads;
            if (p < P) {
                float xr[16];
                pl_ld16(xc, P, p, xr);
#pragma unroll
                for (int co = 0; co < 16; ++co) mid[i][co] += xr[co];
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    R[q * kPmax + p] = make_float4(mid[i][4 * q], mid[i][4 * q + 1], mid[i][4 * q + 2], mid[i][4 * q + 3]);
                pl_st16(oc, P, p, mid[i]);
            }
        }
    }
    __syncthreads();
    ADE_CLK(18);
    // ---- phase C: inter GRNN.  16 lanes per F column: lane = group*8 + unit; GRU(8->8) along T, in place in R
    //      (a column position is read, then overwritten, by its own 16 lanes only).              (:450-455,478-479)
    if (tid < ((kFw * 16 + 63) / 64) * 64) {
        const int f = tid >> 4, q = tid & 15;
        const int grp = q >> 3;
        const bool live = f < kFw;
        const int fc_ = live ? f : kFw - 1;
        const float* pk = w.inter_gru + q * 54;
        float wi[3][8], wh[3][8], bi[3], bh[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) {
#pragma unroll
            for (int k = 0; k < 8; ++k) { wi[g][k] = pk[g * 8 + k]; wh[g][k] = pk[24 + g * 8 + k]; }
            bi[g] = pk[48 + g];
            bh[g] = pk[51 + g];
        }
        float h = 0.0f;
        float4 xa = R[(grp * 2) * kPmax + fc_], xb = R[(grp * 2 + 1) * kPmax + fc_];
        for (int t = 0; t < T; ++t) {
            const int p = t * kFw + fc_;
            const int pn = (t + 1 < T ? t + 1 : t) * kFw + fc_;
            const float xv[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
            xa = R[(grp * 2) * kPmax + pn];          // next step's input: issued now, needed after this step's write
            xb = R[(grp * 2 + 1) * kPmax + pn];
            float gi[3], ga[3], gb[3];
#pragma unroll
            for (int g = 0; g < 3; ++g) { gi[g] = bi[g]; ga[g] = bh[g]; gb[g] = 0.0f; }
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int g = 0; g < 3; ++g) gi[g] += wi[g][k] * xv[k];
#define ADE_INTER_K(K, ACC) { const float lo = row_bcast<K>(h), hi = row_bcast<K + 8>(h); const float hk = grp ? hi : lo; \
                              ACC[0] += wh[0][K] * hk; ACC[1] += wh[1][K] * hk; ACC[2] += wh[2][K] * hk; }
            ADE_INTER_K(0, ga) ADE_INTER_K(1, gb) ADE_INTER_K(2, ga) ADE_INTER_K(3, gb)
            ADE_INTER_K(4, ga) ADE_INTER_K(5, gb) ADE_INTER_K(6, ga) ADE_INTER_K(7, gb)
#undef ADE_INTER_K
            const float r = sigmoid_f(gi[0] + (ga[0] + gb[0]));
            const float z = sigmoid_f(gi[1] + (ga[1] + gb[1]));
            const float n = tanh_f(gi[2] + r * (ga[2] + gb[2]));
            h = (1.0f - z) * n + z * h;
            if (live) Rf[((size_t)(q >> 2) * kPmax + p) * 4 + (q & 3)] = h;
        }
    }
    __syncthreads();
    ADE_CLK(19);
    // ---- phase D: inter Linear + LayerNorm + residual(mid) -> out
    float y[kPosPerThread][16];
    fc_ln_phase(R, red, stat, w.inter_fc, w.inter_fc_b, lnt + 2 * kFw * kCh, lnt + 3 * kFw * kCh, T, P, tid, y);
#pragma unroll
    for (int i = 0; i < kPosPerThread; ++i) {
        const int p = tid + i * kFusedThreads;
        if (p < P) {
            float m[16];
            pl_ld16(oc, P, p, m);     // mid, written by this same thread in phase B
#pragma unroll
            for (int co = 0; co < 16; ++co) y[i][co] += m[co];
            pl_st16(oc, P, p, y[i]);
            if (next_x1) {   // the following GTConvBlock's pointwise input (out + skip)[:, :8] -> LDS planes 2-3 (R is dead)
                float n8[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) n8[k] = y[i][k];
                if (next_skip) {
                    float sk[8];
                    pl_ld8(next_skip + (size_t)chunk * kCh * P, P, p, 0, sk);
#pragma unroll
                    for (int k = 0; k < 8; ++k) n8[k] += sk[k];
                }
                R[2 * kPmax + p] = make_float4(n8[0], n8[1], n8[2], n8[3]);
                R[3 * kPmax + p] = make_float4(n8[4], n8[5], n8[6], n8[7]);
            }
        }
    }
    ADE_CLK(20);
}


}  // namespace stage
}  // namespace ade
