// ade_stage_frontback.h — per-chunk fused FRONT (PCM -> spectrum, e0, e1) and BACK (d2 -> PCM) stage kernels.
//
// Same design as ade_fused.hip: one 1024-thread workgroup owns one audio chunk; everything that is per-frame local
// (STFT, feature build, ERB merge; mask, irFFT, overlap-add, PCM tail) stays in LDS, the inter-stage tensors go to
// HBM in the channel-quad planar layout (coalesced 16 B per lane).
//   FRONT  F1-F7: int16 -> *2^-15 - mean -> reflect pad -> window -> rFFT-512 -> [mag,re,im] -> ERB merge (LDS)
//                 -> SFE + Conv(9->16,1x5,s2)+BN+PReLU -> e0 -> Conv(16->16,g2,1x5,s2)+BN+PReLU -> e1
//   BACK   F11(tail)-F14: (d2+e1) -> ConvT(16->16,g2)+PReLU -> d3 ; (d3+e0) -> ConvT(16->2)+Tanh -> mask -> ERB split
//                 -> complex ratio mask -> irFFT-512 -> window -> overlap-add (LDS) -> /sum(w^2) -> *32767, clamp, trunc
// Reference lines: Export_GTCRN.py:637-647, 594-595, 99-102, 117-141, 159-197, 488-489, 515-516, 104-107, 583-590, 681-690 ;
// STFT_Process.py:303-316, 239-251, 326-336.
#pragma once
#include "ade_stage_net.h"

namespace ade {
namespace stage {


constexpr int kWbuf = 264;   // float2 slots of one wave's FFT / spectrum buffer (257 used)



// X[k] and X[256-k] of the 512-point real FFT from the packed 256-point FFT Z, in place in buf[0..256]
// (Z[256] == Z[0]).  E = (Z[k] + conj Z[256-k])/2, O = -i (Z[k] - conj Z[256-k])/2, X[k] = E + e^{-2 pi i k/512} O.
__device__ __forceinline__ float2 rfft_bin(float2 zk, float2 zp, float2 w) {
    const float2 e = make_float2(0.5f * (zk.x + zp.x), 0.5f * (zk.y - zp.y));
    const float2 d = make_float2(0.5f * (zk.x - zp.x), 0.5f * (zk.y + zp.y));
    const float2 o = make_float2(d.y, -d.x);
    return make_float2(e.x + (w.x * o.x - w.y * o.y), e.y + (w.x * o.y + w.y * o.x));
}

constexpr size_t kFrontFeatFloats = (size_t)kTmaxFused * 3 * kErb;                      // feat[T][3][129]
constexpr size_t kTabFloats = 512 + 2 * 256 + 2 * 264;   // window | tw256 | tw512 staged in LDS (L2 is ~1 us away per dependent load)
constexpr size_t kFrontSmemBytes = kFrontFeatFloats * 4 + (size_t)16 * kWbuf * 8 + 64 + kTabFloats * 4;

// copy the FFT tables into LDS; returns LDS-resident views
struct LdsTabs { const float* win; const float2* tw256; const float2* tw512; };
__device__ __forceinline__ LdsTabs stage_tables(float* dst, const FftTabs& t, int tid) {
    float* win = dst;
    float* tw256 = dst + 512;
    float* tw512 = tw256 + 512;
    for (int i = tid; i < 512; i += kFusedThreads) { win[i] = t.win[i]; tw256[i] = reinterpret_cast<const float*>(t.tw256)[i]; }
    for (int i = tid; i < 514; i += kFusedThreads) tw512[i] = reinterpret_cast<const float*>(t.tw512)[i];
    return LdsTabs{win, reinterpret_cast<const float2*>(tw256), reinterpret_cast<const float2*>(tw512)};
}

__device__ __forceinline__ void front_stage(float* smem, int chunk, const int16_t* __restrict__ pcm, int L, int T, const FftTabs& tabs,
                                            const BandTab& erb, const ConvW& c0, const ConvW& c1, float* __restrict__ spec,
                                            float* __restrict__ e0, float* __restrict__ e1, long long* __restrict__ clk) {
    float* feat = smem;
    float2* wbuf_all = reinterpret_cast<float2*>(smem + kFrontFeatFloats);
    int* red = reinterpret_cast<int*>(wbuf_all + 16 * kWbuf);
    int tid_ = threadIdx.x;
    ADE_OPAQUE_V(tid_);
    const int tid = tid_, wave = tid >> 6, lane = tid & 63;
    float2* buf = wbuf_all + wave * kWbuf;
    const int16_t* row = pcm + (size_t)chunk * L;
    const int P0 = T * kF1, P = T * kFw;
    float* e0c = e0 + (size_t)chunk * kCh * P0;
    float* e1c = e1 + (size_t)chunk * kCh * P;
    float* specc = spec + (size_t)chunk * T * 2 * kBinsPad;
    ADE_CLK(32);
    const LdsTabs lt = stage_tables(reinterpret_cast<float*>(red + 16), tabs, tid);

    // ---- F1: DC mean of THIS chunk (exact integer sum, one rounding)                     (Export_GTCRN.py:645-647)
    {
        int s = 0;
        for (int i = tid; i < L; i += kFusedThreads) s += (int)row[i];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
        if (lane == 0) red[wave] = s;
    }
    __syncthreads();
    float dc;
    {
        long long tot = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) tot += red[i];
        dc = (float)((double)tot / ((double)L * 32768.0));
    }
    ADE_CLK(33);

    // ---- F2-F5: 16 frames per round, one wavefront per frame
    const bool pair_ok = ((L & 1) == 0) && ((reinterpret_cast<size_t>(row) & 3) == 0);
    for (int round = 0; round * 16 < T; ++round) {
        const int t = round * 16 + wave;
        const bool live = t < T;
        float2 v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = lane + 64 * r;
            float s[2];
            const int j0 = kHop * t + 2 * n - kNfft / 2;             // even index of the sample pair (2n, 2n+1) of this frame
            if (live && pair_ok && j0 >= 0 && j0 + 1 < L) {          // interior: one aligned 32-bit load for the pair
                const int w2 = *reinterpret_cast<const int*>(row + j0);
                s[0] = (float)(short)(w2 & 0xffff) * (1.0f / 32768.0f) - dc;
                s[1] = (float)(short)(w2 >> 16) * (1.0f / 32768.0f) - dc;
            } else {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    int j = j0 + q;
                    j = j < 0 ? -j : (j >= L ? 2 * (L - 1) - j : j);     // reflect (STFT_Process.py:306-309)
                    s[q] = live ? (float)row[j] * (1.0f / 32768.0f) - dc : 0.0f;
                }
            }
            s[0] *= lt.win[2 * n];
            s[1] *= lt.win[2 * n + 1];
            v[r] = make_float2(s[0], s[1]);
        }
        fft256_inplace(v, buf, lane, lt.tw256);
        wave_sync();                                       // last pass's reads are done (buffer is wave-private)
#pragma unroll
        for (int r = 0; r < 4; ++r) buf[lane + 64 * r] = v[r];
        wave_sync();
        {   // pairs (k, 256-k), k = lane and lane+64 ; lane 0 also does the self-paired k = 128
            float2 xa[2], xb[2];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int k = lane + 64 * r;
                const float2 zk = buf[k], zp = buf[(256 - k) & 255];
                xa[r] = rfft_bin(zk, zp, lt.tw512[k]);
                xb[r] = rfft_bin(zp, zk, lt.tw512[256 - k]);
            }
            float2 xm = make_float2(0.0f, 0.0f);
            if (lane == 0) { const float2 z = buf[128]; xm = rfft_bin(z, z, lt.tw512[128]); }
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int k = lane + 64 * r;
                buf[k] = xa[r];
                buf[256 - k] = xb[r];          // k = 0 -> slot 256 (Nyquist)
            }
            if (lane == 0) buf[128] = xm;
        }
        wave_sync();
        if (live) {
            float* fr = feat + (size_t)t * 3 * kErb;
#pragma unroll
            for (int r = 0; r < 5; ++r) {
                if (r == 4 && lane != 0) break;
                const int k = r < 4 ? lane + 64 * r : 256;
                const float2 x = buf[k];
                specc[((size_t)t * 2 + 0) * kBinsPad + k] = x.x;
                specc[((size_t)t * 2 + 1) * kBinsPad + k] = x.y;
                if (k < kErbLow) {
                    fr[k] = sqrtf((x.x * x.x + x.y * x.y) + 1e-12f);          // Export_GTCRN.py:594-595
                    fr[kErb + k] = x.x;
                    fr[2 * kErb + k] = x.y;
                }
            }
            // ERB merge, one band per lane: banded sum == the dense 192x64 matmul term for term    (:99-102)
            const int s0 = erb.start[lane];
            float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f;
            for (int n = 0; n < erb.count; ++n) {
                const float wv = erb.w[n * kErbBands + lane];
                const float2 x = buf[kErbLow + min(s0 + n, kErbHigh - 1)];
                a0 += sqrtf((x.x * x.x + x.y * x.y) + 1e-12f) * wv;
                a1 += x.x * wv;
                a2 += x.y * wv;
            }
            fr[kErbLow + lane] = a0;
            fr[kErb + kErbLow + lane] = a1;
            fr[2 * kErb + kErbLow + lane] = a2;
        }
        wave_sync();
    }
    __syncthreads();    // feat (all frames) complete
    ADE_CLK(34);

    // ---- F6-F7a: SFE(3) + Conv2d(9->16,(1,5),s(1,2),p(0,2)) + BN + PReLU, one lane per (t,fo)   (:117-141,159-197,488)
    {
        const cfptr cb = cptr(c0.b);
        for (int idx = tid; idx < P0; idx += kFusedThreads) {
            const int t = idx / kF1, fo = idx - t * kF1;
            const float* fr = feat + (size_t)t * 3 * kErb;
            cfptr cw = cptr(c0.w);
            ADE_KEEP_IN_LOOP(cw);
            float v[3][7];
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int j = 0; j < 7; ++j) {
                    const int q = 2 * fo - 3 + j;
                    v[c][j] = (q >= 0 && q < kErb) ? fr[c * kErb + q] : 0.0f;
                }
            float acc[16];
#pragma unroll
            for (int co = 0; co < 16; ++co) acc[co] = cb[co];
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const int p = 2 * fo - 2 + k;
                const bool pv = p >= 0 && p < kErb;
#pragma unroll
                for (int c = 0; c < 3; ++c)
#pragma unroll
                    for (int o = 0; o < 3; ++o) {
                        const float x = pv ? v[c][k + o] : 0.0f;
#pragma unroll
                        for (int co = 0; co < 16; ++co) acc[co] += cw[(k * 9 + c * 3 + o) * 16 + co] * x;
                    }
            }
#pragma unroll
            for (int co = 0; co < 16; ++co) acc[co] = prelu_f(acc[co], c0.slope);
            pl_st16(e0c, P0, idx, acc);
        }
    }
    __syncthreads();    // e0 (global, written by this workgroup) is visible to the whole workgroup
    ADE_CLK(35);

    // ---- F7b: Conv2d(16->16,(1,5),s2,groups 2) + BN + PReLU                                     (:489)
    {
        const cfptr cb = cptr(c1.b);
        for (int idx = tid; idx < P; idx += kFusedThreads) {
            const int t = idx / kFw, fo = idx - t * kFw;
            cfptr cw = cptr(c1.w);
            ADE_KEEP_IN_LOOP(cw);
            float acc[16];
#pragma unroll
            for (int co = 0; co < 16; ++co) acc[co] = cb[co];
#pragma unroll 1   // one tap live at a time: 5 taps x 16 floats hoisted together would spill
            for (int k = 0; k < 5; ++k) {
                const int fi = 2 * fo - 2 + k;
                float x[16];
                if (fi >= 0 && fi < kF1) pl_ld16(e0c, P0, t * kF1 + fi, x);
                else {
#pragma unroll
                    for (int i = 0; i < 16; ++i) x[i] = 0.0f;
                }
#pragma unroll
                for (int g = 0; g < 2; ++g)
#pragma unroll
                    for (int ci = 0; ci < 8; ++ci)
#pragma unroll
                        for (int co = 0; co < 8; ++co) acc[g * 8 + co] += cw[((k * 2 + g) * 8 + ci) * 8 + co] * x[g * 8 + ci];
            }
#pragma unroll
            for (int co = 0; co < 16; ++co) acc[co] = prelu_f(acc[co], c1.slope);
            pl_st16(e1c, P, idx, acc);
        }
    }
    ADE_CLK(36);
}

// ---------------------------------------------------------------------------------------------------------------
constexpr size_t kBackAccFloats = (size_t)kNfft + (size_t)kHop * (kTmaxFused - 1);
constexpr size_t kBackSmemBytes = (size_t)16 * kWbuf * 8 + kBackAccFloats * 4 + kTabFloats * 4;

__device__ __forceinline__ void back_stage(float* smem, int chunk, const float* __restrict__ x, const float* __restrict__ e1,
                                           const float* __restrict__ e0, const float* __restrict__ spec, const ConvW& c3, const ConvW& c4,
                                           const BandTab& bs, const FftTabs& tabs, float* __restrict__ d3, float* __restrict__ mask,
                                           int16_t* __restrict__ pcm, float* __restrict__ f32, int T, long long* __restrict__ clk) {
    float2* wbuf_all = reinterpret_cast<float2*>(smem);
    float* acc = smem + 16 * kWbuf * 2;
    int tid_ = threadIdx.x;
    ADE_OPAQUE_V(tid_);
    const int tid = tid_, wave = tid >> 6, lane = tid & 63;
    float2* buf = wbuf_all + wave * kWbuf;
    const int P0 = T * kF1, P = T * kFw;
    const float* xc = x + (size_t)chunk * kCh * P;
    const float* e1c = e1 + (size_t)chunk * kCh * P;
    const float* e0c = e0 + (size_t)chunk * kCh * P0;
    float* d3c = d3 + (size_t)chunk * kCh * P0;
    float* maskc = mask + (size_t)chunk * T * 2 * kErbPad;
    const float* specc = spec + (size_t)chunk * T * 2 * kBinsPad;
    ADE_CLK(48);
    const LdsTabs lt = stage_tables(acc + kBackAccFloats, tabs, tid);   // made visible by the barriers below

    // ---- ConvTranspose2d(16->16,(1,5),s(1,2),p(0,2),groups 2) + BN + PReLU on (x + e1): one lane per input column m ->
    //      outputs fo = 2m (taps 0,2,4 <- m+1,m,m-1) and 2m+1 (taps 1,3 <- m+1,m)                  (:515,527)
    {
        const cfptr cb = cptr(c3.b);
        for (int idx = tid; idx < P; idx += kFusedThreads) {
            const int t = idx / kFw, m = idx - t * kFw;
            cfptr cw = cptr(c3.w);
            ADE_KEEP_IN_LOOP(cw);
            float ev[16], od[16];
#pragma unroll
            for (int co = 0; co < 16; ++co) { ev[co] = cb[co]; od[co] = cb[co]; }
#pragma unroll
            for (int dlt = -1; dlt <= 1; ++dlt) {
                const int fi = m + dlt;
                if (fi < 0 || fi >= kFw) continue;
                float xv[16], yv[16];
                pl_ld16(xc, P, idx + dlt, xv);
                pl_ld16(e1c, P, idx + dlt, yv);
#pragma unroll
                for (int i = 0; i < 16; ++i) xv[i] += yv[i];
                const int ke = 2 - 2 * dlt, ko = 3 - 2 * dlt;
#pragma unroll
                for (int g = 0; g < 2; ++g)
#pragma unroll
                    for (int ci = 0; ci < 8; ++ci)
#pragma unroll
                        for (int co = 0; co < 8; ++co) {
                            ev[g * 8 + co] += cw[((ke * 2 + g) * 8 + ci) * 8 + co] * xv[g * 8 + ci];
                            if (dlt >= 0) od[g * 8 + co] += cw[((ko * 2 + g) * 8 + ci) * 8 + co] * xv[g * 8 + ci];
                        }
            }
#pragma unroll
            for (int co = 0; co < 16; ++co) { ev[co] = prelu_f(ev[co], c3.slope); od[co] = prelu_f(od[co], c3.slope); }
            pl_st16(d3c, P0, t * kF1 + 2 * m, ev);
            if (2 * m + 1 < kF1) pl_st16(d3c, P0, t * kF1 + 2 * m + 1, od);
        }
    }
    __syncthreads();
    ADE_CLK(49);
    // ---- ConvTranspose2d(16->2) + BN + Tanh on (d3 + e0) -> mask (T,2,132)                          (:516,528)
    {
        const cfptr cb = cptr(c4.b), cw = cptr(c4.w);
        for (int idx = tid; idx < P0; idx += kFusedThreads) {
            const int t = idx / kF1, m = idx - t * kF1;
            float ev[2] = {cb[0], cb[1]}, od[2] = {cb[0], cb[1]};
#pragma unroll
            for (int dlt = -1; dlt <= 1; ++dlt) {
                const int fi = m + dlt;
                if (fi < 0 || fi >= kF1) continue;
                float xv[16], yv[16];
                pl_ld16(d3c, P0, idx + dlt, xv);
                pl_ld16(e0c, P0, idx + dlt, yv);
#pragma unroll
                for (int i = 0; i < 16; ++i) xv[i] += yv[i];
                const int ke = 2 - 2 * dlt, ko = 3 - 2 * dlt;
#pragma unroll
                for (int ci = 0; ci < 16; ++ci)
#pragma unroll
                    for (int co = 0; co < 2; ++co) {
                        ev[co] += cw[(ke * 16 + ci) * 2 + co] * xv[ci];
                        if (dlt >= 0) od[co] += cw[(ko * 16 + ci) * 2 + co] * xv[ci];
                    }
            }
            float* mr = maskc + (size_t)t * 2 * kErbPad;
#pragma unroll
            for (int co = 0; co < 2; ++co) {
                mr[co * kErbPad + 2 * m] = tanhf(ev[co]);
                if (2 * m + 1 < kErb) mr[co * kErbPad + 2 * m + 1] = tanhf(od[co]);
            }
        }
    }
    const int acc_len = kNfft + kHop * (T - 1);
    for (int i = tid; i < acc_len; i += kFusedThreads) acc[i] = 0.0f;
    __syncthreads();
    ADE_CLK(50);

    // ---- ERB split + complex ratio mask + irFFT-512 + synthesis window + overlap-add in LDS.  Even frames, then odd
    //      frames: two frames of the same parity never overlap, so plain (deterministic) adds suffice.
    const int rounds = (T + 31) / 32;
    for (int par = 0; par < 2; ++par)
        for (int round = 0; round < rounds; ++round) {
            const int t = 2 * (round * 16 + wave) + par;
            const bool live = t < T;
            const int tc = live ? t : T - 1;
            const float* mr = maskc + (size_t)tc * 2 * kErbPad;
#pragma unroll
            for (int r = 0; r < 5; ++r) {
                if (r == 4 && lane != 0) break;
                const int k = r < 4 ? lane + 64 * r : 256;
                const float xr = specc[((size_t)tc * 2 + 0) * kBinsPad + k], xi = specc[((size_t)tc * 2 + 1) * kBinsPad + k];
                float m0, m1;
                if (k < kErbLow) {
                    m0 = mr[k];
                    m1 = mr[kErbPad + k];
                } else {   // ERB.bs: banded == dense 64x192 matmul                                 (:104-107)
                    const int o = k - kErbLow, s0 = bs.start[o];
                    m0 = 0.0f; m1 = 0.0f;
                    for (int n = 0; n < bs.count; ++n) {
                        const float wv = bs.w[n * kErbHigh + o];
                        const int jj = min(s0 + n, kErbBands - 1);
                        m0 += mr[kErbLow + jj] * wv;
                        m1 += mr[kErbPad + kErbLow + jj] * wv;
                    }
                }
                float2 y = make_float2(xr * m0 - xi * m1, xi * m0 + xr * m1);                        // (:585-590)
                if (k == 0 || k == 256) y.y = 0.0f;     // the reference's inverse kernel has sin(0) = sin(pi n) = 0 rows
                buf[k] = y;
            }
            wave_sync();
            float2 v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = lane + 64 * r;
                const float2 yk = buf[k], yp = buf[256 - k];
                const float2 e = make_float2(0.5f * (yk.x + yp.x), 0.5f * (yk.y - yp.y));
                const float2 d = make_float2(0.5f * (yk.x - yp.x), 0.5f * (yk.y + yp.y));
                const float2 o = fcmul(d, make_float2(lt.tw512[k].x, -lt.tw512[k].y));
                v[r] = make_float2(e.x - o.y, -(e.y + o.x));     // conj(E + i O): inverse FFT = conj(FFT(conj Z)) / 256
            }
            wave_sync();
            fft256_inplace(v, buf, lane, lt.tw256);
            if (live) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = lane + 64 * r;
                    float* a = acc + kHop * t + 2 * n;
                    a[0] += v[r].x * (1.0f / 256.0f) * lt.win[2 * n];
                    a[1] += -v[r].y * (1.0f / 256.0f) * lt.win[2 * n + 1];
                }
            }
            wave_sync();
            if (round == rounds - 1) __syncthreads();   // parity switch / end: overlapping frames meet only across this barrier
        }
    ADE_CLK(51);
    // ---- trim N/2, / sum(w^2), * 32767, clamp, truncating cast                          (STFT_Process.py:330-333, Export:681,690)
    {
        const int out_len = kHop * (T - 1);
        int16_t* po = pcm ? pcm + (size_t)chunk * out_len : nullptr;
        float* fo = f32 ? f32 + (size_t)chunk * out_len : nullptr;
        for (int i4 = tid; i4 < out_len / 4; i4 += kFusedThreads) {
            const int n = i4 * 4;
            float v[4], ws[4];
            ld4(acc + kHop + n, v);
            ld4(tabs.win_sum + (n & (kHop - 1)), ws);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = v[i] / ws[i];
            if (fo) st4(fo + n, v);
            if (po) {
                short q[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) q[i] = (short)(int)fminf(fmaxf(v[i] * 32767.0f, -32768.0f), 32767.0f);
                *reinterpret_cast<short4*>(po + n) = make_short4(q[0], q[1], q[2], q[3]);
            }
        }
    }
    ADE_CLK(52);
}


}  // namespace stage
}  // namespace ade
