// ade_stage_frontback.h — per-chunk FRONT (PCM -> spectrum, e0, e1) and BACK (d2 -> PCM) stage bodies.
//
// Same design as ade_stage_net.h: one 1024-thread workgroup owns one audio chunk.  Everything in these two stages is
// local to a frame except the overlap-add, so both walk the chunk in TILES OF 16 FRAMES (one frame per wavefront for the
// FFTs) and keep the whole per-tile pipeline in LDS:
//   FRONT  F1-F7  per tile: int16 -> *2^-15 - mean -> reflect pad -> window -> rFFT-512 -> [mag,re,im] -> ERB merge
//                 -> SFE + Conv(9->16,1x5,s2)+BN+PReLU = e0 (LDS + HBM) -> Conv(16->16,g2,1x5,s2)+BN+PReLU = e1 (HBM)
//   BACK   F11-F14 per tile: (d2+e1) -> ConvT(16->16,g2)+PReLU = d3 (LDS only) ; (d3+e0) -> ConvT(16->2)+Tanh = mask (LDS
//                 only) -> ERB split -> complex ratio mask -> irFFT-512 -> window -> overlap-add (LDS, 256-sample carry
//                 between tiles) -> /sum(w^2) -> *32767, clamp, truncate -> int16
// so a neighbour tap never costs a second trip to L2/HBM and d3 / mask / windowed frames never leave the CU.
// Inter-stage tensors in HBM are channel-quad planar (ade_stage_net.h).
// Reference lines: Export_GTCRN.py:637-647, 594-595, 99-102, 117-141, 159-197, 488-489, 515-516, 104-107, 583-590, 681-690 ;
// STFT_Process.py:303-316, 239-251, 326-336.
#pragma once
#include "ade_stage_net.h"

namespace ade {
namespace stage {

constexpr int kWbuf = 264;            // float2 slots of one wave's FFT / spectrum buffer (257 used)
constexpr int kTileF = 16;            // frames per tile = wavefronts per workgroup
constexpr int kTileP1 = kTileF * kF1; // 1040 positions of width 65
constexpr int kTileP = kTileF * kFw;  // 528 positions of width 33
constexpr size_t kTabFloats = 512 + 2 * 256 + 2 * 264;   // window | tw256 | tw512 staged in LDS

// X[k] of the 512-point real FFT from the packed 256-point FFT Z:  E = (Z[k] + conj Z[256-k])/2,
// O = -i (Z[k] - conj Z[256-k])/2,  X[k] = E + e^{-2 pi i k/512} O.   zp = Z[256-k] (un-conjugated).
__device__ __forceinline__ float2 rfft_bin(float2 zk, float2 zp, float2 w) {
    const float2 e = make_float2(0.5f * (zk.x + zp.x), 0.5f * (zk.y - zp.y));
    const float2 d = make_float2(0.5f * (zk.x - zp.x), 0.5f * (zk.y + zp.y));
    const float2 o = make_float2(d.y, -d.x);
    return make_float2(e.x + (w.x * o.x - w.y * o.y), e.y + (w.x * o.y + w.y * o.x));
}

// copy the FFT tables into LDS (L2 is ~1 us away per dependent load); returns LDS-resident views
struct LdsTabs { const float* win; const float2* tw256; const float2* tw512; };
__device__ __forceinline__ LdsTabs stage_tables(float* dst, const FftTabs& t, int tid) {
    float* win = dst;
    float* tw256 = dst + 512;
    float* tw512 = tw256 + 512;
    for (int i = tid; i < 512; i += kFusedThreads) { win[i] = t.win[i]; tw256[i] = reinterpret_cast<const float*>(t.tw256)[i]; }
    for (int i = tid; i < 514; i += kFusedThreads) tw512[i] = reinterpret_cast<const float*>(t.tw512)[i];
    return LdsTabs{win, reinterpret_cast<const float2*>(tw256), reinterpret_cast<const float2*>(tw512)};
}

// ---------------------------------------------------------------------------------------------------------------
// FRONT.  LDS (floats): wbuf[16][264]x2 | feat[16][3][129] | E0[4][1040]x4 | tabs | red[16]
// ---------------------------------------------------------------------------------------------------------------
constexpr size_t kFrontFeatFloats = (size_t)kTileF * 3 * kErb;
constexpr size_t kFrontSmemBytes = ((size_t)16 * kWbuf * 2 + kFrontFeatFloats + (size_t)4 * kTileP1 * 4 + kTabFloats + 16) * 4;

__device__ __forceinline__ void front_stage(float* smem, int chunk, const int16_t* __restrict__ pcm, int L, int T, const FftTabs& tabs,
                                            const BandTab& erb, const ConvW& c0, const ConvW& c1, float* __restrict__ spec,
                                            float* __restrict__ e0, float* __restrict__ e1, long long* __restrict__ clk) {
    float2* wbuf_all = reinterpret_cast<float2*>(smem);
    float* feat = smem + 16 * kWbuf * 2;
    float4* E0 = reinterpret_cast<float4*>(feat + kFrontFeatFloats);
    float* tabmem = reinterpret_cast<float*>(E0 + 4 * kTileP1);
    int* red = reinterpret_cast<int*>(tabmem + kTabFloats);
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
    const LdsTabs lt = stage_tables(tabmem, tabs, tid);

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
    const bool pair_ok = ((L & 1) == 0) && ((reinterpret_cast<size_t>(row) & 3) == 0);
    const cfptr c0b = cptr(c0.b), c1b = cptr(c1.b);

    for (int t0 = 0; t0 < T; t0 += kTileF) {
        const int nf = T - t0 < kTileF ? T - t0 : kTileF;
        // ---- F2-F5: one wavefront per frame of the tile
        {
            const int t = t0 + wave;
            const bool live = wave < nf;
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
                v[r] = make_float2(s[0] * lt.win[2 * n], s[1] * lt.win[2 * n + 1]);
            }
            fft256_inplace(v, buf, lane, lt.tw256);
            wave_sync();                                       // last pass's reads are done (buffer is wave-private)
#pragma unroll
            for (int r = 0; r < 4; ++r) buf[lane + 64 * r] = v[r];
            wave_sync();
            {   // X[k], X[256-k] in place: pairs k = lane, lane+64 ; lane 0 also does the self-paired k = 128
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
                float* fr = feat + (size_t)wave * 3 * kErb;
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
        }
        __syncthreads();    // feat of the tile complete (and the previous tile's conv1 is done with E0)
        ADE_CLK(34);
        // ---- F6-F7a: SFE(3) + Conv2d(9->16,(1,5),s(1,2),p(0,2)) + BN + PReLU, one lane per (t,fo) -> E0 (LDS) + e0 (HBM)
        for (int idx = tid; idx < nf * kF1; idx += kFusedThreads) {
            const int tl = idx / kF1, fo = idx - tl * kF1;
            const float* fr = feat + (size_t)tl * 3 * kErb;
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
            for (int co = 0; co < 16; ++co) acc[co] = c0b[co];
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const int p = 2 * fo - 2 + k;           // position in the SFE output; the conv zero-pads outside [0,129)
                const bool pv = p >= 0 && p < kErb;
#pragma unroll
                for (int c = 0; c < 3; ++c)
#pragma unroll
                    for (int o = 0; o < 3; ++o) {
                        const float x = pv ? v[c][k + o] : 0.0f;          // SFE channel c*3+o at p = feat[c][p-1+o]
#pragma unroll
                        for (int co = 0; co < 16; ++co) acc[co] += cw[(k * 9 + c * 3 + o) * 16 + co] * x;
                    }
            }
#pragma unroll
            for (int co = 0; co < 16; ++co) acc[co] = prelu_f(acc[co], c0.slope);
#pragma unroll
            for (int q = 0; q < 4; ++q) E0[q * kTileP1 + idx] = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
            pl_st16(e0c, P0, t0 * kF1 + idx, acc);
        }
        __syncthreads();
        ADE_CLK(35);
        // ---- F7b: Conv2d(16->16,(1,5),s2,groups 2) + BN + PReLU from E0 (LDS) -> e1 (HBM)                 (:489)
        for (int idx = tid; idx < nf * kFw; idx += kFusedThreads) {
            const int tl = idx / kFw, fo = idx - tl * kFw;
            cfptr cw = cptr(c1.w);
            ADE_KEEP_IN_LOOP(cw);
            float acc[16];
#pragma unroll
            for (int co = 0; co < 16; ++co) acc[co] = c1b[co];
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const int fi = 2 * fo - 2 + k;
                if (fi < 0 || fi >= kF1) continue;
                const int pp = tl * kF1 + fi;
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const float4 xa = E0[(2 * g) * kTileP1 + pp], xb = E0[(2 * g + 1) * kTileP1 + pp];
                    const float x[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
#pragma unroll
                    for (int ci = 0; ci < 8; ++ci)
#pragma unroll
                        for (int co = 0; co < 8; ++co) acc[g * 8 + co] += cw[((k * 2 + g) * 8 + ci) * 8 + co] * x[ci];
                }
            }
#pragma unroll
            for (int co = 0; co < 16; ++co) acc[co] = prelu_f(acc[co], c1.slope);
            pl_st16(e1c, P, t0 * kFw + idx, acc);
        }
        ADE_CLK(36);
        // no barrier here: the next tile's FFT phase touches neither E0 nor (before its own barrier) anything conv1 reads
    }
}

// ---------------------------------------------------------------------------------------------------------------
// BACK.  LDS (floats): S[4][528]x4 (reused as the FFT buffers wbuf[16][264]x2) | D[4][1040]x4 | M[16][2][132] |
//                      acc[512 + 256*15] | tabs
// ---------------------------------------------------------------------------------------------------------------
constexpr size_t kBackAccFloats = (size_t)kNfft + (size_t)kHop * (kTileF - 1);
constexpr size_t kBackSmemBytes =
    ((size_t)4 * kTileP * 4 + (size_t)4 * kTileP1 * 4 + (size_t)kTileF * 2 * kErbPad + kBackAccFloats + kTabFloats) * 4;
static_assert((size_t)4 * kTileP * 4 >= (size_t)16 * kWbuf * 2, "the S tile must be able to hold the 16 FFT buffers");

__device__ __forceinline__ void back_stage(float* smem, int chunk, const float* __restrict__ x, const float* __restrict__ e1,
                                           const float* __restrict__ e0, const float* __restrict__ spec, const ConvW& c3, const ConvW& c4,
                                           const BandTab& bs, const FftTabs& tabs, float* __restrict__ d3 /*unused: d3 stays in LDS*/,
                                           float* __restrict__ mask /*unused: the mask stays in LDS*/, int16_t* __restrict__ pcm,
                                           float* __restrict__ f32, int T, long long* __restrict__ clk) {
    (void)d3;
    (void)mask;
    float4* S = reinterpret_cast<float4*>(smem);
    float2* wbuf_all = reinterpret_cast<float2*>(smem);
    float4* D = S + 4 * kTileP;
    float* M = reinterpret_cast<float*>(D + 4 * kTileP1);
    float* acc = M + kTileF * 2 * kErbPad;
    float* tabmem = acc + kBackAccFloats;
    int tid_ = threadIdx.x;
    ADE_OPAQUE_V(tid_);
    const int tid = tid_, wave = tid >> 6, lane = tid & 63;
    float2* buf = wbuf_all + wave * kWbuf;
    const int P0 = T * kF1, P = T * kFw;
    const float* xc = x + (size_t)chunk * kCh * P;
    const float* e1c = e1 + (size_t)chunk * kCh * P;
    const float* e0c = e0 + (size_t)chunk * kCh * P0;
    const float* specc = spec + (size_t)chunk * T * 2 * kBinsPad;
    const int out_len = kHop * (T - 1);
    int16_t* po = pcm ? pcm + (size_t)chunk * out_len : nullptr;
    float* fo32 = f32 ? f32 + (size_t)chunk * out_len : nullptr;
    ADE_CLK(48);
    const LdsTabs lt = stage_tables(tabmem, tabs, tid);
    for (int i = tid; i < (int)kBackAccFloats; i += kFusedThreads) acc[i] = 0.0f;
    const cfptr c3b = cptr(c3.b), c4b = cptr(c4.b), c4w = cptr(c4.w);

    for (int t0 = 0; t0 < T; t0 += kTileF) {
        const int nf = T - t0 < kTileF ? T - t0 : kTileF;
        // spectrum of this wavefront's frame: issued now, consumed in the irFFT phase four barriers later (hides L2/HBM)
        float sre[5], sim[5];
        {
            const int tcp = wave < nf ? t0 + wave : t0;
#pragma unroll
            for (int r = 0; r < 5; ++r) {
                const int k = r < 4 ? lane + 64 * r : 256;
                sre[r] = (r < 4 || lane == 0) ? specc[((size_t)tcp * 2 + 0) * kBinsPad + k] : 0.0f;
                sim[r] = (r < 4 || lane == 0) ? specc[((size_t)tcp * 2 + 1) * kBinsPad + k] : 0.0f;
            }
        }
        // ---- stage S = (x + e1) of the tile: own-position, fully coalesced loads                    (:527)
        for (int idx = tid; idx < nf * kFw; idx += kFusedThreads) {
            float a[16], b[16];
            pl_ld16(xc, P, t0 * kFw + idx, a);
            pl_ld16(e1c, P, t0 * kFw + idx, b);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                S[q * kTileP + idx] = make_float4(a[4 * q] + b[4 * q], a[4 * q + 1] + b[4 * q + 1], a[4 * q + 2] + b[4 * q + 2], a[4 * q + 3] + b[4 * q + 3]);
        }
        __syncthreads();    // also: the previous tile's finalize is done with acc, and its FFT buffers (= S) are free
        ADE_CLK(49);
        // ---- ConvTranspose2d(16->16,(1,5),s(1,2),p(0,2),groups 2) + BN + PReLU: one lane per input column m ->
        //      outputs fo = 2m (taps 0,2,4 <- m+1,m,m-1) and 2m+1 (taps 1,3 <- m+1,m) -> D (LDS)      (:515)
        for (int idx = tid; idx < nf * kFw; idx += kFusedThreads) {
            const int tl = idx / kFw, m = idx - tl * kFw;
            cfptr cw = cptr(c3.w);
            ADE_KEEP_IN_LOOP(cw);
            float ev[16], od[16];
#pragma unroll
            for (int co = 0; co < 16; ++co) { ev[co] = c3b[co]; od[co] = c3b[co]; }
#pragma unroll
            for (int dlt = -1; dlt <= 1; ++dlt) {
                const int fi = m + dlt;
                if (fi < 0 || fi >= kFw) continue;
                const int ke = 2 - 2 * dlt, ko = 3 - 2 * dlt;
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const float4 xa = S[(2 * g) * kTileP + idx + dlt], xb = S[(2 * g + 1) * kTileP + idx + dlt];
                    const float xv[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
#pragma unroll
                    for (int ci = 0; ci < 8; ++ci)
#pragma unroll
                        for (int co = 0; co < 8; ++co) {
                            ev[g * 8 + co] += cw[((ke * 2 + g) * 8 + ci) * 8 + co] * xv[ci];
                            if (dlt >= 0) od[g * 8 + co] += cw[((ko * 2 + g) * 8 + ci) * 8 + co] * xv[ci];
                        }
                }
            }
            const int pe = tl * kF1 + 2 * m;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                D[q * kTileP1 + pe] = make_float4(prelu_f(ev[4 * q], c3.slope), prelu_f(ev[4 * q + 1], c3.slope),
                                                  prelu_f(ev[4 * q + 2], c3.slope), prelu_f(ev[4 * q + 3], c3.slope));
                if (2 * m + 1 < kF1)
                    D[q * kTileP1 + pe + 1] = make_float4(prelu_f(od[4 * q], c3.slope), prelu_f(od[4 * q + 1], c3.slope),
                                                          prelu_f(od[4 * q + 2], c3.slope), prelu_f(od[4 * q + 3], c3.slope));
            }
        }
        __syncthreads();
        // ---- D += e0 tile (own-position coalesced loads)                                               (:528)
        for (int idx = tid; idx < nf * kF1; idx += kFusedThreads) {
            float b[16];
            pl_ld16(e0c, P0, t0 * kF1 + idx, b);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 d = D[q * kTileP1 + idx];
                d.x += b[4 * q]; d.y += b[4 * q + 1]; d.z += b[4 * q + 2]; d.w += b[4 * q + 3];
                D[q * kTileP1 + idx] = d;
            }
        }
        __syncthreads();
        ADE_CLK(50);
        // ---- ConvTranspose2d(16->2) + BN + Tanh -> mask tile M (LDS)                                   (:516)
        for (int idx = tid; idx < nf * kF1; idx += kFusedThreads) {
            const int tl = idx / kF1, m = idx - tl * kF1;
            float ev[2] = {c4b[0], c4b[1]}, od[2] = {c4b[0], c4b[1]};
#pragma unroll
            for (int dlt = -1; dlt <= 1; ++dlt) {
                const int fi = m + dlt;
                if (fi < 0 || fi >= kF1) continue;
                const int ke = 2 - 2 * dlt, ko = 3 - 2 * dlt;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 xq = D[q * kTileP1 + idx + dlt];
                    const float xv[4] = {xq.x, xq.y, xq.z, xq.w};
#pragma unroll
                    for (int c = 0; c < 4; ++c)
#pragma unroll
                        for (int co = 0; co < 2; ++co) {
                            ev[co] += c4w[(ke * 16 + 4 * q + c) * 2 + co] * xv[c];
                            if (dlt >= 0) od[co] += c4w[(ko * 16 + 4 * q + c) * 2 + co] * xv[c];
                        }
                }
            }
            float* mr = M + (size_t)tl * 2 * kErbPad;
#pragma unroll
            for (int co = 0; co < 2; ++co) {
                mr[co * kErbPad + 2 * m] = tanhf(ev[co]);
                if (2 * m + 1 < kErb) mr[co * kErbPad + 2 * m + 1] = tanhf(od[co]);
            }
        }
        __syncthreads();    // mask complete; S is dead from here on: its memory becomes the 16 FFT buffers
        ADE_CLK(51);
        // ---- ERB split + complex ratio mask + irFFT-512 + synthesis window + overlap-add, one wavefront per frame.
        //      Neighbouring frames run concurrently and overlap by 256 samples: see the two-parity add below.
        {
            const bool live = wave < nf;
            const float* mr = M + (size_t)(live ? wave : 0) * 2 * kErbPad;
#pragma unroll
            for (int r = 0; r < 5; ++r) {
                if (r == 4 && lane != 0) break;
                const int k = r < 4 ? lane + 64 * r : 256;
                const float xr = sre[r], xi = sim[r];
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
            // overlap-add: frames of equal parity never overlap, so even wavefronts add first, odd ones after a barrier
            // (plain, deterministic adds; every sample gets exactly two addends in total)
#pragma unroll
            for (int par = 0; par < 2; ++par) {
                if (live && (wave & 1) == par) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int n = lane + 64 * r;
                        float2* a = reinterpret_cast<float2*>(acc + kHop * wave + 2 * n);
                        float2 c = *a;
                        c.x += v[r].x * (1.0f / 256.0f) * lt.win[2 * n];
                        c.y += -v[r].y * (1.0f / 256.0f) * lt.win[2 * n + 1];
                        *a = c;
                    }
                }
                __syncthreads();
            }
        }
        ADE_CLK(52);
        // ---- finalize the 256*nf samples this tile completed: raw index m = 256*t0 + i, output n = m - 256 (trim N/2),
        //      / sum(w^2), * 32767, clamp, truncating cast          (STFT_Process.py:330-333, Export_GTCRN.py:681,690)
        for (int i4 = tid; i4 < nf * (kHop / 4); i4 += kFusedThreads) {
            const int i = i4 * 4;
            const int n = kHop * t0 + i - kHop;
            if (n < 0 || n >= out_len) continue;
            float v[4], ws[4];
            ld4(acc + i, v);
            ld4(tabs.win_sum + (i & (kHop - 1)), ws);
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = v[u] / ws[u];
            if (fo32) st4(fo32 + n, v);
            if (po) {
                short q[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) q[u] = (short)(int)fminf(fmaxf(v[u] * 32767.0f, -32768.0f), 32767.0f);
                *reinterpret_cast<short4*>(po + n) = make_short4(q[0], q[1], q[2], q[3]);
            }
        }
        __syncthreads();
        // carry the half-finished last 256 samples to the front of acc, clear the rest
        {
            float carry = 0.0f;
            if (tid < kHop) carry = acc[kHop * nf + tid];
            __syncthreads();
            for (int i = tid; i < (int)kBackAccFloats; i += kFusedThreads) acc[i] = i < kHop ? carry : 0.0f;
        }
        // (the S-staging barrier of the next tile orders these writes before the next tile's atomics)
    }
    ADE_CLK(53);
}

}  // namespace stage
}  // namespace ade
