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
//
// Work decomposition.  16 frames x 65 columns = 1040 and 16 x 33 x 2 groups = 1056 are both just over the 1024 lanes of
// the workgroup, so every conv phase is ONE full round over 16 frames x 64 (or 32 columns x 2 channel groups) = exactly
// 1024 lane-tasks -- no idle wavefronts, no integer division, group index wave-uniform so the weights stay scalar
// operands -- plus a small TAIL for the last column of each frame, spread one output channel per lane over 256 (32)
// lanes.  The 256 workgroups of a batch run in lock-step, so HBM would see bursts at every staging phase; instead
// the loads a tile needs (next tile's PCM pairs, this tile's spectrum and e0, next tile's d2/e1) are issued one or
// more compute phases ahead of their use and land in registers while the FMAs run.
// Inter-stage tensors in HBM are channel-quad planar (ade_stage_net.h).
// Reference lines: Export_GTCRN.py:637-647, 594-595, 99-102, 117-141, 159-197, 488-489, 515-516, 104-107, 583-590, 681-690 ;
// STFT_Process.py:303-316, 239-251, 326-336.
#pragma once
#include "ade_stage_net.h"

namespace ade {
namespace stage {

constexpr int kWbuf = 264;            // float2 slots of one wave's FFT / spectrum buffer (257 used)
// frames per tile = wavefronts per workgroup (G::kTileF: 16 / 8); a tile has kTileF * 65 positions of width 65, kTileF * 33 of width 33
constexpr size_t kTabFloats = 512 + 2 * 256 + 2 * 264;   // window | tw256 | tw512 staged in LDS
constexpr int kBmCap = 16;            // ERB-merge band rows kept in LDS (wider band tables are read from L2)
constexpr int kBsCap = 4;             // ERB-split band rows kept in LDS

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
template <class G>
__device__ __forceinline__ LdsTabs stage_tables(float* dst, const FftTabs& t, int tid) {
    constexpr int kFusedThreads = G::kThreads;
    float* win = dst;
    float* tw256 = dst + 512;
    float* tw512 = tw256 + 512;
    for (int i = tid; i < 512; i += kFusedThreads) { win[i] = t.win[i]; tw256[i] = reinterpret_cast<const float*>(t.tw256)[i]; }
    for (int i = tid; i < 514; i += kFusedThreads) tw512[i] = reinterpret_cast<const float*>(t.tw512)[i];
    return LdsTabs{win, reinterpret_cast<const float2*>(tw256), reinterpret_cast<const float2*>(tw512)};
}

// The 4 sample pairs (2n, 2n+1), n = lane + 64 r, of frame t that this lane windows: packed int16 pairs, reflect-padded
// at the chunk edges (STFT_Process.py:306-309).  Plain loads, no dependence on the DC mean: issued a whole tile ahead.
// centre = 0 (streams): the row is [256 carried samples | the push], frame t reads its samples 256 t .. 256 t + 511 as they are.
__device__ __forceinline__ void load_frame_pairs(const int16_t* __restrict__ row, int L, int t, int lane, bool live, bool pair_ok, int* raw, int centre = kNfft / 2) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int n = lane + 64 * r;
        const int j0 = kHop * t + 2 * n - centre;                // even index of the pair
        if (!live) {
            raw[r] = 0;
        } else if (pair_ok && j0 >= 0 && j0 + 1 < L) {            // interior: one aligned 32-bit load
            raw[r] = *reinterpret_cast<const int*>(row + j0);
        } else {
            int ja = j0, jb = j0 + 1;
            ja = ja < 0 ? -ja : (ja >= L ? 2 * (L - 1) - ja : ja);
            jb = jb < 0 ? -jb : (jb >= L ? 2 * (L - 1) - jb : jb);
            raw[r] = ((int)row[ja] & 0xffff) | ((int)row[jb] << 16);
        }
    }
}

// ERB merge of one frame, one band per lane: banded sum == the dense 192x64 matmul term for term    (Export_GTCRN.py:99-102)
__device__ __forceinline__ void erb_merge(const float* wtab, int count, int s0, const float2* buf, int lane, float* fr) {
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f;
    for (int n = 0; n < count; ++n) {
        const float wv = wtab[n * kErbBands + lane];
        const float2 x = buf[kErbLow + min(s0 + n, kErbHigh - 1)];
        a0 += fast_sqrt((x.x * x.x + x.y * x.y) + 1e-12f) * wv;
        a1 += x.x * wv;
        a2 += x.y * wv;
    }
    fr[kErbLow + lane] = a0;
    fr[kErb + kErbLow + lane] = a1;
    fr[2 * kErb + kErbLow + lane] = a2;
}

// ---------------------------------------------------------------------------------------------------------------
// FRONT.  LDS (floats): wbuf[16][264]x2 | feat[16][3][129] | E0[4][1040]x4 | tabs | ERB band rows | red[16]
// ---------------------------------------------------------------------------------------------------------------
constexpr int kC0TailW = 27 * 16;       // conv0 taps k = 0..2 (all 16 output channels): the weights the fo = 64 tail uses
constexpr int kC1TailW = 3 * 2 * 64;    // conv1 taps k = 0..2
template <class G> constexpr size_t front_smem_bytes() {
    return ((size_t)G::kTileF * kWbuf * 2 + (size_t)G::kTileF * 3 * kErb + (size_t)4 * G::kTileF * kF1 * 4 + kTabFloats +
            (size_t)(G::kLean ? 0 : kBmCap) * kErbBands + kC0TailW + kC1TailW + 16) * 4;
}

// sg: the workgroup's segment of the chunk -- frames sg.t0 .. sg.t0 + sg.nT - 1; every frame of this stage is independent of the others
// (the DC mean is over the whole chunk: each segment's workgroup forms the same exact integer sum).
template <class G>
__device__ __forceinline__ void front_stage(float* smem, int chunk, const Seg& sg, const int16_t* __restrict__ pcm, int L, const FftTabs& tabs,
                                            const BandTab& erb, const ConvW& c0, const ConvW& c1, float* __restrict__ spec,
                                            float* __restrict__ e0, float* __restrict__ e1, long long* __restrict__ clk,
                                            const float* __restrict__ dc_rows = nullptr) {
    constexpr int kFusedThreads = G::kThreads, kTileF = G::kTileF, kTileP1 = kTileF * kF1;
    constexpr size_t kFrontFeatFloats = (size_t)kTileF * 3 * kErb;
    constexpr int kGroupShift = (kFusedThreads == 1024 ? 9 : (kFusedThreads == 512 ? 8 : 7));      // lanes of output-channel group g: [g * NT / 2, (g + 1) * NT / 2)
    static_assert(kFusedThreads == 1024 || kFusedThreads == 512 || kFusedThreads == 256, "front / back conv rounds: 64 lanes per frame");
    constexpr int kBmRows = G::kLean ? 0 : kBmCap;                   // ERB-merge rows kept in LDS
    const int T = sg.T, tbeg = sg.t0, tend = sg.t0 + sg.nT;
    float2* wbuf_all = reinterpret_cast<float2*>(smem);
    float* feat = smem + kTileF * kWbuf * 2;
    float4* E0 = reinterpret_cast<float4*>(feat + kFrontFeatFloats);
    float* E0f = reinterpret_cast<float*>(E0);
    float* tabmem = reinterpret_cast<float*>(E0 + 4 * kTileP1);
    float* erbw = tabmem + kTabFloats;
    float* w0t = erbw + kBmRows * kErbBands;     // tail weights live in LDS: a tail lane owns ONE output channel, so they
    float* w1t = w0t + kC0TailW;                 // cannot be scalar operands, and a VMEM load here would have to wait for
    int* red = reinterpret_cast<int*>(w1t + kC1TailW);   // the (in-order) prefetches in flight
    int tid_ = threadIdx.x;
    ADE_OPAQUE_V(tid_);
    const int tid = tid_, wave = tid >> 6, lane = tid & 63;
    const int16_t* row = pcm + (size_t)chunk * L;
    const int P0 = T * kF1, P = T * kFw;
    float* e0c = e0 + (size_t)chunk * kCh * P0;
    float* e1c = e1 + (size_t)chunk * kCh * P;
    float* specc = spec + (size_t)chunk * T * 2 * kBinsPad;
    ADE_CLK(32);
    const bool pair_ok = ((L & 1) == 0) && ((reinterpret_cast<size_t>(row) & 3) == 0);
    int raw[4];                                            // tile 0's samples: in flight while the mean is computed
    const int centre = sg.stream ? 0 : kNfft / 2;
    load_frame_pairs(row, L, tbeg + wave, lane, tbeg + wave < tend, pair_ok, raw, centre);
    const LdsTabs lt = stage_tables<G>(tabmem, tabs, tid);
    const bool erb_lds = erb.count <= kBmRows;
    if (erb_lds)
        for (int i = tid; i < erb.count * kErbBands; i += kFusedThreads) erbw[i] = erb.w[i];
    const int erb_s0 = erb.start[lane];
    for (int i = tid; i < kC0TailW; i += kFusedThreads) w0t[i] = c0.w[i];          // taps k < 3 are the first 27 rows
    for (int i = tid; i < kC1TailW; i += kFusedThreads) w1t[i] = c1.w[i];          // taps k < 3 are the first 6 (k,g) blocks
    const float b0t = c0.b[tid & 15], b1t = c1.b[tid & 15];                        // tail biases (lane -> channel, see below)

    // ---- F1: DC mean of THIS chunk (exact integer sum, one rounding)                     (Export_GTCRN.py:645-647)
    //      (batch-fold models: the mean is per CALL, over all of its windows, and arrives precomputed in dc_rows)
    if (!dc_rows) {
        int s = 0;
        if (((L & 7) == 0) && ((reinterpret_cast<size_t>(row) & 15) == 0)) {      // 8 samples per 16-byte load
            const int4* r4 = reinterpret_cast<const int4*>(row);
#pragma unroll 2
            for (int i = tid; i < (L >> 3); i += kFusedThreads) {
                const int4 q = r4[i];
                s += (int)(short)(q.x & 0xffff) + (q.x >> 16) + (int)(short)(q.y & 0xffff) + (q.y >> 16)
                   + (int)(short)(q.z & 0xffff) + (q.z >> 16) + (int)(short)(q.w & 0xffff) + (q.w >> 16);
            }
        } else {
            for (int i = tid; i < L; i += kFusedThreads) s += (int)row[i];
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
        if (lane == 0) red[wave] = s;
    }
    __syncthreads();
    float dc;
    if (dc_rows) {
        dc = dc_rows[chunk];
    } else {
        long long tot = 0;
#pragma unroll
        for (int i = 0; i < kTileF; ++i) tot += red[i];
        dc = (float)((double)tot / ((double)L * 32768.0));
    }
    ADE_CLK(33);
    long long clk_prev = ADE_CLK_START();
    const cfptr c0b = cptr(c0.b), c1b = cptr(c1.b);
    const cfptr c0e = cptr(c0.w) + 5 * 9 * 16;           // the merged SFE x conv0 terms [3][8][16] (see the main round of conv0)

    for (int t0 = tbeg; t0 < tend; t0 += kTileF) {
        const int nf = tend - t0 < kTileF ? tend - t0 : kTileF;
        // ---- F2-F5: one wavefront per frame of the tile
        {
            int tq = tid;
            ADE_OPAQUE_V(tq);     // per-iteration copy: keeps the FFT's ~40 loop-invariant LDS addresses from being hoisted
            const int wave = tq >> 6, lane = tq & 63;      // out of the tile loop (where they would be spilled to scratch)
            float2* buf = wbuf_all + wave * kWbuf;
            const int t = t0 + wave;
            const bool live = wave < nf;
            float2 v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = lane + 64 * r;
                const float sa = (float)(short)(raw[r] & 0xffff) * (1.0f / 32768.0f) - dc;
                const float sb = (float)(raw[r] >> 16) * (1.0f / 32768.0f) - dc;
                v[r] = make_float2(sa * lt.win[2 * n], sb * lt.win[2 * n + 1]);
            }
            {   // the next tile's samples: issued now, consumed after this tile's two conv phases
                const int tn = t + kTileF;
                load_frame_pairs(row, L, tn, lane, tn < tend, pair_ok, raw, centre);
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
                wave_sync();
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
                        fr[k] = fast_sqrt((x.x * x.x + x.y * x.y) + 1e-12f);      // Export_GTCRN.py:594-595
                        fr[kErb + k] = x.x;
                        fr[2 * kErb + k] = x.y;
                    }
                }
                if (erb_lds) erb_merge(erbw, erb.count, erb_s0, buf, lane, fr);
                else erb_merge(erb.w, erb.count, erb_s0, buf, lane, fr);
            }
        }
        __syncthreads();    // feat of the tile complete (and the previous tile's conv1 is done with E0)
        ADE_CLK(34);
        ADE_CLK_ACC(40);
        // ---- F6-F7a: SFE(3) + Conv2d(9->16,(1,5),s(1,2),p(0,2)) + BN + PReLU -> E0 (LDS) + e0 (HBM).
        //      Main round on the matrix cores: a wavefront per frame, four tiles of 16 output columns fo < 64; lane (g, j) of a tile owns output
        //      channels 4g .. 4g+3 of column j.  The SFE taps o and the conv taps k that read the same input column 2 fo - 3 + (k + o) are ONE
        //      term with the weights summed on the host (45 -> 21 terms per input channel set, K = 3 x 8 with the padding below), which is exact
        //      for fo >= 1: at fo = 0 the conv zero-pads the SFE OUTPUT at p = -1, whose only non-zero contribution (k = 1, o = 2: feat[c][0]) is
        //      taken back by the K slot that pads each channel's 7 terms to 8 (weight -w[1][c*3+2], input feat[c][0] at fo = 0 and zero elsewhere).
        {
            int tq = tid;
            ADE_OPAQUE_V(tq);                               // keep this phase's index arithmetic out of the other phases' live ranges
            const int tl = tq >> 6, g = (tq >> 4) & 3, j = tq & 15;
            if (tl < nf) {
                const float* fr = feat + (size_t)tl * 3 * kErb;
                float wa[6];
#pragma unroll
                for (int kk = 0; kk < 6; ++kk) wa[kk] = c0e[((kk >> 1) * 8 + 4 * (kk & 1) + g) * 16 + j];
                v4f cb;
#pragma unroll
                for (int r = 0; r < 4; ++r) cb[r] = c0b[4 * g + r];
                const float sel0 = prelu_sel(c0.slope);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const int fo = 16 * mt + j;
                    const int q0 = 2 * fo - 3 + g;                          // terms 0-3 of a channel
                    const bool v0 = q0 >= 0;
                    const int q1 = g == 3 ? 0 : 2 * fo + 1 + g;             // terms 4-6, and the correction slot
                    const bool v1 = g == 3 ? fo == 0 : q1 < kErb;
                    v4f d = cb;
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const float x0 = fr[c * kErb + q0], x1 = fr[c * kErb + q1];
                        d = mfma16x16x4(wa[2 * c], v0 ? x0 : 0.0f, d);
                        d = mfma16x16x4(wa[2 * c + 1], v1 ? x1 : 0.0f, d);
                    }
                    const float4 r4 = make_float4(prelu_m(d[0], d[0] * c0.slope, sel0), prelu_m(d[1], d[1] * c0.slope, sel0),
                                                  prelu_m(d[2], d[2] * c0.slope, sel0), prelu_m(d[3], d[3] * c0.slope, sel0));
                    const int idx = tl * kF1 + fo;
                    E0[g * kTileP1 + idx] = r4;
                    *reinterpret_cast<float4*>(e0c + ((size_t)g * P0 + (size_t)t0 * kF1 + idx) * 4) = r4;
                }
            }
        }
        //      Tail: column fo = 64 of every frame, one lane per (frame, output channel); taps k = 0..2 reach p = 126..128.
        if (tid < kTileF * 16) {
            int tq = tid;
            ADE_OPAQUE_V(tq);
            const int tl = tq >> 4, co = tq & 15;
            if (tl < nf) {
                const float* fr = feat + (size_t)tl * 3 * kErb;
                float acc = b0t;
#pragma unroll
                for (int k = 0; k < 3; ++k)
#pragma unroll
                    for (int c = 0; c < 3; ++c)
#pragma unroll
                        for (int o = 0; o < 3; ++o) {
                            const int q = 2 * (kF1 - 1) - 3 + k + o;   // 125 .. 129
                            const float x = q < kErb ? fr[c * kErb + q] : 0.0f;
                            acc += w0t[(k * 9 + c * 3 + o) * 16 + co] * x;
                        }
                acc = prelu_f(acc, c0.slope);
                const int idx = tl * kF1 + (kF1 - 1);
                E0f[((size_t)(co >> 2) * kTileP1 + idx) * 4 + (co & 3)] = acc;
                e0c[((size_t)(co >> 2) * P0 + t0 * kF1 + idx) * 4 + (co & 3)] = acc;
            }
        }
        __syncthreads();
        ADE_CLK(35);
        ADE_CLK_ACC(41);
        // ---- F7b: Conv2d(16->16,(1,5),s2,groups 2) + BN + PReLU from E0 (LDS) -> e1 (HBM)                 (:489)
        //      Main round: one lane per (group, frame, fo < 32), the group's 8 output channels each.
        {
            int tq = tid;
            ADE_OPAQUE_V(tq);
            const int g = __builtin_amdgcn_readfirstlane(tq >> kGroupShift);     // first half of the wavefronts: group 0, second half: group 1
            const int tl = (tq >> 5) & (kTileF - 1), fo = tq & 31;
            if (tl < nf) {
                cfptr cw = cptr(c1.w);
                ADE_KEEP_IN_LOOP(cw);
                float acc[8];
#pragma unroll
                for (int co = 0; co < 8; ++co) acc[co] = c1b[g * 8 + co];
#pragma unroll
                for (int k = 0; k < 5; ++k) {
                    // (zero padding by select, not by branch: a divergent branch here would turn the wave-uniform weight
                    //  addresses after the join into per-lane values, i.e. the scalar weight loads into VMEM loads)
                    const int fi = 2 * fo - 2 + k;                     // <= 64 for fo < 32
                    const bool ok = fi >= 0;
                    const int pp = tl * kF1 + (ok ? fi : 0);
                    const float4 xa = E0[(2 * g) * kTileP1 + pp], xb = E0[(2 * g + 1) * kTileP1 + pp];
                    const float x[8] = {ok ? xa.x : 0.0f, ok ? xa.y : 0.0f, ok ? xa.z : 0.0f, ok ? xa.w : 0.0f,
                                        ok ? xb.x : 0.0f, ok ? xb.y : 0.0f, ok ? xb.z : 0.0f, ok ? xb.w : 0.0f};
#pragma unroll
                    for (int ci = 0; ci < 8; ++ci)
#pragma unroll
                        for (int co = 0; co < 8; ++co) acc[co] += cw[((k * 2 + g) * 8 + ci) * 8 + co] * x[ci];
                }
#pragma unroll
                for (int co = 0; co < 8; ++co) acc[co] = prelu_f(acc[co], c1.slope);
                const int p = t0 * kFw + tl * kFw + fo;
                st4(e1c + ((size_t)(2 * g) * P + p) * 4, acc);
                st4(e1c + ((size_t)(2 * g + 1) * P + p) * 4, acc + 4);
            }
        }
        //      Tail: column fo = 32, one lane per (frame, group, output channel); taps k = 0..2 reach fi = 62..64.
        if (tid < kTileF * 16) {
            int tq = tid;
            ADE_OPAQUE_V(tq);
            const int tl = tq >> 4, g = (tq >> 3) & 1, co = tq & 7;     // (tq & 15 == g*8 + co: the channel b1t was loaded for)
            if (tl < nf) {
                float acc = b1t;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int pp = tl * kF1 + (kF1 - 3) + k;
#pragma unroll
                    for (int ci = 0; ci < 8; ++ci)
                        acc += w1t[((k * 2 + g) * 8 + ci) * 8 + co] * E0f[((size_t)(2 * g + (ci >> 2)) * kTileP1 + pp) * 4 + (ci & 3)];
                }
                acc = prelu_f(acc, c1.slope);
                const int p = t0 * kFw + tl * kFw + (kFw - 1);
                e1c[((size_t)(2 * g + (co >> 2)) * P + p) * 4 + (co & 3)] = acc;
            }
        }
        ADE_CLK(36);
        ADE_CLK_ACC(42);
        // no barrier here: the next tile's FFT phase touches neither E0 nor (before its own barrier) anything conv1 reads
    }
}

// ---------------------------------------------------------------------------------------------------------------
// BACK.  LDS (floats): S[4][528]x4 (reused as the FFT buffers wbuf[16][264]x2) | D[4][1040]x4 | M[16][2][132] |
//                      carry[256] | tabs | win_sum[256] | ERB-split rows [4][192] | ERB-split starts [192] ; the overlap-add accumulator acc[512 + 256 (kTileF - 1)]
//                      aliases D, which is dead from the end of deconv4 to the next tile's deconv3
// ---------------------------------------------------------------------------------------------------------------
constexpr int kC3TailW = 2 * 2 * 64;    // deconv3 taps 2 and 4, both groups
constexpr int kC4TailW = 2 * 16 * 2;    // deconv4 taps 2 and 4
template <class G> constexpr size_t back_smem_bytes() {      // (the lean geometry keeps win_sum, the ERB-split start indices and `pend` out of LDS)
    return ((size_t)4 * G::kTileF * kFw * 4 + (size_t)4 * G::kTileF * kF1 * 4 + (size_t)G::kTileF * 2 * kErbPad + (size_t)kHop + kTabFloats + (size_t)kBsCap * kErbHigh +
            kC3TailW + kC4TailW + (G::kLean ? 0 : kHop + kErbHigh + kHop)) * 4;
}

// sg: the workgroup's segment.  The overlap-add is the only step that crosses frames: a segment with a successor hands on its last
// 256 half-finished samples; a segment with a predecessor parks the first hop of its first frame (pend) and finishes those samples at the
// very end, when the predecessor's carry has arrived -- each sample still gets exactly its two addends (a + b == b + a: bit-exact).
template <class G>
__device__ __forceinline__ void back_stage(float* smem, int chunk, const Seg& sg, const float* __restrict__ x, const float* __restrict__ e1,
                                           const float* __restrict__ e0, const float* __restrict__ spec, const ConvW& c3, const ConvW& c4,
                                           const BandTab& bs, const FftTabs& tabs, int16_t* __restrict__ pcm,
                                           float* __restrict__ f32, long long* __restrict__ clk) {
    constexpr int kFusedThreads = G::kThreads, kTileF = G::kTileF, kTileP1 = kTileF * kF1, kTileP = kTileF * kFw;
    constexpr size_t kBackAccFloats = (size_t)kNfft + (size_t)kHop * (kTileF - 1);
    static_assert((size_t)4 * kTileP * 4 >= (size_t)kTileF * kWbuf * 2, "the S tile must be able to hold the FFT buffers");
    static_assert((size_t)4 * kTileP1 * 4 >= kBackAccFloats, "the D tile must be able to hold the overlap-add accumulator");
    constexpr int kSUnits = 4 * kTileP;        // float4 slots of one S tile
    constexpr int kSThreads = kSUnits / 3;     // 704 (352) lanes x 3 slots stage a tile
    static_assert(kSThreads * 3 == kSUnits && kSThreads <= kFusedThreads, "S staging split");
    constexpr int kGroupShift = (kFusedThreads == 1024 ? 9 : (kFusedThreads == 512 ? 8 : 7));
    constexpr bool kLean = G::kLean;
    const int T = sg.T, tbeg = sg.t0, tend = sg.t0 + sg.nT;
    float4* S = reinterpret_cast<float4*>(smem);
    float* Sf = smem;
    float2* wbuf_all = reinterpret_cast<float2*>(smem);
    float4* D = S + 4 * kTileP;
    float* Df = reinterpret_cast<float*>(D);
    float* M = reinterpret_cast<float*>(D + 4 * kTileP1);
    float* acc = Df;                                           // overlap-add accumulator of a tile: lives in D between deconv4 and the next tile's deconv3
    float* cbuf = M + kTileF * 2 * kErbPad;                    // [256] the half-finished hop carried from one tile to the next
    float* tabmem = cbuf + kHop;
    float* bsw = tabmem + kTabFloats;                          // ERB-split rows
    float* w3t = bsw + kBsCap * kErbHigh;                      // tail weights (see front_stage): [tap 2 | tap 4] x [g][ci][co]
    float* w4t = w3t + kC3TailW;                               // [tap 2 | tap 4] x [ci][co]
    // (lean geometry, 40 KB per workgroup: win_sum and the ERB-split start indices are fetched from global memory into registers at the top of each tile / once per
    //  stage, and the parked first hop waits in the spare floats behind this segment's own exchange slot)
    float* wsum = kLean ? const_cast<float*>(tabs.win_sum) : w4t + kC4TailW;
    int* bss = reinterpret_cast<int*>(wsum + kHop);
    float* pend = kLean ? sg.xo + kXPendOff : reinterpret_cast<float*>(bss + kErbHigh);  // [256] first hop of a segment that has a predecessor
    int tid_ = threadIdx.x;
    ADE_OPAQUE_V(tid_);
    const int tid = tid_;
    const int P0 = T * kF1, P = T * kFw;
    const float* xc = x + (size_t)chunk * kCh * P;
    const float* e1c = e1 + (size_t)chunk * kCh * P;
    const float* e0c = e0 + (size_t)chunk * kCh * P0;
    const float* specc = spec + (size_t)chunk * T * 2 * kBinsPad;
    const int trim = sg.stream ? 0 : kHop;                     // one-shot calls drop the first half window of the overlap-add (centre padding); streams emit it, one hop behind
    const int out_len = sg.stream ? kHop * T : kHop * (T - 1);
    int16_t* po = pcm ? pcm + (size_t)chunk * out_len : nullptr;
    float* fo32 = f32 ? f32 + (size_t)chunk * out_len : nullptr;
    ADE_CLK(48);

    // S = (d2 + e1) of a tile: 3 float4 slots per lane on 704 lanes, slot u = plane * 528 + tile position (== the LDS index).
    // issue_s() only starts the loads; commit_s() adds and writes the tile (called one or more compute phases later).
    // Every lane's sa/sb are (re)defined by each issue_s(), so nothing is live across the tile loop's back edge.
    auto issue_s = [&](int t0n, bool go, float4* sa, float4* sb) {
        const int nfn = tend - t0n < kTileF ? tend - t0n : kTileF;
        int tq = tid;
        ADE_OPAQUE_V(tq);
        const bool on = go && tq < kSThreads;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int u = tq + kSThreads * i, q = u / kTileP, idx = u - q * kTileP;
            const int idc = idx < nfn * kFw ? idx : 0;                                   // (clamped: the value is unused)
            const size_t off = ((size_t)q * P + (size_t)t0n * kFw + idc) * 4;
            sa[i] = on ? *reinterpret_cast<const float4*>(xc + off) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            sb[i] = on ? *reinterpret_cast<const float4*>(e1c + off) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
    };
    auto commit_s = [&](const float4* sa, const float4* sb) {
        int tq = tid;
        ADE_OPAQUE_V(tq);
        if (tq < kSThreads) {
#pragma unroll
            for (int i = 0; i < 3; ++i)
                S[tq + kSThreads * i] = make_float4(sa[i].x + sb[i].x, sa[i].y + sb[i].y, sa[i].z + sb[i].z, sa[i].w + sb[i].w);
        }
    };
    // e0 addends of a tile's deconv3 main round (lane = (group, frame, column m): outputs 2m, 2m+1 of the group's 2 planes)
    auto issue_e0 = [&](int t0n, bool go, float4* ea, float4* eb) {
        const int nfn = tend - t0n < kTileF ? tend - t0n : kTileF;
        int tq = tid;
        ADE_OPAQUE_V(tq);
        const int g = tq >> kGroupShift, tl = (tq >> 5) & (kTileF - 1), m = tq & 31;
        const bool on = go && tl < nfn;
        const int pe = on ? tl * kF1 + 2 * m : 0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float* src = e0c + ((size_t)(2 * g + h) * P0 + (size_t)t0n * kF1 + pe) * 4;
            ea[h] = on ? *reinterpret_cast<const float4*>(src) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            eb[h] = on ? *reinterpret_cast<const float4*>(src + 4) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
    };
    float4 sa0[3], sb0[3];
    issue_s(tbeg, true, sa0, sb0);
    float4 ea[2], eb[2];                    // loop-carried: tile k issues tile k+1's (every lane redefines them each time)
    issue_e0(tbeg, true, ea, eb);
    const LdsTabs lt = stage_tables<G>(tabmem, tabs, tid);
    if (!kLean)
        for (int i = tid; i < kHop; i += kFusedThreads) wsum[i] = tabs.win_sum[i];
    const bool bs_lds = bs.count <= kBsCap;
    int bs_s[4] = {0, 0, 0, 0};                                // lean: start band of the (at most four) high bins this lane masks: k = lane + 64 r (r = 1 .. 3), k = 256 on lane 0
    if (bs_lds) {
        for (int i = tid; i < bs.count * kErbHigh; i += kFusedThreads) bsw[i] = bs.w[i];
        if (!kLean) {
            for (int i = tid; i < kErbHigh; i += kFusedThreads) bss[i] = bs.start[i];
        } else {
            int tq = tid;
            ADE_OPAQUE_V(tq);
            const int lane = tq & 63;
#pragma unroll
            for (int r = 1; r < 5; ++r) {
                const int k = r < 4 ? lane + 64 * r : 256;
                bs_s[r - 1] = (k >= kErbLow && (r < 4 || lane == 0)) ? bs.start[k - kErbLow] : 0;
            }
        }
    }
    for (int i = tid; i < kC3TailW; i += kFusedThreads) w3t[i] = c3.w[(i < 128 ? 2 : 4) * 128 + (i & 127)];
    for (int i = tid; i < kC4TailW; i += kFusedThreads) w4t[i] = c4.w[(i < 32 ? 2 : 4) * 32 + (i & 31)];
    const float b3t = c3.b[tid & 15], b4t = c4.b[tid & 1];
    for (int i = tid; i < kHop; i += kFusedThreads) cbuf[i] = 0.0f;
    const cfptr c3b = cptr(c3.b), c4b = cptr(c4.b), c4w = cptr(c4.w);
    commit_s(sa0, sb0);
    long long clk_prev = ADE_CLK_START();

    for (int t0 = tbeg; t0 < tend; t0 += kTileF) {
        const int nf = tend - t0 < kTileF ? tend - t0 : kTileF;
        // spectrum of this wavefront's frame: issued now, consumed in the irFFT phase three barriers later
        float sre[5], sim[5];
        float wsr[4] = {1.0f, 1.0f, 1.0f, 1.0f};     // lean: this lane's four win_sum values of the finalize step (nf * 64 float4 slots <= one per lane), requested with the spectrum
        if (kLean) ld4(wsum + ((tid * 4) & (kHop - 1)), wsr);      // (turned into reciprocals where they are used: the division below is a multiplication)
        {
            int tq = tid;
            ADE_OPAQUE_V(tq);
            const int wave = tq >> 6, lane = tq & 63;
            const int tcp = wave < nf ? t0 + wave : t0;
#pragma unroll
            for (int r = 0; r < 5; ++r) {
                const int k = r < 4 ? lane + 64 * r : 256;
                sre[r] = (r < 4 || lane == 0) ? specc[((size_t)tcp * 2 + 0) * kBinsPad + k] : 0.0f;
                sim[r] = (r < 4 || lane == 0) ? specc[((size_t)tcp * 2 + 1) * kBinsPad + k] : 0.0f;
            }
        }
        __syncthreads();    // S of this tile (committed in the prologue / at the end of the previous tile) is visible
        ADE_CLK(49);
        ADE_CLK_ACC(56);
        // ---- ConvTranspose2d(16->16,(1,5),s(1,2),p(0,2),groups 2) + BN + PReLU, + e0 -> D (LDS)       (:515, :528)
        //      Main round: one lane per (group, frame, input column m < 32): outputs fo = 2m (taps 0,2,4 <- m+1,m,m-1)
        //      and 2m+1 (taps 1,3 <- m+1,m) of the group's 8 channels; the e0 addends were requested a tile ago (issue_e0).
        {
            int tq = tid;
            ADE_OPAQUE_V(tq);
            const int g = __builtin_amdgcn_readfirstlane(tq >> kGroupShift);
            const int tl = (tq >> 5) & (kTileF - 1), m = tq & 31;
            if (tl < nf) {
                const int idx = tl * kFw + m, pe = tl * kF1 + 2 * m;
                cfptr cw = cptr(c3.w);
                ADE_KEEP_IN_LOOP(cw);
                float ev[8], od[8];
#pragma unroll
                for (int co = 0; co < 8; ++co) { ev[co] = c3b[g * 8 + co]; od[co] = c3b[g * 8 + co]; }
#pragma unroll
                for (int dlt = -1; dlt <= 1; ++dlt) {
                    const bool ok = m + dlt >= 0;                       // m + dlt <= 32 is always a valid column; padding by select
                    const int ke = 2 - 2 * dlt, ko = 3 - 2 * dlt;
                    const int ps = idx + (ok ? dlt : 0);
                    const float4 xa = S[(2 * g) * kTileP + ps], xb = S[(2 * g + 1) * kTileP + ps];
                    const float xv[8] = {ok ? xa.x : 0.0f, ok ? xa.y : 0.0f, ok ? xa.z : 0.0f, ok ? xa.w : 0.0f,
                                         ok ? xb.x : 0.0f, ok ? xb.y : 0.0f, ok ? xb.z : 0.0f, ok ? xb.w : 0.0f};
#pragma unroll
                    for (int ci = 0; ci < 8; ++ci)
#pragma unroll
                        for (int co = 0; co < 8; ++co) {
                            ev[co] += cw[((ke * 2 + g) * 8 + ci) * 8 + co] * xv[ci];
                            if (dlt >= 0) od[co] += cw[((ko * 2 + g) * 8 + ci) * 8 + co] * xv[ci];
                        }
                }
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    D[(2 * g + h) * kTileP1 + pe] =
                        make_float4(prelu_f(ev[4 * h], c3.slope) + ea[h].x, prelu_f(ev[4 * h + 1], c3.slope) + ea[h].y,
                                    prelu_f(ev[4 * h + 2], c3.slope) + ea[h].z, prelu_f(ev[4 * h + 3], c3.slope) + ea[h].w);
                    D[(2 * g + h) * kTileP1 + pe + 1] =
                        make_float4(prelu_f(od[4 * h], c3.slope) + eb[h].x, prelu_f(od[4 * h + 1], c3.slope) + eb[h].y,
                                    prelu_f(od[4 * h + 2], c3.slope) + eb[h].z, prelu_f(od[4 * h + 3], c3.slope) + eb[h].w);
                }
            }
        }
        //      Tail: input column m = 32 -> output fo = 64 only (taps 2,4 <- m, m-1); one lane per (frame, group, channel).
        if (tid < kTileF * 16) {
            int tq = tid;
            ADE_OPAQUE_V(tq);
            const int tl = tq >> 4, g = (tq >> 3) & 1, co = tq & 7;
            if (tl < nf) {
                const int pe = tl * kF1 + (kF1 - 1);
                const float eadd = e0c[((size_t)(2 * g + (co >> 2)) * P0 + (size_t)t0 * kF1 + pe) * 4 + (co & 3)];
                float ev = b3t;
#pragma unroll
                for (int dlt = -1; dlt <= 0; ++dlt) {
                    const int ps = tl * kFw + (kFw - 1) + dlt;
#pragma unroll
                    for (int ci = 0; ci < 8; ++ci)      // tap ke = 2 - 2 dlt -> w3t block (dlt + 1 ? 0 : 1)
                        ev += w3t[(dlt < 0 ? 128 : 0) + (g * 8 + ci) * 8 + co] * Sf[((size_t)(2 * g + (ci >> 2)) * kTileP + ps) * 4 + (ci & 3)];
                }
                Df[((size_t)(2 * g + (co >> 2)) * kTileP1 + pe) * 4 + (co & 3)] = prelu_f(ev, c3.slope) + eadd;
            }
        }
        __syncthreads();    // D complete; S is dead until commit_s() at the end of the tile
        ADE_CLK(50);
        ADE_CLK_ACC(57);
        // Next tile's d2/e1 and e0 addends: requested here, consumed at the end of this tile / at the end of the next deconv3.
        // (These two tensors are the back stage's HBM stream -- ~130 KB per tile per workgroup, all 256 workgroups in
        // lock-step -- and the request queue back-pressures the issuing wavefronts for about as long as HBM needs to
        // deliver them, wherever in the tile the requests are placed: measured, not assumed.)
        float4 sa[3], sb[3];
        const bool has_next = t0 + kTileF < tend;
        issue_s(has_next ? t0 + kTileF : t0, has_next, sa, sb);
        issue_e0(has_next ? t0 + kTileF : t0, has_next, ea, eb);
        // ---- ConvTranspose2d(16->2) + BN + Tanh -> mask tile M (LDS)                                   (:516)
        //      Main round: one lane per (frame, input column m < 64) -> mask bins 2m, 2m+1.
        {
            int tq = tid;
            ADE_OPAQUE_V(tq);
            const int tl = tq >> 6, m = tq & 63;
            if (tl < nf) {
                const int idx = tl * kF1 + m;
                v2f ev = mk2(c4b[0], c4b[1]), od = ev;                 // the two mask channels of a bin: one packed accumulator (weights are channel pairs)
#pragma unroll
                for (int dlt = -1; dlt <= 1; ++dlt) {
                    const bool ok = m + dlt >= 0;                       // m + dlt <= 64 is always a valid column; padding by select
                    const int ke = 2 - 2 * dlt, ko = 3 - 2 * dlt;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 xq = D[q * kTileP1 + idx + (ok ? dlt : 0)];
                        const float xv[4] = {ok ? xq.x : 0.0f, ok ? xq.y : 0.0f, ok ? xq.z : 0.0f, ok ? xq.w : 0.0f};
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            ev += mk2(c4w[(ke * 16 + 4 * q + c) * 2], c4w[(ke * 16 + 4 * q + c) * 2 + 1]) * xv[c];
                            if (dlt >= 0) od += mk2(c4w[(ko * 16 + 4 * q + c) * 2], c4w[(ko * 16 + 4 * q + c) * 2 + 1]) * xv[c];
                        }
                    }
                }
                float* mr = M + (size_t)tl * 2 * kErbPad;
#pragma unroll
                for (int co = 0; co < 2; ++co) {
                    mr[co * kErbPad + 2 * m] = tanh_f(ev[co]);
                    mr[co * kErbPad + 2 * m + 1] = tanh_f(od[co]);
                }
            }
        }
        //      Tail: input column m = 64 -> mask bin 128 only (taps 2,4 <- m, m-1); one lane per (frame, mask channel).
        if (tid < kTileF * 2) {
            const int tl = tid >> 1, co = tid & 1;
            if (tl < nf) {
                float ev = b4t;
#pragma unroll
                for (int dlt = -1; dlt <= 0; ++dlt) {
                    const int ps = tl * kF1 + (kF1 - 1) + dlt;
#pragma unroll
                    for (int ci = 0; ci < 16; ++ci)
                        ev += w4t[(dlt < 0 ? 32 : 0) + ci * 2 + co] * Df[((size_t)(ci >> 2) * kTileP1 + ps) * 4 + (ci & 3)];
                }
                M[(size_t)tl * 2 * kErbPad + co * kErbPad + (kErb - 1)] = tanh_f(ev);
            }
        }
        __syncthreads();    // mask complete; the S region becomes the 16 FFT buffers
        ADE_CLK(51);
        ADE_CLK_ACC(58);
        // ---- ERB split + complex ratio mask + irFFT-512 + synthesis window + overlap-add, one wavefront per frame.
        //      Neighbouring frames run concurrently and overlap by 256 samples: see the two-parity add below.
        {
            int tq = tid;
            ADE_OPAQUE_V(tq);
            const int wave = tq >> 6, lane = tq & 63;
            float2* buf = wbuf_all + wave * kWbuf;
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
                    const int o = k - kErbLow;
                    m0 = 0.0f; m1 = 0.0f;
                    if (bs_lds) {
                        const int s0 = kLean ? bs_s[r > 0 ? r - 1 : 0] : bss[o];
                        for (int n = 0; n < bs.count; ++n) {
                            const float wv = bsw[n * kErbHigh + o];
                            const int jj = min(s0 + n, kErbBands - 1);
                            m0 += mr[kErbLow + jj] * wv;
                            m1 += mr[kErbPad + kErbLow + jj] * wv;
                        }
                    } else {
                        const int s0 = bs.start[o];
                        for (int n = 0; n < bs.count; ++n) {
                            const float wv = bs.w[n * kErbHigh + o];
                            const int jj = min(s0 + n, kErbBands - 1);
                            m0 += mr[kErbLow + jj] * wv;
                            m1 += mr[kErbPad + kErbLow + jj] * wv;
                        }
                    }
                }
                // (:585-590; which product of a difference is contracted into the FMA is spelled out, so that every instantiation of this stage rounds alike)
                float2 y = make_float2(__fmaf_rn(xr, m0, -(xi * m1)), __fmaf_rn(xi, m0, xr * m1));
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
            //   (the windowed sample is ROUNDED, then added: a sample's two addends then commute, so the result depends neither on the tile
            //    size -- which frame of an overlapping pair adds first -- nor on a segment adding its predecessor's carry last: `park` below.
            //    Left to the compiler's contraction this was a fused multiply-add onto whichever addend happened to be there first.)
            //   The accumulator is never cleared: the even pass WRITES (its frames tile the span without overlap; the first hop adds the carry of the previous
            //   tile), the odd pass adds -- except the last frame's second half when no even frame follows, which it writes.
            const bool park = sg.prev && t0 == tbeg && wave == 0;      // first frame of a segment with a predecessor
#pragma unroll
            for (int par = 0; par < 2; ++par) {
                if (live && (wave & 1) == par) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int n = lane + 64 * r;
                        const float pa = v[r].x * (1.0f / 256.0f), pb = -v[r].y * (1.0f / 256.0f);
                        float wa = pa * lt.win[2 * n], wb = pb * lt.win[2 * n + 1];
                        ADE_OPAQUE_V(wa);                                    // (the products are final: nothing downstream may re-fuse them)
                        ADE_OPAQUE_V(wb);
                        float2* a = reinterpret_cast<float2*>(acc + kHop * wave + 2 * n);
                        if (park && r < 2) {                                 // first hop: waits in `pend` for the predecessor's carry
                            *reinterpret_cast<float2*>(pend + 2 * n) = make_float2(wa, wb);
                        } else if (par == 0) {
                            float2 c = make_float2(0.0f, 0.0f);
                            if (wave == 0 && r < 2) c = *reinterpret_cast<const float2*>(cbuf + 2 * n);     // what the previous tile's last frame left
                            *a = make_float2(c.x + wa, c.y + wb);
                        } else if (r < 2 || wave + 1 < nf) {
                            float2 c = *a;
                            c.x += wa;
                            c.y += wb;
                            *a = c;
                        } else {
                            *a = make_float2(0.0f + wa, 0.0f + wb);
                        }
                    }
                }
                __syncthreads();
            }
        }
        ADE_CLK(52);
        ADE_CLK_ACC(59);
        // the FFT buffers are dead: S of the next tile (loads issued before deconv4) goes in
        if (has_next) commit_s(sa, sb);
        // ---- finalize the 256*nf samples this tile completed: raw index m = 256*t0 + i, output n = m - 256 (trim N/2),
        //      / sum(w^2), * 32767, clamp, truncating cast          (STFT_Process.py:330-333, Export_GTCRN.py:681,690)
        int tf = tid;
        ADE_OPAQUE_V(tf);
        for (int i4 = tf; i4 < nf * (kHop / 4); i4 += kFusedThreads) {
            const int i = i4 * 4;
            const int n = kHop * t0 + i - trim;
            if (n < 0 || n >= out_len) continue;
            float v[4], ws[4];
            ld4(acc + i, v);
            if (sg.stream && sg.first && t0 == 0 && i < kHop) { v[0] = 0.0f; v[1] = 0.0f; v[2] = 0.0f; v[3] = 0.0f; }   // a stream's very first hop has no predecessor: blank
            if (sg.prev && t0 == tbeg && i < kHop) continue;      // this hop waits for the predecessor's carry (after the tile loop)
            if (kLean) { ws[0] = wsr[0]; ws[1] = wsr[1]; ws[2] = wsr[2]; ws[3] = wsr[3]; }       // (i = 4 tid there: one round)
            else ld4(wsum + (i & (kHop - 1)), ws);
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = v[u] * fast_rcp(ws[u]);                            // (STFT_Process.py:330: / sum(w^2); the sum is within [0.99, 1.01])
            if (fo32) st4(fo32 + n, v);
            if (po) {
                short q[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) q[u] = (short)(int)fminf(fmaxf(v[u] * 32767.0f, -32768.0f), 32767.0f);
                *reinterpret_cast<short4*>(po + n) = make_short4(q[0], q[1], q[2], q[3]);
            }
        }
        ADE_CLK_ACC(60);
        // the half-finished last 256 samples wait in cbuf for the next tile (acc itself dies with the next deconv3)
        {
            float carry = 0.0f;
            if (tf < kHop) carry = acc[kHop * nf + tf];
            if (sg.next && !has_next && tf < kHop) xst1(sg.xo + kXOlaOff + tf, carry);     // the last tile's carry belongs to the next segment
            if (tf < kHop) cbuf[tf] = carry;
        }
        // (the barrier at the top of the next tile orders these writes, and commit_s(), before their readers)
        ADE_CLK_ACC(61);
    }
    if (sg.next) {
        xdrain();
        __syncthreads();
        if (tid == 0) xflag_store(sg.fo + kXFlagOla, 1u);
    }
    if (sg.prev) {   // the first hop of this segment: the parked samples + the predecessor's carry; then the same tail as above
        if (tid == 0) xwait(sg.fi + kXFlagOla, sg.err, xcode(kXFlagOla));
        __syncthreads();
        int tf = tid;
        ADE_OPAQUE_V(tf);
        for (int i4 = tf; i4 < kHop / 4; i4 += kFusedThreads) {
            const int i = i4 * 4;
            const int n = kHop * tbeg + i - trim;
            float v[4], ws[4];
            ld4(pend + i, v);
            ld4(wsum + i, ws);
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = (xld1(sg.xi + kXOlaOff + i + u) + v[u]) * fast_rcp(ws[u]);
            if (fo32) st4(fo32 + n, v);
            if (po) {
                short q[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) q[u] = (short)(int)fminf(fmaxf(v[u] * 32767.0f, -32768.0f), 32767.0f);
                *reinterpret_cast<short4*>(po + n) = make_short4(q[0], q[1], q[2], q[3]);
            }
        }
    }
    ADE_CLK(53);
}

}  // namespace stage
}  // namespace ade
