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

#include <type_traits>

#ifndef ADE_DP_INTER_MFMA
#define ADE_DP_INTER_MFMA 0      // 1: the inter-frame GRU batched over columns on the matrix cores (measured slower: DESIGN section 6, round 6)
#endif

namespace ade {
namespace stage {

using namespace dev;


// Geometry of a workgroup: NT threads own up to TMAX consecutive frames of a chunk (see ade_internal.h: two geometries are compiled).
template <int NT, int TMAX, int WAVES_PER_SIMD>
struct Geo {
    static constexpr int kThreads = NT;
    static constexpr int kWavesPerSimd = WAVES_PER_SIMD;          // __launch_bounds__ second argument: workgroups per CU x NT / 256
    static constexpr int kTmax = TMAX;
    static constexpr int kPmax = TMAX * kFw;                      // positions of the LDS-resident activation
    static constexpr int kPosPerThread = (kPmax + NT - 1) / NT;   // 3 for both geometries
    static constexpr int kTileF = NT / 64;                        // frames per front / back tile = wavefronts per workgroup
    static constexpr int kProdBase = ((TMAX * (kFw / 3) + 63) / 64) * 64;   // first lane of the wavefronts that are idle in the conv phases
    static constexpr bool kLean = NT < 512;                       // 40 KB of LDS per workgroup: the front / back stages read their small tables from L1 / L2 instead of LDS copies
    static_assert(kPosPerThread == 3, "the conv phases are written for three positions per lane");
    static_assert(kProdBase < NT, "the history producers need a wavefront outside the conv lanes");
};
typedef Geo<1024, 64, 4> Geo0;     // one workgroup per CU
typedef Geo<512, 32, 4> Geo1;      // two workgroups per CU (4 waves per SIMD = 2 x 512 / 256)
typedef Geo<256, 16, 4> Geo2;      // four workgroups per CU

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
template <class G> constexpr size_t gt_smem_bytes() { return (size_t)4 * G::kPmax * 16 + 2 * G::kTmax * 8 * 4; }

#define ADE_REP16(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)

// Optional phase clocks: when `clk` is non-null, thread 0 of workgroup 0 stamps wall_clock64() at each phase boundary.
#define ADE_CLK(i) do { if (clk && chunk == 0 && threadIdx.x == 0) clk[i] = wall_clock64(); } while (0)
// Accumulating variant for phases inside a tile loop: slot i += time since the previous ADE_CLK_ACC (needs a local
// `long long clk_prev` initialised with ADE_CLK_START()).
#define ADE_CLK_START() ((clk && chunk == 0 && threadIdx.x == 0) ? wall_clock64() : 0)
#define ADE_CLK_ACC(i) do { if (clk && chunk == 0 && threadIdx.x == 0) { const long long n_ = wall_clock64(); clk[i] += n_ - clk_prev; clk_prev = n_; } } while (0)

// x1_in_lds : the previous stage of the same launch already left this block's pointwise input (a+skip)[:, :8] in LDS planes 2-3.
// next_x1   : leave the NEXT GTConvBlock's pointwise input there (out[:, :8] + next_skip[:, :8]); next_skip may be null.
// next_full : leave the whole output (16 channels) in planes 0-3: the input of the DPGRNN that follows in the same launch (its x_in_lds).
// store_lo  : false = planes 0-1 (channels 0-7) of `out` are not written to HBM: their only reader is the next block, which gets them through LDS (next_x1),
//             and no later stage takes this tensor as a skip (decoder blocks 0 and 1, DPGRNN 1).  Option "full_taps" stores them anyway for ade_debug_tap.
// sg / blk  : this workgroup's segment of the chunk and the block's index (0-2 encoder, 3-5 decoder) in the exchange area.  The segment's
//             frames are LDS-local 0 .. sg.nT-1; the HBM tensors are addressed through bases shifted to the segment's first frame and the
//             chunk's plane stride Ps.  A segment with a successor hands on (a) the depthwise convolution's history as PARTIAL SUMS: the
//             successor's first 2 x dilation frames start their accumulators from bias + the taps that reach back into this segment,
//             added here in the very order a whole-chunk workgroup adds them (so the split is bit-exact), computed by wavefronts that
//             idle during the conv phases; (b) the TRA GRU state after its last frame.
template <class G>
__device__ __forceinline__ void gtblock_stage(float4* smem, int chunk, const Seg& sg, int blk, const float* __restrict__ a, const float* __restrict__ skip,
                                              const GtConvW& w, float* __restrict__ out, long long* __restrict__ clk,
                                              bool x1_in_lds = false, bool next_x1 = false, const float* __restrict__ next_skip = nullptr, bool store_lo = true,
                                              bool next_full = false) {
    constexpr int kFusedThreads = G::kThreads, kTmaxFused = G::kTmax, kPmax = G::kPmax, kPosPerThread = G::kPosPerThread;
    float4* H = smem;
    float* zt = reinterpret_cast<float*>(smem + 4 * kPmax);
    float* at = zt + kTmaxFused * 8;
    float* GI = reinterpret_cast<float*>(smem + 2 * kPmax);
    float* HS = GI + kTmaxFused * 48;
    const int T = sg.nT;
    const int P = T * kFw;                 // positions this workgroup owns
    const int Ps = sg.T * kFw;             // plane stride of the chunk's HBM tensors
    int tid_ = threadIdx.x;
    // The conv phases occupy the first 5.5 of a 512-thread workgroup's 8 wavefronts, i.e. (wavefront w sits on SIMD w % 4) two on SIMDs 0 / 1
    // and one on SIMDs 2 / 3.  The two workgroups of a CU would both lean on SIMDs 0 / 1: odd segments swap wavefronts 4, 5 with 6, 7 so
    // that theirs land on SIMDs 2 / 3.  A pure renaming of lanes (whole wavefronts move; every role below is keyed by `tid`).
    if (kFusedThreads == 512 && sg.swap) tid_ ^= (tid_ & 256) >> 1;
    ADE_OPAQUE_V(tid_);
    const int tid = tid_;
    const size_t cbase = (size_t)chunk * kCh * Ps + (size_t)sg.t0 * kFw * 4;
    const float* ac = a + cbase;
    const float* sc = skip ? skip + cbase : nullptr;
    float* oc = out + cbase;
    const cfptr c_pw1_b = cptr(w.pw1_b), c_dw_b = cptr(w.dw_b), c_pw2_b = cptr(w.pw2_b);
    // Private copies of the weight base pointers: `w` arrives as one 16-dword SGPR tuple, and every use of a member
    // would otherwise restore the whole tuple from its spill lanes (16 v_readlane per weight row).
    cfptr c_pw1, c_dw, c_pw2;
    ADE_SCALAR_COPY(c_pw1, cptr(w.pw1));
    ADE_SCALAR_COPY(c_dw, cptr(w.dw));
    ADE_SCALAR_COPY(c_pw2, cptr(w.pw2));
    const float pw1_slope = w.pw1_slope, dw_slope = w.dw_slope;
    const float pw1_sel = prelu_sel(pw1_slope), dw_sel = prelu_sel(dw_slope);
    const int dilation = w.dilation;
    ADE_CLK(0);

    // ---- phase 0: stage the pointwise input x1 = (a + skip)[:, :8] in LDS planes 2-3 (own position, fully coalesced);
    //      skipped when the previous stage of this launch left it there.
    if (!x1_in_lds) {
        // (all of a lane's positions are requested before the first is used: the tensors come from L2 / HBM, a microsecond or two away with every CU busy)
        constexpr int kIt = (kPmax + kFusedThreads - 1) / kFusedThreads;
        float x[kIt][8], y[kIt][8];
#pragma unroll
        for (int i = 0; i < kIt; ++i) {
            const int p = tid + i * kFusedThreads, pc = p < P ? p : P - 1;
            pl_ld8(ac, Ps, pc, 0, x[i]);
            if (sc) pl_ld8(sc, Ps, pc, 0, y[i]);
        }
#pragma unroll
        for (int i = 0; i < kIt; ++i) {
            const int p = tid + i * kFusedThreads;
            if (p >= P) break;
            if (sc) {
#pragma unroll
                for (int k = 0; k < 8; ++k) x[i][k] += y[i][k];
            }
            H[2 * kPmax + p] = make_float4(x[i][0], x[i][1], x[i][2], x[i][3]);
            H[3 * kPmax + p] = make_float4(x[i][4], x[i][5], x[i][6], x[i][7]);
        }
        __syncthreads();
    }
    // ---- conv phases on the matrix cores.  The two pointwise convolutions are GEMMs with the 16 output channels as rows: a wavefront owns
    //      16-POSITION TILES (tile = it * kWaves + wave) and one v_mfma_f32_16x16x4_f32 adds four input channels to the whole 16 x 16 tile.
    //      Lane (g = lane / 16, j = lane % 16) supplies, as B, channel (4 kk + g) of position j -- one ds_read_b32 out of the quad-planar
    //      LDS tile, conflict-free (16 B lane stride within a row of lanes, 4 B between the rows) -- and, as A, the weight of that channel
    //      for output channel j (registers, loaded once per stage); the result registers are output channels 4g .. 4g+3 of position j,
    //      i.e. exactly one float4 of plane g.  The depthwise 3x3 between the two stays on the VALU in the same ownership (a lane = one
    //      position x four channels: 9 taps = 18 packed FMAs), its PReLU output feeds the second GEMM from the registers it is in (K index
    //      = (g, register), the weights are permuted to match), and the 33 tiles spread over ALL wavefronts of the workgroup.
    constexpr int kWaves = kFusedThreads / 64;
    constexpr int kTiles = (kPmax / 16 + kWaves - 1) / kWaves;             // tiles per wavefront: 9 in every geometry
    static_assert(kPmax % 16 == 0, "whole tiles");
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = (tid >> 4) & 3, jn = tid & 15;
    const float* Hf = reinterpret_cast<const float*>(H);
    // ---- phase 1: SFE(3) -> 1x1 (24->16) + BN + PReLU.  Output channels 0-7 (lane rows 0, 1) go straight to planes 0-1; channels 8-15
    //      wait in registers until every wavefront has read its x1 columns out of planes 2-3.  The tile bodies are straight-line code (a
    //      tile past the segment's end works on the clamped last position and stores nothing), so the compiler interleaves the dependent
    //      MFMA chains of neighbouring tiles.                                                                       (:305-310)
    const char* const Hb = reinterpret_cast<const char*>(H);
    constexpr int kZ0 = 4 * kPmax * 16;                    // byte offset of 48 B of zeros (the head of zt, dead during the conv phases): where a tap outside the grid reads
    // Half of a tile's result waits, so the tiles go in PAIRS that share the registers: the odd tile of a pair runs with the rows of A (and the
    // bias) rotated by 8, its channels 8-15 come out in lane rows 0, 1 and wait in the half of hi[] that the even tile leaves unused.
    constexpr int kPairs = (kTiles + 1) / 2;
    float4 hi[kPairs];
    {
        float wa[2][6];                                    // A operands: K block kk = (SFE tap o = kk / 2, input channel 4 (kk % 2) + g), weight row c * 3 + o
#pragma unroll
        for (int kk = 0; kk < 6; ++kk) {
            wa[0][kk] = c_pw1[((4 * (kk & 1) + g) * 3 + (kk >> 1)) * 16 + jn];
            wa[1][kk] = c_pw1[((4 * (kk & 1) + g) * 3 + (kk >> 1)) * 16 + (jn ^ 8)];
        }
        v4f cb[2];
#pragma unroll
        for (int r = 0; r < 4; ++r) { cb[0][r] = c_pw1_b[4 * g + r]; cb[1][r] = c_pw1_b[(4 * g + r) ^ 8]; }
#pragma unroll
        for (int it = 0; it < kTiles; ++it) {
            const int od = it & 1;                         // odd tile of its pair: lane rows 0, 1 hold channels 8-15
            const int pu = (it * kWaves + wave) * 16 + jn, pos = pu < P ? pu : P - 1;
            const int f = pos - (pos / kFw) * kFw;
            const bool lok = f != 0, rok = f != kFw - 1;
            const char* xb = Hb + (pos * 4 + g) * 4;       // channel g of the position's float4; planes and neighbours are constant offsets
            v4f d = cb[od];
#pragma unroll
            for (int kk = 0; kk < 6; ++kk) {
                const int o = kk >> 1;
                float x = *reinterpret_cast<const float*>(xb + ((2 + (kk & 1)) * kPmax + (o - 1)) * 16);
                if (o == 0) x = lok ? x : 0.0f;
                if (o == 2) x = rok ? x : 0.0f;
                d = mfma16x16x4(wa[od][kk], x, d);
            }
            const v2f d0 = mk2(d[0], d[1]), d1 = mk2(d[2], d[3]);
            const v2f a0 = d0 * mk2(pw1_slope, pw1_slope), a1 = d1 * mk2(pw1_slope, pw1_slope);
            const float4 r4 = make_float4(prelu_m(d0[0], a0[0], pw1_sel), prelu_m(d0[1], a0[1], pw1_sel),
                                          prelu_m(d1[0], a1[0], pw1_sel), prelu_m(d1[1], a1[1], pw1_sel));
            const bool now = (g < 2) != (od != 0);         // this lane's four channels are 0-7: plane g (even tile) or g - 2 (odd tile)
            if (now && pu < P) H[(g & 1) * kPmax + pu] = r4;
            if (!od) {
                hi[it >> 1] = r4;
            } else {
                hi[it >> 1] = make_float4(now ? hi[it >> 1].x : r4.x, now ? hi[it >> 1].y : r4.y, now ? hi[it >> 1].z : r4.z, now ? hi[it >> 1].w : r4.w);
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int pr = 0; pr < kPairs; ++pr) {                  // lane rows 2, 3 hold the even tile's channels 8-15, rows 0, 1 the odd tile's
        const int it = 2 * pr + (g < 2 ? 1 : 0);
        const int pu = (it * kWaves + wave) * 16 + jn;
        if (it < kTiles && pu < P) H[(2 + (g & 1)) * kPmax + pu] = hi[pr];
    }
    if (tid < 12) zt[tid] = 0.0f;
    if (sg.prev && tid == 0) xwait(sg.fi + kXFlagHist + blk, sg.err, xcode(kXFlagHist + blk));     // the previous segment's partial sums (normally long there)
    __syncthreads();
    ADE_CLK(1);

    // ---- phase 2: causal dilated depthwise 3x3 + BN + PReLU -> 1x1 (16->8) + BN -> h1 (registers of lane rows 0, 1)   (:311-320)
    //      The second GEMM has 8 output rows: two tiles share one accumulator, the odd tile of a pair through the weights in rows 8-15 (K =
    //      its 16 channels, A block-diagonal), so that lane rows 0, 1 hold h1 of the even tile and rows 2, 3 h1 of the odd tile.
    float4 h1r[kPairs];
    {
        v2f wd[3][3][2];                                   // dw[kt][kf][4g .. 4g+3]
#pragma unroll
        for (int kt = 0; kt < 3; ++kt)
#pragma unroll
            for (int kf = 0; kf < 3; ++kf) {
                const int o = kt * 48 + kf * 16 + 4 * g;
                wd[kt][kf][0] = mk2(c_dw[o], c_dw[o + 1]);
                wd[kt][kf][1] = mk2(c_dw[o + 2], c_dw[o + 3]);
            }
        const v2f db0 = mk2(c_dw_b[4 * g], c_dw_b[4 * g + 1]), db1 = mk2(c_dw_b[4 * g + 2], c_dw_b[4 * g + 3]);
        const int dstep = dilation * kFw * 16;             // bytes between the rows two time taps read
        // The time taps kt < nkt of the depthwise convolution for frame tj (>= T: a frame of the successor), column f, channels 4g .. 4g+3; prow =
        // tj * 33 + f.  kt ascending, kf ascending -- the order of the sum is the same wherever a frame's taps are split between two segments.
        // No branch and no select on the data: a tap whose source frame is not in this segment (zero padding, already in the sum, or the
        // successor's own) or whose column is off the grid reads the zeros at kZ0 instead (one address select each).
        const char* const zp = Hb + kZ0;
        auto dw_taps = [&](v2f& acc0, v2f& acc1, const int tj, const int f, const int prow, const int nkt) {
            const bool lok = f != 0, rok = f != kFw - 1;
            const char* const rb = Hb + ((g * kPmax + prow) * 16 - 16);   // the left neighbour of the position in its own frame
#pragma unroll
            for (int kt = 0; kt < 3; ++kt) {
                if (kt >= nkt) continue;
                const int src = tj - (2 - kt) * dilation;
                // (a frame of this segment reads back into it iff it is late enough; a frame of the successor reads into it iff the tap is old enough)
                const bool tv = nkt == 3 ? (kt == 2 ? true : prow >= (2 - kt) * dilation * kFw) : (src >= 0 && src < T);
                const char* const row = tv ? rb - (2 - kt) * dstep : zp;
                const char* const la = lok ? row : zp;
                const char* const ra = rok ? row : zp;
                const float4 x0 = *reinterpret_cast<const float4*>(la);
                const float4 x1 = *reinterpret_cast<const float4*>(row + 16);
                const float4 x2 = *reinterpret_cast<const float4*>(ra + 32);
                acc0 += wd[kt][0][0] * mk2(x0.x, x0.y);
                acc1 += wd[kt][0][1] * mk2(x0.z, x0.w);
                acc0 += wd[kt][1][0] * mk2(x1.x, x1.y);
                acc1 += wd[kt][1][1] * mk2(x1.z, x1.w);
                acc0 += wd[kt][2][0] * mk2(x2.x, x2.y);
                acc1 += wd[kt][2][1] * mk2(x2.z, x2.w);
            }
        };
        float* const xhist_o = sg.xo + blk * kXHistFloats;
        const float* const xhist_i = sg.xi + blk * kXHistFloats;
        const int NP = 2 * dilation * kFw;                 // positions of the frames whose sums start in the previous segment
        if (sg.next) {
            // History for the NEXT segment: for its first 2 x dilation frames, bias + the time taps that land in THIS segment's frames
            // T + t' - (2 - kt) dilation -- kt ascending, exactly the head of the sum a whole-chunk workgroup forms for that frame.  A segment
            // shorter than the history (a streaming push of two frames) passes on what it received: the sum for frame j = T + t' < 2 x dilation
            // of ITS OWN numbering arrived from its predecessor with the taps older than this segment already in it.
#pragma unroll 1
            for (int pt = wave; pt * 16 < NP; pt += kWaves) {
                const int pu = pt * 16 + jn, pn = pu < NP ? pu : NP - 1;
                const int tn = pn / kFw, f = pn - tn * kFw;
                const int j = tn + T;                      // the successor's frame t' in this segment's numbering
                v2f acc0 = db0, acc1 = db1;
                if (sg.prev && j < 2 * dilation) {         // it starts before this segment's history ends: continue the predecessor's sum
                    const float4 v = xld4(xhist_i, (g * kXHistFrames * kFw + j * kFw + f) * 4);
                    acc0 = mk2(v.x, v.y);
                    acc1 = mk2(v.z, v.w);
                }
                dw_taps(acc0, acc1, j, f, pn + P, 2);
                if (pu < NP) xst4(xhist_o, (g * kXHistFrames * kFw + pn) * 4, make_float4(acc0[0], acc0[1], acc1[0], acc1[1]));
            }
            // The flag goes up HERE, ahead of this segment's own tiles: the successor's depthwise phase waits for nothing else, and the segments of a chunk reach this
            // block within microseconds of each other -- raised after the tile loop, each segment's phase started when its predecessor's ENDED (a chain of whole phases
            // through the chunk's segments: 12 us of waiting in the last segment of a dilation-5 block, profiles/r06_b_phase_latency_256.txt).
            xdrain();
            __syncthreads();
            if (tid == 0) xflag_store(sg.fo + kXFlagHist + blk, 1u);
        }
        float wp[2][4];                                    // A operands of the second GEMM: K block r = (input channel 4g + r); even tile: rows 0-7, odd tile: rows 8-15
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float wv = c_pw2[(4 * g + r) * 8 + (jn & 7)];
            wp[0][r] = jn < 8 ? wv : 0.0f;
            wp[1][r] = jn < 8 ? 0.0f : wv;
        }
        v4f cb2;
#pragma unroll
        for (int r = 0; r < 4; ++r) cb2[r] = c_pw2_b[(4 * g + r) & 7];
        v4f d2 = cb2;
#pragma unroll
        for (int it = 0; it < kTiles; ++it) {
            const int pu = (it * kWaves + wave) * 16 + jn, pos = pu < P ? pu : P - 1;
            const int t = pos / kFw, f = pos - t * kFw;
            v2f acc0 = db0, acc1 = db1;
            if (sg.prev && (it * kWaves + wave) * 16 < NP) {     // (wave-uniform) head of the sum: from the previous segment (bias included)
                const float4 v = xld4(xhist_i, (g * kXHistFrames * kFw + (pos < NP ? pos : NP - 1)) * 4);
                const bool head = pos < NP;
                acc0 = mk2(head ? v.x : db0[0], head ? v.y : db0[1]);
                acc1 = mk2(head ? v.z : db1[0], head ? v.w : db1[1]);
            }
            dw_taps(acc0, acc1, t, f, pos, 3);
            const v2f a0 = acc0 * mk2(dw_slope, dw_slope), a1 = acc1 * mk2(dw_slope, dw_slope);
            const float y[4] = {prelu_m(acc0[0], a0[0], dw_sel), prelu_m(acc0[1], a0[1], dw_sel),
                                prelu_m(acc1[0], a1[0], dw_sel), prelu_m(acc1[1], a1[1], dw_sel)};
            if (!(it & 1)) d2 = cb2;
#pragma unroll
            for (int r = 0; r < 4; ++r) d2 = mfma16x16x4(wp[it & 1][r], y[r], d2);
            if ((it & 1) || it == kTiles - 1) h1r[it >> 1] = make_float4(d2[0], d2[1], d2[2], d2[3]);
        }
    }
    __syncthreads();   // every tap read of H is done: planes may be reused
    ADE_CLK(2);
#pragma unroll
    for (int pr = 0; pr < kPairs; ++pr) {
        const int it = 2 * pr + (g >> 1);
        const int pu = (it * kWaves + wave) * 16 + jn;
        if (it < kTiles && pu < P) H[(g & 1) * kPmax + pu] = h1r[pr];
    }
    __syncthreads();
    ADE_CLK(3);

    // wave 0's recurrent weights for phase 4b: requested NOW -- at the start of 4b the other 15 wavefronts flood the CU's
    // memory queue with their staging loads, and these few loads would wait behind all of them before the recurrence
    // (the critical path) could start
    v2f wr_raw[8];
    float bh_raw = 0.0f;
    {
        const int j = tid & 15, row = (tid >> 4) & 3;
        const int gsel = (row & 1) ? 2 : (row >> 1);
        const float* pk = w.gru + j * 78 + 24 + gsel * 16;           // W_h{gate}[j][:]
#pragma unroll
        for (int m = 0; m < 8; ++m) wr_raw[m] = tid < 64 ? mk2(pk[2 * m], pk[2 * m + 1]) : mk2(0.0f, 0.0f);
        if (tid < 64) bh_raw = w.gru[j * 78 + 75 + gsel];
    }
    // The next block's skip addend (8 channels per position) is requested HERE, two phases before its use: the compiler
    // drains the vector-memory counter before wave 0 enters the serial recurrence (first use of the weights above), and
    // by then -- after the energy phase -- these HBM loads have landed, so the recurrence does not wait for them.
    const float* nsc = (next_x1 && next_skip) ? next_skip + cbase : nullptr;
    float nsk[kPosPerThread][8];
    auto request_next_skip = [&]() {
#pragma unroll
        for (int i = 0; i < kPosPerThread; ++i) {
            const int p = tid + i * kFusedThreads;          // global memory is always walked position-linear (coalesced)
            if (nsc && p < P) {
                pl_ld8(nsc, Ps, p, 0, nsk[i]);
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k) nsk[i][k] = 0.0f;
            }
        }
    };
    request_next_skip();
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
                e[c] = e[c] * (1.0f / (float)kFw);
            }
            const float* pk = w.gru + j * 78;
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                float sacc = pk[72 + g];
#pragma unroll
                for (int k = 0; k < 8; ++k) sacc += pk[g * 8 + k] * e[k];
                if (tr < T) GI[t * 48 + g * 16 + j] = sacc * (g == 2 ? 2.0f * kLog2e : -kLog2e);   // pre-scaled for exp2 (phase 4b)
            }
        }
    }
    if (sg.prev && tid == 0) xwait(sg.fi + kXFlagTra + blk, sg.err, xcode(kXFlagTra + blk));      // the previous segment's last hidden state
    __syncthreads();
    ADE_CLK(4);
    ADE_CLK(5);
    // ---- phase 4b: the serial part, GRU(8->16) over T, on wave 0.  Its step latency IS this stage's critical path
    //      (the other 15 wavefronts wait), so it is written for the length of the dependent chain:
    //      * the three gate mat-vecs run in PARALLEL on the wavefront's 16-lane rows (row 0: r, row 2: z, rows 1/3: W_hn h);
    //      * h_{t-1} lives in 16 SGPRs (v_readlane from the row that produced it), so the mat-vec is 8 packed FMAs with
    //        scalar operands -- no per-step broadcast / rotation traffic;
    //      * gates meet through two cross-row swaps (r -> row 1 for n, n -> row 3 where z already is);
    //      * weights, biases and input projections are pre-scaled by -log2(e) (r, z) and 2 log2(e) (n), so every
    //        activation is exp2 -> add -> rcp with no multiply in front.                              (:149,155)
    if (tid >= 64) {
        // waves 1-15, while wave 0 is busy with the serial recurrence: bypass half (a + skip)[:, 8:] -> planes 0-1 (h1 is dead)
        constexpr int kLanes = kFusedThreads - 64, kIt = (kPmax + kLanes - 1) / kLanes;       // (requests first, as in phase 0)
        float by[kIt][8], y[kIt][8];
#pragma unroll
        for (int i = 0; i < kIt; ++i) {
            const int p = tid - 64 + i * kLanes, pc = p < P ? p : P - 1;
            pl_ld8(ac, Ps, pc, 2, by[i]);
            if (sc) pl_ld8(sc, Ps, pc, 2, y[i]);
        }
#pragma unroll
        for (int i = 0; i < kIt; ++i) {
            const int p = tid - 64 + i * kLanes;
            if (p >= P) break;
            if (sc) {
#pragma unroll
                for (int k = 0; k < 8; ++k) by[i][k] += y[i][k];
            }
            H[p] = make_float4(by[i][0], by[i][1], by[i][2], by[i][3]);
            H[kPmax + p] = make_float4(by[i][4], by[i][5], by[i][6], by[i][7]);
        }
    } else {
        const int j = tid & 15, row = tid >> 4;
        const int gsel = (row & 1) ? 2 : (row >> 1);                 // gate of this row: r, n, z, n
        const float sc = gsel == 2 ? 2.0f * kLog2e : -kLog2e;
        v2f wr[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) wr[m] = wr_raw[m] * sc;
        const float bh = bh_raw * sc;
        int shb[16];                                                 // h_{t-1} (fp32 bit patterns), wave-uniform
        float hv = 0.0f;                                             // h_{t-1}[j] (meaningful in row 3, which computes h_t)
        if (sg.prev) hv = xld1(sg.xi + kXTraOff + blk * 16 + j);     // the recurrence continues from the previous segment's last frame
#define ADE_RL(K) shb[K] = __builtin_amdgcn_readlane(__float_as_int(hv), K);
        ADE_REP16(ADE_RL)
#undef ADE_RL
        float gi = GI[gsel * 16 + j];                                // this row's input projection, fetched one step ahead
        // row 3 stores h_t; the other rows' (meaningless) values go to GI row 0, which is dead once `gi` is loaded --
        // an address select instead of an exec-mask branch in the serial loop
        float* hsp = row == 3 ? HS + j : GI + (tid & 63);
        const int hs_step = row == 3 ? 16 : 0;
        __builtin_amdgcn_s_setprio(3);
        for (int t = 0; t < T; ++t) {
            const int tn = t + 1 < T ? t + 1 : t;
            const float gnx = GI[tn * 48 + gsel * 16 + j];
#pragma unroll
            for (int k = 0; k < 16; ++k) ADE_KEEP_IN_LOOP(shb[k]);   // pin h_{t-1} to scalar registers (no copy back to VGPRs)
            float sh[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) sh[k] = __int_as_float(shb[k]);
            v2f a0 = mk2(bh, 0.0f), a1 = mk2(0.0f, 0.0f);
#pragma unroll
            for (int m = 0; m < 8; m += 2) {
                a0 += wr[m] * mk2(sh[2 * m], sh[2 * m + 1]);
                a1 += wr[m + 1] * mk2(sh[2 * m + 2], sh[2 * m + 3]);
            }
            const v2f a2 = a0 + a1;
            const float a = a2[0] + a2[1];                           // scaled (b_h + W_h{gate} h), every row its own gate
            const float e = __builtin_amdgcn_exp2f(gi + a);
            const float sg = fast_rcp(1.0f + e);                     // rows 0 / 2: r / z
            const float X = swap16(e, sg).a;                         // rows 1 / 3 <- r / z
            const float en = __builtin_amdgcn_exp2f(gi + X * a);     // row 1: e^{2 (gi_n + r (b_hn + W_hn h))}
            const float n = 1.0f - 2.0f * fast_rcp(en + 1.0f);       // row 1: n
            const float n3 = swap32(en, n).a;                        // row 3 <- n (row 3 already holds z in X); `en` is a dead value
            const float hc = n3 + X * (hv - n3);                     // row 3: h_t = (1 - z) n + z h_{t-1}
            hv = hc;
#define ADE_RL(K) shb[K] = __builtin_amdgcn_readlane(__float_as_int(hc), 48 + K);
            ADE_REP16(ADE_RL)
#undef ADE_RL
            *hsp = hc;
            hsp += hs_step;
            gi = gnx;
        }
        set_prio(sg.base_prio);
        if (sg.next) {                                               // hand the state on: row 3 holds h_T
            if (row == 3) xst1(sg.xo + kXTraOff + blk * 16 + j, hv);
            xdrain();
            if (tid == 48) xflag_store(sg.fo + kXFlagTra + blk, 1u);
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
    // ---- phase 5: gate, interleave with the bypass half, store; optionally leave the next block's pointwise input
    //      (out[:, :8] + next_skip[:, :8]) in planes 2-3.  The column-triple owners hold h1 in registers, but HBM wants
    //      position-linear lanes (16 B per lane, 1 KB contiguous per instruction): the gated half goes through LDS planes
    //      2-3 (GI / HS are dead), and every lane then assembles the positions tid, tid+1024, tid+2048.     (:156,324)
#pragma unroll
    for (int pr = 0; pr < kPairs; ++pr) {
        const int it = 2 * pr + (g >> 1);
        const int pu = (it * kWaves + wave) * 16 + jn;
        if (it < kTiles && pu < P) {
            const float4 gt = *reinterpret_cast<const float4*>(at + (pu / kFw) * 8 + 4 * (g & 1));
            H[(2 + (g & 1)) * kPmax + pu] = make_float4(h1r[pr].x * gt.x, h1r[pr].y * gt.y, h1r[pr].z * gt.z, h1r[pr].w * gt.w);
        }
    }
    __syncthreads();
    float n8[kPosPerThread][8];
#pragma unroll
    for (int i = 0; i < kPosPerThread; ++i) {
        const int p = tid + i * kFusedThreads;
        if (p < P) {
            const float4 b0 = H[p], b1 = H[kPmax + p], g0 = H[2 * kPmax + p], g1 = H[3 * kPmax + p];
            const float o[16] = {g0.x, b0.x, g0.y, b0.y, g0.z, b0.z, g0.w, b0.w, g1.x, b1.x, g1.y, b1.y, g1.z, b1.z, g1.w, b1.w};
            if (store_lo) {
                pl_st16(oc, Ps, p, o);
            } else {
                st4(oc + ((size_t)2 * Ps + p) * 4, o + 8);
                st4(oc + ((size_t)3 * Ps + p) * 4, o + 12);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) n8[i][k] = o[k] + nsk[i][k];
            if (next_full) {    // a DPGRNN follows in this launch: its whole input stays in LDS (a position is read and rewritten by its own lane only)
#pragma unroll
                for (int q = 0; q < 4; ++q) H[q * kPmax + p] = make_float4(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]);
            }
        }
    }
    if (next_x1) {
        __syncthreads();    // every lane has read the gated half out of planes 2-3
#pragma unroll
        for (int i = 0; i < kPosPerThread; ++i) {
            const int p = tid + i * kFusedThreads;
            if (p < P) {
                H[2 * kPmax + p] = make_float4(n8[i][0], n8[i][1], n8[i][2], n8[i][3]);
                H[3 * kPmax + p] = make_float4(n8[i][4], n8[i][5], n8[i][6], n8[i][7]);
            }
        }
    }
    ADE_CLK(8);
}

// ---------------------------------------------------------------------------------------------------------
// DPGRNN, whole block for one chunk (Export_GTCRN.py:466-481; GRNN :409-428).
//   x (B,T,33,16) -> intra biGRU along F -> Linear -> LayerNorm((33,16)) -> +x = mid
//                 -> inter GRU along T   -> Linear -> LayerNorm           -> +mid = out
// LDS: R[4][kPmax] float4 (rnn outputs, then mid, then rnn outputs again) | two LayerNorm tables.
// `mid` is parked in the output buffer between the two halves (same thread writes and re-reads it).
// ---------------------------------------------------------------------------------------------------------
template <class G> constexpr size_t dp_smem_bytes() { return (size_t)4 * G::kPmax * 16 + (size_t)2 * kFw * kCh * 4; }   // + 2 LayerNorm tables

// Linear(16,16) on the rnn output + LayerNorm((33,16)) per frame + residual.  The Linear runs on the matrix cores IN PLACE in R: a wavefront owns 16-position tiles
// (tile = it * kWaves + wave), lane (g, j) reads the float4 of plane g at position j -- its four components are the K blocks of four v_mfma_f32_16x16x4_f32 (K index =
// (lane row, component), the weights permuted to match: a dense 16 x 16 product, nothing padded) -- and gets output channels 4g .. 4g+3 of that position back: the float4
// of the same plane and position, which it overwrites.  The LayerNorm of a frame then lives inside ONE 16-lane row: lane j of row t owns columns j
// and 16 + j of frame t (16 channels each) plus channel j of column 32, so that both statistics are a row-rotation all-reduce in registers -- no LDS round trip and
// no barrier after the one that follows the Linear (the element-wise part costs per POSITION: it stays one lane per position, not the Linear's tile ownership).  emit_pos(p, y[16]) / emit_one(p, channel, y): the normalised + residual values of an owned position / of the
// lane's one channel of column 32.  `res`: the residual tensor (quad-planar HBM, plane stride Ps), added here.
template <class G, class EmitPos, class EmitOne>
__device__ __forceinline__ void fc_ln_rows(float4* R, const float* __restrict__ fc, const float* __restrict__ fc_b, const float* __restrict__ ln_w,
                                           const float* __restrict__ ln_b, int T, int P, int tid, const float* __restrict__ res, int Ps, EmitPos emit_pos, EmitOne emit_one) {
    constexpr int kFusedThreads = G::kThreads, kPmax = G::kPmax;
    static_assert(kFusedThreads / 16 >= G::kTmax, "one 16-lane row per frame");
    {
        constexpr int kWaves = kFusedThreads / 64, kTiles = (kPmax / 16 + kWaves - 1) / kWaves;
        int tq = tid;
        ADE_OPAQUE_V(tq);                                    // (this pass's nine tile addresses are its own: not hoisted above the recurrence before the phase)
        const int wave = __builtin_amdgcn_readfirstlane(tq >> 6), g = (tq >> 4) & 3, jn = tq & 15;
        float wa[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) wa[kk] = fc[(4 * g + kk) * 16 + jn];       // A: weight of input channel 4g + kk for output channel jn
        v4f cb;
#pragma unroll
        for (int r = 0; r < 4; ++r) cb[r] = fc_b[4 * g + r];
#pragma unroll
        for (int it = 0; it < kTiles; ++it) {
            const int pu = (it * kWaves + wave) * 16 + jn, pos = pu < P ? pu : P - 1;
            const float4 x = R[g * kPmax + pos];
            v4f d = cb;
            d = mfma16x16x4(wa[0], x.x, d);
            d = mfma16x16x4(wa[1], x.y, d);
            d = mfma16x16x4(wa[2], x.z, d);
            d = mfma16x16x4(wa[3], x.w, d);
            if (pu < P) R[g * kPmax + pu] = make_float4(d[0], d[1], d[2], d[3]);
        }
    }
    __syncthreads();
    int tq = tid;
    ADE_OPAQUE_V(tq);
    const int tr = tq >> 4, j = tq & 15;
    const bool live = tr < T;
    const int t = live ? tr : T - 1;                         // (rows beyond the segment recompute its last frame and emit nothing)
    const int p0 = t * kFw + j, p1 = p0 + 16, p2 = t * kFw + 32;
    float* Rf = reinterpret_cast<float*>(R);
    auto allreduce = [](float s) { s += row_ror<8>(s); s += row_ror<4>(s); s += row_ror<2>(s); s += row_ror<1>(s); return s; };
    // the residual comes from L2 / HBM (a microsecond or two away with every CU busy): requested HERE, ahead of the statistics, used after them
    float rs0[16], rs1[16];
    pl_ld16(res, Ps, p0, rs0);
    const float re = res[((size_t)(j >> 2) * Ps + p2) * 4 + (j & 3)];
    float v0[16], v1[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 a = R[q * kPmax + p0], b = R[q * kPmax + p1];
        v0[4 * q] = a.x; v0[4 * q + 1] = a.y; v0[4 * q + 2] = a.z; v0[4 * q + 3] = a.w;
        v1[4 * q] = b.x; v1[4 * q + 1] = b.y; v1[4 * q + 2] = b.z; v1[4 * q + 3] = b.w;
    }
    const float e = Rf[((size_t)(j >> 2) * kPmax + p2) * 4 + (j & 3)];
    float s = e;
#pragma unroll
    for (int c = 0; c < 16; ++c) s += v0[c] + v1[c];
    const float mean = allreduce(s) * (1.0f / (float)(kFw * kCh));
    const float de = e - mean;
    float q2 = de * de;
#pragma unroll
    for (int c = 0; c < 16; ++c) { const float a = v0[c] - mean, b = v1[c] - mean; q2 += a * a + b * b; }
    const float rstd = fast_rsq(allreduce(q2) * (1.0f / (float)(kFw * kCh)) + 1e-8f);
    {
        float gw[16], gb[16];
        ld16(ln_w + j * kCh, gw);
        ld16(ln_b + j * kCh, gb);
#pragma unroll
        for (int c = 0; c < 16; ++c) v0[c] = ((v0[c] - mean) * rstd * gw[c] + gb[c]) + rs0[c];
        if (live) emit_pos(p0, v0);
    }
    {
        float gw[16], gb[16];
        pl_ld16(res, Ps, p1, rs1);
        ld16(ln_w + (16 + j) * kCh, gw);
        ld16(ln_b + (16 + j) * kCh, gb);
#pragma unroll
        for (int c = 0; c < 16; ++c) v1[c] = ((v1[c] - mean) * rstd * gw[c] + gb[c]) + rs1[c];
        if (live) emit_pos(p1, v1);
    }
    {
        const float ye = (de * rstd * ln_w[32 * kCh + j] + ln_b[32 * kCh + j]) + re;
        if (live) emit_one(p2, j, ye);
    }
}

// The gates of one GRU unit from the four rows (r, z, W_hn h + b_hn, W_in x + b_in) of a batched step, pre-scaled for exp2 (r, z by -log2 e, the n rows by 2 log2 e):
// r = sigmoid, z = sigmoid, n = tanh(nx + r nh), h' = (1 - z) n + z h.  Thirteen vector instructions, six of them exp2 / rcp.
__device__ __forceinline__ float gru_gates(const v4f& d, float h) {
    const float r = fast_rcp(1.0f + __builtin_amdgcn_exp2f(d[0]));
    const float z = fast_rcp(1.0f + __builtin_amdgcn_exp2f(d[1]));
    const float n = 1.0f - 2.0f * fast_rcp(__builtin_amdgcn_exp2f(d[3] + r * d[2]) + 1.0f);
    return n + z * (h - n);
}

// sg / blk: see gtblock_stage.  The intra-frame GRU and both LayerNorms are per frame; only the inter-frame GRU's hidden state (33 x 16)
// crosses from a segment to its successor.
template <class G>
__device__ __forceinline__ void dpgrnn_stage(float4* smem, int chunk, const Seg& sg, int blk, const float* __restrict__ x, const DpW& w, float* __restrict__ out,
                                             long long* __restrict__ clk, bool next_x1 = false,
                                             const float* __restrict__ next_skip = nullptr, bool store_lo = true, bool x_in_lds = false, bool next_full = false) {
    constexpr int kFusedThreads = G::kThreads, kTmaxFused = G::kTmax, kPmax = G::kPmax, kPosPerThread = G::kPosPerThread;
    float4* R = smem;
    float* Rf = reinterpret_cast<float*>(smem);
    float* lnt = reinterpret_cast<float*>(smem + 4 * kPmax);   // LDS copies of two LayerNorm tables [gamma | beta]: the intra pair for phase B, then the inter pair for phase D
    const int T = sg.nT;
    const int P = T * kFw, Ps = sg.T * kFw;
    int tid_ = threadIdx.x;
    ADE_OPAQUE_V(tid_);
    const int tid = tid_;
    for (int i = tid; i < kFw * kCh; i += kFusedThreads) {   // (made visible by the barrier after phase A)
        lnt[i] = w.intra_ln_w[i];
        lnt[kFw * kCh + i] = w.intra_ln_b[i];
    }
    const size_t cbase = (size_t)chunk * kCh * Ps + (size_t)sg.t0 * kFw * 4;
    const float* xc = x + cbase;
    float* oc = out + cbase;

    ADE_CLK(16);
    // ---- phase A: intra GRNN (GRU(8->4) x 2 groups x 2 directions along F, per frame).                      (:441-446,472-473)
    //      BATCHED over frames on the matrix cores: a wavefront owns one (group, direction) pair for 16 frames, and a step of its 16 recurrences is three
    //      v_mfma_f32_16x16x4_f32 whose rows are (unit i, gate c) = 4 i + c with c = r | z | W_hn h | W_in x (the n gate's two halves stay apart because r multiplies
    //      only the first) and whose columns are the 16 frames: two K blocks of input channels (off the dependent chain: issued a step ahead) and one of the four
    //      hidden units.  Lane (g, j) gets (r, z, nh, nx) of unit g of frame j in its four result registers, finishes the gates for that one unit and holds h[g] --
    //      which is exactly the B operand (k = g, column j) of the next step: the hidden state never leaves its lane.  The FMAs of the step (36 per unit) are on
    //      the matrix pipe; the vector pipe keeps the 13 gate instructions.  x is staged in R once (the reverse direction reads columns the forward one has
    //      already passed, so the outputs wait in 33 registers and go to R after a barrier).
    {
        constexpr float kS = -kLog2e, kN = 2.0f * kLog2e;   // pre-scaling for exp2-based activations: r, z by -log2 e, n by 2 log2 e
        if (!x_in_lds) {
            float4 v[kPosPerThread][4];
#pragma unroll
            for (int i = 0; i < kPosPerThread; ++i) {
                const int p = tid + i * kFusedThreads, pc = p < P ? p : P - 1;
#pragma unroll
                for (int q = 0; q < 4; ++q) v[i][q] = *reinterpret_cast<const float4*>(xc + ((size_t)q * Ps + pc) * 4);
            }
#pragma unroll
            for (int i = 0; i < kPosPerThread; ++i) {
                const int p = tid + i * kFusedThreads;
                if (p < P) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) R[q * kPmax + p] = v[i][q];
                }
            }
            __syncthreads();
        }
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int g = (tid >> 4) & 3, j = tid & 15;
        const int grp = (wave >> 1) & 1, dir = wave & 1, t = (wave >> 2) * 16 + j;
        const bool live = t < T;
        const int tc = live ? t : T - 1;
        float ax0, ax1, ah;
        {
            const int c = j & 3;
            const float* pk = w.intra_gru + (grp * 8 + dir * 4 + (j >> 2)) * 42;       // [W_ih 3x8 | W_hh 3x4 | b_ih 3 | b_hh 3] of output row (unit j / 4)
            const float sc_ = c < 2 ? kS : kN;
            const int gi = c == 3 ? 2 : c;                                             // rows r, z, nx read W_i{r, z, n}; row nh reads W_hn
            ax0 = c == 2 ? 0.0f : pk[gi * 8 + g] * sc_;
            ax1 = c == 2 ? 0.0f : pk[gi * 8 + 4 + g] * sc_;
            ah = c == 3 ? 0.0f : pk[24 + c * 4 + g] * sc_;
        }
        v4f cb;
        {
            const float* pb = w.intra_gru + (grp * 8 + dir * 4 + g) * 42;
            cb[0] = (pb[36] + pb[39]) * kS;
            cb[1] = (pb[37] + pb[40]) * kS;
            cb[2] = pb[41] * kN;
            cb[3] = pb[38] * kN;
        }
        const float* const xr = Rf + ((size_t)(grp * 2) * kPmax + tc * kFw) * 4 + g;      // channel 8 grp + g of the frame's column 0; + 4: the next plane
        float hout[kFw];
        auto run = [&](auto dir_c) {
            constexpr int DIR = decltype(dir_c)::value;
            float h = 0.0f;
            __builtin_amdgcn_s_setprio(3);
            v4f dx;
            {
                constexpr int f0 = DIR ? kFw - 1 : 0;
                dx = mfma16x16x4(ax0, xr[f0 * 4], cb);
                dx = mfma16x16x4(ax1, xr[(kPmax + f0) * 4], dx);
            }
#pragma unroll
            for (int s = 0; s < kFw; ++s) {
                v4f d = mfma16x16x4(ah, h, dx);
                if (s + 1 < kFw) {
                    const int fn = DIR ? kFw - 2 - s : s + 1;
                    dx = mfma16x16x4(ax0, xr[fn * 4], cb);
                    dx = mfma16x16x4(ax1, xr[(kPmax + fn) * 4], dx);
                }
                h = gru_gates(d, h);
                hout[s] = h;
            }
            set_prio(sg.base_prio);
            __syncthreads();                                 // every wavefront has read its last x column: R takes the rnn outputs
            float* const hw = Rf + ((size_t)(grp * 2 + DIR) * kPmax + tc * kFw) * 4 + g;
            if (live) {
#pragma unroll
                for (int s = 0; s < kFw; ++s) hw[(DIR ? kFw - 1 - s : s) * 4] = hout[s];
            }
        };
        if (dir) run(std::integral_constant<int, 1>{});
        else run(std::integral_constant<int, 0>{});
    }
    __syncthreads();
    ADE_CLK(17);
    // ---- phase B: intra Linear + LayerNorm + residual -> mid (registers, and back into R for the inter GRU)
    //      (mid is parked in `out` -- L2-resident, re-read by the same thread in phase D -- instead of 48 live VGPRs)
    fc_ln_rows<G>(R, w.intra_fc, w.intra_fc_b, lnt, lnt + kFw * kCh, T, P, tid, xc, Ps,
        [&](int p, const float (&m)[16]) {                  // mid: input of the inter-frame GRU (R) and, parked in the output tensor, the residual of phase D
#pragma unroll
            for (int q = 0; q < 4; ++q) R[q * kPmax + p] = make_float4(m[4 * q], m[4 * q + 1], m[4 * q + 2], m[4 * q + 3]);
            pl_st16(oc, Ps, p, m);
        },
        [&](int p, int c, float m) {
            Rf[((size_t)(c >> 2) * kPmax + p) * 4 + (c & 3)] = m;
            oc[((size_t)(c >> 2) * Ps + p) * 4 + (c & 3)] = m;
        });
    if (sg.prev && tid == 0) xwait(sg.fi + kXFlagInter + blk, sg.err, xcode(kXFlagInter + blk));    // the previous segment's inter-frame GRU state
    __syncthreads();
    for (int i = tid; i < kFw * kCh; i += kFusedThreads) {   // phase B is done with the intra pair: the inter pair takes its place (visible after phase C's barrier)
        lnt[i] = w.inter_ln_w[i];
        lnt[kFw * kCh + i] = w.inter_ln_b[i];
    }
    ADE_CLK(18);
#if ADE_DP_INTER_MFMA
    // ---- phase C: inter GRNN (GRU(8->8) x 2 groups along T, per column), in place in R.                          (:450-455,478-479)
    //      BATCHED over columns on the matrix cores, in the form of phase A: a tile = 16 columns of one group (the 33 columns make two full tiles and one with
    //      column 32 alone, per group: six tiles, dealt to the wavefronts round robin -- a wavefront's tiles share its group, i.e. its weights); the eight units
    //      are two row blocks of (unit, gate) rows, K = 8 input channels + 8 hidden units = four products per row block and step, the two input ones issued a step
    //      ahead.  Lane (g, j) owns units g and 4 + g of column j: its two hidden values are its B operands of the next step, and the positions it reads (channels
    //      g, 4 + g of the group at (t, column j)) are the two it overwrites.
    {
        constexpr float kS = -kLog2e, kN = 2.0f * kLog2e;
        constexpr int kWaves = kFusedThreads / 64, kTilesC = 6;
        static_assert(kWaves % 2 == 0, "a wavefront's tiles share its group");
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int g = (tid >> 4) & 3, j = tid & 15;
        const int grp = wave & 1;
        float ax[2][2], ah[2][2];
        v4f cb[2];
#pragma unroll
        for (int U = 0; U < 2; ++U) {
            const int c = j & 3;
            const float* pk = w.inter_gru + (grp * 8 + 4 * U + (j >> 2)) * 54;      // [W_ih 3x8 | W_hh 3x8 | b_ih 3 | b_hh 3] of output row (unit 4 U + j / 4)
            const float sc_ = c < 2 ? kS : kN;
            const int gi = c == 3 ? 2 : c;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                ax[U][kk] = c == 2 ? 0.0f : pk[gi * 8 + 4 * kk + g] * sc_;
                ah[U][kk] = c == 3 ? 0.0f : pk[24 + c * 8 + 4 * kk + g] * sc_;
            }
            const float* pb = w.inter_gru + (grp * 8 + 4 * U + g) * 54;
            cb[U][0] = (pb[48] + pb[51]) * kS;
            cb[U][1] = (pb[49] + pb[52]) * kS;
            cb[U][2] = pb[53] * kN;
            cb[U][3] = pb[50] * kN;
        }
        auto run = [&](auto nt_c) {
            constexpr int NT = decltype(nt_c)::value;                // tiles of this wavefront
            float h[NT][2];
            bool live[NT];
            float* xp[NT];                                          // channel 8 grp + g of (frame 0, the lane's column); + 4 kPmax floats: channel 8 grp + 4 + g
            float* xs[NT];
            v4f dx[NT][2];
#pragma unroll
            for (int c = 0; c < NT; ++c) {
                const int tile = wave + c * kWaves, f = (tile >> 1) * 16 + j;
                live[c] = f < kFw;                                   // (a lane whose column does not exist recomputes the last one and drops it)
                const int fc = live[c] ? f : kFw - 1;
                xp[c] = Rf + ((size_t)(grp * 2) * kPmax + fc) * 4 + g;
                xs[c] = sg.xi + kXInterOff + blk * (kFw * 16) + fc * 16 + grp;     // hand-off layout: [column][2 unit + group]
                h[c][0] = h[c][1] = 0.0f;
                if (sg.prev) {                                       // the recurrence continues from the previous segment's last frame
                    h[c][0] = xld1(xs[c] + 2 * g);
                    h[c][1] = xld1(xs[c] + 2 * (4 + g));
                }
                const float x0 = xp[c][0], x1 = xp[c][kPmax * 4];
#pragma unroll
                for (int U = 0; U < 2; ++U) {
                    dx[c][U] = mfma16x16x4(ax[U][0], x0, cb[U]);
                    dx[c][U] = mfma16x16x4(ax[U][1], x1, dx[c][U]);
                }
            }
            __builtin_amdgcn_s_setprio(3);
            for (int t = 0; t < T; ++t) {
                const int tn = t + 1 < T ? 1 : 0;
#pragma unroll
                for (int c = 0; c < NT; ++c) {
                    v4f d[2];
#pragma unroll
                    for (int U = 0; U < 2; ++U) {
                        d[U] = mfma16x16x4(ah[U][0], h[c][0], dx[c][U]);
                        d[U] = mfma16x16x4(ah[U][1], h[c][1], d[U]);
                    }
                    float* const xn = xp[c] + tn * (kFw * 4);       // next step's input (the last step re-reads its own: dropped)
                    const float x0 = xn[0], x1 = xn[kPmax * 4];
#pragma unroll
                    for (int U = 0; U < 2; ++U) {
                        dx[c][U] = mfma16x16x4(ax[U][0], x0, cb[U]);
                        dx[c][U] = mfma16x16x4(ax[U][1], x1, dx[c][U]);
                    }
                    h[c][0] = gru_gates(d[0], h[c][0]);
                    h[c][1] = gru_gates(d[1], h[c][1]);
                    if (live[c]) {
                        xp[c][0] = h[c][0];
                        xp[c][kPmax * 4] = h[c][1];
                    }
                    xp[c] = xn;
                }
            }
            set_prio(sg.base_prio);
            if (sg.next) {
#pragma unroll
                for (int c = 0; c < NT; ++c) {
                    float* const xo = sg.xo + (xs[c] - sg.xi);
                    if (live[c]) {
                        xst1(xo + 2 * g, h[c][0]);
                        xst1(xo + 2 * (4 + g), h[c][1]);
                    }
                }
                xdrain();
            }
        };
        const int nt = wave < kTilesC ? (kTilesC - 1 - wave) / kWaves + 1 : 0;      // (wave-uniform)
        static_assert((kTilesC + kWaves - 1) / kWaves <= 2, "at most two tiles per wavefront");
        if (nt == 2) run(std::integral_constant<int, 2>{});
        else if (nt == 1) run(std::integral_constant<int, 1>{});
    }
#else
    // ---- phase C: inter GRNN.  16 lanes per F column: lane = 2*unit + group; GRU(8->8) along T, in place in R
    //      (a column position is read, then overwritten, by its own 16 lanes only).              (:450-455,478-479)
    //      The loop is VALU-issue-bound (9 wavefronts on 4 SIMDs), so it is written for instruction count: r|z gates as
    //      one packed accumulator, the n gate packed over input pairs, and the 8 hidden values of a lane's own group
    //      fetched with 7 row rotations by 2s (group in the lane's low bit => a rotation by 2s stays inside the group);
    //      each lane keeps its recurrent weights pre-rotated to match.
    //      One pass: a 16-lane row walks SEVERAL columns (row, row + 16, ..) through the same step loop -- independent recurrences interleaved instruction by instruction, so
    //      one column's exp / rcp / rotation latencies are another's issue slots.  (Round 3 walked the columns in three passes of eleven: three dependent loops of T steps
    //      where this is one, and the recurrence is what a chunk's four segments wait for.)  The weights depend on the lane's role q only: shared by its columns.
    {
        constexpr int kRows = G::kThreads / 16;                      // 16-lane rows of the workgroup
        const int row = tid >> 4, q = tid & 15;
        const int grp = q & 1, unit = q >> 1;
        const float* pk = w.inter_gru + (grp * 8 + unit) * 54;      // [W_ih 3x8 | W_hh 3x8 | b_ih 3 | b_hh 3] of this output row
        int ks[8];                                                   // ks[s] = hidden index delivered by rotation s (measured, so the
        ks[0] = unit;                                                // rotation direction convention cannot matter)
#define ADE_KS(S) ks[S] = ((int)row_ror<2 * S>((float)q)) >> 1;
        ADE_KS(1) ADE_KS(2) ADE_KS(3) ADE_KS(4) ADE_KS(5) ADE_KS(6) ADE_KS(7)
#undef ADE_KS
        constexpr float kS = -kLog2e, kN = 2.0f * kLog2e;        // pre-scaling for exp2-based activations (see the intra GRU)
        v2f wi_rz[8], wh_rz[8], wi_n[4], wh_n[4];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            wi_rz[k] = mk2(pk[k] * kS, pk[8 + k] * kS);
            wh_rz[k] = mk2(pk[24 + ks[k]] * kS, pk[32 + ks[k]] * kS);
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            wi_n[m] = mk2(pk[16 + 2 * m] * kN, pk[16 + 2 * m + 1] * kN);
            wh_n[m] = mk2(pk[40 + ks[2 * m]] * kN, pk[40 + ks[2 * m + 1]] * kN);
        }
        const v2f b_rz = mk2((pk[48] + pk[51]) * kS, (pk[49] + pk[52]) * kS);
        const float bi_n = pk[50] * kN, bh_n = pk[53] * kN;
        auto run = [&](auto nc_c) {
            constexpr int NC = decltype(nc_c)::value;
            float h[NC];
            float4 xa[NC], xb[NC];
            int fc_[NC];
            bool live[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int f = row + kRows * c;
                live[c] = f < kFw;                                   // (a lane whose column does not exist recomputes the last one and drops it)
                fc_[c] = live[c] ? f : kFw - 1;
                h[c] = 0.0f;
                if (sg.prev) h[c] = xld1(sg.xi + kXInterOff + blk * (kFw * 16) + fc_[c] * 16 + q);      // the recurrence continues from the previous segment's last frame
                xa[c] = R[(grp * 2) * kPmax + fc_[c]];
                xb[c] = R[(grp * 2 + 1) * kPmax + fc_[c]];
            }
            __builtin_amdgcn_s_setprio(3);
            for (int t = 0; t < T; ++t) {
                const int tn = t + 1 < T ? t + 1 : t;
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const int p = t * kFw + fc_[c], pn = tn * kFw + fc_[c];
                    const float xv[8] = {xa[c].x, xa[c].y, xa[c].z, xa[c].w, xb[c].x, xb[c].y, xb[c].z, xb[c].w};
                    xa[c] = R[(grp * 2) * kPmax + pn];          // next step's input: issued now, needed after this step's write
                    xb[c] = R[(grp * 2 + 1) * kPmax + pn];
                    v2f a_rz = b_rz, a_n = mk2(bi_n, 0.0f);
#pragma unroll
                    for (int k = 0; k < 8; ++k) a_rz += wi_rz[k] * xv[k];
#pragma unroll
                    for (int m = 0; m < 4; ++m) a_n += wi_n[m] * mk2(xv[2 * m], xv[2 * m + 1]);
                    float hs[8];
                    hs[0] = h[c];
#define ADE_HS(S) hs[S] = row_ror<2 * S>(h[c]);
                    ADE_HS(1) ADE_HS(2) ADE_HS(3) ADE_HS(4) ADE_HS(5) ADE_HS(6) ADE_HS(7)
#undef ADE_HS
                    v2f c0 = a_rz, c1 = mk2(0.0f, 0.0f), c_n = mk2(bh_n, 0.0f);
#pragma unroll
                    for (int k = 0; k < 8; k += 2) { c0 += wh_rz[k] * hs[k]; c1 += wh_rz[k + 1] * hs[k + 1]; }
#pragma unroll
                    for (int m = 0; m < 4; ++m) c_n += wh_n[m] * mk2(hs[2 * m], hs[2 * m + 1]);
                    const v2f rz = c0 + c1;
                    const float r = fast_rcp(1.0f + __builtin_amdgcn_exp2f(rz[0]));
                    const float z = fast_rcp(1.0f + __builtin_amdgcn_exp2f(rz[1]));
                    const float n = 1.0f - 2.0f * fast_rcp(__builtin_amdgcn_exp2f((a_n[0] + a_n[1]) + r * (c_n[0] + c_n[1])) + 1.0f);
                    h[c] = n + z * (h[c] - n);
                    if (live[c]) Rf[((size_t)(grp * 2 + (unit >> 2)) * kPmax + p) * 4 + (unit & 3)] = h[c];
                }
            }
            set_prio(sg.base_prio);
            if (sg.next) {
#pragma unroll
                for (int c = 0; c < NC; ++c)
                    if (live[c]) xst1(sg.xo + kXInterOff + blk * (kFw * 16) + fc_[c] * 16 + q, h[c]);
                xdrain();
            }
        };
        // columns of this WAVEFRONT's first row decide how many it walks (wave-uniform): rows row0 .. row0 + 3 have columns row + 16 c < 33
        const int row0 = (tid >> 6) * 4, nc = row0 < kFw ? (kFw - 1 - row0) / kRows + 1 : 0;
        static_assert((kFw + kRows - 1) / kRows <= 3, "at most three columns per 16-lane row");
        if (nc == 3) run(std::integral_constant<int, 3>{});
        else if (nc == 2) run(std::integral_constant<int, 2>{});
        else if (nc == 1) run(std::integral_constant<int, 1>{});
    }
#endif
    __syncthreads();
    if (sg.next && tid == 0) xflag_store(sg.fo + kXFlagInter + blk, 1u);
    ADE_CLK(19);
    // ---- phase D: inter Linear + LayerNorm + residual(mid) -> out
    {
        const float* nsc = (next_x1 && next_skip) ? next_skip + cbase : nullptr;
        // (the residual is mid, written by this same lane in phase B: L2-resident)
        fc_ln_rows<G>(R, w.inter_fc, w.inter_fc_b, lnt, lnt + kFw * kCh, T, P, tid, oc, Ps,
            [&](int p, const float (&o)[16]) {
                if (store_lo) {
                    pl_st16(oc, Ps, p, o);
                } else {          // (see gtblock_stage: the following block gets channels 0-7 through LDS)
                    st4(oc + ((size_t)2 * Ps + p) * 4, o + 8);
                    st4(oc + ((size_t)3 * Ps + p) * 4, o + 12);
                }
                if (next_full) {  // the following DPGRNN's input: the whole output, in place (the frame's row is done with these positions)
#pragma unroll
                    for (int q = 0; q < 4; ++q) R[q * kPmax + p] = make_float4(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]);
                }
                if (next_x1) {   // the following GTConvBlock's pointwise input (out + skip)[:, :8] -> LDS planes 2-3 (the frame's row is done with them)
                    float k8[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
                    if (nsc) pl_ld8(nsc, Ps, p, 0, k8);
                    R[2 * kPmax + p] = make_float4(o[0] + k8[0], o[1] + k8[1], o[2] + k8[2], o[3] + k8[3]);
                    R[3 * kPmax + p] = make_float4(o[4] + k8[4], o[5] + k8[5], o[6] + k8[6], o[7] + k8[7]);
                }
            },
            [&](int p, int c, float o) {
                if (store_lo || c >= 8) oc[((size_t)(c >> 2) * Ps + p) * 4 + (c & 3)] = o;
                if (next_full) Rf[((size_t)(c >> 2) * kPmax + p) * 4 + (c & 3)] = o;
                if (next_x1 && c < 8) {
                    const float k = nsc ? nsc[((size_t)(c >> 2) * Ps + p) * 4 + (c & 3)] : 0.0f;
                    Rf[((size_t)(2 + (c >> 2)) * kPmax + p) * 4 + (c & 3)] = o + k;
                }
            });
    }
    ADE_CLK(20);
}


}  // namespace stage
}  // namespace ade
