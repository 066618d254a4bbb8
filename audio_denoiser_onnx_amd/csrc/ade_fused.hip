// ade_fused.hip — kernels of the fused (per-chunk, LDS-resident) path.
//
// The stage bodies live in ade_stage_net.h (GTConvBlock, DPGRNN) and ade_stage_frontback.h (front, back).  They are
// exposed two ways:
//   * k_gtcrn_chunk: ONE launch for the whole network — a 1024-thread workgroup walks its chunk through
//     front -> 3 x GTConvBlock -> 2 x DPGRNN -> 3 x GTConvBlock -> back.  Chunks are independent, so there is no
//     inter-workgroup traffic and no grid barrier; inter-stage tensors (skip connections) go through HBM/L2 in the
//     quad-planar layout and are re-read by the same workgroup after a workgroup barrier.  At batch = 256 this is one
//     workgroup per CU of the MI355X for the whole forward.
//   * one kernel per stage (k_front / k_gtblock / k_dpgrnn / k_back): used for per-stage HIP-event timing and phase
//     clocks (ade_profile_last), and as the implementation the single launch is tested against.
#include "ade_stage_frontback.h"

namespace ade {

using namespace stage;

namespace {

constexpr size_t cmax(size_t a, size_t b) { return a > b ? a : b; }
constexpr size_t kChunkSmemBytes = cmax(cmax(kGtSmemBytes, kDpSmemBytes), cmax(kFrontSmemBytes, kBackSmemBytes));

__global__ __launch_bounds__(kFusedThreads) void k_front(const int16_t* __restrict__ pcm, int L, int T, FftTabs tabs, BandTab erb,
                                                         ConvW c0, ConvW c1, float* __restrict__ spec, float* __restrict__ e0,
                                                         float* __restrict__ e1, long long* __restrict__ clk,
                                                         const float* __restrict__ dc) {
    HIP_DYNAMIC_SHARED(float4, smem)
    front_stage(reinterpret_cast<float*>(smem), blockIdx.x, pcm, L, T, tabs, erb, c0, c1, spec, e0, e1, clk, dc);
}
__global__ __launch_bounds__(kFusedThreads) void k_gtblock(const float* __restrict__ a, const float* __restrict__ skip, GtConvW w,
                                                           float* __restrict__ out, int T, long long* __restrict__ clk) {
    HIP_DYNAMIC_SHARED(float4, smem)
    gtblock_stage(smem, blockIdx.x, a, skip, w, out, T, clk);
}
__global__ __launch_bounds__(kFusedThreads) void k_dpgrnn(const float* __restrict__ x, DpW w, float* __restrict__ out, int T,
                                                          long long* __restrict__ clk) {
    HIP_DYNAMIC_SHARED(float4, smem)
    dpgrnn_stage(smem, blockIdx.x, x, w, out, T, clk);
}
__global__ __launch_bounds__(kFusedThreads) void k_back(const float* __restrict__ x, const float* __restrict__ e1,
                                                        const float* __restrict__ e0, const float* __restrict__ spec, ConvW c3, ConvW c4,
                                                        BandTab bs, FftTabs tabs, float* __restrict__ d3, float* __restrict__ mask,
                                                        int16_t* __restrict__ pcm, float* __restrict__ f32, int T,
                                                        long long* __restrict__ clk) {
    HIP_DYNAMIC_SHARED(float4, smem)
    back_stage(reinterpret_cast<float*>(smem), blockIdx.x, x, e1, e0, spec, c3, c4, bs, tabs, d3, mask, pcm, f32, T, clk);
}

// kClk = false is the shipped kernel (no phase-clock code at all); kClk = true is the same kernel with thread 0 of
// workgroup 0 stamping phase clocks into A.clk, 64 slots per stage: [front | enc 0-2 | dp 0-1 | dec 0-2 | back].
template <bool kClk>
__global__ __launch_bounds__(kFusedThreads) void k_gtcrn_chunk(ChunkArgs A) {
    HIP_DYNAMIC_SHARED(float4, smem)
    long long* const clk0 = kClk ? A.clk : nullptr;
    const int chunk = blockIdx.x;
    float* fsm = reinterpret_cast<float*>(smem);
    // De-phase the workgroups.  Every stage begins and ends with an HBM burst (its inputs / skip tensors in, its output out) and at 256
    // chunks all 256 workgroups -- one per CU -- would issue the same burst at the same instant: measured, the stages run 28 % slower at
    // 256 chunks than at 3 (tools/phase_clock.py).  Holding every other group of 8 workgroups back by about one GTConvBlock (27 us) makes
    // one half's bursts land in the other half's compute phases: -3.5 % per step, the delay included (tools/stagger_probe.py).
    if (A.stagger > 0 && ((chunk >> 3) & 1)) {
        const long long t0 = wall_clock64();                 // 100 MHz, independent of the shader clock
        while (wall_clock64() - t0 < A.stagger) __builtin_amdgcn_s_sleep(8);
    }
    front_stage(fsm, chunk, A.pcm_in, A.L, A.T, A.tabs, A.erb_bm, A.en0, A.en1, A.spec, A.e0, A.e1, clk0, A.dc);
    __syncthreads();
    const float* x = A.e1;
#pragma unroll 1
    for (int i = 0; i < 3; ++i) {       // encoder GTConvBlocks
        gtblock_stage(smem, chunk, x, nullptr, A.en_gt[i], A.xe[i], A.T, kClk ? clk0 + 64 * (1 + i) : nullptr, /*x1_in_lds=*/i > 0,
                      /*next_x1=*/i < 2, nullptr);
        __syncthreads();
        x = A.xe[i];
    }
#pragma unroll 1
    for (int i = 0; i < 2; ++i) {
        dpgrnn_stage(smem, chunk, x, A.dp[i], A.dpo[i], A.T, kClk ? clk0 + 64 * (4 + i) : nullptr, /*next_x1=*/i == 1, /*next_skip=*/A.xe[2]);
        __syncthreads();
        x = A.dpo[i];
    }
#pragma unroll 1
    for (int i = 0; i < 3; ++i) {       // decoder GTConvBlocks on x + en_outs[4 - i]
        gtblock_stage(smem, chunk, x, A.xe[2 - i], A.de_gt[i], A.xd[i], A.T, kClk ? clk0 + 64 * (6 + i) : nullptr, /*x1_in_lds=*/true, /*next_x1=*/i < 2,
                      /*next_skip=*/i < 2 ? A.xe[1 - i] : nullptr);
        __syncthreads();
        x = A.xd[i];
    }
    back_stage(fsm, chunk, x, A.e1, A.e0, A.spec, A.de3, A.de4, A.erb_bs, A.tabs, A.d3, A.mask, A.pcm_out, A.f32_out, A.T,
               kClk ? clk0 + 64 * 9 : nullptr);
}

}  // namespace

bool fused_supported(int T) { return T >= 2 && T <= kTmaxFused; }

hipError_t fused_init() {
    const void* fns[6] = {reinterpret_cast<const void*>(&k_front), reinterpret_cast<const void*>(&k_gtblock),
                          reinterpret_cast<const void*>(&k_dpgrnn), reinterpret_cast<const void*>(&k_back),
                          reinterpret_cast<const void*>(&k_gtcrn_chunk<false>), reinterpret_cast<const void*>(&k_gtcrn_chunk<true>)};
    const size_t bytes[6] = {kFrontSmemBytes, kGtSmemBytes, kDpSmemBytes, kBackSmemBytes, kChunkSmemBytes, kChunkSmemBytes};
    for (int i = 0; i < 6; ++i) {
        hipError_t e = hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes[i]);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

void launch_gtblock(hipStream_t s, const float* a, const float* skip, GtConvW w, float* out, int B, int T, long long* clk) {
    hipLaunchKernelGGL(k_gtblock, dim3(B), dim3(kFusedThreads), kGtSmemBytes, s, a, skip, w, out, T, clk);
}
void launch_dpgrnn(hipStream_t s, const float* x, DpW w, float* out, int B, int T, long long* clk) {
    hipLaunchKernelGGL(k_dpgrnn, dim3(B), dim3(kFusedThreads), kDpSmemBytes, s, x, w, out, T, clk);
}
void launch_front(hipStream_t s, const int16_t* pcm, int B, int L, int T, FftTabs tabs, BandTab erb_bm, ConvW c0, ConvW c1, float* spec,
                  float* e0, float* e1, long long* clk, const float* dc) {
    hipLaunchKernelGGL(k_front, dim3(B), dim3(kFusedThreads), kFrontSmemBytes, s, pcm, L, T, tabs, erb_bm, c0, c1, spec, e0, e1, clk, dc);
}
void launch_back(hipStream_t s, const float* x, const float* e1, const float* e0, const float* spec, ConvW c3, ConvW c4, BandTab erb_bs,
                 FftTabs tabs, float* d3, float* mask, int16_t* pcm, float* f32, int B, int T, long long* clk) {
    hipLaunchKernelGGL(k_back, dim3(B), dim3(kFusedThreads), kBackSmemBytes, s, x, e1, e0, spec, c3, c4, erb_bs, tabs, d3, mask, pcm, f32, T,
                       clk);
}
void launch_gtcrn_chunk(hipStream_t s, const ChunkArgs& args, int B) {
    if (args.clk) hipLaunchKernelGGL(k_gtcrn_chunk<true>, dim3(B), dim3(kFusedThreads), kChunkSmemBytes, s, args);
    else hipLaunchKernelGGL(k_gtcrn_chunk<false>, dim3(B), dim3(kFusedThreads), kChunkSmemBytes, s, args);
}

}  // namespace ade
