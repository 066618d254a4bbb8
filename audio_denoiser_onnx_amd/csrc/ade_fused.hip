// ade_fused.hip — kernels of the fused (per-chunk, LDS-resident) path.
//
// The stage bodies live in ade_stage_net.h (GTConvBlock, DPGRNN) and ade_stage_frontback.h (front, back), templated on the workgroup
// GEOMETRY (ade_internal.h): geometry 0 = 1024 threads owning up to 64 frames (one workgroup per CU), geometry 1 = 512 threads owning up
// to 32 frames (two workgroups per CU).  A chunk with more frames than one workgroup owns is split into SEGMENTS of consecutive frames,
// one workgroup each; segment k + 1 continues the recurrences of segment k through the exchange area (ade_internal.h, dev::xwait).
// Segments are numbered so that all first segments come first in the grid: a segment only ever waits for a workgroup with a LOWER block
// index, and its waits are bounded (dev::xwait) -- nothing depends on the dispatch order for correctness of the protocol's termination.
// The bodies are exposed two ways:
//   * k_gtcrn_chunk: ONE launch for the whole network — each workgroup walks its segment through
//     front -> 3 x GTConvBlock -> 2 x DPGRNN -> 3 x GTConvBlock -> back.  Inter-stage tensors (skip connections) go through HBM/L2 in the
//     quad-planar layout and are re-read by the same workgroup after a workgroup barrier.
//   * one kernel per stage (k_front / k_gtblock / k_dpgrnn / k_back): used for per-stage HIP-event timing and phase
//     clocks (ade_profile_last), and as the implementation the single launch is tested against.
#include "ade_stage_frontback.h"

namespace ade {

using namespace stage;

namespace {

constexpr size_t cmax(size_t a, size_t b) { return a > b ? a : b; }
template <class G> constexpr size_t chunk_smem_bytes() {
    return cmax(cmax(gt_smem_bytes<G>(), dp_smem_bytes<G>()), cmax(front_smem_bytes<G>(), back_smem_bytes<G>()));
}
static_assert(chunk_smem_bytes<Geo1>() * 2 <= 160 * 1024, "two geometry-1 workgroups must fit one CU's LDS");
static_assert(chunk_smem_bytes<Geo2>() * 4 <= 160 * 1024, "four geometry-2 workgroups must fit one CU's LDS");
static_assert(chunk_smem_bytes<Geo0>() <= 160 * 1024, "geometry 0 must fit one CU's LDS");

// Block b of a (B x nseg)-block grid -> (chunk, segment): all first segments first.  Frames are dealt as evenly as possible, the earlier
// segments taking the extra frame (the last segment finishes last; it should not also be the longest).
// chunk0: the launch covers chunks chunk0 .. chunk0 + B - 1 of the engine's batch arrays (ade_process cuts a host batch into sub-batches on separate streams).
__device__ __forceinline__ Seg make_seg(const SegPlan& plan, int B, int T, int block, int& chunk, int chunk0 = 0) {
    const int seg = block / B;
    chunk = chunk0 + block - seg * B;
    const int base = T / plan.nseg, rem = T - base * plan.nseg;
    Seg sg;
    sg.T = T;
    sg.nT = base + (seg < rem ? 1 : 0);
    sg.t0 = seg * base + (seg < rem ? seg : rem);
    sg.prev = seg > 0;
    sg.next = seg + 1 < plan.nseg;
    sg.swap = plan.wave_swap && (seg & 1);
    sg.stream = plan.stream;
    const size_t slot = (size_t)chunk * plan.nseg + seg;
    sg.xo = plan.xchg + slot * kXFloats;
    sg.xi = plan.xchg + (slot - (seg > 0 ? 1 : 0)) * kXFloats;
    sg.fo = plan.flags + slot * kXFlags;
    sg.fi = plan.flags + (slot - (seg > 0 ? 1 : 0)) * kXFlags;
    sg.err = plan.err;
    if (seg == 0 && plan.carry_in) {                 // a stream's push: continue from what the previous push's last segment left
        sg.prev = 1;
        sg.xi = plan.carry_in + (size_t)chunk * kXFloats;
        sg.fi = plan.carry_in_flags + (size_t)chunk * kXFlags;
    }
    if (seg + 1 == plan.nseg && plan.carry_out) {
        sg.next = 1;
        sg.xo = plan.carry_out + (size_t)chunk * kXFloats;
        sg.fo = plan.carry_out_flags + (size_t)chunk * kXFlags;
    }
    if (plan.withhold && block == 0) sg.fo = plan.flags + ((size_t)B * plan.nseg - 1) * kXFlags;   // test hook: see SegPlan::withhold
    sg.first = !sg.prev;
    sg.base_prio = plan.prio == 4 ? (seg < 2 ? 2 - seg : 0) : (seg > 0 ? plan.prio : 0);
    return sg;
}
__device__ __forceinline__ long long* seg_clk(long long* clk, const Seg& sg, int B, int block) {
    return clk ? clk + (size_t)(block / B) * kClkSlotsPerSeg : nullptr;
}

template <class G>
__global__ __launch_bounds__(G::kThreads, G::kWavesPerSimd) void k_front(SegPlan plan, const int16_t* __restrict__ pcm, int B, int L, int T, FftTabs tabs,
                                                                          BandTab erb, ConvW c0, ConvW c1, float* __restrict__ spec, float* __restrict__ e0,
                                                                          float* __restrict__ e1, long long* __restrict__ clk, const float* __restrict__ dc) {
    HIP_DYNAMIC_SHARED(float4, smem)
    int chunk;
    const Seg sg = make_seg(plan, B, T, blockIdx.x, chunk);
    front_stage<G>(reinterpret_cast<float*>(smem), chunk, sg, pcm, L, tabs, erb, c0, c1, spec, e0, e1, seg_clk(clk, sg, B, blockIdx.x), dc);
}
template <class G>
__global__ __launch_bounds__(G::kThreads, G::kWavesPerSimd) void k_gtblock(SegPlan plan, int blk, const float* __restrict__ a, const float* __restrict__ skip,
                                                                            GtConvW w, float* __restrict__ out, int B, int T, long long* __restrict__ clk) {
    HIP_DYNAMIC_SHARED(float4, smem)
    int chunk;
    const Seg sg = make_seg(plan, B, T, blockIdx.x, chunk);
    gtblock_stage<G>(smem, chunk, sg, blk, a, skip, w, out, seg_clk(clk, sg, B, blockIdx.x));
}
template <class G>
__global__ __launch_bounds__(G::kThreads, G::kWavesPerSimd) void k_dpgrnn(SegPlan plan, int blk, const float* __restrict__ x, DpW w, float* __restrict__ out,
                                                                           int B, int T, long long* __restrict__ clk) {
    HIP_DYNAMIC_SHARED(float4, smem)
    int chunk;
    const Seg sg = make_seg(plan, B, T, blockIdx.x, chunk);
    dpgrnn_stage<G>(smem, chunk, sg, blk, x, w, out, seg_clk(clk, sg, B, blockIdx.x));
}
template <class G>
__global__ __launch_bounds__(G::kThreads, G::kWavesPerSimd) void k_back(SegPlan plan, const float* __restrict__ x, const float* __restrict__ e1,
                                                                         const float* __restrict__ e0, const float* __restrict__ spec, ConvW c3, ConvW c4,
                                                                         BandTab bs, FftTabs tabs, int16_t* __restrict__ pcm, float* __restrict__ f32, int B,
                                                                         int T, long long* __restrict__ clk) {
    HIP_DYNAMIC_SHARED(float4, smem)
    int chunk;
    const Seg sg = make_seg(plan, B, T, blockIdx.x, chunk);
    back_stage<G>(reinterpret_cast<float*>(smem), chunk, sg, x, e1, e0, spec, c3, c4, bs, tabs, pcm, f32, seg_clk(clk, sg, B, blockIdx.x));
}

// kClk = false is the shipped kernel (no phase-clock code at all); kClk = true is the DEBUG build of the same kernel: thread 0 of each segment's first
// workgroup stamps phase clocks into C.clk (when given), 64 slots per stage: [front | enc 0-2 | dp 0-1 | dec 0-2 | back], one such set per segment, and every
// inter-stage tensor is stored whole (option "full_taps": the shipped kernel keeps channels 0-7 of x_d0 / x_d1 / dp2 in LDS -- store_lo in ade_stage_net.h).
// The per-engine arguments (weights, workspace) are read from device memory stage by stage: ChunkFixed in ade_internal.h.
template <class G, bool kClk>
__global__ __launch_bounds__(G::kThreads, G::kWavesPerSimd) void k_gtcrn_chunk(ChunkCall C_) {
    HIP_DYNAMIC_SHARED(float4, smem)
    float* fsm = reinterpret_cast<float*>(smem);
    typedef const ChunkFixed ADE_CONSTANT_AS* fixed_ptr;
    typedef const ChunkCall ADE_CONSTANT_AS* call_ptr;
    // Nothing but the block index stays live from one stage to the next: every stage re-reads what it needs of the call (kernel-argument
    // segment) and of the per-engine block (device memory) through the scalar cache and re-derives its segment.  Held in registers across
    // the whole launch, those ~60 scalars were parked in VGPR lanes at every stage boundary.
#if defined(__AMDGCN__)
    call_ptr C = (call_ptr)__builtin_amdgcn_kernarg_segment_ptr();
#else
    call_ptr C = (call_ptr)&C_;
#endif
    int block = blockIdx.x;
#define ADE_STAGE_ENTRY()                                                                                  \
    ADE_KEEP_IN_LOOP(C);                                                                                   \
    ADE_KEEP_IN_LOOP(block);                                                                               \
    int chunk;                                                                                             \
    const Seg sg = make_seg(cload<SegPlan>(&C->plan), C->B, C->T, block, chunk, C->chunk0);                \
    fixed_ptr F = (fixed_ptr)C->fixed;                                                                     \
    long long* const clk0 = kClk ? seg_clk(C->clk, sg, C->B, block) : nullptr;                             \
    (void)clk0
    {
        ADE_STAGE_ENTRY();
        // A later segment is the younger workgroup on its CU and is served last by the oldest-first instruction arbitration; option
        // "seg_prio" raises its wave priority (measured: it only swaps which of the two workgroups of a CU is held back).
        set_prio(sg.base_prio);
        // De-phase the workgroups (geometry 0: one workgroup per CU).  Every stage begins and ends with an HBM burst (its inputs / skip
        // tensors in, its output out) and at 256 chunks all 256 workgroups would issue the same burst at the same instant: measured, the
        // stages run 28 % slower at 256 chunks than at 3 (tools/phase_clock.py).  Holding every other group of 8 workgroups back by about one
        // GTConvBlock makes one half's bursts land in the other half's compute phases (tools/stagger_probe.py).  With two workgroups per
        // CU (geometry 1) the segments of a chunk are out of phase by construction.
        const int stagger = C->stagger;
        if (stagger > 0 && ((chunk >> 3) & 1)) {
            const long long t0 = wall_clock64();                 // 100 MHz, independent of the shader clock
            while (wall_clock64() - t0 < stagger) __builtin_amdgcn_s_sleep(8);
        }
        if (C->in_ready) {       // a host batch streams through this launch: this chunk's PCM may still be on its way (ChunkCall::in_ready)
            if (threadIdx.x == 0) wait_rows_in(C->in_ready + (size_t)((chunk - C->chunk0) / C->group_rows) * kReadyStride, C->epoch, sg.err, xcode(15));
            __syncthreads();
        }
        const FftTabs tabs = cload<FftTabs>(&F->tabs);
        const BandTab erb = cload<BandTab>(&F->erb_bm);
        const ConvW c0 = cload<ConvW>(&F->en0), c1 = cload<ConvW>(&F->en1);
        front_stage<G>(fsm, chunk, sg, C->pcm_in, C->L, tabs, erb, c0, c1, F->spec, F->e0, F->e1, clk0, C->dc);
    }
    __syncthreads();
#pragma unroll 1
    for (int i = 0; i < 3; ++i) {       // encoder GTConvBlocks
        ADE_STAGE_ENTRY();
        const GtConvW w = cload<GtConvW>(&F->en_gt[i]);
        const float* const x = i == 0 ? F->e1 : F->xe[i > 0 ? i - 1 : 0];
        gtblock_stage<G>(smem, chunk, sg, i, x, nullptr, w, F->xe[i], (kClk && clk0) ? clk0 + 64 * (1 + i) : nullptr, /*x1_in_lds=*/i > 0, /*next_x1=*/i < 2, nullptr,
                         /*store_lo=*/true, /*next_full=*/i == 2);
        __syncthreads();
    }
#pragma unroll 1
    for (int i = 0; i < 2; ++i) {
        ADE_STAGE_ENTRY();
        const DpW w = cload<DpW>(&F->dp[i]);
        const float* const x = i == 0 ? F->xe[2] : F->dpo[0];
        dpgrnn_stage<G>(smem, chunk, sg, i, x, w, F->dpo[i], (kClk && clk0) ? clk0 + 64 * (4 + i) : nullptr, /*next_x1=*/i == 1, /*next_skip=*/F->xe[2],
                        /*store_lo=*/i == 0 || kClk, /*x_in_lds=*/true, /*next_full=*/i == 0);
        __syncthreads();
    }
#pragma unroll 1
    for (int i = 0; i < 3; ++i) {       // decoder GTConvBlocks on x + en_outs[4 - i]
        ADE_STAGE_ENTRY();
        const GtConvW w = cload<GtConvW>(&F->de_gt[i]);
        const float* const x = i == 0 ? F->dpo[1] : F->xd[i > 0 ? i - 1 : 0];
        gtblock_stage<G>(smem, chunk, sg, 3 + i, x, F->xe[2 - i], w, F->xd[i], (kClk && clk0) ? clk0 + 64 * (6 + i) : nullptr, /*x1_in_lds=*/true, /*next_x1=*/i < 2,
                         /*next_skip=*/i < 2 ? F->xe[1 - i] : nullptr, /*store_lo=*/i == 2 || kClk);
        __syncthreads();
    }
    {
        ADE_STAGE_ENTRY();
        const FftTabs tabs = cload<FftTabs>(&F->tabs);
        const BandTab erb = cload<BandTab>(&F->erb_bs);
        const ConvW c3 = cload<ConvW>(&F->de3), c4 = cload<ConvW>(&F->de4);
        back_stage<G>(fsm, chunk, sg, F->xd[2], F->e1, F->e0, F->spec, c3, c4, erb, tabs, C->pcm_out, C->f32_out, (kClk && clk0) ? clk0 + 64 * 9 : nullptr);
        if (C->in_ready) {       // this segment's share of the chunk's output is stored: count the workgroup in (ChunkCall::out_done)
            xdrain();
            __syncthreads();
            if (threadIdx.x == 0) {
                const int g = (chunk - C->chunk0) / C->group_rows, rows = C->B - g * C->group_rows < C->group_rows ? C->B - g * C->group_rows : C->group_rows;
                signal_rows_out(C->out_count + g, C->out_done + g, C->epoch, (unsigned)(rows * (int)(gridDim.x / (unsigned)C->B)));
            }
        }
    }
#undef ADE_STAGE_ENTRY
}

template <class G>
hipError_t init_geometry() {
    const void* fns[6] = {reinterpret_cast<const void*>(&k_front<G>),         reinterpret_cast<const void*>(&k_gtblock<G>),
                          reinterpret_cast<const void*>(&k_dpgrnn<G>),        reinterpret_cast<const void*>(&k_back<G>),
                          reinterpret_cast<const void*>(&k_gtcrn_chunk<G, false>), reinterpret_cast<const void*>(&k_gtcrn_chunk<G, true>)};
    const size_t bytes[6] = {front_smem_bytes<G>(), gt_smem_bytes<G>(), dp_smem_bytes<G>(), back_smem_bytes<G>(), chunk_smem_bytes<G>(), chunk_smem_bytes<G>()};
    for (int i = 0; i < 6; ++i) {
        hipError_t e = hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes[i]);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace

int fused_geometries() { return 3; }
int fused_max_frames(int geometry) { return geometry == 2 ? Geo2::kTmax : (geometry == 1 ? Geo1::kTmax : Geo0::kTmax); }
int fused_segments(int T, int geometry) { const int m = fused_max_frames(geometry); return (T + m - 1) / m; }
bool fused_supported(int T, int geometry) {
    if (T < 2 || geometry < 0 || geometry > 2) return false;
    return fused_segments(T, geometry) <= kMaxSegments;      // (segments of any length: a short one passes its predecessor's history on)
}

hipError_t fused_init() {
    hipError_t e = init_geometry<Geo0>();
    if (e == hipSuccess) e = init_geometry<Geo1>();
    return e != hipSuccess ? e : init_geometry<Geo2>();
}

#define ADE_GEO_LAUNCH(kernel_tpl, smem_fn, grid, ...)                                                                              \
    do {                                                                                                                            \
        if (geometry == 2) hipLaunchKernelGGL(kernel_tpl<Geo2>, dim3(grid), dim3(Geo2::kThreads), smem_fn<Geo2>(), s, __VA_ARGS__); \
        else if (geometry == 1) hipLaunchKernelGGL(kernel_tpl<Geo1>, dim3(grid), dim3(Geo1::kThreads), smem_fn<Geo1>(), s, __VA_ARGS__); \
        else hipLaunchKernelGGL(kernel_tpl<Geo0>, dim3(grid), dim3(Geo0::kThreads), smem_fn<Geo0>(), s, __VA_ARGS__);               \
    } while (0)

void launch_gtblock(hipStream_t s, int geometry, SegPlan plan, int blk, const float* a, const float* skip, GtConvW w, float* out, int B, int T, long long* clk) {
    ADE_GEO_LAUNCH(k_gtblock, gt_smem_bytes, B * plan.nseg, plan, blk, a, skip, w, out, B, T, clk);
}
void launch_dpgrnn(hipStream_t s, int geometry, SegPlan plan, int blk, const float* x, DpW w, float* out, int B, int T, long long* clk) {
    ADE_GEO_LAUNCH(k_dpgrnn, dp_smem_bytes, B * plan.nseg, plan, blk, x, w, out, B, T, clk);
}
void launch_front(hipStream_t s, int geometry, SegPlan plan, const int16_t* pcm, int B, int L, int T, FftTabs tabs, BandTab erb_bm, ConvW c0, ConvW c1, float* spec,
                  float* e0, float* e1, long long* clk, const float* dc) {
    ADE_GEO_LAUNCH(k_front, front_smem_bytes, B * plan.nseg, plan, pcm, B, L, T, tabs, erb_bm, c0, c1, spec, e0, e1, clk, dc);
}
void launch_back(hipStream_t s, int geometry, SegPlan plan, const float* x, const float* e1, const float* e0, const float* spec, ConvW c3, ConvW c4, BandTab erb_bs,
                 FftTabs tabs, int16_t* pcm, float* f32, int B, int T, long long* clk) {
    ADE_GEO_LAUNCH(k_back, back_smem_bytes, B * plan.nseg, plan, x, e1, e0, spec, c3, c4, erb_bs, tabs, pcm, f32, B, T, clk);
}
void launch_gtcrn_chunk(hipStream_t s, int geometry, const ChunkCall& call) {
    const int grid = call.B * call.plan.nseg;
    if (geometry == 2) {
        if (call.clk || call.full_taps) hipLaunchKernelGGL((k_gtcrn_chunk<Geo2, true>), dim3(grid), dim3(Geo2::kThreads), chunk_smem_bytes<Geo2>(), s, call);
        else hipLaunchKernelGGL((k_gtcrn_chunk<Geo2, false>), dim3(grid), dim3(Geo2::kThreads), chunk_smem_bytes<Geo2>(), s, call);
    } else if (geometry == 1) {
        if (call.clk || call.full_taps) hipLaunchKernelGGL((k_gtcrn_chunk<Geo1, true>), dim3(grid), dim3(Geo1::kThreads), chunk_smem_bytes<Geo1>(), s, call);
        else hipLaunchKernelGGL((k_gtcrn_chunk<Geo1, false>), dim3(grid), dim3(Geo1::kThreads), chunk_smem_bytes<Geo1>(), s, call);
    } else {
        if (call.clk || call.full_taps) hipLaunchKernelGGL((k_gtcrn_chunk<Geo0, true>), dim3(grid), dim3(Geo0::kThreads), chunk_smem_bytes<Geo0>(), s, call);
        else hipLaunchKernelGGL((k_gtcrn_chunk<Geo0, false>), dim3(grid), dim3(Geo0::kThreads), chunk_smem_bytes<Geo0>(), s, call);
    }
}

}  // namespace ade
