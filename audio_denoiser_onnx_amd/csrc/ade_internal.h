// ade_internal.h — shapes, device-side parameter blocks and launcher prototypes shared by the engine
// (ade_engine.hip) and the gfx950 kernels (ade_kernels.hip).  Not part of the public ABI (include/ade.h).
#pragma once
#include <map>
#include <string>
#include <vector>

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

namespace ade {

// ---- static shapes of the GTCRN path (GTCRN/Export_GTCRN.py:37-39,535,540) ---------------------------------
constexpr int kNfft = 512;
constexpr int kHop = 256;
constexpr int kBins = 257;       // n_fft/2 + 1
constexpr int kBinsPad = 260;    // row pitch of a spectrum row in HBM (16-byte multiple)
constexpr int kErb = 129;        // 65 pass-through + 64 ERB bands
constexpr int kErbPad = 132;
constexpr int kErbLow = 65;
constexpr int kErbBands = 64;
constexpr int kErbHigh = 192;
constexpr int kF1 = 65;          // width after en_convs[0]
constexpr int kFw = 33;          // width after en_convs[1] (DPGRNN "width")
constexpr int kCh = 16;
constexpr int kTileFrames = 7;   // 7 frames x 33 bins = 231 threads of a 256-thread workgroup
constexpr int kTileThreads = kTileFrames * kFw;

// Activation tensor in HBM, channels-last (B, T, F, 16), optionally carrying a deferred TRA gate:
// the logical value of even channel 2i at (b,t,*) is x[...,2i] * at[(b*T+t)*8 + i]   (Export_GTCRN.py:156,324).
struct View {
    const float* x;
    const float* at;   // nullptr: no deferred gate
};

// Banded (sparse) form of the ERB matrices (Export_GTCRN.py:79-107): output o sums `count` consecutive inputs
// starting at start[o]; w is [count][n_out], zero-padded, so the sum equals the dense matmul term for term.
struct BandTab {
    const int* start;
    const float* w;
    int count;
    int n_out;
};

struct FftTabs {
    const float* win;      // [512] analysis == synthesis window (sqrt-hann, periodic)
    const float2* tw256;   // e^{-2 pi i k/256}, k < 256
    const float2* tw512;   // e^{-2 pi i k/512}, k <= 256
    const float* win_sum;  // [256] COLA denominator, one hop period (STFT_Process.py:265-273)
};

struct ConvW {             // ConvBlock after BN fold, weights re-laid out output-channel-fastest
    const float* w;
    const float* b;
    float slope;           // PReLU slope (unused for the Tanh block)
};

struct GtConvW {           // GTConvBlock after BN fold, canonical ("encoder") tap order for both conv flavours
    const float* pw1;      // [24][16]
    const float* pw1_b;    // [16]
    const float* dw;       // [3][3][16]   (kt, kf, c):  y[t,f] += dw[kt][kf][c] * h[t-(2-kt)d, f-1+kf]
    const float* dw_b;     // [16]
    const float* pw2;      // [16][8]
    const float* pw2_b;    // [8]
    const float* gru;      // [16 lanes][78]  TRA GRU rows per lane (3x8 ih | 3x16 hh | 3 b_ih | 3 b_hh)
    const float* fc;       // [8][17]         TRA Linear rows (16 w | bias)
    float pw1_slope, dw_slope;
    int dilation;
    const float* tra_rot;  // [16 lanes][64]  the TRA recurrent rows (3x16) and Linear row (16) of each lane in DPP-rotation order (k_tra)
};

struct DpW {               // one DPGRNN block
    const float* intra_gru;   // [16 lanes][42]   lane = (group, dir, unit)
    const float* inter_gru;   // [16 lanes][54]   lane = (group, unit)
    const float* intra_fc;    // [16][16] (k, co) + [16] bias
    const float* intra_fc_b;
    const float* intra_ln_w;  // [33][16]
    const float* intra_ln_b;
    const float* inter_fc;
    const float* inter_fc_b;
    const float* inter_ln_w;
    const float* inter_ln_b;
    const float* inter_rot;   // [16 lanes][24]   lane = 2 * unit + group: the 3 x 8 recurrent rows in DPP-rotation order (k_inter_gru)
};

// ---- launchers (ade_kernels.hip) ---------------------------------------------------------------------------
void launch_pcm_mean(hipStream_t s, const int16_t* pcm, int B, int L, float* mean, int rows_per_call = 1);
void launch_stft_pcm(hipStream_t s, const int16_t* pcm, const float* mean, int B, int L, int T, FftTabs tabs,
                     BandTab erb_bm, float* spec, float* feat, bool center = true, const float* final_f32 = nullptr);
// GTCRN_CUSTOM's input / output sandwich (float audio, other sample rates, dynamic-length exports; Export_GTCRN.py:636-693): see ade_kernels.hip
void launch_gt_sandwich_in(hipStream_t s, const int16_t* pcm, const float* fin, int rows, int Lin, int L1, int Lm, float lerp1, float lerp2, float gain, float* tmp,
                           float* mean, float* out);
void launch_gt_sandwich_out(hipStream_t s, const float* frames, FftTabs tabs, int rows, int T, int keep, float* wave, int16_t* pcm, float* f32, int Lout, float lerp,
                            bool scale_first);
void launch_stft_ref(hipStream_t s, const float* x, int B, int L, int T, FftTabs tabs, float* ref_spec);
void launch_conv0(hipStream_t s, const float* feat, ConvW w, float* e0, int nframes);
void launch_conv1(hipStream_t s, const float* e0, ConvW w, float* e1, int nframes);
void launch_gt_pw1(hipStream_t s, View a, View skip, GtConvW w, float* h, int nframes);
void launch_gt_dw_pw2(hipStream_t s, const float* h, View a, View skip, GtConvW w, float* xn, float* zt, int B, int T, const float* hist = nullptr);
void launch_tra(hipStream_t s, const float* zt, GtConvW w, float* at, int B, int T, float* state = nullptr);
// +1 if a DPP row_ror:1 hands lane i the value of lane (i + 1) & 15, -1 if of lane (i - 1) & 15 (probed once on the device; the host
// packs the TRA weights of k_tra in the order the rotations deliver the hidden values).  0 = the probe failed.
int dpp_row_ror_direction();

void launch_intra_gru(hipStream_t s, View x, const float* gru, float* rnn, int nframes);
void launch_inter_gru(hipStream_t s, const float* x, const float* gru, float* rnn, int B, int T, float* state = nullptr, const float* rot = nullptr);
void launch_fc_ln_res(hipStream_t s, const float* rnn, View res, const float* fc, const float* fc_b, const float* ln_w,
                      const float* ln_b, float* out, int B, int T);
void launch_deconv3(hipStream_t s, View a, View skip, ConvW w, float* d3, int nframes);
void launch_deconv4(hipStream_t s, const float* d3, const float* e0, ConvW w, float* mask, int nframes);
void launch_istft_masked(hipStream_t s, const float* spec, const float* mask, BandTab erb_bs, FftTabs tabs, float* frames,
                         int nframes);
void launch_istft_ref(hipStream_t s, const float* ref_spec, int B, int T, FftTabs tabs, float* frames);
void launch_ola_pcm(hipStream_t s, const float* frames, FftTabs tabs, int B, int T, int16_t* pcm, float* f32);
// streaming pieces (state carried across pushes; see ade_stream_* in include/ade.h)
void launch_hist_shift(hipStream_t s, const float* hist_in, const float* h, float* hist_out, int B, int T, int depth);
void launch_stream_concat(hipStream_t s, const int16_t* hist, const int16_t* in, int16_t* concat, int B, int P, bool first);
void launch_stream_keep(hipStream_t s, const int16_t* concat, int16_t* hist, int16_t* prev, int B, int P);
void launch_stream_concat_flush(hipStream_t s, const int16_t* hist, const int16_t* prev, int16_t* concat, int B);
// linear resampling of the driver edges (F.interpolate(mode='linear', align_corners=False)); src = scale * (dst + 0.5) - 0.5
void launch_resample_in(hipStream_t s, const int16_t* in, float* out, long long rows, int Lin, int Lout, float scale);
void launch_gt_out(hipStream_t s, const float* wave, int16_t* pcm, float* f32, long long rows, int Lw, int Lout, float lerp, bool scale_first, int nan_to_num /* 0 off, 1 torch.nan_to_num, 2 NaN -> 0 only */);
void launch_resample_in_f32(hipStream_t s, const float* in, float* out, long long rows, int Lin, int Lout, float scale, float gain);
void launch_resample_out(hipStream_t s, const float* in, int16_t* pcm, float* f32, long long rows, int Lin, int Lout, float scale, float pcm_scale, bool truncate_i32,
                         float f32_scale, int nan_to_num /* 0 off, 1 torch.nan_to_num, 2 NaN -> 0 only */);
// IEEE half <-> float tensors of the F16 entry points (exact widening; round-to-nearest-even narrowing, as torch's .to(float16))
void launch_half_to_float(hipStream_t s, const uint16_t* in, float* out, long long n);
void launch_float_to_half(hipStream_t s, const float* in, uint16_t* out, long long n);
void launch_ola_pcm_stream(hipStream_t s, const float* frames, float* carry, FftTabs tabs, int B, int T, bool first, int16_t* pcm, float* f32);

// ---- per-chunk LDS-resident stage kernels (ade_fused.hip) ------------------------------------------------------
// A chunk is walked by one workgroup per SEGMENT of consecutive frames.  Two geometries are compiled from the same stage bodies:
//   geometry 0: 1024-thread workgroups that own up to 64 frames (152 KB of LDS: one workgroup per CU);
//   geometry 1:  512-thread workgroups that own up to 32 frames ( 79 KB of LDS: TWO workgroups per CU, so the serial phases of one --
//                TRA recurrence on one wavefront, inter-frame GRU -- run under the position-parallel phases of the other);
//   geometry 2:  256-thread workgroups that own up to 16 frames ( 40 KB of LDS: FOUR workgroups per CU).
// Everything in GTCRN is causal along time, so segment k + 1 needs from segment k exactly what the streaming path carries between pushes:
// the depthwise-convolution history of each GTConvBlock (handed over as PARTIAL SUMS of the first 2 x dilation frames, see gtblock_stage),
// the six TRA GRU states, the two inter-frame GRU states and the 256-sample overlap-add carry.  They cross in the exchange area below
// (device memory, one slot per (chunk, segment)), each guarded by one flag word; see dev::xwait for the protocol.
constexpr int kXHistFrames = 10;                               // 2 x the largest dilation
constexpr int kXHistFloats = 4 * kXHistFrames * kFw * 4;       // per GTConvBlock: [4 channel quads][10 frames x 33] float4
constexpr int kXTraOff = 6 * kXHistFloats;                     // [6 blocks][16]
constexpr int kXInterOff = kXTraOff + 6 * 16;                  // [2 blocks][33 x 16]
constexpr int kXOlaOff = kXInterOff + 2 * kFw * 16;            // [256]
constexpr int kXPendOff = kXOlaOff + kHop;                     // [256] scratch of the slot's OWN workgroup (geometry 2 parks its first hop here: 40 KB of LDS per workgroup)
constexpr int kXFloats = ((kXPendOff + kHop + 63) / 64) * 64;
constexpr int kXFlags = 16;                                    // hist 0-5 | tra 6-11 | inter 12-13 | ola 14
constexpr int kXFlagHist = 0, kXFlagTra = 6, kXFlagInter = 12, kXFlagOla = 14;
constexpr int kMaxSegments = 8;
constexpr int kClkSlotsPerSeg = 64 * 10;
struct SegPlan {           // how a launch splits its chunks (host-computed, passed by value)
    int wait_ticks;        // bound of one inter-workgroup wait in 10 ns ticks (option "xwait_ms").  MUST stay the first word of the first kernel argument of every kernel that
                           // can wait (k_front .. k_back take a SegPlan first, k_gtcrn_chunk a ChunkCall that begins with its SegPlan): dev::xlimit() reads it from offset 0 of
                           // the kernel-argument segment at the moment of a wait, so that no register holds it across a stage.
    int nseg;              // segments per chunk (1: the whole chunk in one workgroup)
    float* xchg;           // [B][nseg][kXFloats]
    unsigned* flags;       // [B][nseg][kXFlags], zero between launches
    int* err;              // page-locked host word, sticky: the dev::xcode() of the first bounded wait that gave up
    int wave_swap;         // 1: odd segments run their conv lanes on wavefronts 0-3, 6, 7 instead of 0-5 (see gtblock_stage)
    // STREAMS (ade_stream_*): a push is a chunk whose first segment continues the state its last segment left one launch earlier -- the same exchange slots, one per
    // stream, ping-ponged between pushes -- framed without centre padding and emitted one hop behind (include/ade.h).
    float* carry_in;       // [B][kXFloats] state the FIRST segment of every chunk continues from (null: a fresh signal)
    unsigned* carry_in_flags;
    float* carry_out;      // [B][kXFloats] state the LAST segment leaves (null: nothing follows)
    unsigned* carry_out_flags;
    int stream;            // 1: streaming framing -- frame t reads samples 256 t .. 256 t + 511 of the row (256 carried + the push), output sample n = overlap-add sample n
    int prio;              // base wave priority of the workgroups by segment (option "seg_prio"): 0 none; 1-3 = that level for every later segment; 4 = earlier segments first (2, 1, 0, 0)
    int withhold;          // TEST HOOK (option "xchg_withhold"): block 0 raises its flags in the launch's LAST slot, which nobody polls, so its successor's bounded waits give up
};
struct Seg {               // one workgroup's share (device-side)
    int t0, nT, T;         // first frame, frames owned, frames of the chunk
    int prev, next;        // there is a segment before / after this one
    int swap;              // this workgroup permutes its upper wavefronts (SegPlan::wave_swap, odd segments)
    int stream;            // SegPlan::stream
    int first;             // nothing precedes this segment at all: a chunk's / stream's very first frames
    int base_prio;         // wave priority outside the recurrences (which run at 3)
    float* xo;             // exchange slot this segment fills (for the next one)
    float* xi;             // exchange slot of the previous segment
    unsigned* fo;          // flags this segment raises
    unsigned* fi;          // flags of the previous segment (polled, then lowered)
    int* err;
};
int fused_geometries();                       // 3
int fused_max_frames(int geometry);           // frames one workgroup can own
bool fused_supported(int T, int geometry);    // T >= 2 and ceil(T / max_frames) <= kMaxSegments
int fused_segments(int T, int geometry);
hipError_t fused_init();   // raises the dynamic-LDS limit of the stage kernels (once per process/device)
void launch_gtblock(hipStream_t s, int geometry, SegPlan plan, int blk, const float* a, const float* skip, GtConvW w, float* out, int B, int T, long long* clk);
void launch_dpgrnn(hipStream_t s, int geometry, SegPlan plan, int blk, const float* x, DpW w, float* out, int B, int T, long long* clk);
// front / back stages (ade_stage_frontback.h).  All (B,.,.,16) tensors of the fused path are channel-quad planar:
// X[b][q][p] = float4(channels 4q..4q+3 of position p).
void launch_front(hipStream_t s, int geometry, SegPlan plan, const int16_t* pcm, int B, int L, int T, FftTabs tabs, BandTab erb_bm, ConvW c0, ConvW c1, float* spec,
                  float* e0, float* e1, long long* clk, const float* dc = nullptr);
void launch_back(hipStream_t s, int geometry, SegPlan plan, const float* x, const float* e1, const float* e0, const float* spec, ConvW c3, ConvW c4, BandTab erb_bs,
                 FftTabs tabs, int16_t* pcm, float* f32, int B, int T, long long* clk);

// What the single-launch chunk kernel needs.  The per-engine part lives in DEVICE memory (uploaded when the workspace is reserved) and is
// fetched stage by stage through the scalar cache: as a by-value kernel argument its ~250 dwords were all loaded at kernel entry and parked
// in VGPR lanes for the whole launch (232 SGPR spills at the 128-VGPR ceiling).
struct ChunkFixed {
    FftTabs tabs;
    BandTab erb_bm, erb_bs;
    ConvW en0, en1, de3, de4;
    GtConvW en_gt[3], de_gt[3];
    DpW dp[2];
    float *spec, *e0, *e1, *xe[3], *dpo[2], *xd[3];
};
constexpr int kReadyStride = 16384;      // words: 64 KB, the smallest copy this runtime gives to a copy engine
struct ChunkCall {           // per call, by value
    SegPlan plan;            // (first: see SegPlan::wait_ticks)
    const ChunkFixed* fixed;
    const int16_t* pcm_in;
    int16_t* pcm_out;
    float* f32_out;          // optional pre-PCM waveform
    const float* dc;         // optional per-row DC means computed upstream (batch-fold: one mean per call, shared by its windows)
    long long* clk;          // optional phase clocks, kClkSlotsPerSeg per segment
    int L, T, B;
    int chunk0;              // first chunk of this launch in the batch arrays (pcm_in / pcm_out / f32_out, the workspace tensors and the exchange slots are indexed by chunk0 + the
                             // launch's own chunk number): 0 except for the sub-batch launches of ade_process
    int full_taps;           // 1: launch the debug build, which stores every inter-stage tensor whole (option "full_taps", for ade_debug_tap); 0: the three tensors whose
                             // channels 0-7 only the following block reads -- through LDS -- keep those planes out of HBM (x_d0, x_d1, dp2).  Read by the launcher only.
    int stagger;             // > 0: every other group of 8 workgroups starts this many 10 ns ticks late (geometry 0: the workgroups would all
                             // hit HBM with the same stage's burst at the same instant; ade_set_option "stagger_us")
    // Host batches streamed THROUGH one launch (ade_process): the rows are dealt into groups of `group_rows`; a group's workgroups wait until a copy engine has delivered
    // the group's PCM (in_ready[g * kReadyStride] == epoch: the first word of a 64 KB block copied behind the group's PCM on the same copy stream) and, once all of them
    // have stored their output, the last one writes out_done[g] = epoch into page-locked host memory, which the host thread is polling to start the group's copy-out --
    // the copies of the other groups run under this launch's arithmetic.  in_ready null: PCM is resident, nothing waits and nothing is signalled.
    // (Why not hipStreamWriteValue32 / hipStreamWaitValue32 or a 4-byte copy: measured on this runtime, those and every copy below 64 KB are executed by a shader and cannot
    //  start while this launch fills every wavefront slot -- tools/ubench/copy_under_full_gpu_probe.hip; copies from 64 KB up run on the SDMA engines.)
    const unsigned* in_ready;      // fine-grained device memory, one block of kReadyStride words per group
    unsigned* out_done;            // [groups] page-locked host memory
    unsigned* out_count;           // [groups] device memory: workgroups of the group that are done (reset by the last)
    unsigned epoch;
    int group_rows;
};
void launch_gtcrn_chunk(hipStream_t s, int geometry, const ChunkCall& call);

// ---- a tensor of the parsed ADEWGT01 weight blob (host memory, owned by the engine while it is being built)
struct Tensor {
    std::vector<int> dims;
    const float* data = nullptr;
    size_t count = 0;
};

// ---- model families other than GTCRN are sub-engines behind one interface; ade_create builds them from the same manifest +
//      blob format and ade_process / ade_run_device / ade_debug_tap forward to them.  Every GEMM-shaped step of these models
//      runs on the matrix cores (csrc/ade_gemm.h).  PCM rows are [batch][channels()][in_len()] int16.
struct SubEngine {
    virtual ~SubEngine() {}
    virtual int frames() const = 0;
    virtual int in_len() const = 0;      // samples per channel row
    virtual int out_len() const = 0;
    virtual int channels() const { return 1; }
    virtual int out_channels() const { return channels(); }   // H-GTCRN: two microphones in, one channel out
    // Resampled input (in_sample_rate != model_sample_rate): when set, run() reads its PCM from here -- floats in int16 units, same
    // [batch][channels()][in_len()] layout -- instead of d_in (the reference interpolates `audio.float()` before anything else).
    const float* float_in = nullptr;
    int float_src_len = 0;               // samples per channel row of the caller-rate PCM behind float_in (set once by the engine; H-GTCRN's DC mean)
    const float* float_src = nullptr;    // the caller-rate FLOAT tensor behind float_in when the call came in through a float entry (d_in is then not PCM)
    float float_src_gain = 1.0f;         // ... times this = int16 units (32768 for normalised samples)
    virtual bool accepts_float_input() const { return false; }
    virtual int n_outputs() const { return 1; }   // output tensors per call; PCM out rows are [batch][n_outputs()][out_channels()][out_len()]
    virtual int reserve(int batch, std::string& err) = 0;                                                                          // ade_status
    virtual int run(hipStream_t s, const int16_t* d_in, int batch, int16_t* d_out, float* d_f32, std::string& err) = 0;
    virtual int tap(hipStream_t s, const char* name, int batch, float* out, size_t count, size_t* written, std::string& err) = 0;
    // A sub-engine whose stages hand states between workgroups through bounded waits (H-GTCRN's fused network middle) reports a wait that gave up HERE: the engine calls this
    // after its stream synchronise on every entry point and at the entry of the next call -- outside any captured graph, so a replayed launch sequence is covered too.
    // -> 0, or the dev::xcode() the device left; a non-zero answer has drained the device, cleared the word and lowered every flag (a late producer may have raised one
    // that nobody consumed).  xwait: the engine's option "xwait_ms" in 10 ns ticks.
    virtual int exchange_error_and_reset() { return 0; }
    virtual void set_exchange_wait_ticks(int) {}
};
// model_family "dfsmn" (DFSMN/Export_DFSMN.py:71-246), csrc/ade_dfsmn.hip
int dfsmn_create(const std::map<std::string, Tensor>& tensors, int window_len, int n_win, int device, SubEngine** out, std::string& err);
// model_family "mel_band_roformer" (Mel_Band_Roformer/Stereo/Export_MelBandRoformer.py:262-680), csrc/ade_melband.hip
// model_family "mossformer2_ss" (MossFormer2_SS_16K/Export_MossFormer2_SS_16K.py:84-662), csrc/ade_mossformer.hip
int mossformer_create(const std::map<std::string, Tensor>& tensors, int window_len, int n_win, bool dynamic, int device, SubEngine** out, std::string& err);
// model_family "ul_unas" (UL-UNAS/Export_UL_UNAS.py:51-913), csrc/ade_ulunas.hip
int ulunas_create(const std::map<std::string, Tensor>& tensors, int window_len, int n_win, int dynamic_keep /* 0: static export; > 0: dynamic, the caller-rate input length */, int device, SubEngine** out, std::string& err);
// model_family "h_gtcrn" (H-GTCRN/Export_H_GTCRN.py:428-1063), csrc/ade_hgtcrn.hip
int hgtcrn_create(const std::map<std::string, Tensor>& tensors, int window_len, int n_win, bool dynamic, int device, SubEngine** out, std::string& err);
// model_family "zipenhancer" (ZipEnhancer/Export_ZipEnhancer.py:357-927), csrc/ade_zipenhancer.hip
int zipenhancer_create(const std::map<std::string, Tensor>& tensors, int window_len, int n_win, bool exact_dft, bool bf16 /* ade_gemm_dtype = bf16: csrc/ade_zip16.h */, bool dynamic /* DYNAMIC_AXES export: divide by the overlap-add denominator */, int device, SubEngine** out, std::string& err);
int melband_create(const std::map<std::string, Tensor>& tensors, int window_len, int n_win, bool exact_dft, bool bf16, bool dynamic, int device, SubEngine** out, std::string& err);

}  // namespace ade
