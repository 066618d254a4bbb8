// ade_engine.hip — host side of libade: the C ABI of include/ade.h, manifest/weight loading, HBM workspace,
// launch sequence (optionally replayed from a captured hipGraph) and parity/timing taps.
//
// One engine = one HIP device, one stream, one resident copy of the weights (191 KB) and a workspace sized for
// `capacity` independent chunks (~2.6 MB of fp32 activations per 1 s chunk; 288 GB of HBM3E holds >100k chunks).
// There is no CPU execution mode: without a HIP device ade_create fails with ADE_ERR_DEVICE.
#include "ade_internal.h"
#include "ade_gtcrn_pack.h"

#include "../../include/ade.h"

#if !defined(HIPSIM)
#include <dlfcn.h>      // ade_stitch_device opens librccl on first use
#endif

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

using namespace ade;

namespace {

thread_local std::string g_create_error;

struct KernelStat {
    std::string name;
    float ms = 0.f;
    int launches = 0;
};

struct GraphEntry {
    const void* in;
    void* out_pcm;
    void* out_f32;
    int batch;
    hipGraphExec_t exec;
    hipGraph_t graph;
};

}  // namespace

constexpr int kXErrWords = 8;      // = the sub-batch streams of ade_process
struct ade_engine {
    int device = 0;
    int in_len = 0, T = 0, out_len = 0;   // per WINDOW (== per call unless batch-fold)
    int n_win = 1;                        // windows per call: USE_BATCH_FOLD folds (1,1,n_win*W) into (n_win,1,W), Export_GTCRN.py:656-660
    int sample_rate = 16000, in_rate = 0, out_rate = 0;   // model rate; caller-side rates when they differ (0 = same)
    std::string last_error;
    std::map<std::string, std::string> meta;
    std::vector<float> blob_storage;
    std::map<std::string, Tensor> tensors;

    std::vector<struct ade_stream*> live_streams;   // streams created on this engine: ade_destroy releases their device state and orphans them
    ade::SubEngine* sub = nullptr;        // model_family "dfsmn" / "mel_band_roformer": a sub-engine (everything below is GTCRN's)
    int channels = 1, out_channels = 1, n_outputs = 1;   // in_len / out_len below count one batch item: channels * samples in, n_outputs * out_channels * samples out
    // driver-edge resampling around a sub-engine (in / out sample rate != model rate): caller-side lengths above, model-side below
    bool resample = false;
    int rs_model_in = 0, rs_model_out = 0;     // per channel row, at the model rate
    float rs_scale_in = 1.0f, rs_scale_out = 1.0f, rs_pcm_scale = 1.0f;
    bool rs_truncate_i32 = false;
    bool rs_sandwich_out = false;        // Mel-Band: the GTCRN-style output sandwich (interpolate before the PCM scale when down-sampling, after it when up-sampling)
    bool rs_scale_first = false;
    int rs_nan_to_num = 0;               // ahead of the PCM tail: 1 = torch.nan_to_num (ZipEnhancer always; UL-UNAS for float input), 2 = where(isnan, 0) only (H-GTCRN keeps its infinities)
    float rs_in_gain = 32768.0f;         // float audio input: normalised samples -> the PCM units the sub-engines read (MossFormer2 is fed as it is: 1)
    float rs_f32_scale = 1.0f;           // the export's F32 / F16 output from the model-rate waveform (2^-15 where that is in PCM units: MossFormer2, ZipEnhancer)
    float *rs_in = nullptr, *rs_out = nullptr;

    // GTCRN_CUSTOM's input / output sandwich (Export_GTCRN.py:636-693): float audio in, other sample rates and dynamic-length exports.  in_len / out_len above are the
    // CALLER-side lengths; the model-rate ones live here.  The sandwich takes the multi-kernel launch sequence (its STFT reads the prepared fp32 waveform).
    bool gt_sand = false, gt_float_in = false, gt_scale_first = false;
    int gt_l1 = 0, gt_lm = 0, gt_keep = 0;            // stage-1 length, model-rate input length, model-rate output samples kept (256 T when dynamic)
    float gt_lerp1 = 0.0f, gt_lerp2 = 0.0f, gt_gain = 1.0f, gt_lerp_out = 0.0f;
    float *gt_tmp = nullptr, *gt_in = nullptr, *gt_wave = nullptr, *gt_mean = nullptr;
    float* d_f32_in = nullptr;                        // staging of ade_process_f32
    uint16_t *d_f16_in = nullptr, *d_f16_out = nullptr;   // staging of ade_process_f16 (allocated by its first call)
    uint16_t *h_f16_in = nullptr, *h_f16_out = nullptr;
    int f16_capacity = 0;
    const float* cur_fin = nullptr;                   // the float input of the call being enqueued (float-input engines)

    hipStream_t stream = nullptr;
    hipStream_t sub_streams[8] = {};      // ade_process: one stream per sub-batch (created on first use)
    int host_split = 0;                   // option "host_split": sub-batches of a host batch (0 = per call: two from 128 rows; 1 = never; see ade_process)
    // a host batch streamed through ONE launch (ade_process; ChunkCall::in_ready)
    static constexpr int kMaxGroups = 8;
    int host_stream = 0;                  // option "host_stream": row groups of the streamed launch (0 = per call: four from 128 rows; 1 = off: the sub-batch path)
    int sio_state = 0;                    // 0: not set up yet, 1: ready, -1: unavailable (an allocation was refused)
    hipStream_t s_in = nullptr, s_out = nullptr;
    unsigned* d_sio_ready = nullptr;      // fine-grained device memory [kMaxGroups][kReadyStride]: block g is overwritten (from h_sio_epoch) behind group g's copy-in
    unsigned* h_sio_epoch = nullptr;      // page-locked [kReadyStride]: word 0 = the call's epoch, the source of those copies
    unsigned* h_sio_done = nullptr;       // page-locked [kMaxGroups]: written by a group's last workgroup, polled by the host thread
    unsigned* d_sio_count = nullptr;      // device memory [kMaxGroups]
    unsigned sio_epoch = 0;
    // ade_submit / ade_wait: a ring of `pipe_depth` submissions in flight -- H2D of call k + 1 on s_pin, the kernels of call k on `stream`, D2H of call k - 1 on s_pout
    struct PipeSlot {
        int16_t *d_in = nullptr, *d_out = nullptr, *h_in = nullptr, *h_out = nullptr;     // device buffers of the slot; page-locked staging for pageable callers
        float *d_f32 = nullptr, *h_f32 = nullptr;
        hipEvent_t ev_in = nullptr, ev_k = nullptr, ev_out = nullptr;
        unsigned long long ticket = 0;
        int state = 0;                    // 0 free, 1 in flight, 2 finished (its status waits for ade_wait)
        int rows = 0;
        int16_t* out_pcm = nullptr;       // where a pageable caller's output goes when the slot is finished (null: it was DMA'd directly)
        float* out_f32 = nullptr;
        bool d2h_pending = false;         // the copy-out has not been enqueued yet (it goes out behind the NEXT submission's copy-in: see pipe_copy_out)
        int16_t* dst_pcm = nullptr;       // its destinations (the caller's page-locked buffer or the slot's staging; null: not asked for)
        float* dst_f32 = nullptr;
        int status = 0;
        std::string error;
    };
    static constexpr int kMaxPipe = 4;
    PipeSlot pipe[kMaxPipe];
    int pipe_depth = 3;                   // option "pipe_depth" (2 .. 4): at 2 the copy-in of call k + 1 can only be enqueued once call k - 1 has been waited for, i.e. behind its copy-out
    int pipe_capacity = 0;                // rows every slot's buffers hold
    unsigned long long pipe_next = 1;     // the next ticket
    hipStream_t s_pin = nullptr, s_pout = nullptr;
    float* d_weights = nullptr;
    int* d_ints = nullptr;
    FftTabs tabs{};
    BandTab erb_bm{}, erb_bs{};
    ConvW en0{}, en1{}, de3{}, de4{};
    GtConvW en_gt[3]{}, de_gt[3]{};
    DpW dp[2]{};

    // workspace
    int capacity = 0;
    int last_batch = 0;
    int16_t* d_pcm_in = nullptr;
    int16_t* d_pcm_out = nullptr;
    float* d_f32_out = nullptr;
    float* ws = nullptr;   // one slab, carved below
    float *mean = nullptr, *spec = nullptr, *feat = nullptr, *e0 = nullptr, *e1 = nullptr, *h = nullptr, *zt = nullptr;
    float *xe[3] = {}, *ate[3] = {}, *xd[3] = {}, *atd[3] = {};
    float *rnn = nullptr, *dpm[2] = {}, *dpo[2] = {};
    float *d3 = nullptr, *mask = nullptr, *frames = nullptr;
    int16_t* h_pcm_in = nullptr;    // pinned staging for ade_process
    int16_t* h_pcm_out = nullptr;
    float* h_f32_out = nullptr;

    int stagger_ticks = 2750;             // 27.5 us, applied when a launch has enough chunks to load the memory system (see enqueue; geometry 0 only)
    int wave_swap = 0;                    // option "wave_swap": odd segments run their conv lanes on wavefronts 0-3, 6, 7 (measured neutral to -1 %: off)
    int seg_prio = 0;                     // base wave priority of the workgroups by segment (option "seg_prio", 0-4: SegPlan::prio)
    int xwait_ticks = 20000000;           // bound of one inter-workgroup wait in 10 ns ticks (option "xwait_ms"; 0.2 s)
    int full_taps = 0;                    // option "full_taps": the single-launch kernel stores every inter-stage tensor whole (ChunkCall::full_taps)
    int xchg_withhold = 0;                // test hook (option "xchg_withhold"): SegPlan::withhold
    int xwait_retry = 1;                  // option "xwait_retry": ade_process re-runs a call whose segmented launch timed out ONCE on the path without hand-offs (see ade_process)
    bool timed_out = false;               // exchange_status() found a time-out (as opposed to any other ADE_ERR_DEVICE)
    int retries = 0;                      // calls re-run that way (debug tap "xwait_retries")
    int geometry = -1;                    // fused-path workgroup geometry (ade_internal.h): -1 = choose per call, 0 = 1024 threads x 64 frames, 1 = 512 x 32
    ade::ChunkFixed* d_fixed = nullptr;   // device copy of the chunk kernel's per-engine arguments (rebuilt by reserve)
    float* d_xchg = nullptr;              // segment exchange area [capacity][kMaxSegments slots as needed][kXFloats]
    unsigned* d_xflags = nullptr;         // its flags (zero between launches)
    int* d_xerr = nullptr;                // page-locked host words the kernels see: the dev::xcode() of the first bounded inter-workgroup wait that gave up.  kXErrWords words: a launch
                                          // reports into word 0, the k-th concurrent sub-batch launch of ade_process into word k (its block indices count from ITS first chunk)
    int xl_B[kXErrWords] = {}, xl_chunk0[kXErrWords] = {};      // chunks / first chunk of the launch that reports into word k (xl_B 0: one launch over the whole batch, last_batch)
    int xchg_capacity = 0;                // chunks the exchange area holds
    int xchg_segments = 0;                // slots per chunk the exchange area was sized for
    bool use_graph = true;
    bool graph_supported = true;
    bool use_fused = true;      // per-chunk LDS-resident stage kernels when T <= 64 (ade_fused.hip)
    bool use_single = true;     // ... as ONE launch (k_gtcrn_chunk); profile mode always uses the per-stage kernels
    bool last_fused = false;
    int last_geometry = -1;
    long long* d_clk = nullptr;   // phase-clock slots (profile modes 1 and 3 only)
    std::vector<GraphEntry> graphs;

    bool profile = false;
    int profile_mode = 1;       // 1: one kernel per stage (+ phase clocks); 2: the shipped launch sequence, timed as launched
    std::vector<KernelStat> stats;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    std::vector<int> event_stat;
    int events_used = 0;
};

namespace {

ade_status fail(ade_engine* e, ade_status st, const std::string& msg) {
    if (e) e->last_error = msg;
    else g_create_error = msg;
    return st;
}

#define HIP_TRY(e, expr)                                                                                        \
    do {                                                                                                        \
        hipError_t _err = (expr);                                                                               \
        if (_err != hipSuccess)                                                                                 \
            return fail((e), ADE_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_err));              \
    } while (0)

// ---- flat JSON object of string / number / bool values -> string map ------------------------------------
bool parse_manifest(const char* s, std::map<std::string, std::string>& out, std::string& err) {
    size_t i = 0, n = strlen(s);
    auto ws = [&]() { while (i < n && (s[i] == ' ' || s[i] == '\n' || s[i] == '\t' || s[i] == '\r')) ++i; };
    auto str = [&](std::string& v) -> bool {
        if (s[i] != '"') return false;
        ++i;
        v.clear();
        while (i < n && s[i] != '"') {
            if (s[i] == '\\' && i + 1 < n) {
                ++i;
                switch (s[i]) {
                    case 'n': v += '\n'; break;
                    case 't': v += '\t'; break;
                    case 'u': v += '?'; i += 4; break;
                    default: v += s[i];
                }
                ++i;
            } else v += s[i++];
        }
        if (i >= n) return false;
        ++i;
        return true;
    };
    ws();
    if (i >= n || s[i] != '{') { err = "manifest is not a JSON object"; return false; }
    ++i;
    ws();
    if (i < n && s[i] == '}') return true;
    while (i < n) {
        ws();
        std::string key, val;
        if (!str(key)) { err = "manifest: expected a string key"; return false; }
        ws();
        if (i >= n || s[i] != ':') { err = "manifest: expected ':' after key " + key; return false; }
        ++i;
        ws();
        if (i < n && s[i] == '"') {
            if (!str(val)) { err = "manifest: bad string value for " + key; return false; }
        } else {
            size_t j = i;
            while (j < n && s[j] != ',' && s[j] != '}' && s[j] != ' ' && s[j] != '\n') ++j;
            val.assign(s + i, j - i);
            i = j;
            if (val == "true") val = "1";
            else if (val == "false") val = "0";
            else if (val == "null") val.clear();
        }
        out[key] = val;
        ws();
        if (i < n && s[i] == ',') { ++i; continue; }
        if (i < n && s[i] == '}') return true;
        err = "manifest: expected ',' or '}' after " + key;
        return false;
    }
    err = "manifest: unterminated object";
    return false;
}

// the reference's REQUIRED_AUDIO_METADATA_KEYS (audio_onnx_metadata.py:8-26)
const char* kRequiredKeys[] = {"audio_metadata_version", "producer", "model_name", "task", "model_family", "dynamic_axes", "opset",
                               "input_audio_dtype", "output_audio_dtype", "in_sample_rate", "out_sample_rate", "model_sample_rate",
                               "input_audio_length", "input_to_output_scale", "max_dynamic_audio_seconds",
                               "normalize_audio_default", "normalize_target_rms"};

// _parse_bool (audio_onnx_metadata.py:281-287)
bool parse_bool(const std::string& v, bool* out) {
    std::string s;
    for (char c : v) s += (char)tolower(c);
    if (s == "1" || s == "true" || s == "yes" || s == "on") { *out = true; return true; }
    if (s == "0" || s == "false" || s == "no" || s == "off") { *out = false; return true; }
    return false;
}

bool parse_int(const std::string& v, long* out) {
    char* end = nullptr;
    long x = strtol(v.c_str(), &end, 10);
    if (end == v.c_str() || *end) return false;
    *out = x;
    return true;
}

ade_status parse_blob(ade_engine* e, const void* blob, size_t nbytes) {
    const unsigned char* p = (const unsigned char*)blob;
    if (!blob || nbytes < 12 || memcmp(p, "ADEWGT01", 8) != 0) return fail(e, ADE_ERR_BAD_VALUE, "weights: not an ADEWGT01 blob");
    uint32_t n;
    memcpy(&n, p + 8, 4);
    size_t pos = 12;
    struct Ent { std::string name; std::vector<int> dims; uint64_t off, nb; size_t count; };
    std::vector<Ent> ents;
    for (uint32_t i = 0; i < n; ++i) {
        uint16_t ln;
        if (pos + 2 > nbytes) return fail(e, ADE_ERR_BAD_VALUE, "weights: truncated header");
        memcpy(&ln, p + pos, 2);
        pos += 2;
        if (pos + ln + 2 > nbytes) return fail(e, ADE_ERR_BAD_VALUE, "weights: truncated header");
        Ent en;
        en.name.assign((const char*)p + pos, ln);
        pos += ln;
        const int dtype = p[pos], ndim = p[pos + 1];
        pos += 2;
        if (dtype != 0 || ndim > 8 || pos + 4u * ndim + 16 > nbytes) return fail(e, ADE_ERR_BAD_VALUE, "weights: bad tensor header " + en.name);
        en.count = 1;
        for (int d = 0; d < ndim; ++d) {
            uint32_t v;
            memcpy(&v, p + pos, 4);
            pos += 4;
            en.dims.push_back((int)v);
            if (v != 0 && en.count > (size_t)0x3fffffffffffull / v) return fail(e, ADE_ERR_BAD_VALUE, "weights: tensor too large " + en.name);
            en.count *= v;
        }
        memcpy(&en.off, p + pos, 8);
        memcpy(&en.nb, p + pos + 8, 8);
        pos += 16;
        if (en.nb != en.count * 4) return fail(e, ADE_ERR_BAD_VALUE, "weights: bad extent " + en.name);
        ents.push_back(en);
    }
    const size_t data0 = (pos + 63) & ~(size_t)63;
    if (data0 > nbytes) return fail(e, ADE_ERR_BAD_VALUE, "weights: truncated header");
    const size_t avail = nbytes - data0;
    size_t total = 0;
    for (auto& en : ents) {
        if (en.off > avail || en.nb > avail - en.off) return fail(e, ADE_ERR_BAD_VALUE, "weights: data out of range " + en.name);
        total += en.count;
    }
    e->blob_storage.resize(total + 1);
    size_t w = 0;
    for (auto& en : ents) {
        memcpy(e->blob_storage.data() + w, p + data0 + en.off, en.nb);
        Tensor t;
        t.dims = en.dims;
        t.data = e->blob_storage.data() + w;
        t.count = en.count;
        e->tensors[en.name] = t;
        w += en.count;
    }
    return ADE_OK;
}

struct Loader {
    ade_engine* e;
    ade_status st = ADE_OK;
    const float* get(const std::string& name, std::initializer_list<int> dims) {
        auto it = e->tensors.find(name);
        if (it == e->tensors.end()) {
            if (st == ADE_OK) st = fail(e, ADE_ERR_MISSING_KEY, "weights: tensor missing: " + name);
            return nullptr;
        }
        std::vector<int> want(dims);
        if (it->second.dims != want) {
            if (st == ADE_OK) st = fail(e, ADE_ERR_SHAPE_MISMATCH, "weights: tensor has the wrong shape: " + name);
            return nullptr;
        }
        return it->second.data;
    }
};

constexpr int kClkSlots = ade::kClkSlotsPerSeg * ade::kMaxSegments;   // phase clocks of every segment of chunk 0 (the stage bodies stamp seg * kClkSlotsPerSeg + slot)

ade_status build_device_constants(ade_engine* e) {
    Loader L{e};
    Arena A;
    // --- FFT tables.  Window exactly as the reference builds it in fp32 (STFT_Process.py:93 'hann_sqrt' periodic).
    const size_t o_win = A.alloc(kNfft), o_tw256 = A.alloc(512), o_tw512 = A.alloc(2 * 257 + 2), o_ws = A.alloc(kHop);
    {
        const float step = (float)(2.0 * M_PI / (double)kNfft);
        for (int n = 0; n < kNfft; ++n) A.f[o_win + n] = sqrtf(cosf((float)n * step) * -0.5f + 0.5f);
        for (int k = 0; k < 256; ++k) {
            A.f[o_tw256 + 2 * k] = (float)cos(2.0 * M_PI * k / 256.0);
            A.f[o_tw256 + 2 * k + 1] = (float)-sin(2.0 * M_PI * k / 256.0);
        }
        for (int k = 0; k <= 256; ++k) {
            A.f[o_tw512 + 2 * k] = (float)cos(2.0 * M_PI * k / 512.0);
            A.f[o_tw512 + 2 * k + 1] = (float)-sin(2.0 * M_PI * k / 512.0);
        }
        A.f[o_tw512 + 2 * 256] = -1.0f;
        A.f[o_tw512 + 2 * 256 + 1] = 0.0f;
        A.f[o_tw512 + 2 * 128] = 0.0f;   // e^{-i pi/2} = -i exactly
        A.f[o_tw256 + 2 * 64] = 0.0f;
        A.f[o_tw256 + 2 * 128 + 1] = 0.0f;
        A.f[o_tw256 + 2 * 192] = 0.0f;
        // COLA denominator over one hop period: conv_transpose1d(ones, w^2) in fp32 (STFT_Process.py:254-273)
        for (int r = 0; r < kHop; ++r) {
            const float a = A.f[o_win + r] * A.f[o_win + r], b = A.f[o_win + kHop + r] * A.f[o_win + kHop + r];
            A.f[o_ws + r] = a + b;
        }
    }
    // --- ERB matrices -> banded tables
    const float* erb_t = L.get("erb.erb_weight_t", {kErbHigh, kErbBands});
    const float* ierb_t = L.get("erb.ierb_weight_t", {kErbBands, kErbHigh});
    if (L.st != ADE_OK) return L.st;
    std::vector<int> bm_start, bs_start;
    std::vector<float> bm_w, bs_w;
    int bm_count = 0, bs_count = 0;
    band_table(erb_t, kErbHigh, kErbBands, bm_start, bm_w, bm_count);
    band_table(ierb_t, kErbBands, kErbHigh, bs_start, bs_w, bs_count);
    const size_t o_bm = A.alloc(bm_w.size()), o_bs = A.alloc(bs_w.size());
    memcpy(&A.f[o_bm], bm_w.data(), bm_w.size() * 4);
    memcpy(&A.f[o_bs], bs_w.data(), bs_w.size() * 4);
    // --- conv blocks
    const float* w0 = L.get("encoder.en_convs.0.conv.weight", {16, 9, 1, 5});
    const float* b0 = L.get("encoder.en_convs.0.conv.bias", {16});
    const float* a0 = L.get("encoder.en_convs.0.act.weight", {1});
    const float* w1 = L.get("encoder.en_convs.1.conv.weight", {16, 8, 1, 5});
    const float* b1 = L.get("encoder.en_convs.1.conv.bias", {16});
    const float* a1 = L.get("encoder.en_convs.1.act.weight", {1});
    const float* w3 = L.get("decoder.de_convs.3.conv.weight", {16, 8, 1, 5});
    const float* b3 = L.get("decoder.de_convs.3.conv.bias", {16});
    const float* a3 = L.get("decoder.de_convs.3.act.weight", {1});
    const float* w4 = L.get("decoder.de_convs.4.conv.weight", {16, 2, 1, 5});
    const float* b4 = L.get("decoder.de_convs.4.conv.bias", {2});
    if (L.st != ADE_OK) return L.st;
    const size_t o_w0 = A.alloc(5 * 9 * 16 + 3 * 8 * 16), o_b0 = A.alloc(16), o_w1 = A.alloc(5 * 2 * 8 * 8), o_b1 = A.alloc(16);
    const size_t o_w3 = A.alloc(5 * 2 * 8 * 8), o_b3 = A.alloc(16), o_w4 = A.alloc(5 * 16 * 2), o_b4 = A.alloc(2);
    for (int co = 0; co < 16; ++co)
        for (int ci = 0; ci < 9; ++ci)
            for (int k = 0; k < 5; ++k) A.f[o_w0 + (k * 9 + ci) * 16 + co] = w0[(co * 9 + ci) * 5 + k];
    // conv0 on the matrix cores (front stage, main round): the SFE tap o and the conv tap k with the same k + o read the same input column, so
    // their weights are one term; the eighth slot of each input channel takes the k = 1, o = 2 term back at fo = 0 (its SFE position is padding).
    for (int co = 0; co < 16; ++co)
        for (int c = 0; c < 3; ++c) {
            for (int jj = 0; jj < 7; ++jj) {
                float sum = 0.0f;
                for (int k = 0; k < 5; ++k) {
                    const int o = jj - k;
                    if (o >= 0 && o < 3) sum += w0[(co * 9 + c * 3 + o) * 5 + k];
                }
                A.f[o_w0 + 720 + (c * 8 + jj) * 16 + co] = sum;
            }
            A.f[o_w0 + 720 + (c * 8 + 7) * 16 + co] = -w0[(co * 9 + c * 3 + 2) * 5 + 1];
        }
    for (int g = 0; g < 2; ++g)
        for (int co = 0; co < 8; ++co)
            for (int ci = 0; ci < 8; ++ci)
                for (int k = 0; k < 5; ++k) {
                    A.f[o_w1 + ((k * 2 + g) * 8 + ci) * 8 + co] = w1[((g * 8 + co) * 8 + ci) * 5 + k];   // Conv2d (Cout, Cin/g,1,5)
                    A.f[o_w3 + ((k * 2 + g) * 8 + ci) * 8 + co] = w3[((g * 8 + ci) * 8 + co) * 5 + k];   // ConvT  (Cin, Cout/g,1,5)
                }
    for (int ci = 0; ci < 16; ++ci)
        for (int co = 0; co < 2; ++co)
            for (int k = 0; k < 5; ++k) A.f[o_w4 + (k * 16 + ci) * 2 + co] = w4[(ci * 2 + co) * 5 + k];
    memcpy(&A.f[o_b0], b0, 64);
    memcpy(&A.f[o_b1], b1, 64);
    memcpy(&A.f[o_b3], b3, 64);
    memcpy(&A.f[o_b4], b4, 8);
    GtOff gte[3], gtd[3];
    DpOff dpo[2];
    static const int en_dil[3] = {1, 2, 5}, de_dil[3] = {5, 2, 1};   // Export_GTCRN.py:490-492,512-514
    for (int i = 0; i < 3; ++i) {
        if (!load_gt(L, A, "encoder.en_convs." + std::to_string(i + 2) + ".", false, gte[i])) return L.st;
        if (!load_gt(L, A, "decoder.de_convs." + std::to_string(i) + ".", true, gtd[i])) return L.st;
    }
    if (!load_dp(L, A, "dpgrnn1.", dpo[0]) || !load_dp(L, A, "dpgrnn2.", dpo[1])) return L.st;

    // --- upload
    HIP_TRY(e, hipMalloc((void**)&e->d_weights, A.f.size() * sizeof(float)));
    HIP_TRY(e, hipMemcpy(e->d_weights, A.f.data(), A.f.size() * sizeof(float), hipMemcpyHostToDevice));
    std::vector<int> ints(bm_start);
    ints.insert(ints.end(), bs_start.begin(), bs_start.end());
    HIP_TRY(e, hipMalloc((void**)&e->d_ints, ints.size() * sizeof(int)));
    HIP_TRY(e, hipMalloc((void**)&e->d_clk, kClkSlots * sizeof(long long)));
    HIP_TRY(e, hipMemset(e->d_clk, 0, kClkSlots * sizeof(long long)));
    HIP_TRY(e, hipMemcpy(e->d_ints, ints.data(), ints.size() * sizeof(int), hipMemcpyHostToDevice));
    const float* W = e->d_weights;
    e->tabs.win = W + o_win;
    e->tabs.tw256 = reinterpret_cast<const float2*>(W + o_tw256);
    e->tabs.tw512 = reinterpret_cast<const float2*>(W + o_tw512);
    e->tabs.win_sum = W + o_ws;
    e->erb_bm = BandTab{e->d_ints, W + o_bm, bm_count, kErbBands};
    e->erb_bs = BandTab{e->d_ints + kErbBands, W + o_bs, bs_count, kErbHigh};
    e->en0 = ConvW{W + o_w0, W + o_b0, a0[0]};
    e->en1 = ConvW{W + o_w1, W + o_b1, a1[0]};
    e->de3 = ConvW{W + o_w3, W + o_b3, a3[0]};
    e->de4 = ConvW{W + o_w4, W + o_b4, 0.0f};
    for (int i = 0; i < 3; ++i) {
        const GtOff* src[2] = {&gte[i], &gtd[i]};
        GtConvW* dst[2] = {&e->en_gt[i], &e->de_gt[i]};
        for (int k = 0; k < 2; ++k) {
            const GtOff& o = *src[k];
            *dst[k] = GtConvW{W + o.pw1, W + o.pw1_b, W + o.dw, W + o.dw_b, W + o.pw2, W + o.pw2_b, W + o.gru, W + o.fc,
                              o.s1, o.s2, k == 0 ? en_dil[i] : de_dil[i], W + o.tra_rot};
        }
    }
    for (int i = 0; i < 2; ++i) {
        const DpOff& o = dpo[i];
        e->dp[i] = DpW{W + o.intra_gru, W + o.inter_gru, W + o.fc[0], W + o.fc_b[0], W + o.ln_w[0], W + o.ln_b[0],
                       W + o.fc[1], W + o.fc_b[1], W + o.ln_w[1], W + o.ln_b[1], W + o.inter_rot};
    }
    return ADE_OK;
}

void free_graphs(ade_engine* e) {
    for (auto& g : e->graphs) {
        if (g.exec) hipGraphExecDestroy(g.exec);
        if (g.graph) hipGraphDestroy(g.graph);
    }
    e->graphs.clear();
}

void free_exchange(ade_engine* e) {
    if (e->d_xchg) hipFree(e->d_xchg);
    if (e->d_xflags) hipFree(e->d_xflags);
    e->d_xchg = nullptr; e->d_xflags = nullptr;
    e->xchg_capacity = 0; e->xchg_segments = 0;
}

void free_workspace(ade_engine* e) {
    free_graphs(e);
    if (e->ws) hipFree(e->ws);
    if (e->d_pcm_in) hipFree(e->d_pcm_in);
    if (e->d_pcm_out) hipFree(e->d_pcm_out);
    if (e->d_f32_out) hipFree(e->d_f32_out);
    if (e->h_pcm_in) hipHostFree(e->h_pcm_in);
    if (e->h_pcm_out) hipHostFree(e->h_pcm_out);
    if (e->h_f32_out) hipHostFree(e->h_f32_out);
    if (e->rs_in) hipFree(e->rs_in);
    if (e->rs_out) hipFree(e->rs_out);
    if (e->d_fixed) hipFree(e->d_fixed);
    e->d_fixed = nullptr;
    free_exchange(e);
    for (float** p : {&e->gt_tmp, &e->gt_in, &e->gt_wave, &e->gt_mean, &e->d_f32_in}) { if (*p) hipFree(*p); *p = nullptr; }
    for (uint16_t** p : {&e->d_f16_in, &e->d_f16_out}) { if (*p) hipFree(*p); *p = nullptr; }
    for (uint16_t** p : {&e->h_f16_in, &e->h_f16_out}) { if (*p) hipHostFree(*p); *p = nullptr; }
    e->f16_capacity = 0;
    e->rs_in = e->rs_out = nullptr;
    e->ws = nullptr;
    e->d_pcm_in = e->d_pcm_out = nullptr;
    e->d_f32_out = nullptr;
    e->h_pcm_in = e->h_pcm_out = nullptr;
    e->h_f32_out = nullptr;
    e->capacity = 0;
}

int pick_geometry(const ade_engine* e, int B);

// The segment exchange area of the fused path (ade_internal.h: kX*), sized for the geometry the next call will actually take: nothing for the sandwich exports and for
// chunks one workgroup walks alone (no slot is ever touched there), [capacity][segments] slots otherwise (132 KB each: 135 MB at 256 x 1 s in four segments).
ade_status ensure_exchange(ade_engine* e) {
    if (e->sub) return ADE_OK;
    if (!e->d_xerr) {
        HIP_TRY(e, hipHostMalloc((void**)&e->d_xerr, kXErrWords * sizeof(int), hipHostMallocDefault));   // page-locked, device-visible: the host reads it after a synchronise
        for (int k = 0; k < kXErrWords; ++k) e->d_xerr[k] = 0;
    }
    int nseg = 0;
    if (e->use_fused && !e->gt_sand) {
        const int g = pick_geometry(e, e->capacity);
        if (g >= 0) nseg = fused_segments(e->T, g);
    }
    if (nseg <= 1) nseg = 0;
    if (nseg == e->xchg_segments && (nseg == 0 || e->xchg_capacity >= e->capacity)) return ADE_OK;
    // a call that needs NO slots (or fewer segments) keeps a larger area that is already there: ade_process's retry flips the geometry to 0 and back, and freeing /
    // re-allocating 135 MB with two device synchronisations per retried call is exactly what a pre-empted GPU does not need.  The flags are zero between launches.
    if (e->d_xchg && nseg <= e->xchg_segments && e->xchg_capacity >= e->capacity) return ADE_OK;
    if (e->stream) HIP_TRY(e, hipStreamSynchronize(e->stream));
    free_graphs(e);
    free_exchange(e);
    if (nseg) {
        const size_t B = (size_t)e->capacity;
        HIP_TRY(e, hipMalloc((void**)&e->d_xchg, B * nseg * (size_t)kXFloats * sizeof(float)));
        HIP_TRY(e, hipMalloc((void**)&e->d_xflags, B * nseg * (size_t)kXFlags * sizeof(unsigned)));
        HIP_TRY(e, hipMemset(e->d_xflags, 0, B * nseg * (size_t)kXFlags * sizeof(unsigned)));
        e->xchg_capacity = e->capacity;
    }
    e->xchg_segments = nseg;
    return ADE_OK;
}

ade_status reserve(ade_engine* e, int batch) {
    if (batch <= e->capacity) return ensure_exchange(e);
    HIP_TRY(e, hipSetDevice(e->device));
    if (e->stream) HIP_TRY(e, hipStreamSynchronize(e->stream));
    free_workspace(e);
    const size_t B = (size_t)batch, T = (size_t)e->T;
    if (e->sub) {   // the sub-engine owns its activations; only the I/O staging of ade_process lives here
        std::string derr;
        const int rc = e->sub->reserve(batch, derr);
        if (rc != ADE_OK) return fail(e, (ade_status)rc, derr);
        HIP_TRY(e, hipMalloc((void**)&e->d_pcm_in, B * e->in_len * sizeof(int16_t)));
        HIP_TRY(e, hipMalloc((void**)&e->d_pcm_out, B * e->out_len * sizeof(int16_t)));
        HIP_TRY(e, hipMalloc((void**)&e->d_f32_out, B * e->out_len * sizeof(float)));
        HIP_TRY(e, hipHostMalloc((void**)&e->h_pcm_in, B * e->in_len * sizeof(int16_t), hipHostMallocDefault));
        HIP_TRY(e, hipHostMalloc((void**)&e->h_pcm_out, B * e->out_len * sizeof(int16_t), hipHostMallocDefault));
        HIP_TRY(e, hipHostMalloc((void**)&e->h_f32_out, B * e->out_len * sizeof(float), hipHostMallocDefault));
        if (e->resample) {
            HIP_TRY(e, hipMalloc((void**)&e->rs_in, B * e->channels * e->rs_model_in * sizeof(float)));
            HIP_TRY(e, hipMalloc((void**)&e->rs_out, B * e->out_channels * e->n_outputs * e->rs_model_out * sizeof(float)));
        }
        if (e->gt_float_in) HIP_TRY(e, hipMalloc((void**)&e->d_f32_in, B * (size_t)e->in_len * sizeof(float)));
        e->capacity = batch;
        return ADE_OK;
    }
    const size_t nfr = B * T;
    struct Carve { float** p; size_t n; };
    std::vector<Carve> cs = {
        {&e->mean, B}, {&e->spec, nfr * 2 * kBinsPad}, {&e->feat, nfr * 3 * kErbPad}, {&e->e0, nfr * kF1 * kCh},
        {&e->e1, nfr * kFw * kCh}, {&e->h, nfr * kFw * kCh}, {&e->zt, nfr * 8}, {&e->rnn, nfr * kFw * kCh},
        {&e->d3, nfr * kF1 * kCh}, {&e->mask, nfr * 2 * kErbPad}, {&e->frames, nfr * kNfft}};
    for (int i = 0; i < 3; ++i) {
        cs.push_back({&e->xe[i], nfr * kFw * kCh});
        cs.push_back({&e->ate[i], nfr * 8});
        cs.push_back({&e->xd[i], nfr * kFw * kCh});
        cs.push_back({&e->atd[i], nfr * 8});
    }
    for (int i = 0; i < 2; ++i) {
        cs.push_back({&e->dpm[i], nfr * kFw * kCh});
        cs.push_back({&e->dpo[i], nfr * kFw * kCh});
    }
    size_t total = 0;
    for (auto& c : cs) total += (c.n + 63) & ~(size_t)63;
    HIP_TRY(e, hipMalloc((void**)&e->ws, total * sizeof(float)));
    size_t off = 0;
    for (auto& c : cs) {
        *c.p = e->ws + off;
        off += (c.n + 63) & ~(size_t)63;
    }
    HIP_TRY(e, hipMalloc((void**)&e->d_pcm_in, B * e->in_len * sizeof(int16_t)));
    HIP_TRY(e, hipMalloc((void**)&e->d_pcm_out, B * e->out_len * sizeof(int16_t)));
    HIP_TRY(e, hipMalloc((void**)&e->d_f32_out, B * e->out_len * sizeof(float)));
    HIP_TRY(e, hipHostMalloc((void**)&e->h_pcm_in, B * e->in_len * sizeof(int16_t), hipHostMallocDefault));
    HIP_TRY(e, hipHostMalloc((void**)&e->h_pcm_out, B * e->out_len * sizeof(int16_t), hipHostMallocDefault));
    HIP_TRY(e, hipHostMalloc((void**)&e->h_f32_out, B * e->out_len * sizeof(float), hipHostMallocDefault));
    {   // the fused path's device-resident argument block (the segment exchange area: ensure_exchange)
        ChunkFixed F{};
        F.tabs = e->tabs; F.erb_bm = e->erb_bm; F.erb_bs = e->erb_bs;
        F.en0 = e->en0; F.en1 = e->en1; F.de3 = e->de3; F.de4 = e->de4;
        for (int i = 0; i < 3; ++i) { F.en_gt[i] = e->en_gt[i]; F.de_gt[i] = e->de_gt[i]; F.xe[i] = e->xe[i]; F.xd[i] = e->xd[i]; }
        for (int i = 0; i < 2; ++i) { F.dp[i] = e->dp[i]; F.dpo[i] = e->dpo[i]; }
        F.spec = e->spec; F.e0 = e->e0; F.e1 = e->e1;
        HIP_TRY(e, hipMalloc((void**)&e->d_fixed, sizeof(ChunkFixed)));
        HIP_TRY(e, hipMemcpy(e->d_fixed, &F, sizeof(ChunkFixed), hipMemcpyHostToDevice));
    }
    if (e->gt_sand) {
        HIP_TRY(e, hipMalloc((void**)&e->gt_tmp, B * (size_t)e->gt_l1 * sizeof(float)));
        HIP_TRY(e, hipMalloc((void**)&e->gt_in, B * (size_t)e->gt_lm * sizeof(float)));
        HIP_TRY(e, hipMalloc((void**)&e->gt_wave, B * (size_t)e->gt_keep * sizeof(float)));
        HIP_TRY(e, hipMalloc((void**)&e->gt_mean, B * sizeof(float)));
        if (e->gt_float_in) HIP_TRY(e, hipMalloc((void**)&e->d_f32_in, B * (size_t)e->in_len * sizeof(float)));
    }
    e->capacity = batch;
    return ensure_exchange(e);
}

// ---- the launch sequence -------------------------------------------------------------------------------
struct Seq {
    ade_engine* e;
    hipStream_t s;
    bool prof;
    int cursor = 0;
    void begin(const char* name) {
        if (!prof) return;
        int si = -1;
        for (size_t i = 0; i < e->stats.size(); ++i)
            if (e->stats[i].name == name) si = (int)i;
        if (si < 0) { e->stats.push_back(KernelStat{name, 0.f, 0}); si = (int)e->stats.size() - 1; }
        if ((int)e->events.size() <= cursor) {
            hipEvent_t a, b;
            hipEventCreate(&a);
            hipEventCreate(&b);
            e->events.push_back({a, b});
            e->event_stat.push_back(si);
        }
        e->event_stat[cursor] = si;
        hipEventRecord(e->events[cursor].first, s);
    }
    void end() {
        if (!prof) return;
        hipEventRecord(e->events[cursor].second, s);
        ++cursor;
        e->events_used = cursor;
    }
};

// Fused-path geometry of a call (-1: the frame count fits none).  Several small workgroups per CU overlap one workgroup's serial phases (TRA recurrence on one
// wavefront, the GRUs) with the others' parallel ones: measured on one box at 256 x 1 s, 0.451 ms (geometry 0: one 1024-thread workgroup per CU), 0.418 ms
// (geometry 1: two of 512 threads) and 0.408 ms (geometry 2: four of 256), identical bits.  The finest split the frame count allows wins; the option
// "geometry" pins one.
int pick_geometry(const ade_engine* e, int /*B*/) {
    if (e->geometry >= 0) return fused_supported(e->T, e->geometry) ? e->geometry : -1;
    for (int g = fused_geometries() - 1; g >= 0; --g)
        if (fused_supported(e->T, g)) return g;
    return -1;
}

void enqueue(ade_engine* e, hipStream_t s, const int16_t* d_in, int B, int16_t* d_out, float* d_f32, bool prof) {
    const int T = e->T, nfr = B * T;
    Seq q{e, s, prof};
    const View none{nullptr, nullptr};
    const int geo = pick_geometry(e, B);
    const bool fused = e->use_fused && geo >= 0 && !e->gt_sand;      // the sandwich (float audio / other rates / dynamic length) takes the multi-kernel sequence
    e->last_fused = fused;
    View x{e->e1, nullptr};
    if (fused) {
        // ---- fused path: one workgroup per chunk SEGMENT per stage (ade_internal.h: geometries), activations LDS-resident, inter-stage
        //      tensors channel-quad planar in HBM, TRA gates applied inside the stage (every tensor is plain).  1 launch, or 10.
        SegPlan plan{};
        plan.nseg = fused_segments(T, geo); plan.xchg = e->d_xchg; plan.flags = e->d_xflags; plan.err = e->d_xerr; plan.wave_swap = e->wave_swap;
        plan.prio = e->seg_prio;
        plan.withhold = e->xchg_withhold;
        plan.wait_ticks = e->xwait_ticks;
        e->last_geometry = geo;
        long long* clk = prof ? e->d_clk : nullptr;
        if (clk) (void)hipMemsetAsync(e->d_clk, 0, kClkSlots * sizeof(long long), s);   // phase accumulators start from zero
        const float* dc = nullptr;
        if (e->n_win > 1) {   // batch-fold: one DC mean per call, shared by its windows
            q.begin("pcm_mean"); launch_pcm_mean(s, d_in, B, e->in_len, e->mean, e->n_win); q.end();
            dc = e->mean;
        }
        if (e->use_single && (!prof || e->profile_mode >= 2)) {
            ChunkCall C{};
            C.fixed = e->d_fixed;
            C.dc = dc;
            // geometry 0, measured: the stagger pays from ~three quarters of a chunk per CU (B = 128: +4.6 %, 256: -4.3 %, 512 / 1024: -2.8 %)
            C.stagger = (geo == 0 && B >= 192) ? e->stagger_ticks : 0;
            C.pcm_in = d_in; C.pcm_out = d_out; C.f32_out = d_f32; C.L = e->in_len; C.T = T; C.B = B;
            C.plan = plan;
            C.full_taps = e->full_taps;
            C.clk = (prof && e->profile_mode == 3) ? e->d_clk : nullptr;   // mode 3: the phase-clock build of the same kernel
            q.begin("gtcrn_chunk"); launch_gtcrn_chunk(s, geo, C); q.end();
            return;
        }
        q.begin("front"); launch_front(s, geo, plan, d_in, B, e->in_len, T, e->tabs, e->erb_bm, e->en0, e->en1, e->spec, e->e0, e->e1, clk, dc); q.end();
        for (int i = 0; i < 3; ++i) {
            q.begin("gtblock"); launch_gtblock(s, geo, plan, i, x.x, nullptr, e->en_gt[i], e->xe[i], B, T, (prof && i == 0) ? e->d_clk : nullptr); q.end();
            x = View{e->xe[i], nullptr};
        }
        for (int i = 0; i < 2; ++i) {
            q.begin("dpgrnn"); launch_dpgrnn(s, geo, plan, i, x.x, e->dp[i], e->dpo[i], B, T, (prof && i == 0) ? e->d_clk : nullptr); q.end();
            x = View{e->dpo[i], nullptr};
        }
        for (int i = 0; i < 3; ++i) {
            q.begin("gtblock"); launch_gtblock(s, geo, plan, 3 + i, x.x, e->xe[2 - i], e->de_gt[i], e->xd[i], B, T, nullptr); q.end();
            x = View{e->xd[i], nullptr};
        }
        q.begin("back"); launch_back(s, geo, plan, x.x, e->e1, e->e0, e->spec, e->de3, e->de4, e->erb_bs, e->tabs, d_out, d_f32, B, T, clk); q.end();
        return;
    }
    // ---- multi-kernel path (any T): channels-last tensors, deferred TRA gates (View)
    if (e->gt_sand) {   // GTCRN_CUSTOM's input sandwich -> the final fp32 waveform at the model rate, which the STFT reads as it is   (Export_GTCRN.py:636-655)
        q.begin("sandwich_in");
        // (batch-fold: one mean per CALL of n_win windows, :645-660; a call's windows are contiguous, so the stage runs on B / n_win rows of n_win * W samples)
        launch_gt_sandwich_in(s, e->cur_fin ? nullptr : d_in, e->cur_fin, B / e->n_win, e->in_len * e->n_win, e->gt_l1 * e->n_win, e->gt_lm * e->n_win, e->gt_lerp1,
                              e->gt_lerp2, e->gt_gain, e->gt_tmp, e->gt_mean, e->gt_in);
        q.end();
        q.begin("stft_feat"); launch_stft_pcm(s, nullptr, nullptr, B, e->gt_lm, T, e->tabs, e->erb_bm, e->spec, e->feat, true, e->gt_in); q.end();
    } else {
        q.begin("pcm_mean"); launch_pcm_mean(s, d_in, B, e->in_len, e->mean, e->n_win); q.end();
        q.begin("stft_feat"); launch_stft_pcm(s, d_in, e->mean, B, e->in_len, T, e->tabs, e->erb_bm, e->spec, e->feat); q.end();
    }
    q.begin("conv0"); launch_conv0(s, e->feat, e->en0, e->e0, nfr); q.end();
    q.begin("conv1"); launch_conv1(s, e->e0, e->en1, e->e1, nfr); q.end();
    for (int i = 0; i < 3; ++i) {   // Encoder GTConvBlocks (Export_GTCRN.py:502-504)
        q.begin("gt_pw1"); launch_gt_pw1(s, x, none, e->en_gt[i], e->h, nfr); q.end();
        q.begin("gt_dw_pw2"); launch_gt_dw_pw2(s, e->h, x, none, e->en_gt[i], e->xe[i], e->zt, B, T); q.end();
        q.begin("tra_gru"); launch_tra(s, e->zt, e->en_gt[i], e->ate[i], B, T); q.end();
        x = View{e->xe[i], e->ate[i]};
    }
    for (int i = 0; i < 2; ++i) {   // DPGRNN x2 (Export_GTCRN.py:577-578)
        q.begin("intra_gru"); launch_intra_gru(s, x, e->dp[i].intra_gru, e->rnn, nfr); q.end();
        q.begin("fc_ln_res"); launch_fc_ln_res(s, e->rnn, x, e->dp[i].intra_fc, e->dp[i].intra_fc_b, e->dp[i].intra_ln_w,
                                               e->dp[i].intra_ln_b, e->dpm[i], B, T); q.end();
        q.begin("inter_gru"); launch_inter_gru(s, e->dpm[i], e->dp[i].inter_gru, e->rnn, B, T, nullptr, e->dp[i].inter_rot); q.end();
        q.begin("fc_ln_res"); launch_fc_ln_res(s, e->rnn, View{e->dpm[i], nullptr}, e->dp[i].inter_fc, e->dp[i].inter_fc_b,
                                               e->dp[i].inter_ln_w, e->dp[i].inter_ln_b, e->dpo[i], B, T); q.end();
        x = View{e->dpo[i], nullptr};
    }
    for (int i = 0; i < 3; ++i) {   // Decoder GTConvBlocks on x + en_outs[4-i] (Export_GTCRN.py:524-526)
        const View skip{e->xe[2 - i], e->ate[2 - i]};
        q.begin("gt_pw1"); launch_gt_pw1(s, x, skip, e->de_gt[i], e->h, nfr); q.end();
        q.begin("gt_dw_pw2"); launch_gt_dw_pw2(s, e->h, x, skip, e->de_gt[i], e->xd[i], e->zt, B, T); q.end();
        q.begin("tra_gru"); launch_tra(s, e->zt, e->de_gt[i], e->atd[i], B, T); q.end();
        x = View{e->xd[i], e->atd[i]};
    }
    q.begin("deconv3"); launch_deconv3(s, x, View{e->e1, nullptr}, e->de3, e->d3, nfr); q.end();
    q.begin("deconv4"); launch_deconv4(s, e->d3, e->e0, e->de4, e->mask, nfr); q.end();
    q.begin("istft_mask"); launch_istft_masked(s, e->spec, e->mask, e->erb_bs, e->tabs, e->frames, nfr); q.end();
    if (e->gt_sand) {   // overlap-add (static or dynamic-length trim) and the output sandwich   (STFT_Process.py:326-341, Export_GTCRN.py:673-693)
        q.begin("sandwich_out");
        launch_gt_sandwich_out(s, e->frames, e->tabs, B, T, e->gt_keep, e->gt_wave, d_out, d_f32, e->out_len, e->gt_lerp_out, e->gt_scale_first);
        q.end();
        return;
    }
    q.begin("ola_pcm"); launch_ola_pcm(s, e->frames, e->tabs, B, T, d_out, d_f32); q.end();
}

ade_status run(ade_engine* e, hipStream_t s, const int16_t* d_in, int B, int16_t* d_out, float* d_f32) {
    if (B == 0) return ADE_OK;
    e->last_batch = B;
    std::string sub_err;
    int sub_rc = ADE_OK;
    // the launch sequence of one call: a sub-engine's (hundreds of GEMM / row kernels) or GTCRN's
    auto launch_all = [&](bool timed) {
        if (e->sub && e->resample) {   // interpolate to the model rate, run on floats, interpolate the float waveform back and apply the PCM tail
            const long long rows_in = (long long)B * e->channels, rows_out = (long long)B * e->out_channels * e->n_outputs;
            if (e->cur_fin) launch_resample_in_f32(s, e->cur_fin, e->rs_in, rows_in, e->in_len / e->channels, e->rs_model_in, e->rs_scale_in, e->rs_in_gain);
            else launch_resample_in(s, d_in, e->rs_in, rows_in, e->in_len / e->channels, e->rs_model_in, e->rs_scale_in);
            e->sub->float_in = e->rs_in;
            e->sub->float_src = e->cur_fin;
            e->sub->float_src_gain = e->rs_in_gain;
            sub_rc = e->sub->run(s, d_in, B, nullptr, e->rs_out, sub_err);
            e->sub->float_in = nullptr;
            e->sub->float_src = nullptr;
            if (e->rs_sandwich_out)
                launch_gt_out(s, e->rs_out, d_out, d_f32, rows_out, e->rs_model_out, e->out_len / (e->out_channels * e->n_outputs), e->rs_scale_out, e->rs_scale_first,
                              e->rs_nan_to_num);
            else
                launch_resample_out(s, e->rs_out, d_out, d_f32, rows_out, e->rs_model_out, e->out_len / (e->out_channels * e->n_outputs), e->rs_scale_out,
                                    e->rs_pcm_scale, e->rs_truncate_i32, e->rs_f32_scale, e->rs_nan_to_num);
        } else if (e->sub) sub_rc = e->sub->run(s, d_in, B, d_out, d_f32, sub_err);
        else enqueue(e, s, d_in, B, d_out, d_f32, timed);
    };
    if (e->sub && (e->profile || !e->use_graph || !e->graph_supported)) {
        launch_all(false);
        return sub_rc == ADE_OK ? ADE_OK : fail(e, (ade_status)sub_rc, sub_err);
    }
    if (e->profile) {
        for (auto& st : e->stats) { st.ms = 0.f; st.launches = 0; }
        e->events_used = 0;
        enqueue(e, s, d_in, B, d_out, d_f32, true);
        HIP_TRY(e, hipStreamSynchronize(s));
        for (size_t i = 0; i < (size_t)e->events_used; ++i) {   // only the events recorded by THIS run
            float ms = 0.f;
            hipEventElapsedTime(&ms, e->events[i].first, e->events[i].second);
            e->stats[e->event_stat[i]].ms += ms;
            e->stats[e->event_stat[i]].launches += 1;
        }
        return ADE_OK;
    }
    // A captured graph pays off for the multi-kernel launch sequences (10 or 34 launches).  The single-launch path is one
    // kernel: a plain launch has less per-step overhead than a one-node graph (measured: 0.462 vs 0.475 ms per step).
    // Sub-engines enqueue 300 - 800 launches per call; the captured sequence is replayed so that the host cost of issuing them does not
    // depend on the caller's CPU (their workspace is reserved before capture; reserve() drops the graphs when it reallocates).  On the
    // bench host the replay measured the same as plain launches (MossFormer2, 1 window: 26.9 ms both ways): small batches are bound by
    // GPU-side kernel latency and by the few single-workgroup reductions, not by the host.
    const bool one_kernel = !e->sub && e->use_fused && e->use_single && pick_geometry(e, B) >= 0 && !e->gt_sand;
    if (e->use_graph && e->graph_supported && !one_kernel) {
        GraphEntry* hit = nullptr;
        for (auto& g : e->graphs)
            if (g.in == d_in && g.out_pcm == d_out && g.out_f32 == d_f32 && g.batch == B) hit = &g;
        if (!hit) {
            if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess) {
                launch_all(false);
                hipGraph_t graph = nullptr;
                hipGraphExec_t exec = nullptr;
                if (hipStreamEndCapture(s, &graph) == hipSuccess && graph && sub_rc == ADE_OK &&
                    hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess) {
                    if (e->graphs.size() >= 8) {
                        hipGraphExecDestroy(e->graphs.front().exec);
                        hipGraphDestroy(e->graphs.front().graph);
                        e->graphs.erase(e->graphs.begin());
                    }
                    e->graphs.push_back(GraphEntry{d_in, d_out, d_f32, B, exec, graph});
                    hit = &e->graphs.back();
                } else {
                    if (graph) hipGraphDestroy(graph);
                    e->graph_supported = false;
                }
            } else {
                (void)hipGetLastError();
                e->graph_supported = false;
            }
        }
        if (hit) {
            HIP_TRY(e, hipGraphLaunch(hit->exec, s));
            return ADE_OK;
        }
    }
    launch_all(false);
    if (sub_rc != ADE_OK) return fail(e, (ade_status)sub_rc, sub_err);
    HIP_TRY(e, hipGetLastError());
    return ADE_OK;
}

// The segmented fused path's bounded waits (dev::xwait): a wait that gave up left its dev::xcode() in the page-locked error word and the launch's output is not
// to be trusted.  Every entry point calls this (a) on entry -- launches on a CALLER's stream are not synchronised by the engine, so a failure of an earlier call is
// reported by the next one -- and (b) after it has synchronised its own launch, BEFORE any PCM is handed out.  On a failure the device is drained (the waiting
// workgroups of the failed launch run on for up to their own bounds), every flag is lowered -- a late producer may have raised one that nobody consumed -- and the
// call fails with ADE_ERR_DEVICE naming the chunk, the segment and the hand-off that did not arrive.
const char* xflag_name(int idx) {
    return idx < kXFlagTra ? "depthwise-convolution history" : (idx < kXFlagInter ? "TRA state" : (idx < kXFlagOla ? "inter-frame GRU state" : "overlap-add carry"));
}
ade_status exchange_status(ade_engine* h, const char* who, bool earlier) {
    if (h->sub) {        // a sub-engine's own bounded hand-offs (H-GTCRN's fused network middle: SubEngine::exchange_error_and_reset); checked here, outside the captured graph
        const int code = h->sub->exchange_error_and_reset();
        if (code) {
            h->timed_out = true;
            char msg[320];
            snprintf(msg, sizeof msg, "%s: %s: a segment hand-off (%s) of the fused network stages timed out (%.1f ms)%s; the call's output is not valid "
                     "(ADE_HG_FUSED=0 runs the multi-kernel sequence, option xwait_ms raises the bound)", who, "sub-engine", xflag_name(code & 15), h->xwait_ticks * 1e-5,
                     earlier ? " in an earlier call on a caller-provided stream" : "");
            return fail(h, ADE_ERR_DEVICE, msg);
        }
    }
    if (!h->d_xerr) return ADE_OK;
    int code = 0, word = 0;
    for (int k = 0; k < kXErrWords && !code; ++k) { code = ((volatile int*)h->d_xerr)[k]; word = k; }
    if (!code) return ADE_OK;
    (void)hipDeviceSynchronize();
    for (int k = 0; k < kXErrWords; ++k) h->d_xerr[k] = 0;
    if (h->d_xflags) (void)hipMemset(h->d_xflags, 0, (size_t)h->xchg_capacity * h->xchg_segments * kXFlags * sizeof(unsigned));
    // the block index counts within the reporting launch: segment = block / (chunks of that launch), chunk = its first chunk + block % (chunks of that launch)
    const int block = (code >> 4) - 1, idx = code & 15, B = h->xl_B[word] > 0 ? h->xl_B[word] : (h->last_batch > 0 ? h->last_batch : 1), chunk0 = h->xl_B[word] > 0 ? h->xl_chunk0[word] : 0;
    h->timed_out = true;
    char msg[384];
    if (idx == 15)      // dev::wait_rows_in: a streamed host batch (ade_process) whose rows were not delivered in time
        snprintf(msg, sizeof msg, "%s: fused path%s: segment %d of chunk %d timed out (%.1f ms) waiting for its rows of the host batch to be copied in; no output was produced "
                 "(option host_stream=1 copies the batch before the launch, option xwait_ms raises the bound)",
                 who, earlier ? " (an earlier call on a caller-provided stream)" : "", block / B, chunk0 + block % B, h->xwait_ticks * 1e-5);
    else
        snprintf(msg, sizeof msg,
                 "%s: fused path%s: segment %d of chunk %d timed out (%.1f ms) waiting for the %s of its predecessor workgroup; no output was produced "
                 "(option geometry=0 runs whole chunks per workgroup, option xwait_ms raises the bound)",
                 who, earlier ? " (an earlier call on a caller-provided stream)" : "", block / B, chunk0 + block % B, h->xwait_ticks * 1e-5, xflag_name(idx));
    return fail(h, ADE_ERR_DEVICE, msg);
}

// ---- streaming: the multi-kernel launch sequence over pushes of N frames with carried state -------
}  // namespace

struct ade_stream {
    ade_engine* e = nullptr;     // nullptr once the engine has been destroyed (orphan: every call but destroy answers BAD_VALUE)
    int device = 0;
    int S = 0, N = 0;
    bool first = true, flushed = false;
    int16_t *pcm_prev = nullptr, *pcm_hist = nullptr, *concat = nullptr, *d_in = nullptr, *d_out = nullptr, *h_in = nullptr, *h_out = nullptr;
    float *d_f32 = nullptr, *h_f32 = nullptr;
    float* state = nullptr;      // one allocation: dc | conv histories (ping-pong) | TRA hidden | inter-GRU hidden | OLA carry
    size_t state_floats = 0;
    float* dc = nullptr;
    float* hist[6][2] = {};
    int hist_cur[6] = {};
    float* tra_h[6] = {};
    float* inter_h[2] = {};
    float* carry = nullptr;
    // fused pushes (the single-launch chunk kernel with a carried state, ade_internal.h SegPlan::carry_in / carry_out): decided when the stream is created
    bool fused = false;
    int geo = -1, geo_flush = -1;
    ade::ChunkFixed* d_fixed = nullptr;   // the chunk kernel's per-engine block with THIS stream's push workspace
    float* xq = nullptr;                  // exchange slots between the segments of one push [S][segments][kXFloats]
    unsigned* xq_flags = nullptr;
    float* xstate[2] = {};                // the state a push leaves for the next one, ping-ponged [S][kXFloats]
    unsigned* xstate_flags[2] = {};
    int xcur = 0;
    int* xerr = nullptr;                  // page-locked: a bounded inter-workgroup wait gave up
    float* ws = nullptr;         // activations of one push, same tensor set as the engine's multi-kernel workspace
    float *spec = nullptr, *feat = nullptr, *e0 = nullptr, *e1 = nullptr, *h = nullptr, *zt = nullptr, *rnn = nullptr, *d3 = nullptr, *mask = nullptr,
          *frames = nullptr, *xe[3] = {}, *ate[3] = {}, *xd[3] = {}, *atd[3] = {}, *dpm[2] = {}, *dpo[2] = {};
};

namespace {

// flush = true: the single frame past the end of the signal (its second half is the end reflection); produces the stream's last hop
void enqueue_stream(ade_stream* st, hipStream_t s, const int16_t* d_in, int16_t* d_out, float* d_f32, bool flush = false) {
    ade_engine* e = st->e;
    const int B = st->S, T = flush ? 1 : st->N, nfr = B * T, P = T * kHop;
    if (flush) launch_stream_concat_flush(s, st->pcm_hist, st->pcm_prev, st->concat, B);
    else {
        launch_stream_concat(s, st->pcm_hist, d_in, st->concat, B, P, st->first);
        launch_stream_keep(s, st->concat, st->pcm_hist, st->pcm_prev, B, P);
    }
    if (st->fused) {
        // ONE launch for the whole push: the chunk kernel on rows [256 carried samples | the push], its first segment continuing from the exchange slot the previous
        // push's last segment filled (conv partial sums, TRA / inter-GRU states, overlap-add carry), its last segment filling the other slot for the next push.
        const int geo = flush ? st->geo_flush : st->geo;
        ChunkCall C{};
        C.fixed = st->d_fixed;
        C.pcm_in = st->concat; C.pcm_out = d_out; C.f32_out = d_f32; C.dc = st->dc;
        C.L = P + kHop; C.T = T; C.B = B;
        SegPlan plan{};
        plan.nseg = fused_segments(T, geo);
        plan.xchg = st->xq; plan.flags = st->xq_flags; plan.err = st->xerr;
        plan.wait_ticks = e->xwait_ticks;
        plan.carry_in = st->first ? nullptr : st->xstate[st->xcur];
        plan.carry_in_flags = st->first ? nullptr : st->xstate_flags[st->xcur];
        plan.carry_out = st->xstate[st->xcur ^ 1];
        plan.carry_out_flags = st->xstate_flags[st->xcur ^ 1];
        plan.stream = 1;
        C.plan = plan;
        launch_gtcrn_chunk(s, geo, C);
        st->xcur ^= 1;
        st->first = false;
        return;
    }
    launch_stft_pcm(s, st->concat, st->dc, B, P + kHop, T, e->tabs, e->erb_bm, st->spec, st->feat, /*center=*/false);
    launch_conv0(s, st->feat, e->en0, st->e0, nfr);
    launch_conv1(s, st->e0, e->en1, st->e1, nfr);
    const View none{nullptr, nullptr};
    View x{st->e1, nullptr};
    auto gt_block = [&](int gi, const GtConvW& w, View in, View skip, float* xn, float* at) {
        launch_gt_pw1(s, in, skip, w, st->h, nfr);
        const int cur = st->hist_cur[gi];
        launch_gt_dw_pw2(s, st->h, in, skip, w, xn, st->zt, B, T, st->hist[gi][cur]);
        launch_hist_shift(s, st->hist[gi][cur], st->h, st->hist[gi][cur ^ 1], B, T, 2 * w.dilation);
        st->hist_cur[gi] = cur ^ 1;
        launch_tra(s, st->zt, w, at, B, T, st->tra_h[gi]);
    };
    for (int i = 0; i < 3; ++i) {
        gt_block(i, e->en_gt[i], x, none, st->xe[i], st->ate[i]);
        x = View{st->xe[i], st->ate[i]};
    }
    for (int i = 0; i < 2; ++i) {
        launch_intra_gru(s, x, e->dp[i].intra_gru, st->rnn, nfr);
        launch_fc_ln_res(s, st->rnn, x, e->dp[i].intra_fc, e->dp[i].intra_fc_b, e->dp[i].intra_ln_w, e->dp[i].intra_ln_b, st->dpm[i], B, T);
        launch_inter_gru(s, st->dpm[i], e->dp[i].inter_gru, st->rnn, B, T, st->inter_h[i], e->dp[i].inter_rot);
        launch_fc_ln_res(s, st->rnn, View{st->dpm[i], nullptr}, e->dp[i].inter_fc, e->dp[i].inter_fc_b, e->dp[i].inter_ln_w, e->dp[i].inter_ln_b,
                         st->dpo[i], B, T);
        x = View{st->dpo[i], nullptr};
    }
    for (int i = 0; i < 3; ++i) {
        const View skip{st->xe[2 - i], st->ate[2 - i]};
        gt_block(3 + i, e->de_gt[i], x, skip, st->xd[i], st->atd[i]);
        x = View{st->xd[i], st->atd[i]};
    }
    launch_deconv3(s, x, View{st->e1, nullptr}, e->de3, st->d3, nfr);
    launch_deconv4(s, st->d3, st->e0, e->de4, st->mask, nfr);
    launch_istft_masked(s, st->spec, st->mask, e->erb_bs, e->tabs, st->frames, nfr);
    launch_ola_pcm_stream(s, st->frames, st->carry, e->tabs, B, T, st->first, d_out, d_f32);
    st->first = false;
}

}  // namespace

// ======================================= C ABI ================================================================
extern "C" {

ade_status ade_create(const char* manifest_json, const void* weights, size_t weights_nbytes, int device, ade_handle* out) {
    if (!out) return fail(nullptr, ADE_ERR_BAD_VALUE, "ade_create: out is NULL");
    *out = nullptr;
    if (!manifest_json) return fail(nullptr, ADE_ERR_NOT_FOUND, "ade_create: manifest is NULL (metadata carrier missing)");
    if (!weights) return fail(nullptr, ADE_ERR_NOT_FOUND, "ade_create: weights are NULL");
    ade_engine* e = new ade_engine();
    auto bail = [&](ade_status st) {
        g_create_error = e->last_error;
        ade_destroy(e);
        return st;
    };
    std::string err;
    if (!parse_manifest(manifest_json, e->meta, err)) return bail(fail(e, ADE_ERR_BAD_VALUE, err));
    // load_runtime_metadata: every required key present and non-empty (audio_onnx_metadata.py:251-256,300-302)
    for (const char* k : kRequiredKeys) {
        auto it = e->meta.find(k);
        if (it == e->meta.end() || it->second.empty())
            return bail(fail(e, ADE_ERR_MISSING_KEY, std::string("Required metadata key ") + k + " is missing."));
    }
    const bool fam_dfsmn = e->meta["model_family"] == "dfsmn", fam_melband = e->meta["model_family"] == "mel_band_roformer",
               fam_moss = e->meta["model_family"] == "mossformer2_ss", fam_ulu = e->meta["model_family"] == "ul_unas",
               fam_hg = e->meta["model_family"] == "h_gtcrn", fam_zip = e->meta["model_family"] == "zipenhancer";
    if (fam_dfsmn || fam_melband || fam_moss || fam_ulu || fam_hg || fam_zip) {   // DFSMN/Export_DFSMN.py (48 kHz mono) / Mel_Band_Roformer/Stereo/Export_MelBandRoformer.py (44.1 kHz stereo)
        const std::string fam = e->meta["model_family"];
        const long rate = fam_dfsmn ? 48000 : fam_melband ? 44100 : 16000;
        bool dyn_d = false, fold_d = false;
        if (!parse_bool(e->meta["dynamic_axes"], &dyn_d))
            return bail(fail(e, ADE_ERR_BAD_VALUE, "Metadata key dynamic_axes must be a boolean encoded as 1/0, got '" + e->meta["dynamic_axes"] + "'."));
        // A dynamic_axes export of Mel-Band-Roformer takes its frame count from the waveform and keeps everything after the first half window of the overlap-add
        // (Stereo/STFT_Process.py:296-306), UL-UNAS's slices that to the caller-rate input length (Export_UL_UNAS.py:851, 888-889); the engine still serves ONE input
        // length per handle.  The other families' dynamic exports are not built (their static exports resample consistently).
        const bool fam_sand = fam_melband || fam_ulu;       // families whose resampling follows GTCRN_CUSTOM's scale-factor sandwich and needs dynamic axes
        // DFSMN's dynamic export (Export_DFSMN.py:28, :186-187, :236-237, :274) differs from its static one in two places only: the edges interpolate by SCALE FACTOR
        // (floor(n * factor) samples, source step 1 / factor), and the ISTFT builds its overlap-add denominator from the actual frame count -- which IS the static
        // denominator of that count (no centre padding: nothing is trimmed).  Any input length of at least one frame.
        // ZipEnhancer's (Export_ZipEnhancer.py:31, :61, :828-829, :898-899, :907-908; STFT_Process.py:294-299): the same two places -- its ISTFT trims half a window on
        // both sides in either mode, so the output is still 100 (T - 1) samples, cut to the model-rate input length (a no-op), but DIVIDED by the denominator instead of
        // multiplied by its precomputed reciprocal; the position tables are slices of the same 1024-frame table either way.
        // MossFormer2-SS's (Export_MossFormer2_SS_16K.py:24, :183, :430, :500-501, :565-577, :634-646): scale-factor edges again, and 1 / frames is applied to the reduced
        // linear-attention product at run time instead of being folded into the OffsetScale row of the linear keys; tables are slices of the 6 s ones, the rotary
        // half-rotation is the same arithmetic in either form, any window of 16 + 8 k samples.
        const bool dyn_sf = dyn_d && (fam_dfsmn || fam_zip || fam_moss);
        // H-GTCRN's (Export_H_GTCRN.py:27, :1075, :1097, :1110): frame counts from the waveform and the ISTFT's dynamic trim -- half a window of tail is kept, L + 256 samples out;
        // its edges interpolate by scale factor in either mode.
        if (dyn_d && !fam_sand && !dyn_sf && !fam_hg)
            return bail(fail(e, ADE_ERR_UNSUPPORTED, "dynamic_axes=1 is not implemented for " + fam));
        if (e->meta.count("use_batch_fold") && !e->meta["use_batch_fold"].empty() && !parse_bool(e->meta["use_batch_fold"], &fold_d))
            return bail(fail(e, ADE_ERR_BAD_VALUE, "Metadata key use_batch_fold must be a boolean encoded as 1/0."));
        if (fold_d && fam_dfsmn) {   // a folded window must reconstruct itself: raw overlap-add length 1920 + 960 (T - 1) == W  (Export_DFSMN.py:54)
            long fw = 0;
            if (!e->meta.count("fold_window_length") || !parse_int(e->meta["fold_window_length"], &fw) || fw < 1920 || fw % 960)
                return bail(fail(e, ADE_ERR_BAD_VALUE, "dfsmn: use_batch_fold=1 needs fold_window_length: a multiple of the 960-sample hop, at least 1920"));
        }
        long sri = 0, sro = 0, srm = 0, Ld = 0;
        if (!parse_int(e->meta["in_sample_rate"], &sri) || !parse_int(e->meta["out_sample_rate"], &sro) ||
            !parse_int(e->meta["model_sample_rate"], &srm) || !parse_int(e->meta["input_audio_length"], &Ld))
            return bail(fail(e, ADE_ERR_BAD_VALUE, "manifest: sample rates / input_audio_length must be integers"));
        const bool rates_differ = sri != srm || sro != srm;
        if (srm != rate) return bail(fail(e, ADE_ERR_UNSUPPORTED, fam + " runs at a model rate of " + std::to_string(rate) + " Hz"));
        // Resampling edges exist where the reference's STATIC export is self-consistent: MossFormer2 and DFSMN size their frames from the
        // model-rate length (Export_MossFormer2_SS_16K.py:36-37,99-104; Export_DFSMN.py:48,67).  Mel-Band-Roformer (like GTCRN) and UL-UNAS size
        // the static frame count from the INPUT-rate length (Export_MelBandRoformer.py:52), which only agrees with the STFT at equal rates.
        // H-GTCRN's static export is consistent too (frames from MODEL_AUDIO_LENGTH, Export_H_GTCRN.py:45-46); it interpolates by SCALE FACTOR.
        // ZipEnhancer sizes its frames from MODEL_AUDIO_LENGTH too (Export_ZipEnhancer.py:55, 61) and interpolates by size (:826-832, :905-911).
        if (rates_differ && !fam_moss && !fam_dfsmn && !fam_hg && !fam_zip && !(fam_sand && dyn_d))
            return bail(fail(e, ADE_ERR_UNSUPPORTED, fam + " runs at " + std::to_string(rate) + " Hz in, model and out (its static export has no consistent resampling path" +
                                                     (fam_sand ? ": export with dynamic_axes=1 for other rates)" : ")")));
        if (dyn_d && fold_d) return bail(fail(e, ADE_ERR_BAD_VALUE, "Batch folding requires a static shape (dynamic_axes=0)."));     // (Export_MelBandRoformer.py:46)
        if (rates_differ && (sri < 1000 || sro < 1000 || sri > 384000 || sro > 384000)) return bail(fail(e, ADE_ERR_BAD_VALUE, "manifest: sample rates out of range"));
        // Float audio tensors (IN / OUT_AUDIO_DTYPE F32 / F16; F16 crosses the C ABI as fp32, the host layer converts): the graph is the same with the int16 scale steps
        // left out, so such handles run through the resampling edges below (identity where the rates agree) with the family's input gain / output scale.
        auto dtype_ok_d = [](const std::string& d) { return d == "INT16" || d == "F32" || d == "F16"; };
        if (!dtype_ok_d(e->meta["input_audio_dtype"]) || !dtype_ok_d(e->meta["output_audio_dtype"]))
            return bail(fail(e, ADE_ERR_BAD_VALUE, "manifest: input_audio_dtype / output_audio_dtype must be INT16, F32 or F16"));
        const bool float_in_d = e->meta["input_audio_dtype"] != "INT16", float_io = float_in_d || e->meta["output_audio_dtype"] != "INT16";
        if (Ld < 16 || Ld > (1 << 24)) return bail(fail(e, ADE_ERR_SHAPE_MISMATCH, "input_audio_length out of range"));
        long sub_win = 1;
        const long caller_len = Ld;
        if (rates_differ) {   // MODEL_AUDIO_LENGTH = round(L * model / in) (:36); batch-fold needs equal rates (:92-93)
            if (fold_d) return bail(fail(e, ADE_ERR_BAD_VALUE, "Batch folding requires equal input/model/output sample rates."));
            Ld = dyn_sf ? (long)floor((double)caller_len * ((double)srm / (double)sri))               // F.interpolate(scale_factor = float(MODEL / IN))
                 : fam_hg ? (long)((double)caller_len * (double)srm / (double)sri)                      // int(EXPORT_AUDIO_LENGTH * MODEL / IN) (:45)
                 : fam_melband ? (long)floor((double)caller_len * ((double)srm / (double)sri))       // F.interpolate(scale_factor = float(MODEL / IN)) (Export_MelBandRoformer.py:52, 631-644)
                 : fam_ulu ? (long)floor((double)caller_len * (1.0 / ((double)sri / 16000.0)))       // scale_factor = 1 / (in_sample_rate / 16000.0) (Export_UL_UNAS.py:835-837, 852-868)
                        : (long)nearbyint((double)caller_len * (double)srm / (double)sri)   /* Python round(): half to even */;
        }
        if (fold_d) {   // the graph input is ceil(L / W) whole windows of W model-rate samples, folded into the batch inside the model
            long fw = 0;    //                                                            (Export_MelBandRoformer.py:47-51, 644-647)
            if (!e->meta.count("fold_window_length") || !parse_int(e->meta["fold_window_length"], &fw) || fw <= 0)
                return bail(fail(e, ADE_ERR_BAD_VALUE, "use_batch_fold=1 needs fold_window_length"));
            const long want = (Ld + fw - 1) / fw * fw;
            long exp_len = 0;
            if (e->meta.count("export_audio_length") && !e->meta["export_audio_length"].empty() &&
                (!parse_int(e->meta["export_audio_length"], &exp_len) || exp_len != want))
                return bail(fail(e, ADE_ERR_SHAPE_MISMATCH, "export_audio_length is not input_audio_length rounded up to whole fold windows"));
            sub_win = want / fw;
            Ld = fw;
        }
        ade_status std_ = parse_blob(e, weights, weights_nbytes);
        if (std_ != ADE_OK) return bail(std_);
        int ndev_d = 0;
        if (hipGetDeviceCount(&ndev_d) != hipSuccess || ndev_d <= 0) {
            (void)hipGetLastError();
            return bail(fail(e, ADE_ERR_DEVICE, "no HIP device visible: libade has no CPU execution mode"));
        }
        if (device < 0 || device >= ndev_d) return bail(fail(e, ADE_ERR_DEVICE, "device ordinal out of range"));
        e->device = device;
        if (hipSetDevice(device) != hipSuccess) return bail(fail(e, ADE_ERR_DEVICE, "hipSetDevice failed"));
        if (hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess) return bail(fail(e, ADE_ERR_DEVICE, "hipStreamCreate failed"));
        std::string derr;
        bool exact_dft = false;
        if (e->meta.count("ade_dft_tables")) {
            if (e->meta["ade_dft_tables"] == "exact") exact_dft = true;
            else if (e->meta["ade_dft_tables"] != "reference") return bail(fail(e, ADE_ERR_BAD_VALUE, "manifest: ade_dft_tables must be 'reference' or 'exact'"));
        }
        // ade_gemm_dtype: "f32" (default: exact fp32 matrix-core products, the parity path)
        //               | "bf16" (mel_band_roformer, zipenhancer: bf16 activations and weights STORED in HBM, gfx950's full-rate bf16 matrix instructions -- csrc/ade_gemm16.h, ade_zip16.h).
        // (Rounds 2 - 3 had a "bf16" MODE for the other transformer families -- fp32 operands in HBM rounded on their way into LDS, the half-rate 16x16x16 instruction:
        //  1.5 - 1.8 x at 28 - 34 dB.  It was a rounding mode of the fp32 kernels, not a bf16 data path, and was removed in round 4 when the real path went in.)
        bool gemm_bf16 = false;
        if (e->meta.count("ade_gemm_dtype") && !e->meta["ade_gemm_dtype"].empty()) {
            const std::string& dt = e->meta["ade_gemm_dtype"];
            if (dt == "bf16") {
                if (!fam_melband && !fam_zip) return bail(fail(e, ADE_ERR_UNSUPPORTED, "ade_gemm_dtype = bf16 (bf16 stored in HBM) is implemented for mel_band_roformer and zipenhancer; the other families run f32"));
                gemm_bf16 = true;
            } else if (dt != "f32") {
                return bail(fail(e, ADE_ERR_BAD_VALUE, "manifest: ade_gemm_dtype must be 'f32' or 'bf16'"));
            }
        }
        const int rc = fam_dfsmn     ? ade::dfsmn_create(e->tensors, (int)Ld, (int)sub_win, device, &e->sub, derr)
                       : fam_melband ? ade::melband_create(e->tensors, (int)Ld, (int)sub_win, exact_dft, gemm_bf16, dyn_d, device, &e->sub, derr)
                       : fam_ulu     ? ade::ulunas_create(e->tensors, (int)Ld, (int)sub_win, dyn_d ? (int)caller_len : 0, device, &e->sub, derr)
                       : fam_hg      ? ade::hgtcrn_create(e->tensors, (int)Ld, (int)sub_win, dyn_d, device, &e->sub, derr)
                       : fam_zip     ? ade::zipenhancer_create(e->tensors, (int)Ld, (int)sub_win, exact_dft, gemm_bf16, dyn_d, device, &e->sub, derr)
                                     : ade::mossformer_create(e->tensors, (int)Ld, (int)sub_win, dyn_d, device, &e->sub, derr);
        if (rc != ADE_OK) return bail(fail(e, (ade_status)rc, derr));
        e->channels = e->sub->channels();
        e->out_channels = e->sub->out_channels();
        e->n_outputs = e->sub->n_outputs();
        e->in_len = e->sub->in_len() * e->channels;
        e->T = e->sub->frames();
        e->out_len = e->sub->out_len() * e->out_channels * e->n_outputs;
        if (float_io && !rates_differ) {   // the same edges with identity interpolation (source step 1: every sample is reproduced exactly)
            e->resample = true;
            e->rs_model_in = e->sub->in_len();
            e->rs_model_out = e->sub->out_len();
            e->rs_scale_in = 1.0f;
            e->rs_scale_out = fam_sand ? 0.0f : 1.0f;
            e->rs_sandwich_out = fam_sand;
            e->rs_pcm_scale = fam_dfsmn ? 32768.0f : fam_hg ? 32767.0f : 1.0f;     // as on the resampling path below
            e->rs_truncate_i32 = !fam_dfsmn && !fam_hg;
        }
        if (float_io) {
            e->gt_float_in = float_in_d;
            e->rs_in_gain = fam_moss ? 1.0f : 32768.0f;                 // MossFormer2 normalises its input itself and returns its units (Export_MossFormer2_SS_16K.py:563, 585)
            e->rs_f32_scale = (fam_moss || fam_zip) ? (float)(1.0 / 32768.0) : 1.0f;     // (:655; Export_ZipEnhancer.py:920-922)
        }
        e->rs_nan_to_num = fam_hg ? 2 : ((fam_zip || (fam_ulu && float_in_d)) ? 1 : 0);          // (Export_ZipEnhancer.py:913-920; Export_H_GTCRN.py:1056; Export_UL_UNAS.py:906-907)
        if (rates_differ) {   // F.interpolate(size = ...) on both edges (:562-571, :625-640): source scale = source length / target length
            e->resample = true;
            e->rs_model_in = e->sub->in_len();
            e->rs_model_out = e->sub->out_len();
            long out_caller;
            if (fam_hg || dyn_sf) {   // F.interpolate(scale_factor = ...) (Export_H_GTCRN.py:953-970, 1036-1052; DFSMN's dynamic export): floor(length * factor) samples, source step 1 / factor
                out_caller = (long)floor((double)e->rs_model_out * ((double)sro / (double)srm));
                e->rs_scale_in = (float)((double)sri / (double)srm);
                e->rs_scale_out = (float)((double)srm / (double)sro);
                e->sub->float_src_len = (int)caller_len;
            } else if (fam_sand) {   // scale_factor on both edges; down-sampling precedes the * 32767 of an int16 output, up-sampling follows it (Export_MelBandRoformer.py:660-680,
                //                                                                                                                     Export_UL_UNAS.py:890-905)
                const double f_in = fam_ulu ? 1.0 / ((double)sri / 16000.0) : (double)srm / (double)sri, f_out = fam_ulu ? (double)sro / 16000.0 : (double)sro / (double)srm;
                out_caller = sro == srm ? (long)e->rs_model_out : (long)floor((double)e->rs_model_out * f_out);
                e->rs_scale_in = sri == srm ? 1.0f : (float)(1.0 / f_in);
                e->rs_scale_out = sro == srm ? 0.0f : (float)(1.0 / f_out);
                e->rs_sandwich_out = true;
                e->rs_scale_first = sro > srm;
                if (out_caller < 1) return bail(fail(e, ADE_ERR_SHAPE_MISMATCH, fam + ": the output-rate waveform is empty"));
            } else {
                out_caller = (long)nearbyint((double)caller_len * (double)sro / (double)sri);     // OUTPUT_AUDIO_LENGTH (:37)
                e->rs_scale_in = (float)((double)caller_len / (double)e->rs_model_in);
                e->rs_scale_out = (float)((double)e->rs_model_out / (double)out_caller);
            }
            e->rs_pcm_scale = fam_dfsmn ? 32768.0f : fam_hg ? 32767.0f : 1.0f;   // DFSMN: * 32768 after the interpolation (Export_DFSMN.py:241-243); H-GTCRN: * 32767
            e->rs_truncate_i32 = !fam_dfsmn && !fam_hg;      // (:1045); MossFormer2's waveform is already in PCM units and goes through .to(int32).clamp().to(int16) (:645)
            e->in_len = (int)caller_len * e->channels;
            e->out_len = (int)out_caller * e->out_channels * e->n_outputs;
        }
        e->sample_rate = (int)rate;
        e->in_rate = (int)sri;
        e->out_rate = (int)sro;
        e->blob_storage.clear();
        e->blob_storage.shrink_to_fit();
        e->tensors.clear();
        *out = e;
        return ADE_OK;
    }
    if (e->meta["model_family"] != "gtcrn")
        return bail(fail(e, ADE_ERR_UNSUPPORTED, "model_family '" + e->meta["model_family"] + "' is not implemented (gtcrn, h_gtcrn, dfsmn, mel_band_roformer, mossformer2_ss, ul_unas, zipenhancer)"));
    bool dyn = false;
    if (!parse_bool(e->meta["dynamic_axes"], &dyn))
        return bail(fail(e, ADE_ERR_BAD_VALUE, "Metadata key dynamic_axes must be a boolean encoded as 1/0, got '" + e->meta["dynamic_axes"] + "'."));
    long sr_in = 0, sr_out = 0, sr_model = 0, L = 0;
    if (!parse_int(e->meta["in_sample_rate"], &sr_in) || !parse_int(e->meta["out_sample_rate"], &sr_out) ||
        !parse_int(e->meta["model_sample_rate"], &sr_model) || !parse_int(e->meta["input_audio_length"], &L))
        return bail(fail(e, ADE_ERR_BAD_VALUE, "manifest: sample rates / input_audio_length must be integers"));
    // GTCRN_CUSTOM's sandwich (Export_GTCRN.py:636-693).  A dynamic_axes export takes its frame count from the model-rate waveform and keeps 256 T samples of the
    // overlap-add (STFT_Process.py:337-341); the engine still serves ONE input length per handle (the manifest's input_audio_length).  The STATIC export sizes its frame
    // count from the input-rate length (:45), so other sample rates are only self-consistent with dynamic_axes = 1.
    const bool rates_differ_g = sr_in != sr_model || sr_out != sr_model;
    auto dtype_ok = [](const std::string& d) { return d == "INT16" || d == "F32" || d == "F16"; };
    if (!dtype_ok(e->meta["input_audio_dtype"]) || !dtype_ok(e->meta["output_audio_dtype"]))
        return bail(fail(e, ADE_ERR_BAD_VALUE, "manifest: input_audio_dtype / output_audio_dtype must be INT16, F32 or F16"));
    const bool float_in = e->meta["input_audio_dtype"] != "INT16";       // F16 tensors cross the C ABI as fp32 (the host layer converts): the graph computes in fp32
    if (rates_differ_g && !dyn)
        return bail(fail(e, ADE_ERR_UNSUPPORTED, "gtcrn: in / out sample rates other than the model rate need dynamic_axes=1 (the static export sizes its frame count from "
                                                 "the input-rate length, Export_GTCRN.py:45)"));
    if (rates_differ_g && (sr_model != 16000 || sr_in < 1000 || sr_out < 1000 || sr_in > 384000 || sr_out > 384000))
        return bail(fail(e, ADE_ERR_BAD_VALUE, "manifest: sample rates out of range"));
    const bool sandwich = dyn || rates_differ_g || float_in;
    auto opt = [&](const char* k, const char* want) {
        auto it = e->meta.find(k);
        return it == e->meta.end() || it->second.empty() || it->second == want;
    };
    if (!opt("nfft", "512") || !opt("hop_length", "256") || !opt("window_length", "512") || !opt("window_type", "hann_sqrt") ||
        !opt("pad_mode", "reflect") || !opt("center_pad", "1"))
        return bail(fail(e, ADE_ERR_UNSUPPORTED, "STFT configuration other than 512/512/256 hann_sqrt reflect centre-pad"));
    bool fold = false;
    if (e->meta.count("use_batch_fold") && !e->meta["use_batch_fold"].empty()) {
        if (!parse_bool(e->meta["use_batch_fold"], &fold))
            return bail(fail(e, ADE_ERR_BAD_VALUE, "Metadata key use_batch_fold must be a boolean encoded as 1/0."));
    }
    long fold_w = 0;
    if (fold) {   // the call's input is n_win windows of fold_window_length model-rate samples (a multiple of the hop)
        if (!e->meta.count("fold_window_length") || !parse_int(e->meta["fold_window_length"], &fold_w) || fold_w <= 0 || fold_w % kHop != 0)
            return bail(fail(e, ADE_ERR_BAD_VALUE, "use_batch_fold=1 needs fold_window_length: a positive multiple of the hop length"));
    }
    // validate_audio_metadata (audio_onnx_metadata.py:322-351): export length / channels must agree with the "graph"
    long v = 0;
    if (fold) {   // the graph input is EXPORT_AUDIO_LENGTH = ceil(INPUT_AUDIO_LENGTH / W) * W            (Export_GTCRN.py:43)
        const long want = (L + fold_w - 1) / fold_w * fold_w;
        if (e->meta.count("export_audio_length") && !e->meta["export_audio_length"].empty()) {
            if (!parse_int(e->meta["export_audio_length"], &v) || v != want)
                return bail(fail(e, ADE_ERR_SHAPE_MISMATCH, "export_audio_length is not input_audio_length rounded up to whole fold windows"));
        }
        e->n_win = (int)(want / fold_w);
        L = fold_w;
    } else if (e->meta.count("export_audio_length") && !e->meta["export_audio_length"].empty()) {
        if (!parse_int(e->meta["export_audio_length"], &v) || v != L)
            return bail(fail(e, ADE_ERR_SHAPE_MISMATCH, "input length does not match metadata export_audio_length"));
    }
    for (const char* k : {"input_channels", "output_channels", "num_audio_inputs"})
        if (e->meta.count(k) && !e->meta[k].empty() && e->meta[k] != "1")
            return bail(fail(e, ADE_ERR_SHAPE_MISMATCH, std::string("metadata ") + k + " must be 1 for GTCRN"));
    if (L < kNfft / 2 + 2 || L > (1 << 24)) return bail(fail(e, ADE_ERR_SHAPE_MISMATCH, "input_audio_length out of range"));
    e->in_len = (int)L;
    e->sample_rate = (int)sr_model;
    if (sandwich) {
        // batch-fold needs a static shape and equal rates in the reference (Export_GTCRN.py:41); with a float input tensor the fold is the same reshape after the
        // centring of the whole call (:645-660): the input stage below runs on whole calls, everything after it on their windows
        if (fold && (dyn || rates_differ_g)) return bail(fail(e, ADE_ERR_BAD_VALUE, "Batch folding requires a static shape and equal input/model/output sample rates."));
        // F.interpolate(scale_factor = f): floor(length * f) samples, source step 1 / f (the same evaluation order as the module's attributes, :623-626)
        long Lm = L;
        e->gt_l1 = (int)L;
        if (sr_in != sr_model) {
            const double f_in = 1.0 / ((double)sr_in / 16000.0);
            Lm = (long)floor((double)L * f_in);
            if (sr_in > sr_model) { e->gt_lerp1 = (float)(1.0 / f_in); e->gt_l1 = (int)Lm; }     // down: interpolate, scale, centre
            else e->gt_lerp2 = (float)(1.0 / f_in);                                               // up: scale, centre, interpolate
        }
        if (Lm < kNfft / 2 + 2) return bail(fail(e, ADE_ERR_SHAPE_MISMATCH, "input_audio_length is too short at the model rate"));
        e->gt_sand = true;
        e->gt_float_in = float_in;
        e->gt_gain = float_in ? 1.0f : (float)(1.0 / 32768.0);
        e->gt_lm = (int)Lm;
        e->T = e->gt_lm / kHop + 1;
        e->gt_keep = dyn ? kHop * e->T : kHop * (e->T - 1);                                       // STFT_Process.py:337-341 vs :169-176
        long Lout = e->gt_keep;
        if (sr_out != sr_model) {
            const double f_out = (double)sr_out / 16000.0;
            Lout = (long)floor((double)e->gt_keep * f_out);
            e->gt_lerp_out = (float)(1.0 / f_out);
            e->gt_scale_first = sr_out > sr_model;                                                // up: * 32767 BEFORE the interpolation (:680-688)
        }
        if (Lout < 1 || Lout > (1 << 24)) return bail(fail(e, ADE_ERR_SHAPE_MISMATCH, "output length out of range"));
        e->out_len = (int)Lout;
        e->in_rate = (int)sr_in;
        e->out_rate = (int)sr_out;
    } else {
        e->T = e->in_len / kHop + 1;                 // STATIC_SIGNAL_LENGTH, Export_GTCRN.py:45
        e->out_len = kHop * (e->T - 1);              // STFT_Process.py:169-176 with max_frames = T
        if (e->meta.count("max_signal_length") && !e->meta["max_signal_length"].empty()) {
            if (!parse_int(e->meta["max_signal_length"], &v) || v != e->T)
                return bail(fail(e, ADE_ERR_SHAPE_MISMATCH, "max_signal_length does not equal input_audio_length // hop + 1"));
        }
    }
    ade_status st = parse_blob(e, weights, weights_nbytes);
    if (st != ADE_OK) return bail(st);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        (void)hipGetLastError();
        return bail(fail(e, ADE_ERR_DEVICE, "no HIP device visible: libade has no CPU execution mode"));
    }
    if (device < 0 || device >= ndev) return bail(fail(e, ADE_ERR_DEVICE, "device ordinal out of range"));
    e->device = device;
    if (hipSetDevice(device) != hipSuccess) return bail(fail(e, ADE_ERR_DEVICE, "hipSetDevice failed"));
    if (hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess)
        return bail(fail(e, ADE_ERR_DEVICE, "hipStreamCreate failed"));
    st = build_device_constants(e);
    if (st != ADE_OK) return bail(st);
    if (fused_init() != hipSuccess) {
        (void)hipGetLastError();
        e->use_fused = false;   // keep the multi-kernel path if the 140 KB dynamic-LDS request is refused
    }
    e->blob_storage.clear();
    e->blob_storage.shrink_to_fit();
    e->tensors.clear();
    const char* env = getenv("ADE_GRAPH");
    if (env && env[0] == '0') e->use_graph = false;
    *out = e;
    return ADE_OK;
}

ade_status ade_get_io(ade_handle h, ade_io_desc* d) {
    if (!h || !d) return ADE_ERR_BAD_VALUE;
    d->abi_version = ADE_ABI_VERSION;
    d->in_channels = h->channels;
    d->out_channels = h->out_channels;
    d->n_outputs = h->n_outputs;
    d->in_len = h->in_len / h->channels * h->n_win;       // per channel; what one call sees (the fold is internal)
    d->out_len = h->out_len / (h->out_channels * h->n_outputs) * h->n_win;
    d->model_sample_rate = h->sample_rate;
    d->in_sample_rate = h->in_rate ? h->in_rate : h->sample_rate;
    d->out_sample_rate = h->out_rate ? h->out_rate : h->sample_rate;
    d->frames = h->T;
    d->max_batch = h->capacity / h->n_win;
    d->device = h->device;
    return ADE_OK;
}

ade_status ade_reserve(ade_handle h, int batch) {
    if (!h || batch < 0) return ADE_ERR_BAD_VALUE;
    return reserve(h, batch * h->n_win);
}

ade_status ade_set_option(ade_handle h, const char* key, const char* value) {
    if (!h || !key || !value) return ADE_ERR_BAD_VALUE;
    if (strcmp(key, "stagger_us") == 0) {      // delay of every other workgroup group in the single-launch kernel, microseconds (0 = off)
        char* end = nullptr;
        const double us = strtod(value, &end);
        if (!value[0] || *end || us < 0.0 || us > 1000.0) return fail(h, ADE_ERR_BAD_VALUE, "option stagger_us: 0..1000");
        h->stagger_ticks = (int)(us * 100.0 + 0.5);
        return ADE_OK;
    }
    if (strcmp(key, "wave_swap") == 0) {
        bool b;
        if (!parse_bool(value, &b)) return fail(h, ADE_ERR_BAD_VALUE, "option wave_swap must be 0/1");
        h->wave_swap = b;
        return ADE_OK;
    }
    if (strcmp(key, "seg_prio") == 0) {        // base wave priority by segment: 0 none, 1-3 that level for every later segment, 4 earlier segments first (SegPlan::prio)
        if (value[0] < '0' || value[0] > '4' || value[1]) return fail(h, ADE_ERR_BAD_VALUE, "option seg_prio: 0..4");
        h->seg_prio = value[0] - '0';
        return ADE_OK;
    }
    if (strcmp(key, "xwait_ms") == 0) {        // bound of one inter-workgroup wait of the segmented fused path, milliseconds (default 200)
        char* end = nullptr;
        const double ms = strtod(value, &end);
        if (!value[0] || *end || !(ms >= 0.001) || ms > 20000.0) return fail(h, ADE_ERR_BAD_VALUE, "option xwait_ms: 0.001..20000");
        h->xwait_ticks = (int)(ms * 1e5 + 0.5);
        if (h->sub) h->sub->set_exchange_wait_ticks(h->xwait_ticks);
        free_graphs(h);
        return ADE_OK;
    }
    if (strcmp(key, "pipe_depth") == 0) {      // submissions ade_submit keeps in flight (2 .. 4); takes effect while nothing is in flight
        if (value[0] < '2' || value[0] > '4' || value[1]) return fail(h, ADE_ERR_BAD_VALUE, "option pipe_depth: 2..4");
        for (const auto& sl : h->pipe) if (sl.state == 1) return fail(h, ADE_ERR_BAD_VALUE, "option pipe_depth: submissions are in flight (ade_wait them first)");
        h->pipe_depth = value[0] - '0';
        return ADE_OK;
    }
    if (strcmp(key, "xwait_retry") == 0) {     // "1" (default): ade_process re-runs a timed-out call once without hand-offs; "0": it fails
        if ((value[0] != '0' && value[0] != '1') || value[1]) return fail(h, ADE_ERR_BAD_VALUE, "option xwait_retry: 0 or 1");
        h->xwait_retry = value[0] - '0';
        return ADE_OK;
    }
    if (strcmp(key, "host_stream") == 0) {     // row groups a host batch is streamed through ONE launch in (ade_process); 0 = per call (four from 128 rows), 1 = off
        if (value[0] < '0' || value[0] > '8' || value[1]) return fail(h, ADE_ERR_BAD_VALUE, "option host_stream: 0..8");
        h->host_stream = value[0] - '0';
        return ADE_OK;
    }
    if (strcmp(key, "host_split") == 0) {      // sub-batches ade_process cuts a host batch into (the copies of one overlap the kernel of another); 0 = per call (two from 128 rows), 1 = never
        if (value[0] < '0' || value[0] > '8' || value[1]) return fail(h, ADE_ERR_BAD_VALUE, "option host_split: 0..8");
        h->host_split = value[0] - '0';
        return ADE_OK;
    }
    if (strcmp(key, "full_taps") == 0) {       // debugging: x_d0 / x_d1 / dp2 complete in device memory after a single-launch call (their channels 0-7 otherwise never leave LDS)
        bool b;
        if (!parse_bool(value, &b)) return fail(h, ADE_ERR_BAD_VALUE, "option full_taps must be 0/1");
        h->full_taps = b;
        return ADE_OK;
    }
    if (strcmp(key, "xchg_withhold") == 0) {   // TEST HOOK: the first workgroup of the next fused launches raises its hand-off flags where nobody looks (SegPlan::withhold)
        bool b;
        if (!parse_bool(value, &b)) return fail(h, ADE_ERR_BAD_VALUE, "option xchg_withhold must be 0/1");
        h->xchg_withhold = b;
        free_graphs(h);
        return ADE_OK;
    }
    if (strcmp(key, "geometry") == 0) {        // fused-path workgroup geometry: "auto", "0" (1024 threads x 64 frames), "1" (512 x 32, two per CU)
        int g = -2;
        if (strcmp(value, "auto") == 0) g = -1;
        else if (strcmp(value, "0") == 0) g = 0;
        else if (strcmp(value, "1") == 0) g = 1;
        else if (strcmp(value, "2") == 0) g = 2;
        if (g == -2) return fail(h, ADE_ERR_BAD_VALUE, "option geometry: auto, 0, 1 or 2");
        h->geometry = g;
        free_graphs(h);
        return ADE_OK;
    }
    if (strcmp(key, "graph") == 0 || strcmp(key, "fused") == 0 || strcmp(key, "single_launch") == 0) {
        bool b;
        if (!parse_bool(value, &b)) return fail(h, ADE_ERR_BAD_VALUE, std::string("option ") + key + " must be 0/1");
        if (key[0] == 'g') h->use_graph = b;
        else if (key[0] == 'f') h->use_fused = b;
        else h->use_single = b;
        free_graphs(h);
        return ADE_OK;
    }
    return fail(h, ADE_ERR_MISSING_KEY, std::string("unknown option: ") + key);
}

// Submissions in flight (ade_submit) own the exchange area's time-out word until their ade_wait has looked at it: a synchronous entry point that ran beside them would
// read -- and clear -- a time-out that belongs to one of their tickets, which would then report ADE_OK for garbage.  Every ade_process* entry therefore completes the ring
// first (each submission's status is kept for its ade_wait, exactly as when a full ring completes its oldest one).
namespace { void pipe_finish(ade_engine* h, ade_engine::PipeSlot& sl); }
static void pipe_drain(ade_engine* h) {
    for (;;) {
        ade_engine::PipeSlot* oldest = nullptr;
        for (auto& sl : h->pipe) if (sl.state == 1 && (!oldest || sl.ticket < oldest->ticket)) oldest = &sl;
        if (!oldest) return;
        pipe_finish(h, *oldest);
    }
}

ade_status ade_process_device(ade_handle h, const int16_t* d_in, int batch, int16_t* d_out, float* d_f32, void* hip_stream) {
    if (!h) return ADE_ERR_BAD_VALUE;
    if (batch < 0 || (batch > 0 && (!d_in || (!d_out && !d_f32)))) return fail(h, ADE_ERR_BAD_VALUE, "ade_process_device: bad arguments");
    if (h->gt_float_in) return fail(h, ADE_ERR_BAD_VALUE, "this handle's input_audio_dtype is F32 / F16: call ade_process_device_f32");
    HIP_TRY(h, hipSetDevice(h->device));
    pipe_drain(h);
    { const ade_status xs = exchange_status(h, "ade_process_device", true); if (xs != ADE_OK) return xs; }
    const int rows = batch * h->n_win;      // batch-fold: every call is n_win internal rows (windows)
    ade_status st = reserve(h, rows);
    if (st != ADE_OK) return st;
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : h->stream;
    st = run(h, s, d_in, rows, d_out, d_f32);
    if (st != ADE_OK) return st;
    if (!hip_stream) {
        HIP_TRY(h, hipStreamSynchronize(s));
        return exchange_status(h, "ade_process_device", false);
    }
    return ADE_OK;
}

// fp32 audio in (input_audio_dtype F32 / F16): GTCRN handles take the sandwich path, sub-engine handles their edge kernels; out_pcm / out_f32 as in ade_process_device
ade_status ade_process_device_f32(ade_handle h, const float* d_in, int batch, int16_t* d_out, float* d_f32, void* hip_stream) {
    if (!h) return ADE_ERR_BAD_VALUE;
    if (batch < 0 || (batch > 0 && (!d_in || (!d_out && !d_f32)))) return fail(h, ADE_ERR_BAD_VALUE, "ade_process_device_f32: bad arguments");
    if (!h->gt_float_in) return fail(h, ADE_ERR_BAD_VALUE, "this handle's input_audio_dtype is INT16: call ade_process_device");
    if (batch == 0) return ADE_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    pipe_drain(h);
    { const ade_status xs = exchange_status(h, "ade_process_device_f32", true); if (xs != ADE_OK) return xs; }
    const int rows = batch * h->n_win;      // batch-fold: every call is n_win internal rows (windows)
    ade_status st = reserve(h, rows);
    if (st != ADE_OK) return st;
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : h->stream;
    h->cur_fin = d_in;
    st = run(h, s, reinterpret_cast<const int16_t*>(d_in), rows, d_out, d_f32);      // (the pointer only keys the graph cache: enqueue reads cur_fin)
    h->cur_fin = nullptr;
    if (st != ADE_OK) return st;
    if (!hip_stream) {
        HIP_TRY(h, hipStreamSynchronize(s));
        return exchange_status(h, "ade_process_device_f32", false);
    }
    return ADE_OK;
}

// ---- multi-GPU stitch: one RCCL all-gather of the rank's output rows on the caller's communicator and stream (librccl opened on first use)
ade_status ade_stitch_device(ade_handle h, const int16_t* d_local, int rows, int16_t* d_all, void* nccl_comm, void* hip_stream) {
    if (!h) return ADE_ERR_BAD_VALUE;
    if (rows < 0 || (rows > 0 && (!d_local || !d_all)) || !nccl_comm) return fail(h, ADE_ERR_BAD_VALUE, "ade_stitch_device: bad arguments");
    if (rows == 0) return ADE_OK;
#if defined(HIPSIM)
    (void)hip_stream;
    return fail(h, ADE_ERR_UNSUPPORTED, "ade_stitch_device: no RCCL under the host simulator");
#else
    typedef int (*all_gather_fn)(const void*, void*, size_t, int, void*, hipStream_t);      // ncclResult_t ncclAllGather(send, recv, count, ncclDataType_t, ncclComm_t, stream)
    static all_gather_fn all_gather = nullptr;
    static std::once_flag once;                       // (two handles may stitch from two threads)
    std::call_once(once, [] {
        void* lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (lib) all_gather = reinterpret_cast<all_gather_fn>(dlsym(lib, "ncclAllGather"));
    });
    if (!all_gather) return fail(h, ADE_ERR_UNSUPPORTED, "ade_stitch_device: librccl.so / ncclAllGather not found");
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t bytes = (size_t)rows * h->out_len * sizeof(int16_t);
    const int rc = all_gather(d_local, d_all, bytes, /*ncclUint8*/ 1, nccl_comm, hip_stream ? (hipStream_t)hip_stream : h->stream);
    if (rc != 0) return fail(h, ADE_ERR_DEVICE, "ade_stitch_device: ncclAllGather failed with ncclResult_t " + std::to_string(rc));
    if (!hip_stream) HIP_TRY(h, hipStreamSynchronize(h->stream));
    return ADE_OK;
#endif
}

// ---- IEEE-half tensors at the boundary: widen into the fp32 entry's staging, run, narrow the float output
static ade_status reserve_f16(ade_handle h, int rows) {
    if (rows <= h->f16_capacity && h->d_f16_out) return ADE_OK;
    for (uint16_t** p : {&h->d_f16_in, &h->d_f16_out}) { if (*p) hipFree(*p); *p = nullptr; }
    for (uint16_t** p : {&h->h_f16_in, &h->h_f16_out}) { if (*p) hipHostFree(*p); *p = nullptr; }
    const size_t cap = (size_t)std::max(rows, h->capacity);
    HIP_TRY(h, hipMalloc((void**)&h->d_f16_in, cap * h->in_len * sizeof(uint16_t)));
    HIP_TRY(h, hipMalloc((void**)&h->d_f16_out, cap * h->out_len * sizeof(uint16_t)));
    HIP_TRY(h, hipHostMalloc((void**)&h->h_f16_in, cap * h->in_len * sizeof(uint16_t), hipHostMallocDefault));
    HIP_TRY(h, hipHostMalloc((void**)&h->h_f16_out, cap * h->out_len * sizeof(uint16_t), hipHostMallocDefault));
    h->f16_capacity = (int)cap;
    return ADE_OK;
}

ade_status ade_process_device_f16(ade_handle h, const void* d_in, int batch, int16_t* d_out, uint16_t* d_f16, void* hip_stream) {
    if (!h) return ADE_ERR_BAD_VALUE;
    if (batch < 0 || (batch > 0 && (!d_in || (!d_out && !d_f16)))) return fail(h, ADE_ERR_BAD_VALUE, "ade_process_device_f16: bad arguments");
    if (batch == 0) return ADE_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    pipe_drain(h);
    { const ade_status xs = exchange_status(h, "ade_process_device_f16", true); if (xs != ADE_OK) return xs; }
    const int rows = batch * h->n_win;
    ade_status st = reserve(h, rows);
    if (st != ADE_OK) return st;
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : h->stream;
    float* f32 = d_f16 ? h->d_f32_out : nullptr;
    if (h->gt_float_in) {
        launch_half_to_float(s, static_cast<const uint16_t*>(d_in), h->d_f32_in, (long long)rows * h->in_len);
        h->cur_fin = h->d_f32_in;
        st = run(h, s, reinterpret_cast<const int16_t*>(h->d_f32_in), rows, d_out, f32);
        h->cur_fin = nullptr;
    } else {
        st = run(h, s, static_cast<const int16_t*>(d_in), rows, d_out, f32);
    }
    if (st != ADE_OK) return st;
    if (d_f16) launch_float_to_half(s, h->d_f32_out, d_f16, (long long)rows * h->out_len);
    HIP_TRY(h, hipGetLastError());
    if (!hip_stream) {
        HIP_TRY(h, hipStreamSynchronize(s));
        return exchange_status(h, "ade_process_device_f16", false);
    }
    return ADE_OK;
}

ade_status ade_process_f16(ade_handle h, const void* in, int batch, int16_t* out_pcm, uint16_t* out_f16) {
    if (!h) return ADE_ERR_BAD_VALUE;
    if (batch < 0 || (batch > 0 && (!in || (!out_pcm && !out_f16)))) return fail(h, ADE_ERR_BAD_VALUE, "ade_process_f16: bad arguments");
    if (batch == 0) return ADE_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    pipe_drain(h);
    { const ade_status xs = exchange_status(h, "ade_process_f16", true); if (xs != ADE_OK) return xs; }
    const int rows = batch * h->n_win;
    ade_status st = reserve(h, rows);
    if (st != ADE_OK) return st;
    st = reserve_f16(h, rows);
    if (st != ADE_OK) return st;
    const size_t nin = (size_t)rows * h->in_len, nout = (size_t)rows * h->out_len;
    const void* d_in;
    if (h->gt_float_in) {
        memcpy(h->h_f16_in, in, nin * sizeof(uint16_t));
        HIP_TRY(h, hipMemcpyAsync(h->d_f16_in, h->h_f16_in, nin * sizeof(uint16_t), hipMemcpyHostToDevice, h->stream));
        d_in = h->d_f16_in;
    } else {
        memcpy(h->h_pcm_in, in, nin * sizeof(int16_t));
        HIP_TRY(h, hipMemcpyAsync(h->d_pcm_in, h->h_pcm_in, nin * sizeof(int16_t), hipMemcpyHostToDevice, h->stream));
        d_in = h->d_pcm_in;
    }
    st = ade_process_device_f16(h, d_in, batch, out_pcm ? h->d_pcm_out : nullptr, out_f16 ? h->d_f16_out : nullptr, (void*)h->stream);
    if (st != ADE_OK) return st;
    if (out_pcm) HIP_TRY(h, hipMemcpyAsync(h->h_pcm_out, h->d_pcm_out, nout * sizeof(int16_t), hipMemcpyDeviceToHost, h->stream));
    if (out_f16) HIP_TRY(h, hipMemcpyAsync(h->h_f16_out, h->d_f16_out, nout * sizeof(uint16_t), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    { const ade_status xs = exchange_status(h, "ade_process_f16", false); if (xs != ADE_OK) return xs; }
    if (out_pcm) memcpy(out_pcm, h->h_pcm_out, nout * sizeof(int16_t));
    if (out_f16) memcpy(out_f16, h->h_f16_out, nout * sizeof(uint16_t));
    return ADE_OK;
}

ade_status ade_process_f32(ade_handle h, const float* in, int batch, int16_t* out_pcm, float* out_f32) {
    if (!h) return ADE_ERR_BAD_VALUE;
    if (batch < 0 || (batch > 0 && (!in || (!out_pcm && !out_f32)))) return fail(h, ADE_ERR_BAD_VALUE, "ade_process_f32: bad arguments");
    if (!h->gt_float_in) return fail(h, ADE_ERR_BAD_VALUE, "this handle's input_audio_dtype is INT16: call ade_process");
    if (batch == 0) return ADE_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    pipe_drain(h);
    { const ade_status xs = exchange_status(h, "ade_process_f32", true); if (xs != ADE_OK) return xs; }
    const int rows = batch * h->n_win;
    ade_status st = reserve(h, rows);
    if (st != ADE_OK) return st;
    const size_t nin = (size_t)rows * h->in_len, nout = (size_t)rows * h->out_len;
    HIP_TRY(h, hipMemcpyAsync(h->d_f32_in, in, nin * sizeof(float), hipMemcpyHostToDevice, h->stream));
    h->cur_fin = h->d_f32_in;
    st = run(h, h->stream, reinterpret_cast<const int16_t*>(h->d_f32_in), rows, h->d_pcm_out, out_f32 ? h->d_f32_out : nullptr);
    h->cur_fin = nullptr;
    if (st != ADE_OK) return st;
    if (out_pcm) HIP_TRY(h, hipMemcpyAsync(out_pcm, h->d_pcm_out, nout * sizeof(int16_t), hipMemcpyDeviceToHost, h->stream));
    if (out_f32) HIP_TRY(h, hipMemcpyAsync(out_f32, h->d_f32_out, nout * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    {
        const ade_status xs = exchange_status(h, "ade_process_f32", false);
        if (xs != ADE_OK) {
            if (out_pcm) memset(out_pcm, 0, nout * sizeof(int16_t));
            if (out_f32) memset(out_f32, 0, nout * sizeof(float));
            return xs;
        }
    }
    return ADE_OK;
}

static ade_status process_once(ade_handle h, const int16_t* in, int batch, int16_t* out_pcm, float* out_f32);

// The host-buffer entry.  A bounded inter-workgroup wait that gives up fails the launch (exchange_status); on a shared or pre-empted GPU, under a debugger or a
// serialising profiler that can happen to a perfectly valid call.  This entry is synchronous and owns its buffers for the duration of the call, so it re-runs such a
// call ONCE on the path that has no hand-offs -- whole chunks per workgroup (geometry 0, the same bits), or the multi-kernel sequence where a chunk has more than 64
// frames -- and reports success; ade_last_error then says that a retry happened.  Option "xwait_retry" = "0" turns this off; the test hook that withholds a flag does too.
ade_status ade_process(ade_handle h, const int16_t* in, int batch, int16_t* out_pcm, float* out_f32) {
    if (!h) return ADE_ERR_BAD_VALUE;
    // An EARLIER call's failure (an unsynchronised ade_process_device launch on a caller-provided stream that timed out: include/ade.h says the next call on the handle
    // reports it) is checked HERE and surfaces as ADE_ERR_DEVICE -- that call's output was garbage and nothing this call does can repair it.  Only a time-out of THIS call's
    // own launch (found by process_once after its synchronise) is re-run below.
    if (hipSetDevice(h->device) == hipSuccess) pipe_drain(h);
    if (hipSetDevice(h->device) == hipSuccess) { const ade_status xs = exchange_status(h, "ade_process", true); if (xs != ADE_OK) return xs; }
    h->timed_out = false;
    ade_status st = process_once(h, in, batch, out_pcm, out_f32);
    if (st == ADE_ERR_DEVICE && h->timed_out && h->xwait_retry && !h->xchg_withhold && !h->sub) {
        const std::string first = h->last_error;
        const int keep_geo = h->geometry, keep_fused = h->use_fused;
        if (fused_supported(h->T, 0)) h->geometry = 0; else h->use_fused = 0;
        free_graphs(h);
        h->timed_out = false;
        st = process_once(h, in, batch, out_pcm, out_f32);
        h->geometry = keep_geo; h->use_fused = keep_fused;
        free_graphs(h);
        if (st == ADE_OK) { ++h->retries; h->last_error = "ade_process: re-run without inter-workgroup hand-offs after: " + first; }
    }
    return st;
}

static ade_status process_once(ade_handle h, const int16_t* in, int batch, int16_t* out_pcm, float* out_f32) {
    if (batch < 0 || (batch > 0 && (!in || (!out_pcm && !out_f32)))) return fail(h, ADE_ERR_BAD_VALUE, "ade_process: bad arguments");
    if (h->gt_float_in) return fail(h, ADE_ERR_BAD_VALUE, "this handle's input_audio_dtype is F32 / F16: call ade_process_f32");
    if (batch == 0) return ADE_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    pipe_drain(h);
    { const ade_status xs = exchange_status(h, "ade_process", true); if (xs != ADE_OK) return xs; }
    const int rows = batch * h->n_win;
    ade_status st = reserve(h, rows);
    if (st != ADE_OK) return st;
    const size_t nin = (size_t)rows * h->in_len, nout = (size_t)rows * h->out_len;
    // Page-locked caller buffers (hipHostMalloc / hipHostRegister / a pinned torch tensor) are DMA'd directly; pageable ones go through the
    // engine's own page-locked staging buffers (an async copy from pageable memory would be staged by the runtime anyway, synchronously).
    auto page_locked = [](const void* p) {
        hipPointerAttribute_t a;
        if (!p || hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
        return a.type == hipMemoryTypeHost;
    };
    const bool in_direct = page_locked(in), pcm_direct = page_locked(out_pcm), f32_direct = page_locked(out_f32);
    // The reference's own timing convention is this call (wall clock around copies + compute, Inference_GTCRN_ONNX.py:323-343).  Run as ONE copy in, ONE launch, ONE copy out
    // it is the sum of the three (0.16 + 0.36 + 0.16 ms at 256 x 1 s).  A batch of 128 rows or more is cut into TWO sub-batches (option "host_split" = n: n of them), each on
    // its own stream: chunks are independent calls of the graph, the kernel indexes everything by ChunkCall::chunk0 + its own chunk number, so the sub-batches share the
    // batch arrays and the workspace without touching each other's rows -- same bits as the single launch (tests/test_gpu_parity.py).  The second sub-batch's copy-in runs
    // under the first one's kernel, the two kernels run side by side, the first copy-out under the second kernel's tail.  MEASURED (profiles/r04_b_host_split_probe*.txt,
    // 256 x 1 s, page-locked buffers): 0.678 ms unsplit, 0.598 ms in two; 0.79 / 1.14 ms in 4 / 8 -- this process's streams land on two hardware queues, so a third and
    // fourth sub-batch queue up behind the first two (rocprofv3 timeline in DESIGN.md section 6), and a launch of 64 chunks is latency-bound at ~0.23 ms however little it carries.
    // ---- one launch, the batch streamed through it (ChunkCall::in_ready): the rows are dealt into groups; every group's copy-in is followed on the copy stream by a 64 KB
    // copy whose first word is this call's epoch, the group's workgroups wait for that word before their first PCM read, the group's last workgroup writes the epoch into
    // page-locked host memory, and this thread, polling that word, starts the group's copy-out.  The kernel is launched ONCE, at once: group 0 starts when its 1 / n of the
    // input has arrived, the other copy-ins run under its arithmetic, every copy-out but the last under the later groups' arithmetic.  Unlike the sub-batch path below this
    // needs no kernel concurrency between streams, i.e. it does not depend on which hardware queues the process's streams were dealt (the sub-batch path measured 0.60 ms
    // in a process with three streams and 0.80 ms in bench.py's, which has five).
    {
        const int geo = pick_geometry(h, rows);
        const int n_grp = h->host_stream ? h->host_stream : (rows >= 128 ? 4 : 1);
        const bool plain = !h->sub && !h->gt_sand && h->use_fused && h->use_single && geo >= 0 && !h->profile && h->n_win == 1;
        const int per = n_grp > 0 ? (rows + n_grp - 1) / n_grp : rows;
        const bool big = (size_t)per * h->in_len * sizeof(int16_t) >= 65536 && (size_t)per * h->out_len * sizeof(int16_t) >= 65536;      // copies a copy engine takes
        if (plain && n_grp > 1 && rows >= 2 * n_grp && big && h->sio_state == 0) {
            h->sio_state = -1;
            const size_t blk = (size_t)kReadyStride * sizeof(unsigned);
            bool ok = hipStreamCreateWithFlags(&h->s_in, hipStreamNonBlocking) == hipSuccess && hipStreamCreateWithFlags(&h->s_out, hipStreamNonBlocking) == hipSuccess &&
                      hipExtMallocWithFlags((void**)&h->d_sio_ready, ade_engine::kMaxGroups * blk, hipDeviceMallocFinegrained) == hipSuccess &&
                      hipMalloc((void**)&h->d_sio_count, ade_engine::kMaxGroups * sizeof(unsigned)) == hipSuccess &&
                      hipHostMalloc((void**)&h->h_sio_epoch, blk, hipHostMallocDefault) == hipSuccess &&
                      hipHostMalloc((void**)&h->h_sio_done, ade_engine::kMaxGroups * sizeof(unsigned), hipHostMallocDefault) == hipSuccess;
            if (ok) {
                memset(h->h_sio_epoch, 0, blk);
                memset(h->h_sio_done, 0, ade_engine::kMaxGroups * sizeof(unsigned));
                ok = hipMemset(h->d_sio_ready, 0, ade_engine::kMaxGroups * blk) == hipSuccess && hipMemset(h->d_sio_count, 0, ade_engine::kMaxGroups * sizeof(unsigned)) == hipSuccess &&
                     hipDeviceSynchronize() == hipSuccess;
            }
            (void)hipGetLastError();
            if (ok) h->sio_state = 1;
        }
        if (plain && n_grp > 1 && rows >= 2 * n_grp && big && h->sio_state == 1) {
            const int used = (rows + per - 1) / per;
            // A HIP call that fails once the launch is out must not return with the kernel still polling for rows that will never arrive and copies in flight: drain the
            // three streams (the waiting workgroups give up after their own bound), clear the time-out word and the flags, THEN report the original error.
            bool launched = false;
#define SIO_TRY(expr)                                                                                                        \
            do {                                                                                                             \
                hipError_t _err = (expr);                                                                                    \
                if (_err != hipSuccess) {                                                                                    \
                    const std::string _what = std::string(#expr) + ": " + hipGetErrorString(_err);                           \
                    if (launched) {                                                                                          \
                        (void)hipStreamSynchronize(h->stream); (void)hipStreamSynchronize(h->s_in); (void)hipStreamSynchronize(h->s_out); \
                        (void)exchange_status(h, "ade_process", false);                                                      \
                        (void)hipGetLastError();                                                                             \
                    }                                                                                                        \
                    return fail(h, ADE_ERR_DEVICE, _what);                                                                   \
                }                                                                                                            \
            } while (0)
            const unsigned epoch = ++h->sio_epoch ? h->sio_epoch : ++h->sio_epoch;            // (never 0: the words start at 0)
            h->last_batch = rows;
            h->last_fused = true;
            h->last_geometry = geo;
            h->h_sio_epoch[0] = epoch;                                                           // (the previous call's copies of this block are complete: every call ends synchronised)
            for (int g = 0; g < used; ++g) {
                const int r0 = g * per, nr = std::min(per, rows - r0);
                const size_t i0 = (size_t)r0 * h->in_len;
                const int16_t* src = in + i0;
                if (!in_direct) { memcpy(h->h_pcm_in + i0, in + i0, (size_t)nr * h->in_len * sizeof(int16_t)); src = h->h_pcm_in + i0; }
                SIO_TRY(hipMemcpyAsync(h->d_pcm_in + i0, src, (size_t)nr * h->in_len * sizeof(int16_t), hipMemcpyHostToDevice, h->s_in));
                SIO_TRY(hipMemcpyAsync(h->d_sio_ready + (size_t)g * kReadyStride, h->h_sio_epoch, (size_t)kReadyStride * sizeof(unsigned), hipMemcpyHostToDevice, h->s_in));
                if (g == 0) {       // the launch goes out behind the FIRST group's copies (enqueued, not complete): enqueueing the other groups' copies -- ~25 us of API time
                                    // each -- runs under the first copy; launching only after all of them measured 0.735 ms per call against 0.635.  A later group's
                                    // workgroups therefore also wait out this thread's progress through the loop: their bound (option "xwait_ms", 200 ms) is four orders
                                    // of magnitude above it, and a call that does time out is re-run (ade_process above)
                    ChunkCall C{};
                    C.plan.nseg = fused_segments(h->T, geo); C.plan.xchg = h->d_xchg; C.plan.flags = h->d_xflags; C.plan.err = h->d_xerr; C.plan.wave_swap = h->wave_swap;
                    C.plan.prio = h->seg_prio; C.plan.withhold = h->xchg_withhold; C.plan.wait_ticks = h->xwait_ticks;
                    C.fixed = h->d_fixed;
                    C.pcm_in = h->d_pcm_in; C.pcm_out = h->d_pcm_out; C.f32_out = out_f32 ? h->d_f32_out : nullptr;
                    C.L = h->in_len; C.T = h->T; C.B = rows; C.chunk0 = 0;
                    C.full_taps = h->full_taps;
                    C.in_ready = h->d_sio_ready; C.out_done = h->h_sio_done; C.out_count = h->d_sio_count; C.epoch = epoch; C.group_rows = per;
                    launch_gtcrn_chunk(h->stream, geo, C);
                    launched = true;
                    SIO_TRY(hipGetLastError());
                }
            }
            // copy-outs: started by this thread as the groups finish (a word per group in page-locked memory, written by the group's last workgroup)
            volatile unsigned* const done = h->h_sio_done;
            bool finished = false;      // the launch has completed (normally after every group was seen; a failed launch never raises its words)
            for (int g = 0; g < used; ++g) {
                const int r0 = g * per, nr = std::min(per, rows - r0);
                const size_t o0 = (size_t)r0 * h->out_len;
                for (unsigned spins = 0; done[g] != epoch && !finished; ++spins)
                    if ((spins & 0x3ffu) == 0x3ffu && hipStreamQuery(h->stream) == hipSuccess) finished = true;
                if (out_pcm) SIO_TRY(hipMemcpyAsync((pcm_direct ? out_pcm : h->h_pcm_out) + o0, h->d_pcm_out + o0, (size_t)nr * h->out_len * sizeof(int16_t), hipMemcpyDeviceToHost, h->s_out));
                if (out_f32) SIO_TRY(hipMemcpyAsync((f32_direct ? out_f32 : h->h_f32_out) + o0, h->d_f32_out + o0, (size_t)nr * h->out_len * sizeof(float), hipMemcpyDeviceToHost, h->s_out));
            }
#undef SIO_TRY
            (void)hipGetLastError();
            HIP_TRY(h, hipStreamSynchronize(h->stream));
            HIP_TRY(h, hipStreamSynchronize(h->s_out));
            HIP_TRY(h, hipStreamSynchronize(h->s_in));
            st = exchange_status(h, "ade_process", false);
            if (st != ADE_OK) {
                if (out_pcm && pcm_direct) memset(out_pcm, 0, nout * sizeof(int16_t));
                if (out_f32 && f32_direct) memset(out_f32, 0, nout * sizeof(float));
                return st;
            }
            if (out_pcm && !pcm_direct) memcpy(out_pcm, h->h_pcm_out, nout * sizeof(int16_t));
            if (out_f32 && !f32_direct) memcpy(out_f32, h->h_f32_out, nout * sizeof(float));
            return ADE_OK;
        }
    }
    {
        const int geo = pick_geometry(h, rows);
        int n_sub = h->host_split ? h->host_split : (rows >= 128 ? 2 : 1);
        const bool plain = !h->sub && !h->gt_sand && h->use_fused && h->use_single && geo >= 0 && !h->profile && h->n_win == 1;
        if (plain && n_sub > 1 && rows >= 2 * n_sub) {
            const int per = (rows + n_sub - 1) / n_sub;
            h->last_batch = rows;
            h->last_fused = true;
            h->last_geometry = geo;
            // Enqueue order matters: the runtime's copy queue is served in order of submission across streams (traced: with copy-in / launch / copy-out enqueued sub-batch by
            // sub-batch, the next sub-batch's copy-in waited behind the previous one's copy-out, i.e. behind its kernel -- a fully serial chain).  So: every copy-in first,
            // then every launch, then every copy-out.
            const int used = (rows + per - 1) / per;
            for (int k = 0; k < used; ++k) {
                const int r0 = k * per, nr = std::min(per, rows - r0);
                if (!h->sub_streams[k]) HIP_TRY(h, hipStreamCreateWithFlags(&h->sub_streams[k], hipStreamNonBlocking));
                const size_t i0 = (size_t)r0 * h->in_len;
                const int16_t* src = in + i0;
                if (!in_direct) { memcpy(h->h_pcm_in + i0, in + i0, (size_t)nr * h->in_len * sizeof(int16_t)); src = h->h_pcm_in + i0; }
                HIP_TRY(h, hipMemcpyAsync(h->d_pcm_in + i0, src, (size_t)nr * h->in_len * sizeof(int16_t), hipMemcpyHostToDevice, h->sub_streams[k]));
            }
            for (int k = 0; k < used; ++k) {
                const int r0 = k * per, nr = std::min(per, rows - r0);
                ChunkCall C{};
                C.plan.nseg = fused_segments(h->T, geo); C.plan.xchg = h->d_xchg; C.plan.flags = h->d_xflags; C.plan.wave_swap = h->wave_swap;
                C.plan.err = h->d_xerr + k;                               // its own word: exchange_status names the chunk with THIS launch's first chunk and size
                h->xl_B[k] = nr; h->xl_chunk0[k] = r0;
                C.plan.prio = h->seg_prio; C.plan.withhold = r0 == 0 ? h->xchg_withhold : 0; C.plan.wait_ticks = h->xwait_ticks;
                C.fixed = h->d_fixed;
                C.pcm_in = h->d_pcm_in; C.pcm_out = h->d_pcm_out; C.f32_out = out_f32 ? h->d_f32_out : nullptr;
                C.L = h->in_len; C.T = h->T; C.B = nr; C.chunk0 = r0;
                C.full_taps = h->full_taps;
                launch_gtcrn_chunk(h->sub_streams[k], geo, C);
                HIP_TRY(h, hipGetLastError());
            }
            for (int k = 0; k < used; ++k) {
                const int r0 = k * per, nr = std::min(per, rows - r0);
                const size_t o0 = (size_t)r0 * h->out_len;
                hipStream_t s = h->sub_streams[k];
                if (out_pcm) HIP_TRY(h, hipMemcpyAsync((pcm_direct ? out_pcm : h->h_pcm_out) + o0, h->d_pcm_out + o0, (size_t)nr * h->out_len * sizeof(int16_t), hipMemcpyDeviceToHost, s));
                if (out_f32) HIP_TRY(h, hipMemcpyAsync((f32_direct ? out_f32 : h->h_f32_out) + o0, h->d_f32_out + o0, (size_t)nr * h->out_len * sizeof(float), hipMemcpyDeviceToHost, s));
            }
            for (int k = 0; k < used; ++k) HIP_TRY(h, hipStreamSynchronize(h->sub_streams[k]));
            st = exchange_status(h, "ade_process", false);
            for (int k = 0; k < kXErrWords; ++k) h->xl_B[k] = 0;                 // (every other launch reports into word 0 with the whole batch's block numbering)
            if (st != ADE_OK) {
                if (out_pcm && pcm_direct) memset(out_pcm, 0, nout * sizeof(int16_t));
                if (out_f32 && f32_direct) memset(out_f32, 0, nout * sizeof(float));
                return st;
            }
            if (out_pcm && !pcm_direct) memcpy(out_pcm, h->h_pcm_out, nout * sizeof(int16_t));
            if (out_f32 && !f32_direct) memcpy(out_f32, h->h_f32_out, nout * sizeof(float));
            return ADE_OK;
        }
    }
    const int16_t* src = in;
    if (!in_direct) { memcpy(h->h_pcm_in, in, nin * sizeof(int16_t)); src = h->h_pcm_in; }
    HIP_TRY(h, hipMemcpyAsync(h->d_pcm_in, src, nin * sizeof(int16_t), hipMemcpyHostToDevice, h->stream));
    st = run(h, h->stream, h->d_pcm_in, rows, h->d_pcm_out, out_f32 ? h->d_f32_out : nullptr);
    if (st != ADE_OK) return st;
    if (out_pcm) HIP_TRY(h, hipMemcpyAsync(pcm_direct ? out_pcm : h->h_pcm_out, h->d_pcm_out, nout * sizeof(int16_t), hipMemcpyDeviceToHost, h->stream));
    if (out_f32) HIP_TRY(h, hipMemcpyAsync(f32_direct ? out_f32 : h->h_f32_out, h->d_f32_out, nout * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    st = exchange_status(h, "ade_process", false);
    if (st != ADE_OK) {   // a page-locked destination has already received the failed launch's bytes: blank it
        if (out_pcm && pcm_direct) memset(out_pcm, 0, nout * sizeof(int16_t));
        if (out_f32 && f32_direct) memset(out_f32, 0, nout * sizeof(float));
        return st;
    }
    if (out_pcm && !pcm_direct) memcpy(out_pcm, h->h_pcm_out, nout * sizeof(int16_t));
    if (out_f32 && !f32_direct) memcpy(out_f32, h->h_f32_out, nout * sizeof(float));
    return ADE_OK;
}

// ---- the pipelined host entry (include/ade.h: ade_submit / ade_wait) -----------------------------------------------------------------------------------------------------
// The reference's driver times a LOOP of session runs over the slices of a file (Inference_GTCRN_ONNX.py:314-333); ade_process is one such run and pays copy-in + kernels +
// copy-out in sequence.  A submission is the same call cut into its three legs on three streams, tied by events: s_pin [H2D -> ev_in], stream [wait ev_in -> kernels -> ev_k],
// s_pout [wait ev_k -> D2H -> ev_out].  With two or more submissions in flight the copy engines move call k + 1 in and call k - 1 out under call k's kernels; the kernels of
// consecutive calls are serialised on `stream`, so the workspace and the exchange area are shared exactly as between two ade_process calls.  Each slot of the ring owns its
// device PCM buffers (a call's input must survive until its kernels ran, its output until it was copied out).
namespace {
void pipe_free(ade_engine* e) {
    for (auto& sl : e->pipe) {
        if (sl.d_in) hipFree(sl.d_in);
        if (sl.d_out) hipFree(sl.d_out);
        if (sl.d_f32) hipFree(sl.d_f32);
        if (sl.h_in) hipHostFree(sl.h_in);
        if (sl.h_out) hipHostFree(sl.h_out);
        if (sl.h_f32) hipHostFree(sl.h_f32);
        sl.d_in = sl.d_out = sl.h_in = sl.h_out = nullptr; sl.d_f32 = sl.h_f32 = nullptr;
    }
    e->pipe_capacity = 0;
}
// A submission's copy-out is enqueued LATE: behind the next submission's copy-in, or when it is waited for.  The runtime serves its copy engines in order of submission
// across streams (measured: ade_process's sub-batch path), so a copy-out enqueued at submit time -- it waits for the kernels -- holds back the NEXT
// call's copy-in, and the three legs run one after the other (0.76 ms per 256 x 1 s batch instead of 0.42).
hipError_t pipe_copy_out(ade_engine* h, ade_engine::PipeSlot& sl) {
    if (!sl.d2h_pending) return hipSuccess;
    sl.d2h_pending = false;
    const size_t nout = (size_t)sl.rows * h->out_len;
    hipError_t e = hipStreamWaitEvent(h->s_pout, sl.ev_k, 0);
    if (e == hipSuccess && sl.dst_pcm) e = hipMemcpyAsync(sl.dst_pcm, sl.d_out, nout * sizeof(int16_t), hipMemcpyDeviceToHost, h->s_pout);
    if (e == hipSuccess && sl.dst_f32) e = hipMemcpyAsync(sl.dst_f32, sl.d_f32, nout * sizeof(float), hipMemcpyDeviceToHost, h->s_pout);
    if (e == hipSuccess) e = hipEventRecord(sl.ev_out, h->s_pout);
    return e;
}
// Finish a slot in flight: wait for its copy-out, check the launch's time-out word, hand a pageable caller its bytes.  The status is kept for ade_wait.
void pipe_finish(ade_engine* h, ade_engine::PipeSlot& sl) {
    if (sl.state != 1) return;
    ade_status st = ADE_OK;
    hipError_t e = hipSuccess;
    for (unsigned long long t = h->pipe_next > (unsigned long long)ade_engine::kMaxPipe ? h->pipe_next - ade_engine::kMaxPipe : 1; t <= sl.ticket && e == hipSuccess; ++t)
        for (auto& o : h->pipe)                        // (copy-outs leave in ticket order: this one and every older one still pending)
            if (o.state == 1 && o.ticket == t) e = pipe_copy_out(h, o);
    if (e == hipSuccess) e = hipEventSynchronize(sl.ev_out);
    if (e != hipSuccess) st = fail(h, ADE_ERR_DEVICE, std::string("ade_wait: hipEventSynchronize: ") + hipGetErrorString(e));
    if (st == ADE_OK) st = exchange_status(h, "ade_wait", false);
    const size_t nout = (size_t)sl.rows * h->out_len;
    if (st == ADE_OK) {
        if (sl.out_pcm) memcpy(sl.out_pcm, sl.h_out, nout * sizeof(int16_t));
        if (sl.out_f32) memcpy(sl.out_f32, sl.h_f32, nout * sizeof(float));
    } else {
        // a time-out drained the device and lowered every flag (exchange_status): the other submissions in flight ran on a disturbed exchange area -- they fail with it
        for (auto& o : h->pipe)
            if (&o != &sl && o.state == 1) { o.d2h_pending = false; (void)hipStreamSynchronize(h->stream); o.state = 2; o.status = st; o.error = h->last_error + " (a submission in flight beside the one that failed)"; }
    }
    sl.state = 2;
    sl.status = st;
    sl.error = st == ADE_OK ? std::string() : h->last_error;
}
}  // namespace

ade_status ade_submit(ade_handle h, const int16_t* in, int batch, int16_t* out_pcm, float* out_f32, uint64_t* ticket) {
    if (!h) return ADE_ERR_BAD_VALUE;
    if (!ticket || batch <= 0 || !in || (!out_pcm && !out_f32)) return fail(h, ADE_ERR_BAD_VALUE, "ade_submit: bad arguments");
    if (h->gt_float_in) return fail(h, ADE_ERR_BAD_VALUE, "ade_submit: this handle's input_audio_dtype is F32 / F16 (use ade_process_f32)");
    if (h->profile) return fail(h, ADE_ERR_BAD_VALUE, "ade_submit: not while ade_profile_last is on");
    HIP_TRY(h, hipSetDevice(h->device));
    const int rows = batch * h->n_win;
    bool busy = false;
    for (const auto& sl : h->pipe) busy = busy || sl.state == 1;
    if (rows > h->capacity || rows > h->pipe_capacity) {      // growing: nothing may be in flight while buffers are re-allocated
        for (auto& sl : h->pipe) pipe_finish(h, sl);
        busy = false;
    }
    if (!busy) { const ade_status xs = exchange_status(h, "ade_submit", true); if (xs != ADE_OK) return xs; }     // (an earlier device-pointer call's failure)
    ade_status st = reserve(h, rows);
    if (st != ADE_OK) return st;
    if (!h->s_pin) {
        HIP_TRY(h, hipStreamCreateWithFlags(&h->s_pin, hipStreamNonBlocking));
        HIP_TRY(h, hipStreamCreateWithFlags(&h->s_pout, hipStreamNonBlocking));
        for (auto& sl : h->pipe) {
            HIP_TRY(h, hipEventCreateWithFlags(&sl.ev_in, hipEventDisableTiming));
            HIP_TRY(h, hipEventCreateWithFlags(&sl.ev_k, hipEventDisableTiming));
            HIP_TRY(h, hipEventCreateWithFlags(&sl.ev_out, hipEventDisableTiming));
        }
    }
    if (rows > h->pipe_capacity) {
        pipe_free(h);
        const size_t nin = (size_t)rows * h->in_len, nout = (size_t)rows * h->out_len;
        for (auto& sl : h->pipe) {
            HIP_TRY(h, hipMalloc((void**)&sl.d_in, nin * sizeof(int16_t)));
            HIP_TRY(h, hipMalloc((void**)&sl.d_out, nout * sizeof(int16_t)));
            HIP_TRY(h, hipMalloc((void**)&sl.d_f32, nout * sizeof(float)));
            HIP_TRY(h, hipHostMalloc((void**)&sl.h_in, nin * sizeof(int16_t), hipHostMallocDefault));
            HIP_TRY(h, hipHostMalloc((void**)&sl.h_out, nout * sizeof(int16_t), hipHostMallocDefault));
            HIP_TRY(h, hipHostMalloc((void**)&sl.h_f32, nout * sizeof(float), hipHostMallocDefault));
        }
        h->pipe_capacity = rows;
    }
    const unsigned long long t = h->pipe_next;
    // ANY free slot of the first pipe_depth (tickets may be waited for in any order, so the free one need not be t % depth).  None free: the oldest submission in flight is
    // completed (its status still waits for ade_wait) and the call is refused -- at most pipe_depth tickets may be outstanding.
    ade_engine::PipeSlot* slp = nullptr;
    for (int i = 0; i < h->pipe_depth && !slp; ++i) if (h->pipe[i].state == 0) slp = &h->pipe[i];
    if (!slp) {
        ade_engine::PipeSlot* oldest = nullptr;
        for (int i = 0; i < h->pipe_depth; ++i) if (!oldest || h->pipe[i].ticket < oldest->ticket) oldest = &h->pipe[i];
        if (oldest->state == 1) pipe_finish(h, *oldest);
        return fail(h, ADE_ERR_BAD_VALUE, "ade_submit: ticket " + std::to_string(oldest->ticket) + " has not been waited for (at most pipe_depth submissions between waits)");
    }
    ade_engine::PipeSlot& sl = *slp;
    auto page_locked = [](const void* p) {
        hipPointerAttribute_t a;
        if (!p || hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
        return a.type == hipMemoryTypeHost;
    };
    const bool in_direct = page_locked(in), pcm_direct = page_locked(out_pcm), f32_direct = page_locked(out_f32);
    const size_t nin = (size_t)rows * h->in_len, nout = (size_t)rows * h->out_len;
    const int16_t* src = in;
    if (!in_direct) { memcpy(sl.h_in, in, nin * sizeof(int16_t)); src = sl.h_in; }
    HIP_TRY(h, hipMemcpyAsync(sl.d_in, src, nin * sizeof(int16_t), hipMemcpyHostToDevice, h->s_pin));
    HIP_TRY(h, hipEventRecord(sl.ev_in, h->s_pin));
    HIP_TRY(h, hipStreamWaitEvent(h->stream, sl.ev_in, 0));
    st = run(h, h->stream, sl.d_in, rows, out_pcm ? sl.d_out : nullptr, out_f32 ? sl.d_f32 : nullptr);
    if (st != ADE_OK) { (void)hipStreamSynchronize(h->s_pin); (void)hipStreamSynchronize(h->stream); return st; }
    // From here on this call's kernels are enqueued into the slot's buffers: a HIP failure must not return with them running and the slot unmarked.  Every stream is
    // drained, every submission in flight fails with this one (their copies may not have been enqueued), and no ticket is issued.
    hipError_t pe = hipEventRecord(sl.ev_k, h->stream);
    // the copy-outs of the OLDER submissions go out now, behind this one's copy-in; this one's waits for the next submission or for its ade_wait
    for (unsigned long long o = t > (unsigned long long)ade_engine::kMaxPipe ? t - ade_engine::kMaxPipe : 1; o < t && pe == hipSuccess; ++o)
        for (auto& other : h->pipe)
            if (other.state == 1 && other.ticket == o && pe == hipSuccess) pe = pipe_copy_out(h, other);
    if (pe != hipSuccess) {
        (void)hipStreamSynchronize(h->s_pin); (void)hipStreamSynchronize(h->stream); (void)hipStreamSynchronize(h->s_pout);
        const ade_status fs = fail(h, ADE_ERR_DEVICE, std::string("ade_submit: after the launch: ") + hipGetErrorString(pe));
        for (auto& other : h->pipe)
            if (other.state == 1) { other.d2h_pending = false; other.state = 2; other.status = fs; other.error = h->last_error + " (a submission in flight beside the one that failed)"; }
        return fs;
    }
    sl.dst_pcm = out_pcm ? (pcm_direct ? out_pcm : sl.h_out) : nullptr;
    sl.dst_f32 = out_f32 ? (f32_direct ? out_f32 : sl.h_f32) : nullptr;
    sl.d2h_pending = true;
    (void)nout;
    sl.ticket = t; sl.state = 1; sl.rows = rows;
    sl.out_pcm = (out_pcm && !pcm_direct) ? out_pcm : nullptr;
    sl.out_f32 = (out_f32 && !f32_direct) ? out_f32 : nullptr;
    sl.status = ADE_OK; sl.error.clear();
    h->pipe_next = t + 1;
    *ticket = (uint64_t)t;
    return ADE_OK;
}

ade_status ade_wait(ade_handle h, uint64_t ticket) {
    if (!h) return ADE_ERR_BAD_VALUE;
    for (auto& sl : h->pipe) {
        if (sl.state == 0 || sl.ticket != (unsigned long long)ticket) continue;
        HIP_TRY(h, hipSetDevice(h->device));
        pipe_finish(h, sl);
        const ade_status st = (ade_status)sl.status;
        if (st != ADE_OK) h->last_error = sl.error;
        sl.state = 0;
        return st;
    }
    return fail(h, ADE_ERR_BAD_VALUE, "ade_wait: unknown ticket " + std::to_string((unsigned long long)ticket) + " (never issued, or already waited for)");
}

ade_status ade_debug_tap(ade_handle h, const char* name, float* out, size_t count, size_t* written) {
    if (!h || !name || !out || !written) return ADE_ERR_BAD_VALUE;
    if (h->sub) {
        std::string derr;
        HIP_TRY(h, hipSetDevice(h->device));
        const int rc = h->sub->tap(h->stream, name, h->last_batch, out, count, written, derr);
        return rc == ADE_OK ? ADE_OK : fail(h, (ade_status)rc, derr);
    }
    const size_t nfr = (size_t)h->last_batch * h->T;
    struct Tap { const char* name; const float* p; size_t n; };
    const Tap taps[] = {
        {"mean", h->mean, (size_t)h->last_batch}, {"spec", h->spec, nfr * 2 * kBinsPad}, {"feat", h->feat, nfr * 3 * kErbPad},
        {"e0", h->e0, nfr * kF1 * kCh}, {"e1", h->e1, nfr * kFw * kCh}, {"h", h->h, nfr * kFw * kCh}, {"zt", h->zt, nfr * 8},
        {"x_e2", h->xe[0], nfr * kFw * kCh}, {"x_e3", h->xe[1], nfr * kFw * kCh}, {"x_e4", h->xe[2], nfr * kFw * kCh},
        {"at_e2", h->ate[0], nfr * 8}, {"at_e3", h->ate[1], nfr * 8}, {"at_e4", h->ate[2], nfr * 8},
        {"x_d0", h->xd[0], nfr * kFw * kCh}, {"x_d1", h->xd[1], nfr * kFw * kCh}, {"x_d2", h->xd[2], nfr * kFw * kCh},
        {"at_d0", h->atd[0], nfr * 8}, {"at_d1", h->atd[1], nfr * 8}, {"at_d2", h->atd[2], nfr * 8},
        {"rnn", h->rnn, nfr * kFw * kCh}, {"dp1_mid", h->dpm[0], nfr * kFw * kCh}, {"dp1", h->dpo[0], nfr * kFw * kCh},
        {"dp2_mid", h->dpm[1], nfr * kFw * kCh}, {"dp2", h->dpo[1], nfr * kFw * kCh}, {"d3", h->d3, nfr * kF1 * kCh},
        {"mask", h->mask, nfr * 2 * kErbPad}, {"frames", h->frames, nfr * kNfft}};
    if (strcmp(name, "phase_clock") == 0) {   // wall_clock64() stamps of workgroup 0's phases (profile mode), as tick deltas
        if (count < 64) return fail(h, ADE_ERR_SHAPE_MISMATCH, "tap buffer too small");
        // Per-stage kernels (profile mode 1) use the first 64 slots: gtblock 0-15, dpgrnn 16-31, front 32-47, back 48-63.
        // The single-launch clock build (mode 3) uses one such 64-slot page per stage: [front | enc 0-2 | dp 0-1 | dec 0-2 | back].
        // Stamps are returned relative to their group of 16; slots 8-15 of the front / back groups are per-phase
        // accumulators over the tile loops (raw tick sums).
        // One 640-slot set per segment (the first two segments of chunk 0 are clocked).
        const size_t n = std::min(count, (size_t)kClkSlots) / 64 * 64;
        long long raw[kClkSlots];
        HIP_TRY(h, hipMemcpy(raw, h->d_clk, n * sizeof(long long), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < n; ++i)
            out[i] = (i & 15) >= 8 && (i & 63) >= 32 ? (float)raw[i] : (float)(raw[i] - raw[(i / 16) * 16]);
        *written = n;
        return ADE_OK;
    }
    if (strcmp(name, "phase_clock_abs") == 0) {   // the same stamps in ticks since segment 0 entered the front stage (slot 32); -1 = never stamped
        const size_t n = std::min(count, (size_t)kClkSlots) / 64 * 64;
        if (n < 64) return fail(h, ADE_ERR_SHAPE_MISMATCH, "tap buffer too small");
        long long raw[kClkSlots];
        HIP_TRY(h, hipMemcpy(raw, h->d_clk, n * sizeof(long long), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < n; ++i) out[i] = raw[i] ? (float)(raw[i] - raw[32]) : -1.0f;
        *written = n;
        return ADE_OK;
    }
    if (strcmp(name, "fused_geometry") == 0) {    // [geometry of the last fused call (-1: none), workgroups per chunk]
        if (count < 1) return fail(h, ADE_ERR_SHAPE_MISMATCH, "tap buffer too small");
        out[0] = h->last_fused ? (float)h->last_geometry : -1.0f;
        if (count > 1) out[1] = h->last_fused ? (float)fused_segments(h->T, h->last_geometry) : 0.0f;
        *written = count > 1 ? 2 : 1;
        return ADE_OK;
    }
    if (strcmp(name, "xwait_retries") == 0) {     // calls ade_process re-ran without hand-offs after a time-out (option "xwait_retry")
        if (count < 1) return fail(h, ADE_ERR_SHAPE_MISMATCH, "tap buffer too small");
        out[0] = (float)h->retries;
        *written = 1;
        return ADE_OK;
    }
    if (strcmp(name, "xchg_error") == 0) {        // non-zero (the dev::xcode of the wait) after a bounded inter-workgroup wait of the segmented fused path gave up; sticky until an entry point reports it
        if (count < 1) return fail(h, ADE_ERR_SHAPE_MISMATCH, "tap buffer too small");
        int v = 0;
        if (h->d_xerr) {
            HIP_TRY(h, hipStreamSynchronize(h->stream));
            for (int k = 0; k < kXErrWords && !v; ++k) v = ((volatile int*)h->d_xerr)[k];
        }
        out[0] = (float)v;
        *written = 1;
        return ADE_OK;
    }
    for (const Tap& t : taps)
        if (strcmp(t.name, name) == 0) {
            if (!t.p || t.n == 0) return fail(h, ADE_ERR_NOT_FOUND, "tap has no data yet");
            if (h->last_fused)
                for (const char* lds_only : {"mean", "feat", "h", "zt", "rnn", "dp1_mid", "dp2_mid", "frames", "d3", "mask"})
                    if (strcmp(name, lds_only) == 0)
                        return fail(h, ADE_ERR_NOT_FOUND, std::string("tap lives in LDS on the fused path (set option fused=0): ") + name);
            if (count < t.n) return fail(h, ADE_ERR_SHAPE_MISMATCH, "tap buffer too small");
            HIP_TRY(h, hipSetDevice(h->device));
            HIP_TRY(h, hipStreamSynchronize(h->stream));
            if (h->last_fused && strncmp(name, "at_", 3) == 0) {
                for (size_t i = 0; i < t.n; ++i) out[i] = 1.0f;   // fused stages apply the TRA gate in-kernel: x_* is already gated
            } else {
                HIP_TRY(h, hipMemcpy(out, t.p, t.n * sizeof(float), hipMemcpyDeviceToHost));
                const bool is16 = !strcmp(name, "e0") || !strcmp(name, "e1") || !strcmp(name, "d3") || !strncmp(name, "x_", 2) ||
                                  !strcmp(name, "dp1") || !strcmp(name, "dp2");
                if (h->last_fused && is16) {
                    // fused path keeps (B,.,.,16) tensors channel-quad planar [b][q][p][4]: hand back channels-last [b][p][16]
                    const size_t per = t.n / (size_t)h->last_batch, P = per / 16;
                    std::vector<float> tmp(per);
                    for (int b = 0; b < h->last_batch; ++b) {
                        float* o = out + (size_t)b * per;
                        memcpy(tmp.data(), o, per * sizeof(float));
                        for (size_t q4 = 0; q4 < 4; ++q4)
                            for (size_t p = 0; p < P; ++p)
                                for (int c = 0; c < 4; ++c) o[p * 16 + q4 * 4 + c] = tmp[(q4 * P + p) * 4 + c];
                    }
                }
            }
            *written = t.n;
            return ADE_OK;
        }
    return fail(h, ADE_ERR_MISSING_KEY, std::string("unknown tap: ") + name);
}

int ade_kernel_count(ade_handle h) { return h ? (int)h->stats.size() : 0; }
const char* ade_kernel_name(ade_handle h, int i) { return (h && i >= 0 && i < (int)h->stats.size()) ? h->stats[i].name.c_str() : ""; }
ade_status ade_profile_last(ade_handle h, int enable) {
    if (!h) return ADE_ERR_BAD_VALUE;
    h->profile = enable != 0;
    if (enable) h->profile_mode = (enable == 2 || enable == 3) ? enable : 1;
    return ADE_OK;
}
ade_status ade_kernel_ms(ade_handle h, int i, float* total_ms, int* launches) {
    if (!h || i < 0 || i >= (int)h->stats.size()) return ADE_ERR_BAD_VALUE;
    if (total_ms) *total_ms = h->stats[i].ms;
    if (launches) *launches = h->stats[i].launches;
    return ADE_OK;
}

const char* ade_last_error(ade_handle h) { return h ? h->last_error.c_str() : g_create_error.c_str(); }

static void ade_orphan_streams(ade_handle h);

void ade_destroy(ade_handle h) {
    if (!h) return;
    if (h->stream) {
        hipSetDevice(h->device);
        hipStreamSynchronize(h->stream);
    }
    ade_orphan_streams(h);
    for (auto& sl : h->pipe) {
        if (sl.state == 1 && sl.ev_out && !sl.d2h_pending) hipEventSynchronize(sl.ev_out);
        if (sl.ev_in) hipEventDestroy(sl.ev_in);
        if (sl.ev_k) hipEventDestroy(sl.ev_k);
        if (sl.ev_out) hipEventDestroy(sl.ev_out);
    }
    pipe_free(h);
    if (h->s_pin) hipStreamDestroy(h->s_pin);
    if (h->s_pout) hipStreamDestroy(h->s_pout);
    free_workspace(h);
    delete h->sub;
    for (auto& ev : h->events) {
        hipEventDestroy(ev.first);
        hipEventDestroy(ev.second);
    }
    if (h->d_weights) hipFree(h->d_weights);
    if (h->d_ints) hipFree(h->d_ints);
    if (h->d_clk) hipFree(h->d_clk);
    if (h->d_xerr) hipHostFree(h->d_xerr);
    for (hipStream_t s : h->sub_streams) if (s) hipStreamDestroy(s);
    if (h->s_in) hipStreamDestroy(h->s_in);
    if (h->s_out) hipStreamDestroy(h->s_out);
    if (h->d_sio_ready) hipFree(h->d_sio_ready);
    if (h->d_sio_count) hipFree(h->d_sio_count);
    if (h->h_sio_epoch) hipHostFree(h->h_sio_epoch);
    if (h->h_sio_done) hipHostFree(h->h_sio_done);
    if (h->stream) hipStreamDestroy(h->stream);
    delete h;
}

ade_status ade_stft_forward(ade_handle h, const float* d_x, int batch, int length, float* d_spec, void* hip_stream) {
    if (h && h->sub) return fail(h, ADE_ERR_UNSUPPORTED, "ade_stft_forward: GTCRN handles only (use ade_stft_create for a generic STFT)");
    if (!h || batch < 0 || (batch > 0 && (!d_x || !d_spec))) return ADE_ERR_BAD_VALUE;
    if (length < kNfft / 2 + 2) return fail(h, ADE_ERR_SHAPE_MISMATCH, "ade_stft_forward: length too short for reflect padding");
    if (batch == 0) return ADE_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : h->stream;
    launch_stft_ref(s, d_x, batch, length, length / kHop + 1, h->tabs, d_spec);
    HIP_TRY(h, hipGetLastError());
    if (!hip_stream) HIP_TRY(h, hipStreamSynchronize(s));
    return ADE_OK;
}

ade_status ade_istft_forward(ade_handle h, const float* d_spec, int batch, int frames, float* d_y, void* hip_stream) {
    if (h && h->sub) return fail(h, ADE_ERR_UNSUPPORTED, "ade_istft_forward: GTCRN handles only (use ade_stft_create for a generic STFT)");
    if (!h || batch < 0 || frames < 2 || (batch > 0 && (!d_spec || !d_y))) return ADE_ERR_BAD_VALUE;
    if (batch == 0) return ADE_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : h->stream;
    float* tmp = nullptr;
    HIP_TRY(h, hipMalloc((void**)&tmp, (size_t)batch * frames * kNfft * sizeof(float)));
    launch_istft_ref(s, d_spec, batch, frames, h->tabs, tmp);
    launch_ola_pcm(s, tmp, h->tabs, batch, frames, nullptr, d_y);
    hipError_t err = hipGetLastError();
    hipStreamSynchronize(s);
    hipFree(tmp);
    if (err != hipSuccess) return fail(h, ADE_ERR_DEVICE, hipGetErrorString(err));
    return ADE_OK;
}


// ---- streaming entry points ---------------------------------------------------------------------------------------------
ade_status ade_stream_create(ade_handle h, int n_streams, int frames_per_push, ade_stream_handle* out) {
    if (!out) return fail(h, ADE_ERR_BAD_VALUE, "ade_stream_create: out is NULL");
    *out = nullptr;
    if (!h) return ADE_ERR_BAD_VALUE;
    if (h->sub || h->n_win != 1 || h->gt_sand) return fail(h, ADE_ERR_UNSUPPORTED, "ade_stream_create: streaming is implemented for plain GTCRN handles (int16 audio at the model rate)");
    if (n_streams < 1 || frames_per_push < 2 || frames_per_push > 4096)
        return fail(h, ADE_ERR_BAD_VALUE, "ade_stream_create: need n_streams >= 1 and 2 <= frames_per_push <= 4096 (the first push reflects 257 samples)");
    HIP_TRY(h, hipSetDevice(h->device));
    ade_stream* st = new ade_stream();
    st->e = h; st->device = h->device; st->S = n_streams; st->N = frames_per_push;
    const size_t S = (size_t)n_streams, nfr = S * frames_per_push, P = (size_t)frames_per_push * kHop;
    auto bail = [&](const char* what) { h->last_error = what; ade_stream_destroy(st); return ADE_ERR_DEVICE; };
    // state
    const GtConvW* gts[6] = {&h->en_gt[0], &h->en_gt[1], &h->en_gt[2], &h->de_gt[0], &h->de_gt[1], &h->de_gt[2]};
    size_t total = (S + 63) & ~(size_t)63;
    size_t o_hist[6][2], o_tra[6], o_inter[2], o_carry;
    auto take = [&](size_t n) { const size_t at = total; total += (n + 63) & ~(size_t)63; return at; };
    for (int i = 0; i < 6; ++i) {
        const size_t n = S * 2 * gts[i]->dilation * kFw * kCh;
        o_hist[i][0] = take(n); o_hist[i][1] = take(n);
        o_tra[i] = take(S * 16);
    }
    for (int i = 0; i < 2; ++i) o_inter[i] = take(S * kFw * kCh);
    o_carry = take(S * kHop);
    st->state_floats = total;
    if (hipMalloc((void**)&st->state, total * sizeof(float)) != hipSuccess) return bail("ade_stream_create: hipMalloc of the stream state failed");
    st->dc = st->state;
    for (int i = 0; i < 6; ++i) { st->hist[i][0] = st->state + o_hist[i][0]; st->hist[i][1] = st->state + o_hist[i][1]; st->tra_h[i] = st->state + o_tra[i]; }
    for (int i = 0; i < 2; ++i) st->inter_h[i] = st->state + o_inter[i];
    st->carry = st->state + o_carry;
    // activations of one push
    struct Carve { float** p; size_t n; };
    std::vector<Carve> cs = {{&st->spec, nfr * 2 * kBinsPad}, {&st->feat, nfr * 3 * kErbPad}, {&st->e0, nfr * kF1 * kCh}, {&st->e1, nfr * kFw * kCh},
                             {&st->h, nfr * kFw * kCh}, {&st->zt, nfr * 8}, {&st->rnn, nfr * kFw * kCh}, {&st->d3, nfr * kF1 * kCh},
                             {&st->mask, nfr * 2 * kErbPad}, {&st->frames, nfr * kNfft}};
    for (int i = 0; i < 3; ++i) {
        cs.push_back({&st->xe[i], nfr * kFw * kCh}); cs.push_back({&st->ate[i], nfr * 8});
        cs.push_back({&st->xd[i], nfr * kFw * kCh}); cs.push_back({&st->atd[i], nfr * 8});
    }
    for (int i = 0; i < 2; ++i) { cs.push_back({&st->dpm[i], nfr * kFw * kCh}); cs.push_back({&st->dpo[i], nfr * kFw * kCh}); }
    size_t wtotal = 0;
    for (auto& c : cs) wtotal += (c.n + 63) & ~(size_t)63;
    if (hipMalloc((void**)&st->ws, wtotal * sizeof(float)) != hipSuccess) return bail("ade_stream_create: hipMalloc of the push workspace failed");
    size_t at = 0;
    for (auto& c : cs) { *c.p = st->ws + at; at += (c.n + 63) & ~(size_t)63; }
    if (hipMalloc((void**)&st->pcm_prev, S * sizeof(int16_t)) != hipSuccess || hipMalloc((void**)&st->pcm_hist, S * kHop * sizeof(int16_t)) != hipSuccess || hipMalloc((void**)&st->concat, S * (P + kHop) * sizeof(int16_t)) != hipSuccess ||
        hipMalloc((void**)&st->d_in, S * P * sizeof(int16_t)) != hipSuccess || hipMalloc((void**)&st->d_out, S * P * sizeof(int16_t)) != hipSuccess ||
        hipMalloc((void**)&st->d_f32, S * P * sizeof(float)) != hipSuccess ||
        hipHostMalloc((void**)&st->h_in, S * P * sizeof(int16_t), hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void**)&st->h_out, S * P * sizeof(int16_t), hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void**)&st->h_f32, S * P * sizeof(float), hipHostMallocDefault) != hipSuccess)
        return bail("ade_stream_create: allocation of the PCM staging buffers failed");
    // Fused pushes: possible whenever the frame count of a push fits a fused geometry (every push of up to 512 frames) and the engine's fused path is on.
    {
        int geo = -1, geo1 = -1;
        for (int g = fused_geometries() - 1; g >= 0 && geo < 0; --g)
            if ((h->geometry < 0 || h->geometry == g) && fused_supported(frames_per_push, g)) geo = g;
        for (int g = fused_geometries() - 1; g >= 0 && geo1 < 0; --g)
            if (fused_supported(2, g)) geo1 = g;                       // the flush is a one-frame push
        if (h->use_fused && h->use_single && geo >= 0 && geo1 >= 0) {
            const int nseg = fused_segments(frames_per_push, geo);
            ChunkFixed F{};
            F.tabs = h->tabs; F.erb_bm = h->erb_bm; F.erb_bs = h->erb_bs;
            F.en0 = h->en0; F.en1 = h->en1; F.de3 = h->de3; F.de4 = h->de4;
            for (int i = 0; i < 3; ++i) { F.en_gt[i] = h->en_gt[i]; F.de_gt[i] = h->de_gt[i]; F.xe[i] = st->xe[i]; F.xd[i] = st->xd[i]; }
            for (int i = 0; i < 2; ++i) { F.dp[i] = h->dp[i]; F.dpo[i] = st->dpo[i]; }
            F.spec = st->spec; F.e0 = st->e0; F.e1 = st->e1;
            if (hipMalloc((void**)&st->d_fixed, sizeof(ChunkFixed)) != hipSuccess || hipMemcpy(st->d_fixed, &F, sizeof(ChunkFixed), hipMemcpyHostToDevice) != hipSuccess ||
                hipMalloc((void**)&st->xq, S * nseg * (size_t)kXFloats * sizeof(float)) != hipSuccess ||
                hipMalloc((void**)&st->xq_flags, S * nseg * (size_t)kXFlags * sizeof(unsigned)) != hipSuccess ||
                hipMalloc((void**)&st->xstate[0], S * (size_t)kXFloats * sizeof(float)) != hipSuccess || hipMalloc((void**)&st->xstate[1], S * (size_t)kXFloats * sizeof(float)) != hipSuccess ||
                hipMalloc((void**)&st->xstate_flags[0], S * (size_t)kXFlags * sizeof(unsigned)) != hipSuccess ||
                hipMalloc((void**)&st->xstate_flags[1], S * (size_t)kXFlags * sizeof(unsigned)) != hipSuccess ||
                hipHostMalloc((void**)&st->xerr, 4 * sizeof(int), hipHostMallocDefault) != hipSuccess)
                return bail("ade_stream_create: allocation of the fused-push state failed");
            st->xerr[0] = 0;
            st->fused = true; st->geo = geo; st->geo_flush = geo1;
        }
    }
    h->live_streams.push_back(st);
    const ade_status rc = ade_stream_reset(st);
    if (rc != ADE_OK) {
        ade_stream_destroy(st);
        return rc;
    }
    *out = st;
    return ADE_OK;
}

ade_status ade_stream_reset(ade_stream_handle st) {
    if (!st || !st->e) return ADE_ERR_BAD_VALUE;
    ade_engine* h = st->e;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, hipMemset(st->state, 0, st->state_floats * sizeof(float)));
    HIP_TRY(h, hipMemset(st->pcm_hist, 0, (size_t)st->S * kHop * sizeof(int16_t)));
    HIP_TRY(h, hipMemset(st->pcm_prev, 0, (size_t)st->S * sizeof(int16_t)));
    for (int i = 0; i < 6; ++i) st->hist_cur[i] = 0;
    if (st->fused) {           // no flag of an abandoned signal may survive into the next one
        const int nseg = fused_segments(st->N, st->geo);
        HIP_TRY(h, hipMemset(st->xq_flags, 0, (size_t)st->S * nseg * kXFlags * sizeof(unsigned)));
        for (int i = 0; i < 2; ++i) HIP_TRY(h, hipMemset(st->xstate_flags[i], 0, (size_t)st->S * kXFlags * sizeof(unsigned)));
        st->xcur = 0;
        *st->xerr = 0;
    }
    st->first = true;
    st->flushed = false;
    return ADE_OK;
}

ade_status ade_stream_push_device(ade_stream_handle st, const int16_t* d_in, int16_t* d_out_pcm, float* d_out_f32, void* hip_stream) {
    if (!st || !st->e || !d_in || (!d_out_pcm && !d_out_f32)) return ADE_ERR_BAD_VALUE;
    ade_engine* h = st->e;
    if (st->flushed) return fail(h, ADE_ERR_BAD_VALUE, "ade_stream_push: the stream was flushed; reset it first");
    HIP_TRY(h, hipSetDevice(h->device));
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : h->stream;
    enqueue_stream(st, s, d_in, d_out_pcm, d_out_f32);
    HIP_TRY(h, hipGetLastError());
    if (!hip_stream) HIP_TRY(h, hipStreamSynchronize(s));
    return ADE_OK;
}

ade_status ade_stream_push(ade_stream_handle st, const int16_t* in, int16_t* out_pcm, float* out_f32) {
    if (!st || !st->e || !in || (!out_pcm && !out_f32)) return ADE_ERR_BAD_VALUE;
    ade_engine* h = st->e;
    if (st->flushed) return fail(h, ADE_ERR_BAD_VALUE, "ade_stream_push: the stream was flushed; reset it first");
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t n = (size_t)st->S * st->N * kHop;
    memcpy(st->h_in, in, n * sizeof(int16_t));
    HIP_TRY(h, hipMemcpyAsync(st->d_in, st->h_in, n * sizeof(int16_t), hipMemcpyHostToDevice, h->stream));
    enqueue_stream(st, h->stream, st->d_in, st->d_out, out_f32 ? st->d_f32 : nullptr);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipMemcpyAsync(st->h_out, st->d_out, n * sizeof(int16_t), hipMemcpyDeviceToHost, h->stream));
    if (out_f32) HIP_TRY(h, hipMemcpyAsync(st->h_f32, st->d_f32, n * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (st->xerr && *(volatile int*)st->xerr) return fail(h, ADE_ERR_DEVICE, "ade_stream_push: a segment of the fused push timed out waiting for its predecessor");
    if (out_pcm) memcpy(out_pcm, st->h_out, n * sizeof(int16_t));
    if (out_f32) memcpy(out_f32, st->h_f32, n * sizeof(float));
    return ADE_OK;
}

ade_status ade_stream_flush(ade_stream_handle st, int16_t* out_pcm, float* out_f32) {
    if (!st || !st->e || (!out_pcm && !out_f32)) return ADE_ERR_BAD_VALUE;
    ade_engine* h = st->e;
    if (st->first || st->flushed) return fail(h, ADE_ERR_BAD_VALUE, "ade_stream_flush: nothing to flush (no push since the last reset, or already flushed)");
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t n = (size_t)st->S * kHop;
    enqueue_stream(st, h->stream, nullptr, st->d_out, out_f32 ? st->d_f32 : nullptr, /*flush=*/true);
    st->flushed = true;
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipMemcpyAsync(st->h_out, st->d_out, n * sizeof(int16_t), hipMemcpyDeviceToHost, h->stream));
    if (out_f32) HIP_TRY(h, hipMemcpyAsync(st->h_f32, st->d_f32, n * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (st->xerr && *(volatile int*)st->xerr) return fail(h, ADE_ERR_DEVICE, "ade_stream_flush: a segment of the fused push timed out waiting for its predecessor");
    if (out_pcm) memcpy(out_pcm, st->h_out, n * sizeof(int16_t));
    if (out_f32) memcpy(out_f32, st->h_f32, n * sizeof(float));
    return ADE_OK;
}

namespace {
void release_stream_buffers(ade_stream* st) {
    (void)hipSetDevice(st->device);
    (void)hipDeviceSynchronize();
    if (st->state) (void)hipFree(st->state);
    if (st->ws) (void)hipFree(st->ws);
    if (st->pcm_hist) (void)hipFree(st->pcm_hist);
    if (st->pcm_prev) (void)hipFree(st->pcm_prev);
    if (st->concat) (void)hipFree(st->concat);
    if (st->d_in) (void)hipFree(st->d_in);
    if (st->d_out) (void)hipFree(st->d_out);
    if (st->d_f32) (void)hipFree(st->d_f32);
    if (st->h_in) (void)hipHostFree(st->h_in);
    if (st->h_out) (void)hipHostFree(st->h_out);
    if (st->h_f32) (void)hipHostFree(st->h_f32);
    if (st->d_fixed) (void)hipFree(st->d_fixed);
    if (st->xq) (void)hipFree(st->xq);
    if (st->xq_flags) (void)hipFree(st->xq_flags);
    for (int i = 0; i < 2; ++i) { if (st->xstate[i]) (void)hipFree(st->xstate[i]); if (st->xstate_flags[i]) (void)hipFree(st->xstate_flags[i]); st->xstate[i] = nullptr; st->xstate_flags[i] = nullptr; }
    if (st->xerr) (void)hipHostFree(st->xerr);
    st->d_fixed = nullptr; st->xq = nullptr; st->xq_flags = nullptr; st->xerr = nullptr;
    st->state = st->ws = nullptr;
    st->pcm_hist = st->pcm_prev = st->concat = st->d_in = st->d_out = st->h_in = st->h_out = nullptr;
    st->d_f32 = st->h_f32 = nullptr;
}
}  // namespace

void ade_stream_destroy(ade_stream_handle st) {
    if (!st) return;
    if (st->e) {                 // still attached: release the device state and leave the engine's list
        release_stream_buffers(st);
        auto& v = st->e->live_streams;
        v.erase(std::remove(v.begin(), v.end(), st), v.end());
    }
    delete st;
}

// called by ade_destroy: the engine goes away first -> its streams keep only their shell (the caller still owns the handle)
static void ade_orphan_streams(ade_handle h) {
    for (ade_stream* st : h->live_streams) {
        release_stream_buffers(st);
        st->e = nullptr;
    }
    h->live_streams.clear();
}

}  // extern "C"
