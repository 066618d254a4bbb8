/*
 * ade.h — C ABI of libade, the MI355X (gfx950) engine for the reference's per-chunk denoise call.
 *
 * The reference (DakeQQ/Audio-Denoiser-ONNX) has no plugin / FFI layer: its hot-path boundary is the tensor
 * contract of the exported ONNX graph as driven by GTCRN/Inference_GTCRN_ONNX.py:307-317
 *     input  "noisy_audio"    int16 (1, 1, L)         (GTCRN/Export_GTCRN.py:768)
 *     output "denoised_audio" int16 (1, 1, L_out)     (GTCRN/Export_GTCRN.py:769)
 * with static shapes (DYNAMIC_AXES=False, Export_GTCRN.py:28), caller-owned pre-bound buffers re-used for every
 * slice (Inference_GTCRN_ONNX.py:307-311), synchronous single-threaded execution (:156), no state carried between
 * calls (zero GRU state, Export_GTCRN.py:339-352), plus the string metadata map of audio_onnx_metadata.py:8-26.
 * Each entry point below names the reference interface it replaces.  Plain pointers and sizes only.
 *
 * A batch of B rows is B INDEPENDENT reference calls (per-row DC mean, Export_GTCRN.py:647): row b of the output
 * equals what the reference's session.run returns for row b alone.
 */
#ifndef ADE_H
#define ADE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ADE_ABI_VERSION 1

/* Status codes map 1:1 onto the exception classes the reference raises at this boundary
 * (audio_onnx_metadata.py:251-256 KeyError, :281-287 / :325-351 ValueError, :293-297 FileNotFoundError). */
typedef enum ade_status {
    ADE_OK = 0,
    ADE_ERR_NOT_FOUND = 1,      /* FileNotFoundError: weights / manifest carrier missing            */
    ADE_ERR_MISSING_KEY = 2,    /* KeyError: required metadata key or weight tensor absent          */
    ADE_ERR_SHAPE_MISMATCH = 3, /* ValueError: length / channel / tensor-shape disagreement         */
    ADE_ERR_BAD_VALUE = 4,      /* ValueError: malformed value (bad bool, bad blob, bad argument)   */
    ADE_ERR_DEVICE = 5,         /* no usable gfx950 device / HIP runtime failure (no CPU fallback)  */
    ADE_ERR_UNSUPPORTED = 6     /* model family or configuration this build does not implement      */
} ade_status;

typedef struct ade_engine* ade_handle;

/* What session.get_inputs()/get_outputs() report for the bound tensors (Inference_GTCRN_ONNX.py:262-267,276-277). */
typedef struct ade_io_desc {
    int32_t abi_version;
    int32_t in_channels;       /* 1; 2 for Mel-Band-Roformer stereo (Export_MelBandRoformer.py:714)                  */
    int32_t out_channels;      /* = in_channels, except H-GTCRN: 2 in, 1 out (Export_H_GTCRN.py:1181-1182)             */
    int32_t n_outputs;         /* 1 ("denoised_audio"); 2 for MossFormer2-SS ("separated_0/1", Export_MossFormer2_SS_16K.py:689) */
    int32_t in_len;            /* L: static input length in samples                      */
    int32_t out_len;           /* L_out = 256 * (L / 256): 15872 for L = 16000           */
    int32_t in_sample_rate;
    int32_t out_sample_rate;
    int32_t model_sample_rate;
    int32_t frames;            /* T = L / hop + 1 (Export_GTCRN.py:45)                    */
    int32_t max_batch;         /* rows the current workspace holds without re-allocation */
    int32_t device;            /* HIP device ordinal                                      */
} ade_io_desc;

/* Replaces onnxruntime.InferenceSession(model) + load_runtime_metadata/validate_audio_metadata
 * (Inference_GTCRN_ONNX.py:237-239).  `manifest_json`: flat JSON object of string values carrying the reference's
 * metadata key set (audio_onnx_metadata.py:161-203); `weights`: ADEWGT01 blob of the BN-folded tensors under the
 * reference's state_dict names.  `device` must be a gfx950 HIP device ordinal (there is no CPU mode).
 * The manifest key `model_family` selects the engine: "gtcrn" (GTCRN/Export_GTCRN.py), "dfsmn" (DFSMN/Export_DFSMN.py),
 * "mel_band_roformer" (Mel_Band_Roformer/Stereo/Export_MelBandRoformer.py), "mossformer2_ss"
 * (MossFormer2_SS_16K/Export_MossFormer2_SS_16K.py), "zipenhancer" (ZipEnhancer/Export_ZipEnhancer.py), "ul_unas"
 * (UL-UNAS/Export_UL_UNAS.py) or "h_gtcrn" (H-GTCRN/Export_H_GTCRN.py: two microphones in, one channel out); the blob then
 * carries that export's fused buffers. */
ade_status ade_create(const char* manifest_json, const void* weights, size_t weights_nbytes, int device,
                      ade_handle* out);

/* Replaces session.get_inputs()/get_outputs() shape queries. */
ade_status ade_get_io(ade_handle h, ade_io_desc* desc);

/* Replaces one (or B) `_update_ortvalue + run_with_iobinding + output copy` rounds
 * (Inference_GTCRN_ONNX.py:314-317) on caller-owned HOST buffers.  Synchronous.
 *   in        [B][in_channels][in_len] int16 ;  out_pcm [B][n_outputs][out_channels][out_len] int16 (channel-planar) ;
 *   out_f32   optional, same layout as out_pcm, float: the waveform before the PCM tail (scale / clamp / truncate; parity tap). */
ade_status ade_process(ade_handle h, const int16_t* in, int batch, int16_t* out_pcm, float* out_f32);

/* Same call on DEVICE buffers (what a GPU execution provider's io-binding does, Inference_GTCRN_ONNX.py:69-92,
 * 193-198): enqueues on `hip_stream` (a hipStream_t, NULL = the engine's own stream) and returns without
 * synchronising when a stream is given. */
ade_status ade_process_device(ade_handle h, const int16_t* d_in, int batch, int16_t* d_out_pcm, float* d_out_f32,
                              void* hip_stream);

/* The same call PIPELINED: what the reference's driver does around its session is a LOOP of runs over the slices of a file (Inference_GTCRN_ONNX.py:314-333,
 * timed as a whole, :323-343).  ade_submit enqueues one ade_process call -- copy-in, kernels, copy-out on three streams tied by events -- and returns a ticket
 * without waiting; ade_wait(ticket) blocks until that call's output is in the caller's buffers and returns ITS status (a time-out of its launch included).
 * Up to `pipe_depth` (option, 2..4, default 3) submissions are in flight: the copy engines move call k + 1 in and call k - 1 out under call k's kernels, so a
 * file of many batches runs at max(kernel, copies) per batch instead of their sum.  Rules: `in`, `out_pcm`, `out_f32` stay the caller's and must stay valid and
 * untouched from ade_submit until ade_wait of the same ticket (page-locked buffers are DMA'd directly, pageable ones go through the slot's page-locked staging);
 * tickets may be waited for in any order, each exactly once; submitting into a full ring completes its oldest submission first (its status still waits for
 * ade_wait).  Results are bit-identical to ade_process on the same rows.  int16 handles of every model family (float-input manifests: ade_process_f32). */
ade_status ade_submit(ade_handle h, const int16_t* in, int batch, int16_t* out_pcm, float* out_f32, uint64_t* ticket);
ade_status ade_wait(ade_handle h, uint64_t ticket);

/* fp32 audio IN (manifest input_audio_dtype "F32" or "F16": normalised samples, no 2^-15 scale -- GTCRN/Export_GTCRN.py:645-646; an F16 tensor crosses this ABI as
 * fp32, the graph computes in fp32 either way).  Every family: the sub-engine families run such handles through their resampling-edge kernels with the family's input gain
 * (each export script's own IN_AUDIO_DTYPE branch).  The outputs are the same pair as above: out_f32 is the export's F32 / F16 output
 * (no * 32767, no clamp, :680-693), out_pcm its INT16 output; either may be NULL. */
ade_status ade_process_f32(ade_handle h, const float* in, int batch, int16_t* out_pcm, float* out_f32);
ade_status ade_process_device_f32(ade_handle h, const float* d_in, int batch, int16_t* d_out_pcm, float* d_out_f32, void* hip_stream);

/* IEEE-half audio tensors at the boundary (manifest input_audio_dtype and / or output_audio_dtype "F16": `noisy_audio` / `denoised_audio` are float16 graph tensors,
 * GTCRN/Export_GTCRN.py:47-48, 645-646, 691-693; the graph itself computes in fp32 either way).  `in`: the input tensor in the handle's INPUT dtype -- half for an F16 / F32
 * input manifest (normalised samples, widened exactly), int16 PCM otherwise; out_f16: the export's float output rounded to half (`audio_out.to(torch.float16)`,
 * round to nearest even), out_pcm its INT16 output; either may be NULL.  Layouts as above. */
ade_status ade_process_f16(ade_handle h, const void* in, int batch, int16_t* out_pcm, uint16_t* out_f16);
ade_status ade_process_device_f16(ade_handle h, const void* d_in, int batch, int16_t* d_out_pcm, uint16_t* d_out_f16, void* hip_stream);

/* The stitch step of a multi-GPU job (chunks are dealt to the ranks in contiguous blocks, the only exchange is the final concatenation,
 * Inference_GTCRN_ONNX.py:326-332): all-gather this rank's `rows` output rows, d_local [rows][n_outputs * out_channels * out_len] int16, into
 * d_all [world * rows][...] on every rank over the CALLER's RCCL communicator (`nccl_comm` = its ncclComm_t), enqueued on `hip_stream` behind the
 * ade_process_device call that filled d_local -- one ncclAllGather of bytes (RCCL has no int16 type).  librccl is opened on first use; ADE_ERR_UNSUPPORTED when
 * it is not installed.  Ranks with fewer rows pad their block (the gathered array is then trimmed by the caller). */
ade_status ade_stitch_device(ade_handle h, const int16_t* d_local, int rows, int16_t* d_all, void* nccl_comm, void* hip_stream);

/* Grow the workspace so `batch` rows run without allocation inside the timed path. */
ade_status ade_reserve(ade_handle h, int batch);

/* Options: "graph" = "0"/"1" replay the launch sequence from a captured hipGraph (default 1); "fused" / "single_launch" = "0"/"1" GTCRN's per-chunk LDS-resident path,
 * as one launch; "geometry" = "auto"/"0"/"1"/"2" its workgroup geometry (0: one 1024-thread workgroup per chunk; 1: 512-thread workgroups that each own a run of at most 32
 * frames, two per CU; 2: 256-thread workgroups of at most 16 frames, four per CU -- the default where the frame count allows; identical bits in all three);
 * "stagger_us" (geometry 0), "seg_prio" = "0".."4" (base wave priority by segment; the recurrences always run at priority 3), "wave_swap": measurement knobs, see DESIGN.md.
 * "xwait_ms" = bound of one inter-workgroup wait of the segmented fused path in milliseconds (default 200).  When a wait gives up, the CALL fails with ADE_ERR_DEVICE
 * (message: chunk, segment and the hand-off that did not arrive), no PCM is handed out, every hand-off flag is cleared and the next call starts clean; a launch on a
 * caller-provided stream is not synchronised by the engine, so its failure is reported by the NEXT call on the handle (or by the debug tap "xchg_error").
 * "xwait_retry" = "1" (default) / "0": ade_process (synchronous, host buffers) re-runs a call whose launch timed out ONCE on the path without hand-offs (whole chunks per
 * workgroup: the same bits) and returns ADE_OK -- a pre-empted or profiled GPU must not fail a valid call; ade_last_error then names the retry, the tap "xwait_retries" counts
 * them.  The device-pointer entries and "0" keep the failure.
 * "full_taps" = "0"/"1": 1 launches the debug build of the single-launch kernel, which stores every inter-stage tensor whole (the shipped kernel keeps channels 0-7 of
 * x_d0 / x_d1 / dp2 in LDS -- their only reader is the next block); set it before a call whose ade_debug_tap results are compared channel by channel.
 * "host_stream" = "0".."8": row groups ade_process streams a host batch THROUGH ONE LAUNCH in (GTCRN's fused path): every group's copy-in is followed by a 64 KB copy carrying
 * the call's epoch, the group's workgroups wait for it before their first PCM read, the group's last workgroup tells the host thread, which starts the group's copy-out -- the
 * copies of the other groups run under the launch's arithmetic (0 = per call: four groups from 128 rows; 1 = off; same bits as the device-resident launch).
 * "host_split" = "0".."8": with "host_stream" = "1": sub-batches ade_process cuts a host batch into instead, each with its own launch on its own stream (0 = per call: two
 * for 128 rows or more; 1 = never).  DESIGN.md section 6 has the measurements of both.
 * "xchg_withhold" = "0"/"1": TEST HOOK, makes the first workgroup of the next launches raise its hand-off flags where nobody polls (forces the time-out path). */
ade_status ade_set_option(ade_handle h, const char* key, const char* value);

/* Parity taps: copy a named intermediate of the LAST processed batch to host (engine-native layout, see
 * DESIGN.md "Data layout").  `count` = capacity of `out` in floats; `*written` = floats copied. */
ade_status ade_debug_tap(ade_handle h, const char* name, float* out, size_t count, size_t* written);

/* Timing tap: device time (ms) of the kernels launched by the last ade_process_device/ade_process call, measured with
 * hipEvents on the stream the kernels ran on.  ade_profile_last(h, 2): time the shipped launch sequence as launched
 * (one "gtcrn_chunk" kernel on the fused path); (h, 1): one kernel per network stage + in-kernel phase clocks;
 * (h, 3): the single-launch kernel's phase-clock build (debug tap "phase_clock", 640 slots); (h, 0): off.
 * Index results by ade_kernel_name(i). */
int ade_kernel_count(ade_handle h);
const char* ade_kernel_name(ade_handle h, int index);
ade_status ade_profile_last(ade_handle h, int enable);
ade_status ade_kernel_ms(ade_handle h, int index, float* total_ms, int* launches);

/* Message of the last failing call on this handle (or of ade_create when h == NULL). */
const char* ade_last_error(ade_handle h);

void ade_destroy(ade_handle h);

/* ---- STFT_Process operator (GTCRN/STFT_Process.py:129-341), FFT-based, on device buffers -----------------
 * stft_B packed:  x [B][L] float -> spec [B][2*(n_fft/2+1)][T]   (reference layout, re rows then im rows)
 * istft_B packed: spec [B][2F][T] -> y [B][hop*(T-1)] (center_pad=True, static_norm=True).
 * This build implements n_fft = 512, hop = 256, window "hann_sqrt", center pad, reflect. */
ade_status ade_stft_forward(ade_handle h, const float* d_x, int batch, int length, float* d_spec, void* hip_stream);
ade_status ade_istft_forward(ade_handle h, const float* d_spec, int batch, int frames, float* d_y, void* hip_stream);

/* ---- stateful streaming over the GTCRN path -----------------------------------------------------
 * The reference resets every state at each slice (zero GRU state, zero causal padding, reflected STFT edges:
 * Inference_GTCRN_ONNX.py:307-317 calls a stateless graph), so slice edges are audible and a slice cannot be shorter than the
 * network's memory.  A stream carries, per independent audio stream: the last 256 input samples (STFT overlap), each GTConvBlock's
 * last 2 * dilation frames of depthwise input, the six TRA GRU and two inter-frame GRU hidden states, and the ISTFT overlap.
 * Pushing a long signal in pieces of frames_per_push * 256 samples then produces EXACTLY what the reference's graph would produce
 * on the whole signal in one call (every op of the network is causal in time), with two stated differences: the output is one
 * hop (256 samples, 16 ms) behind the input -- a frame is complete one hop after its centre -- with the stream's first hop zero;
 * and the per-call DC removal of GTCRN_CUSTOM.forward (Export_GTCRN.py:647, a mean over the WHOLE call, not computable causally)
 * is not applied.  GTCRN handles only (not batch-fold, not the other model families).  All streams of a handle advance together.
 * A push of up to 512 frames is ONE kernel launch (the fused chunk kernel continuing from the state its previous launch left); longer pushes, or a
 * stream created while the option "fused" is 0, take the multi-kernel sequence. */
typedef struct ade_stream* ade_stream_handle;
ade_status ade_stream_create(ade_handle h, int n_streams, int frames_per_push, ade_stream_handle* out);
/* in / out: [n_streams][frames_per_push * 256] int16 (out_f32 optional float, pre-PCM-tail), caller-owned HOST buffers; synchronous. */
ade_status ade_stream_push(ade_stream_handle s, const int16_t* in, int16_t* out_pcm, float* out_f32);
/* the same on DEVICE buffers; enqueues on `hip_stream` (NULL = the engine's stream, then synchronous). */
ade_status ade_stream_push_device(ade_stream_handle s, const int16_t* d_in, int16_t* d_out_pcm, float* d_out_f32, void* hip_stream);
/* End of the signal: the last hop (out: [n_streams][256]), computed from the one frame the reference's graph evaluates past the end
 * (its second half is the reflection of the last 257 samples).  Pushes + flush then reproduce the one-shot output sample for sample,
 * shifted by one hop.  The stream must be reset before it is pushed again. */
ade_status ade_stream_flush(ade_stream_handle s, int16_t* out_pcm, float* out_f32);
ade_status ade_stream_reset(ade_stream_handle s);      /* back to a fresh stream (zero state, next push reflects its head) */
void ade_stream_destroy(ade_stream_handle s);           /* before ade_destroy of its engine */

/* ---- generic STFT_Process operator: any n_fft / win_length / hop / window, for the other model families -------
 * Replaces the reference's STFT_Process module in its 'stft_B' (packed) and 'istft_B' (packed, static_norm=True) forms
 * as instantiated by GTCRN (512/512/256 hann_sqrt, GTCRN/Export_GTCRN.py:719-741), ZipEnhancer (400/400/100 hann,
 * ZipEnhancer/Export_ZipEnhancer.py:947-948), Mel-Band-Roformer (2048/2048/441 hann,
 * Mel_Band_Roformer/Stereo/Export_MelBandRoformer.py:695-696) and DFSMN (1920/1920/960 symmetric hamming analysis,
 * periodic hamming synthesis, no centre pad, DFSMN/Export_DFSMN.py:273-274); class at GTCRN/STFT_Process.py:129-341.
 * Window names: "hann", "hann_sqrt", "hamming" (torch periodic=True), with a "_sym" suffix for periodic=False;
 * "hamming_periodic" is accepted as an alias of "hamming".  Layouts are the reference's: x [B][L] float ->
 * spec [B][2*(n_fft/2+1)][T] (re rows, then im rows) -> y [B][out_len].  5-smooth transform sizes (every folder's: 512, 400, 2048, 1920) run as mixed-radix
 * Stockham FFTs in LDS; other sizes as the reference's own dense windowed DFT on the fp32 matrix cores. */
typedef struct ade_stft_plan* ade_stft_handle;
typedef struct ade_stft_config {
    int n_fft, win_length, hop;
    const char* window;             /* analysis window */
    const char* synthesis_window;   /* NULL: same as the analysis window */
    int center_pad;                 /* 1: pad n_fft/2 on both sides (pad_mode), ISTFT trims them again */
    const char* pad_mode;           /* "reflect" | "constant" (NULL = "reflect") */
} ade_stft_config;
ade_status ade_stft_create(const ade_stft_config* cfg, int device, ade_stft_handle* out);
ade_status ade_stft_frames(ade_stft_handle h, int length, int* frames);             /* T for an input of `length` samples */
ade_status ade_stft_output_length(ade_stft_handle h, int frames, int* out_len);     /* ISTFT output length for T frames */
/* The dynamic-length trim of a DYNAMIC_AXES export: the reference's ISTFT slices [n_fft/2 : out_end(max_frames)] and is fed fewer frames than max_frames
 * (UL-UNAS/STFT_Process.py:170-177, :317-326; Mel_Band_Roformer/Stereo/STFT_Process.py:296-306), so the output keeps the second half of the last frame:
 * hop (T - 1) + n_fft / 2 samples, each divided by the squared-window sum of the frames that cover it.  keep_tail = 0 (default) is the static trim hop (T - 1). */
ade_status ade_stft_keep_tail(ade_stft_handle h, int keep_tail);
ade_status ade_stft_analyze(ade_stft_handle h, const float* d_x, int batch, int length, float* d_spec, void* hip_stream);
ade_status ade_stft_synthesize(ade_stft_handle h, const float* d_spec, int batch, int frames, float* d_y, void* hip_stream);
/* The polar form (model_type "istft_A", STFT_Process.py:343-361): magnitude and phase, each [batch][F][T]; real = mag cos(phase),
 * imag = mag sin(phase) are formed inside the GEMM's operand loader. */
ade_status ade_stft_synthesize_polar(ade_stft_handle h, const float* d_mag, const float* d_phase, int batch, int frames, float* d_y, void* hip_stream);
const char* ade_stft_last_error(ade_stft_handle h);   /* h == NULL: the last failing ade_stft_create of this thread */
void ade_stft_destroy(ade_stft_handle h);

#ifdef __cplusplus
}
#endif
#endif /* ADE_H */
