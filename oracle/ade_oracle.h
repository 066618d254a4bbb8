/*
 * ade_oracle.h — CPU ORACLE for the GTCRN hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C, fp32 restatement of the reference's algorithm for the path
 *   int16 PCM -> STFT -> GTCRN -> complex mask -> ISTFT/OLA -> int16 PCM
 * (GTCRN/Export_GTCRN.py:636-693, GTCRN/STFT_Process.py:213-341), kept in the
 * reference's own tensor layouts (NCHW, packed re|im spectra, dense windowed
 * DFT tables) so each function reads against the file:line it cites.
 *
 * Pinned (tests/test_oracle_golden.py) against fixtures produced by running the
 * reference itself in the build container (tools/make_golden_gtcrn.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may link
 * or load this library.  The product (libade / audio_denoiser_onnx_amd) never does.
 */
#ifndef ADE_ORACLE_H
#define ADE_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ade_oracle ade_oracle;

/* Parse an ADEWGT01 blob (reference state_dict names, BN already folded) and build the
 * 512/512/256 sqrt-hann STFT tables for a static chunk length `in_len` (T = in_len/256+1). */
int ade_oracle_create(const void* blob, size_t nbytes, int in_len, ade_oracle** out);
void ade_oracle_destroy(ade_oracle* o);
/* test knob: exact (double-angle) DFT tables instead of the reference's fp32-angle tables; see ade_oracle.c */
void ade_oracle_set_exact_dft(ade_oracle* o, int exact);
int ade_oracle_in_len(const ade_oracle* o);
int ade_oracle_out_len(const ade_oracle* o);
const char* ade_oracle_last_error(void);

/* B independent reference calls (per-row DC mean, zero GRU state).  in: [B][in_len] int16;
 * out_pcm: [B][out_len] int16 (may be NULL); out_f32: [B][out_len] pre-PCM-scale waveform (may be NULL).
 * n_threads <= 1 runs serially; > 1 uses OpenMP over chunks. */
int ade_oracle_process(ade_oracle* o, const int16_t* in, int B, int16_t* out_pcm, float* out_f32, int n_threads);

/* USE_BATCH_FOLD=True exports: n_calls calls of n_win windows each ([n_calls][n_win][in_len] -> [n_calls][n_win][out_len]);
 * the DC mean is taken over the whole call before the fold (Export_GTCRN.py:647,656-660). */
int ade_oracle_process_fold(ade_oracle* o, const int16_t* in, int n_calls, int n_win, int16_t* out_pcm, float* out_f32, int n_threads);

/* Taps (reference layouts) of chunk 0 of the most recent ade_oracle_process call. */
int ade_oracle_tap(const ade_oracle* o, const char* name, const float** data, size_t* count);

/* Generic STFT_Process restatement ('stft_B' packed / 'istft_B' packed, static_norm=True).
 * window: "hann","hann_sqrt","hamming" (periodic) ; pad_mode: "reflect" or "constant". */
void ade_oracle_set_generic_exact_dft(int exact);   /* test knob: exact DFT tables in ade_oracle_stft / ade_oracle_istft */
int ade_oracle_stft(const float* x, int B, int L, int n_fft, int win_length, int hop, const char* window,
                    int center_pad, const char* pad_mode, float* out /* [B][2*(n_fft/2+1)][T] */, int* T_out);
int ade_oracle_istft(const float* spec /* [B][2F][T] */, int B, int T, int n_fft, int win_length, int hop,
                     const char* window, int center_pad, float* out /* [B][out_len] */, int* out_len);

#ifdef __cplusplus
}
#endif
#endif
