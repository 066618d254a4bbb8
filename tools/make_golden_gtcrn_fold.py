#!/usr/bin/env python3
"""Golden vector for the USE_BATCH_FOLD export mode, produced by RUNNING THE REFERENCE here (build container only).

USE_BATCH_FOLD=True (GTCRN/Export_GTCRN.py:41-45): the graph input is EXPORT_AUDIO_LENGTH = whole fold windows of
FOLD_WINDOW_LENGTH = 24064 samples (1.5 s rounded up to the hop); the DC mean is taken over the whole input, then the
audio is folded to (num_window, 1, W), run as a batch and stitched back (:647,656-660,671-672).

    python tools/make_golden_gtcrn_fold.py      # writes tests/golden/gtcrn_seed0_fold.npz
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)

import make_golden_gtcrn as mg  # noqa: E402
from ref_import import REF_ROOT, import_gtcrn_namespace, import_stft_process  # noqa: E402


def build_fold_reference(seed: int, input_len: int, extra: dict | None = None):
    ns = import_gtcrn_namespace(input_len, dict({"USE_BATCH_FOLD": True}, **(extra or {})))
    assert ns["USE_BATCH_FOLD"] and ns["STATIC_MODEL_BATCH"] == ns["EXPORT_AUDIO_LENGTH"] // ns["FOLD_WINDOW_LENGTH"]
    # the same seeded weights as gtcrn_seed<seed>.adew: weights do not depend on the length constants
    _, custom_plain, _ = mg.build_reference(seed, 16000)
    g = ns["GTCRN"]().eval()
    g.load_state_dict(mg_unfused(seed), strict=True)
    g.prepare_for_export_()
    STFT_Process = import_stft_process("GTCRN").STFT_Process
    stft = STFT_Process("stft_B", ns["NFFT"], ns["WINDOW_LENGTH"], ns["HOP_LENGTH"], 0, ns["WINDOW_TYPE"], True, ns["PAD_MODE"]).eval()
    istft = STFT_Process("istft_B", ns["NFFT"], ns["WINDOW_LENGTH"], ns["HOP_LENGTH"], ns["MAX_SIGNAL_LENGTH"], ns["WINDOW_TYPE"], True,
                         ns["PAD_MODE"], static_norm=True).eval()
    custom = ns["GTCRN_CUSTOM"](g.float(), stft, istft, 16000, 16000, True, ns["FOLD_WINDOW_LENGTH"]).eval()
    return ns, custom


def mg_unfused(seed):
    sd = np.load(os.path.join(mg.GOLD, f"gtcrn_seed{seed}_unfused_state_dict.npz"))
    return {k: torch.from_numpy(sd[k]) for k in sd.files}


def main():
    ns, custom = build_fold_reference(0, 48000)
    W, n_win, L = ns["FOLD_WINDOW_LENGTH"], ns["STATIC_MODEL_BATCH"], ns["EXPORT_AUDIO_LENGTH"]
    wav = mg.load_wav_i16(os.path.join(REF_ROOT, "Test_Examples", "denoise", "gtcrn_mix.wav"))
    pcm = wav[30000:30000 + L].copy()
    pcm[:2000] += 700            # a DC step: the per-call mean differs visibly from the per-window means
    with torch.inference_mode():
        out = custom(torch.from_numpy(pcm.reshape(1, 1, -1))).numpy().reshape(-1)
    np.savez_compressed(os.path.join(mg.GOLD, "gtcrn_seed0_fold.npz"), pcm_in=pcm, pcm_out=out,
                        fold_window_length=np.int64(W), n_windows=np.int64(n_win), export_audio_length=np.int64(L),
                        input_audio_length=np.int64(48000))
    print("fold golden:", W, n_win, L, out.shape, int(np.abs(out.astype(np.int32)).max()))


def fold_float():
    """USE_BATCH_FOLD with float audio tensors (IN / OUT_AUDIO_DTYPE F32): the whole call is centred (:645-647), then folded (:656-660).  BATCH_WINDOW_SECONDS = 0.256
    -> W = 4096 (17 frames); INPUT_AUDIO_LENGTH = 10000 -> 3 windows = 12288 samples.  tests/golden/gtcrn_seed0_fold_f32.npz; weights = gtcrn_seed0.adew."""
    z = np.load(os.path.join(mg.GOLD, "gtcrn_seed0_fold.npz"))
    out = {}
    for tag, din, dout in (("f32_i16", "F32", "INT16"), ("f32_f32", "F32", "F32")):
        ns, custom = build_fold_reference(0, 10000, {"BATCH_WINDOW_SECONDS": 0.256, "IN_AUDIO_DTYPE": din, "OUT_AUDIO_DTYPE": dout})
        W, n_win, L = ns["FOLD_WINDOW_LENGTH"], ns["STATIC_MODEL_BATCH"], ns["EXPORT_AUDIO_LENGTH"]
        assert (W, n_win, L) == (4096, 3, 12288), (W, n_win, L)
        pcm = z["pcm_in"][:L].copy()
        x = (pcm.astype(np.float32) / np.float32(32768.0)).astype(np.float32)
        with torch.inference_mode():
            y = custom(torch.from_numpy(x.reshape(1, 1, -1))).numpy().reshape(-1)
        out["x_in"], out[tag] = x, y
        print(tag, y.shape, y.dtype, float(np.abs(y).max()))
    np.savez_compressed(os.path.join(mg.GOLD, "gtcrn_seed0_fold_f32.npz"), fold_window_length=np.int64(4096), input_audio_length=np.int64(10000),
                        batch_window_seconds=np.float64(0.256), **out)


if __name__ == "__main__" and "--float" in sys.argv:
    fold_float()
    sys.exit(0)

if __name__ == "__main__":
    main()
