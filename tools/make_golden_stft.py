#!/usr/bin/env python3
"""STFT_Process golden vectors for the analysis/synthesis configurations of the starred models, produced by RUNNING each
model folder's own STFT_Process copy here (build container only).

    GTCRN       512 / 512 / 256   hann_sqrt (periodic)   centre, reflect      GTCRN/STFT_Process.py
    ZipEnhancer 400 / 400 / 100   hann (periodic)        centre, reflect      ZipEnhancer/STFT_Process.py
    Mel-Band    2048 / 2048 / 441 hann (periodic)        centre, reflect      Mel_Band_Roformer/Stereo/STFT_Process.py
    DFSMN       1920 / 1920 / 960 hamming (symmetric) analysis, hamming_periodic synthesis, no centre pad   DFSMN/STFT_Process.py

Each fixture holds x (B, L), the packed spectrum stft_B(x) (B, 2F, T) and istft_B(spectrum) (B, L_out).

    python tools/make_golden_stft.py      # writes tests/golden/stft_<name>.npz
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, HERE)
from ref_import import import_stft_process  # noqa: E402

GOLD = os.path.join(REPO, "tests", "golden")

CASES = [
    # name, model dir, n_fft, win, hop, analysis window, synthesis window, center, pad_mode, L, static kwarg
    ("gtcrn", "GTCRN", 512, 512, 256, "hann_sqrt", "hann_sqrt", True, "reflect", 6000, "static_norm"),
    ("zipenhancer", "ZipEnhancer", 400, 400, 100, "hann", "hann", True, "reflect", 4000, "static_norm"),
    ("melband", "Mel_Band_Roformer/Stereo", 2048, 2048, 441, "hann", "hann", True, "reflect", 13230, "static_frames"),
    ("dfsmn", "DFSMN", 1920, 1920, 960, "hamming", "hamming_periodic", False, "constant", 12480, "static_norm"),
]


def main():
    for name, mdir, n_fft, win, hop, wa, ws, center, pad, L, static_kw in CASES:
        mod = import_stft_process(mdir)
        torch.manual_seed(1234)
        x = torch.randn(2, 1, L) * 0.25
        stft = mod.STFT_Process("stft_B", n_fft, win, hop, 0, wa, center, pad).eval()
        with torch.inference_mode():
            spec = stft._stft_B_packed_forward(x) if hasattr(stft, "_stft_B_packed_forward") else torch.cat(stft(x), dim=1)
            T = spec.shape[2]
            istft = mod.STFT_Process("istft_B", n_fft, win, hop, T, ws, center, pad, **{static_kw: True}).eval()
            F = n_fft // 2 + 1
            if hasattr(istft, "_istft_B_packed_forward"):
                y = istft._istft_B_packed_forward(spec)
            else:
                y = istft(spec[:, :F], spec[:, F:])
        y = y.reshape(2, -1)
        np.savez_compressed(os.path.join(GOLD, f"stft_{name}.npz"), x=x.numpy().reshape(2, L), spec=spec.numpy(), y=y.numpy(),
                            n_fft=np.int64(n_fft), win_length=np.int64(win), hop=np.int64(hop), center=np.int64(center),
                            analysis_window=np.array(wa), synthesis_window=np.array(ws), pad_mode=np.array(pad))
        err = float((y - x.reshape(2, L)[:, :y.shape[1]]).abs().max()) if center else float("nan")
        print(f"{name}: spec {tuple(spec.shape)} y {tuple(y.shape)} roundtrip err {err:.2e}")


def polar():
    """model_type 'stft_A' (real rows only, :285-296) and 'istft_A' (polar synthesis, :343-361) of GTCRN's copy, on the gtcrn case's input."""
    mod = import_stft_process("GTCRN")
    z = np.load(os.path.join(GOLD, "stft_gtcrn.npz"))
    x = torch.from_numpy(z["x"]).reshape(2, 1, -1)
    spec = torch.from_numpy(z["spec"])
    F, T = 257, spec.shape[2]
    with torch.inference_mode():
        real_a = mod.STFT_Process("stft_A", 512, 512, 256, 0, "hann_sqrt", True, "reflect").eval()(x)
        mag = torch.sqrt(spec[:, :F] ** 2 + spec[:, F:] ** 2)
        phase = torch.atan2(spec[:, F:], spec[:, :F])
        y = mod.STFT_Process("istft_A", 512, 512, 256, T, "hann_sqrt", True, "reflect", static_norm=True).eval()(mag, phase)
    np.savez_compressed(os.path.join(GOLD, "stft_gtcrn_polar.npz"), real_a=real_a.numpy(), magnitude=mag.numpy(), phase=phase.numpy(), y=y.numpy().reshape(2, -1))
    print("polar:", tuple(real_a.shape), tuple(y.shape), float((y.reshape(2, -1) - torch.from_numpy(z["y"])).abs().max()))


def dynamic_tail():
    """The ISTFT of a DYNAMIC_AXES export: built with max_frames above the frame count it is fed and the per-call window sum (static_norm / static_frames False), the
    slice [n_fft / 2 : out_end(max_frames)] keeps the second half of the last frame (GTCRN/STFT_Process.py:337-341; Mel_Band_Roformer/Stereo/STFT_Process.py:296-306).
    On the spectra of the gtcrn and melband cases; the fixture holds only the kept tail (the samples before it equal the static output, checked here)."""
    out = {}
    for name, mdir, n_fft, win, hop, ws, static_kw, max_frames in (("gtcrn", "GTCRN", 512, 512, 256, "hann_sqrt", "static_norm", 4096),
                                                                   ("melband", "Mel_Band_Roformer/Stereo", 2048, 2048, 441, "hann", "static_frames", 2048)):
        mod = import_stft_process(mdir)
        z = np.load(os.path.join(GOLD, f"stft_{name}.npz"))
        spec = torch.from_numpy(z["spec"])
        F, T = n_fft // 2 + 1, spec.shape[2]
        with torch.inference_mode():
            istft = mod.STFT_Process("istft_B", n_fft, win, hop, max_frames, ws, True, "reflect", **{static_kw: False}).eval()
            y = istft._istft_B_packed_forward(spec) if hasattr(istft, "_istft_B_packed_forward") else istft(spec[:, :F], spec[:, F:])
        y = y.numpy().reshape(2, -1)
        static_len = hop * (T - 1)
        assert y.shape[1] == static_len + n_fft // 2, (y.shape, static_len)
        print(f"dynamic {name}: {y.shape}, prefix vs static max diff {np.abs(y[:, :static_len] - z['y']).max():.2e}, tail max {np.abs(y[:, static_len:]).max():.3g}")
        out[name + "_tail"] = y[:, static_len:]
    np.savez_compressed(os.path.join(GOLD, "stft_dynamic_tail.npz"), **out)


if __name__ == "__main__" and "--dynamic" in sys.argv:
    dynamic_tail()
    sys.exit(0)

if __name__ == "__main__":
    main()
    polar()
