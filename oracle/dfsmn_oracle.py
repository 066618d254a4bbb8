"""dfsmn_oracle.py — CPU ORACLE for the DFSMN hot path.  TEST INFRASTRUCTURE ONLY.

A numpy fp32 restatement of ``DFSMN.forward`` (DFSMN/Export_DFSMN.py:180-246) and of the buffers its constructor builds
(:86-178), in the reference's own layouts (channels-first (C, frames), packed re|im spectra), each step citing the
lines it follows.  Pinned (tests/test_dfsmn.py) against fixtures produced by running the reference module itself in the
build container (tools/make_golden_dfsmn.py; seeded fake parameter tree + this package's Kaldi mel bank, see there).
Only tests/ may import this module; the product (libade / audio_denoiser_onnx_amd) never does.
"""
from __future__ import annotations

import numpy as np

F32 = np.float32
KALDI_NFFT, FRAME, HOP, PREEMPH = 2048, 1920, 960, 0.97
NFFT_STFT = 1920


def _hamming(n: int, periodic: bool, dtype) -> np.ndarray:
    """torch.hamming_window(n, periodic, alpha=0.54, beta=0.46): arange * (2 pi / denom) -> cos -> * -beta + alpha."""
    denom = n if periodic else n - 1
    k = np.arange(n, dtype=dtype)
    return (np.cos(k * dtype(2.0 * np.pi / denom)) * dtype(-0.46) + dtype(0.54)).astype(dtype)


def fbank_kernel() -> np.ndarray:
    """(2 * 1025, 1920): per-frame DC removal -> 0.97 pre-emphasis -> symmetric hamming -> 2048-pt DFT, folded into one
    matrix in float64 and rounded once (Export_DFSMN.py:97-120)."""
    n = FRAME
    win = _hamming(n, False, np.float64)
    t = np.arange(n, dtype=np.float64)[None, :]
    freq = np.arange(KALDI_NFFT // 2 + 1, dtype=np.float64)[:, None]
    omega = (2.0 * np.pi / KALDI_NFFT) * freq * t
    out = []
    for basis in (np.cos(omega) * win[None, :], -np.sin(omega) * win[None, :]):
        filt = np.concatenate(((1.0 - PREEMPH) * basis[:, :1] - PREEMPH * basis[:, 1:2], basis[:, 1:-1] - PREEMPH * basis[:, 2:],
                               basis[:, -1:]), axis=1)
        out.append(filt - filt.mean(axis=1, keepdims=True))
    return np.concatenate(out, axis=0).astype(F32)


def stft_kernels(exact: bool = False):
    """Forward (2*961, 1920) with the symmetric hamming and inverse (2*961, 1920) with the periodic hamming, as
    STFT_Process builds them (DFSMN/STFT_Process.py: _build_stft_kernels / _build_istft_kernels; fp32 angles).
    exact=True: exact angles instead (test knob, isolates kernel error from the reference's table error)."""
    n = NFFT_STFT
    fb = n // 2 + 1
    wa, ws = _hamming(n, False, F32), _hamming(n, True, F32)
    if exact:
        k = (np.arange(fb, dtype=np.int64)[:, None] * np.arange(n, dtype=np.int64)[None, :]) % n
        ang = 2.0 * np.pi * k.astype(np.float64) / n
        c, s = np.cos(ang).astype(F32), np.sin(ang).astype(F32)
    else:
        omega = (F32(2.0 * np.pi / n) * np.arange(fb, dtype=F32)[:, None]) * np.arange(n, dtype=F32)[None, :]
        c, s = np.cos(omega).astype(F32), np.sin(omega).astype(F32)
    fwd = np.concatenate((c * wa[None, :], -s * wa[None, :]), axis=0).astype(F32)
    scale = np.full((fb, 1), 2.0, F32)
    scale[0] = 1.0
    scale[fb - 1] = 1.0
    inv_n = F32(1.0 / n)
    inv = np.concatenate(((scale * c * inv_n) * ws[None, :], (scale * -s * inv_n) * ws[None, :]), axis=0).astype(F32)
    return fwd, inv, ws


def interpolate_linear(x, out_len):
    """F.interpolate(x, size=out_len, mode='linear', align_corners=False) over the last axis in fp32: src = (in / out) (dst + 0.5) - 0.5
    clamped at 0, y = (1 - l) x[i0] + l x[min(i0 + 1, in - 1)]."""
    n = x.shape[-1]
    src = np.maximum(F32(n / out_len) * (np.arange(out_len, dtype=F32) + F32(0.5)) - F32(0.5), F32(0.0)).astype(F32)
    i0 = np.minimum(src.astype(np.int64), n - 1)
    i1 = np.minimum(i0 + 1, n - 1)
    l1 = (src - i0.astype(F32)).astype(F32)
    return ((F32(1.0) - l1) * x[..., i0] + l1 * x[..., i1]).astype(F32)


def interpolate_scale(x, scale_factor: float):
    """F.interpolate(x, scale_factor=s, mode='linear', align_corners=False): floor(n * s) samples, src = (1 / s) (dst + 0.5) - 0.5 with the step held in fp32
    (the DYNAMIC_AXES edges, Export_DFSMN.py:186-187, 236-237)."""
    n = x.shape[-1]
    out_len = int(np.floor(float(n) * float(scale_factor)))
    src = np.maximum(F32(1.0 / float(scale_factor)) * (np.arange(out_len, dtype=F32) + F32(0.5)) - F32(0.5), F32(0.0)).astype(F32)
    i0 = np.minimum(src.astype(np.int64), n - 1)
    i1 = np.minimum(i0 + 1, n - 1)
    l1 = (src - i0.astype(F32)).astype(F32)
    return ((F32(1.0) - l1) * x[..., i0] + l1 * x[..., i1]).astype(F32)


def process_dynamic(tensors: dict, pcm: np.ndarray, in_rate: int = 48000, out_rate: int = 48000, exact_dft: bool = False) -> np.ndarray:
    """The DYNAMIC_AXES export on one int16 row of any length (Export_DFSMN.py:28, :186-187, :236-237, :274): scale-factor interpolation to 48 kHz, frames from the
    model-rate length (snip edges), the ISTFT's overlap-add denominator built from that frame count -- the static one of the same count --, scale-factor interpolation out."""
    x = np.asarray(pcm, np.int16).astype(F32)
    if in_rate != 48000:
        x = interpolate_scale(x, float(48000 / in_rate))
    o = DfsmnOracle(tensors, x.shape[-1], exact_dft)
    y = o._one(x, False)
    if out_rate != 48000:
        y = interpolate_scale(y, float(out_rate / 48000))
    return np.clip(y * F32(32768.0), F32(-32768.0), F32(32767.0)).astype(np.int16)


class DfsmnOracle:
    def __init__(self, tensors: dict, in_len: int, exact_dft: bool = False):
        self.w = {k: np.ascontiguousarray(v, F32) for k, v in tensors.items()}
        self.depth = sum(1 for k in self.w if k.startswith("uf_lin_w_"))
        self.in_len = int(in_len)
        self.frames = (self.in_len - FRAME) // HOP + 1
        self.out_len = NFFT_STFT + HOP * (self.frames - 1)
        self.kfb = fbank_kernel()
        self.kst, self.kinv, ws = stft_kernels(exact_dft)
        # static COLA normalisation, centre_pad=False: full raw length (DFSMN/STFT_Process.py static_norm branch)
        wsum = np.zeros(self.out_len, F32)
        for t in range(self.frames):
            wsum[t * HOP:t * HOP + NFFT_STFT] += ws * ws
        self.win_sum = wsum
        self.taps = {}

    def process(self, pcm: np.ndarray):
        pcm = np.ascontiguousarray(pcm, np.int16).reshape(-1, self.in_len)
        out_pcm = np.empty((pcm.shape[0], self.out_len), np.int16)
        out_f32 = np.empty((pcm.shape[0], self.out_len), F32)
        for b in range(pcm.shape[0]):
            out_f32[b] = self._one(pcm[b], b == 0)
            out_pcm[b] = np.clip(out_f32[b] * F32(32768.0), F32(-32768.0), F32(32767.0)).astype(np.int16)   # trunc toward zero (:243-244)
        return out_pcm, out_f32

    def process_resampled(self, pcm: np.ndarray, out_len: int):
        """Resampling edges (:186-193, :233-240): int16 (n,) at another input rate -> F.interpolate(size = in_len) -> model ->
        F.interpolate(size = out_len) -> * 32768, clamp, truncate."""
        x = interpolate_linear(np.asarray(pcm, np.int16).astype(F32), self.in_len)
        y = interpolate_linear(self._one(x, False), out_len)
        return np.clip(y * F32(32768.0), F32(-32768.0), F32(32767.0)).astype(np.int16)

    def _one(self, pcm: np.ndarray, keep: bool) -> np.ndarray:
        w, T = self.w, self.frames
        audio = pcm.astype(F32) * F32(1.0 / 32768.0)                                            # :186-190 (pcm: int16, or floats in int16 units)
        frames = np.stack([audio[t * HOP:t * HOP + FRAME] for t in range(T)], axis=1)           # (1920, T): conv1d stride 960, no pad
        fb = self.kfb @ frames                                                                  # :205-209 (fused analysis conv)
        spec = self.kst @ frames                                                                # (2*961, T)
        nb = KALDI_NFFT // 2 + 1
        power = (fb[:nb] * fb[:nb] + fb[nb:] * fb[nb:]) * F32(32768.0 * 32768.0)                # :216
        x = np.log(np.maximum(w["mel_banks"] @ power, np.finfo(F32).eps)).astype(F32)           # :217
        if keep:
            self.taps["logmel"] = x.copy()
        x = np.maximum(w["lin1_w"] @ x + w["lin1_b"][:, None], 0.0).astype(F32)                 # :224
        for i in range(self.depth):                                                             # :225-229
            f1 = np.maximum(w[f"uf_lin_w_{i}"] @ x + w[f"uf_lin_b_{i}"][:, None], 0.0).astype(F32)
            p1 = (w[f"uf_proj_w_{i}"] @ f1).astype(F32)
            cw = w[f"uf_conv_w_{i}"]                                                            # (256, lorder), inner residual folded in
            lo = cw.shape[1]
            pad = np.concatenate((np.zeros((p1.shape[0], lo - 1), F32), p1), axis=1)            # causal left pad (fsmn_pad)
            mem = np.zeros_like(p1)
            for k in range(lo):                                                                 # cross-correlation, like F.conv1d
                mem += cw[:, k:k + 1] * pad[:, k:k + T]
            x = (x + mem).astype(F32)
        mask = (1.0 / (1.0 + np.exp(-(w["lin2_w"] @ x + w["lin2_b"][:, None])))).astype(F32)    # :230
        if keep:
            self.taps["mask"] = mask.copy()
        masked = spec * np.concatenate((mask, mask), axis=0)                                    # :236-237
        raw = np.zeros(self.out_len, F32)                                                       # conv_transpose1d stride 960
        fr = (masked.T @ self.kinv).astype(F32)                                                 # (T, 1920)
        for t in range(T):
            raw[t * HOP:t * HOP + NFFT_STFT] += fr[t]
        return (raw / self.win_sum).astype(F32)
