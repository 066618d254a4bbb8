"""melband_oracle.py — CPU ORACLE for the Mel-Band-Roformer hot path.  TEST INFRASTRUCTURE ONLY.

A numpy fp32 restatement of ``MelBandRoformer.forward`` (Mel_Band_Roformer/Stereo/Export_MelBandRoformer.py:626-680) and
``_core`` (:588-624) over the FUSED buffers the reference's constructor registers (:330-538) -- the same tensors the
exported graph carries -- each step citing the lines it follows.  Pinned (tests/test_melband.py) against a fixture made by
running the reference's own forward in the build container (tools/make_golden_melband.py: fused buffers filled by
audio_denoiser_onnx_amd/weightgen.py because no checkpoint is available; the checkpoint -> buffer fusion algebra
(:459-538) is therefore NOT pinned, everything the forward computes is).
Only tests/ may import this module; the product (libade / audio_denoiser_onnx_amd) never does.
"""
from __future__ import annotations

import numpy as np

F32 = np.float32
NFFT, HOP, SR = 2048, 441, 44100         # Export_MelBandRoformer.py:42-45 ('hann', win_length = n_fft)
FBINS = NFFT // 2 + 1
EPS = F32(1e-12)                          # norm_eps (:455)


def hann_periodic(n: int) -> np.ndarray:
    """torch.hann_window(n, periodic=True) (Stereo/STFT_Process.py:93): arange * (2 pi / n) -> cos -> * -0.5 + 0.5 in fp32."""
    k = np.arange(n, dtype=F32)
    return (np.cos(k * F32(2.0 * np.pi / n)) * F32(-0.5) + F32(0.5)).astype(F32)


def stft_kernels(exact: bool = False):
    """Forward (2*1025, 2048) and inverse (2*1025, 2048) windowed DFT matrices (Stereo/STFT_Process.py:205-243; fp32 angles
    omega = fp32(2 pi / N) * f * t).  exact=True swaps in exactly reduced angles (test knob)."""
    n, fb = NFFT, FBINS
    w = hann_periodic(n)
    if exact:
        k = (np.arange(fb, dtype=np.int64)[:, None] * np.arange(n, dtype=np.int64)[None, :]) % n
        ang = 2.0 * np.pi * k.astype(np.float64) / n
        c, s = np.cos(ang).astype(F32), np.sin(ang).astype(F32)
    else:
        omega = (F32(2.0 * np.pi / n) * np.arange(fb, dtype=F32)[:, None]) * np.arange(n, dtype=F32)[None, :]
        c, s = np.cos(omega).astype(F32), np.sin(omega).astype(F32)
    fwd = np.concatenate((c * w[None, :], -s * w[None, :]), axis=0).astype(F32)
    scale = np.full((fb, 1), 2.0, F32)
    scale[0] = 1.0
    scale[fb - 1] = 1.0
    inv_n = F32(1.0 / n)
    inv = np.concatenate((((scale * c) * inv_n) * w[None, :], ((scale * -s) * inv_n) * w[None, :]), axis=0).astype(F32)
    return fwd, inv, w


def rotary_tables(n_pos: int, dim_head: int, half_round: bool):
    """cos / sign-folded sin tables (n_pos, dim_head) (:395-401, :438-449).  The TIME tables pass through fp16
    (m.cos_rotary_pos_emb.half() :413-414) before the static slice converts them back; the FREQ tables stay fp32."""
    pos = np.arange(n_pos, dtype=F32)[:, None]
    inv_freq = (F32(10000.0) ** -(np.arange(0, dim_head, 2, dtype=F32) / F32(dim_head))).astype(F32)
    rot = np.repeat(pos * inv_freq[None, :], 2, axis=1).astype(F32)
    c, s = np.cos(rot).astype(F32), np.sin(rot).astype(F32)
    if half_round:
        c, s = c.astype(np.float16).astype(F32), s.astype(np.float16).astype(F32)
    sign = np.ones(dim_head, F32)
    sign[0::2] = -1.0
    return c, (s * sign[None, :]).astype(F32)


def _normalize(x: np.ndarray) -> np.ndarray:
    """x / max(||x||_2, 1e-12) over the last axis (:533-538)."""
    n = np.sqrt((x * x).sum(axis=-1, keepdims=True, dtype=F32)).astype(F32)
    return (x / np.maximum(n, EPS)).astype(F32)


def _gelu(x: np.ndarray) -> np.ndarray:
    """F.gelu default (erf form) (:564)."""
    from scipy.special import erf
    return (F32(0.5) * x * (F32(1.0) + erf(x.astype(np.float64) * 0.7071067811865476))).astype(F32)


class MelBandOracle:
    """tensors: the fused buffers by their registered names (bs_w_i, time0_in_w, ..., me_w3_i); freq_indices / dim_inputs:
    the reference's band tables; frames: static frame count T; in_len = (T - 1) * 441.
    dynamic=True restates a DYNAMIC_AXES export (:33, :50, :696): `length` model-rate samples per channel (any length >= 2048; T = length // 441 + 1
    frames) and the ISTFT keeps everything after the first half window, divided by the window sum of those T frames (Stereo/STFT_Process.py:296-306)."""

    def __init__(self, tensors: dict, freq_indices: np.ndarray, dim_inputs: np.ndarray, frames: int, depth: int,
                 heads: int = 8, dim_head: int = 64, exact_dft: bool = False, dynamic: bool = False, length: int | None = None):
        self.w = {k: np.asarray(v, F32) for k, v in tensors.items()}
        self.fi = np.asarray(freq_indices, np.int64)
        self.dims = [int(d) for d in dim_inputs]
        self.T, self.depth, self.heads, self.dh = int(frames), int(depth), heads, dim_head
        self.di = heads * dim_head
        self.nb = len(self.dims)
        self.L = (self.T - 1) * HOP if length is None else int(length)
        assert self.L // HOP + 1 == self.T and (dynamic or self.L % HOP == 0)
        self.Lo = (self.T - 1) * HOP + NFFT // 2 if dynamic else self.L
        self.fwd, self.inv, win = stft_kernels(exact_dft)
        self.tcos, self.tsin = rotary_tables(self.T, dim_head, True)
        self.fcos, self.fsin = rotary_tables(self.nb, dim_head, False)
        # static COLA denominator (Stereo/STFT_Process.py:245-256)
        raw = np.zeros(NFFT + HOP * (self.T - 1), F32)
        w2 = (win * win).astype(F32)
        for t in range(self.T):
            raw[t * HOP:t * HOP + NFFT] += w2
        self.win_sum = raw[NFFT // 2:NFFT // 2 + self.Lo].copy()
        self.taps = {}

    # ---- transformer pieces (:540-572) ----
    def _attention(self, x, p, rcos, rsin):
        b, n, _ = x.shape
        qkvg = (_normalize(x) @ self.w[p + "_in_w"].T + self.w[p + "_in_b"]).astype(F32)
        qkv, gates = qkvg[..., :3 * self.di], qkvg[..., 3 * self.di:]
        qkv = qkv.reshape(b, n, 3, self.heads, self.dh).transpose(2, 0, 3, 1, 4)      # (3, b, heads, n, dh)
        perm = np.arange(self.dh).reshape(-1, 2)[:, ::-1].reshape(-1)                # rotate_indices (:450-453)
        qk = (qkv[:2] * rcos + qkv[:2][..., perm] * rsin).astype(F32)
        q, k, v = qk[0], qk[1], qkv[2]
        s = (q @ k.transpose(0, 1, 3, 2)).astype(F32)
        s = s - s.max(axis=-1, keepdims=True)
        e = np.exp(s).astype(F32)
        a = (e / e.sum(axis=-1, keepdims=True, dtype=F32)).astype(F32)
        out = (a @ v).astype(F32).transpose(0, 2, 1, 3)                               # (b, n, heads, dh)
        g = (F32(1.0) / (F32(1.0) + np.exp(-gates))).astype(F32)[..., None]
        out = (out * g).reshape(b, n, self.di)
        return (out @ self.w[p + "_out_w"].T).astype(F32)

    def _transformer(self, x, p, rcos, rsin):
        x = (x + self._attention(x, p, rcos, rsin)).astype(F32)
        h = _gelu((_normalize(x) @ self.w[p + "_ff1_w"].T + self.w[p + "_ff1_b"]).astype(F32))
        x = (x + (h @ self.w[p + "_ff2_w"].T + self.w[p + "_ff2_b"])).astype(F32)
        return (_normalize(x) * self.w[p + "_out_g"]).astype(F32)

    def process(self, pcm: np.ndarray) -> np.ndarray:
        """pcm int16 (2, L) -> int16 (2, Lo)   (forward :626-680, no fold, all rates 44.1 kHz)."""
        assert pcm.shape == (2, self.L) and pcm.dtype == np.int16
        wav = self.process_wave(pcm.astype(F32))
        return np.clip(wav * F32(32767.0), -32768.0, 32767.0).astype(np.int16)             # (:667, :676) trunc toward zero

    def process_rates(self, pcm: np.ndarray, in_rate: int, out_rate: int) -> np.ndarray:
        """The resampling sandwich of forward (:630-644, :660-680) around the network: F.interpolate(scale_factor = float(MODEL / IN)) on audio.float(),
        down-sampling before the * 32767 of the int16 output and up-sampling after it.  pcm int16 (2, n) with floor(n * MODEL / IN) == self.L."""
        from gtcrn_sandwich import interpolate_scale
        x = pcm.astype(F32)                                                                # :630
        if in_rate != SR:
            x = interpolate_scale(x, float(SR / in_rate))                                  # :52, :631-644
        assert x.shape == (2, self.L), (x.shape, self.L)
        wav = self.process_wave(x)
        f_out = float(out_rate / SR)                                                       # :53
        if out_rate < SR:
            wav = interpolate_scale(wav, f_out)                                            # :660-666
        wav = (wav * F32(32767.0)).astype(F32)                                             # :667-668
        if out_rate > SR:
            wav = interpolate_scale(wav, f_out)                                            # :669-675
        return np.clip(wav, -32768.0, 32767.0).astype(np.int16)                            # :676-677

    def process_wave(self, samples: np.ndarray) -> np.ndarray:
        """fp32 samples in PCM units (2, L) -> the normalised fp32 waveform (2, Lo) the ISTFT returns."""
        assert samples.shape == (2, self.L) and samples.dtype == F32
        T, half = self.T, NFFT // 2
        x = samples * F32(1.0 / 32768.0)                         # INV_INT16 folded into the STFT kernel (:326-327)
        xp = np.concatenate((x[:, 1:half + 1][:, ::-1], x, x[:, -(half + 1):-1][:, ::-1]), axis=1)
        frames = np.stack([xp[:, t * HOP:t * HOP + NFFT] for t in range(T)], axis=1)      # (2, T, 2048)
        spec = (frames @ self.fwd.T).astype(F32)                                           # (2, T, 2*1025)
        re, im = spec[..., :FBINS], spec[..., FBINS:]
        # (chan, F, T, 2) -> (F*chan, T, 2), channel-minor (:596)
        rep = np.stack((re, im), axis=-1).transpose(2, 0, 1, 3).reshape(FBINS * 2, T, 2)
        self.taps["spec"] = rep.copy()
        sel = rep[self.fi]                                                                 # (S, T, 2) (:597)
        xb = sel.transpose(1, 0, 2).reshape(T, -1)                                         # (T, 2S) (:598)
        outs, off = [], 0
        for i, d in enumerate(self.dims):                                                  # _band_split (:574-577)
            outs.append((_normalize(xb[:, off:off + d]) @ self.w[f"bs_w_{i}"].T + self.w[f"bs_b_{i}"]).astype(F32))
            off += d
        h = np.stack(outs, axis=0)                                                         # (nb, T, dim)
        self.taps["band_split"] = h.copy()
        self.taps["layers"] = []                                                           # the token tensor after each (time, freq) pair: tests print the error growth with depth
        for i in range(self.depth):                                                        # axial transformers (:609-614)
            h = self._transformer(h, f"time{i}", self.tcos, self.tsin)
            h = h.transpose(1, 0, 2)                                                       # (T, nb, dim)
            h = self._transformer(h, f"freq{i}", self.fcos, self.fsin)
            h = h.transpose(1, 0, 2)
            self.taps["layers"].append(h.copy())
        self.taps["tf_out"] = h.copy()
        m = np.tanh(h @ self.w["me_w1t"] + self.w["me_b1"]).astype(F32)                    # _mask_estimator (:579-585)
        m = np.tanh(m @ self.w["me_w2t"] + self.w["me_b2"]).astype(F32)
        parts = []
        for i, d in enumerate(self.dims):
            y = (m[i] @ self.w[f"me_w3_{i}"].T + self.w[f"me_b3_{i}"]).astype(F32)
            parts.append((y[:, :d] * (F32(1.0) / (F32(1.0) + np.exp(-y[:, d:])))).astype(F32))   # GLU
        masks = np.concatenate(parts, axis=-1)                                             # (T, 2S)
        self.taps["masks"] = masks.copy()
        masks = masks.reshape(T, -1, 2).transpose(1, 0, 2)                                 # (S, T, 2) (:616)
        avg = np.zeros_like(rep)
        np.add.at(avg, self.fi, masks)                                                     # scatter_add (:617-619)
        self.taps["mask_avg"] = avg.copy()
        mr, mi = avg[..., 0], avg[..., 1]
        out_re = (rep[..., 0] * mr - rep[..., 1] * mi).astype(F32)                         # (:621-624)
        out_im = (rep[..., 0] * mi + rep[..., 1] * mr).astype(F32)
        out_re = out_re.reshape(FBINS, 2, T).transpose(1, 2, 0)                            # (chan, T, F)
        out_im = out_im.reshape(FBINS, 2, T).transpose(1, 2, 0)
        fr = (np.concatenate((out_re, out_im), axis=-1) @ self.inv).astype(F32)            # (chan, T, 2048): irDFT + window
        raw = np.zeros((2, NFFT + HOP * (T - 1)), F32)
        for t in range(T):
            raw[:, t * HOP:t * HOP + NFFT] += fr[:, t]
        return (raw[:, half:half + self.Lo] / self.win_sum).astype(F32)

    def process_fold(self, pcm: np.ndarray, n_win: int) -> np.ndarray:
        """USE_BATCH_FOLD (:644-647, :663-664): (2, n_win * W) -> n_win independent stereo clips of W -> stitched back per channel."""
        assert pcm.shape == (2, n_win * self.L)
        out = np.empty_like(pcm)
        for w in range(n_win):
            out[:, w * self.L:(w + 1) * self.L] = self.process(np.ascontiguousarray(pcm[:, w * self.L:(w + 1) * self.L]))
        return out
