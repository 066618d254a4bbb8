"""hgtcrn_oracle.py — CPU ORACLE for the H-GTCRN hot path.  TEST INFRASTRUCTURE ONLY.

A numpy fp32 restatement of ``H_GTCRN_CUSTOM.forward`` (H-GTCRN/Export_H_GTCRN.py:941-1063): int16 stereo -> /32768, minus the mean of the
whole call -> [fold] -> STFT (512 / 256, periodic hann, reflect) -> WPE dereverberation with a 6-step conjugate-gradient solve
(``OnnxFriendlyWPE`` :581-757, ``batched_complex_solve_cg`` :499-555) -> AuxIVA, 10 iterations, 2x2 Cramer solves (``OnnxFriendlyAuxIVA``
:760-900, ``solve_2x2_complex`` :557-598) -> six features -> the GTCRN_IVA network (:428-494, blocks :75-425) -> complex ratio mask on
microphone 0 -> ISTFT -> x32767, NaN -> 0, clamp, int16.  Works on the FOLDED tensors of ``audio_denoiser_onnx_amd.hgtcrn.fold_state_dict``
(BatchNorm folded into the convolutions; GTCRN's folded names).  Pinned (tests/test_hgtcrn.py) against fixtures made by running the
reference in the build container (tools/make_golden_hgtcrn.py), which pins the fold as well.  Only tests/ may import this module; the
product never does.
"""
from __future__ import annotations

import numpy as np

from ulunas_oracle import UlunasOracle, _gru, _sig, stft_tables

F32 = np.float32
NFFT, HOP, FB = 512, 256, 257
LG, DELAY, CG_ITER, IVA_ITER = 18, 2, 6, 10          # int(0.3 * 16000 / 256), WPE_DELAY, CG_SOLVE_ITER, IVA_ITER (:50-54, :610)


def _mm(a, b):
    return np.matmul(a, b).astype(F32)


def wpe(Xr, Xi, dtype=F32):
    """(B, F, 2, T) -> (B, F, 2, T), one iteration (:686-757).  dtype=np.float64 is a TEST KNOB: the conjugate-gradient solve is ill-conditioned in
    fp32 for some bins (the reference's own fp32 and fp64 runs differ there by O(1)); comparing the two marks the bins where parity is defined."""
    F32 = dtype                                                        # noqa: N806  (shadows the module constant inside this function only)
    Xr, Xi = Xr.astype(F32), Xi.astype(F32)
    _mm = lambda a, b: np.matmul(a, b).astype(F32)                     # noqa: E731
    B, Fq, M, T = Xr.shape
    Dr = np.zeros((B, Fq, LG, M, T), F32)
    Di = np.zeros_like(Dr)
    for l in range(LG):                                                # delay bank (:627-684): row l * M + m = channel m delayed DELAY + l frames
        sh = DELAY + l
        if sh < T:
            Dr[:, :, l, :, sh:] = Xr[..., :T - sh]
            Di[:, :, l, :, sh:] = Xi[..., :T - sh]
    Dr, Di = Dr.reshape(B, Fq, LG * M, T), Di.reshape(B, Fq, LG * M, T)
    mag = (Xr * Xr + Xi * Xi).astype(F32)
    eps = (F32(1e-3) * mag.max(axis=(-2, -1)).mean(axis=-1, dtype=F32)).astype(F32).reshape(B, 1, 1, 1)     # (:690-691)
    ypow = np.maximum(mag.mean(axis=2, keepdims=True, dtype=F32), eps).astype(F32)
    inv = (F32(1.0) / ypow).astype(F32)
    tr, ti = (Dr * inv).astype(F32), (Di * inv).astype(F32)
    DrT, DiT = Dr.transpose(0, 1, 3, 2), Di.transpose(0, 1, 3, 2)
    XrT, XiT = Xr.transpose(0, 1, 3, 2), Xi.transpose(0, 1, 3, 2)
    Rr = (_mm(tr, DrT) + _mm(ti, DiT)).astype(F32)
    Ri = (_mm(ti, DrT) - _mm(tr, DiT)).astype(F32)
    Pr = (_mm(tr, XrT) + _mm(ti, XiT)).astype(F32)
    Pi = (_mm(ti, XrT) - _mm(tr, XiT)).astype(F32)
    Rr = (Rr + eps * np.eye(LG * M, dtype=F32)).astype(F32)
    # conjugate gradient on R G = P, column by column (:499-555)
    xr = np.zeros_like(Pr); xi = np.zeros_like(Pi)
    rr_, ri_, pr, pi = Pr, Pi, Pr, Pi
    rr = ((rr_ * rr_ + ri_ * ri_).sum(axis=-2, dtype=F32) + F32(1e-12)).astype(F32)
    for _ in range(CG_ITER):
        Apr = (_mm(Rr, pr) - _mm(Ri, pi)).astype(F32)
        Api = (_mm(Rr, pi) + _mm(Ri, pr)).astype(F32)
        pAp = ((pr * Apr + pi * Api).sum(axis=-2, dtype=F32) + F32(1e-12)).astype(F32)
        alpha = (rr / pAp).astype(F32)[..., None, :]
        xr, xi = (xr + alpha * pr).astype(F32), (xi + alpha * pi).astype(F32)
        rr_, ri_ = (rr_ - alpha * Apr).astype(F32), (ri_ - alpha * Api).astype(F32)
        rr_new = ((rr_ * rr_ + ri_ * ri_).sum(axis=-2, dtype=F32) + F32(1e-12)).astype(F32)
        beta = (rr_new / rr).astype(F32)[..., None, :]
        pr, pi = (rr_ + beta * pr).astype(F32), (ri_ + beta * pi).astype(F32)
        rr = rr_new
    Gr, Gi = xr.transpose(0, 1, 3, 2), (-xi).transpose(0, 1, 3, 2)     # conj(G)^T
    pred_r = (_mm(Gr, Dr) - _mm(Gi, Di)).astype(F32)
    pred_i = (_mm(Gi, Dr) + _mm(Gr, Di)).astype(F32)
    return (Xr - pred_r).astype(F32), (Xi - pred_i).astype(F32)


def _solve2(Ar, Ai, s):
    """Cramer's rule for A x = e_s, A (..., 2, 2) complex (:557-598); returns x (..., 2) real, imaginary."""
    a_r, b_r, c_r, d_r = Ar[..., 0, 0], Ar[..., 0, 1], Ar[..., 1, 0], Ar[..., 1, 1]
    a_i, b_i, c_i, d_i = Ai[..., 0, 0], Ai[..., 0, 1], Ai[..., 1, 0], Ai[..., 1, 1]
    det_r = ((a_r * d_r - a_i * d_i) - (b_r * c_r - b_i * c_i)).astype(F32)
    det_i = ((a_r * d_i + a_i * d_r) - (b_r * c_i + b_i * c_r)).astype(F32)
    q = (F32(1.0) / ((det_r * det_r + det_i * det_i) + F32(1e-12))).astype(F32)
    ir, ii = (det_r * q).astype(F32), (-det_i * q).astype(F32)
    one, zero = np.ones_like(a_r), np.zeros_like(a_r)
    b0r, b1r = (one, zero) if s == 0 else (zero, one)
    b0i = b1i = zero
    n0r = ((d_r * b0r - d_i * b0i) - (b_r * b1r - b_i * b1i)).astype(F32)
    n0i = ((d_r * b0i + d_i * b0r) - (b_r * b1i + b_i * b1r)).astype(F32)
    n1r = ((a_r * b1r - a_i * b1i) - (c_r * b0r - c_i * b0i)).astype(F32)
    n1i = ((a_r * b1i + a_i * b1r) - (c_r * b0i + c_i * b0r)).astype(F32)
    xr = np.stack(((n0r * ir - n0i * ii), (n1r * ir - n1i * ii)), axis=-1).astype(F32)
    xi = np.stack(((n0r * ii + n0i * ir), (n1r * ii + n1i * ir)), axis=-1).astype(F32)
    return xr, xi


def auxiva(Xr, Xi):
    """(B, F, 2, T) -> (B, F, 2, T): AuxIVA with the projection back onto microphone 0 (:795-900)."""
    B, Fq, M, T = Xr.shape
    inv_t, eps = F32(1.0 / T), F32(1e-10)
    XrT, XiT = Xr.transpose(0, 1, 3, 2), Xi.transpose(0, 1, 3, 2)
    Wr = np.broadcast_to(np.eye(2, dtype=F32), (B, Fq, 2, 2)).copy()
    Wi = np.zeros((B, Fq, 2, 2), F32)
    Yr, Yi = Xr, Xi
    for it in range(IVA_ITER):
        r = (F32(2.0) * np.sqrt((Yr * Yr + Yi * Yi).astype(F32).sum(axis=1, dtype=F32) + eps)).astype(F32)        # (B, 2, T)
        r_inv = (F32(1.0) / r).astype(F32)
        for s in range(2):
            w = r_inv[:, s][:, None, None, :]
            wr, wi = (Xr * w).astype(F32), (Xi * w).astype(F32)
            Vr = ((_mm(wr, XrT) + _mm(wi, XiT)) * inv_t).astype(F32)
            Vi = ((_mm(wi, XrT) - _mm(wr, XiT)) * inv_t).astype(F32)
            if it == 0 and s == 0:
                WVr, WVi = Vr, Vi
            else:
                WVr = (_mm(Wr, Vr) - _mm(Wi, Vi)).astype(F32)
                WVi = (_mm(Wr, Vi) + _mm(Wi, Vr)).astype(F32)
            WVr = (WVr + eps * np.eye(2, dtype=F32)).astype(F32)
            nr, ni = _solve2(WVr, WVi, s)                                                                         # (B, F, 2)
            Vwr = (_mm(Vr, nr[..., None]) - _mm(Vi, ni[..., None])).astype(F32)[..., 0]
            Vwi = (_mm(Vr, ni[..., None]) + _mm(Vi, nr[..., None])).astype(F32)[..., 0]
            cr, ci = nr, (-ni).astype(F32)
            den = (cr * Vwr - ci * Vwi).astype(F32).sum(axis=-1, keepdims=True, dtype=F32)
            sc = (F32(1.0) / np.sqrt(np.maximum(den, F32(0.0)) + eps)).astype(F32)
            Wr[:, :, s, :] = (cr * sc).astype(F32)
            Wi[:, :, s, :] = (ci * sc).astype(F32)
        Yr = (_mm(Wr, Xr) - _mm(Wi, Xi)).astype(F32)
        Yi = (_mm(Wr, Xi) + _mm(Wi, Xr)).astype(F32)
    ref_r, ref_i = Xr[:, :, :1], Xi[:, :, :1]
    num_r = (ref_r * Yr + ref_i * Yi).astype(F32).sum(axis=-1, dtype=F32)
    num_i = (ref_r * Yi - ref_i * Yr).astype(F32).sum(axis=-1, dtype=F32)
    den = (Yr * Yr + Yi * Yi).astype(F32).sum(axis=-1, dtype=F32)
    valid = den > 0
    safe = (F32(1.0) / np.where(valid, den, F32(1.0))).astype(F32)
    c_r = np.where(valid, num_r * safe, F32(1.0)).astype(F32)[..., None]
    c_i = np.where(valid, num_i * safe, F32(0.0)).astype(F32)[..., None]
    return (c_r * Yr + c_i * Yi).astype(F32), (c_r * Yi - c_i * Yr).astype(F32)


class HgtcrnOracle:
    def __init__(self, tensors: dict, window_len: int, n_win: int = 1, exact_dft: bool = False, dynamic: bool = False):
        self.w = {k: np.asarray(v, F32) for k, v in tensors.items()}
        self.W, self.n_win = int(window_len), int(n_win)
        self.T = self.W // HOP + 1
        self.fwd, self.inv, win = stft_tables(exact_dft)
        raw = np.zeros(NFFT + HOP * (self.T - 1), F32)
        for t in range(self.T):
            raw[t * HOP:t * HOP + NFFT] += (win * win).astype(F32)
        # dynamic: the DYNAMIC_AXES export (Export_H_GTCRN.py:27, :1097): the ISTFT's slice end is sized for 4096 frames, so everything after the leading half window
        # stays -- half a window more than the static trim -- and the window-square sum is that of the actual frames (STFT_Process.py:318-327): in the kept tail only the last
        # frame contributes, so the samples are x / w with w running down to hann(511) ~ 4e-5 (the periodic window's zero is at sample 0 of a frame): they saturate the int16 clamp.
        self.dynamic = bool(dynamic)
        assert not (self.dynamic and self.n_win != 1), "Batch folding requires a static shape"
        self.out_len = HOP * (self.T - 1) + (NFFT // 2 if self.dynamic else 0)
        self.win_sum = raw[NFFT // 2:NFFT // 2 + self.out_len].copy()                     # COLA table (STFT_Process.py:262-274; the same values when computed per call)
        self.taps = {}

    # ---- network blocks (B, C, T, F) -----------------------------------------------------------------------------------------
    @staticmethod
    def _prelu(x, a):
        return np.where(x >= 0, x, x * F32(a)).astype(F32)

    @staticmethod
    def _sfe(x):
        """3-tap neighbourhood on F, channel c * 3 + o (:136-162)."""
        B, C, T, Fq = x.shape
        p = np.zeros((B, C, T, Fq + 2), F32)
        p[..., 1:-1] = x
        return np.stack((p[..., :Fq], p[..., 1:Fq + 1], p[..., 2:Fq + 2]), axis=2).reshape(B, C * 3, T, Fq)

    def _convblock(self, x, p, stride, groups, deconv, last=False):
        y = UlunasOracle._conv(x, self.w[p + "conv.weight"], self.w[p + "conv.bias"], (1, 5), stride, groups, deconv)
        return np.tanh(y).astype(F32) if last else self._prelu(y, self.w[p + "act.weight"][0])

    def _gt(self, x, p, dil):
        """GTConvBlock (:262-290): SFE + 1x1 + causal dilated depthwise 3x3 + 1x1 on the first half, TRA gate, interleave with the second half."""
        w = self.w
        B, C, T, Fq = x.shape
        x1, x2 = x[:, :8], x[:, 8:]
        h = np.einsum("bitf,oi->botf", self._sfe(x1), w[p + "point_conv1.weight"][:, :, 0, 0]).astype(F32) + w[p + "point_conv1.bias"][None, :, None, None]
        h = self._prelu(h.astype(F32), w[p + "point_act.weight"][0])
        hp = np.zeros((B, 16, T + 2 * dil, Fq + 2), F32)
        hp[:, :, 2 * dil:, 1:-1] = h
        dw = w[p + "depth_conv.weight"][:, 0]                                             # (16, 3, 3)
        y = np.zeros((B, 16, T, Fq), F32)
        for a in range(3):
            for b in range(3):
                y += hp[:, :, a * dil:a * dil + T, b:b + Fq] * dw[None, :, a, b, None, None]
        y = self._prelu((y + w[p + "depth_conv.bias"][None, :, None, None]).astype(F32), w[p + "depth_act.weight"][0])
        h1 = (np.einsum("bitf,oi->botf", y, w[p + "point_conv2.weight"][:, :, 0, 0]).astype(F32) + w[p + "point_conv2.bias"][None, :, None, None]).astype(F32)
        zt = (h1 * h1).astype(F32).mean(axis=-1, dtype=F32).transpose(2, 0, 1)                # (T, B, 8)   TRA (:165-178)
        g = _gru(zt, w[p + "tra.att_gru.weight_ih_l0"], w[p + "tra.att_gru.weight_hh_l0"], w[p + "tra.att_gru.bias_ih_l0"], w[p + "tra.att_gru.bias_hh_l0"])
        at = _sig((g @ w[p + "tra.att_fc.weight"].T + w[p + "tra.att_fc.bias"]).astype(F32)).transpose(1, 2, 0)[..., None]
        h1 = (h1 * at).astype(F32)
        return np.stack((h1, x2), axis=2).reshape(B, 16, T, Fq)

    def _grnn(self, x, p, bidirectional):
        half = x.shape[-1] // 2
        outs = []
        for name, xs in (("rnn1", x[..., :half]), ("rnn2", x[..., half:])):
            q = f"{p}{name}."
            y = _gru(xs, self.w[q + "weight_ih_l0"], self.w[q + "weight_hh_l0"], self.w[q + "bias_ih_l0"], self.w[q + "bias_hh_l0"])
            if bidirectional:
                yb = _gru(xs, self.w[q + "weight_ih_l0_reverse"], self.w[q + "weight_hh_l0_reverse"], self.w[q + "bias_ih_l0_reverse"],
                          self.w[q + "bias_hh_l0_reverse"], reverse=True)
                y = np.concatenate((y, yb), axis=-1)
            outs.append(y)
        return np.concatenate(outs, axis=-1)

    def _dpgrnn(self, x, p):
        """DPGRNN (:339-384), x (B, T, F, C)."""
        B, T, Fq, C = x.shape
        w = self.w
        intra_in = x.transpose(2, 0, 1, 3).reshape(Fq, B * T, C)
        y = (self._grnn(intra_in, p + "intra_rnn.", True) @ w[p + "intra_fc.weight"].T + w[p + "intra_fc.bias"]).astype(F32)
        y = y.reshape(Fq, B, T, C).transpose(1, 2, 0, 3)
        intra_out = (x + UlunasOracle._ln(y, w[p + "intra_ln.weight"], w[p + "intra_ln.bias"])).astype(F32)
        inter_in = intra_out.transpose(1, 0, 2, 3).reshape(T, B * Fq, C)
        y = (self._grnn(inter_in, p + "inter_rnn.", False) @ w[p + "inter_fc.weight"].T + w[p + "inter_fc.bias"]).astype(F32)
        y = y.reshape(T, B, Fq, C).transpose(1, 0, 2, 3)
        return (intra_out + UlunasOracle._ln(y, w[p + "inter_ln.weight"], w[p + "inter_ln.bias"])).astype(F32)

    def network(self, feat):
        """GTCRN_IVA.forward (:462-494) up to the mask: (B, 6, T, 257) -> (B, 2, T, 257)."""
        w = self.w
        x = np.concatenate((feat[..., :65], _mm(feat[..., 65:], w["erb.erb_weight_t"])), axis=-1).astype(F32)
        x = self._sfe(x)                                                                   # (B, 18, T, 129)
        e0 = self._convblock(x, "encoder.en_convs.0.", 2, 1, False)
        e1 = self._convblock(e0, "encoder.en_convs.1.", 2, 2, False)
        e2 = self._gt(e1, "encoder.en_convs.2.", 1)
        e3 = self._gt(e2, "encoder.en_convs.3.", 2)
        e4 = self._gt(e3, "encoder.en_convs.4.", 5)
        h = e4.transpose(0, 2, 3, 1)
        h = self._dpgrnn(h, "dpgrnn1.")
        h = self._dpgrnn(h, "dpgrnn2.").transpose(0, 3, 1, 2)
        h = self._gt((h + e4).astype(F32), "decoder.de_convs.0.", 5)
        h = self._gt((h + e3).astype(F32), "decoder.de_convs.1.", 2)
        h = self._gt((h + e2).astype(F32), "decoder.de_convs.2.", 1)
        h = self._convblock((h + e1).astype(F32), "decoder.de_convs.3.", 2, 2, True)
        m = self._convblock((h + e0).astype(F32), "decoder.de_convs.4.", 2, 1, True, last=True)       # (B, 2, T, 129)
        return np.concatenate((m[..., :65], _mm(m[..., 65:], w["erb.ierb_weight_t"])), axis=-1).astype(F32)

    # ---- the call ------------------------------------------------------------------------------------------------------------
    def process(self, pcm: np.ndarray, inject_wpe=None, float_out: bool = False) -> np.ndarray:
        """pcm int16 (calls, 2, n_win * W) -> int16 (calls, n_win * out_len).  inject_wpe = (re, im), each (B, 2, F, T): continue from a given WPE
        output instead of this module's own (how the tests pin everything downstream of the ill-conditioned solve on identical inputs).
        A float32 `pcm` is an IN_AUDIO_DTYPE F32 / F16 tensor: normalised samples, the * inv_int16 left out (:965-966); float_out: OUT_AUDIO_DTYPE F32 / F16, the waveform
        without the * 32767 and the clamp (:1042-1043, :1057-1063)."""
        assert pcm.ndim == 3 and pcm.shape[1] == 2 and pcm.shape[2] == self.W * self.n_win and pcm.dtype in (np.int16, np.float32)
        calls = pcm.shape[0]
        x = (pcm.astype(F32) * F32(1.0 / 32768.0)).astype(F32) if pcm.dtype == np.int16 else pcm
        x = (x - x.reshape(calls, -1).mean(axis=1, dtype=F32)[:, None, None]).astype(F32)                 # (:963-964) the mean of the whole call
        with np.errstate(all="ignore"):
            y = self._core(x, inject_wpe)
            if not float_out:
                y = (y * F32(32767.0)).astype(F32)
            y = np.where(np.isnan(y), F32(0.0), y)                                                         # (:1054)
        return y.astype(F32) if float_out else np.clip(y, -32768.0, 32767.0).astype(np.int16)

    @staticmethod
    def _interp(x, factor):
        """F.interpolate(x, scale_factor=factor, mode='linear', align_corners=False) over the last axis: floor(n * factor) samples, source
        coordinate (dst + 0.5) / factor - 0.5 clamped at 0 (the given factor is used as is)."""
        n = x.shape[-1]
        m = int(np.floor(n * factor))
        step = F32(1.0 / factor)
        src = np.maximum(step * (np.arange(m, dtype=F32) + F32(0.5)) - F32(0.5), F32(0.0)).astype(F32)
        i0 = np.minimum(src.astype(np.int64), n - 1)
        i1 = np.minimum(i0 + 1, n - 1)
        l1 = (src - i0.astype(F32)).astype(F32)
        return ((F32(1.0) - l1) * x[..., i0] + l1 * x[..., i1]).astype(F32)

    def process_resampled(self, pcm: np.ndarray, in_rate: int, out_rate: int, inject_wpe=None) -> np.ndarray:
        """The resampling edges (:953-970, :1036-1052): interpolate to 16 kHz before the scaling / centring when the input rate is higher,
        after them when it is lower; on the way out interpolate before the PCM scale when the output rate is lower, after it when higher."""
        assert self.n_win == 1 and pcm.ndim == 3 and pcm.shape[1] == 2 and pcm.dtype == np.int16
        calls = pcm.shape[0]
        x = pcm.astype(F32)
        if in_rate > 16000:
            x = self._interp(x, 16000.0 / in_rate)
        x = (x * F32(1.0 / 32768.0)).astype(F32)
        x = (x - x.reshape(calls, -1).mean(axis=1, dtype=F32)[:, None, None]).astype(F32)
        if in_rate < 16000:
            x = self._interp(x, 16000.0 / in_rate)
        assert x.shape[2] == self.W, x.shape
        with np.errstate(all="ignore"):
            y = self._core(x, inject_wpe)
            if out_rate < 16000:
                y = self._interp(y, out_rate / 16000.0)
            y = (y * F32(32767.0)).astype(F32)
            if out_rate > 16000:
                y = self._interp(y, out_rate / 16000.0)
            y = np.where(np.isnan(y), F32(0.0), y)
        return np.clip(y, -32768.0, 32767.0).astype(np.int16)

    def _core(self, x, inject_wpe=None):
        """Centred model-rate waveform (calls, 2, n_win * W) -> enhanced waveform (calls, n_win * out_len), everything between the two PCM edges."""
        calls, T, half = x.shape[0], self.T, NFFT // 2
        x = x.reshape(calls, 2, self.n_win, self.W).transpose(0, 2, 1, 3).reshape(-1, self.W)             # (:972-981) row = (window, channel)
        xp = np.concatenate((x[:, 1:half + 1][:, ::-1], x, x[:, -(half + 1):-1][:, ::-1]), axis=1)
        frames = np.stack([xp[:, t * HOP:t * HOP + NFFT] for t in range(T)], axis=1)
        spec = (frames @ self.fwd.T).astype(F32)                                                           # (2B, T, 514)
        B = calls * self.n_win
        re = spec[..., :FB].reshape(B, 2, T, FB).transpose(0, 3, 1, 2)                                     # (B, F, 2, T)
        im = spec[..., FB:].reshape(B, 2, T, FB).transpose(0, 3, 1, 2)
        if True:
            self.taps.update(stft_r=re.transpose(0, 2, 1, 3), stft_i=im.transpose(0, 2, 1, 3))
            if inject_wpe is None:
                dr, di = wpe(re, im)
            else:
                dr = np.ascontiguousarray(np.asarray(inject_wpe[0], F32).transpose(0, 2, 1, 3))
                di = np.ascontiguousarray(np.asarray(inject_wpe[1], F32).transpose(0, 2, 1, 3))
            yr, yi = auxiva(dr, di)
            self.taps.update(wpe_r=dr.transpose(0, 2, 1, 3), wpe_i=di.transpose(0, 2, 1, 3), iva_r=yr.transpose(0, 2, 1, 3), iva_i=yi.transpose(0, 2, 1, 3))
            power = (yr * yr + yi * yi).astype(F32)                                                        # (B, F, 2, T)
            energy = power.sum(axis=(1, 3), dtype=F32)                                                     # (B, 2)   (:1000-1003)
            pred = (energy[:, 0] < energy[:, 1])[:, None, None]
            logm = (F32(0.5) * np.log10(np.maximum(power, F32(1e-24)))).astype(F32)                        # (:1010)
            sel = np.where(pred, logm[:, :, 0], logm[:, :, 1])
            uns = np.where(pred, logm[:, :, 1], logm[:, :, 0])
            feat = np.stack((re[:, :, 0], im[:, :, 0], re[:, :, 1], im[:, :, 1], sel, uns), axis=1).transpose(0, 1, 3, 2).astype(F32)   # (B, 6, T, F)
            self.taps["features"] = feat
            m = self.network(feat)
            ref_r, ref_i = feat[:, 0], feat[:, 1]                                                          # (B, T, F)
            s_r = (ref_r * m[:, 0] - ref_i * m[:, 1]).astype(F32)
            s_i = (ref_i * m[:, 0] + ref_r * m[:, 1]).astype(F32)
            self.taps.update(s_r=s_r.transpose(0, 2, 1), s_i=s_i.transpose(0, 2, 1))
            fr = (np.concatenate((s_r, s_i), axis=-1) @ self.inv).astype(F32)
            raw = np.zeros((B, NFFT + HOP * (T - 1)), F32)
            for t in range(T):
                raw[:, t * HOP:t * HOP + NFFT] += fr[:, t]
            wav = (raw[:, half:half + self.out_len] / self.win_sum).astype(F32)
            self.taps["wav"] = wav.reshape(calls, -1)
        return wav.reshape(calls, self.n_win * self.out_len)
