"""ulunas_oracle.py — CPU ORACLE for the UL-UNAS hot path.  TEST INFRASTRUCTURE ONLY.

A numpy fp32 restatement of ``ULUNAS_CUSTOM.forward`` (UL-UNAS/Export_UL_UNAS.py:848-913) and ``ULUNAS.forward`` with its blocks
(:111-739) over the FOLDED tensors ``audio_denoiser_onnx_amd.ulunas.fold_state_dict`` produces (BatchNorm folded into the
convolutions, AffinePReLU as positive / negative slope tables, the two half-width GRUs of each grouped GRU), in the reference's
(B, C, T, F) layout, each step citing the lines it follows.  Pinned (tests/test_ulunas.py) against fixtures made by running the
reference's own export path -- ``prepare_for_export_`` + wrapper forward -- in the build container (tools/make_golden_ulunas.py),
which pins the fold as well.  Only tests/ may import this module; the product never does.
"""
from __future__ import annotations

import numpy as np

F32 = np.float32
NFFT, HOP, FB = 512, 256, 257


def _sig(x):
    with np.errstate(over="ignore"):            # exp(-x) -> inf for very negative x: 1 / (1 + inf) = 0, the limit
        return (F32(1.0) / (F32(1.0) + np.exp(-x))).astype(F32)


def _gru(x, wih, whh, bih, bhh, reverse=False):
    """nn.GRU, one layer, zero initial state.  x (S, N, I) -> (S, N, H); gate order r, z, n; n = tanh(W_in x + b_in + r (W_hn h + b_hn))."""
    S, N, _ = x.shape
    H = whh.shape[1]
    gi = (x @ wih.T + bih).astype(F32)
    h = np.zeros((N, H), F32)
    out = np.zeros((S, N, H), F32)
    order = range(S - 1, -1, -1) if reverse else range(S)
    for s in order:
        gh = (h @ whh.T + bhh).astype(F32)
        r = _sig(gi[s][:, :H] + gh[:, :H])
        z = _sig(gi[s][:, H:2 * H] + gh[:, H:2 * H])
        n = np.tanh(gi[s][:, 2 * H:] + r * gh[:, 2 * H:]).astype(F32)
        h = ((F32(1.0) - z) * n + z * h).astype(F32)
        out[s] = h
    return out


def stft_tables(exact: bool = False):
    """512 / 256 periodic-hann forward and inverse matrices with the reference's fp32 angles (UL-UNAS/STFT_Process.py, 'hann' :93);
    exact=True: exactly reduced angles (test knob: the HIP path's tables, isolates kernel error from the reference's table error)."""
    n = NFFT
    w = (np.cos(np.arange(n, dtype=F32) * F32(2.0 * np.pi / n)) * F32(-0.5) + F32(0.5)).astype(F32)
    if exact:
        k = (np.arange(FB, dtype=np.int64)[:, None] * np.arange(n, dtype=np.int64)[None, :]) % n
        ang = 2.0 * np.pi * k.astype(np.float64) / n
        c, s = np.cos(ang).astype(F32), np.sin(ang).astype(F32)
    else:
        omega = (F32(2.0 * np.pi / n) * np.arange(FB, dtype=F32)[:, None]) * np.arange(n, dtype=F32)[None, :]
        c, s = np.cos(omega).astype(F32), np.sin(omega).astype(F32)
    fwd = np.concatenate((c * w, -s * w), axis=0).astype(F32)
    scale = np.full((FB, 1), 2.0, F32)
    scale[0] = 1.0
    scale[FB - 1] = 1.0
    inv = np.concatenate((((scale * c) * F32(1.0 / n)) * w, ((scale * -s) * F32(1.0 / n)) * w), axis=0).astype(F32)
    return fwd, inv, w


class UlunasOracle:
    def __init__(self, tensors: dict, plan: list, in_len: int, exact_dft: bool = False, dynamic_keep: int = 0):
        """dynamic_keep > 0 restates a DYNAMIC_AXES export (:26, :41-43): the ISTFT keeps everything after the first half window (256 T samples,
        UL-UNAS/STFT_Process.py:170-177, 317-326) and the wrapper slices audio[..., :audio_len] with audio_len = dynamic_keep, the caller-rate input length (:851, :888-889)."""
        self.w = {k: np.asarray(v, F32) for k, v in tensors.items()}
        self.plan, self.L = plan, int(in_len)
        self.T = self.L // HOP + 1
        self.fwd, self.inv, win = stft_tables(exact_dft)
        raw = np.zeros(NFFT + HOP * (self.T - 1), F32)
        for t in range(self.T):
            raw[t * HOP:t * HOP + NFFT] += (win * win).astype(F32)
        self.out_len = min(HOP * self.T, int(dynamic_keep)) if dynamic_keep > 0 else HOP * (self.T - 1)
        self.win_sum = raw[NFFT // 2:NFFT // 2 + self.out_len].copy()
        self.taps = {}

    # ---- building blocks -------------------------------------------------------------------------------------------------
    @staticmethod
    def _conv(x, w, b, kshape, stride, groups, deconv):
        """Causal (in T) Conv2d / ConvTranspose2d over (B, C, T, F) with stride on F, padding kf // 2 on F (:222-238, 264-267):
        conv: padding (kt - 1) on T then the last kt - 1 frames dropped; deconv: no T padding, last kt - 1 frames dropped."""
        B, Cin, T, Fi = x.shape
        kt, kf = kshape
        pf = kf // 2
        if deconv:
            cog = w.shape[1]
            Cout, cig = cog * groups, Cin // groups
            Fo = (Fi - 1) * stride - 2 * pf + kf
            out = np.zeros((B, Cout, T, Fo), F32)
            for g in range(groups):
                xs = x[:, g * cig:(g + 1) * cig]
                wg = w[g * cig:(g + 1) * cig]                       # (cig, cog, kt, kf)
                for a in range(kt):
                    xa = np.zeros_like(xs)
                    xa[:, :, a:] = xs[:, :, :T - a] if a else xs     # out[t] += x[t - a] * W[a]
                    for bb in range(kf):
                        contrib = np.einsum("bitf,io->botf", xa, wg[:, :, a, bb]).astype(F32)
                        for fi in range(Fi):
                            fo = fi * stride - pf + bb
                            if 0 <= fo < Fo:
                                out[:, g * cog:(g + 1) * cog, :, fo] += contrib[:, :, :, fi]
        else:
            Cout, cig = w.shape[0], w.shape[1]
            cog = Cout // groups
            Fo = (Fi + 2 * pf - kf) // stride + 1
            out = np.zeros((B, Cout, T, Fo), F32)
            xp = np.zeros((B, Cin, T + kt - 1, Fi + 2 * pf), F32)
            xp[:, :, kt - 1:, pf:pf + Fi] = x
            for g in range(groups):
                xs = xp[:, g * cig:(g + 1) * cig]
                wg = w[g * cog:(g + 1) * cog]                       # (cog, cig, kt, kf)
                for a in range(kt):
                    for bb in range(kf):
                        sl = xs[:, :, a:a + T, bb:bb + stride * (Fo - 1) + 1:stride]
                        out[:, g * cog:(g + 1) * cog] += np.einsum("bitf,oi->botf", sl, wg[:, :, a, bb]).astype(F32)
        return (out + b[None, :, None, None]).astype(F32)

    def _act(self, x, p):
        """AffinePReLU after fuse_for_export_ (:128-130): where(x > 0, pos, neg) * x + bias, tables per (channel, bin)."""
        pos, neg, bias = self.w[p + "pos"][None, :, None, :], self.w[p + "neg"][None, :, None, :], self.w[p + "bias"][None, :, None, :]
        return (np.where(x > 0, pos, neg) * x + bias).astype(F32)

    @staticmethod
    def _shuffle(x):
        """Shuffle (:197-208): out[:, j] = x[:, idx[j]], idx = interleave(0..C/2-1, C/2..C-1)."""
        C = x.shape[1]
        idx = np.stack((np.arange(C // 2), np.arange(C // 2) + C // 2), axis=1).reshape(-1)
        return x[:, idx]

    def _ctfa(self, x, p):
        """cTFA (:173-194) with FA (:132-170)."""
        w = self.w
        B, C, T, Fq = x.shape
        power = (x * x).astype(F32)
        zt = power.mean(axis=-1, dtype=F32)                                                   # (B, C, T)
        at = _gru(zt.transpose(2, 0, 1), w[p + "ta_weight_ih_l0"], w[p + "ta_weight_hh_l0"], w[p + "ta_bias_ih_l0"], w[p + "ta_bias_hh_l0"])
        at = (at @ w[p + "ta_fc_w"].T + w[p + "ta_fc_b"]).astype(F32).transpose(1, 2, 0)     # (B, C, T)
        at = _sig(at)[..., None]
        r = 4
        pad = (r - Fq % r) % r
        xf = power.mean(axis=1, dtype=F32)                                                    # (B, T, F)
        xf = np.concatenate((xf, np.zeros((B, T, pad), F32)), axis=-1)
        H = (Fq + pad) // r
        seq = xf.reshape(-1, H, r).transpose(1, 0, 2)                                         # (H, B*T, r)
        fwd = _gru(seq, w[p + "fa_weight_ih_l0"], w[p + "fa_weight_hh_l0"], w[p + "fa_bias_ih_l0"], w[p + "fa_bias_hh_l0"])
        bwd = _gru(seq, w[p + "fa_weight_ih_l0_reverse"], w[p + "fa_weight_hh_l0_reverse"], w[p + "fa_bias_ih_l0_reverse"],
                   w[p + "fa_bias_hh_l0_reverse"], reverse=True)
        af = (np.concatenate((fwd, bwd), axis=-1) @ w[p + "fa_fc_w"].T + w[p + "fa_fc_b"]).astype(F32)
        af = af.transpose(1, 0, 2).reshape(B, 1, T, H * r)[..., :Fq]
        return ((at * x) * _sig(af)).astype(F32)                                              # at * x * af (:194)

    def _block(self, x, spec):
        p, typ, cin, cout, width, k, stride, groups, deconv, last = spec
        w = self.w
        if typ == 0:                                                                           # XConvBlock (:264-273)
            x = self._conv(x, w[p + "conv_w"], w[p + "conv_b"], k, stride, groups, deconv)
            if not last:
                x = self._act(x, p + "act_")
            x = self._ctfa(x, p + "ctfa_")
            if not last and groups == 2:
                x = self._shuffle(x)
            return x
        if typ == 1:                                                                           # XDWSBlock (:342-357)
            h = self._conv(x, w[p + "pconv_w"], w[p + "pconv_b"], (1, 1), 1, groups, False)
            h = self._act(h, p + "pconv_act_")
            if groups == 2:
                h = self._shuffle(h)
            h = self._conv(h, w[p + "dconv_w"], w[p + "dconv_b"], k, stride, cout, deconv)
            if not last:
                h = self._act(h, p + "dconv_act_")
            return self._ctfa(h, p + "ctfa_")
        inp = x                                                                                # XMBBlocks (:433-453)
        x = self._conv(x, w[p + "pconv1_w"], w[p + "pconv1_b"], (1, 1), 1, groups, False)
        x = self._act(x, p + "pconv1_act_")
        if groups == 2:
            x = self._shuffle(x)
        x = self._conv(x, w[p + "dconv_w"], w[p + "dconv_b"], k, stride, cout, deconv)
        x = self._act(x, p + "dconv_act_")
        x = self._conv(x, w[p + "pconv2_w"], w[p + "pconv2_b"], (1, 1), 1, groups, False)
        x = self._ctfa(x, p + "ctfa_")
        if cin == cout and stride == 1:
            x = (x + inp).astype(F32)
        if not last and groups == 2:
            x = self._shuffle(x)
        return x

    def _grnn(self, x, p, bidirectional):
        """GRNN, unfused form (:518-524): two GRUs on the channel halves, outputs concatenated."""
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

    @staticmethod
    def _ln(x, wt, b):
        """LayerNorm((width, C), eps 1e-8) over the last two axes."""
        mu = x.mean(axis=(-2, -1), keepdims=True, dtype=F32)
        xc = (x - mu).astype(F32)
        var = (xc * xc).mean(axis=(-2, -1), keepdims=True, dtype=F32)
        return (xc / np.sqrt(var + F32(1e-8)) * wt + b).astype(F32)

    def _dpgrnn(self, x, p):
        """DPGRNN (:561-574), x (B, T, F, C)."""
        B, T, Fq, C = x.shape
        w = self.w
        intra_in = x.transpose(2, 0, 1, 3).reshape(Fq, B * T, C)
        y = (self._grnn(intra_in, p + "intra_rnn.", True) @ w[p + "intra_fc.weight"].T + w[p + "intra_fc.bias"]).astype(F32)
        y = y.reshape(Fq, B, T, C).transpose(1, 2, 0, 3)
        intra_out = (x + self._ln(y, w[p + "intra_ln.weight"], w[p + "intra_ln.bias"])).astype(F32)
        inter_in = intra_out.transpose(1, 0, 2, 3).reshape(T, B * Fq, C)
        y = (self._grnn(inter_in, p + "inter_rnn.", False) @ w[p + "inter_fc.weight"].T + w[p + "inter_fc.bias"]).astype(F32)
        y = y.reshape(T, B, Fq, C).transpose(1, 0, 2, 3)
        return (intra_out + self._ln(y, w[p + "inter_ln.weight"], w[p + "inter_ln.bias"])).astype(F32)

    # ---- the call ------------------------------------------------------------------------------------------------------------
    def process(self, pcm: np.ndarray) -> np.ndarray:
        """pcm int16 (B, L) -> int16 (B, 256 * (T - 1))   (ULUNAS_CUSTOM.forward, static shapes, 16 kHz in and out, no DC removal)."""
        assert pcm.ndim == 2 and pcm.shape[1] == self.L and pcm.dtype == np.int16
        wav = self.process_wave(pcm.astype(F32))
        return np.clip(wav * F32(32767.0), -32768.0, 32767.0).astype(np.int16)                # output_scale folded into the ISTFT kernel (:955)

    def process_rates(self, pcm: np.ndarray, in_rate: int, out_rate: int) -> np.ndarray:
        """The resampling sandwich of a dynamic export (:835-845, :851-868, :890-905) around the network.  pcm int16 (B, n), floor(n / (in_rate / 16000)) == in_len."""
        from gtcrn_sandwich import interpolate_scale
        x = pcm.astype(F32)                                                                    # :850
        in_scale = in_rate / 16000.0                                                           # :835
        if in_scale != 1.0:     # :852-868: interpolate, * INV_INT16 (down) or * INV_INT16, interpolate (up) -- the same numbers: a power of two commutes with the rounding
            x = interpolate_scale(x, 1.0 / in_scale)
        wav = self.process_wave(x)
        out_scale = out_rate / 16000.0                                                         # :836
        if out_scale < 1.0:                                                                    # :890-896
            wav = interpolate_scale(wav, out_scale)
        wav = (wav * F32(32767.0)).astype(F32)                                                 # :897-898 (or folded into the ISTFT at the model rate, :955)
        if out_scale > 1.0:                                                                    # :899-905
            wav = interpolate_scale(wav, out_scale)
        return np.clip(wav, -32768.0, 32767.0).astype(np.int16)

    def process_wave(self, samples: np.ndarray) -> np.ndarray:
        """fp32 samples in PCM units (B, L) -> the normalised fp32 waveform (B, out_len)."""
        assert samples.ndim == 2 and samples.shape[1] == self.L and samples.dtype == F32
        B, T, half = samples.shape[0], self.T, NFFT // 2
        x = samples * F32(1.0 / 32768.0)                                                      # input_scale folded into the STFT kernel (:944)
        xp = np.concatenate((x[:, 1:half + 1][:, ::-1], x, x[:, -(half + 1):-1][:, ::-1]), axis=1)
        frames = np.stack([xp[:, t * HOP:t * HOP + NFFT] for t in range(T)], axis=1)          # (B, T, 512)
        spec = (frames @ self.fwd.T).astype(F32)                                               # (B, T, 514)
        re, im = spec[..., :FB], spec[..., FB:]
        power = (re * re + im * im).astype(F32)                                                # (:877)
        erb = self.w["erb_filters"]                                                            # (64, 192)
        feat = np.log(np.maximum(power, F32(1e-24))).astype(F32)                               # (:725) the 0.5 / ln 10 lives in the first conv
        feat = np.concatenate((feat[..., :65], feat[..., 65:] @ erb.T), axis=-1).astype(F32)[:, None]   # ERB.bm (:97-100): (B, 1, T, 129)
        en_outs = []
        h = feat
        for spec_ in self.plan[:5]:
            h = self._block(h, spec_)
            en_outs.append(h)
        h = h.transpose(0, 2, 3, 1)
        for i in range(2):
            h = self._dpgrnn(h, f"dpgrnn.{i}.")
        h = h.transpose(0, 3, 1, 2)
        for i, spec_ in enumerate(self.plan[5:]):
            h = self._block((h + en_outs[4 - i]).astype(F32), spec_)                           # (:647-648)
        m = _sig(h)[:, 0]                                                                      # (B, T, 129)
        mask = np.concatenate((m[..., :65], m[..., 65:] @ erb), axis=-1).astype(F32)           # ERB.bs (:102-105): (B, T, 257)
        self.taps["mask"] = mask.copy()
        out_re, out_im = (re * mask).astype(F32), (im * mask).astype(F32)                      # (:880)
        fr = (np.concatenate((out_re, out_im), axis=-1) @ self.inv).astype(F32)                # (B, T, 512)
        raw = np.zeros((B, NFFT + HOP * (T - 1)), F32)
        for t in range(T):
            raw[:, t * HOP:t * HOP + NFFT] += fr[:, t]
        wav = (raw[:, half:half + self.out_len] / self.win_sum).astype(F32)
        self.taps["wav"] = wav.copy()
        return wav
