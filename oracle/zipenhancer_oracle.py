"""zipenhancer_oracle.py — CPU ORACLE for the ZipEnhancer hot path.  TEST INFRASTRUCTURE ONLY.

A numpy fp32 restatement of ``ZipEnhancer.forward`` (ZipEnhancer/Export_ZipEnhancer.py:818-927) over the FUSED tensors its
constructor registers (:437-664) -- the tensor set of ``audio_denoiser_onnx_amd/zipenhancer.py::blob_tensors`` -- each step citing
the reference lines it follows: the ten forward overrides (:118-339), the causal dense blocks (:701-757), the sub-pixel decoder
(:759-780), the dual-path wiring (:782-816) and the STFT / ISTFT pair (ZipEnhancer/STFT_Process.py:268-300).

Pinned (tests/test_zipenhancer.py) against fixtures made by RUNNING the reference's own ``ZipEnhancer`` class here
(tools/make_golden_zipenhancer.py) over a stand-in network tree: the reference takes its network from the modelscope package
(absent here), so the tool supplies modules with the attribute paths the reference reads, standard torch leaves (Conv2d,
InstanceNorm2d, PReLU, Linear, Conv1d) and the reference's OWN forwards installed on them exactly as ``apply_onnx_export_patches``
does (:342-355).  Pinned: every line of the reference that runs (constructor folds, overrides, wrapper forward).  PARITY UNPINNED:
the leaf geometry (channel / head / kernel sizes, the CompactRelPositionalEncoding table formula, FeedforwardModule's
in_proj -> out_proj composition), which lives in modelscope; see audio_denoiser_onnx_amd/zipenhancer.py.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product never does.
"""
from __future__ import annotations

import math

import numpy as np

F32 = np.float32
NFFT, HOP = 400, 100                      # Export_ZipEnhancer.py:47-49
FBINS = NFFT // 2 + 1


def hann_periodic(n: int) -> np.ndarray:
    """torch.hann_window(n, periodic=True) in fp32 (ZipEnhancer/STFT_Process.py:93)."""
    k = np.arange(n, dtype=F32)
    return (np.cos(k * F32(2.0 * np.pi / n)) * F32(-0.5) + F32(0.5)).astype(F32)


def stft_kernels(exact: bool = False):
    """Forward / inverse windowed-DFT matrices (2 * 201, 400) (STFT_Process.py:205-243): angles fp32(2 pi / N) * f * t in fp32.
    exact=True: exactly reduced angles (test knob)."""
    w = hann_periodic(NFFT)
    if exact:
        k = (np.arange(FBINS, dtype=np.int64)[:, None] * np.arange(NFFT, dtype=np.int64)[None, :]) % NFFT
        ang = 2.0 * np.pi * k.astype(np.float64) / NFFT
        c, s = np.cos(ang).astype(F32), np.sin(ang).astype(F32)
    else:
        omega = (F32(2.0 * np.pi / NFFT) * np.arange(FBINS, dtype=F32)[:, None]) * np.arange(NFFT, dtype=F32)[None, :]
        c, s = np.cos(omega).astype(F32), np.sin(omega).astype(F32)
    fwd = np.concatenate((c * w[None, :], -s * w[None, :]), axis=0).astype(F32)
    scale = np.full((FBINS, 1), 2.0, F32)
    scale[0] = 1.0
    scale[FBINS - 1] = 1.0
    inv_n = F32(1.0 / NFFT)
    inv = np.concatenate((((scale * c) * inv_n) * w[None, :], ((scale * -s) * inv_n) * w[None, :]), axis=0).astype(F32)
    return fwd, inv, w


def pos_table(pos_dim: int, length: int) -> np.ndarray:
    """Rows x = -(length - 1) .. length - 1 of CompactRelPositionalEncoding's table (the slice ``_pos_enc`` takes, :690-699).
    The table formula is modelscope's (= icefall Zipformer2): compressed-log position -> atan -> cos / sin at integer
    frequencies, last column 1.  Evaluated in fp32 like torch."""
    x = np.arange(-(length - 1), length, dtype=F32)[:, None]
    freqs = (1 + np.arange(pos_dim // 2)).astype(F32)
    cl = F32(pos_dim ** 0.5)
    xc = (cl * np.sign(x) * (np.log(np.abs(x) + cl) - F32(math.log(pos_dim ** 0.5)))).astype(F32)
    length_scale = F32(1.0 * pos_dim / (2.0 * math.pi))
    xa = np.arctan(xc / length_scale).astype(F32)
    pe = np.zeros((x.shape[0], pos_dim), F32)
    pe[:, 0::2] = np.cos(xa * freqs)
    pe[:, 1::2] = np.sin(xa * freqs)
    pe[:, -1] = 1.0
    return pe


def softplus(x: np.ndarray) -> np.ndarray:
    """F.softplus (beta 1, threshold 20)."""
    return np.where(x > 20.0, x, np.log1p(np.exp(np.minimum(x, F32(20.0))))).astype(F32)


def swoosh_l(x):
    return (softplus(x - F32(4.0)) - F32(0.08) * x).astype(F32)        # offset folded into the next bias (:135-136)


def swoosh_r(x):
    return (softplus(x - F32(1.0)) - F32(0.08) * x).astype(F32)        # (:138)


def sigmoid(x):
    return (F32(1.0) / (F32(1.0) + np.exp(-x))).astype(F32)


def linear(x, w, b=None):
    y = x @ w.T
    return (y + b).astype(F32) if b is not None else y.astype(F32)


def instance_norm(x, w, b, eps=1e-5):
    """F.instance_norm over (T, F) of a channels-last (B, T, F, C) map: biased variance, affine (:741-749)."""
    m = x.mean(axis=(1, 2), keepdims=True, dtype=np.float64)
    v = ((x - m) ** 2).mean(axis=(1, 2), keepdims=True, dtype=np.float64)
    return (((x - m) / np.sqrt(v + eps)) * w + b).astype(F32)


def prelu(x, a):
    return np.where(x >= 0, x, x * a).astype(F32)


def shift(x, axis, by):
    """y[i] = x[i - by] along ``axis`` with zero fill."""
    if by == 0:
        return x
    y = np.zeros_like(x)
    src = [slice(None)] * x.ndim
    dst = [slice(None)] * x.ndim
    n = x.shape[axis]
    if by > 0:
        src[axis], dst[axis] = slice(0, n - by), slice(by, n)
    else:
        src[axis], dst[axis] = slice(-by, n), slice(0, n + by)
    if abs(by) < n:
        y[tuple(dst)] = x[tuple(src)]
    return y


def causal_conv_2x3(x, w, b, dilation):
    """conv2d(x, w, padding=(d, 1), dilation=(d, 1))[:, :, :-d] on channels-last x (B, T, F, Cin); w (Cout, Cin, 2, 3) (:711-719):
    out[t, f] = sum_{kt, kf} w[:, :, kt, kf] . x[t - (1 - kt) d, f + kf - 1]."""
    acc = None
    for kt in range(2):
        xt = shift(x, 1, (1 - kt) * dilation)
        for kf in range(3):
            y = shift(xt, 2, 1 - kf) @ w[:, :, kt, kf].T
            acc = y if acc is None else acc + y
    return (acc + b).astype(F32)


class ZipEnhancerOracle:
    """tensors: blob tensors by name; window_len: samples per window (whole hops); n_win: windows per call (batch-fold)."""

    def __init__(self, tensors: dict, window_len: int, n_win: int = 1, exact_dft: bool = False, dynamic: bool = False):
        self.w = {k: np.asarray(v, F32) for k, v in tensors.items()}
        c = [int(round(float(v))) for v in self.w["zip_config"].reshape(-1)]
        (self.C, self.H, self.q, self.p, self.v, self.pos_dim, self.ff, self.K, self.dt1, self.df1, self.dt2, self.df2, self.r, self.depth) = c[:14]
        self.hid = self.C * 3 // 4
        self.ff1 = self.ff * 3 // 4
        # dynamic = the DYNAMIC_AXES export (Export_ZipEnhancer.py:31, :61, :898-899; STFT_Process.py:297-299): any length, T = L // hop + 1 frames, 100 (T - 1) samples
        # out, the overlap-add DIVIDED by its denominator instead of multiplied by the precomputed reciprocal
        if window_len % HOP and not dynamic:
            raise ValueError("window_len must be whole hops")
        self.dynamic = bool(dynamic)
        self.L, self.n_win, self.T = window_len, n_win, window_len // HOP + 1
        self.fwd, self.inv, win = stft_kernels(exact_dft)
        raw = np.zeros(NFFT + HOP * (self.T - 1), F32)
        for t in range(self.T):
            raw[t * HOP:t * HOP + NFFT] += win * win
        self.inv_win_sum = (F32(1.0) / raw[NFFT // 2:raw.size - NFFT // 2]).astype(F32)     # static_norm (STFT_Process.py:245-249)
        self.win_sum = raw[NFFT // 2:raw.size - NFFT // 2].astype(F32)
        self._pos_cache = {}

    # ---- STFT / ISTFT (STFT_Process.py:268-281, 291-296) -------------------------------------------------------------------
    def stft(self, x):
        xp = np.pad(x, ((0, 0), (NFFT // 2, NFFT // 2)), mode="reflect")
        idx = np.arange(self.T)[:, None] * HOP + np.arange(NFFT)[None, :]
        spec = np.einsum("btn,cn->bct", xp[:, idx], self.fwd, dtype=F32, optimize=True).astype(F32)
        return spec[:, :FBINS], spec[:, FBINS:]

    def istft(self, packed):
        B = packed.shape[0]
        fr = np.einsum("bct,cn->btn", packed, self.inv, dtype=F32, optimize=True).astype(F32)
        raw = np.zeros((B, NFFT + HOP * (self.T - 1)), F32)
        for t in range(self.T):
            raw[:, t * HOP:t * HOP + NFFT] += fr[:, t]
        if self.dynamic:
            return (raw[:, NFFT // 2:raw.shape[1] - NFFT // 2] / self.win_sum).astype(F32)
        return (raw[:, NFFT // 2:raw.shape[1] - NFFT // 2] * self.inv_win_sum).astype(F32)

    # ---- attention weights with the relative-position term (:232-289) -------------------------------------------------------
    def pos_proj(self, prefix, n):
        key = (prefix, n)
        if key not in self._pos_cache:
            pe = pos_table(self.pos_dim, n)                                           # (2n - 1, pos_dim)
            pos = linear(pe, self.w[prefix + "pos_w"])                                # (:599)
            self._pos_cache[key] = pos.reshape(2 * n - 1, self.H, self.p).transpose(1, 2, 0).copy()   # (head, pos_head_dim, 2n - 1) (:600-602)
        return self._pos_cache[key]

    def attn_weights(self, prefix, proj):
        """proj (S, n, heads * (2 q + p)) with per-head [q | k | p] rows -> softmax weights (S, heads, n, n)."""
        S, n, _ = proj.shape
        x = proj.reshape(S, n, self.H, 2 * self.q + self.p).transpose(0, 2, 1, 3)     # (:250)
        qv, kv, pv = x[..., :self.q], x[..., self.q:2 * self.q], x[..., 2 * self.q:]
        scores = qv @ kv.transpose(0, 1, 3, 2)                                        # (:259)
        ps = pv @ self.pos_proj(prefix, n)[None]                                      # (S, H, n, 2n - 1) (:269)
        i = np.arange(n)[:, None]
        j = np.arange(n)[None, :]
        rel = np.take_along_axis(ps, np.broadcast_to((n - 1 - i + j)[None, None], (S, self.H, n, n)), axis=3)   # out[i, j] = ps[i, n-1-i+j] (:270-284)
        s = (scores + rel).astype(F32)
        s = s - s.max(axis=-1, keepdims=True)
        e = np.exp(s).astype(F32)
        return (e / e.sum(axis=-1, keepdims=True, dtype=F32)).astype(F32)             # (:289)

    # ---- one fused Zipformer2 encoder layer on (S sequences, n positions, C) (:143-187) --------------------------------------
    def layer(self, prefix, src):
        w = self.w
        g = lambda k: w[prefix + k]
        S, n, C = src.shape
        orig = src
        proj = linear(src, g("attn_ff1_w"), g("attn_ff1_b"))                          # (:148-149)
        attn_dim = self.H * (2 * self.q + self.p)
        aw = self.attn_weights(prefix, proj[..., :attn_dim])                          # (:154-159)
        src = src + linear(swoosh_l(proj[..., attn_dim:]), g("ff1_out_w"), g("ff1_out_b"))    # (:160, :131-140)
        # NonlinAttention on head 0 (:167, :304-317)
        x = linear(src, g("nonlin_in_w"), g("nonlin_in_b"))
        s, xm, y = x[..., :self.hid], x[..., self.hid:2 * self.hid], x[..., 2 * self.hid:]
        xm = (aw[:, 0] @ (xm * np.tanh(s))).astype(F32) * y
        src = src + linear(xm, g("nonlin_out_w"), g("nonlin_out_b"))

        def self_attn(i, src):                                                        # (:292-301)
            v = linear(src, g(f"sa{i}_in_w"), g(f"sa{i}_in_b")).reshape(S, n, self.H, self.v).transpose(0, 2, 1, 3)
            o = (aw @ v).transpose(0, 2, 1, 3).reshape(S, n, self.H * self.v)
            return linear(o.astype(F32), g(f"sa{i}_out_w"), g(f"sa{i}_out_b"))

        def conv_module(i, src):                                                      # (:320-339); depthwise Conv1d(k, padding k // 2) along the sequence
            x = linear(src, g(f"conv{i}_in_w"), g(f"conv{i}_in_b"))
            xm = x[..., :C] * sigmoid(x[..., C:])
            dw, half = g(f"conv{i}_dw_w"), self.K // 2
            acc = np.zeros_like(xm)
            for k in range(self.K):
                acc += shift(xm, 1, half - k) * dw[:, k]
            xm = (acc + g(f"conv{i}_dw_b")).astype(F32)
            return linear(swoosh_r(xm), g(f"conv{i}_out_w"), g(f"conv{i}_out_b"))

        def ff(i, src):                                                               # FeedforwardModule: in_proj -> SwooshL + Linear
            return linear(swoosh_l(linear(src, g(f"ff{i}_in_w"), g(f"ff{i}_in_b"))), g(f"ff{i}_out_w"), g(f"ff{i}_out_b"))

        src = src + self_attn(1, src)                                                 # (:168)
        src = src + conv_module(1, src)                                               # (:169)
        src = src + ff(2, src)                                                        # (:170)
        src = (orig + (src - orig) * g("bypass_mid")).astype(F32)                     # (:171, :190-191)
        src = src + self_attn(2, src)                                                 # (:172)
        src = src + conv_module(2, src)                                               # (:173)
        src = src + ff(3, src)                                                        # (:174)
        d = src - g("norm_bias")
        norm = np.sqrt((d * d).sum(axis=-1, keepdims=True, dtype=F32)).astype(F32)    # (:178-179)
        return ((src / norm) * g("final_norm_scale") + orig * g("final_residual_scale")).astype(F32)   # (:180-183)

    def dualpath(self, e, x):
        """x (B, T', F', C) channels-last: frequency-path layer over each frame's sub-bands, then time-path layer (:782-792)."""
        B, T, F, C = x.shape
        x = self.layer(f"enc{e}_f_", x.reshape(B * T, F, C)).reshape(B, T, F, C)
        x = self.layer(f"enc{e}_t_", x.transpose(0, 2, 1, 3).reshape(B * F, T, C)).reshape(B, F, T, C)
        return np.ascontiguousarray(x.transpose(0, 2, 1, 3))

    @staticmethod
    def _down(x, axis, ds, wts):
        """SimpleDownsample along ``axis``: pad to a multiple by repeating the last position, weighted sum of each group (:194-218)."""
        n = x.shape[axis]
        dn = -(-n // ds)
        pad = dn * ds - n
        if pad:
            last = np.take(x, [n - 1], axis=axis)
            x = np.concatenate([x] + [last] * pad, axis=axis)
        shp = list(x.shape)
        shp[axis:axis + 1] = [dn, ds]
        wshape = [1] * len(shp)
        wshape[axis + 1] = ds
        return (x.reshape(shp) * wts.reshape(wshape)).sum(axis=axis + 1, dtype=F32).astype(F32)

    def downsampled(self, e, x):
        """(:794-816)"""
        B, T, F, C = x.shape
        w = self.w
        y = self._down(x, 1, self.dt2, w[f"enc{e}_down_t_w"])                         # time first (:799)
        y = self._down(y, 2, self.df2, w[f"enc{e}_down_f_w"])                         # then sub-bands (:801)
        y = self.dualpath(e, y)
        y = y * w[f"enc{e}_out_scale"]                                                # (:812)
        y = np.repeat(y, self.df2, axis=2)[:, :, :F]                                  # upsample_f, crop (:813-814)
        y = np.repeat(y, self.dt2, axis=1)[:, :T]                                     # upsample_t, crop (:814-815)
        return (x * w[f"enc{e}_res_scale"] + y).astype(F32)                           # (:816)

    def dense_block(self, pre, x, groups=1):
        """DenseBlockV2, causal in time (:701-723); groups = 2: the fused mask | phase pair (:725-757)."""
        w = self.w
        C = self.C
        skips = [[x] for _ in range(groups)]
        out = None
        for i in range(self.depth):
            wt, b = w[f"{pre}{i}_w"], w[f"{pre}{i}_b"]
            ys = []
            for gi in range(groups):
                inp = np.concatenate(skips[gi], axis=-1)                              # newest first (:722, :753-756)
                ys.append(causal_conv_2x3(inp, wt[gi * C:(gi + 1) * C], b[gi * C:(gi + 1) * C], 1 << i))
            y = np.concatenate(ys, axis=-1)
            y = prelu(instance_norm(y, w[f"{pre}{i}_nw"], w[f"{pre}{i}_nb"]), w[f"{pre}{i}_pr"])
            for gi in range(groups):
                skips[gi].insert(0, y[..., gi * C:(gi + 1) * C])
            out = y
        return out

    def process(self, pcm, taps=False, spectrum=None):
        """int16 (B, n_win * L) -> (int16 (B, n_win * L), fp32 pre-cast waveform, taps).
        ``spectrum`` = (re, im), each (windows, 201, T): continue from a given STFT instead of this oracle's own.  The phase feature
        atan2(im, re + 1e-5) (:844) has a branch cut on the negative real axis, and the two reflect-padded edge frames are symmetric, so
        their spectra are real up to round-off: for a low bin with re < 0 the SIGN of that round-off -- i.e. +pi or -pi -- depends on the
        summation order of whoever computed the STFT (torch's conv1d, ONNX Runtime, this numpy einsum and the HIP GEMM all differ).
        Tests therefore pin the network on identical spectra and the spectrum separately."""
        w = self.w
        C = self.C
        pcm = np.asarray(pcm)
        Bc = pcm.shape[0]
        audio = pcm.astype(F32).reshape(Bc * self.n_win, self.L)                      # fold (:837); int16 amplitude (:819)
        norm = np.sqrt(np.mean(audio * audio, axis=-1, keepdims=True, dtype=F32) + F32(1e-6)).astype(F32)   # (:839)
        audio = (audio / norm).astype(F32)
        re, im = self.stft(audio) if spectrum is None else (np.asarray(spectrum[0], F32), np.asarray(spectrum[1], F32))
        if taps:
            tp0 = {"spec_re": re, "spec_im": im}
        mag = np.power(re * re + im * im + F32(1e-9), F32(0.15)).astype(F32)           # (:843)
        pha = np.arctan2(im, re + F32(1e-5)).astype(F32)                              # (:844)
        x = np.stack((mag, pha), axis=-1).transpose(0, 2, 1, 3)                       # (B, T, 201, 2) channels-last (:850)
        x = prelu(instance_norm(linear(x, w["enc_conv1_w"], w["enc_conv1_b"]), w["enc_norm1_w"], w["enc_norm1_b"]), w["enc_prelu1"])   # (:851)
        x = self.dense_block("enc_dense", x)                                          # (:852)
        w2 = w["enc_conv2_w"]                                                         # Conv2d (1, 3), stride (1, 2), padding (0, 1) (:853)
        xp = np.pad(x, ((0, 0), (0, 0), (1, 1), (0, 0)))
        F = (FBINS + 2 - 3) // 2 + 1
        y = sum(xp[:, :, kf:kf + 2 * F - 1:2] @ w2[:, :, 0, kf].T for kf in range(3)) + w["enc_conv2_b"]
        x = prelu(instance_norm(y.astype(F32), w["enc_norm2_w"], w["enc_norm2_b"]), w["enc_prelu2"])
        tp = dict(tp0, enc_in=x) if taps else {}
        for e in range(4):                                                            # (:860-863)
            x = self.downsampled(e, x) if e in (1, 2) else self.dualpath(e, x)
            if taps:
                tp[f"enc{e}"] = x
        d = self.dense_block("dec_dense", x, groups=2)                                # (:864)
        # sub-pixel up-sampling pair (:759-780): conv (1, 3) pad 1 to C * r channels per group, channel c * r + u -> sub-band f * r + u
        r = self.r
        dp = np.pad(d, ((0, 0), (0, 0), (1, 1), (0, 0)))
        ups = []
        for gi in range(2):
            wt = w["dec_up_w"][gi * C * r:(gi + 1) * C * r]
            y = sum(dp[:, :, kf:kf + F, gi * C:(gi + 1) * C] @ wt[:, :, 0, kf].T for kf in range(3)) + w["dec_up_b"][gi * C * r:(gi + 1) * C * r]
            B_, T_ = y.shape[0], y.shape[1]
            ups.append(y.reshape(B_, T_, F, C, r).transpose(0, 1, 2, 4, 3).reshape(B_, T_, F * r, C))
        u = np.concatenate(ups, axis=-1).astype(F32)
        u = prelu(instance_norm(u, w["dec_up_nw"], w["dec_up_nb"]), w["dec_up_pr"])
        mx, px = u[..., :C], u[..., C:]
        Fo = F * r - 1
        m = sum(mx[:, :, kf:kf + Fo] @ w["mask_out_w"][:, :, 0, kf].T for kf in range(2)) + w["mask_out_b"]        # (B, T, 201, 1) (:868)
        ph = sum(px[:, :, kf:kf + Fo] @ w["phase_out_w"][:, :, 0, kf].T for kf in range(2)) + w["phase_out_b"]     # (B, T, 201, 2) (:874-877)
        m, ph = m.astype(F32), ph.astype(F32)
        if taps:
            tp["mask"], tp["phase_ri"] = m[..., 0], ph
        magnitude = np.power(np.maximum(m, F32(0.0)), F32(1.0 / 0.3)).astype(F32)      # (:882-883)
        pn = np.sqrt((ph * ph).sum(axis=-1, keepdims=True, dtype=F32)).astype(F32)     # (:885)
        has = pn > 0.0
        ph = np.where(has, ph, np.array([1.0, 0.0], F32))                              # (:886-887)
        pn = np.where(has, pn, F32(1.0))
        ri = (ph * (magnitude / pn)).astype(F32)                                       # (:891)
        packed = np.concatenate((ri[..., 0].transpose(0, 2, 1), ri[..., 1].transpose(0, 2, 1)), axis=1)   # (B, 402, T) (:892)
        wave = (self.istft(np.ascontiguousarray(packed)) * norm).astype(F32)          # (:893, :900)
        wave = wave.reshape(Bc, -1)                                                   # (:902); 100 (T - 1) samples per window (== L for whole-hop windows)
        y = np.where(np.isnan(wave), F32(0.0), wave)                                  # (:917)
        out = np.clip(y, -32768.0, 32767.0).astype(np.int16)                          # truncation toward zero (:918)
        if taps:
            tp["packed"] = packed
        return out, wave, tp
