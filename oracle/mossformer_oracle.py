"""mossformer_oracle.py — CPU ORACLE for the MossFormer2-SS-16K hot path.  TEST INFRASTRUCTURE ONLY.

A numpy fp32 restatement of ``MOSSFORMER_SS.forward`` / ``_run_mdl`` / ``norm_audio`` / ``group_norm_static``
(MossFormer2_SS_16K/Export_MossFormer2_SS_16K.py:398-662) over the FUSED buffers its constructor registers (:130-395) and the
scalar attributes it derives (eps values, slopes, group geometry: passed in as ``scalars``), each step citing the lines it
follows.  Pinned (tests/test_mossformer.py) against fixtures made by running the reference's own constructor + forward in the
build container (tools/make_golden_mossformer.py: stand-in network tree of the published geometry, fused buffers filled by
audio_denoiser_onnx_amd/weightgen.py).  Only tests/ may import this module; the product never does.
"""
from __future__ import annotations

import numpy as np

F32 = np.float32
ENC_K, ENC_S = 16, 8                      # encoder / decoder kernel and stride (:56-57 of the reference's argument block)


def _silu(x):
    return (x / (F32(1.0) + np.exp(-x))).astype(F32)


def _sigmoid(x):
    return (F32(1.0) / (F32(1.0) + np.exp(-x))).astype(F32)


def _layer_norm(x, w, b, eps):
    """F.layer_norm over the last axis (biased variance)."""
    mu = x.mean(axis=-1, keepdims=True, dtype=F32)
    xc = (x - mu).astype(F32)
    var = (xc * xc).mean(axis=-1, keepdims=True, dtype=F32)
    y = (xc / np.sqrt(var + F32(eps))).astype(F32)
    if w is not None:
        y = (y * w + b).astype(F32)
    return y


def _dwconv(x, c, pad):
    """x (B, n, C), c (C, 1, K): depthwise cross-correlation over time with zero padding (F.conv1d groups=C)."""
    B, n, C = x.shape
    K = c.shape[-1]
    xp = np.zeros((B, n + 2 * pad, C), F32)
    xp[:, pad:pad + n] = x
    out = np.zeros_like(x)
    for k in range(K):
        out += xp[:, k:k + n] * c[:, 0, k]
    return out.astype(F32)


def interpolate_linear(x, out_len):
    """F.interpolate(x, size=out_len, mode='linear', align_corners=False) over the last axis, fp32 (the export's resampling edges :562-571,
    :625-640): src = (in / out) * (dst + 0.5) - 0.5 clamped at 0, y = (1 - l) x[i0] + l x[min(i0 + 1, in - 1)]."""
    n = x.shape[-1]
    scale = F32(n / out_len)
    src = np.maximum(scale * (np.arange(out_len, dtype=F32) + F32(0.5)) - F32(0.5), F32(0.0)).astype(F32)
    i0 = np.minimum(src.astype(np.int64), n - 1)
    i1 = np.minimum(i0 + 1, n - 1)
    l1 = (src - i0.astype(F32)).astype(F32)
    return ((F32(1.0) - l1) * x[..., i0] + l1 * x[..., i1]).astype(F32)


def interpolate_scale(x, factor):
    """F.interpolate(x, scale_factor=factor, mode='linear', align_corners=False): floor(n * factor) samples, source step 1 / factor in fp32 (the DYNAMIC_AXES export's
    edges, :565-577, :634-646; recompute_scale_factor is left at None, so the given factor -- not out / in -- is the scale)."""
    n = x.shape[-1]
    out_len = int(np.floor(float(n) * float(factor)))
    src = np.maximum(F32(1.0 / float(factor)) * (np.arange(out_len, dtype=F32) + F32(0.5)) - F32(0.5), F32(0.0)).astype(F32)
    i0 = np.minimum(src.astype(np.int64), n - 1)
    i1 = np.minimum(i0 + 1, n - 1)
    l1 = (src - i0.astype(F32)).astype(F32)
    return ((F32(1.0) - l1) * x[..., i0] + l1 * x[..., i1]).astype(F32)


class MossFormerOracle:
    """tensors: the fused buffers by registered name (encoder_w, front_w, fl_in_w_i, ..., tail_gate_w, decoder_w) plus emb_pos
    (1, 512, n) and rot_cos / rot_signed_sin (1, n, 1, rot_dim); scalars: the dict the golden tool stores."""

    def __init__(self, tensors: dict, scalars: dict, layers: int, window: int, dynamic: bool = False):
        # dynamic: the DYNAMIC_AXES export (:24): 1 / frames multiplies the reduced linear-attention product at run time (:430, :500-501) instead of being folded into
        # the linear keys' OffsetScale row (:183, :236-241) -- `tensors` then hold the UNfolded row --, and the resampling edges interpolate by scale factor.
        self.dynamic = bool(dynamic)
        self.w = {k: np.asarray(v, F32) for k, v in tensors.items()}
        self.s = scalars
        self.layers, self.W = int(layers), int(window)
        self.n = (self.W - ENC_K) // ENC_S + 1
        self.group = int(scalars["flash_group_size"])
        self.pad = (self.group - self.n % self.group) % self.group
        self.taps = {}

    # group_norm_static (:398-401): one instance-norm over all of (512, n) per window
    @staticmethod
    def _window_norm(x, eps):
        B = x.shape[0]
        y = x.reshape(B, -1)
        mu = y.mean(axis=1, keepdims=True, dtype=F32)
        yc = (y - mu).astype(F32)
        var = (yc * yc).mean(axis=1, keepdims=True, dtype=F32)
        return (yc / np.sqrt(var + F32(eps))).astype(F32).reshape(x.shape)

    def _norm_audio(self, pcm):
        """(B, W) int16 -> normalised (B, W), rms_in (B,)   (:403-423)."""
        eps, nf = F32(1e-6), F32(self.s["norm_factor"])
        x = pcm.astype(F32) * F32(1.0 / 32768.0)            # pcm: int16, or floats in int16 units after the input interpolation
        p = (x * x).astype(F32)
        avg = p.mean(axis=1, keepdims=True, dtype=F32)
        rms = np.sqrt(avg).astype(F32)
        scalar = (nf / (rms + eps)).astype(F32)
        m = (p > avg).astype(F32)
        high = np.sqrt((p * m).sum(axis=1, keepdims=True, dtype=F32) / np.maximum(m.sum(axis=1, keepdims=True, dtype=F32), F32(1.0))).astype(F32)
        scalarx = (nf / (high * scalar + eps)).astype(F32)
        x = ((x * scalar) * scalarx).astype(F32)
        gp = (scalar * scalarx).astype(F32)
        undo = (F32(1.0) / (gp + eps)).astype(F32)
        return x, (rms * gp * undo * F32(32767.0)).astype(F32)[:, 0]

    def _flash(self, h, i):
        w, s, n, G, g = self.w, self.s, self.n, (self.n + self.pad) // self.group, self.group
        B = h.shape[0]
        rd = int(s["rot_dim"])
        shift = np.concatenate((np.zeros((B, 1, 256), F32), h[:, :-1, :256]), axis=1)                       # token shift (:454-456)
        nx = np.concatenate((shift, h[..., 256:]), axis=-1)
        base = (nx / np.maximum(np.sqrt((nx * nx).sum(-1, keepdims=True, dtype=F32)), F32(s["fl_norm_eps"]))).astype(F32)
        proj = _silu((base @ w[f"fl_in_w_{i}"].T + w[f"fl_in_b_{i}"]).astype(F32))                           # (:459)
        proj = (proj + _dwconv(proj, w[f"fl_in_c_{i}"], int(s["dw_pad"]))).astype(F32)                      # (:460)
        vu, qk = proj[..., :2048], proj[..., 2048:]
        v, u = vu[..., :1024], vu[..., 1024:]
        sc = (qk[:, :, None, :] * w[f"qkos_gamma_{i}"] + w[f"qkos_beta_{i}"]).astype(F32)                   # (B, n, 4, 128) (:466)
        mid = sc[..., :rd]
        perm = np.arange(rd).reshape(-1, 2)[:, ::-1].reshape(-1)
        rot = (mid * w["rot_cos"][:, :n] + mid[..., perm] * w["rot_signed_sin"][:, :n]).astype(F32)        # (:467-472)
        sc = np.concatenate((rot, sc[..., rd:]), axis=-1)
        if self.pad:
            sc = np.concatenate((sc, np.zeros((B, self.pad, 4, sc.shape[-1]), F32)), axis=1)               # (:473-474)
            vug = np.concatenate((vu, np.zeros((B, self.pad, 2048), F32)), axis=1)
        else:
            vug = vu
        sc = sc.reshape(B, G, g, 4, -1)
        vug = vug.reshape(B, G, g, 2048)
        qq, lq, qk_, lk = sc[:, :, :, 0], sc[:, :, :, 1], sc[:, :, :, 2], sc[:, :, :, 3]
        attn = np.maximum(qq @ qk_.transpose(0, 1, 3, 2), F32(0.0)).astype(F32)                             # (:487)
        quad = ((attn * attn) @ vug).astype(F32)                                                            # (:488)
        lkf = lk.reshape(B, G * g, -1).transpose(0, 2, 1)                                                   # (B, 128, padded)
        lin_kvu = (lkf @ vug.reshape(B, G * g, 2048)).astype(F32)                                           # (:491-492), 1/n folded into head 3
        if self.dynamic:
            lin_kvu = (lin_kvu * F32(1.0 / n)).astype(F32)                                                  # (:500-501)
        lin = (lq @ lin_kvu[:, None]).astype(F32)                                                           # (:495)
        att = (quad + lin).reshape(B, G * g, 2048)[:, :n]
        av, au = att[..., :1024], att[..., 1024:]
        out = ((au * v) * _sigmoid(av * u)).astype(F32)                                                     # (:499)
        y = (out / np.maximum(np.sqrt((out * out).sum(-1, keepdims=True, dtype=F32)), F32(s["fl_out_norm_eps"]))).astype(F32)
        y = _silu((y @ w[f"fl_out_w_{i}"].T + w[f"fl_out_b_{i}"]).astype(F32))                              # (:503)
        y = (y + _dwconv(y, w[f"fl_out_c_{i}"], int(s["dw_pad"]))).astype(F32)
        return (h + y).astype(F32)                                                                          # (:505)

    def _fsmn(self, h, i):
        w, s, n = self.w, self.s, self.n
        B = h.shape[0]
        c1 = (h @ w[f"fs_front_w_{i}"].T + w[f"fs_front_b_{i}"]).astype(F32)
        c1 = np.where(c1 >= 0, c1, c1 * F32(s["fs_front_alpha"][i])).astype(F32)                            # (:509)
        gf = _layer_norm(c1, w[f"fs_n1_w_{i}"], w[f"fs_n1_b_{i}"], s["fs_n1_eps"])
        xn = _layer_norm(gf, None, None, s["fs_ln_eps"])
        proj = _silu((xn @ w[f"fs_uv_w_{i}"].T + w[f"fs_uv_b_{i}"]).astype(F32))
        proj = (proj + _dwconv(proj, w[f"fs_uv_c_{i}"], int(s["dw_pad"]))).astype(F32)
        xu, xv = proj[..., :256], proj[..., 256:]
        f1 = np.maximum((xu @ w[f"fs_mem_linear_w_{i}"].T + w[f"fs_mem_linear_b_{i}"]).astype(F32), F32(0.0))
        dense = (f1 @ w[f"fs_mem_project_w_{i}"].T).astype(F32).transpose(0, 2, 1)                          # (B, 256, n)
        depth = int(s["fs_mem_depth"])
        mem = None
        for j in range(depth):                                                                              # dilated dense memory (:521-535)
            wj = w[f"fs_mem_w_{i}_{j}"]                                                                     # (256, j + 1, 39)
            pad, dil = int(s["fs_mem_paddings"][j]), int(s["fs_mem_dilations"][j])
            C, cin, K = wj.shape
            dp = np.zeros((B, dense.shape[1], n + 2 * pad), F32)
            dp[:, :, pad:pad + n] = dense
            mem = np.zeros((B, C, n), F32)
            for c in range(cin):                                                                            # group g reads channels g * cin + c
                src = dp[:, c::cin]
                for k in range(K):
                    mem += src[:, :, k * dil:k * dil + n] * wj[None, :, c, k, None]
            mu = mem.mean(axis=2, keepdims=True, dtype=F32)
            mc = (mem - mu).astype(F32)
            var = (mc * mc).mean(axis=2, keepdims=True, dtype=F32)
            mem = (mc / np.sqrt(var + F32(s["fs_mem_norm_eps"][j])) * w[f"fs_mem_norm_w_{i}_{j}"][None, :, None]
                   + w[f"fs_mem_norm_b_{i}_{j}"][None, :, None]).astype(F32)
            mem = np.where(mem >= 0, mem, mem * w[f"fs_mem_prelu_{i}_{j}"][None, :, None]).astype(F32)
            if j + 1 < depth:
                dense = np.concatenate((mem, dense), axis=1)
        xu = (xu + mem.transpose(0, 2, 1)).astype(F32)
        y = (xv * xu + gf).astype(F32)                                                                      # (:538)
        n2 = _layer_norm(y, w[f"fs_n2_w_{i}"], w[f"fs_n2_b_{i}"], s["fs_n2_eps"])
        return (n2 @ w[f"fs_back_w_{i}"].T + w[f"fs_back_b_{i}"] + h).astype(F32)                          # (:541)

    def process_dynamic(self, pcm: np.ndarray, in_rate: int = 16000, out_rate: int = 16000) -> np.ndarray:
        """The DYNAMIC_AXES graph on (B, L) input-rate samples: scale-factor interpolation to the model rate (floor(L * 16000 / in) samples must be this oracle's
        window), the network, scale-factor interpolation to the output rate."""
        assert self.dynamic
        xin = pcm.astype(F32)
        if in_rate != 16000:
            xin = interpolate_scale(xin, float(16000 / in_rate))
        assert xin.shape[-1] == self.W, (xin.shape, self.W)
        self.process(xin)
        out = self.taps["wav"]
        if out_rate != 16000:
            out = interpolate_scale(out, float(out_rate / 16000))
            self.taps["wav"] = out.copy()
        return np.clip(np.trunc(out.astype(np.float64)), -32768, 32767).astype(np.int16)

    tap_after = ()       # layer counts k whose truncated-network output is kept in taps["mdl_out_after"][k]

    def process(self, pcm: np.ndarray, out_len: int = 0) -> np.ndarray:
        """pcm int16 (B, L): B independent windows -> int16 (B, 2, L_out).  L != W or out_len: the resampling edges (in / out rate != 16 kHz):
        the int16 samples are interpolated to W model-rate samples as floats, the restored waveform to out_len before the int cast."""
        assert pcm.ndim == 2 and pcm.dtype in (np.int16, np.float32)      # a float input tensor is read as it is (audio.float(), :563); taps["wav"] * 2^-15 is the float output (:655)
        w, s, n = self.w, self.s, self.n
        B = pcm.shape[0]
        xin = pcm.astype(F32)
        if pcm.shape[1] != self.W:
            xin = interpolate_linear(xin, self.W)
        x, rms_in = self._norm_audio(xin)
        fr = np.stack([x[:, ENC_S * t:ENC_S * t + ENC_K] for t in range(n)], axis=1)                        # (B, n, 16)
        x_enc = np.maximum(fr @ w["encoder_w"][:, 0, :].T, F32(0.0)).astype(F32).transpose(0, 2, 1)        # (B, 512, n) (:579-582)
        normed = self._window_norm(x_enc, s["front_norm_eps"])
        mask = (np.einsum("oc,bcn->bon", w["front_w"][:, :, 0], normed) + w["front_b"][None, :, None]).astype(F32)
        mdl_in = (mask + w["emb_pos"][..., :n]).astype(F32)                                                 # (:590-591)
        self.taps["mdl_in"] = mdl_in.copy()
        h = mdl_in.transpose(0, 2, 1)
        def tail(hh):
            hh = _layer_norm(hh, w["mm_norm_w"], w["mm_norm_b"], s["mm_norm_eps"]).transpose(0, 2, 1)       # (:544-545)
            hh = self._window_norm(hh, s["intra_norm_eps"])
            return (hh * w["intra_norm_w"][None, :, None] + w["intra_norm_b"][None, :, None] + mdl_in).astype(F32)  # (:549-551)
        self.taps["mdl_out_after"] = {}                  # k -> what "mdl_out" would be if the network ended after k layers (tests print the error growth with depth)
        for i in range(self.layers):
            h = self._flash(h, i)
            h = self._fsmn(h, i)
            if (i + 1) in self.tap_after:
                self.taps["mdl_out_after"][i + 1] = tail(h)
        h = tail(h)
        self.taps["mdl_out"] = h.copy()
        m = np.where(h >= 0, h, h * F32(s["tail_prelu_alpha"])).astype(F32)                                 # (:599)
        gp = (np.einsum("oc,bcn->bon", w["tail_gate_w"][:, :, 0], m) + w["tail_gate_b"][None, :, None]).astype(F32)
        gp = gp.reshape(B * 2, 1024, n)
        mk = (np.tanh(gp[:, :512]) * _sigmoid(gp[:, 512:])).astype(F32)                                     # (:604-605)
        mk = np.maximum(np.einsum("oc,bcn->bon", w["mask_decoder_w"][:, :, 0], mk), F32(0.0)).astype(F32)
        sep = (x_enc[:, None] * mk.reshape(B, 2, 512, n)).reshape(B * 2, 512, n)                            # (:610-611)
        fr = np.einsum("bcn,ck->bnk", sep, w["decoder_w"][:, 0, :]).astype(F32)                             # (B*2, n, 16)
        wav = np.zeros((B * 2, self.W), F32)
        for t in range(n):
            wav[:, ENC_S * t:ENC_S * t + ENC_K] += fr[:, t]
        wav = wav.reshape(B, 2, self.W)
        rms_out = np.sqrt((wav * wav).mean(axis=2, keepdims=True, dtype=F32)).astype(F32)
        with np.errstate(divide="ignore", invalid="ignore"):
            gain = np.where(rms_out > 0, rms_in[:, None, None] / rms_out, F32(0.0)).astype(F32)             # (:622)
        out = (wav * gain).astype(F32)
        if out_len and out_len != self.W:
            out = interpolate_linear(out, out_len)
        self.taps["wav"] = out.copy()
        return np.clip(np.trunc(out.astype(np.float64)), -32768, 32767).astype(np.int16)                    # (:645) int32 cast truncates

    def process_fold(self, pcm: np.ndarray) -> np.ndarray:
        """USE_BATCH_FOLD (:572-576, :655-656): (n_win * W,) -> (2, n_win * W), each window an independent call."""
        out = self.process(np.ascontiguousarray(pcm.reshape(-1, self.W)))                                   # (n_win, 2, W)
        return np.ascontiguousarray(out.transpose(1, 0, 2).reshape(2, -1))
