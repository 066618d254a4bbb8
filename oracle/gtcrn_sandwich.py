"""CPU restatement (TEST INFRASTRUCTURE: only tests/ may import this) of GTCRN_CUSTOM.forward's input / output sandwich around the network --
float audio, other input / output sample rates and the dynamic-length export -- GTCRN/Export_GTCRN.py:636-693.  The network itself (STFT -> GTCRN ->
ISTFT on a final fp32 waveform at the model rate) is oracle/ade_oracle.c::ade_oracle_process_model_f32.

Pinned against the reference itself: tests/golden/gtcrn_sandwich_seed0.npz (tools/make_golden_gtcrn_sandwich.py runs the reference class on seeded weights).
"""
import math

import numpy as np

F32 = np.float32
INV_INT16 = F32(1.0 / 32768.0)          # :49
MODEL_RATE = 16000.0                     # :30, :623-624


def interpolate_scale(x: np.ndarray, scale_factor: float) -> np.ndarray:
    """F.interpolate(x, scale_factor=f, mode='linear', align_corners=False) over the last axis in fp32 (:639-644, :650-655, :674-688): floor(n f) samples,
    source coordinate (1 / f) (dst + 0.5) - 0.5 clamped at 0, y = (1 - l) x[i0] + l x[min(i0 + 1, n - 1)]."""
    n = x.shape[-1]
    out = int(math.floor(n * scale_factor))
    step = F32(1.0 / scale_factor)
    src = np.maximum(step * (np.arange(out, dtype=F32) + F32(0.5)) - F32(0.5), F32(0.0)).astype(F32)
    i0 = np.minimum(src.astype(np.int64), n - 1)
    i1 = np.minimum(i0 + 1, n - 1)
    l1 = (src - i0.astype(F32)).astype(F32)
    return ((F32(1.0) - l1) * x[..., i0] + l1 * x[..., i1]).astype(F32)


def model_length(n: int, in_rate: int) -> int:
    return n if in_rate == 16000 else int(math.floor(n * (1.0 / (in_rate / MODEL_RATE))))


def forward(network, audio: np.ndarray, in_rate: int, out_rate: int, float_out: bool) -> np.ndarray:
    """`audio`: one call's samples, int16 PCM or normalised float32; `network(x)`: final fp32 model-rate waveform (Lm,) -> normalised waveform (static or dynamic trim).
    Returns int16 PCM, or float32 when float_out."""
    int_in = audio.dtype == np.int16
    x = audio.astype(F32)                                         # :637
    in_scale = in_rate / MODEL_RATE                               # :623
    model_rate_scale = 1.0 / in_scale                             # :625
    if in_scale > 1.0:                                            # :638-644 down-sample BEFORE centring
        x = interpolate_scale(x, model_rate_scale)
    if int_in:                                                    # :645-646
        x = (x * INV_INT16).astype(F32)
    x = (x - F32(np.mean(x, dtype=np.float64))).astype(F32)       # :647 (torch.mean: pairwise fp32; the fp64 mean is its exact value to < 1 ulp of the mean)
    if in_scale < 1.0:                                            # :648-655 up-sample AFTER centring
        x = interpolate_scale(x, model_rate_scale)
    y = np.asarray(network(x), F32)
    out_scale = out_rate / MODEL_RATE                             # :624
    if out_scale < 1.0:                                           # :673-679
        y = interpolate_scale(y, out_scale)
    if not float_out:                                             # :680-681
        y = (y * F32(32767.0)).astype(F32)
    if out_scale > 1.0:                                           # :682-688
        y = interpolate_scale(y, out_scale)
    if float_out:                                                 # :691-693
        return y
    return np.clip(y, F32(-32768.0), F32(32767.0)).astype(np.int16)   # :689-690 (clamp, truncating cast)
