#!/usr/bin/env python3
"""Generate tests/golden/gtcrn_sandwich_seed0.npz by RUNNING THE REFERENCE here: GTCRN_CUSTOM.forward (GTCRN/Export_GTCRN.py:636-693) with float audio,
other input / output sample rates and the dynamic-length export (DYNAMIC_AXES = True: frame count from the model-rate waveform, ISTFT trim of
STFT_Process.py:337-341).  Same seeded weights as tests/golden/gtcrn_seed0.adew (tools/make_golden_gtcrn.py::build_reference, seed 0).

    python tools/make_golden_gtcrn_sandwich.py          # needs /root/reference; writes tests/golden/
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)

from ref_import import import_gtcrn_namespace, import_stft_process  # noqa: E402
import make_golden_gtcrn as base  # noqa: E402

GOLD = os.path.join(REPO, "tests", "golden")

# name -> (input length, in rate, out rate, in dtype, out dtype, dynamic)
CASES = {
    "f32_static": (16000, 16000, 16000, "F32", "F32", False),
    "i16_dynamic": (16000, 16000, 16000, "INT16", "INT16", True),
    "dyn_48k_to_8k": (48000, 48000, 8000, "INT16", "INT16", True),
    "dyn_8k_to_48k": (8000, 8000, 48000, "INT16", "INT16", True),
    "dyn_22500_f32_to_44000": (22500, 22500, 44000, "F32", "INT16", True),
    "dyn_24k_to_24k_f32": (30000, 24000, 24000, "INT16", "F32", True),
}


def seeded_module(ns):
    """The seed-0 GTCRN of make_golden_gtcrn.build_reference, in the namespace `ns` (its constants hold this case's overrides)."""
    torch.manual_seed(0)
    g = ns["GTCRN"]().eval()
    gen = torch.Generator().manual_seed(1000)
    with torch.no_grad():
        for m in g.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=gen) * 0.1)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=gen) + 0.5)
                m.weight.copy_(torch.rand(m.weight.shape, generator=gen) + 0.5)
                m.bias.copy_(torch.randn(m.bias.shape, generator=gen) * 0.1)
            elif isinstance(m, torch.nn.PReLU):
                m.weight.copy_(torch.rand(m.weight.shape, generator=gen) * 0.4 + 0.05)
            elif isinstance(m, torch.nn.LayerNorm):
                m.weight.copy_(torch.rand(m.weight.shape, generator=gen) + 0.5)
                m.bias.copy_(torch.randn(m.bias.shape, generator=gen) * 0.1)
    g.prepare_for_export_()
    return g


def signal(rng, n, rate):
    t = np.arange(n) / rate
    x = 0.25 * np.sin(2 * np.pi * 220.0 * t) + 0.12 * np.sin(2 * np.pi * 1370.0 * t + 0.4) + 0.05 * rng.standard_normal(n) + 0.01     # a small DC offset on purpose
    return x.astype(np.float32)


def main():
    STFT_Process = import_stft_process("GTCRN").STFT_Process
    out = {"names": np.array(sorted(CASES))}
    rng = np.random.default_rng(11)
    blob_ref = None
    for name in sorted(CASES):
        L, sri, sro, din, dout, dyn = CASES[name]
        ns = import_gtcrn_namespace(L, {"DYNAMIC_AXES": dyn, "IN_SAMPLE_RATE": sri, "OUT_SAMPLE_RATE": sro, "IN_AUDIO_DTYPE": din, "OUT_AUDIO_DTYPE": dout})
        assert ns["STATIC_SIGNAL_LENGTH"] == (None if dyn else L // 256 + 1)
        g = seeded_module(ns)
        fused = base.fused_tensors(type("C", (), {"gtcrn": g})())
        if blob_ref is None:
            blob_ref = fused
        else:
            assert all(np.array_equal(fused[k], blob_ref[k]) for k in blob_ref)
        stft = STFT_Process("stft_B", ns["NFFT"], ns["WINDOW_LENGTH"], ns["HOP_LENGTH"], 0, ns["WINDOW_TYPE"], True, ns["PAD_MODE"]).eval()
        istft = STFT_Process("istft_B", ns["NFFT"], ns["WINDOW_LENGTH"], ns["HOP_LENGTH"], ns["MAX_SIGNAL_LENGTH"], ns["WINDOW_TYPE"], True, ns["PAD_MODE"],
                             static_norm=not dyn).eval()
        custom = ns["GTCRN_CUSTOM"](g.float(), stft, istft, sri, sro, False, 0).eval()
        x = signal(rng, L, sri)
        xin = x if din == "F32" else np.clip(np.round(x * 32767.0), -32768, 32767).astype(np.int16)
        with torch.no_grad():
            y = custom(torch.from_numpy(xin).reshape(1, 1, L)).numpy().reshape(-1)
        assert y.dtype == (np.float32 if dout == "F32" else np.int16), y.dtype
        out[name + ":in"] = xin
        out[name + ":out"] = y
        out[name + ":cfg"] = np.array([L, sri, sro, int(din == "F32"), int(dout == "F32"), int(dyn)], np.int64)
        print(f"{name:26s} in {xin.dtype} {L} @ {sri} -> out {y.dtype} {y.size} @ {sro}  |out| max {np.abs(y.astype(np.float64)).max():.4g}")
    # the seed-0 blob of the static fixtures must be this very model
    from audio_denoiser_onnx_amd.weights import load_blob
    committed = load_blob(os.path.join(GOLD, "gtcrn_seed0.adew"))
    assert all(np.array_equal(np.asarray(blob_ref[k], np.float32).reshape(-1), np.asarray(committed[k], np.float32).reshape(-1)) for k in committed), "seed-0 blob mismatch"
    np.savez_compressed(os.path.join(GOLD, "gtcrn_sandwich_seed0.npz"), **out)
    print("wrote", os.path.join(GOLD, "gtcrn_sandwich_seed0.npz"))


if __name__ == "__main__":
    main()
