#!/usr/bin/env python3
"""Error budget of ZipEnhancer's bf16 path (ade_gemm_dtype = "bf16"): SNR in dB of the bf16 engine against the f32 engine of the same build at the encoder taps, after
every sub-module of the first Zipformer layer, at the mask and at the waveform -- and the waveform's distance from the REFERENCE's fixture clips -- with each part of the
network switched back to f32 in turn (ADE_ZIP16_PARTS: bit 0 dense encoder block, bit 1 the eight Zipformer layers, bit 2 the decoder pair's dense block + sub-pixel
convolution; read when the engine is created).   usage: python tools/zip_bf16_budget.py  [> profiles/rNN_zip_bf16_budget.txt]      (GPU box)"""
import os
import sys
import time

sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from audio_denoiser_onnx_amd import zipenhancer as zp
from audio_denoiser_onnx_amd.session import InferenceSession
from audio_denoiser_onnx_amd.weights import pack_blob

F, C = 101, 64


def snr(x, ref):
    e, s = np.asarray(x, np.float64) - np.asarray(ref, np.float64), np.asarray(ref, np.float64)
    return float(10 * np.log10(max((s ** 2).mean(), 1e-30) / max((e ** 2).mean(), 1e-30)))


def run(blob, pcm, L, dtype, parts=None, layer_taps=True, dense_f16=True):
    os.environ["ADE_ZIP_LAYER_TAPS"] = "1" if layer_taps else "0"
    os.environ["ADE_ZIP_DENSE_F16"] = "1" if dense_f16 else "0"
    if parts is None:
        os.environ.pop("ADE_ZIP16_PARTS", None)
    else:
        os.environ["ADE_ZIP16_PARTS"] = str(parts)
    with InferenceSession(weights=blob, metadata=zp.metadata(L, gemm_dtype=dtype)) as s:
        out, wave = s.process(pcm, want_f32=True)
        T = s.frames
        taps = {}
        for k in ["enc_in", "enc0", "enc1", "enc2", "enc3"] + ["l0_%d" % i for i in range(8)]:
            try:
                taps[k] = s.tap(k, pcm.shape[0] * T * F * C).copy()
            except Exception as e:       # a tap the handle does not keep
                taps[k] = None
        try:
            taps["mask"] = s.tap("mask", pcm.shape[0] * T * 201).copy()
        except Exception:
            taps["mask"] = None
        t0 = time.perf_counter()
        for _ in range(3):
            s.process(pcm)
        ms = (time.perf_counter() - t0) / 3 * 1e3
    return out, wave, taps, ms


def main():
    z = np.load(os.path.join("tests", "golden", "zipenhancer_seed0_io.npz"))
    cfg = zp.ZipConfig.from_tensor(z["config"])
    t = zp.fuse_state_dict(zp.synthetic_state_dict(cfg, int(z["seed"])), cfg)
    blob = pack_blob(t)
    L = int(z["length"])
    names = ("wav0", "randn")
    pcm = np.stack([z["in_" + n] for n in names])
    o32, w32, t32, _ = run(blob, pcm, L, "f32")
    sub = ["ff1", "nonlin-attention", "self-attention 1", "convolution 1", "ff2 + bypass", "self-attention 2", "convolution 2", "ff3 + final norm"]
    print("ZipEnhancer bf16 error budget: dB from the f32 engine (taps, both fixture clips together) and from the reference's fixture waveforms (per clip)")
    print("parts = ADE_ZIP16_PARTS: 1 dense encoder block | 2 Zipformer layers | 4 decoder dense block + sub-pixel convolution on bf16 operands (7 = the shipped bf16 path)")
    print("dense = the 16-bit type of the three causal dense blocks (ADE_ZIP_DENSE_F16): IEEE half (the default) or bf16")
    for parts, half in ((7, True), (7, False), (6, True), (5, True), (3, True), (1, True), (2, True), (4, True), (1, False), (4, False)):
        o, w, tp, ms = run(blob, pcm, L, "bf16", parts, dense_f16=half)
        row = {k: (round(snr(v, t32[k]), 1) if v is not None and t32[k] is not None else None) for k, v in tp.items()}
        print(f"\nparts = {parts} ({'+'.join(n for b, n in ((1, 'enc-dense'), (2, 'layers'), (4, 'dec-dense')) if parts & b)} on 16-bit operands), dense = {'half' if half else 'bf16'}")
        print("   encoder taps  enc_in %s | after dual-path block 0..3: %s %s %s %s | mask %s" % (row["enc_in"], row["enc0"], row["enc1"], row["enc2"], row["enc3"], row["mask"]))
        if parts & 2:
            print("   first layer (encoder 0, frequency path), residual stream after: " + " | ".join(f"{n} {row['l0_%d' % i]}" for i, n in enumerate(sub)))
        for i, n in enumerate(names):
            peak = int(np.abs(z["out_" + n]).max())
            dl = int(np.abs(o[i].astype(np.int32) - z["out_" + n].astype(np.int32)).max())
            print(f"   {n}: wave vs f32 engine {snr(w[i], w32[i]):.1f} dB | vs reference wave {snr(w[i], z['wave_' + n]):.1f} dB | vs reference PCM {snr(o[i], z['out_' + n]):.1f} dB | "
                  f"max deviation {dl} LSB at peak {peak} = {20 * np.log10(max(dl, 1) / peak):.1f} dB of peak")
    print("\nf32 engine vs reference: " + " | ".join(f"{n} {snr(w32[i], z['wave_' + n]):.1f} dB" for i, n in enumerate(names)))


if __name__ == "__main__":
    main()
