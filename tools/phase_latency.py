#!/usr/bin/env python3
"""Per-phase wall time of the single-launch chunk kernel (its phase-clock build), segments 0 and 1 of chunk 0, at a given batch.   python tools/phase_latency.py B [geometry]

B = 1 shows each phase's LATENCY (one chunk alone: four workgroups on one CU, nothing to contend with); B = 256 the same phases with every CU full.  Ticks are 10 ns."""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from ade_testlib import make_session
from audio_denoiser_onnx_amd.synth import synth_batch

B = int(sys.argv[1]); geo = sys.argv[2] if len(sys.argv) > 2 else "2"
x = synth_batch(B)
s = make_session()
s.set_option("geometry", geo)
s.reserve(B)
for _ in range(5):
    s.process(x)
s.profile(3); s.process(x); s.process(x)
rel = s.tap('phase_clock', 5120).reshape(-1, 10, 64)
c = s.tap('phase_clock_abs', 5120).reshape(-1, 10, 64)
s.profile(0)
names = ['front', 'enc0', 'enc1', 'enc2', 'dp0', 'dp1', 'dec0', 'dec1', 'dec2', 'back']
first = [32, 0, 0, 0, 16, 16, 0, 0, 0, 48]
last = [36, 8, 8, 8, 20, 20, 8, 8, 8, 53]
gt_ph = ["x1+pw1", "dw+pw2", "h1->lds", "energy+gi", "-", "tra+bypass", "gate-linear", "gate+store"]
dp_ph = ["intra-gru", "linear+ln", "inter-gru", "linear+ln"]
for seg in range(min(c.shape[0], 4)):
    if not c[seg].any():
        continue
    print(f"B = {B} geometry {geo} segment {seg}: stage [start, end] us:", ' '.join(f"{n}[{c[seg, i, first[i]] / 100:.1f},{c[seg, i, last[i]] / 100:.1f}]" for i, n in enumerate(names)))
    print(f"   front  mean {float(rel[seg, 0, 33] - rel[seg, 0, 32]) / 100:.1f} | summed over tiles: fft+erb {rel[seg, 0, 40] / 100:.1f} conv0 {rel[seg, 0, 41] / 100:.1f} conv1 {rel[seg, 0, 42] / 100:.1f}")
    for i in range(1, 9):
        ph = np.diff(c[seg, i, first[i]:last[i] + 1] / 100)
        lab = gt_ph if len(ph) == 8 else dp_ph
        print(f"   {names[i]:5s} " + "  ".join(f"{l} {v:.1f}" for l, v in zip(lab, ph)))
    print(f"   back   summed over tiles: top {rel[seg, 9, 56] / 100:.1f} deconv3 {rel[seg, 9, 57] / 100:.1f} issue+deconv4 {rel[seg, 9, 58] / 100:.1f} mask+irfft+ola {rel[seg, 9, 59] / 100:.1f} "
          f"commit+finalize {rel[seg, 9, 60] / 100:.1f} carry {rel[seg, 9, 61] / 100:.1f}")
