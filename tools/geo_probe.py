#!/usr/bin/env python3
"""A/B of the fused path's two workgroup geometries on one box: bit-equality of the outputs, ms per step, phase clocks of the first two segments.

    python tools/geo_probe.py [B] [steps]
"""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import torch
from ade_testlib import make_session
from audio_denoiser_onnx_amd.synth import synth_batch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
x = synth_batch(B)
d_in = torch.from_numpy(x).cuda()
outs = {}
stream = torch.cuda.Stream()
torch.cuda.set_stream(stream)
TAPS = ("spec", "e0", "e1", "x_e2", "x_e3", "x_e4", "dp1", "dp2", "x_d0", "x_d1", "x_d2")
configs = [("0", "0"), ("1", "0"), ("2", "0")]      # (geometry, wave_swap)
for rep in range(2):
    for geo, prio in configs:
        s = make_session()
        s.set_option("geometry", geo)
        s.set_option("full_taps", "1")
        s.set_option("wave_swap", prio)
        s.reserve(B)
        d_out = torch.empty((B, s.row_out), dtype=torch.int16, device="cuda")
        for _ in range(30):
            s.run_device(d_in, d_out, stream=stream.cuda_stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            s.run_device(d_in, d_out, stream=stream.cuda_stream)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        pcm, f32 = s.process(x, want_f32=True)
        err = s.tap("xchg_error", 1)[0]
        taps = {}
        for n in TAPS:
            try:
                taps[n] = s.tap(n, B * 63 * 65 * 16).copy()
            except Exception as ex:
                pass
        outs[geo] = (pcm, f32, d_out.cpu().numpy(), taps)
        print(f"rep {rep} geometry {geo} wave_swap {prio}: {ms:.4f} ms/step  ({B * 0.992 / ms * 1e3:.0f} audio-s/s)  xchg_error {err}", flush=True)
        if rep == 1:
            s.profile(3); s.process(x); s.process(x)
            rel = s.tap('phase_clock', 5120).reshape(-1, 10, 64)
            for seg in range(2):
                print(f"  seg {seg} front acc over tiles [stft+feat, conv0, conv1] us:", np.round(rel[seg, 0, 40:43] / 100, 1).tolist(), " mean", round(float(rel[seg, 0, 33] - rel[seg, 0, 32]) / 100, 1),
                      "| back acc [top, deconv3, s-issue+deconv4, mask+irfft+ola, commit+finalize, carry] us:", np.round(rel[seg, 9, 56:62] / 100, 1).tolist())
            c = s.tap('phase_clock_abs', 5120).reshape(-1, 10, 64)
            s.profile(0)
            for seg in range(min(c.shape[0], 4)):
                if (c[seg, 0, 32] < 0 or not c[seg].any()) and seg > 0:
                    continue
                names = ['front', 'enc0', 'enc1', 'enc2', 'dp0', 'dp1', 'dec0', 'dec1', 'dec2', 'back']
                first = [32, 0, 0, 0, 16, 16, 0, 0, 0, 48]
                last = [36, 8, 8, 8, 20, 20, 8, 8, 8, 53]
                print(f"  seg {seg} stage [start, end] in us since seg 0 entered front:",
                      ' '.join(f"{n}[{c[seg, i, first[i]] / 100:.1f},{c[seg, i, last[i]] / 100:.1f}]" for i, n in enumerate(names)), flush=True)
                for i in range(1, 9):
                    ph = c[seg, i, first[i]:last[i] + 1] / 100
                    print(f"    {names[i]} phase durations:", np.round(np.diff(ph), 1).tolist())
ks = sorted(outs)
for k in ks[1:]:
    print(f"geometry {k} == geometry {ks[0]}:  pcm", np.array_equal(outs[ks[0]][0], outs[k][0]), " f32", np.array_equal(outs[ks[0]][1], outs[k][1]), " device-path pcm",
          np.array_equal(outs[ks[0]][2], outs[k][2]), " taps", all(np.array_equal(outs[ks[0]][3][n], outs[k][3][n]) for n in outs[ks[0]][3]))
