#!/usr/bin/env python3
"""Generate tests/golden/gtcrn_*.npz + weight blobs by RUNNING THE REFERENCE here.

Runs only in the build container (needs /root/reference); the outputs are
committed as fixtures and are the pin for oracle/ and for the HIP path.
Every vector is produced by a B=1 call of the reference's own
``GTCRN_CUSTOM.forward`` (GTCRN/Export_GTCRN.py:636-693) on seeded weights —
the reference ships no checkpoint — so "parity" below
means seeded-weights parity.

    python tools/make_golden_gtcrn.py            # writes tests/golden/
"""
from __future__ import annotations

import os
import sys
import wave

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)

from ref_import import REF_ROOT, import_gtcrn_namespace, import_stft_process  # noqa: E402
from audio_denoiser_onnx_amd.weights import save_blob  # noqa: E402

GOLD = os.path.join(REPO, "tests", "golden")
L_IN = 16000


def build_reference(seed: int, length: int = L_IN):
    """Seeded reference module in its export-ready (BN-folded) state."""
    ns = import_gtcrn_namespace(length)
    stft_mod = import_stft_process("GTCRN")
    STFT_Process = stft_mod.STFT_Process
    torch.manual_seed(seed)
    g = ns["GTCRN"]().eval()
    gen = torch.Generator().manual_seed(1000 + seed)
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
    unfused = {k: v.detach().clone().numpy() for k, v in g.state_dict().items()}
    g.prepare_for_export_()
    frames = length // ns["HOP_LENGTH"] + 1
    stft = STFT_Process("stft_B", ns["NFFT"], ns["WINDOW_LENGTH"], ns["HOP_LENGTH"], 0,
                        ns["WINDOW_TYPE"], True, ns["PAD_MODE"]).eval()
    istft = STFT_Process("istft_B", ns["NFFT"], ns["WINDOW_LENGTH"], ns["HOP_LENGTH"], frames,
                         ns["WINDOW_TYPE"], True, ns["PAD_MODE"], static_norm=True).eval()
    custom = ns["GTCRN_CUSTOM"](g.float(), stft, istft, 16000, 16000, False, 0).eval()
    return ns, custom, unfused


def fused_tensors(custom) -> dict:
    """Post-fold tensors under the reference's state_dict names (+ the two ERB buffers)."""
    g = custom.gtcrn
    out = {k: v.detach().numpy() for k, v in g.state_dict().items()}
    out["erb.erb_weight_t"] = g.erb.erb_weight_t.detach().numpy()      # (192, 64)  Export_GTCRN.py:111
    out["erb.ierb_weight_t"] = g.erb.ierb_weight_t.detach().numpy()    # (64, 192)  Export_GTCRN.py:112
    return out


def run_with_taps(custom, pcm: np.ndarray, want_taps: bool):
    taps = {}
    hooks = []
    g = custom.gtcrn

    def rec(name):
        def _h(_m, _i, o):
            taps[name] = (o[0] if isinstance(o, tuple) else o).detach().numpy().copy()
        return _h

    if want_taps:
        for i, m in enumerate(g.encoder.en_convs):
            hooks.append(m.register_forward_hook(rec(f"e{i}")))
        for i, m in enumerate(g.decoder.de_convs):
            hooks.append(m.register_forward_hook(rec(f"d{i}")))
        blk = g.encoder.en_convs[2]
        hooks.append(blk.point_act.register_forward_hook(rec("e2.pw1")))
        hooks.append(blk.depth_act.register_forward_hook(rec("e2.dw")))
        hooks.append(blk.point_conv2.register_forward_hook(rec("e2.pw2")))
        hooks.append(blk.tra.att_gru.register_forward_hook(rec("e2.tra_gru")))
        hooks.append(blk.tra.register_forward_hook(rec("e2.tra")))
        dblk = g.decoder.de_convs[0]
        hooks.append(dblk.point_act.register_forward_hook(rec("d0.pw1")))
        hooks.append(dblk.depth_act.register_forward_hook(rec("d0.dw")))
        hooks.append(dblk.point_conv2.register_forward_hook(rec("d0.pw2")))
        for n, dp in (("dp1", g.dpgrnn1), ("dp2", g.dpgrnn2)):
            hooks.append(dp.register_forward_hook(rec(n)))
            hooks.append(dp.intra_rnn.register_forward_hook(rec(n + ".intra_rnn")))
            hooks.append(dp.intra_ln.register_forward_hook(rec(n + ".intra_ln")))
            hooks.append(dp.inter_rnn.register_forward_hook(rec(n + ".inter_rnn")))
            hooks.append(dp.inter_ln.register_forward_hook(rec(n + ".inter_ln")))
        hooks.append(g.sfe.register_forward_hook(
            lambda _m, i, _o: taps.__setitem__("feat_erb", i[0].detach().numpy().copy())))

    stft_fn = custom.stft_model._stft_B_packed_forward
    istft_fn = custom.istft_model._istft_B_packed_forward
    fwd_packed = g.forward_packed

    def stft_wrap(x):
        taps["audio_f32"] = x.detach().numpy().copy()
        y = stft_fn(x)
        taps["spec"] = y.detach().numpy().copy()
        return y

    def fp_wrap(s):
        y = fwd_packed(s)
        taps["spec_enh"] = y.detach().numpy().copy()
        return y

    def istft_wrap(x):
        y = istft_fn(x)
        taps["wave_f32"] = y.detach().numpy().copy()
        return y

    custom.stft_model._stft_B_packed_forward = stft_wrap
    custom.istft_model._istft_B_packed_forward = istft_wrap
    g.forward_packed = fp_wrap
    try:
        with torch.inference_mode():
            out = custom(torch.from_numpy(pcm.reshape(1, 1, -1)))
    finally:
        custom.stft_model._stft_B_packed_forward = stft_fn
        custom.istft_model._istft_B_packed_forward = istft_fn
        g.forward_packed = fwd_packed
        for h in hooks:
            h.remove()
    taps["pcm_out"] = out.numpy().reshape(-1).copy()
    return taps


def load_wav_i16(path):
    with wave.open(path, "rb") as w:
        assert w.getsampwidth() == 2 and w.getnchannels() == 1
        return np.frombuffer(w.readframes(w.getnframes()), dtype="<i2").copy()


def make_inputs():
    wav = load_wav_i16(os.path.join(REF_ROOT, "Test_Examples", "denoise", "gtcrn_mix.wav"))
    ins = {}
    for i in range(3):   # slices exactly as Inference_GTCRN_ONNX.py:287-330 cuts them (stride 15872)
        ins[f"wav{i}"] = wav[i * 15872:i * 15872 + L_IN].copy()
    torch.manual_seed(1234)   # the seed the reference's own STFT self-test uses (STFT_Process.py:465-468)
    ins["randn"] = (torch.randn(L_IN) * 3000.0).clamp(-32768, 32767).to(torch.int16).numpy()
    ins["zeros"] = np.zeros(L_IN, np.int16)
    sq = np.where((np.arange(L_IN) // 40) % 2 == 0, 32767, -32767).astype(np.int16)
    ins["square_fs"] = sq
    for pos in (0, 8000, L_IN - 1):
        imp = np.zeros(L_IN, np.int16)
        imp[pos] = 20000
        ins[f"impulse{pos}"] = imp
    ins["dc_min"] = np.full(L_IN, -32768, np.int16)
    return ins


def main():
    os.makedirs(GOLD, exist_ok=True)
    inputs = make_inputs()
    np.savez_compressed(os.path.join(GOLD, "gtcrn_inputs.npz"), **inputs)
    TAP_KEEP_F16 = ()   # every tap is kept in float32
    for seed in (0, 1, 2):
        ns, custom, unfused = build_reference(seed)
        fused = fused_tensors(custom)
        save_blob(os.path.join(GOLD, f"gtcrn_seed{seed}.adew"), fused)
        if seed == 0:
            # pre-fold checkpoint-format state_dict: pins the importer's own BN fold (tools/import_gtcrn_checkpoint.py)
            np.savez_compressed(os.path.join(GOLD, "gtcrn_seed0_unfused_state_dict.npz"), **unfused)
        outs = {}
        for name, pcm in inputs.items():
            if seed != 0 and name not in ("wav0", "randn", "square_fs"):
                continue
            want = seed == 0 and name == "wav0"
            taps = run_with_taps(custom, pcm, want)
            outs[f"{name}.wave_f32"] = taps["wave_f32"].reshape(-1)
            outs[f"{name}.pcm_out"] = taps["pcm_out"]
            if want:
                np.savez_compressed(os.path.join(GOLD, "gtcrn_seed0_wav0_taps.npz"),
                                    **{k: v.astype(np.float32) if v.dtype != np.int16 else v
                                       for k, v in taps.items() if k not in TAP_KEEP_F16})
        np.savez_compressed(os.path.join(GOLD, f"gtcrn_seed{seed}_outputs.npz"), **outs)
        print(f"seed {seed}: {len(fused)} tensors, {sum(v.size for v in fused.values())} floats, "
              f"{len(outs) // 2} input cases")
    # a 2 s case (T=126, the reference's default INPUT_AUDIO_LENGTH=32000) for the length-generic path
    ns, custom, _ = build_reference(0, 32000)
    wav = load_wav_i16(os.path.join(REF_ROOT, "Test_Examples", "denoise", "gtcrn_mix.wav"))
    pcm = wav[20000:52000].copy()
    taps = run_with_taps(custom, pcm, False)
    np.savez_compressed(os.path.join(GOLD, "gtcrn_seed0_len32000.npz"), pcm_in=pcm,
                        wave_f32=taps["wave_f32"].reshape(-1), pcm_out=taps["pcm_out"])
    print("done ->", GOLD)


if __name__ == "__main__":
    main()
