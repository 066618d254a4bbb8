#!/usr/bin/env python3
"""DFSMN golden vectors, produced by RUNNING THE REFERENCE's ``DFSMN.forward`` (DFSMN/Export_DFSMN.py:71-246) here.

The reference obtains its network from modelscope (absent, unpinned) and its mel bank from torchaudio (absent):
  * the network is a *parameter container* only (``linear1.linear``, ``linear2.linear``, ``deepfsmn[i].{linear, project,
    conv1, output_dim, lorder}``, Export_DFSMN.py:150-178) -- a seeded fake tree with those attribute paths and the shapes
    the reference documents (120 -> 256, 9 x [256 -> 256, 256 -> 256, depthwise lorder 20], 256 -> 961) stands in;
  * ``torchaudio.compliance.kaldi.get_mel_banks`` is served by this package's restatement of the published Kaldi
    algorithm (audio_denoiser_onnx_amd/kaldi_mel.py); the bank is stored in the weight blob, so the fixture, the oracle
    and the engine all use the same table (real-bank parity is unpinned until torchaudio is available: DESIGN.md).
Everything else -- the fused fbank|STFT analysis kernel, log-mel, the inlined mask network, the masked ISTFT and the PCM
tail -- is the reference's own code path.

    python tools/make_golden_dfsmn.py      # writes tests/golden/dfsmn_seed0.adew + dfsmn_seed0_io.npz
"""
import ast
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)

import make_golden_gtcrn as mg  # noqa: E402
from ref_import import REF_ROOT, import_stft_process  # noqa: E402
from audio_denoiser_onnx_amd import kaldi_mel  # noqa: E402
from audio_denoiser_onnx_amd.weights import save_blob  # noqa: E402

DEPTH, LORDER, HID, NMEL, NBINS = 9, 20, 256, 120, 961


def import_dfsmn_namespace(input_audio_length: int, overrides: dict = None) -> dict:
    path = os.path.join(REF_ROOT, "DFSMN", "Export_DFSMN.py")
    with open(path) as f:
        tree = ast.parse(f.read(), filename=path)
    keep = []
    for node in tree.body:
        if isinstance(node, ast.ClassDef):
            keep.append(node)
        elif isinstance(node, ast.Assign):
            names = [t.id for t in node.targets if isinstance(t, ast.Name)]
            if names and all(n.upper() == n for n in names):
                if names == ["INPUT_AUDIO_LENGTH"]:
                    node = ast.parse(f"INPUT_AUDIO_LENGTH = {int(input_audio_length)}").body[0]
                elif overrides and len(names) == 1 and names[0] in overrides:
                    node = ast.parse(f"{names[0]} = {overrides[names[0]]!r}").body[0]
                keep.append(node)
        elif isinstance(node, ast.If):      # the HOP_LENGTH > INPUT_AUDIO_LENGTH guard
            keep.append(node)
    module = ast.Module(body=keep, type_ignores=[])
    ast.fix_missing_locations(module)
    ta = types.ModuleType("torchaudio")
    ta.compliance = types.SimpleNamespace(kaldi=types.SimpleNamespace(
        get_mel_banks=lambda *a: (torch.from_numpy(kaldi_mel.get_mel_banks(*a)), None)))
    ns = {"torch": torch, "F": torch.nn.functional, "torchaudio": ta, "__name__": "ref_export_dfsmn"}
    exec(compile(module, path, "exec"), ns)
    return ns


def fake_dfsmn(seed: int):
    torch.manual_seed(seed)
    gen = torch.Generator().manual_seed(2000 + seed)

    def lin(i, o, bias=True, gain=1.0):
        m = nn.Linear(i, o, bias=bias)
        with torch.no_grad():
            m.weight.copy_(torch.randn(o, i, generator=gen) * (gain / np.sqrt(i)))
            if bias:
                m.bias.copy_(torch.randn(o, generator=gen) * 0.1)
        return m

    layers = []
    for _ in range(DEPTH):
        conv = nn.Conv2d(HID, HID, (LORDER, 1), groups=HID, bias=False)
        with torch.no_grad():
            conv.weight.copy_(torch.randn(conv.weight.shape, generator=gen) * 0.08)
        layers.append(types.SimpleNamespace(linear=lin(HID, HID, gain=1.2), project=lin(HID, HID, bias=False, gain=0.7), conv1=conv,
                                            output_dim=HID, lorder=LORDER))
    return types.SimpleNamespace(linear1=types.SimpleNamespace(linear=lin(NMEL, HID, gain=0.15)),
                                 linear2=types.SimpleNamespace(linear=lin(HID, NBINS, gain=0.5)), deepfsmn=layers)


def build(seed: int, length: int, overrides: dict = None):
    ns = import_dfsmn_namespace(length, overrides)
    STFT_Process = import_stft_process("DFSMN").STFT_Process
    stft = STFT_Process("stft_B", ns["NFFT_STFT"], ns["WINDOW_LENGTH"], ns["HOP_LENGTH"], 0, ns["WINDOW_TYPE"], False, "constant").eval()
    istft = STFT_Process("istft_B", ns["NFFT_STFT"], ns["WINDOW_LENGTH"], ns["HOP_LENGTH"], ns["MAX_SIGNAL_LENGTH"],
                         ns["ISTFT_WINDOW_TYPE"], False, "constant", static_norm=not ns["DYNAMIC_AXES"]).eval()      # (Export_DFSMN.py:274)
    net = fake_dfsmn(seed)
    model = ns["DFSMN"](net, stft, istft, ns["NFFT_STFT"], ns["N_MELS"], ns["IN_SAMPLE_RATE"], ns["OUT_SAMPLE_RATE"], ns["USE_BATCH_FOLD"],
                        ns["FOLD_WINDOW_LENGTH"] if ns["USE_BATCH_FOLD"] else 0, ns["STATIC_MODEL_BATCH"]).eval()
    return ns, model


def blob_tensors(model) -> dict:
    """The tensors libade loads, under the reference's buffer names (Export_DFSMN.py:150-178) + the mel bank."""
    out = {"lin1_w": model.lin1_w.squeeze(-1), "lin1_b": model.lin1_b, "lin2_w": model.lin2_w.squeeze(-1), "lin2_b": model.lin2_b,
           "mel_banks": model.mel_banks.squeeze(0)}
    for i in range(model.fsmn_depth):
        out[f"uf_lin_w_{i}"] = getattr(model, f"uf_lin_w_{i}").squeeze(-1)
        out[f"uf_lin_b_{i}"] = getattr(model, f"uf_lin_b_{i}")
        out[f"uf_proj_w_{i}"] = getattr(model, f"uf_proj_w_{i}").squeeze(-1)
        out[f"uf_conv_w_{i}"] = getattr(model, f"uf_conv_w_{i}").squeeze(1)           # (256, lorder), inner residual already folded
    return {k: v.detach().float().numpy().copy() for k, v in out.items()}


def main():
    L = 24000                                   # 0.5 s @ 48 kHz -> 24 frames
    ns, model = build(0, L)
    save_blob(os.path.join(mg.GOLD, "dfsmn_seed0.adew"), blob_tensors(model))
    wav = mg.load_wav_i16(os.path.join(REF_ROOT, "Test_Examples", "denoise", "speech_with_noise_48k.wav"))
    torch.manual_seed(1234)
    ins = {"speech0": wav[48000:48000 + L].copy(), "speech1": wav[120000:120000 + L].copy(),
           "randn": (torch.randn(L) * 2500.0).clamp(-32768, 32767).to(torch.int16).numpy(), "zeros": np.zeros(L, np.int16)}
    out = {"input_audio_length": np.int64(L), "frames": np.int64(ns["STFT_SIGNAL_LENGTH"])}
    taps = {}
    hooks = []
    # taps: log-mel feature (input of lin1) and the mask, via the reference's own F.conv1d calls
    orig_conv1d = torch.nn.functional.conv1d

    def spy(x, w, b=None, *a, **k):
        y = orig_conv1d(x, w, b, *a, **k)
        if w is model.lin1_w:
            taps["logmel"] = x.detach().numpy().copy()
        if w is model.lin2_w:
            taps["mask_pre"] = y.detach().numpy().copy()
        return y

    for name, pcm in ins.items():
        taps.clear()
        ns["F"].conv1d = spy
        try:
            with torch.inference_mode():
                y = model(torch.from_numpy(pcm.reshape(1, 1, -1)))
        finally:
            ns["F"].conv1d = orig_conv1d
        out[f"{name}.pcm_in"] = pcm
        out[f"{name}.pcm_out"] = y.numpy().reshape(-1)
        if name == "speech0":
            out["speech0.logmel"] = taps["logmel"].reshape(NMEL, -1)
            out["speech0.mask"] = 1.0 / (1.0 + np.exp(-taps["mask_pre"].reshape(NBINS, -1)))
        print(name, y.shape, int(np.abs(y.numpy()).max()))
    np.savez_compressed(os.path.join(mg.GOLD, "dfsmn_seed0_io.npz"), **out)

    # USE_BATCH_FOLD (:53-57, :194-198, :231-232): BATCH_WINDOW_SECONDS = 0.2 -> W = 9600 (9 frames); INPUT_AUDIO_LENGTH = 20000 -> 3 windows
    ns, model = build(0, 20000, {"USE_BATCH_FOLD": True, "BATCH_WINDOW_SECONDS": 0.2})
    assert ns["FOLD_WINDOW_LENGTH"] == 9600 and ns["EXPORT_AUDIO_LENGTH"] == 28800 and ns["STFT_SIGNAL_LENGTH"] == 9
    pcm = wav[60000:60000 + 28800].copy()
    with torch.inference_mode():
        y = model(torch.from_numpy(pcm.reshape(1, 1, -1))).numpy().reshape(-1)
    # resampling edges (:186-193, :233-240): 16 kHz in -> 48 kHz model (8000 -> 24000 samples, 24 frames) -> 24 kHz out (12000)
    ns, model = build(0, 8000, {"IN_SAMPLE_RATE": 16000, "OUT_SAMPLE_RATE": 24000})
    assert ns["MODEL_AUDIO_LENGTH"] == 24000 and ns["OUTPUT_AUDIO_LENGTH"] == 12000
    pcm2 = np.ascontiguousarray(wav[48000:48000 + 24000:3])
    with torch.inference_mode():
        y2 = model(torch.from_numpy(pcm2.reshape(1, 1, -1))).numpy().reshape(-1)
    np.savez_compressed(os.path.join(mg.GOLD, "dfsmn_seed0_edges.npz"), fold_in=pcm, fold_out=y, fold_input_audio_length=np.int64(20000),
                        fold_window_length=np.int64(9600), fold_batch_window_seconds=np.float64(0.2), rs_in=pcm2, rs_out=y2, rs_in_rate=np.int64(16000),
                        rs_out_rate=np.int64(24000))
    print("fold", y.shape, int(np.abs(y).max()), "resample", y2.shape, int(np.abs(y2).max()))


def dynamic_fixture():
    """DYNAMIC_AXES = True (:28, :48-49, :68, :186-187, :236-237, :274): the input length is free, the edges interpolate by SCALE FACTOR (floor(n * factor) samples,
    source step 1 / factor) and the ISTFT builds its overlap-add denominator from the actual frame count (2048-frame bound).  Two runs of the reference's forward:
    48 kHz throughout on a length that is not a whole number of hops, and 22.05 kHz -> 48 kHz -> 16 kHz."""
    wav = mg.load_wav_i16(os.path.join(REF_ROOT, "Test_Examples", "denoise", "speech_with_noise_48k.wav"))
    out = {}
    ns, model = build(0, 10000, {"DYNAMIC_AXES": True})
    assert ns["MAX_SIGNAL_LENGTH"] == 2048 and not hasattr(model.istft_model, "static_win_sum")
    x = wav[60000:60000 + 10000].copy()
    with torch.inference_mode():
        out["eq_in"], out["eq_out"] = x, model(torch.from_numpy(x.reshape(1, 1, -1))).numpy().reshape(-1)
    ns, model = build(0, 6000, {"DYNAMIC_AXES": True, "IN_SAMPLE_RATE": 22050, "OUT_SAMPLE_RATE": 16000})
    x = np.ascontiguousarray(wav[48000:48000 + 12000:2])
    with torch.inference_mode():
        out["rs_in"], out["rs_out"] = x, model(torch.from_numpy(x.reshape(1, 1, -1))).numpy().reshape(-1)
    out["rs_in_rate"], out["rs_out_rate"] = np.int64(22050), np.int64(16000)
    np.savez_compressed(os.path.join(mg.GOLD, "dfsmn_dynamic_seed0.npz"), **out)
    print("dynamic: equal rates", out["eq_in"].shape, "->", out["eq_out"].shape, "; 22.05k -> 16k", out["rs_in"].shape, "->", out["rs_out"].shape)


def float_io_fixture():
    """IN / OUT_AUDIO_DTYPE other than INT16 (:43-44): a float input skips the * INV_INT16 (:178-182), a float output the * 32768 and the clamp (:241-247).
    tests/golden/dfsmn_float_io_seed0.npz; the network is dfsmn_seed0.adew's."""
    L = 24000
    wav = mg.load_wav_i16(os.path.join(REF_ROOT, "Test_Examples", "denoise", "speech_with_noise_48k.wav"))
    pcm = wav[48000:48000 + L].copy()
    x = (pcm.astype(np.float32) / np.float32(32768.0)).astype(np.float32)
    out = {"pcm_in": pcm, "x_in": x}
    for tag, din, dout in (("f32_f32", "F32", "F32"), ("f32_i16", "F32", "INT16"), ("i16_f32", "INT16", "F32")):
        ns, model = build(0, L, {"IN_AUDIO_DTYPE": din, "OUT_AUDIO_DTYPE": dout})
        src = pcm if din == "INT16" else x
        with torch.inference_mode():
            y = model(torch.from_numpy(src.reshape(1, 1, -1).copy())).numpy().reshape(-1)
        out[tag] = y
        print(tag, y.shape, y.dtype, float(np.abs(y).max()))
    np.savez_compressed(os.path.join(mg.GOLD, "dfsmn_float_io_seed0.npz"), **out)


if __name__ == "__main__" and "--dynamic" in sys.argv:
    dynamic_fixture()
    sys.exit(0)

if __name__ == "__main__" and "--float-io" in sys.argv:
    float_io_fixture()
    sys.exit(0)

if __name__ == "__main__":
    main()
