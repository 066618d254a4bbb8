#!/usr/bin/env python3
"""UL-UNAS golden vectors, produced by RUNNING THE REFERENCE (UL-UNAS/Export_UL_UNAS.py: the ULUNAS network :51-739, its
``prepare_for_export_`` folds :697-717 and the ``ULUNAS_CUSTOM`` wrapper :826-913, with the folder's own STFT_Process) here.

The reference ships no checkpoint, so the network is seeded: every parameter and BatchNorm statistic of ``ULUNAS()`` is filled
from torch's generator (seed in the file name) at scales that keep the activations O(1).  The fixture holds the CHECKPOINT-format
state_dict (convolutions and BatchNorms separate, AffinePReLU affine / slope, the two half-width GRUs of every GRNN): what
``audio_denoiser_onnx_amd.ulunas.fold_state_dict`` folds for the engine, so the parity test pins that fold against the reference's
own ``prepare_for_export_`` as well as the forward.

    python tools/make_golden_ulunas.py     # writes tests/golden/ulunas_seed0.npz
"""
import ast
import json
import os
import sys

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)

import make_golden_gtcrn as mg  # noqa: E402
from ref_import import REF_ROOT, _stub_absent_modules, import_stft_process  # noqa: E402

L = 16000


def import_namespace(length: int, fold: bool = False, window_seconds: float = 1.5, extra: dict | None = None) -> dict:
    _stub_absent_modules()
    path = os.path.join(REF_ROOT, "UL-UNAS", "Export_UL_UNAS.py")
    with open(path) as f:
        tree = ast.parse(f.read(), filename=path)
    over = {"INPUT_AUDIO_LENGTH": length, "USE_BATCH_FOLD": fold, "BATCH_WINDOW_SECONDS": window_seconds}
    over.update(extra or {})
    keep = []
    for node in tree.body:
        if isinstance(node, (ast.ClassDef, ast.FunctionDef)):
            if isinstance(node, ast.FunctionDef) and node.name == "_run_inference_demo":
                continue
            keep.append(node)
        elif isinstance(node, ast.Assign):
            names = [t.id for t in node.targets if isinstance(t, ast.Name)]
            if names and all(n.upper() == n for n in names):
                if len(names) == 1 and names[0] in over:
                    node = ast.parse(f"{names[0]} = {over[names[0]]!r}").body[0]
                keep.append(node)
    module = ast.Module(body=keep, type_ignores=[])
    ast.fix_missing_locations(module)
    ns = {"np": np, "torch": torch, "nn": nn, "__name__": "ref_export_ulunas"}
    exec(compile(module, path, "exec"), ns)
    return ns


def seed_network(net, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in net.named_parameters():
            if name.startswith("erb."):
                continue                                             # the ERB filter bank is a fixed table
            if name.endswith("slope_weight"):
                p.copy_(0.05 + 0.4 * torch.rand(p.shape, generator=g))
            elif name.endswith("affine_weight"):
                p.copy_(0.4 * (torch.rand(p.shape, generator=g) - 0.5))
            elif name.endswith("affine_bias"):
                p.copy_(0.2 * (torch.rand(p.shape, generator=g) - 0.5))
            elif ".bn" in name or "_bn." in name or "_ln." in name:
                p.copy_(1.0 + 0.3 * (torch.rand(p.shape, generator=g) - 0.5) if name.endswith("weight") else 0.2 * (torch.rand(p.shape, generator=g) - 0.5))
            else:
                fan = int(np.prod(p.shape[1:])) if p.dim() > 1 else p.shape[0]
                p.copy_((torch.rand(p.shape, generator=g) - 0.5) * (2.0 * 1.8 / max(fan, 1) ** 0.5))
        for name, b in net.named_buffers():
            if name.endswith("running_mean"):
                b.copy_(0.2 * (torch.rand(b.shape, generator=g) - 0.5))
            elif name.endswith("running_var"):
                b.copy_(0.6 + 0.8 * torch.rand(b.shape, generator=g))


def main(seed=0):
    ns = import_namespace(L)
    STFT_Process = import_stft_process("UL-UNAS").STFT_Process
    stft = STFT_Process(model_type="stft_B", n_fft=ns["NFFT"], hop_len=ns["HOP_LENGTH"], win_length=ns["WINDOW_LENGTH"], max_frames=0,
                        window_type=ns["WINDOW_TYPE"], center_pad=True, pad_mode=ns["STFT_PAD_MODE"], input_scale=ns["INV_INT16"]).eval()
    istft = STFT_Process(model_type="istft_B", n_fft=ns["NFFT"], hop_len=ns["HOP_LENGTH"], win_length=ns["WINDOW_LENGTH"],
                         max_frames=ns["MAX_SIGNAL_LENGTH"], window_type=ns["WINDOW_TYPE"], center_pad=True, pad_mode=ns["STFT_PAD_MODE"],
                         output_scale=32767.0, static_norm=True).eval()
    torch.manual_seed(seed)
    net = ns["ULUNAS"]().eval()
    seed_network(net, seed)
    state = {k: v.detach().clone().numpy() for k, v in net.state_dict().items() if v.dtype.is_floating_point and not k.startswith("erb.")}
    net.prepare_for_export_()
    model = ns["ULUNAS_CUSTOM"](net.float(), stft, istft, 16000, 16000, remove_dc_offset=False, use_batch_fold=False, fold_window=0,
                                input_scale_folded=True, output_scale_folded=True).eval()
    rng = np.random.default_rng(1234)
    wav = None
    try:
        from scipy.io import wavfile
        sr, data = wavfile.read(os.path.join(REF_ROOT, "Test_Examples", "denoise", "ul_unas_0174.wav"))
        data = data.reshape(len(data), -1)[:, 0]
        if sr == 16000 and data.dtype == np.int16 and len(data) >= 2 * L:
            wav = np.ascontiguousarray(data[L // 2:L // 2 + L])
    except Exception as e:                                            # noqa: BLE001
        print("test wav not usable:", e)
    rows = [wav if wav is not None else (rng.standard_normal(L) * 2000).astype(np.int16), (rng.standard_normal(L) * 3000).astype(np.int16),
            np.zeros(L, np.int16)]
    taps = {}
    orig = net.forward
    def tapped(power):
        taps["power"] = power.clone()
        m = orig(power)
        taps["mask"] = m.clone()
        return m
    net.forward = tapped
    outs, masks = [], []
    with torch.inference_mode():
        for r in rows:
            outs.append(model(torch.from_numpy(r.reshape(1, 1, -1).copy())).numpy().reshape(-1))
            masks.append(taps["mask"].numpy().reshape(257, -1).copy())
    np.savez_compressed(os.path.join(mg.GOLD, f"ulunas_seed{seed}.npz"), pcm_in=np.stack(rows), pcm_out=np.stack(outs), mask0=masks[0],
                        keys=np.array(list(state)), **{"w:" + k: v for k, v in state.items()})
    print("state tensors", len(state), "floats", sum(v.size for v in state.values()), "out max", [int(np.abs(o).max()) for o in outs],
          "mask mean", float(masks[0].mean()), "mask std", float(masks[0].std()))


def dynamic_fixture(seed=0):
    """DYNAMIC_AXES = True exports (:26, :41-43): any input length, other input / output sample rates (:835-845, :851-868, :890-905), the ISTFT's dynamic trim
    (UL-UNAS/STFT_Process.py:170-177, 317-326) sliced to the caller-rate input length (:851, :888-889).  Built exactly as the export's main does (:936-975):
    the int16 scales are folded into the STFT / ISTFT kernels only where the rate equals the model rate.  tests/golden/ulunas_dynamic_seed{seed}.npz; the network is
    ulunas_seed{seed}.npz's."""
    cases = [("dyn_16000", 7000, 16000, 16000),            # not a multiple of the hop: 28 frames, 7000 samples back
             ("dyn_48000_to_8000", 15000, 48000, 8000),    # down-sample on both edges: 5000 model-rate samples, the full 256 T = 5120 tail is kept (audio_len = 15000 > 5120)
             ("dyn_8000_to_48000", 3500, 8000, 48000),     # up-sample on both edges: 7000 model-rate samples, sliced to audio_len = 3500 of them (:888-889), then x 3
             ("dyn_16000_to_24000", 6144, 16000, 24000)]   # output edge only, whole hops
    z = np.load(os.path.join(mg.GOLD, f"ulunas_seed{seed}.npz"))
    out = {}
    for tag, n, sri, sro in cases:
        ns = import_namespace(n, extra={"DYNAMIC_AXES": True, "IN_SAMPLE_RATE": sri, "OUT_SAMPLE_RATE": sro})
        assert ns["STATIC_SIGNAL_LENGTH"] is None and ns["MAX_SIGNAL_LENGTH"] == 4096
        STFT_Process = import_stft_process("UL-UNAS").STFT_Process
        stft = STFT_Process(model_type="stft_B", n_fft=ns["NFFT"], hop_len=ns["HOP_LENGTH"], win_length=ns["WINDOW_LENGTH"], max_frames=0,
                            window_type=ns["WINDOW_TYPE"], center_pad=True, pad_mode=ns["STFT_PAD_MODE"],
                            input_scale=ns["INV_INT16"] if sri == 16000 else 1.0).eval()
        istft = STFT_Process(model_type="istft_B", n_fft=ns["NFFT"], hop_len=ns["HOP_LENGTH"], win_length=ns["WINDOW_LENGTH"],
                             max_frames=ns["MAX_SIGNAL_LENGTH"], window_type=ns["WINDOW_TYPE"], center_pad=True, pad_mode=ns["STFT_PAD_MODE"],
                             output_scale=32767.0 if sro == 16000 else 1.0, static_norm=False).eval()
        torch.manual_seed(seed)
        net = ns["ULUNAS"]().eval()
        seed_network(net, seed)
        net.prepare_for_export_()
        model = ns["ULUNAS_CUSTOM"](net.float(), stft, istft, sri, sro, remove_dc_offset=False, use_batch_fold=False, fold_window=0,
                                    input_scale_folded=sri == 16000, output_scale_folded=sro == 16000).eval()
        pcm = np.ascontiguousarray(z["pcm_in"][0][1000:1000 + n])
        assert pcm.shape == (n,)
        with torch.inference_mode():
            y = model(torch.from_numpy(pcm.reshape(1, 1, -1).copy())).numpy().reshape(-1)
        out[tag + "_in"], out[tag + "_out"], out[tag + "_rates"] = pcm, y, np.asarray([sri, sro], np.int64)
        print(tag, "in", pcm.shape, "out", y.shape, y.dtype, "max", int(np.abs(y).max()))
    np.savez_compressed(os.path.join(mg.GOLD, f"ulunas_dynamic_seed{seed}.npz"), cases=np.array(json.dumps([c[0] for c in cases])), **out)


def float_io_fixture(seed=0):
    """IN / OUT_AUDIO_DTYPE other than INT16 (:45-46, :858-859, :897-898, :906-912): normalised float tensors in and / or out of the STATIC 16 kHz export, built as the
    export's main does (:936-975: the int16 scales are folded into the STFT / ISTFT kernels only for int16 tensors).  tests/golden/ulunas_float_io_seed{seed}.npz."""
    n = 4096
    z = np.load(os.path.join(mg.GOLD, f"ulunas_seed{seed}.npz"))
    pcm = np.ascontiguousarray(z["pcm_in"][0][2000:2000 + n])
    x = (pcm.astype(np.float32) / np.float32(32768.0)).astype(np.float32)
    out = {"pcm_in": pcm, "x_in": x}
    for tag, din, dout in (("f32_f32", "F32", "F32"), ("f32_i16", "F32", "INT16"), ("i16_f32", "INT16", "F32")):
        ns = import_namespace(n, extra={"IN_AUDIO_DTYPE": din, "OUT_AUDIO_DTYPE": dout})
        STFT_Process = import_stft_process("UL-UNAS").STFT_Process
        stft = STFT_Process(model_type="stft_B", n_fft=ns["NFFT"], hop_len=ns["HOP_LENGTH"], win_length=ns["WINDOW_LENGTH"], max_frames=0,
                            window_type=ns["WINDOW_TYPE"], center_pad=True, pad_mode=ns["STFT_PAD_MODE"], input_scale=ns["INV_INT16"] if din == "INT16" else 1.0).eval()
        istft = STFT_Process(model_type="istft_B", n_fft=ns["NFFT"], hop_len=ns["HOP_LENGTH"], win_length=ns["WINDOW_LENGTH"],
                             max_frames=ns["MAX_SIGNAL_LENGTH"], window_type=ns["WINDOW_TYPE"], center_pad=True, pad_mode=ns["STFT_PAD_MODE"],
                             output_scale=32767.0 if dout == "INT16" else 1.0, static_norm=True).eval()
        torch.manual_seed(seed)
        net = ns["ULUNAS"]().eval()
        seed_network(net, seed)
        net.prepare_for_export_()
        model = ns["ULUNAS_CUSTOM"](net.float(), stft, istft, 16000, 16000, remove_dc_offset=False, use_batch_fold=False, fold_window=0,
                                    input_scale_folded=din == "INT16", output_scale_folded=dout == "INT16").eval()
        src = pcm if din == "INT16" else x
        with torch.inference_mode():
            y = model(torch.from_numpy(src.reshape(1, 1, -1).copy())).numpy().reshape(-1)
        out[tag] = y
        print(tag, y.shape, y.dtype, float(np.abs(y).max()))
    np.savez_compressed(os.path.join(mg.GOLD, f"ulunas_float_io_seed{seed}.npz"), **out)


def fold_fixture(seed=0):
    """USE_BATCH_FOLD = True (:41-44, :866-871, :886-887): BATCH_WINDOW_SECONDS = 0.256 -> W = 4096 (17 frames); INPUT_AUDIO_LENGTH = 10000 ->
    the graph input is 3 whole windows = 12288 samples, folded into the batch; same seeded network as the plain fixture."""
    ns = import_namespace(10000, True, 0.256)
    assert ns["FOLD_WINDOW_LENGTH"] == 4096 and ns["EXPORT_AUDIO_LENGTH"] == 12288 and ns["STATIC_SIGNAL_LENGTH"] == 17
    STFT_Process = import_stft_process("UL-UNAS").STFT_Process
    stft = STFT_Process(model_type="stft_B", n_fft=ns["NFFT"], hop_len=ns["HOP_LENGTH"], win_length=ns["WINDOW_LENGTH"], max_frames=0,
                        window_type=ns["WINDOW_TYPE"], center_pad=True, pad_mode=ns["STFT_PAD_MODE"], input_scale=ns["INV_INT16"]).eval()
    istft = STFT_Process(model_type="istft_B", n_fft=ns["NFFT"], hop_len=ns["HOP_LENGTH"], win_length=ns["WINDOW_LENGTH"],
                         max_frames=ns["MAX_SIGNAL_LENGTH"], window_type=ns["WINDOW_TYPE"], center_pad=True, pad_mode=ns["STFT_PAD_MODE"],
                         output_scale=32767.0, static_norm=True).eval()
    torch.manual_seed(seed)
    net = ns["ULUNAS"]().eval()
    seed_network(net, seed)
    net.prepare_for_export_()
    model = ns["ULUNAS_CUSTOM"](net.float(), stft, istft, 16000, 16000, remove_dc_offset=False, use_batch_fold=True, fold_window=4096,
                                input_scale_folded=True, output_scale_folded=True).eval()
    z = np.load(os.path.join(mg.GOLD, f"ulunas_seed{seed}.npz"))
    pcm = np.ascontiguousarray(np.concatenate((z["pcm_in"][0][:6000], z["pcm_in"][1][:6288])))
    with torch.inference_mode():
        out = model(torch.from_numpy(pcm.reshape(1, 1, -1).copy())).numpy().reshape(-1)
    np.savez_compressed(os.path.join(mg.GOLD, f"ulunas_seed{seed}_fold.npz"), pcm_in=pcm, pcm_out=out, input_audio_length=np.int64(10000),
                        fold_window_length=np.int64(4096), batch_window_seconds=np.float64(0.256))
    print("fold out", out.shape, int(np.abs(out).max()))


if __name__ == "__main__" and "--float-io" in sys.argv:
    float_io_fixture()
    sys.exit(0)

if __name__ == "__main__" and "--dynamic" in sys.argv:
    dynamic_fixture()
    sys.exit(0)

if __name__ == "__main__":
    main()
    fold_fixture()
