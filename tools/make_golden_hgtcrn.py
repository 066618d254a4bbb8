#!/usr/bin/env python3
"""H-GTCRN golden vectors, produced by RUNNING THE REFERENCE (H-GTCRN/Export_H_GTCRN.py: the GTCRN_IVA network :428-494, the WPE :581-757 and
AuxIVA :760-900 front-ends, the ``H_GTCRN_CUSTOM`` wrapper :903-1063, with the folder's own STFT_Process) in this container.

The reference ships no checkpoint, so the network is seeded: every parameter and BatchNorm statistic of ``GTCRN_IVA()`` is filled from
torch's generator (seed in the file name).  The fixture holds the CHECKPOINT-format state_dict (convolutions and BatchNorms separate):
what ``audio_denoiser_onnx_amd.hgtcrn.fold_state_dict`` folds for the engine, so the parity test pins that fold against the reference's
own ``fuse_bn_`` as well as the forward.

    python tools/make_golden_hgtcrn.py     # writes tests/golden/hgtcrn_seed0.npz and hgtcrn_seed0_fold.npz
"""
import ast
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

L = 16384                                                            # a whole number of hops (Export_H_GTCRN.py:33) -> 65 frames


def import_namespace(length: int, fold: bool = False, window_seconds: float = 1.5, in_rate: int = 16000, out_rate: int = 16000, extra: dict | None = None) -> dict:
    _stub_absent_modules()
    path = os.path.join(REF_ROOT, "H-GTCRN", "Export_H_GTCRN.py")
    with open(path) as f:
        tree = ast.parse(f.read(), filename=path)
    over = {"INPUT_AUDIO_LENGTH": length, "USE_BATCH_FOLD": fold, "BATCH_WINDOW_SECONDS": window_seconds, "IN_SAMPLE_RATE": in_rate,
            "OUT_SAMPLE_RATE": out_rate}
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
    ns = {"np": np, "torch": torch, "nn": nn, "__name__": "ref_export_hgtcrn"}
    exec(compile(module, path, "exec"), ns)
    return ns


def seed_network(net, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in net.named_parameters():
            if name.startswith("erb."):
                continue                                             # the ERB filter bank is a fixed table
            if ".bn." in name:
                p.copy_(1.0 + 0.3 * (torch.rand(p.shape, generator=g) - 0.5) if name.endswith("weight") else 0.2 * (torch.rand(p.shape, generator=g) - 0.5))
            elif ".act." in name:
                p.copy_(0.05 + 0.4 * torch.rand(p.shape, generator=g))
            elif "_ln." in name:
                p.copy_(1.0 + 0.3 * (torch.rand(p.shape, generator=g) - 0.5) if name.endswith("weight") else 0.2 * (torch.rand(p.shape, generator=g) - 0.5))
            else:
                fan = int(np.prod(p.shape[1:])) if p.dim() > 1 else p.shape[0]
                p.copy_((torch.rand(p.shape, generator=g) - 0.5) * (2.0 * 1.6 / max(fan, 1) ** 0.5))
        for name, b in net.named_buffers():
            if name.endswith("running_mean"):
                b.copy_(0.2 * (torch.rand(b.shape, generator=g) - 0.5))
            elif name.endswith("running_var"):
                b.copy_(0.6 + 0.8 * torch.rand(b.shape, generator=g))


def build(ns, seed, fold_window=0, dynamic=False):
    STFT_Process = import_stft_process("H-GTCRN").STFT_Process
    stft = STFT_Process(model_type="stft_B", n_fft=ns["NFFT"], hop_len=ns["HOP_LENGTH"], win_length=ns["WINDOW_LENGTH"], max_frames=0,
                        window_type=ns["WINDOW_TYPE"], center_pad=True, pad_mode=ns["PAD_MODE"], input_scale=1.0).eval()
    istft = STFT_Process(model_type="istft_B", n_fft=ns["NFFT"], hop_len=ns["HOP_LENGTH"], win_length=ns["WINDOW_LENGTH"],
                         max_frames=ns["MAX_SIGNAL_LENGTH"], window_type=ns["WINDOW_TYPE"], center_pad=True, pad_mode=ns["PAD_MODE"],
                         output_scale=1.0, static_cola=not dynamic).eval()                                                  # (:1097)
    frames, fb = ns["MAX_SIGNAL_LENGTH"], ns["FRONTEND_BATCH"]
    static_frames = None if dynamic else frames                                                                                # (:1075)
    wpe = ns["OnnxFriendlyWPE"](n_channels=2, rt60=ns["WPE_RT60"], hop_length=ns["HOP_LENGTH"], delay=ns["WPE_DELAY"], sample_rate=16000,
                                num_iter=ns["WPE_ITER"], ns_iter=ns["CG_SOLVE_ITER"], n_freq_bins=257, max_frames=frames, batch_size=fb,
                                dynamic_frames=dynamic).eval()                                                                 # (:1110)
    iva = ns["OnnxFriendlyAuxIVA"](n_iter=ns["IVA_ITER"], n_channels=2, batch_size=fb, n_frames=static_frames).eval()
    torch.manual_seed(seed)
    net = ns["GTCRN_IVA"](batch_size=fb, n_frames=static_frames).eval()
    seed_network(net, seed)
    state = {k: v.detach().clone().numpy() for k, v in net.state_dict().items()
             if v.dtype.is_floating_point and not k.endswith("_weight_t") and not k.endswith("_h0") and "zero" not in k}
    net.fuse_bn_()
    model = ns["H_GTCRN_CUSTOM"](net, stft, istft, wpe, iva, n_fft=512, in_sample_rate=ns["IN_SAMPLE_RATE"], out_sample_rate=ns["OUT_SAMPLE_RATE"],
                                 use_batch_fold=bool(fold_window), fold_window=fold_window, model_audio_length=ns["MODEL_AUDIO_LENGTH"],
                                 n_frames=static_frames, frontend_batch=fb, fold_input_pcm_scale=False, fold_output_pcm_scale=False).eval()
    return model, net, wpe, iva, state


def stereo_rows(n_rows, length):
    """Row 0: the reference's own stereo example; the others: a 'room' made here (two sources, each reaching the two microphones through a
    different short decaying filter, plus sensor noise) so that WPE and AuxIVA have something to do."""
    rng = np.random.default_rng(1234)
    rows = []
    try:
        from scipy.io import wavfile
        sr, data = wavfile.read(os.path.join(REF_ROOT, "Test_Examples", "denoise", "h_gtcrn_noisy.wav"))
        if sr == 16000 and data.dtype == np.int16 and data.ndim == 2 and data.shape[1] == 2 and len(data) >= 8000 + length:
            rows.append(np.ascontiguousarray(data[8000:8000 + length].T))
    except Exception as e:                                            # noqa: BLE001
        print("test wav not usable:", e)
    while len(rows) < n_rows:
        t = np.arange(length) / 16000.0
        f0 = rng.uniform(100, 240)
        voice = sum(np.sin(2 * np.pi * k * f0 * t + rng.uniform(0, 6.28)) / k for k in range(1, 16)) * (0.5 - 0.5 * np.cos(2 * np.pi * 3.0 * t))
        other = rng.standard_normal(length)
        mics = []
        for m in range(2):
            h1 = rng.standard_normal(1200) * np.exp(-np.arange(1200) / 300.0); h1[0] = 3.0
            h2 = rng.standard_normal(1200) * np.exp(-np.arange(1200) / 300.0); h2[0] = 3.0
            mics.append(np.convolve(voice, h1)[:length] * 600 + np.convolve(other, h2)[:length] * 150 + rng.standard_normal(length) * 40)
        rows.append(np.clip(np.round(np.stack(mics)), -32768, 32767).astype(np.int16))
    return rows


def main(seed=0):
    ns = import_namespace(L)
    assert ns["MAX_SIGNAL_LENGTH"] == 65 and ns["FRONTEND_BATCH"] == 1
    model, net, wpe, iva, state = build(ns, seed)
    rows = stereo_rows(3, L) + [np.zeros((2, L), np.int16)]
    taps = {}
    import copy
    wpe64 = copy.deepcopy(wpe).double()                            # the reference's OWN WPE module in float64: its distance from the
    spread = []                                                     # fp32 run, per bin, is what decides which bins are well-conditioned

    def wrap(mod, name):
        orig = mod.forward
        def f(*a):
            y = orig(*a)
            taps[name] = [t.clone() for t in y] if isinstance(y, tuple) else y.clone()
            if name == "net":
                taps["features"] = a[0].clone()
            if name == "wpe":
                taps["wpe_in"] = [t.clone() for t in a]
            return y
        mod.forward = f
    wrap(wpe, "wpe"); wrap(iva, "iva"); wrap(net, "net")
    outs, saved, wr, wi = [], {}, [], []
    with torch.inference_mode():
        for i, r in enumerate(rows):
            outs.append(model(torch.from_numpy(r.reshape(1, 2, -1).copy())).numpy().reshape(-1))
            wr.append(taps["wpe"][0].numpy()[0].copy()); wi.append(taps["wpe"][1].numpy()[0].copy())
            y64 = wpe64(*[t.double() for t in taps["wpe_in"]])
            d = torch.maximum((y64[0] - taps["wpe"][0].double()).abs(), (y64[1] - taps["wpe"][1].double()).abs())[0]     # (2, F, T) or (F, ...)
            spread.append(torch.nan_to_num(d, nan=float("inf")).reshape(2, 257, -1).amax(dim=(0, 2)).numpy())
            if i == 1:
                saved = {"tap_iva_r": taps["iva"][0].numpy()[0],
                         "tap_iva_i": taps["iva"][1].numpy()[0], "tap_features": taps["features"].numpy()[0],
                         "tap_s_r": taps["net"][0].numpy()[0], "tap_s_i": taps["net"][1].numpy()[0]}
    np.savez_compressed(os.path.join(mg.GOLD, f"hgtcrn_seed{seed}.npz"), pcm_in=np.stack(rows), pcm_out=np.stack(outs), wpe_r=np.stack(wr), wpe_i=np.stack(wi),
                        wpe_ref_spread=np.stack(spread).astype(np.float32), keys=np.array(list(state)), **saved, **{"w:" + k: v for k, v in state.items()})
    print("reference WPE fp32-vs-fp64: bins with spread >= 1e-4 per row:", [int((sp >= 1e-4).sum()) for sp in spread])
    print("state tensors", len(state), "floats", sum(v.size for v in state.values()), "out max", [int(np.abs(o).max()) for o in outs],
          "in max", [int(np.abs(r).max()) for r in rows])


def fold_fixture(seed=0):
    """USE_BATCH_FOLD = True (:43-47, :953-961, :1014-1016): BATCH_WINDOW_SECONDS = 0.512 -> W = 8192 (33 frames); INPUT_AUDIO_LENGTH = 20000
    -> the graph input is 3 whole windows = 24576 samples; WPE / AuxIVA run per window.  Same seeded network as the plain fixture."""
    ns = import_namespace(20000, True, 0.512)
    assert ns["FOLD_WINDOW_LENGTH"] == 8192 and ns["EXPORT_AUDIO_LENGTH"] == 24576 and ns["MAX_SIGNAL_LENGTH"] == 33 and ns["FRONTEND_BATCH"] == 3
    model, _, wpe, *_ = build(ns, seed, fold_window=8192)
    taps = {}
    orig = wpe.forward
    def tapped(*a):
        y = orig(*a)
        taps["wpe"] = [t.clone().numpy() for t in y]
        return y
    wpe.forward = tapped
    z = np.load(os.path.join(mg.GOLD, f"hgtcrn_seed{seed}.npz"))
    pcm = np.ascontiguousarray(np.concatenate((z["pcm_in"][0][:, :12288], z["pcm_in"][1][:, :12288]), axis=1))
    with torch.inference_mode():
        out = model(torch.from_numpy(pcm.reshape(1, 2, -1).copy())).numpy().reshape(-1)
    np.savez_compressed(os.path.join(mg.GOLD, f"hgtcrn_seed{seed}_fold.npz"), pcm_in=pcm, pcm_out=out, wpe_r=taps["wpe"][0], wpe_i=taps["wpe"][1], input_audio_length=np.int64(20000),
                        fold_window_length=np.int64(8192), batch_window_seconds=np.float64(0.512))
    print("fold out", out.shape, int(np.abs(out).max()))


def resample_fixture(seed=0):
    """The resampling edges (:953-970, :1036-1052) with the same seeded network: 24 kHz in / 8 kHz out (both interpolations BEFORE the scale /
    centring and the PCM scale) and 8 kHz in / 48 kHz out (both AFTER); model length 8192 samples = 33 frames in either case."""
    z = np.load(os.path.join(mg.GOLD, f"hgtcrn_seed{seed}.npz"))
    out = {}
    for tag, in_rate, out_rate, length in (("down", 24000, 8000, 12288), ("up", 8000, 48000, 4096)):
        ns = import_namespace(length, in_rate=in_rate, out_rate=out_rate)
        assert ns["MODEL_AUDIO_LENGTH"] == 8192 and ns["MAX_SIGNAL_LENGTH"] == 33
        model, _, wpe, *_ = build(ns, seed)
        taps = {}
        orig = wpe.forward
        def tapped(*a, orig=orig, taps=taps):
            y = orig(*a)
            taps["wpe"] = [t.clone().numpy() for t in y]
            return y
        wpe.forward = tapped
        pcm = np.ascontiguousarray(z["pcm_in"][1 if tag == "down" else 2][:, 1000:1000 + length])
        with torch.inference_mode():
            y = model(torch.from_numpy(pcm.reshape(1, 2, -1).copy())).numpy().reshape(-1)
        out.update({f"{tag}_pcm_in": pcm, f"{tag}_pcm_out": y, f"{tag}_wpe_r": taps["wpe"][0], f"{tag}_wpe_i": taps["wpe"][1],
                    f"{tag}_rates": np.array([in_rate, out_rate], np.int64)})
        print(tag, "in", pcm.shape, "out", y.shape, int(np.abs(y).max()))
    np.savez_compressed(os.path.join(mg.GOLD, f"hgtcrn_seed{seed}_resample.npz"), **out)


def dynamic_fixture(seed=0):
    """DYNAMIC_AXES = True (:27, :43-44, :1075, :1097, :1110): frame counts come from the waveform (MAX_SIGNAL_LENGTH = 4096 only sizes tables), and the ISTFT keeps
    everything after the leading half window -- half a window more than the static trim -- dividing by the window-square sum of the ACTUAL frames
    (H-GTCRN/STFT_Process.py:318-327; in that tail only the last frame contributes: x / w with w running out, most of it saturates the int16 clamp).  One module instance on two lengths (8192 and 12288 samples), the same seeded
    network as hgtcrn_seed{seed}.npz; the WPE outputs are kept for the two-part contract.  tests/golden/hgtcrn_seed{seed}_dynamic.npz"""
    z = np.load(os.path.join(mg.GOLD, f"hgtcrn_seed{seed}.npz"))
    ns = import_namespace(8192, extra={"DYNAMIC_AXES": True})
    assert ns["MAX_SIGNAL_LENGTH"] == 4096 and ns["MODEL_AUDIO_LENGTH"] == 0
    model, _, wpe, *_ = build(ns, seed, dynamic=True)
    out = {}
    for tag, row, length in (("a", 1, 8192), ("b", 2, 12288)):
        taps = {}
        orig = type(wpe).forward
        def tapped(*a, taps=taps):
            y = orig(wpe, *a)
            taps["wpe"] = [t.clone().numpy() for t in y]
            return y
        wpe.forward = tapped
        pcm = np.ascontiguousarray(z["pcm_in"][row][:, 500:500 + length])
        with torch.inference_mode():
            y = model(torch.from_numpy(pcm.reshape(1, 2, -1).copy())).numpy().reshape(-1)
        assert y.shape[0] == length + 256, y.shape
        out.update({f"{tag}_pcm_in": pcm, f"{tag}_pcm_out": y, f"{tag}_wpe_r": taps["wpe"][0], f"{tag}_wpe_i": taps["wpe"][1]})
        print("dynamic", tag, "in", pcm.shape, "out", y.shape, int(np.abs(y).max()), "tail max", int(np.abs(y[-256:]).max()))
    np.savez_compressed(os.path.join(mg.GOLD, f"hgtcrn_seed{seed}_dynamic.npz"), **out)


def float_io_fixture(seed=0):
    """IN / OUT_AUDIO_DTYPE other than INT16 (:52-53): a float input skips the * inv_int16 (:965-966), a float output the * 32767 and the clamp (:1042-1063).
    tests/golden/hgtcrn_float_io_seed{seed}.npz; the network and the input row are hgtcrn_seed{seed}.npz's (row 1)."""
    z = np.load(os.path.join(mg.GOLD, f"hgtcrn_seed{seed}.npz"))
    pcm = np.ascontiguousarray(z["pcm_in"][1])
    x = (pcm.astype(np.float32) / np.float32(32768.0)).astype(np.float32)
    out = {"pcm_in": pcm, "x_in": x}
    for tag, din, dout in (("f32_f32", "F32", "F32"), ("f32_i16", "F32", "INT16"), ("i16_f32", "INT16", "F32")):
        ns = import_namespace(pcm.shape[1], extra={"IN_AUDIO_DTYPE": din, "OUT_AUDIO_DTYPE": dout})
        model, *_ = build(ns, seed)
        src = pcm if din == "INT16" else x
        with torch.inference_mode():
            y = model(torch.from_numpy(src.reshape(1, 2, -1).copy())).numpy().reshape(-1)
        out[tag] = y
        print(tag, y.shape, y.dtype, float(np.abs(y).max()))
    np.savez_compressed(os.path.join(mg.GOLD, f"hgtcrn_float_io_seed{seed}.npz"), **out)


if __name__ == "__main__" and "--dynamic" in sys.argv:
    dynamic_fixture()
    sys.exit(0)

if __name__ == "__main__" and "--float-io" in sys.argv:
    float_io_fixture()
    sys.exit(0)

if __name__ == "__main__":
    main()
    fold_fixture()
    resample_fixture()
