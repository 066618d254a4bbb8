#!/usr/bin/env python3
"""ZipEnhancer golden vectors, produced by RUNNING THE REFERENCE's own ``ZipEnhancer`` wrapper -- constructor folds
(ZipEnhancer/Export_ZipEnhancer.py:385-699), forward overrides (:118-355, installed by its own ``apply_onnx_export_patches``)
and wrapper forward (:701-927) -- here, in the build container.

The reference takes its network from ``modelscope`` (``Model.from_pretrained`` :949; package and checkpoint both absent).  It only
(a) reads attributes of that network, (b) calls leaf modules and (c) replaces ten forwards with its own.  This tool therefore
registers a stand-in ``modelscope.models.audio.ans.zipenhancer_layers.{scaling, zipformer}`` whose classes carry exactly the
attributes the reference reads and standard torch leaves (Linear, Conv1d, Conv2d, InstanceNorm2d, PReLU); the forwards that run
are the reference's.  Code of this tool that takes part in the arithmetic and is NOT the reference's: ``FeedforwardModule.forward``
(in_proj -> out_proj), ``CompactRelPositionalEncoding``'s table (the published Zipformer2 formula), and the geometry
(``audio_denoiser_onnx_amd.zipenhancer.ZipConfig``): those are the parity-unpinned part.
Parameters are filled from ``zipenhancer.synthetic_state_dict`` (counter-based generator: the fixture carries the config and seed,
not 2.1 M floats), so tests rebuild the same checkpoint, fold it with ``zipenhancer.fuse_state_dict`` and must land on the
reference's outputs: this pins the fold AND the forward.

    python tools/make_golden_zipenhancer.py     # writes tests/golden/zipenhancer_seed0_io.npz, zipenhancer_seed0_fold_io.npz
"""
import ast
import math
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

from ref_import import REF_ROOT, _stub_absent_modules, import_stft_process  # noqa: E402
from audio_denoiser_onnx_amd import zipenhancer as zp  # noqa: E402

GOLD = os.path.join(REPO, "tests", "golden")


# ---- stand-in modelscope layers: attributes + leaves only; forwards come from the reference ------------------------------
class BiasNorm(nn.Module):
    def __init__(self, num_channels, channel_dim=-1):
        super().__init__()
        self.num_channels, self.channel_dim = num_channels, channel_dim
        self.log_scale = nn.Parameter(torch.tensor(0.0))
        self.bias = nn.Parameter(torch.zeros(num_channels))


class ActivationDropoutAndLinear(nn.Module):
    def __init__(self, in_channels, out_channels, activation):
        super().__init__()
        self.activation = activation
        self.weight = nn.Parameter(torch.zeros(out_channels, in_channels))
        self.bias = nn.Parameter(torch.zeros(out_channels))


class BypassModule(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.bypass_scale = nn.Parameter(torch.full((dim,), 0.5))


class SimpleDownsample(nn.Module):
    def __init__(self, ds):
        super().__init__()
        self.downsample = ds
        self.bias = nn.Parameter(torch.zeros(ds))


class SimpleUpsample(nn.Module):
    def __init__(self, us):
        super().__init__()
        self.upsample = us


class CompactRelPositionalEncoding(nn.Module):
    """The published Zipformer2 table (not in the reference; the reference slices it, :690-699)."""

    def __init__(self, embed_dim, max_len=1000, length_factor=1.0):
        super().__init__()
        T = max_len
        x = torch.arange(-(T - 1), T).to(torch.float32).unsqueeze(1)
        freqs = 1 + torch.arange(embed_dim // 2)
        compression_length = embed_dim ** 0.5
        x_compressed = compression_length * x.sign() * ((x.abs() + compression_length).log() - math.log(compression_length))
        length_scale = length_factor * embed_dim / (2.0 * math.pi)
        x_atan = (x_compressed / length_scale).atan()
        pe = torch.zeros(x.shape[0], embed_dim)
        pe[:, 0::2] = (x_atan * freqs).cos()
        pe[:, 1::2] = (x_atan * freqs).sin()
        pe[:, -1] = 1.0
        self.pe = pe


class RelPositionMultiheadAttentionWeights(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.num_heads, self.query_head_dim, self.pos_head_dim = c.heads, c.query_head_dim, c.pos_head_dim
        self.in_proj = nn.Linear(c.channels, c.attn_dim)
        self.linear_pos = nn.Linear(c.pos_dim, c.heads * c.pos_head_dim, bias=False)


class SelfAttention(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.in_proj = nn.Linear(c.channels, c.value_dim)
        self.out_proj = nn.Linear(c.value_dim, c.channels)
        self.whiten = nn.Identity()


class NonlinAttention(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.hidden_channels = c.hidden
        self.in_proj = nn.Linear(c.channels, 3 * c.hidden)
        self.out_proj = nn.Linear(c.hidden, c.channels)
        self.tanh, self.balancer, self.whiten1, self.whiten2 = nn.Tanh(), nn.Identity(), nn.Identity(), nn.Identity()


class ConvolutionModule(nn.Module):
    def __init__(self, c):
        super().__init__()
        C = c.channels
        self.in_proj = nn.Linear(C, 2 * C)
        self.sigmoid = nn.Sigmoid()
        self.balancer1 = self.balancer2 = self.activation1 = self.activation2 = self.whiten = nn.Identity()
        self.depthwise_conv = nn.Conv1d(C, C, c.conv_kernel, padding=c.conv_kernel // 2, groups=C)
        self.out_proj = ActivationDropoutAndLinear(C, C, "SwooshR")


class FeedforwardModule(nn.Module):
    def __init__(self, c, dim):
        super().__init__()
        self.in_proj = nn.Linear(c.channels, dim)
        self.out_proj = ActivationDropoutAndLinear(dim, c.channels, "SwooshL")

    def forward(self, x):                  # modelscope / icefall: in_proj -> (balancer) -> activation + out_proj -> (whiten)
        return self.out_proj(self.in_proj(x))


class Zipformer2EncoderLayer(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.self_attn_weights = RelPositionMultiheadAttentionWeights(c)
        self.feed_forward1, self.feed_forward2, self.feed_forward3 = FeedforwardModule(c, c.ff1), FeedforwardModule(c, c.ff_dim), FeedforwardModule(c, c.ff3)
        self.nonlin_attention = NonlinAttention(c)
        self.self_attn1, self.self_attn2 = SelfAttention(c), SelfAttention(c)
        self.conv_module1, self.conv_module2 = ConvolutionModule(c), ConvolutionModule(c)
        self.bypass_mid, self.bypass = BypassModule(c.channels), BypassModule(c.channels)
        self.norm = BiasNorm(c.channels)


class DualPathZipformer2Encoder(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.f_layers = nn.ModuleList([Zipformer2EncoderLayer(c)])
        self.t_layers = nn.ModuleList([Zipformer2EncoderLayer(c)])
        self.bypass_layers = nn.ModuleList([BypassModule(c.channels), BypassModule(c.channels)])
        self.encoder_pos = CompactRelPositionalEncoding(c.pos_dim)


class DualPathDownsampledZipformer2Encoder(nn.Module):
    def __init__(self, c, ds_t, ds_f):
        super().__init__()
        self.encoder = DualPathZipformer2Encoder(c)
        self.t_downsample_factor, self.f_downsample_factor = ds_t, ds_f
        self.downsample_t, self.downsample_f = SimpleDownsample(ds_t), SimpleDownsample(ds_f)
        self.upsample_t, self.upsample_f = SimpleUpsample(ds_t), SimpleUpsample(ds_f)
        self.out_combiner = BypassModule(c.channels)


class DenseBlockV2(nn.Module):
    def __init__(self, c):
        super().__init__()
        C = c.channels
        self.dense_block = nn.ModuleList([
            nn.Sequential(nn.ConstantPad2d((1, 1, 1 << i, 0), 0.0), nn.Conv2d(C * (i + 1), C, (2, 3), dilation=(1 << i, 1)),
                          nn.InstanceNorm2d(C, affine=True), nn.PReLU(C)) for i in range(c.dense_depth)])


class DenseEncoder(nn.Module):
    def __init__(self, c):
        super().__init__()
        C = c.channels
        self.dense_conv_1 = nn.Sequential(nn.Conv2d(2, C, (1, 1)), nn.InstanceNorm2d(C, affine=True), nn.PReLU(C))
        self.dense_block = DenseBlockV2(c)
        self.dense_conv_2 = nn.Sequential(nn.Conv2d(C, C, (1, 3), (1, 2), padding=(0, 1)), nn.InstanceNorm2d(C, affine=True), nn.PReLU(C))


class SPConvTranspose2d(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.upscale_width_factor = c.upscale
        self.conv1 = nn.Conv2d(c.channels, c.channels * c.upscale, (1, 3))


class MaskDecoder(nn.Module):
    def __init__(self, c):
        super().__init__()
        C = c.channels
        self.dense_block = DenseBlockV2(c)
        self.mask_conv = nn.Sequential(SPConvTranspose2d(c), nn.InstanceNorm2d(C, affine=True), nn.PReLU(C), nn.Conv2d(C, 1, (1, 2)))
        self.relu = nn.ReLU()


class PhaseDecoder(nn.Module):
    def __init__(self, c):
        super().__init__()
        C = c.channels
        self.dense_block = DenseBlockV2(c)
        self.phase_conv = nn.Sequential(SPConvTranspose2d(c), nn.InstanceNorm2d(C, affine=True), nn.PReLU(C))
        self.phase_conv_r, self.phase_conv_i = nn.Conv2d(C, 1, (1, 2)), nn.Conv2d(C, 1, (1, 2))


class TSConformer(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.encoders = nn.ModuleList([DualPathZipformer2Encoder(c), DualPathDownsampledZipformer2Encoder(c, c.down_t2, c.down_f2),
                                       DualPathDownsampledZipformer2Encoder(c, c.down_t2, c.down_f2), DualPathZipformer2Encoder(c)])


class StandInZipEnhancer(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.dense_encoder, self.TSConformer = DenseEncoder(c), TSConformer(c)
        self.mask_decoder, self.phase_decoder = MaskDecoder(c), PhaseDecoder(c)


def install_fake_modelscope():
    names = ["modelscope", "modelscope.models", "modelscope.models.base", "modelscope.models.audio", "modelscope.models.audio.ans",
             "modelscope.models.audio.ans.zipenhancer_layers", "modelscope.models.audio.ans.zipenhancer_layers.scaling",
             "modelscope.models.audio.ans.zipenhancer_layers.zipformer"]
    mods = {n: types.ModuleType(n) for n in names}
    for n, m in mods.items():
        sys.modules[n] = m
        if "." in n:
            setattr(mods[n.rsplit(".", 1)[0]], n.rsplit(".", 1)[1], m)
    scaling, zipformer = mods[names[-2]], mods[names[-1]]
    scaling.BiasNorm, scaling.ActivationDropoutAndLinear = BiasNorm, ActivationDropoutAndLinear
    for cls in (Zipformer2EncoderLayer, BypassModule, SimpleDownsample, SimpleUpsample, RelPositionMultiheadAttentionWeights, SelfAttention,
                NonlinAttention, ConvolutionModule, CompactRelPositionalEncoding):
        setattr(zipformer, cls.__name__, cls)
    mods["modelscope.models.base"].Model = object


def import_namespace(length: int, fold: bool, window_seconds: float = 1.5, extra: dict | None = None) -> dict:
    _stub_absent_modules()
    install_fake_modelscope()
    path = os.path.join(REF_ROOT, "ZipEnhancer", "Export_ZipEnhancer.py")
    with open(path) as f:
        tree = ast.parse(f.read(), filename=path)
    over = {"INPUT_AUDIO_LENGTH": length, "USE_BATCH_FOLD": fold, "BATCH_WINDOW_SECONDS": window_seconds}
    over.update(extra or {})
    keep = []
    for node in tree.body:
        if isinstance(node, (ast.ClassDef, ast.FunctionDef)):
            keep.append(node)
        elif isinstance(node, ast.Assign):
            names = [t.id for t in node.targets if isinstance(t, ast.Name)]
            if names and all(n.upper() == n for n in names):
                if len(names) == 1 and names[0] in over:
                    node = ast.parse(f"{names[0]} = {over[names[0]]!r}").body[0]
                keep.append(node)
    module = ast.Module(body=keep, type_ignores=[])
    ast.fix_missing_locations(module)
    ns = {"torch": torch, "math": math, "__name__": "ref_export_zipenhancer"}
    exec(compile(module, path, "exec"), ns)
    ns["_validate_export_configuration"]()
    ns["apply_onnx_export_patches"]()                                      # the reference installs ITS forwards on the stand-in classes
    return ns


def build_reference(cfg: zp.ZipConfig, seed: int, length: int, fold: bool, extra: dict | None = None):
    ns = import_namespace(length, fold, extra=extra)
    net = StandInZipEnhancer(cfg).eval()
    sd = zp.synthetic_state_dict(cfg, seed)
    own = net.state_dict()
    assert set(own) == set(sd), (sorted(set(own) ^ set(sd))[:8], len(own), len(sd))
    net.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)).reshape(own[k].shape) for k, v in sd.items()})
    stft_mod = import_stft_process("ZipEnhancer")
    with torch.inference_mode():
        stft = stft_mod.STFT_Process(model_type="stft_B", n_fft=ns["NFFT"], hop_len=ns["HOP_LENGTH"], win_length=ns["WINDOW_LENGTH"], max_frames=0,
                                     window_type=ns["WINDOW_TYPE"], center_pad=True, pad_mode="reflect").eval()
        istft = stft_mod.STFT_Process(model_type="istft_B", n_fft=ns["NFFT"], hop_len=ns["HOP_LENGTH"], win_length=ns["WINDOW_LENGTH"],
                                      max_frames=ns["MAX_SIGNAL_LENGTH"], window_type=ns["WINDOW_TYPE"], center_pad=True, pad_mode="reflect",
                                      static_norm=ns["STATIC_SHAPE"]).eval()
        model = ns["ZipEnhancer"](net, stft, istft, ns["IN_SAMPLE_RATE"], ns["OUT_SAMPLE_RATE"], use_batch_fold=ns["USE_BATCH_FOLD"],
                                  fold_window=ns["FOLD_WINDOW_LENGTH"], use_rectangular_istft=ns["USE_RECTANGULAR_ISTFT"]).eval()
    return ns, model, sd


def sub(x):
    """(b, c, t, f) -> channels-last, sub-sampled (b, t::4, f::5, c::8) -- what the fixture stores of each encoder tap."""
    return np.ascontiguousarray(x.permute(0, 2, 3, 1).numpy()[:, ::4, ::5, ::8])


def run_with_taps(model, pcm: np.ndarray):
    """One reference call on int16 (1, 1, n); returns (int16 out, fp32 pre-cast waveform, taps)."""
    taps = {}
    enc_calls = []
    orig_dual, orig_down, orig_inv = model._dualpath_encoder, model._downsampled_encoder, model.istft_model.inverse_packed

    def dual(e, x, b, c, t, f):
        if not enc_calls:
            taps["enc_in"] = sub(x)
        y = orig_dual(e, x, b, c, t, f)
        enc_calls.append(y)
        return y

    def down(e, x, b, c, t, f):
        y = orig_down(e, x, b, c, t, f)
        enc_calls.append(y)
        return y

    def inv(packed):
        taps["packed"] = packed.numpy().copy()
        y = orig_inv(packed)
        taps["istft"] = y.clone()
        return y
    model._dualpath_encoder, model._downsampled_encoder, model.istft_model.inverse_packed = dual, down, inv
    hook = model.zip_enhancer.mask_decoder.mask_conv[3].register_forward_hook(lambda m, i, o: taps.__setitem__("mask", o[:, 0].numpy().copy()))
    try:
        with torch.inference_mode():
            x = torch.from_numpy(pcm.reshape(1, 1, -1).copy())
            out = model(x)
            a = x.float()
            if model.use_batch_fold:
                a = a.reshape(-1, 1, model.fold_window)
            norm = torch.sqrt(torch.mean(a * a, dim=-1, keepdim=True) + 1e-6)          # (:839), recomputed for the fp32 tap
            wave = (taps.pop("istft") * norm).reshape(1, -1)                             # (:900-902)
    finally:
        hook.remove()
        model._dualpath_encoder, model._downsampled_encoder, model.istft_model.inverse_packed = orig_dual, orig_down, orig_inv
    for i, y in enumerate(enc_calls):
        taps[f"enc{i}"] = sub(y)
    return out.numpy().reshape(-1), wave.numpy().reshape(-1), taps


def load_wav_i16(path):
    import wave
    with wave.open(path, "rb") as w:
        assert w.getsampwidth() == 2 and w.getnchannels() == 1, path
        return np.frombuffer(w.readframes(w.getnframes()), dtype="<i2").copy(), w.getframerate()


def main():
    cfg, seed = zp.ZipConfig(), 0
    wav, sr = load_wav_i16(os.path.join(REF_ROOT, "Test_Examples", "denoise", "speech_with_noise1.wav"))
    assert sr == 16000
    torch.manual_seed(1234)
    randn = (torch.randn(16000) * 3000).clamp(-32768, 32767).to(torch.int16).numpy()
    # ---- plain static export at the BASELINE chunk: 1 s = 161 frames x 101 sub-bands
    L = 16000
    ns, model, sd = build_reference(cfg, seed, L, fold=False)
    fused_ref = {}                                               # the reference constructor's folds, for the fuse_state_dict pin
    lay = model.zip_enhancer.TSConformer.encoders[1].encoder.t_layers[0]
    fused_ref["enc1_t_attn_ff1_w"] = lay.onnx_attn_ff1_weight.numpy().copy()
    fused_ref["enc1_t_attn_ff1_b"] = lay.onnx_attn_ff1_bias.numpy().copy()
    fused_ref["enc1_t_final_norm_scale"] = lay.onnx_final_norm_scale.numpy().copy()
    fused_ref["enc1_t_final_residual_scale"] = lay.onnx_final_residual_scale.numpy().copy()
    fused_ref["enc1_t_ff2_out_b"] = lay.feed_forward2.out_proj.onnx_bias.numpy().copy()
    fused_ref["enc1_t_conv2_out_b"] = lay.conv_module2.out_proj.onnx_bias.numpy().copy()
    fused_ref["enc1_t_pos_table"] = lay.self_attn_weights.onnx_linear_pos.numpy().copy()          # (1, heads, pos_head_dim, 2 * 81 - 1)
    fused_ref["enc2_down_t_w"] = model.zip_enhancer.TSConformer.encoders[2].downsample_t.onnx_downsample_weights.numpy().reshape(-1).copy()
    fused_ref["enc2_res_scale"] = model.zip_enhancer.TSConformer.encoders[2].out_combiner.onnx_residual_scale.numpy().copy()
    fused_ref["dec_dense2_w"] = model.decoder_dense_conv_weight_2.numpy()[::8].copy()          # every 8th output row
    fused_ref["dec_up_w"] = model.decoder_up_weight.numpy()[::8].copy()
    fused_ref["phase_out_w"] = model.phase_output_weight.numpy().copy()
    rows = {"wav0": wav[16000:32000].copy(), "randn": randn, "zeros": np.zeros(L, np.int16)}
    out = {"config": cfg.as_tensor(), "seed": np.int64(seed), "length": np.int64(L)}
    for name, pcm in rows.items():
        o, wv, taps = run_with_taps(model, pcm)
        out[f"in_{name}"], out[f"out_{name}"], out[f"wave_{name}"] = pcm, o, wv
        if name == "wav0":
            for k, v in taps.items():
                out[f"tap_{k}"] = v[:, :, ::2] if k == "packed" else v
        print(name, "out rms", float(np.sqrt(np.mean(o.astype(np.float64) ** 2))), "in rms", float(np.sqrt(np.mean(pcm.astype(np.float64) ** 2))))
    out.update({f"fused_{k}": v for k, v in fused_ref.items()})
    np.savez_compressed(os.path.join(GOLD, "zipenhancer_seed0_io.npz"), **out)
    # ---- batch-fold export: 40000 samples -> 2 windows of 24000 (241 frames), the tail zero-padded outside the model
    ns, model, _ = build_reference(cfg, seed, 40000, fold=True)
    n = ns["EXPORT_AUDIO_LENGTH"]
    pcm = np.zeros(n, np.int16)
    pcm[:40000] = wav[8000:48000]
    o, wv, _ = run_with_taps(model, pcm)
    np.savez_compressed(os.path.join(GOLD, "zipenhancer_seed0_fold_io.npz"), config=cfg.as_tensor(), seed=np.int64(seed), length=np.int64(40000),
                        export_length=np.int64(n), fold_window=np.int64(ns["FOLD_WINDOW_LENGTH"]), pcm_in=pcm, pcm_out=o, wave=wv)
    print("fold: windows", n // ns["FOLD_WINDOW_LENGTH"], "out rms", float(np.sqrt(np.mean(o.astype(np.float64) ** 2))))


def float_io_fixture():
    """IN / OUT_AUDIO_DTYPE other than INT16 (:35-36): a float input is lifted by * 32768 (:820-821), a float output is nan_to_num(waveform) * 2^-15 (:920-926).
    tests/golden/zipenhancer_float_io_seed0.npz; the network is zipenhancer_seed0_io.npz's (seed 0), the clip its wav0 row cut to 0.5 s."""
    cfg, seed, L = zp.ZipConfig(), 0, 8000
    z = np.load(os.path.join(GOLD, "zipenhancer_seed0_io.npz"))
    pcm = np.ascontiguousarray(z["in_wav0"][4000:4000 + L])
    x = (pcm.astype(np.float32) / np.float32(32768.0)).astype(np.float32)
    out = {"pcm_in": pcm, "x_in": x}
    for tag, din, dout in (("f32_f32", "F32", "F32"), ("f32_i16", "F32", "INT16"), ("i16_f32", "INT16", "F32"), ("i16_i16", "INT16", "INT16")):
        ns, model, _ = build_reference(cfg, seed, L, fold=False, extra={"IN_AUDIO_DTYPE": din, "OUT_AUDIO_DTYPE": dout})
        src = pcm if din == "INT16" else x
        with torch.inference_mode():
            y = model(torch.from_numpy(src.reshape(1, 1, -1).copy())).numpy().reshape(-1)
        out[tag] = y
        print(tag, y.shape, y.dtype, float(np.abs(y).max()))
    np.savez_compressed(os.path.join(GOLD, "zipenhancer_float_io_seed0.npz"), **out)


def dynamic_fixture():
    """DYNAMIC_AXES = True (:31, :61 MAX_SIGNAL_LENGTH = 1024, :828-829, :898-899, :907-908): a free input length, scale-factor edges, position tables sliced from
    the 1024-frame table, ISTFT denominator built from the frame count and divided.  tests/golden/zipenhancer_dynamic_seed0.npz: 8 050 samples at 16 kHz throughout
    (80 hops + 50 samples: 81 frames -> 8 000 out) and 6 000 samples at 12 kHz -> 16 kHz -> 24 kHz; the network is seed 0's."""
    cfg, seed = zp.ZipConfig(), 0
    z = np.load(os.path.join(GOLD, "zipenhancer_seed0_io.npz"))
    out = {}
    ns, model, _ = build_reference(cfg, seed, 8050, fold=False, extra={"DYNAMIC_AXES": True})
    assert ns["MAX_SIGNAL_LENGTH"] == 1024 and not ns["STATIC_SHAPE"]
    def run(model, x, tag):       # the reference's output and its analysis spectrum (the network is pinned on identical spectra: the edge frames' phase is ill-conditioned)
        spec = {}
        hook = model.stft_model.register_forward_hook(lambda m, i, o: spec.update(re=o[0].numpy().copy(), im=o[1].numpy().copy()))
        try:
            with torch.inference_mode():
                y = model(torch.from_numpy(x.reshape(1, 1, -1).copy())).numpy().reshape(-1)
        finally:
            hook.remove()
        out[tag + "_in"], out[tag + "_out"], out[tag + "_spec_re"], out[tag + "_spec_im"] = x, y, spec["re"][0], spec["im"][0]
    x = np.ascontiguousarray(z["in_wav0"][3000:3000 + 8050])
    run(model, x, "eq")
    ns, model, _ = build_reference(cfg, seed, 6000, fold=False, extra={"DYNAMIC_AXES": True, "IN_SAMPLE_RATE": 12000, "OUT_SAMPLE_RATE": 24000})
    x = np.ascontiguousarray(z["in_wav0"][2000:2000 + 6000])
    run(model, x, "rs")
    out["rs_in_rate"], out["rs_out_rate"] = np.int64(12000), np.int64(24000)
    np.savez_compressed(os.path.join(GOLD, "zipenhancer_dynamic_seed0.npz"), **out)
    print("dynamic: equal rates", out["eq_in"].shape, "->", out["eq_out"].shape, "; 12k -> 24k", out["rs_in"].shape, "->", out["rs_out"].shape)


if __name__ == "__main__" and "--dynamic" in sys.argv:
    dynamic_fixture()
    sys.exit(0)

if __name__ == "__main__" and "--float-io" in sys.argv:
    float_io_fixture()
    sys.exit(0)

if __name__ == "__main__":
    main()
