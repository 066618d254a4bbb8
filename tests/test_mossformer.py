"""MossFormer2-SS-16K: oracle pin (CPU) and HIP parity through the C ABI (GPU).

Fixtures: tests/golden/mossformer_seed0_io.npz / mossformer_seed0_fold_io.npz = the reference's own constructor + forward run in
the build container over a stand-in network tree with generator-filled fused buffers (tools/make_golden_mossformer.py); the 6 M
weights are regenerated here from (name, shape, scale) (audio_denoiser_onnx_amd.mossformer.synthetic_tensor).
"""
import json
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(REPO, "oracle"))

from audio_denoiser_onnx_amd import mossformer  # noqa: E402

GOLD = os.path.join(HERE, "golden", "mossformer_seed0_io.npz")
GOLD_FOLD = os.path.join(HERE, "golden", "mossformer_seed0_fold_io.npz")
GOLD_RESAMPLE = os.path.join(HERE, "golden", "mossformer_seed0_resample_io.npz")


@pytest.fixture(scope="module")
def fixture():
    z = np.load(GOLD)
    spec = json.loads(str(z["spec"]))
    scalars = json.loads(str(z["scalars"]))
    W = z["pcm_in"].shape[0]
    frames = mossformer.frames_of(W)
    fused = {n: mossformer.synthetic_tensor(n, s, sc, frames, int(scalars["flash_group_size"])) for n, s, sc in spec}
    return z, fused, scalars, W


def _oracle(fixture):
    from mossformer_oracle import MossFormerOracle
    z, fused, scalars, W = fixture
    tensors = dict(fused)
    tensors.update(mossformer.position_tables(mossformer.frames_of(W), int(scalars["rot_dim"])))
    return MossFormerOracle(tensors, scalars, int(z["layers"]), W)


def test_oracle_matches_reference_forward(fixture):
    z = fixture[0]
    o = _oracle(fixture)
    out = o.process(z["pcm_in"][None])[0]
    assert np.abs(o.taps["mdl_in"][0][:, ::7] - z["mdl_in"]).max() < 5e-4
    assert np.abs(o.taps["mdl_out"][0][:, ::7] - z["mdl_out"]).max() < 1e-3
    d = out.astype(np.int32) - z["pcm_out"].astype(np.int32)
    assert np.abs(d).max() <= 1 and (d != 0).mean() < 0.02
    assert np.abs(z["pcm_out"]).max(axis=1).min() > 2000                      # both separated sources carry signal


def test_oracle_batch_fold_matches_reference_forward(fixture):
    """USE_BATCH_FOLD: 3 windows, per-window RMS normalisation and restore, the last window partly silent."""
    zf = np.load(GOLD_FOLD)
    out = _oracle(fixture).process_fold(zf["pcm_in"])
    d = out.astype(np.int32) - zf["pcm_out"].astype(np.int32)
    assert np.abs(d).max() <= 1 and (d != 0).mean() < 0.02


def test_oracle_resampling_edges_match_reference_forward(fixture):
    """IN 8 kHz -> model 16 kHz -> OUT 48 kHz in the reference (F.interpolate on both edges) against the oracle."""
    zr = np.load(GOLD_RESAMPLE)
    out = _oracle(fixture).process(zr["pcm_in"][None], out_len=zr["pcm_out"].shape[1])[0]
    d = out.astype(np.int32) - zr["pcm_out"].astype(np.int32)
    assert np.abs(d).max() <= 1 and (d != 0).mean() < 0.02


def test_position_tables_and_hyper_layout(fixture):
    _, fused, scalars, W = fixture
    t = mossformer.model_tensors(fused, scalars, W)
    n = mossformer.frames_of(W)
    assert t["emb_pos"].shape == (1, 512, n) and t["rot_cos"].shape == (1, n, 1, 32) and t["rot_signed_sin"].shape == (1, n, 1, 32)
    assert np.allclose(t["emb_pos"][0, 256:, 0], 1.0) and np.allclose(t["emb_pos"][0, :256, 0], 0.0)          # sin | cos halves at t = 0
    assert t["rot_signed_sin"][0, 5, 0, 0] == -t["rot_signed_sin"][0, 5, 0, 1]                                # sign folded pairwise
    h = dict(zip(mossformer.HYPER_KEYS, t["hyper"]))
    assert h["flash_group_size"] == 256 and h["fs_mem_lorder"] == 20 and h["fs_mem_depth"] == 2 and h["dw_pad"] == 8
    assert t["fs_front_alpha"].shape == (2,)


# ---- GPU: the HIP engine through the C ABI ------------------------------------------------------------------------------
def _session(fixture, length, **meta_kw):
    from audio_denoiser_onnx_amd.session import InferenceSession
    from audio_denoiser_onnx_amd.weights import pack_blob
    _, fused, scalars, W = fixture
    return InferenceSession(weights=pack_blob(mossformer.model_tensors(fused, scalars, W)), metadata=mossformer.metadata(length, **meta_kw))


@pytest.mark.gpu
def test_gpu_matches_reference_fixture(fixture):
    """int16 (1, 1, W) through libade vs the reference's own forward: two outputs, within 2 LSB."""
    z, _, _, W = fixture
    with _session(fixture, W) as sess:
        assert sess.n_outputs == 2 and [o.name for o in sess.get_outputs()] == ["separated_0", "separated_1"] and sess.get_inputs()[0].name == "mix_audio"
        outs = sess.run(None, {"mix_audio": z["pcm_in"][None, None]})
        mdl_in = sess.tap("mdl_in", sess.frames * 512).reshape(sess.frames, 512).T
        mdl_out = sess.tap("mdl_out", sess.frames * 512).reshape(sess.frames, 512).T
    assert len(outs) == 2 and outs[0].shape == (1, 1, W) and outs[0].dtype == np.int16
    print("mossformer taps vs the reference: mdl_in %.2e, mdl_out %.2e (|mdl_out| max %.1f)"
          % (np.abs(mdl_in[:, ::7] - z["mdl_in"]).max(), np.abs(mdl_out[:, ::7] - z["mdl_out"]).max(), np.abs(z["mdl_out"]).max()))
    assert np.abs(mdl_in[:, ::7] - z["mdl_in"]).max() < 3e-4          # observed 5.9e-5
    assert np.abs(mdl_out[:, ::7] - z["mdl_out"]).max() < 3e-4         # observed 6.2e-5 on values up to 11.9
    for spk in range(2):
        d = outs[spk][0, 0].astype(np.int32) - z["pcm_out"][spk].astype(np.int32)
        assert np.abs(d).max() <= 2 and (d != 0).mean() < 0.05, (spk, np.abs(d).max(), (d != 0).mean())


@pytest.mark.gpu
def test_gpu_batch_fold_matches_reference_fixture(fixture):
    """use_batch_fold = 1: one call = (1, 1, 3 * W), per-window RMS normalisation / restore, a partly silent last window."""
    W = fixture[3]
    zf = np.load(GOLD_FOLD)
    meta_len = int(zf["input_audio_length"])
    with _session(fixture, meta_len, use_batch_fold=True, batch_window_seconds=W / 16000.0) as sess:
        assert sess.in_len == zf["pcm_in"].shape[0] and sess.frames == mossformer.frames_of(W)
        outs = sess.run(None, {"mix_audio": np.stack((zf["pcm_in"], zf["pcm_in"]))[:, None]})
    for spk in range(2):
        d = outs[spk][0, 0].astype(np.int32) - zf["pcm_out"][spk].astype(np.int32)
        assert np.abs(d).max() <= 2 and (d != 0).mean() < 0.05, (spk, np.abs(d).max(), (d != 0).mean())
        assert np.array_equal(outs[spk][0], outs[spk][1])


@pytest.mark.gpu
def test_gpu_batch_rows_and_oracle_on_other_inputs(fixture):
    """Batch rows are independent calls; seeded-noise, silent and full-scale windows against the oracle."""
    z, _, _, W = fixture
    rng = np.random.default_rng(5)
    rows = np.stack([z["pcm_in"], (rng.standard_normal(W) * 4000).astype(np.int16), np.zeros(W, np.int16),
                     (np.sign(np.sin(np.arange(W) * 0.05)) * 32767).astype(np.int16)])
    want = _oracle(fixture).process(rows)                                   # (4, 2, W)
    with _session(fixture, W) as sess:
        outs = sess.run(None, {"mix_audio": rows[:, None]})
        one = sess.run(None, {"mix_audio": rows[1:2, None]})
    got = np.stack((outs[0][:, 0], outs[1][:, 0]), axis=1)
    d = got.astype(np.int32) - want.astype(np.int32)
    assert np.abs(d).max() <= 3 and (d != 0).mean() < 0.05, (np.abs(d).max(), (d != 0).mean())
    assert np.all(got[2] == 0)                                               # a silent window stays silent (gain 0 / 0 -> 0, :618-622)
    assert np.array_equal(one[0][0], outs[0][1]) and np.array_equal(one[1][0], outs[1][1])


def test_checkpoint_fusion_matches_reference():
    """mossformer.fuse_checkpoint (clearvoice state_dict -> fused buffers + scalars) against the reference's own export constructor
    run over a one-layer stand-in tree with generator-filled parameters (tools/make_golden_mossformer.py::fusion_fixture)."""
    from audio_denoiser_onnx_amd import weightgen
    z = np.load(os.path.join(HERE, "golden", "mossformer_fusion.npz"))
    state = {}
    for key, shape, scale in json.loads(str(z["spec"])):
        v = weightgen.tensor(key, shape, scale)
        state[key] = np.abs(v) + np.float32(0.4) if scale == 0.6 else v
    want_sc = json.loads(str(z["scalars"]))
    fused, sc = mossformer.fuse_checkpoint(state, int(want_sc["static_frames"]))
    fused.update({k: v for k, v in mossformer.position_tables(int(want_sc["static_frames"]), 32, sc["pos_scale"]).items() if k == "emb_pos"})
    names = json.loads(str(z["names"]))
    assert set(names) == set(fused), set(names) ^ set(fused)
    for name in names:
        v = fused[name].reshape(-1).astype(np.float64)
        got = np.concatenate((v[::max(1, len(v) // 64)][:64], [v.sum()]))
        want = z[f"s_{name}"]
        tol = 2e-4 if name == "emb_pos" else 2e-6                  # emb_pos: sin / cos of fp32 angles up to 300 rad, libm vs torch
        assert np.allclose(got[:-1], want[:-1], rtol=tol, atol=tol), name
        assert abs(got[-1] - want[-1]) <= 1e-5 * max(1.0, np.abs(v).sum()), name
    for k in ("tail_prelu_alpha", "fl_norm_eps", "fl_out_norm_eps", "front_norm_eps", "mm_norm_eps", "intra_norm_eps", "fs_ln_eps", "fs_n1_eps", "fs_n2_eps",
              "norm_factor", "flash_group_size", "rot_dim", "dw_pad", "fs_mem_depth"):
        assert abs(float(sc[k]) - float(want_sc[k])) <= 1e-6 * abs(float(want_sc[k])), k
    assert np.allclose(sc["fs_front_alpha"], want_sc["fs_front_alpha"])
    # the DYNAMIC_AXES constructor (:24, :183): the same buffers except the OffsetScale rows, whose linear-key row carries no 1 / frames
    dyn, _ = mossformer.fuse_checkpoint(state, int(want_sc["static_frames"]), fold_inv_n=False)
    for name in fused:
        if name.startswith("qkos_"):
            v = dyn[name].reshape(-1).astype(np.float64)
            got = np.concatenate((v[::max(1, len(v) // 64)][:64], [v.sum()]))
            assert np.allclose(got, z[f"dyn_{name}"], rtol=2e-6, atol=2e-6), name
            assert not np.array_equal(dyn[name][3], fused[name][3]) and np.array_equal(dyn[name][:3], fused[name][:3]), name
        elif name != "emb_pos":
            assert np.array_equal(dyn[name], fused[name]), name


@pytest.mark.gpu
def test_gpu_resampling_edges_match_reference_fixture(fixture):
    """in_sample_rate 8000 / out_sample_rate 48000 manifest: linear interpolation on both edges around the 16 kHz model, like the export."""
    zr = np.load(GOLD_RESAMPLE)
    L_in, L_out = zr["pcm_in"].shape[0], zr["pcm_out"].shape[1]
    with _session(fixture, L_in, in_sample_rate=int(zr["in_rate"]), out_sample_rate=int(zr["out_rate"])) as sess:
        assert sess.in_len == L_in and sess.out_len == L_out and sess.frames == mossformer.frames_of(fixture[3])
        outs = sess.run(None, {"mix_audio": np.stack((zr["pcm_in"], zr["pcm_in"]))[:, None]})
    for spk in range(2):
        d = outs[spk][0, 0].astype(np.int32) - zr["pcm_out"][spk].astype(np.int32)
        assert np.abs(d).max() <= 2 and (d != 0).mean() < 0.05, (spk, np.abs(d).max(), (d != 0).mean())
        assert np.array_equal(outs[spk][0], outs[spk][1])


# ---- production-relevant sizes: 4 layers x 2999 frames (12 FLASH groups, the last padded) and 2 layers x one 4 s window
#      (7999 frames, 32 groups = BASELINE configs[4]'s window).  Reference-run fixtures: tools/make_golden_mossformer.py --production-size
def _big_fixture(tag):
    z = np.load(os.path.join(HERE, "golden", f"mossformer_seed0_{tag}_io.npz"))
    spec, scalars = json.loads(str(z["spec"])), json.loads(str(z["scalars"]))
    W = z["pcm_in"].shape[0]
    fused = {n: mossformer.synthetic_tensor(n, s, sc, mossformer.frames_of(W), int(scalars["flash_group_size"])) for n, s, sc in spec}
    return z, fused, scalars, W


def test_oracle_full_depth_matches_reference_forward():
    """The numpy restatement against the reference's own forward at the BASELINE depth: 24 layers (Export_MossFormer2_SS_16K.py:460-550) x one 1 s window (1999 frames)."""
    from mossformer_oracle import MossFormerOracle
    z, fused, scalars, W = _big_fixture("l24_1999")
    assert int(z["layers"]) == 24
    tensors = dict(fused)
    tensors.update(mossformer.position_tables(mossformer.frames_of(W), int(scalars["rot_dim"])))
    o = MossFormerOracle(tensors, scalars, 24, W)
    out = o.process(z["pcm_in"][None])[0]
    tap = float(np.abs(o.taps["mdl_out"][0][::8, ::7] - z["mdl_out"]).max())
    wave = float(np.abs(o.taps["wav"][0][:, ::int(z["wave_step"])] - z["wave"]).max())
    d = out.astype(np.int32) - z["pcm_out"].astype(np.int32)
    print("oracle 24 layers vs reference: mdl_out max %.2e, wave max %.3f (int16 units), PCM max %d LSB" % (tap, wave, np.abs(d).max()))
    # (24 layers of fp32 round-off between numpy's and torch's summation orders: 1.85 units = 5.7e-5 of full scale on the waveform, so the truncating cast may differ by 2)
    assert tap < 5e-3 and wave <= 3.3 and np.abs(d).max() <= 2 and (d != 0).mean() < 0.10


@pytest.mark.gpu
def test_gpu_full_depth_error_growth_vs_oracle():
    """VERDICT r05 missing #1: HIP vs the numpy oracle at the depth `bench.py --workload mossformer` times (24 layers) on one 1 s window, with the growth of the
    masking network's output error printed for networks cut after k = 1, 2, 4, 8, 16, 24 layers (the engine has one "mdl_out" tap, so it is built once per k from the
    first k layers of the same weights; the oracle keeps the truncated outputs of one pass).  Gates at 24 layers: fp32 waveform within 1e-4 of full scale (3.3 in
    these int16 units), PCM <= 2 LSB."""
    from mossformer_oracle import MossFormerOracle
    z, fused, scalars, W = _big_fixture("l24_1999")
    n = mossformer.frames_of(W)
    tensors = dict(fused)
    tensors.update(mossformer.position_tables(n, int(scalars["rot_dim"])))
    o = MossFormerOracle(tensors, scalars, 24, W)
    o.tap_after = (1, 2, 4, 8, 16, 24)
    want = o.process(z["pcm_in"][None])[0]
    want_wave = o.taps["wav"][0]

    def layer_of(name):
        parts = name.split("_")
        idx = [p for p in parts if p.isdigit()]
        return int(idx[0]) if idx and name.startswith(("fl_", "fs_", "qkos_")) else -1
    rows = []
    for k in o.tap_after:
        fk = {nm: v for nm, v in fused.items() if layer_of(nm) < k}
        sk = dict(scalars, fs_front_alpha=list(scalars["fs_front_alpha"])[:k])
        with _session((z, fk, sk, W), W) as sess:
            pcm, f32 = sess.process(z["pcm_in"][None], want_f32=True)
            got = sess.tap("mdl_out", n * 512).reshape(n, 512).T
        ref = o.taps["mdl_out_after"][k][0]
        e = np.abs(got - ref)
        rows.append((k, float(np.median(e)), float(e.max()), float(np.sqrt((ref.astype(np.float64) ** 2).mean()))))
    print("mossformer2 HIP vs oracle, |mdl_out error| of the network cut after k layers (median | max | rms of the tensor):")
    for r in rows:
        print("   k = %2d: %.2e | %.2e | %.2f" % r)
    pcm, f32 = pcm.reshape(2, W), f32.reshape(2, W)
    wave_err = float(np.abs(f32 - want_wave).max())
    d = np.abs(pcm.astype(np.int32) - want.astype(np.int32))
    print("   24 layers: wave max|d| %.3f (int16 units; 1e-4 of full scale = 3.3), PCM max %d LSB, %.3f of samples differ" % (wave_err, d.max(), (d != 0).mean()))
    assert rows[-1][2] < 5e-3, rows
    assert wave_err <= 3.3 and d.max() <= 2 and (d != 0).mean() < 0.05, (wave_err, d.max())


@pytest.mark.gpu
@pytest.mark.parametrize("tag,frames", [("l4_2999", 2999), ("l2_7999", 7999), ("l24_1999", 1999)])
def test_gpu_production_size_windows_match_reference(tag, frames):
    """HIP vs the reference's own forward: both speakers' PCM <= 2 LSB AND the fp32 waveform before the integer cast within 1e-4 of full scale
    (3.3 in these int16 units), taps of the masking network at every 8th channel / 7th frame.  l24_1999 (round 6) = ALL 24 layers on a 1 s window."""
    fx = _big_fixture(tag)
    z, _, _, W = fx
    with _session(fx, W) as sess:
        assert sess.frames == frames
        pcm, f32 = sess.process(z["pcm_in"][None], want_f32=True)
        mdl_out = sess.tap("mdl_out", sess.frames * 512).reshape(sess.frames, 512).T
    pcm, f32 = pcm.reshape(2, W), f32.reshape(2, W)
    assert np.abs(mdl_out[::8, ::7] - z["mdl_out"]).max() < 5e-3
    assert np.abs(f32[:, ::int(z["wave_step"])] - z["wave"]).max() <= 3.3
    d = pcm.astype(np.int32) - z["pcm_out"].astype(np.int32)
    # (at 24 layers the numpy oracle itself sits 1.85 units / 2 LSB from torch's forward -- summation order; the engine is 0.2 units from the oracle,
    #  test_gpu_full_depth_error_growth_vs_oracle -- so the share of samples whose truncating cast lands on the other integer is larger there)
    assert np.abs(d).max() <= 2 and (d != 0).mean() < (0.12 if tag == "l24_1999" else 0.05), (np.abs(d).max(), (d != 0).mean())


@pytest.mark.gpu
def test_gpu_baseline_batch_64_windows_of_4s_properties():
    """BASELINE configs[4]'s shape: 64 x 4 s (7999 frames) in one call on the 2-layer model: finite, silent rows stay silent, rows equal their solo run."""
    fx = _big_fixture("l2_7999")
    z, _, _, W = fx
    from audio_denoiser_onnx_amd.synth import synth_chunk
    rows = np.stack([z["pcm_in"] if i == 0 else (np.zeros(W, np.int16) if i == 5 else synth_chunk(300 + i, W)) for i in range(64)])
    with _session(fx, W) as sess:
        outs, f32 = sess.process(rows, want_f32=True)
        solo, _ = sess.process(rows[:1])
    outs = outs.reshape(64, 2, W)
    assert np.isfinite(f32).all() and not outs[5].any()
    d = outs[0].astype(np.int32) - z["pcm_out"].astype(np.int32)
    assert np.abs(d).max() <= 2
    assert np.array_equal(outs[0], solo.reshape(2, W))          # a row's bits do not depend on its batch (measured at B = 2, 3, 8, 33, 64: identical fp32 waveforms)


@pytest.mark.gpu
def test_gpu_baseline_batch_64_windows_of_4s_on_the_24_layer_model():
    """BASELINE configs[4] exactly -- 64 x 4 s (7999 frames), 24 layers, ONE call -- on the random-init weights `bench.py --workload mossformer` times (the
    64-row test above runs 2 layers).  Size-independent properties: finite, a silent row exactly silent, rows equal their solo run and the reversed batch."""
    from audio_denoiser_onnx_amd.session import InferenceSession
    from audio_denoiser_onnx_amd.synth import synth_chunk
    from audio_denoiser_onnx_amd.weights import pack_blob
    L, layers, B = 64000, 24, 64
    frames = mossformer.frames_of(L)
    fused = {n: mossformer.synthetic_tensor(n, sh, sc, frames) for n, sh, sc in mossformer.synthetic_spec(layers)}
    scalars = dict(mossformer.DEFAULT_SCALARS, fs_front_alpha=[0.25] * layers)
    blob = pack_blob(mossformer.model_tensors(fused, scalars, L))
    del fused
    x = np.stack([np.zeros(L, np.int16) if i == 5 else synth_chunk(700 + i, L) for i in range(B)])
    with InferenceSession(weights=blob, metadata=mossformer.metadata(L)) as sess:
        assert sess.frames == 7999
        out, f32 = sess.process(x, want_f32=True)
        solo, _ = sess.process(x[17:18])
        rev, _ = sess.process(x[::-1].copy())
    out = out.reshape(B, 2, -1)
    assert np.isfinite(f32).all() and not out[5].any()
    assert (np.abs(np.delete(out, 5, axis=0)).max(axis=(1, 2)) > 50).all()
    assert np.array_equal(out[17], solo.reshape(2, -1)), "row 17 depends on its neighbours at batch 64"
    assert np.array_equal(rev.reshape(B, 2, -1), out[::-1])


@pytest.mark.gpu
def test_gpu_reduced_precision_manifests_are_refused(fixture):
    """MossFormer2-SS runs f32 (BASELINE.json's dtype for it): the bf16-in-HBM path exists for Mel-Band-Roformer only, and the round-2 mode that rounded fp32 operands on
    their way into LDS is gone -- a manifest asking for either is refused at create, not served by something else."""
    W = fixture[3]
    for dt in ("bf16", "bf16_inputs"):
        with pytest.raises(Exception):
            _session(fixture, W, gemm_dtype=dt)


@pytest.mark.gpu
def test_gpu_file_driver_head_padding_two_outputs(fixture, tmp_path):
    """inference_mossformer.main() end to end on the GPU (this driver was only CPU-tested): model directory from the blob + manifest, 8000 zeros of
    head padding, static slices, one batched call, head drop + trim, two wav files; equals the per-slice session calls (Inference_MossFormer_SS_ONNX.py:273-340)."""
    from audio_denoiser_onnx_amd import inference_mossformer
    from audio_denoiser_onnx_amd.inference_gtcrn import read_wav_int16, write_wav_int16
    from audio_denoiser_onnx_amd.metadata import write_metadata
    from audio_denoiser_onnx_amd.session import InferenceSession
    from audio_denoiser_onnx_amd.synth import synth_chunk
    from audio_denoiser_onnx_amd.weights import save_blob
    _, fused, scalars, W = fixture
    model = tmp_path / "MossFormer2_SS_16K.adew"
    save_blob(model, mossformer.model_tensors(fused, scalars, W))
    write_metadata(model, mossformer.metadata(W, use_batch_fold=False))
    audio = (synth_chunk(7, 5000).astype(np.int32) + synth_chunk(8, 5000)).clip(-32768, 32767).astype(np.int16)
    write_wav_int16(tmp_path / "mix.wav", audio, 16000)
    assert inference_mossformer.main([str(model), str(tmp_path / "mix.wav"), str(tmp_path / "sep"), "--seed", "3"]) == 0
    outs = [read_wav_int16(tmp_path / f"sep_{i}.wav", 16000) for i in range(2)]
    assert all(o.shape == audio.shape for o in outs)
    padded = np.concatenate((np.zeros(8000, np.int16), audio))
    slices = inference_mossformer.cut_slices(padded, W, False, np.random.default_rng(3))
    with InferenceSession(str(model)) as sess:
        want = sess.run(None, {"mix_audio": slices[:, None, :]})
    for spk in range(2):
        assert np.array_equal(outs[spk], want[spk].reshape(-1)[8000:len(padded)])


@pytest.mark.gpu
def test_gpu_full_depth_full_length_properties():
    """BASELINE configs[4]'s network and window -- 24 layers, 4 s windows (7999 frames) -- on the random-init weights `bench.py --workload mossformer` times: finite
    outputs, a silent row stays silent, a row's bits do not depend on the rows beside it."""
    from audio_denoiser_onnx_amd import mossformer
    from audio_denoiser_onnx_amd.session import InferenceSession
    from audio_denoiser_onnx_amd.synth import synth_chunk
    from audio_denoiser_onnx_amd.weights import pack_blob
    L, layers = 64000, 24
    frames = mossformer.frames_of(L)
    fused = {n: mossformer.synthetic_tensor(n, sh, sc, frames) for n, sh, sc in mossformer.synthetic_spec(layers)}
    scalars = dict(mossformer.DEFAULT_SCALARS, fs_front_alpha=[0.25] * layers)
    blob = pack_blob(mossformer.model_tensors(fused, scalars, L))
    del fused
    x = np.stack([synth_chunk(500, L), np.zeros(L, np.int16), synth_chunk(501, L)])
    with InferenceSession(weights=blob, metadata=mossformer.metadata(L)) as sess:
        assert sess.frames == 7999
        out, f32 = sess.process(x, want_f32=True)
        solo, _ = sess.process(x[2:3])
    out = out.reshape(3, 2, -1)
    assert np.isfinite(f32).all() and np.abs(out[0]).max() > 50 and not out[1].any()
    assert np.array_equal(out[2], solo.reshape(2, -1))


# ---- DYNAMIC_AXES export (Export_MossFormer2_SS_16K.py:24): run-time 1 / frames, scale-factor edges; one reference module instance run on two lengths ------------------
GOLD_DYN = os.path.join(HERE, "golden", "mossformer_dynamic_seed0.npz")


def _dynamic_case(tag):
    z = np.load(GOLD_DYN)
    spec, scalars = json.loads(str(z["spec"])), json.loads(str(z["scalars"]))
    pcm, ref = z["pcm_in_" + tag], z["pcm_out_" + tag]
    in_rate, out_rate = (8000, 48000) if tag == "c" else (16000, 16000)
    W = int(np.floor(pcm.shape[0] * (16000 / in_rate)))
    frames = mossformer.frames_of(W)
    fused = {n: mossformer.synthetic_tensor(n, s, sc, frames, int(scalars["flash_group_size"]), fold_inv_n=False) for n, s, sc in spec}
    return pcm, ref, fused, scalars, W, int(z["layers"]), in_rate, out_rate


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_oracle_dynamic_axes_match_reference(tag):
    from mossformer_oracle import MossFormerOracle
    pcm, ref, fused, scalars, W, layers, in_rate, out_rate = _dynamic_case(tag)
    tensors = dict(fused)
    tensors.update(mossformer.position_tables(mossformer.frames_of(W), int(scalars["rot_dim"])))
    out = MossFormerOracle(tensors, scalars, layers, W, dynamic=True).process_dynamic(pcm[None], in_rate, out_rate)[0]
    assert out.shape == ref.shape
    d = out.astype(np.int32) - ref.astype(np.int32)
    assert np.abs(d).max() <= 1 and (d != 0).mean() < 0.02, (np.abs(d).max(), (d != 0).mean())
    # the static arithmetic on the same (unfolded) weights is a different function: the run-time factor matters
    if tag == "a":
        stat = MossFormerOracle(tensors, scalars, layers, W, dynamic=False).process(pcm[None])[0]
        assert np.abs(stat.astype(np.int32) - ref.astype(np.int32)).max() > 50


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_gpu_dynamic_axes_match_reference(tag):
    from audio_denoiser_onnx_amd.session import InferenceSession
    from audio_denoiser_onnx_amd.weights import pack_blob
    pcm, ref, fused, scalars, W, layers, in_rate, out_rate = _dynamic_case(tag)
    meta = mossformer.metadata(pcm.shape[0], in_sample_rate=in_rate, out_sample_rate=out_rate, dynamic_axes=True)
    with InferenceSession(weights=pack_blob(mossformer.model_tensors(fused, scalars, W)), metadata=meta) as sess:
        assert sess.frames == mossformer.frames_of(W)
        out, _ = sess.process(pcm[None], want_f32=True)
    out = out.reshape(2, -1)
    assert out.shape == ref.shape
    d = out.astype(np.int32) - ref.astype(np.int32)
    assert np.abs(d).max() <= 2 and (d != 0).mean() < 0.05, (np.abs(d).max(), (d != 0).mean())
    with pytest.raises(Exception):                         # a folded manifest is static by definition (:97)
        mossformer.metadata(3 * 2408, use_batch_fold=True, dynamic_axes=True)
