"""UL-UNAS: oracle + checkpoint fold pinned to the reference (CPU) and HIP parity through the C ABI (GPU).

Fixture: tests/golden/ulunas_seed0.npz = the reference's own export path (seeded ULUNAS() -> prepare_for_export_ -> ULUNAS_CUSTOM
forward with its STFT_Process) run in the build container (tools/make_golden_ulunas.py); it holds the checkpoint-format state_dict,
three input rows (its test wav, seeded noise, silence), their outputs and the mask of row 0.
"""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(REPO, "oracle"))

from audio_denoiser_onnx_amd import ulunas  # noqa: E402

GOLD = os.path.join(HERE, "golden", "ulunas_seed0.npz")


@pytest.fixture(scope="module")
def fixture():
    z = np.load(GOLD)
    state = {str(k): z["w:" + str(k)] for k in z["keys"]}
    return z, ulunas.fold_state_dict(state)


def _oracle(fused, length=16000, exact_dft=False):
    from ulunas_oracle import UlunasOracle
    return UlunasOracle(fused, ulunas.block_plan(), length, exact_dft)


def test_fold_and_oracle_match_reference_export_path(fixture):
    z, fused = fixture
    o = _oracle(fused)
    out = o.process(z["pcm_in"])
    assert np.abs(o.taps["mask"][0].T - z["mask0"]).max() < 5e-5
    d = out.astype(np.int32) - z["pcm_out"].astype(np.int32)
    assert np.abs(d).max() <= 1 and (d != 0).mean() < 0.01
    assert not out[2].any() and np.abs(z["pcm_out"][1]).max() > 1000          # silence stays silent; the noise row carries signal


def test_erb_matrix_and_plan():
    e = ulunas.erb_matrix()
    assert e.shape == (64, 192) and e.min() >= 0 and np.all((e > 0).sum(axis=0) >= 1)
    plan = ulunas.block_plan()
    assert [p[1] for p in plan] == [0, 2, 1, 2, 1, 1, 2, 1, 2, 0] and plan[-1][3] == 1 and plan[-1][4] == 129 and plan[-1][-1]


def _session(fused, length=16000, library=None):
    from audio_denoiser_onnx_amd.session import InferenceSession
    from audio_denoiser_onnx_amd.weights import pack_blob
    return InferenceSession(weights=pack_blob(fused), metadata=ulunas.metadata(length), library=library)


@pytest.mark.hipsim
def test_hipsim_short_clip_matches_oracle(fixture):
    """The same csrc/ade_ulunas.hip compiled for the host simulator: a 9-frame clip, mask and PCM against the oracle."""
    from ade_testlib import hipsim_library
    z, fused = fixture
    L = 2048
    pcm = np.ascontiguousarray(z["pcm_in"][:2, 3000:3000 + L])
    o = _oracle(fused, L, exact_dft=True)                  # the engine's DFT tables use exactly reduced angles
    want = o.process(pcm)
    with _session(fused, L, hipsim_library()) as sess:
        got = sess.run(None, {"noisy_audio": pcm[:, None]})[0][:, 0]
        mask = sess.tap("mask", 2 * sess.frames * 257).reshape(2, sess.frames, 257)
    assert np.abs(mask - o.taps["mask"]).max() < 2e-4
    assert np.abs(got.astype(np.int32) - want.astype(np.int32)).max() <= 1


@pytest.mark.gpu
def test_gpu_matches_reference_fixture(fixture):
    z, fused = fixture
    with _session(fused) as sess:
        assert sess.frames == 63 and sess.out_len == 15872 and sess.channels == 1
        out = sess.run(None, {"noisy_audio": z["pcm_in"][:, None]})[0][:, 0]
        mask = sess.tap("mask", 3 * 63 * 257).reshape(3, 63, 257)
        one = sess.run(None, {"noisy_audio": z["pcm_in"][1:2, None]})[0][0, 0]
    assert np.abs(mask[0].T - z["mask0"]).max() < 5e-3          # log-power features of near-empty bins amplify the reference's DFT-table error
    d = out.astype(np.int32) - z["pcm_out"].astype(np.int32)
    assert np.abs(d).max() <= 1 and (d != 0).mean() < 0.02, (np.abs(d).max(), (d != 0).mean())
    assert not out[2].any() and np.array_equal(one, out[1])


@pytest.mark.gpu
def test_gpu_two_second_clips_match_oracle(fixture):
    """A length the fixture does not cover (T = 126): HIP vs oracle on seeded inputs."""
    _, fused = fixture
    rng = np.random.default_rng(3)
    L = 32000
    pcm = (rng.standard_normal((3, L)) * np.array([[300.0], [3000.0], [12000.0]])).astype(np.int16)
    want = _oracle(fused, L, exact_dft=True).process(pcm)
    with _session(fused, L) as sess:
        got = sess.run(None, {"noisy_audio": pcm[:, None]})[0][:, 0]
    d = got.astype(np.int32) - want.astype(np.int32)
    assert np.abs(d).max() <= 2 and (d != 0).mean() < 0.02, (np.abs(d).max(), (d != 0).mean())


def test_upstream_checkpoint_keys_convert_and_export(fixture, tmp_path):
    """Upstream (nn.Sequential) key names -> optimised names -> the same folded tensors; export writes blob + manifest."""
    from audio_denoiser_onnx_amd import export
    from audio_denoiser_onnx_amd.weights import load_blob
    z, fused = fixture
    state = {str(k): z["w:" + str(k)] for k in z["keys"]}
    back = {"conv.": "ops.1.", "bn.": "ops.2.", "act.": "ops.3.", "ctfa.": "ops.4."}
    upstream = {}
    for k, v in state.items():
        if k.startswith("encoder.en_convs.0.") or k.startswith("decoder.de_convs.4."):           # the two XConvBlocks
            head, rest = k[:19], k[19:]
            for new, old in back.items():
                if rest.startswith(new):
                    k = head + old + rest[len(new):]
                    break
            if k.endswith(("affine_weight", "affine_bias")):
                v = v[0, :, 0, :]
            elif k.endswith("slope_weight"):
                v = v[0, :, 0, :]
        upstream[k] = v
    assert any(".ops.1." in k for k in upstream)
    again = ulunas.fold_state_dict(ulunas.convert_state_dict(upstream))
    assert set(again) == set(fused) and all(np.array_equal(again[k], fused[k]) for k in fused)
    np.savez(tmp_path / "ckpt.npz", **upstream)
    path = export.export_ulunas(tmp_path / "ckpt.npz", tmp_path / "out", 16000)
    blob = load_blob(path)
    assert set(blob) == set(fused) and (tmp_path / "out" / "UL_UNAS_Metadata.json").exists()


def _fold_oracle_out(fused, zf):
    W = int(zf["fold_window_length"])
    rows = np.ascontiguousarray(zf["pcm_in"].reshape(-1, W))
    return _oracle(fused, W).process(rows).reshape(-1)


def test_oracle_batch_fold_matches_reference_forward(fixture):
    """USE_BATCH_FOLD in the reference (3 windows of 4096 samples) = the oracle on the windows as independent clips, stitched."""
    zf = np.load(os.path.join(HERE, "golden", "ulunas_seed0_fold.npz"))
    d = _fold_oracle_out(fixture[1], zf).astype(np.int32) - zf["pcm_out"].astype(np.int32)
    assert np.abs(d).max() <= 1 and (d != 0).mean() < 0.01


@pytest.mark.gpu
def test_gpu_batch_fold_matches_reference_fixture(fixture):
    from audio_denoiser_onnx_amd.session import InferenceSession
    from audio_denoiser_onnx_amd.weights import pack_blob
    zf = np.load(os.path.join(HERE, "golden", "ulunas_seed0_fold.npz"))
    meta = ulunas.metadata(int(zf["input_audio_length"]), use_batch_fold=True, batch_window_seconds=float(zf["batch_window_seconds"]))
    assert int(meta["fold_window_length"]) == int(zf["fold_window_length"]) and int(meta["export_audio_length"]) == zf["pcm_in"].shape[0]
    with InferenceSession(weights=pack_blob(fixture[1]), metadata=meta) as sess:
        assert sess.in_len == 12288 and sess.out_len == 12288 and sess.frames == 17
        out = sess.run(None, {"noisy_audio": np.stack((zf["pcm_in"], zf["pcm_in"]))[:, None]})[0][:, 0]
    d = out[0].astype(np.int32) - zf["pcm_out"].astype(np.int32)
    assert np.abs(d).max() <= 1 and (d != 0).mean() < 0.02 and np.array_equal(out[0], out[1])


def _variants_identical(fused, pcm, library, rows_tap):
    """Row groups on side streams (round 5, ADE_ULU_GROUPS; read when the engine is created) are the same launches on the same rows: PCM and mask must come back
    bit-identical to the one-stream form."""
    outs = []
    for groups in ("1", "2", "3", "4"):
        os.environ["ADE_ULU_GROUPS"] = groups
        try:
            with _session(fused, pcm.shape[1], library) as sess:
                pcm_out = sess.run(None, {"noisy_audio": pcm[:, None]})[0][:, 0].copy()
                outs.append((pcm_out, sess.tap("mask", rows_tap * sess.frames * 257).copy()))
        finally:
            del os.environ["ADE_ULU_GROUPS"]
    for k in range(1, len(outs)):
        assert np.array_equal(outs[0][0], outs[k][0]) and np.array_equal(outs[0][1], outs[k][1]), f"variant {k}"
    assert outs[0][0].any()


@pytest.mark.hipsim
def test_hipsim_row_groups_are_bit_identical(fixture):
    from ade_testlib import hipsim_library
    z, fused = fixture
    pcm = np.ascontiguousarray(z["pcm_in"][:3, 3000:3000 + 2048])
    _variants_identical(fused, pcm, hipsim_library(), 3)


@pytest.mark.gpu
def test_gpu_row_groups_are_bit_identical(fixture):
    _, fused = fixture
    rng = np.random.default_rng(11)
    pcm = (rng.standard_normal((70, 16000)) * rng.uniform(100.0, 9000.0, (70, 1))).astype(np.int16)       # 70 rows: row groups of 35 / 24 + 24 + 22
    _variants_identical(fused, pcm, None, 70)


@pytest.mark.gpu
def test_gpu_pipelined_entry_on_a_sub_engine(fixture):
    """ade_submit / ade_wait sit on every family's run(): three submissions in flight of 40 rows each (two row groups on side streams inside each) equal ade_process."""
    _, fused = fixture
    rng = np.random.default_rng(5)
    batches = [(rng.standard_normal((40, 16000)) * 4000.0).astype(np.int16) for _ in range(7)]
    with _session(fused) as sess:
        refs = [sess.process(b)[0] for b in batches]
        outs = [np.empty((40, sess.row_out), np.int16) for _ in batches]
        tickets = []
        for b, o in zip(batches, outs):
            if len(tickets) >= 3:
                sess.wait(tickets.pop(0))
            tickets.append(sess.submit(b, o))
        for t in tickets:
            sess.wait(t)
        assert all(np.array_equal(o, r) for o, r in zip(outs, refs)) and outs[3].any()
