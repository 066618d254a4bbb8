"""Kernel logic under the host-side HIP simulator vs the oracle (CPU, test-only build of csrc/*.hip).

This does NOT exercise the product library: it compiles the same kernel and engine sources with g++ against
tests/hipsim (fibers per HIP thread) so that indexing / barrier / shuffle data-flow bugs are caught before GPU time
is spent.  The GPU parity tests proper are in test_gpu_parity.py.
"""
import numpy as np
import pytest

from ade_testlib import compare_taps, golden_blob, golden_inputs, hipsim_library, make_session
from oracle_lib import GtcrnOracle

pytestmark = pytest.mark.hipsim


@pytest.fixture(scope="module")
def simlib():
    return hipsim_library()


def test_hipsim_taps_and_waveform(simlib):
    ins = golden_inputs()
    sess = make_session(simlib, seed=0)
    pcm_in = np.stack([ins["randn"], ins["wav0"]])
    lean_pcm, lean_f32 = sess.process(pcm_in, want_f32=True)      # the shipped launch: channels 0-7 of x_d0 / x_d1 / dp2 never leave LDS
    sess.set_option("full_taps", "1")                             # the same launch with every inter-stage tensor stored whole (the taps below read them)
    pcm, f32 = sess.process(pcm_in, want_f32=True)
    assert np.array_equal(pcm, lean_pcm) and np.array_equal(f32, lean_f32)
    o = GtcrnOracle(golden_blob(0), 16000)
    # (1) against the oracle with the reference's own (fp32-angle) DFT table: the documented 1e-4 contract
    opcm, of32 = o.process(pcm_in)
    assert np.abs(f32 - of32).max() <= 1e-4
    assert np.abs(pcm.astype(np.int32) - opcm.astype(np.int32)).max() <= 1
    # (2) against the oracle with exact DFT tables: every tap to fp32 round-off (isolates kernel error)
    o.set_exact_dft(True)
    opcm, of32 = o.process(pcm_in[1:2])
    res = compare_taps(sess, o, batch=2, row=1)
    for name, (err, scale) in res.items():
        assert err <= 2e-6 * max(1.0, scale) + 4e-6, f"{name}: {err:.3e} (scale {scale:.3g})"
    assert np.abs(f32[1] - of32[0]).max() <= 5e-6


def test_hipsim_edge_inputs_and_batch_tail(simlib):
    ins = golden_inputs()
    sess = make_session(simlib, seed=1)
    names = ["zeros", "impulse15999", "dc_min"]          # B=3: frame count not a multiple of the 4-frame workgroup
    pcm_in = np.stack([ins[n] for n in names])
    pcm, f32 = sess.process(pcm_in, want_f32=True)
    o = GtcrnOracle(golden_blob(1), 16000)
    opcm, of32 = o.process(pcm_in)
    assert np.abs(f32 - of32).max() <= 1e-4
    assert np.abs(pcm.astype(np.int32) - opcm.astype(np.int32)).max() <= 1
    assert not pcm[0].any() and not pcm[2].any()        # silence and pure DC come out exactly silent
    empty, _ = sess.process(np.zeros((0, 16000), np.int16))
    assert empty.shape == (0, 15872)


def test_hipsim_batch_fold_fused_window(simlib):
    """USE_BATCH_FOLD with a 1.0 s window (W = 16128 -> T = 64): the single-launch path with a per-CALL DC mean."""
    import os
    from ade_testlib import GOLD, default_meta
    from audio_denoiser_onnx_amd.metadata import build_audio_metadata
    from audio_denoiser_onnx_amd.session import InferenceSession
    g = np.load(os.path.join(GOLD, "gtcrn_seed0_fold.npz"))
    W = 16128
    meta = build_audio_metadata(producer="tests", model_name="GTCRN", task="denoise", model_family="gtcrn",
                                input_audio_length=2 * W, use_batch_fold=True, batch_window_seconds=1.0)
    assert int(meta["fold_window_length"]) == W and int(meta["export_audio_length"]) == 2 * W
    sess = InferenceSession(weights=golden_blob(0), metadata=meta, library=simlib)
    assert (sess.in_len, sess.out_len, sess.frames) == (2 * W, 2 * W, 64)
    pcm_in = g["pcm_in"][:2 * W]
    pcm, f32 = sess.process(pcm_in[None], want_f32=True)
    o = GtcrnOracle(golden_blob(0), W)
    opcm, of32 = o.process_fold(pcm_in, 2)
    assert np.abs(f32 - of32).max() <= 1e-4
    assert np.abs(pcm.astype(np.int32) - opcm.astype(np.int32)).max() <= 1
