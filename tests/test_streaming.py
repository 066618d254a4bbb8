"""Stateful streaming over the GTCRN path (include/ade.h ade_stream_*).

The contract: pushing a long signal in pieces equals the reference's stateless graph applied to the WHOLE signal in one call, one hop
(256 samples) later, with no whole-call DC removal.  The oracle restates the reference including its DC removal, so the test signals
are built with an exactly zero integer sum: the oracle's mean is then exactly 0 and both sides compute the same thing.
"""
import numpy as np
import pytest

from ade_testlib import golden_blob, hipsim_library, make_session
from audio_denoiser_onnx_amd.session import StreamingSession
from oracle_lib import GtcrnOracle

HOP = 256


def zero_sum_signal(rng, n):
    t = np.arange(n) / 16000.0
    x = 3000.0 * np.sin(2 * np.pi * 220.0 * t + rng.uniform(0, 6)) * (0.6 + 0.4 * np.sin(2 * np.pi * 3.0 * t)) + 800.0 * rng.standard_normal(n)
    x = np.round(x).astype(np.int64)
    x -= int(x.sum()) // n
    rem = int(x.sum())                                  # 0 <= rem < n: take it off one LSB at a time
    x[:rem] -= 1                                        # exact zero integer sum -> the reference's DC term is exactly 0
    assert x.sum() == 0 and np.abs(x).max() < 32768
    return x.astype(np.int16)


def run_stream(sess, signals, frames_per_push):
    P = frames_per_push * HOP
    n_push = signals.shape[1] // P
    with StreamingSession(sess, signals.shape[0], frames_per_push) as st:
        parts = [st.push(signals[:, i * P:(i + 1) * P], want_f32=True) for i in range(n_push)]
        parts.append(st.flush(want_f32=True))              # the last hop: the frame past the end, reflected
        with pytest.raises(ValueError):
            st.push(signals[:, :P])                        # a flushed stream must be reset first
        pcm = np.concatenate([p[0] for p in parts], axis=1)
        f32 = np.concatenate([p[1] for p in parts], axis=1)
        # a reset stream starts over: same first push again
        st.reset()
        again = st.push(signals[:, :P])
    assert np.array_equal(again, pcm[:, :P])
    return pcm, f32


def check_equivalence(library, frames_per_push, n_frames=40, seed=0):
    rng = np.random.default_rng(seed)
    L = n_frames * HOP                                   # one-shot call: T = n_frames + 1 frames, the last one reflected at the end
    signals = np.stack([zero_sum_signal(rng, L), zero_sum_signal(rng, L)])
    sess = make_session(library, seed=seed, length=L)
    pcm, f32 = run_stream(sess, signals, frames_per_push)
    o = GtcrnOracle(golden_blob(seed), L)
    o.set_exact_dft(True)
    opcm, of32 = o.process(signals)
    assert pcm.shape[1] == L + HOP                                                   # pushes + flush = the whole one-shot output, one hop later
    assert not pcm[:, :HOP].any() and not f32[:, :HOP].any()                         # the stream's first hop is blank
    assert np.abs(f32[:, HOP:] - of32).max() <= 1e-5
    assert np.abs(pcm[:, HOP:].astype(np.int32) - opcm.astype(np.int32)).max() <= 1
    return sess, signals, pcm


@pytest.mark.hipsim
def test_hipsim_streaming_equals_one_shot():
    """CPU (host simulator of the same kernels): 5 pushes of 8 frames, and 20 pushes of 2 frames (shorter than every conv history)."""
    lib = hipsim_library()
    check_equivalence(lib, 8)
    check_equivalence(lib, 2, n_frames=24, seed=1)


@pytest.mark.gpu
def test_gpu_streaming_equals_one_shot():
    sess, signals, pcm = check_equivalence(None, 8)
    check_equivalence(None, 5, seed=2)
    check_equivalence(None, 2, n_frames=24, seed=1)
    # the push size does not change the stream: 8-frame pushes == 4-frame pushes, bit for bit
    pcm4, _ = run_stream(sess, signals, 4)
    assert np.array_equal(pcm4, pcm)


def fused_vs_multikernel(library, frames_per_push, n_push=4):
    """A push as ONE launch of the chunk kernel (carried exchange slot) vs the 48-launch multi-kernel push (option fused = 0 when the stream is created): the same stream."""
    rng = np.random.default_rng(7)
    P = frames_per_push * HOP
    sig = np.stack([zero_sum_signal(rng, n_push * P) for _ in range(3)])
    outs = []
    for fused in ("1", "0"):
        sess = make_session(library, seed=2, length=16000)
        sess.set_option("fused", fused)
        with StreamingSession(sess, 3, frames_per_push) as st:
            parts = [st.push(sig[:, i * P:(i + 1) * P], want_f32=True) for i in range(n_push)] + [st.flush(want_f32=True)]
        outs.append((np.concatenate([p[0] for p in parts], axis=1), np.concatenate([p[1] for p in parts], axis=1)))
    assert np.abs(outs[0][1] - outs[1][1]).max() <= 2e-5
    assert np.abs(outs[0][0].astype(np.int32) - outs[1][0].astype(np.int32)).max() <= 1
    assert not np.array_equal(outs[0][1], outs[1][1])          # (two implementations of the same arithmetic: if they were identical the option did nothing)


@pytest.mark.hipsim
def test_hipsim_fused_push_equals_multikernel_push():
    fused_vs_multikernel(hipsim_library(), 3, n_push=3)


@pytest.mark.gpu
def test_gpu_fused_push_equals_multikernel_push_and_segments():
    fused_vs_multikernel(None, 2)
    fused_vs_multikernel(None, 40)             # 40 frames per push: three 16-frame segments per stream and launch, the last one carrying into the next push
    # many streams, more workgroups than the chip holds: 3000 streams x 2 frames
    rng = np.random.default_rng(3)
    sig = np.stack([zero_sum_signal(rng, 4 * 2 * HOP)] * 2 + [zero_sum_signal(rng, 4 * 2 * HOP)])
    big = np.concatenate([sig] * 1000)
    sess = make_session(None, seed=0)
    with StreamingSession(sess, 3000, 2) as st:
        a = np.concatenate([st.push(big[:, i * 512:(i + 1) * 512]) for i in range(4)], axis=1)
    with StreamingSession(sess, 3, 2) as st:
        b = np.concatenate([st.push(sig[:, i * 512:(i + 1) * 512]) for i in range(4)], axis=1)
    assert np.array_equal(a[:3], b) and np.array_equal(a[2997:], b)


@pytest.mark.gpu
def test_gpu_streaming_rejects_other_families_and_bad_sizes():
    sess = make_session(None, seed=0)
    with pytest.raises(ValueError):
        StreamingSession(sess, 1, 1)                     # the first push must hold the 257 samples the head reflection reads
    with pytest.raises(ValueError):
        with StreamingSession(sess, 2, 4) as st:
            st.push(np.zeros((2, 100), np.int16))


@pytest.mark.gpu
def test_gpu_file_driver_streaming_mode_has_no_slice_edges():
    """``inference_gtcrn --stream``: a 2.3 s file through one stream equals the one-shot oracle on the (zero-padded) whole file."""
    from audio_denoiser_onnx_amd import inference_gtcrn as drv
    rng = np.random.default_rng(4)
    n = 36000                                            # not a whole number of pushes: the driver zero-pads to 5 pushes of 32 frames = 40960
    audio = zero_sum_signal(rng, n)                      # integer sum 0 -> the oracle's whole-call DC term on the padded file is exactly 0
    padded = np.zeros(40960, np.int16)
    padded[:n] = audio
    sess = make_session(None, seed=0)                    # the session's own static length does not matter for streaming
    out = drv.denoise_streaming(sess, audio, frames_per_push=32)
    assert out.shape == audio.shape and out.dtype == np.int16
    o = GtcrnOracle(golden_blob(0), 40960)
    o.set_exact_dft(True)
    opcm, _ = o.process(padded[None])
    assert np.abs(out.astype(np.int32) - opcm[0, :len(audio)].astype(np.int32)).max() <= 1
