"""The fused path's chunk SEGMENTS (csrc/ade_internal.h: geometries): a chunk walked by several workgroups, each owning a run of
consecutive frames and continuing its predecessor's recurrences, must give the very bits a single workgroup gives.

CPU part (host simulator, test-only build of the same csrc/*.hip): data flow of the hand-offs -- depthwise-convolution partial sums, TRA and
inter-frame GRU states, overlap-add carry -- for 2, 3 and 4 segments.  GPU part: the same on the MI355X at the benchmark batch (every CU holds two
workgroups that exchange through device memory), with more workgroups than the chip holds at once, and the sticky time-out word."""
import re

import numpy as np
import pytest

from ade_testlib import golden_blob, golden_inputs, hipsim_library, make_session
from audio_denoiser_onnx_amd.synth import synth_batch
from oracle_lib import GtcrnOracle

TAPS = ("spec", "e0", "e1", "x_e2", "x_e3", "x_e4", "dp1", "dp2", "x_d0", "x_d1", "x_d2")


def run(sess, x, geometry, single="1"):
    sess.set_option("geometry", geometry)
    sess.set_option("single_launch", single)
    sess.set_option("full_taps", "1")                      # (the single launch otherwise keeps channels 0-7 of x_d0 / x_d1 / dp2 in LDS: test_gpu_lean_stores)
    pcm, f32 = sess.process(x, want_f32=True)
    taps = {n: sess.tap(n, x.shape[0] * sess.frames * 65 * 16).copy() for n in TAPS}
    assert sess.tap("xchg_error", 1)[0] == 0.0
    return pcm, f32, taps


def assert_same(a, b, what):
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), f"{what}: output differs, max |d f32| = {np.abs(a[1] - b[1]).max():.3g}"
    for n in TAPS:
        assert np.array_equal(a[2][n], b[2][n]), f"{what}: tap {n} differs"


@pytest.mark.hipsim
def test_hipsim_two_segments_equal_one_workgroup():
    """T = 63: geometry 1 = 32 + 31 frames on two 512-thread workgroups; geometry 0 = one 1024-thread workgroup."""
    lib = hipsim_library()
    ins = golden_inputs()
    x = np.stack([ins["wav0"], ins["randn"]])
    sess = make_session(lib, seed=0)
    whole = run(sess, x, "0")
    assert_same(whole, run(sess, x, "1"), "two segments, one launch")
    assert_same(whole, run(sess, x, "1", single="0"), "two segments, one launch per stage")
    assert_same(whole, run(sess, x, "2"), "four segments of 16 frames (256-thread workgroups), one launch")
    opcm, of32 = GtcrnOracle(golden_blob(0), 16000).process(x)
    assert np.abs(whole[1] - of32).max() <= 1e-4 and np.abs(whole[0].astype(np.int32) - opcm.astype(np.int32)).max() <= 1


@pytest.mark.hipsim
@pytest.mark.parametrize("length", [16384, 32000])
def test_hipsim_three_and_four_segments_vs_oracle(length):
    """T = 65 -> 22 + 22 + 21 frames (geometry 1) / 33 + 32 (geometry 0); T = 126 -> 4 x 32 / 2 x 63: bit-equal to each other, and the oracle's."""
    lib = hipsim_library()
    x = synth_batch(1, length)
    sess = make_session(lib, seed=2, length=length)
    a, b = run(sess, x, "0"), run(sess, x, "1")
    assert_same(a, b, f"length {length}")
    if length == 16384:
        assert_same(a, run(sess, x, "2"), f"length {length}: five segments of 13 frames")
    opcm, of32 = GtcrnOracle(golden_blob(2), length).process(x)
    assert np.abs(a[1] - of32).max() <= 1e-4 and np.abs(a[0].astype(np.int32) - opcm.astype(np.int32)).max() <= 1


@pytest.mark.hipsim
def test_hipsim_segments_shorter_than_the_convolution_history():
    """T = 17 -> 9 + 8 frames under geometry 2: the dilation-5 blocks reach 10 frames back, over a whole segment -- the partial sums are passed on."""
    lib = hipsim_library()
    x = synth_batch(2, 4096)
    sess = make_session(lib, seed=1, length=4096)
    assert sess.frames == 17
    a = run(sess, x, "0")
    assert_same(a, run(sess, x, "2"), "T = 17, segments of 9 and 8 frames")
    opcm, of32 = GtcrnOracle(golden_blob(1), 4096).process(x)
    assert np.abs(a[1] - of32).max() <= 1e-4 and np.abs(a[0].astype(np.int32) - opcm.astype(np.int32)).max() <= 1


@pytest.mark.gpu
def test_gpu_segments_bit_equal_at_benchmark_batch():
    """256 x 1 s: geometry 1 (512 workgroups, two per CU, hand-offs through device memory) == geometry 0 (one workgroup per chunk), every tap."""
    x = synth_batch(256)
    sess = make_session(None, seed=0)
    whole = run(sess, x, "0")
    for rep in range(3):                                   # the hand-offs race differently every launch
        assert_same(whole, run(sess, x, "1"), f"256 chunks, two segments, launch {rep}")
        assert_same(whole, run(sess, x, "2"), f"256 chunks, four segments, launch {rep}")
    assert_same(whole, run(sess, x, "1", single="0"), "256 chunks, one launch per stage")
    assert_same(whole, run(sess, x, "2", single="0"), "256 chunks, four segments, one launch per stage")


@pytest.mark.gpu
@pytest.mark.parametrize("batch", [1, 3, 300, 700])
def test_gpu_segments_more_workgroups_than_the_chip_holds(batch):
    """600 / 1400 workgroups on 512 slots: later segments start while earlier chunks are still running, or long after their predecessor ended."""
    x = synth_batch(batch)
    sess = make_session(None, seed=1)
    whole = run(sess, x, "0")
    assert_same(whole, run(sess, x, "1"), f"batch {batch}, two segments")
    assert_same(whole, run(sess, x, "2"), f"batch {batch}, four segments")


@pytest.mark.gpu
@pytest.mark.parametrize("length", [16384, 20000, 32000, 60000])
def test_gpu_three_to_eight_segments(length):
    """T = 65, 79, 126, 235 frames: 3, 3, 4, 8 segments of geometry 1 (5, 5, 8, - of geometry 2) against 2 - 4 segments of geometry 0 and the multi-kernel path."""
    x = synth_batch(40, length)
    sess = make_session(None, seed=2, length=length)
    a, b = run(sess, x, "0"), run(sess, x, "1")
    assert_same(a, b, f"length {length}")
    if length <= 32000:
        assert_same(a, run(sess, x, "2"), f"length {length}, 16-frame segments")
    sess.set_option("fused", "0")
    m_pcm, m_f32 = sess.process(x, want_f32=True)
    sess.set_option("fused", "1")
    assert np.abs(a[1] - m_f32).max() <= 2e-5 and np.abs(a[0].astype(np.int32) - m_pcm.astype(np.int32)).max() <= 1
    o = GtcrnOracle(golden_blob(2), length)
    opcm, of32 = o.process(x[:4], threads=4)
    assert np.abs(a[1][:4] - of32).max() <= 1e-4


@pytest.mark.gpu
def test_gpu_priority_options_and_every_segments_clocks():
    """The scheduling knobs do not touch a bit ("seg_prio" 0 .. 4: base wave priority by segment; the recurrences always run at priority 3), values outside the range are
    refused, and the clock build stamps EVERY segment of chunk 0 (four of them in geometry 2: one 640-slot set each)."""
    x = synth_batch(64)
    sess = make_session(None, seed=0)
    ref = run(sess, x, "2")
    for level in "1234":
        sess.set_option("seg_prio", level)
        assert_same(ref, run(sess, x, "2"), f"seg_prio {level}")
    with pytest.raises(Exception):
        sess.set_option("seg_prio", "7")
    sess.set_option("seg_prio", "0")
    sess.profile(3)
    sess.process(x)
    sess.process(x)
    clocks = sess.tap("phase_clock_abs", 8 * 640).reshape(-1, 10, 64)
    sess.profile(0)
    assert clocks.shape[0] == 8
    for seg in range(4):
        # the back stage of every segment ended after its front stage began; the zero of the clocks is segment 0's entry, and a workgroup of another segment -- on another
        # XCD -- may enter a few ticks (10 ns each) BEFORE block 0 does: HIP promises no dispatch order (seen once: -4 ticks)
        assert clocks[seg, 9, 53] > clocks[seg, 0, 32] >= -100, (seg, clocks[seg, 0, 32:37].tolist(), clocks[seg, 9, 48:54].tolist())
    assert (clocks[4:] == -1).all()                                        # no fifth segment at 63 frames
    ends = [clocks[seg, 9, 53] for seg in range(4)]
    assert ends == sorted(ends)                                            # a segment cannot finish before its predecessor (its overlap-add carry comes from it)


def _forced_timeout_contract(sess, x, device_path=None):
    """Option "xchg_withhold" makes block 0 raise its hand-off flags where nobody polls; its successor's bounded waits ("xwait_ms") must give up and the CALL must fail
    with ADE_ERR_DEVICE naming the segment -- no PCM -- and the next call, hook off, must be clean and bit-equal to the whole-chunk geometry."""
    from audio_denoiser_onnx_amd._lib import AdeDeviceError
    ref = run(sess, x, "0")
    sess.set_option("geometry", "2")
    sess.set_option("xwait_ms", "2")
    sess.set_option("xchg_withhold", "1")
    out = np.full((x.shape[0], sess.row_out), 12345, np.int16)
    with pytest.raises(AdeDeviceError, match=r"segment [123] of chunk 0 timed out .* depthwise-convolution history"):
        sess.process_into(x, out)
    assert (out == 12345).all(), "a failed call must not hand out PCM"
    if device_path is not None:                      # the caller-stream entry cannot synchronise: the NEXT call on the handle reports it
        d_in, d_out, stream = device_path(x)
        sess.run_device(d_in, d_out, stream=stream)  # enqueued, returns ADE_OK
        import torch
        torch.cuda.synchronize()                     # (the caller's own synchronisation: the launch has failed by now)
        with pytest.raises(AdeDeviceError, match=r"earlier call on a caller-provided stream"):
            sess.run_device(d_in, d_out)
        with pytest.raises(AdeDeviceError, match=r"segment [123] of chunk 0"):      # engine-stream device entry: synchronises, reports at once
            sess.run_device(d_in, d_out)
    sess.set_option("xchg_withhold", "0")
    sess.set_option("xwait_ms", "200")
    assert sess.tap("xchg_error", 1)[0] == 0.0       # reported once, then cleared together with every flag
    for rep in range(2):
        assert_same(ref, run(sess, x, "2"), f"call {rep} after the forced time-out")


@pytest.mark.hipsim
def test_hipsim_withheld_flag_fails_the_call():
    lib = hipsim_library()
    ins = golden_inputs()
    _forced_timeout_contract(make_session(lib, seed=0), np.stack([ins["wav0"]]))


@pytest.mark.gpu
def test_gpu_withheld_flag_fails_the_call_on_every_entry_point():
    import torch
    x = synth_batch(5)
    sess = make_session(None, seed=0)
    keep = []

    def device_path(x):
        d_in = torch.from_numpy(x).cuda()
        d_out = torch.zeros((x.shape[0], sess.row_out), dtype=torch.int16, device="cuda")
        side = torch.cuda.Stream()                     # (the default stream's handle is 0 = "run synchronously on the engine's stream")
        side.wait_stream(torch.cuda.current_stream())
        keep.append(side)
        return d_in, d_out, side.cuda_stream
    _forced_timeout_contract(sess, x, device_path)


@pytest.mark.gpu
def test_gpu_lean_stores():
    """The shipped single launch does not write channels 0-7 of x_d0 / x_d1 / dp2 (their only reader is the next block, through LDS): same PCM / waveform bit for bit as
    with option full_taps, the stored half (channels 8-15) equal, every other inter-stage tensor equal."""
    x = synth_batch(37)
    sess = make_session(None, seed=2)
    for geometry in "210":
        full = run(sess, x, geometry)
        sess.set_option("full_taps", "0")
        pcm, f32 = sess.process(x, want_f32=True)
        assert np.array_equal(pcm, full[0]) and np.array_equal(f32, full[1]), f"geometry {geometry}"
        for name in TAPS:
            got = sess.tap(name, x.shape[0] * sess.frames * 65 * 16)
            if name in ("x_d0", "x_d1", "dp2"):
                got, want = got.reshape(-1, 16)[:, 8:], full[2][name].reshape(-1, 16)[:, 8:]
            else:
                want = full[2][name]
            assert np.array_equal(got, want), f"geometry {geometry}: tap {name}"


@pytest.mark.gpu
def test_gpu_timed_out_call_is_rerun_without_handoffs_by_the_host_entry():
    """A bounded inter-workgroup wait that gives up on a VALID call (a pre-empted or profiled GPU; forced here by a 1 us bound, which no successor segment can meet) must not
    fail ade_process: the synchronous host entry re-runs the call once on the path without hand-offs -- the same bits -- and says so; with "xwait_retry" = "0", and on the
    device-pointer entries, the failure stands."""
    import torch
    from audio_denoiser_onnx_amd._lib import AdeDeviceError
    x = synth_batch(9)
    sess = make_session(None, seed=0)
    ref = run(sess, x, "0")
    sess.set_option("geometry", "2")
    sess.set_option("xwait_ms", "0.001")
    out = np.zeros((x.shape[0], sess.row_out), np.int16)
    before = sess.tap("xwait_retries", 1)[0]
    sess.process_into(x, out)
    assert np.array_equal(out, ref[0]), "the re-run must produce the whole-chunk geometry's bits"
    assert sess.tap("xwait_retries", 1)[0] == before + 1
    sess.set_option("xwait_retry", "0")
    with pytest.raises(AdeDeviceError, match=r"timed out"):
        sess.process_into(x, out)
    # two sub-batches on two streams (option host_split): each launch reports into its own word, and the message names the segment and the chunk of the whole batch
    # (decoded with the batch's size, block 5 of the 5-chunk launch read "segment 0 of chunk 5" -- a segment that waits for nobody)
    sess.set_option("host_split", "2")
    for rep in range(3):
        with pytest.raises(AdeDeviceError, match=r"timed out") as ei:
            sess.process_into(x, out)
        m = re.search(r"segment (\d+) of chunk (\d+)", str(ei.value))
        assert m and 1 <= int(m.group(1)) < 4 and 0 <= int(m.group(2)) < x.shape[0], str(ei.value)
    sess.set_option("host_split", "0")
    sess.set_option("xwait_retry", "1")
    d_in, d_out = torch.from_numpy(x).cuda(), torch.zeros((x.shape[0], sess.row_out), dtype=torch.int16, device="cuda")
    with pytest.raises(AdeDeviceError, match=r"timed out"):
        sess.run_device(d_in, d_out)                 # device-pointer entry: reports, does not retry
    sess.set_option("xwait_ms", "200")
    for rep in range(2):
        assert_same(ref, run(sess, x, "2"), f"call {rep} after the forced time-outs")
