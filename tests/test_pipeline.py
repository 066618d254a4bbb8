"""The pipelined host entry (include/ade.h: ade_submit / ade_wait): the reference's loop of session runs over a file's slices (Inference_GTCRN_ONNX.py:314-333) with
the copy-in of call k + 1, the kernels of call k and the copy-out of call k - 1 overlapped.  Bit-identical to ade_process on the same rows."""
import numpy as np
import pytest

from ade_testlib import golden_inputs, hipsim_library, make_session
from audio_denoiser_onnx_amd.synth import synth_batch


def _run_pipeline(sess, batches, depth=3, want_f32=False):
    outs = [np.empty((len(b), sess.row_out), np.int16) for b in batches]
    f32s = [np.empty((len(b), sess.row_out), np.float32) if want_f32 else None for b in batches]
    tickets = []
    for k, b in enumerate(batches):
        if len(tickets) >= depth:
            sess.wait(tickets.pop(0))
        tickets.append(sess.submit(b, outs[k], f32s[k]))
    for t in tickets:
        sess.wait(t)
    return outs, f32s


@pytest.mark.hipsim
def test_hipsim_submit_wait_equals_process():
    """Host simulator (CPU): three submissions, two in flight, inputs that differ per call; then the ticket rules."""
    sess = make_session(hipsim_library(), seed=0)
    ins = golden_inputs()
    batches = [ins[a][None].copy() for a in ("wav0", "randn", "impulse15999")]          # (one row per call: a simulated row costs ~10 s of CPU)
    outs, f32s = _run_pipeline(sess, batches, want_f32=True)
    for b, o, f in zip(batches, outs, f32s):
        ref, ref32 = sess.process(b, want_f32=True)
        assert np.array_equal(o, ref) and np.array_equal(f, ref32)
    with pytest.raises(ValueError):
        sess.wait(12345)                      # never issued
    o = np.empty((1, sess.row_out), np.int16)
    # the ring refuses one more un-waited ticket than its depth (the slot's status has not been collected); a ticket is waited for exactly once
    sess.set_option("pipe_depth", "2")
    t1 = sess.submit(batches[0], np.empty_like(o)); t2 = sess.submit(batches[1], np.empty_like(o))
    with pytest.raises(ValueError):
        sess.submit(batches[2], np.empty_like(o))
    sess.wait(t1); sess.wait(t2)
    with pytest.raises(ValueError):
        sess.wait(t2)
    # tickets may be waited for in ANY order: waiting for the newest one frees ITS slot for the next submission (the slot is any free one, not ticket % depth)
    o1, o2, o3 = np.empty_like(o), np.empty_like(o), np.empty_like(o)
    t1 = sess.submit(batches[0], o1); t2 = sess.submit(batches[1], o2)
    sess.wait(t2)
    t3 = sess.submit(batches[2], o3)
    # a synchronous call beside submissions in flight completes them first and leaves their statuses for their own ade_wait
    ref, _ = sess.process(batches[0])
    sess.wait(t3); sess.wait(t1)
    assert np.array_equal(o1, outs[0]) and np.array_equal(o2, outs[1]) and np.array_equal(o3, outs[2]) and np.array_equal(ref, outs[0])


def test_process_rows_splits_large_files_into_submissions():
    """inference_gtcrn.process_rows: one process call up to a batch, a three-deep pipeline of submissions beyond it, rows stitched in order (host logic, no engine)."""
    from audio_denoiser_onnx_amd.inference_gtcrn import process_rows

    class Fake:
        row_out = 4

        def __init__(self):
            self.calls, self.pending, self.max_pending = [], {}, 0

        def process(self, rows):
            self.calls.append(("process", len(rows)))
            return rows[:, :4] + 1, None

        def submit(self, rows, out, f32=None):
            t = len(self.calls)
            self.calls.append(("submit", len(rows)))
            self.pending[t] = (rows, out)
            self.max_pending = max(self.max_pending, len(self.pending))
            return t

        def wait(self, t):
            rows, out = self.pending.pop(t)
            out[:] = rows[:, :4] + 1
            return out, None

    rows = np.arange(10 * 6, dtype=np.int16).reshape(10, 6)
    f = Fake()
    assert np.array_equal(process_rows(f, rows, batch=16), rows[:, :4] + 1) and f.calls == [("process", 10)]
    f = Fake()
    assert np.array_equal(process_rows(f, rows, batch=4), rows[:, :4] + 1)
    assert f.calls == [("submit", 4), ("submit", 4), ("submit", 2)] and f.max_pending == 3 and not f.pending


@pytest.mark.gpu
def test_gpu_submit_wait_equals_process_on_changing_inputs():
    """MI355X: 24 back-to-back submissions of 256 x 1 s with inputs that change every call, page-locked and pageable buffers, depth 2 and 3: every batch equals ade_process."""
    import torch
    sess = make_session()
    B, n = 256, 24
    sess.reserve(B)
    pool = synth_batch(B + n)
    batches = [np.ascontiguousarray(pool[k:k + B]) for k in range(n)]          # a sliding window: every call differs from its neighbours
    refs = [sess.process(b)[0] for b in batches[:6]] + [None] * (n - 12) + [sess.process(b)[0] for b in batches[-6:]]
    # pageable numpy buffers
    sess.set_option("pipe_depth", "2")
    outs, _ = _run_pipeline(sess, batches, depth=2)
    for o, r in zip(outs, refs):
        assert r is None or np.array_equal(o, r)
    # page-locked buffers (direct DMA), three in flight, the fp32 waveform as well
    sess.set_option("pipe_depth", "3")
    pin_in = [torch.from_numpy(b).pin_memory() for b in batches[:8]]
    pin_out = [torch.empty((B, sess.row_out), dtype=torch.int16).pin_memory() for _ in range(8)]
    pin_f32 = [torch.empty((B, sess.row_out), dtype=torch.float32).pin_memory() for _ in range(8)]
    tickets = []
    for k in range(8):
        if len(tickets) >= 3:
            sess.wait(tickets.pop(0))
        tickets.append(sess.submit(pin_in[k].numpy(), pin_out[k].numpy(), pin_f32[k].numpy()))
    for t in tickets:
        sess.wait(t)
    for k in range(6):
        ref, ref32 = sess.process(batches[k], want_f32=True)
        assert np.array_equal(pin_out[k].numpy(), ref) and np.array_equal(pin_f32[k].numpy(), ref32)
    # a smaller batch and a mixed sequence with ade_process in between
    t = sess.submit(batches[0][:7], np.empty((7, sess.row_out), np.int16))
    mid = sess.process(batches[1][:3])[0]
    o7, _ = sess.wait(t)
    assert np.array_equal(o7, refs[0][:7]) and np.array_equal(mid, refs[1][:3])


@pytest.mark.gpu
def test_gpu_submit_timeout_fails_at_its_wait():
    """The time-out path of the pipelined entry: a submission whose launch times out fails at ITS ade_wait with ADE_ERR_DEVICE."""
    from audio_denoiser_onnx_amd._lib import AdeDeviceError
    sess = make_session()
    x = synth_batch(8)
    sess.set_option("xwait_ms", "5")
    sess.set_option("xchg_withhold", "1")
    t = sess.submit(x, np.empty((8, sess.row_out), np.int16))
    with pytest.raises(AdeDeviceError):
        sess.wait(t)
    sess.set_option("xchg_withhold", "0")
    sess.set_option("xwait_ms", "200")
    out = np.empty((8, sess.row_out), np.int16)
    sess.wait(sess.submit(x, out))
    assert np.array_equal(out, sess.process(x)[0])
