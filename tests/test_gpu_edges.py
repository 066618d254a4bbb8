"""GPU edge cases of the chunk path vs the oracle: chunk lengths around the fused/multi-kernel switch (T = 64 / 65),
a very short chunk, batch sizes 0 / 1 / > #CUs, and the file driver end to end (ragged tail, zero padding)."""
import numpy as np
import pytest

from ade_testlib import golden_blob, make_session
from audio_denoiser_onnx_amd.inference_gtcrn import denoise, plan_slices
from audio_denoiser_onnx_amd.synth import synth_batch, synth_chunk
from oracle_lib import GtcrnOracle

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("length", [1024, 4000, 16128, 16383, 16384, 20000])
def test_chunk_lengths_vs_oracle(length):
    """T = L//256 + 1: 5, 16, 64 (largest fused), 64, 65 (first multi-kernel), 79 frames; L not a multiple of the hop."""
    sess = make_session(None, seed=2, length=length)
    assert sess.frames == length // 256 + 1 and sess.out_len == 256 * (length // 256)
    x = synth_batch(3, length)
    pcm, f32 = sess.process(x, want_f32=True)
    o = GtcrnOracle(golden_blob(2), length)
    opcm, of32 = o.process(x, threads=3)
    assert np.abs(f32 - of32).max() <= 1e-4, length
    assert np.abs(pcm.astype(np.int32) - opcm.astype(np.int32)).max() <= 1, length


def test_batch_sizes():
    sess = make_session(None, seed=1)
    empty, _ = sess.process(np.zeros((0, 16000), np.int16))
    assert empty.shape == (0, 15872)
    x = synth_batch(300)                       # more chunks than CUs: the grid wraps, rows stay independent
    big, _ = sess.process(x)
    one, _ = sess.process(x[299:300])
    assert np.array_equal(big[299], one[0])
    first, _ = sess.process(x[:1])
    assert np.array_equal(big[0], first[0])
    with pytest.raises(ValueError):
        sess.process(np.zeros((2, 15999), np.int16))          # wrong static length (validate: ValueError in the reference)
    with pytest.raises(ValueError):
        sess.run(None, {"noisy_audio": np.zeros((1, 1, 16000), np.float32)})


def test_file_driver_ragged_tail_matches_per_slice_oracle():
    """denoise(): slices at stride out_len, zero-padded tail, concat, trim (Inference_GTCRN_ONNX.py:287-332)."""
    sess = make_session(None, seed=0)
    audio = np.concatenate([synth_chunk(i) for i in range(4)])[:50007]      # 3.125 s: not a multiple of anything
    out = denoise(sess, audio)
    assert out.shape == audio.shape and out.dtype == np.int16
    seq = denoise(sess, audio, sequential=True)                            # the reference's one-slice-per-call loop
    assert np.array_equal(out, seq)
    stride, n, total = plan_slices(len(audio), 16000, 15872)
    padded = np.zeros(total, np.int16)
    padded[:len(audio)] = audio
    o = GtcrnOracle(golden_blob(0), 16000)
    ref = np.concatenate([o.process(padded[i * stride:i * stride + 16000])[0][0] for i in range(n)])[:len(audio)]
    assert np.abs(out.astype(np.int32) - ref.astype(np.int32)).max() <= 1


def test_stitch_device_over_rccl_single_rank_communicator():
    """ade_stitch_device = one ncclAllGather of the rank's int16 rows on the caller's communicator and stream.  A 1-GPU box can only build a world-size-1 communicator
    (the all-gather is then a device copy through RCCL), which still exercises the library lookup, the call signature, the byte count and the stream order."""
    import ctypes as C
    import torch
    try:
        rccl = C.CDLL("librccl.so.1")
    except OSError:
        pytest.skip("librccl.so.1 not installed")
    class UniqueId(C.Structure):                                  # ncclUniqueId is passed BY VALUE
        _fields_ = [("internal", C.c_char * 128)]
    uid = UniqueId()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    torch.cuda.set_device(0)
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    try:
        sess = make_session(None, seed=1)
        x = synth_batch(5)
        d_in = torch.from_numpy(x).cuda()
        d_out = torch.zeros((5, sess.row_out), dtype=torch.int16, device="cuda")
        d_all = torch.zeros((5, sess.row_out), dtype=torch.int16, device="cuda")
        stream = torch.cuda.Stream()
        sess.run_device(d_in, d_out, stream=stream.cuda_stream)
        st = sess._lib.c.ade_stitch_device(sess._h, C.c_void_p(d_out.data_ptr()), 5, C.c_void_p(d_all.data_ptr()), comm, C.c_void_p(stream.cuda_stream))
        sess._lib.check(st, sess._h)
        stream.synchronize()
        want, _ = sess.process(x)
        assert np.array_equal(d_all.cpu().numpy(), want)
    finally:
        rccl.ncclCommDestroy.argtypes = [C.c_void_p]
        rccl.ncclCommDestroy(comm)
