"""Size-independent properties of the GEMM-family engines at their PRODUCTION shapes (random-init weights of the full architectures:
no oracle is affordable here, the fixtures pin the arithmetic at small shapes): batch rows are independent calls bit for bit, row order
does not matter, outputs are finite and carry signal, silence stays bounded."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _signal(rng, shape, amp):
    t = np.arange(shape[-1]) / 16000.0
    x = amp * np.sin(2 * np.pi * 310.0 * t) * (0.5 + 0.5 * np.sin(2 * np.pi * 2.0 * t)) + 0.2 * amp * rng.standard_normal(shape)
    return np.clip(x, -32768, 32767).astype(np.int16)


def test_mossformer_24_layers_one_and_a_half_second_windows():
    from audio_denoiser_onnx_amd import mossformer
    from audio_denoiser_onnx_amd.session import InferenceSession
    from audio_denoiser_onnx_amd.weights import pack_blob
    W = 24000                                                     # 2999 frames: 12 FLASH groups, the last one padded
    frames = mossformer.frames_of(W)
    fused = {n: mossformer.synthetic_tensor(n, s, sc, frames) for n, s, sc in mossformer.synthetic_spec(24)}
    scalars = dict(mossformer.DEFAULT_SCALARS, fs_front_alpha=[0.25] * 24)
    rng = np.random.default_rng(0)
    rows = np.stack([_signal(rng, (W,), 5000.0), _signal(rng, (W,), 800.0), np.zeros(W, np.int16)])
    with InferenceSession(weights=pack_blob(mossformer.model_tensors(fused, scalars, W)), metadata=mossformer.metadata(W)) as sess:
        assert sess.frames == frames
        a = sess.run(None, {"mix_audio": rows[:, None]})
        b = sess.run(None, {"mix_audio": rows[::-1][:, None].copy()})
        one = sess.run(None, {"mix_audio": rows[1:2, None]})
    for spk in range(2):
        assert np.array_equal(a[spk][::-1], b[spk])              # permutation of the rows = permutation of the results
        assert np.array_equal(one[spk][0], a[spk][1])             # a row alone = the same row inside a batch
        assert np.abs(a[spk][0]).max() > 200 and not a[spk][2].any()     # signal comes out; a silent window stays silent


def test_melband_depth_6_fold_window_and_eight_second_clip():
    from audio_denoiser_onnx_amd import melband, weightgen
    from audio_denoiser_onnx_amd.session import InferenceSession
    from audio_denoiser_onnx_amd.weights import pack_blob
    blob = pack_blob(melband.model_tensors(weightgen.materialise(melband.synthetic_spec(6))))
    rng = np.random.default_rng(1)
    for L, frames in ((66150, 151), (352800, 801)):               # the 1.5 s fold window; BASELINE's 8 s clip (801 keys stream through LDS)
        rows = np.stack([np.stack((_signal(rng, (L,), 6000.0), _signal(rng, (L,), 4000.0))), np.stack((_signal(rng, (L,), 500.0), np.zeros(L, np.int16)))])
        with InferenceSession(weights=blob, metadata=melband.metadata(L)) as sess:
            assert sess.frames == frames
            a = sess.run(None, {"noisy_audio": rows})[0]
            b = sess.run(None, {"noisy_audio": rows[::-1].copy()})[0]
            one = sess.run(None, {"noisy_audio": rows[1:2]})[0]
        assert np.array_equal(a[::-1], b) and np.array_equal(one[0], a[1])
        assert np.isfinite(a.astype(np.float64)).all() and np.abs(a[0]).max() > 200


def test_hgtcrn_and_ulunas_256_calls():
    """The two GTCRN-kernel families at batch 256 (H-GTCRN: 2 s stereo calls, the export's default length; UL-UNAS: 1 s): a row inside the
    big batch equals the same row in a batch of three, reversed order gives reversed results, silence stays silent."""
    import os
    from audio_denoiser_onnx_amd import hgtcrn, ulunas
    from audio_denoiser_onnx_amd.session import InferenceSession
    from audio_denoiser_onnx_amd.weights import pack_blob
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    rng = np.random.default_rng(2)
    z = np.load(os.path.join(gold, "hgtcrn_seed0.npz"))
    fused = hgtcrn.fold_state_dict({str(k): z["w:" + str(k)] for k in z["keys"]})
    L = 32000
    rows = np.stack([np.stack((_signal(rng, (L,), 200.0 + 40.0 * i), _signal(rng, (L,), 150.0 + 30.0 * i))) for i in range(256)])
    rows[17] = 0
    with InferenceSession(weights=pack_blob(fused), metadata=hgtcrn.metadata(L)) as sess:
        assert sess.frames == 126 and sess.out_len == L
        a = sess.run(None, {"noisy_audio": rows})[0][:, 0]
        b = sess.run(None, {"noisy_audio": rows[::-1].copy()})[0][:, 0]
        few = sess.run(None, {"noisy_audio": rows[[200, 17, 3]]})[0][:, 0]
    assert np.array_equal(a[::-1], b) and np.array_equal(few, a[[200, 17, 3]])
    assert not a[17].any() and np.abs(a[200]).max() > 100
    z = np.load(os.path.join(gold, "ulunas_seed0.npz"))
    fused = ulunas.fold_state_dict({str(k): z["w:" + str(k)] for k in z["keys"]})
    rows = np.stack([_signal(rng, (16000,), 300.0 + 50.0 * i) for i in range(256)])
    rows[5] = 0
    with InferenceSession(weights=pack_blob(fused), metadata=ulunas.metadata(16000)) as sess:
        a = sess.run(None, {"noisy_audio": rows[:, None]})[0][:, 0]
        b = sess.run(None, {"noisy_audio": rows[::-1][:, None].copy()})[0][:, 0]
        few = sess.run(None, {"noisy_audio": rows[[250, 5, 0]][:, None]})[0][:, 0]
    assert np.array_equal(a[::-1], b) and np.array_equal(few, a[[250, 5, 0]])
    assert not a[5].any() and np.abs(a[250]).max() > 100
