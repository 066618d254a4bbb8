"""Pin the CPU oracle (oracle/ade_oracle.c) to the reference.

The fixtures were produced by RUNNING the reference's GTCRN_CUSTOM.forward in the build
container (tools/make_golden_gtcrn.py); the reference ships no tests with thresholds for
this path, so these seeded-weights vectors are the pin.
Tolerances: per-tap 2e-5 abs at tap magnitudes <= 50 (4e-7 relative observed),
waveform 1e-5, int16 within 1 LSB (truncating cast, Export_GTCRN.py:690).
"""
import os

import numpy as np
import pytest

from oracle_lib import GtcrnOracle, oracle_istft, oracle_stft

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _blob(seed):
    with open(os.path.join(GOLD, f"gtcrn_seed{seed}.adew"), "rb") as f:
        return f.read()


@pytest.fixture(scope="module")
def inputs():
    return dict(np.load(os.path.join(GOLD, "gtcrn_inputs.npz")))


def test_oracle_taps_match_reference(inputs):
    taps = np.load(os.path.join(GOLD, "gtcrn_seed0_wav0_taps.npz"))
    o = GtcrnOracle(_blob(0), 16000)
    pcm, f32 = o.process(inputs["wav0"])
    checked = 0
    for name in taps.files:
        if name == "pcm_out":
            continue
        ref = taps[name].reshape(-1)
        got = o.tap(name)
        assert got.shape == ref.shape, name
        scale = max(1.0, float(np.abs(ref).max()))
        err = float(np.abs(got - ref).max())
        assert err <= 1e-6 * scale + 2e-6, f"tap {name}: max|d|={err:.3e} (scale {scale:.3g})"
        checked += 1
    assert checked >= 30
    assert np.abs(pcm[0].astype(np.int32) - taps["pcm_out"].astype(np.int32)).max() <= 1


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_oracle_outputs_match_reference(seed, inputs):
    outs = np.load(os.path.join(GOLD, f"gtcrn_seed{seed}_outputs.npz"))
    names = sorted({k.split(".")[0] for k in outs.files})
    o = GtcrnOracle(_blob(seed), 16000)
    batch = np.stack([inputs[n] for n in names])
    # B>1 call == B independent reference calls (per-row DC mean)
    pcm, f32 = o.process(batch, threads=2)
    for i, n in enumerate(names):
        ref_w = outs[f"{n}.wave_f32"]
        ref_p = outs[f"{n}.pcm_out"]
        assert np.abs(f32[i] - ref_w).max() <= 1e-5, n
        assert np.abs(pcm[i].astype(np.int32) - ref_p.astype(np.int32)).max() <= 1, n


def test_oracle_length_32000():
    g = np.load(os.path.join(GOLD, "gtcrn_seed0_len32000.npz"))
    o = GtcrnOracle(_blob(0), 32000)
    assert o.out_len == 32000
    pcm, f32 = o.process(g["pcm_in"])
    assert np.abs(f32[0] - g["wave_f32"]).max() <= 1e-5
    assert np.abs(pcm[0].astype(np.int32) - g["pcm_out"].astype(np.int32)).max() <= 1


def test_oracle_edge_cases(inputs):
    o = GtcrnOracle(_blob(0), 16000)
    pcm, f32 = o.process(np.stack([inputs["zeros"], inputs["dc_min"]]))
    # all-zero and constant inputs are exactly silent after per-call DC removal
    assert not pcm.any()
    assert np.abs(f32).max() == 0.0
    empty_pcm, empty_f32 = o.process(np.zeros((0, 16000), np.int16))
    assert empty_pcm.shape == (0, 15872)


def test_oracle_stft_istft_roundtrip_and_dft_truth():
    rng = np.random.default_rng(1234)
    x = rng.standard_normal((2, 16000)).astype(np.float32)
    for n_fft, win, hop, wt in ((512, 512, 256, "hann_sqrt"), (400, 400, 100, "hann")):
        spec = oracle_stft(x, n_fft, win, hop, wt)
        F = n_fft // 2 + 1
        # against an exact rFFT of the same frames: the reference's fp32 angle table is only ~2e-4 accurate
        import scipy.signal as ss
        w = ss.get_window("hann", win, fftbins=True)
        if wt == "hann_sqrt":
            w = np.sqrt(w)
        xp = np.pad(x.astype(np.float64), ((0, 0), (n_fft // 2, n_fft // 2)), mode="reflect")
        T = spec.shape[2]
        fr = np.stack([xp[:, t * hop:t * hop + n_fft] * w for t in range(T)], axis=2)
        truth = np.fft.rfft(fr, axis=1)
        tol = 1e-4 * np.abs(truth).max()   # 4e-5 relative observed
        assert np.abs(spec[:, :F] - truth.real).max() < tol
        assert np.abs(spec[:, F:] - truth.imag).max() < tol
        if wt == "hann_sqrt":
            y = oracle_istft(spec, n_fft, win, hop, wt)
            n = y.shape[1]
            assert np.abs(y - x[:, :n]).max() < 2e-3


def test_oracle_batch_fold_matches_reference():
    """USE_BATCH_FOLD export mode (Export_GTCRN.py:41-45,647-660): one call = whole fold windows, ONE DC mean for the
    call, windows run as a batch and stitched.  Fixture: the reference itself with USE_BATCH_FOLD=True on 48128 samples
    (tools/make_golden_gtcrn_fold.py); the input carries a DC step so that per-window means would be visibly wrong."""
    g = np.load(os.path.join(GOLD, "gtcrn_seed0_fold.npz"))
    W, n_win = int(g["fold_window_length"]), int(g["n_windows"])
    assert (W, n_win, g["pcm_in"].size) == (24064, 2, 48128)
    o = GtcrnOracle(_blob(0), W)
    assert o.out_len == W                                   # W is a multiple of the hop: every window reconstructs to W samples
    pcm, _ = o.process_fold(g["pcm_in"], n_win, threads=2)
    assert np.abs(pcm[0].astype(np.int32) - g["pcm_out"].astype(np.int32)).max() <= 1
    # the same windows as independent calls (per-window means) do NOT reproduce the fold output
    pcm_rows, _ = o.process(g["pcm_in"].reshape(n_win, W), threads=2)
    assert np.abs(pcm_rows.reshape(-1).astype(np.int32) - g["pcm_out"].astype(np.int32)).max() > 1


# analysis / synthesis window names in the oracle's vocabulary for each fixture of tools/make_golden_stft.py
STFT_CASES = {
    "gtcrn": ("hann_sqrt", "hann_sqrt"),
    "zipenhancer": ("hann", "hann"),
    "melband": ("hann", "hann"),
    "dfsmn": ("hamming_sym", "hamming_periodic"),     # DFSMN's registry binds 'hamming' to periodic=False
}


@pytest.mark.parametrize("name", sorted(STFT_CASES))
def test_oracle_stft_process_configs(name):
    """STFT_Process restatement vs each starred model folder's own STFT_Process copy (rows a1-a4): 512/256 sqrt-hann,
    400/100 hann, 2048/441 hann (centre, reflect) and 1920/960 hamming without centre padding."""
    g = np.load(os.path.join(GOLD, f"stft_{name}.npz"))
    n_fft, win, hop, center = int(g["n_fft"]), int(g["win_length"]), int(g["hop"]), bool(g["center"])
    wa, ws = STFT_CASES[name]
    spec = oracle_stft(g["x"], n_fft, win, hop, wa, center, str(g["pad_mode"]))
    assert spec.shape == g["spec"].shape
    # fp32 sums over n_fft terms in a different order than torch's conv1d: a few 1e-6 relative at n_fft = 2048
    assert np.abs(spec - g["spec"]).max() <= 1e-5 * np.abs(g["spec"]).max()
    y = oracle_istft(g["spec"], n_fft, win, hop, ws, center)
    assert y.shape == g["y"].shape
    assert np.abs(y - g["y"]).max() <= 1e-5
