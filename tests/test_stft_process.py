"""Generic STFT_Process operator (csrc/ade_stft.hip): the dense windowed DFT as fp32 MFMA
GEMMs for the analysis / synthesis configurations of the starred model folders.

Fixtures: tests/golden/stft_<name>.npz, produced by each folder's own STFT_Process copy (tools/make_golden_stft.py).
The operator uses exact-angle DFT tables; the reference evaluates cos/sin of fp32 angles (up to 1e-4 relative table
error), so two comparisons are made: tight against the oracle with exact tables (kernel error only) and
loose against the reference fixture (kernel + the reference's table error).
"""
import ctypes as C
import os

import numpy as np
import pytest

from ade_testlib import GOLD, hipsim_library
from audio_denoiser_onnx_amd import _lib
from oracle_lib import oracle_istft, oracle_set_generic_exact_dft, oracle_stft

CASES = {   # name -> (analysis window, synthesis window) in the canonical vocabulary
    "gtcrn": ("hann_sqrt", "hann_sqrt"),
    "zipenhancer": ("hann", "hann"),
    "melband": ("hann", "hann"),
    "dfsmn": ("hamming_sym", "hamming_periodic"),
}


def _load(name):
    g = np.load(os.path.join(GOLD, f"stft_{name}.npz"))
    return g, int(g["n_fft"]), int(g["win_length"]), int(g["hop"]), bool(g["center"]), str(g["pad_mode"])


@pytest.mark.hipsim
def test_hipsim_generic_stft_small():
    """Kernel logic under the host simulator (MFMA emulated): ZipEnhancer's 400/100 hann on a short input."""
    lib = hipsim_library()
    g, n_fft, win, hop, center, pad = _load("zipenhancer")
    x = np.ascontiguousarray(g["x"][:, :1300])
    cfg = _lib.StftConfig(n_fft, win, hop, b"hann", None, int(center), pad.encode())
    h = C.c_void_p()
    assert lib.c.ade_stft_create(C.byref(cfg), 0, C.byref(h)) == 0
    t = C.c_int()
    assert lib.c.ade_stft_frames(h, x.shape[1], C.byref(t)) == 0 and t.value == 14
    spec = np.empty((2, n_fft + 2, t.value), np.float32)
    assert lib.c.ade_stft_analyze(h, x.ctypes.data, 2, x.shape[1], spec.ctypes.data, None) == 0     # (simulated device memory = host memory)
    oracle_set_generic_exact_dft(True)
    try:
        ref = oracle_stft(x, n_fft, win, hop, "hann", center, pad)
        assert np.abs(spec - ref).max() <= 2e-6 * np.abs(ref).max()
        n = C.c_int()
        assert lib.c.ade_stft_output_length(h, t.value, C.byref(n)) == 0 and n.value == 1300
        y = np.empty((2, n.value), np.float32)
        assert lib.c.ade_stft_synthesize(h, spec.ctypes.data, 2, t.value, y.ctypes.data, None) == 0
        assert np.abs(y - oracle_istft(ref, n_fft, win, hop, "hann", center)).max() <= 2e-6
        assert np.abs(y - x).max() <= 2e-6                                                         # perfect reconstruction
    finally:
        oracle_set_generic_exact_dft(False)
        lib.c.ade_stft_destroy(h)
    bad = _lib.StftConfig(400, 400, 100, b"kaiser", None, 1, b"reflect")
    assert lib.c.ade_stft_create(C.byref(bad), 0, C.byref(h)) == _lib.ADE_ERR_UNSUPPORTED
    assert b"kaiser" in lib.c.ade_stft_last_error(None)


@pytest.mark.hipsim
@pytest.mark.parametrize("n_fft,hop,length,center", [(60, 15, 333, True),      # 4 * 3 * 5: FFT path, odd frame count (last pair half empty)
                                                      (45, 9, 200, True),       # odd n_fft (no Nyquist bin), radices 3 and 5 only
                                                      (64, 32, 256, False),     # no centre padding
                                                      (44, 11, 180, True)])     # 4 * 11: not 5-smooth -> the dense-table MFMA path
def test_hipsim_generic_stft_fft_and_dense_paths(n_fft, hop, length, center):
    """Both formulations of the operator (LDS FFT for 5-smooth sizes, dense tables otherwise) against the oracle with exact tables, including the polar entry."""
    lib = hipsim_library()
    rng = np.random.default_rng(n_fft)
    x = rng.standard_normal((3, length)).astype(np.float32)
    cfg = _lib.StftConfig(n_fft, n_fft, hop, b"hamming", None, int(center), b"reflect")
    h = C.c_void_p()
    assert lib.c.ade_stft_create(C.byref(cfg), 0, C.byref(h)) == 0
    oracle_set_generic_exact_dft(True)
    try:
        t = C.c_int()
        assert lib.c.ade_stft_frames(h, length, C.byref(t)) == 0
        ref = oracle_stft(x, n_fft, n_fft, hop, "hamming", center, "reflect")
        assert ref.shape[2] == t.value
        spec = np.empty_like(ref)
        assert lib.c.ade_stft_analyze(h, x.ctypes.data, 3, length, spec.ctypes.data, None) == 0
        assert np.abs(spec - ref).max() <= 2e-6 * np.abs(ref).max()
        n = C.c_int()
        assert lib.c.ade_stft_output_length(h, t.value, C.byref(n)) == 0
        want = oracle_istft(ref, n_fft, n_fft, hop, "hamming", center)
        assert want.shape[1] == n.value
        y = np.empty_like(want)
        assert lib.c.ade_stft_synthesize(h, ref.ctypes.data, 3, t.value, y.ctypes.data, None) == 0
        tol = 3e-6 if center else 5e-5                  # without centre padding the first / last hop divides by a tiny sum(w^2)
        assert np.abs(y - want).max() <= tol
        F = n_fft // 2 + 1
        mag = np.ascontiguousarray(np.hypot(ref[:, :F], ref[:, F:]))
        ph = np.ascontiguousarray(np.arctan2(ref[:, F:], ref[:, :F]))
        y2 = np.empty_like(want)
        assert lib.c.ade_stft_synthesize_polar(h, mag.ctypes.data, ph.ctypes.data, 3, t.value, y2.ctypes.data, None) == 0
        assert np.abs(y2 - want).max() <= 4 * tol
    finally:
        oracle_set_generic_exact_dft(False)
        lib.c.ade_stft_destroy(h)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_gpu_stft_process_configs(name):
    import torch
    from audio_denoiser_onnx_amd.stft_process import STFT_Process
    g, n_fft, win, hop, center, pad = _load(name)
    wa, ws = CASES[name]
    stft = STFT_Process("stft_B", n_fft, win, hop, 0, wa, center, pad)
    x = torch.from_numpy(g["x"]).cuda()
    spec = stft(x[:, None, :])
    assert tuple(spec.shape) == g["spec"].shape
    spec_h = spec.cpu().numpy()
    smax = float(np.abs(g["spec"]).max())
    # loose: the reference fixture (its fp32-angle table error included: up to ~1e-4 relative at n_fft = 2048)
    assert np.abs(spec_h - g["spec"]).max() <= 2e-4 * smax
    oracle_set_generic_exact_dft(True)
    try:
        ref = oracle_stft(g["x"], n_fft, win, hop, wa, center, pad)
        assert np.abs(spec_h - ref).max() <= 1e-5 * smax                       # tight: kernel error only
        istft = STFT_Process("istft_B", n_fft, win, hop, spec.shape[2], ws, center, pad, static_norm=True)
        y = istft(torch.from_numpy(g["spec"]).cuda()).cpu().numpy().reshape(2, -1)
        assert y.shape == g["y"].shape
        # vs the reference fixture.  Without centre padding the first / last hop is divided by a tiny sum(w^2)
        # (hamming ends at 0.08), which amplifies the reference's own table error there (7e-4 observed)
        assert np.abs(y - g["y"]).max() <= (1e-4 if center else 2e-3)
        assert np.abs(y - oracle_istft(g["spec"], n_fft, win, hop, ws, center)).max() <= 1e-5
    finally:
        oracle_set_generic_exact_dft(False)
    if center:    # analysis -> synthesis of the operator's own spectrum reconstructs the input
        y2 = STFT_Process("istft_B", n_fft, win, hop, spec.shape[2], ws, center, pad)(spec).cpu().numpy().reshape(2, -1)
        assert np.abs(y2 - g["x"][:, :y2.shape[1]]).max() <= 1e-5


@pytest.mark.gpu
def test_gpu_stft_process_large_batch_properties():
    """Mel-Band geometry at a realistic size (stereo 1.5 s windows): linearity and batch independence."""
    import torch
    from audio_denoiser_onnx_amd.stft_process import STFT_Process
    stft = STFT_Process("stft_B", 2048, 2048, 441, 0, "hann", True, "reflect")
    gen = torch.Generator(device="cpu").manual_seed(7)
    x = (torch.randn(16, 66150, generator=gen) * 0.1).cuda()
    a = stft(x)
    assert a.shape == (16, 2050, 151)
    b = stft(x[3:5])
    assert torch.equal(a[3:5], b)                                              # rows are independent, kernels deterministic
    c = stft(2.0 * x[:2] + x[2:4])
    assert float((c - (2.0 * a[:2] + a[2:4])).abs().max()) <= 1e-4 * float(a.abs().max())


@pytest.mark.gpu
def test_gpu_stft_a_and_polar_istft_a():
    """model_type 'stft_A' (real rows), the split pair of 'stft_B', and 'istft_A' (magnitude, phase) against GTCRN's own STFT_Process copy
    (tests/golden/stft_gtcrn_polar.npz, tools/make_golden_stft.py::polar)."""
    import torch
    from audio_denoiser_onnx_amd.stft_process import STFT_Process
    g, n_fft, win, hop, center, pad = _load("gtcrn")
    p = np.load(os.path.join(GOLD, "stft_gtcrn_polar.npz"))
    x = torch.from_numpy(g["x"]).cuda()
    real_a = STFT_Process("stft_A", n_fft, win, hop, 0, "hann_sqrt", center, pad)(x[:, None, :])
    assert tuple(real_a.shape) == p["real_a"].shape and np.abs(real_a.cpu().numpy() - p["real_a"]).max() <= 2e-4 * float(np.abs(p["real_a"]).max())
    packed = STFT_Process("stft_B", n_fft, win, hop, 0, "hann_sqrt", center, pad)(x[:, None, :])
    re, im = STFT_Process.split(packed)
    assert torch.equal(re, real_a) and tuple(im.shape) == tuple(re.shape)
    T = p["magnitude"].shape[2]
    ist = STFT_Process("istft_A", n_fft, win, hop, T, "hann_sqrt", center, pad, static_norm=True)
    y = ist(torch.from_numpy(p["magnitude"]).cuda(), torch.from_numpy(p["phase"]).cuda()).cpu().numpy().reshape(2, -1)
    assert y.shape == p["y"].shape and np.abs(y - p["y"]).max() <= 1e-4
    with pytest.raises(TypeError):
        ist.forward(torch.from_numpy(p["magnitude"]).cuda())
    with pytest.raises(ValueError):
        ist(torch.from_numpy(p["magnitude"][:, :100]).cuda(), torch.from_numpy(p["phase"][:, :100]).cuda())


@pytest.mark.gpu
@pytest.mark.parametrize("n_fft,hop,length", [(512, 256, 300), (512, 256, 513), (400, 100, 250), (2048, 441, 1100), (1920, 960, 1921)])
def test_gpu_stft_fft_path_few_frames(n_fft, hop, length):
    """One to a handful of frames (a half-empty last pair, a single workgroup with idle pair groups) on the FFT path, against the oracle with exact tables."""
    import torch
    from audio_denoiser_onnx_amd.stft_process import STFT_Process
    rng = np.random.default_rng(length)
    x = (rng.standard_normal((3, length)) * 0.2).astype(np.float32)
    stft = STFT_Process("stft_B", n_fft, n_fft, hop, 0, "hann", True, "reflect")
    spec = stft(torch.from_numpy(x).cuda()[:, None, :]).cpu().numpy()
    oracle_set_generic_exact_dft(True)
    try:
        ref = oracle_stft(x, n_fft, n_fft, hop, "hann", True, "reflect")
        assert spec.shape == ref.shape and np.abs(spec - ref).max() <= 1e-5 * max(1.0, float(np.abs(ref).max()))
        y = STFT_Process("istft_B", n_fft, n_fft, hop, spec.shape[2], "hann", True, "reflect")(torch.from_numpy(ref).cuda()).cpu().numpy().reshape(3, -1)
        want = oracle_istft(ref, n_fft, n_fft, hop, "hann", True)
        assert y.shape == want.shape and np.abs(y - want).max() <= 1e-5
    finally:
        oracle_set_generic_exact_dft(False)


@pytest.mark.gpu
@pytest.mark.parametrize("name,max_frames", [("gtcrn", 4096), ("melband", 2048)])
def test_gpu_istft_dynamic_trim_keeps_the_tail(name, max_frames):
    """An ISTFT built for a DYNAMIC_AXES export (max_frames above the frame count, per-call window sum): the reference's slice keeps the second half of the last frame
    (GTCRN/STFT_Process.py:337-341, Mel_Band_Roformer/Stereo/STFT_Process.py:296-306).  tests/golden/stft_dynamic_tail.npz = that tail from each folder's own copy; the
    tail divides the last frame alone by its squared window (down to 3.8e-5 / 5.5e-12 at the very end), so it is compared relative to the signal."""
    import torch
    from audio_denoiser_onnx_amd.stft_process import STFT_Process
    g, n_fft, win, hop, center, pad = _load(name)
    tail = np.load(os.path.join(GOLD, "stft_dynamic_tail.npz"))[name + "_tail"]
    ws = CASES[name][1]
    spec = torch.from_numpy(g["spec"]).cuda()
    T = spec.shape[2]
    y = STFT_Process("istft_B", n_fft, win, hop, max_frames, ws, center, pad, static_norm=False)(spec).cpu().numpy().reshape(2, -1)
    static_len = hop * (T - 1)
    assert y.shape == (2, static_len + n_fft // 2) and tail.shape == (2, n_fft // 2)
    assert np.abs(y[:, :static_len] - g["y"]).max() <= 1e-4                            # the static part is unchanged
    # the tail is (last frame) / w^2: an absolute error e of the frame (fp32 round-off here, the fp32-angle tables in the reference: ~3e-7 of the signal) becomes e / w^2
    n = np.arange(n_fft // 2, n_fft)
    hann = 0.5 - 0.5 * np.cos(2.0 * np.pi * n / n_fft)
    w2 = hann if ws == "hann_sqrt" else hann * hann
    # (the reference's fp32-angle tables are off by up to ~1e-4 of the signal at n_fft = 2048, see test_gpu_stft_process_configs; the operator's are exact)
    assert (np.abs(y[:, static_len:] - tail) <= 1e-4 + 1e-7 * n_fft * np.abs(g["x"]).max() / w2).all(), np.abs(y[:, static_len:] - tail).max()
    body = slice(0, n_fft // 4)                                                         # where w^2 is still O(1): tight
    assert np.abs(y[:, static_len:][:, body] - tail[:, body]).max() <= 1e-4
    # max_frames equal to the frame count is the static trim again; fewer frames than fed truncates (the same slice)
    y_eq = STFT_Process("istft_B", n_fft, win, hop, T, ws, center, pad, static_norm=False)(spec).cpu().numpy().reshape(2, -1)
    assert y_eq.shape == (2, static_len) and np.array_equal(y_eq, y[:, :static_len])
    y_lt = STFT_Process("istft_B", n_fft, win, hop, T - 2, ws, center, pad, static_norm=False)(spec).cpu().numpy().reshape(2, -1)
    assert y_lt.shape == (2, hop * (T - 3)) and np.array_equal(y_lt, y[:, :hop * (T - 3)])
