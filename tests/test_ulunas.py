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


def _oracle(fused, length=16000):
    from ulunas_oracle import UlunasOracle
    return UlunasOracle(fused, ulunas.block_plan(), length)


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
