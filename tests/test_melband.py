"""Mel-Band-Roformer: oracle pin (CPU) and HIP parity through the C ABI (GPU).

Fixture: tests/golden/melband_seed0_io.npz = the reference's own forward run in the build container over generator-filled
fused buffers (tools/make_golden_melband.py).  The 208 M weights are regenerated here from (name, shape, scale) with
audio_denoiser_onnx_amd.weightgen -- a pure function of name and index -- so they never need to be committed.
"""
import json
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(REPO, "oracle"))

from audio_denoiser_onnx_amd import weightgen  # noqa: E402

GOLD = os.path.join(HERE, "golden", "melband_seed0_io.npz")


@pytest.fixture(scope="module")
def fixture():
    z = np.load(GOLD)
    spec = [(n, s, sc) for n, s, sc in json.loads(str(z["spec"]))]
    return z, spec, weightgen.materialise(spec)


def test_weightgen_is_a_pure_function_of_name_and_index():
    a = weightgen.tensor("me_w1t", (4, 7, 5), 0.1)
    b = weightgen.tensor("me_w1t", (140,), 0.1)
    assert np.array_equal(a.reshape(-1), b) and a.dtype == np.float32
    assert np.abs(a).max() < 0.1 and not np.array_equal(a.reshape(-1), weightgen.tensor("me_w2t", (140,), 0.1))
    big = weightgen.tensor("x", (1 << 16,), 1.0)
    assert abs(float(big.mean())) < 0.02 and abs(float(big.std()) - 3 ** -0.5) < 0.01      # uniform(-1, 1)


def test_oracle_matches_reference_forward(fixture):
    """The numpy restatement against the reference's own forward: taps at fp32 round-off (the L2-normalised band inputs of
    the near-silent top bands amplify it to ~1e-4), PCM within 1 LSB."""
    from melband_oracle import MelBandOracle
    z, _, w = fixture
    o = MelBandOracle(w, z["freq_indices"], z["dim_inputs"], int(z["frames"]), int(z["depth"]))
    out = o.process(z["pcm_in"])
    assert np.abs(o.taps["band_split"][0] - z["band_split_b0"]).max() < 5e-5
    assert np.abs(o.taps["tf_out"][7] - z["tf_out_b7"]).max() < 1e-3
    assert np.abs(o.taps["masks"][:, :256] - z["masks"]).max() < 1e-3
    d = out.astype(np.int32) - z["pcm_out"].astype(np.int32)
    assert np.abs(d).max() <= 1 and (d != 0).mean() < 0.05
    assert np.abs(z["pcm_out"]).max() > 2000                                              # a non-trivial signal came out
