"""DFSMN: oracle pinned to the reference-generated fixture; engine vs oracle / fixture.

Fixture: tools/make_golden_dfsmn.py runs the reference's own ``DFSMN.forward`` on a seeded parameter tree (the reference
gets its parameters from modelscope, absent) with this package's restatement of the Kaldi mel bank standing in for
torchaudio's (absent): both are INPUTS of the forward being pinned, not part of it.
"""
import os
import sys

import numpy as np
import pytest

from ade_testlib import GOLD
from audio_denoiser_onnx_amd.weights import load_blob

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
from dfsmn_oracle import DfsmnOracle  # noqa: E402


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLD, "dfsmn_seed0_io.npz"))


@pytest.fixture(scope="module")
def tensors():
    return load_blob(os.path.join(GOLD, "dfsmn_seed0.adew"))


def test_dfsmn_oracle_matches_reference(gold, tensors):
    L = int(gold["input_audio_length"])
    o = DfsmnOracle(tensors, L)
    assert o.frames == int(gold["frames"]) == 24 and o.out_len == L
    names = ["speech0", "speech1", "randn", "zeros"]
    pcm, _ = o.process(np.stack([gold[f"{n}.pcm_in"] for n in names]))
    assert np.abs(o.taps["logmel"] - gold["speech0.logmel"]).max() <= 2e-4            # log of a 1025-term fp32 sum
    assert np.abs(o.taps["mask"] - gold["speech0.mask"]).max() <= 1e-4
    for i, n in enumerate(names):
        d = np.abs(pcm[i].astype(np.int32) - gold[f"{n}.pcm_out"].astype(np.int32)).max()
        assert d <= 1, (n, d)
    assert not pcm[3].any()
    m = gold["speech0.mask"]
    assert float(m.std()) > 0.05 and float(m.min()) < 0.2 and float(m.max()) > 0.8      # a non-degenerate mask
