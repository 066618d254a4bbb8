"""MossFormer2-SS-16K: oracle pin (CPU) and HIP parity through the C ABI (GPU).

Fixtures: tests/golden/mossformer_seed0_io.npz / mossformer_seed0_fold_io.npz = the reference's own constructor + forward run in
the build container over a stand-in network tree with generator-filled fused buffers (tools/make_golden_mossformer.py); the 6 M
weights are regenerated here from (name, shape, scale) (audio_denoiser_onnx_amd.mossformer.synthetic_tensor).
"""
import json
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(REPO, "oracle"))

from audio_denoiser_onnx_amd import mossformer  # noqa: E402

GOLD = os.path.join(HERE, "golden", "mossformer_seed0_io.npz")
GOLD_FOLD = os.path.join(HERE, "golden", "mossformer_seed0_fold_io.npz")


@pytest.fixture(scope="module")
def fixture():
    z = np.load(GOLD)
    spec = json.loads(str(z["spec"]))
    scalars = json.loads(str(z["scalars"]))
    W = z["pcm_in"].shape[0]
    frames = mossformer.frames_of(W)
    fused = {n: mossformer.synthetic_tensor(n, s, sc, frames, int(scalars["flash_group_size"])) for n, s, sc in spec}
    return z, fused, scalars, W


def _oracle(fixture):
    from mossformer_oracle import MossFormerOracle
    z, fused, scalars, W = fixture
    tensors = dict(fused)
    tensors.update(mossformer.position_tables(mossformer.frames_of(W), int(scalars["rot_dim"])))
    return MossFormerOracle(tensors, scalars, int(z["layers"]), W)


def test_oracle_matches_reference_forward(fixture):
    z = fixture[0]
    o = _oracle(fixture)
    out = o.process(z["pcm_in"][None])[0]
    assert np.abs(o.taps["mdl_in"][0][:, ::7] - z["mdl_in"]).max() < 5e-4
    assert np.abs(o.taps["mdl_out"][0][:, ::7] - z["mdl_out"]).max() < 1e-3
    d = out.astype(np.int32) - z["pcm_out"].astype(np.int32)
    assert np.abs(d).max() <= 1 and (d != 0).mean() < 0.02
    assert np.abs(z["pcm_out"]).max(axis=1).min() > 2000                      # both separated sources carry signal


def test_oracle_batch_fold_matches_reference_forward(fixture):
    """USE_BATCH_FOLD: 3 windows, per-window RMS normalisation and restore, the last window partly silent."""
    zf = np.load(GOLD_FOLD)
    out = _oracle(fixture).process_fold(zf["pcm_in"])
    d = out.astype(np.int32) - zf["pcm_out"].astype(np.int32)
    assert np.abs(d).max() <= 1 and (d != 0).mean() < 0.02


def test_position_tables_and_hyper_layout(fixture):
    _, fused, scalars, W = fixture
    t = mossformer.model_tensors(fused, scalars, W)
    n = mossformer.frames_of(W)
    assert t["emb_pos"].shape == (1, 512, n) and t["rot_cos"].shape == (1, n, 1, 32) and t["rot_signed_sin"].shape == (1, n, 1, 32)
    assert np.allclose(t["emb_pos"][0, 256:, 0], 1.0) and np.allclose(t["emb_pos"][0, :256, 0], 0.0)          # sin | cos halves at t = 0
    assert t["rot_signed_sin"][0, 5, 0, 0] == -t["rot_signed_sin"][0, 5, 0, 1]                                # sign folded pairwise
    h = dict(zip(mossformer.HYPER_KEYS, t["hyper"]))
    assert h["flash_group_size"] == 256 and h["fs_mem_lorder"] == 20 and h["fs_mem_depth"] == 2 and h["dw_pad"] == 8
    assert t["fs_front_alpha"].shape == (2,)
