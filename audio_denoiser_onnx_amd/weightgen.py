"""Counter-based weight generator: reproducible "seeded weights" for models whose tensors are too large to commit.

Mel-Band-Roformer's fused buffers hold ~208 M floats at depth 1 (the 60-band mask estimator alone is 141.6 M,
Mel_Band_Roformer/Stereo/Export_MelBandRoformer.py:499-502) -- 0.8 GB cannot be a fixture, and torch's RNG stream is
not reproducible outside torch.  Every value is instead a pure function of (tensor name, flat index, seed):

    value[i] = scale * (2 * u - 1),  u = top 24 bits of splitmix64(fnv1a64(name) ^ seed * GOLDEN + i) / 2^24

so the golden-vector script (which overwrites the reference module's buffers with these values before running it), the
oracle and the engine-side tests all materialise identical tensors from a (name, shape, scale) list.
"""
from __future__ import annotations

from typing import Dict, Iterable, Sequence, Tuple

import numpy as np

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)


def fnv1a64(name: str) -> np.uint64:
    h = 0xCBF29CE484222325
    for b in name.encode():
        h = ((h ^ b) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return np.uint64(h)


def splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = x + _GOLDEN
        z = x
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def tensor(name: str, shape: Sequence[int], scale: float, seed: int = 0) -> np.ndarray:
    n = int(np.prod(shape)) if len(shape) else 1
    with np.errstate(over="ignore"):
        base = fnv1a64(name) ^ (np.uint64(seed) * _GOLDEN)
        out = np.empty(n, np.float32)
        step = 1 << 24
        for lo in range(0, n, step):                       # chunked: bounds the uint64 scratch for the 141 M-float tensor
            idx = np.arange(lo, min(n, lo + step), dtype=np.uint64)
            bits = splitmix64(base + idx) >> np.uint64(40)                     # top 24 bits
            out[lo:lo + idx.size] = (bits.astype(np.float32) * np.float32(2.0 / (1 << 24)) - np.float32(1.0)) * np.float32(scale)
    return out.reshape(tuple(shape))


def materialise(spec: Iterable[Tuple[str, Sequence[int], float]], seed: int = 0) -> Dict[str, np.ndarray]:
    return {name: tensor(name, shape, scale, seed) for name, shape, scale in spec}
