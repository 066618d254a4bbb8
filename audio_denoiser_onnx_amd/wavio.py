"""Minimal RIFF/WAVE reader and writer for the file drivers: 16-bit PCM in plain (``WAVE_FORMAT_PCM``) or
``WAVE_FORMAT_EXTENSIBLE`` containers, any channel count.

The reference reads with pydub and writes Mel-Band output with ``soundfile.write(..., format='WAVEX', subtype='PCM_16')``
(Mel_Band_Roformer/Stereo/Inference_MelBandRoformer_ONNX.py:268-276, 353); neither package is needed for 16-bit PCM.
The standard library's ``wave`` module rejects EXTENSIBLE files (the shipped Test_Examples/denoise/mel_band_roformer.wav
is one), hence this module.
"""
from __future__ import annotations

import struct
from typing import Tuple

import numpy as np

_PCM_GUID = bytes.fromhex("0100000000001000800000aa00389b71")    # KSDATAFORMAT_SUBTYPE_PCM


def read_pcm16(path) -> Tuple[np.ndarray, int]:
    """-> (int16 array (channels, frames), sample_rate).  ``ValueError`` for anything but 16-bit integer PCM."""
    with open(path, "rb") as f:
        data = f.read()
    if len(data) < 12 or data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise ValueError(f"{path}: not a RIFF/WAVE file")
    pos, fmt, pcm = 12, None, None
    while pos + 8 <= len(data):
        tag, size = data[pos:pos + 4], struct.unpack_from("<I", data, pos + 4)[0]
        body = data[pos + 8:pos + 8 + size]
        if tag == b"fmt ":
            fmt = body
        elif tag == b"data":
            pcm = body
        pos += 8 + size + (size & 1)
    if fmt is None or pcm is None or len(fmt) < 16:
        raise ValueError(f"{path}: missing fmt / data chunk")
    code, channels, rate, _, _, bits = struct.unpack_from("<HHIIHH", fmt, 0)
    if code == 0xFFFE:
        if len(fmt) < 40 or fmt[24:40] != _PCM_GUID:
            raise ValueError(f"{path}: WAVE_FORMAT_EXTENSIBLE with a non-PCM sub-format")
    elif code != 1:
        raise ValueError(f"{path}: unsupported wav format tag {code:#x} (integer PCM only)")
    if bits != 16 or channels < 1:
        raise ValueError(f"{path}: only 16-bit PCM wav is supported, got {bits}-bit")
    frames = len(pcm) // (2 * channels)
    x = np.frombuffer(pcm, dtype="<i2", count=frames * channels).reshape(frames, channels)
    return np.ascontiguousarray(x.T, dtype=np.int16), int(rate)


def write_pcm16(path, pcm: np.ndarray, sample_rate: int, extensible: bool = False) -> None:
    """``pcm`` int16 (frames,) or (channels, frames).  ``extensible=True`` writes the WAVEX header soundfile's
    ``format='WAVEX'`` produces (front-left/right speaker mask for stereo)."""
    x = np.asarray(pcm, dtype=np.int16)
    if x.ndim == 1:
        x = x[None]
    channels, frames = x.shape
    body = np.ascontiguousarray(x.T, dtype="<i2").tobytes()
    block = 2 * channels
    if extensible:
        mask = {1: 0x4, 2: 0x3}.get(channels, 0)
        fmt = struct.pack("<HHIIHHHHI", 0xFFFE, channels, sample_rate, sample_rate * block, block, 16, 22, 16, mask) + _PCM_GUID
    else:
        fmt = struct.pack("<HHIIHH", 1, channels, sample_rate, sample_rate * block, block, 16)
    chunks = b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"data" + struct.pack("<I", len(body)) + body + (b"\0" if len(body) & 1 else b"")
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 4 + len(chunks)) + b"WAVE" + chunks)
