"""Weight-blob container shared by the HIP engine (libade) and the CPU oracle.

The reference bakes its (BatchNorm-folded) weights into ONNX initializers at
export time (GTCRN/Export_GTCRN.py:742-777, fold at :171-194 and :244-270).
This engine has no ONNX, so the same post-fold tensors travel in a flat
little-endian container, keyed by the reference's own ``state_dict`` names
(PyTorch layouts untouched; libade re-lays them out for its kernels at load).

Layout ("ADEWGT01"):
    8   magic  b"ADEWGT01"
    u32 n_tensors
    per tensor:  u16 name_len | name utf-8 | u8 dtype(0=f32) | u8 ndim |
                 u32 dims[ndim] | u64 offset (from data start) | u64 nbytes
    zero pad to a 64-byte boundary, then the data section (each tensor 64-B aligned).
"""
from __future__ import annotations

import struct
from collections import OrderedDict
from typing import Dict, Mapping

import numpy as np

MAGIC = b"ADEWGT01"
_ALIGN = 64


def _pad(n: int) -> int:
    return (-n) % _ALIGN


def pack_blob(tensors: Mapping[str, np.ndarray]) -> bytes:
    """Serialise ``name -> float32 ndarray`` into the blob format."""
    header = bytearray()
    header += MAGIC
    header += struct.pack("<I", len(tensors))
    payload = bytearray()
    for name, arr in tensors.items():
        a = np.ascontiguousarray(np.asarray(arr), dtype="<f4")
        raw = a.tobytes()
        payload += b"\0" * _pad(len(payload))
        offset = len(payload)
        payload += raw
        nb = name.encode("utf-8")
        header += struct.pack("<H", len(nb)) + nb
        header += struct.pack("<BB", 0, a.ndim)
        header += struct.pack("<%dI" % a.ndim, *a.shape) if a.ndim else b""
        header += struct.pack("<QQ", offset, len(raw))
    header += b"\0" * _pad(len(header))
    return bytes(header) + bytes(payload)


def unpack_blob(blob: bytes) -> "OrderedDict[str, np.ndarray]":
    """Inverse of :func:`pack_blob` (raises ``ValueError`` on a malformed blob)."""
    if len(blob) < 12 or blob[:8] != MAGIC:
        raise ValueError("not an ADEWGT01 weight blob")
    (n,) = struct.unpack_from("<I", blob, 8)
    pos = 12
    entries = []
    for _ in range(n):
        (ln,) = struct.unpack_from("<H", blob, pos)
        pos += 2
        name = blob[pos:pos + ln].decode("utf-8")
        pos += ln
        dtype, ndim = struct.unpack_from("<BB", blob, pos)
        pos += 2
        if dtype != 0:
            raise ValueError(f"tensor {name}: unsupported dtype code {dtype}")
        dims = struct.unpack_from("<%dI" % ndim, blob, pos) if ndim else ()
        pos += 4 * ndim
        off, nbytes = struct.unpack_from("<QQ", blob, pos)
        pos += 16
        entries.append((name, dims, off, nbytes))
    data0 = pos + _pad(pos)
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for name, dims, off, nbytes in entries:
        count = int(np.prod(dims)) if dims else 1
        if count * 4 != nbytes or data0 + off + nbytes > len(blob):
            raise ValueError(f"tensor {name}: bad extent")
        out[name] = np.frombuffer(blob, dtype="<f4", count=count, offset=data0 + off).reshape(dims).copy()
    return out


def save_blob(path, tensors: Mapping[str, np.ndarray]) -> None:
    with open(path, "wb") as f:
        f.write(pack_blob(tensors))


def load_blob(path) -> Dict[str, np.ndarray]:
    with open(path, "rb") as f:
        return unpack_blob(f.read())
