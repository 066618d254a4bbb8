"""MI355X-native engine for the DakeQQ/Audio-Denoiser-ONNX per-chunk denoise path (GTCRN first).

Package layout (only what the hot path needs):
  csrc/          hand-written gfx950 kernels + the C-ABI engine (built to libade.so by __graft_entry__.build)
  _lib.py        ctypes binding of include/ade.h
  session.py     InferenceSession — mirror of the ORT session surface the reference driver uses
  metadata.py    the reference's audio metadata contract on a JSON carrier
  weights.py     ADEWGT01 weight-blob container
  export.py      checkpoint/state_dict -> BN-folded blob + manifest
  inference_gtcrn.py   the Inference_GTCRN_ONNX.py call surface (wav in -> slices -> wav out, RTF)
"""
from .metadata import (REQUIRED_AUDIO_METADATA_KEYS, MetadataReader, build_audio_metadata, load_runtime_metadata,  # noqa: F401
                       runtime_config_from_metadata, validate_audio_metadata)
from .weights import load_blob, pack_blob, save_blob, unpack_blob  # noqa: F401


def InferenceSession(*args, **kwargs):
    """Lazy constructor so importing the package never needs the built library; creating a session does."""
    from .session import InferenceSession as _S
    return _S(*args, **kwargs)


__all__ = ["InferenceSession", "REQUIRED_AUDIO_METADATA_KEYS", "MetadataReader", "build_audio_metadata",
           "load_runtime_metadata", "runtime_config_from_metadata", "validate_audio_metadata", "load_blob", "pack_blob",
           "save_blob", "unpack_blob"]
