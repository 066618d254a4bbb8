"""Model manifest = the reference's audio metadata contract without the ONNX carrier.

The reference stores a flat ``str -> str`` map in ``model.metadata_props`` plus a sidecar
``<name>_Metadata.onnx`` and validates it at load (audio_onnx_metadata.py:8-26 required keys, :161-203 key
set, :247-287 typed reader, :290-303 load, :315-351 validate, :354-386 runtime config).  This engine has no
ONNX, so the same key/value map lives in ``<name>_Metadata.json`` next to the ``<name>.adew`` weight blob.
Function names, argument meaning and the exception classes raised follow the reference so the driver and
the tests read the same way.
"""
from __future__ import annotations

import json
from pathlib import Path
from typing import Any, Dict, Iterable, Mapping, Optional

AUDIO_METADATA_VERSION = 1

# audio_onnx_metadata.py:8-26
REQUIRED_AUDIO_METADATA_KEYS = (
    "audio_metadata_version", "producer", "model_name", "task", "model_family", "dynamic_axes", "opset",
    "input_audio_dtype", "output_audio_dtype", "in_sample_rate", "out_sample_rate", "model_sample_rate",
    "input_audio_length", "input_to_output_scale", "max_dynamic_audio_seconds", "normalize_audio_default",
    "normalize_target_rms",
)

_TRUE = {"1", "true", "yes", "on"}
_FALSE = {"0", "false", "no", "off"}


def metadata_path_for_model(model_path) -> Path:
    """``foo/GTCRN.adew`` -> ``foo/GTCRN_Metadata.json`` (audio_onnx_metadata.py:37-39 with a JSON carrier)."""
    p = Path(model_path)
    return p.with_name(p.stem + "_Metadata.json")


def _encode(value: Any) -> str:
    if isinstance(value, bool):
        return "1" if value else "0"
    if isinstance(value, (list, tuple)):
        return ",".join(str(v) for v in value)
    return str(value)


def build_model_metadata(*sections: Optional[Mapping[str, Any]]) -> Dict[str, str]:
    """Merge sections into one string map; ``None`` values are dropped, bools become ``"1"/"0"``."""
    merged: Dict[str, str] = {}
    for section in sections:
        for key, value in (section or {}).items():
            if value is not None:
                merged[str(key)] = _encode(value)
    return merged


def build_audio_metadata(*, producer: str, model_name: str, task: str, model_family: str, input_audio_length: int,
                         in_sample_rate: int = 16000, out_sample_rate: Optional[int] = None,
                         model_sample_rate: Optional[int] = None, nfft: int = 512, window_length: int = 512,
                         hop_length: int = 256, window_type: str = "hann_sqrt", center_pad: bool = True,
                         pad_mode: str = "reflect", dynamic_axes: bool = False, opset: int = 0,
                         input_audio_dtype: str = "INT16", output_audio_dtype: str = "INT16",
                         max_dynamic_audio_seconds: int = 30, normalize_audio_default: bool = False,
                         normalize_target_rms: float = 4096.0, batch_window_seconds: float = 1.5,
                         use_batch_fold: bool = False, input_channels: int = 1, output_channels: int = 1,
                         num_audio_inputs: int = 1, feature_kind: str = "stft",
                         extra: Optional[Mapping[str, Any]] = None) -> Dict[str, str]:
    """The key set ``build_audio_metadata_from_globals`` stamps (audio_onnx_metadata.py:115-205), from explicit args."""
    out_sample_rate = in_sample_rate if out_sample_rate is None else out_sample_rate
    model_sample_rate = out_sample_rate if model_sample_rate is None else model_sample_rate
    fold_window = ((int(batch_window_seconds * model_sample_rate) + hop_length - 1) // hop_length) * hop_length
    # USE_BATCH_FOLD (Export_GTCRN.py:41-45): the graph input is rounded UP to whole fold windows and the static frame
    # count is that of ONE window
    export_length = ((input_audio_length + fold_window - 1) // fold_window) * fold_window if use_batch_fold else input_audio_length
    frames = (fold_window if use_batch_fold else export_length) // hop_length + 1
    return build_model_metadata({
        "audio_metadata_version": AUDIO_METADATA_VERSION,
        "producer": producer,
        "model_name": model_name,
        "task": task,
        "model_family": model_family,
        "dynamic_axes": dynamic_axes,
        "opset": opset,
        "input_audio_dtype": input_audio_dtype,
        "output_audio_dtype": output_audio_dtype,
        "in_sample_rate": in_sample_rate,
        "out_sample_rate": out_sample_rate,
        "model_sample_rate": model_sample_rate,
        "input_audio_length": input_audio_length,
        "export_audio_length": export_length,
        "model_audio_length": int(round(input_audio_length * model_sample_rate / in_sample_rate)),
        "output_audio_length": int(round(input_audio_length * out_sample_rate / in_sample_rate)),
        "input_to_output_scale": float(out_sample_rate / in_sample_rate),
        "batch_window_seconds": batch_window_seconds,
        "use_batch_fold": use_batch_fold,
        "batch_fold_inference_default": use_batch_fold,
        "fold_window_length": fold_window,
        "fold_input_length": max(1, int(round(fold_window * in_sample_rate / model_sample_rate))),
        "max_dynamic_audio_seconds": max_dynamic_audio_seconds,
        "normalize_audio_default": normalize_audio_default,
        "normalize_target_rms": normalize_target_rms,
        "window_type": window_type,
        "nfft": nfft,
        "window_length": window_length,
        "hop_length": hop_length,
        "max_signal_length": frames,
        "center_pad": center_pad,
        "pad_mode": pad_mode,
        "feature_kind": feature_kind,
        "input_channels": input_channels,
        "output_channels": output_channels,
        "num_audio_inputs": num_audio_inputs,
    }, extra)


def write_metadata(model_path, metadata: Mapping[str, Any]) -> Path:
    """Write the manifest next to the weight blob (stamp_export_metadata's role, audio_onnx_metadata.py:107-112)."""
    path = metadata_path_for_model(model_path)
    path.parent.mkdir(parents=True, exist_ok=True)
    with open(path, "w") as f:
        json.dump(build_model_metadata(metadata), f, indent=1, sort_keys=True)
    return path


def read_metadata(model_path) -> Dict[str, str]:
    with open(metadata_path_for_model(model_path), "r") as f:
        raw = json.load(f)
    if not isinstance(raw, dict):
        raise ValueError("metadata manifest must be a JSON object")
    return build_model_metadata(raw)


_REQUIRED = object()


def _as_bool(text: str, key: str) -> bool:
    low = text.strip().lower()
    if low in _TRUE or low in _FALSE:
        return low in _TRUE
    raise ValueError(f"Metadata key {key} must be a boolean encoded as 1/0, got {text!r}.")


class MetadataReader:
    """Typed view over the manifest's string map.  One accessor, ``get(key, kind, default)``; the ``required_* / optional_*`` names the reference's
    drivers call (audio_onnx_metadata.py:247-287) are generated from it.  A missing REQUIRED key is a ``KeyError`` (the reference's contract),
    an unparsable value a ``ValueError``."""

    _KINDS = {"int": int, "float": float, "string": str}

    def __init__(self, metadata: Optional[Mapping[str, str]]):
        self.metadata = dict(metadata or {})

    def get(self, key: str, kind: str = "string", default=_REQUIRED):
        raw = self.metadata.get(key)
        if raw is None or raw == "":
            if default is _REQUIRED:
                raise KeyError(f"Required metadata key {key} is missing. Re-export the model to regenerate its manifest.")
            return default
        return _as_bool(str(raw), key) if kind == "bool" else self._KINDS[kind](raw)

    def string(self, key, default=None, required=False):
        return self.get(key, "string", _REQUIRED if required else default)

    def to_json(self) -> str:
        return json.dumps(self.metadata, sort_keys=True)


for _kind in ("int", "float", "bool"):
    setattr(MetadataReader, f"required_{_kind}", (lambda k: lambda self, key: self.get(key, k))(_kind))
    setattr(MetadataReader, f"optional_{_kind}", (lambda k: lambda self, key, default=None: self.get(key, k, default))(_kind))


def load_runtime_metadata(model_path, required_keys: Iterable[str] = REQUIRED_AUDIO_METADATA_KEYS) -> MetadataReader:
    """``FileNotFoundError`` if the manifest carrier is absent, ``KeyError`` per missing required key
    (audio_onnx_metadata.py:290-303)."""
    path = metadata_path_for_model(model_path)
    if not path.exists():
        raise FileNotFoundError(f"Required metadata manifest is missing: {path}. Re-export the model.")
    reader = MetadataReader(read_metadata(model_path))
    for key in required_keys:
        reader.string(key, required=True)
    return reader


def validate_audio_metadata(reader: MetadataReader, session) -> None:
    """Static input length / channels / #inputs of the session must agree with the manifest
    (``ValueError`` otherwise; audio_onnx_metadata.py:315-351)."""
    inputs, outputs = session.get_inputs(), session.get_outputs()
    if not inputs:
        return
    shape = list(inputs[0].shape)
    want_len = reader.optional_int("export_audio_length", reader.optional_int("input_audio_length"))
    if shape and isinstance(shape[-1], int) and want_len is not None and shape[-1] != want_len:
        raise ValueError(f"Model input length {shape[-1]} does not match metadata input length {want_len}.")
    want_in = reader.optional_int("input_channels")
    if want_in is not None and len(shape) >= 3 and isinstance(shape[-2], int) and shape[-2] != want_in:
        raise ValueError(f"Model input channels {shape[-2]} do not match metadata input_channels={want_in}.")
    want_out = reader.optional_int("output_channels")
    if want_out is not None and outputs:
        oshape = list(outputs[0].shape)
        if len(oshape) >= 3 and isinstance(oshape[-2], int) and oshape[-2] != want_out:
            raise ValueError(f"Model output channels {oshape[-2]} do not match metadata output_channels={want_out}.")
    n_in = reader.optional_int("num_audio_inputs")
    if n_in is not None and len(inputs) < n_in:
        raise ValueError(f"Model has {len(inputs)} inputs, metadata num_audio_inputs={n_in}.")


# manifest key -> (runtime constant of the reference's drivers, kind, default; _REQUIRED = must be present)   (audio_onnx_metadata.py:354-386)
_RUNTIME_FIELDS = (
    ("in_sample_rate", "IN_SAMPLE_RATE", "int", _REQUIRED), ("out_sample_rate", "OUT_SAMPLE_RATE", "int", _REQUIRED),
    ("model_sample_rate", "MODEL_SAMPLE_RATE", "int", _REQUIRED), ("input_to_output_scale", "INPUT_TO_OUTPUT_SCALE", "float", _REQUIRED),
    ("max_dynamic_audio_seconds", "MAX_DYNAMIC_AUDIO_SECONDS", "int", _REQUIRED), ("normalize_audio_default", "NORMALIZE_AUDIO", "bool", _REQUIRED),
    ("normalize_target_rms", "NORMALIZE_TARGET_RMS", "float", _REQUIRED), ("batch_window_seconds", "BATCH_WINDOW_SECONDS", "float", 0.0),
    ("hop_length", "HOP_LENGTH", "int", 0), ("fold_window_length", "FOLD_WINDOW_LENGTH", "int", 0),
    ("batch_fold_inference_default", "BATCH_FOLD_INFERENCE", "bool", False), ("input_channels", "INPUT_CHANNELS", "int", 1),
    ("output_channels", "OUTPUT_CHANNELS", "int", 1), ("input_channels", "N_CHANNELS", "int", 1), ("num_audio_inputs", "NUM_AUDIO_INPUTS", "int", 1),
    ("pad_head", "PAD_HEAD", "int", 0), ("enc_stride", "ENC_STRIDE", "int", 0), ("output_sources", "OUTPUT_SOURCES", "int", 1),
)


def runtime_config_from_metadata(reader: MetadataReader) -> Dict[str, Any]:
    """The UPPER_CASE runtime constants an inference script repopulates from the manifest; rate-derived defaults are filled in afterwards."""
    cfg = {const: reader.get(key, kind, default) for key, const, kind, default in _RUNTIME_FIELDS}
    in_sr, out_sr, model_sr, window = cfg["IN_SAMPLE_RATE"], cfg["OUT_SAMPLE_RATE"], cfg["MODEL_SAMPLE_RATE"], cfg["FOLD_WINDOW_LENGTH"]
    cfg["FOLD_INPUT_LENGTH"] = reader.get("fold_input_length", "int", max(1, int(round(window * in_sr / model_sr))) if window else 0)
    cfg["ORIGINAL_SAMPLE_RATE"] = reader.get("original_sample_rate", "int", in_sr)
    cfg["SUPER_SAMPLE_RATE"] = reader.get("super_sample_rate", "int", out_sr)
    cfg["SCALE_FACTOR"] = reader.get("scale_factor", "float", float(out_sr / in_sr))
    return cfg
