"""Import the reference's GTCRN path in THIS container (never on the GPU box).

Used only by the golden-vector generators under tools/.  Nothing here is
shipped or imported by the product; /root/reference is read-only and absent
on the GPU box.  Recipe = SURVEY.md section 8(c1):

* ``onnxruntime`` / ``onnx`` are absent and ``STFT_Process.py`` imports
  onnxruntime at module top, so both are stubbed with empty modules;
* ``Export_GTCRN.py`` runs its export at module level, so only its class
  definitions and UPPER_CASE constants are executed (``ast`` filter), with
  ``INPUT_AUDIO_LENGTH`` overridden to the requested static chunk length.
"""
from __future__ import annotations

import ast
import os
import sys
import types

REF_ROOT = os.environ.get("ADE_REFERENCE_ROOT", "/root/reference")


def _stub_absent_modules():
    for name in ("onnxruntime", "onnx", "onnxslim"):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                sys.modules[name] = types.ModuleType(name)
                if name == "onnxslim":      # `from onnxslim import slim` at the top of some STFT_Process copies
                    sys.modules[name].slim = lambda *a, **k: None


def import_stft_process(model_dir: str = "GTCRN"):
    """Return the ``STFT_Process`` module of ``<reference>/<model_dir>``."""
    sys.dont_write_bytecode = True
    _stub_absent_modules()
    import importlib.util

    path = os.path.join(REF_ROOT, model_dir, "STFT_Process.py")
    spec = importlib.util.spec_from_file_location(f"ref_stft_{model_dir.replace('/', '_')}", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def import_gtcrn_namespace(input_audio_length: int = 16000, overrides: dict = None) -> dict:
    """Exec the class defs + constants of GTCRN/Export_GTCRN.py; return the namespace."""
    import numpy as np
    import torch
    import torch.nn as nn

    sys.dont_write_bytecode = True
    _stub_absent_modules()
    path = os.path.join(REF_ROOT, "GTCRN", "Export_GTCRN.py")
    with open(path, "r") as f:
        tree = ast.parse(f.read(), filename=path)
    keep = []
    for node in tree.body:
        if isinstance(node, ast.ClassDef):
            keep.append(node)
        elif isinstance(node, ast.Assign):
            names = [t.id for t in node.targets if isinstance(t, ast.Name)]
            if names and all(n.isupper() or "_" in n and n.upper() == n for n in names):
                if names == ["INPUT_AUDIO_LENGTH"]:
                    node = ast.parse(f"INPUT_AUDIO_LENGTH = {int(input_audio_length)}").body[0]
                elif overrides and len(names) == 1 and names[0] in overrides:      # e.g. {"USE_BATCH_FOLD": True}
                    node = ast.parse(f"{names[0]} = {overrides[names[0]]!r}").body[0]
                keep.append(node)
    module = ast.Module(body=keep, type_ignores=[])
    ast.fix_missing_locations(module)
    ns = {"np": np, "torch": torch, "nn": nn, "__name__": "ref_export_gtcrn"}
    exec(compile(module, path, "exec"), ns)
    return ns
