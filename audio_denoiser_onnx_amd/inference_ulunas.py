#!/usr/bin/env python3
"""The reference's ``UL-UNAS/Inference_UL_UNAS_ONNX.py`` call surface on the MI355X engine.

    python -m audio_denoiser_onnx_amd.inference_ulunas <model_dir_or_.adew> [noisy_16k.wav] [denoised.wav]

The reference driver has GTCRN's skeleton (mono int16 at 16 kHz, slices of the static input length at a stride of the output length
when they differ :328-334, zero-padded tail :145,165, concatenate, trim), so this is ``inference_gtcrn``'s loop on a ``ul_unas`` model:
all slices of the file go to the GPU as one batch.
"""
from __future__ import annotations

import sys

from . import inference_gtcrn
from .session import resolve_model_path
from .metadata import load_runtime_metadata


def main(argv=None) -> int:
    argv = list(sys.argv[1:] if argv is None else argv)
    args = [a for a in argv if not a.startswith("--")]
    if not args:
        print(__doc__)
        return 2
    family = load_runtime_metadata(resolve_model_path(args[0])).metadata.get("model_family")
    if family != "ul_unas":
        raise ValueError(f"this driver expects a model_family=ul_unas manifest, got {family!r}")
    if len(args) == 1:
        argv = argv + [str(inference_gtcrn.example_audio("denoise", "ul_unas_0174.wav"))]
    return inference_gtcrn.main(argv)


if __name__ == "__main__":
    raise SystemExit(main())
