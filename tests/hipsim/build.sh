#!/bin/bash
# Build the host-simulated engine (TEST INFRASTRUCTURE): same csrc/*.hip sources, g++ + tests/hipsim shim.
set -e
cd "$(dirname "$0")/../.."
mkdir -p tests/hipsim/_build
g++ -std=c++17 -O2 -g -fPIC -shared -Wall -Wno-unused-function -Wno-unknown-pragmas -I tests/hipsim \
    -x c++ audio_denoiser_onnx_amd/csrc/ade_kernels.hip audio_denoiser_onnx_amd/csrc/ade_fused.hip audio_denoiser_onnx_amd/csrc/ade_engine.hip audio_denoiser_onnx_amd/csrc/ade_stft.hip audio_denoiser_onnx_amd/csrc/ade_dfsmn.hip audio_denoiser_onnx_amd/csrc/ade_melband.hip audio_denoiser_onnx_amd/csrc/ade_mossformer.hip audio_denoiser_onnx_amd/csrc/ade_ulunas.hip audio_denoiser_onnx_amd/csrc/ade_hgtcrn.hip audio_denoiser_onnx_amd/csrc/ade_zipenhancer.hip \
    -x c++ tests/hipsim/hipsim.cpp -o tests/hipsim/_build/libade_hipsim.so
