#!/bin/bash
# A variant of the product library for same-box A/B runs: ONE source re-compiled with extra -D flags, linked with the other objects of the last build().
# Usage: tools/build_variant.sh <name> <source stem, e.g. ade_zipenhancer> [-DFLAG=1 ...]     ->  _ab/libade_<name>.so
set -e
N=$1; S=$2; shift 2
C=audio_denoiser_onnx_amd/csrc; mkdir -p _ab
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -ffp-contract=fast -c -I include "$@" $C/$S.hip -o _ab/$S.$N.o
OBJS=$(ls $C/_obj/*.o | grep -v "/$S.hip.o")
hipcc --offload-arch=gfx950 -shared -fPIC $OBJS _ab/$S.$N.o -o _ab/libade_$N.so
echo "_ab/libade_$N.so"
