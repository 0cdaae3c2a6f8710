#!/bin/bash
# build_variant.sh NAME [-DJDSP_...=.. ...]  ->  build/variants/NAME.so: a complete libjdsp.so with extra flags
# (timing-only A/B builds; tools/tune_stft.py, tools/cold_input_probe.py and JDSP_LIB= load them).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
mkdir -p "$R/build/variants"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared "$@" -o "$R/build/variants/$name.so" "$R"/jeicyboodsp_amd/csrc/*.hip
echo "built build/variants/$name.so ($*)"
