#!/bin/bash
# Diagnostic: build a variant of the product library into build_ab/<name>.so with extra compiler flags, for tools/ab.py
# usage: tools/build_variant.sh <name> [flags ...]
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build_ab
S=corintho_ai_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
  -fno-fast-math -Wno-unused-function -Wno-unused-variable -Wno-unknown-pragmas "$@" -o build_ab/$name.so \
  $S/engine.hip $S/nn_mlp.hip $S/nn_mlp_split.hip $S/nn_rescnn.hip
echo build_ab/$name.so
