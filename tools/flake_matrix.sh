#!/bin/bash
# builds the -DGDSP_STREAMING=0 variant of the library (and a copy of the driver beside it: its rpath starts at $ORIGIN)
# here, where hipcc cross-compiles; the GPU box then runs tools/flake_matrix.py
set -e
cd "$(dirname "$0")/.."
mkdir -p build/variant_nostream
make -s -C genodsp_amd/csrc BUILD=../../build/variant_nostream_obj OUT=../../build/variant_nostream/libgenodsp_hip.so \
     HIPFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-result -I../../include -DGDSP_STREAMING=0" -j6
make -s -C genodsp_amd/host
cp genodsp_amd/genodsp_hip build/variant_nostream/genodsp_hip
