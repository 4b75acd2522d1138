#!/bin/bash
# Developer tool: build a variant of the HIP library with extra compiler definitions (timing-only ablations EINCM_ABL_*, experiments)
# into tools/variants/libeincm_<name>.so; run anything against it with EINCM_LIB=tools/variants/libeincm_<name>.so.
# usage: tools/build_variant.sh <name> "<extra hipcc flags>"
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $ROOT/tools/variants
hipcc -O3 --offload-arch=gfx950 -std=c++17 -shared -fPIC -fno-slp-vectorize -Wall -Wno-unused-function $2 \
  -I $ROOT/include -o $ROOT/tools/variants/libeincm_$1.so $ROOT/edge-informed-contrast-maximization_amd/csrc/eincm_api.hip
echo built tools/variants/libeincm_$1.so
