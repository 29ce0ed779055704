#!/bin/bash
# developer tool: libdcamd with ONE source rebuilt under extra flags (the other objects as built by `make`) -> tools/dev/_build/libdcamd_NAME.so
# usage: build_one_alt.sh NAME FILE.hip [flags ...]      e.g.  build_one_alt.sh xreg_noslp igemm_xreg.hip -fno-slp-vectorize
set -e
cd "$(dirname "$0")/../.."
name=$1; file=$2; shift 2
src=diffusion-classifier_amd/csrc
mkdir -p tools/dev/_build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Iinclude -Wno-unused-function "$@" -c $src/$file -o tools/dev/_build/${file%.hip}_$name.o
objs=$(ls $src/build/*.o | grep -v "/${file%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/dev/_build/libdcamd_$name.so $objs tools/dev/_build/${file%.hip}_$name.o
