#!/bin/bash
# developer tool: libdcamd with csrc/tblock.hip rebuilt under extra -D flags (the other objects as built by `make`) -> tools/dev/_build/libdcamd_NAME.so
# usage: build_tb_alt.sh NAME [-DFLAG ...]
set -e
cd "$(dirname "$0")/../.."
name=$1; shift
src=diffusion-classifier_amd/csrc
mkdir -p tools/dev/_build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Iinclude -Wno-unused-function "$@" -c $src/tblock.hip -o tools/dev/_build/tblock_$name.o
objs=$(ls $src/build/*.o | grep -v '/tblock.o')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/dev/_build/libdcamd_$name.so $objs tools/dev/_build/tblock_$name.o
