#!/bin/bash
# developer tool: build a second libdcamd into gpurun_out/ with extra -D flags (A/B of compile-time variants in one gpurun session)
# usage: build_alt.sh NAME -DFLAG...   -> gpurun_out/libdcamd_NAME.so
name=$1; shift
src=diffusion-classifier_amd/csrc
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -shared -Wno-unused-function "$@" -Iinclude -o gpurun_out/libdcamd_$name.so $src/*.hip
