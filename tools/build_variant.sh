#!/bin/bash
# Diagnostic: builds gnuspeech_amd/libtrm_var_NAME.so = the product library with trm_quad.hip (and, with WIDE=1,
# trm_kernels.hip) recompiled under extra -D flags.   usage: build_variant.sh NAME "-DTRM_EXPERIMENTS -D..."
set -e
cd "$(dirname "$0")/../gnuspeech_amd/csrc"
make -s
NAME=$1; shift
FLAGS="-O3 -std=c++17 -fPIC -Wall -Wno-unused-function -fno-slp-vectorize"
mkdir -p build/var_$NAME
hipcc --offload-arch=gfx950 $FLAGS -mllvm -amdgpu-sched-strategy=iterative-ilp "$@" -c trm_quad.hip -o build/var_$NAME/trm_quad.o
OBJS="build/trm_kernels.o build/trm_tracks.o build/trm_capi.o build/trm_setup.o build/trm_io.o"
if [ -n "$WIDE" ]; then hipcc --offload-arch=gfx950 $FLAGS "$@" -c trm_kernels.hip -o build/var_$NAME/trm_kernels.o; OBJS="build/var_$NAME/trm_kernels.o build/trm_tracks.o build/trm_capi.o build/trm_setup.o build/trm_io.o"; fi
hipcc --offload-arch=gfx950 -shared -o ../libtrm_var_$NAME.so build/var_$NAME/trm_quad.o $OBJS
echo built ../libtrm_var_$NAME.so
