#!/bin/bash
# Diagnostic: builds gnuspeech_amd/libtrm_var_NAME.so = the product library with ONE kernel source recompiled under
# extra -D flags: trm_quad.hip by default, SRC=trm_oct / SRC=trm_kernels picks another (WIDE=1 = SRC=trm_kernels).
#   usage: [SRC=trm_oct] [SCHED=max-ilp] build_variant.sh NAME "-DTRM_EXPERIMENTS -D..."
set -e
cd "$(dirname "$0")/../gnuspeech_amd/csrc"
make -s
NAME=$1; shift
SRC=${SRC:-trm_quad}
[ -n "$WIDE" ] && SRC=trm_kernels
FLAGS="-O3 -std=c++17 -fPIC -Wall -Wno-unused-function -fno-slp-vectorize"
SCHED=${SCHED:-iterative-ilp}      # the scheduler strategy of the multi-lane kernels (Makefile: QUADFLAGS); SCHED=none = the compiler's default
[ "$SRC" != trm_kernels ] && [ "$SCHED" != none ] && FLAGS="$FLAGS -mllvm -amdgpu-sched-strategy=$SCHED"
mkdir -p build/var_$NAME
hipcc --offload-arch=gfx950 $FLAGS "$@" -c $SRC.hip -o build/var_$NAME/$SRC.o
OBJS=""
for o in trm_kernels trm_quad trm_oct trm_tracks trm_capi trm_setup trm_io; do
  if [ $o = $SRC ]; then OBJS="$OBJS build/var_$NAME/$o.o"; else OBJS="$OBJS build/$o.o"; fi
done
hipcc --offload-arch=gfx950 -shared -o ../libtrm_var_$NAME.so $OBJS
echo built ../libtrm_var_$NAME.so
