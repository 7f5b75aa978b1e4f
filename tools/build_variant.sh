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
# per-file scheduler flags as in the Makefile (QUADFLAGS / OCTFLAGS / WIDEFLAGS); SCHED=<strategy> overrides the strategy, SCHED=none = the
# compiler's default scheduling
case $SRC in
  trm_kernels) DEF="-mllvm -amdgpu-sched-strategy=${SCHED:-max-ilp} -mllvm -enable-post-misched=false" ;;
  trm_quad)    DEF="-mllvm -amdgpu-sched-strategy=${SCHED:-iterative-ilp} -mllvm -enable-post-misched=false" ;;
  *)           DEF="-mllvm -amdgpu-sched-strategy=${SCHED:-iterative-ilp}" ;;
esac
[ "$SCHED" != none ] && FLAGS="$FLAGS $DEF"
mkdir -p build/var_$NAME
hipcc --offload-arch=gfx950 $FLAGS "$@" -c $SRC.hip -o build/var_$NAME/$SRC.o
OBJS=""
for o in trm_kernels trm_quad trm_oct trm_tracks trm_capi trm_setup trm_io; do
  if [ $o = $SRC ]; then OBJS="$OBJS build/var_$NAME/$o.o"; else OBJS="$OBJS build/$o.o"; fi
done
hipcc --offload-arch=gfx950 -shared -o ../libtrm_var_$NAME.so $OBJS
echo built ../libtrm_var_$NAME.so
