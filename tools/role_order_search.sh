#!/bin/bash
# Diagnostic (profiles/ab_r03.txt, "wave order and issue priorities"): times the one-voice-per-lane kernel under other wave orders and
# role priorities.  Two steps, because the GPU box has no spare cores for 50 compilations:
#   here:        tools/role_order_search.sh build LIST   LIST: lines "NAME -DTRM_ROLE_PERM=5,6,4,0,2,3,1 -DTRM_PRIO_CVT=3 ..." (any of the
#                                                         TRM_EXPERIMENTS switches of trm_kernels.hip) -> gnuspeech_amd/libtrm_var_NAME.so each
#   on the box:  tools/role_order_search.sh run LIST     -> one line per variant: ms per launch at 65 536 and 12 288 voices
set -e
cd "$(dirname "$0")/.."
MODE=$1; LIST=$2
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=max-ilp -mllvm -enable-post-misched=false -DTRM_EXPERIMENTS"
one() { TRM_LIB=$PWD/gnuspeech_amd/libtrm_var_$1.so python bench.py --steps 20 --warmup 3 --no-cpu-baseline --kernel wide --voices $2 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.3f"%d["roofline"]["avg_launch_ms"])'; }
if [ "$MODE" = build ]; then
  make -s -C gnuspeech_amd/csrc
  cp gnuspeech_amd/libtrm_hip.so gnuspeech_amd/libtrm_var_product.so
  build_one() {
    n=$1; shift; cd gnuspeech_amd/csrc; mkdir -p build/var_$n
    hipcc $FLAGS "$@" -c trm_kernels.hip -o build/var_$n/trm_kernels.o 2>/dev/null
    hipcc --offload-arch=gfx950 -shared -o ../libtrm_var_$n.so build/var_$n/trm_kernels.o build/trm_quad.o build/trm_oct.o build/trm_tracks.o build/trm_capi.o build/trm_setup.o build/trm_io.o
  }
  export -f build_one; export FLAGS
  xargs -P 6 -L 1 bash -c 'build_one "$@"' _ < "$LIST"
else
  echo "product - $(one product 65536) $(one product 12288)"
  while read n rest; do echo "$n $rest $(one $n 65536) $(one $n 12288)"; done < "$LIST"
  echo "product - $(one product 65536) $(one product 12288)"
fi
