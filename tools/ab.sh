#!/bin/bash
# usage: ab.sh "<bench args>" lib1 lib2 ... : alternates the libraries, 3 rounds, prints ms per step
ARGS="$1"; shift
for round in 1 2 3; do
  for L in "$@"; do
    r=$(TRM_LIB=$PWD/gnuspeech_amd/libtrm_var_$L.so python bench.py --steps 30 --warmup 3 --no-cpu-baseline $ARGS 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.3f ms kernel %.3f ms  %.3e /s %s"%(d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["value"], d["config"]["kernel_form"]))')
    echo "[$ARGS] $L: $r"
  done
done
