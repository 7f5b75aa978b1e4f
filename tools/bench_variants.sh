#!/bin/bash
# Diagnostic: bench every gnuspeech_amd/libtrm_var_*.so (kernel experiments built with -D overrides).
for f in gnuspeech_amd/libtrm_var_*.so; do
  r=$(TRM_LIB=$PWD/$f python bench.py --steps 5 --warmup 2 --no-cpu-baseline --kernel ${KERNEL:-quad} ${BENCH_ARGS} 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.3f ms  %.3e samples/s"%(d["ms_per_step"], d["value"]))')
  echo "$f $r"
done
