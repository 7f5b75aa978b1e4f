#!/bin/bash
# on the GPU box: times every variant library named in gnuspeech_amd/csrc/build/variants.list (bench batch, ms per launch;
# VOICES / WORKLOAD override the batch)
while read -r name rest; do
  [ -z "$name" ] && continue
  r=$(TRM_LIB=gnuspeech_amd/libtrm_var_$name.so python bench.py --steps 5 --warmup 2 --no-cpu-baseline ${VOICES:+--voices $VOICES} ${WORKLOAD:+--workload $WORKLOAD} 2>/dev/null | python -c 'import sys,json; print("%.3f" % json.loads(sys.stdin.read())["ms_per_step"])')
  echo "$name  $rest  $r ms"
done < gnuspeech_amd/csrc/build/variants.list
