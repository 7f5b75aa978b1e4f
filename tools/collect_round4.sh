#!/bin/bash
# copies what tools/profile_round4.sh left in gpurun_out/ into profiles/ under the round's names and rebuilds
# profiles/traffic_r04.json (one entry per profiled workload, stamped with the current kernel sources' hash).
set -e
cd "$(dirname "$0")/.."
R=gpurun_out/prof_r04
for t in config1 config1_whole config2 config3 config4 config4_whole wide65536; do
  cp gpurun_out/prof_r04_$t/summary.txt profiles/rocprof_r04_$t.txt
  cp gpurun_out/prof_r04_$t/kernel_stats.csv profiles/rocprof_r04_${t}_kernel_stats.csv
done
cp $R/single_voice.txt profiles/single_voice_r04.txt
cp $R/rates.txt profiles/rates_r04.txt
cp $R/sweep_auto.txt profiles/sweep_forms_r04.txt
cat $R/fuzz_parity.txt $R/fuzz_parity_broad.txt | grep -v "^seed .*oracle refuses" | tail -12 > profiles/fuzz_r04.txt
(echo "tools/fuzz_stream.py 0 60 and 0 30 tract, TRM_TUBE_KERNEL=wide then quad (chunked == single push bit for bit, == one-shot / oracle to rounding):"; cat $R/fuzz_stream.txt) >> profiles/fuzz_r04.txt
rm -f profiles/traffic_r04.json
# <tag> <voices> <frames per voice (ragged: the longest)> <kind> <kernel form>
for a in "config1 4096 251 static wide/split" "config1_whole 4096 251 static oct" "config2 4096 251 timevarying wide/split" "config3 1024 1498 ragged wide/split" "config4 8192 251 timevarying wide/split" "config4_whole 8192 251 timevarying quad" "wide65536 65536 251 static wide"; do
  set -- $a
  python tools/make_traffic.py profiles/rocprof_r04_$1.txt $2 $3 $4 $5 mix > /dev/null
done
echo "profiles/traffic_r04.json: $(python -c "import json; print(len(json.load(open('profiles/traffic_r04.json'))['entries']))") entries"
# the bench lines (they quote the traffic file just built): run tools/bench_lines_r04.sh on the GPU box next
