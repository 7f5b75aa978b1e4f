#!/bin/bash
# copies what tools/profile_round3.sh left in gpurun_out/ into profiles/ under the round's names and rebuilds
# profiles/traffic_r03.json (one entry per profiled workload, stamped with the current kernel sources' hash).
set -e
cd "$(dirname "$0")/.."
R=gpurun_out/prof_r03
for t in oct4096 oct4096_tv quad8192_tv wide65536 wide131072; do
  cp gpurun_out/prof_r03_$t/summary.txt profiles/rocprof_r03_$t.txt
  cp gpurun_out/prof_r03_$t/bench.json profiles/bench_r03_$t.json
done
cp gpurun_out/prof_r03_oct4096/kernel_stats.csv profiles/rocprof_r03_kernel_stats.csv
cp gpurun_out/prof_r03_oct4096/summary.txt profiles/rocprof_r03_summary.txt
cp $R/stage_oct_4096.txt profiles/stage_profile_r03_oct.txt
cp $R/stage_quad_8192.txt profiles/stage_profile_r03_quad_8192voices.txt
cp $R/stage_wide_65536.txt profiles/stage_profile_r03_wide_65536voices.txt
cp $R/stage_wide_12288.txt profiles/stage_profile_r03_wide_12288voices.txt
cp $R/sweep_auto.txt profiles/sweep_forms_r03.txt
cp $R/configs.txt profiles/configs_r03.txt
rm -f profiles/traffic_r03.json
for a in "oct4096 4096 static oct" "oct4096_tv 4096 timevarying oct" "quad8192_tv 8192 timevarying quad" "wide65536 65536 static wide" "wide131072 131072 static wide"; do
  set -- $a
  python tools/make_traffic.py profiles/rocprof_r03_$1.txt $2 251 $3 $4 auto > /dev/null
done
echo "profiles/traffic_r03.json: $(python -c "import json; print(len(json.load(open('profiles/traffic_r03.json'))['entries']))") entries"
