#!/bin/bash
# copies what tools/profile_round.sh left in gpurun_out/prof_$TAG into profiles/ under the round's names and re-stamps
# profiles/traffic_r02.json with the current kernel sources' hash.   usage: collect_profiles.sh TAG [issue cycles per VALU]
set -e
cd "$(dirname "$0")/.."
O=gpurun_out/prof_$1; CYC=${2:-3.40}
cp $O/summary.txt profiles/rocprof_r02_summary.txt
cp $O/kernel_stats.csv profiles/rocprof_r02_kernel_stats.csv
cp $O/stage_profile_oct.txt profiles/stage_profile_r02_oct.txt
cp $O/stage_profile_quad.txt profiles/stage_profile_r02_quad.txt
cp $O/stage_profile_quad_8192.txt profiles/stage_profile_r02_quad_8192voices.txt
cp $O/stage_profile_wide.txt profiles/stage_profile_r02_wide.txt
cp $O/sweep_forms.txt profiles/sweep_forms_r02.txt
cp $O/sweep_oct.txt profiles/sweep_oct_r02.txt
cp $O/configs.txt profiles/configs_r02.txt
cp $O/host_path.txt profiles/host_path_r02.txt
cp $O/bench_8192_timevarying.json profiles/bench_r02_8192voices_timevarying.json
cp $O/bench_4096_timevarying.json profiles/bench_r02_4096voices_timevarying.json
cp $O/bench_65536.json profiles/bench_r02_saturated_65536voices.json
cp $O/bench_wide4096.json profiles/bench_r02_wide_form_4096voices.json
cp $O/bench_quad4096.json profiles/bench_r02_quad_form_4096voices.json
python tools/make_traffic.py profiles/rocprof_r02_summary.txt 4096 251 static oct $CYC profiles/traffic_r02.json | grep -E "sha16|SQ_INSTS|traffic_bytes"
