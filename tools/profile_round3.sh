#!/bin/bash
# Round 3's measurements on the GPU box -> gpurun_out/prof_r03/ (copy what is to be judged into profiles/):
# kernel trace + PMC passes per workload and kernel form (tools/profile_workload.sh), stage profiles, the batch-size sweep,
# randomized parity / stream runs, the real-time voice count.
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
R=gpurun_out/prof_r03
mkdir -p $R
bash tools/profile_workload.sh r03_oct4096 > $R/prof_oct4096.log 2>&1
bash tools/profile_workload.sh r03_oct4096_tv --config 2 > $R/prof_oct4096_tv.log 2>&1
bash tools/profile_workload.sh r03_quad8192_tv --config 4 > $R/prof_quad8192_tv.log 2>&1
bash tools/profile_workload.sh r03_wide65536 --voices 65536 --kernel wide > $R/prof_wide65536.log 2>&1
TRM_SUMMARY_DISPATCHES=2 bash tools/profile_workload.sh r03_wide131072 --voices 131072 --kernel wide > $R/prof_wide131072.log 2>&1
echo "profiles done"
for w in "4096 0.25 static oct" "8192 0.25 timevarying quad" "65536 0.25 static wide" "12288 0.25 static wide"; do
  set -- $w; python tools/stage_profile.py $w > $R/stage_$4_$1.txt 2>/dev/null
done
echo "stage profiles done"
for v in 1024 2048 4096 6144 8192 10240 12288 14336 16384 32768 65536 98304 131072; do
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline --voices $v 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%6d voices  %s  %.3f ms  %.3e samples/s  %.2f %% of HBM peak"%(d["config"]["voices_per_gpu"], d["config"]["kernel_form"], d["ms_per_step"], d["value"], 100*d["roofline"]["frac"]))'
done > $R/sweep_auto.txt
cat $R/sweep_auto.txt
python tools/fuzz_parity.py 0 150 > $R/fuzz_parity.txt 2>&1; tail -2 $R/fuzz_parity.txt
python tools/fuzz_parity.py 0 60 300 broad > $R/fuzz_parity_broad.txt 2>&1; tail -2 $R/fuzz_parity_broad.txt
TRM_TUBE_KERNEL=quad python tools/fuzz_stream.py 0 200 > $R/fuzz_stream_quad.txt 2>&1; tail -1 $R/fuzz_stream_quad.txt
TRM_TUBE_KERNEL=wide python tools/fuzz_stream.py 0 200 > $R/fuzz_stream_wide.txt 2>&1; tail -1 $R/fuzz_stream_wide.txt
python tools/bench_configs.py > $R/configs.txt 2>/dev/null; tail -8 $R/configs.txt
