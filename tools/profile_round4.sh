#!/bin/bash
# Round 4's measurements on the GPU box -> gpurun_out/prof_r04/ (tools/collect_round4.sh copies what is to be judged into
# profiles/): kernel trace + PMC passes per BASELINE config (tools/profile_workload.sh), the bench line of every config,
# the batch-size sweep with AUTO (time split on) and with whole utterances, randomized parity incl. time-split launches.
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
R=gpurun_out/prof_r04
mkdir -p $R
# (a time-split launch = two trm_tube_kernel dispatches: the segment instance and the gated whole-utterance one that returns at once)
TRM_SUMMARY_DISPATCHES=2 bash tools/profile_workload.sh r04_config1 --no-stream > $R/prof_config1.log 2>&1
TRM_SUMMARY_DISPATCHES=2 bash tools/profile_workload.sh r04_config2 --config 2 > $R/prof_config2.log 2>&1
bash tools/profile_workload.sh r04_config1_whole --no-stream --split off > $R/prof_config1_whole.log 2>&1
TRM_SUMMARY_DISPATCHES=2 bash tools/profile_workload.sh r04_config3 --config 3 --no-end-to-end > $R/prof_config3.log 2>&1
TRM_SUMMARY_DISPATCHES=2 bash tools/profile_workload.sh r04_config4 --config 4 > $R/prof_config4.log 2>&1
bash tools/profile_workload.sh r04_config4_whole --config 4 --split off > $R/prof_config4_whole.log 2>&1
bash tools/profile_workload.sh r04_wide65536 --voices 65536 --kernel wide > $R/prof_wide65536.log 2>&1
echo "profiles done"
for v in 64 256 512 1024 1536 2048 3072 4096 5120 6144 7168 8192 9216 10240 11264 12288 13312 14336 15360 16384 20480 24576 28672 32768 49152 65536 98304 131072; do
  for sp in auto off; do
    python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-stream --voices $v --split $sp 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%6d voices  split %-4s %-10s %s  %.3f ms  %.3e samples/s  %.2f %% of HBM peak"%(d["config"]["voices_per_gpu"], sys.argv[1], d["config"]["kernel_form"], d["config"]["time_split"] or "", d["ms_per_step"], d["value"], 100*d["roofline"]["frac"]))' $sp
  done
done > $R/sweep_auto.txt
cat $R/sweep_auto.txt
python tools/single_voice_latency.py > $R/single_voice.txt 2>&1; tail -6 $R/single_voice.txt
python tools/bench_rates.py > $R/rates.txt 2>&1; head -5 $R/rates.txt
python tools/fuzz_parity.py 0 120 > $R/fuzz_parity.txt 2>&1; tail -4 $R/fuzz_parity.txt
python tools/fuzz_parity.py 0 60 300 broad > $R/fuzz_parity_broad.txt 2>&1; tail -4 $R/fuzz_parity_broad.txt
for f in wide quad; do TRM_TUBE_KERNEL=$f python tools/fuzz_stream.py 0 60 2>&1 | tail -1; TRM_TUBE_KERNEL=$f python tools/fuzz_stream.py 0 30 tract 2>&1 | tail -1; done > $R/fuzz_stream.txt; cat $R/fuzz_stream.txt
