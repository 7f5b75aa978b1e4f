#!/bin/bash
# rocprofv3 kernel trace + the PMC passes (HBM bytes, instruction counts, LDS; each in its own run) of ONE bench workload
# on the GPU box -> gpurun_out/prof_$TAG/summary.txt (+ the bench line of the same command).
#   usage: [TRM_SUMMARY_DISPATCHES=2] tools/profile_workload.sh TAG <bench.py workload flags...>       e.g.  wide65536 --voices 65536 --kernel wide
#   (TRM_SUMMARY_DISPATCHES: dispatches per launch -- 2 for 131 072 voices of the one-voice-per-lane kernel, tools/rocprof_summary.py)
# tools/make_traffic.py turns summary.txt into an entry of profiles/traffic_r04.json.
set -e
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
TAG=$1; shift
O=gpurun_out/prof_$TAG
rm -rf $O; mkdir -p $O
B="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o run -- python3 bench.py --steps 25 --warmup 3 --no-cpu-baseline "$@" > $O/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o run -- $B > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o run -- $B > $O/pmc_write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/pmc_sq -o run -- $B > $O/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_lds -o run -- $B > $O/pmc_lds.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_INT32 --output-format csv -d $O/pmc_cls -o run -- $B > $O/pmc_cls.log 2>&1
python tools/rocprof_summary.py $O/summary.txt $O/trace $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_lds $O/pmc_cls > /dev/null
cp $O/trace/run_kernel_stats.csv $O/kernel_stats.csv 2>/dev/null || true
grep '^{' $O/trace.log > $O/bench_under_trace.json || true
python bench.py --steps 50 --warmup 5 --no-cpu-baseline "$@" > $O/bench.json 2>$O/bench.err
rm -rf $O/trace $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_lds $O/pmc_cls $O/pmc_*.log
echo "== $TAG: bench.py $*"; cat $O/summary.txt; cut -c1-400 $O/bench.json
