#!/bin/bash
# rocprofv3 kernel trace + HBM / instruction PMC passes of configs[4]'s per-GPU batch (8192 time-varying voices: the
# one-block-per-step instance, two workgroups per CU) -> gpurun_out/prof_8192/summary.txt
set -e
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
O=gpurun_out/prof_8192
rm -rf $O; mkdir -p $O
B="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --voices 8192 --workload timevarying"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o run -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --voices 8192 --workload timevarying > $O/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o run -- $B > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o run -- $B > $O/pmc_write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/pmc_sq -o run -- $B > $O/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_lds -o run -- $B > $O/pmc_lds.log 2>&1
python tools/rocprof_summary.py $O/summary.txt $O/trace $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_lds > /dev/null
rm -rf $O/trace $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_lds $O/*.log
cat $O/summary.txt
