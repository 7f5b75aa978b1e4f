set -e
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
O=gpurun_out/pmc_wide
rm -rf $O; mkdir -p $O
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_LDS --output-format csv -d $O/pmc_sq -o run -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --voices 65536 > $O/log.txt 2>&1
python tools/rocprof_summary.py $O/summary.txt $O/none $O/pmc_sq > /dev/null
cat $O/summary.txt
rm -rf $O/pmc_sq
