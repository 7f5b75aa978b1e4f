set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/prof_r01c
rm -rf $O; mkdir -p $O
python bench.py --steps 5 --warmup 2 > $O/bench_default.json 2>$O/bench_default.err
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --voices 65536 > $O/bench_65536.json 2>/dev/null
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --kernel wide > $O/bench_wide4096.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o run -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o run -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o run -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/pmc_sq -o run -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_lds -o run -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_lds.log 2>&1
python tools/rocprof_summary.py $O/summary.txt $O/trace $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_lds > /dev/null
find $O -name "*.db" -delete; find $O -name "*_kernel_trace.csv" -size +2M -delete
cat $O/summary.txt | head -60
cat $O/bench_default.json
