#!/bin/bash
# Collects the round's measurements on the GPU box into gpurun_out/prof_$TAG (copy what is to be judged into profiles/):
#   bench lines (default = BASELINE configs[1]; configs[4]'s per-GPU batch; the other kernel form; a saturating batch),
#   rocprofv3 kernel trace + stats, PMC passes (HBM bytes, instruction classes, LDS) each in its own run, the stage
#   profile, the form sweep, the other configs, the host path.
set -e
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
TAG=${1:-r02}
O=gpurun_out/prof_$TAG
rm -rf $O; mkdir -p $O
B="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o run -- python3 bench.py --no-cpu-baseline > $O/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o run -- $B > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o run -- $B > $O/pmc_write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/pmc_sq -o run -- $B > $O/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_lds -o run -- $B > $O/pmc_lds.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_INT32 --output-format csv -d $O/pmc_cls -o run -- $B > $O/pmc_cls.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT64 SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU --output-format csv -d $O/pmc_cls2 -o run -- $B > $O/pmc_cls2.log 2>&1
python tools/rocprof_summary.py $O/summary.txt $O/trace $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_lds $O/pmc_cls $O/pmc_cls2 > /dev/null
cp $O/trace/run_kernel_stats.csv $O/kernel_stats.csv
python bench.py --steps 10 --warmup 2 > $O/bench_default.json 2>$O/bench_default.err
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --voices 8192 --workload timevarying > $O/bench_8192_timevarying.json 2>/dev/null
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --workload timevarying > $O/bench_4096_timevarying.json 2>/dev/null
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --voices 65536 > $O/bench_65536.json 2>/dev/null
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --kernel wide > $O/bench_wide4096.json 2>/dev/null
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --kernel quad > $O/bench_quad4096.json 2>/dev/null
bash tools/sweep_oct.sh > $O/sweep_oct.txt 2>/dev/null
python tools/stage_profile.py 4096 0.25 static oct > $O/stage_profile_oct.txt 2>/dev/null
python tools/stage_profile.py 4096 0.25 static quad > $O/stage_profile_quad.txt 2>/dev/null
python tools/stage_profile.py 8192 0.25 static quad > $O/stage_profile_quad_8192.txt 2>/dev/null
python tools/stage_profile.py 4096 0.25 static wide > $O/stage_profile_wide.txt 2>/dev/null
bash tools/sweep_forms.sh > $O/sweep_forms.txt 2>/dev/null
python tools/bench_configs.py > $O/configs.txt 2>/dev/null
python tools/host_path_rate.py > $O/host_path.txt 2>/dev/null
tools/ubench/valu_ceiling > $O/valu_ceiling.txt 2>/dev/null || true
rm -rf $O/trace $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_lds $O/pmc_cls $O/pmc_cls2 $O/*.log
cat $O/summary.txt; cat $O/stage_profile_oct.txt; cat $O/sweep_forms.txt; cut -c1-250 $O/bench_default.json $O/bench_65536.json $O/bench_wide4096.json
