#!/bin/bash
# what does SQ_ACTIVE_INST_VALU count?  The VALU microbenchmark (tools/ubench/valu_ceiling: one
# workgroup on one CU, W = 1..4 waves per SIMD, every wave the same stream of one instruction class, in-kernel cycle count
# known) under rocprofv3 --pmc; tools/valu_pmc_join.py puts the counters beside the benchmark's own cycles.
set -e
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
O=gpurun_out/valu_pmc
rm -rf $O; mkdir -p $O
hipcc --offload-arch=gfx950 -O2 -o tools/ubench/valu_ceiling tools/ubench/valu_ceiling.hip
./tools/ubench/valu_ceiling > $O/plain.txt
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/pmc -o run -- ./tools/ubench/valu_ceiling > $O/under_pmc.txt 2>$O/pmc.err
python3 tools/valu_pmc_join.py $O/under_pmc.txt $O/pmc > $O/calibration.txt
cat $O/calibration.txt
