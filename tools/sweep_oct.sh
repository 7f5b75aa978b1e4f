#!/bin/bash
# small batches: TRM_KERNEL_QUAD (16 voices per workgroup, four lanes per voice) against TRM_KERNEL_OCT (8 voices per
# workgroup, eight lanes per voice); ms per launch.  usage: sweep_oct.sh [sizes...]
SIZES=${@:-16 128 512 1024 2048 3072 4096}
for wl in static timevarying; do for v in $SIZES; do for o in quad oct; do
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline --kernel $o --voices $v --workload $wl 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%-12s %6d voices  %s  %.3f ms  %.3e samples/s" % ("'$wl'", d["config"]["voices_per_gpu"], d["config"]["kernel_form"], d["ms_per_step"], d["value"]))'
done; done; done
