# batch-size sweep over both kernel forms (ms per launch, output samples/s); usage: sweep_forms.sh [sizes...]
SIZES=${@:-1024 2048 4096 6144 8192 12288 16384 32768}
for v in $SIZES; do for k in quad wide; do python bench.py --steps 3 --warmup 1 --no-cpu-baseline --kernel $k --voices $v 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%6d %s %.3f ms  %.3e samples/s"%(d["config"]["voices_per_gpu"], d["config"]["kernel_form"], d["ms_per_step"], d["value"]))'; done; done
