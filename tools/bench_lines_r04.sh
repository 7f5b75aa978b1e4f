#!/bin/bash
# The round's bench lines, one per BASELINE config, with the traffic file of the same kernel sources in place
# (tools/profile_round4.sh -> tools/collect_round4.sh first) -> profiles/bench_r04_config{1,2,3,4}.json (+ the saturating batch
# and the whole-utterance lines for comparison).
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/bench_r04
python bench.py > gpurun_out/bench_r04/config1.json 2> gpurun_out/bench_r04/config1.err
for c in 2 3 4; do python bench.py --config $c > gpurun_out/bench_r04/config$c.json 2> gpurun_out/bench_r04/config$c.err; done
python bench.py --split off --no-stream --no-cpu-baseline > gpurun_out/bench_r04/config1_whole.json 2>/dev/null
python bench.py --config 4 --split off --no-cpu-baseline > gpurun_out/bench_r04/config4_whole.json 2>/dev/null
python bench.py --voices 65536 --kernel wide --no-stream --no-cpu-baseline > gpurun_out/bench_r04/wide65536.json 2>/dev/null
for f in gpurun_out/bench_r04/*.json; do python -c "import sys,json; d=json.loads(open(sys.argv[1]).read()); r=d['roofline']; print('%-22s %.3f ms  %.3e samples/s  frac %.4f  traffic %s  valu %s' % (sys.argv[1].split('/')[-1], d['ms_per_step'], d['value'], r['frac'], r['traffic'], (r['valu_issue'] or {}).get('frac_of_issue_slots')))" $f; done
