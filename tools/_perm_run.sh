#!/bin/bash
# one pass over the variants: ms per launch at two batch sizes
one() { TRM_LIB=$PWD/gnuspeech_amd/libtrm_var_$1.so python bench.py --steps 20 --warmup 3 --no-cpu-baseline --kernel wide --voices $2 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.3f"%d["roofline"]["avg_launch_ms"])'; }
echo "base - $(one s1 65536) $(one s1 12288)"
while read i p; do echo "p$i $p $(one p$i 65536) $(one p$i 12288)"; done < tools/_perm_idx.txt
echo "base - $(one s1 65536) $(one s1 12288)"
