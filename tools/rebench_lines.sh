#!/bin/bash
# The committed bench lines again (after the PMC csvs of a collection sit in profiles/<round>/): bash tools/rebench_lines.sh <tag> -> gpurun_out/<tag>/
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:?usage: rebench_lines.sh <tag>}; mkdir -p $OUT; cd $R
timeout 400 python3 bench.py > $OUT/bench_random.json 2> /dev/null
timeout 400 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_random_20steps.json 2>/dev/null
timeout 400 python3 bench.py --agent-orders 128 --no-cpu-baseline --no-facade > $OUT/bench_random_agent128.json 2>/dev/null
for w in teradactyl ad deep50 rs norm; do timeout 400 python3 bench.py --workload $w --no-cpu-baseline --steps 200 > $OUT/bench_$w.json 2>/dev/null; done
timeout 400 python3 bench.py --workload deep50 --agent-orders 128 --no-cpu-baseline --no-facade --steps 200 > $OUT/bench_deep50_agent128.json 2>/dev/null
timeout 400 python3 bench.py --generic-kernels --no-cpu-baseline --no-facade --steps 200 > $OUT/bench_generic.json 2>/dev/null
for n in 8192 16384; do timeout 400 python3 bench.py --envs $n --no-cpu-baseline --no-facade --steps 200 > $OUT/bench_random_${n}envs.json 2>/dev/null; done
for f in $OUT/bench_*.json; do python3 -c "
import json,sys
d=json.load(open('$f')); r=d['roofline']
print('$f'.split('/')[-1], 'value %.1f M'%(d['value']/1e6), 'kernel_ms %.5f'%r['kernel_ms'], 'frac %.4f'%r['frac'], 'traffic', r['traffic'], r['traffic_source'])
"; done
