#!/bin/bash
# Everything profiles/<round>/ holds, in one GPU call: bash tools/collect_profiles.sh <tag>  -> gpurun_out/<tag>/
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$1; mkdir -p $OUT
cd $R
python3 bench.py > $OUT/bench_random.json 2> $OUT/bench_random.err
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_random_20steps.json 2>/dev/null
for w in teradactyl ad deep50; do python3 bench.py --workload $w --no-cpu-baseline --steps 200 > $OUT/bench_$w.json 2> $OUT/bench_$w.err; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 100 --warmup 10 --repeats 2 --no-cpu-baseline --no-facade > $OUT/stats.log 2>&1
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
cd $R
bash tools/pmc_passes.sh $1/pmc > /dev/null 2>&1
python3 tools/pmc_csv.py $OUT/pmc $OUT/pmc_step_kernel.csv > /dev/null
rm -rf $OUT/pmc $OUT/stats $OUT/pmc.*.log
for f in $OUT/bench_*.json; do echo $f; cut -c1-260 $f; done; cat $OUT/kernel_stats.csv | head -4; cat $OUT/pmc_step_kernel.csv | grep -E "FETCH|WRITE|WAVES,|WAVE_CYCLES|WAIT_ANY|INSTS_VALU|INSTS_SALU"
