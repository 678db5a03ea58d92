#!/bin/bash
# Everything profiles/<round>/ holds, in one GPU call: bash tools/collect_profiles.sh <tag>  -> gpurun_out/<tag>/
# (needs build/variants/prof.so for the tail profiles: tools/build_variant.sh prof -DLOBSIM_PROFILE -DLOBSIM_STEP_WAVES=4 in the build container;
#  roofline.traffic of the bench lines comes from profiles/<round>/pmc_step_kernel*.csv: copy the csvs there, then bash tools/rebench_lines.sh)
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$1; mkdir -p $OUT
cd $R
# ---- bench lines: configs[1] in the default and in the driver's form, the other configs' shards, the kernels other configurations run
timeout 400 python3 bench.py > $OUT/bench_random.json 2> $OUT/bench_random.err
timeout 400 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_random_20steps.json 2>/dev/null
timeout 400 python3 bench.py --agent-orders 128 --no-cpu-baseline --no-facade > $OUT/bench_random_agent128.json 2>/dev/null
for w in teradactyl ad deep50 rs norm; do timeout 400 python3 bench.py --workload $w --no-cpu-baseline --steps 200 > $OUT/bench_$w.json 2> $OUT/bench_$w.err; done
timeout 400 python3 bench.py --workload deep50 --agent-orders 128 --no-cpu-baseline --no-facade --steps 200 > $OUT/bench_deep50_agent128.json 2>/dev/null
timeout 400 python3 bench.py --generic-kernels --no-cpu-baseline --no-facade --steps 200 > $OUT/bench_generic.json 2>/dev/null
for n in 8192 16384; do timeout 400 python3 bench.py --envs $n --no-cpu-baseline --no-facade --steps 200 > $OUT/bench_random_${n}envs.json 2>/dev/null; done
# ---- rocprofv3 kernel trace (per-kernel time)
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 100 --warmup 10 --repeats 2 --no-cpu-baseline --no-facade > $OUT/stats.log 2>&1
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
cd $R
# ---- PMC counters of the step kernel, per workload (one rocprofv3 run per counter group)
PMC_ARGS="" bash tools/pmc_passes.sh $1/pmcdir_random > /dev/null 2>&1; python3 tools/pmc_csv.py $OUT/pmcdir_random $OUT/pmc_step_kernel.csv > /dev/null
for w in teradactyl ad deep50 rs norm; do
  PMC_ONLY="sq1 sq2 fetch write" PMC_ARGS="--workload $w" bash tools/pmc_passes.sh $1/pmcdir_$w > /dev/null 2>&1
  python3 tools/pmc_csv.py $OUT/pmcdir_$w $OUT/pmc_step_kernel_$w.csv > /dev/null
done
PMC_ONLY="sq1 sq2 fetch write" PMC_ARGS="--generic-kernels" bash tools/pmc_passes.sh $1/pmcdir_generic > /dev/null 2>&1
python3 tools/pmc_csv.py $OUT/pmcdir_generic $OUT/pmc_step_kernel_generic.csv > /dev/null
rm -rf $OUT/pmcdir_* $OUT/stats $OUT/*.log
# ---- what the slowest wave of a launch does (stamp build), per workload
if [ -f build/variants/prof.so ]; then bash tools/tail_profiles.sh $1 > /dev/null 2>&1; fi
# ---- the bench lines again, now that the PMC files of this collection exist (roofline.traffic reads them from profiles/<round>/): done by
# copying gpurun_out/<tag>/pmc_step_kernel*.csv into profiles/<round>/ and re-running the lines that are committed
for f in $OUT/bench_*.json; do echo $f; cut -c1-260 $f; done; cat $OUT/kernel_stats.csv | head -4; grep -E "FETCH|WRITE|WAVES,|WAVE_CYCLES|WAIT_ANY|INSTS_VALU|INSTS_SALU" $OUT/pmc_step_kernel.csv
