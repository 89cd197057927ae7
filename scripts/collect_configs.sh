#!/bin/bash
# rocprofv3 kernel statistics of the OTHER BASELINE configurations (cfg4 is scripts/collect_profiles.sh's):
#   bash scripts/collect_configs.sh <tag>     -> gpurun_out/prof_<tag>/kernel_stats_<cfg>.csv
TAG=${1:-final}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in cfg1 cfg2 cfg3 cfg5; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$c -o t -- python3 $R/bench.py --config $c --steps 30 --warmup 3 --no-cpu-baseline --no-extra-pass > $OUT/bench_under_rocprof_$c.json 2> $OUT/trace_$c.log || echo "$c failed"
  cp $OUT/trace_$c/t_kernel_stats.csv $OUT/kernel_stats_$c.csv 2>/dev/null
done
ls $OUT | grep kernel_stats
