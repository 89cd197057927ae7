#!/bin/bash
# Copy what scripts/collect_round.sh + collect_configs.sh left under gpurun_out/prof_<tag>/ (and the test run's own JSON
# files under gpurun_out/) into profiles/<tag>/ -- the summaries only, not the rocprofv3 trace directories -- and put the
# stamped PMC figures where bench.py looks for them (profiles/k_ramp_traffic.json, valu_issue.json: quoted only while
# their csrc hash matches the tree's).
#   bash scripts/publish_round.sh <tag>
set -e
TAG=${1:?usage: publish_round.sh <tag>}
R=$(cd "$(dirname "$0")/.." && pwd)
SRC=$R/gpurun_out/prof_$TAG
DST=$R/profiles/$TAG
mkdir -p $DST
for f in bench.json bench_4ranks_one_gpu.json bench_under_rocprof.json bench_under_rocprof_thrower_electron.json \
         configs.txt ablate.txt timeline_gaps.txt pmc_sq_wide.txt ramp_vs_reads.txt thrower_vs_electrons.txt \
         host_path_cfg1.txt example_visit.txt fits_full_array.txt pmc_hbm.json pmc_sq.json k_ramp_traffic.json \
         valu_issue.json kernel_stats_cfg1.csv kernel_stats_cfg2.csv kernel_stats_cfg3.csv kernel_stats_cfg5.csv; do
  [ -f $SRC/$f ] && cp $SRC/$f $DST/$f || echo "missing: $f"
done
# the per-kernel statistics of the two traced bench runs (rocprofv3 writes <dir>/<host>/<pid>_kernel_stats.csv)
for pair in trace_split:kernel_stats.csv trace_electron:kernel_stats_thrower_electron.csv; do
  d=${pair%%:*}; o=${pair##*:}
  s=$(find $SRC/$d -name "*kernel_stats.csv" | head -1)
  [ -n "$s" ] && cp $s $DST/$o || echo "missing: $o"
done
cp $SRC/k_ramp_traffic.json $SRC/valu_issue.json $R/profiles/
for f in extremes.json fullsize_parity.json gpu_suite.log; do
  [ -f $R/gpurun_out/$f ] && cp $R/gpurun_out/$f $DST/$f
done
[ -f $R/gpurun_out/ensemble_parity.json ] && cp $R/gpurun_out/ensemble_parity.json $DST/ensemble_parity_default.json
python3 - <<EOF
import json, sys
sys.path.insert(0, "$R")
import bench
t = json.load(open("$R/profiles/k_ramp_traffic.json"))
print("csrc hash: tree", bench.csrc_hash(), " stamp", t.get("csrc_hash"), " commit", t.get("commit"))
EOF
