#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   bash scripts/collect_profiles.sh <tag> <commit>     -> gpurun_out/prof_<tag>/...
# (<commit> = `git rev-parse HEAD` of the tree that was snapshotted: the GPU box has no .git)
# kernel trace + stats, HBM traffic counters and SQ counters in SEPARATE passes (never --pmc with a trace).
set -e
TAG=${1:-final}
COMMIT=${2:-unknown}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-extra-pass"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_split -o t -- $B --steps 30 --warmup 3 > $OUT/bench_under_rocprof.json 2> $OUT/trace_split.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_electron -o t -- $B --steps 30 --warmup 3 --thrower electron > $OUT/bench_under_rocprof_thrower_electron.json 2> $OUT/trace_electron.log
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -o p -- $B --steps 4 --warmup 1 > /dev/null 2> $OUT/pmc_$c.log
done
for c in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS_F32 SQ_WAVES SQ_BUSY_CU_CYCLES" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_WAVE_CYCLES"; do
  n=$(echo $c | tr " " "_")
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $OUT/sq_$n -o p -- $B --steps 4 --warmup 1 > /dev/null 2> $OUT/sq_$n.log || echo "pmc pass $n failed (see $OUT/sq_$n.log)"
done
python3 $R/scripts/summarize_pmc.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE > $OUT/pmc_hbm.json
python3 $R/scripts/summarize_pmc.py $OUT/sq_* > $OUT/pmc_sq.json
# the two files bench.py quotes, stamped with the commit and the hash of wayne_amd/csrc (copy them to profiles/)
python3 $R/scripts/make_profile_stamps.py $OUT $COMMIT
find $OUT -name "*_kernel_stats.csv" | head
echo collected $OUT
