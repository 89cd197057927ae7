#!/bin/bash
# One gpurun call that collects a round's evidence (run from the repo root on the GPU box):
#   bash scripts/collect_round.sh <tag> <commit>    -> gpurun_out/prof_<tag>/...
# = scripts/collect_profiles.sh (kernel trace + stats, HBM and SQ counters, stamped) + the bench line with the CPU
# baseline + every configuration + the stage ablation + the stream's idle gaps + the wide SQ counter set.
TAG=${1:-final}
COMMIT=${2:-unknown}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
bash $R/scripts/collect_profiles.sh $TAG $COMMIT > $OUT/collect.log 2>&1 || echo "collect_profiles failed" >> $OUT/collect.log
cd $R
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
python3 scripts/bench_configs.py 40 > $OUT/configs.txt 2> $OUT/configs.err
python3 scripts/ablate.py cfg4 5 > $OUT/ablate.txt 2> $OUT/ablate.err
T=$(find $OUT/trace_split -name "*kernel_trace.csv" | head -1)
[ -n "$T" ] && python3 scripts/timeline_gaps.py $T > $OUT/timeline_gaps.txt 2>&1
bash scripts/pmc_quick.sh $TAG > $OUT/pmc_sq_wide.txt 2>&1
ls $OUT
python3 scripts/ramp_vs_reads.py 30 > $OUT/ramp_vs_reads.txt 2>&1
python3 scripts/thrower_vs_electrons.py 20 > $OUT/thrower_vs_electrons.txt 2>&1
python3 scripts/profile_host_path.py cfg1 1000 > $OUT/host_path_cfg1.txt 2>&1
python3 scripts/run_example_visit.py 3000 > $OUT/example_visit.txt 2>&1
python3 scripts/time_fits_pipeline.py > $OUT/fits_full_array.txt 2>&1
# the 4-ranks-on-one-card rehearsal of `bench.py --gpus N` (delivered / end_to_end per rank; not a scaling figure)
WAYNE_BENCH_SHARE_GPU=1 python3 bench.py --gpus 4 --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_4ranks_one_gpu.json 2> $OUT/bench_4ranks_one_gpu.err
ls $OUT
