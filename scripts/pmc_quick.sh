#!/bin/bash
# Quick SQ counter passes over `bench.py --no-extra-pass` (no fork, so kernels do not overlap):
#   bash scripts/pmc_quick.sh <tag> [extra bench args]  -> gpurun_out/pmcq_<tag>.json
TAG=${1:-q}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmcq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export WAYNE_FORK_NARROW=0
B="python3 $R/bench.py --no-cpu-baseline --no-extra-pass --steps 4 --warmup 1 $@"
for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS_F32" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_FLAT SQ_ACTIVE_INST_SCA"; do
  n=$(echo $c | tr " " "_")
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $OUT/$n -o p -- $B > /dev/null 2> $OUT/$n.log || echo "pass $n failed"
done
python3 $R/scripts/summarize_pmc.py $OUT/SQ_* > $R/gpurun_out/pmcq_$TAG.json
python3 - <<PY
import json
d=json.load(open("$R/gpurun_out/pmcq_$TAG.json"))
for k,v in d.items():
    if "wayne::" not in k: continue
    print(k, {c: round(x["mean"]) for c,x in v.items()})
PY
