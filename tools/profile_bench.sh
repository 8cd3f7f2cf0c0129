#!/bin/bash
# Collects the round's rocprofv3 evidence for bench.py on the GPU box (run through gpurun from the repo root):
#   kernel-trace stats, then separate --pmc passes (no other trace domains with --pmc).
# Usage: bash tools/profile_bench.sh <tag>      -> gpurun_out/prof_<tag>/...
set -u
TAG=${1:-run}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
ARGS="$R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-culled --no-fitting --no-training --no-c1 --no-f16"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ARGS > $OUT/stats.log 2>&1
for P in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA"; do
  N=$(echo $P | tr ' ' '_')
  rocprofv3 --pmc $P --output-format csv -d $OUT/pmc_$N -- python3 $ARGS > $OUT/pmc_$N.log 2>&1
done
cd $R
python3 tools/pmc_summary.py 'k_field2_hand<1>' $OUT/pmc_summary.json $OUT/pmc_*/ > $OUT/pmc_summary.log 2>&1
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
# the raw traces are large: keep the summaries only
find $OUT -name "*.csv" -size +2M -delete
tail -3 $OUT/stats.log
