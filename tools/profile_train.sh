#!/bin/bash
# kernel stats of the training iteration (tools/train_step_bench.py), then separate --pmc passes (HBM bytes, MFMA-busy) for the two kernels
# of its backward pass.  Usage: bash tools/profile_train.sh <tag> <kind>   -> gpurun_out/prof_train_<tag>/
set -u
TAG=${1:-run}
KINDS=${2:-obj}        # obj | hand | hand_dense
FIELD=${KINDS%%_*}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_train_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
W=$(mktemp -d /tmp/hn_prof_XXXXXX)
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $W/stats -- python3 $R/tools/train_step_bench.py --kinds $KINDS --steps 20 --warmup 3 > $OUT/stats.log 2>&1
find $W/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
for P in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  N=$(echo $P | tr ' ' '_')
  rocprofv3 --pmc $P --output-format csv -d $W/pmc_$N -- python3 $R/tools/train_step_bench.py --kinds $KINDS --steps 4 --warmup 2 > $OUT/pmc_$N.log 2>&1
done
cd $R
for K in 'k_outer_group' "k_field2_$FIELD<5>" "k_field2_$FIELD<3>"; do
  T=$(echo $K | tr -d '<>' )
  python3 tools/pmc_summary.py "$K" $OUT/pmc_$T.json $W/pmc_*/ > /dev/null 2>&1
done
head -25 $OUT/kernel_stats.csv | cut -c1-200
