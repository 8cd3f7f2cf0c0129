#!/bin/bash
# kernel stats of the training iteration (tools/train_step_bench.py).  Usage: bash tools/profile_train.sh <tag> <kinds>
set -u
TAG=${1:-run}
KINDS=${2:-obj}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_train_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
W=$(mktemp -d /tmp/hn_prof_XXXXXX)
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $W/stats -- python3 $R/tools/train_step_bench.py --kinds $KINDS --steps 20 --warmup 3 > $OUT/stats.log 2>&1
find $W/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
cd $R
head -25 $OUT/kernel_stats.csv | cut -c1-200
