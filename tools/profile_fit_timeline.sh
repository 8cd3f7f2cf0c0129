#!/bin/bash
# kernel-trace of the fitting step + per-step timeline and busy/idle summary.
# Usage (through gpurun, from the repo root): bash tools/profile_fit_timeline.sh <tag>  -> gpurun_out/prof_fit_<tag>/
set -u
TAG=${1:-run}
FRAMES=${2:-1}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_fit_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
W=$(mktemp -d /tmp/hn_prof_XXXXXX)   # a directory of this invocation's own: a box may be re-used by later calls, and a tag twice
cd /tmp
python3 $R/tools/fit_profile.py 40 halo pipe $FRAMES > $OUT/unprofiled.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $W/stats -- python3 $R/tools/fit_profile.py 20 halo pipe $FRAMES > $OUT/stats.log 2>&1
find $W/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
T=$(find $W/stats -name "*kernel_trace.csv" | head -1)
python3 $R/tools/trace_gaps.py $T 12 4 > $OUT/busy_idle.txt 2>&1
python3 $R/tools/trace_timeline.py $T 8 15 > $OUT/timeline.txt 2>&1
python3 $R/tools/trace_timeline.py $T 8 0 > $OUT/timeline_all.txt 2>&1
cd $R
tail -1 $OUT/unprofiled.log
head -8 $OUT/busy_idle.txt
