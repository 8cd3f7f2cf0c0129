#!/bin/bash
# kernel-trace of the fitting_video window step + timeline.  Usage: bash tools/profile_fit_video_timeline.sh <tag>
set -u
TAG=${1:-run}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_fitv_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
W=$(mktemp -d /tmp/hn_prof_XXXXXX)   # a directory of this invocation's own: a box may be re-used by later calls, and a tag twice
cd /tmp
python3 $R/tools/fit_profile_video.py 40 > $OUT/unprofiled.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $W/stats -- python3 $R/tools/fit_profile_video.py 20 > $OUT/stats.log 2>&1
find $W/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
export HN_TRACE_MARK=k_sort_rows   # once per step (the stable term's taped evaluation is a second k_field2_hand<3> per step)
T=$(find $W/stats -name "*kernel_trace.csv" | head -1)
python3 $R/tools/trace_gaps.py $T 12 4 > $OUT/busy_idle.txt 2>&1
python3 $R/tools/trace_timeline.py $T 8 15 > $OUT/timeline.txt 2>&1
python3 $R/tools/trace_timeline.py $T 8 0 > $OUT/timeline_all.txt 2>&1
cd $R
tail -1 $OUT/unprofiled.log
head -16 $OUT/busy_idle.txt
