#!/bin/bash
# rocprofv3 evidence for the fitting step (tools/fit_profile.py: fitting_single, fit type 12, 196 rays x 192 depths):
# kernel-trace stats + busy/idle analysis + per-step timelines, then separate --pmc passes for the four persistent field kernels of the step.
# Usage (through gpurun, from the repo root): bash tools/profile_fit.sh <tag>   -> gpurun_out/prof_fit_<tag>/
set -u
TAG=${1:-run}
FRAMES=${2:-1}      # frames side by side (fitting.fit_frames_batched's step) 
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
for P in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  N=$(echo $P | tr ' ' '_')
  rocprofv3 --pmc $P --output-format csv -d $W/pmc_$N -- python3 $R/tools/fit_profile.py 6 halo pipe $FRAMES > $OUT/pmc_$N.log 2>&1
done
cd $R
for K in 'k_field2_hand<4>' 'k_field2_hand<3>' 'k_field2_obj<4>' 'k_field2_obj<3>'; do
  T=$(echo $K | tr -d '<>' )
  python3 tools/pmc_summary.py "$K" $OUT/pmc_$T.json $W/pmc_*/ > /dev/null 2>&1
done
cat $OUT/unprofiled.log | tail -1
cat $OUT/busy_idle.txt | head -12
