#!/bin/bash
# rocprofv3 evidence for the training iteration (tools/train_step_bench.py, hand nets): kernel-trace stats, then
# separate --pmc passes for the two GEMM kernels of the backward pass.
# Usage (through gpurun, from the repo root): bash tools/profile_train.sh <tag>   -> gpurun_out/prof_train_<tag>/
set -u
TAG=${1:-run}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_train_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for K in obj hand; do
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt_stats_$K -- python3 $R/tools/train_step_bench.py --steps 10 --warmup 3 --kinds $K > $OUT/stats_$K.log 2>&1
  find /tmp/pt_stats_$K -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/train_step_${K}_kernel_stats.csv
done
for P in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  N=$(echo $P | tr ' ' '_')
  rocprofv3 --pmc $P --output-format csv -d /tmp/pt_pmc_$N -- python3 $R/tools/train_step_bench.py --steps 3 --warmup 2 --kinds hand > $OUT/pmc_$N.log 2>&1
done
cd $R
python3 tools/pmc_summary.py 'k_outer' $OUT/pmc_train_hand_k_outer.json /tmp/pt_pmc_*/
python3 tools/pmc_summary.py 'k_dense' $OUT/pmc_train_hand_k_dense.json /tmp/pt_pmc_*/
tail -1 $OUT/stats_hand.log | cut -c1-200
