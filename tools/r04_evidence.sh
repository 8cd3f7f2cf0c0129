#!/bin/bash
# The round's evidence in one gpurun call: GPU suite + parity report, bench.py profiles (stats + PMC), fitting step profiles, the bench line.
# Usage: gpurun --timeout 3000 -- 'bash tools/r04_evidence.sh'   -> gpurun_out/r04/, gpurun_out/prof_*r04/
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
(timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r04/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/r04/pytest_gpu.log)
tail -3 gpurun_out/r04/pytest_gpu.log
cp gpurun_out/parity_report.json gpurun_out/r04/parity_report.json 2>/dev/null
timeout 900 bash tools/profile_bench.sh r04 > gpurun_out/r04/profile_bench.log 2>&1
timeout 600 bash tools/profile_fit.sh r04 > gpurun_out/r04/profile_fit.log 2>&1
timeout 400 bash tools/profile_fit_video_timeline.sh r04 > gpurun_out/r04/profile_fitv.log 2>&1
cd $GRAFT_REPO_ROOT
mkdir -p profiles/r04 && cp gpurun_out/prof_r04/pmc_summary.json profiles/r04/pmc_bench_field2_hand_full_r04.json && cp profiles/r04/pmc_bench_field2_hand_full_r04.json gpurun_out/r04/
(timeout 1200 python bench.py > gpurun_out/r04/bench_line.json 2> gpurun_out/r04/bench_err.log; echo "bench exit $?" >> gpurun_out/r04/bench_err.log)
# the N > 1 code path, functionally: two gloo ranks on this one GPU (timings mean nothing)
(HONERF_BENCH_SHARE_GPU=1 timeout 900 python bench.py --gpus 2 --steps 2 --warmup 1 --fit-quick --no-cpu-baseline --no-culled --no-f16 --no-c1 --no-training > gpurun_out/r04/bench_2ranks_one_gpu_gloo_functional.json 2> gpurun_out/r04/bench_2ranks_err.log; echo "2-rank exit $?" >> gpurun_out/r04/bench_2ranks_err.log)
tail -2 gpurun_out/r04/bench_2ranks_err.log
tail -c 1500 gpurun_out/r04/bench_line.json
tail -2 gpurun_out/r04/bench_err.log
