#!/bin/bash
# The round's evidence in one gpurun call: GPU suite + parity report, bench.py profiles (stats + PMC), fitting step profiles (one frame and
# 8 frames side by side), the sequence-loop reproducibility diagnosis, scan kernels, the secondary row, the bench line.
# Usage: gpurun --timeout 3300 -- 'bash tools/r05_evidence.sh'   -> gpurun_out/r05/, gpurun_out/prof_*r05*/
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05
mkdir -p $O
python tools/srchash.py > $O/csrc_sha16.txt
(timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log)
tail -3 $O/pytest_gpu.log
cp gpurun_out/parity_report.json $O/parity_report.json 2>/dev/null
timeout 900 bash tools/profile_bench.sh r05 > $O/profile_bench.log 2>&1
timeout 600 bash tools/profile_fit.sh r05 1 > $O/profile_fit.log 2>&1
timeout 900 bash tools/profile_fit.sh r05_f8 8 > $O/profile_fit_f8.log 2>&1
timeout 400 bash tools/profile_fit_video_timeline.sh r05 > $O/profile_fitv.log 2>&1
cd $GRAFT_REPO_ROOT
cp gpurun_out/prof_r05/pmc_summary.json $O/pmc_bench_field2_hand_full_r05.json
mkdir -p profiles/r05 && cp $O/pmc_bench_field2_hand_full_r05.json profiles/r05/      # (bench.py quotes roofline.traffic from it when the source hash matches)
# the training iteration: kernel stats per field kind, the fused parameter-gradient path beside the generic sequence
timeout 400 bash tools/profile_train.sh r05_obj obj > $O/profile_train_obj.log 2>&1
timeout 400 bash tools/profile_train.sh r05_hand hand > $O/profile_train_hand.log 2>&1
timeout 400 bash tools/profile_train.sh r05_hand_dense hand_dense > $O/profile_train_hand_dense.log 2>&1
# (PMC summaries of the backward pass's two kernels: train_step_bench.py / bench.py quote roofline_hbm.traffic from them when the source hash matches)
for k in k_outer_group k_field2_obj5 k_field2_obj3; do cp gpurun_out/prof_train_r05_obj/pmc_$k.json $O/pmc_train_obj_$k.json; cp $O/pmc_train_obj_$k.json profiles/r05/; done
for k in k_outer_group k_field2_hand5 k_field2_hand3; do cp gpurun_out/prof_train_r05_hand_dense/pmc_$k.json $O/pmc_train_hand_dense_$k.json; cp $O/pmc_train_hand_dense_$k.json profiles/r05/; done
cp gpurun_out/prof_train_r05_obj/kernel_stats.csv $O/train_step_obj_kernel_stats.csv
cp gpurun_out/prof_train_r05_hand/kernel_stats.csv $O/train_step_hand_kernel_stats.csv
(timeout 300 python tools/train_fused_ab.py 56448 obj 2>&1 | grep -v amdgpu > $O/train_fused_ab_obj.txt)
(timeout 300 python tools/train_fused_ab.py 56448 hand 2>&1 | grep -v amdgpu > $O/train_fused_ab_hand.txt)
(timeout 300 python tools/train_grad_ab.py obj 0 2>&1 | grep -v "amdgpu\|UserWarning\|Consider\|np.savez" > $O/train_grad_ab_obj.txt)
(timeout 300 python tools/train_grad_ab.py hand 1 2>&1 | grep -v "amdgpu\|UserWarning\|Consider\|np.savez" > $O/train_grad_ab_hand.txt)
(for d in 0 2 4 6; do echo "HN_DBG_OUTER=$d (2: no MFMAs, 4: no splits / LDS stores after the first step; results wrong, timing only)"; HN_DBG_OUTER=$d timeout 200 python tools/train_fused_ab.py 56448 obj 2>&1 | grep "^n ="; done > $O/outer_group_parts.txt)
(timeout 400 python tools/train_soak.py hand 300 2>&1 | grep -v "amdgpu\|Warning\|Consider\|float(" > $O/train_soak_hand.txt)
(timeout 400 python tools/train_soak.py obj 300 2>&1 | grep -v "amdgpu\|Warning\|Consider\|float(" > $O/train_soak_obj.txt)
timeout 300 python tools/scan_bench.py $O/scan_kernels_hbm.json > $O/scan_kernels.log 2>&1
timeout 600 python tools/secondary_bench.py $O/secondary_bench.json > $O/secondary_bench.log 2>&1
# the sequence loop's reproducibility: as shipped (Jacobian launch behind the stable term), and with the round-4 order
timeout 300 python tools/seq_repro_diag.py --frames 6 --runs 4 --check > $O/seq_repro_shipped.txt 2>&1
timeout 300 python tools/seq_repro_diag.py --frames 6 --runs 4 --check --no-defer > $O/seq_repro_jacobian_first.txt 2>&1
timeout 300 python tools/host_time_probe.py > $O/host_time_probe.txt 2>&1
for f in 1 2 4 8; do timeout 200 python tools/fit_profile.py 60 halo pipe $f 2>&1 | tail -1 >> $O/frames_side_by_side.txt; done
(HN_TAPED_GRID=cus timeout 200 python tools/fit_profile.py 60 halo pipe 4 2>&1 | tail -1 | sed 's/^/persistent grids (HN_TAPED_GRID=cus): /' >> $O/frames_side_by_side.txt)
(timeout 1500 python bench.py > $O/bench_line.json 2> $O/bench_err.log; echo "bench exit $?" >> $O/bench_err.log)
# the N > 1 code path, functionally: two gloo ranks on this one GPU (timings mean nothing)
(HONERF_BENCH_SHARE_GPU=1 timeout 900 python bench.py --gpus 2 --steps 2 --warmup 1 --fit-quick --no-cpu-baseline --no-culled --no-f16 --no-c1 --no-training > $O/bench_2ranks_one_gpu_gloo_functional.json 2> $O/bench_2ranks_err.log; echo "2-rank exit $?" >> $O/bench_2ranks_err.log)
tail -2 $O/bench_2ranks_err.log
tail -c 1200 $O/bench_line.json
tail -2 $O/bench_err.log
