#!/bin/bash
# gpurun_out/ (scratch, merged back from the GPU box by tools/r05_evidence.sh) -> profiles/r05/ (tracked).  Usage: bash tools/collect_r05.sh
set -eu
cd "$(dirname "$0")/.."
O=gpurun_out/r05
P=profiles/r05
mkdir -p $P
cp $O/csrc_sha16.txt $O/parity_report.json $O/frames_side_by_side.txt $O/host_time_probe.txt $O/scan_kernels_hbm.json $O/secondary_bench.json \
   $O/seq_repro_jacobian_first.txt $O/seq_repro_shipped.txt $O/bench_2ranks_one_gpu_gloo_functional.json $O/pmc_bench_field2_hand_full_r05.json \
   $O/train_step_obj_kernel_stats.csv $O/train_step_hand_kernel_stats.csv $O/train_fused_ab_obj.txt $O/train_fused_ab_hand.txt \
   $O/train_grad_ab_obj.txt $O/train_grad_ab_hand.txt $O/outer_group_parts.txt $O/train_soak_hand.txt $O/train_soak_obj.txt $P/
cp $O/pmc_train_*.json $P/
grep '^{' $O/bench_line.json | tail -1 > $P/bench_r05_line.json
cp gpurun_out/prof_r05/kernel_stats.csv $P/bench_r05_kernel_stats.csv
for f in busy_idle.txt kernel_stats.csv timeline.txt timeline_all.txt; do
    cp gpurun_out/prof_fit_r05/$f $P/fit_step_$f
    cp gpurun_out/prof_fit_r05_f8/$f $P/fit_step_8_frames_$f
done
for f in busy_idle.txt kernel_stats.csv timeline.txt; do cp gpurun_out/prof_fitv_r05/$f $P/fit_video_step_$f; done
for k in hand3 hand4 obj3 obj4; do
    cp gpurun_out/prof_fit_r05/pmc_k_field2_$k.json $P/pmc_fit_k_field2_$k.json
    cp gpurun_out/prof_fit_r05_f8/pmc_k_field2_$k.json $P/pmc_fit_8_frames_k_field2_$k.json
done
tail -3 $O/pytest_gpu.log | head -2
cat $P/csrc_sha16.txt
