#!/bin/bash
# Copy the summaries of tools/profile_round.sh (gpurun_out/prof_<tag>/) into profiles/<tag>/ under the names its README uses.
set -e
TAG=${1:-r02}
S=gpurun_out/prof_$TAG
D=profiles/$TAG
mkdir -p "$D"
cp "$S/bench_line.json" "$D/bench_n1_under_rocprof.json"
cp "$S/bench/bench_kernel_stats.csv" "$D/bench_kernel_stats.csv"
cp "$S/bench_medians.txt" "$D/bench_kernel_medians.txt"
cp "$S/sweep_step_breakdown.txt" "$S/sweep_step_breakdown_graphs.txt" "$D/"
cp "$S/bench_lanes1_line.json" "$D/bench_n1_lanes1_under_rocprof.json"
cp "$S/bench_lanes1/bench_kernel_stats.csv" "$D/bench_lanes1_kernel_stats.csv"
cp "$S/bench_lanes1_medians.txt" "$D/bench_lanes1_kernel_medians.txt"
cp "$S/kbench/kb_kernel_stats.csv" "$D/kbench_kernel_stats.csv"
cp "$S/kbench_medians.txt" "$D/kbench_kernel_medians.txt"
cp "$S/kbench_fwd/kb_kernel_stats.csv" "$D/kbench_forward_kernel_stats.csv"
cp "$S/kbench_fwd_medians.txt" "$D/kbench_forward_kernel_medians.txt"
grep "^{" "$S/kbench_fwd.log" > "$D/kbench_forward_kernels_under_rocprof.jsonl" || true
cp "$S/pmc_traffic.json" "$S/pmc_traffic_small_cases.json" "$S/pmc_traffic_forward_kernels.json" "$D/"
for f in conv1x1_pairs_fused_vs_two_launches.jsonl conv3x3_library_vs_miopen.jsonl bench_n1.json bench_n1_no_graphs.json host_profile_step.txt kd_finetune_step.json bench_2ranks_gloo_one_gpu.json; do
    [ -s "$S/$f" ] && cp "$S/$f" "$D/$f"
done
echo copied
