#!/bin/bash
# The round's measurement pass on the GPU box: kernel traces and the two PMC passes, summaries into
# gpurun_out/prof_<tag>/ (copy what is to be kept into profiles/).  Usage: tools/profile_round.sh r02
set -e
TAG=${1:-rXX}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
# 1. every kernel alone, per-dispatch trace
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kbench" -o kb -- python3 tools/kbench.py --iters 25 > "$OUT/kbench.log" 2>&1
python3 tools/trace_medians.py "$OUT/kbench" > "$OUT/kbench_medians.txt"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kbench_fwd" -o kb -- python3 tools/kbench.py --only upsample,depthwise,bnact,conv1x1,stem --iters 25 > "$OUT/kbench_fwd.log" 2>&1
python3 tools/trace_medians.py "$OUT/kbench_fwd" > "$OUT/kbench_fwd_medians.txt"
# 2. the bench command itself (two candidates in flight: kernels of the two lanes overlap on the device) ...
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench" -o bench -- python3 bench.py --no-cpu-baseline > "$OUT/bench_line.json" 2> "$OUT/bench.err"
python3 tools/trace_medians.py "$OUT/bench" > "$OUT/bench_medians.txt"
# ... and with one candidate on the device at a time: every kernel's duration by itself, and the step breakdown
# (every launch queued by the host, so that the trace holds a candidate's launches in the order they were issued)
export SPICE_BENCH_MARK=1
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench_lanes1" -o bench -- python3 bench.py --no-cpu-baseline --lanes 1 --no-graphs > "$OUT/bench_lanes1_line.json" 2> "$OUT/bench_lanes1.err"
python3 tools/trace_medians.py "$OUT/bench_lanes1" > "$OUT/bench_lanes1_medians.txt"
python3 tools/stepbreakdown.py "$(find "$OUT/bench_lanes1" -name "*kernel_trace.csv" | head -1)" > "$OUT/sweep_step_breakdown.txt"
# ... and the same with the captured graphs (the graphs' kernels appear in the trace like any other launch)
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d "$OUT/bench_lanes1_graphs" -o bench -- python3 bench.py --no-cpu-baseline --no-kernel-table --no-strong-block --lanes 1 > "$OUT/bench_lanes1_graphs_line.json" 2> "$OUT/bench_lanes1_graphs.err"
python3 tools/stepbreakdown.py "$(find "$OUT/bench_lanes1_graphs" -name "*kernel_trace.csv" | head -1)" > "$OUT/sweep_step_breakdown_graphs.txt"
unset SPICE_BENCH_MARK
# 3. HBM traffic: one counter per pass, nothing else traced
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 tools/kbench.py --iters 6 > "$OUT/pmc_fetch.log" 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 tools/kbench.py --iters 6 > "$OUT/pmc_write.log" 2>&1
python3 tools/pmc_summary.py "$OUT/pmc_fetch" "$OUT/pmc_write" > "$OUT/pmc_traffic.json"
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch_small" -- python3 tools/kbench.py --only gather_cols,confusion_small --iters 6 > "$OUT/pmc_fetch_small.log" 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write_small" -- python3 tools/kbench.py --only gather_cols,confusion_small --iters 6 > "$OUT/pmc_write_small.log" 2>&1
python3 tools/pmc_summary.py "$OUT/pmc_fetch_small" "$OUT/pmc_write_small" > "$OUT/pmc_traffic_small_cases.json"
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch_fwd" -- python3 tools/kbench.py --only conv1x1_pmc,stem --iters 6 > "$OUT/pmc_fetch_fwd.log" 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write_fwd" -- python3 tools/kbench.py --only conv1x1_pmc,stem --iters 6 > "$OUT/pmc_write_fwd.log" 2>&1
python3 tools/pmc_summary.py "$OUT/pmc_fetch_fwd" "$OUT/pmc_write_fwd" > "$OUT/pmc_traffic_forward_kernels.json"
rm -rf "$OUT"/pmc_fetch "$OUT"/pmc_write "$OUT"/pmc_fetch_small "$OUT"/pmc_write_small "$OUT"/pmc_fetch_fwd "$OUT"/pmc_write_fwd
find "$OUT" -name "*kernel_trace.csv" -size +20M -delete

# 4. the convolution kernels against the libraries, shape by shape (HIP events, no profiler)
timeout -k 10 300 python3 tools/conv1x1_bench.py --plain > "$OUT/conv1x1_pairs_fused_vs_two_launches.jsonl" 2>/dev/null
timeout -k 10 300 python3 tools/conv3x3_bench.py > "$OUT/conv3x3_library_vs_miopen.jsonl" 2>/dev/null
timeout -k 10 100 python3 tools/conv7x7_bench.py >> "$OUT/conv3x3_library_vs_miopen.jsonl" 2>/dev/null
timeout -k 10 300 python3 bench.py > "$OUT/bench_n1.json" 2> "$OUT/bench_n1.err"
timeout -k 10 300 python3 bench.py --no-graphs --no-strong-block --no-cpu-baseline > "$OUT/bench_n1_no_graphs.json" 2> /dev/null
timeout -k 10 300 python3 tools/host_profile_step.py > "$OUT/host_profile_step.txt" 2> /dev/null
timeout -k 10 300 python3 tools/ftbench.py --steps 10 2> /dev/null | tail -1 > "$OUT/kd_finetune_step.json"
SPICE_DIST_BACKEND=gloo timeout -k 10 400 python3 bench.py --gpus 2 --steps 20 --warmup 4 --no-cpu-baseline > "$OUT/bench_2ranks_gloo_one_gpu.json" 2> /dev/null
echo convolution benches done
echo profile pass done
