set -e
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_r02b
mkdir -p $OUT
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench" -o bench -- python3 bench.py --no-cpu-baseline --no-strong-block > "$OUT/bench_line.json" 2> "$OUT/bench.err"
python3 tools/trace_medians.py "$OUT/bench" > "$OUT/bench_medians.txt"
python3 tools/stepbreakdown.py "$OUT/bench/bench_kernel_trace.csv" > "$OUT/sweep_step_breakdown.txt"
head -30 "$OUT/sweep_step_breakdown.txt"
