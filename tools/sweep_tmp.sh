set -e
timeout -k 10 800 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread 2>&1 | tail -5
timeout -k 10 400 python bench.py --no-cpu-baseline --no-kernel-table > gpurun_out/bench_pair.json 2> gpurun_out/bench_pair.err
head -c 400 gpurun_out/bench_pair.json
