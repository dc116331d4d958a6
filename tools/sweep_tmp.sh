set -e
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread -k "bn_act or bottleneck or model_forward or engine or fused_modules or half_precision or torch_forward or sweep or config" 2>&1 | tail -5
timeout -k 10 400 python bench.py --no-cpu-baseline --no-kernel-table > gpurun_out/bench_dual.json 2> gpurun_out/bench_dual.err
head -c 400 gpurun_out/bench_dual.json
