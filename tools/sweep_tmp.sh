set -e
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 400 python bench.py --no-cpu-baseline --no-kernel-table > gpurun_out/bench_weak2.json 2> gpurun_out/bench_weak2.err
cat gpurun_out/bench_weak2.json
timeout -k 10 300 python bench.py --strong > gpurun_out/bench_strong_n1.json 2> gpurun_out/bench_strong_n1.err
cat gpurun_out/bench_strong_n1.json
SPICE_DIST_BACKEND=gloo timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 2 --strong > gpurun_out/bench_strong_gloo2.json 2> gpurun_out/bench_strong_gloo2.err
cat gpurun_out/bench_strong_gloo2.json
timeout -k 10 300 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread -k "sweep" 2>&1 | tail -3
