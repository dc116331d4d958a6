set -e
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 400 python bench.py > gpurun_out/bench_weak.json 2> gpurun_out/bench_weak.err
tail -c 3000 gpurun_out/bench_weak.json
timeout -k 10 300 python bench.py --strong > gpurun_out/bench_strong_n1.json 2> gpurun_out/bench_strong_n1.err
cat gpurun_out/bench_strong_n1.json
SPICE_DIST_BACKEND=gloo timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 2 --strong > gpurun_out/bench_strong_gloo2.json 2> gpurun_out/bench_strong_gloo2.err
cat gpurun_out/bench_strong_gloo2.json
