set -e
timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread -k "kdlite or loss or config3 or training_step or prune_model" 2>&1 | tail -3
tools/rocprof_kbench.sh kdlite_grad,kdlite5 r02_kdpre 20 | grep -v "^ *n "
