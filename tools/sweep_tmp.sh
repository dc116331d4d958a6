set -e
timeout -k 10 300 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread -k "gather or engine or config1" 2>&1 | tail -3
for t in 4096 8192 16384; do echo "== STAGE_TARGET=$t"; SPICE_GATHER_STAGE_TARGET_ELEMS=$t tools/rocprof_kbench.sh gather r02_gq_$t 30 | grep -v "^ *n "; done
for t in 2048 4096; do echo "== ROWS TARGET=$t"; SPICE_GATHER_TARGET_ELEMS=$t tools/rocprof_kbench.sh gather r02_gr_$t 30 | grep -v "^ *n "; done
