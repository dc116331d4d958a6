#!/bin/bash
# Print VGPR / SGPR / occupancy / LDS per kernel of one .hip file.
f=$1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Rpass-analysis=kernel-resource-usage -c $f -o /tmp/ru.o 2>&1 | \
 awk '/Function Name/ {n=$NF} / VGPRs:/ {v=$(NF-1)} /AGPRs/ {a=$(NF-1)} /Occupancy/ {o=$(NF-1)} /ScratchSize/ {s=$(NF-1)} /LDS Size/ {l=$(NF-1); printf "%-80s vgpr=%s occ=%s scratch=%s lds=%s\n", n, v, o, s, l}'
