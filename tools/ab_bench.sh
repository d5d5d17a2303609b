#!/bin/bash
# Two (or more) builds of the library against each other on ONE box, alternating, the headline workload:
# boxes differ by 2-3 %, so a change of 1 % only shows when both builds run on the same one.
#   tools/ab_bench.sh path/to/libA.so path/to/libB.so [...]      (on the GPU box, from the repository root)
# A build of an older commit:  git archive <commit> slam_amd/csrc include | tar -x -C /tmp/old &&
#   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -shared -std=c++17 -o tools/diag/libold.so /tmp/old/slam_amd/csrc/mcmcgpu.hip
set -e
run() {
  timeout -k 10 200 python3 bench.py --no-live-traffic --no-cpu-baseline --no-plaza --steps 10 --warmup 3 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4g units/s, kernel %.2f ms' % (d['value'], d['roofline']['kernel_ms']))"
}
for round in 1 2 3; do
  for lib in "$@"; do
    echo -n "$lib: "
    MCMCGPU_LIB=$(realpath "$lib") run
  done
done
