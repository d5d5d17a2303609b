#!/bin/bash
# Runs on the GPU box (from the repository root): phase shares of the diagnostic build (benchmark workload and
# both Plaza replays, with the next update selected ahead and without), the rebuild scan's rates, two soak runs
# and the host program's Plaza replays.  Everything lands in gpurun_out/<tag>/.
#   tools/profile_extras.sh r3
set -e
TAG=${1:-run}
R=$(pwd)
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
timeout -k 10 300 python3 tools/phase_profile.py > "$OUT/${TAG}_phases.txt" 2>&1
for p in Plaza1 Plaza2; do
  n=$(echo $p | tr 'P' 'p')
  { echo "== serial_program 0 (the library's choice here: compiled in, next update selected ahead)";
    timeout -k 10 120 python3 tools/phase_profile.py --plaza $p;
    echo "== serial_program 2 (compiled in, nothing selected ahead): the parts of selection and proposal";
    timeout -k 10 120 python3 tools/phase_profile.py --plaza $p --serial-program 2; } > "$OUT/${TAG}_phases_$n.txt" 2>&1
done
timeout -k 10 300 python3 tools/scan_bench.py > "$OUT/${TAG}_scan.txt" 2>&1
{ echo "== benchmark problem, 30 launches of 1000 updates x 1024 chains, rebuild every 5";
  timeout -k 10 400 python3 tools/soak.py 30 1000 5 1024;
  echo "== commit-heavy problem (one wavefront per chain, serial program compiled in), 60 launches";
  timeout -k 10 400 python3 tools/soak.py 60 1000 5 1024 0 commit-heavy; } > "$OUT/${TAG}_soak.txt" 2>&1
for p in Plaza1 Plaza2; do
  echo "== slam_hip --dataset $p --mcmc-slam --mcmc-steps 10 --mcmc-end-steps 1000 --sensor-range-stddev 1 --seed 42"
  timeout -k 10 120 slam_amd/host/slam_hip --dataset $p --dataset-dir tests/golden/plaza --mcmc-slam --mcmc-steps 10 \
    --mcmc-end-steps 1000 --sensor-range-stddev 1 --seed 42 2>&1 | tail -4 || true
done > "$OUT/${TAG}_host_program.txt" 2>&1
