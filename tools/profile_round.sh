#!/bin/bash
# Runs on the GPU box (from the repository root): the benchmark, a rocprofv3
# kernel-trace pass of the same command, the two HBM counter passes and the
# phase profile.  Everything lands in gpurun_out/<tag>/; copy what should be
# kept into profiles/.
#   tools/profile_round.sh r1 [extra bench.py arguments, e.g. --workload range-bearing]
set -e
TAG=${1:-run}
shift || true
EXTRA="$*"
R=$(pwd)
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
BENCH_ARGS="--steps 6 --warmup 2 --no-cpu-baseline --no-plaza --no-live-traffic $EXTRA"
cd /tmp
export TMPDIR=/tmp
timeout -k 10 300 python3 "$R/bench.py" --steps 10 --warmup 2 $EXTRA > "$OUT/bench.json" 2> "$OUT/bench.err"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- \
  python3 "$R/bench.py" $BENCH_ARGS > "$OUT/stats.log" 2>&1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -- \
  python3 "$R/bench.py" $BENCH_ARGS > "$OUT/fetch.log" 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -- \
  python3 "$R/bench.py" $BENCH_ARGS > "$OUT/write.log" 2>&1
if [ -z "$EXTRA" ]; then timeout -k 10 300 python3 "$R/tools/phase_profile.py" > "$OUT/phases.txt" 2>&1; fi
cd "$R"
python3 tools/summarise_profile.py "$OUT" > "$OUT/summary.txt"
cat "$OUT/summary.txt"
