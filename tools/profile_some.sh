#!/bin/bash
# tools/profile_all.sh for a subset of its workloads (a GPU call is limited to 20 minutes):
#   tools/profile_some.sh r3 chains sigma01 ...
set -e
TAG=$1; shift
R=$(pwd)
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp
export TMPDIR=/tmp
run() {  # name, seconds allowed, bench arguments...
  local name=$1 limit=$2; shift 2
  echo "== $name: bench.py $*"
  timeout -k 10 "$limit" python3 "$R/bench.py" "$@" --keep-counter-csv "$OUT/${TAG}_$name" \
    > "$OUT/${TAG}_bench_$name.json" 2> "$OUT/${TAG}_bench_$name.err"
  timeout -k 10 "$limit" rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_$name" -- \
    python3 "$R/bench.py" "$@" --steps 4 --warmup 1 --no-cpu-baseline --no-plaza --no-live-traffic > "$OUT/stats_$name.log" 2>&1
  cp "$(ls "$OUT"/stats_$name/*/*kernel_stats.csv | head -1)" "$OUT/${TAG}_${name}_kernel_stats.csv"
  python3 - "$OUT/${TAG}_bench_$name.json" <<'PY'
import json, sys
r = json.loads([ln for ln in open(sys.argv[1]) if ln.startswith("{")][-1])
f = r["roofline"]
print(f"   value {r['value']:.4g} {r['unit']}, {r['ms_per_step']:.2f} ms per step, accept rate {r['accept_rate']:.4f}")
print(f"   roofline.frac {f['frac']} (traffic {f['traffic']} B per launch, kernel {f['kernel_ms']:.3f} ms), "
      f"algorithmic_frac {f['algorithmic_frac']:.3f}")
PY
}
for w in "$@"; do
  case $w in
    chains) run chains 400 ;;
    sigma01) run sigma01 300 --range-stddev 0.1 --no-cpu-baseline --no-plaza ;;
    range_bearing) run range_bearing 300 --workload range-bearing --steps 10 --warmup 2 ;;
    commit_heavy) run commit_heavy 300 --workload commit-heavy --steps 10 --warmup 2 ;;
    burn_in) run burn_in 300 --workload burn-in --beta 1e-3 --steps 10 --warmup 2 ;;
    single_chain) run single_chain 300 --chains 1 --steps 5 --warmup 2 --no-cpu-baseline --no-plaza ;;
    single_chain_split64) run single_chain_split64 300 --chains 1 --split-slices 64 --steps 5 --warmup 2 --no-cpu-baseline --no-plaza ;;
    tempering_split256) run tempering_split256 600 --workload tempering --split-slices 256 --steps 5 --warmup 2 --live-traffic-seconds 400 ;;
  esac
done
