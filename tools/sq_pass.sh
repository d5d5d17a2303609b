#!/bin/bash
# One SQ counter pass over the benchmark command (runs on the GPU box, from the repository root):
#   tools/sq_pass.sh <tag> <counters...>
set -e
TAG=$1; shift
R=$(pwd)
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp
export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/sq" -- \
  python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-plaza --no-live-traffic > "$OUT/sq.log" 2>&1
cd "$R"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/sq/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "advance_kernel" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"{k:28s} {sum(v) / len(v):.6g}  ({len(v)} launches)")
PY
