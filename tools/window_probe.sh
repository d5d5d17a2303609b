#!/bin/bash
# How much of advance_kernel's time is memory traffic?  Three diagnostic builds of the library, all
# with every proposal rejected (so the chains stay at their starting state and the three runs do
# the same work): as it is; with every pose gather confined to a 16 KB window of the chain's table;
# with every record load confined to a 48 KB window (which confines the pose gathers to those
# records' timesteps as well); and with 8-byte records whose timestep is arithmetic on the segment
# (the bound on a stream over runs of consecutive timesteps: no dependent load, 4 bytes fewer).  Results are wrong by construction; rates and FETCH_SIZE are what
# is read off.
#   tools/window_probe.sh build      here (hipcc cross-compiles): tools/diag/libmcmcgpu_win_*.so
#   tools/window_probe.sh            on the GPU box, from the repository root -> gpurun_out/window_probe/
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
if [ "$1" = build ]; then
  mkdir -p "$R/tools/diag"
  for v in NONE POSE RECORD RUN; do
    ${HIPCC:-/opt/rocm/bin/hipcc} -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -shared -std=c++17 \
      -DMG_DIAG_${v}_WINDOW -DMG_DIAG_${v}_RECORDS -DMG_DIAG_REJECT_ALL -Wall -Wno-unused-function \
      -o "$R/tools/diag/libmcmcgpu_win_$v.so" "$R/slam_amd/csrc/mcmcgpu.hip"
  done
  exit 0
fi
O=$R/gpurun_out/window_probe
mkdir -p "$O"
cd /tmp
export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-plaza --no-live-traffic"
for v in NONE POSE RECORD RUN; do
  export MCMCGPU_LIB=$R/tools/diag/libmcmcgpu_win_$v.so
  timeout -k 10 120 python3 "$R/bench.py" $ARGS > "$O/bench_$v.json" 2> "$O/bench_$v.err"
  timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$O/fetch_$v" -- \
    python3 "$R/bench.py" $ARGS > "$O/fetch_$v.log" 2>&1
done
cd "$R"
python3 - <<'PY' | tee gpurun_out/window_probe/summary.txt
import csv, datetime, glob, json, os, sys
sys.path.insert(0, os.getcwd())
import bench as bench_module
names = {"NONE": "as it is", "POSE": "pose gathers in a 16 KB window", "RECORD": "record loads in a 48 KB window",
         "RUN": "8-byte records, timestep from the segment"}
per_unit, rate, cfg = {}, {}, None
for v in ("NONE", "POSE", "RECORD", "RUN"):
    b = json.loads([ln for ln in open(f"gpurun_out/window_probe/bench_{v}.json") if ln.startswith("{")][-1])
    cfg = b["config"]
    units = b["value"] * b["ms_per_step"] / 1e3
    f = max(glob.glob(f"gpurun_out/window_probe/fetch_{v}/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
    tot = n = 0
    for r in csv.DictReader(open(f)):
        if "advance_kernel" in r["Kernel_Name"]:
            tot += float(r["Counter_Value"])
            n += 1
    reads = 2 * tot / max(n, 1) * 1024  # KiB as reported, 128-byte requests tallied at 64
    per_unit[v], rate[v] = reads / units, b["value"]
    print(f"{names[v]:34s} {b['value']:.4g} units/s, kernel {b['roofline']['kernel_ms']:.1f} ms, "
          f"reads leaving L2 {reads:.4g} B per launch = {reads / units:.1f} B per unit")
# what bench.py brackets HBM-only traffic with: pose lines = what confining the pose gathers removes, records =
# what is left once the record loads (and with them the pose gathers) are confined
rec = {"workload": cfg["workload"], "updates_per_step": cfg["updates_per_step"], "library_sha16": bench_module.library_sha16(),
       "date": datetime.date.today().isoformat(), "reads_bytes_per_unit": per_unit, "units_per_s": rate,
       "pose_line_bytes_per_unit": per_unit["NONE"] - per_unit["POSE"], "record_bytes_per_unit": per_unit["RECORD"]}
json.dump(rec, open("gpurun_out/window_probe/window_probe.json", "w"), indent=1)
print(f"pose-table lines {rec['pose_line_bytes_per_unit']:.1f} B per unit, reading records {rec['record_bytes_per_unit']:.1f} B per unit")
PY
