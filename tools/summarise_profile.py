#!/usr/bin/env python3
"""Condenses one tools/profile_round.sh output directory into a text summary:
the benchmark line, per-kernel time from rocprofv3 --kernel-trace --stats, HBM
traffic of the dominant kernel from the FETCH_SIZE / WRITE_SIZE counter passes
(with the gfx950 correction: FETCH_SIZE reports half of a wide coalesced read
stream, so the read side is given as a [x1, x2] bracket), and the phase shares."""
import csv
import glob
import json
import os
import sys


def first(pattern):
    files = glob.glob(pattern, recursive=True)
    return files[0] if files else None


def counter_mean(path, kernel, counter):
    vals = []
    with open(path) as f:
        for r in csv.DictReader(f):
            if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter:
                vals.append(float(r["Counter_Value"]))
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)


def main():
    out = sys.argv[1]
    bench = None
    try:
        with open(os.path.join(out, "bench.json")) as f:
            lines = [ln for ln in f.read().splitlines() if ln.startswith("{")]
        bench = json.loads(lines[-1])
    except Exception as e:  # noqa: BLE001
        print("no bench line:", e)
    if bench:
        r = bench["roofline"]
        print("== bench ==")
        print(f"value {bench['value']:.4g} {bench['unit']}  ms/step {bench['ms_per_step']:.2f}  "
              f"updates/step {bench['config']['updates_per_step']}")
        if r.get("frac") is not None:
            print(f"roofline (counters, bytes leaving L2): {r['achieved']:.1f} GB/s of {r['peak']:.0f} = {r['frac']:.3f}  "
                  f"traffic/launch {r['traffic']:.4g} B  ({(r.get('traffic_source') or {}).get('method')})")
        print(f"algorithmic: {r['algorithmic_achieved']:.1f} GB/s = {r['algorithmic_frac']:.3f} of peak  "
              f"kernel {r['kernel_ms']:.3f} ms x {r['launches']}  algorithmic bytes/launch {r['algorithmic_bytes_per_launch']:.4g}")
        if r.get("hbm_only"):
            print(f"HBM only (bracket): {r['hbm_only']['low_frac']:.3f} .. {r['hbm_only']['high_frac']:.3f} of peak")
        if "cpu_baseline" in bench:
            c = bench["cpu_baseline"]
            print(f"cpu_baseline: {c['value']:.4g} {c['unit']} on {c['cores']} core ({c['sample']})")
        for name in ("plaza1", "plaza2"):
            if name in bench:
                print(name + ":", bench[name])
        if "cpu_baseline" in bench and "all_cores" in bench["cpu_baseline"]:
            a = bench["cpu_baseline"]["all_cores"]
            print(f"cpu, one chain per thread: {a['value']:.4g} on {a['cores']} threads ({a['sample']})")
    stats = first(os.path.join(out, "stats", "**", "*kernel_stats.csv"))
    if stats:
        print("== rocprofv3 --kernel-trace --stats (same bench command, 2 warm-up + 6 timed launches) ==")
        with open(stats) as f:
            for r in csv.DictReader(f):
                if float(r["Percentage"]) >= 0.01:
                    print(f"{r['Name'][:60]:60s} calls {r['Calls']:>4s}  avg {float(r['AverageNs']) / 1e6:10.3f} ms  "
                          f"{float(r['Percentage']):6.2f} %")
    fetch = first(os.path.join(out, "fetch", "**", "*counter_collection.csv"))
    write = first(os.path.join(out, "write", "**", "*counter_collection.csv"))
    if fetch and write:
        fv, n = counter_mean(fetch, "advance_kernel", "FETCH_SIZE")
        wv, _ = counter_mean(write, "advance_kernel", "WRITE_SIZE")
        if fv is not None and wv is not None:
            print("== HBM traffic of advance_kernel per launch (rocprofv3 --pmc, KiB units -> bytes) ==")
            print(f"FETCH_SIZE {fv * 1024:.4g} B as reported ({n} launches); gfx950 reports half of a wide "
                  f"coalesced stream, so reads are in [{fv * 1024:.4g}, {2 * fv * 1024:.4g}] B")
            print(f"WRITE_SIZE {wv * 1024:.4g} B")
            if bench:
                # what bench.py quotes as roofline.traffic for the same workload: reads doubled (calibrated on
                # this gather pattern with tools/stream_probe: FETCH_SIZE tallies it at about half) + writes
                import datetime
                sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
                import bench as bench_module
                sha = bench_module.library_sha16()
                rec = {"workload": bench["config"]["workload"], "updates_per_step": bench["config"]["updates_per_step"],
                       "library_sha16": sha, "date": datetime.date.today().isoformat(), "commit": os.environ.get("GRAFT_COMMIT"),
                       "fetch_size_reported_bytes": fv * 1024, "write_size_bytes": wv * 1024,
                       "traffic_bytes_per_launch": 2 * fv * 1024 + wv * 1024, "launches_averaged": n}
                with open(os.path.join(out, "hbm_traffic.json"), "w") as f:
                    json.dump(rec, f, indent=1)
    ph = os.path.join(out, "phases.txt")
    if os.path.exists(ph):
        print("== phase shares (diagnostic build, lane-0 shader-clock stamps; shares only) ==")
        print(open(ph).read().strip())


if __name__ == "__main__":
    main()
