#!/usr/bin/env python3
"""Per-phase cycle shares of the advance kernel on the benchmark workload.

Uses the diagnostic twin of the library (libmcmcgpu_diag.so, built with
-DMG_PHASE_TIMING); its numbers are shares of lane-0 time per phase, never a
throughput figure.  Run on the GPU box:  python tools/phase_profile.py
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("MCMCGPU_LIB", os.path.join(ROOT, "slam_amd", "libmcmcgpu_diag.so"))
sys.path.insert(0, ROOT)

from slam_amd import capi, synthetic  # noqa: E402

NAMES = ["select+propose", "state: landmark set-up", "state: reading stream", "landmark edge eval",
         "reduce + accept", "commit", "whole kernel", "-"]


FINE = [(8, "weight totals"), (9, "uniform that picks the tree"), (10, "state tree descent (+ its uniform)"),
        (11, "three normals"), (12, "inv_observe"), (13, "control log density")]


def fine(cyc, nprop, nstate):
    """Inside selection and proposal: lane-0 cycles per state-edge proposal of each part."""
    nstate = max(nstate, 1)
    print("  inside select+propose, per state-edge proposal:")
    for slot, name in FINE[2:]:
        print(f"    {name:36s} {cyc[slot] / nstate:10.0f} cycles")
    print(f"    {'landmark proposal, descent to end':36s} {cyc[14] / max(nprop - nstate, 1):10.0f} cycles per landmark proposal")
    for slot, name in FINE[:2]:
        print(f"    {name:36s} {cyc[slot] / max(nprop, 1):10.0f} cycles per update")
    print(f"    {'growth (new edge, arrivals)':36s} {cyc[15] / max(nprop, 1):10.0f} cycles per update")
    inside = float(sum(cyc[k] for k in range(8, 16)))
    # slot 0 + slot 4 = every span from before the serial program's call to after the barrier behind it;
    # slot 7 = the same spans measured inside the program, first to last instruction
    call = float(cyc[0]) + float(cyc[4]) - float(cyc[7])
    print(f"    {'the call itself (save, reload, barrier)':36s} {call / max(nprop, 1):10.0f} cycles per update")
    print(f"    {'driver state machine and stores':36s} {(float(cyc[0]) - inside - call) / max(nprop, 1):10.0f} cycles per update")


def plaza(a):
    """One chain, --mcmc-steps 10 --mcmc-end-steps 1000 over the whole log, as bench.py times it."""
    import subprocess
    import tempfile
    import time

    import numpy as np

    from slam_amd import dataset_cache
    data_dir = os.path.join(ROOT, "tests", "golden", "plaza")
    exe = os.path.join(ROOT, "slam_amd", "host", "slam_hip")
    with tempfile.TemporaryDirectory() as tmp:
        cache = os.path.join(tmp, a.plaza + ".slamro")
        subprocess.check_call([exe, "--dataset", a.plaza, "--dataset-dir", data_dir, "--dataset-cache", cache,
                               "--dump-loader"], stdout=subprocess.DEVNULL)
        data = dataset_cache.read(cache)
    T, N = data["T"], data["N"]
    ctrl_mean, ctrl_std = synthetic.control_models(data["controls"], data["dt"])
    obs_f, order = dataset_cache.dense_landmarks(data["obs_id"])
    seeds = capi.seed_sequence(42, 6)
    g = capi.McmcGpu(1, T, len(order), N, threads_per_chain=a.threads_per_chain, serial_program=a.serial_program)
    g.upload_problem(ctrl_mean, ctrl_std, data["obs_t"], obs_f, data["obs_range"], np.full(N, 1.0), len(order))
    g.init_chains(np.array([seeds[2]], dtype=np.uint32), capi.uniform_sequence(int(seeds[1]), len(order)))
    g.debug_phase_cycles(reset=True)
    t0 = time.perf_counter()
    g.advance(T, steps_per_timestep=10, end_steps=1000)
    g.sync()
    wall = time.perf_counter() - t0
    cyc = g.debug_phase_cycles()
    u = g.counters()
    s = g.traffic_counters()
    nprop, nacc, units = int(u[0][0]), int(u[1][0]), int(u[2][0])
    whole = float(cyc[6])
    print(f"{a.plaza}: {T} timesteps, {N} readings, wall {wall:.3f} s, {nprop} updates ({int(s[0][0])} state edges), "
          f"{nacc} accepted, {units / max(nprop, 1):.0f} units per update, {int(s[1][0]) / max(nacc, 1):.0f} poses moved per accept")
    print(f"whole kernel {whole / 1e6:.1f} Mcycles = {whole / max(nprop, 1):.0f} cycles per update")
    if a.serial_program in (0, 3) and (a.threads_per_chain == 0 or a.threads_per_chain >= 128):
        # the run-ahead variant books its cycles differently (advance_ahead in chain_kernels.hpp)
        ev = max(int(cyc[8]), 1)
        for slot, name in [(0, "first wavefront: serial step"), (1, "first wavefront: work ahead"),
                           (2, "first wavefront: waiting for the evaluation"), (3, "others: evaluation"),
                           (4, "others: waiting for the first wavefront"), (5, "commit")]:
            print(f"  {name:44s} {100.0 * cyc[slot] / whole:6.2f} % of kernel   {cyc[slot] / max(nprop, 1):10.0f} cycles per update")
        for slot, name in [(7, "  serial step: end of the update (decide, apply)"), (14, "  serial step: on to an update selected ahead"),
                           (15, "  serial step: on to growth or a fresh selection")]:
            print(f"  {name:52s} {cyc[slot] / max(nprop, 1):10.0f} cycles per update")
        print(f"  evaluations {int(cyc[8])}: next update selected ahead {int(cyc[9])} ({100.0 * cyc[9] / ev:.1f} %), taken {int(cyc[10])} "
              f"({100.0 * cyc[10] / ev:.1f} %), given up early {int(cyc[11])}, finished but dropped {int(cyc[12])}, "
              f"out of generator words {int(cyc[13])}")
        return
    for i in range(6):
        print(f"  {NAMES[i]:28s} {100.0 * cyc[i] / whole:6.2f} % of kernel   {cyc[i] / max(nprop, 1):10.0f} cycles per update")
    rest = whole - float(cyc[:6].sum())
    print(f"  {'growth and the rest':28s} {100.0 * rest / whole:6.2f} % of kernel   {rest / max(nprop, 1):10.0f} cycles per update")
    fine(cyc, nprop, int(s[0][0]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chains", type=int, default=1024)
    ap.add_argument("--updates", type=int, default=100)
    ap.add_argument("--timesteps", type=int, default=100_000)
    ap.add_argument("--landmarks", type=int, default=1000)
    ap.add_argument("--k-nearest", type=int, default=10)
    ap.add_argument("--range-stddev", type=float, default=1.0)
    ap.add_argument("--threads-per-chain", type=int, default=0)
    ap.add_argument("--serial-program", type=int, default=0, help="mcmcgpu_config.serial_program")
    ap.add_argument("--plaza", default="", help="Plaza1 or Plaza2: the online replay of bench.py's plaza line instead")
    a = ap.parse_args()
    if a.plaza:
        return plaza(a)
    p = synthetic.make_problem(T=a.timesteps, n_landmarks=a.landmarks, k_nearest=a.k_nearest, range_stddev=a.range_stddev)
    g = capi.McmcGpu(a.chains, p["T"], p["F"], p["N"], threads_per_chain=a.threads_per_chain, serial_program=a.serial_program)
    g.upload_problem(p["ctrl_mean"], p["ctrl_std"], p["obs_t"], p["obs_f"], p["obs_mean"], p["obs_std"], p["F"])
    g.init_chains(capi.seed_sequence(12347, a.chains))
    g.load_state(p["true_rel"], p["landmarks"])
    g.run(8)
    g.sync()
    g.debug_phase_cycles(reset=True)
    g.kernel_time()
    u0 = g.counters()
    s0 = g.traffic_counters()
    g.run(a.updates)
    g.sync()
    ms, _ = g.kernel_time()
    cyc = g.debug_phase_cycles()
    u1 = g.counters()
    s1 = g.traffic_counters()
    units = int((u1[2] - u0[2]).sum())
    nstate = int((s1[0] - s0[0]).sum())
    nprop = int((u1[0] - u0[0]).sum())
    whole = float(cyc[6])
    print(f"kernel {ms:.2f} ms, {units / ms / 1e6:.1f} G units/s, {nprop} proposals ({nstate} state edges), "
          f"{units / max(nstate, 1):.0f} units per state edge")
    busy = float(cyc[:6].sum())
    print(f"mean chain duration {whole / a.chains / 1e6:.2f} Mcycles vs launch {ms:.2f} ms "
          f"-> {whole / a.chains / (ms * 1e3):.0f} cycles/us if every chain ran the whole launch "
          f"(the shader clock is ~2100-2400; lower = chains idle waiting for the slowest)")
    for i in range(6):
        per = cyc[i] / max(nprop, 1)
        print(f"  {NAMES[i]:28s} {100.0 * cyc[i] / busy:6.2f} % of busy   {per:12.0f} cycles per proposal")
    print(f"  reading stream: {cyc[2] / max(units, 1) * 256:.1f} lane-0 cycles per 256 readings")
    fine(cyc, nprop, nstate)


if __name__ == "__main__":
    main()
