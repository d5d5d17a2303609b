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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chains", type=int, default=1024)
    ap.add_argument("--updates", type=int, default=100)
    ap.add_argument("--timesteps", type=int, default=100_000)
    ap.add_argument("--landmarks", type=int, default=1000)
    ap.add_argument("--threads-per-chain", type=int, default=0)
    a = ap.parse_args()
    p = synthetic.make_problem(T=a.timesteps, n_landmarks=a.landmarks)
    g = capi.McmcGpu(a.chains, p["T"], p["F"], p["N"], threads_per_chain=a.threads_per_chain)
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
    launches = 2  # the warm-up launch was reset away; slot 7 counts the timed launch only after the reset
    hist = [(int(cyc[7]) >> (16 * k)) & 0xFFFF for k in range(4)]
    print(f"  driver wavefronts per SIMD id (one per chain and launch): {hist}")


if __name__ == "__main__":
    main()
