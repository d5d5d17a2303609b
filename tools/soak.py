#!/usr/bin/env python3
"""Long run on the benchmark's problem with the invariants checked as it goes: after every launch the
incremental log likelihood of every chain against the from-scratch one (1e-9 relative), no chain stopped,
counters consistent.  tools/soak.py [launches=50] [updates per launch=1000] [rebuild every=5] [chains=1024]
[split slices=0: the persistent kernel; >0: split mode with that many slices] [problem=bench: the benchmark's;
commit-heavy: 10 000 poses, 2 nearest landmarks, sigma 3 m -- one proposal in eight is accepted, so the
pose table, the kept log densities and the landmarks' parent copies are rewritten all the time]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_amd import capi, synthetic  # noqa: E402


def main():
    launches = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    per = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    rebuild_every = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    heavy = len(sys.argv) > 6 and sys.argv[6] == "commit-heavy"
    if heavy:
        p = synthetic.make_problem(T=10_000, n_landmarks=1000, k_nearest=2, seed=12345, range_stddev=3.0)
    else:
        p = synthetic.make_problem(T=100_000, n_landmarks=1000, k_nearest=10, seed=12345)
    C = int(sys.argv[4]) if len(sys.argv) > 4 else 1024
    slices = int(sys.argv[5]) if len(sys.argv) > 5 else 0
    g = capi.McmcGpu(C, p["T"], p["F"], p["N"], max_split_slices=slices, split_commit_min_poses=1 if slices else 0)
    g.upload_problem(p["ctrl_mean"], p["ctrl_std"], p["obs_t"], p["obs_f"], p["obs_mean"], p["obs_std"], p["F"])
    g.init_chains(capi.seed_sequence(777, C))
    g.load_state(p["true_rel"], p["landmarks"])
    worst = 0.0
    t0 = time.perf_counter()
    for i in range(launches):
        if slices:
            g.run_split(per, slices)
        else:
            g.run(per)
        if rebuild_every and (i + 1) % rebuild_every == 0:
            g.rebuild()
        g.sync()  # raises if a chain set its error flag
        ll, full = g.log_likelihood(), g.full_log_likelihood()
        rel = np.abs(ll - full) / np.abs(full)
        worst = max(worst, float(rel.max()))
        upd, acc, _ = g.counters()
        assert np.all(upd == (i + 1) * per), "update counters"
        print(f"launch {i + 1}: drift {rel.max():.2e} (worst {worst:.2e}), accepts per chain "
              f"{acc.mean():.1f}, best ll {ll.max():.3f}, {time.perf_counter() - t0:.0f} s", flush=True)
        assert rel.max() < 1e-9, "incremental log likelihood drifted from the from-scratch value"
    mode = f"split mode, {slices} slices, pose updates by split_commit_kernel" if slices else "persistent kernel"
    print(f"soak ok ({mode}): {launches * per} updates on each of {C} chains, worst drift {worst:.2e}")


if __name__ == "__main__":
    main()
