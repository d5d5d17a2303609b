#!/usr/bin/env python3
"""Rate of the pose-table rebuild (mcmcgpu_rebuild: the SE(2) prefix scan) on its own, at the shapes the
benchmarks use: 48 bytes per pose (24 read, 24 written) over the time the library's own HIP events measure.
Run on the GPU box:  python tools/scan_bench.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from slam_amd import capi, synthetic  # noqa: E402


def main():
    for chains, T in ((1024, 100_000), (256, 100_000), (64, 1_000_000), (8, 1_000_000), (1, 1_000_000)):
        p = synthetic.make_problem(T=T, n_landmarks=4, k_nearest=1, seed=5, max_tries=1)
        g = capi.McmcGpu(chains, p["T"], p["F"], p["N"])
        g.upload_problem(p["ctrl_mean"], p["ctrl_std"], p["obs_t"], p["obs_f"], p["obs_mean"], p["obs_std"], p["F"])
        g.init_chains(capi.seed_sequence(1, chains))
        g.load_state(p["true_rel"], p["landmarks"])
        g.sync()
        for _ in range(3):
            g.rebuild()
        g.sync()
        g.rebuild_time()
        for _ in range(20):
            g.rebuild()
        g.sync()
        ms, n = g.rebuild_time()
        cum = g.get_state(chains - 1)["cum"]
        err = np.abs(cum - p["true_cum"]).max()
        gb = 48.0 * T * chains / 1e9
        print(f"{chains:5d} chains x {T:8d} poses: {ms:8.3f} ms per rebuild ({n} timed), {gb / ms:7.2f} TB/s at 48 B per pose, "
              f"max |difference| to a sequential composition {err:.2e}", flush=True)
        g.close()


if __name__ == "__main__":
    main()
