#!/usr/bin/env python3
"""Generates tests/golden/*.json from the CPU oracle.

The reference ships no golden vectors for the sampler, the models or the loader
(its only assertions are in test/bitree.cpp) and cannot be built here, so these
fixtures are produced by the restated algorithm itself: they pin the oracle
against regressions and give the GPU tests something stable to compare with.
Run from the repository root after `make -C oracle`:

    python tests/golden/make_golden.py

Contents per dataset (Plaza1/Plaza2, --mcmc-slam --mcmc-steps 10 --mcmc-end-steps 1000
--sensor-range-stddev 1 --seed 42): loader known answers, the first 2000 step
records and every 1000th after, an FNV-1a hash of the edge-index sequence, the
accept count, final cumulative poses every 64th timestep, the final map, and the
final incremental and full log likelihoods.  Floats are stored as hex strings.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib  # noqa: E402

SEED = 42


def fnv1a64(values):
    h = 0xCBF29CE484222325
    for v in values:
        for b in int(v).to_bytes(8, "little"):
            h ^= b
            h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def hexf(x):
    return float(x).hex()


def record(r):
    return {
        "kind": int(r["kind"]),
        "index": int(r["index"]),
        "accepted": int(r["accepted"]),
        "proposed": [hexf(v) for v in r["proposed"]],
        "log_ratio": hexf(r["log_ratio"]),
        "log_likelihood": hexf(r["log_likelihood"]),
    }


def run(name):
    w = oracle_lib.World()
    w.load_plaza(os.path.join(HERE, "plaza"), name, 1.0)
    inp = w.get_input()
    T, N = w.dims()
    seeds = oracle_lib.seed_sequence(SEED, 6)
    w.set_initialiser(int(seeds[1]))
    c = w.add_chain(int(seeds[2]), steps_per_timestep=10, end_steps=1000, use_initialiser=True, trace=True)
    w.replay(T, finish=True)
    tr = w.trace(c)
    st = w.state(c)
    keep = list(range(min(2000, len(tr)))) + list(range(2000, len(tr), 1000))
    out = {
        "dataset": name,
        "seed": SEED,
        "loader": {
            "T": T,
            "N": N,
            "first": [[int(inp["obs_t"][i]), int(inp["obs_id"][i]), hexf(inp["obs_mean"][i])] for i in range(5)],
            "last": [[int(inp["obs_t"][i]), int(inp["obs_id"][i]), hexf(inp["obs_mean"][i])]
                     for i in range(N - 5, N)],
            "sum_ctrl_mean": [hexf(v) for v in inp["ctrl_mean"].sum(axis=0)],
            "sum_ctrl_std": [hexf(v) for v in inp["ctrl_std"].sum(axis=0)],
        },
        "steps": len(tr),
        "accepted": int(tr["accepted"].sum()),
        "index_hash": "%016x" % fnv1a64(tr["index"]),
        "kind_hash": "%016x" % fnv1a64(tr["kind"]),
        "accept_hash": "%016x" % fnv1a64(tr["accepted"]),
        "records": {str(i): record(tr[i]) for i in keep},
        "cum_every_64": [[hexf(v) for v in st["cum"][k]] for k in range(0, T + 1, 64)],
        "feat_id": [int(v) for v in st["feat_id"]],
        "feat_parent": [int(v) for v in st["feat_parent"]],
        "feat_est": [[hexf(v) for v in row] for row in st["feat_est"]],
        "log_likelihood": hexf(w.log_likelihood(c)),
        "full_log_likelihood": hexf(w.full_log_likelihood(c)),
        "units": int(w.units(c)),
    }
    with open(os.path.join(HERE, f"{name.lower()}_mcmc_slam.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(name, "steps", len(tr), "accepted", out["accepted"], "LL", w.log_likelihood(c))


if __name__ == "__main__":
    for n in ("Plaza2", "Plaza1"):
        run(n)
