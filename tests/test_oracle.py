"""CPU tests of the oracle (the restated reference algorithm): the reference's own
tree known-answer checks, the pinned random stream, the committed golden vectors,
and the invariants the reference offers (incremental == full log likelihood)."""
import json
import os
import subprocess

import numpy as np
import pytest

import oracle_lib
from golden.make_golden import SEED, fnv1a64
from slam_amd import synthetic

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_tree_known_answers():
    """All four checks of the reference's test/bitree.cpp, same sizes and seed."""
    oracle_lib.lib()  # builds oracle/ if needed
    exe = os.path.join(ROOT, "oracle", "bitree_kat")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.split() == ["check", "1:", "1", "check", "2:", "1", "check", "3:", "1", "check", "4:", "1"]


def test_random_stream_matches_libstdcxx():
    """Explicit mt19937 / generate_canonical / Marsaglia-polar against <random> of this toolchain."""
    for seed in (1, 5489, 123456789):
        assert oracle_lib.lib().ora_selftest_rng(seed, 20000) == 0


def test_seed_sequence_known_values():
    # first outputs of mt19937(5489), the standard's check values
    s = oracle_lib.seed_sequence(5489, 3)
    assert list(s) == [3499211612, 581869302, 3890346734]


@pytest.mark.parametrize("name", ["Plaza1", "Plaza2"])
def test_loader_and_pose_arithmetic_against_the_reference_controls_txt(name, plaza_dir):
    """The one output of the reference program that ships with it: input/<name>/controls.txt, written by the
    reference's own loader (src/dataset.cpp:96-109) -- per timestep dt, velocity_slip_model::observe(ground-truth
    pose change) / dt and the commanded control, at the stream's default precision of 6 significant digits.
    The oracle recomputes all 9 657 + 4 090 rows from the logs; that pins its DR/GT loader and dt, bitree<pose>
    push_back_accumulated / get (bitree.hpp:256-264, 293-300), pose inverse and composition (pose.hpp:67-76),
    velocity_slip_model::observe (velocity_model.hpp:37-49) and wrap_angle against the reference itself."""
    rows = oracle_lib.plaza_controls_rows(plaza_dir, name)
    with open(os.path.join(plaza_dir, name, "controls.txt")) as f:
        lines = f.read().splitlines()
    assert lines[0].startswith("#")
    ref_lines = lines[1:]
    assert len(ref_lines) == rows.shape[0] == {"Plaza1": 9657, "Plaza2": 4090}[name]
    ref = np.array([[float(x) for x in line.split()] for line in ref_lines])
    # half a unit of the sixth significant digit is 5e-6 relative; nothing may be further away
    err = np.abs(rows - ref) / np.maximum(np.abs(ref), 1e-300)
    assert err.max() <= 5.0e-6, f"row {np.unravel_index(np.argmax(err), err.shape)}: rel err {err.max():.3e}"
    # ... and printed the way operator<< prints a double, the rows are the reference's text
    mine = ["%g %g %g %g %g %g" % tuple(r) for r in rows]
    different = [t for t, (a, b) in enumerate(zip(mine, ref_lines)) if a != b]
    assert not different, f"{len(different)} rows differ as text, first: {different[0]}: {mine[different[0]]!r}"


def unhex(x):
    return float.fromhex(x)


@pytest.mark.parametrize("name", ["Plaza2", "Plaza1"])
def test_golden_vectors(name, plaza_dir):
    with open(os.path.join(ROOT, "tests", "golden", f"{name.lower()}_mcmc_slam.json")) as f:
        g = json.load(f)
    w = oracle_lib.World()
    w.load_plaza(plaza_dir, name, 1.0)
    inp = w.get_input()
    T, N = w.dims()
    assert (T, N) == (g["loader"]["T"], g["loader"]["N"])
    for i, (t, fid, mu) in enumerate(g["loader"]["first"]):
        assert (int(inp["obs_t"][i]), int(inp["obs_id"][i]), float(inp["obs_mean"][i])) == (t, fid, unhex(mu))
    for j, (t, fid, mu) in enumerate(g["loader"]["last"]):
        i = N - 5 + j
        assert (int(inp["obs_t"][i]), int(inp["obs_id"][i]), float(inp["obs_mean"][i])) == (t, fid, unhex(mu))
    assert [v.hex() for v in inp["ctrl_mean"].sum(axis=0)] == g["loader"]["sum_ctrl_mean"]
    assert [v.hex() for v in inp["ctrl_std"].sum(axis=0)] == g["loader"]["sum_ctrl_std"]
    seeds = oracle_lib.seed_sequence(SEED, 6)
    w.set_initialiser(int(seeds[1]))
    c = w.add_chain(int(seeds[2]), steps_per_timestep=10, end_steps=1000, use_initialiser=True, trace=True)
    w.replay(T, finish=True)
    tr = w.trace(c)
    assert len(tr) == g["steps"]
    assert int(tr["accepted"].sum()) == g["accepted"]
    assert "%016x" % fnv1a64(tr["index"]) == g["index_hash"]
    assert "%016x" % fnv1a64(tr["kind"]) == g["kind_hash"]
    assert "%016x" % fnv1a64(tr["accepted"]) == g["accept_hash"]
    for key, rec in g["records"].items():
        r = tr[int(key)]
        assert (int(r["kind"]), int(r["index"]), int(r["accepted"])) == (rec["kind"], rec["index"], rec["accepted"])
        assert [float(v).hex() for v in r["proposed"]] == rec["proposed"]
        assert float(r["log_ratio"]).hex() == rec["log_ratio"]
        assert float(r["log_likelihood"]).hex() == rec["log_likelihood"]
    st = w.state(c)
    assert [[float(v).hex() for v in st["cum"][k]] for k in range(0, T + 1, 64)] == g["cum_every_64"]
    assert [int(v) for v in st["feat_id"]] == g["feat_id"]
    assert [int(v) for v in st["feat_parent"]] == g["feat_parent"]
    assert float(w.log_likelihood(c)).hex() == g["log_likelihood"]
    assert float(w.full_log_likelihood(c)).hex() == g["full_log_likelihood"]
    assert w.units(c) == g["units"]
    # the one invariant the reference offers: incremental == full log likelihood
    assert abs(w.log_likelihood(c) - w.full_log_likelihood(c)) <= 1e-9 * abs(w.log_likelihood(c))


def test_plaza_dataset_facts(plaza_dir):
    """Sizes of the shipped datasets: controls, readings after (timestep, beacon) de-duplication, beacons."""
    for name, T, N in (("Plaza1", 9657, 3496), ("Plaza2", 4090, 1816)):
        w = oracle_lib.World()
        w.load_plaza(plaza_dir, name, 1.0)
        assert w.dims() == (T, N)
        inp = w.get_input()
        assert len(set(inp["obs_id"].tolist())) == 4
        assert np.all(np.diff(inp["obs_t"].astype(np.int64)) >= 0)
        assert np.all(inp["ctrl_std"] > 0)


def small(seed=3, T=300):
    return synthetic.make_problem(T=T, n_landmarks=12, k_nearest=4, seed=seed, lap_steps=150, wobble_period=500,
                                  range_stddev=2.0, landmark_spread=6.0)


def test_incremental_equals_full_log_likelihood_synthetic():
    p = small()
    w = oracle_lib.World()
    w.set_input(p["ctrl_mean"], p["ctrl_std"], p["obs_t"], p["obs_f"], p["obs_mean"], p["obs_std"])
    for s in oracle_lib.seed_sequence(9, 3):
        w.add_chain(int(s))
    w.replay(p["T"])
    for c in range(3):
        assert w.update(c, 1500) >= 0
        a, b = w.log_likelihood(c), w.full_log_likelihood(c)
        assert abs(a - b) <= 1e-9 * max(1.0, abs(a))


def test_draw_order_switch_changes_the_stream_only_for_landmarks():
    p = small(seed=5)
    traces = []
    for uniform_first in (True, False):
        w = oracle_lib.World()
        w.set_input(p["ctrl_mean"], p["ctrl_std"], p["obs_t"], p["obs_f"], p["obs_mean"], p["obs_std"])
        c = w.add_chain(77, uniform_first=uniform_first, trace=True)
        w.replay(p["T"])
        w.update(c, 200)
        traces.append(w.trace(c))
    a, b = traces
    first_landmark = int(np.argmax(a["kind"] == 1))
    assert np.array_equal(a["proposed"][:first_landmark], b["proposed"][:first_landmark])
    assert not np.array_equal(a["proposed"][first_landmark], b["proposed"][first_landmark])


def test_load_state_is_consistent():
    p = small(seed=8, T=500)
    w = oracle_lib.World()
    w.set_input(p["ctrl_mean"], p["ctrl_std"], p["obs_t"], p["obs_f"], p["obs_mean"], p["obs_std"])
    w.replay(p["T"])
    c = w.add_detached_chain(3)
    w.load_state(c, p["true_rel"], np.arange(p["F"]), p["landmarks"])
    assert w.log_likelihood(c) == w.full_log_likelihood(c)
    st = w.state(c)
    np.testing.assert_allclose(st["cum"], p["true_cum"], rtol=0, atol=1e-9)
    # parent = the closest reading, first one on ties
    for f in range(p["F"]):
        sel = p["obs_f"] == f
        t, mu = p["obs_t"][sel], np.abs(p["obs_mean"][sel])
        assert st["feat_parent"][f] == t[np.argmin(mu)]
    assert w.update(c, 300) >= 0
    assert abs(w.log_likelihood(c) - w.full_log_likelihood(c)) <= 1e-9 * abs(w.log_likelihood(c))


def test_range_bearing_incremental_equals_full():
    """The reference's one invariant (incremental log likelihood == slam_log_likelihood) for the
    range-and-bearing observation model, which the reference's default build does not compile."""
    from slam_amd import synthetic

    p = synthetic.make_problem(T=200, n_landmarks=8, k_nearest=3, seed=17, lap_steps=150, wobble_period=500,
                               range_stddev=2.0, landmark_spread=6.0, bearing_stddev=0.05)
    w = oracle_lib.World()
    w.set_input(p["ctrl_mean"], p["ctrl_std"], p["obs_t"], p["obs_f"], p["obs_mean"], p["obs_std"])
    w.set_bearings(p["obs_bearing"], p["obs_bearing_std"])
    c = w.add_chain(123, steps_per_timestep=3, end_steps=300, trace=True)
    w.replay(p["T"], finish=True)
    inc, full = w.log_likelihood(c), w.full_log_likelihood(c)
    assert abs(inc - full) <= 1e-9 * abs(full), (inc, full)
    tr = w.trace(c)
    assert 0 < tr["accepted"].sum() < len(tr)
