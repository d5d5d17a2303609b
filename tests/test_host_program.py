"""The C++ host program (slam_amd/host): the reference's command line over the C ABI.
CPU: its Plaza loader against the golden loader facts.  GPU: Plaza2 end to end
through the drop-in classes against the golden run of the oracle."""
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "slam_amd", "host", "slam_hip")
EXE_RB = os.path.join(ROOT, "slam_amd", "host", "slam_hip_rb")  # built with the range-and-bearing model
PLAZA = os.path.join(ROOT, "tests", "golden", "plaza")


def golden(name):
    with open(os.path.join(ROOT, "tests", "golden", f"{name.lower()}_mcmc_slam.json")) as f:
        return json.load(f)


def run(*args, timeout=600):
    out = subprocess.run([EXE, *args], capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, out.stdout + out.stderr
    return out.stdout


@pytest.mark.parametrize("name", ["Plaza1", "Plaza2"])
def test_loader_matches_golden_facts(name):
    """Same timesteps, same de-duplicated readings, same control models as the oracle's loader, bit for bit."""
    g = golden(name)["loader"]
    lines = run("--dataset", name, "--dataset-dir", PLAZA, "--sensor-range-stddev", "1", "--dump-loader").splitlines()
    kv = {ln.split()[0]: ln.split()[1:] for ln in lines if not ln.startswith("obs")}
    assert int(kv["T"][0]) == g["T"] and int(kv["N"][0]) == g["N"]
    obs = {int(ln.split()[1]): ln.split()[2:] for ln in lines if ln.startswith("obs")}
    for i, (t, fid, mu) in enumerate(g["first"]):
        assert (int(obs[i][0]), int(obs[i][1]), float.fromhex(obs[i][2])) == (t, fid, float.fromhex(mu))
    for j, (t, fid, mu) in enumerate(g["last"]):
        i = g["N"] - 5 + j
        assert (int(obs[i][0]), int(obs[i][1]), float.fromhex(obs[i][2])) == (t, fid, float.fromhex(mu))
    assert [float.fromhex(v) for v in kv["sum_ctrl_mean"]] == [float.fromhex(v) for v in g["sum_ctrl_mean"]]
    assert [float.fromhex(v) for v in kv["sum_ctrl_std"]] == [float.fromhex(v) for v in g["sum_ctrl_std"]]


@pytest.mark.parametrize("name", ["Plaza1", "Plaza2"])
def test_dataset_cache_round_trip(name, tmp_path):
    """Text logs -> binary cache -> loader: the same dataset, bit for bit, in C++ and in Python."""
    from slam_amd import dataset_cache

    cache = str(tmp_path / f"{name}.slamro")
    dump = ("--sensor-range-stddev", "1", "--dump-loader")
    from_text = run("--dataset", name, "--dataset-dir", PLAZA, *dump)
    writing = run("--dataset", name, "--dataset-dir", PLAZA, "--dataset-cache", cache, *dump)
    assert os.path.exists(cache) and writing == from_text
    from_cache = run("--dataset", name, "--dataset-dir", str(tmp_path / "nowhere"), "--dataset-cache", cache, *dump)
    assert from_cache == from_text
    # the Python reader sees what the oracle's text loader sees
    import oracle_lib

    d = dataset_cache.read(cache)
    w = oracle_lib.World()
    w.load_plaza(PLAZA, name, 1.0)
    inp = w.get_input()
    assert (d["T"], d["N"]) == w.dims()
    assert np.array_equal(d["obs_t"], inp["obs_t"]) and np.array_equal(d["obs_id"], inp["obs_id"])
    assert np.array_equal(d["obs_range"], inp["obs_mean"])
    assert np.array_equal(d["dt"][:, None] * d["controls"], inp["ctrl_mean"])
    # the Python control-model builder and landmark numbering (what bench.py feeds the GPU) equal the oracle's
    from slam_amd import synthetic

    cm, cs = synthetic.control_models(d["controls"], d["dt"])
    assert np.array_equal(cm, inp["ctrl_mean"]) and np.array_equal(cs, inp["ctrl_std"])
    dense, order = dataset_cache.dense_landmarks(d["obs_id"])
    dense_o, order_o = oracle_lib.dense_features(inp["obs_id"])
    assert np.array_equal(dense, dense_o) and np.array_equal(order, order_o)
    # and a cache written from Python reads back identically in C++
    again = str(tmp_path / "again.slamro")
    dataset_cache.write(again, d["dt"], d["controls"], d["obs_t"], d["obs_id"], d["obs_range"])
    assert open(again, "rb").read() == open(cache, "rb").read()
    assert run("--dataset", name, "--dataset-cache", again, *dump) == from_text


EXAMPLE = os.path.join(ROOT, "tests", "golden", "example")


@pytest.mark.parametrize("seed,repetitions", [(5, 1.0), (77, 2.5)])
def test_simulator_matches_oracle(seed, repetitions, tmp_path):
    """The simulated data source (waypoint controller + landmark sensor + simulator loop): the C++
    host's run equals the oracle's restatement of the reference's, bit for bit."""
    import oracle_lib
    from slam_amd import dataset_cache

    wp, lm = os.path.join(EXAMPLE, "waypoints.txt"), os.path.join(EXAMPLE, "landmarks.txt")
    cache = str(tmp_path / "sim.slamro")
    run("--waypoint-file", wp, "--landmark-file", lm, "--sim-seed", str(seed), "--waypoint-repetitions",
        str(repetitions), "--sensor-range-stddev", "1", "--dataset-cache", cache, "--dump-loader")
    d = dataset_cache.read(cache)
    w = oracle_lib.World()
    w.simulate(wp, lm, seed=seed, repetitions=repetitions, range_stddev=1.0)
    inp = w.get_input()
    assert (d["T"], d["N"]) == w.dims() and d["T"] > 500 and d["N"] > 100
    assert np.array_equal(d["obs_t"], inp["obs_t"]) and np.array_equal(d["obs_id"], inp["obs_id"])
    assert np.array_equal(d["obs_range"], inp["obs_mean"])
    assert np.array_equal(d["dt"][:, None] * d["controls"], inp["ctrl_mean"])


@pytest.mark.parametrize("name", ["Plaza1", "Plaza2"])
def test_host_loader_reproduces_the_reference_controls_txt(name, tmp_path):
    """The host loader's ground truth, pose arithmetic and bitree<pose> against the controls.txt the reference's
    own loader wrote (src/dataset.cpp:96-109, shipped as input/<name>/controls.txt): byte for byte."""
    out = tmp_path / "controls.txt"
    run("--dataset", name, "--dataset-dir", PLAZA, "--dump-loader", "--controls-out", str(out))
    with open(os.path.join(PLAZA, name, "controls.txt"), "rb") as f:
        assert out.read_bytes() == f.read()


def test_dataset_cache_rejects_garbage(tmp_path):
    bad = tmp_path / "bad.slamro"
    bad.write_bytes(b"not a cache at all" * 8)
    out = subprocess.run([EXE, "--dataset", "x", "--dataset-cache", str(bad), "--dump-loader"], capture_output=True,
                         text=True)
    assert out.returncode == 1 and "not a version-1" in out.stderr


def test_unknown_option_is_an_error():
    out = subprocess.run([EXE, "--dataset", "Plaza2", "--no-such-flag"], capture_output=True, text=True)
    assert out.returncode == 1 and "unknown option" in out.stderr


@pytest.mark.gpu
def test_plaza2_through_the_drop_in_classes():
    """--mcmc-slam with the dead-reckoning initialiser, event by event through slam_data, 10 updates per
    timestep and 1000 at the end: the final log likelihood equals the golden run's to 1e-9 relative."""
    g = golden("Plaza2")
    out = run("--dataset", "Plaza2", "--dataset-dir", PLAZA, "--mcmc-slam", "--mcmc-steps", "10", "--mcmc-end-steps",
              "1000", "--sensor-range-stddev", "1", "--seed", str(g["seed"]))
    ll = float([ln for ln in out.splitlines() if ln.startswith("MCMC-SLAM log likelihood")][0].split()[-1])
    ref = float.fromhex(g["log_likelihood"])
    assert abs(ll - ref) <= 1e-9 * abs(ref), (ll, ref)
    assert "timesteps: %d landmarks: 4" % g["loader"]["T"] in out


@pytest.mark.gpu
def test_multi_mcmc_report(tmp_path):
    out = subprocess.run([EXE, "--dataset", "Plaza2", "--dataset-dir", PLAZA, "--multi-mcmc", "--multi-mcmc-chains", "8",
                          "--multi-mcmc-end-steps", "250", "--sensor-range-stddev", "4", "--seed", "7"],
                         capture_output=True, text=True, timeout=600, cwd=tmp_path)
    assert out.returncode == 0, out.stdout + out.stderr
    rows = [ln.split("\t") for ln in open(tmp_path / "multi-mcmc-report.txt").read().splitlines()]
    assert [float(r[0]) for r in rows] == [100.0, 200.0, 250.0]  # updates per chain at each report
    best = [float(r[2]) for r in rows]
    assert best == sorted(best)  # the best log likelihood never decreases
    assert "Multi-MCMC-SLAM Chains: 8" in out.stdout
    # the report lines against the reference's loop (multi_mcmc.hpp:106-139) played with the oracle: chain after
    # chain, each compared right after ITS updates with the best chain so far
    import oracle_lib

    seeds = oracle_lib.seed_sequence(int(oracle_lib.seed_sequence(7, 6)[5]), 8)
    w = oracle_lib.World()
    w.load_plaza(PLAZA, "Plaza2", 4.0)
    for s in seeds:
        w.add_chain(int(s))
    w.replay(w.dims()[0])
    best, updates, accepted = 0, 0, 0
    for k, n in enumerate([100, 100, 50]):
        for c in range(8):
            acc = w.update(c, n)
            assert acc >= 0
            accepted += acc
            updates += n
            if w.log_likelihood(c) > w.log_likelihood(best):
                best = c
        assert float(rows[k][0]) == updates / 8
        assert abs(float(rows[k][1]) - accepted / 8) < 1e-4  # printed with six significant digits
        ref = w.log_likelihood(best)
        assert abs(float(rows[k][2]) - ref) <= 2e-6 * abs(ref), (k, rows[k][2], ref)
    avg = [ln for ln in out.stdout.splitlines() if ln.startswith("Averaged-Multi-MCMC log likelihood")]
    assert len(avg) == 1 and np.isfinite(float(avg[0].split()[-1])), out.stdout


@pytest.mark.gpu
def test_simulated_run_through_the_drop_in_classes():
    """No --dataset: the simulator feeds slam_data; --mcmc-slam with the initialiser, 5 updates per timestep and
    500 at the end; the final log likelihood equals the oracle's run on the oracle's simulation."""
    import oracle_lib

    wp, lm = os.path.join(EXAMPLE, "waypoints.txt"), os.path.join(EXAMPLE, "landmarks.txt")
    out = run("--waypoint-file", wp, "--landmark-file", lm, "--mcmc-slam", "--mcmc-steps", "5", "--mcmc-end-steps", "500",
              "--sensor-range-stddev", "1", "--seed", "9")
    ll = float([ln for ln in out.splitlines() if ln.startswith("MCMC-SLAM log likelihood")][0].split()[-1])
    seeds = oracle_lib.seed_sequence(9, 6)  # sim, init, mcmc-slam, hmc, fastslam, multi-mcmc
    w = oracle_lib.World()
    w.simulate(wp, lm, seed=int(seeds[0]), range_stddev=1.0)
    w.set_initialiser(int(seeds[1]))
    c = w.add_chain(int(seeds[2]), steps_per_timestep=5, end_steps=500, use_initialiser=True)
    w.replay(w.dims()[0], finish=True)
    ref = w.log_likelihood(c)
    assert abs(ll - ref) <= 1e-9 * abs(ref), (ll, ref)


def test_range_bearing_simulator_matches_oracle():
    """slam_hip_rb (the reference's other compile-time observation model): same simulated readings as the oracle."""
    import oracle_lib

    wp, lm = os.path.join(EXAMPLE, "waypoints.txt"), os.path.join(EXAMPLE, "landmarks.txt")
    out = subprocess.run([EXE_RB, "--waypoint-file", wp, "--landmark-file", lm, "--sim-seed", "31", "--sensor-range-stddev",
                          "1", "--sensor-bearing-stddev", "0.05", "--dump-loader"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    w = oracle_lib.World()
    w.simulate(wp, lm, seed=31, range_stddev=1.0, bearing_stddev=0.05)
    inp, (bm, _) = w.get_input(), w.get_bearings()
    lines = out.stdout.splitlines()
    kv = {ln.split()[0]: ln.split()[1:] for ln in lines if not ln.startswith("obs")}
    assert (int(kv["T"][0]), int(kv["N"][0])) == w.dims()
    rows = [ln.split()[1:] for ln in lines if ln.startswith("obs")]
    assert len(rows) == 10
    for i, t, fid, r, b in rows:
        i = int(i)
        assert (int(t), int(fid)) == (int(inp["obs_t"][i]), int(inp["obs_id"][i]))
        assert (float.fromhex(r), float.fromhex(b)) == (float(inp["obs_mean"][i]), float(bm[i]))
    assert [float.fromhex(v) for v in kv["sum_ctrl_mean"]] == [float(v) for v in inp["ctrl_mean"].sum(axis=0)]
    err = subprocess.run([EXE_RB, "--dataset", "Plaza1", "--dump-loader"], capture_output=True, text=True)
    assert err.returncode == 1 and "range-only" in err.stderr


@pytest.mark.gpu
def test_range_bearing_run_through_the_drop_in_classes():
    """slam::mcmc_slam<velocity_slip_model, range_bearing_model> on the GPU, event by event, against the oracle."""
    import oracle_lib

    wp, lm = os.path.join(EXAMPLE, "waypoints.txt"), os.path.join(EXAMPLE, "landmarks.txt")
    out = subprocess.run([EXE_RB, "--waypoint-file", wp, "--landmark-file", lm, "--mcmc-slam", "--mcmc-steps", "5",
                          "--mcmc-end-steps", "500", "--sensor-range-stddev", "1", "--sensor-bearing-stddev", "0.05",
                          "--seed", "12"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    ll = float([ln for ln in out.stdout.splitlines() if ln.startswith("MCMC-SLAM log likelihood")][0].split()[-1])
    seeds = oracle_lib.seed_sequence(12, 6)
    w = oracle_lib.World()
    w.simulate(wp, lm, seed=int(seeds[0]), range_stddev=1.0, bearing_stddev=0.05)
    w.set_initialiser(int(seeds[1]))
    c = w.add_chain(int(seeds[2]), steps_per_timestep=5, end_steps=500, use_initialiser=True)
    w.replay(w.dims()[0], finish=True)
    ref = w.log_likelihood(c)
    assert abs(ll - ref) <= 1e-9 * abs(ref), (ll, ref)


@pytest.mark.gpu
def test_single_chain_split_mode_gives_the_same_run():
    """--mcmc-slam through the drop-in class with the reading stream of every proposal spread over several
    workgroups (forced on for this small data set): same final log likelihood as the one-workgroup run."""
    args = ["--dataset", "Plaza2", "--dataset-dir", PLAZA, "--mcmc-slam", "--mcmc-steps", "3", "--mcmc-end-steps", "400",
            "--sensor-range-stddev", "1", "--seed", "42"]
    lls = []
    for extra in (["--split-slices", "0"], ["--split-slices", "4", "--split-min-observations", "1"]):
        out = run(*args, *extra)
        lls.append(float([ln for ln in out.splitlines() if ln.startswith("MCMC-SLAM log likelihood")][0].split()[-1]))
    assert abs(lls[0] - lls[1]) <= 1e-9 * abs(lls[0]), lls


@pytest.mark.gpu
def test_multi_mcmc_with_few_chains_in_split_mode_writes_the_same_report(tmp_path):
    """--multi-mcmc with a handful of chains switches to split mode on large data (forced on here): the
    report lines are those of the one-workgroup-per-chain run."""
    reports = []
    for k, extra in enumerate((["--split-slices", "0"], ["--split-slices", "4", "--split-min-observations", "1"])):
        d = tmp_path / str(k)
        d.mkdir()
        out = subprocess.run([EXE, "--dataset", "Plaza2", "--dataset-dir", PLAZA, "--multi-mcmc", "--multi-mcmc-chains", "6",
                              "--multi-mcmc-end-steps", "250", "--sensor-range-stddev", "4", "--seed", "7", *extra],
                             capture_output=True, text=True, timeout=600, cwd=d)
        assert out.returncode == 0, out.stdout + out.stderr
        reports.append([[float(x) for x in ln.split("\t")] for ln in open(d / "multi-mcmc-report.txt").read().splitlines()])
    assert len(reports[0]) == len(reports[1]) == 3
    for a, b in zip(*reports):
        assert a[0] == b[0] and a[1] == b[1]  # updates and accepts per chain
        assert abs(a[2] - b[2]) <= 1e-9 * abs(a[2])  # best log likelihood


def _multi_mcmc_run(tmp_path, tag, *extra):
    d = tmp_path / tag
    d.mkdir()
    out = subprocess.run([EXE, "--dataset", "Plaza2", "--dataset-dir", PLAZA, "--multi-mcmc", "--multi-mcmc-chains", "7",
                          "--multi-mcmc-end-steps", "250", "--sensor-range-stddev", "4", "--seed", "7", *extra],
                         capture_output=True, text=True, timeout=600, cwd=d)
    assert out.returncode == 0, out.stdout + out.stderr
    report = open(d / "multi-mcmc-report.txt").read()
    keep = [ln for ln in out.stdout.splitlines() if ln.startswith(("Multi-MCMC-SLAM log", "Multi-MCMC-SLAM best", "Averaged"))]
    return report, keep, out.stdout


@pytest.mark.gpu
def test_multi_mcmc_through_a_one_rank_rccl_communicator(tmp_path):
    """--exchange rccl: the chain summaries, the best chain's state and the averaging sums go through
    ncclAllGather / ncclBroadcast / ncclAllReduce of a communicator of one rank (all a one-GPU box can hold) --
    the same report, best chain and average as without an exchange."""
    plain = _multi_mcmc_run(tmp_path, "plain")
    rccl = _multi_mcmc_run(tmp_path, "rccl", "--exchange", "rccl")
    assert "exchange through RCCL" in rccl[2] and "exchange through nothing" in plain[2]
    assert rccl[0] == plain[0] and rccl[1] == plain[1]


@pytest.mark.gpu
def test_multi_mcmc_sharded_over_handles_gives_the_same_run(tmp_path):
    """7 chains as 4 + 3 on two handles (both on GPU 0, so the exchange goes through host memory; on GPUs of
    their own the same code path uses RCCL): seeds, chain order, report lines, best chain and average are
    those of one handle with 7 chains (src/slam/multi_mcmc.hpp:105-139, 160-166)."""
    one = _multi_mcmc_run(tmp_path, "one")
    two = _multi_mcmc_run(tmp_path, "two", "--devices", "0,0")
    three = _multi_mcmc_run(tmp_path, "three", "--devices", "0,0,0")
    assert "7 on 2 handle(s), exchange through host memory" in two[2]
    assert "7 on 3 handle(s)" in three[2]
    for sharded in (two, three):
        assert sharded[0] == one[0]
        assert sharded[1][:2] == one[1][:2]  # best log likelihood and best chain's estimate: bit for bit
        # the average adds the per-handle partial sums in another order
        a, b = float(sharded[1][2].split()[-1]), float(one[1][2].split()[-1])
        assert abs(a - b) <= 1e-9 * abs(b)


@pytest.mark.gpu
def test_multi_mcmc_with_8192_chains_on_eight_handles_gives_the_same_run(tmp_path):
    """The sharded configuration's chain count through the C++ host: 8192 chains as 8 handles of 1024 (all on
    GPU 0, exchange through host memory) against one handle of 8192 -- same report lines, same best chain and
    estimate (src/slam/multi_mcmc.hpp:105-139, 160-166); the average differs by the order of the partial sums."""
    def job(tag, *extra):
        d = tmp_path / tag
        d.mkdir()
        out = subprocess.run([EXE, "--dataset", "Plaza2", "--dataset-dir", PLAZA, "--multi-mcmc", "--multi-mcmc-chains",
                              "8192", "--multi-mcmc-end-steps", "200", "--sensor-range-stddev", "4", "--seed", "7", *extra],
                             capture_output=True, text=True, timeout=900, cwd=d)
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
        keep = [ln for ln in out.stdout.splitlines() if ln.startswith(("Multi-MCMC-SLAM log", "Multi-MCMC-SLAM best", "Averaged"))]
        return open(d / "multi-mcmc-report.txt").read(), keep, out.stdout

    one = job("one")
    eight = job("eight", "--devices", "0,0,0,0,0,0,0,0", "--exchange", "host")
    assert "8192 on 8 handle(s), exchange through host memory" in eight[2]
    assert len(one[0].splitlines()) == 2
    assert eight[0] == one[0]
    assert eight[1][:2] == one[1][:2]
    a, b = float(eight[1][2].split()[-1]), float(one[1][2].split()[-1])
    assert abs(a - b) <= 1e-9 * abs(b)
