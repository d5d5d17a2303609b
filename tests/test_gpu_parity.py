"""GPU parity: the HIP path through the C ABI against the CPU oracle on the same
seeded inputs.  Bars: edge kinds, edge indices and accept bits identical; chain
log likelihoods and cumulative poses within 1e-9 relative (fp64)."""
import numpy as np
import pytest
import torch

import oracle_lib
from slam_amd import capi, synthetic

pytestmark = pytest.mark.gpu

# torch brings a HIP runtime of its own; a test below hands the library a torch tensor, so torch's runtime
# has to be the one this process initialises first (also when this file is run on its own)
torch.cuda.is_available()

REL = 1e-9  # the tolerance the task states for log likelihoods and poses


def small_problem(T=400, F=16, k=4, seed=3, range_stddev=2.0):
    return synthetic.make_problem(T=T, n_landmarks=F, k_nearest=k, seed=seed, lap_steps=150, wobble_period=500,
                                  range_stddev=range_stddev, landmark_spread=6.0)


def oracle_world(p, seeds, trace=True, init_seed=None, steps_per_timestep=0, end_steps=0):
    w = oracle_lib.World()
    w.set_input(p["ctrl_mean"], p["ctrl_std"], p["obs_t"], p["obs_f"], p["obs_mean"], p["obs_std"])
    if init_seed is not None:
        w.set_initialiser(init_seed)
    for s in seeds:
        w.add_chain(int(s), steps_per_timestep=steps_per_timestep, end_steps=end_steps,
                    use_initialiser=init_seed is not None, trace=trace)
    return w


def gpu_handle(p, seeds, trace_capacity, init_uniforms=None, **kw):
    g = capi.McmcGpu(len(seeds), p["T"], p["F"], p["N"], trace_capacity=trace_capacity, **kw)
    g.upload_problem(p["ctrl_mean"], p["ctrl_std"], p["obs_t"], p["obs_f"], p["obs_mean"], p["obs_std"], p["F"])
    g.init_chains(seeds, init_uniforms)
    return g


def assert_close(a, b, rel, what, floor=1.0):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    scale = np.maximum(np.maximum(np.abs(a), np.abs(b)), floor)
    err = np.abs(a - b) / scale
    assert np.all(err <= rel), f"{what}: max rel err {err.max():.3e} at {np.argmax(err)}"


# A log ratio is the signed sum of log densities: the control model's at the proposed and at the current
# relative pose, and every affected reading's at the moved and at the current landmark position.  Two
# computations of it (flat pose table here; O(log T) tree walks and a relative pose reconstructed from tree
# nodes in the oracle, src/utility/bitree.hpp:256-264) agree on the scale of those terms, not on the scale
# of their difference.  The oracle records that scale per step (term_scale = sum of the terms' sizes); the
# bar is REL times it, with a floor of 1.  On the Plaza logs term_scale is a few thousand where the chain's
# log likelihood is 10^5, so this is some 30 times tighter than REL x |log likelihood|.


def log_ratio_errors(tg, to):
    """Largest |difference| of two traces' log ratios, and the largest in units of max(1, term_scale)."""
    a, b = np.asarray(tg["log_ratio"], dtype=np.float64), np.asarray(to["log_ratio"], dtype=np.float64)
    if a.size == 0:
        return 0.0, 0.0
    err = np.abs(a - b)
    return float(err.max()), float((err / np.maximum(to["term_scale"], 1.0)).max())


def compare_traces(tg, to, ll_scale=None):
    """tg: the device's trace, to: the oracle's (ll_scale: unused, kept for callers that still pass it)."""
    assert len(tg) == len(to), f"trace lengths differ: gpu {len(tg)} oracle {len(to)}"
    assert np.array_equal(tg["kind"], to["kind"]), "edge kinds differ"
    assert np.array_equal(tg["index"], to["index"]), "edge indices differ"
    assert np.array_equal(tg["accepted"], to["accepted"]), "accept decisions differ"
    assert_close(tg["proposed"], to["proposed"], 1e-11, "proposals")
    a, b = np.asarray(tg["log_ratio"], dtype=np.float64), np.asarray(to["log_ratio"], dtype=np.float64)
    bad = np.abs(a - b) > REL * np.maximum(to["term_scale"], 1.0)
    assert not bad.any(), (f"log ratios: {int(bad.sum())} of {a.size} differ by more than {REL} x max(1, sum of |terms|); "
                           f"worst |difference| {np.abs(a - b).max():.3e} where the terms sum to {to['term_scale'][np.argmax(np.abs(a - b))]:.3e}")
    assert_close(tg["log_likelihood"], to["log_likelihood"], REL, "running log likelihood")


def test_scan_affine_group_exact():
    """The scan template on the reference tree test's non-commutative integer group."""
    rng = np.random.default_rng(1)
    P = 32749
    for n in (1, 63, 64, 65, 255, 256, 257, 2000, 10000):
        a = rng.integers(0, P, size=n).astype(np.int32)
        b = rng.integers(1, P, size=n).astype(np.int32)
        oa, ob = capi.selftest_scan(a, b, P)
        ra, rb = np.empty(n, dtype=np.int64), np.empty(n, dtype=np.int64)
        ca, cb = 0, 1
        for i in range(n):
            ca, cb = (ca + cb * int(a[i])) % P, (cb * int(b[i])) % P
            ra[i], rb[i] = ca, cb
        assert np.array_equal(oa, ra) and np.array_equal(ob, rb), f"n={n}"


def test_seed_helpers_match_oracle():
    assert np.array_equal(capi.seed_sequence(99, 50), oracle_lib.seed_sequence(99, 50))
    assert np.array_equal(capi.uniform_sequence(7, 50), oracle_lib.uniform_sequence(7, 50))


@pytest.mark.parametrize("uniform_first,serial_program", [(True, 1), (False, 1), (True, 2), (False, 2), (True, 3), (False, 3)])
def test_grow_then_run_matches_oracle(uniform_first, serial_program):
    """multi_mcmc's way: chains grow over the whole data set without updates, then step.  Every build of the
    serial part (mcmcgpu_config.serial_program: a called function, compiled into the kernel, compiled in with
    the first wavefront selecting the next update ahead of time)."""
    p = small_problem()
    seeds = capi.seed_sequence(2024, 4)
    steps = 600
    w = oracle_lib.World()
    w.set_input(p["ctrl_mean"], p["ctrl_std"], p["obs_t"], p["obs_f"], p["obs_mean"], p["obs_std"])
    for s in seeds:
        w.add_chain(int(s), uniform_first=uniform_first, trace=True)
    w.replay(p["T"])
    g = gpu_handle(p, seeds, trace_capacity=4096, feature_uniform_first=uniform_first, serial_program=serial_program)
    g.grow(p["T"])
    g.sync()
    ll_grow = g.log_likelihood()
    for c in range(len(seeds)):
        assert_close(ll_grow[c], w.log_likelihood(c), REL, f"chain {c} log likelihood after growth")
    for c in range(len(seeds)):
        assert w.update(c, steps) >= 0
    g.run(steps)
    g.sync()
    ll = g.log_likelihood()
    full = g.full_log_likelihood()
    upd, acc, units = g.counters()
    for c in range(len(seeds)):
        to, tg = w.trace(c), g.trace(c)
        compare_traces(tg, to, abs(w.log_likelihood(c)))
        assert_close(ll[c], w.log_likelihood(c), REL, f"chain {c} final log likelihood")
        assert_close(full[c], w.full_log_likelihood(c), REL, f"chain {c} full log likelihood")
        assert_close(ll[c], full[c], 1e-8, f"chain {c} incremental vs full")
        so, sg = w.state(c), g.get_state(c)
        assert np.array_equal(sg["feat_parent"], so["feat_parent"])
        assert_close(sg["cum"], so["cum"], REL, "cumulative poses")
        assert_close(sg["rel"], so["rel"], REL, "relative poses")
        assert_close(sg["feat_est"], so["feat_est"], REL, "landmark estimates")
        assert_close(sg["state_w"], so["state_w"], 1e-9, "state edge weights", floor=1e-300)
        assert_close(sg["feat_w"], so["feat_w"], 1e-9, "landmark edge weights", floor=1e-300)
        assert units[c] == w.units(c), "evaluated-observation counts differ"
        assert upd[c] == steps


@pytest.mark.parametrize("serial_program", [1, 2, 3])
def test_online_with_initialiser_matches_oracle(serial_program):
    """--mcmc-slam's way: initialiser, updates after every timestep, then end steps."""
    p = small_problem(T=300, F=10, k=3, seed=11)
    seeds = capi.seed_sequence(5, 6)  # module seeds, src/main.cpp:133-147
    init_seed, chain_seed = int(seeds[1]), int(seeds[2])
    w = oracle_world(p, [chain_seed], init_seed=init_seed, steps_per_timestep=5, end_steps=200)
    w.replay(p["T"], finish=True)
    g = gpu_handle(p, np.array([chain_seed], dtype=np.uint32), trace_capacity=4096,
                   init_uniforms=capi.uniform_sequence(init_seed, p["F"]), serial_program=serial_program)
    g.advance(p["T"], steps_per_timestep=5, end_steps=200)
    g.sync()
    compare_traces(g.trace(0), w.trace(0), abs(w.log_likelihood(0)))
    assert_close(g.log_likelihood()[0], w.log_likelihood(0), REL, "final log likelihood")
    assert_close(g.get_state(0)["cum"], w.state(0)["cum"], REL, "cumulative poses")


@pytest.mark.parametrize("lanes,landmarks,k", [(128, 12, 4), (128, 90, 6), (256, 90, 6), (256, 300, 12), (128, 300, 12)])
def test_selection_ahead_of_time_matches_oracle(lanes, landmarks, k):
    """serial_program = 3: the first wavefront draws the accept test's uniform and selects and proposes the next
    update while the other wavefronts evaluate the current one; an accepted update drops what was drawn ahead.
    Step for step against the oracle with one and with three evaluating wavefronts, with a wavefront's worth
    of landmarks (set-up inside one wavefront), more (set-up by all, meeting at the evaluating wavefronts' own
    barrier) and more than a round of segments holds -- online (growth and re-parenting between the updates)
    and with end steps; the chains accept between a tenth and a third of their proposals, so both outcomes of
    the work ahead occur hundreds of times."""
    p = synthetic.make_problem(T=260, n_landmarks=landmarks, k_nearest=k, seed=17 + landmarks, lap_steps=130,
                               wobble_period=400, range_stddev=2.5, landmark_spread=7.0)
    seeds = capi.seed_sequence(31 + lanes, 3)
    w = oracle_world(p, [int(x) for x in seeds], steps_per_timestep=4, end_steps=300)
    w.replay(p["T"], finish=True)
    g = gpu_handle(p, seeds, trace_capacity=4096, threads_per_chain=lanes, serial_program=3)
    g.advance(p["T"], steps_per_timestep=4, end_steps=300)
    g.sync()  # (raises when a chain has stopped with its error flag set)
    upd, acc, units = g.counters()
    for c in range(len(seeds)):
        compare_traces(g.trace(c), w.trace(c))
        assert_close(g.log_likelihood()[c], w.log_likelihood(c), REL, f"chain {c} final log likelihood")
        assert_close(g.get_state(c)["cum"], w.state(c)["cum"], REL, "cumulative poses")
        assert units[c] == w.units(c), "evaluated-observation counts differ"
        assert 0.05 * upd[c] < acc[c] < 0.6 * upd[c], f"chain {c}: {acc[c]} of {upd[c]} accepted"
    full = g.full_log_likelihood()
    assert_close(g.log_likelihood(), full, 1e-8, "incremental vs from-scratch log likelihood")


@pytest.mark.parametrize("uniform_sigma", [True, False])
def test_one_wavefront_per_chain_with_readings_fetched_ahead(uniform_sigma):
    """64 lanes and the serial program compiled in -- what many chains on sparse data get by default: the
    wavefront that draws a landmark proposal asks for the landmark's readings and their poses while it draws
    (lists of at most 64; here 15 on average, growing from one as the chain grows) and evaluates them in
    registers.  Online, step for step against the oracle, with one sigma for all readings and with a sigma per
    reading (the per-reading factor is then fetched ahead as well)."""
    p = synthetic.make_problem(T=300, n_landmarks=60, k_nearest=3, seed=23, lap_steps=140, wobble_period=450,
                               range_stddev=2.0, landmark_spread=7.0)
    if not uniform_sigma:
        p = dict(p)
        p["obs_std"] = p["obs_std"] * (1.0 + 0.3 * np.random.default_rng(4).random(p["N"]))
    seeds = capi.seed_sequence(77, 3)
    w = oracle_world(p, [int(x) for x in seeds], steps_per_timestep=3, end_steps=400)
    w.replay(p["T"], finish=True)
    g = gpu_handle(p, seeds, trace_capacity=4096, threads_per_chain=64, serial_program=2)
    g.advance(p["T"], steps_per_timestep=3, end_steps=400)
    g.sync()
    upd, acc, units = g.counters()
    for c in range(len(seeds)):
        compare_traces(g.trace(c), w.trace(c))
        assert_close(g.log_likelihood()[c], w.log_likelihood(c), REL, f"chain {c} final log likelihood")
        assert units[c] == w.units(c), "evaluated-observation counts differ"
    assert_close(g.log_likelihood(), g.full_log_likelihood(), 1e-8, "incremental vs from-scratch log likelihood")


def test_online_in_pieces_equals_one_call():
    """Growing timestep by timestep with re-uploads of the growing problem gives the same chain."""
    p = small_problem(T=120, F=8, k=3, seed=5)
    seeds = np.array([77], dtype=np.uint32)
    g1 = gpu_handle(p, seeds, trace_capacity=2048)
    g1.advance(p["T"], steps_per_timestep=3, end_steps=50)
    g1.sync()
    g2 = capi.McmcGpu(1, p["T"], p["F"], p["N"], trace_capacity=2048)
    g2.upload_problem(p["ctrl_mean"][:0], p["ctrl_std"][:0], p["obs_t"][:0], p["obs_f"][:0], p["obs_mean"][:0],
                      p["obs_std"][:0], 0)
    g2.init_chains(seeds)
    for t in range(0, p["T"] + 1):
        n = int(np.searchsorted(p["obs_t"], t, side="right"))
        F = int(p["obs_f"][:n].max()) + 1 if n else 0
        g2.upload_problem(p["ctrl_mean"][:t], p["ctrl_std"][:t], p["obs_t"][:n], p["obs_f"][:n], p["obs_mean"][:n],
                          p["obs_std"][:n], F)
        g2.advance(t, steps_per_timestep=3, end_steps=50 if t == p["T"] else 0)
    g2.sync()
    t1, t2 = g1.trace(0), g2.trace(0)
    assert len(t1) == len(t2)
    for name in ("kind", "index", "accepted", "log_ratio", "log_likelihood"):
        assert np.array_equal(t1[name], t2[name]), name


def test_eval_and_commit_entry_points():
    """edge_log_likelihood_ratio for recorded proposals, and the accept branch applied from outside."""
    p = small_problem(T=250, F=10, k=4, seed=21)
    seeds = np.array([4242], dtype=np.uint32)
    w = oracle_world(p, seeds)
    w.replay(p["T"])
    g = gpu_handle(p, seeds, trace_capacity=0)
    g.grow(p["T"])
    rng = np.random.default_rng(0)
    so = w.state(0)
    for it in range(60):
        if it % 3 == 2:
            f = int(rng.integers(0, p["F"]))
            q = so["feat_est"][f] + rng.normal(0, 0.5, 2)
            a, b = g.eval_llr(0, capi.FEATURE_EDGE, f, q), w.eval(0, 1, f, q)
        else:
            t = int(rng.integers(0, p["T"]))
            q = so["rel"][t] + rng.normal(0, [0.02, 0.02, 0.005])
            a, b = g.eval_llr(0, capi.STATE_EDGE, t, q), w.eval(0, 0, t, q)
        assert_close(a, b, REL, f"log ratio #{it}", floor=abs(w.log_likelihood(0)))
    # commit a state edge and a landmark edge by hand; the incremental likelihood must track the full one
    t, q = 100, so["rel"][100] + np.array([0.01, -0.02, 0.003])
    llr = g.eval_llr(0, capi.STATE_EDGE, t, q)
    g.commit(0, capi.STATE_EDGE, t, q, 1.25, llr)
    f, qf = 3, so["feat_est"][3] + np.array([0.3, -0.2])
    llr = g.eval_llr(0, capi.FEATURE_EDGE, f, qf)
    g.commit(0, capi.FEATURE_EDGE, f, qf, 0.75, llr)
    g.sync()
    st = g.get_state(0)
    assert np.allclose(st["rel"][t], q, rtol=0, atol=0)
    assert np.allclose(st["feat_est"][f], qf, rtol=0, atol=0)
    assert_close(st["state_w"][t], 1.25, 1e-12, "committed state weight")
    assert_close(st["feat_w"][f], 0.75, 1e-12, "committed landmark weight")
    assert_close(g.log_likelihood()[0], g.full_log_likelihood()[0], 1e-9, "incremental vs full after commits")
    # ... and its split into control part and readings' part (what tempering weighs) tracks a from-scratch one
    fresh = gpu_handle(p, seeds, trace_capacity=0)
    fresh.load_state(st["rel"], st["feat_world"])
    fresh.sync()
    assert_close(g.ladders()[2][0], fresh.ladders()[2][0], 1e-9, "readings' part of the log likelihood after commits")
    # cumulative poses after the commit equal a from-scratch rebuild
    before = st["cum"].copy()
    g.rebuild()
    assert_close(g.get_state(0)["cum"], before, 1e-11, "cumulative poses vs rebuild")


def test_rebuild_scan_matches_oracle_accumulate():
    p = small_problem(T=700, F=12, k=3, seed=8)
    seeds = np.array([1], dtype=np.uint32)
    w = oracle_world(p, seeds, trace=False)
    w.replay(p["T"])
    g = gpu_handle(p, seeds, trace_capacity=0)
    g.grow(p["T"])
    grown = g.get_state(0)["cum"].copy()
    g.rebuild()
    assert_close(g.get_state(0)["cum"], w.state(0)["cum"], REL, "scan vs bitree accumulate")
    assert_close(g.get_state(0)["cum"], grown, 1e-11, "scan vs poses appended one by one")


@pytest.mark.parametrize("T,chains", [(20000, 2), (9000, 3), (5000, 1), (260000, 2), (400000, 1)])
def test_rebuild_scan_in_segments_matches_oracle_accumulate(T, chains):
    """Few chains on a long trajectory: the rebuild cuts every chain into pieces, one workgroup each, which pass
    on what lies before them (the two longest cases: more than 64 pieces per chain, so the prefix is handed on
    from window to window and tickets run across the chains); against bitree<pose>::accumulate of the oracle,
    every chain its own trajectory."""
    p = synthetic.make_problem(T=T, n_landmarks=12, k_nearest=3, seed=T, lap_steps=400, wobble_period=900,
                               range_stddev=1.0, landmark_spread=8.0)
    seeds = capi.seed_sequence(T, chains)
    rng = np.random.default_rng(T)
    w = oracle_lib.World()
    w.set_input(p["ctrl_mean"], p["ctrl_std"], p["obs_t"], p["obs_f"], p["obs_mean"], p["obs_std"])
    w.replay(p["T"])
    g = gpu_handle(p, seeds, trace_capacity=0)
    g.load_state(p["true_rel"], p["landmarks"])
    for c in range(chains):
        rel = p["true_rel"] + rng.normal(0.0, [1e-3, 1e-3, 1e-4], size=p["true_rel"].shape)
        k = w.add_detached_chain(int(seeds[c]))
        w.load_state(k, rel, np.arange(p["F"]), p["landmarks"])
        g.load_state(rel, p["landmarks"], first=c, count=1)
    g.rebuild()
    g.sync()
    for c in range(chains):
        assert_close(g.get_state(c)["cum"], w.state(c)["cum"], REL, f"chain {c}: segmented scan vs bitree accumulate")


@pytest.mark.parametrize("threads", [64, 128, 512, 1024])
def test_other_workgroup_sizes(threads):
    p = small_problem(T=200, F=9, k=3, seed=13)
    seeds = capi.seed_sequence(1, 2)
    w = oracle_world(p, seeds)
    w.replay(p["T"])
    g = gpu_handle(p, seeds, trace_capacity=1024, threads_per_chain=threads)
    g.grow(p["T"])
    g.run(300)
    g.sync()
    for c in range(2):
        w.update(c, 300)
        compare_traces(g.trace(c), w.trace(c), abs(w.log_likelihood(c)))


@pytest.mark.parametrize("name,serial_program", [("Plaza2", 1), ("Plaza2", 0), ("Plaza1", 0)])
def test_plaza_single_chain(plaza_dir, name, serial_program):
    """Plaza1 / Plaza2, one chain on the GPU, step for step: --mcmc-slam --mcmc-steps 10 --mcmc-end-steps 1000 --sensor-range-stddev 1."""
    w = oracle_lib.World()
    w.load_plaza(plaza_dir, name, 1.0)
    inp = w.get_input()
    obs_f, order = oracle_lib.dense_features(inp["obs_id"])
    T, N = w.dims()
    seeds = capi.seed_sequence(42, 6)
    init_seed, chain_seed = int(seeds[1]), int(seeds[2])
    w.set_initialiser(init_seed)
    w.add_chain(chain_seed, steps_per_timestep=10, end_steps=1000, use_initialiser=True, trace=True)
    w.replay(T, finish=True)
    g = capi.McmcGpu(1, T, len(order), N, trace_capacity=T * 10 + 1000 + 4 * N, serial_program=serial_program)
    g.upload_problem(inp["ctrl_mean"], inp["ctrl_std"], inp["obs_t"], obs_f, inp["obs_mean"], inp["obs_std"],
                     len(order))
    g.init_chains(np.array([chain_seed], dtype=np.uint32), capi.uniform_sequence(init_seed, len(order)))
    g.advance(T, steps_per_timestep=10, end_steps=1000)
    g.sync()
    compare_traces(g.trace(0), w.trace(0))
    # how close the run came to a different decision: the smallest relative distance between the two sides
    # of the accept test (src/slam/mcmc_slam.hpp:408) over all steps, on the oracle's side, against the
    # largest difference of the two implementations' log ratios (which shifts exp(accept_log_ratio) by
    # that relative amount).  A flip needs the second to reach the first.
    margin = w.min_accept_margin(0)
    worst_abs, worst_rel = log_ratio_errors(g.trace(0), w.trace(0))
    print(f"{name}: smallest relative accept margin {margin:.3e} over {len(w.trace(0))} steps; "
          f"largest log-ratio difference {worst_abs:.3e} absolute, {worst_rel:.3e} of the terms' sum")
    assert margin > 0 and worst_abs * 10 < margin, (margin, worst_abs)
    assert_close(g.log_likelihood()[0], w.log_likelihood(0), REL, f"{name} final log likelihood")
    assert_close(g.full_log_likelihood()[0], w.full_log_likelihood(0), REL, f"{name} full log likelihood")
    assert_close(g.get_state(0)["cum"], w.state(0)["cum"], REL, f"{name} cumulative poses")


def test_load_state_then_run_matches_oracle():
    """Chains started from a given state (the benchmark's way) step exactly like the oracle's."""
    p = synthetic.make_problem(T=3000, n_landmarks=60, k_nearest=5, seed=17, lap_steps=400, wobble_period=900,
                               range_stddev=1.0, landmark_spread=8.0)
    seeds = capi.seed_sequence(31, 3)
    steps = 400
    w = oracle_lib.World()
    w.set_input(p["ctrl_mean"], p["ctrl_std"], p["obs_t"], p["obs_f"], p["obs_mean"], p["obs_std"])
    w.replay(p["T"])
    for s in seeds:
        c = w.add_detached_chain(int(s), trace=True)
        w.load_state(c, p["true_rel"], np.arange(p["F"]), p["landmarks"])
    g = gpu_handle(p, seeds, trace_capacity=steps)
    g.load_state(p["true_rel"], p["landmarks"])
    ll0 = g.log_likelihood()
    for c in range(len(seeds)):
        assert_close(ll0[c], w.log_likelihood(c), REL, "log likelihood of the loaded state")
        so, sg = w.state(c), g.get_state(c)
        assert np.array_equal(sg["feat_parent"], so["feat_parent"])
        assert_close(sg["cum"], so["cum"], REL, "cumulative poses of the loaded state")
        assert_close(sg["feat_est"], so["feat_est"], REL, "landmark estimates of the loaded state")
        assert_close(sg["state_w"], so["state_w"], 1e-9, "state weights", floor=1e-300)
        assert_close(sg["feat_w"], so["feat_w"], 1e-9, "landmark weights", floor=1e-300)
    g.run(steps)
    g.sync()
    ll = g.log_likelihood()
    _, _, units = g.counters()
    for c in range(len(seeds)):
        assert w.update(c, steps) >= 0
        compare_traces(g.trace(c), w.trace(c), abs(w.log_likelihood(c)))
        assert_close(ll[c], w.log_likelihood(c), REL, "final log likelihood")
        assert units[c] == w.units(c)
    # drift control: a rebuild changes nothing beyond rounding
    before = g.get_state(0)["cum"].copy()
    g.rebuild()
    assert_close(g.get_state(0)["cum"], before, 1e-11, "rebuild after many commits")


def _summaries_to_numpy(t):
    return t.cpu().numpy().view(capi.CHAIN_SUMMARY).reshape(t.shape[:-1])


def test_one_reported_launch_equals_many_short_ones():
    """multi_mcmc::completed reports every 100 updates; one long launch leaves the same report records."""
    import torch

    p = small_problem(T=300, F=10, k=3, seed=23)
    seeds = capi.seed_sequence(3, 5)
    C, steps, every = len(seeds), 250, 100
    a = gpu_handle(p, seeds, trace_capacity=0)
    a.grow(p["T"])
    short = []
    buf = torch.zeros((C, 2), dtype=torch.float64, device="cuda")
    for n in (100, 100, 50):
        a.run(n)
        a.write_summaries(buf.data_ptr())
        a.sync()
        short.append(_summaries_to_numpy(buf).copy())
    b = gpu_handle(p, seeds, trace_capacity=0)
    b.grow(p["T"])
    out = torch.zeros((3, C, 2), dtype=torch.float64, device="cuda")
    b.run_reported(steps, every, out.data_ptr())
    b.sync()
    long = _summaries_to_numpy(out)
    for k in range(3):
        for name in ("log_likelihood", "updates", "accepts"):
            assert np.array_equal(short[k][name], long[k][name]), (k, name)
    assert list(long[2]["updates"]) == [steps] * C
    assert np.array_equal(a.log_likelihood(), b.log_likelihood())


def test_export_import_moves_a_chain():
    """The best chain's state travels as a packed device buffer (what gets broadcast between GPUs)."""
    import torch

    p = small_problem(T=260, F=9, k=3, seed=29)
    seeds = capi.seed_sequence(11, 2)
    g = gpu_handle(p, seeds, trace_capacity=0)
    g.grow(p["T"])
    g.run(400)
    g.sync()
    ll = g.log_likelihood()
    packed = torch.zeros(3 * p["T"] + 2 * p["F"], dtype=torch.float64, device="cuda")
    g.export_state(0, packed.data_ptr())
    g.import_state(1, packed.data_ptr(), float(ll[0]))
    g.sync()
    s0, s1 = g.get_state(0), g.get_state(1)
    assert np.array_equal(s0["rel"], s1["rel"]) and np.array_equal(s0["feat_est"], s1["feat_est"])
    assert_close(s1["cum"], s0["cum"], 1e-12, "cumulative poses of the imported chain")
    assert_close(s1["state_w"], s0["state_w"], 1e-12, "state weights of the imported chain", floor=1e-300)
    assert_close(s1["feat_w"], s0["feat_w"], 1e-12, "landmark weights of the imported chain", floor=1e-300)
    assert g.log_likelihood()[1] == ll[0]
    assert_close(g.full_log_likelihood()[1], ll[0], 1e-9, "imported chain: incremental vs full log likelihood")
    rll = g.ladders()[2]  # the readings' part, which tempering weighs, came along too
    assert_close(rll[1], rll[0], 1e-9, "imported chain: readings' part of the log likelihood")
    g.run(100)  # both keep stepping (with their own generators)
    g.sync()


def test_average_over_chains_matches_oracle():
    """multi_mcmc::get_average: on one handle, and split over two handles the way two GPUs would."""
    import torch

    p = small_problem(T=250, F=9, k=3, seed=21)
    seeds = capi.seed_sequence(31, 6)
    steps = 300
    w = oracle_world(p, seeds, trace=False)
    w.replay(p["T"])
    for c in range(len(seeds)):
        assert w.update(c, steps) >= 0
    acc_o, feat_o = w.average(p["T"], p["F"])
    g = gpu_handle(p, seeds, trace_capacity=0)
    g.grow(p["T"])
    g.run(steps)
    acc_g, feat_g = g.average()
    assert_close(acc_g, acc_o, REL, "averaged accumulated poses")
    assert_close(feat_g, feat_o, REL, "averaged landmarks")
    # chains 0-2 on one handle, 3-5 on another: base from the first, sums added, finished on either
    n = 3 * p["T"] + 2 * p["F"]
    halves = []
    for part in (seeds[:3], seeds[3:]):
        h = gpu_handle(p, part, trace_capacity=0)
        h.grow(p["T"])
        h.run(steps)
        h.sync()
        halves.append(h)
    base = torch.zeros(n, dtype=torch.float64, device="cuda")
    sums = [torch.zeros(n, dtype=torch.float64, device="cuda") for _ in halves]
    torch.cuda.synchronize()
    halves[0].average_base(0, base.data_ptr())
    halves[0].sync()
    halves[0].average_sum(base.data_ptr(), 0, sums[0].data_ptr())
    halves[1].average_sum(base.data_ptr(), -1, sums[1].data_ptr())
    halves[0].sync()
    halves[1].sync()
    total = sums[0] + sums[1]
    torch.cuda.synchronize()
    halves[1].average_finish(base.data_ptr(), total.data_ptr(), len(seeds))
    halves[1].sync()
    out = base.cpu().numpy()
    assert_close(out[:3 * p["T"]].reshape(-1, 3), acc_g, 1e-12, "split average, poses")
    assert_close(out[3 * p["T"]:].reshape(-1, 2), feat_g, 1e-12, "split average, landmarks")


def _rb_problem(**kw):
    p = small_problem(**kw)
    q = synthetic.make_problem(T=p["T"], n_landmarks=p["F"], k_nearest=kw.get("k", 4), seed=kw.get("seed", 3),
                               lap_steps=150, wobble_period=500, range_stddev=kw.get("range_stddev", 2.0),
                               landmark_spread=6.0, bearing_stddev=0.05)
    return q


def test_range_bearing_model_matches_oracle():
    """The reference's other observation model (range_bearing_model): growth, landmark proposals from two
    normals, wrapped bearing residuals in the ratio and in the from-scratch likelihood."""
    p = _rb_problem()
    seeds = capi.seed_sequence(909, 3)
    steps = 500
    w = oracle_lib.World()
    w.set_input(p["ctrl_mean"], p["ctrl_std"], p["obs_t"], p["obs_f"], p["obs_mean"], p["obs_std"])
    w.set_bearings(p["obs_bearing"], p["obs_bearing_std"])
    for s in seeds:
        w.add_chain(int(s), trace=True)
    w.replay(p["T"])
    g = capi.McmcGpu(len(seeds), p["T"], p["F"], p["N"], trace_capacity=4096, range_bearing=True)
    g.upload_problem(p["ctrl_mean"], p["ctrl_std"], p["obs_t"], p["obs_f"], p["obs_mean"], p["obs_std"], p["F"])
    with pytest.raises(RuntimeError, match="mcmcgpu_set_bearings"):
        g.init_chains(seeds)
        g.grow(p["T"])
    g.set_bearings(p["obs_bearing"], p["obs_bearing_std"])
    g.init_chains(seeds)
    g.grow(p["T"])
    g.sync()
    ll_grow = g.log_likelihood()
    for c in range(len(seeds)):
        assert_close(ll_grow[c], w.log_likelihood(c), REL, f"chain {c} log likelihood after growth")
    for c in range(len(seeds)):
        assert w.update(c, steps) >= 0
    g.run(steps)
    g.sync()
    ll, full = g.log_likelihood(), g.full_log_likelihood()
    for c in range(len(seeds)):
        compare_traces(g.trace(c), w.trace(c), abs(w.log_likelihood(c)))
        assert_close(ll[c], w.log_likelihood(c), REL, f"chain {c} final log likelihood")
        assert_close(full[c], w.full_log_likelihood(c), REL, f"chain {c} full log likelihood")
        so, sg = w.state(c), g.get_state(c)
        assert np.array_equal(sg["feat_parent"], so["feat_parent"])
        assert_close(sg["cum"], so["cum"], REL, "cumulative poses")
        assert_close(sg["feat_est"], so["feat_est"], REL, "landmark estimates")
        assert g.counters()[2][c] == w.units(c)
    # a range-only handle refuses bearings
    g2 = gpu_handle(p, seeds, trace_capacity=0)
    with pytest.raises(RuntimeError, match="range-only"):
        g2.set_bearings(p["obs_bearing"], p["obs_bearing_std"])


def test_range_bearing_online_with_updates_matches_oracle():
    """--mcmc-slam's way with bearings: 5 updates after every timestep (re-parenting updates included)."""
    p = _rb_problem(T=250, F=10, k=3, seed=13)
    seeds = capi.seed_sequence(4711, 2)
    w = oracle_lib.World()
    w.set_input(p["ctrl_mean"], p["ctrl_std"], p["obs_t"], p["obs_f"], p["obs_mean"], p["obs_std"])
    w.set_bearings(p["obs_bearing"], p["obs_bearing_std"])
    for s in seeds:
        w.add_chain(int(s), steps_per_timestep=5, end_steps=200, trace=True)
    w.replay(p["T"], finish=True)
    g = capi.McmcGpu(len(seeds), p["T"], p["F"], p["N"], trace_capacity=4096, range_bearing=True)
    g.upload_problem(p["ctrl_mean"], p["ctrl_std"], p["obs_t"], p["obs_f"], p["obs_mean"], p["obs_std"], p["F"])
    g.set_bearings(p["obs_bearing"], p["obs_bearing_std"])
    g.init_chains(seeds)
    g.advance(p["T"], steps_per_timestep=5, end_steps=200)
    g.sync()
    ll = g.log_likelihood()
    for c in range(len(seeds)):
        compare_traces(g.trace(c), w.trace(c), abs(w.log_likelihood(c)))
        assert_close(ll[c], w.log_likelihood(c), REL, f"chain {c} final log likelihood")


def test_device_range_is_accurate():
    """|(dx, dy)| of the reading stream (reciprocal square root + fused refinement) against numpy."""
    rng = np.random.default_rng(8)
    n = 200_000
    mag = 10.0 ** rng.uniform(-6, 5, size=n)
    ang = rng.uniform(0, 2 * np.pi, size=n)
    dx, dy = mag * np.cos(ang), mag * np.sin(ang)
    dx[:4], dy[:4] = [0.0, 3.0, 0.0, 1e-150], [0.0, 4.0, -2.5, 0.0]
    got = capi.selftest_range(dx, dy)
    ref = np.hypot(dx, dy)
    assert got[0] < 1e-139 and got[1] == 5.0 and got[2] == 2.5  # a zero distance stays finite
    rel = np.abs(got[4:] - ref[4:]) / ref[4:]
    print("range_of max relative error: %.3e (ulp = 1.1e-16)" % rel.max())
    assert rel.max() < 3.5e-16


def test_snapshot_is_the_state_at_its_place_in_the_stream():
    """snapshot_begin / snapshot_wait: the estimate as of the point where the snapshot was queued, although more
    updates were queued behind it before it was collected (the hand-off loop of the reference's main.cpp:331-337)."""
    p = small_problem()
    seeds = capi.seed_sequence(61, 2)
    a, b = gpu_handle(p, seeds, trace_capacity=0), gpu_handle(p, seeds, trace_capacity=0)
    for g in (a, b):
        g.grow(p["T"])
        g.run(300)
    a.snapshot_begin(1)
    a.run(300)  # queued behind the snapshot
    with pytest.raises(RuntimeError, match="has not been collected"):
        a.snapshot_begin(0)
    rel, world, ll = a.snapshot_wait(p["T"], p["F"])
    b.sync()
    sb = b.get_state(1)
    assert np.array_equal(rel, sb["rel"]) and np.array_equal(world, sb["feat_world"])
    assert ll == b.log_likelihood()[1]
    # the updates queued behind it did run
    a.sync()
    b.run(300)
    b.sync()
    assert np.array_equal(a.log_likelihood(), b.log_likelihood())
    with pytest.raises(RuntimeError, match="no snapshot"):
        a.snapshot_wait(p["T"], p["F"])


def _ragged_problem(seed, T=120, F=7):
    """Hand-made awkward data: timesteps without readings, a landmark seen exactly once, readings at
    timestep 0 and at the last timestep, ranges that shrink and grow (so re-parenting happens both ways)."""
    rng = np.random.default_rng(seed)
    base = small_problem(T=T, F=F, k=3, seed=seed)
    keep = np.ones(base["N"], dtype=bool)
    t = base["obs_t"]
    keep &= ~((t % 7 == 3) | (t % 11 == 5))          # whole timesteps without any reading
    keep &= rng.random(base["N"]) > 0.35             # ragged per-landmark lists
    once = int(base["obs_f"][np.flatnonzero(keep)[-1]])
    last_of_once = np.flatnonzero(keep & (base["obs_f"] == once))
    keep[last_of_once[:-1]] = False                  # this landmark is seen exactly once, at the very end
    idx = np.flatnonzero(keep)
    obs_raw = base["obs_f"][idx]
    _, first_pos = np.unique(obs_raw, return_index=True)  # renumber densely by first appearance
    order = obs_raw[np.sort(first_pos)]
    remap = np.full(base["F"], -1, dtype=np.int64)
    remap[order] = np.arange(order.size)
    p = dict(base)
    p.update(obs_t=base["obs_t"][idx], obs_f=remap[obs_raw].astype(np.uint32), obs_mean=base["obs_mean"][idx],
             obs_std=base["obs_std"][idx], N=int(idx.size), F=int(order.size))
    return p


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_ragged_data_online_matches_oracle(seed):
    p = _ragged_problem(seed)
    assert p["obs_t"][0] == 0 and np.setdiff1d(np.arange(p["T"]), p["obs_t"]).size > 10
    seeds = capi.seed_sequence(100 + seed, 3)
    w = oracle_world(p, seeds, steps_per_timestep=4, end_steps=150)
    w.replay(p["T"], finish=True)
    g = gpu_handle(p, seeds, trace_capacity=2048)
    g.advance(p["T"], steps_per_timestep=4, end_steps=150)
    g.sync()
    ll, full = g.log_likelihood(), g.full_log_likelihood()
    for c in range(len(seeds)):
        compare_traces(g.trace(c), w.trace(c), abs(w.log_likelihood(c)))
        assert_close(ll[c], w.log_likelihood(c), REL, f"chain {c} final log likelihood")
        assert_close(full[c], w.full_log_likelihood(c), REL, f"chain {c} full log likelihood")
        assert np.array_equal(g.get_state(c)["feat_parent"], w.state(c)["feat_parent"])


def test_no_readings_at_all_and_a_single_timestep():
    """Only state edges (no landmark ever seen), and the smallest problem there is."""
    p = small_problem(T=60, F=4, k=2, seed=9)
    empty = np.zeros(0)
    seeds = capi.seed_sequence(5, 2)
    for T in (60, 1):
        w = oracle_lib.World()
        w.set_input(p["ctrl_mean"][:T], p["ctrl_std"][:T], empty.astype(np.uint32), empty.astype(np.uint32), empty, empty)
        for s in seeds:
            w.add_chain(int(s), steps_per_timestep=2, end_steps=50, trace=True)
        w.replay(T, finish=True)
        g = capi.McmcGpu(len(seeds), T, 1, 1, trace_capacity=512)
        g.upload_problem(p["ctrl_mean"][:T], p["ctrl_std"][:T], empty, empty, empty, empty, 0)
        g.init_chains(seeds)
        g.advance(T, steps_per_timestep=2, end_steps=50)
        g.sync()
        for c in range(len(seeds)):
            compare_traces(g.trace(c), w.trace(c), max(1.0, abs(w.log_likelihood(c))))
            assert_close(g.log_likelihood()[c], w.log_likelihood(c), REL, "log likelihood")
        assert g.dims()[1] == 0


def test_capacity_and_argument_errors_are_reported():
    p = small_problem(T=50, F=4, k=2, seed=2)
    g = capi.McmcGpu(2, 40, 4, p["N"])  # too few timesteps
    with pytest.raises(RuntimeError, match="capacities"):
        g.upload_problem(p["ctrl_mean"], p["ctrl_std"], p["obs_t"], p["obs_f"], p["obs_mean"], p["obs_std"], p["F"])
    g = capi.McmcGpu(2, p["T"], p["F"], p["N"])
    with pytest.raises(RuntimeError, match="arrival order"):
        g.upload_problem(p["ctrl_mean"], p["ctrl_std"], p["obs_t"][::-1].copy(), p["obs_f"], p["obs_mean"], p["obs_std"],
                         p["F"])
    with pytest.raises(RuntimeError, match="first appearance"):
        g.upload_problem(p["ctrl_mean"], p["ctrl_std"], p["obs_t"], (p["F"] - 1 - p["obs_f"]).astype(np.uint32),
                         p["obs_mean"], p["obs_std"], p["F"])
    with pytest.raises(RuntimeError, match="timestep 0|upload a problem"):
        g.init_chains(capi.seed_sequence(1, 2))
        g.run(5)
    g.upload_problem(p["ctrl_mean"], p["ctrl_std"], p["obs_t"], p["obs_f"], p["obs_mean"], p["obs_std"], p["F"])
    with pytest.raises(RuntimeError, match="out of range"):
        g.init_chains(capi.seed_sequence(1, 3))  # more chains than the handle was created for
    with pytest.raises(RuntimeError, match="power of two"):
        capi.McmcGpu(1, 10, 1, 1, threads_per_chain=96)


def test_full_size_problem_properties_and_oracle_spot_check():
    """The benchmark's problem size (100 000 poses, 1 000 landmarks, 1 000 000 readings): two chains step
    for step against the oracle, and on all chains the size-independent properties -- incremental equals
    from-scratch log likelihood, a rebuild of the pose table changes nothing, counters add up."""
    p = synthetic.make_problem(T=100_000, n_landmarks=1000, k_nearest=10, seed=12345)
    C, steps = 32, 120
    seeds = capi.seed_sequence(12347, C)  # chains 0, 15 and 31 are replayed by the oracle
    g = capi.McmcGpu(C, p["T"], p["F"], p["N"], trace_capacity=steps)
    g.upload_problem(p["ctrl_mean"], p["ctrl_std"], p["obs_t"], p["obs_f"], p["obs_mean"], p["obs_std"], p["F"])
    g.init_chains(seeds)
    g.load_state(p["true_rel"], p["landmarks"])
    g.sync()
    ll0 = g.log_likelihood()
    assert np.all(ll0 == ll0[0])  # every chain starts from the same state
    g.run(steps)
    g.sync()
    ll, full = g.log_likelihood(), g.full_log_likelihood()
    assert_close(ll, full, REL, "incremental vs from-scratch log likelihood, all chains")
    upd, acc, units = g.counters()
    assert np.all(upd == steps) and np.all(acc <= steps) and units.sum() > 1e8
    cum_before = g.get_state(3)["cum"]
    g.rebuild()
    g.sync()
    assert_close(g.get_state(3)["cum"], cum_before, REL, "pose table after a rebuild")
    assert_close(g.full_log_likelihood(), full, REL, "log likelihood after a rebuild")
    w = oracle_lib.World()
    w.set_input(p["ctrl_mean"], p["ctrl_std"], p["obs_t"], p["obs_f"], p["obs_mean"], p["obs_std"])
    w.replay(p["T"])
    for c in (0, C // 2 - 1, C - 1):
        k = w.add_detached_chain(int(seeds[c]), trace=True)
        w.load_state(k, p["true_rel"], np.arange(p["F"]), p["landmarks"])
        assert_close(ll0[c], w.log_likelihood(k), REL, "log likelihood of the loaded state")
        assert w.update(k, steps) >= 0
        print("chain", c, "log-ratio differences (absolute, relative):", log_ratio_errors(g.trace(c), w.trace(k)))
        compare_traces(g.trace(c), w.trace(k))
        assert units[c] == w.units(k)


@pytest.mark.parametrize("slices,uniform_sigma", [(1, True), (3, True), (8, False)])
def test_split_mode_matches_oracle(slices, uniform_sigma):
    """mcmcgpu_run_split: every update as a sequence of kernels with the reading stream cut into slices --
    the same steps as the oracle (and so as the one-workgroup-per-chain kernel)."""
    p = small_problem(T=600, F=20, k=5, seed=23)
    if not uniform_sigma:
        p = dict(p)
        p["obs_std"] = p["obs_std"] * (1.0 + 0.3 * np.sin(np.arange(p["N"])))
    seeds = capi.seed_sequence(808, 3)
    steps = 400
    w = oracle_world(p, seeds)
    w.replay(p["T"])
    g = gpu_handle(p, seeds, trace_capacity=4096, max_split_slices=8)
    g.grow(p["T"])
    g.run_split(150, slices)
    g.run(100)                      # the two ways of running mix freely
    g.run_split(steps - 250, slices)
    g.sync()
    ll, full = g.log_likelihood(), g.full_log_likelihood()
    upd, acc, units = g.counters()
    for c in range(len(seeds)):
        assert w.update(c, steps) >= 0
        compare_traces(g.trace(c), w.trace(c), abs(w.log_likelihood(c)))
        assert_close(ll[c], w.log_likelihood(c), REL, f"chain {c} final log likelihood")
        assert_close(ll[c], full[c], 1e-8, f"chain {c} incremental vs full")
        assert_close(g.get_state(c)["cum"], w.state(c)["cum"], REL, "cumulative poses")
        assert units[c] == w.units(c) and upd[c] == steps
    plain = gpu_handle(p, seeds, trace_capacity=0)
    with pytest.raises(RuntimeError, match="without split mode"):
        plain.grow(p["T"])
        plain.run_split(5, 2)


def test_split_mode_with_range_bearing_readings():
    p = _rb_problem(T=300, F=12, k=4, seed=13)
    seeds = capi.seed_sequence(333, 2)
    steps = 300
    w = oracle_lib.World()
    w.set_input(p["ctrl_mean"], p["ctrl_std"], p["obs_t"], p["obs_f"], p["obs_mean"], p["obs_std"])
    w.set_bearings(p["obs_bearing"], p["obs_bearing_std"])
    for s in seeds:
        w.add_chain(int(s), trace=True)
    w.replay(p["T"])
    g = capi.McmcGpu(len(seeds), p["T"], p["F"], p["N"], trace_capacity=4096, range_bearing=True, max_split_slices=4)
    g.upload_problem(p["ctrl_mean"], p["ctrl_std"], p["obs_t"], p["obs_f"], p["obs_mean"], p["obs_std"], p["F"])
    g.set_bearings(p["obs_bearing"], p["obs_bearing_std"])
    g.init_chains(seeds)
    g.grow(p["T"])
    g.run_split(steps, 4)
    g.sync()
    ll = g.log_likelihood()
    for c in range(len(seeds)):
        assert w.update(c, steps) >= 0
        compare_traces(g.trace(c), w.trace(c), abs(w.log_likelihood(c)))
        assert_close(ll[c], w.log_likelihood(c), REL, f"chain {c} final log likelihood")


@pytest.mark.parametrize("split", [False, True])
def test_range_bearing_with_a_sigma_per_reading(split):
    """Readings whose range and bearing deviations differ from one to the next: the stream then loads
    both per reading instead of holding one pair in registers."""
    p = _rb_problem(T=300, F=12, k=4, seed=13)
    rng = np.random.default_rng(5)
    p["obs_std"] = p["obs_std"] * rng.uniform(1.0, 1.6, size=p["N"])
    p["obs_bearing_std"] = p["obs_bearing_std"] * rng.uniform(1.0, 1.6, size=p["N"])
    seeds = capi.seed_sequence(808, 2)
    steps = 300
    w = oracle_lib.World()
    w.set_input(p["ctrl_mean"], p["ctrl_std"], p["obs_t"], p["obs_f"], p["obs_mean"], p["obs_std"])
    w.set_bearings(p["obs_bearing"], p["obs_bearing_std"])
    for s in seeds:
        w.add_chain(int(s), trace=True)
    w.replay(p["T"])
    g = capi.McmcGpu(len(seeds), p["T"], p["F"], p["N"], trace_capacity=4096, range_bearing=True,
                     max_split_slices=4 if split else 0)
    g.upload_problem(p["ctrl_mean"], p["ctrl_std"], p["obs_t"], p["obs_f"], p["obs_mean"], p["obs_std"], p["F"])
    g.set_bearings(p["obs_bearing"], p["obs_bearing_std"])
    g.init_chains(seeds)
    g.grow(p["T"])
    if split:
        g.run_split(steps, 4)
    else:
        g.run(steps)
    g.sync()
    ll, full = g.log_likelihood(), g.full_log_likelihood()
    for c in range(len(seeds)):
        assert w.update(c, steps) >= 0
        compare_traces(g.trace(c), w.trace(c), abs(w.log_likelihood(c)))
        assert_close(ll[c], w.log_likelihood(c), REL, f"chain {c} final log likelihood")
        assert_close(full[c], w.full_log_likelihood(c), REL, f"chain {c} full log likelihood")


def _thinned(p, rng, keep):
    """A copy of problem `p` with a random subset of its readings: landmarks renumbered by first
    appearance (what slam_data does), some timesteps left without readings, some landmarks seen once."""
    sel = rng.random(p["N"]) < keep
    if not sel.any():
        sel[0] = True
    q = dict(p)
    t, f = p["obs_t"][sel], p["obs_f"][sel]
    ids, first = {}, []
    for x in f:
        if int(x) not in ids:
            ids[int(x)] = len(ids)
            first.append(int(x))
    q["obs_t"] = np.ascontiguousarray(t)
    q["obs_f"] = np.array([ids[int(x)] for x in f], dtype=p["obs_f"].dtype)
    q["obs_mean"] = np.ascontiguousarray(p["obs_mean"][sel])
    q["obs_std"] = np.ascontiguousarray(p["obs_std"][sel])
    q["N"], q["F"] = int(sel.sum()), len(ids)
    return q


@pytest.mark.parametrize("case", range(12))
def test_random_small_problems_match_oracle(case):
    """A sweep over small random shapes -- 1 to 90 timesteps, 1 to 9 landmarks, 15 to 100 % of the
    readings kept, 0 to 3 updates per timestep, both draw orders, 64 to 256 lanes -- each replayed
    online and compared with the oracle step for step, alternately with either build of the serial part."""
    rng = np.random.default_rng(1000 + case)
    T = int(rng.integers(1, 91))
    F = int(rng.integers(1, 10))
    p = synthetic.make_problem(T=T, n_landmarks=F, k_nearest=int(rng.integers(1, F + 1)), seed=int(rng.integers(1, 1 << 20)),
                               lap_steps=int(rng.integers(20, 120)), wobble_period=500, range_stddev=2.0,
                               landmark_spread=6.0)
    p = _thinned(p, rng, keep=float(rng.uniform(0.15, 1.0)))
    per_ts, end_steps = int(rng.integers(0, 4)), int(rng.integers(0, 120))
    uniform_first = bool(rng.integers(0, 2))
    lanes = int(rng.choice([64, 128, 256]))
    seeds = capi.seed_sequence(77 + case, 2)
    w = oracle_lib.World()
    w.set_input(p["ctrl_mean"], p["ctrl_std"], p["obs_t"], p["obs_f"], p["obs_mean"], p["obs_std"])
    for s in seeds:
        w.add_chain(int(s), steps_per_timestep=per_ts, end_steps=end_steps, uniform_first=uniform_first, trace=True)
    w.replay(p["T"], finish=True)
    g = gpu_handle(p, seeds, trace_capacity=4096, feature_uniform_first=uniform_first, threads_per_chain=lanes,
                   serial_program=1 + case % 2)  # the serial part as a called function / compiled into the kernel
    g.advance(p["T"], steps_per_timestep=per_ts, end_steps=end_steps)
    g.sync()
    ll, full = g.log_likelihood(), g.full_log_likelihood()
    for c in range(len(seeds)):
        compare_traces(g.trace(c), w.trace(c), max(abs(w.log_likelihood(c)), 1.0))
        assert_close(ll[c], w.log_likelihood(c), REL, f"chain {c} final log likelihood")
        assert_close(full[c], w.full_log_likelihood(c), REL, f"chain {c} from-scratch log likelihood")
        so, sg = w.state(c), g.get_state(c)
        assert np.array_equal(sg["feat_parent"], so["feat_parent"])
        assert_close(sg["cum"], so["cum"], REL, "cumulative poses")
        assert_close(sg["feat_est"], so["feat_est"], REL, "landmark estimates")


def _append_all(g, p, upto_from=0, bearings=False, per_ts=3, end_steps=50, init_rel=None, init_feat=None):
    """Feeds problem `p` to handle `g` one timestep at a time with mcmcgpu_append_timestep, advancing after each."""
    seen = 0
    for t in range(upto_from, p["T"] + 1):
        lo = int(np.searchsorted(p["obs_t"], t, side="left"))
        hi = int(np.searchsorted(p["obs_t"], t, side="right"))
        f = p["obs_f"][lo:hi]
        new = [int(x) for x in f if int(x) >= seen]
        kw = {}
        if bearings:
            kw.update(bearing_mean=p["obs_bearing"][lo:hi], bearing_std=p["obs_bearing_std"][lo:hi])
        if init_rel is not None and t > 0:
            kw.update(init_state_rel=init_rel[t - 1])
        if init_feat is not None and new:
            kw.update(init_feat_rel=np.concatenate([init_feat[x] for x in new]))
        g.append_timestep(None if t == 0 else p["ctrl_mean"][t - 1], None if t == 0 else p["ctrl_std"][t - 1], f,
                          p["obs_mean"][lo:hi], p["obs_std"][lo:hi], **kw)
        seen += len(new)
        g.advance(t, steps_per_timestep=per_ts, end_steps=end_steps if t == p["T"] else 0)


def test_appended_timesteps_equal_one_upload():
    """mcmcgpu_append_timestep, timestep by timestep from an empty handle (in-place appends and the
    re-layouts that make room), gives the chain of one upload followed by one online replay."""
    p = small_problem(T=200, F=9, k=3, seed=5)
    seeds = np.array([77, 78], dtype=np.uint32)
    g1 = gpu_handle(p, seeds, trace_capacity=2048)
    g1.advance(p["T"], steps_per_timestep=3, end_steps=50)
    g1.sync()
    g2 = capi.McmcGpu(2, p["T"], p["F"], p["N"], trace_capacity=2048)
    g2.init_chains(seeds)
    _append_all(g2, p)
    g2.sync()
    for c in range(2):
        t1, t2 = g1.trace(c), g2.trace(c)
        assert len(t1) == len(t2)
        for name in ("kind", "index", "accepted", "log_ratio", "log_likelihood", "proposed"):
            assert np.array_equal(t1[name], t2[name]), name
    assert np.array_equal(g1.full_log_likelihood(), g2.full_log_likelihood())
    # the lists that outgrew their slots were moved (twice each here): a second replay from scratch over the
    # appended data walks the arrival index and the moved lists, and must give the same chains again
    g2.init_chains(seeds)
    g2.advance(p["T"], steps_per_timestep=3, end_steps=50)
    g2.sync()
    for c in range(2):
        t1, t2 = g1.trace(c), g2.trace(c)
        assert len(t1) == len(t2)
        for name in ("kind", "index", "accepted", "log_ratio", "log_likelihood", "proposed"):
            assert np.array_equal(t1[name], t2[name]), "replay after moves: " + name
    # ... and continuing an uploaded problem by appends (compact layout first, then room to grow)
    half = 100
    n = int(np.searchsorted(p["obs_t"], half, side="right"))
    F = int(p["obs_f"][:n].max()) + 1
    g3 = capi.McmcGpu(2, p["T"] + 4, p["F"], p["N"] + 8, trace_capacity=2048)
    g3.upload_problem(p["ctrl_mean"][:half], p["ctrl_std"][:half], p["obs_t"][:n], p["obs_f"][:n], p["obs_mean"][:n],
                      p["obs_std"][:n], F)
    g3.init_chains(seeds)
    g3.advance(half, steps_per_timestep=3, end_steps=0)
    q = dict(p)
    _append_all(g3, q, upto_from=half + 1)
    g3.sync()
    # same chain as g1 only up to the hand-over (g1's replay interleaves identically): compare end state
    for c in range(2):
        t1, t3 = g1.trace(c), g3.trace(c)
        assert len(t1) == len(t3)
        assert np.array_equal(t1["accepted"], t3["accepted"]) and np.array_equal(t1["log_likelihood"], t3["log_likelihood"])
    with pytest.raises(RuntimeError, match="dense"):
        g3.append_timestep(p["ctrl_mean"][0], p["ctrl_std"][0], [p["F"] + 5], [1.0], [1.0])
    with pytest.raises(RuntimeError, match="two readings"):
        g3.append_timestep(p["ctrl_mean"][0], p["ctrl_std"][0], [0, 0], [1.0, 1.0], [1.0, 1.0])


def test_appended_timesteps_with_bearings_and_initial_estimates():
    """The same for range-and-bearing readings, with an initialiser's estimates handed over edge by edge."""
    p = _rb_problem(T=150, F=8, k=3, seed=13)
    seeds = np.array([31], dtype=np.uint32)
    rng = np.random.default_rng(4)
    init_rel = p["true_rel"] + rng.normal(0, [0.01, 0.01, 0.002], size=p["true_rel"].shape)
    init_feat = rng.normal(0, 3.0, size=(p["F"], 2))
    g1 = capi.McmcGpu(1, p["T"], p["F"], p["N"], trace_capacity=2048, range_bearing=True)
    g1.upload_problem(p["ctrl_mean"], p["ctrl_std"], p["obs_t"], p["obs_f"], p["obs_mean"], p["obs_std"], p["F"])
    g1.set_bearings(p["obs_bearing"], p["obs_bearing_std"])
    g1.init_chains(seeds)
    g1.set_initial_estimates(init_rel, init_feat)
    g1.advance(p["T"], steps_per_timestep=2, end_steps=40)
    g1.sync()
    g2 = capi.McmcGpu(1, p["T"], p["F"], p["N"], trace_capacity=2048, range_bearing=True)
    g2.init_chains(seeds)
    _append_all(g2, p, bearings=True, per_ts=2, end_steps=40, init_rel=init_rel, init_feat=init_feat)
    g2.sync()
    t1, t2 = g1.trace(0), g2.trace(0)
    assert len(t1) == len(t2)
    for name in ("kind", "index", "accepted", "log_ratio", "log_likelihood", "proposed"):
        assert np.array_equal(t1[name], t2[name]), name
    assert np.array_equal(g1.get_state(0)["cum"], g2.get_state(0)["cum"])


def test_split_mode_with_the_pose_update_as_a_kernel_of_its_own():
    """Accepted state edges moved by split_commit_kernel (the path long trajectories take) give the
    chain the ends kernel's own update gives, bit for bit."""
    p = small_problem(T=500, F=14, k=4, seed=9, range_stddev=3.0)
    seeds = capi.seed_sequence(555, 3)
    steps = 400
    out = []
    for min_poses in (0, 1):  # 0: the default threshold (200 000 poses, not reached here); 1: always
        g = gpu_handle(p, seeds, trace_capacity=4096, max_split_slices=8, split_commit_min_poses=min_poses)
        g.grow(p["T"])
        g.run_split(steps, 8)
        g.sync()
        out.append((g.trace(0), g.trace(2), g.log_likelihood(), g.full_log_likelihood(), g.get_state(1)["cum"]))
    assert out[0][0]["accepted"].sum() > 5 and (out[0][0]["kind"][out[0][0]["accepted"] == 1] == 0).any()
    for a, b in zip(out[0][:2], out[1][:2]):
        for name in ("kind", "index", "accepted", "log_ratio", "log_likelihood", "proposed"):
            assert np.array_equal(a[name], b[name]), name
    for a, b in zip(out[0][2:], out[1][2:]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("slices", [8, 1])
def test_split_mode_with_more_landmarks_than_one_workgroup_prepares(slices):
    """More than 1024 landmarks: the segments are prepared by several workgroups per chain and the
    prefix sums by a kernel of their own (up to 1024 one kernel does both).  With a single slice the
    stream kernel meets more segments than it stages in LDS and walks the tables in global memory."""
    p = synthetic.make_problem(T=1500, n_landmarks=1400, k_nearest=8, seed=3, lap_steps=150, wobble_period=500,
                               range_stddev=2.0, landmark_spread=6.0)
    assert p["F"] > 1024
    seeds = capi.seed_sequence(4321, 2)
    steps = 150
    w = oracle_world(p, seeds)
    w.replay(p["T"])
    g = gpu_handle(p, seeds, trace_capacity=4096, max_split_slices=8)
    g.grow(p["T"])
    g.run_split(steps, slices)
    g.sync()
    ll = g.log_likelihood()
    for c in range(len(seeds)):
        assert w.update(c, steps) >= 0
        compare_traces(g.trace(c), w.trace(c), abs(w.log_likelihood(c)))
        assert_close(ll[c], w.log_likelihood(c), REL, f"chain {c} final log likelihood")


def test_split_mode_leaves_the_report_records_of_a_reported_launch():
    """mcmcgpu_run_split_reported against mcmcgpu_run_reported: the same summary per chain and interval."""
    p = small_problem(T=300, F=12, k=4, seed=7)
    seeds = capi.seed_sequence(90, 3)
    a = gpu_handle(p, seeds, trace_capacity=0)
    a.grow(p["T"])
    a.run_reported(250, 100, None)
    ra = a.get_reports()
    b = gpu_handle(p, seeds, trace_capacity=0, max_split_slices=4)
    b.grow(p["T"])
    b.run_split_reported(250, 4, 100, None)
    rb = b.get_reports()
    assert ra.shape == rb.shape == (3, 3)
    assert np.array_equal(ra["updates"], rb["updates"]) and np.array_equal(ra["accepts"], rb["accepts"])
    assert_close(rb["log_likelihood"], ra["log_likelihood"], REL, "log likelihoods at the interval boundaries")
    assert list(ra["updates"][:, 0]) == [100, 200, 250]


@pytest.mark.parametrize("seed", [1, 5489, 20240607])
def test_wavefront_forms_of_generator_and_descent_equal_the_serial_ones(seed):
    """The serial part compiled into the kernel draws from the chain's generator through a 64-output window
    (polar trials side by side) and descends the weight trees six levels at a time; both must hand out exactly
    what the one-lane code does: same doubles, same indices.  4000 draws cross the generator's 624-word blocks
    some twenty times; the trees range from one node to Plaza1's size and include zero weights."""
    rng = np.random.default_rng(seed)
    kinds = (rng.random(4000) < 0.45).astype(np.uint8)
    kinds[rng.random(4000) < 0.2] = 2  # three normals as one batch (a state-edge proposal)
    kinds[rng.random(4000) < 0.05] = 3  # two (a range-and-bearing landmark proposal)
    for tree_n in (1, 2, 3, 5, 64, 65, 1000, 9657):
        weights = rng.exponential(1.0, tree_n) * (rng.random(tree_n) > 0.1)
        weights[0] = max(weights[0], 1e-3)
        u = np.concatenate([rng.random(300), [0.0, 1.0, 1.0 - 2.0 ** -53, 0.5, 1.0 + 2.0 ** -40]])
        ds, dw, i_s, i_w = capi.selftest_wave(seed + tree_n, kinds, weights, u)
        assert np.array_equal(ds, dw), f"draws differ first at {np.nonzero(ds != dw)[0][:5]}"
        assert np.array_equal(i_s, i_w), f"tree of {tree_n}: indices differ at {np.nonzero(i_s != i_w)[0][:5]}"
        assert np.all(i_s <= tree_n)


def test_tree_tops_at_full_capacity_and_many_narrow_workgroups():
    """The LDS copies of the weight trees' tops at the edges of their planning.  (a) 1024 chains whose capacity
    equals the data (max_timesteps = T = 4096, a multiple of the copy's stride): the copy of every second node
    must end at the last valid node, not one stride further -- in the last chain that would be past the
    allocation.  (b) 2304 chains of 64 lanes: nine workgroups share a CU and the static block is all the LDS
    each gets; the launch must still run, without copies.  First and last chain step for step against the oracle."""
    for chains, lanes, T, steps in ((1024, 0, 4096, 40), (2304, 64, 96, 60)):
        p = synthetic.make_problem(T=T, n_landmarks=24, k_nearest=3, seed=29, lap_steps=300, wobble_period=700,
                                   range_stddev=1.5, landmark_spread=8.0)
        assert p["T"] == T
        seeds = capi.seed_sequence(4242, chains)
        g = capi.McmcGpu(chains, p["T"], p["F"], p["N"], trace_capacity=steps, threads_per_chain=lanes)
        g.upload_problem(p["ctrl_mean"], p["ctrl_std"], p["obs_t"], p["obs_f"], p["obs_mean"], p["obs_std"], p["F"])
        g.init_chains(seeds)
        g.load_state(p["true_rel"], p["landmarks"])
        g.run(steps)
        g.sync()
        ll = g.log_likelihood()
        w = oracle_lib.World()
        w.set_input(p["ctrl_mean"], p["ctrl_std"], p["obs_t"], p["obs_f"], p["obs_mean"], p["obs_std"])
        w.replay(p["T"])
        for c in (0, chains - 1):
            k = w.add_detached_chain(int(seeds[c]), trace=True)
            w.load_state(k, p["true_rel"], np.arange(p["F"]), p["landmarks"])
            assert w.update(k, steps) >= 0
            compare_traces(g.trace(c), w.trace(k))
            assert_close(ll[c], w.log_likelihood(k), REL, f"{chains} chains: chain {c} log likelihood")
        g.close()
