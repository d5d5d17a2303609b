"""The two size regimes with code paths of their own, at their real sizes (the other GPU tests use small
problems): a ladder of tempered replicas on a 1 000 000-pose trajectory in split mode, and 1024 chains in one
reported launch on the benchmark's problem.  Step-for-step comparisons against the CPU oracle where it
finishes in seconds, size-independent properties everywhere else."""
import numpy as np
import pytest
import torch

import oracle_lib
from slam_amd import capi, synthetic
from test_gpu_parity import REL, assert_close, compare_traces, log_ratio_errors

pytestmark = pytest.mark.gpu

torch.cuda.is_available()  # torch's HIP runtime first (the summaries buffer below is a torch tensor)


def test_million_pose_ladder_in_split_mode():
    """1 000 000 poses, 10 000 000 readings, one ladder of 64 replicas (the ladder bench.py --workload tempering
    runs), mcmcgpu_run_split with the DEFAULT
    threshold for the many-workgroup pose-table update (split_commit_kernel runs from 200 000 poses on), then
    replica exchanges.  The beta = 1 chain and the hottest chain step for step against the oracle; incremental
    against from-scratch log likelihood and against a rebuild of the pose tables on all chains (the
    reference's own invariant, src/slam/slam_likelihood.hpp:26-58 vs mcmc_slam.hpp:173)."""
    p = synthetic.make_problem(T=1_000_000, n_landmarks=1000, k_nearest=10, seed=12345, max_tries=2)
    T, F, N = p["T"], p["F"], p["N"]
    assert (T, N) == (1_000_000, 10_000_000)
    R, steps, slices = 64, 12, 256
    betas = np.geomspace(1.0, 1e-5, R)  # the hot end accepts most state edges: the pose-table kernel gets work
    seeds = capi.seed_sequence(4711, R)
    g = capi.McmcGpu(R, T, F, N, trace_capacity=2 * steps, max_split_slices=slices)  # split_commit_min_poses = default
    g.upload_problem(p["ctrl_mean"], p["ctrl_std"], p["obs_t"], p["obs_f"], p["obs_mean"], p["obs_std"], F)
    g.init_chains(seeds)
    g.load_state(p["true_rel"], p["landmarks"])
    g.set_ladders(betas, exchange_seed=99)
    g.run_split(steps, slices)
    g.sync()
    traces = [g.trace(c) for c in range(R)]
    accepted_state_edges = sum(int(np.sum((t["kind"] == 0) & (t["accepted"] == 1))) for t in traces)
    assert accepted_state_edges >= 1, "no state edge was accepted: the pose-table kernel never ran"
    ll, full = g.log_likelihood(), g.full_log_likelihood()
    assert_close(ll, full, REL, "incremental vs from-scratch log likelihood after split-mode updates")
    mover = max(range(R), key=lambda c: int(np.sum((traces[c]["kind"] == 0) & (traces[c]["accepted"] == 1))))
    before = g.get_state(mover)["cum"]
    g.rebuild()
    g.sync()
    assert_close(g.get_state(mover)["cum"], before, REL, "pose table moved by accepted edges vs a rebuild")
    assert_close(g.full_log_likelihood(), full, REL, "log likelihood after the rebuild")
    # the oracle on the same two chains, same temperatures
    w = oracle_lib.World()
    w.set_input(p["ctrl_mean"], p["ctrl_std"], p["obs_t"], p["obs_f"], p["obs_mean"], p["obs_std"])
    w.replay(T)
    _, _, units = g.counters()
    for c in (0, R - 1):
        k = w.add_detached_chain(int(seeds[c]), trace=True)
        w.load_state(k, p["true_rel"], np.arange(F), p["landmarks"])
        w.set_beta(k, betas[c])
        assert w.update(k, steps) >= 0
        print("chain", c, "log-ratio differences (absolute, in units of the terms' sum):", log_ratio_errors(traces[c], w.trace(k)))
        compare_traces(traces[c], w.trace(k))
        assert units[c] == w.units(k), "evaluated-observation counts differ"
        if c == mover:
            assert_close(before, w.state(k)["cum"], REL, "cumulative poses vs the oracle's tree")
    # replica exchange, then more updates at the swapped temperatures: the invariants still hold
    g.exchange(0)
    g.exchange(1)
    beta_now, slot, _, attempts, swaps = g.ladders()
    assert attempts == R // 2 + (R - 1) // 2 and sorted(beta_now) == sorted(betas)
    assert np.array_equal(np.sort(slot), np.arange(R))
    g.run_split(6, slices)
    g.sync()
    assert_close(g.log_likelihood(), g.full_log_likelihood(), REL, "incremental vs from-scratch after the exchange")
    upd, _, _ = g.counters()
    assert np.all(upd == steps + 6)


def test_benchmark_launch_shape_1024_chains():
    """The benchmark's launch as it is timed: 1024 chains on 100 000 poses / 1 000 000 readings, ONE
    mcmcgpu_run_reported(1000, 100) launch and a rebuild.  On all chains: report records grow by 100 updates
    per interval and end at the chain's final state, accepted never exceeds updates, incremental equals
    from-scratch log likelihood before and after the rebuild."""
    p = synthetic.make_problem(T=100_000, n_landmarks=1000, k_nearest=10, seed=12345)
    C, steps, every = 1024, 1000, 100
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)
    g = capi.McmcGpu(C, p["T"], p["F"], p["N"], stream=stream.cuda_stream)
    g.upload_problem(p["ctrl_mean"], p["ctrl_std"], p["obs_t"], p["obs_f"], p["obs_mean"], p["obs_std"], p["F"])
    g.init_chains(capi.seed_sequence(12347, C))
    g.load_state(p["true_rel"], p["landmarks"])
    out = torch.zeros((steps // every, C, 2), dtype=torch.float64, device=dev)
    g.run_reported(steps, every, out.data_ptr())
    g.sync()
    rec = out.cpu().numpy().view(capi.CHAIN_SUMMARY).reshape(steps // every, C)
    assert np.array_equal(rec["updates"], np.repeat(np.arange(every, steps + 1, every)[:, None], C, axis=1))
    assert np.all(np.diff(rec["accepts"].astype(np.int64), axis=0) >= 0) and np.all(rec["accepts"] <= rec["updates"])
    ll, full = g.log_likelihood(), g.full_log_likelihood()
    assert np.array_equal(rec["log_likelihood"][-1], ll)
    assert_close(ll, full, REL, "incremental vs from-scratch log likelihood, 1024 chains x 1000 updates")
    upd, acc, units = g.counters()
    assert np.all(upd == steps) and np.array_equal(acc, rec["accepts"][-1]) and units.sum() > 5e10
    g.rebuild()
    g.sync()
    assert_close(g.full_log_likelihood(), full, REL, "from-scratch log likelihood after the rebuild")
    torch.cuda.set_stream(torch.cuda.default_stream(dev))
