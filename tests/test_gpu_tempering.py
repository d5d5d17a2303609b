"""Tempered replicas on the GPU.  The reference has no tempering, so parity with
it is pinned at beta = 1 only (every other GPU parity test runs there); here a
heated chain is checked step for step against the oracle with the same beta, and
the replica exchange against its own rule recomputed in numpy."""
import numpy as np
import pytest

import oracle_lib
from slam_amd import capi
from test_gpu_parity import REL, assert_close, compare_traces, gpu_handle, small_problem

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("lanes,serial_program", [(0, 0), (64, 2), (128, 3)])
def test_heated_chains_match_oracle(lanes, serial_program):
    """Library defaults; one wavefront per chain with the serial program compiled in (a landmark proposal's
    accept test is then prepared before its evaluation, in two parts that both use the temperature); and two
    wavefronts with the next update selected ahead (the test decided by the evaluating wavefront)."""
    p = small_problem()
    betas = np.array([1.0, 0.5, 0.1, 0.02])
    seeds = capi.seed_sequence(515, len(betas))
    steps = 500
    w = oracle_lib.World()
    w.set_input(p["ctrl_mean"], p["ctrl_std"], p["obs_t"], p["obs_f"], p["obs_mean"], p["obs_std"])
    for s in seeds:
        w.add_chain(int(s), trace=True)
    w.replay(p["T"])
    g = gpu_handle(p, seeds, trace_capacity=4096, threads_per_chain=lanes, serial_program=serial_program)
    g.grow(p["T"])
    g.set_ladders(betas, exchange_seed=1)
    beta, slot, _, att, swp = g.ladders()
    assert np.array_equal(beta, betas) and np.array_equal(slot, np.arange(4)) and att == 0 and swp == 0
    for c in range(len(seeds)):
        w.set_beta(c, float(betas[c]))
        assert w.update(c, steps) >= 0
    g.run(steps)
    g.sync()
    ll = g.log_likelihood()
    accepted = []  # differs by temperature; not monotone, since the proposals stay the reference's (untempered) ones
    for c in range(len(seeds)):
        to, tg = w.trace(c), g.trace(c)
        compare_traces(tg, to, abs(w.log_likelihood(c)))
        assert_close(ll[c], w.log_likelihood(c), REL, f"chain {c} final log likelihood")
        accepted.append(int(tg["accepted"][-steps:].sum()))
    assert len(set(accepted)) > 1, f"temperatures should change the accept counts: {accepted}"
    # control part + readings' part = the whole
    assert np.all(g.ladders()[2] < ll), "control log densities are positive here, so the readings' part is below ll"


def _expected_sweep(ll, beta, slot, R, phase, u):
    """One exchange sweep restated on the host: same pair order, same rule."""
    beta, slot = beta.copy(), slot.copy()
    per = (R - phase) // 2
    swaps = 0
    for pair in range(per * (len(ll) // R)):
        s = (pair // per) * R + phase + 2 * (pair % per)
        a, b = slot[s], slot[s + 1]
        if u[pair] < np.exp((beta[a] - beta[b]) * (ll[b] - ll[a])):
            beta[a], beta[b] = beta[b], beta[a]
            slot[s], slot[s + 1] = b, a
            swaps += 1
    return beta, slot, per * (len(ll) // R), swaps


def test_exchange_follows_its_rule():
    p = small_problem(T=200, F=8, k=3, seed=5)
    R, ladders = 5, 3
    betas = np.geomspace(1.0, 0.05, R)
    seeds = capi.seed_sequence(77, R * ladders)
    g = gpu_handle(p, seeds, trace_capacity=0)
    g.grow(p["T"])
    exchange_seed = 4242
    g.set_ladders(betas, exchange_seed)
    u_all = capi.uniform_sequence(exchange_seed, 1000)
    used = 0
    beta = np.tile(betas, ladders)
    slot = np.arange(R * ladders, dtype=np.uint32)
    attempts = swaps = 0
    for sweep in range(12):
        g.run(20)
        g.sync()
        ll = g.ladders()[2]  # the readings' part of the log likelihood, which the temperatures weigh
        phase = sweep % 2
        g.exchange(phase)
        npairs = ((R - phase) // 2) * ladders
        beta, slot, a, s = _expected_sweep(ll, beta, slot, R, phase, u_all[used:used + npairs])
        used += npairs
        attempts += a
        swaps += s
        gb, gs, _, ga, gw = g.ladders()
        assert np.array_equal(gb, beta), f"sweep {sweep}: temperatures differ"
        assert np.array_equal(gs, slot), f"sweep {sweep}: slot owners differ"
        assert (ga, gw) == (attempts, swaps)
        # every ladder still holds each temperature exactly once, and beta follows the slots
        for l in range(ladders):
            assert sorted(gb[l * R:(l + 1) * R]) == sorted(betas)
            assert np.array_equal(gb[gs[l * R:(l + 1) * R]], betas)
    assert 0 < swaps < attempts
    # the log likelihoods are the untempered ones: the from-scratch value agrees
    assert_close(g.log_likelihood(), g.full_log_likelihood(), 1e-8, "incremental vs full under tempering")
