"""CPU tests of the multi-GPU layer (slam_amd/parallel.py) on the gloo backend,
world size 2: seed sharding, the summary all-gather, the best-chain arg-max with
its tie rule, and the broadcast of the winner's state.  Also the synthetic
generator's contract."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib
from slam_amd import capi, parallel, synthetic


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _summaries(lls, updates, accepts):
    rec = np.zeros(len(lls), dtype=capi.CHAIN_SUMMARY)
    rec["log_likelihood"] = lls
    rec["updates"] = updates
    rec["accepts"] = accepts
    return torch.from_numpy(rec.view(np.float64).reshape(-1, 2).copy())


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    C = 4
    all_seeds = capi.seed_sequence(99, C * world)
    mine = parallel.shard_seeds(all_seeds, C, rank)
    # log likelihoods: the global maximum -1.0 appears on BOTH ranks (global chains 2 and 5)
    lls = np.array([-5.0, -3.0, -1.0, -2.0]) if rank == 0 else np.array([-4.0, -1.0, -6.0, -7.0])
    local = _summaries(lls, np.arange(C) + 10 * rank, np.arange(C))
    gathered = parallel.gather_summaries(local, world)
    ll, gid, owner, idx = parallel.best_chain(gathered, C)
    upd, acc = parallel.counts_of(gathered)
    state = torch.full((6,), float(rank + 1), dtype=torch.float64)
    parallel.broadcast_from_owner(state, owner, world)
    out[rank] = (mine.tolist(), gathered[:, 0].tolist(), ll, gid, owner, idx, upd.tolist(), acc.tolist(),
                 state.tolist())
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_gloo():
    world = 2
    port = _free_port()
    with mp.Manager() as m:
        out = m.dict()
        mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
        res = dict(out)
    seeds = capi.seed_sequence(99, 8).tolist()
    assert res[0][0] == seeds[:4] and res[1][0] == seeds[4:]  # disjoint blocks of ONE sequence
    for r in (0, 1):
        _, lls, ll, gid, owner, idx, upd, acc, state = res[r]
        assert lls == [-5.0, -3.0, -1.0, -2.0, -4.0, -1.0, -6.0, -7.0]  # global chain order
        assert (ll, gid, owner, idx) == (-1.0, 2, 0, 2)  # tie -> lowest global chain id
        assert upd == [0, 1, 2, 3, 10, 11, 12, 13] and acc == [0, 1, 2, 3, 0, 1, 2, 3]
        assert state == [1.0] * 6  # everyone holds the owner's (rank 0) state


def test_single_rank_needs_no_collective():
    local = _summaries(np.array([-2.0, -1.5, -9.0]), np.array([5, 5, 5]), np.array([1, 2, 3]))
    gathered = parallel.gather_summaries(local, 1)
    assert parallel.best_chain(gathered, 3) == (-1.5, 1, 0, 1)


def test_seed_sharding_matches_reference_multi_mcmc_order():
    """Chain i of the job gets the i-th draw of random_source(seed), whatever the GPU count."""
    seeds = oracle_lib.seed_sequence(7, 16)
    for world in (1, 2, 4, 8):
        C = 16 // world
        got = np.concatenate([parallel.shard_seeds(seeds, C, r) for r in range(world)])
        assert np.array_equal(got, seeds)


def test_synthetic_generator_contract():
    p = synthetic.make_problem(T=600, n_landmarks=25, k_nearest=5, seed=4, lap_steps=200, wobble_period=700)
    assert p["N"] == 600 * 5 and p["obs_t"].size == p["N"]
    assert np.all(np.diff(p["obs_t"].astype(np.int64)) >= 0)  # arrival order
    # dense landmark numbers in order of first appearance
    _, first = np.unique(p["obs_f"], return_index=True)
    assert np.array_equal(p["obs_f"][np.sort(first)], np.arange(p["F"]))
    # no landmark ranged twice in one timestep
    key = p["obs_t"].astype(np.int64) * p["F"] + p["obs_f"]
    assert np.unique(key).size == key.size
    assert np.all(p["ctrl_std"] > 0)
    q = synthetic.make_problem(T=600, n_landmarks=25, k_nearest=5, seed=4, lap_steps=200, wobble_period=700)
    assert np.array_equal(p["obs_mean"], q["obs_mean"]) and np.array_equal(p["ctrl_mean"], q["ctrl_mean"])
    # the readings are the true ranges plus noise of the stated size
    true = np.linalg.norm(p["true_cum"][p["obs_t"], :2] - p["landmarks"][p["obs_f"]], axis=1)
    assert abs(np.std(p["obs_mean"] - true) - 1.0) < 0.1
    np.testing.assert_allclose(synthetic.cumulative(p["true_rel"]), p["true_cum"])
