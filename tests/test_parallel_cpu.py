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


class _HostChains:
    """Stands in for a GPU handle in the averaging exchange: the three calls of the C ABI
    (mcmcgpu_average_base / _sum / _finish) restated with numpy on host memory."""

    def __init__(self, vectors, T, F):  # vectors: [chains, 3T + 2F]
        self.v, self.T, self.F = vectors, T, F

    def dims(self):
        return self.T, self.F, self.v.shape[0]

    def _view(self, ptr):
        import ctypes
        n = 3 * self.T + 2 * self.F
        return np.ctypeslib.as_array((ctypes.c_double * n).from_address(ptr))

    @staticmethod
    def _wrap(a):
        return (a + np.pi) % (2 * np.pi) - np.pi

    def average_base(self, chain, ptr):
        self._view(ptr)[:] = self.v[chain]

    def average_sum(self, base_ptr, skip, sum_ptr):
        base, out = self._view(base_ptr), self._view(sum_ptr)
        out[:] = 0.0
        angle = np.zeros(base.size, dtype=bool)
        angle[2:3 * self.T:3] = True
        for c in range(self.v.shape[0]):
            if c == skip:
                continue
            diff = base - self.v[c]
            out += np.where(angle, self._wrap(diff), diff)

    def average_finish(self, base_ptr, sum_ptr, n):
        base, total = self._view(base_ptr), self._view(sum_ptr)
        angle = np.zeros(base.size, dtype=bool)
        angle[2:3 * self.T:3] = True
        res = base - total / n
        base[:] = np.where(angle, self._wrap(res), res)


def _chain_vectors(T, F, chains, seed):
    rng = np.random.default_rng(seed)
    v = rng.normal(size=(chains, 3 * T + 2 * F))
    v[:, 2:3 * T:3] = rng.uniform(-3.1, 3.1, size=(chains, T))  # angles near the wrap on purpose
    return v


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
    # per-interval records of one long launch: [intervals, C, 2] -> [intervals, world * C, 2]
    stacked = torch.stack([local, local + 100.0])
    wide = parallel.gather_summaries(stacked, world)
    assert wide.shape == (2, C * world, 2)
    assert torch.equal(wide[0], gathered) and torch.equal(wide[1], gathered + 100.0)
    # multi_mcmc::get_average across ranks: base from global chain 0, sums all-reduced
    T, F = 5, 3
    vectors = _chain_vectors(T, F, C * world, seed=4)
    g = _HostChains(vectors[rank * C:(rank + 1) * C].copy(), T, F)
    acc_avg, feat_avg = parallel.average_over_ranks(g, C, rank, world, torch.device("cpu"))
    out[rank] = (mine.tolist(), gathered[:, 0].tolist(), ll, gid, owner, idx, upd.tolist(), acc.tolist(),
                 state.tolist(), acc_avg.numpy().copy(), feat_avg.numpy().copy())
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
        _, lls, ll, gid, owner, idx, upd, acc, state, acc_avg, feat_avg = res[r]
        # the average equals the serial accumulator of average_slam_result.hpp:27-54 over all 8 chains
        T, F = 5, 3
        vectors = _chain_vectors(T, F, 8, seed=4)
        one = _HostChains(vectors, T, F)
        import ctypes
        base = np.zeros(3 * T + 2 * F)
        total = np.zeros_like(base)
        one.average_base(0, base.ctypes.data)
        one.average_sum(base.ctypes.data, 0, total.ctypes.data)
        one.average_finish(base.ctypes.data, total.ctypes.data, 8)
        assert np.allclose(acc_avg.reshape(-1), base[:3 * T], rtol=0, atol=1e-14)
        assert np.allclose(feat_avg.reshape(-1), base[3 * T:], rtol=0, atol=1e-14)
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
