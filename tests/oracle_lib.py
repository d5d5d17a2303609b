"""ctypes wrapper of oracle/liboracle.so -- the CPU restatement of the reference
algorithm.  TEST INFRASTRUCTURE ONLY: imported by tests, smoke() and the
benchmark's cpu_baseline leg, never by the shipped package."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "oracle", "liboracle.so")

RECORD = np.dtype(
    [
        ("kind", np.int32),
        ("accepted", np.int32),
        ("index", np.uint64),
        ("proposed", np.float64, (3,)),
        ("log_ratio", np.float64),
        ("log_likelihood", np.float64),
        ("term_scale", np.float64),  # sum of |log density| over the terms of log_ratio
    ],
    align=True,
)

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
        L = ctypes.CDLL(LIB)
        L.ora_world_new.restype = ctypes.c_void_p
        L.ora_world_free.restype = None
        L.ora_world_error.restype = ctypes.c_char_p
        L.ora_chain_log_likelihood.restype = ctypes.c_double
        L.ora_chain_full_log_likelihood.restype = ctypes.c_double
        L.ora_chain_eval.restype = ctypes.c_double
        L.ora_chain_units.restype = ctypes.c_uint64
        L.ora_chain_min_accept_margin.restype = ctypes.c_double
        L.ora_world_update_all.restype = ctypes.c_uint64
        L.ora_chain_trace_size.restype = ctypes.c_uint32
        for name in ("ora_world_dims", "ora_world_get_input", "ora_chain_trace_enable", "ora_chain_trace_get",
                     "ora_chain_dims", "ora_chain_get_state", "ora_chain_get_weight_nodes", "ora_chain_get_rng",
                     "ora_seed_sequence", "ora_uniform_sequence", "ora_chain_set_beta", "ora_world_average", "ora_world_simulated_truth", "ora_world_get_bearings"):
            getattr(L, name).restype = None
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def seed_sequence(seed, n):
    out = np.empty(n, dtype=np.uint32)
    lib().ora_seed_sequence(ctypes.c_uint32(seed), ctypes.c_uint32(n), _p(out))
    return out


def uniform_sequence(seed, n):
    out = np.empty(n, dtype=np.float64)
    lib().ora_uniform_sequence(ctypes.c_uint32(seed), ctypes.c_uint32(n), _p(out))
    return out


def plaza_controls_rows(plaza_dir, name):
    """The six columns of the reference loader's controls.txt (src/dataset.cpp:96-109), recomputed
    by the oracle from the logs: [T, 6] = dt, actual v, w, g, commanded v, w."""
    n = lib().ora_plaza_controls_rows(plaza_dir.encode(), name.encode(), None, ctypes.c_uint32(0))
    if n < 0:
        raise RuntimeError(f"oracle could not load {name} under {plaza_dir}")
    rows = np.empty((n, 6), dtype=np.float64)
    got = lib().ora_plaza_controls_rows(plaza_dir.encode(), name.encode(), _p(rows), ctypes.c_uint32(n))
    assert got == n
    return rows


class World:
    def __init__(self):
        self.L = lib()
        self.w = ctypes.c_void_p(self.L.ora_world_new())
        self.nchains = 0

    def __del__(self):
        try:
            self.L.ora_world_free(self.w)
        except Exception:
            pass

    def set_input(self, ctrl_mean, ctrl_std, obs_t, obs_id, obs_mean, obs_std):
        cm = np.ascontiguousarray(ctrl_mean, dtype=np.float64).reshape(-1, 3)
        cs = np.ascontiguousarray(ctrl_std, dtype=np.float64).reshape(-1, 3)
        ot = np.ascontiguousarray(obs_t, dtype=np.uint32)
        oi = np.ascontiguousarray(obs_id, dtype=np.uint32)
        om = np.ascontiguousarray(obs_mean, dtype=np.float64)
        os_ = np.ascontiguousarray(obs_std, dtype=np.float64)
        rc = self.L.ora_world_set_input(self.w, ctypes.c_uint32(cm.shape[0]), _p(cm), _p(cs), ctypes.c_uint32(ot.size),
                                        _p(ot), _p(oi), _p(om), _p(os_))
        assert rc == 0

    def set_bearings(self, bearing_mean, bearing_std):
        """Makes every reading of set_input a range-and-bearing observation."""
        bm = np.ascontiguousarray(bearing_mean, dtype=np.float64)
        bs = np.ascontiguousarray(bearing_std, dtype=np.float64)
        assert self.L.ora_world_set_bearings(self.w, ctypes.c_uint32(bm.size), _p(bm), _p(bs)) == 0

    def simulate(self, waypoint_file, landmark_file, seed, repetitions=1.0, dt=0.25, sensor_skip=8, range_stddev=0.1,
                 bearing_stddev=0.0):
        """The reference's simulator as the data source (range-only unless bearing_stddev > 0)."""
        rc = self.L.ora_world_simulate(self.w, waypoint_file.encode(), landmark_file.encode(), ctypes.c_uint32(seed),
                                       ctypes.c_double(repetitions), ctypes.c_double(dt), ctypes.c_uint32(sensor_skip),
                                       ctypes.c_double(range_stddev), ctypes.c_double(bearing_stddev))
        if rc != 0:
            raise RuntimeError(self.L.ora_world_error(self.w).decode())

    def simulated_truth(self):
        T, _ = self.dims()
        init, cum = np.empty(3), np.empty((T + 1, 3))
        self.L.ora_world_simulated_truth(self.w, _p(init), _p(cum))
        return init, cum

    def get_bearings(self):
        _, N = self.dims()
        m, s = np.empty(N), np.empty(N)
        self.L.ora_world_get_bearings(self.w, _p(m), _p(s))
        return m, s

    def load_plaza(self, directory, name, range_stddev):
        rc = self.L.ora_world_load_plaza(self.w, directory.encode(), name.encode(), ctypes.c_double(range_stddev), None)
        if rc != 0:
            raise RuntimeError(self.L.ora_world_error(self.w).decode())

    def dims(self):
        T, N = ctypes.c_uint32(), ctypes.c_uint32()
        self.L.ora_world_dims(self.w, ctypes.byref(T), ctypes.byref(N))
        return T.value, N.value

    def get_input(self):
        T, N = self.dims()
        cm, cs = np.empty((T, 3)), np.empty((T, 3))
        ot, oi = np.empty(N, dtype=np.uint32), np.empty(N, dtype=np.uint32)
        om, os_ = np.empty(N), np.empty(N)
        self.L.ora_world_get_input(self.w, _p(cm), _p(cs), _p(ot), _p(oi), _p(om), _p(os_))
        return {"ctrl_mean": cm, "ctrl_std": cs, "obs_t": ot, "obs_id": oi, "obs_mean": om, "obs_std": os_}

    def set_initialiser(self, seed):
        assert self.L.ora_world_set_initialiser(self.w, ctypes.c_uint32(seed)) == 0

    def add_chain(self, seed, steps_per_timestep=0, end_steps=0, use_initialiser=False, uniform_first=True, trace=False):
        c = self.L.ora_world_add_chain(self.w, ctypes.c_uint32(seed), ctypes.c_uint32(steps_per_timestep),
                                       ctypes.c_uint32(end_steps), int(use_initialiser), int(uniform_first))
        assert c >= 0
        if trace:
            self.L.ora_chain_trace_enable(self.w, c)
        self.nchains = c + 1
        return c

    def add_detached_chain(self, seed, uniform_first=True, trace=False):
        """A chain that does not listen to the data; bring it to life with load_state()."""
        c = self.L.ora_world_add_detached_chain(self.w, ctypes.c_uint32(seed), int(uniform_first))
        if trace:
            self.L.ora_chain_trace_enable(self.w, c)
        self.nchains = c + 1
        return c

    def load_state(self, chain, rel, feat_id, feat_world):
        rel = np.ascontiguousarray(rel, dtype=np.float64)
        fid = np.ascontiguousarray(feat_id, dtype=np.uint32)
        fw = np.ascontiguousarray(feat_world, dtype=np.float64)
        rc = self.L.ora_chain_load_state(self.w, chain, ctypes.c_uint32(rel.shape[0]), _p(rel), ctypes.c_uint32(fid.size),
                                         _p(fid), _p(fw))
        if rc != 0:
            raise RuntimeError(f"ora_chain_load_state failed ({rc}): {self.L.ora_world_error(self.w).decode()}")

    def average(self, T, F):
        """multi_mcmc::get_average over all chains: accumulated poses [T,3], landmarks [F,2] by ascending id."""
        acc, feat = np.empty((T, 3)), np.empty((F, 2))
        self.L.ora_world_average(self.w, _p(acc), _p(feat))
        return acc, feat

    def set_beta(self, chain, beta):
        self.L.ora_chain_set_beta(self.w, chain, ctypes.c_double(beta))

    def replay(self, upto, finish=False):
        rc = self.L.ora_world_replay(self.w, ctypes.c_uint32(upto), int(finish))
        if rc != 0:
            raise RuntimeError(self.L.ora_world_error(self.w).decode())

    def update(self, chain, nsteps):
        return self.L.ora_chain_update(self.w, chain, ctypes.c_uint32(nsteps))

    def update_all(self, nsteps, nthreads=1):
        return self.L.ora_world_update_all(self.w, ctypes.c_uint32(nsteps), ctypes.c_uint32(nthreads))

    def trace(self, chain):
        n = self.L.ora_chain_trace_size(self.w, chain)
        out = np.zeros(n, dtype=RECORD)
        if n:
            self.L.ora_chain_trace_get(self.w, chain, ctypes.c_uint32(0), ctypes.c_uint32(n), _p(out))
        return out

    def log_likelihood(self, chain):
        return self.L.ora_chain_log_likelihood(self.w, chain)

    def full_log_likelihood(self, chain):
        return self.L.ora_chain_full_log_likelihood(self.w, chain)

    def units(self, chain):
        return self.L.ora_chain_units(self.w, chain)

    def min_accept_margin(self, chain):
        """Smallest relative distance of an accept test's two sides over the chain's steps so far."""
        return self.L.ora_chain_min_accept_margin(self.w, chain)

    def state(self, chain):
        T, F = ctypes.c_uint32(), ctypes.c_uint32()
        self.L.ora_chain_dims(self.w, chain, ctypes.byref(T), ctypes.byref(F))
        T, F = T.value, F.value
        st = {
            "rel": np.empty((T, 3)),
            "cum": np.empty((T + 1, 3)),
            "feat_id": np.empty(F, dtype=np.uint32),
            "feat_parent": np.empty(F, dtype=np.uint32),
            "feat_est": np.empty((F, 2)),
            "state_w": np.empty(T),
            "feat_w": np.empty(F),
        }
        self.L.ora_chain_get_state(self.w, chain, _p(st["rel"]), _p(st["cum"]), _p(st["feat_id"]),
                                   _p(st["feat_parent"]), _p(st["feat_est"]), _p(st["state_w"]), _p(st["feat_w"]))
        return st

    def eval(self, chain, kind, index, proposed):
        q = np.zeros(3)
        q[:len(proposed)] = proposed
        return self.L.ora_chain_eval(self.w, chain, int(kind), ctypes.c_uint32(index), _p(q))


def dense_features(obs_id):
    """Landmark ids -> dense indices numbered by first appearance (what the C ABI takes)."""
    obs_id = np.asarray(obs_id)
    _, first = np.unique(obs_id, return_index=True)
    order = obs_id[np.sort(first)]
    remap = {int(v): i for i, v in enumerate(order)}
    return np.array([remap[int(v)] for v in obs_id], dtype=np.uint32), order
