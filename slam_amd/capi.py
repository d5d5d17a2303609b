"""ctypes binding of the C ABI in include/mcmcgpu.h (libmcmcgpu.so, gfx950).

Plumbing only: numpy arrays in, numpy arrays out.  There is no CPU fallback --
if the library is missing or no MI355X is visible the calls raise.
"""
import ctypes
import os

import numpy as np

_LIB_PATH = os.environ.get("MCMCGPU_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmcmcgpu.so")

STATE_EDGE = 0
FEATURE_EDGE = 1


class Config(ctypes.Structure):
    _fields_ = [
        ("device", ctypes.c_int32),
        ("max_chains", ctypes.c_uint32),
        ("max_timesteps", ctypes.c_uint32),
        ("max_features", ctypes.c_uint32),
        ("max_observations", ctypes.c_uint32),
        ("threads_per_chain", ctypes.c_uint32),
        ("feature_uniform_first", ctypes.c_uint32),
        ("trace_capacity", ctypes.c_uint32),
        ("stream", ctypes.c_void_p),
        ("range_bearing", ctypes.c_uint32),
        ("max_split_slices", ctypes.c_uint32),
        ("split_commit_min_poses", ctypes.c_uint32),
        ("serial_program", ctypes.c_uint32),
    ]


STEP_RECORD = np.dtype(
    [
        ("kind", np.int32),
        ("accepted", np.int32),
        ("index", np.uint64),
        ("proposed", np.float64, (3,)),
        ("log_ratio", np.float64),
        ("log_likelihood", np.float64),
    ],
    align=True,
)

CHAIN_SUMMARY = np.dtype([("log_likelihood", np.float64), ("updates", np.uint32), ("accepts", np.uint32)], align=True)

# Every symbol include/mcmcgpu.h declares (the header is the source of truth;
# tests/test_abi.py checks the two lists against each other).
EXPORTS = [
    "mcmcgpu_last_error",
    "mcmcgpu_create",
    "mcmcgpu_destroy",
    "mcmcgpu_upload_problem",
    "mcmcgpu_append_timestep",
    "mcmcgpu_set_bearings",
    "mcmcgpu_init_chains",
    "mcmcgpu_set_initial_estimates",
    "mcmcgpu_advance",
    "mcmcgpu_grow",
    "mcmcgpu_run",
    "mcmcgpu_run_reported",
    "mcmcgpu_get_reports",
    "mcmcgpu_run_split",
    "mcmcgpu_run_split_reported",
    "mcmcgpu_set_ladders",
    "mcmcgpu_exchange",
    "mcmcgpu_get_ladders",
    "mcmcgpu_eval_llr",
    "mcmcgpu_commit",
    "mcmcgpu_rebuild",
    "mcmcgpu_scan_poses_static",
    "mcmcgpu_get_log_likelihood",
    "mcmcgpu_full_log_likelihood",
    "mcmcgpu_get_state",
    "mcmcgpu_get_dims",
    "mcmcgpu_get_counters",
    "mcmcgpu_get_traffic_counters",
    "mcmcgpu_get_trace",
    "mcmcgpu_write_summaries",
    "mcmcgpu_comm_unique_id",
    "mcmcgpu_comm_init",
    "mcmcgpu_comm_init_all",
    "mcmcgpu_comm_info",
    "mcmcgpu_group_begin",
    "mcmcgpu_group_end",
    "mcmcgpu_allgather_summaries",
    "mcmcgpu_allgather_reports",
    "mcmcgpu_get_gathered",
    "mcmcgpu_broadcast_state",
    "mcmcgpu_get_broadcast_state",
    "mcmcgpu_average_collective",
    "mcmcgpu_average_buffer",
    "mcmcgpu_snapshot_begin",
    "mcmcgpu_snapshot_wait",
    "mcmcgpu_average_base",
    "mcmcgpu_average_sum",
    "mcmcgpu_average_finish",
    "mcmcgpu_get_average",
    "mcmcgpu_export_state",
    "mcmcgpu_import_state",
    "mcmcgpu_load_state",
    "mcmcgpu_sync",
    "mcmcgpu_kernel_time",
    "mcmcgpu_rebuild_time",
    "mcmcgpu_seed_sequence",
    "mcmcgpu_uniform_sequence",
    "mcmcgpu_debug_phase_cycles",
    "mcmcgpu_selftest_scan",
    "mcmcgpu_selftest_range",
    "mcmcgpu_selftest_wave",
]

_lib = None


def load_library():
    """Loads libmcmcgpu.so (built by build.sh / __graft_entry__.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError(f"{_LIB_PATH} is missing: run ./build.sh (hipcc, gfx950) first")
        lib = ctypes.CDLL(_LIB_PATH)
        lib.mcmcgpu_last_error.restype = ctypes.c_char_p
        lib.mcmcgpu_last_error.argtypes = [ctypes.c_void_p]
        lib.mcmcgpu_destroy.restype = None
        lib.mcmcgpu_seed_sequence.restype = None
        lib.mcmcgpu_uniform_sequence.restype = None
        _lib = lib
    return _lib


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def seed_sequence(seed, n):
    out = np.empty(n, dtype=np.uint32)
    load_library().mcmcgpu_seed_sequence(ctypes.c_uint32(seed), ctypes.c_uint32(n), _ptr(out))
    return out


def uniform_sequence(seed, n):
    out = np.empty(n, dtype=np.float64)
    load_library().mcmcgpu_uniform_sequence(ctypes.c_uint32(seed), ctypes.c_uint32(n), _ptr(out))
    return out


def selftest_range(dx, dy, device=0):
    """The device's |(dx, dy)| exactly as the reading stream computes it."""
    lib = load_library()
    dx = np.ascontiguousarray(dx, dtype=np.float64)
    dy = np.ascontiguousarray(dy, dtype=np.float64)
    out = np.empty(dx.size, dtype=np.float64)
    rc = lib.mcmcgpu_selftest_range(ctypes.c_int32(device), ctypes.c_uint32(dx.size), _ptr(dx), _ptr(dy), _ptr(out))
    if rc != 0:
        raise RuntimeError(f"mcmcgpu_selftest_range failed ({rc}): {lib.mcmcgpu_last_error(None).decode()}")
    return out


def selftest_wave(seed, kinds, weights, search_u, device=0):
    """The generator's and the weight tree's one-lane and 64-lane forms side by side; returns
    (draws_serial, draws_wave, index_serial, index_wave)."""
    lib = load_library()
    kinds = np.ascontiguousarray(kinds, dtype=np.uint8)
    weights = np.ascontiguousarray(weights, dtype=np.float64)
    search_u = np.ascontiguousarray(search_u, dtype=np.float64)
    ds, dw = np.zeros(kinds.size), np.zeros(kinds.size)
    i_s, i_w = np.zeros(search_u.size, dtype=np.uint32), np.zeros(search_u.size, dtype=np.uint32)
    rc = lib.mcmcgpu_selftest_wave(ctypes.c_int32(device), ctypes.c_uint32(seed), ctypes.c_uint32(kinds.size), _ptr(kinds),
                                   _ptr(ds), _ptr(dw), ctypes.c_uint32(weights.size), _ptr(weights),
                                   ctypes.c_uint32(search_u.size), _ptr(search_u), _ptr(i_s), _ptr(i_w))
    if rc != 0:
        raise RuntimeError(f"mcmcgpu_selftest_wave failed ({rc}): {lib.mcmcgpu_last_error(None).decode()}")
    return ds, dw, i_s, i_w


def selftest_scan(a, b, p, device=0):
    a = np.ascontiguousarray(a, dtype=np.int32)
    b = np.ascontiguousarray(b, dtype=np.int32)
    oa = np.empty_like(a)
    ob = np.empty_like(b)
    lib = load_library()
    rc = lib.mcmcgpu_selftest_scan(ctypes.c_int32(device), ctypes.c_uint32(a.size), ctypes.c_int32(p), _ptr(a), _ptr(b),
                                   _ptr(oa), _ptr(ob))
    if rc != 0:
        raise RuntimeError(f"mcmcgpu_selftest_scan failed ({rc}): {lib.mcmcgpu_last_error(None).decode()}")
    return oa, ob


def comm_unique_id():
    """ncclGetUniqueId through the library: 128 bytes for mcmcgpu_comm_init on every rank."""
    lib = load_library()
    buf = (ctypes.c_ubyte * 128)()
    rc = lib.mcmcgpu_comm_unique_id(buf)
    if rc != 0:
        raise RuntimeError(f"mcmcgpu_comm_unique_id failed ({rc}): {lib.mcmcgpu_last_error(None).decode()}")
    return bytes(buf)


def scan_poses_static(chains, n, rel_x_ptr, rel_y_ptr, rel_th_ptr, rel_stride, cum_xy_ptr, cum_th_ptr, cum_stride,
                      device=0, stream=None):
    """mcmcgpu_scan_poses_static: cumulative poses from relative poses, device pointers in and out,
    nothing allocated (the scan's partial results live in a static device array of the library)."""
    lib = load_library()
    rc = lib.mcmcgpu_scan_poses_static(ctypes.c_int32(device), ctypes.c_void_p(stream), ctypes.c_uint32(chains),
                                       ctypes.c_uint32(n), ctypes.c_void_p(rel_x_ptr), ctypes.c_void_p(rel_y_ptr),
                                       ctypes.c_void_p(rel_th_ptr), ctypes.c_uint32(rel_stride),
                                       ctypes.c_void_p(cum_xy_ptr), ctypes.c_void_p(cum_th_ptr),
                                       ctypes.c_uint32(cum_stride))
    if rc != 0:
        raise RuntimeError(f"mcmcgpu_scan_poses_static failed ({rc}): {lib.mcmcgpu_last_error(None).decode()}")


class McmcGpu:
    """One handle = C chains on one GPU."""

    def __init__(self, max_chains, max_timesteps, max_features, max_observations, device=0,
                 threads_per_chain=0, feature_uniform_first=True, trace_capacity=0, stream=None, range_bearing=False,
                 max_split_slices=0, split_commit_min_poses=0, serial_program=0):
        self.lib = load_library()
        cfg = Config(device, max_chains, max_timesteps, max_features, max_observations, threads_per_chain,
                     1 if feature_uniform_first else 0, trace_capacity, stream, 1 if range_bearing else 0, max_split_slices,
                     split_commit_min_poses, serial_program)
        self.h = ctypes.c_void_p()
        rc = self.lib.mcmcgpu_create(ctypes.byref(cfg), ctypes.byref(self.h))
        if rc != 0:
            raise RuntimeError(f"mcmcgpu_create failed ({rc}): {self.lib.mcmcgpu_last_error(None).decode()}")
        self.trace_capacity = trace_capacity
        self.chains = 0

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed ({rc}): {self.lib.mcmcgpu_last_error(self.h).decode()}")

    def close(self):
        if self.h:
            self.lib.mcmcgpu_destroy(self.h)
            self.h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload_problem(self, ctrl_mean, ctrl_std, obs_t, obs_f, obs_mean, obs_std, F):
        cm = np.ascontiguousarray(ctrl_mean, dtype=np.float64).reshape(-1, 3)
        cs = np.ascontiguousarray(ctrl_std, dtype=np.float64).reshape(-1, 3)
        ot = np.ascontiguousarray(obs_t, dtype=np.uint32)
        of = np.ascontiguousarray(obs_f, dtype=np.uint32)
        om = np.ascontiguousarray(obs_mean, dtype=np.float64)
        os_ = np.ascontiguousarray(obs_std, dtype=np.float64)
        self._check(
            self.lib.mcmcgpu_upload_problem(self.h, ctypes.c_uint32(cm.shape[0]), _ptr(cm), _ptr(cs),
                                            ctypes.c_uint32(ot.size), _ptr(ot), _ptr(of), _ptr(om), _ptr(os_),
                                            ctypes.c_uint32(F)), "mcmcgpu_upload_problem")

    def append_timestep(self, ctrl_mean=None, ctrl_std=None, obs_f=(), obs_mean=(), obs_std=(), bearing_mean=None,
                        bearing_std=None, init_state_rel=None, init_feat_rel=None):
        """One more timestep: its control (None for timestep 0 of an empty problem) and the readings taken there."""
        def arr(x, dt):
            return None if x is None else np.ascontiguousarray(x, dtype=dt)
        cm, cs = arr(ctrl_mean, np.float64), arr(ctrl_std, np.float64)
        of, om, os_ = arr(obs_f, np.uint32), arr(obs_mean, np.float64), arr(obs_std, np.float64)
        bm, bs = arr(bearing_mean, np.float64), arr(bearing_std, np.float64)
        ir, if_ = arr(init_state_rel, np.float64), arr(init_feat_rel, np.float64)
        self._check(
            self.lib.mcmcgpu_append_timestep(self.h, _ptr(cm), _ptr(cs), ctypes.c_uint32(of.size), _ptr(of), _ptr(om),
                                             _ptr(os_), _ptr(bm), _ptr(bs), _ptr(ir), _ptr(if_)),
            "mcmcgpu_append_timestep")

    def set_bearings(self, bearing_mean, bearing_std):
        """The bearing half of every reading (arrival order); range-and-bearing handles, after every upload."""
        bm = np.ascontiguousarray(bearing_mean, dtype=np.float64)
        bs = np.ascontiguousarray(bearing_std, dtype=np.float64)
        self._check(self.lib.mcmcgpu_set_bearings(self.h, ctypes.c_uint32(bm.size), _ptr(bm), _ptr(bs)),
                    "mcmcgpu_set_bearings")

    def init_chains(self, seeds, init_uniforms=None):
        seeds = np.ascontiguousarray(seeds, dtype=np.uint32)
        iu = None if init_uniforms is None else np.ascontiguousarray(init_uniforms, dtype=np.float64)
        self._check(
            self.lib.mcmcgpu_init_chains(self.h, ctypes.c_uint32(seeds.size), _ptr(seeds), _ptr(iu),
                                         ctypes.c_uint32(0 if iu is None else iu.size)), "mcmcgpu_init_chains")
        self.chains = int(seeds.size)

    def set_initial_estimates(self, state_rel=None, feat_rel=None):
        sr = None if state_rel is None else np.ascontiguousarray(state_rel, dtype=np.float64).reshape(-1, 3)
        fr = None if feat_rel is None else np.ascontiguousarray(feat_rel, dtype=np.float64).reshape(-1, 2)
        self._check(
            self.lib.mcmcgpu_set_initial_estimates(self.h, ctypes.c_uint32(0 if sr is None else sr.shape[0]), _ptr(sr),
                                                   ctypes.c_uint32(0 if fr is None else fr.shape[0]), _ptr(fr)),
            "mcmcgpu_set_initial_estimates")

    def advance(self, upto, steps_per_timestep=0, end_steps=0):
        self._check(
            self.lib.mcmcgpu_advance(self.h, ctypes.c_uint32(upto), ctypes.c_uint32(steps_per_timestep),
                                     ctypes.c_uint32(end_steps)), "mcmcgpu_advance")

    def grow(self, upto):
        self._check(self.lib.mcmcgpu_grow(self.h, ctypes.c_uint32(upto)), "mcmcgpu_grow")

    def run(self, steps):
        self._check(self.lib.mcmcgpu_run(self.h, ctypes.c_uint32(steps)), "mcmcgpu_run")

    def run_reported(self, steps, report_every, dev_ptr):
        """`steps` updates per chain in one launch; summaries [ceil(steps/report_every)][C] to device memory."""
        self._check(
            self.lib.mcmcgpu_run_reported(self.h, ctypes.c_uint32(steps), ctypes.c_uint32(report_every),
                                          ctypes.c_void_p(dev_ptr) if dev_ptr else None), "mcmcgpu_run_reported")

    def run_split(self, steps, slices):
        """`steps` updates per chain, each proposal's reading stream cut into `slices` workgroups."""
        self._check(self.lib.mcmcgpu_run_split(self.h, ctypes.c_uint32(steps), ctypes.c_uint32(slices)), "mcmcgpu_run_split")

    def run_split_reported(self, steps, slices, report_every, dev_ptr):
        """run_split that leaves the summaries of run_reported at every interval boundary."""
        self._check(
            self.lib.mcmcgpu_run_split_reported(self.h, ctypes.c_uint32(steps), ctypes.c_uint32(slices),
                                                ctypes.c_uint32(report_every),
                                                ctypes.c_void_p(dev_ptr) if dev_ptr else None),
            "mcmcgpu_run_split_reported")

    def get_reports(self):
        """Records of the last run_reported(..., dev_ptr=None): structured array [intervals, C]."""
        n = ctypes.c_uint32()
        self._check(self.lib.mcmcgpu_get_reports(self.h, None, ctypes.c_uint32(0), ctypes.byref(n)), "mcmcgpu_get_reports")
        out = np.zeros((n.value, self.chains), dtype=CHAIN_SUMMARY)
        self._check(self.lib.mcmcgpu_get_reports(self.h, _ptr(out), ctypes.c_uint32(n.value), ctypes.byref(n)),
                    "mcmcgpu_get_reports")
        return out

    def set_ladders(self, betas, exchange_seed):
        """Tempered replicas: ladders of len(betas) consecutive chains, slot s at inverse temperature betas[s]."""
        b = np.ascontiguousarray(betas, dtype=np.float64)
        self._check(self.lib.mcmcgpu_set_ladders(self.h, ctypes.c_uint32(b.size), _ptr(b), ctypes.c_uint32(exchange_seed)),
                    "mcmcgpu_set_ladders")

    def exchange(self, phase):
        self._check(self.lib.mcmcgpu_exchange(self.h, ctypes.c_uint32(phase)), "mcmcgpu_exchange")

    def ladders(self):
        """(beta per chain, chain per temperature slot, readings' log likelihood per chain, attempted swaps,
        accepted swaps)."""
        beta = np.empty(self.chains, dtype=np.float64)
        slot = np.empty(self.chains, dtype=np.uint32)
        rll = np.empty(self.chains, dtype=np.float64)
        att, swp = ctypes.c_uint64(), ctypes.c_uint64()
        self._check(self.lib.mcmcgpu_get_ladders(self.h, _ptr(beta), _ptr(slot), _ptr(rll), ctypes.byref(att),
                                                 ctypes.byref(swp)), "mcmcgpu_get_ladders")
        return beta, slot, rll, att.value, swp.value

    # ---- exchange between handles (RCCL inside the library) ----
    def comm_init(self, nranks, rank, unique_id, slots_per_rank):
        """ncclCommInitRank for this handle; unique_id = the 128 bytes of comm_unique_id() made on rank 0."""
        buf = (ctypes.c_ubyte * 128).from_buffer_copy(bytes(unique_id))
        self._check(self.lib.mcmcgpu_comm_init(self.h, ctypes.c_int(nranks), ctypes.c_int(rank), buf,
                                               ctypes.c_uint32(slots_per_rank)), "mcmcgpu_comm_init")

    def comm_info(self):
        n, r, s = ctypes.c_int(), ctypes.c_int(), ctypes.c_uint32()
        self._check(self.lib.mcmcgpu_comm_info(self.h, ctypes.byref(n), ctypes.byref(r), ctypes.byref(s)), "mcmcgpu_comm_info")
        return n.value, r.value, s.value

    def allgather_summaries(self):
        self._check(self.lib.mcmcgpu_allgather_summaries(self.h), "mcmcgpu_allgather_summaries")

    def allgather_reports(self):
        self._check(self.lib.mcmcgpu_allgather_reports(self.h), "mcmcgpu_allgather_reports")

    def gathered(self):
        """What the last all-gather brought: structured array [ranks, intervals, slots]."""
        r, k, s = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint32()
        self._check(self.lib.mcmcgpu_get_gathered(self.h, None, ctypes.c_uint32(0), ctypes.byref(r), ctypes.byref(k),
                                                  ctypes.byref(s)), "mcmcgpu_get_gathered")
        out = np.zeros((r.value, k.value, s.value), dtype=CHAIN_SUMMARY)
        self._check(self.lib.mcmcgpu_get_gathered(self.h, _ptr(out), ctypes.c_uint32(out.size), ctypes.byref(r),
                                                  ctypes.byref(k), ctypes.byref(s)), "mcmcgpu_get_gathered")
        return out

    def broadcast_state(self, root, chain):
        self._check(self.lib.mcmcgpu_broadcast_state(self.h, ctypes.c_int(root), ctypes.c_uint32(chain)),
                    "mcmcgpu_broadcast_state")

    def get_broadcast_state(self):
        T, F, _ = self.dims()
        rel, est = np.empty((T, 3)), np.empty((F, 2))
        self._check(self.lib.mcmcgpu_get_broadcast_state(self.h, _ptr(rel), _ptr(est)), "mcmcgpu_get_broadcast_state")
        return rel, est

    def average_collective(self, total_chains):
        """All three phases in a row: right for one handle per process (or a single handle)."""
        T, F, _ = self.dims()
        cum, feat = np.empty((T, 3)), np.empty((F, 2))
        for phase in (0, 1, 2):
            self._check(self.lib.mcmcgpu_average_collective(self.h, ctypes.c_int(phase), ctypes.c_uint32(total_chains),
                                                            _ptr(cum), _ptr(feat)), "mcmcgpu_average_collective")
        return cum, feat

    def snapshot_begin(self, chain):
        """Queue a copy of one chain's estimate to pinned host memory; later launches may be queued behind it."""
        self._check(self.lib.mcmcgpu_snapshot_begin(self.h, ctypes.c_uint32(chain)), "mcmcgpu_snapshot_begin")

    def snapshot_wait(self, max_timesteps, max_features):
        """(relative poses [T,3], world landmarks [F',2], log likelihood) of the snapshot begun last."""
        rel = np.empty((max_timesteps, 3), dtype=np.float64)
        world = np.empty((max_features, 2), dtype=np.float64)
        T, F, ll = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_double()
        self._check(self.lib.mcmcgpu_snapshot_wait(self.h, ctypes.byref(T), _ptr(rel), ctypes.byref(F), _ptr(world),
                                                   ctypes.byref(ll)), "mcmcgpu_snapshot_wait")
        return rel[:T.value].copy(), world[:F.value].copy(), ll.value

    def average_base(self, chain, dev_ptr):
        self._check(self.lib.mcmcgpu_average_base(self.h, ctypes.c_uint32(chain), ctypes.c_void_p(dev_ptr)),
                    "mcmcgpu_average_base")

    def average_sum(self, base_ptr, skip_chain, sum_ptr):
        self._check(self.lib.mcmcgpu_average_sum(self.h, ctypes.c_void_p(base_ptr), ctypes.c_int32(skip_chain),
                                                 ctypes.c_void_p(sum_ptr)), "mcmcgpu_average_sum")

    def average_finish(self, base_ptr, sum_ptr, total_chains):
        self._check(self.lib.mcmcgpu_average_finish(self.h, ctypes.c_void_p(base_ptr), ctypes.c_void_p(sum_ptr),
                                                    ctypes.c_uint32(total_chains)), "mcmcgpu_average_finish")

    def average(self):
        """multi_mcmc::get_average over this handle's chains: (accumulated poses [T,3], landmarks [F',2])."""
        T, F, _ = self.dims()
        cum = np.empty((T, 3), dtype=np.float64)
        feat = np.empty((F, 2), dtype=np.float64)
        self._check(self.lib.mcmcgpu_get_average(self.h, _ptr(cum), _ptr(feat)), "mcmcgpu_get_average")
        return cum, feat

    def eval_llr(self, chain, kind, index, proposed):
        q = np.zeros(3, dtype=np.float64)
        q[:len(proposed)] = proposed
        out = ctypes.c_double()
        self._check(
            self.lib.mcmcgpu_eval_llr(self.h, ctypes.c_uint32(chain), ctypes.c_int(kind), ctypes.c_uint32(index),
                                      _ptr(q), ctypes.byref(out)), "mcmcgpu_eval_llr")
        return out.value

    def commit(self, chain, kind, index, proposed, new_weight, log_ratio):
        q = np.zeros(3, dtype=np.float64)
        q[:len(proposed)] = proposed
        self._check(
            self.lib.mcmcgpu_commit(self.h, ctypes.c_uint32(chain), ctypes.c_int(kind), ctypes.c_uint32(index), _ptr(q),
                                    ctypes.c_double(new_weight), ctypes.c_double(log_ratio)), "mcmcgpu_commit")

    def rebuild(self):
        self._check(self.lib.mcmcgpu_rebuild(self.h), "mcmcgpu_rebuild")

    def log_likelihood(self):
        out = np.empty(self.chains, dtype=np.float64)
        self._check(self.lib.mcmcgpu_get_log_likelihood(self.h, _ptr(out)), "mcmcgpu_get_log_likelihood")
        return out

    def full_log_likelihood(self):
        out = np.empty(self.chains, dtype=np.float64)
        self._check(self.lib.mcmcgpu_full_log_likelihood(self.h, _ptr(out)), "mcmcgpu_full_log_likelihood")
        return out

    def dims(self):
        t, f, c = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint32()
        self._check(self.lib.mcmcgpu_get_dims(self.h, ctypes.byref(t), ctypes.byref(f), ctypes.byref(c)),
                    "mcmcgpu_get_dims")
        return t.value, f.value, c.value

    def get_state(self, chain):
        T, F, _ = self.dims()
        st = {
            "rel": np.empty((T, 3)),
            "cum": np.empty((T + 1, 3)),
            "feat_parent": np.empty(F, dtype=np.uint32),
            "feat_est": np.empty((F, 2)),
            "feat_world": np.empty((F, 2)),
            "state_w": np.empty(T),
            "feat_w": np.empty(F),
        }
        self._check(
            self.lib.mcmcgpu_get_state(self.h, ctypes.c_uint32(chain), _ptr(st["rel"]), _ptr(st["cum"]),
                                       _ptr(st["feat_parent"]), _ptr(st["feat_est"]), _ptr(st["feat_world"]),
                                       _ptr(st["state_w"]), _ptr(st["feat_w"])), "mcmcgpu_get_state")
        return st

    def counters(self):
        u = np.empty(self.chains, dtype=np.uint64)
        a = np.empty(self.chains, dtype=np.uint64)
        n = np.empty(self.chains, dtype=np.uint64)
        self._check(self.lib.mcmcgpu_get_counters(self.h, _ptr(u), _ptr(a), _ptr(n)), "mcmcgpu_get_counters")
        return u, a, n

    def traffic_counters(self):
        s = np.empty(self.chains, dtype=np.uint64)
        m = np.empty(self.chains, dtype=np.uint64)
        self._check(self.lib.mcmcgpu_get_traffic_counters(self.h, _ptr(s), _ptr(m)), "mcmcgpu_get_traffic_counters")
        return s, m

    def trace(self, chain):
        out = np.zeros(max(self.trace_capacity, 1), dtype=STEP_RECORD)
        n = ctypes.c_uint32()
        self._check(
            self.lib.mcmcgpu_get_trace(self.h, ctypes.c_uint32(chain), _ptr(out), ctypes.c_uint32(self.trace_capacity),
                                       ctypes.byref(n)), "mcmcgpu_get_trace")
        return out[:n.value]

    def write_summaries(self, dev_ptr):
        self._check(self.lib.mcmcgpu_write_summaries(self.h, ctypes.c_void_p(dev_ptr)), "mcmcgpu_write_summaries")

    def export_state(self, chain, dev_ptr):
        self._check(self.lib.mcmcgpu_export_state(self.h, ctypes.c_uint32(chain), ctypes.c_void_p(dev_ptr)),
                    "mcmcgpu_export_state")

    def import_state(self, chain, dev_ptr, log_likelihood):
        self._check(
            self.lib.mcmcgpu_import_state(self.h, ctypes.c_uint32(chain), ctypes.c_void_p(dev_ptr),
                                          ctypes.c_double(log_likelihood)), "mcmcgpu_import_state")

    def load_state(self, rel_poses, feat_world, first=0, count=None):
        rel = np.ascontiguousarray(rel_poses, dtype=np.float64)
        fw = np.ascontiguousarray(feat_world, dtype=np.float64)
        count = self.chains - first if count is None else count
        self._check(
            self.lib.mcmcgpu_load_state(self.h, ctypes.c_uint32(first), ctypes.c_uint32(count), _ptr(rel), _ptr(fw)),
            "mcmcgpu_load_state")

    def debug_phase_cycles(self, reset=True):
        out = np.zeros(16, dtype=np.uint64)
        self._check(self.lib.mcmcgpu_debug_phase_cycles(self.h, _ptr(out), ctypes.c_int(1 if reset else 0)),
                    "mcmcgpu_debug_phase_cycles")
        return out

    def sync(self):
        self._check(self.lib.mcmcgpu_sync(self.h), "mcmcgpu_sync")

    def kernel_time(self):
        ms, n = ctypes.c_double(), ctypes.c_uint32()
        self._check(self.lib.mcmcgpu_kernel_time(self.h, ctypes.byref(ms), ctypes.byref(n)), "mcmcgpu_kernel_time")
        return ms.value, n.value

    def rebuild_time(self):
        ms, n = ctypes.c_double(), ctypes.c_uint32()
        self._check(self.lib.mcmcgpu_rebuild_time(self.h, ctypes.byref(ms), ctypes.byref(n)), "mcmcgpu_rebuild_time")
        return ms.value, n.value
