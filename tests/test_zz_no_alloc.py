"""GPU test of mcmcgpu_scan_poses_static, the entry point that allocates nothing: its cumulative
poses against mcmcgpu_rebuild's (bit for bit: same kernels, same plan) and against a sequential
fp64 composition in numpy (1e-9 relative, the tolerance stated for poses), on calls in a row with
shrinking inputs -- all calls share one static scratch array per GPU."""
import numpy as np
import pytest
import torch

from slam_amd import capi, synthetic

# torch brings a HIP runtime of its own: the device pointers below come from torch, so its runtime
# has to be the one this process initialises first (the library then shares it)
torch.cuda.is_available()

pytestmark = pytest.mark.gpu

REL = 1e-9


def sequential_poses(rel):
    """P[k+1] = P[k] o rel[k] in plain fp64, one pose after the other (src/planar_robot/pose.hpp:67-71)."""
    n = rel.shape[0]
    out = np.zeros((n + 1, 3))
    x = y = th = 0.0
    for k in range(n):
        c, s = np.cos(th), np.sin(th)
        x, y, th = x + (c * rel[k, 0] - s * rel[k, 1]), y + (s * rel[k, 0] + c * rel[k, 1]), th + rel[k, 2]
        out[k + 1] = (x, y, th)
    return out


_warm = set()


def warm_up(dev):
    """The first call on a device makes the runtime load the library's code object (and with it the static
    array): that one-time cost is not the call's own, so it is paid before free memory is compared."""
    if dev.index in _warm:
        return
    z = torch.zeros(8, dtype=torch.float64, device=dev)
    capi.scan_poses_static(1, 1, z.data_ptr(), z.data_ptr(), z.data_ptr(), 1, z[2:].data_ptr(), z.data_ptr(), 2,
                           device=dev.index, stream=torch.cuda.current_stream(dev).cuda_stream)
    torch.cuda.synchronize(dev)
    _warm.add(dev.index)


def static_scan(rel_by_chain, pad=0):
    """Runs the entry point on device copies of rel_by_chain [C, n, 3]; returns cum [C, n+1, 3]."""
    C, n, _ = rel_by_chain.shape
    rstride, cstride = n + pad, n + 1 + pad
    dev = torch.device("cuda", 0)
    comp = []
    for k in range(3):
        a = np.zeros((C, rstride))
        a[:, :n] = rel_by_chain[:, :, k]
        comp.append(torch.from_numpy(a).to(dev))
    cum_xy = torch.full((C, cstride, 2), float("nan"), dtype=torch.float64, device=dev)
    cum_th = torch.full((C, cstride), float("nan"), dtype=torch.float64, device=dev)
    warm_up(dev)
    torch.cuda.synchronize(dev)
    # free bytes of the device as the driver counts them (hipMemGetInfo): a hipMalloc or hipFree inside
    # the library would show here, which torch's own allocator statistics cannot see
    before = torch.cuda.mem_get_info(dev)[0]
    capi.scan_poses_static(C, n, comp[0].data_ptr(), comp[1].data_ptr(), comp[2].data_ptr(), rstride,
                           cum_xy.data_ptr(), cum_th.data_ptr(), cstride, device=0,
                           stream=torch.cuda.current_stream(dev).cuda_stream)
    torch.cuda.synchronize(dev)
    assert torch.cuda.mem_get_info(dev)[0] == before, "the call changed the device's free memory"
    out = np.empty((C, n + 1, 3))
    out[:, :, :2] = cum_xy[:, :n + 1].cpu().numpy()
    out[:, :, 2] = cum_th[:, :n + 1].cpu().numpy()
    return out


def rebuilt_by_handle(T, chains, seed):
    """A handle whose chains hold different trajectories; returns (rel [C,T,3], cum [C,T+1,3]) after mcmcgpu_rebuild."""
    p = synthetic.make_problem(T=T, n_landmarks=12, k_nearest=3, seed=seed, lap_steps=400, wobble_period=900,
                               range_stddev=1.0, landmark_spread=8.0)
    g = capi.McmcGpu(chains, p["T"], p["F"], p["N"])
    g.upload_problem(p["ctrl_mean"], p["ctrl_std"], p["obs_t"], p["obs_f"], p["obs_mean"], p["obs_std"], p["F"])
    g.init_chains(capi.seed_sequence(seed, chains))
    rng = np.random.default_rng(seed)
    g.load_state(p["true_rel"], p["landmarks"])
    for c in range(chains):  # every chain its own trajectory
        g.load_state(p["true_rel"] + rng.normal(0.0, [1e-3, 1e-3, 1e-4], size=p["true_rel"].shape), p["landmarks"],
                     first=c, count=1)
    g.rebuild()
    g.sync()
    states = [g.get_state(c) for c in range(chains)]
    g.close()
    return np.stack([s["rel"] for s in states]), np.stack([s["cum"] for s in states])


def test_static_scan_equals_rebuild_on_calls_in_a_row():
    # 2 chains x 20 000 edges: seven pieces per chain scanned in one pass (the scratch holds the ticket counter
    # and the pieces' totals); then 3 x 9 000: three pieces; then 2 x 700: one workgroup per chain, the scratch
    # is not touched
    for T, chains, seed in ((20000, 2, 5), (9000, 3, 6), (700, 2, 7)):
        rel, cum = rebuilt_by_handle(T, chains, seed)
        got = static_scan(rel, pad=3)
        assert np.array_equal(got, cum), f"T={T}: static scan differs from mcmcgpu_rebuild"
        # the single-pass scan composes the pieces' totals in an order that depends on the piece only, not on
        # which workgroup got there first: a second call gives the same bits
        assert np.array_equal(static_scan(rel, pad=3), got), f"T={T}: two calls on the same data differ"
        ref = sequential_poses(rel[0])
        scale = np.maximum(np.maximum(np.abs(got[0]), np.abs(ref)), 1.0)
        assert np.all(np.abs(got[0] - ref) / scale <= REL), f"T={T}: static scan vs sequential composition"


def test_static_scan_long_single_trajectory_then_short():
    rng = np.random.default_rng(3)
    for n in (300_000, 5_000, 1, 0):
        rel = np.zeros((1, n, 3))
        rel[0, :, 0] = 0.75 + rng.normal(0, 0.01, n)
        rel[0, :, 1] = rng.normal(0, 0.01, n)
        rel[0, :, 2] = 0.03 * np.sin(np.arange(n) / 700.0) + rng.normal(0, 1e-3, n)
        got = static_scan(rel, pad=1)[0]
        ref = sequential_poses(rel[0])
        scale = np.maximum(np.maximum(np.abs(got), np.abs(ref)), 1.0)
        assert np.all(np.abs(got - ref) / scale <= REL), f"n={n}"


def test_static_scan_rejects_what_it_cannot_hold():
    dev = torch.device("cuda", 0)
    buf = torch.zeros(64, dtype=torch.float64, device=dev)
    p = buf.data_ptr()
    with pytest.raises(RuntimeError, match="at most 4096 chains"):
        capi.scan_poses_static(4097, 1, p, p, p, 1, p, p, 2)
    with pytest.raises(RuntimeError, match="strides"):
        capi.scan_poses_static(1, 10, p, p, p, 9, p, p, 11)
    with pytest.raises(RuntimeError, match="null"):
        capi.scan_poses_static(1, 10, 0, p, p, 10, p, p, 11)
    with pytest.raises(RuntimeError, match="relative poses per chain"):
        capi.scan_poses_static(1, 0xFFFF0001, p, p, p, 0xFFFF0001, p, p, 0xFFFF0002)
