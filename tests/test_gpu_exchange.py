"""The library's own collectives (RCCL behind the C ABI) on a communicator of one rank -- all a one-GPU box can
hold -- against the plain entry points they wrap; and a handle without a communicator, which must behave as
one rank without touching RCCL."""
import numpy as np
import pytest

from slam_amd import capi, synthetic

pytestmark = pytest.mark.gpu


def grown_handle(chains, max_chains=None):
    p = synthetic.make_problem(T=300, n_landmarks=10, k_nearest=3, seed=11, lap_steps=150, wobble_period=500,
                               range_stddev=2.0, landmark_spread=6.0)
    g = capi.McmcGpu(max_chains or chains, p["T"], p["F"], p["N"])
    g.upload_problem(p["ctrl_mean"], p["ctrl_std"], p["obs_t"], p["obs_f"], p["obs_mean"], p["obs_std"], p["F"])
    g.init_chains(capi.seed_sequence(77, chains))
    g.grow(p["T"])
    return g, p


@pytest.mark.parametrize("with_comm", [True, False])
def test_collectives_of_one_rank_equal_the_local_entry_points(with_comm):
    g, p = grown_handle(5, max_chains=8)
    if with_comm:
        g.comm_init(1, 0, capi.comm_unique_id(), slots_per_rank=8)
        assert g.comm_info() == (1, 0, 8)
    g.run_reported(250, 100, None)
    local = g.get_reports()  # [3, 5]
    g.allgather_reports()
    got = g.gathered()
    assert got.shape == (1, 3, 8)
    assert np.array_equal(got[0, :, :5], local)
    assert np.all(np.isneginf(got[0, :, 5:]["log_likelihood"])) and np.all(got[0, :, 5:]["updates"] == 0)
    g.allgather_summaries()
    now = g.gathered()
    assert now.shape == (1, 1, 8)
    assert np.array_equal(now[0, 0, :5]["log_likelihood"], g.log_likelihood())
    assert np.all(now[0, 0, :5]["updates"] == 250)
    # the current summaries went through a row of their own: the reported run's records are still there
    assert np.array_equal(g.get_reports(), local)
    # the best chain's state, broadcast from its owner (here: to itself)
    best = int(np.argmax(g.log_likelihood()))
    g.broadcast_state(0, best)
    rel, est = g.get_broadcast_state()
    st = g.get_state(best)
    assert np.array_equal(rel, st["rel"]) and np.array_equal(est, st["feat_est"])
    # averaging through the three collective phases == mcmcgpu_get_average
    cum, feat = g.average_collective(total_chains=5)
    ref_cum, ref_feat = g.average()
    assert np.array_equal(cum, ref_cum) and np.array_equal(feat, ref_feat)
    g.close()


def test_a_second_communicator_and_too_few_slots_are_refused():
    g, _ = grown_handle(3)
    uid = capi.comm_unique_id()
    with pytest.raises(RuntimeError, match="slots_per_rank"):
        g.comm_init(1, 0, uid, slots_per_rank=2)
    g.comm_init(1, 0, uid, slots_per_rank=3)
    with pytest.raises(RuntimeError, match="already has a communicator"):
        g.comm_init(1, 0, capi.comm_unique_id(), slots_per_rank=3)
    g.close()
