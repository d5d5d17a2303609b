"""The multi-rank path with real handles: a job's result must not depend on how many ranks its chains are
spread over.  multi_mcmc seeds chain i with the i-th draw of ONE generator and steps the chains independently
(src/slam/multi_mcmc.hpp:105-119, 160-166), so global chain i has the same history on 1 x 2C and on 2 x C.
Both ranks share GPU 0 and exchange over gloo (RCCL refuses two ranks on one device); the exchange code is
slam_amd/parallel.py, the same the benchmark runs over RCCL."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "helpers", "multirank_worker.py")


def run_job(world, total_chains, out):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, WORKER, "--total-chains", str(total_chains), "--out", str(out)],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    for p in procs:
        text, _ = p.communicate(timeout=240)
        assert p.returncode == 0, text
    return np.load(out)


@pytest.mark.gpu
def test_result_does_not_depend_on_the_rank_count(tmp_path):
    one = run_job(1, 6, tmp_path / "one.npz")
    two = run_job(2, 6, tmp_path / "two.npz")
    # per-chain histories: identical bit for bit at every report interval, in global chain order
    assert np.array_equal(one["log_likelihood"], two["log_likelihood"])
    assert np.array_equal(one["updates"], two["updates"]) and np.array_equal(one["accepts"], two["accepts"])
    assert one["updates"].shape == (3, 6) and np.all(one["updates"][-1] == 300)
    # the same best chain (first maximum in global chain order) and the same state of it on every rank
    assert np.array_equal(one["best"], two["best"])
    assert np.array_equal(one["winner"], two["winner"])
    # the average over all chains: partial sums per rank are added in another order, nothing more
    np.testing.assert_allclose(two["average_poses"], one["average_poses"], rtol=0, atol=1e-11)
    np.testing.assert_allclose(two["average_landmarks"], one["average_landmarks"], rtol=0, atol=1e-11)


def test_bench_refuses_a_gpu_count_that_disagrees_with_the_launcher():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"], env=env, capture_output=True,
                         text=True, timeout=60)
    assert out.returncode != 0 and "WORLD_SIZE" in out.stderr


@pytest.mark.gpu
def test_bench_starts_its_own_ranks(tmp_path):
    """python bench.py --gpus 2 without a launcher: two child ranks (here both on GPU 0, gloo) and one JSON line
    that says n_gpus = 2."""
    import json

    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--rehearse-on-one-gpu", "--timesteps", "3000", "--landmarks", "40", "--k-nearest", "4",
                          "--chains", "8", "--updates-per-step", "100", "--no-cpu-baseline", "--no-plaza"],
                         capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stdout + out.stderr
    line = [x for x in out.stdout.splitlines() if x.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == 2 and rec["config"]["chains_total"] == 16 and rec["value"] > 0


@pytest.mark.gpu
def test_bench_exchanges_through_rccl_and_prints_one_line():
    """The bench with a process group of one rank on the "nccl" backend: every exchange of a step (all-gather
    of the report records, barrier, broadcast of the winner, all-reduces of the totals) goes through RCCL on
    device buffers, and standard output carries the JSON line and nothing else (RCCL's banner goes to stderr)."""
    import json

    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--collectives-with-one-rank", "--steps", "2",
                          "--warmup", "1", "--timesteps", "3000", "--landmarks", "40", "--k-nearest", "4", "--chains", "8",
                          "--updates-per-step", "100", "--no-cpu-baseline", "--no-plaza"],
                         capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [x for x in out.stdout.splitlines() if x.strip()]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 1 and rec["config"]["chains_total"] == 8 and rec["value"] > 0


def _bench_job(tmp_path, tag, gpus, *extra):
    import json

    dump = tmp_path / (tag + ".npz")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--total-chains", "8192",
                          "--updates-per-step", "1000", "--report-every", "100", "--steps", "1", "--warmup", "0",
                          "--rebuild-every", "0", "--no-cpu-baseline", "--no-plaza", "--no-live-traffic",
                          "--dump-records", str(dump), *extra], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    rec = json.loads([x for x in out.stdout.splitlines() if x.startswith("{")][-1])
    return rec, np.load(dump)


@pytest.mark.gpu
def test_sharded_configuration_at_its_own_shape(tmp_path):
    """The baseline's sharded configuration at its real size: 100 000 poses, 1 000 landmarks, 10^6 readings, 8192
    chains as 8 shards of 1024 (eight ranks that share GPU 0 and exchange over gloo; 53 GB of chain state) against
    ONE rank with all 8192 chains.  One launch of 1000 updates leaves 10 report rows, so the exchange is the
    [10 x 8192] gather of multi_mcmc's report loop (src/slam/multi_mcmc.hpp:121-139).  Global chain i is seeded with
    the i-th draw of one generator (:160-166) and stepped independently (:105-119): every record of every interval,
    the best chain (first maximum in global chain order, strict `>` at :115) and its broadcast state must be
    identical bit for bit whatever the rank count."""
    rec8, eight = _bench_job(tmp_path, "eight", 8, "--rehearse-on-one-gpu")
    rec1, one = _bench_job(tmp_path, "one", 1)
    assert rec8["n_gpus"] == 8 and rec8["config"]["chains_total"] == 8192 and rec8["scaling"] == "strong"
    assert rec8["collective"]["world_size"] == 8 and rec8["collective"]["backend"] == "gloo"
    assert len(rec8["collective"]["kernel_ms_per_rank"]["all"]) == 8
    assert rec1["collective"]["world_size"] == 1
    a = eight["records"].view(capi_summary()).reshape(eight["records"].shape[:-1])
    b = one["records"].view(capi_summary()).reshape(one["records"].shape[:-1])
    assert a.shape == b.shape == (1, 10, 8192)
    assert np.array_equal(a["log_likelihood"], b["log_likelihood"])
    assert np.array_equal(a["updates"], b["updates"]) and np.array_equal(a["accepts"], b["accepts"])
    assert np.all(a["updates"][0, -1] == 1000) and np.all(a["updates"][0, 0] == 100)
    assert np.array_equal(eight["best"], one["best"])
    assert np.array_equal(eight["winner"], one["winner"]) and np.any(one["winner"] != 0)
    assert rec8["best_chain"] == rec1["best_chain"] and rec8["best_log_likelihood"] == rec1["best_log_likelihood"]


def capi_summary():
    sys.path.insert(0, ROOT)
    from slam_amd import capi

    return capi.CHAIN_SUMMARY
