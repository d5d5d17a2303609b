#!/usr/bin/env python3
"""Benchmark of the MCMC-SLAM proposal-evaluation path on MI355X.

Metric: evaluated MCMC proposals*observations per second -- one
unit is one execution of the reference's inner loop body
(src/slam/mcmc_slam.hpp:487-502), i.e. one range reading evaluated for one
proposal of one chain -- summed over all chains of all GPUs of the node.

Workload (per GPU): synthetic planar robot, 100k poses,
1k landmarks, 1M range readings, 1024 independent chains.  A bench "step" is one
launch of the on-device Metropolis-Hastings loop doing `--updates-per-step`
updates (mcmc_slam::update) on every chain; the launch leaves one summary record
per chain per `--report-every` updates (multi_mcmc's report interval), and the
step ends with the exchange of those records (all-gather over RCCL when N > 1)
and the best-chain maximum per interval.  N GPUs run N x 1024 chains (weak scaling), one process per GPU.

Launch: python bench.py [--gpus N --steps K --warmup W].  For N > 1 either under
python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ... (the ranks
read RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment), or plainly as
python bench.py --gpus N: the process then starts the N ranks itself, as fresh child
processes, before it has touched a GPU.  --gpus that disagrees with WORLD_SIZE is an error.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak
BYTES_PER_UNIT = 32  # pose translation 16 B + range 8 B + timestep 4 B + landmark id 4 B
BYTES_PER_LANDMARK = 20  # estimate 16 B + parent 4 B, read once per state-edge proposal
BYTES_PER_MOVED_POSE = 48  # read + write of one 24 B cumulative pose on an accepted state edge


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=["chains", "tempering", "range-bearing", "commit-heavy", "burn-in"], default="chains",
                    help="chains: the headline workload; tempering: one ladder of "
                         "tempered replicas per GPU on a 1M-pose trajectory with a replica exchange after every launch; "
                         "range-bearing: the headline workload with the reference's other observation model (48 B per unit); "
                         "commit-heavy: 10 000 poses with few, noisy readings per pose (2 nearest landmarks, sigma 3 m), "
                         "the regime of the Plaza logs: an eighth of the proposals is accepted and every accepted state "
                         "edge re-positions all later poses, a quarter of the algorithmic bytes; "
                         "burn-in: the headline problem with every chain heated to --beta (readings' part of the accept "
                         "test weighed by it, src/slam/mcmc_slam.hpp:408 otherwise), so that a large share of the "
                         "100 000-pose proposals is accepted: what the commit path costs where it can hurt")
    ap.add_argument("--beta", type=float, default=1e-3, help="burn-in: inverse temperature of every chain")
    ap.add_argument("--range-stddev", type=float, default=None, help="sigma of the range readings, m (1; commit-heavy: 3)")
    ap.add_argument("--bearing-stddev", type=float, default=0.05, help="range-bearing: sigma of the bearings, rad")
    ap.add_argument("--chains", type=int, default=None, help="chains per GPU (1024; tempering: 64)")
    ap.add_argument("--total-chains", type=int, default=0,
                    help="chains of the whole job, divided over the GPUs (strong scaling: 8192 over 2/4/8 GPUs is "
                         "the sharded configuration of the baseline); overrides --chains")
    ap.add_argument("--updates-per-step", type=int, default=None,
                    help="updates per chain per launch (1000; tempering: 200)")
    ap.add_argument("--report-every", type=int, default=100,
                    help="updates between per-chain summary records; 100 = the report interval of multi_mcmc.hpp:125")
    ap.add_argument("--timesteps", type=int, default=None, help="poses (100 000; tempering: 1 000 000)")
    ap.add_argument("--beta-min", type=float, default=0.01, help="tempering: coldest..hottest = 1..beta_min, geometric")
    ap.add_argument("--landmarks", type=int, default=1000)
    ap.add_argument("--k-nearest", type=int, default=None, help="landmarks ranged per timestep (10; commit-heavy: 2)")
    ap.add_argument("--seed", type=int, default=12345)
    ap.add_argument("--rebuild-every", type=int, default=1, help="steps between cumulative-pose rebuilds")
    ap.add_argument("--threads-per-chain", type=int, default=0)
    ap.add_argument("--serial-program", type=int, default=0, choices=[0, 1, 2, 3],
                    help="mcmcgpu_config.serial_program: 0 = the library's choice, 1 = called, 2 = compiled into the kernel, 3 = compiled in and selecting ahead")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target duration of the CPU baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-plaza", action="store_true")
    ap.add_argument("--split-slices", type=int, default=0,
                    help="run every update as a sequence of kernels with each proposal's reading stream cut into "
                         "this many workgroups (mcmcgpu_run_split): for few, long chains")
    ap.add_argument("--collectives-with-one-rank", action="store_true",
                    help="testing aid: with one rank, still set up an RCCL process group of that one rank and "
                         "run every exchange through it")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not run the two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE) over a short child run of "
                         "this workload; roofline.traffic then comes from profiles/hbm_traffic*.json when that was taken on "
                         "the same library, else it is null")
    ap.add_argument("--live-traffic-seconds", type=float, default=150.0,
                    help="time allowed for the two counter passes together; a pass that would not fit is not started")
    ap.add_argument("--keep-counter-csv", default=None,
                    help="prefix for copies of the counter passes' rows (the dominant kernel's dispatches only): "
                         "<prefix>_pmc_fetch_size.csv and <prefix>_pmc_write_size.csv, what roofline.traffic was summed from")
    ap.add_argument("--dump-records", default=None,
                    help="rank 0 writes what the job as a whole produced (per-interval chain records of every timed step, "
                         "best chain, the winner's broadcast state) to this .npz file")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="testing aid: all ranks share GPU 0 and exchange over gloo (RCCL refuses two ranks on one "
                         "device); exercises the N > 1 code path on a one-GPU box, its numbers mean nothing")
    args = ap.parse_args()
    tempering = args.workload == "tempering"
    heavy = args.workload == "commit-heavy"
    if args.k_nearest is None:
        args.k_nearest = 2 if heavy else 10
    if args.range_stddev is None:
        args.range_stddev = 3.0 if heavy else 1.0
    if heavy or args.workload == "burn-in":
        args.no_cpu_baseline = args.no_plaza = True
    if args.chains is None:
        args.chains = 64 if tempering else 1024
    if args.updates_per_step is None:
        args.updates_per_step = 200 if tempering else 1000
    if args.timesteps is None:
        args.timesteps = 1_000_000 if tempering else 10_000 if heavy else 100_000
    if args.workload == "range-bearing":
        args.no_cpu_baseline = args.no_plaza = True
    if tempering:
        if args.threads_per_chain == 0:
            args.threads_per_chain = 1024  # few, long chains: the widest workgroup per chain
        args.no_cpu_baseline = args.no_plaza = True
    return args


def library_sha16():
    """Identifies the library's code: a digest of its sources (slam_amd/csrc, include/mcmcgpu.h).  Profiles quote
    it; bench.py quotes a profile's counter figures only when they were taken on the same code."""
    import glob
    import hashlib

    h = hashlib.sha256()
    for path in sorted(glob.glob(os.path.join(ROOT, "slam_amd", "csrc", "*")) + [os.path.join(ROOT, "include", "mcmcgpu.h")]):
        with open(path, "rb") as f:
            h.update(os.path.basename(path).encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def cpu_baseline(problem, seed, target_seconds, all_seeds=None):
    """The CPU oracle (restated reference algorithm, one thread, one chain) on a bounded
    sample of the same workload: same problem, same starting state, same chain seed."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib

    w = oracle_lib.World()
    w.set_input(problem["ctrl_mean"], problem["ctrl_std"], problem["obs_t"], problem["obs_f"], problem["obs_mean"],
                problem["obs_std"])
    w.replay(problem["T"])
    c = w.add_detached_chain(int(seed))
    w.load_state(c, problem["true_rel"], np.arange(problem["F"]), problem["landmarks"])
    probe = 8
    t0 = time.perf_counter()
    w.update(c, probe)
    dt = max(time.perf_counter() - t0, 1e-6)
    n = int(max(8, min(5000, target_seconds / (dt / probe))))
    u0 = w.units(c)
    t0 = time.perf_counter()
    w.update(c, n)
    dt = time.perf_counter() - t0
    units = w.units(c) - u0
    out = {
        "value": units / dt,
        "unit": "proposals*observations/s",
        "cores": 1,
        "kind": "port",
        "sample": f"1 chain x {n} updates ({units} units, {dt:.1f} s) from the GPU run's starting state",
    }
    # the same oracle with one chain per host thread (chains are independent): what the box's cores
    # give when the reference's serial multi_mcmc loop (multi_mcmc.hpp:109-118) is simply parallelised
    # every core this process may run on (the scheduler's affinity mask, which is what a container's share of
    # the host shows as): not capped
    try:
        cores = max(1, len(os.sched_getaffinity(0)))
    except AttributeError:
        cores = max(1, os.cpu_count() or 1)
    if cores > 1 and all_seeds is not None:
        chains = [c] + [w.add_detached_chain(int(s)) for s in all_seeds[1:cores]]
        for k in chains[1:]:
            w.load_state(k, problem["true_rel"], np.arange(problem["F"]), problem["landmarks"])
        per_chain = max(8, int(n * min(1.0, target_seconds / max(dt, 1e-6))))
        t0 = time.perf_counter()
        many = w.update_all(per_chain, nthreads=cores)
        dt_all = time.perf_counter() - t0
        out["all_cores"] = {
            "value": many / dt_all,
            "cores": cores,
            "host_logical_cpus": os.cpu_count(),
            "sample": f"{len(chains)} chains x {per_chain} updates on {cores} threads = every core this process may use "
                      f"({many} units, {dt_all:.1f} s)",
        }
    return out


def child_argv(args):
    """The command line of a short run of the same workload (the counter passes below profile it)."""
    argv = [os.path.abspath(__file__), "--workload", args.workload, "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
            "--no-plaza", "--no-live-traffic", "--chains", str(args.chains), "--updates-per-step", str(args.updates_per_step),
            "--report-every", str(args.report_every), "--timesteps", str(args.timesteps), "--landmarks", str(args.landmarks),
            "--k-nearest", str(args.k_nearest), "--seed", str(args.seed), "--rebuild-every", str(args.rebuild_every),
            "--threads-per-chain", str(args.threads_per_chain), "--serial-program", str(args.serial_program), "--split-slices", str(args.split_slices),
            "--range-stddev", repr(args.range_stddev), "--bearing-stddev", repr(args.bearing_stddev),
            "--beta-min", repr(args.beta_min), "--beta", repr(args.beta)]
    return argv, 3  # launches (steps + warm-up) of that run


def live_traffic(args, split_mode, budget_s=150.0):
    """HBM-side bytes per launch of the dominant kernel, measured now: two rocprofv3 --pmc passes (FETCH_SIZE,
    WRITE_SIZE; one counter each, they do not fit one pass) over a short child run of this very workload on the
    library this process runs.  FETCH_SIZE and WRITE_SIZE are reported in KiB; gfx950 tallies a 128-byte read
    request at 64 bytes, and every read this kernel sends out of L2 is a 128-byte one, so reads count double.
    Requests leaving L2 are counted, Infinity-Cache hits included.  Returns (bytes per launch or None, record)."""
    import csv
    import glob
    import shutil
    import signal
    import subprocess
    import tempfile

    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    rec = {"method": "rocprofv3 --pmc, one pass per counter, over a child run of this workload (2 timed + 1 warm-up launches)"}
    if not os.path.exists(prof):
        rec["failed"] = "rocprofv3 not found"
        return None, rec
    argv, launches = child_argv(args)
    names = ("split_",) if split_mode else ("advance_kernel",)
    got = {}
    t_begin = time.perf_counter()
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        out = tempfile.mkdtemp(prefix="bench_pmc_", dir="/tmp")
        cmd = [prof, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", out, "--", sys.executable] + argv
        env = dict(os.environ, TMPDIR="/tmp")
        for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
            env.pop(k, None)
        try:
            left = budget_s - (time.perf_counter() - t_begin)
            if left < 20:
                raise TimeoutError("time budget of the counter passes used up")
            proc = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                                    start_new_session=True)
            try:
                proc.wait(timeout=left)
            except subprocess.TimeoutExpired:
                os.killpg(proc.pid, signal.SIGKILL)  # the process group this call started, nothing else
                proc.wait()
                raise TimeoutError(f"{counter} pass did not finish in {left:.0f} s")
            if proc.returncode != 0:
                raise RuntimeError(f"{counter} pass ended with code {proc.returncode}")
            files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            if not files:
                raise RuntimeError(f"{counter} pass left no counter_collection.csv")
            total, rows, kept = 0.0, 0, []
            with open(files[0]) as f:
                reader = csv.DictReader(f)
                for r in reader:
                    if r["Counter_Name"] == counter and any(n in r["Kernel_Name"] for n in names):
                        total += float(r["Counter_Value"])
                        rows += 1
                        kept.append(r)
            if args.keep_counter_csv and kept:
                with open(f"{args.keep_counter_csv}_pmc_{counter.lower()}.csv", "w", newline="") as f:
                    if len(kept) <= 64:
                        w = csv.DictWriter(f, fieldnames=reader.fieldnames)
                        w.writeheader()
                        w.writerows(kept)
                    else:  # split mode: thousands of dispatches -- one row per kernel
                        per = {}
                        for r in kept:
                            n, v = per.get(r["Kernel_Name"], (0, 0.0))
                            per[r["Kernel_Name"]] = (n + 1, v + float(r["Counter_Value"]))
                        w = csv.writer(f)
                        w.writerow(["Kernel_Name", "Counter_Name", "Dispatches", "Counter_Value_Sum", "Launches_Of_The_Run"])
                        for name, (n, v) in sorted(per.items()):
                            w.writerow([name, counter, n, v, launches])
            if rows == 0:
                raise RuntimeError(f"{counter} pass saw none of the kernels {names}")
            got[counter] = total * 1024.0 / launches
            rec[counter.lower() + "_reported_bytes_per_launch"] = got[counter]
            rec[counter.lower() + "_dispatches"] = rows
        except Exception as e:  # noqa: BLE001 -- the bench line must come out whatever the profiler does
            rec["failed"] = f"{type(e).__name__}: {e}"
            return None, rec
        finally:
            shutil.rmtree(out, ignore_errors=True)
    rec["seconds"] = time.perf_counter() - t_begin
    rec["counters"] = "2 x FETCH_SIZE + WRITE_SIZE: requests leaving L2, Infinity-Cache hits included"
    return 2.0 * got["FETCH_SIZE"] + got["WRITE_SIZE"], rec


def quoted_traffic(args, workload_name):
    """roofline.traffic from the last committed profiling round (profiles/hbm_traffic*.json), only when that was
    taken on the library that is running now."""
    import glob

    running_sha = library_sha16()
    traffic, source = None, None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "hbm_traffic*.json"))):
        try:
            with open(path) as f:
                rec = json.load(f)
            if rec["updates_per_step"] != args.updates_per_step or rec["workload"] != workload_name:
                continue
            source = {"method": "quoted from " + os.path.relpath(path, ROOT), "date": rec.get("date"),
                      "commit": rec.get("commit"), "library_sha16": rec.get("library_sha16"),
                      "running_library_sha16": running_sha,
                      "counters": "2 x FETCH_SIZE + WRITE_SIZE: requests leaving L2, Infinity-Cache hits included"}
            if rec.get("library_sha16") == running_sha:
                traffic = rec["traffic_bytes_per_launch"]
            else:
                source["withheld"] = "the profile was taken on another build of the library"
        except (OSError, ValueError, KeyError):
            pass
    return traffic, source


def hbm_only_bracket(workload_name, updates_per_step, units_per_launch, traffic):
    """Which part of `traffic` (requests leaving L2) goes on to HBM and which the 256 MB Infinity Cache serves
    cannot be counted on this GPU (rocprofv3 lists no counter beyond the L2's memory-side interface), so it is
    bracketed from the window probe (tools/window_probe.sh -> profiles/window_probe.json): the reading records
    (12 MB, shared by all chains, re-read by every proposal) stay in the Infinity Cache, the pose tables
    (1.6 MB per chain, 1.6 GB in all) do not fit.  low = pose lines only; high = everything but the records."""
    path = os.path.join(ROOT, "profiles", "window_probe.json")
    try:
        with open(path) as f:
            rec = json.load(f)
        if rec["workload"] != workload_name or rec["updates_per_step"] != updates_per_step:
            return None
        if rec.get("library_sha16") != library_sha16():  # taken on another build of the library: not quoted
            return {"withheld": "profiles/window_probe.json was taken on another build of the library",
                    "library_sha16": rec.get("library_sha16"), "running_library_sha16": library_sha16()}
        pose, records = rec["pose_line_bytes_per_unit"], rec["record_bytes_per_unit"]
        low = pose * units_per_launch
        high = traffic - records * units_per_launch
        return {"low_bytes_per_launch": low, "high_bytes_per_launch": high, "source": os.path.relpath(path, ROOT),
                "library_sha16": rec.get("library_sha16"), "running_library_sha16": library_sha16(),
                "basis": "bracket: pose-table lines only .. all requests but the reading records' (which the Infinity Cache holds)"}
    except (OSError, ValueError, KeyError):
        return None


def plaza_wall(name, device, with_cpu):
    """Plaza1 / Plaza2 --mcmc-slam --mcmc-steps 10 --mcmc-end-steps 1000 --sensor-range-stddev 1: wall time of the
    whole online replay, one chain on the GPU (and the CPU oracle beside it).  The GPU leg reads the logs
    with the shipped loader (the host program leaves a binary cache behind) and never touches the oracle."""
    import subprocess
    import tempfile

    from slam_amd import capi, dataset_cache, synthetic

    plaza = os.path.join(ROOT, "tests", "golden", "plaza")
    exe = os.path.join(ROOT, "slam_amd", "host", "slam_hip")
    with tempfile.TemporaryDirectory() as tmp:
        cache = os.path.join(tmp, name + ".slamro")
        subprocess.check_call([exe, "--dataset", name, "--dataset-dir", plaza, "--dataset-cache", cache,
                               "--dump-loader"], stdout=subprocess.DEVNULL)
        data = dataset_cache.read(cache)
    T, N = data["T"], data["N"]
    ctrl_mean, ctrl_std = synthetic.control_models(data["controls"], data["dt"])
    obs_f, order = dataset_cache.dense_landmarks(data["obs_id"])
    obs_std = np.full(N, 1.0)
    seeds = capi.seed_sequence(42, 6)
    out = {}
    g = capi.McmcGpu(1, T, len(order), N, device=device)
    t0 = time.perf_counter()
    g.upload_problem(ctrl_mean, ctrl_std, data["obs_t"], obs_f, data["obs_range"], obs_std, len(order))
    g.init_chains(np.array([seeds[2]], dtype=np.uint32), capi.uniform_sequence(int(seeds[1]), len(order)))
    g.advance(T, steps_per_timestep=10, end_steps=1000)
    g.sync()
    out["gpu_wall_s"] = time.perf_counter() - t0
    out["gpu_log_likelihood"] = float(g.log_likelihood()[0])
    g.close()
    if with_cpu:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib

        w = oracle_lib.World()
        w.load_plaza(plaza, name, 1.0)
        w.set_initialiser(int(seeds[1]))
        w.add_chain(int(seeds[2]), steps_per_timestep=10, end_steps=1000, use_initialiser=True)
        t0 = time.perf_counter()
        w.replay(T, finish=True)
        out["cpu_wall_s"] = time.perf_counter() - t0
        out["cpu_log_likelihood"] = w.log_likelihood(0)
    return out


def launch_ranks(n):
    """python bench.py --gpus N outside a launcher: N fresh child processes, one rank each, rendezvous on
    127.0.0.1.  This process has not touched a GPU (nothing GPU-related is imported before this point);
    it only waits for the children and hands on the worst exit code.  Rank 0 prints the JSON line."""
    import socket
    import subprocess

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    children = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        children.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    codes = [c.wait() for c in children]
    return max((abs(c) for c in codes), default=0)


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args.gpus))
    # Libraries write to the process's standard output behind Python's back (RCCL prints a version
    # banner there when a communicator is made).  The JSON line is the only thing this program may
    # print there, so file descriptor 1 points at standard error from here on and the line goes to a
    # duplicate of the original descriptor.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE is {world}: start one rank per GPU "
                         f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...) "
                         "or run python bench.py --gpus N without a launcher")
    import torch
    import torch.distributed as dist

    from slam_amd import capi, parallel, synthetic

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible")
    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    elif args.collectives_with_one_rank:
        import socket

        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)

    tempering = args.workload == "tempering"
    bearing = args.workload == "range-bearing"
    burn_in = args.workload == "burn-in"
    problem = synthetic.make_problem(T=args.timesteps, n_landmarks=args.landmarks, k_nearest=args.k_nearest,
                                     seed=args.seed, max_tries=2 if tempering else 8, range_stddev=args.range_stddev,
                                     bearing_stddev=args.bearing_stddev if bearing else None)
    if args.total_chains:
        if args.total_chains % world:
            raise SystemExit("--total-chains must be a multiple of the number of GPUs")
        args.chains = args.total_chains // world
    T, F, N, C = problem["T"], problem["F"], problem["N"], args.chains
    # one explicit stream for the library's kernels, torch's reductions and the collectives:
    # a zero stream handle would make the library create a private stream of its own
    stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)
    g = capi.McmcGpu(C, T, F, N, device=local_rank,
                     threads_per_chain=args.threads_per_chain, stream=stream.cuda_stream, range_bearing=bearing,
                     max_split_slices=args.split_slices, serial_program=args.serial_program)
    g.upload_problem(problem["ctrl_mean"], problem["ctrl_std"], problem["obs_t"], problem["obs_f"],
                     problem["obs_mean"], problem["obs_std"], F)
    if bearing:
        g.set_bearings(problem["obs_bearing"], problem["obs_bearing_std"])
    # chain seeds the way multi_mcmc draws them (successive engine outputs), global chain numbering
    all_seeds = capi.seed_sequence(args.seed + 2, C * world)
    g.init_chains(parallel.shard_seeds(all_seeds, C, rank))
    g.load_state(problem["true_rel"], problem["landmarks"])
    if tempering:  # one ladder per GPU: slot 0 is the untempered chain
        g.set_ladders(np.geomspace(1.0, args.beta_min, C), exchange_seed=args.seed + 3 + rank)
    if burn_in:  # every chain a "ladder" of its own at one inverse temperature; nothing is ever exchanged
        g.set_ladders(np.array([args.beta]), exchange_seed=args.seed + 3 + rank)
    g.sync()

    intervals = -(-args.updates_per_step // args.report_every)
    # mcmcgpu_chain_summary records, one row of C per report interval
    summaries = torch.zeros((intervals, C, 2), dtype=torch.float64, device=dev)
    best = {"ll": float("-inf"), "chain": -1}

    gather_events = []  # (before, after) the exchange of every timed step, on the stream everything runs on
    dumped = []         # --dump-records: the gathered records of every timed step

    def one_step(i, timed=False):
        if args.split_slices:
            g.run_split_reported(args.updates_per_step, args.split_slices, args.report_every, summaries.data_ptr())
        else:
            g.run_reported(args.updates_per_step, args.report_every, summaries.data_ptr())
        if args.rebuild_every and (i + 1) % args.rebuild_every == 0:
            g.rebuild()
        if tempering:
            g.exchange(i % 2)
        if timed:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        gathered = parallel.gather_summaries(summaries, world)  # RCCL all-gather when world > 1
        if timed:
            ev[1].record()
            gather_events.append(ev)
            if args.dump_records and rank == 0:
                dumped.append(gathered.cpu().numpy().copy())
        # best log likelihood of every report interval, stays on the device
        return gathered[-1], torch.max(gathered[:, :, 0], dim=1).values

    def barrier():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize(dev)

    def note(msg):  # progress lines for rehearsals, where a hang must be located from a log
        if args.rehearse_on_one_gpu:
            print(f"[rank {rank}] {msg}", file=sys.stderr, flush=True)

    note("problem uploaded, chains loaded")
    for i in range(args.warmup):
        one_step(i)
        note(f"warm-up step {i} done")
    barrier()
    g.kernel_time()  # reset the library's kernel timers
    g.rebuild_time()
    u0 = g.counters()
    s0 = g.traffic_counters()
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        gathered, ll = one_step(args.warmup + i, timed=True)
    barrier()
    elapsed = time.perf_counter() - t0
    note("timed steps done")
    # a result is asked for: arg-max with multi_mcmc's tie rule, then the winner's state from its owner
    best["ll"], best["chain"], owner, owner_idx = parallel.best_chain(gathered, C)
    packed = torch.zeros(3 * T + 2 * F, dtype=torch.float64, device=dev)
    if rank == owner:
        g.export_state(owner_idx, packed.data_ptr())
    parallel.broadcast_from_owner(packed, owner, world)
    torch.cuda.synchronize(dev)
    kern_ms, launches = g.kernel_time()
    rebuild_ms, rebuilds = g.rebuild_time()
    u1 = g.counters()
    s1 = g.traffic_counters()
    d_updates = int((u1[0] - u0[0]).sum())
    d_accepts = int((u1[1] - u0[1]).sum())
    d_units = int((u1[2] - u0[2]).sum())
    d_state = int((s1[0] - s0[0]).sum())
    d_moved = int((s1[1] - s0[1]).sum())

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    tot = torch.tensor([d_units, d_updates, d_accepts], dtype=torch.float64, device=dev)
    parallel.all_reduce(t, dist.ReduceOp.MAX, world)
    parallel.all_reduce(tot, dist.ReduceOp.SUM, world)
    elapsed_max = float(t.item())
    units_total, updates_total, accepts_total = (float(x) for x in tot.tolist())
    # what every rank saw, gathered through the same collective the steps use: the record of an N-GPU run then
    # shows by itself that the backend had N ranks and where any loss against N x one GPU is (kernel tail or exchange)
    gather_ms = [a.elapsed_time(b) for a, b in gather_events]
    props = torch.cuda.get_device_properties(dev)
    mine = torch.tensor([[rank, local_rank, kern_ms, launches, 1e3 * elapsed / max(args.steps, 1),
                          sum(gather_ms) / max(len(gather_ms), 1), max(gather_ms, default=0.0), d_units,
                          float(getattr(props, "pci_bus_id", -1)), float(getattr(props, "pci_device_id", -1))]],
                        dtype=torch.float64, device=dev)
    per_rank = parallel.gather_summaries(mine, world).cpu().numpy()

    workload_name = ("synthetic planar robot: %d poses, %d landmarks, %d range readings, %d %s per GPU" %
                     (T, F, N, C, "tempered replicas (one ladder)" if tempering else
                      "chains, range-and-bearing readings" if bearing else
                      "chains, commit-heavy (%d nearest, sigma %g m)" % (args.k_nearest, args.range_stddev)
                      if args.workload == "commit-heavy" else
                      "chains heated to beta %g (burn-in)" % args.beta if burn_in else "chains"))
    if rank == 0:
        # HBM-side bytes per launch: measured now by counter passes over a short child run (one GPU only), else
        # quoted from the last committed profiling round when that was taken on this library, else null
        traffic, traffic_source = None, None
        if world == 1 and not args.no_live_traffic:
            g.sync()
            traffic, traffic_source = live_traffic(args, bool(args.split_slices), args.live_traffic_seconds)
        if traffic is None:
            failed = traffic_source
            traffic, traffic_source = quoted_traffic(args, workload_name)
            if failed is not None and traffic_source is not None:
                traffic_source["live_pass"] = failed
            elif failed is not None:
                traffic_source = failed
        # a range-and-bearing reading also needs the pose's angle and the measured bearing: 48 bytes
        unit_bytes = BYTES_PER_UNIT + 16 if bearing else BYTES_PER_UNIT
        commit_bytes = d_moved * BYTES_PER_MOVED_POSE / max(launches, 1)
        bytes_per_launch = (d_units * unit_bytes + d_state * F * BYTES_PER_LANDMARK) / max(launches, 1) + commit_bytes
        achieved = bytes_per_launch / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
        scan_bytes = 48.0 * T * C  # a rebuild reads 24 B and writes 24 B per pose
        other_kernels = [{
            "kernel": "commit_state_edge (inside advance_kernel; split_commit_kernel in split mode)",
            "moved_poses_per_launch": d_moved / max(launches, 1),
            "bytes_per_launch": commit_bytes,
            "share_of_algorithmic_bytes": commit_bytes / bytes_per_launch if bytes_per_launch else 0.0,
            "note": "48 B per re-positioned pose; no duration of its own inside the persistent kernel",
        }]
        if rebuilds:
            other_kernels.append({
                "kernel": "scan_chain_kernel / scan_pieces_kernel (mcmcgpu_rebuild)", "ms": rebuild_ms, "launches": rebuilds,
                "bytes_per_launch": scan_bytes, "achieved": scan_bytes / (rebuild_ms * 1e-3) / 1e9 if rebuild_ms > 0 else 0.0,
                "frac": scan_bytes / (rebuild_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if rebuild_ms > 0 else 0.0,
            })
        kern_s = kern_ms * 1e-3
        counter_gbs = traffic / kern_s / 1e9 if traffic and kern_s > 0 else None
        roofline = {
            "bound": "hbm",
            # frac: bytes the counters saw leave L2 per launch (HBM + Infinity Cache) over the kernel's device time
            # and the HBM peak -- a physical fraction, at most 1.  algorithmic_frac: the per-unit byte figure of the
            # metric's definition (32 B per proposal*observation, 48 B with bearings, + per-proposal landmark table
            # + 48 B per re-positioned pose) over the same time; it exceeds 1 where caches serve shared records
            # and neighbouring pose lines, and says how much work is done per second, not how busy HBM is.
            "achieved": counter_gbs,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": counter_gbs / HBM_PEAK_GBS if counter_gbs is not None else None,
            "traffic": traffic,
            "traffic_source": traffic_source,
            "basis": "achieved = roofline.traffic (2 x FETCH_SIZE + WRITE_SIZE of the dominant kernel per launch: requests "
                     "that leave L2, Infinity-Cache hits included) / kernel_ms; null when no counter pass is available",
            "algorithmic_bytes_per_launch": bytes_per_launch,
            "algorithmic_achieved": achieved,
            "algorithmic_frac": achieved / HBM_PEAK_GBS,
            "traffic_over_algorithmic": traffic / bytes_per_launch if traffic and bytes_per_launch else None,
            "kernel": "split_* kernel sequence of one step" if args.split_slices else "advance_kernel",
            "kernel_ms": kern_ms,
            "launches": launches,
            "other_kernels": other_kernels,
        }
        if traffic:
            only = hbm_only_bracket(workload_name, args.updates_per_step, d_units / max(launches, 1), traffic)
            if only and "withheld" in only:
                roofline["hbm_only"] = only
            elif only:
                only["low_frac"] = only["low_bytes_per_launch"] / kern_s / 1e9 / HBM_PEAK_GBS
                only["high_frac"] = only["high_bytes_per_launch"] / kern_s / 1e9 / HBM_PEAK_GBS
                roofline["hbm_only"] = only
        collective = {
            "backend": dist.get_backend() if dist.is_initialized() else "none (one rank, no process group)",
            "world_size": dist.get_world_size() if dist.is_initialized() else 1,
            "exchange": "all-gather of %d report rows x %d chains x 16 B per rank every step = %d B gathered" %
                        (intervals, C, intervals * C * 16 * world),
            "allgather_ms_per_step": {"mean_over_ranks_max": float(per_rank[:, 5].max()), "worst_step": float(per_rank[:, 6].max())},
            "kernel_ms_per_rank": {"min": float(per_rank[:, 2].min()), "max": float(per_rank[:, 2].max()),
                                   "all": [float(x) for x in per_rank[:, 2]]},
            "step_ms_per_rank": {"min": float(per_rank[:, 4].min()), "max": float(per_rank[:, 4].max())},
            "units_per_rank": [float(x) for x in per_rank[:, 7]],
            "ranks": [{"rank": int(r[0]), "local_rank": int(r[1]), "pci_bus_id": int(r[8]), "pci_device_id": int(r[9])}
                      for r in per_rank],
        }
        out = {
            "metric": "MCMC proposals*observations/sec (node) on 100k-pose synthetic",
            "value": units_total / elapsed_max,
            "unit": "proposals*observations/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed_max / args.steps,
            "higher_is_better": True,
            "scaling": "strong" if args.total_chains else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": workload_name,
                "chains_total": C * world,
                "updates_per_step": args.updates_per_step,
                "report_every": args.report_every,
                "split_slices": args.split_slices,
                "parallelism": "chains sharded %dx%d, per-interval summaries all-gathered every step" % (world, C),
                "start_state": "generating trajectory and landmark positions (mcmcgpu_load_state)",
                "range_stddev": args.range_stddev,
                "k_nearest": args.k_nearest,
                **({"beta": args.beta} if burn_in else {}),
            },
            "roofline": roofline,
            "collective": collective,
            "proposals_per_s": updates_total / elapsed_max,
            "accept_rate": accepts_total / max(updates_total, 1.0),
            "units_per_proposal": units_total / max(updates_total, 1.0),
            "best_log_likelihood": best["ll"],
            "best_chain": best["chain"],
        }
        if tempering:
            beta, slot, _, attempts, swaps = g.ladders()
            out["config"]["beta"] = "geometric 1 .. %g over %d slots, exchange after every launch" % (args.beta_min, C)
            out["config"]["threads_per_chain"] = args.threads_per_chain
            out["exchange"] = {"attempts": attempts, "swaps": swaps, "chain_at_beta_1": int(slot[0])}
        if args.dump_records:
            np.savez(args.dump_records, records=np.stack(dumped), best=np.array([best["ll"], best["chain"]]),
                     winner=packed.cpu().numpy())
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(problem, all_seeds[0], args.cpu_seconds, all_seeds)
        if world == 1 and not args.no_plaza:
            g.close()
            out["plaza1"] = plaza_wall("Plaza1", local_rank, with_cpu=not args.no_cpu_baseline)
            out["plaza2"] = plaza_wall("Plaza2", local_rank, with_cpu=not args.no_cpu_baseline)
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
