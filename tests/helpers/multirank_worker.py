"""One rank of a multi-rank run on ONE GPU (all ranks on device 0, exchange over gloo): the N > 1 code path of
slam_amd/parallel.py driven by real handles.  Started by tests/test_gpu_multirank.py, once as 2 ranks x C
chains and once as 1 rank x 2C chains; rank 0 writes what the job as a whole produced to --out (npz).
Usage: RANK=.. WORLD_SIZE=.. MASTER_ADDR=127.0.0.1 MASTER_PORT=.. python multirank_worker.py --total-chains 6 --out f.npz
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--total-chains", type=int, required=True)
    ap.add_argument("--updates", type=int, default=300)
    ap.add_argument("--report-every", type=int, default=100)
    ap.add_argument("--out", required=True)
    args = ap.parse_args()
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    import torch
    import torch.distributed as dist

    from slam_amd import capi, parallel, synthetic

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    if world > 1:
        dist.init_process_group("gloo")
    C = args.total_chains // world
    p = synthetic.make_problem(T=300, n_landmarks=10, k_nearest=3, seed=11, lap_steps=150, wobble_period=500,
                               range_stddev=2.0, landmark_spread=6.0)
    T, F = p["T"], p["F"]
    stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)
    g = capi.McmcGpu(C, T, F, p["N"], device=0, stream=stream.cuda_stream)
    g.upload_problem(p["ctrl_mean"], p["ctrl_std"], p["obs_t"], p["obs_f"], p["obs_mean"], p["obs_std"], F)
    all_seeds = capi.seed_sequence(77, args.total_chains)  # ONE sequence for the job, multi_mcmc.hpp:160-166
    g.init_chains(parallel.shard_seeds(all_seeds, C, rank))
    g.grow(T)
    intervals = -(-args.updates // args.report_every)
    local = torch.zeros((intervals, C, 2), dtype=torch.float64, device=dev)
    g.run_reported(args.updates, args.report_every, local.data_ptr())
    gathered = parallel.gather_summaries(local, world)  # [intervals, world * C, 2]
    ll, gid, owner, idx = parallel.best_chain(gathered[-1], C)
    packed = torch.zeros(3 * T + 2 * F, dtype=torch.float64, device=dev)
    if rank == owner:
        g.export_state(idx, packed.data_ptr())
    parallel.broadcast_from_owner(packed, owner, world)
    acc_avg, feat_avg = parallel.average_over_ranks(g, C, rank, world, dev)
    torch.cuda.synchronize(dev)
    g.sync()
    if rank == 0:
        records = gathered.cpu().numpy().view(capi.CHAIN_SUMMARY).reshape(intervals, world * C)
        np.savez(args.out, log_likelihood=records["log_likelihood"], updates=records["updates"],
                 accepts=records["accepts"], best=np.array([ll, gid]), winner=packed.cpu().numpy(),
                 average_poses=acc_avg.cpu().numpy(), average_landmarks=feat_avg.cpu().numpy())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
