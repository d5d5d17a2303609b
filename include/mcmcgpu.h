/*
 * mcmcgpu -- MI355X (gfx950) implementation of the MCMC-SLAM proposal
 * evaluation path, behind a plain C ABI.
 *
 * The reference (tree: roshanshariff/slam) has no FFI: the seam is the C++
 * class slam::mcmc_slam<velocity_slip_model, range_only_model>
 * (src/slam/mcmc_slam.hpp:37-223, instantiated at src/slam/mcmc_slam.cpp:11).
 * A host-side class with the same public surface (slam_amd/host/) keeps its
 * whole sampler state on the GPU behind one of these handles; every entry
 * point below names the reference member it stands in for.
 *
 * Conventions: every function returns 0 on success and a negative
 * mcmcgpu_status on failure; mcmcgpu_last_error() gives the message.  All
 * pointers are HOST pointers unless the parameter name starts with `dev_`.
 * All floating point is IEEE binary64.  A handle is not thread-safe; all work
 * of a handle is ordered on one HIP stream.
 */
#ifndef MCMCGPU_H
#define MCMCGPU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mcmcgpu_ctx* mcmcgpu_handle;

enum mcmcgpu_status {
  MCMCGPU_OK = 0,
  MCMCGPU_ERR_ARGUMENT = -1,
  MCMCGPU_ERR_CAPACITY = -2,
  MCMCGPU_ERR_HIP = -3,
  MCMCGPU_ERR_STATE = -4,
  MCMCGPU_ERR_DEVICE_FAULT = -5 /* a chain reported an internal error flag */
};

enum mcmcgpu_edge_kind { MCMCGPU_STATE_EDGE = 0, MCMCGPU_FEATURE_EDGE = 1 };

typedef struct mcmcgpu_config {
  int32_t device;             /* HIP device ordinal */
  uint32_t max_chains;        /* C: independent chains held by this handle */
  uint32_t max_timesteps;     /* capacity for controls (state edges) */
  uint32_t max_features;      /* capacity for landmarks */
  uint32_t max_observations;  /* capacity for range readings */
  uint32_t threads_per_chain; /* workgroup size; 0 -> 256 (64 when max_observations <= 32768 and max_chains > 256); power of two in [64, 1024] */
  /* Order of the two draws inside a landmark proposal.  The reference leaves it
   * unsequenced (src/planar_robot/range_bearing_model.hpp:155-159): 1 = uniform
   * then normal (what g++ emits), 0 = normal then uniform (clang). */
  uint32_t feature_uniform_first;
  uint32_t trace_capacity;    /* MH steps recorded per chain (0 = none) */
  void* stream;               /* hipStream_t to run on; NULL = a stream of its own */
  /* Observation model (the reference fixes it at compile time, src/main.hpp:22):
   * 0 = range_only_model (range_bearing_model.hpp:119-196), what the reference
   * ships with; 1 = range_bearing_model (:21-117), every reading also carries a
   * bearing (mcmcgpu_set_bearings). */
  uint32_t range_bearing;
  /* Split mode (mcmcgpu_run_split): the most slices one proposal's reading
   * stream may be cut into; 0 = no split mode, and no memory set aside for it
   * (44 bytes per chain and landmark). */
  uint32_t max_split_slices;
  /* Split mode: from this many poses on, the pose-table update of an accepted
   * state edge runs as a kernel of its own with many workgroups per chain
   * instead of inside the chain's own workgroup; 0 = the default (200 000). */
  uint32_t split_commit_min_poses;
  /* Where the serial part of an update (selection, proposal, accept test; the reference's
   * src/slam/mcmc_slam.hpp:339-418 outside the loop over readings) is compiled: 0 = the library
   * chooses (compiled into the kernel when a launch has at most 1024 wavefronts -- 256 chains of 256
   * lanes, 1024 chains of 64 -- so that every wavefront has a SIMD to itself and the serial part is
   * the launch's critical path; a called function
   * otherwise, which keeps the registers of the loop over readings free); 1 = always called;
   * 2 = compiled in whenever the workgroup has at most 256 lanes; 3 = compiled in, and the first
   * wavefront selects and proposes the next update while the others evaluate the current one
   * (workgroups of 128 or 256 lanes; what 0 chooses for few chains on data of at most 32 768
   * readings, where an update's serial part outlasts its evaluation).  Results do not depend on it. */
  uint32_t serial_program;
} mcmcgpu_config;

/* One Metropolis-Hastings step as recorded when trace_capacity > 0. */
typedef struct mcmcgpu_step_record {
  int32_t kind;       /* mcmcgpu_edge_kind */
  int32_t accepted;   /* 0 / 1 */
  uint64_t index;     /* timestep of the state edge, or landmark index */
  double proposed[3]; /* pose (x, y, bearing) or position (x, y, 0) */
  double log_ratio;   /* value of edge_log_likelihood_ratio, mcmc_slam.hpp:385 */
  double log_likelihood; /* chain log likelihood after the step */
} mcmcgpu_step_record;

/* Per-chain summary, the unit of the multi-GPU all-gather (the serial arg-max
 * of src/slam/multi_mcmc.hpp:109-118 becomes a reduction over these). */
typedef struct mcmcgpu_chain_summary {
  double log_likelihood;
  uint32_t updates;
  uint32_t accepts;
} mcmcgpu_chain_summary;

const char* mcmcgpu_last_error(mcmcgpu_handle h); /* h may be NULL: last create error */

/* Allocates every device buffer for the stated capacities; nothing else
 * allocates later.  Stands in for constructing mcmc_slam objects
 * (mcmc_slam.hpp:165-166) and for multi_mcmc's chain vector (multi_mcmc.hpp:164-166). */
int mcmcgpu_create(const mcmcgpu_config* cfg, mcmcgpu_handle* out);
void mcmcgpu_destroy(mcmcgpu_handle h);

/* The data the sampler reads -- the contents of slam_data
 * (src/slam/slam_data.hpp:28-148): one normal model per control
 * (mean/stddev, 3 doubles each, src/planar_robot/velocity_model.hpp:176-180)
 * and one per range reading (range_bearing_model.hpp:192-194), readings in
 * ARRIVAL order (non-decreasing timestep, 0..T).  obs_feature holds dense
 * landmark indices numbered in order of first appearance (the index into
 * mcmc_slam::feature_estimates, mcmc_slam.hpp:125-129).  May be called again
 * with a longer problem whose prefix is unchanged (online growth). */
int mcmcgpu_upload_problem(mcmcgpu_handle h, uint32_t T, const double* ctrl_mean, const double* ctrl_stddev,
                           uint32_t N, const uint32_t* obs_timestep, const uint32_t* obs_feature,
                           const double* obs_mean, const double* obs_stddev, uint32_t F);

/* Online growth, one timestep at a time: what slam_data::add_control / add_observation hand to
 * the listeners before timestep_listener::timestep(t) (src/slam/slam_data.hpp:76-101,
 * mcmc_slam.hpp:272-336), without uploading the problem again.  Appends the control of the edge to
 * the new timestep (3 + 3 doubles; NULL for timestep 0 of an empty problem, which has none) and the
 * n_obs readings taken there (obs_feature as in mcmcgpu_upload_problem: a landmark seen for the
 * first time has the next unused index; bearing arrays only on a range-and-bearing handle).
 * init_state_rel (3 doubles) and init_feat_rel (2 doubles per landmark that is NEW in this call,
 * in order of appearance) continue the arrays of mcmcgpu_set_initial_estimates when they reach
 * exactly up to the new edge / landmark; NULL = none.  The readings go into free slots next to
 * their landmark's list; a landmark whose slots are used up is moved, alone, to a region of twice
 * the size (amortised O(1) per reading); only when the slot space itself runs out is the whole
 * problem laid out again.  Follow with mcmcgpu_grow / mcmcgpu_advance. */
int mcmcgpu_append_timestep(mcmcgpu_handle h, const double* ctrl_mean, const double* ctrl_stddev, uint32_t n_obs,
                            const uint32_t* obs_feature, const double* obs_mean, const double* obs_stddev,
                            const double* bearing_mean, const double* bearing_stddev, const double* init_state_rel,
                            const double* init_feat_rel);

/* Range-and-bearing handles only: the bearing half of every reading's model
 * (range_bearing_model::builder, range_bearing_model.hpp:100-116), in the
 * ARRIVAL order of the mcmcgpu_upload_problem call it follows; required after
 * every upload.  A landmark proposal is then inv_observe of two normals
 * (:84-86), a new landmark starts at inv_observe(mean) without a draw (:93-95),
 * and the bearing residual is wrapped (:33-36). */
int mcmcgpu_set_bearings(mcmcgpu_handle h, uint32_t N, const double* bearing_mean, const double* bearing_stddev);

/* Creates C empty chains (timestep 0 not yet processed), chain i seeded like
 * random_source(seeds[i]) (src/utility/random.hpp:24).  init_uniforms, when
 * not NULL, holds one uniform per landmark: the draws a slam_initialiser
 * (src/slam/slam_initialiser.hpp:40-49) would have made; landmarks are then
 * placed from those instead of the chain's own generator, which is what
 * mcmc_slam::set_initialiser changes (mcmc_slam.hpp:255-259). */
int mcmcgpu_init_chains(mcmcgpu_handle h, uint32_t C, const uint32_t* seeds, const double* init_uniforms,
                        uint32_t n_init_uniforms);

/* What mcmc_slam::set_initialiser changes for ANY initialiser (mcmc_slam.hpp:233-237,
 * 255-259): the first n_state state edges start from state_rel[t] (= -init.get_state(t)
 * + init.get_state(t+1)) and the first n_feat landmarks from feat_rel[f] (=
 * -init.get_state(t_first) + init.get_feature(id)) instead of the proposals' initial
 * values; such edges draw nothing from the chain's generator.  Call again with longer
 * arrays as data arrives; edges beyond the arrays fall back to the defaults. */
int mcmcgpu_set_initial_estimates(mcmcgpu_handle h, uint32_t n_state, const double* state_rel, uint32_t n_feat,
                                  const double* feat_rel);

/* The whole of slam_data::add_dataset as the sampler sees it
 * (slam_data.hpp:191-215 driving mcmc_slam::timestep, mcmc_slam.hpp:272-336,
 * and mcmc_slam::updater, :200-222): for every not yet processed timestep up
 * to `upto` (inclusive) grow the graph, then do `steps_per_timestep` updates;
 * afterwards `end_steps` more.  mcmcgpu_grow == advance(upto, 0, 0);
 * mcmcgpu_run == advance(current, 0, steps) == steps x mcmc_slam::update()
 * (mcmc_slam.hpp:339-368) on every chain.  Asynchronous on the stream. */
int mcmcgpu_advance(mcmcgpu_handle h, uint32_t upto, uint32_t steps_per_timestep, uint32_t end_steps);
int mcmcgpu_grow(mcmcgpu_handle h, uint32_t upto);
int mcmcgpu_run(mcmcgpu_handle h, uint32_t steps);

/* mcmcgpu_run in ONE launch with the report lines of multi_mcmc::completed
 * (src/slam/multi_mcmc.hpp:121-139: a report after every 100 updates): after
 * every `report_every` updates, and after the last one, each chain writes its
 * mcmcgpu_chain_summary to DEVICE memory dev_out[interval * C + chain];
 * ceil(steps / report_every) intervals.  Chains cost different amounts of work
 * per update, so a launch lasts as long as its slowest chain; long launches
 * keep the idle tail small without giving up the report cadence.
 * dev_out == NULL: the records go to a buffer of the library (at most 256
 * intervals per launch) and mcmcgpu_get_reports copies them to HOST memory,
 * out[interval * C + chain], after the launch has finished; *intervals = how
 * many there are (out may be NULL to ask only that). */
int mcmcgpu_run_reported(mcmcgpu_handle h, uint32_t steps, uint32_t report_every, void* dev_out);
int mcmcgpu_get_reports(mcmcgpu_handle h, mcmcgpu_chain_summary* out, uint32_t capacity_intervals,
                        uint32_t* intervals);

/* Parallel tempering.  NO counterpart in the reference (its sampler is the
 * case beta = 1); this is the "tempered replicas" batching mode.  The chains
 * are grouped into ladders of `replicas` consecutive chains; the chain in
 * temperature slot s of its ladder samples controls(x) * readings(x)^betas[s]:
 * in the accept test of mcmc_slam.hpp:408 the readings' part of log_ratio
 * (the sums of :476-505) is multiplied by beta, the control term of :436-438
 * is not; with beta = 1 it is the reference's test bit for bit.  Initially slot s is held by chain ladder*replicas + s.
 * mcmcgpu_exchange attempts, in every ladder, to swap the temperatures of the
 * chains in slots (phase + 2k, phase + 2k + 1), phase 0 or 1, each with
 * probability min(1, exp((beta_a - beta_b) (L_b - L_a))), L = the readings'
 * part of the chain's log likelihood; one uniform per
 * pair is always drawn from a library-owned generator (same engine and
 * uniform transform as random_source, src/utility/random.hpp:11-42, seeded
 * with `exchange_seed`), in (ladder, k) order.  States stay where they are;
 * the log likelihoods reported everywhere are untempered.  mcmcgpu_get_ladders
 * returns each chain's beta, the chain in each slot, each chain's L, and the
 * totals of attempted and accepted swaps; null pointers are skipped.  init_chains resets
 * every chain to beta = 1. */
int mcmcgpu_set_ladders(mcmcgpu_handle h, uint32_t replicas, const double* betas, uint32_t exchange_seed);
int mcmcgpu_exchange(mcmcgpu_handle h, uint32_t phase);
int mcmcgpu_get_ladders(mcmcgpu_handle h, double* beta, uint32_t* slot_chain, double* reading_ll, uint64_t* attempts,
                        uint64_t* swaps);

/* mcmcgpu_run with every update issued as a sequence of kernels, the reading
 * stream of each proposal (the loop of mcmc_slam.hpp:476-505) cut into up to
 * `slices` equal parts that run on different workgroups (no part shorter than
 * 1024 readings: a small proposal takes few workgroups).  mcmcgpu_run gives a
 * chain one workgroup, which is the right shape for hundreds of chains; a
 * single mcmc_slam instance or a handful of tempered replicas on a long
 * trajectory would leave most of the GPU idle.  Same proposals, same accept
 * decisions, same random stream as mcmcgpu_run (the partial sums are added in
 * slice order, so log ratios agree to rounding).  For updates after growth;
 * slices <= max_split_slices of the config. */
int mcmcgpu_run_split(mcmcgpu_handle h, uint32_t steps, uint32_t slices);

/* mcmcgpu_run_split with the report records of mcmcgpu_run_reported: after every
 * report_every updates (and after the last) every chain's summary goes to row k
 * of dev_out ([intervals][chains] records in device memory), or to the library's
 * own buffer when dev_out is NULL (mcmcgpu_get_reports). */
int mcmcgpu_run_split_reported(mcmcgpu_handle h, uint32_t steps, uint32_t slices, uint32_t report_every, void* dev_out);

/* edge_log_likelihood_ratio (mcmc_slam.hpp:432-474) for one given proposal on
 * one chain, without changing the chain.  For a state edge the value includes
 * the control term (:436-438). */
int mcmcgpu_eval_llr(mcmcgpu_handle h, uint32_t chain, int kind, uint32_t index, const double proposed[3],
                     double* out);

/* The accept branch of mcmc_slam::update(EdgeType&&, bool), :409-413: store
 * the proposal, set the edge weight, add log_ratio to the chain's likelihood. */
int mcmcgpu_commit(mcmcgpu_handle h, uint32_t chain, int kind, uint32_t index, const double proposed[3],
                   double new_weight, double log_ratio);

/* Rebuilds every chain's cumulative poses from its relative poses with the
 * SE(2) prefix scan (replaces re-deriving bitree<pose> nodes,
 * src/utility/bitree.hpp:182-184); removes accumulated rounding drift. */
int mcmcgpu_rebuild(mcmcgpu_handle h);

/* The scan of mcmcgpu_rebuild for callers that cannot allocate.  NO counterpart in the reference
 * and no handle: it exists for code that must not call hipMalloc or have a library do it -- work
 * under HIP graph capture, tight serving loops, harnesses that forbid allocation.  The caller
 * passes inputs and outputs as DEVICE pointers: relative poses of `chains` trajectories of n
 * edges each, component arrays [chains][rel_stride]; out come the cumulative poses, translation
 * dev_cum_xy [chains][cum_stride] as (x, y) pairs (16-byte aligned) and unwrapped angle dev_cum_th
 * [chains][cum_stride], slot k = the pose after k edges (slot 0 = identity), exactly the numbers
 * mcmcgpu_rebuild leaves in a handle's pose tables for the same chains and relative poses.  What
 * the scan needs when few chains are cut into many workgroups -- the pieces' totals and the counter
 * that hands the pieces out -- lives in a static device array of the library (1 MiB per GPU, held
 * until the process ends) instead of allocated memory; the call allocates and frees nothing and runs on `stream` (NULL = the null
 * stream) of `device`.  Limits: chains <= 4096 and n <= 0xFFFF0000 (MCMCGPU_ERR_CAPACITY beyond),
 * rel_stride >= n, cum_stride >= n + 1.  All calls on one GPU share that array: calls that may
 * overlap on one GPU (different streams or threads) race and must be serialised by the caller.
 * Errors are reported through mcmcgpu_last_error(NULL). */
int mcmcgpu_scan_poses_static(int32_t device, void* stream, uint32_t chains, uint32_t n, const double* dev_rel_x,
                              const double* dev_rel_y, const double* dev_rel_th, uint32_t rel_stride,
                              double* dev_cum_xy, double* dev_cum_th, uint32_t cum_stride);

/* get_log_likelihood (mcmc_slam.hpp:173) for chains [0, C). */
int mcmcgpu_get_log_likelihood(mcmcgpu_handle h, double* out);
/* slam_log_likelihood (src/slam/slam_likelihood.hpp:26-58) recomputed from
 * scratch on the device for every chain. */
int mcmcgpu_full_log_likelihood(mcmcgpu_handle h, double* out);

/* Chain read-out -- what get_trajectory / get_state / get_feature /
 * get_feature_map return (mcmc_slam.hpp:189-198, 507-525).  Any pointer may
 * be NULL.  rel_poses: [T][3] relative pose per state edge; cum_poses:
 * [T+1][3] = get_state(k); feat_parent: [F] parent timestep; feat_estimate:
 * [F][2] parent-relative estimate; feat_world: [F][2] = get_feature();
 * state_weights [T], feat_weights [F]: edge weights. */
int mcmcgpu_get_state(mcmcgpu_handle h, uint32_t chain, double* rel_poses, double* cum_poses,
                      uint32_t* feat_parent, double* feat_estimate, double* feat_world, double* state_weights,
                      double* feat_weights);
/* Current sizes: timesteps grown (state edges), landmarks known, chains. */
int mcmcgpu_get_dims(mcmcgpu_handle h, uint32_t* timesteps, uint32_t* features, uint32_t* chains);

/* Counters since init: updates, accepted updates and evaluated observations
 * (executions of the loop body mcmc_slam.hpp:487-502), per chain. */
int mcmcgpu_get_counters(mcmcgpu_handle h, uint64_t* updates, uint64_t* accepts, uint64_t* units);

/* Counters for traffic accounting, per chain since init: state-edge proposals
 * evaluated (each reads the landmark table once) and cumulative-pose entries
 * rewritten by accepted state edges (block-local poses + block offsets). */
int mcmcgpu_get_traffic_counters(mcmcgpu_handle h, uint64_t* state_proposals, uint64_t* moved_poses);

/* Recorded steps of one chain (trace_capacity > 0); *count = number returned. */
int mcmcgpu_get_trace(mcmcgpu_handle h, uint32_t chain, mcmcgpu_step_record* out, uint32_t capacity,
                      uint32_t* count);

/* Writes C mcmcgpu_chain_summary records to DEVICE memory (stream-ordered),
 * ready to be all-gathered across GPUs. */
int mcmcgpu_write_summaries(mcmcgpu_handle h, void* dev_out);
/* Packs one chain's state (relative poses [T][3], then feat estimates [F][2])
 * as doubles into DEVICE memory, for a broadcast of the best chain;
 * mcmcgpu_import_state is the receiving side. */
int mcmcgpu_export_state(mcmcgpu_handle h, uint32_t chain, void* dev_out);
int mcmcgpu_import_state(mcmcgpu_handle h, uint32_t chain, const void* dev_in, double log_likelihood);

/* ---- Exchange between handles: RCCL over xGMI -------------------------------------------------
 * The reference runs all chains in one loop and compares them there (multi_mcmc::update,
 * src/slam/multi_mcmc.hpp:105-119), reports after every 100 updates (:121-139) and averages them
 * (:71-73).  With the chains of one job sharded over several handles -- one per GPU, in one process
 * or in one process per GPU -- those three become collectives, issued by the library on each
 * handle's stream: an all-gather of the 16-byte chain summaries, a broadcast of one chain's state
 * from the rank that owns it, an all-reduce of the averaging sums.  Rank r holds global chains
 * [r * slots_per_rank, r * slots_per_rank + its chain count); ranks may hold fewer chains than
 * slots_per_rank, the unused slots come back with log_likelihood = -infinity.  librccl is opened
 * when the first of these calls needs it.  A handle WITHOUT a communicator behaves as a
 * communicator of one rank (slots_per_rank = max_chains): every call below then works locally,
 * without RCCL.  Attaching a communicator allocates the exchange buffers (the one allocation
 * outside mcmcgpu_create).
 *
 *   mcmcgpu_comm_unique_id   ncclGetUniqueId: 128 bytes rank 0 hands to the other processes;
 *   mcmcgpu_comm_init        ncclCommInitRank for one handle per process;
 *   mcmcgpu_comm_init_all    ncclCommInitAll for n handles of ONE process on n different devices,
 *                            rank = position in the array; collectives of such handles issued from
 *                            one thread go between mcmcgpu_group_begin / mcmcgpu_group_end
 *                            (ncclGroupStart / ncclGroupEnd);
 *   mcmcgpu_comm_info        ranks, this handle's rank, slots per rank. */
#define MCMCGPU_UNIQUE_ID_BYTES 128
int mcmcgpu_comm_unique_id(unsigned char* id);
int mcmcgpu_comm_init(mcmcgpu_handle h, int nranks, int rank, const unsigned char* id, uint32_t slots_per_rank);
int mcmcgpu_comm_init_all(mcmcgpu_handle* handles, int n, uint32_t slots_per_rank);
int mcmcgpu_comm_info(mcmcgpu_handle h, int* nranks, int* rank, uint32_t* slots_per_rank);
int mcmcgpu_group_begin(void);
int mcmcgpu_group_end(void);

/* All-gather of chain summaries (stream-ordered, every rank calls it):
 *   mcmcgpu_allgather_summaries  the chains' current summaries, one interval;
 *   mcmcgpu_allgather_reports    the report records of the last mcmcgpu_run_reported /
 *                                mcmcgpu_run_split_reported that used the library's own buffer
 *                                (dev_out == NULL), all its intervals;
 *   mcmcgpu_get_gathered         waits and copies what was gathered to HOST memory:
 *                                out[(rank * intervals + interval) * slots + slot]; out may be NULL
 *                                to ask for the three sizes only. */
int mcmcgpu_allgather_summaries(mcmcgpu_handle h);
int mcmcgpu_allgather_reports(mcmcgpu_handle h);
int mcmcgpu_get_gathered(mcmcgpu_handle h, mcmcgpu_chain_summary* out, uint32_t capacity_records, uint32_t* ranks,
                         uint32_t* intervals, uint32_t* slots_per_rank);

/* The best chain's state to every rank (what max_likelihood points at, multi_mcmc.hpp:115-117, once
 * a result is asked for): rank `root` packs its chain `chain` (ignored on the other ranks), the
 * library broadcasts it into a buffer of each handle; mcmcgpu_get_broadcast_state copies that to
 * HOST memory as rel_poses [T][3] and parent-relative landmark estimates [F][2] (either may be NULL). */
int mcmcgpu_broadcast_state(mcmcgpu_handle h, int root, uint32_t chain);
int mcmcgpu_get_broadcast_state(mcmcgpu_handle h, double* rel_poses, double* feat_estimate);

/* multi_mcmc::get_average over the chains of ALL ranks (src/slam/average_slam_result.hpp:27-54),
 * in three phases every rank goes through in turn: 0 = the base vectors (global chain 0) computed
 * on rank 0 and broadcast; 1 = each rank's sum of differences from the base, all-reduced; 2 = the
 * average itself, copied to HOST memory in the layout of mcmcgpu_get_average (total_chains = chains
 * of all ranks; the pointers are only used in phase 2 and may be NULL).  mcmcgpu_average_buffer
 * copies the base (which = 0) or the sum (1) between the handle's buffer and HOST memory (write = 0:
 * to the host, 1: from it): for handles that share a GPU, where RCCL cannot run, the caller carries
 * the two vectors between them itself. */
int mcmcgpu_average_collective(mcmcgpu_handle h, int phase, uint32_t total_chains, double* cumulative_poses,
                               double* features_world);
int mcmcgpu_average_buffer(mcmcgpu_handle h, int which, int write, double* host);

/* Hand-off of one chain's estimate to a CPU consumer WITHOUT stalling the
 * sampler -- the candidate loop of src/main.cpp:331-337 (10 000 updates, then
 * g2o_clustering::add(*mcmc_slam), which reads get_trajectory / get_feature_map
 * through g2o_slam::reinitialise, src/slam/g2o_slam.hpp:270-286, and optimises
 * on the CPU).  mcmcgpu_snapshot_begin queues, in stream order, a copy of the
 * chain's relative poses, world landmark positions and log likelihood into a
 * pinned host buffer of the handle; the caller may queue the next
 * mcmcgpu_run right behind it.  mcmcgpu_snapshot_wait blocks until THAT copy
 * has landed (not until the stream is idle) and hands it out: rel_poses
 * [T][3], feat_world [F'][2]; null pointers are skipped.  One snapshot can be
 * outstanding per handle. */
int mcmcgpu_snapshot_begin(mcmcgpu_handle h, uint32_t chain);
int mcmcgpu_snapshot_wait(mcmcgpu_handle h, uint32_t* timesteps, double* rel_poses, uint32_t* features,
                          double* feat_world, double* log_likelihood);

/* multi_mcmc::get_average = average_slam_result over the chains
 * (src/slam/multi_mcmc.hpp:71-73, src/slam/average_slam_result.hpp:58-110).
 * What is averaged, packed as doubles: for t < T the accumulated pose
 * trajectory.accumulate(t+1) as (x, y, angle) -> [T][3], then each landmark's
 * world position -> [F'][2] (F' = landmarks met so far).  The reference keeps
 * the first result's vectors as `base`, sums subtract(base, v) over the others
 * (angle differences wrapped) and returns subtract(base, sum / n)
 * (average_slam_result.hpp:27-54).
 *   mcmcgpu_average_base   writes one chain's vectors to DEVICE memory;
 *   mcmcgpu_average_sum    writes the sum over this handle's chains (in chain
 *                          order, leaving out `skip_chain` unless it is < 0) of
 *                          subtract(base, v) to DEVICE memory -- across GPUs
 *                          these sums are all-reduced (SUM);
 *   mcmcgpu_average_finish turns `base` into the average in place, n =
 *                          total_chains over all GPUs;
 *   mcmcgpu_get_average    does all three for this handle's chains (base =
 *                          chain 0) and copies the result to HOST memory; the
 *                          caller rebuilds the relative-pose tree from the
 *                          accumulated poses as push_back_accumulated does
 *                          (src/utility/bitree.hpp:294-300).
 * A NULL dev_base / dev_sum stands for a buffer of the handle itself (the ones
 * mcmcgpu_average_collective and mcmcgpu_average_buffer work on). */
int mcmcgpu_average_base(mcmcgpu_handle h, uint32_t chain, void* dev_base);
int mcmcgpu_average_sum(mcmcgpu_handle h, const void* dev_base, int32_t skip_chain, void* dev_sum);
int mcmcgpu_average_finish(mcmcgpu_handle h, void* dev_base, const void* dev_sum, uint32_t total_chains);
int mcmcgpu_get_average(mcmcgpu_handle h, double* cumulative_poses, double* features_world);

/* Puts chains [first, first+count) into one given state with the WHOLE uploaded
 * problem already seen (restoring a checkpoint; also how a run is started from
 * a known estimate, e.g. another back end's result as g2o_slam::reinitialise
 * does for g2o, src/slam/g2o_slam.hpp:270-286): relative poses rel_poses[T][3]
 * and world landmark positions feat_world[F][2].  Parents follow the rule of
 * mcmc_slam::timestep (mcmc_slam.hpp:310-318: the reading with the smallest
 * |range|, earliest on ties); edge weights (:240-241, 262-263), cumulative
 * poses and the log likelihood (slam_likelihood.hpp:26-58) are derived on the
 * device.  The chains' generators stay as mcmcgpu_init_chains seeded them. */
int mcmcgpu_load_state(mcmcgpu_handle h, uint32_t first, uint32_t count, const double* rel_poses,
                       const double* feat_world);

/* Blocks until all work queued on the handle's stream is done and reports a
 * device-side error flag if any chain set one. */
int mcmcgpu_sync(mcmcgpu_handle h);
/* Average device duration (ms) of the advance kernel launches timed with HIP
 * events since the last call (reset on read), and how many there were. */
int mcmcgpu_kernel_time(mcmcgpu_handle h, double* avg_ms, uint32_t* launches);
/* The same for the scan kernels of mcmcgpu_rebuild calls (one rebuild = one timed launch). */
int mcmcgpu_rebuild_time(mcmcgpu_handle h, double* avg_ms, uint32_t* launches);

/* Host helpers for seeding, same generator as the chains: n successive 32-bit
 * draws of random_source(seed) -- the chain seeds of multi_mcmc
 * (src/slam/multi_mcmc.hpp:160-166) and the module seeds of src/main.cpp:133-147
 * -- and n successive uniform() draws (what a slam_initialiser consumes, one
 * per new landmark, src/slam/slam_initialiser.hpp:44-47). */
void mcmcgpu_seed_sequence(uint32_t seed, uint32_t n, uint32_t* out);
void mcmcgpu_uniform_sequence(uint32_t seed, uint32_t n, double* out);

/* Diagnostic builds only (-DMG_PHASE_TIMING): shader-clock cycles lane 0 of every
 * chain spent per phase of the advance kernel, summed over chains -- [0] serial
 * selection/proposal, [1] state edge: landmark set-up, [2] state edge: reading
 * stream, [3] landmark edge evaluation, [4] reduction + accept test, [5] commit,
 * [6] whole kernel, [7] SIMD of the driver's wavefront (four 16-bit counters);
 * inside [0], for state edges: [8] weight totals, [9] the uniform that picks the
 * tree, [10] tree descent, [11] three normals, [12] inv_observe, [13] control log
 * density; [14] a landmark proposal from the descent on; [15] growth (new state
 * edge, arrivals).  A normal build returns
 * MCMCGPU_ERR_STATE. */
int mcmcgpu_debug_phase_cycles(mcmcgpu_handle h, uint64_t out[16], int reset);

/* Test hook for the scan building block: inclusive prefix "sums" of n affine
 * maps x -> a + b*x (mod p) under composition, computed by the same wavefront
 * scan template that builds cumulative poses (exact integer arithmetic; the
 * group of the reference's own tree test, test/bitree.cpp:36-54). */
int mcmcgpu_selftest_scan(int32_t device, uint32_t n, int32_t p, const int32_t* a, const int32_t* b,
                          int32_t* out_a, int32_t* out_b);

/* Self-test: out[i] = the device's |(dx[i], dy[i])| as the reading stream
 * computes it (position::distance, src/planar_robot/position.hpp:33, by a
 * reciprocal-square-root seed and fused refinement steps instead of sqrt). */
int mcmcgpu_selftest_range(int32_t device, uint32_t n, const double* dx, const double* dy, double* out);

/* Self-test of the serial part's wavefront forms (used when it is compiled into the kernel,
 * mcmcgpu_config.serial_program): n draws of the chain generator seeded with `seed`, kinds[i] = 0 a
 * uniform (random_source::uniform, src/utility/random.hpp:25-31), 1 a normal (:33-39), 2 / 3 three / two
 * normals drawn as one batch (output: first + 2 second + 4 third), once by the
 * one-lane code and once by the 64-lane code; and for each of n_search numbers u in [0, 1] the index
 * bitree<double>::binary_search(total * u) (src/utility/bitree.hpp:302-321) finds in a tree of
 * tree_n weights, by both forms.  The two outputs of each pair must be identical. */
int mcmcgpu_selftest_wave(int32_t device, uint32_t seed, uint32_t n, const uint8_t* kinds, double* out_serial, double* out_wave,
                          uint32_t tree_n, const double* weights, uint32_t n_search, const double* search_u,
                          uint32_t* index_serial, uint32_t* index_wave);

#ifdef __cplusplus
}
#endif
#endif /* MCMCGPU_H */
