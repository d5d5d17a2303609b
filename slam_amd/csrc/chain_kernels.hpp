// Device kernels of the MCMC-SLAM path for gfx950 (MI355X).
//
// Execution model: ONE WORKGROUP PER CHAIN.  Chains never exchange data, so a
// chain's whole Metropolis-Hastings loop -- edge selection, proposal, the
// observation log-likelihood ratio, the accept test and the commit -- runs
// inside one persistent workgroup for as many steps as the launch asks for.
// Lane 0 of the workgroup is the "driver": it executes the short serial parts
// of the reference algorithm (src/slam/mcmc_slam.hpp:339-418 and the graph
// growth :227-336) and posts requests; all lanes then serve the data-parallel
// parts together:
//   * eval_state_edge / eval_feature_edge -- the fused observation
//     log-likelihood-ratio reduction (mcmc_slam.hpp:432-505);
//   * commit_state_edge -- the rigid re-positioning of the cumulative poses
//     behind an accepted state edge.
//
// Cumulative poses (what the reference keeps in bitree<pose>) are a flat
// per-chain table P[k] in world coordinates: translation as one 16-byte record
// (so a reading costs ONE 16-byte gather) and the unwrapped angle beside it.
// The table is built by the SE(2) prefix scan in scan.hpp and kept current by
// commit_state_edge, which rewrites the poses after an accepted edge (a point
// update followed by re-evaluation of the affected prefix sums).  A range
// reading only needs the TRANSLATION of P[s]: |P[s]^-1 w| = |w - t_s| for a
// world point w, so the hot loop never touches an angle.
#ifndef SLAM_AMD_CHAIN_KERNELS_HPP
#define SLAM_AMD_CHAIN_KERNELS_HPP

#include "../../include/mcmcgpu.h"
#include "device_math.hpp"
#include "scan.hpp"

namespace mg {

struct alignas(16) PoseXY {
  double x, y;
};

// One range reading, landmark-major order: 16 bytes, one load.
struct alignas(16) ObsRec {
  double mu;   // measured range
  uint32_t t;  // timestep it was taken at
  uint32_t f;  // landmark (dense index)
};

// A reading as the streams load it: range and timestep in 12 bytes (one three-dword load).
struct __attribute__((packed, aligned(4))) StreamRec {
  double mu;
  uint32_t t;
};
static_assert(sizeof(StreamRec) == 12, "stream records are 12 bytes");

// Where the readings of one landmark stand relative to one block of 2^tblk_shift timesteps: how many
// were taken before the block, how many inside it, and -- when those inside were taken at consecutive
// timesteps (one visit of the robot) -- the first of these timesteps, which turns "how many readings
// up to timestep t" into arithmetic.  One 16-byte load.
struct alignas(16) TimeBlock {
  uint32_t before, count;
  uint32_t run_from;  // timestep of the block's first reading when they are consecutive, else kNoRun
  uint32_t pad;
};
constexpr uint32_t kNoRun = 0xFFFFFFFFu;

// Everything a kernel needs, passed by value.
struct Dev {
  // ---- problem (read-only), shared by all chains -------------------------
  uint32_t T, F, N;           // uploaded sizes
  uint32_t Tcap, Fcap, Ncap;  // capacities (array strides)
  const double* cmu;          // [3][Tcap] control means
  const double* csd;          // [3][Tcap] control std deviations
  const double* cld;          // [Tcap] log csd[0] + log csd[1] + log csd[2]: the log-density's constant
  const uint32_t* foff;       // [Fcap+1] readings of landmark f: [foff[f], foff[f+1]), time-sorted
  const ObsRec* orec;         // [Ncap] readings, landmark-major
  // the same readings without the landmark, what the reading streams load: 12 bytes per reading
  // instead of the record's 16 (a stream knows its landmark from the segment it is in)
  const StreamRec* osr;       // [Ncap]
  const double* osd;          // [Ncap] std deviation of a reading
  const double* oisd;         // [Ncap] 1 / std deviation
  const uint32_t* opiv;       // [Ncap/32+1] timestep of every 32nd reading (search pivots)
  const uint32_t* otime;      // [Ncap] timestep of every reading (32 to a cache line, for searches)
  // [tblk_rows][Fcap] landmark f against the block of timesteps [b << tblk_shift, (b + 1) << tblk_shift),
  // for the blocks b = 0 .. tblk_n - 1 that the last layout of the problem covered; row tblk_n holds
  // the readings before the end of the last block.  Block-major: one proposal asks for the same
  // block of every landmark, so neighbouring lanes load neighbouring entries.
  const TimeBlock* tblk;
  uint32_t tblk_shift, tblk_n, tblk_rows;
  const uint32_t* toff;       // [Tcap+2] arrivals at timestep t: tobs[toff[t] .. toff[t+1])
  const uint32_t* tobs;       // [Ncap] arrival order -> landmark-major position
  const uint32_t* ffirst;     // [Fcap] timestep of a landmark's first reading
  const double* init_u;       // [n_init_u] initialiser draws per landmark, or null
  uint32_t n_init_u;
  const double* init_rel;     // [n_init_rel][3] initial relative pose per state edge (from an initialiser)
  const double* init_feat;    // [n_init_feat][2] initial landmark estimate relative to its first pose
  uint32_t n_init_rel, n_init_feat;
  double usig_inv;            // 1/sigma when every reading has the same sigma, else 0
  // range-and-bearing readings (null obear = range only); landmark-major like orec
  const double* obear;        // [Ncap] measured bearing
  const double* obsd;         // [Ncap] its std deviation
  const double* obisd;        // [Ncap] 1 / that
  double ubsig_inv;           // 1/sigma of the bearings when they all have the same sigma, else 0
  // ---- chains --------------------------------------------------------------
  uint32_t C;
  double *relx, *rely, *relth;  // [C][Tcap]   relative pose of state edge t
  double* clp;                  // [C][Tcap]   the control model's log density of that relative pose
  PoseXY* pxy;                  // [C][Tcap+1] world translation of pose k
  double* pth;                  // [C][Tcap+1] its (unwrapped) angle
  double *sw, *fw;              // [C][Tcap], [C][Fcap] weight trees
  double *fex, *fey;            // [C][Fcap]   landmark estimate relative to its parent pose
  uint32_t* fpar;               // [C][Fcap]   parent reading, as position inside the landmark's list
  uint32_t* fcnt;               // [C][Fcap]   readings of the landmark seen so far
  uint32_t* fpt;                // [C][Fcap]   parent timestep
  // [C][Fcap] the parent's world pose: a copy of pose-table entry fpt, rewritten by the same
  // arithmetic whenever that entry is, so a proposal reads its landmarks' parents with neighbouring
  // lanes on neighbouring addresses instead of one scattered table line per landmark; with it the
  // cosine and sine of its angle (taken anew whenever the angle changes)
  double *fpx, *fpy, *fpth, *fpc, *fps;
  // [C][Fcap] the parent READING as a distribution (measured range and its sigma; bearing and its sigma,
  // null on a range-only handle): a copy of what foff[f] + fpar indexes, so that a landmark proposal
  // finds its distribution in one load instead of three dependent ones
  double *fpmu, *fpsd, *fpbmu, *fpbsd;
  double *fplsd, *fplbsd;  // [C][Fcap] log of fpsd / fpbsd, taken when the copy is (fplbsd null on a range-only handle)

  uint32_t* flast;              // [C][Fcap]   timestep of the latest reading seen
  uint32_t* nf;                 // [C]         landmarks known
  double* ll;                   // [C]         chain log likelihood
  uint32_t* mt;                 // [C][624]
  int32_t* mti;                 // [C]
  int32_t* nflag;               // [C]
  double* nsaved;               // [C]
  unsigned long long *n_updates, *n_accepts, *n_units;  // [C]
  unsigned long long *n_state_props, *n_moved;          // [C] state-edge proposals; poses rewritten by commits
  uint32_t* errflag;            // [C]
  mcmcgpu_step_record* trace;   // [C][trace_cap] or null
  uint32_t trace_cap;
  uint32_t* trace_n;  // [C]
  int32_t uniform_first;
  mcmcgpu_chain_summary* snap_out;  // [intervals][C] per-interval summaries of the current launch, or null
  uint32_t snap_every;
  // tempered replicas: chain c samples p(x)^beta[c]; slot_chain[l*R + s] = the chain that holds
  // temperature slot s of ladder l
  double* beta;          // [C]
  double* llc;           // [C] the control-model part of ll (tempering leaves it at full weight)
  uint32_t* slot_chain;  // [C]
  // split mode: a chain's update as a sequence of kernels, its reading stream cut into slices
  struct ChainCtl* ctl;  // [C] what the pieces of one update hand to each other
  uint32_t* sp_seglo;    // [C][Fcap]   first record of landmark f's affected readings
  uint32_t* sp_segpre;   // [C][Fcap+1] their count, then (after the scan) the prefix sums
  double* sp_segw;       // [C][Fcap][4] old and moved landmark position
  double* sp_partial;    // [C][max_slices]
  uint32_t* sp_slice_seg;  // [C][max_slices+1] segment of each slice's first reading; entry n = last segment
  uint32_t max_slices;
};

// The scalars an update's kernels hand on in split mode (in the persistent kernel they live in LDS).
struct ChainCtl {
  double Ws, Wf, new_pll, old_pll, q[3];
  int32_t w_dirty, op, pending, use_weight;
  uint32_t idx, total;
  // an accepted state edge whose pose-table update is left to split_commit_kernel: poses from
  // move_from on turn by (move_sin, move_cos) about move_pivot and land at move_to; angle + move_dth
  uint32_t move_from;  // 0 = nothing to move
  double move_pivot[2], move_to[2], move_sin, move_cos, move_dth;
};

constexpr int kMaxSeg = 256;   // landmark segments evaluated per pass
constexpr int kActCap = 1024;  // most landmarks examined per round
constexpr int kMaxWaves = 16;
constexpr uint32_t kMaxRetry = 1u << 16;

// Slices a chain's stream of `total` readings is cut into in split mode: as many as the launch
// offers, but no slice shorter than kSplitMinChunk readings -- a small proposal (one landmark's
// list) occupies a few workgroups, a large one all of them.  It depends on the chain's own
// proposal only, so a chain's arithmetic does not depend on which other chains share the launch.
constexpr uint32_t kSplitMinChunk = 1024;
constexpr uint32_t kSplitWindow = 1024;  // segments of one slice whose prefix sums are staged in LDS
__device__ __forceinline__ uint32_t active_slices(uint32_t total, uint32_t offered) {
  const uint32_t n = (total + kSplitMinChunk - 1) / kSplitMinChunk;
  return n < 1 ? 1 : (n > offered ? offered : n);
}

// Segment (landmark) that holds the first reading of every slice of a chain's stream, found in the
// exclusive prefix sums `pre_at(0 .. nf-1)`; entry `n` (one past the last slice) is the last
// segment.  The stream kernel stages the prefix sums of its own segments [entry s, entry s+1] in
// LDS instead of searching all of them in global memory.
template <class PreAt>
__device__ __forceinline__ void write_slice_segments(uint32_t* out, uint32_t nf, uint32_t total, uint32_t offered,
                                                     PreAt pre_at) {
  const uint32_t n = active_slices(total, offered), per = (total + n - 1) / n;
  for (uint32_t s = threadIdx.x; s <= n; s += blockDim.x) {
    uint32_t a = 0;
    if (s == n) {
      a = nf ? nf - 1 : 0;
    } else if (nf) {  // last f with pre[f] <= i
      const uint32_t i = s * per;
      uint32_t b = nf;
      while (a + 1 < b) {
        const uint32_t m = (a + b) >> 1;
        if (pre_at(m) <= i)
          a = m;
        else
          b = m;
      }
    }
    out[s] = a;
  }
}

enum Op : int { OP_END = 0, OP_EVAL_STATE = 1, OP_EVAL_FEATURE = 2, OP_COMMIT = 3 };
enum Pending : int { PEND_NONE = 0, PEND_UPDATE = 1, PEND_REPARENT = 2 };
enum Stage : int { ST_EDGE = 0, ST_ARRIVALS = 1, ST_STEPS = 2, ST_NEXT = 3, ST_END_STEPS = 4, ST_DONE = 5 };

// Diagnostic builds (-DMG_PHASE_TIMING) stamp the shader clock around the phases of a
// step in lane 0 and sum the differences per chain; normal builds compile the stamps out.
#ifdef MG_PHASE_TIMING
#define MG_STAMP(v) const long long v = clock64()
#define MG_PHASE_ADD(S, slot, a, b)                     \
  do {                                                  \
    if (threadIdx.x == 0) (S).phase[slot] += (b) - (a); \
  } while (0)
// inside lane 0's serial code: time since the last mark goes to `slot`
#define MG_MARK_BEGIN() long long mark_last_ = clock64()
#define MG_MARK(S, slot)                      \
  do {                                        \
    const long long now_ = clock64();         \
    (S).phase[slot] += now_ - mark_last_;     \
    mark_last_ = now_;                        \
  } while (0)
__device__ unsigned long long g_phase_cycles[16];
#else
#define MG_STAMP(v)
#define MG_PHASE_ADD(S, slot, a, b)
#define MG_MARK_BEGIN()
#define MG_MARK(S, slot)
#endif

// What the serial program (the "driver") keeps between its steps.  When the program is a called piece run
// by one lane it lives in the chain's LDS block (Shared::drv); when it is compiled into the kernel and run
// by a whole wavefront on the same data (advance_kernel's SOLO variant) it is a local object of the kernel,
// i.e. registers: a lone wavefront waits out every LDS round trip, and the program reads these all the time.
// One update as selection and proposal leave it for the evaluation (mcmc_slam.hpp:339-382).
struct Request {
  int op;
  uint32_t idx;
  double q[3];
  double new_pll, old_pll;
  double which;  // the uniform that chose between the two weight trees (:346)
  int plain;     // the index was found at the first draw (no retry, :349-364)
};
// A landmark proposal's readings fetched while the proposal is still being drawn (one wavefront per chain,
// range-only readings, a list of at most 64): lane L holds reading L of the landmark's list and the pose it
// was taken at, every lane the landmark's parent frame and current estimate.  The evaluation is then
// arithmetic on registers and one reduction -- no set-up, no LDS tables, no memory round trip after the
// proposal (three in a row otherwise: landmark, records, poses).
constexpr int kDirtyStates = 1, kDirtyLandmarks = 2;
struct ListAhead {
  int valid;
  uint32_t count;
  Frame parent;
  Vec2 current;    // the landmark's estimate, relative to its parent pose
  double mu, k;    // this lane's reading: measured range, 1 / (2 sigma^2)
  PoseXY pose;     // where it was taken
};
struct DriverState {
  int stage;
  uint32_t tn, upto, arr, arr_end, steps_left, steps_per_ts, end_left, end_done, end_inflight;
  uint32_t awaiting;  // an evaluation has been posted and its result not yet taken (step_with)
  uint32_t Tcur, nf;
  double ll;
  double beta;    // inverse temperature of this chain (1 = the reference's sampler)
  double llc;     // control-model part of ll
  double Ws, Wf;  // weight totals of the two trees
  int w_dirty;    // which of them are out of date: kDirtyStates | kDirtyLandmarks
  // LDS copies of the tops of the two weight trees (advance_kernel, dynamic LDS): byte address in
  // LDS of each copy (0 = none) and the stride 2^shift of the nodes it holds
  uint32_t sw_top, fw_top, sw_shift, fw_shift;
  int pending, use_weight;
  double new_pll, old_pll;
  uint32_t rp_obs;
  unsigned long long n_upd, n_acc, n_units, n_state, n_moved;
  uint32_t failed;
  // request posted by the driver
  int op;
  uint32_t idx;
  double q[3];
  int accept;
  // the NEXT update, selected and proposed ahead of time on the assumption that the update in flight is
  // rejected (advance_kernel's SPEC variant); driver_begin_update takes it instead of drawing
  int spec_ready;
  Request spec;
  ListAhead list;  // with a landmark request posted by driver_begin_update: see ListAhead
};

// An update as far as its accept test is concerned (mcmc_slam.hpp:385-408).
struct TestInputs {
  int is_state, use_weight;
  double new_pll, old_pll;
  double beta;        // the chain's inverse temperature
  double weight_sum;  // both trees' totals (:400-401), while the update is in flight
  double u;           // the uniform of :408 (nothing else draws between a proposal and that uniform)
};

// A request as the SPEC variant's first wavefront posts it for the others.
struct Posted {
  int op;
  uint32_t idx;
  double q[3];
  uint32_t Tcur, nf;
  uint32_t join;  // with a commit: every wavefront meets again when the pose table is moved
  TestInputs test;  // with an evaluation: what its accept test needs besides the evaluation's result
};

struct Shared {
  long long phase[16];
  RngState rng;
  DriverState drv;
  // evaluation scratch
  uint32_t nact;
  uint32_t act[kActCap];
  uint32_t segpre[kMaxSeg + 1];
  uint32_t seglo[kMaxSeg];
  // per segment, what a stream's cursor needs when it enters it, in one 8-byte read: where the segment
  // ends in the flattened index space (segpre[k + 1]) and what turns such an index into a record
  // position (seglo[k] - segpre[k])
  uint2 segcur[kMaxSeg];
  double segw[kMaxSeg][4];  // old landmark (x,y), new landmark (x,y), world frame
#ifdef MG_DIAG_RUN_RECORDS  // diagnostic builds (tools/window_probe.sh): timestep of a segment's first reading
  uint32_t segt[kMaxSeg];
#endif
  uint32_t wcount[kMaxWaves];
  double red[kMaxWaves];
  // the four rigid motions of a state-edge proposal (ProposalFrames), kept here between the rounds of its
  // evaluation: sixteen doubles held in registers across the reading stream cost it a spill
  double frames[16];
  // advance_kernel's SPEC variant: the barrier of the wavefronts after the first (TailWaves), and what they
  // hand the first one when an evaluation is done: its number, its sum and the readings it went through
  uint32_t tail_arrivals, eval_arrived;
  unsigned long long eval_units;
  Posted post[2];
  // ... and the accept test as the last of them decides it: the number of the evaluation it belongs to
  // (written last), the decision, the new edge weight, the evaluation's sum
  uint32_t verdict_of;
  int verdict_accepted;
  double verdict_new_weight, verdict_sum;
};

// Separately compiled pieces (see advance_kernel) take the kernel-argument block and the chain's
// LDS block as address-space-qualified pointers.  A kernel that reaches a piece has its Dev as
// FIRST parameter, so the argument block starts with it, and reads its Dev THROUGH that block
// (MG_KERNEL_DEV) -- a by-value parameter whose address is taken would be a private copy, and a
// piece handed that address as an argument-block pointer would read garbage.
using DevArgs = const __attribute__((address_space(4))) Dev*;
using SharedLds = __attribute__((address_space(3))) Shared*;
#define MG_PIECE __device__ __noinline__
// only inside a __global__ function whose first parameter is the Dev
#define MG_KERNEL_DEV(name) const Dev& name = *(const Dev*)(DevArgs)__builtin_amdgcn_kernarg_segment_ptr()

// ---------------------------------------------------------------------------
// cumulative pose access

MG_DEV Pose world_pose(const Dev& d, uint32_t c, uint32_t k) {
  const size_t i = size_t(c) * (d.Tcap + 1) + k;
  const PoseXY p = d.pxy[i];
  return Pose{p.x, p.y, d.pth[i]};
}

// The parent pose of landmark entry fi = c * Fcap + f, from its copy beside the landmark.
MG_DEV Frame parent_frame(const Dev& d, size_t fi) { return Frame{d.fpx[fi], d.fpy[fi], d.fpc[fi], d.fps[fi]}; }
// Takes the copy anew from the pose table (the landmark got a parent, or the table was rebuilt).
MG_DEV void keep_parent_pose(const Dev& d, uint32_t c, size_t fi, uint32_t pt) {
  const Pose p = world_pose(d, c, pt);
  const Frame f = frame_of(p);
  d.fpx[fi] = p.x;
  d.fpy[fi] = p.y;
  d.fpth[fi] = p.th;
  d.fpc[fi] = f.c;
  d.fps[fi] = f.s;
}

// Appends pose Tcur+1 = P[Tcur] o rel (lane 0, graph growth).
MG_DEV void append_pose(const Dev& d, uint32_t c, uint32_t Tcur, const Pose& rel) {
  const size_t i = size_t(c) * (d.Tcap + 1) + Tcur;
  const Pose n = compose(world_pose(d, c, Tcur), rel);
  d.pxy[i + 1] = PoseXY{n.x, n.y};
  d.pth[i + 1] = n.th;
}

// The chain's two weight trees as the driver sees them: global arrays, with their tops in LDS when
// the kernel has set copies up (see advance_kernel).
MG_DEV wt::Cached state_tree(const Dev& d, uint32_t c, const DriverState& D) {
  return wt::Cached{d.sw + size_t(c) * d.Tcap, D.sw_top, D.sw_shift};
}
MG_DEV wt::Cached landmark_tree(const Dev& d, uint32_t c, const DriverState& D) {
  return wt::Cached{d.fw + size_t(c) * d.Fcap, D.fw_top, D.fw_shift};
}

// ---------------------------------------------------------------------------
// workgroup helpers

// The lanes that run the data-parallel parts of an update together: the whole workgroup, or -- while the
// first wavefront runs ahead with the serial program (advance_kernel's SPEC variant) -- the wavefronts
// after the first, which then meet at a barrier of their own in LDS (the hardware barrier counts every
// wavefront of the workgroup).
struct WholeBlock {
  MG_DEV uint32_t tid() const { return threadIdx.x; }
  MG_DEV uint32_t size() const { return blockDim.x; }
  MG_DEV void sync() { __syncthreads(); }
  MG_DEV double sum(double v, struct Shared& S);  // over the team, in a fixed order; valid in its first lane
};
struct TailWaves {
  uint32_t* arrivals;   // in LDS, zero at launch: wavefronts that have arrived at any barrier so far
  uint32_t passed = 0;  // barriers this wavefront has passed
  MG_DEV explicit TailWaves(uint32_t* counter) : arrivals(counter) {}
  MG_DEV uint32_t tid() const { return threadIdx.x - 64u; }
  MG_DEV uint32_t size() const { return blockDim.x - 64u; }
  // Leaves every wavefront's sum in S.red for the first wavefront of the workgroup to add up (in the order
  // WholeBlock::sum adds them) and counts the wavefront in S.eval_arrived; returns nothing of use.
  MG_DEV double sum(double v, struct Shared& S);
  MG_DEV void sync() {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t nw = (blockDim.x >> 6) - 1u;
    ++passed;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (nw > 1u) {
      if ((threadIdx.x & 63u) == 0) __hip_atomic_fetch_add(arrivals, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      const uint32_t want = passed * nw;
      // every wavefront of a workgroup is resident, so the ones still to arrive make progress
      while (int32_t(__hip_atomic_load(arrivals, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) - want) < 0) {
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#endif
  }
};

// Inclusive prefix sums over a wavefront (lane 63 ends up with the wavefront's sum) by data-parallel
// primitives: lanes shifted inside their row of 16 by 1, 2, 4, 8, then lane 15 of a row added to the
// row after it, then lane 31 to the upper half.  Six register-to-register steps; the shuffle
// instructions these replace go through the LDS crossbar and cost a lone wavefront an LDS latency each.
#if defined(__HIP_DEVICE_COMPILE__)
template <int CTRL, int ROWS>
MG_DEV uint32_t dpp_take(uint32_t v) {  // the value of the lane that CTRL names, 0 where there is none
  return uint32_t(__builtin_amdgcn_update_dpp(0, int(v), CTRL, ROWS, 0xF, false));
}
template <int CTRL, int ROWS>
MG_DEV double dpp_take(double v) {
  return __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROWS, 0xF, false),
                          __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROWS, 0xF, false));
}
template <class T>
MG_DEV T wave_prefix_sums(T v) {
  v += dpp_take<0x111, 0xF>(v);  // row_shr:1
  v += dpp_take<0x112, 0xF>(v);  // row_shr:2
  v += dpp_take<0x114, 0xF>(v);  // row_shr:4
  v += dpp_take<0x118, 0xF>(v);  // row_shr:8
  v += dpp_take<0x142, 0xA>(v);  // row_bcast:15 into rows 1 and 3
  v += dpp_take<0x143, 0xC>(v);  // row_bcast:31 into rows 2 and 3
  return v;
}
#else
template <class T>
MG_DEV T wave_prefix_sums(T v) { return v; }
#endif

template <class Team>
MG_DEV uint32_t block_scan_u32(uint32_t v, Shared& S, uint32_t& total, Team& tm) {
  const int lane = tm.tid() & 63, wave = tm.tid() >> 6, nwaves = tm.size() >> 6;
  v = wave_prefix_sums(v);
  if (lane == 63) S.wcount[wave] = v;
  tm.sync();
  uint32_t before = 0, all = 0;
  for (int w = 0; w < nwaves; ++w) {
    if (w == wave) before = all;
    all += S.wcount[w];
  }
  total = all;
  tm.sync();
  return v + before;  // inclusive
}
MG_DEV uint32_t block_scan_u32(uint32_t v, Shared& S, uint32_t& total) {
  WholeBlock tm;
  return block_scan_u32(v, S, total, tm);
}

// Sum over the team in a fixed order; the result is valid in its first lane.
template <class Team>
MG_DEV double block_sum(double v, Shared& S, Team& tm) {
  const int lane = tm.tid() & 63, wave = tm.tid() >> 6, nwaves = tm.size() >> 6;
  v = wave_prefix_sums(v);
  if (lane == 63) S.red[wave] = v;
  tm.sync();
  double r = 0.0;
  if (tm.tid() == 0)
    for (int w = 0; w < nwaves; ++w) r += S.red[w];
  tm.sync();
  return r;
}
MG_DEV double block_sum(double v, Shared& S) {
  WholeBlock tm;
  return block_sum(v, S, tm);
}
MG_DEV double WholeBlock::sum(double v, Shared& S) { return block_sum(v, S, *this); }
MG_DEV double TailWaves::sum(double v, Shared& S) {
#if defined(__HIP_DEVICE_COMPILE__)
  v = wave_prefix_sums(v);
  if ((threadIdx.x & 63u) == 63u) {
    S.red[tid() >> 6] = v;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __hip_atomic_fetch_add(&S.eval_arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
#endif
  return 0.0;
}
// What TailWaves::sum left, added up by the reader.
MG_DEV double tail_waves_sum(const Shared& S) {
  double r = 0.0;
  const int nwaves = int(blockDim.x >> 6) - 1;
  for (int w = 0; w < nwaves; ++w) r += S.red[w];
  return r;
}

// |(dx, dy)| to within 2 ulp (measured: 2.2e-16 relative at worst): v_rsq_f64 seed y = (1 + e) / sqrt(x),
// good to about 24 bits, then ONE third-order correction: with s = x y = sqrt(x) (1 + e) and
// r = 1/2 - (y/2) s = -e - e^2/2 the root is s (1 + r + 3/2 r^2) up to e^3 -- six instructions after the
// seed.  (Two second-order Goldschmidt steps: seven instructions, 2.8e-16, 1.3 % slower on the whole
// benchmark; one such step: four instructions, 4e-15, 2.2 % faster -- not taken, the stream's accuracy
// stays at the rounding level.)  The correctly rounded sqrt the compiler emits costs about twice the
// instructions.  Inputs are lengths in metres; the 1e-280 keeps a zero distance finite (it vanishes
// against any squared distance above 1e-264).
MG_DEV double range_of(double dx, double dy) {
#if defined(__HIP_DEVICE_COMPILE__)
  const double x = fma(dx, dx, fma(dy, dy, 1e-280));
  const double y = __builtin_amdgcn_rsq(x);
  const double s = x * y, h = 0.5 * y;
  const double r = fma(-h, s, 0.5);
  return fma(s, r * fma(1.5, r, 1.0), s);
#else
  return sqrt(dx * dx + dy * dy);
#endif
}

// range_of for a batch of vectors: accuracy self-test of the device square root
__global__ void range_selftest_kernel(uint32_t n, const double* dx, const double* dy, double* out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = range_of(dx[i], dy[i]);
}

// a - 2 pi rint(a / 2 pi): the angle brought into [-pi, pi] (2 pi in two parts, so the reduction
// of an angle of hundreds of radians keeps its last bits).  The reference's wrap_angle
// (src/utility/geometry.hpp:8-16) maps to [-pi, pi); the two differ only at +-pi itself, and the
// residual is squared.
MG_DEV double wrap_residual(double a) {
  constexpr double kTwoPiHi = 6.283185307179586232, kTwoPiLo = 2.4492935982947064e-16;
  const double k = rint(a * (1.0 / kTwoPiHi));
  return fma(-k, kTwoPiLo, fma(-k, kTwoPiHi, a));
}

// ---------------------------------------------------------------------------
// The hot loop: sum over the prepared segments of
//   ll(|w_new - t_s|) - ll(|w_old - t_s|)                      (mcmc_slam.hpp:487-502)
// With z = (r - mu)/sigma the term is -(zn^2 - zo^2)/2, evaluated as
// (ro - rn) (ro + rn - 2 mu) / (2 sigma^2): the same number without cancelling two large squares.
// Per reading: one 16-byte record, one 16-byte pose.  It runs as a software
// pipeline.  A reading needs two dependent loads (its record,
// then the pose of the record's timestep), each about 1.5 us away under load,
// against ~1000 cycles of fp64 work per group of readings; with four
// wavefronts per SIMD that latency is only hidden if both loads of LATER
// groups are in flight while a group is evaluated:
//   step s:  copy out range/segment of group s        (registers R[s&1] -> C)
//            request the poses of group s+1           (records R[(s+1)&1] -> P[(s+1)&1])
//            find and request the records of group s+2 (-> R[s&1])
//            evaluate group s                         (P[s&1], C)
// The two buffer parities are written out as two half-steps so every buffer
// is a fixed set of registers.  The segment cursor lives in registers; LDS is
// read again only when a lane crosses into another segment.
template <bool USIG>
struct RatioStream {
  static constexpr int U = 4;  // readings per lane per group
  struct Recs {
    uint32_t t[U], sg[U], o[U];
    double mu[U];
  };
  struct Poses {
    PoseXY p[U];
  };
  const Dev& d;
  const Shared& S;
  const PoseXY* __restrict__ pxy;
  const uint32_t total, stride, first;
  const double half_isd2;
  uint32_t cur = 0, seg_end, seg_base;
  double acc = 0.0;
#ifdef MG_DIAG_RUN_RECORDS
  uint32_t seg_tbase;
#endif

  // the team's lane `first_` of `stride_`
  MG_DEV RatioStream(const Dev& d_, uint32_t c, const Shared& S_, uint32_t nseg, uint32_t first_, uint32_t stride_)
      : d(d_), S(S_), pxy(d_.pxy + size_t(c) * (d_.Tcap + 1)), total(S_.segpre[nseg]), stride(stride_), first(first_),
        half_isd2(0.5 * d_.usig_inv * d_.usig_inv), seg_end(S_.segcur[0].x), seg_base(S_.segcur[0].y) {
#ifdef MG_DIAG_RUN_RECORDS
    seg_tbase = S_.segt[0];
#endif
  }

  MG_DEV void request_records(uint32_t i0, Recs& r) {
    if (i0 >= total) return;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t i = i0 + u * stride;
      if (i < total) {
        if (i >= seg_end) {
          uint2 entry;
          do {
            ++cur;
            entry = S.segcur[cur];
          } while (i >= entry.x);
          seg_end = entry.x;
          seg_base = entry.y;
#ifdef MG_DIAG_RUN_RECORDS
          seg_tbase = S.segt[cur] - S.segpre[cur];
#endif
        }
        r.o[u] = seg_base + i;
      } else {
        r.o[u] = S.seglo[0];
      }
      r.sg[u] = cur;
#ifdef MG_DIAG_RUN_RECORDS
      // what a stream over runs of consecutive timesteps would do: the timestep is arithmetic on the
      // segment (here: as if every segment were one run -- wrong results), the record 8 bytes
      r.t[u] = i < total ? min(seg_tbase + i, d.T) : 0u;
#endif
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
#if defined(MG_DIAG_RUN_RECORDS)
      r.mu[u] = d.osd[r.o[u]];  // an 8-byte array with one entry per reading stands in for the ranges
#elif defined(MG_DIAG_RECORD_WINDOW)  // diagnostic builds (tools/window_probe.sh): wrong results, known traffic
      const StreamRec rec = d.osr[r.o[u] & 4095u];
      r.t[u] = rec.t;
      r.mu[u] = rec.mu;
#else
      const StreamRec rec = d.osr[r.o[u]];
      r.t[u] = rec.t;
      r.mu[u] = rec.mu;
#endif
    }
  }
  MG_DEV void request_poses(uint32_t i0, const Recs& r, Poses& p) {
    if (i0 >= total) return;
#pragma unroll
#ifdef MG_DIAG_POSE_WINDOW
    for (int u = 0; u < U; ++u) p.p[u] = pxy[r.t[u] & 1023u];
#else
    for (int u = 0; u < U; ++u) p.p[u] = pxy[r.t[u]];
#endif
  }
  MG_DEV void step(uint32_t i0, Recs& r_this, const Recs& r_next, const Poses& p_this, Poses& p_next) {
    const uint32_t group = U * stride;
    double mu[U], k[U];
    uint32_t sg[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      mu[u] = r_this.mu[u];
      sg[u] = r_this.sg[u];
      k[u] = half_isd2;
      if (!USIG) {
        const double isd = d.oisd[r_this.o[u]];
        k[u] = 0.5 * isd * isd;
      }
    }
    request_poses(i0 + group, r_next, p_next);
    request_records(i0 + 2 * group, r_this);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const double* w = S.segw[sg[u]];
      const double wox = w[0], woy = w[1], wnx = w[2], wny = w[3];  // both positions asked for before either is used
      const double ro = range_of(wox - p_this.p[u].x, woy - p_this.p[u].y);
      const double rn = range_of(wnx - p_this.p[u].x, wny - p_this.p[u].y);
      const double term = (ro - rn) * fma(-2.0, mu[u], ro + rn) * k[u];
      if (i0 + u * stride < total) acc += term;
    }
  }
  MG_DEV double run() {
    const uint32_t group = U * stride;
    Recs r0, r1;
    Poses p0, p1;
    request_records(first, r0);
    request_records(first + group, r1);
    request_poses(first, r0, p0);
    for (uint32_t i0 = first; i0 < total; i0 += 2 * group) {
      step(i0, r0, r1, p0, p1);
      if (i0 + group < total) step(i0 + group, r1, r0, p1, p0);
    }
    return acc;
  }
};

// The same pipeline for range-AND-bearing readings (range_bearing_model,
// src/planar_robot/range_bearing_model.hpp:21-117).  Seen from pose s the landmark is at
// R(-th_s) (w - t_s): its range is |w - t_s| and its bearing atan2(w - t_s) - th_s, so a record
// brings its measured bearing along and a pose its angle -- per reading 16 B record + 8 B bearing +
// 16 B translation + 8 B angle = 48 bytes.  With z_r = (r - mu_r) / sigma_r and z_b =
// wrap(b - mu_b) / sigma_b the term is -(z_r'^2 + z_b'^2) / 2 + (z_r^2 + z_b^2) / 2, each
// difference of squares taken as a product so that nothing large cancels.  USIG here means that the ranges and the
// bearings each have one sigma for all readings.  A struct of its own, so that the range-only
// stream's code does not change with it (sharing one template cost the range-only stream 5 %).
template <bool USIG>
struct BearingStream {
  static constexpr int U = 2;  // readings per lane per group (one: -4 %; four spill: -50 %)
  struct Recs {
    uint32_t t[U], sg[U], o[U];
    double mu[U];
    double bm[U];
  };
  struct Poses {
    PoseXY p[U];
    double th[U];
  };
  const Dev& d;
  const Shared& S;
  const PoseXY* __restrict__ pxy;
  const double* __restrict__ pth;
  const uint32_t total, stride, first;
  const double half_isd2, half_ibsd2;
  uint32_t cur = 0, seg_end, seg_base;
  double acc = 0.0;

  MG_DEV BearingStream(const Dev& d_, uint32_t c, const Shared& S_, uint32_t nseg, uint32_t first_, uint32_t stride_)
      : d(d_), S(S_), pxy(d_.pxy + size_t(c) * (d_.Tcap + 1)), pth(d_.pth + size_t(c) * (d_.Tcap + 1)),
        total(S_.segpre[nseg]), stride(stride_), first(first_), half_isd2(0.5 * d_.usig_inv * d_.usig_inv),
        half_ibsd2(0.5 * d_.ubsig_inv * d_.ubsig_inv), seg_end(S_.segpre[1]), seg_base(S_.seglo[0]) {}

  MG_DEV void request_records(uint32_t i0, Recs& r) {
    if (i0 >= total) return;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t i = i0 + u * stride;
      if (i < total) {
        if (i >= seg_end) {
          do {
            ++cur;
            seg_end = S.segpre[cur + 1];
          } while (i >= seg_end);
          seg_base = S.seglo[cur] - S.segpre[cur];
        }
        r.o[u] = seg_base + i;
      } else {
        r.o[u] = S.seglo[0];
      }
      r.sg[u] = cur;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const StreamRec rec = d.osr[r.o[u]];
      r.t[u] = rec.t;
      r.mu[u] = rec.mu;
      r.bm[u] = d.obear[r.o[u]];
    }
  }
  MG_DEV void request_poses(uint32_t i0, const Recs& r, Poses& p) {
    if (i0 >= total) return;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      p.p[u] = pxy[r.t[u]];
      p.th[u] = pth[r.t[u]];
    }
  }
  MG_DEV void step(uint32_t i0, Recs& r_this, const Recs& r_next, const Poses& p_this, Poses& p_next) {
    const uint32_t group = U * stride;
    double mu[U], k[U], bm[U], kb[U];
    uint32_t sg[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      mu[u] = r_this.mu[u];
      sg[u] = r_this.sg[u];
      bm[u] = r_this.bm[u];
      k[u] = half_isd2;
      kb[u] = half_ibsd2;
      if (!USIG) {
        const double isd = d.oisd[r_this.o[u]], ibsd = d.obisd[r_this.o[u]];
        k[u] = 0.5 * isd * isd;
        kb[u] = 0.5 * ibsd * ibsd;
      }
    }
    request_poses(i0 + group, r_next, p_next);
    request_records(i0 + 2 * group, r_this);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const double* w = S.segw[sg[u]];
      const double ox = w[0] - p_this.p[u].x, oy = w[1] - p_this.p[u].y;
      const double nx = w[2] - p_this.p[u].x, ny = w[3] - p_this.p[u].y;
      const double ro = range_of(ox, oy);
      const double rn = range_of(nx, ny);
      const double eo = wrap_residual(atan2(oy, ox) - p_this.th[u] - bm[u]);
      const double en = wrap_residual(atan2(ny, nx) - p_this.th[u] - bm[u]);
      const double term = (ro - rn) * fma(-2.0, mu[u], ro + rn) * k[u] + (eo - en) * (eo + en) * kb[u];
      if (i0 + u * stride < total) acc += term;
    }
  }
  MG_DEV double run() {
    const uint32_t group = U * stride;
    Recs r0, r1;
    Poses p0, p1;
    request_records(first, r0);
    request_records(first + group, r1);
    request_poses(first, r0, p0);
    for (uint32_t i0 = first; i0 < total; i0 += 2 * group) {
      step(i0, r0, r1, p0, p1);
      if (i0 + group < total) step(i0 + group, r1, r0, p1, p0);
    }
    return acc;
  }
};

// The stream of one pass.  RB (range-and-bearing readings) is a compile-time choice made where the
// kernel calls its evaluation piece: each model's evaluation is then a piece with registers of its
// own.  (As a run-time branch to a bearing piece from inside the range-only evaluation it cost the
// range-only stream 5 %: live values of the caller saved around a call it never makes.)
// The bearing stream is once more a piece inside its evaluation piece: inlined there, next to the
// landmark set-up, it spills (1.0e11 instead of 1.2e11 units/s).
MG_PIECE double piece_bearing_stream(DevArgs dp, SharedLds sp, uint32_t c, uint32_t nseg, uint32_t first, uint32_t stride) {
  const Dev& d = *(const Dev*)dp;
  const Shared& S = *(const Shared*)sp;
  // one sigma for all ranges and one for all bearings (what the reference's model holds,
  // range_bearing_model.hpp:21-117) -> no per-reading loads of them
  if (d.usig_inv > 0.0 && d.ubsig_inv > 0.0) return BearingStream<true>(d, c, S, nseg, first, stride).run();
  return BearingStream<false>(d, c, S, nseg, first, stride).run();
}

template <bool RB>
MG_DEV double eval_ratio(const Dev& d, uint32_t c, const Shared& S, uint32_t nseg, uint32_t first, uint32_t stride) {
  // `d` is the argument block itself here (see MG_KERNEL_DEV), S the chain's LDS block
  if (RB) return piece_bearing_stream((DevArgs)&d, (SharedLds)&S, c, nseg, first, stride);
  // every reading has the same sigma (what the reference's builder produces) -> no per-reading load of it
  if (d.usig_inv > 0.0) return RatioStream<true>(d, c, S, nseg, first, stride).run();
  return RatioStream<false>(d, c, S, nseg, first, stride).run();
}

// ---------------------------------------------------------------------------
// The from-scratch likelihood over prepared segments (slam_likelihood.hpp:48-54): all lanes stream
// the readings, gather the pose translation of each reading's timestep and accumulate
// ll(|w - t_s|) with its constants.  Plain four-deep loop; it runs once per read-out, not per proposal.
MG_DEV double full_ll_segments(const Dev& d, uint32_t c, const Shared& S, uint32_t nseg) {
  const uint32_t total = S.segpre[nseg];
  const PoseXY* __restrict__ pxy = d.pxy + size_t(c) * (d.Tcap + 1);
  const ObsRec* __restrict__ orec = d.orec;
  const double* __restrict__ osd = d.osd;
  double acc = 0.0;
  uint32_t cur = 0;
  // kUnroll readings per lane are in flight at once to cover the latency of the
  // dependent gather (record -> pose).
  constexpr int kUnroll = 4;
  const uint32_t stride = blockDim.x;
  for (uint32_t i0 = threadIdx.x; i0 < total; i0 += kUnroll * stride) {
    uint32_t o[kUnroll], sg[kUnroll];
    bool ok[kUnroll];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const uint32_t i = i0 + u * stride;
      ok[u] = i < total;
      if (ok[u]) {
        while (i >= S.segpre[cur + 1]) ++cur;
        o[u] = S.seglo[cur] + (i - S.segpre[cur]);
      } else {
        o[u] = S.seglo[0];
      }
      sg[u] = cur;
    }
    ObsRec rec[kUnroll];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) rec[u] = orec[o[u]];
    PoseXY p[kUnroll];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) p[u] = pxy[rec[u].t];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const double ro = range_of(S.segw[sg[u]][0] - p[u].x, S.segw[sg[u]][1] - p[u].y);
      const double sd = osd[o[u]];
      const double zo = (ro - rec[u].mu) / sd;
      const double term = -0.5 * (zo * zo) - 0.5 * log(2 * kPi) - log(sd);
      if (ok[u]) acc += term;
    }
  }
  return acc;
}

// The same for range-AND-bearing readings (range_bearing_model,
// src/planar_robot/range_bearing_model.hpp:21-117): the log density of src/utility/random.hpp:116-121
// in two dimensions, residuals z_r = (r - mu_r) / sigma_r, z_b = wrap(b - mu_b) / sigma_b.
MG_DEV double full_ll_segments_rb(const Dev& d, uint32_t c, const Shared& S, uint32_t nseg) {
  const uint32_t total = S.segpre[nseg];
  const PoseXY* __restrict__ pxy = d.pxy + size_t(c) * (d.Tcap + 1);
  const double* __restrict__ pth = d.pth + size_t(c) * (d.Tcap + 1);
  const ObsRec* __restrict__ orec = d.orec;
  const double* __restrict__ obear = d.obear;
  double acc = 0.0;
  uint32_t cur = 0;
  constexpr int kUnroll = 2;
  const uint32_t stride = blockDim.x;
  for (uint32_t i0 = threadIdx.x; i0 < total; i0 += kUnroll * stride) {
    uint32_t o[kUnroll], sg[kUnroll];
    bool ok[kUnroll];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const uint32_t i = i0 + u * stride;
      ok[u] = i < total;
      if (ok[u]) {
        while (i >= S.segpre[cur + 1]) ++cur;
        o[u] = S.seglo[cur] + (i - S.segpre[cur]);
      } else {
        o[u] = S.seglo[0];
      }
      sg[u] = cur;
    }
    ObsRec rec[kUnroll];
    double bm[kUnroll];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      rec[u] = orec[o[u]];
      bm[u] = obear[o[u]];
    }
    PoseXY p[kUnroll];
    double th[kUnroll];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      p[u] = pxy[rec[u].t];
      th[u] = pth[rec[u].t];
    }
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const double* w = S.segw[sg[u]];
      const double isd = d.oisd[o[u]], ibsd = d.obisd[o[u]];
      const double ox = w[0] - p[u].x, oy = w[1] - p[u].y;
      const double ro = range_of(ox, oy);
      const double eo = wrap_residual(atan2(oy, ox) - th[u] - bm[u]);
      const double zr = (ro - rec[u].mu) * isd, zb = eo * ibsd;
      const double term = -0.5 * (zr * zr + zb * zb) - log(2 * kPi) - (log(d.osd[o[u]]) + log(d.obsd[o[u]]));
      if (ok[u]) acc += term;
    }
  }
  return acc;
}

// Landmark edge (mcmc_slam.hpp:468-474): every reading of landmark f seen so
// far, old estimate against `e_new`, both relative to the parent pose.
template <bool RB, class Team>
MG_DEV double eval_feature_edge(const Dev& d, uint32_t c, Shared& S, uint32_t f, const Vec2& e_new,
                                unsigned long long& units, Team& tm) {
  const size_t fi = size_t(c) * d.Fcap + f;
  if (tm.tid() == 0) {
    const Frame Pp = parent_frame(d, fi);
    const Vec2 wo = act(Pp, Vec2{d.fex[fi], d.fey[fi]});
    const Vec2 wn = act(Pp, e_new);
    S.seglo[0] = d.foff[f];
    S.segpre[0] = 0;
    S.segpre[1] = d.fcnt[fi];
    S.segcur[0] = make_uint2(d.fcnt[fi], d.foff[f]);
    S.segw[0][0] = wo.x;
    S.segw[0][1] = wo.y;
    S.segw[0][2] = wn.x;
    S.segw[0][3] = wn.y;
  }
  tm.sync();
  units += S.segpre[1];
  const double acc = eval_ratio<RB>(d, c, S, 1, tm.tid(), tm.size());
  return tm.sum(acc, S);
}
template <bool RB>
MG_DEV double eval_feature_edge(const Dev& d, uint32_t c, Shared& S, uint32_t f, const Vec2& e_new,
                                unsigned long long& units) {
  WholeBlock tm;
  return eval_feature_edge<RB>(d, c, S, f, e_new, units, tm);
}

// upper_bound of timestep t in the time-sorted readings [lo, hi) of landmark f.  Two tables cut the
// search short.  Per landmark and block of 2^tblk_shift timesteps, a TimeBlock: when the block's
// readings are one run of consecutive timesteps (the usual case: one visit) the answer is
// arithmetic on that one entry, else the entry leaves the readings of one block to search.  And,
// for timesteps the table does not cover yet (data appended since the last layout), the timestep of
// every 32nd reading, which narrows the range to at most 32 readings, so the search touches a few
// cache lines instead of log2(hi - lo) scattered ones.
// The table entry that split_readings consults for landmark f and timestep t (any entry when there is no table).
MG_DEV TimeBlock time_block_of(const Dev& d, uint32_t f, uint32_t t) {
  if (!d.tblk_n) return TimeBlock{0, 0, kNoRun, 0};
  const uint32_t q = t >> d.tblk_shift;
  return d.tblk[size_t(q < d.tblk_n ? q : d.tblk_n) * d.Fcap + f];
}
MG_DEV uint32_t split_readings(const Dev& d, uint32_t f, uint32_t lo, uint32_t hi, uint32_t t, const TimeBlock& tb) {
  uint32_t a = lo, b = hi;
  if (d.tblk_n) {
    const uint32_t q = t >> d.tblk_shift;
    const uint32_t ra = lo + tb.before;
    if (q < d.tblk_n) {
      if (tb.run_from != kNoRun) {  // readings ra, ra + 1, ... were taken at run_from, run_from + 1, ...
        uint32_t upto = t >= tb.run_from ? t - tb.run_from + 1 : 0;
        if (upto > tb.count) upto = tb.count;
        const uint32_t r = ra + upto;
        return r < hi ? r : hi;
      }
      const uint32_t rb = ra + tb.count;
      a = ra < hi ? ra : hi;
      b = rb < hi ? rb : hi;
    } else {
      a = ra < hi ? ra : hi;
    }
  }
  if (b - a > 64) {
    uint32_t pa = (a + 31) >> 5, pb = (b + 31) >> 5;  // pivots at positions 32*i in [a, b)
    const uint32_t pa0 = pa;
    while (pa < pb) {  // first pivot with timestep > t
      const uint32_t m = (pa + pb) >> 1;
      if (d.opiv[m] <= t)
        pa = m + 1;
      else
        pb = m;
    }
    if (pa > pa0) a = (pa - 1) << 5;  // reading 32*(pa-1) is <= t
    if ((pa << 5) < b) b = pa << 5;   // reading 32*pa is > t
  }
  // what is left spans a few lines of the timestep array: ask for them together, then search
  if (b - a > 32) {
    uint32_t touch = 0;
    for (uint32_t j = a + 32; j < b; j += 32) touch += d.otime[j];
    asm volatile("" ::"v"(touch));
  }
  while (a < b) {
    const uint32_t m = (a + b) >> 1;
    if (d.otime[m] <= t)
      a = m + 1;
    else
      b = m;
  }
  return a;
}

MG_DEV uint32_t split_readings(const Dev& d, uint32_t f, uint32_t lo, uint32_t hi, uint32_t t) {
  return split_readings(d, f, lo, hi, t, time_block_of(d, f, t));
}

// The four rigid motions of a state-edge proposal, each with the cosine and sine of its angle: poses
// t and t + 1, the proposed relative pose and its inverse.  The four sincos evaluations are
// independent, so lanes 4k .. 4k+3 of every wavefront take one each and exchange the results.
struct ProposalFrames {
  Frame Pt, Pt1, q, qinv;
};
MG_DEV ProposalFrames proposal_frames(const Pose& Pt, const Pose& Pt1, const Pose& q) {
  const uint32_t which = threadIdx.x & 3u;
  const double angle = which == 0 ? Pt.th : which == 1 ? Pt1.th : which == 2 ? q.th : -q.th;
  double sn, cs;
  sincos(angle, &sn, &cs);
  double c[4], s[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    c[k] = __shfl(cs, k, 4);
    s[k] = __shfl(sn, k, 4);
  }
  return ProposalFrames{Frame{Pt.x, Pt.y, c[0], s[0]}, Frame{Pt1.x, Pt1.y, c[1], s[1]}, Frame{q.x, q.y, c[2], s[2]},
                        inverse_frame(q, c[3], s[3])};
}

// State edge (mcmc_slam.hpp:432-466), observation part only.  Replacing the
// relative pose of edge t by q moves every pose after t rigidly.  For a
// landmark whose parent pose is after t the landmark moves with it and the
// readings taken up to t are affected; otherwise the landmark stays and the
// readings after t are affected.  Either way the affected readings compare
// |w_new - t_s| with |w_old - t_s| on the UNCHANGED pose table.
template <bool RB, class Team>
MG_DEV double eval_state_edge(const Dev& d, uint32_t c, Shared& S, uint32_t t, const Pose& q, uint32_t nf,
                              unsigned long long& units, Team& tm) {
  const uint32_t tid = tm.tid(), nthr = tm.size();
  const uint32_t segcap = nthr < uint32_t(kMaxSeg) ? nthr : uint32_t(kMaxSeg);
  const size_t fbase = size_t(c) * d.Fcap;
  if (nf <= 64) {
    // A wavefront's worth of landmarks: the team's first wavefront prepares them all, every landmark a
    // segment (those without affected readings an empty one), with the prefix sums taken inside the
    // wavefront -- one barrier instead of three.  Everything the set-up reads -- poses t and t + 1, the
    // lane's landmark, its time-block entry -- is asked for before anything is computed: one wait for
    // memory instead of three in a row (a lone wavefront has nothing else to do meanwhile).
    MG_STAMP(ts0);
    if (tid < 64) {
      const Pose Pt = world_pose(d, c, t), Pt1 = world_pose(d, c, t + 1);
      const uint32_t f = tid < nf ? tid : 0;
      const uint32_t lo = d.foff[f], cnt = d.fcnt[fbase + f], pt = d.fpt[fbase + f];
      const uint32_t first = d.ffirst[f], last = d.flast[fbase + f];
      const Vec2 e{d.fex[fbase + f], d.fey[fbase + f]};
      const Frame Pp = parent_frame(d, fbase + f);
      const TimeBlock tb = time_block_of(d, f, t);
#if defined(__HIP_DEVICE_COMPILE__)
      asm volatile("" ::"v"(Pt.x), "v"(Pt1.x), "v"(Pt.th), "v"(Pt1.th), "v"(lo), "v"(cnt), "v"(pt), "v"(first), "v"(last),
                   "v"(e.x), "v"(e.y), "v"(Pp.x), "v"(Pp.y), "v"(Pp.c), "v"(Pp.s), "v"(tb.before), "v"(tb.run_from));
#endif
      const ProposalFrames fr = proposal_frames(Pt, Pt1, q);
      uint32_t slo = lo, len = 0;
      Vec2 w{0.0, 0.0}, wn{0.0, 0.0};
      if (tid < nf && first <= t && t < last) {
        const uint32_t hi = lo + cnt;
        const uint32_t a = split_readings(d, f, lo, hi, t, tb);
        w = act(Pp, e);
        if (t < pt) {  // landmark hangs off a pose that moves
          wn = act(fr.Pt, act(fr.q, act_inverse(fr.Pt1, w)));
          len = a - lo;
        } else {  // poses after t move, seen from them the landmark moves by the inverse
          wn = act(fr.Pt1, act(fr.qinv, act_inverse(fr.Pt, w)));
          slo = a;
          len = hi - a;
        }
      }
      const uint32_t incl = wave_prefix_sums(len);
      if (tid < nf) {
        S.seglo[tid] = slo;
        S.segw[tid][0] = w.x;
        S.segw[tid][1] = w.y;
        S.segw[tid][2] = wn.x;
        S.segw[tid][3] = wn.y;
        S.segpre[tid + 1] = incl;
        S.segcur[tid] = make_uint2(incl, slo - (incl - len));
#ifdef MG_DIAG_RUN_RECORDS
        S.segt[tid] = len ? d.otime[slo] : 0u;
#endif
      }
      if (tid == 0) S.segpre[0] = 0;
    }
    tm.sync();
    units += S.segpre[nf];
    MG_STAMP(ts1);
    MG_PHASE_ADD(S, 1, ts0, ts1);
    const double sum = eval_ratio<RB>(d, c, S, nf, tid, nthr);
    MG_STAMP(ts2);
    MG_PHASE_ADD(S, 2, ts1, ts2);
    return tm.sum(sum, S);
  }
  {
    const ProposalFrames once = proposal_frames(world_pose(d, c, t), world_pose(d, c, t + 1), q);
    if (tid == 0) {
      const Frame* f4[4] = {&once.Pt, &once.Pt1, &once.q, &once.qinv};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        S.frames[4 * k] = f4[k]->x;
        S.frames[4 * k + 1] = f4[k]->y;
        S.frames[4 * k + 2] = f4[k]->c;
        S.frames[4 * k + 3] = f4[k]->s;
      }
    }
    tm.sync();
  }
  double acc = 0.0;
#ifdef MG_PHASE_TIMING
  long long tab = clock64();
#endif
  // B: one lane per landmark with affected readings: split its reading list at t, move it
  auto prepare = [&](uint32_t f, bool affected) -> uint32_t {
    const uint32_t lo = d.foff[f];
    uint32_t slo = lo, len = 0;
    Vec2 w{0.0, 0.0}, wn{0.0, 0.0};
    if (affected) {
      const double* fm = S.frames;
      const ProposalFrames fr{Frame{fm[0], fm[1], fm[2], fm[3]}, Frame{fm[4], fm[5], fm[6], fm[7]},
                              Frame{fm[8], fm[9], fm[10], fm[11]}, Frame{fm[12], fm[13], fm[14], fm[15]}};
      const uint32_t hi = lo + d.fcnt[fbase + f];
      const uint32_t pt = d.fpt[fbase + f];
      const Vec2 e{d.fex[fbase + f], d.fey[fbase + f]};
      const Frame Pp = parent_frame(d, fbase + f);
      const uint32_t a = split_readings(d, f, lo, hi, t);
      w = act(Pp, e);
      uint32_t shi;
      if (t < pt) {  // landmark hangs off a pose that moves
        wn = act(fr.Pt, act(fr.q, act_inverse(fr.Pt1, w)));
        slo = lo;
        shi = a;
      } else {  // poses after t move, seen from them the landmark moves by the inverse
        wn = act(fr.Pt1, act(fr.qinv, act_inverse(fr.Pt, w)));
        slo = a;
        shi = hi;
      }
      len = shi - slo;
    }
    S.seglo[tid] = slo;
    S.segw[tid][0] = w.x;
    S.segw[tid][1] = w.y;
    S.segw[tid][2] = wn.x;
    S.segw[tid][3] = wn.y;
#ifdef MG_DIAG_RUN_RECORDS
    S.segt[tid] = len ? d.otime[slo] : 0u;
#endif
    return len;
  };
  auto stream = [&](uint32_t len, uint32_t nseg) {
    uint32_t total;
    const uint32_t incl = block_scan_u32(len, S, total, tm);
    if (tid < nseg) {
      S.segpre[tid + 1] = incl;
      S.segcur[tid] = make_uint2(incl, S.seglo[tid] - (incl - len));  // this lane wrote seglo[tid] itself
    }
    if (tid == 0) S.segpre[0] = 0;
    tm.sync();
    units += total;
    MG_STAMP(tc0);
    MG_PHASE_ADD(S, 1, tab, tc0);
    // C: stream the affected readings
    acc += eval_ratio<RB>(d, c, S, nseg, tid, nthr);
    tm.sync();
    MG_STAMP(tc1);
    MG_PHASE_ADD(S, 2, tc0, tc1);
#ifdef MG_PHASE_TIMING
    tab = tc1;
#endif
  };
  if (nf <= segcap) {
    // few landmarks: every one is a segment, those without affected readings an empty one (the
    // readings come in the same order as below, so the sums are the same)
    uint32_t len = 0;
    if (tid < nf) len = prepare(tid, d.ffirst[tid] <= t && t < d.flast[fbase + tid]);
    stream(len, nf);
    return tm.sum(acc, S);
  }
  const uint32_t per = uint32_t(kActCap) / nthr;  // landmarks examined per lane and round (1..16)
  for (uint32_t f0 = 0; f0 < nf; f0 += per * nthr) {  // (a team of 192 lanes examines 960 per round)
    // A: which landmarks have affected readings?  first <= t < last
    uint32_t flags = 0;
    for (uint32_t k = 0; k < per; ++k) {
      const uint32_t f = f0 + tid * per + k;
      if (f < nf) {
        const uint32_t first = d.ffirst[f], last = d.flast[fbase + f];
        if (first <= t && t < last) flags |= 1u << k;
      }
    }
    const uint32_t cnt = __popc(flags);
    uint32_t nact;
    uint32_t at = block_scan_u32(cnt, S, nact, tm) - cnt;
    for (uint32_t k = 0; k < per; ++k)
      if ((flags >> k) & 1u) S.act[at++] = f0 + tid * per + k;
    tm.sync();
    for (uint32_t a0 = 0; a0 < nact; a0 += segcap) {
      const uint32_t nseg = (nact - a0) < segcap ? (nact - a0) : segcap;
      uint32_t len = 0;
      if (tid < nseg) len = prepare(S.act[a0 + tid], true);
      stream(len, nseg);
    }
  }
  return tm.sum(acc, S);
}
template <bool RB>
MG_DEV double eval_state_edge(const Dev& d, uint32_t c, Shared& S, uint32_t t, const Pose& q, uint32_t nf,
                              unsigned long long& units) {
  WholeBlock tm;
  return eval_state_edge<RB>(d, c, S, t, q, nf, units, tm);
}

// Accepted state edge t <- q: P'[k] = P[k] (k <= t), D o P[k] (k > t) with
// D = P[t] o q o P[t+1]^-1, applied as a rotation about the old P[t+1] so no
// large coordinates cancel: P'[k].t = A.t + R(D.th) (P[k].t - P[t+1].t) with
// A = P[t] o q the new pose t+1.
struct PoseMove {
  double pivot_x, pivot_y, to_x, to_y, s, c, dth;
};
MG_DEV PoseMove pose_move_of(const Dev& d, uint32_t c, uint32_t t, const Pose& q) {
  const Pose Pt = world_pose(d, c, t), Pt1 = world_pose(d, c, t + 1);
  const Pose A = compose(Pt, q);
  PoseMove m;
  m.dth = q.th - d.relth[size_t(c) * d.Tcap + t];
  sincos(m.dth, &m.s, &m.c);
  m.pivot_x = Pt1.x;
  m.pivot_y = Pt1.y;
  m.to_x = A.x;
  m.to_y = A.y;
  return m;
}
MG_DEV PoseXY moved(const PoseMove& m, const PoseXY& p) {
  const double dx = p.x - m.pivot_x, dy = p.y - m.pivot_y;
  return PoseXY{m.to_x + (m.c * dx - m.s * dy), m.to_y + (m.s * dx + m.c * dy)};
}
MG_DEV void move_pose(const Dev& d, size_t i, const PoseMove& m) {
  d.pxy[i] = moved(m, d.pxy[i]);
  d.pth[i] += m.dth;
}
// Poses base + k for k = first, first + stride, ... below `end`: four per lane are loaded before the
// first is stored, so a lane waits for memory once per four poses.
MG_DEV void move_poses(const Dev& d, size_t base, uint32_t first, uint32_t end, uint32_t stride, const PoseMove& m) {
  uint32_t k = first;
  for (; k < end && end - k > 3 * stride; k += 4 * stride) {
    PoseXY p[4];
    double th[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      p[u] = d.pxy[base + k + u * stride];
      th[u] = d.pth[base + k + u * stride];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      d.pxy[base + k + u * stride] = moved(m, p[u]);
      d.pth[base + k + u * stride] = th[u] + m.dth;
    }
  }
  for (; k < end; k += stride) move_pose(d, base + k, m);
}
// The copies of parent poses after edge t move exactly as their table entries do (all lanes).
MG_DEV void move_parent_poses(const Dev& d, uint32_t c, uint32_t t, uint32_t nf, const PoseMove& m, uint32_t first,
                              uint32_t stride) {
  const size_t fbase = size_t(c) * d.Fcap;
  for (uint32_t f = first; f < nf; f += stride) {
    const size_t fi = fbase + f;
    if (d.fpt[fi] > t) {
      const PoseXY p = moved(m, PoseXY{d.fpx[fi], d.fpy[fi]});
      const double th = d.fpth[fi] + m.dth;
      double sn, cs;
      sincos(th, &sn, &cs);
      d.fpx[fi] = p.x;
      d.fpy[fi] = p.y;
      d.fpth[fi] = th;
      d.fpc[fi] = cs;
      d.fps[fi] = sn;
    }
  }
}
MG_DEV void move_parent_poses(const Dev& d, uint32_t c, uint32_t t, uint32_t nf, const PoseMove& m) {
  move_parent_poses(d, c, t, nf, m, threadIdx.x, blockDim.x);
}
template <class Team>
MG_DEV void commit_state_edge(const Dev& d, uint32_t c, uint32_t t, const Pose& q, uint32_t Tcur, uint32_t nf, Team& tm) {
  const size_t base = size_t(c) * (d.Tcap + 1);
  const PoseMove m = pose_move_of(d, c, t, q);
  tm.sync();  // every lane holds the old pivot before anything is overwritten
  move_poses(d, base, t + 1 + tm.tid(), Tcur + 1, tm.size(), m);
  move_parent_poses(d, c, t, nf, m, tm.tid(), tm.size());
}
MG_DEV void commit_state_edge(const Dev& d, uint32_t c, uint32_t t, const Pose& q, uint32_t Tcur, uint32_t nf) {
  WholeBlock tm;
  commit_state_edge(d, c, t, q, Tcur, nf, tm);
}

// ---------------------------------------------------------------------------
// The driver: lane 0's resumable restatement of mcmc_slam::timestep / update.

MG_DEV Reading load_reading(const Dev& d, uint32_t o) {
  Reading m;
  m.mu = d.orec[o].mu;
  m.sd = d.osd[o];
  m.bearing = d.obear != nullptr;
  m.bmu = m.bearing ? d.obear[o] : 0.0;
  m.bsd = m.bearing ? d.obsd[o] : 1.0;
  m.log_sd = log(m.sd);
  m.log_bsd = m.bearing ? log(m.bsd) : 0.0;
  return m;
}

// The parent reading of landmark entry fi, from its copy beside the landmark / the copy taken anew.
MG_DEV Reading parent_reading(const Dev& d, size_t fi) {
  Reading m;
  m.mu = d.fpmu[fi];
  m.sd = d.fpsd[fi];
  m.bearing = d.obear != nullptr;
  m.bmu = m.bearing ? d.fpbmu[fi] : 0.0;
  m.bsd = m.bearing ? d.fpbsd[fi] : 1.0;
  m.log_sd = d.fplsd[fi];
  m.log_bsd = m.bearing ? d.fplbsd[fi] : 0.0;
  return m;
}
MG_DEV void keep_parent_reading(const Dev& d, size_t fi, const Reading& m) {
  d.fpmu[fi] = m.mu;
  d.fpsd[fi] = m.sd;
  d.fplsd[fi] = m.log_sd;
  if (m.bearing) {
    d.fpbmu[fi] = m.bmu;
    d.fpbsd[fi] = m.bsd;
    d.fplbsd[fi] = m.log_bsd;
  }
}

// proposal_dist::operator() of the reading's model.  Range only
// (range_bearing_model.hpp:155-159): polar(N(mu, sigma), U 2 pi), draw order by flag.  Range and
// bearing (:84-86): inv_observe of two normals drawn in index order (random.hpp:94-100).
template <class R>
MG_DEV Vec2 propose_landmark(const Dev& d, R& rng, const Reading& m);

MG_DEV void load_control(const Dev& d, uint32_t t, double mu[3], double sd[3], double& log_det) {
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    mu[i] = d.cmu[size_t(i) * d.Tcap + t];
    sd[i] = d.csd[size_t(i) * d.Tcap + t];
  }
  log_det = d.cld[t];
}

MG_DEV void write_trace(const Dev& d, uint32_t c, const DriverState& D, int kind, uint32_t idx, const double q[3], double llr,
                        int accepted) {
  if (!d.trace) return;
  const uint32_t n = d.trace_n[c];
  if (n < d.trace_cap) {
    mcmcgpu_step_record& r = d.trace[size_t(c) * d.trace_cap + n];
    r.kind = kind;
    r.accepted = accepted;
    r.index = idx;
    r.proposed[0] = q[0];
    r.proposed[1] = q[1];
    r.proposed[2] = q[2];
    r.log_ratio = llr;
    r.log_likelihood = D.ll;
  }
  d.trace_n[c] = n + 1;
}

template <class R>
MG_DEV Vec2 propose_landmark(const Dev& d, R& rng, const Reading& m) {
  if (m.bearing) {
    double v[2];
    rng.normals(v, 2);
    return polar(v[0] * m.sd + m.mu, v[1] * m.bsd + m.bmu);
  }
  double r, u;
  if (d.uniform_first) {
    u = rng.uniform();
    r = rng.normal() * m.sd + m.mu;
  } else {
    r = rng.normal() * m.sd + m.mu;
    u = rng.uniform();
  }
  return polar(r, u * 2 * kPi);
}

// mcmc_slam::add_state_edge (mcmc_slam.hpp:227-246) for edge Tcur
MG_DEV void driver_add_state_edge(const Dev& d, uint32_t c, DriverState& D) {
  const uint32_t t = D.Tcur;
  double mu[3], sd[3], ld;
  load_control(d, t, mu, sd, ld);
  // with an initialiser its estimate is taken, else the proposal's initial value (:233-237)
  const Pose est = t < d.n_init_rel ? Pose{d.init_rel[3 * size_t(t)], d.init_rel[3 * size_t(t) + 1], d.init_rel[3 * size_t(t) + 2]}
                                    : control_inv_observe(mu);
  const size_t ri = size_t(c) * d.Tcap + t;
  d.relx[ri] = est.x;
  d.rely[ri] = est.y;
  d.relth[ri] = est.th;
  append_pose(d, c, t, est);
  const double lp = control_log_pdf(mu, sd, ld, est);
  d.clp[ri] = lp;
  wt::push_back(state_tree(d, c, D), t, edge_weight_from_log_pdf(lp, 3.0, nullptr));
  D.w_dirty |= kDirtyStates;
  D.ll += lp;
  D.llc += lp;
  D.Tcur = t + 1;
}

// Weight totals of both trees (mcmc_slam.hpp:342-343, :400-401).  They only
// change when a weight changes, so they are kept until then; recomputing them
// gives bit-identical values.
MG_DEV void refresh_totals(const Dev& d, uint32_t c, DriverState& D) {
  if (!D.w_dirty) return;
  if (D.w_dirty & kDirtyStates) D.Ws = wt::total(state_tree(d, c, D), D.Tcur);
  if (D.w_dirty & kDirtyLandmarks) D.Wf = wt::total(landmark_tree(d, c, D), D.nf);
  D.w_dirty = 0;
}

// Selection and proposal of one update (mcmc_slam.hpp:339-368, :375-382).
// Returns false when nothing can be updated (both weight totals zero, :345).
// Index draw on a weight tree.  The wavefront form of the descent (wt::search_wave: six levels per
// memory latency) is bit-identical but not faster where it could be used: a lone wavefront issues a
// dependent instruction about every ten cycles, and what six levels save in LDS latency they spend
// on the walk through the probed nodes (Plaza1, 14 levels: 5 900 cycles against the serial 4 600).
constexpr bool kWaveDescent = false;
template <class R>
MG_DEV uint32_t draw_index(const wt::Cached& tree, uint32_t n, double value) {
#if defined(__HIP_DEVICE_COMPILE__)
  if (R::kWave && kWaveDescent) return wt::search_wave(tree, n, value);
#endif
  return wt::search(tree, n, value);
}

// `watch()` is asked between the parts of the work whether to give up (a selection made ahead of time
// whose premise has just turned out false); the plain call never gives up.
struct NeverGiveUp {
  MG_DEV bool operator()() { return false; }
};
enum BeginStatus : int { BEGIN_OK = 1, BEGIN_NOTHING = 0, BEGIN_FAILED = -1, BEGIN_GIVEN_UP = -2 };
// Writes the update to `out` and touches nothing of D but the cached weight totals.
template <class R, class Watch>
MG_DEV int begin_update_core(const Dev& d, uint32_t c, Shared& S, DriverState& D, R& rng, Request& out, Watch& watch) {
  const wt::Cached sw = state_tree(d, c, D), fw = landmark_tree(d, c, D);
  MG_MARK_BEGIN();
  refresh_totals(d, c, D);
  const double Ws = D.Ws, Wf = D.Wf;
  if (Ws == 0 && Wf == 0) return BEGIN_NOTHING;
  uint32_t retry = 0;
  MG_MARK(S, 8);
  const double which = rng.uniform();
  MG_MARK(S, 9);
  if ((Ws + Wf) * which < Ws) {
    uint32_t t;
    do {
      t = draw_index<R>(sw, D.Tcur, Ws * rng.uniform());
    } while (t >= D.Tcur && ++retry < kMaxRetry);
    if (t >= D.Tcur) return BEGIN_FAILED;
    MG_MARK(S, 10);
    if (watch()) return BEGIN_GIVEN_UP;
    double mu[3], sd[3], v[3], ld;
    load_control(d, t, mu, sd, ld);
    // the current relative pose's log density is kept from when that pose was set
    const double old_pll = d.clp[size_t(c) * d.Tcap + t];
    rng.normals(v, 3);
    for (int i = 0; i < 3; ++i) v[i] = v[i] * sd[i] + mu[i];
    MG_MARK(S, 11);
    if (watch()) return BEGIN_GIVEN_UP;
    const Pose q = control_inv_observe(v);
    MG_MARK(S, 12);
    out.new_pll = control_log_pdf(mu, sd, ld, q);
    out.old_pll = old_pll;
    MG_MARK(S, 13);
    D.list.valid = 0;
    out.op = OP_EVAL_STATE;
    out.idx = t;
    out.q[0] = q.x;
    out.q[1] = q.y;
    out.q[2] = q.th;
  } else {
    uint32_t f;
    do {
      f = draw_index<R>(fw, D.nf, Wf * rng.uniform());
    } while (f >= D.nf && ++retry < kMaxRetry);
    if (f >= D.nf) return BEGIN_FAILED;
    if (watch()) return BEGIN_GIVEN_UP;
    const size_t fi = size_t(c) * d.Fcap + f;
    const Reading m = parent_reading(d, fi);
    const Vec2 current{d.fex[fi], d.fey[fi]};
#if defined(__HIP_DEVICE_COMPILE__)
    // One wavefront per chain (it runs this program AND the evaluation): what the evaluation of this
    // proposal will read is asked for now, so that its three memory round trips pass while the proposal
    // is drawn -- the landmark's list and parent frame with the parent reading, ...
    const bool fetch_ahead = R::kWave && blockDim.x == 64 && d.obear == nullptr;
    uint32_t list_lo = 0, list_count = 0;
    Frame list_parent{0.0, 0.0, 1.0, 0.0};
    if (fetch_ahead) {
      list_lo = d.foff[f];
      list_count = d.fcnt[fi];
      list_parent = parent_frame(d, fi);
    }
    // the landmark's current estimate is asked for WITH its parent reading: one wait for memory, not a
    // second one after the draws (the compiler leaves a load where its value is used)
    asm volatile("" ::"v"(m.mu), "v"(m.sd), "v"(current.x), "v"(current.y), "v"(list_lo), "v"(list_count), "v"(list_parent.x),
                 "v"(list_parent.y), "v"(list_parent.c), "v"(list_parent.s));
    // ... lane L's reading of the list before the draws, ...
    const bool list_fits = fetch_ahead && list_count <= 64u;
    const uint32_t lane = threadIdx.x & 63u;
    StreamRec list_rec{0.0, 0u};
    double list_k = 0.5 * d.usig_inv * d.usig_inv;
    if (list_fits) {
      const uint32_t o = list_lo + (lane < list_count ? lane : 0u);
      list_rec = d.osr[o];
      if (!(d.usig_inv > 0.0)) {
        const double isd = d.oisd[o];
        list_k = 0.5 * isd * isd;
      }
    }
#endif
    const Vec2 q = propose_landmark(d, rng, m);
#if defined(__HIP_DEVICE_COMPILE__)
    // ... and the pose it was taken at between the draws and the proposal's densities
    D.list.valid = 0;
    // (every earlier load has come back before the pose is asked for: where the two paths meet again, the
    // compiler would otherwise wait for "whatever is outstanding" at the next use of any of them -- the pose)
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    if (list_fits) {
      asm volatile("" ::"v"(list_rec.t), "v"(list_rec.mu), "v"(list_k));
      D.list.pose = d.pxy[size_t(c) * (d.Tcap + 1) + list_rec.t];
      D.list.valid = 1;
      D.list.count = list_count;
      D.list.parent = list_parent;
      D.list.current = current;
      D.list.mu = list_rec.mu;
      D.list.k = list_k;
    }
#endif
    out.new_pll = reading_proposal_log_pdf(m, q);
    out.old_pll = reading_proposal_log_pdf(m, current);
    out.op = OP_EVAL_FEATURE;
    out.idx = f;
    out.q[0] = q.x;
    out.q[1] = q.y;
    out.q[2] = 0.0;
    MG_MARK(S, 14);
  }
  out.which = which;
  out.plain = retry == 0 ? 1 : 0;
  return BEGIN_OK;
}

MG_DEV void post_request(DriverState& D, const Request& rq) {
  D.op = rq.op;
  D.idx = rq.idx;
  D.q[0] = rq.q[0];
  D.q[1] = rq.q[1];
  D.q[2] = rq.q[2];
  D.new_pll = rq.new_pll;
  D.old_pll = rq.old_pll;
  D.pending = PEND_UPDATE;
  D.use_weight = 1;
}

template <class R>
MG_DEV bool driver_begin_update(const Dev& d, uint32_t c, Shared& S, DriverState& D, R& rng) {
  if (D.spec_ready) {  // selected and proposed while the last update was evaluated, and that one was rejected
    D.spec_ready = 0;
    post_request(D, D.spec);
    return true;
  }
  Request rq;
  NeverGiveUp never;
  const int st = begin_update_core(d, c, S, D, rng, rq, never);
  if (st == BEGIN_FAILED) D.failed = 1;
  if (st != BEGIN_OK) return false;
  post_request(D, rq);
  return true;
}

// The rest of mcmc_slam::update(EdgeType&&, bool) (:385-418) once the
// observation part of the log ratio is known.  Returns the accept decision;
// the pose table of an accepted state edge is moved afterwards by all lanes.
template <class R>
MG_DEV bool driver_finish_update(const Dev& d, uint32_t c, DriverState& D, R& rng, double obs_ratio) {
  const bool is_state = (D.op == OP_EVAL_STATE);
  const double dim = is_state ? 3.0 : 2.0;
  const double proposal_log_ratio = D.new_pll - D.old_pll;
  // state edge: the control term of :436-438 is the same pair of numbers
  const double log_ratio = is_state ? (D.new_pll - D.old_pll) + obs_ratio : obs_ratio;
  // the three exponentials of the step (:389, :397, :408) have their arguments ready together: a
  // wavefront takes one each in three lanes, one lane takes them one after the other; same values
  // edge_log_weight, :160-162 (a landmark edge's division by 2 is an exact halving)
  const double new_log_weight = log(dim) - (is_state ? D.new_pll / 3.0 : 0.5 * D.new_pll);
  const double old_log_weight = log(dim) - (is_state ? D.old_pll / 3.0 : 0.5 * D.old_pll);
  double normaliser = 1.0;
  // a tempered replica targets controls x readings^beta: the readings' part of the ratio is
  // weighed by the inverse temperature; with beta = 1 this is log_ratio bit for bit
  const double tempered = is_state ? (D.new_pll - D.old_pll) + D.beta * obs_ratio : D.beta * obs_ratio;
  double accept_log_ratio = tempered - proposal_log_ratio;
  if (D.use_weight) accept_log_ratio += new_log_weight - old_log_weight;
  double new_weight, old_weight, accept_bar;
#if defined(__HIP_DEVICE_COMPILE__)
  if (R::kWave) {
    const uint32_t lane = threadIdx.x & 63u;
    const double e = exp(lane == 1 ? old_log_weight : lane == 2 ? accept_log_ratio : new_log_weight);
    new_weight = wt::readlane_f64(e, 0);
    old_weight = wt::readlane_f64(e, 1);
    accept_bar = wt::readlane_f64(e, 2);
  } else
#endif
  {
    new_weight = exp(new_log_weight);
    old_weight = D.use_weight ? exp(old_log_weight) : 0.0;
    accept_bar = exp(accept_log_ratio);
  }
  const wt::Cached sw = state_tree(d, c, D), fw = landmark_tree(d, c, D);
  if (D.use_weight) {
    refresh_totals(d, c, D);
    const double weight_sum = D.Ws + D.Wf;
    normaliser += (new_weight - old_weight) / weight_sum;
  }
#ifdef MG_DIAG_REJECT_ALL  // diagnostic builds: the chains stay where they are whatever the streams computed
  const bool accept = (normaliser * rng.uniform() < accept_bar) && false;
#else
  const bool accept = normaliser * rng.uniform() < accept_bar;
#endif
  if (accept) {
    if (is_state) {
      D.llc += D.new_pll - D.old_pll;
      d.clp[size_t(c) * d.Tcap + D.idx] = D.new_pll;
      wt::set(sw, D.Tcur, D.idx, new_weight);
      D.w_dirty |= kDirtyStates;
      // the relative pose itself is stored after commit_state_edge has moved
      // the pose table (that needs the old angle)
    } else {
      const size_t fi = size_t(c) * d.Fcap + D.idx;
      d.fex[fi] = D.q[0];
      d.fey[fi] = D.q[1];
      wt::set(fw, D.nf, D.idx, new_weight);
      D.w_dirty |= kDirtyLandmarks;
    }
    D.ll += log_ratio;
  }
  write_trace(d, c, D, is_state ? 0 : 1, D.idx, D.q, log_ratio, accept ? 1 : 0);
  return accept;
}

// driver_finish_update in three parts (advance_kernel's SPEC variant, where the accept test is prepared
// and decided by a wavefront that does not hold the driver's state): what does not need the evaluation's
// result -- the two weights, the normaliser times the uniform of the accept test (:389-401, :408);
// the decision once the result is there; and what the decision changes.  Same operations on the same
// numbers as the one-piece form.
struct FinishPrepared {
  double new_weight;
  double log_weight_change;  // new_log_weight - old_log_weight
  double bar_factor;         // normaliser x the accept test's uniform
};
#if defined(__HIP_DEVICE_COMPILE__)
MG_DEV FinishPrepared finish_prepare(const TestInputs& in) {  // by a whole wavefront on the same data
  const double dim = in.is_state ? 3.0 : 2.0;
  const double new_log_weight = log(dim) - (in.is_state ? in.new_pll / 3.0 : 0.5 * in.new_pll);
  const double old_log_weight = log(dim) - (in.is_state ? in.old_pll / 3.0 : 0.5 * in.old_pll);
  const double e = exp((threadIdx.x & 63u) == 1 ? old_log_weight : new_log_weight);
  const double new_weight = wt::readlane_f64(e, 0), old_weight = wt::readlane_f64(e, 1);
  double normaliser = 1.0;
  if (in.use_weight) normaliser += (new_weight - old_weight) / in.weight_sum;
  return FinishPrepared{new_weight, new_log_weight - old_log_weight, normaliser * in.u};
}
#endif
MG_DEV bool finish_decide(const TestInputs& in, const FinishPrepared& pre, double obs_ratio) {
  const double proposal_log_ratio = in.new_pll - in.old_pll;
  const double tempered = in.is_state ? (in.new_pll - in.old_pll) + in.beta * obs_ratio : in.beta * obs_ratio;
  double accept_log_ratio = tempered - proposal_log_ratio;
  if (in.use_weight) accept_log_ratio += pre.log_weight_change;
#ifdef MG_DIAG_REJECT_ALL
  return (pre.bar_factor < exp(accept_log_ratio)) && false;
#else
  return pre.bar_factor < exp(accept_log_ratio);
#endif
}
MG_DEV void finish_apply(const Dev& d, uint32_t c, DriverState& D, double new_weight, double obs_ratio, bool accept) {
  const bool is_state = (D.op == OP_EVAL_STATE);
  const double log_ratio = is_state ? (D.new_pll - D.old_pll) + obs_ratio : obs_ratio;
  if (accept) {
    if (is_state) {
      D.llc += D.new_pll - D.old_pll;
      d.clp[size_t(c) * d.Tcap + D.idx] = D.new_pll;
      wt::set(state_tree(d, c, D), D.Tcur, D.idx, new_weight);
      D.w_dirty |= kDirtyStates;
    } else {
      const size_t fi = size_t(c) * d.Fcap + D.idx;
      d.fex[fi] = D.q[0];
      d.fey[fi] = D.q[1];
      wt::set(landmark_tree(d, c, D), D.nf, D.idx, new_weight);
      D.w_dirty |= kDirtyLandmarks;
    }
    D.ll += log_ratio;
  }
  write_trace(d, c, D, is_state ? 0 : 1, D.idx, D.q, log_ratio, accept ? 1 : 0);
}

// One arrival during growth (mcmc_slam.hpp:283-331).  Returns true when a
// re-parenting update was posted and needs the workgroup.
template <class R>
MG_DEV bool driver_arrival(const Dev& d, uint32_t c, DriverState& D, R& rng, uint32_t o) {
  const ObsRec rec = d.orec[o];
  const uint32_t f = rec.f;
  const size_t fi = size_t(c) * d.Fcap + f;
  const uint32_t t = D.tn;
  const Reading m = load_reading(d, o);
  const double mu = rec.mu;
  if (o == d.foff[f]) {  // first reading of this landmark: add_feature_edge, :248-270
    Vec2 est;
    if (f < d.n_init_feat) {  // an initialiser's estimate (:255-258): no draw from the chain
      est = Vec2{d.init_feat[2 * size_t(f)], d.init_feat[2 * size_t(f) + 1]};
    } else if (m.bearing) {  // proposal_dist::initial_value, range_bearing_model.hpp:93-95: no draw at all
      est = polar(m.mu, m.bmu);
    } else {
      const double u = (d.init_u && f < d.n_init_u) ? d.init_u[f] : rng.uniform();
      est = polar(mu, u * 2 * kPi);
    }
    d.fex[fi] = est.x;
    d.fey[fi] = est.y;
    d.fpar[fi] = 0;
    keep_parent_reading(d, fi, m);
    d.fcnt[fi] = 1;
    d.fpt[fi] = t;
    keep_parent_pose(d, c, fi, t);
    d.flast[fi] = t;
    wt::push_back(landmark_tree(d, c, D), f, edge_weight_from_log_pdf(reading_proposal_log_pdf(m, est), 2.0, nullptr));
    D.w_dirty |= kDirtyLandmarks;
    D.ll += reading_log_pdf(m, est);
    D.nf = f + 1;
    return false;
  }
  const uint32_t rel = o - d.foff[f];
  d.fcnt[fi] = rel + 1;
  d.flast[fi] = t;
  const Vec2 w = act(parent_frame(d, fi), Vec2{d.fex[fi], d.fey[fi]});
  const Vec2 est = act_inverse(world_pose(d, c, t), w);
  D.ll += reading_log_pdf(m, est);
  const double par_mu = d.fpmu[fi];
  if (fabs(mu) < fabs(par_mu)) {  // more_accurate_than, range_bearing_model.hpp:141-143
    d.fpar[fi] = rel;
    keep_parent_reading(d, fi, m);
    d.fpt[fi] = t;
    keep_parent_pose(d, c, fi, t);
    d.fex[fi] = est.x;
    d.fey[fi] = est.y;
    // update(feature_edge, false), :320
    const Vec2 q = propose_landmark(d, rng, m);
    D.new_pll = reading_proposal_log_pdf(m, q);
    D.old_pll = reading_proposal_log_pdf(m, est);
    D.op = OP_EVAL_FEATURE;
    D.idx = f;
    D.q[0] = q.x;
    D.q[1] = q.y;
    D.q[2] = 0.0;
    D.pending = PEND_REPARENT;
    D.use_weight = 0;
    D.list.valid = 0;
    D.rp_obs = o;
    return true;
  }
  return false;
}

// Report interval bookkeeping of a launch's end steps (multi_mcmc::completed,
// multi_mcmc.hpp:121-139, reports after every 100 updates): when a launch runs
// more updates than one interval, every chain leaves a summary of itself at
// each interval boundary, so the caller gets the same report lines from one
// long launch as from many short ones.
MG_DEV void driver_report(const Dev& d, uint32_t c, const DriverState& D) {
  if (!d.snap_out || d.snap_every == 0) return;
  if (D.end_done % d.snap_every != 0 && D.end_left != 0) return;
  const uint32_t slot = (D.end_done + d.snap_every - 1) / d.snap_every - 1;
  mcmcgpu_chain_summary& r = d.snap_out[size_t(slot) * d.C + c];
  r.log_likelihood = D.ll;
  r.updates = uint32_t(d.n_updates[c] + D.n_upd);
  r.accepts = uint32_t(d.n_accepts[c] + D.n_acc);
}

// Runs lane 0's program until it needs the workgroup (a request is posted in
// D.op) or is finished (D.op = OP_END).
template <class R>
MG_DEV void driver_advance(const Dev& d, uint32_t c, Shared& S, DriverState& D, R& rng) {
  for (;;) {
    if (D.failed) {
      D.op = OP_END;
      return;
    }
    switch (D.stage) {
      case ST_EDGE: {
        if (D.tn > D.upto) {
          D.stage = ST_END_STEPS;
          break;
        }
        MG_MARK_BEGIN();
        if (D.tn > 0) driver_add_state_edge(d, c, D);
        D.arr = d.toff[D.tn];
        D.arr_end = d.toff[D.tn + 1];
        D.stage = ST_ARRIVALS;
        MG_MARK(S, 15);
        break;
      }
      case ST_ARRIVALS: {
        MG_MARK_BEGIN();
        while (D.arr < D.arr_end) {
          const uint32_t o = d.tobs[D.arr++];
          if (driver_arrival(d, c, D, rng, o)) {
            MG_MARK(S, 15);
            return;
          }
        }
        D.steps_left = D.steps_per_ts;
        D.stage = ST_STEPS;
        MG_MARK(S, 15);
        break;
      }
      case ST_STEPS:
        while (D.steps_left > 0) {
          --D.steps_left;
          ++D.n_upd;
          if (driver_begin_update(d, c, S, D, rng)) return;
        }
        D.stage = ST_NEXT;
        break;
      case ST_NEXT:
        ++D.tn;
        D.stage = ST_EDGE;
        break;
      case ST_END_STEPS:
        for (;;) {
          if (D.end_inflight) {  // the update posted last time round has been finished by the workgroup
            D.end_inflight = 0;
            driver_report(d, c, D);
          }
          if (D.end_left == 0) break;
          --D.end_left;
          ++D.n_upd;
          ++D.end_done;
          if (driver_begin_update(d, c, S, D, rng)) {
            D.end_inflight = 1;
            return;
          }
          driver_report(d, c, D);  // nothing could be updated: the step is over already
        }
        D.stage = ST_DONE;
        break;
      default:
        D.op = OP_END;
        return;
    }
  }
}

// ---------------------------------------------------------------------------
// kernels

// Loads / stores the per-chain state that lives in LDS while a kernel runs.
// D is the driver's state: the block's copy in LDS, filled by lane 0, or (EVERY_LANE) an object of the
// kernel that every lane fills for itself.
template <bool EVERY_LANE = false>
MG_DEV void chain_state_load(const Dev& d, uint32_t c, Shared& S, DriverState& D) {
  for (uint32_t i = threadIdx.x; i < 624; i += blockDim.x) S.rng.mt[i] = d.mt[size_t(c) * 624 + i];
  if (threadIdx.x == 0) {
    S.rng.idx = d.mti[c];
    S.rng.has_saved = d.nflag[c];
    S.rng.saved = d.nsaved[c];
    for (int i = 0; i < 16; ++i) S.phase[i] = 0;
  }
  if (EVERY_LANE || threadIdx.x == 0) {
    D.nf = d.nf[c];
    D.ll = d.ll[c];
    D.beta = d.beta[c];
    D.llc = d.llc[c];
    D.n_upd = 0;
    D.n_acc = 0;
    D.n_units = 0;
    D.n_state = 0;
    D.n_moved = 0;
    // a chain that has stopped (edge selection did not terminate: an infinite weight, where the reference
    // prints "stuck in ... select loop" for ever, mcmc_slam.hpp:350-354) stays stopped: a later launch must
    // not spend its 65 536 redraws again (a replay queues one launch per timestep)
    D.failed = d.errflag[c];
    D.w_dirty = kDirtyStates | kDirtyLandmarks;
    D.sw_top = D.fw_top = 0;
    D.sw_shift = D.fw_shift = 0;
    D.pending = PEND_NONE;
    D.spec_ready = 0;
    D.list.valid = 0;
  }
}
MG_DEV void chain_state_store(const Dev& d, uint32_t c, Shared& S, const DriverState& D) {
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < 624; i += blockDim.x) d.mt[size_t(c) * 624 + i] = S.rng.mt[i];
  if (threadIdx.x == 0) {
    d.mti[c] = S.rng.idx;
    d.nflag[c] = S.rng.has_saved;
    d.nsaved[c] = S.rng.saved;
    d.nf[c] = D.nf;
    d.ll[c] = D.ll;
    d.llc[c] = D.llc;
    d.n_updates[c] += D.n_upd;
    d.n_accepts[c] += D.n_acc;
    d.n_units[c] += D.n_units;
    d.n_state_props[c] += D.n_state;
    d.n_moved[c] += D.n_moved;
    if (D.failed) d.errflag[c] = 1;
  }
}

// The persistent kernel below is built from separately compiled pieces: the
// serial driver (lane 0) needs many registers and may spill, the reading stream
// must not -- a scratch reload inside its software pipeline waits on the same
// counter as the prefetched loads and stalls it.  Each piece is therefore a
// real call with its own register allocation.  The pieces get the kernel's
// argument block and the chain's LDS block as address-space-qualified
// pointers, so that after inlining inside each piece every access is still a
// scalar load resp. an LDS operation.


// The evaluation is compiled INTO the kernel (one kernel per observation model), the serial pieces
// around it are calls: a called evaluation saves and restores the callee-saved half of its 128
// registers in every lane at every proposal -- about 100 KB each way per proposal, an eighth of
// the kernel's memory traffic -- while the serial pieces, called by one lane, save next to nothing.
// `units`: readings the evaluation went through (the same number in every lane).
template <bool RB, class Team>
MG_DEV double inline_eval(DevArgs dp, SharedLds sp, uint32_t c, int op, uint32_t idx, double qx, double qy,
                          double qth, uint32_t nf, unsigned long long& units, Team& tm) {
  const Dev& d = *(const Dev*)dp;
  Shared& S = *(Shared*)sp;
  units = 0;
  if (op == OP_EVAL_STATE) return eval_state_edge<RB>(d, c, S, idx, Pose{qx, qy, qth}, nf, units, tm);
  return eval_feature_edge<RB>(d, c, S, idx, Vec2{qx, qy}, units, tm);
}

// Lane 0's serial program between two requests to the workgroup, as ONE call: the end of the update
// whose evaluation has just come back (`r`; accept test and bookkeeping), then -- unless an accepted
// state edge first needs all lanes for the pose table (OP_COMMIT) -- on to the next request.  One
// call per update instead of two: a call costs a lone lane the save and reload of its registers.
template <class R>
MG_DEV void step_with(const Dev& d, uint32_t c, Shared& S, DriverState& D, R& rng, double r) {
  if (D.awaiting) {
    D.awaiting = 0;
#ifdef MG_PHASE_TIMING
    const long long tf0 = clock64();
#endif
    const uint32_t idx = D.idx;
    const bool accept = driver_finish_update(d, c, D, rng, r);
    D.accept = accept ? 1 : 0;
    if (D.pending == PEND_UPDATE) {
      if (accept) ++D.n_acc;
    } else if (!accept) {  // rejected re-parenting update: reset the edge weight, :321-322
      const uint32_t o = D.rp_obs;
      const size_t fi = size_t(c) * d.Fcap + idx;
      wt::set(landmark_tree(d, c, D), D.nf, idx,
              edge_weight_from_log_pdf(reading_proposal_log_pdf(load_reading(d, o), Vec2{d.fex[fi], d.fey[fi]}), 2.0,
                                       nullptr));
      D.w_dirty |= kDirtyLandmarks;
    }
    D.pending = PEND_NONE;
#ifdef MG_PHASE_TIMING
    const long long tf1 = clock64();
    S.phase[4] += tf1 - tf0;  // the accept test's share of this call; the kernel books the whole call to slot 0
    S.phase[0] -= tf1 - tf0;
#endif
    if (D.op == OP_EVAL_STATE && accept) {
      D.op = OP_COMMIT;
      return;
    }
  }
  driver_advance(d, c, S, D, rng);
  if (D.op == OP_EVAL_STATE || D.op == OP_EVAL_FEATURE) D.awaiting = 1;
}

// WAVE: the body is run by all 64 lanes of a wavefront on the same data (every lane computes and
// stores the same values), which lets the generator's window and the weight trees' descents use
// the lanes (WaveRng, wt::search_wave); otherwise by one lane.  Same results either way.
// The body run by ONE lane (the called piece below).
MG_DEV void step_body(DevArgs dp, SharedLds sp, uint32_t c, double r) {
  const Dev& d = *(const Dev*)dp;
  Shared& S = *(Shared*)sp;
#ifdef MG_PHASE_TIMING
  const long long t_in = clock64();
#endif
  SerialRng rng{S.rng};
  step_with(d, c, S, S.drv, rng, r);
#ifdef MG_PHASE_TIMING
  S.phase[7] += clock64() - t_in;  // slot 7: first to last instruction of the body (the call is the rest)
#endif
}

#if defined(__HIP_DEVICE_COMPILE__)
// The body run by all 64 lanes of a wavefront on the same data (every lane computes and stores the
// same values), which lets the generator hand out its words from a window held in the lanes
// (WaveRng).  The window object and the driver's state D live in the kernel's registers from launch
// to end: one fetch of 64 words serves several updates, and the state machine never waits for LDS.
// Same results as the one-lane body.
MG_DEV void step_body_wave(DevArgs dp, SharedLds sp, uint32_t c, double r, WaveRng& rng, DriverState& D) {
  const Dev& d = *(const Dev*)dp;
  Shared& S = *(Shared*)sp;
#ifdef MG_PHASE_TIMING
  const long long t_in = clock64();
#endif
  step_with(d, c, S, D, rng, r);
#ifdef MG_PHASE_TIMING
  S.phase[7] += clock64() - t_in;
#endif
}
#endif

MG_PIECE void piece_step(DevArgs dp, SharedLds sp, uint32_t c, double r) { step_body(dp, sp, c, r); }

// The pose table behind an accepted state edge and the edge's new relative pose (all lanes).
template <class Team>
MG_DEV void commit_body(DevArgs dp, uint32_t c, uint32_t idx, double qx, double qy, double qth, uint32_t Tcur,
                        uint32_t nf, Team& tm) {
  const Dev& d = *(const Dev*)dp;
  commit_state_edge(d, c, idx, Pose{qx, qy, qth}, Tcur, nf, tm);
  if (tm.tid() == 0) {
    const size_t ri = size_t(c) * d.Tcap + idx;
    d.relx[ri] = qx;
    d.rely[ri] = qy;
    d.relth[ri] = qth;
  }
}
MG_PIECE void piece_commit(DevArgs dp, uint32_t c, uint32_t idx, double qx, double qy, double qth, uint32_t Tcur,
                           uint32_t nf) {
  WholeBlock tm;
  commit_body(dp, c, idx, qx, qy, qth, Tcur, nf, tm);
}

#if defined(__HIP_DEVICE_COMPILE__)
// The first wavefront's view of the evaluation that runs beside it (advance_ahead): asked between the
// parts of the selection drawn ahead, it looks whether the update in flight has been decided, and says
// "give up" when it was accepted.
struct EvalWatch {
  Shared& S;
  uint32_t number;  // of the evaluation in flight
  bool decided = false, accepted = false;
  MG_DEV bool operator()() {
    if (decided) return false;
    if (__hip_atomic_load(&S.verdict_of, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != number) return false;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    decided = true;
    accepted = S.verdict_accepted != 0;
    return accepted;
  }
};

// The loop of advance_kernel's SPEC variant (see there).  Two hardware barriers per update: one after
// the request is posted, one when both the evaluation and the work ahead are over.
//   * The first wavefront posts a request together with what its accept test needs -- the proposal's two
//     log densities, the weight totals, and the test's uniform, which it draws right behind the proposal
//     (nothing else draws between the two, mcmc_slam.hpp:375-408) -- and then selects and proposes the
//     NEXT update as it will be if this one is rejected.
//   * The other wavefronts evaluate.  The last of them has nothing to do while the first of them sets the
//     segments up: it prepares the accept test meanwhile (the two edge weights, the normaliser), and when
//     every wavefront has left its sum it decides the test and leaves the verdict for the first wavefront,
//     which by then only has to apply it.
//   * The pose table behind an accepted state edge is moved by the evaluating wavefronts alone while the
//     first one already selects the next update (selection and proposal read weights, controls and
//     landmark estimates, never a pose) -- unless the driver's next act is growth, which does read poses
//     and waits for them.
template <bool RB>
MG_DEV void advance_ahead(DevArgs dp, SharedLds sp, uint32_t c, DriverState& D, WaveRng& rng) {
  const Dev& d = *(const Dev*)dp;
  Shared& S = *(Shared*)sp;
  const bool first_wave = threadIdx.x < 64;
  const uint32_t tail_waves = (blockDim.x >> 6) - 1u;
  const bool last_wave = (threadIdx.x >> 6) == tail_waves;
  TailWaves tail(&S.tail_arrivals);
  uint32_t evals = 0;  // evaluations started so far
  uint32_t round = 0;  // requests posted so far: they alternate between two slots
  // the first wavefront's: the request selected ahead, with its accept test's uniform; the generator as it
  // stood behind the uniform of the update in flight
  bool ahead_ok = false;
  double ahead_u = 0.0;
  WaveRng::Mark after_test = rng.mark();
#ifdef MG_PHASE_TIMING  // diagnostic build: where the two parties' cycles go, and how the work ahead fares
  long long dg_serial = 0, dg_ahead = 0, dg_wait0 = 0, dg_eval = 0, dg_wait1 = 0, dg_commit = 0;
  long long dg_evals = 0, dg_tried = 0, dg_taken = 0, dg_given_up = 0, dg_wasted = 0, dg_ran_out = 0;
  long long dg_finish = 0, dg_adopt = 0, dg_afresh = 0;
#define MG_AHEAD_CLOCK(v) const long long v = clock64()
#define MG_AHEAD_ADD(acc, a, b) acc += (b) - (a)
#define MG_AHEAD_COUNT(acc) ++acc
#else
#define MG_AHEAD_CLOCK(v)
#define MG_AHEAD_ADD(acc, a, b)
#define MG_AHEAD_COUNT(acc)
#endif
  for (;;) {
    MG_AHEAD_CLOCK(ta0);
    if (first_wave) {
      bool commit = false;
      if (D.awaiting) {  // the end of the update whose evaluation and verdict are back: step_with's first half
        D.awaiting = 0;
        const uint32_t idx = D.idx;
        const bool accept = S.verdict_accepted != 0;
        finish_apply(d, c, D, S.verdict_new_weight, S.verdict_sum, accept);
        if (accept && ahead_ok) {
          // Drawn ahead on a false premise -- which does not always spoil it.  An accepted update changes ONE
          // weight tree (and that tree's total).  An update drawn ahead on the OTHER tree took its index from
          // a tree and a total that are what they were, and its proposal from draws that are the same either
          // way (the generator stood where it stands); all that the accepted update can change for it is the
          // choice between the trees, :346, which is looked at again with the new totals.
          bool keep = false;
          if (D.spec.plain && D.spec.op != D.op) {
            refresh_totals(d, c, D);
            keep = ((D.Ws + D.Wf) * D.spec.which < D.Ws) == (D.spec.op == OP_EVAL_STATE);
          }
          if (!keep) {
            rng.restore(after_test);
            ahead_ok = false;
            MG_AHEAD_COUNT(dg_wasted);
          }
        }
        D.accept = accept ? 1 : 0;
        if (D.pending == PEND_UPDATE) {
          if (accept) ++D.n_acc;
        } else if (!accept) {  // rejected re-parenting update: reset the edge weight, :321-322
          const uint32_t o = D.rp_obs;
          const size_t fi = size_t(c) * d.Fcap + idx;
          wt::set(landmark_tree(d, c, D), D.nf, idx,
                  edge_weight_from_log_pdf(reading_proposal_log_pdf(load_reading(d, o), Vec2{d.fex[fi], d.fey[fi]}), 2.0,
                                           nullptr));
          D.w_dirty |= kDirtyLandmarks;
        }
        D.pending = PEND_NONE;
        if (D.op == OP_EVAL_STATE && accept) {
          D.op = OP_COMMIT;
          commit = true;
        }
      }
      MG_AHEAD_CLOCK(tf1);
      MG_AHEAD_ADD(dg_finish, ta0, tf1);
      TestInputs test{0, 0, 0.0, 0.0, 1.0, 1.0, 0.0};
      if (!commit) {
        const bool adopting = ahead_ok;
        D.spec_ready = ahead_ok ? 1 : 0;
        if (ahead_ok) MG_AHEAD_COUNT(dg_taken);
        ahead_ok = false;
        driver_advance(d, c, S, D, rng);
        if (D.spec_ready) {  // (cannot happen: the driver's next act was checked to be a plain update)
          D.spec_ready = 0;
          D.failed = 1;
        }
        if (D.op == OP_EVAL_STATE || D.op == OP_EVAL_FEATURE) {
          D.awaiting = 1;
          // the accept test's uniform: drawn behind the proposal -- with it when the request was selected ahead
          test.u = adopting ? ahead_u : rng.uniform();
          after_test = rng.mark();
          if (D.use_weight) refresh_totals(d, c, D);
          test.is_state = D.op == OP_EVAL_STATE ? 1 : 0;
          test.use_weight = D.use_weight;
          test.new_pll = D.new_pll;
          test.old_pll = D.old_pll;
          test.beta = D.beta;
          test.weight_sum = D.Ws + D.Wf;
        }
#ifdef MG_PHASE_TIMING
        const long long tf2 = clock64();
        if (adopting)
          dg_adopt += tf2 - tf1;
        else
          dg_afresh += tf2 - tf1;
#endif
      }
      if (threadIdx.x == 0) {  // the request, where the other wavefronts find it
        Posted& post = S.post[round & 1u];
        post.op = D.op;
        post.idx = D.idx;
        post.q[0] = D.q[0];
        post.q[1] = D.q[1];
        post.q[2] = D.q[2];
        post.Tcur = D.Tcur;
        post.nf = D.nf;
        // a commit: does the driver go on to something that reads the pose table?
        post.join = !(D.stage == ST_STEPS && D.steps_left > 0) && !(D.stage == ST_END_STEPS && D.end_left > 0);
        post.test = test;
      }
    }
    __syncthreads();
    MG_AHEAD_CLOCK(ta1);
    MG_AHEAD_ADD(dg_serial, ta0, ta1);
    const Posted& post = S.post[round & 1u];
    ++round;
    const int op = post.op;
    if (op == OP_END) break;
    const uint32_t idx = post.idx;
    const double qx = post.q[0], qy = post.q[1], qth = post.q[2];
    const uint32_t Tcur = post.Tcur, nf = post.nf;
    if (op == OP_COMMIT) {  // the state edge just accepted: the other wavefronts move the pose table
      const uint32_t join = post.join;
      if (!first_wave) commit_body(dp, c, idx, qx, qy, qth, Tcur, nf, tail);
      D.n_moved += Tcur - idx;
      if (join) __syncthreads();
      MG_AHEAD_CLOCK(tc1);
      MG_AHEAD_ADD(dg_commit, ta1, tc1);
      continue;
    }
    ++evals;
    MG_AHEAD_COUNT(dg_evals);
    if (first_wave) {
      ahead_ok = false;
      // the driver's next act is another plain update: the one in flight is one, and more are due
      const bool more = (D.stage == ST_STEPS && D.steps_left > 0) || (D.stage == ST_END_STEPS && D.end_left > 0);
      if (D.pending == PEND_UPDATE && more && !D.failed) {
        EvalWatch watch{S, evals};
        rng.ahead = true;
        rng.ran_out = false;
        MG_AHEAD_COUNT(dg_tried);
        const int st = begin_update_core(d, c, S, D, rng, D.spec, watch);
        if (st == BEGIN_OK) ahead_u = rng.uniform();  // its accept test's uniform, behind its proposal
        ahead_ok = st == BEGIN_OK && !rng.ran_out;
        if (st == BEGIN_GIVEN_UP) MG_AHEAD_COUNT(dg_given_up);
        if (rng.ran_out) MG_AHEAD_COUNT(dg_ran_out);
        rng.ahead = false;
        rng.ran_out = false;
        if (!ahead_ok) rng.restore(after_test);
      }
      MG_AHEAD_CLOCK(tb1);
      MG_AHEAD_ADD(dg_ahead, ta1, tb1);
    } else {
      // the accept test as far as it can be prepared: by the last evaluating wavefront, which would
      // otherwise wait for the first of them to set the segments up
      FinishPrepared pre{0.0, 0.0, 0.0};
      if (last_wave) pre = finish_prepare(post.test);
      unsigned long long units;
      inline_eval<RB>(dp, sp, c, op, idx, qx, qy, qth, nf, units, tail);  // leaves the sums in S.red
      if (tail.tid() == 0) S.eval_units = units;
      if (last_wave) {  // ... and its decision, once every wavefront's sum is there
        const uint32_t due = evals * tail_waves;
        while (__hip_atomic_load(&S.eval_arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != due) {
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const double sum = tail_waves_sum(S);
        const bool accept = finish_decide(post.test, pre, sum);
        if ((threadIdx.x & 63u) == 0) {
          S.verdict_accepted = accept ? 1 : 0;
          S.verdict_new_weight = pre.new_weight;
          S.verdict_sum = sum;
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          __hip_atomic_store(&S.verdict_of, evals, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
      MG_AHEAD_CLOCK(tb1);
      MG_AHEAD_ADD(dg_eval, ta1, tb1);
    }
    MG_AHEAD_CLOCK(tw0);
    __syncthreads();
    MG_AHEAD_CLOCK(tw1);
    if (first_wave)
      MG_AHEAD_ADD(dg_wait0, tw0, tw1);
    else
      MG_AHEAD_ADD(dg_wait1, tw0, tw1);
    D.n_units += S.eval_units;
    if (op == OP_EVAL_STATE) ++D.n_state;
  }
#ifdef MG_PHASE_TIMING
  // slots of this variant: 0 serial step, 1 work ahead, 2 first wavefront waiting for the evaluation,
  // 3 evaluation, 4 the others waiting for the first, 5 commit; 8.. counts: evaluations, selections
  // tried ahead, taken, given up early, completed but dropped, out of generator words
  __syncthreads();
  if (threadIdx.x == 0) {
    S.phase[0] = dg_serial, S.phase[1] = dg_ahead, S.phase[2] = dg_wait0, S.phase[5] = dg_commit;
    S.phase[8] = dg_evals, S.phase[9] = dg_tried, S.phase[10] = dg_taken, S.phase[11] = dg_given_up;
    S.phase[12] = dg_wasted, S.phase[13] = dg_ran_out, S.phase[14] = dg_adopt, S.phase[15] = dg_afresh, S.phase[7] = dg_finish;
  }
  if (threadIdx.x == 64) S.phase[3] = dg_eval, S.phase[4] = dg_wait1;
  __syncthreads();
#endif
}
#endif

// grid = chains, block = threads_per_chain.  first_tn = the next timestep the
// chains have not processed yet; Tcur0 = state edges grown so far.  `d` must
// stay the first parameter: the pieces read it through the argument block.
// What the launch sets aside in dynamic LDS for copies of the weight trees' tops: sw_n / fw_n nodes
// (0 = no copy) at strides 2^sw_shift / 2^fw_shift; the state tree's copy comes first.
struct TreeTops {
  uint32_t sw_n, sw_shift, fw_n, fw_shift;
};
// SOLO: the serial program compiled into the kernel as well, for launches of so few chains that a chain
// has its CU to itself and the serial program IS the launch's critical path (one mcmc_slam instance on
// a Plaza-sized log): no call, no registers saved and reloaded around it.
//
// SPEC (with SOLO, workgroups of at least two wavefronts): while the wavefronts after the first evaluate
// update k, the first one runs ahead -- it draws the uniform of k's accept test (nothing else draws
// between a proposal and that uniform, mcmc_slam.hpp:375-408), prepares the parts of the test that do not
// need the evaluation, and then selects and proposes update k + 1 as it will be IF k is rejected (a
// rejected update changes nothing but the generator).  It looks now and then whether k's evaluation has
// come back; if k turns out accepted it drops what it has drawn ahead, puts the generator back to just
// after the accept test's uniform and selects afresh once k is applied.  Every draw, weight and sum is the
// one the sequential order produces; only the time at which it is computed differs.  For data on which
// an update's serial part outlasts its evaluation (a Plaza-sized log: 650 readings per update).
template <bool RB, bool SOLO = false, bool SPEC = false>
__global__ void __launch_bounds__(SOLO ? 256 : 1024) advance_kernel(Dev d, uint32_t first_tn, uint32_t Tcur0, uint32_t upto, uint32_t steps_per_ts,
                               uint32_t end_steps, uint32_t grow, TreeTops tops) {
  static_assert(SOLO || !SPEC, "running ahead needs the serial program compiled in");
  __shared__ Shared S;
  extern __shared__ double tree_tops[];
  const DevArgs dp = (DevArgs)__builtin_amdgcn_kernarg_segment_ptr();
  const SharedLds sp = (SharedLds)&S;
  const uint32_t c = blockIdx.x;
  // the driver's state: in LDS for the called program; SOLO: in registers, every lane its own copy (the
  // first wavefront's lanes run the program together and keep theirs equal; the other lanes never use it)
  DriverState own;
  DriverState& D = SOLO ? own : S.drv;
  chain_state_load<SOLO>(d, c, S, D);
  {  // nodes beyond a tree's current size hold nothing yet; they are written (through) before they are read
    const double* sw = d.sw + size_t(c) * d.Tcap;
    const double* fw = d.fw + size_t(c) * d.Fcap;
    for (uint32_t j = threadIdx.x; j < tops.sw_n; j += blockDim.x) tree_tops[j] = sw[size_t(j) << tops.sw_shift];
    for (uint32_t j = threadIdx.x; j < tops.fw_n; j += blockDim.x) tree_tops[tops.sw_n + j] = fw[size_t(j) << tops.fw_shift];
  }
  if (SOLO || threadIdx.x == 0) {
    D.sw_top = tops.sw_n ? uint32_t(uintptr_t((wt::LdsDoubles)&tree_tops[0])) : 0;
    D.fw_top = tops.fw_n ? uint32_t(uintptr_t((wt::LdsDoubles)&tree_tops[tops.sw_n])) : 0;
    D.sw_shift = tops.sw_shift;
    D.fw_shift = tops.fw_shift;
    D.Tcur = Tcur0;
    D.tn = first_tn;
    D.upto = upto;
    D.steps_per_ts = steps_per_ts;
    D.end_left = end_steps;
    D.end_done = 0;
    D.end_inflight = 0;
    D.awaiting = 0;
    D.stage = grow ? ST_EDGE : ST_END_STEPS;
    D.op = OP_END;
    D.idx = 0;
    D.q[0] = D.q[1] = D.q[2] = 0.0;
    D.accept = 0;
    D.use_weight = 0;
    D.new_pll = D.old_pll = 0.0;
    D.rp_obs = 0;
    D.Ws = D.Wf = 0.0;
    D.arr = D.arr_end = D.steps_left = 0;
  }
  if (threadIdx.x == 0) {
    S.tail_arrivals = 0;
    S.eval_arrived = 0;
    S.verdict_of = 0;
  }
  __syncthreads();
  MG_STAMP(tk0);
  double r = 0.0;
  const bool lone = SOLO && blockDim.x == 64;  // the chain's only wavefront runs the program AND evaluates
#if defined(__HIP_DEVICE_COMPILE__)
  WaveRng wave_rng(S.rng);  // used by the first wavefront of the SOLO variant only
  if (SPEC) {
    advance_ahead<RB>(dp, sp, c, D, wave_rng);
  } else
#endif
  for (;;) {
    MG_STAMP(td0);
    if (SOLO) {
#if defined(__HIP_DEVICE_COMPILE__)
      if (threadIdx.x < 64) {  // the first wavefront, all lanes on the same data
        step_body_wave(dp, sp, c, r, wave_rng, D);
        if (threadIdx.x == 0 && !lone) {  // the request, where the other wavefronts find it
          S.drv.op = D.op;
          S.drv.idx = D.idx;
          S.drv.q[0] = D.q[0];
          S.drv.q[1] = D.q[1];
          S.drv.q[2] = D.q[2];
          S.drv.Tcur = D.Tcur;
          S.drv.nf = D.nf;
        }
      }
#endif
    } else {
      if (threadIdx.x == 0) piece_step(dp, sp, c, r);
    }
    if (!lone) __syncthreads();
    MG_STAMP(td1);
    MG_PHASE_ADD(S, 0, td0, td1);
    // (a lone wavefront has the request in its registers: nothing to hand over, nobody to wait for)
    const int op = lone ? D.op : S.drv.op;
    if (op == OP_END) break;
    const uint32_t idx = lone ? D.idx : S.drv.idx;
    const double qx = lone ? D.q[0] : S.drv.q[0], qy = lone ? D.q[1] : S.drv.q[1], qth = lone ? D.q[2] : S.drv.q[2];
    const uint32_t Tcur = lone ? D.Tcur : S.drv.Tcur, nf = lone ? D.nf : S.drv.nf;
    if (op == OP_COMMIT) {  // the state edge just accepted: all lanes move the pose table, then lane 0 goes on
      if (SOLO) {
        WholeBlock everyone;
        commit_body(dp, c, idx, qx, qy, qth, Tcur, nf, everyone);
      } else
        piece_commit(dp, c, idx, qx, qy, qth, Tcur, nf);
      if (SOLO || threadIdx.x == 0) D.n_moved += Tcur - idx;  // poses idx+1 .. Tcur rewritten
      __syncthreads();
      MG_STAMP(tg1);
      MG_PHASE_ADD(S, 5, td1, tg1);
      continue;
    }
    unsigned long long units;
#if defined(__HIP_DEVICE_COMPILE__)
    if (SOLO && !RB && lone && op == OP_EVAL_FEATURE && D.list.valid) {
      // The readings are in the lanes already (ListAhead), all but the poses they were taken at, which are
      // still on their way: the accept test is prepared meanwhile -- its uniform (nothing else draws between
      // a proposal and that uniform, mcmc_slam.hpp:375-408), the two edge weights, the normaliser -- then the
      // stream's arithmetic per reading and one reduction, the decision and what it changes: step_with's
      // first half, here because the test's first part must come BEFORE the evaluation to overlap anything.
      TestInputs test;
      test.u = wave_rng.uniform();
      if (D.use_weight) refresh_totals(d, c, D);
      test.is_state = 0;
      test.use_weight = D.use_weight;
      test.new_pll = D.new_pll;
      test.old_pll = D.old_pll;
      test.beta = D.beta;
      test.weight_sum = D.Ws + D.Wf;
      const FinishPrepared pre = finish_prepare(test);
      MG_STAMP(tp1);
      MG_PHASE_ADD(S, 4, td1, tp1);
      const ListAhead& L = D.list;
      const Vec2 wo = act(L.parent, L.current), wn = act(L.parent, Vec2{qx, qy});
      const double ro = range_of(wo.x - L.pose.x, wo.y - L.pose.y), rn = range_of(wn.x - L.pose.x, wn.y - L.pose.y);
      const double term = (ro - rn) * fma(-2.0, L.mu, ro + rn) * L.k;
      r = wave_prefix_sums((threadIdx.x & 63u) < L.count ? term : 0.0);  // lane 63: the sum
      r = wt::readlane_f64(r, 63);
      units = L.count;
      D.n_units += units;
      MG_STAMP(te1);
      MG_PHASE_ADD(S, 3, tp1, te1);
      const bool accept = finish_decide(test, pre, r);
      finish_apply(d, c, D, pre.new_weight, r, accept);
      D.awaiting = 0;  // (the list is fetched for plain updates only: no re-parenting to undo)
      D.accept = accept ? 1 : 0;
      if (accept) ++D.n_acc;
      D.pending = PEND_NONE;
      MG_STAMP(tf1);
      MG_PHASE_ADD(S, 4, te1, tf1);
      continue;
    }
#endif
    WholeBlock everyone;
    r = inline_eval<RB>(dp, sp, c, op, idx, qx, qy, qth, nf, units, everyone);
    if (SOLO || threadIdx.x == 0) {
      D.n_units += units;
      if (op == OP_EVAL_STATE) ++D.n_state;
    }
#if defined(__HIP_DEVICE_COMPILE__)
    // the sum is valid in lane 0; the first wavefront's lanes all run the serial program on it
    if (SOLO) r = wt::readlane_f64(r, 0);
#endif
    MG_STAMP(te1);
    if (op != OP_EVAL_STATE) MG_PHASE_ADD(S, 3, td1, te1);
  }
#if defined(__HIP_DEVICE_COMPILE__)
  if (SOLO && threadIdx.x < 64) wave_rng.flush();  // chain_state_store begins with a barrier
#endif
  MG_STAMP(tk1);
  MG_PHASE_ADD(S, 6, tk0, tk1);
#ifdef MG_PHASE_TIMING
  if (threadIdx.x == 0) {
    for (int i = 0; i < 16; ++i) atomicAdd(&g_phase_cycles[i], (unsigned long long)S.phase[i]);
  }
#endif
  chain_state_store(d, c, S, D);
}

// ---------------------------------------------------------------------------
// Split mode: ONE update of every chain as four kernels, so that a chain's reading stream can use
// many workgroups.  One workgroup per chain cannot fill the GPU when there are few chains (a
// single mcmc_slam instance, a ladder of tempered replicas on a long trajectory); here the stream
// of one proposal is cut into `slices` equal parts.  Same arithmetic per reading as the persistent
// kernel; the partial sums of the slices are added in a fixed order.  Updates after growth.

MG_DEV void ctl_to_shared(const Dev& d, uint32_t c, DriverState& D, uint32_t Tcur) {
  const ChainCtl& k = d.ctl[c];
  D.Tcur = Tcur;
  D.Ws = k.Ws;
  D.Wf = k.Wf;
  D.w_dirty = k.w_dirty;
  D.new_pll = k.new_pll;
  D.old_pll = k.old_pll;
  D.q[0] = k.q[0];
  D.q[1] = k.q[1];
  D.q[2] = k.q[2];
  D.op = k.op;
  D.idx = k.idx;
  D.pending = k.pending;
  D.use_weight = k.use_weight;
}
MG_DEV void shared_to_ctl(const Dev& d, uint32_t c, const DriverState& D) {
  ChainCtl& k = d.ctl[c];
  k.Ws = D.Ws;
  k.Wf = D.Wf;
  k.w_dirty = D.w_dirty;
  k.new_pll = D.new_pll;
  k.old_pll = D.old_pll;
  k.q[0] = D.q[0];
  k.q[1] = D.q[1];
  k.q[2] = D.q[2];
  k.op = D.op;
  k.idx = D.idx;
  k.pending = D.pending;
  k.use_weight = D.use_weight;
}

// The serial ends of an update, one workgroup per chain: the END of update k -- sum of the slices,
// accept test (mcmc_slam.hpp:385-418), pose table of an accepted state edge -- and the BEGINNING of
// update k + 1 -- selection and proposal (:339-382) -- in one kernel, so that the chain's generator
// and scalars are fetched once for both.  finish = 0 on the first call of a run (nothing to end;
// the weight totals are computed afresh), begin = 0 on the last.
__global__ void split_ends_kernel(Dev d_by_value, uint32_t Tcur, uint32_t slices, uint32_t finish, uint32_t begin,
                                  uint32_t defer_commit) {
  MG_KERNEL_DEV(d);
  __shared__ Shared S;
  DriverState& D = S.drv;
  const uint32_t c = blockIdx.x;
  chain_state_load(d, c, S, D);
  // the slices' partial sums, added by all lanes in a fixed order (lane-strided, then the
  // wavefront tree, then wavefront after wavefront): up to max_slices loads that one lane would
  // wait for one after the other
  double part = 0.0;
  if (finish && d.ctl[c].op != OP_END) {
    const uint32_t used = active_slices(d.ctl[c].total, slices);
    for (uint32_t k = threadIdx.x; k < used; k += blockDim.x) part += d.sp_partial[size_t(c) * d.max_slices + k];
  }
  const double slices_sum = block_sum(part, S);
  if (threadIdx.x == 0) {
    D.Tcur = Tcur;
    D.accept = 0;
    D.op = OP_END;
    if (finish) {
      ctl_to_shared(d, c, D, Tcur);
      D.n_upd = 1;
      if (D.op != OP_END) {
        const double r = slices_sum;
        D.n_units = d.ctl[c].total;
        if (D.op == OP_EVAL_STATE) D.n_state = 1;
        SerialRng rng{S.rng};
        const bool accept = driver_finish_update(d, c, D, rng, r);
        D.accept = accept ? 1 : 0;
        if (accept) D.n_acc = 1;
        D.pending = PEND_NONE;
      }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) d.ctl[c].move_from = 0;
  if (D.op == OP_EVAL_STATE && D.accept) {
    const Pose q{D.q[0], D.q[1], D.q[2]};
    if (defer_commit) {  // split_commit_kernel moves the poses, with many workgroups
      const PoseMove m = pose_move_of(d, c, D.idx, q);
      move_parent_poses(d, c, D.idx, D.nf, m);  // few enough for this workgroup
      if (threadIdx.x == 0) {
        ChainCtl& k = d.ctl[c];
        k.move_from = D.idx + 1;
        k.move_pivot[0] = m.pivot_x;
        k.move_pivot[1] = m.pivot_y;
        k.move_to[0] = m.to_x;
        k.move_to[1] = m.to_y;
        k.move_sin = m.s;
        k.move_cos = m.c;
        k.move_dth = m.dth;
      }
    } else {
      commit_state_edge(d, c, D.idx, q, Tcur, D.nf);
    }
    __syncthreads();  // the old relative angle has been read
    if (threadIdx.x == 0) {
      const size_t ri = size_t(c) * d.Tcap + D.idx;
      d.relx[ri] = q.x;
      d.rely[ri] = q.y;
      d.relth[ri] = q.th;
      D.n_moved = Tcur - D.idx;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    D.op = OP_END;
    SerialRng rng{S.rng};
    if (begin && !D.failed && !d.errflag[c] && !driver_begin_update(d, c, S, D, rng)) D.op = OP_END;
    shared_to_ctl(d, c, D);
  }
  chain_state_store(d, c, S, D);
}

// The pose table of the chains whose state-edge proposal split_ends_kernel has just accepted
// (ChainCtl::move_from != 0); grid = (pieces, chains).  A million poses are one workgroup's work for
// milliseconds, which the other chains of the launch would wait for.
__global__ void split_commit_kernel(Dev d_by_value, uint32_t Tcur) {
  MG_KERNEL_DEV(d);
  const uint32_t c = blockIdx.y;
  const ChainCtl& k = d.ctl[c];
  const uint32_t from = k.move_from;
  if (from == 0 || from > Tcur) return;
  const PoseMove m{k.move_pivot[0], k.move_pivot[1], k.move_to[0], k.move_to[1], k.move_sin, k.move_cos, k.move_dth};
  const size_t base = size_t(c) * (d.Tcap + 1);
  const uint32_t n = Tcur - from + 1, per = (n + gridDim.x - 1) / gridDim.x;
  const uint32_t lo = from + blockIdx.x * per, hi = min(Tcur + 1, lo + per);
  move_poses(d, base, lo + threadIdx.x, hi, blockDim.x, m);
}

// Landmark f of chain c under the proposal in `k`: first record and number of its affected readings,
// its position and where the proposal moves it (phases A and B of eval_state_edge, or the single
// segment of eval_feature_edge).  Returns the count and stores the rest in the split-mode tables.
MG_DEV uint32_t split_setup_landmark(const Dev& d, uint32_t c, uint32_t f, const ChainCtl& k) {
  const size_t fi = size_t(c) * d.Fcap + f;
  uint32_t slo = d.foff[f], len = 0;
  Vec2 w{0.0, 0.0}, wn{0.0, 0.0};
  if (k.op == OP_EVAL_STATE) {
    const uint32_t t = k.idx;
    if (d.ffirst[f] <= t && t < d.flast[fi]) {
      const Pose q{k.q[0], k.q[1], k.q[2]};
      const Pose Pt = world_pose(d, c, t), Pt1 = world_pose(d, c, t + 1);
      const uint32_t lo = d.foff[f], hi = lo + d.fcnt[fi], pt = d.fpt[fi];
      const uint32_t a = split_readings(d, f, lo, hi, t);
      w = act(parent_frame(d, fi), Vec2{d.fex[fi], d.fey[fi]});
      if (t < pt) {
        wn = act(Pt, act(q, act_inverse(Pt1, w)));
        slo = lo;
        len = a - lo;
      } else {
        wn = act(Pt1, act(inverse(q), act_inverse(Pt, w)));
        slo = a;
        len = hi - a;
      }
    }
  } else if (k.op == OP_EVAL_FEATURE && f == k.idx) {
    const Frame Pp = parent_frame(d, fi);
    w = act(Pp, Vec2{d.fex[fi], d.fey[fi]});
    wn = act(Pp, Vec2{k.q[0], k.q[1]});
    len = d.fcnt[fi];
  }
  d.sp_seglo[fi] = slo;
  double* o = d.sp_segw + 4 * fi;
  o[0] = w.x;
  o[1] = w.y;
  o[2] = wn.x;
  o[3] = wn.y;
  return len;
}

// One lane per landmark; grid = (ceil(nf / 256), chains).  Leaves the counts for split_scan_kernel.
__global__ void split_setup_kernel(Dev d_by_value) {
  MG_KERNEL_DEV(d);
  const uint32_t c = blockIdx.y, f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= d.nf[c]) return;
  d.sp_segpre[size_t(c) * (d.Fcap + 1) + f] = split_setup_landmark(d, c, f, d.ctl[c]);
}

// The same and the prefix sums in one kernel, for chains of at most blockDim.x (1024) landmarks:
// one launch and one pass through memory fewer per update.  grid = chains.
__global__ void __launch_bounds__(1024) split_setup_scan_kernel(Dev d_by_value, uint32_t slices) {
  MG_KERNEL_DEV(d);
  __shared__ Shared S;
  const uint32_t c = blockIdx.x, f = threadIdx.x, nf = d.nf[c];
  const uint32_t len = f < nf ? split_setup_landmark(d, c, f, d.ctl[c]) : 0;
  uint32_t total;
  const uint32_t incl = block_scan_u32(len, S, total);
  uint32_t* pre = d.sp_segpre + size_t(c) * (d.Fcap + 1);
  if (f < nf) {
    pre[f] = incl - len;
    S.act[f] = incl - len;  // kActCap = 1024 entries
  }
  if (f == 0) {
    pre[nf] = total;
    d.ctl[c].total = total;
  }
  __syncthreads();
  write_slice_segments(d.sp_slice_seg + size_t(c) * (d.max_slices + 1), nf, total, slices,
                       [](uint32_t m) { return S.act[m]; });
}

// Counts -> exclusive prefix sums, in place; grid = chains.
__global__ void split_scan_kernel(Dev d_by_value, uint32_t slices) {
  MG_KERNEL_DEV(d);
  __shared__ Shared S;
  const uint32_t c = blockIdx.x, nf = d.nf[c];
  uint32_t* pre = d.sp_segpre + size_t(c) * (d.Fcap + 1);
  uint32_t carry = 0;
  for (uint32_t f0 = 0; f0 < nf; f0 += blockDim.x) {
    const uint32_t f = f0 + threadIdx.x;
    const uint32_t v = f < nf ? pre[f] : 0;
    uint32_t total;
    const uint32_t incl = block_scan_u32(v, S, total);
    if (f < nf) pre[f] = carry + incl - v;
    carry += total;
  }
  if (threadIdx.x == 0) {
    pre[nf] = carry;
    d.ctl[c].total = carry;
  }
  __syncthreads();  // the prefix sums written above are read back by other lanes below
  const uint32_t* cpre = pre;
  write_slice_segments(d.sp_slice_seg + size_t(c) * (d.max_slices + 1), nf, carry, slices,
                       [cpre](uint32_t m) { return cpre[m]; });
}

// The reading stream of one proposal, slice blockIdx.x of chain blockIdx.y (see active_slices).
// Two readings per lane per group; the records of the next group are requested before the
// current group is evaluated, so that their latency hides behind its arithmetic.  BEAR: range-and-
// bearing readings (the arithmetic of BearingStream).
template <bool BEAR>
__global__ void __launch_bounds__(256) split_eval_kernel(Dev d_by_value) {
  MG_KERNEL_DEV(d);
  __shared__ double red[4];
  __shared__ uint32_t wpre[kSplitWindow + 1], wlo[kSplitWindow];
  const uint32_t c = blockIdx.y, nf = d.nf[c];
  const uint32_t* __restrict__ segpre = d.sp_segpre + size_t(c) * (d.Fcap + 1);
  const uint32_t* __restrict__ seglo = d.sp_seglo + size_t(c) * d.Fcap;
  const double* __restrict__ w4 = d.sp_segw + 4 * size_t(c) * d.Fcap;
  const PoseXY* __restrict__ poses = d.pxy + size_t(c) * (d.Tcap + 1);
  const double* __restrict__ pth = d.pth + size_t(c) * (d.Tcap + 1);
  const StreamRec* __restrict__ osr = d.osr;
  const double* __restrict__ obear = d.obear;
  const uint32_t total = nf ? segpre[nf] : 0;
  const uint32_t slices = active_slices(total, gridDim.x);
  if (blockIdx.x >= slices) return;
  const uint32_t per = (total + slices - 1) / slices;
  const uint32_t lo = min(total, blockIdx.x * per), hi = min(total, lo + per);
  // this slice's segments [seg0, seg0 + nwin): their prefix sums and first records go to LDS when
  // they fit, so that the search for a lane's first segment and the cursor after it stay out of
  // global memory
  const uint32_t* sseg = d.sp_slice_seg + size_t(c) * (d.max_slices + 1);
  uint32_t seg0 = sseg[blockIdx.x], nwin = sseg[blockIdx.x + 1] - seg0 + 1;
  const bool staged = nwin <= kSplitWindow;
  if (staged) {
    for (uint32_t j = threadIdx.x; j <= nwin; j += 256) wpre[j] = segpre[seg0 + j];
    for (uint32_t j = threadIdx.x; j < nwin; j += 256) wlo[j] = seglo[seg0 + j];
    __syncthreads();
  } else {
    seg0 = 0;
    nwin = nf;
  }
  const uint32_t fallback_record = seglo[0];
  // one sigma for all ranges (and, with bearings, one for all bearings): nothing to load per reading
  const bool usig = d.usig_inv > 0.0 && (!BEAR || d.ubsig_inv > 0.0);
  const double half_isd2 = 0.5 * d.usig_inv * d.usig_inv, half_ibsd2 = 0.5 * d.ubsig_inv * d.ubsig_inv;
  const double* __restrict__ oisd = d.oisd;
  const double* __restrict__ obisd = d.obisd;
  // pre_at(j), lo_at(j): prefix sum and first record of segment seg0 + j
  auto stream = [&](auto pre_at, auto lo_at) {
    double acc = 0.0;
    if (lo + threadIdx.x >= hi) return acc;
    uint32_t cur;
    {  // the segment that holds this lane's first reading: last j with pre_at(j) <= i
      uint32_t a = 0, b = nwin;
      const uint32_t i = lo + threadIdx.x;
      while (a + 1 < b) {
        const uint32_t m = (a + b) >> 1;
        if (pre_at(m) <= i)
          a = m;
        else
          b = m;
      }
      cur = a;
    }
    uint32_t seg_end = pre_at(cur + 1), seg_base = lo_at(cur) - pre_at(cur);
    constexpr int U = 2;
    constexpr uint32_t stride = 256;
    StreamRec rec[U];
    double bm[U];
    uint32_t o[U], sg[U];
    auto fetch = [&](uint32_t i0) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t i = i0 + u * stride;
        if (i < hi) {
          while (i >= seg_end) {
            ++cur;
            seg_end = pre_at(cur + 1);
            seg_base = lo_at(cur) - pre_at(cur);
          }
          o[u] = seg_base + i;
        } else {
          o[u] = fallback_record;
        }
        sg[u] = seg0 + cur;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        rec[u] = osr[o[u]];
        if (BEAR) bm[u] = obear[o[u]];
      }
    };
    fetch(lo + threadIdx.x);
    for (uint32_t i0 = lo + threadIdx.x; i0 < hi; i0 += U * stride) {
      PoseXY p[U];
      double mu[U], k[U], th[U], bmu[U], kb[U];
      uint32_t s_[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        p[u] = poses[rec[u].t];
        mu[u] = rec[u].mu;
        s_[u] = sg[u];
        k[u] = half_isd2;
        if (BEAR) {
          th[u] = pth[rec[u].t];
          bmu[u] = bm[u];
          kb[u] = half_ibsd2;
        }
        if (!usig) {
          const double isd = oisd[o[u]];
          k[u] = 0.5 * isd * isd;
          if (BEAR) {
            const double ibsd = obisd[o[u]];
            kb[u] = 0.5 * ibsd * ibsd;
          }
        }
      }
      if (i0 + U * stride < hi) fetch(i0 + U * stride);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const double* w = w4 + 4 * size_t(s_[u]);
        const double ox = w[0] - p[u].x, oy = w[1] - p[u].y, nx = w[2] - p[u].x, ny = w[3] - p[u].y;
        const double ro = range_of(ox, oy), rn = range_of(nx, ny);
        double term = (ro - rn) * fma(-2.0, mu[u], ro + rn) * k[u];
        if (BEAR) {
          const double eo = wrap_residual(atan2(oy, ox) - th[u] - bmu[u]);
          const double en = wrap_residual(atan2(ny, nx) - th[u] - bmu[u]);
          term = term + (eo - en) * (eo + en) * kb[u];
        }
        if (i0 + u * stride < hi) acc += term;
      }
    }
    return acc;
  };
  double acc = staged ? stream([&](uint32_t j) { return wpre[j]; }, [&](uint32_t j) { return wlo[j]; })
                      : stream([&](uint32_t j) { return segpre[j]; }, [&](uint32_t j) { return seglo[j]; });
#pragma unroll
  for (int dd = 32; dd > 0; dd >>= 1) acc += __shfl_down(acc, dd);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) d.sp_partial[size_t(c) * d.max_slices + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// edge_log_likelihood_ratio for a caller-supplied proposal; grid = 1.
__global__ void eval_kernel(Dev d_by_value, uint32_t c, int kind, uint32_t idx, double q0, double q1, double q2,
                            double* out) {
  MG_KERNEL_DEV(d);
  __shared__ Shared S;
  unsigned long long units = 0;
  double r;
  if (kind == MCMCGPU_STATE_EDGE) {
    const Pose q{q0, q1, q2};
    r = d.obear ? eval_state_edge<true>(d, c, S, idx, q, d.nf[c], units)
                : eval_state_edge<false>(d, c, S, idx, q, d.nf[c], units);
    if (threadIdx.x == 0) {
      double mu[3], sd[3], ld;
      load_control(d, idx, mu, sd, ld);
      const size_t ri = size_t(c) * d.Tcap + idx;
      const Pose cur{d.relx[ri], d.rely[ri], d.relth[ri]};
      r = (control_log_pdf(mu, sd, ld, q) - control_log_pdf(mu, sd, ld, cur)) + r;
    }
  } else {
    r = d.obear ? eval_feature_edge<true>(d, c, S, idx, Vec2{q0, q1}, units)
                : eval_feature_edge<false>(d, c, S, idx, Vec2{q0, q1}, units);
  }
  if (threadIdx.x == 0) *out = r;
}

// The accept branch applied from outside; grid = 1.
__global__ void commit_kernel(Dev d, uint32_t c, int kind, uint32_t idx, double q0, double q1, double q2,
                              double new_weight, double log_ratio, uint32_t Tcur) {
  if (kind == MCMCGPU_STATE_EDGE) {
    const Pose q{q0, q1, q2};
    double control_part = 0.0;  // what the edge's own control term changes by (kept apart for tempering)
    if (threadIdx.x == 0) {
      double mu[3], sd[3], ld;
      load_control(d, idx, mu, sd, ld);
      const size_t ri = size_t(c) * d.Tcap + idx;
      const double new_part = control_log_pdf(mu, sd, ld, q);
      control_part = new_part - control_log_pdf(mu, sd, ld, Pose{d.relx[ri], d.rely[ri], d.relth[ri]});
      d.clp[ri] = new_part;
    }
    commit_state_edge(d, c, idx, q, Tcur, d.nf[c]);
    if (threadIdx.x == 0) {
      const size_t ri = size_t(c) * d.Tcap + idx;
      d.relx[ri] = q.x;
      d.rely[ri] = q.y;
      d.relth[ri] = q.th;
      wt::set(wt::Plain{d.sw + size_t(c) * d.Tcap}, Tcur, idx, new_weight);
      d.ll[c] += log_ratio;
      d.llc[c] += control_part;
    }
  } else if (threadIdx.x == 0) {
    const size_t fi = size_t(c) * d.Fcap + idx;
    d.fex[fi] = q0;
    d.fey[fi] = q1;
    wt::set(wt::Plain{d.fw + size_t(c) * d.Fcap}, d.nf[c], idx, new_weight);
    d.ll[c] += log_ratio;
  }
}

// ---- cumulative pose rebuild: SE(2) scan ---------------------------------------
// P[k+1] = rel[0] o rel[1] o ... o rel[k] for every chain (P[0] = identity).  A chain is cut into
// `nseg` segments of `seg_len` relative poses, one workgroup each (grid = (nseg, chains)), which
// walks its segment tile by tile with the running composition in registers (chained_scan).  With
// many chains nseg = 1: one kernel, every pose read once and written once.  With few chains
// (a handful of replicas on a long trajectory) a first pass (WRITE = false) leaves only each
// segment's total in `tot`, scan_offsets_kernel turns the totals into the world pose at each
// segment's start, and the second pass (WRITE = true) starts from there.
struct ScanArgs {
  const double *relx, *rely, *relth;  // [chains][rstride] relative pose of edge k
  PoseXY* pxy;                        // [chains][pstride] slot k + 1 = world translation after edge k
  double* pth;                        // [chains][pstride] its unwrapped angle
  uint32_t rstride, pstride;
  uint32_t n;        // relative poses per chain
  uint32_t nseg;     // segments per chain
  uint32_t seg_len;  // relative poses per segment, a multiple of the tile
  Pose* tot;         // [chains][nseg] segment totals / offsets; unused when nseg == 1
};
constexpr int kScanLanes = 256, kScanPerLane = 4;          // tile = 1024 poses
constexpr uint32_t kScanTargetGroups = 1024;               // workgroups a rebuild should offer the GPU
constexpr uint32_t kScanMinSegmentTiles = 4;               // no segment shorter than this many tiles

// Four relative-pose components in a row as two 16-byte loads (8-byte aligned: a chain's slice may start
// on an odd element); element by element the compiler issues four 8-byte loads whose lanes are 32 bytes apart.
struct __attribute__((packed, aligned(8))) Double2 {
  double a, b;
};
MG_DEV void load4(const double* __restrict__ p, double out[4]) {
  const Double2 lo = *reinterpret_cast<const Double2*>(p), hi = *reinterpret_cast<const Double2*>(p + 2);
  out[0] = lo.a;
  out[1] = lo.b;
  out[2] = hi.a;
  out[3] = hi.b;
}

// (No occupancy bound: held to 128 registers for a fourth workgroup per CU the kernel spills and
// runs a quarter slower.)
template <bool WRITE>
__global__ void __launch_bounds__(kScanLanes) scan_chain_kernel(ScanArgs a) {
  __shared__ PoseCS stage[2][kScanMaxWaves];
  const uint32_t seg = blockIdx.x, c = blockIdx.y;
  const uint32_t lo = seg * a.seg_len;
  const uint32_t hi = a.n - lo < a.seg_len ? a.n : lo + a.seg_len;  // lo <= n by construction of the plan
  const double* __restrict__ rx = a.relx + size_t(c) * a.rstride;
  const double* __restrict__ ry = a.rely + size_t(c) * a.rstride;
  const double* __restrict__ rth = a.relth + size_t(c) * a.rstride;
  PoseXY* __restrict__ pxy = a.pxy + size_t(c) * a.pstride;
  double* __restrict__ pth = a.pth + size_t(c) * a.pstride;
  const PoseCSOp op;
  PoseCS carry = op.identity();
  if (WRITE && a.nseg > 1) carry = op.lift(a.tot[size_t(c) * a.nseg + seg]);
  if (WRITE && seg == 0 && threadIdx.x == 0) {
    pxy[0] = PoseXY{0.0, 0.0};
    pth[0] = 0.0;
  }
  static_assert(kScanPerLane == 4, "the row loader below fetches four elements");
  carry = chained_scan<kScanPerLane>(
      op, lo, hi, carry, [&](uint32_t i) { return Pose{rx[i], ry[i], rth[i]}; },
      [&](uint32_t first, Pose* out) {
        double x[4], y[4], th[4];
        load4(rx + first, x);
        load4(ry + first, y);
        load4(rth + first, th);
#pragma unroll
        for (int k = 0; k < 4; ++k) out[k] = Pose{x[k], y[k], th[k]};
      },
      [&](uint32_t i, const PoseCS& v) {
        if (WRITE) {
          pxy[i + 1] = PoseXY{v.x, v.y};
          pth[i + 1] = v.th;
        }
      },
      stage);
  if (!WRITE && threadIdx.x == 0) a.tot[size_t(c) * a.nseg + seg] = Pose{carry.x, carry.y, carry.th};
}

// ---- the same scan in ONE pass over memory, for few long chains ------------------------------------
// The two-pass form above reads every relative pose twice (72 bytes of traffic per pose instead of 48).
// Here a chain is cut into pieces of kPieceTiles tiles whose scanned poses a workgroup can keep in
// registers: it scans its piece relative to the piece's start, publishes the piece's total, composes the
// totals of the pieces before it in its chain in a fixed order -- oldest first, 64 at a time -- and
// places its poses.  The order of compositions depends on the piece's number only, not on timing, so two
// rebuilds of the same data give the same bits.  Pieces are handed out by a ticket counter: a workgroup
// only ever waits for pieces with lower tickets, which are running or done.  Reading ALL earlier totals
// would be quadratic in the pieces of a chain (at 326 pieces a third as many memory transactions again as
// the poses themselves), so the prefix is handed on once per window of 64 pieces; see the kernel.
constexpr int kPieceTiles = 3;                 // 3072 poses per piece, 12 per lane
constexpr uint32_t kPieceMaxPerChain = 1024;   // at most 16 hand-overs of the prefix along a chain
// A piece's total (x, y, cos, sin, angle) as five 64-bit words, each written and read by one device-wide
// atomic access of its own.  Before the launch every word holds kPieceEmpty (all ones: a NaN no total
// contains); a reader takes a total once none of its five words is that any more.  No word waits for
// another, so nothing needs a cache write-back or invalidation to be ordered: each access is coherent by
// itself (a release fence of device scope would write back the whole L2 of its XCD, per piece).
struct alignas(16) PieceSlot {
  unsigned long long word[5];
  unsigned long long pad;
};
static_assert(sizeof(PieceSlot) == 48, "piece slots are 48 bytes");
constexpr unsigned long long kPieceEmpty = ~0ull;
constexpr uint32_t kPieceHeaderBytes = 256;  // the scratch starts with the ticket counter

__device__ __forceinline__ void publish_piece_word(unsigned long long* word, double part) {
  unsigned long long bits = (unsigned long long)__double_as_longlong(part);
  // (input that carries this very NaN through the arithmetic must not look like an empty slot: readers would wait for ever)
  if (bits == kPieceEmpty) bits = 0x7FF8000000000000ull;
  __hip_atomic_store(word, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ PoseCS read_piece_slot(const PieceSlot* sl) {
  unsigned long long w[5];
  for (;;) {
    bool full = true;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      w[k] = __hip_atomic_load(&sl->word[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      full = full && w[k] != kPieceEmpty;
    }
    if (full) break;
    __builtin_amdgcn_s_sleep(2);
  }
  return PoseCS{__longlong_as_double((long long)w[0]), __longlong_as_double((long long)w[1]), __longlong_as_double((long long)w[2]),
                __longlong_as_double((long long)w[3]), __longlong_as_double((long long)w[4])};
}
// Slots the scratch holds behind its header for `chains` chains of `pieces` pieces: one per piece, then one
// per window of 64 pieces and chain for the prefix at that window's start.
__host__ __device__ constexpr uint64_t piece_slots_needed(uint64_t chains, uint64_t pieces) {
  return chains * (pieces + (pieces + 63) / 64);
}

// a.nseg = pieces per chain, a.seg_len = kPieceTiles tiles; `scratch` = [ticket = 0][chains * nseg slots of kPieceEmpty].
__global__ void __launch_bounds__(kScanLanes) scan_pieces_kernel(ScanArgs a, unsigned char* scratch) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int K = kScanPerLane, J = kPieceTiles, W = kScanLanes / 64;
  __shared__ PoseCS stage[J][W];
  __shared__ PoseCS before_piece;
  __shared__ uint32_t my_ticket;
  uint32_t* ticket = reinterpret_cast<uint32_t*>(scratch);
  PieceSlot* slots = reinterpret_cast<PieceSlot*>(scratch + kPieceHeaderBytes);
  if (threadIdx.x == 0) my_ticket = atomicAdd(ticket, 1u) + 1u;  // the counter starts at all ones like the rest of the scratch
  __syncthreads();
  // Tickets run piece after piece across the chains when a chain has more than one window of pieces, so
  // that the pieces in flight belong to many chains and few of them wait for a prefix to be handed on;
  // chain after chain otherwise.  Either way a workgroup only waits for lower tickets.
  const uint32_t chains = gridDim.x / a.nseg;
  const bool across = a.nseg > 64 || gridDim.x < 16384u;
  const uint32_t c = across ? my_ticket % chains : my_ticket / a.nseg;
  const uint32_t piece = across ? my_ticket / chains : my_ticket % a.nseg;
  const uint32_t windows_per_chain = (a.nseg + 63) / 64;
  PieceSlot* prefixes = slots + size_t(chains) * a.nseg;
  const uint32_t tile = uint32_t(kScanLanes) * K;
  const uint32_t lo = piece * a.seg_len;
  const uint32_t hi = a.n - lo < a.seg_len ? a.n : lo + a.seg_len;
  const double* __restrict__ rx = a.relx + size_t(c) * a.rstride;
  const double* __restrict__ ry = a.rely + size_t(c) * a.rstride;
  const double* __restrict__ rth = a.relth + size_t(c) * a.rstride;
  PoseXY* __restrict__ pxy = a.pxy + size_t(c) * a.pstride;
  double* __restrict__ pth = a.pth + size_t(c) * a.pstride;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const PoseCSOp op;
  // every load of the piece goes out before anything is computed
  double x[J][K], y[J][K], th[J][K];
#pragma unroll
  for (int j = 0; j < J; ++j) {
    const uint32_t first = lo + j * tile + threadIdx.x * K;
    if (first + K <= hi) {
      load4(rx + first, x[j]);
      load4(ry + first, y[j]);
      load4(rth + first, th[j]);
    } else {
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const bool in = first + k < hi;
        x[j][k] = in ? rx[first + k] : 0.0;
        y[j][k] = in ? ry[first + k] : 0.0;
        th[j][k] = in ? rth[first + k] : 0.0;
      }
    }
  }
  // the piece relative to its own start: lane runs, wavefront scans, one barrier for all tiles
  PoseCS excl[J];
#pragma unroll
  for (int j = 0; j < J; ++j) {
    const uint32_t first = lo + j * tile + threadIdx.x * K;
    PoseCS mine = op.identity();
#pragma unroll
    for (int k = 0; k < K; ++k) {
      // elements past the end are identities: (0, 0, 0) lifts to one
      const PoseCS v = op.lift(Pose{x[j][k], y[j][k], th[j][k]});
      mine = (k == 0) ? v : op(mine, v);
      // the lane's running composition takes the raw element's place (the angle stays in th)
      x[j][k] = mine.x;
      y[j][k] = mine.y;
      th[j][k] = mine.th;
      // its cosine and sine are needed once more below: for the last element of the run only
    }
    (void)first;
    const PoseCS inc = wave_scan_inclusive(mine, op);
    if (lane == 63) stage[j][wave] = inc;
    excl[j] = shfl_up_any(inc, 1);
    if (lane == 0) excl[j] = op.identity();
  }
  __syncthreads();
  // excl[j] becomes everything of the piece before this lane's run of tile j
  PoseCS all = op.identity();
#pragma unroll
  for (int j = 0; j < J; ++j) {
#pragma unroll
    for (int w = 0; w < W; ++w) {
      if (w == wave) excl[j] = op(all, excl[j]);
      all = op(all, stage[j][w]);
    }
  }
  // `all` is the piece's total: publish it, then collect what lies before the piece
  if (threadIdx.x < 5)
    publish_piece_word(&slots[size_t(c) * a.nseg + piece].word[threadIdx.x],
                       threadIdx.x == 0 ? all.x : threadIdx.x == 1 ? all.y : threadIdx.x == 2 ? all.c : threadIdx.x == 3 ? all.s : all.th);
  // What lies before the piece: the prefix of its chain up to the start of a window of 64 pieces, which
  // the first piece of the NEXT window has published (it needs that very value itself), composed with
  // the totals of the pieces from there on -- at most 64, one per lane.  So a piece reads at most 65
  // slots however long its chain is, and the compositions are the same whoever runs when.
  if (wave == 0) {
    const uint32_t base = piece == 0 ? 0u : ((piece - 1) & ~63u);  // a window's first piece looks back over the whole window before it
    PoseCS prefix = op.identity();
    if (base > 0) prefix = read_piece_slot(prefixes + size_t(c) * windows_per_chain + (base >> 6));
    PoseCS v = op.identity();
    if (base + lane < piece) v = read_piece_slot(slots + size_t(c) * a.nseg + base + lane);
    const PoseCS inc = wave_scan_inclusive(v, op);  // identities behind the last piece change nothing
    const PoseCS window = PoseCS{__shfl(inc.x, 63), __shfl(inc.y, 63), __shfl(inc.c, 63), __shfl(inc.s, 63), __shfl(inc.th, 63)};
    prefix = op(prefix, window);
    if (piece > 0 && (piece & 63u) == 0 && lane < 5)
      publish_piece_word(&prefixes[size_t(c) * windows_per_chain + (piece >> 6)].word[lane],
                         lane == 0 ? prefix.x : lane == 1 ? prefix.y : lane == 2 ? prefix.c : lane == 3 ? prefix.s : prefix.th);
    if (lane == 0) before_piece = prefix;
  }
  __syncthreads();
  const PoseCS P = before_piece;
  if (piece == 0 && threadIdx.x == 0) {
    pxy[0] = PoseXY{0.0, 0.0};
    pth[0] = 0.0;
  }
#pragma unroll
  for (int j = 0; j < J; ++j) {
    const uint32_t first = lo + j * tile + threadIdx.x * K;
    const PoseCS start = op(P, excl[j]);  // everything before this lane's run
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if (first + k < hi) {
        pxy[first + k + 1] = PoseXY{fma(start.c, x[j][k], fma(-start.s, y[j][k], start.x)),
                                    fma(start.s, x[j][k], fma(start.c, y[j][k], start.y))};
        pth[first + k + 1] = start.th + th[j][k];
      }
    }
  }
#endif
}

// grid = chains.  Exclusive scan of a chain's segment totals, in place: tot[s] becomes the world
// pose at the start of segment s.
__global__ void scan_offsets_kernel(uint32_t nseg, Pose* tot) {
  __shared__ Pose stage[kMaxWaves];
  __shared__ Pose incl[256];
  Pose* t = tot + size_t(blockIdx.x) * nseg;
  Pose carry = pose_identity();
  for (uint32_t b0 = 0; b0 < nseg; b0 += blockDim.x) {
    const uint32_t b = b0 + threadIdx.x;
    Pose v = pose_identity();
    if (b < nseg) v = t[b];
    Pose total;
    incl[threadIdx.x] = block_scan_inclusive(v, PoseOp{}, stage, total);
    __syncthreads();
    // exclusive value of lane j = carry o inclusive value of lane j-1
    const Pose excl = threadIdx.x == 0 ? carry : compose(carry, incl[threadIdx.x - 1]);
    __syncthreads();
    if (b < nseg) t[b] = excl;
    carry = compose(carry, total);
  }
}

// World landmark positions and plain edge weights of one chain.
__global__ void readout_kernel(Dev d, uint32_t c, uint32_t Tcur, double* feat_world, double* state_w,
                               double* feat_w) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t nf = d.nf[c];
  if (i < nf) {
    const size_t fi = size_t(c) * d.Fcap + i;
    const Vec2 w = act(world_pose(d, c, d.fpt[fi]), Vec2{d.fex[fi], d.fey[fi]});
    feat_world[2 * size_t(i)] = w.x;
    feat_world[2 * size_t(i) + 1] = w.y;
    feat_w[i] = wt::get(wt::Plain{d.fw + size_t(c) * d.Fcap}, i);
  }
  if (i < Tcur) state_w[i] = wt::get(wt::Plain{d.sw + size_t(c) * d.Tcap}, i);
}

// One chain's estimate packed for a hand-off to the host: relative poses [Tcur][3], world landmark
// positions [nf][2] (get_trajectory / get_feature_map, mcmc_slam.hpp:194-196, :507-525), then the
// log likelihood and nf as a double.  One thread per pose or landmark.
__global__ void snapshot_kernel(Dev d, uint32_t c, uint32_t Tcur, double* out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t nf = d.nf[c];
  if (i < Tcur) {
    const size_t ri = size_t(c) * d.Tcap + i;
    out[3 * size_t(i)] = d.relx[ri];
    out[3 * size_t(i) + 1] = d.rely[ri];
    out[3 * size_t(i) + 2] = d.relth[ri];
  }
  if (i < nf) {
    const size_t fi = size_t(c) * d.Fcap + i;
    const Vec2 w = act(world_pose(d, c, d.fpt[fi]), Vec2{d.fex[fi], d.fey[fi]});
    out[3 * size_t(Tcur) + 2 * size_t(i)] = w.x;
    out[3 * size_t(Tcur) + 2 * size_t(i) + 1] = w.y;
  }
  if (i == 0) {
    out[3 * size_t(Tcur) + 2 * size_t(nf)] = d.ll[c];
    out[3 * size_t(Tcur) + 2 * size_t(d.Fcap) + 1] = double(nf);  // fixed place: the host does not know nf yet
  }
}

// slam_log_likelihood (src/slam/slam_likelihood.hpp:26-58); grid = chains from `first`.
// out[c] = the log likelihood from scratch; out_controls[c] (optional) = its control-model part.
__global__ void full_ll_kernel(Dev d, uint32_t first, uint32_t Tcur, double* out, double* out_controls) {
  __shared__ Shared S;
  const uint32_t c = first + blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
  const uint32_t segcap = nthr < uint32_t(kMaxSeg) ? nthr : uint32_t(kMaxSeg);
  double acc = 0.0;
  for (uint32_t t = tid; t < Tcur; t += nthr) {
    double mu[3], sd[3], ld;
    load_control(d, t, mu, sd, ld);
    const size_t ri = size_t(c) * d.Tcap + t;
    acc += control_log_pdf(mu, sd, ld, Pose{d.relx[ri], d.rely[ri], d.relth[ri]});
  }
  if (out_controls) {
    const double controls = block_sum(acc, S);
    if (tid == 0) out_controls[c] = controls;
    __syncthreads();
  }
  const uint32_t nf = d.nf[c];
  const size_t fbase = size_t(c) * d.Fcap;
  for (uint32_t f0 = 0; f0 < nf; f0 += segcap) {
    const uint32_t nseg = (nf - f0) < segcap ? (nf - f0) : segcap;
    uint32_t len = 0;
    if (tid < nseg) {
      const uint32_t f = f0 + tid;
      const Vec2 w = act(world_pose(d, c, d.fpt[fbase + f]), Vec2{d.fex[fbase + f], d.fey[fbase + f]});
      S.seglo[tid] = d.foff[f];
      S.segw[tid][0] = w.x;
      S.segw[tid][1] = w.y;
      S.segw[tid][2] = w.x;
      S.segw[tid][3] = w.y;
      len = d.fcnt[fbase + f];
    }
    uint32_t total;
    const uint32_t incl = block_scan_u32(len, S, total);
    if (tid < nseg) S.segpre[tid + 1] = incl;
    if (tid == 0) S.segpre[0] = 0;
    __syncthreads();
    acc += d.obear ? full_ll_segments_rb(d, c, S, nseg) : full_ll_segments(d, c, S, nseg);
    __syncthreads();
  }
  const double r = block_sum(acc, S);
  if (tid == 0) out[c] = r;
}

// One appended timestep (or a piece of one), handed over as kernel arguments: the new control and
// up to kAppendMax readings with the slots the host has chosen for them.  Writes what
// mcmcgpu_upload_problem would have put there, in place; the arrays are read-only for every other
// kernel, so the const is cast away here and nowhere else.
constexpr int kAppendMax = 16;
struct AppendArgs {
  uint32_t t;            // timestep of the readings (and of the pose the control leads to)
  uint32_t has_control;  // control of edge t-1 -> t is in cm / cs
  double cm[3], cs[3];
  uint32_t has_init_rel;  // initial estimate of that edge is in irel (set_initialiser)
  double irel[3];
  uint32_t n;             // readings in this piece
  uint32_t arrival0;      // arrival index of the first of them
  uint32_t arrivals_end;  // arrival index one past the last reading of timestep t so far
  uint32_t slot[kAppendMax], f[kAppendMax];
  uint32_t is_new[kAppendMax];    // first reading of its landmark
  uint32_t has_ifeat[kAppendMax];  // an initial estimate for that new landmark is in ifeat
  double mu[kAppendMax], sd[kAppendMax], bmu[kAppendMax], bsd[kAppendMax];
  double ifeat[kAppendMax][2];
};
// The constant of every control's log density, from the uploaded deviations; one lane per control.
__global__ void control_log_det_kernel(Dev d, uint32_t n) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const_cast<double*>(d.cld)[t] = control_log_det(d.csd[t], d.csd[size_t(d.Tcap) + t], d.csd[2 * size_t(d.Tcap) + t]);
}

__global__ void append_kernel(Dev d, AppendArgs a) {
  const uint32_t i = threadIdx.x;
  if (i == 0) {
    if (a.has_control) {
      for (int k = 0; k < 3; ++k) {
        const_cast<double*>(d.cmu)[size_t(k) * d.Tcap + (a.t - 1)] = a.cm[k];
        const_cast<double*>(d.csd)[size_t(k) * d.Tcap + (a.t - 1)] = a.cs[k];
        if (a.has_init_rel) const_cast<double*>(d.init_rel)[3 * size_t(a.t - 1) + k] = a.irel[k];
      }
      const_cast<double*>(d.cld)[a.t - 1] = control_log_det(a.cs[0], a.cs[1], a.cs[2]);
    }
    const_cast<uint32_t*>(d.toff)[a.t + 1] = a.arrivals_end;
  }
  if (i < a.n) {
    const uint32_t s = a.slot[i], f = a.f[i];
    const_cast<ObsRec*>(d.orec)[s] = ObsRec{a.mu[i], a.t, f};
    const_cast<StreamRec*>(d.osr)[s] = StreamRec{a.mu[i], a.t};
    const_cast<double*>(d.osd)[s] = a.sd[i];
    const_cast<double*>(d.oisd)[s] = 1.0 / a.sd[i];
    if (d.obear) {
      const_cast<double*>(d.obear)[s] = a.bmu[i];
      const_cast<double*>(d.obsd)[s] = a.bsd[i];
      const_cast<double*>(d.obisd)[s] = 1.0 / a.bsd[i];
    }
    const_cast<uint32_t*>(d.tobs)[a.arrival0 + i] = s;
    if ((s & 31u) == 0) const_cast<uint32_t*>(d.opiv)[s >> 5] = a.t;
    const_cast<uint32_t*>(d.otime)[s] = a.t;
    if (a.is_new[i]) {
      const_cast<uint32_t*>(d.foff)[f] = s;
      const_cast<uint32_t*>(d.ffirst)[f] = a.t;
      if (a.has_ifeat[i]) {
        const_cast<double*>(d.init_feat)[2 * size_t(f)] = a.ifeat[i][0];
        const_cast<double*>(d.init_feat)[2 * size_t(f) + 1] = a.ifeat[i][1];
      }
    }
  }
}

// One landmark's list of `count` readings moved from slot `from` to slot `to` (a free region: the
// two do not overlap): records, deviations, pivots of the new region, the landmark's start, and
// the arrival index, in which every moved reading is looked up among the arrivals of its timestep.
__global__ void move_list_kernel(Dev d, uint32_t f, uint32_t from, uint32_t to, uint32_t count) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) {
    const ObsRec r = d.orec[from + i];
    const uint32_t s = to + i;
    const_cast<ObsRec*>(d.orec)[s] = r;
    const_cast<StreamRec*>(d.osr)[s] = StreamRec{r.mu, r.t};
    const_cast<double*>(d.osd)[s] = d.osd[from + i];
    const_cast<double*>(d.oisd)[s] = d.oisd[from + i];
    if (d.obear) {
      const_cast<double*>(d.obear)[s] = d.obear[from + i];
      const_cast<double*>(d.obsd)[s] = d.obsd[from + i];
      const_cast<double*>(d.obisd)[s] = d.obisd[from + i];
    }
    if ((s & 31u) == 0) const_cast<uint32_t*>(d.opiv)[s >> 5] = r.t;
    const_cast<uint32_t*>(d.otime)[s] = r.t;
    for (uint32_t a = d.toff[r.t]; a < d.toff[r.t + 1]; ++a)
      if (d.tobs[a] == from + i) {
        const_cast<uint32_t*>(d.tobs)[a] = s;
        break;
      }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) const_cast<uint32_t*>(d.foff)[f] = to;
}

__global__ void init_chains_kernel(Dev d, const uint32_t* seeds) {
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= d.C) return;
  uint32_t* mt = d.mt + size_t(c) * 624;
  mt[0] = seeds[c];
  for (int i = 1; i < 624; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + uint32_t(i);
  d.mti[c] = 624;
  d.nflag[c] = 0;
  d.nsaved[c] = 0.0;
  d.nf[c] = 0;
  d.ll[c] = 0.0;
  d.n_updates[c] = 0;
  d.n_accepts[c] = 0;
  d.n_units[c] = 0;
  d.n_state_props[c] = 0;
  d.n_moved[c] = 0;
  d.errflag[c] = 0;
  d.beta[c] = 1.0;
  d.llc[c] = 0.0;
  d.slot_chain[c] = c;
  if (d.trace_n) d.trace_n[c] = 0;
  d.pxy[size_t(c) * (d.Tcap + 1)] = PoseXY{0.0, 0.0};
  d.pth[size_t(c) * (d.Tcap + 1)] = 0.0;
}

// Replica exchange between neighbouring temperature slots (phase + 2k, phase + 2k + 1) of every
// ladder of R slots; one thread per pair.  The two chains swap their temperatures (not their
// states) with probability min(1, exp((beta_a - beta_b) (L_b - L_a))), L = the readings' part of
// the log likelihood (ll - llc), which is what the temperatures weigh; u[pair] is the uniform of
// that pair.  counts[0] += attempts, counts[1] += swaps.
__global__ void exchange_kernel(Dev d, uint32_t R, uint32_t phase, const double* u, unsigned long long* counts) {
  const uint32_t per_ladder = (R - phase) / 2;
  const uint32_t pair = blockIdx.x * blockDim.x + threadIdx.x;
  if (pair >= per_ladder * (d.C / R)) return;
  const uint32_t s = (pair / per_ladder) * R + phase + 2 * (pair % per_ladder);
  const uint32_t a = d.slot_chain[s], b = d.slot_chain[s + 1];
  const double ba = d.beta[a], bb = d.beta[b];
  const double log_ratio = (ba - bb) * ((d.ll[b] - d.llc[b]) - (d.ll[a] - d.llc[a]));
  atomicAdd(&counts[0], 1ull);
  if (u[pair] < exp(log_ratio)) {
    d.beta[a] = bb;
    d.beta[b] = ba;
    d.slot_chain[s] = b;
    d.slot_chain[s + 1] = a;
    atomicAdd(&counts[1], 1ull);
  }
}

// average_slam_result (src/slam/average_slam_result.hpp:58-110) in two kernels.  The vectors that
// are averaged: per timestep t the accumulated pose trajectory.accumulate(t+1) as (x, y, angle)
// (:81-83), per landmark its world position (:85-87, initial state = identity); packed as
// [Tcur][3] then [nf][2] doubles.  One thread per vector.
MG_DEV void average_vector(const Dev& d, uint32_t c, uint32_t Tcur, uint32_t e, double v[3]) {
  if (e < Tcur) {
    const Pose p = world_pose(d, c, e + 1);
    v[0] = p.x;
    v[1] = p.y;
    v[2] = p.th;
  } else {
    const size_t fi = size_t(c) * d.Fcap + (e - Tcur);
    const Vec2 w = act(world_pose(d, c, d.fpt[fi]), Vec2{d.fex[fi], d.fey[fi]});
    v[0] = w.x;
    v[1] = w.y;
    v[2] = 0.0;
  }
}

// The first result's vectors: avg_acc::base (:42-44).
__global__ void average_base_kernel(Dev d, uint32_t c, uint32_t Tcur, uint32_t nf, double* base) {
  const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= Tcur + nf) return;
  double v[3];
  average_vector(d, c, Tcur, e, v);
  if (e < Tcur) {
    base[3 * size_t(e)] = v[0];
    base[3 * size_t(e) + 1] = v[1];
    base[3 * size_t(e) + 2] = v[2];
  } else {
    base[3 * size_t(Tcur) + 2 * size_t(e - Tcur)] = v[0];
    base[3 * size_t(Tcur) + 2 * size_t(e - Tcur) + 1] = v[1];
  }
}

// sum_innov (:45-46): sum over this handle's chains, in chain order and without `skip`, of
// subtract(base, vector) -- componentwise, the angle difference wrapped (pose.hpp:63-65).
__global__ void average_sum_kernel(Dev d, uint32_t Tcur, uint32_t nf, const double* base, int32_t skip, double* sum) {
  const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= Tcur + nf) return;
  const bool is_pose = e < Tcur;
  const size_t at = is_pose ? 3 * size_t(e) : 3 * size_t(Tcur) + 2 * size_t(e - Tcur);
  const double b0 = base[at], b1 = base[at + 1], b2 = is_pose ? base[at + 2] : 0.0;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0;
  for (uint32_t c = 0; c < d.C; ++c) {
    if (int32_t(c) == skip) continue;
    double v[3];
    average_vector(d, c, Tcur, e, v);
    s0 += b0 - v[0];
    s1 += b1 - v[1];
    if (is_pose) s2 += wrap_angle(b2 - v[2]);
  }
  sum[at] = s0;
  sum[at + 1] = s1;
  if (is_pose) sum[at + 2] = s2;
}

// avg_acc::get (:50-53): subtract(base, sum_innov / n), in place in `base`.
__global__ void average_finish_kernel(uint32_t Tcur, uint32_t nf, double* base, const double* sum, double n) {
  const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= Tcur + nf) return;
  const bool is_pose = e < Tcur;
  const size_t at = is_pose ? 3 * size_t(e) : 3 * size_t(Tcur) + 2 * size_t(e - Tcur);
  base[at] = base[at] - sum[at] / n;
  base[at + 1] = base[at + 1] - sum[at + 1] / n;
  if (is_pose) base[at + 2] = wrap_angle(base[at + 2] - sum[at + 2] / n);
}

__global__ void summaries_kernel(Dev d, mcmcgpu_chain_summary* out) {
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= d.C) return;
  out[c].log_likelihood = d.ll[c];
  out[c].updates = uint32_t(d.n_updates[c]);
  out[c].accepts = uint32_t(d.n_accepts[c]);
}

// Report records [intervals][C] laid out for an all-gather between handles that may hold different
// numbers of chains: [intervals][slots] with slots >= C, the unused slots filled with a record no
// arg-max can pick (log likelihood -inf, no updates).
__global__ void pack_records_kernel(const mcmcgpu_chain_summary* src, uint32_t C, uint32_t intervals,
                                    mcmcgpu_chain_summary* dst, uint32_t slots) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= intervals * slots) return;
  const uint32_t k = i / slots, s = i % slots;
  mcmcgpu_chain_summary r;
  r.log_likelihood = -__builtin_huge_val();
  r.updates = 0;
  r.accepts = 0;
  if (s < C) r = src[size_t(k) * C + s];
  dst[i] = r;
}

// Recomputes both weight trees of chains [first, first + gridDim.x) from their
// current estimates (each edge weight is a function of the edge's current
// value: mcmc_slam.hpp:240-241, 262-263, 388-390).  A tree node only depends on
// nodes whose lowest set index bit is smaller, so the tree is filled level by
// level; inside a node the additions happen in the order push_back would use.
MG_DEV void build_tree_levels(double* node, uint32_t n) {
  for (uint32_t low = 2; low < n; low <<= 1) {
    __syncthreads();
    for (uint32_t i = low * (2 * threadIdx.x + 1); i < n; i += 2 * low * blockDim.x)
      node[i] = wt::sum_to_ancestor(wt::Plain{node}, i - 1, wt::parent(i)) + node[i];
  }
  __syncthreads();
}
__global__ void reweigh_kernel(Dev d, uint32_t first, uint32_t Tcur) {
  const uint32_t c = first + blockIdx.x;
  double* sw = d.sw + size_t(c) * d.Tcap;
  for (uint32_t t = threadIdx.x; t < Tcur; t += blockDim.x) {
    double mu[3], sd[3], ld;
    load_control(d, t, mu, sd, ld);
    const size_t ri = size_t(c) * d.Tcap + t;
    const double lp = control_log_pdf(mu, sd, ld, Pose{d.relx[ri], d.rely[ri], d.relth[ri]});
    d.clp[ri] = lp;
    sw[t] = edge_weight_from_log_pdf(lp, 3.0, nullptr);
  }
  build_tree_levels(sw, Tcur);
  double* fw = d.fw + size_t(c) * d.Fcap;
  const uint32_t nf = d.nf[c];
  for (uint32_t f = threadIdx.x; f < nf; f += blockDim.x) {
    const size_t fi = size_t(c) * d.Fcap + f;
    const Reading parent = load_reading(d, d.foff[f] + d.fpar[fi]);
    keep_parent_reading(d, fi, parent);  // the landmarks' state has just been loaded: take the copy anew
    const double lp = reading_proposal_log_pdf(parent, Vec2{d.fex[fi], d.fey[fi]});
    fw[f] = edge_weight_from_log_pdf(lp, 2.0, nullptr);
  }
  build_tree_levels(fw, nf);
}

// Copies one staged state into chains [first, first + gridDim.x): relative
// poses rel[T][3] and the per-landmark bookkeeping fmeta[4][F] (parent reading,
// readings seen, parent timestep, last timestep).
__global__ void load_state_kernel(Dev d, uint32_t first, uint32_t T, uint32_t F, const double* rel,
                                  const uint32_t* fmeta) {
  const uint32_t c = first + blockIdx.x;
  const size_t rbase = size_t(c) * d.Tcap, fbase = size_t(c) * d.Fcap;
  for (uint32_t t = threadIdx.x; t < T; t += blockDim.x) {
    d.relx[rbase + t] = rel[3 * size_t(t)];
    d.rely[rbase + t] = rel[3 * size_t(t) + 1];
    d.relth[rbase + t] = rel[3 * size_t(t) + 2];
  }
  for (uint32_t f = threadIdx.x; f < F; f += blockDim.x) {
    d.fpar[fbase + f] = fmeta[f];
    d.fcnt[fbase + f] = fmeta[size_t(F) + f];
    d.fpt[fbase + f] = fmeta[2 * size_t(F) + f];
    d.flast[fbase + f] = fmeta[3 * size_t(F) + f];
  }
  if (threadIdx.x == 0) d.nf[c] = F;
}

// Landmark estimates relative to their parent poses from world positions;
// grid = (ceil(F / block), chains).
__global__ void features_from_world_kernel(Dev d, uint32_t first, uint32_t F, const double* world) {
  const uint32_t c = first + blockIdx.y, f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= F) return;
  const size_t fi = size_t(c) * d.Fcap + f;
  const Vec2 e = act_inverse(world_pose(d, c, d.fpt[fi]), Vec2{world[2 * size_t(f)], world[2 * size_t(f) + 1]});
  d.fex[fi] = e.x;
  d.fey[fi] = e.y;
}

// The copies of the landmarks' parent poses, taken anew from pose tables that a scan has just
// rewritten; grid = (ceil(Fcap / block), chains), chains first .. first + gridDim.y - 1.
__global__ void parent_poses_kernel(Dev d, uint32_t first) {
  const uint32_t c = first + blockIdx.y, f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= d.nf[c]) return;
  const size_t fi = size_t(c) * d.Fcap + f;
  keep_parent_pose(d, c, fi, d.fpt[fi]);
}

// Scan self-test over the integer affine group (exact): the chained scan that builds the pose
// tables, one workgroup, on the group of the reference's own tree test.
__global__ void __launch_bounds__(kScanLanes) scan_affine_kernel(uint32_t n, int32_t p, const int32_t* a,
                                                                  const int32_t* b, int32_t* oa, int32_t* ob) {
  __shared__ Affine stage[2][kScanMaxWaves];
  const AffineOp op{p};
  chained_scan<kScanPerLane>(
      op, 0u, n, op.identity(), [&](uint32_t i) { return Affine{a[i], b[i]}; },
      row_by_elements<kScanPerLane, Affine>([&](uint32_t i) { return Affine{a[i], b[i]}; }),
      [&](uint32_t i, const Affine& v) {
        oa[i] = v.a;
        ob[i] = v.b;
      },
      stage);
}

// Self-tests of the wavefront forms of the generator and of the weight-tree descent against the serial
// ones they must equal bit for bit.  One workgroup of 64 lanes each.  kinds[i]: 0 = uniform, 1 = normal,
// 2 / 3 = three / two normals drawn as a batch (their weighted sum is the output).
struct SelftestLds {
  double pad[16];  // the generator's state sits where a chain's does: not at the start of the workgroup's LDS
  RngState rng;
};
template <class R>
MG_DEV double selftest_draw(R& r, uint8_t kind) {
  if (kind == 0) return r.uniform();
  if (kind == 1) return r.normal();
  double v[3] = {0.0, 0.0, 0.0};
  r.normals(v, kind == 2 ? 3 : 2);
  return v[0] + 2.0 * v[1] + 4.0 * v[2];
}
__global__ void __launch_bounds__(64) serial_draws_selftest_kernel(uint32_t seed, uint32_t n, const uint8_t* kinds, double* out) {
  __shared__ SelftestLds L;
  if (threadIdx.x == 0) {
    rng_seed(L.rng, seed);
    SerialRng r{L.rng};
    for (uint32_t i = 0; i < n; ++i) out[i] = selftest_draw(r, kinds[i]);
  }
}
__global__ void __launch_bounds__(64) wave_draws_selftest_kernel(uint32_t seed, uint32_t n, const uint8_t* kinds, double* out) {
#if defined(__HIP_DEVICE_COMPILE__)  // the wavefront forms exist in device code only
  __shared__ SelftestLds L;
  if (threadIdx.x == 0) rng_seed(L.rng, seed);
  __syncthreads();
  for (uint32_t i0 = 0; i0 < n; i0 += 7) {  // a fresh window object every few draws, as one per update
    WaveRng r(L.rng);
    for (uint32_t i = i0; i < n && i < i0 + 7; ++i) {
      const double v = selftest_draw(r, kinds[i]);
      if (threadIdx.x == 0) out[i] = v;
    }
    r.flush();
  }
#endif
}
// nodes: the tree in the reference's layout (built on the host); index pairs out: serial, wave
__global__ void __launch_bounds__(64) search_selftest_kernel(uint32_t tree_n, double* nodes, uint32_t ns, const double* us,
                                                             uint32_t* is, uint32_t* iw) {
#if defined(__HIP_DEVICE_COMPILE__)
  const wt::Plain t{nodes};
  const double total = wt::total(t, tree_n);
  for (uint32_t j = 0; j < ns; ++j) {
    const double value = total * us[j];
    const uint32_t s = wt::search(t, tree_n, value), w = wt::search_wave(t, tree_n, value);
    if (threadIdx.x == 0) {
      is[j] = s;
      iw[j] = w;
    }
  }
#endif
}

}  // namespace mg

#endif  // SLAM_AMD_CHAIN_KERNELS_HPP
