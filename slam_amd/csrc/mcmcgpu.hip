// Host side of the C ABI declared in include/mcmcgpu.h: device memory layout,
// problem flattening (per-landmark CSR + arrival index) and kernel launches.
// All device memory is allocated once in mcmcgpu_create for the capacities the
// caller states; no other entry point allocates device memory.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <set>
#include <string>
#include <type_traits>
#include <vector>

#include "chain_kernels.hpp"
#include "rccl_loader.hpp"

using namespace mg;

namespace {

thread_local std::string g_create_error;

struct EventPair {
  hipEvent_t start, stop;
  int kind;  // 0 = the update launches (advance kernel or a split-mode sequence), 1 = a pose-table rebuild
};

}  // namespace

struct mcmcgpu_ctx {
  mcmcgpu_config cfg{};
  hipStream_t stream = nullptr;
  bool own_stream = false;
  Dev d{};
  std::vector<void*> allocations;
  Pose* scan_tot = nullptr;  // [scan_total_slots(C)] segment totals / offsets of the rebuild scan
  uint32_t* seed_stage = nullptr;  // [C]
  double* scratch = nullptr;  // readout staging: max(3*(Tcap+1), ...) doubles
  size_t scratch_doubles = 0;
  uint32_t nthreads = 256;
  std::set<const void*> advance_lds_raised;  // advance_kernel variants whose dynamic LDS limit has been raised above 48 KB
  // host-side bookkeeping
  uint32_t chains = 0;
  bool problem_loaded = false;
  uint32_t next_timestep = 0;  // first timestep the chains have not processed
  bool bearings_loaded = false;  // range-and-bearing handles: every reading of the current problem has its bearing
  // The problem as it was given, in arrival order: kept for re-layouts (see lay_out) and appends.
  std::vector<double> a_cmean, a_cstd;  // [T][3]
  std::vector<uint32_t> a_t, a_f;       // [N]
  std::vector<double> a_mean, a_sd, a_bmean, a_bsd;
  // Landmark-major slots on the device: landmark f owns slots [f_start[f], f_start[f] + f_cap[f]), the
  // first f_count[f] of them filled, in time order.  A compact layout has f_cap == f_count; after an
  // append has outgrown it, capacities are doubled so that most appends write one slot in place.
  std::vector<uint32_t> f_start, f_count, f_cap;
  uint32_t slots_used = 0;
  std::vector<uint32_t> h_tobs;  // arrival order -> slot
  std::vector<std::vector<uint32_t>> f_arrivals;  // landmark -> arrival indices of its readings, in time order
  std::vector<uint32_t> h_ot;    // slot -> timestep (filled slots)
  std::vector<double> h_omu;     // slot -> measured range
  double sigma_all = 0.0, bsigma_all = 0.0;  // the one sigma every range / bearing has so far, or 0
  std::vector<EventPair> timings;      // launches not yet folded into the totals below
  std::vector<EventPair> free_events;  // pairs ready for reuse
  double timed_ms[2] = {0.0, 0.0};     // finished launches already folded (see fold_timings), per kind
  uint32_t timed_launches[2] = {0, 0};
  // report records of the last mcmcgpu_run_reported that used the library's own buffer
  mcmcgpu_chain_summary* reports = nullptr;  // [kMaxOwnReports + 1][max_chains]; the last row stages mcmcgpu_allgather_summaries
  uint32_t report_intervals = 0;
  // hand-off snapshot: device staging, pinned host landing buffer, completion event
  double* snap_dev = nullptr;
  double* snap_host = nullptr;
  size_t snap_doubles = 0;
  hipEvent_t snap_event = nullptr;
  bool snap_pending = false;
  uint32_t snap_T = 0;
  // tempered replicas
  uint32_t ladder = 0;  // replicas per ladder, 0 = no ladders set
  RngState exchange_rng{};
  double* exch_u = nullptr;                 // [C/2 + 1] uniforms of one exchange sweep
  unsigned long long* exch_counts = nullptr;  // attempts, swaps
  // exchange with other handles (mcmcgpu_comm_*): communicator, this handle's place in it, buffers
  ncclComm_t comm = nullptr;
  int comm_ranks = 0, comm_rank = 0;
  uint32_t comm_slots = 0;                         // records per rank and report interval in a gather
  mcmcgpu_chain_summary* gather_send = nullptr;    // [kMaxOwnReports][comm_slots]
  mcmcgpu_chain_summary* gather_recv = nullptr;    // [comm_ranks][kMaxOwnReports][comm_slots]
  uint32_t gathered_intervals = 0;
  double* avg_base = nullptr;                      // [3 Tcap + 2 Fcap] averaging: the base vectors
  double* avg_sum = nullptr;                       // [3 Tcap + 2 Fcap] averaging: the summed differences
  double* bcast = nullptr;                         // [3 Tcap + 2 Fcap] a chain's packed state, as broadcast
  std::string error;
};

namespace {

int fail(mcmcgpu_handle h, int code, const std::string& msg) {
  if (h)
    h->error = msg;
  else
    g_create_error = msg;
  return code;
}

#define HIP_TRY(h, expr)                                                                    \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess)                                                                   \
      return fail(h, MCMCGPU_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

template <class T>
int dev_alloc(mcmcgpu_handle h, T** out, size_t count) {
  void* p = nullptr;
  const size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
  hipError_t e = hipMalloc(&p, bytes);
  if (e != hipSuccess)
    return fail(h, MCMCGPU_ERR_HIP, "hipMalloc of " + std::to_string(bytes) + " bytes: " + hipGetErrorString(e));
  e = hipMemsetAsync(p, 0, bytes, h->stream);
  if (e != hipSuccess) return fail(h, MCMCGPU_ERR_HIP, std::string("hipMemsetAsync: ") + hipGetErrorString(e));
  h->allocations.push_back(p);
  *out = static_cast<T*>(p);
  return 0;
}

uint32_t edges_grown(const mcmcgpu_ctx* h) { return h->next_timestep == 0 ? 0 : h->next_timestep - 1; }

constexpr uint32_t kDeferCommitPoses = 200000;  // split mode: trajectories from here on get split_commit_kernel
constexpr uint32_t kTimeBlockShift = 8;  // timesteps per block of the per-landmark split table: 256
constexpr uint32_t kMaxOwnReports = 256;  // intervals of one launch the library can hold itself

bool is_pow2(uint32_t v) { return v && !(v & (v - 1)); }

int check_device_flags(mcmcgpu_handle h) {
  std::vector<uint32_t> flags(h->chains);
  if (h->chains == 0) return 0;
  HIP_TRY(h, hipMemcpyAsync(flags.data(), h->d.errflag, sizeof(uint32_t) * h->chains, hipMemcpyDeviceToHost,
                            h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  for (uint32_t c = 0; c < h->chains; ++c)
    if (flags[c])
      return fail(h, MCMCGPU_ERR_DEVICE_FAULT,
                  "chain " + std::to_string(c) + " stopped: edge selection did not terminate");
  return 0;
}

constexpr size_t kMaxPendingTimings = 1024;

// Adds the durations of finished launches to the handle's totals and frees their event pairs for
// reuse; with `wait` the stream is drained first so that every launch is finished.  Keeps a caller
// that never asks for mcmcgpu_kernel_time (an online replay of 100 000 timesteps, one launch each)
// from holding an event pair per launch.
int fold_timings(mcmcgpu_handle h, bool wait) {
  if (wait) HIP_TRY(h, hipStreamSynchronize(h->stream));
  size_t kept = 0;
  for (size_t i = 0; i < h->timings.size(); ++i) {
    const EventPair ev = h->timings[i];
    // launches finish in stream order: after the first unfinished one nothing later is done either
    if (kept == 0 && (wait || hipEventQuery(ev.stop) == hipSuccess)) {
      float ms = 0.f;
      HIP_TRY(h, hipEventElapsedTime(&ms, ev.start, ev.stop));
      h->timed_ms[ev.kind] += ms;
      ++h->timed_launches[ev.kind];
      h->free_events.push_back(ev);
    } else {
      h->timings[kept++] = ev;
    }
  }
  h->timings.resize(kept);
  return 0;
}

// An event pair for one timed launch, `start` already recorded on the stream.
int begin_timed(mcmcgpu_handle h, EventPair* ev, int kind = 0) {
  if (h->timings.size() >= kMaxPendingTimings)
    if (int rc = fold_timings(h, false)) return rc;
  if (!h->free_events.empty()) {
    *ev = h->free_events.back();
    h->free_events.pop_back();
  } else {
    HIP_TRY(h, hipEventCreate(&ev->start));
    hipError_t e = hipEventCreate(&ev->stop);
    if (e != hipSuccess) {
      (void)hipEventDestroy(ev->start);
      return fail(h, MCMCGPU_ERR_HIP, std::string("hipEventCreate: ") + hipGetErrorString(e));
    }
  }
  ev->kind = kind;
  hipError_t e = hipEventRecord(ev->start, h->stream);
  if (e != hipSuccess) {
    h->free_events.push_back(*ev);
    return fail(h, MCMCGPU_ERR_HIP, std::string("hipEventRecord: ") + hipGetErrorString(e));
  }
  return 0;
}

// Closes a timed launch: records `stop` when the launches went out (launch_error == hipSuccess),
// gives the pair back otherwise.
int end_timed(mcmcgpu_handle h, const EventPair& ev, hipError_t launch_error, const char* what) {
  if (launch_error == hipSuccess) launch_error = hipEventRecord(ev.stop, h->stream);
  if (launch_error != hipSuccess) {
    h->free_events.push_back(ev);
    return fail(h, MCMCGPU_ERR_HIP, std::string(what) + ": " + hipGetErrorString(launch_error));
  }
  h->timings.push_back(ev);
  return 0;
}

// Places the problem held in the handle's arrival-order arrays into landmark-major slots and
// uploads everything.  with_slack: every landmark gets room for twice its readings (at least 32
// slots) as far as the slot capacity allows, so that later appends find a free slot next to
// their landmark's list.  Chain state refers to readings by their rank inside a landmark's list,
// never by slot, so chains that have already grown survive a new layout.
int lay_out(mcmcgpu_handle h, bool with_slack) {
  Dev& d = h->d;
  const uint32_t T = uint32_t(h->a_cmean.size() / 3), N = uint32_t(h->a_t.size()), F = uint32_t(h->f_count.size());
  h->f_cap = h->f_count;
  if (with_slack) {
    uint64_t want = 0;
    for (uint32_t f = 0; f < F; ++f) want += std::max<uint32_t>(32, 2 * h->f_count[f]);
    if (want <= d.Ncap) {
      for (uint32_t f = 0; f < F; ++f) h->f_cap[f] = std::max<uint32_t>(32, 2 * h->f_count[f]);
    } else {  // share what is left evenly
      const uint32_t extra = F ? (d.Ncap - N) / F : 0;
      for (uint32_t f = 0; f < F; ++f) h->f_cap[f] = h->f_count[f] + extra;
    }
  }
  h->f_start.assign(size_t(F) + 1, 0);
  for (uint32_t f = 0; f < F; ++f) h->f_start[f + 1] = h->f_start[f] + h->f_cap[f];
  const uint32_t S = h->f_start[F];  // slots in use, gaps included
  h->slots_used = S;
  // controls: [t][3] -> [3][Tcap]
  std::vector<double> row(T);
  auto up = [&](const void* dst, const void* src, size_t bytes) {
    return bytes ? hipMemcpyAsync(const_cast<void*>(dst), src, bytes, hipMemcpyHostToDevice, h->stream) : hipSuccess;
  };
  for (int i = 0; i < 3; ++i) {
    for (uint32_t t = 0; t < T; ++t) row[t] = h->a_cmean[3 * size_t(t) + i];
    HIP_TRY(h, up(d.cmu + size_t(i) * d.Tcap, row.data(), size_t(T) * sizeof(double)));
    HIP_TRY(h, hipStreamSynchronize(h->stream));  // `row` is reused
    for (uint32_t t = 0; t < T; ++t) row[t] = h->a_cstd[3 * size_t(t) + i];
    HIP_TRY(h, up(d.csd + size_t(i) * d.Tcap, row.data(), size_t(T) * sizeof(double)));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
  }
  if (T) control_log_det_kernel<<<(T + 255) / 256, 256, 0, h->stream>>>(d, T);
  HIP_TRY(h, hipGetLastError());
  // readings: arrival order -> slots, time-sorted inside a landmark
  std::vector<uint32_t> toff(size_t(T) + 2, 0), fill(h->f_start.begin(), h->f_start.end() - 1), tobs(N), ffirst(F, 0),
      opiv(size_t(S) / 32 + 1, 0);
  h->h_ot.assign(S, 0);
  h->h_omu.assign(S, 0.0);
  std::vector<ObsRec> orec(S, ObsRec{0.0, 0, 0});
  std::vector<double> oisd(S, 1.0), osd(S, 1.0);
  bool uniform_sigma = N > 0;
  for (uint32_t i = 0; i < N; ++i) {
    const uint32_t f = h->a_f[i], t = h->a_t[i];
    const uint32_t pos = fill[f]++;
    if (pos == h->f_start[f]) ffirst[f] = t;
    ++toff[t + 1];
    h->h_ot[pos] = t;
    h->h_omu[pos] = h->a_mean[i];
    osd[pos] = h->a_sd[i];
    oisd[pos] = 1.0 / h->a_sd[i];
    orec[pos] = ObsRec{h->a_mean[i], t, f};
    tobs[i] = pos;  // arrivals are already grouped by timestep, in order
    uniform_sigma = uniform_sigma && h->a_sd[i] == h->a_sd[0];
  }
  for (uint32_t t = 0; t <= T; ++t) toff[t + 1] += toff[t];
  for (size_t i = 0; 32 * i < S; ++i) opiv[i] = h->h_ot[32 * i];
  // per block of 2^shift timesteps and landmark: readings before the block, inside it, and whether
  // those inside are one run of consecutive timesteps.  The last block ends at or before T + 1
  // (readings appended later are all past it, so the counts stay true); row nb = before its end.
  const uint32_t nb = (T + 1) >> d.tblk_shift;
  std::vector<TimeBlock> tblk(size_t(nb + 1) * d.Fcap, TimeBlock{0, 0, 0, 0});
  for (uint32_t f = 0; f < F; ++f) {
    uint32_t seen = 0;
    const uint32_t first = h->f_start[f], count = h->f_count[f];
    for (uint32_t blk = 0; blk <= nb; ++blk) {
      TimeBlock& tb = tblk[size_t(blk) * d.Fcap + f];
      tb.before = seen;
      if (blk == nb) break;
      const uint32_t end = (blk + 1) << d.tblk_shift;
      bool run = true;
      while (seen < count && h->h_ot[first + seen] < end) {
        run = run && h->h_ot[first + seen] == h->h_ot[first + tb.before] + (seen - tb.before);
        ++seen;
      }
      tb.count = seen - tb.before;
      tb.run_from = !run ? kNoRun : (tb.count ? h->h_ot[first + tb.before] : 0);
    }
  }
  HIP_TRY(h, up(d.tblk, tblk.data(), tblk.size() * sizeof(TimeBlock)));
  HIP_TRY(h, up(d.foff, h->f_start.data(), h->f_start.size() * sizeof(uint32_t)));
  HIP_TRY(h, up(d.toff, toff.data(), toff.size() * sizeof(uint32_t)));
  HIP_TRY(h, up(d.orec, orec.data(), size_t(S) * sizeof(ObsRec)));
  std::vector<StreamRec> osr(S);
  for (uint32_t s = 0; s < S; ++s) osr[s] = StreamRec{h->h_omu[s], h->h_ot[s]};
  HIP_TRY(h, up(d.osr, osr.data(), size_t(S) * sizeof(StreamRec)));
  HIP_TRY(h, up(d.oisd, oisd.data(), size_t(S) * sizeof(double)));
  HIP_TRY(h, up(d.opiv, opiv.data(), opiv.size() * sizeof(uint32_t)));
  HIP_TRY(h, up(d.otime, h->h_ot.data(), size_t(S) * sizeof(uint32_t)));
  HIP_TRY(h, up(d.tobs, tobs.data(), size_t(N) * sizeof(uint32_t)));
  HIP_TRY(h, up(d.osd, osd.data(), size_t(S) * sizeof(double)));
  HIP_TRY(h, up(d.ffirst, ffirst.data(), size_t(F) * sizeof(uint32_t)));
  std::vector<double> bmu, bsd, bisd;
  bool uniform_bsigma = false;
  if (h->cfg.range_bearing && h->a_bmean.size() == N && N) {
    bmu.assign(S, 0.0);
    bsd.assign(S, 1.0);
    bisd.assign(S, 1.0);
    uniform_bsigma = true;
    for (uint32_t i = 0; i < N; ++i) {
      const uint32_t pos = tobs[i];
      bmu[pos] = h->a_bmean[i];
      bsd[pos] = h->a_bsd[i];
      bisd[pos] = 1.0 / h->a_bsd[i];
      uniform_bsigma = uniform_bsigma && h->a_bsd[i] == h->a_bsd[0];
    }
    HIP_TRY(h, up(d.obear, bmu.data(), size_t(S) * sizeof(double)));
    HIP_TRY(h, up(d.obsd, bsd.data(), size_t(S) * sizeof(double)));
    HIP_TRY(h, up(d.obisd, bisd.data(), size_t(S) * sizeof(double)));
  }
  HIP_TRY(h, hipStreamSynchronize(h->stream));  // host staging vectors die here
  d.T = T;
  d.N = N;
  d.F = F;
  d.tblk_n = nb;
  h->sigma_all = uniform_sigma ? h->a_sd[0] : 0.0;
  d.usig_inv = uniform_sigma ? 1.0 / h->a_sd[0] : 0.0;
  h->bsigma_all = uniform_bsigma ? h->a_bsd[0] : 0.0;
  d.ubsig_inv = uniform_bsigma ? 1.0 / h->a_bsd[0] : 0.0;
  h->h_tobs = tobs;
  h->problem_loaded = true;
  return 0;
}

// ---- scratch that no call allocates -------------------------------------------------------------
// mcmcgpu_scan_poses_static keeps the segment totals of its scan here instead of in memory a
// handle owns.  Every GPU the library runs on holds its own copy of this array, at its own
// address, until the process ends; in host code the array's NAME is only a stand-in for it, so
// the address is asked for once per device (static_scratch_on) and only that pointer is ever
// handed to a kernel.  All calls on one GPU share the array: calls that overlap there race.
constexpr uint32_t kStaticScanChains = 4096;  // most chains one mcmcgpu_scan_poses_static call takes
constexpr uint32_t kStaticScanMaxPoses = 0xFFFF0000u;  // and most relative poses per chain
// One MiB: the two-pass scan's segment totals ((1024 + 4096) x 24 bytes at most) or the single-pass scan's
// ticket and slots (256 bytes, then 48 per piece of 3072 poses and 48 per window of 64 pieces: 64 chains of
// 1 000 000 poses take 1 020 160 bytes).
constexpr size_t kScanScratchBytes = size_t(1) << 20;
constexpr size_t kStaticScratchBytes = kScanScratchBytes;
static_assert((size_t(kScanTargetGroups) + kStaticScanChains) * sizeof(Pose) <= kScanScratchBytes, "two-pass totals fit");
static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];

constexpr int kMaxDevices = 64;
constexpr uint32_t kSoloMaxChains = 256;  // chains (= CUs) up to which a handle's default workgroup is 256 lanes wide
// advance_kernel's SOLO variant (serial program compiled in, 256 registers: one wavefront per SIMD) is taken
// by launches of at most this many wavefronts, i.e. when every wavefront has a SIMD to itself anyway and the
// serial program is the launch's critical path: 256 chains of 256 lanes, 1024 chains of 64
constexpr uint32_t kSoloMaxWaves = 1024;
// ... and its run-ahead form (SPEC) by those whose data holds at most this many readings
constexpr uint32_t kAheadMaxReadings = 32768;
std::mutex g_static_scratch_mutex;
void* g_static_scratch[kMaxDevices] = {};

// Device address of `scratch` on `device`, which must be the current device.
hipError_t static_scratch_on(int device, void** out) {
  if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
  std::lock_guard<std::mutex> lock(g_static_scratch_mutex);
  if (!g_static_scratch[device]) {
    void* p = nullptr;
    const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess) return e;
    if (!p) return hipErrorInvalidSymbol;
    g_static_scratch[device] = p;
  }
  *out = g_static_scratch[device];
  return hipSuccess;
}

// LDS of a CU and what advance_kernel declares statically (its Shared block, rounded up).
#ifdef MG_DIAG_RUN_RECORDS
constexpr size_t kLdsPerCu = 160 * 1024, kAdvanceStaticLds = 22 * 1024;  // the diagnostic build's extra table
#else
constexpr size_t kLdsPerCu = 160 * 1024, kAdvanceStaticLds = 20 * 1024;
#endif
static_assert(sizeof(Shared) <= kAdvanceStaticLds, "advance_kernel's static LDS outgrew its allowance");

// Copies of the weight trees' tops for one launch of advance_kernel: every node whose index is a
// multiple of 2^shift, with the smallest shifts whose copies fit into the LDS left beside the
// workgroups that share a CU (as many as the chain count and the kernel's 128 registers allow).
// The landmark tree comes first and may take half (most proposals are a landmark's where landmarks are
// many, and a tree held whole is searched without a look at memory).  One chain alone on a CU keeps a
// Plaza-sized tree whole; 1024 chains of 100 000 poses and 1000 landmarks keep the whole landmark tree and
// every 128th node of the state tree.
TreeTops plan_tree_tops(const mcmcgpu_ctx* h, uint32_t state_nodes) {
  const uint32_t waves = h->nthreads / 64;
  const uint32_t by_registers = std::max<uint32_t>(1, 16 / waves);  // 4 wavefronts per SIMD at 128 registers
  const uint32_t per_cu = std::max<uint32_t>(1, std::min<uint32_t>(by_registers, (h->chains + 255) / 256));
  TreeTops t{0, 0, 0, 0};
  // narrow workgroups and many chains: so many workgroups share a CU that the static block is all each gets
  if (kLdsPerCu / per_cu < kAdvanceStaticLds + 512 + 3 * sizeof(double)) return t;
  const size_t budget = (kLdsPerCu / per_cu - kAdvanceStaticLds - 512) / sizeof(double);  // doubles per workgroup
  // a tree of n nodes has the indices 0 .. n - 1: the copy at stride 2^s holds ((n - 1) >> s) + 1 of them
  const auto held = [](uint32_t nodes, uint32_t shift) { return nodes ? size_t((nodes - 1) >> shift) + 1 : size_t(0); };
  const uint32_t F = h->d.F;
  uint32_t fs = 0;
  while (fs < 12 && held(F, fs) > budget / 2) ++fs;
  if (held(F, fs) <= budget / 2) {
    t.fw_shift = fs;
    t.fw_n = uint32_t(held(F, fs));
  }
  const size_t left = budget - t.fw_n;
  uint32_t ss = 0;
  while (ss < 12 && held(state_nodes, ss) > left) ++ss;
  if (held(state_nodes, ss) <= left) {
    t.sw_shift = ss;
    t.sw_n = uint32_t(held(state_nodes, ss));
  }
  return t;
}

// How a rebuild of `chains` pose tables of n relative poses each is cut into workgroups: whole
// chains when there are enough of them to fill the GPU, otherwise segments of at least
// kScanMinSegmentTiles tiles.  A function of (chains, n) only, so that two rebuilds of the same
// data compose in the same order.
struct ScanPlan {
  uint32_t nseg, seg_len;
  bool single_pass;  // few long chains: scan_pieces_kernel (nseg = pieces per chain) instead of two passes
};
ScanPlan scan_plan(uint32_t chains, uint32_t n) {
  const uint32_t tile = uint32_t(kScanLanes) * kScanPerLane;
  const uint32_t want = chains >= kScanTargetGroups ? 1 : (kScanTargetGroups + chains - 1) / std::max<uint32_t>(chains, 1);
  const uint32_t longest = std::max<uint32_t>(1, n / (kScanMinSegmentTiles * tile));
  uint32_t nseg = std::min(want, longest);
  const uint32_t per = (n + nseg - 1) / nseg;
  const uint32_t seg_len = std::max<uint32_t>(tile, (per + tile - 1) / tile * tile);
  nseg = std::max<uint32_t>(1, (n + seg_len - 1) / seg_len);  // no empty segment at the end
  if (nseg > 1) {
    // chains too few to fill the GPU with whole chains: one pass over memory when the pieces' slots fit
    // the scratch (the handle's and the static one are the same size, so both entry points plan alike)
    const uint32_t piece_len = uint32_t(kPieceTiles) * tile;
    const uint64_t pieces = (uint64_t(n) + piece_len - 1) / piece_len;
    if (pieces <= kPieceMaxPerChain && kPieceHeaderBytes + piece_slots_needed(chains, pieces) * sizeof(PieceSlot) <= kScanScratchBytes)
      return ScanPlan{uint32_t(pieces), piece_len, true};
  }
  return ScanPlan{nseg, seg_len, false};
}
// Segment totals a rebuild of `chains` chains can need: chains * nseg < kScanTargetGroups + chains.
size_t scan_total_slots(size_t chains) { return kScanTargetGroups + chains; }

// The scan's launches on `stream`; `tot` must hold chains * plan.nseg poses when plan.nseg > 1.
hipError_t launch_scan(hipStream_t stream, ScanArgs a, uint32_t chains, bool single_pass) {
  if (chains == 0) return hipSuccess;
  if (single_pass) {  // `tot` is the whole scratch: ticket counter and slots, every byte set to 0xFF for this launch
    unsigned char* scratch_bytes = reinterpret_cast<unsigned char*>(a.tot);
    const size_t used = kPieceHeaderBytes + size_t(piece_slots_needed(chains, a.nseg)) * sizeof(PieceSlot);
    if (hipError_t e = hipMemsetAsync(scratch_bytes, 0xFF, used, stream); e != hipSuccess) return e;
    hipLaunchKernelGGL(scan_pieces_kernel, dim3(chains * a.nseg), dim3(kScanLanes), 0, stream, a, scratch_bytes);
    return hipGetLastError();
  }
  const dim3 grid(a.nseg, chains), lanes(kScanLanes);
  if (a.nseg > 1) {
    hipLaunchKernelGGL(scan_chain_kernel<false>, grid, lanes, 0, stream, a);
    hipLaunchKernelGGL(scan_offsets_kernel, dim3(chains), dim3(256), 0, stream, a.nseg, a.tot);
  }
  hipLaunchKernelGGL(scan_chain_kernel<true>, grid, lanes, 0, stream, a);
  return hipGetLastError();
}

// Cumulative poses of chains [first, first + count) from their relative poses.
int rebuild_impl(mcmcgpu_handle h, uint32_t first, uint32_t count) {
  const Dev& d = h->d;
  const uint32_t Tcur = edges_grown(h);
  const ScanPlan plan = scan_plan(count, Tcur);
  ScanArgs a{};
  a.relx = d.relx + size_t(first) * d.Tcap;
  a.rely = d.rely + size_t(first) * d.Tcap;
  a.relth = d.relth + size_t(first) * d.Tcap;
  a.pxy = d.pxy + size_t(first) * (d.Tcap + 1);
  a.pth = d.pth + size_t(first) * (d.Tcap + 1);
  a.rstride = d.Tcap;
  a.pstride = d.Tcap + 1;
  a.n = Tcur;
  a.nseg = plan.nseg;
  a.seg_len = plan.seg_len;
  a.tot = h->scan_tot;
  HIP_TRY(h, launch_scan(h->stream, a, count, plan.single_pass));
  // the landmarks' copies of their parent poses follow the rewritten tables
  parent_poses_kernel<<<dim3((d.Fcap + 255) / 256, count), 256, 0, h->stream>>>(d, first);
  HIP_TRY(h, hipGetLastError());
  return 0;
}

// ---- exchange between handles ---------------------------------------------------------------------
#define RCCL_TRY(h, expr)                                                                          \
  do {                                                                                             \
    ncclResult_t r_ = (expr);                                                                      \
    if (r_ != ncclSuccess)                                                                         \
      return fail(h, MCMCGPU_ERR_HIP, std::string(#expr) + ": " + rccl().GetErrorString(r_));      \
  } while (0)

int need_rccl(mcmcgpu_handle h) {
  if (!rccl().error.empty()) return fail(h, MCMCGPU_ERR_STATE, rccl().error);
  return 0;
}

// Doubles of one packed estimate: accumulated poses [T][3], landmark positions [F][2].
size_t packed_capacity(const mcmcgpu_ctx* h) { return 3 * size_t(h->d.Tcap) + 2 * size_t(h->d.Fcap); }

// Device buffers of the exchange entry points, for `ranks` ranks of `slots` records each.  The only
// allocation outside mcmcgpu_create: made when a communicator is attached, or at the first exchange
// call of a handle without one (which behaves as a communicator of one rank).
int ensure_exchange_buffers(mcmcgpu_handle h, int ranks, uint32_t slots) {
  if (h->gather_recv && h->comm_ranks == ranks && h->comm_slots == slots) return 0;
  for (void* p : {static_cast<void*>(h->gather_send), static_cast<void*>(h->gather_recv)}) {
    if (!p) continue;
    h->allocations.erase(std::find(h->allocations.begin(), h->allocations.end(), p));
    HIP_TRY(h, hipFree(p));
  }
  h->gather_send = h->gather_recv = nullptr;
  h->comm_ranks = ranks;
  h->comm_slots = slots;
  int rc = dev_alloc(h, &h->gather_send, size_t(kMaxOwnReports) * slots);
  if (!rc) rc = dev_alloc(h, &h->gather_recv, size_t(ranks) * kMaxOwnReports * slots);
  if (!rc && hipMemset(h->gather_recv, 0, size_t(ranks) * kMaxOwnReports * slots * sizeof(mcmcgpu_chain_summary)) != hipSuccess)
    rc = fail(h, MCMCGPU_ERR_HIP, "hipMemset of the gather buffer");
  if (!rc && !h->avg_base) rc = dev_alloc(h, &h->avg_base, packed_capacity(h));
  if (!rc && !h->avg_sum) rc = dev_alloc(h, &h->avg_sum, packed_capacity(h));
  if (!rc && !h->bcast) rc = dev_alloc(h, &h->bcast, packed_capacity(h));
  return rc;
}
int default_exchange_buffers(mcmcgpu_handle h) {  // a handle without a communicator: one rank, all its chains
  if (h->gather_recv) return 0;
  return ensure_exchange_buffers(h, 1, h->cfg.max_chains);
}

// `intervals` rows of this handle's records (src: [intervals][chains]) to every rank; afterwards
// gather_recv holds [ranks][intervals][slots].
int gather_records(mcmcgpu_handle h, const mcmcgpu_chain_summary* src, uint32_t intervals) {
  if (int rc = default_exchange_buffers(h)) return rc;
  if (h->chains > h->comm_slots) return fail(h, MCMCGPU_ERR_CAPACITY, "more chains than the communicator's slots per rank");
  const uint32_t n = intervals * h->comm_slots;
  mcmcgpu_chain_summary* send = h->comm ? h->gather_send : h->gather_recv;  // alone: packed straight into place
  if (n) {
    hipLaunchKernelGGL(pack_records_kernel, dim3((n + 255) / 256), dim3(256), 0, h->stream, src, h->chains, intervals, send,
                       h->comm_slots);
    HIP_TRY(h, hipGetLastError());
    if (h->comm)
      RCCL_TRY(h, rccl().AllGather(send, h->gather_recv, size_t(n) * sizeof(mcmcgpu_chain_summary), ncclChar, h->comm,
                                   h->stream));
  }
  h->gathered_intervals = intervals;
  return 0;
}

}  // namespace

extern "C" {

const char* mcmcgpu_last_error(mcmcgpu_handle h) { return h ? h->error.c_str() : g_create_error.c_str(); }

int mcmcgpu_create(const mcmcgpu_config* cfg, mcmcgpu_handle* out) {
  if (!cfg || !out) return fail(nullptr, MCMCGPU_ERR_ARGUMENT, "null argument");
  if (cfg->max_chains == 0) return fail(nullptr, MCMCGPU_ERR_ARGUMENT, "max_chains must be positive");
  // default width: 256 lanes.  Many chains on sparse data (at most 32 768 readings in all, so a proposal
  // touches a few dozen to a few thousand) take ONE wavefront per chain: such an update is its serial part,
  // four chains per CU then run it on a SIMD each and no barrier joins wavefronts (commit-heavy benchmark,
  // 1024 chains x 10 000 poses, 52 readings per proposal: 5.3e9 units/s against 3.2e9 with 128 or 256 lanes).
  // Few chains keep 256 lanes: their stream is spread over the CU they have to themselves (Plaza1, one
  // chain: 28 200 cycles per update with 256 lanes, 29 500 with 128).
  const uint32_t nthr = cfg->threads_per_chain
                            ? cfg->threads_per_chain
                            : (cfg->max_observations <= 32768 && cfg->max_chains > kSoloMaxChains ? 64 : 256);
  if (nthr < 64 || nthr > 1024 || !is_pow2(nthr))
    return fail(nullptr, MCMCGPU_ERR_ARGUMENT, "threads_per_chain must be a power of two in [64, 1024]");
  if (cfg->serial_program > 3) return fail(nullptr, MCMCGPU_ERR_ARGUMENT, "serial_program must be 0, 1, 2 or 3");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(nullptr, MCMCGPU_ERR_HIP, "no HIP device available (this library has no CPU path)");
  if (cfg->device < 0 || cfg->device >= ndev) return fail(nullptr, MCMCGPU_ERR_ARGUMENT, "device ordinal out of range");
  // slots for readings: twice the readings plus a first allotment per landmark, so that appended
  // readings find room next to their landmark's list (see lay_out)
  const uint64_t slots = 2 * uint64_t(std::max<uint32_t>(cfg->max_observations, 1)) +
                         32 * uint64_t(std::max<uint32_t>(cfg->max_features, 1));
  if (slots > 0xFFFFFFFFull) return fail(nullptr, MCMCGPU_ERR_CAPACITY, "max_observations too large");
  hipError_t e = hipSetDevice(cfg->device);
  if (e != hipSuccess) return fail(nullptr, MCMCGPU_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(e));
  {  // the library's set-up call on this device: where mcmcgpu_scan_poses_static's scratch lives here
    void* unused = nullptr;
    e = static_scratch_on(cfg->device, &unused);
    if (e != hipSuccess)
      return fail(nullptr, MCMCGPU_ERR_HIP, std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(e));
  }

  mcmcgpu_handle h = new mcmcgpu_ctx();
  h->cfg = *cfg;
  h->nthreads = nthr;
  if (cfg->stream) {
    h->stream = static_cast<hipStream_t>(cfg->stream);
  } else {
    e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
      delete h;
      return fail(nullptr, MCMCGPU_ERR_HIP, std::string("hipStreamCreate: ") + hipGetErrorString(e));
    }
    h->own_stream = true;
  }
  Dev& d = h->d;
  d.Tcap = std::max<uint32_t>(cfg->max_timesteps, 1);
  d.Fcap = std::max<uint32_t>(cfg->max_features, 1);
  d.Ncap = uint32_t(slots);
  d.C = 0;
  d.uniform_first = cfg->feature_uniform_first ? 1 : 0;
  d.trace_cap = cfg->trace_capacity;
  const size_t C = cfg->max_chains;
  int rc = 0;
#define ALLOC(ptr, count)                                                         \
  do {                                                                            \
    if (!rc) rc = dev_alloc(h, const_cast<std::remove_const_t<std::remove_pointer_t<decltype(ptr)>>**>(&(ptr)), count); \
  } while (0)
  ALLOC(d.cmu, 3 * size_t(d.Tcap));
  ALLOC(d.csd, 3 * size_t(d.Tcap));
  ALLOC(d.cld, d.Tcap);
  ALLOC(d.foff, size_t(d.Fcap) + 1);
  ALLOC(d.orec, d.Ncap);
  ALLOC(d.osr, d.Ncap);
  ALLOC(d.oisd, d.Ncap);
  ALLOC(d.opiv, size_t(d.Ncap) / 32 + 2);
  ALLOC(d.otime, d.Ncap);
  d.tblk_shift = kTimeBlockShift;
  d.tblk_rows = (d.Tcap >> kTimeBlockShift) + 3;
  d.tblk_n = 0;
  ALLOC(d.tblk, size_t(d.Fcap) * d.tblk_rows);
  ALLOC(d.osd, d.Ncap);
  if (cfg->range_bearing) {
    ALLOC(d.obear, d.Ncap);
    ALLOC(d.obsd, d.Ncap);
    ALLOC(d.obisd, d.Ncap);
  }
  ALLOC(d.toff, size_t(d.Tcap) + 2);
  ALLOC(d.tobs, d.Ncap);
  ALLOC(d.ffirst, d.Fcap);
  ALLOC(d.init_u, d.Fcap);
  ALLOC(d.init_rel, 3 * size_t(d.Tcap));
  ALLOC(d.init_feat, 2 * size_t(d.Fcap));
  ALLOC(d.relx, C * d.Tcap);
  ALLOC(d.rely, C * d.Tcap);
  ALLOC(d.relth, C * d.Tcap);
  ALLOC(d.clp, C * d.Tcap);
  ALLOC(d.pxy, C * (size_t(d.Tcap) + 1));
  ALLOC(d.pth, C * (size_t(d.Tcap) + 1));
  ALLOC(d.sw, C * d.Tcap);
  ALLOC(d.fw, C * d.Fcap);
  ALLOC(d.fex, C * d.Fcap);
  ALLOC(d.fey, C * d.Fcap);
  ALLOC(d.fpar, C * d.Fcap);
  ALLOC(d.fcnt, C * d.Fcap);
  ALLOC(d.fpt, C * d.Fcap);
  ALLOC(d.fpx, C * d.Fcap);
  ALLOC(d.fpy, C * d.Fcap);
  ALLOC(d.fpth, C * d.Fcap);
  ALLOC(d.fpc, C * d.Fcap);
  ALLOC(d.fps, C * d.Fcap);
  ALLOC(d.fpmu, C * d.Fcap);
  ALLOC(d.fpsd, C * d.Fcap);
  ALLOC(d.fplsd, C * d.Fcap);
  if (cfg->range_bearing) {
    ALLOC(d.fpbmu, C * d.Fcap);
    ALLOC(d.fpbsd, C * d.Fcap);
    ALLOC(d.fplbsd, C * d.Fcap);
  }
  ALLOC(d.flast, C * d.Fcap);
  ALLOC(d.nf, C);
  ALLOC(d.ll, C);
  ALLOC(d.mt, C * 624);
  ALLOC(d.mti, C);
  ALLOC(d.nflag, C);
  ALLOC(d.nsaved, C);
  ALLOC(d.n_updates, C);
  ALLOC(d.n_accepts, C);
  ALLOC(d.n_units, C);
  ALLOC(d.n_state_props, C);
  ALLOC(d.n_moved, C);
  ALLOC(d.errflag, C);
  ALLOC(d.trace_n, C);
  ALLOC(d.beta, C);
  ALLOC(d.llc, C);
  ALLOC(d.slot_chain, C);
  ALLOC(h->exch_u, C / 2 + 1);
  ALLOC(h->exch_counts, 2);
  if (d.trace_cap) ALLOC(d.trace, C * d.trace_cap);
  ALLOC(h->scan_tot, std::max(scan_total_slots(C), kScanScratchBytes / sizeof(Pose) + 1));
  ALLOC(h->seed_stage, C);
  ALLOC(h->reports, size_t(kMaxOwnReports + 1) * C);
  d.max_slices = cfg->max_split_slices;
  if (d.max_slices) {
    ALLOC(d.ctl, C);
    ALLOC(d.sp_seglo, C * d.Fcap);
    ALLOC(d.sp_segpre, C * (size_t(d.Fcap) + 1));
    ALLOC(d.sp_segw, 4 * C * d.Fcap);
    ALLOC(d.sp_partial, C * d.max_slices);
    ALLOC(d.sp_slice_seg, C * (size_t(d.max_slices) + 1));
  }
  h->snap_doubles = 3 * size_t(d.Tcap) + 2 * size_t(d.Fcap) + 2;
  ALLOC(h->snap_dev, h->snap_doubles);
  if (!rc) {
    hipError_t he = hipHostMalloc(reinterpret_cast<void**>(&h->snap_host), h->snap_doubles * sizeof(double), 0);
    if (he == hipSuccess) he = hipEventCreateWithFlags(&h->snap_event, hipEventDisableTiming);
    if (he != hipSuccess) rc = fail(h, MCMCGPU_ERR_HIP, std::string("snapshot buffers: ") + hipGetErrorString(he));
  }
  h->scratch_doubles = 6 * (size_t(d.Tcap) + 1) + 4 * size_t(d.Fcap) + 16;
  ALLOC(h->scratch, h->scratch_doubles);
#undef ALLOC
  if (rc) {
    g_create_error = h->error;
    mcmcgpu_destroy(h);
    return rc;
  }
  d.n_init_u = 0;
  *out = h;
  return 0;
}

void mcmcgpu_destroy(mcmcgpu_handle h) {
  if (!h) return;
  (void)hipSetDevice(h->cfg.device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  for (auto& ev : h->timings) {
    (void)hipEventDestroy(ev.start);
    (void)hipEventDestroy(ev.stop);
  }
  for (auto& ev : h->free_events) {
    (void)hipEventDestroy(ev.start);
    (void)hipEventDestroy(ev.stop);
  }
  if (h->comm && rccl().CommDestroy) (void)rccl().CommDestroy(h->comm);
  if (h->snap_host) (void)hipHostFree(h->snap_host);
  if (h->snap_event) (void)hipEventDestroy(h->snap_event);
  for (void* p : h->allocations) (void)hipFree(p);
  if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

int mcmcgpu_upload_problem(mcmcgpu_handle h, uint32_t T, const double* ctrl_mean, const double* ctrl_stddev,
                           uint32_t N, const uint32_t* obs_timestep, const uint32_t* obs_feature,
                           const double* obs_mean, const double* obs_stddev, uint32_t F) {
  if (!h) return MCMCGPU_ERR_ARGUMENT;
  if (T > h->cfg.max_timesteps || N > h->cfg.max_observations || F > h->cfg.max_features)
    return fail(h, MCMCGPU_ERR_CAPACITY, "problem exceeds the capacities given to mcmcgpu_create");
  if ((T && (!ctrl_mean || !ctrl_stddev)) || (N && (!obs_timestep || !obs_feature || !obs_mean || !obs_stddev)))
    return fail(h, MCMCGPU_ERR_ARGUMENT, "null problem array");
  if (T + 1 < h->next_timestep) return fail(h, MCMCGPU_ERR_STATE, "problem shorter than what the chains have processed");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  // arrival order: non-decreasing timesteps, landmarks numbered by first appearance, at most one
  // reading per landmark and timestep
  uint32_t seen = 0;
  std::vector<uint32_t> last_t(F, 0);
  for (uint32_t i = 0; i < N; ++i) {
    const uint32_t f = obs_feature[i], t = obs_timestep[i];
    if (t > T) return fail(h, MCMCGPU_ERR_ARGUMENT, "reading " + std::to_string(i) + " has a timestep beyond T");
    if (i && t < obs_timestep[i - 1]) return fail(h, MCMCGPU_ERR_ARGUMENT, "readings are not in arrival order");
    if (f > seen || f >= F)
      return fail(h, MCMCGPU_ERR_ARGUMENT, "landmark indices must be dense and numbered by first appearance");
    if (f == seen)
      ++seen;
    else if (last_t[f] >= t)
      return fail(h, MCMCGPU_ERR_ARGUMENT, "a landmark has two readings at one timestep");
    last_t[f] = t;
    if (!(obs_stddev[i] > 0)) return fail(h, MCMCGPU_ERR_ARGUMENT, "non-positive reading stddev");
  }
  if (seen != F) return fail(h, MCMCGPU_ERR_ARGUMENT, "F does not match the landmarks that appear in the readings");
  h->a_cmean.assign(ctrl_mean, ctrl_mean + 3 * size_t(T));
  h->a_cstd.assign(ctrl_stddev, ctrl_stddev + 3 * size_t(T));
  h->a_t.assign(obs_timestep, obs_timestep + N);
  h->a_f.assign(obs_feature, obs_feature + N);
  h->a_mean.assign(obs_mean, obs_mean + N);
  h->a_sd.assign(obs_stddev, obs_stddev + N);
  h->a_bmean.clear();
  h->a_bsd.clear();
  h->bearings_loaded = false;
  h->f_count.assign(F, 0);
  h->f_arrivals.assign(F, {});
  for (uint32_t i = 0; i < N; ++i) {
    ++h->f_count[obs_feature[i]];
    h->f_arrivals[obs_feature[i]].push_back(i);
  }
  return lay_out(h, false);
}

int mcmcgpu_append_timestep(mcmcgpu_handle h, const double* ctrl_mean, const double* ctrl_stddev, uint32_t n_obs,
                            const uint32_t* obs_feature, const double* obs_mean, const double* obs_stddev,
                            const double* bearing_mean, const double* bearing_stddev, const double* init_state_rel,
                            const double* init_feat_rel) {
  if (!h) return MCMCGPU_ERR_ARGUMENT;
  if ((ctrl_mean == nullptr) != (ctrl_stddev == nullptr)) return fail(h, MCMCGPU_ERR_ARGUMENT, "control mean and stddev go together");
  if (n_obs && (!obs_feature || !obs_mean || !obs_stddev)) return fail(h, MCMCGPU_ERR_ARGUMENT, "null reading array");
  const bool rb = h->cfg.range_bearing != 0;
  if (rb && n_obs && (!bearing_mean || !bearing_stddev))
    return fail(h, MCMCGPU_ERR_ARGUMENT, "a range-and-bearing handle needs the bearings of the appended readings");
  Dev& d = h->d;
  const uint32_t T0 = uint32_t(h->a_cmean.size() / 3), N0 = uint32_t(h->a_t.size()), F0 = uint32_t(h->f_count.size());
  if (!ctrl_mean && (T0 != 0 || N0 != 0))
    return fail(h, MCMCGPU_ERR_STATE, "only timestep 0 of an empty problem comes without a control");
  if (rb && N0 && !h->bearings_loaded)
    return fail(h, MCMCGPU_ERR_STATE, "mcmcgpu_set_bearings must follow mcmcgpu_upload_problem before readings are appended");
  const uint32_t t = ctrl_mean ? T0 + 1 : 0;  // timestep of the appended readings
  if (t > h->cfg.max_timesteps || uint64_t(N0) + n_obs > h->cfg.max_observations)
    return fail(h, MCMCGPU_ERR_CAPACITY, "appended data exceeds the capacities given to mcmcgpu_create");
  // the same checks mcmcgpu_upload_problem makes, for the new readings only
  uint32_t F1 = F0;
  for (uint32_t i = 0; i < n_obs; ++i) {
    const uint32_t f = obs_feature[i];
    if (f > F1) return fail(h, MCMCGPU_ERR_ARGUMENT, "landmark indices must be dense and numbered by first appearance");
    if (f == F1) {
      if (F1 + 1 > h->cfg.max_features) return fail(h, MCMCGPU_ERR_CAPACITY, "more landmarks than max_features");
      ++F1;
    } else if (f < F0) {
      if (h->h_ot[h->f_start[f] + h->f_count[f] - 1] >= t)
        return fail(h, MCMCGPU_ERR_ARGUMENT, "a landmark has two readings at one timestep");
    }
    for (uint32_t k = 0; k < i; ++k)
      if (obs_feature[k] == f) return fail(h, MCMCGPU_ERR_ARGUMENT, "a landmark has two readings at one timestep");
    if (!(obs_stddev[i] > 0) || (rb && !(bearing_stddev[i] > 0))) return fail(h, MCMCGPU_ERR_ARGUMENT, "non-positive reading stddev");
  }
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  // initial estimates continue the arrays of mcmcgpu_set_initial_estimates only without a hole
  const bool take_irel = init_state_rel && ctrl_mean && d.n_init_rel == T0;
  const bool take_ifeat = init_feat_rel && d.n_init_feat == F0;
  // A landmark whose slots are used up moves, alone, to a region of twice the size behind everything
  // else (the old region is left as a gap until the next full layout): every reading is moved
  // O(1) times on average, and nothing else is touched.
  for (uint32_t i = 0; i < n_obs && h->problem_loaded; ++i) {
    const uint32_t f = obs_feature[i];
    if (f >= F0 || h->f_count[f] < h->f_cap[f]) continue;
    const uint32_t count = h->f_count[f], from = h->f_start[f], to = h->slots_used;
    const uint64_t room = std::max<uint64_t>(32, 2 * uint64_t(h->f_cap[f]));
    if (uint64_t(to) + room + 32 * uint64_t(F1 - F0) > d.Ncap) break;  // no room left: the full layout below compacts
    hipLaunchKernelGGL(move_list_kernel, dim3(std::max<uint32_t>(1, std::min<uint32_t>(64, (count + 255) / 256))), dim3(256), 0,
                       h->stream, d, f, from, to, count);
    HIP_TRY(h, hipGetLastError());
    h->slots_used = to + uint32_t(room);
    h->h_ot.resize(h->slots_used, 0);
    h->h_omu.resize(h->slots_used, 0.0);
    std::copy_n(h->h_ot.begin() + from, count, h->h_ot.begin() + to);
    std::copy_n(h->h_omu.begin() + from, count, h->h_omu.begin() + to);
    for (const uint32_t arrival : h->f_arrivals[f]) h->h_tobs[arrival] = h->h_tobs[arrival] - from + to;
    h->f_start[f] = to;
    h->f_cap[f] = uint32_t(room);
    h->f_start[F0] = h->slots_used;
  }
  // does every reading find a free slot next to its landmark's list?
  bool fits = h->problem_loaded || (T0 == 0 && N0 == 0);
  std::vector<uint32_t> new_start(F1 - F0, 0);
  uint32_t end = h->slots_used;
  for (uint32_t i = 0; i < n_obs && fits; ++i) {
    const uint32_t f = obs_feature[i];
    if (f < F0) {
      fits = h->f_count[f] < h->f_cap[f];
    } else {  // a new landmark takes 32 slots at the end
      fits = uint64_t(end) + 32 <= d.Ncap;
      new_start[f - F0] = end;
      end += 32;
    }
  }
  // the arrival-order copy grows either way
  if (ctrl_mean) {
    h->a_cmean.insert(h->a_cmean.end(), ctrl_mean, ctrl_mean + 3);
    h->a_cstd.insert(h->a_cstd.end(), ctrl_stddev, ctrl_stddev + 3);
  }
  h->f_arrivals.resize(F1);
  for (uint32_t i = 0; i < n_obs; ++i) {
    h->f_arrivals[obs_feature[i]].push_back(uint32_t(h->a_t.size()));
    h->a_t.push_back(t);
    h->a_f.push_back(obs_feature[i]);
    h->a_mean.push_back(obs_mean[i]);
    h->a_sd.push_back(obs_stddev[i]);
    if (rb) {
      h->a_bmean.push_back(bearing_mean[i]);
      h->a_bsd.push_back(bearing_stddev[i]);
    }
  }
  if (rb) h->bearings_loaded = true;
  auto take_estimates = [&]() {  // into the arrays of mcmcgpu_set_initial_estimates
    if (take_irel) {
      HIP_TRY(h, hipMemcpyAsync(const_cast<double*>(d.init_rel) + 3 * size_t(T0), init_state_rel, 3 * sizeof(double),
                                hipMemcpyHostToDevice, h->stream));
      d.n_init_rel = T0 + 1;
    }
    if (take_ifeat && F1 > F0) {
      HIP_TRY(h, hipMemcpyAsync(const_cast<double*>(d.init_feat) + 2 * size_t(F0), init_feat_rel,
                                2 * size_t(F1 - F0) * sizeof(double), hipMemcpyHostToDevice, h->stream));
      d.n_init_feat = F1;
    }
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return 0;
  };
  if (!fits) {  // a landmark's slots are used up (or nothing is laid out yet): new layout, with room to grow
    h->f_count.resize(F1, 0);
    for (uint32_t i = 0; i < n_obs; ++i) ++h->f_count[obs_feature[i]];
    if (int rc = lay_out(h, true)) return rc;
    return take_estimates();
  }
  // in place: one small kernel per kAppendMax readings carries the data as its arguments
  h->f_count.resize(F1, 0);
  h->f_cap.resize(F1, 32);
  h->f_start.resize(size_t(F1) + 1);
  for (uint32_t f = F0; f < F1; ++f) h->f_start[f] = new_start[f - F0];
  h->f_start[F1] = end;
  h->slots_used = end;
  h->h_ot.resize(end, 0);
  h->h_omu.resize(end, 0.0);
  uint32_t done = 0, new_seen = 0;
  do {
    AppendArgs a{};
    a.t = t;
    a.has_control = ctrl_mean && done == 0;
    if (a.has_control)
      for (int k = 0; k < 3; ++k) {
        a.cm[k] = ctrl_mean[k];
        a.cs[k] = ctrl_stddev[k];
      }
    a.has_init_rel = a.has_control && take_irel;
    if (a.has_init_rel)
      for (int k = 0; k < 3; ++k) a.irel[k] = init_state_rel[k];
    a.n = std::min<uint32_t>(kAppendMax, n_obs - done);
    a.arrival0 = N0 + done;
    a.arrivals_end = N0 + done + a.n;
    for (uint32_t i = 0; i < a.n; ++i) {
      const uint32_t j = done + i, f = obs_feature[j];
      const uint32_t slot = h->f_start[f] + h->f_count[f];
      a.slot[i] = slot;
      a.f[i] = f;
      a.is_new[i] = f >= F0;
      a.mu[i] = obs_mean[j];
      a.sd[i] = obs_stddev[j];
      a.bmu[i] = rb ? bearing_mean[j] : 0.0;
      a.bsd[i] = rb ? bearing_stddev[j] : 1.0;
      if (a.is_new[i]) {
        a.has_ifeat[i] = take_ifeat;
        if (take_ifeat) {
          a.ifeat[i][0] = init_feat_rel[2 * size_t(new_seen)];
          a.ifeat[i][1] = init_feat_rel[2 * size_t(new_seen) + 1];
        }
        ++new_seen;
      }
      ++h->f_count[f];
      h->h_ot[slot] = t;
      h->h_omu[slot] = obs_mean[j];
      h->h_tobs.push_back(slot);
      if (N0 + j == 0)
        h->sigma_all = obs_stddev[j], h->bsigma_all = rb ? bearing_stddev[j] : 0.0;
      if (obs_stddev[j] != h->sigma_all) h->sigma_all = 0.0;
      if (rb && bearing_stddev[j] != h->bsigma_all) h->bsigma_all = 0.0;
    }
    hipLaunchKernelGGL(append_kernel, dim3(1), dim3(64), 0, h->stream, d, a);
    HIP_TRY(h, hipGetLastError());
    done += a.n;
  } while (done < n_obs);
  if (take_irel) d.n_init_rel = T0 + 1;
  if (take_ifeat && F1 > F0) d.n_init_feat = F1;
  d.T = uint32_t(h->a_cmean.size() / 3);
  d.N = uint32_t(h->a_t.size());
  d.F = F1;
  d.usig_inv = h->sigma_all > 0 ? 1.0 / h->sigma_all : 0.0;
  d.ubsig_inv = h->bsigma_all > 0 ? 1.0 / h->bsigma_all : 0.0;
  h->problem_loaded = true;
  return 0;
}

int mcmcgpu_set_bearings(mcmcgpu_handle h, uint32_t N, const double* bearing_mean, const double* bearing_stddev) {
  if (!h || !bearing_mean || !bearing_stddev) return fail(h, MCMCGPU_ERR_ARGUMENT, "null argument");
  if (!h->cfg.range_bearing) return fail(h, MCMCGPU_ERR_STATE, "the handle was created for range-only readings");
  if (!h->problem_loaded || N != h->d.N)
    return fail(h, MCMCGPU_ERR_ARGUMENT, "one bearing per reading of the uploaded problem is required");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  for (uint32_t i = 0; i < N; ++i)
    if (!(bearing_stddev[i] > 0)) return fail(h, MCMCGPU_ERR_ARGUMENT, "non-positive bearing stddev");
  h->a_bmean.assign(bearing_mean, bearing_mean + N);
  h->a_bsd.assign(bearing_stddev, bearing_stddev + N);
  std::vector<double> mu(h->slots_used, 0.0), sd(h->slots_used, 1.0), isd(h->slots_used, 1.0);
  bool uniform_sigma = N > 0;
  for (uint32_t i = 0; i < N; ++i) {
    uniform_sigma = uniform_sigma && bearing_stddev[i] == bearing_stddev[0];
    const uint32_t pos = h->h_tobs[i];
    mu[pos] = bearing_mean[i];
    sd[pos] = bearing_stddev[i];
    isd[pos] = 1.0 / bearing_stddev[i];
  }
  const size_t bytes = size_t(h->slots_used) * sizeof(double);
  if (bytes) {
    HIP_TRY(h, hipMemcpyAsync(const_cast<double*>(h->d.obear), mu.data(), bytes, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(const_cast<double*>(h->d.obsd), sd.data(), bytes, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(const_cast<double*>(h->d.obisd), isd.data(), bytes, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
  }
  h->bsigma_all = uniform_sigma ? bearing_stddev[0] : 0.0;
  h->d.ubsig_inv = uniform_sigma ? 1.0 / bearing_stddev[0] : 0.0;
  h->bearings_loaded = true;
  return 0;
}

int mcmcgpu_init_chains(mcmcgpu_handle h, uint32_t C, const uint32_t* seeds, const double* init_uniforms,
                        uint32_t n_init_uniforms) {
  if (!h || !seeds) return fail(h, MCMCGPU_ERR_ARGUMENT, "null argument");
  if (C == 0 || C > h->cfg.max_chains) return fail(h, MCMCGPU_ERR_CAPACITY, "chain count out of range");
  if (init_uniforms && n_init_uniforms > h->d.Fcap) return fail(h, MCMCGPU_ERR_CAPACITY, "too many initialiser draws");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  Dev& d = h->d;
  d.C = C;
  h->chains = C;
  h->next_timestep = 0;
  h->ladder = 0;
  uint32_t* dseeds = h->seed_stage;
  HIP_TRY(h, hipMemcpyAsync(dseeds, seeds, size_t(C) * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream));
  d.n_init_u = 0;
  d.n_init_rel = 0;
  d.n_init_feat = 0;
  if (init_uniforms && n_init_uniforms) {
    HIP_TRY(h, hipMemcpyAsync(const_cast<double*>(d.init_u), init_uniforms, size_t(n_init_uniforms) * sizeof(double),
                              hipMemcpyHostToDevice, h->stream));
    d.n_init_u = n_init_uniforms;
  }
  hipLaunchKernelGGL(init_chains_kernel, dim3((C + 255) / 256), dim3(256), 0, h->stream, d, dseeds);
  HIP_TRY(h, hipGetLastError());
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return 0;
}

int mcmcgpu_set_initial_estimates(mcmcgpu_handle h, uint32_t n_state, const double* state_rel, uint32_t n_feat,
                                   const double* feat_rel) {
  if (!h) return MCMCGPU_ERR_ARGUMENT;
  if ((n_state && !state_rel) || (n_feat && !feat_rel)) return fail(h, MCMCGPU_ERR_ARGUMENT, "null argument");
  if (n_state > h->d.Tcap || n_feat > h->d.Fcap) return fail(h, MCMCGPU_ERR_CAPACITY, "more estimates than capacity");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  if (n_state)
    HIP_TRY(h, hipMemcpyAsync(const_cast<double*>(h->d.init_rel), state_rel, sizeof(double) * 3 * n_state,
                              hipMemcpyHostToDevice, h->stream));
  if (n_feat)
    HIP_TRY(h, hipMemcpyAsync(const_cast<double*>(h->d.init_feat), feat_rel, sizeof(double) * 2 * n_feat,
                              hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  h->d.n_init_rel = n_state;
  h->d.n_init_feat = n_feat;
  return 0;
}

int mcmcgpu_advance(mcmcgpu_handle h, uint32_t upto, uint32_t steps_per_timestep, uint32_t end_steps) {
  if (!h) return MCMCGPU_ERR_ARGUMENT;
  if (!h->problem_loaded || h->chains == 0) return fail(h, MCMCGPU_ERR_STATE, "upload a problem and init chains first");
  if (upto > h->d.T) return fail(h, MCMCGPU_ERR_ARGUMENT, "timestep beyond the uploaded problem");
  if (h->cfg.range_bearing && !h->bearings_loaded)
    return fail(h, MCMCGPU_ERR_STATE, "mcmcgpu_set_bearings must follow mcmcgpu_upload_problem on a range-and-bearing handle");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  const bool grow = upto + 1 > h->next_timestep;
  if (!grow && end_steps == 0) return 0;
  // LDS copies of the tops of the weight trees: as many nodes as fit beside the workgroups that share a CU
  // state edges the chains have by the end of this launch: tree nodes 0 .. upto - 1
  const TreeTops tops = plan_tree_tops(h, std::max(upto, edges_grown(h)));
  const size_t dyn_lds = (size_t(tops.sw_n) + tops.fw_n) * sizeof(double);
  // one kernel per observation model; few chains take the variant with the serial program compiled in
  const bool solo = h->nthreads <= 256 &&  // that variant is built for at most 256 lanes
                    (h->cfg.serial_program >= 2 ||
                     (h->cfg.serial_program == 0 && uint64_t(h->chains) * (h->nthreads / 64) <= kSoloMaxWaves));
  // ... and of those, launches on sparse data (an update's serial part outlasts its evaluation) let the first
  // wavefront run ahead with the next update's selection while the others evaluate
  const bool ahead = solo && h->nthreads >= 128 &&
                     (h->cfg.serial_program == 3 || (h->cfg.serial_program == 0 && h->d.N <= kAheadMaxReadings));
  auto* const kernel =
      h->cfg.range_bearing
          ? (ahead ? advance_kernel<true, true, true> : solo ? advance_kernel<true, true, false> : advance_kernel<true, false, false>)
          : (ahead ? advance_kernel<false, true, true> : solo ? advance_kernel<false, true, false> : advance_kernel<false, false, false>);
  if (dyn_lds > 48 * 1024 && h->advance_lds_raised.insert(reinterpret_cast<const void*>(kernel)).second)
    HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   int(kLdsPerCu - kAdvanceStaticLds)));
  EventPair ev;
  if (int rc = begin_timed(h, &ev)) return rc;
  hipLaunchKernelGGL(kernel, dim3(h->chains), dim3(h->nthreads), dyn_lds, h->stream, h->d, h->next_timestep,
                     edges_grown(h), upto, steps_per_timestep, end_steps, grow ? 1u : 0u, tops);
  if (int rc = end_timed(h, ev, hipGetLastError(), "advance_kernel")) return rc;
  if (grow) h->next_timestep = upto + 1;
  return 0;
}

int mcmcgpu_grow(mcmcgpu_handle h, uint32_t upto) { return mcmcgpu_advance(h, upto, 0, 0); }

int mcmcgpu_run(mcmcgpu_handle h, uint32_t steps) {
  if (!h) return MCMCGPU_ERR_ARGUMENT;
  if (h->next_timestep == 0) return fail(h, MCMCGPU_ERR_STATE, "chains have not processed timestep 0 yet");
  return mcmcgpu_advance(h, h->next_timestep - 1, 0, steps);
}

int mcmcgpu_run_reported(mcmcgpu_handle h, uint32_t steps, uint32_t report_every, void* dev_out) {
  if (!h) return MCMCGPU_ERR_ARGUMENT;
  if (report_every == 0) return fail(h, MCMCGPU_ERR_ARGUMENT, "the report interval must be positive");
  const uint32_t intervals = (steps + report_every - 1) / report_every;
  if (!dev_out) {  // the library's own buffer, read back with mcmcgpu_get_reports
    if (intervals > kMaxOwnReports)
      return fail(h, MCMCGPU_ERR_CAPACITY, "more report intervals than the library's buffer holds (" +
                                               std::to_string(kMaxOwnReports) + "); pass a buffer or split the run");
    dev_out = h->reports;
    h->report_intervals = intervals;
  }
  h->d.snap_out = static_cast<mcmcgpu_chain_summary*>(dev_out);
  h->d.snap_every = report_every;
  const int rc = mcmcgpu_run(h, steps);
  h->d.snap_out = nullptr;  // the launch took its copy of the descriptor
  h->d.snap_every = 0;
  return rc;
}

int mcmcgpu_set_ladders(mcmcgpu_handle h, uint32_t replicas, const double* betas, uint32_t exchange_seed) {
  if (!h || !betas) return fail(h, MCMCGPU_ERR_ARGUMENT, "null argument");
  if (h->chains == 0) return fail(h, MCMCGPU_ERR_STATE, "no chains initialised");
  if (replicas == 0 || h->chains % replicas != 0)
    return fail(h, MCMCGPU_ERR_ARGUMENT, "the chain count must be a multiple of the replicas per ladder");
  for (uint32_t s = 0; s < replicas; ++s)
    if (!(betas[s] > 0)) return fail(h, MCMCGPU_ERR_ARGUMENT, "inverse temperatures must be positive");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  std::vector<double> beta(h->chains);
  std::vector<uint32_t> slot(h->chains);
  for (uint32_t c = 0; c < h->chains; ++c) {
    beta[c] = betas[c % replicas];
    slot[c] = c;
  }
  HIP_TRY(h, hipMemcpyAsync(h->d.beta, beta.data(), beta.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipMemcpyAsync(h->d.slot_chain, slot.data(), slot.size() * sizeof(uint32_t), hipMemcpyHostToDevice,
                            h->stream));
  HIP_TRY(h, hipMemsetAsync(h->exch_counts, 0, 2 * sizeof(unsigned long long), h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  h->ladder = replicas;
  rng_seed(h->exchange_rng, exchange_seed);
  return 0;
}

int mcmcgpu_exchange(mcmcgpu_handle h, uint32_t phase) {
  if (!h) return MCMCGPU_ERR_ARGUMENT;
  if (h->ladder == 0) return fail(h, MCMCGPU_ERR_STATE, "mcmcgpu_set_ladders has not been called");
  if (phase > 1) return fail(h, MCMCGPU_ERR_ARGUMENT, "phase must be 0 or 1");
  const uint32_t R = h->ladder;
  const uint32_t pairs = ((R - phase) / 2) * (h->chains / R);
  if (pairs == 0) return 0;
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  std::vector<double> u(pairs);
  for (uint32_t i = 0; i < pairs; ++i) u[i] = rng_uniform(h->exchange_rng);  // always drawn, pair order
  HIP_TRY(h, hipMemcpyAsync(h->exch_u, u.data(), u.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));  // `u` dies here
  hipLaunchKernelGGL(exchange_kernel, dim3((pairs + 255) / 256), dim3(256), 0, h->stream, h->d, R, phase, h->exch_u,
                     h->exch_counts);
  HIP_TRY(h, hipGetLastError());
  return 0;
}

int mcmcgpu_get_ladders(mcmcgpu_handle h, double* beta, uint32_t* slot_chain, double* reading_ll, uint64_t* attempts,
                        uint64_t* swaps) {
  if (!h) return MCMCGPU_ERR_ARGUMENT;
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  unsigned long long counts[2] = {0, 0};
  if (beta)
    HIP_TRY(h, hipMemcpyAsync(beta, h->d.beta, h->chains * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if (slot_chain)
    HIP_TRY(h, hipMemcpyAsync(slot_chain, h->d.slot_chain, h->chains * sizeof(uint32_t), hipMemcpyDeviceToHost,
                              h->stream));
  std::vector<double> ll, llc;
  if (reading_ll) {
    ll.resize(h->chains);
    llc.resize(h->chains);
    HIP_TRY(h, hipMemcpyAsync(ll.data(), h->d.ll, h->chains * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(llc.data(), h->d.llc, h->chains * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  }
  HIP_TRY(h, hipMemcpyAsync(counts, h->exch_counts, sizeof counts, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  for (uint32_t c = 0; reading_ll && c < h->chains; ++c) reading_ll[c] = ll[c] - llc[c];
  if (attempts) *attempts = counts[0];
  if (swaps) *swaps = counts[1];
  return 0;
}

int mcmcgpu_run_split(mcmcgpu_handle h, uint32_t steps, uint32_t slices) {
  return mcmcgpu_run_split_reported(h, steps, slices, 0, nullptr);
}

int mcmcgpu_run_split_reported(mcmcgpu_handle h, uint32_t steps, uint32_t slices, uint32_t report_every, void* dev_out) {
  if (!h) return MCMCGPU_ERR_ARGUMENT;
  if (!h->problem_loaded || h->chains == 0) return fail(h, MCMCGPU_ERR_STATE, "upload a problem and init chains first");
  if (h->next_timestep == 0) return fail(h, MCMCGPU_ERR_STATE, "chains have not processed timestep 0 yet");
  if (h->d.max_slices == 0) return fail(h, MCMCGPU_ERR_STATE, "the handle was created without split mode (max_split_slices)");
  if (slices == 0 || slices > h->d.max_slices) return fail(h, MCMCGPU_ERR_ARGUMENT, "slices out of range");
  if (h->cfg.range_bearing && !h->bearings_loaded)
    return fail(h, MCMCGPU_ERR_STATE, "mcmcgpu_set_bearings must follow mcmcgpu_upload_problem on a range-and-bearing handle");
  mcmcgpu_chain_summary* reports = nullptr;  // one row of C records per report interval
  if (report_every) {
    const uint32_t intervals = (steps + report_every - 1) / report_every;
    reports = static_cast<mcmcgpu_chain_summary*>(dev_out);
    if (!reports) {  // the library's own buffer, read back with mcmcgpu_get_reports
      if (intervals > kMaxOwnReports)
        return fail(h, MCMCGPU_ERR_CAPACITY, "more report intervals than the library's buffer holds (" +
                                                 std::to_string(kMaxOwnReports) + "); pass a buffer or split the run");
      reports = h->reports;
      h->report_intervals = intervals;
    }
  }
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  const uint32_t T = edges_grown(h), C = h->chains;
  const uint32_t fblocks = std::max<uint32_t>(1, (h->d.F + 255) / 256);
  EventPair ev;  // the whole sequence counts as one timed launch (mcmcgpu_kernel_time)
  if (int rc = begin_timed(h, &ev)) return rc;
  hipError_t launch_error = hipSuccess;
  // ends(begin 1) | setup, scan, eval, ends(finish k, begin k+1) [, commit] | ... : 1 + 4 (5) * steps launches
  // an accepted state edge moves every later pose: on a long trajectory that is a kernel of its own
  // with many workgroups per chain, on a short one the ends kernel does it on the way
  const uint32_t defer_from = h->cfg.split_commit_min_poses ? h->cfg.split_commit_min_poses : kDeferCommitPoses;
  const uint32_t defer = T >= defer_from ? 1u : 0u;
  const uint32_t pieces = std::max<uint32_t>(1, std::min<uint32_t>(64, (T + 4095) / 4096));
  if (steps) hipLaunchKernelGGL(split_ends_kernel, dim3(C), dim3(256), 0, h->stream, h->d, T, slices, 0u, 1u, defer);
  for (uint32_t i = 0; i < steps; ++i) {
    if (h->d.F <= 1024) {
      hipLaunchKernelGGL(split_setup_scan_kernel, dim3(C), dim3(1024), 0, h->stream, h->d, slices);
    } else {
      hipLaunchKernelGGL(split_setup_kernel, dim3(fblocks, C), dim3(256), 0, h->stream, h->d);
      hipLaunchKernelGGL(split_scan_kernel, dim3(C), dim3(256), 0, h->stream, h->d, slices);
    }
    if (h->cfg.range_bearing)
      hipLaunchKernelGGL(split_eval_kernel<true>, dim3(slices, C), dim3(256), 0, h->stream, h->d);
    else
      hipLaunchKernelGGL(split_eval_kernel<false>, dim3(slices, C), dim3(256), 0, h->stream, h->d);
    hipLaunchKernelGGL(split_ends_kernel, dim3(C), dim3(256), 0, h->stream, h->d, T, slices, 1u, i + 1 < steps ? 1u : 0u,
                       defer);
    if (defer) hipLaunchKernelGGL(split_commit_kernel, dim3(pieces, C), dim3(256), 0, h->stream, h->d, T);
    if (reports && ((i + 1) % report_every == 0 || i + 1 == steps))  // the chains' summaries at an interval boundary
      hipLaunchKernelGGL(summaries_kernel, dim3((C + 255) / 256), dim3(256), 0, h->stream, h->d,
                         reports + size_t(i / report_every) * C);
    if ((launch_error = hipGetLastError()) != hipSuccess) break;
  }
  return end_timed(h, ev, launch_error, "split-mode launch");
}

int mcmcgpu_get_reports(mcmcgpu_handle h, mcmcgpu_chain_summary* out, uint32_t capacity, uint32_t* intervals) {
  if (!h || !intervals) return fail(h, MCMCGPU_ERR_ARGUMENT, "null argument");
  *intervals = h->report_intervals;
  if (!out) return 0;
  if (capacity < h->report_intervals) return fail(h, MCMCGPU_ERR_CAPACITY, "report array too small");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  const size_t n = size_t(h->report_intervals) * h->chains;
  if (n) HIP_TRY(h, hipMemcpyAsync(out, h->reports, n * sizeof(mcmcgpu_chain_summary), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return check_device_flags(h);
}

int mcmcgpu_eval_llr(mcmcgpu_handle h, uint32_t chain, int kind, uint32_t index, const double proposed[3],
                     double* out) {
  if (!h || !proposed || !out) return fail(h, MCMCGPU_ERR_ARGUMENT, "null argument");
  if (chain >= h->chains) return fail(h, MCMCGPU_ERR_ARGUMENT, "chain out of range");
  if (kind == MCMCGPU_STATE_EDGE) {
    if (index >= edges_grown(h)) return fail(h, MCMCGPU_ERR_ARGUMENT, "state edge out of range");
  } else if (kind == MCMCGPU_FEATURE_EDGE) {
    if (index >= h->d.F) return fail(h, MCMCGPU_ERR_ARGUMENT, "landmark out of range");
  } else {
    return fail(h, MCMCGPU_ERR_ARGUMENT, "unknown edge kind");
  }
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  if (kind == MCMCGPU_FEATURE_EDGE) {
    uint32_t nf = 0;
    HIP_TRY(h, hipMemcpyAsync(&nf, h->d.nf + chain, sizeof nf, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (index >= nf) return fail(h, MCMCGPU_ERR_ARGUMENT, "landmark not yet observed by this chain");
  }
  hipLaunchKernelGGL(eval_kernel, dim3(1), dim3(h->nthreads), 0, h->stream, h->d, chain, kind, index, proposed[0],
                     proposed[1], proposed[2], h->scratch);
  HIP_TRY(h, hipGetLastError());
  HIP_TRY(h, hipMemcpyAsync(out, h->scratch, sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return 0;
}

int mcmcgpu_commit(mcmcgpu_handle h, uint32_t chain, int kind, uint32_t index, const double proposed[3],
                   double new_weight, double log_ratio) {
  if (!h || !proposed) return fail(h, MCMCGPU_ERR_ARGUMENT, "null argument");
  if (chain >= h->chains) return fail(h, MCMCGPU_ERR_ARGUMENT, "chain out of range");
  if (kind == MCMCGPU_STATE_EDGE) {
    if (index >= edges_grown(h)) return fail(h, MCMCGPU_ERR_ARGUMENT, "state edge out of range");
  } else if (kind == MCMCGPU_FEATURE_EDGE) {
    uint32_t nf = 0;
    HIP_TRY(h, hipMemcpyAsync(&nf, h->d.nf + chain, sizeof nf, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (index >= nf) return fail(h, MCMCGPU_ERR_ARGUMENT, "landmark out of range");
  } else {
    return fail(h, MCMCGPU_ERR_ARGUMENT, "unknown edge kind");
  }
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  hipLaunchKernelGGL(commit_kernel, dim3(1), dim3(h->nthreads), 0, h->stream, h->d, chain, kind, index,
                     proposed[0], proposed[1], proposed[2], new_weight, log_ratio, edges_grown(h));
  HIP_TRY(h, hipGetLastError());
  return 0;
}

int mcmcgpu_rebuild(mcmcgpu_handle h) {
  if (!h) return MCMCGPU_ERR_ARGUMENT;
  if (h->chains == 0) return fail(h, MCMCGPU_ERR_STATE, "no chains");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  EventPair ev;
  if (int rc = begin_timed(h, &ev, 1)) return rc;
  const int rc = rebuild_impl(h, 0, h->chains);
  if (int rc2 = end_timed(h, ev, rc ? hipErrorUnknown : hipSuccess, "rebuild")) return rc ? rc : rc2;
  return 0;
}

int mcmcgpu_scan_poses_static(int32_t device, void* stream, uint32_t chains, uint32_t n, const double* dev_rel_x,
                              const double* dev_rel_y, const double* dev_rel_th, uint32_t rel_stride,
                              double* dev_cum_xy, double* dev_cum_th, uint32_t cum_stride) {
  if (!dev_rel_x || !dev_rel_y || !dev_rel_th || !dev_cum_xy || !dev_cum_th)
    return fail(nullptr, MCMCGPU_ERR_ARGUMENT, "null argument");
  if (chains == 0) return fail(nullptr, MCMCGPU_ERR_ARGUMENT, "chains must be positive");
  if (chains > kStaticScanChains)
    return fail(nullptr, MCMCGPU_ERR_CAPACITY, "mcmcgpu_scan_poses_static takes at most " +
                                                   std::to_string(kStaticScanChains) + " chains per call");
  // the plan rounds segment lengths up to whole tiles in 32-bit arithmetic: keep a tile's worth of headroom
  if (n > kStaticScanMaxPoses)
    return fail(nullptr, MCMCGPU_ERR_CAPACITY, "mcmcgpu_scan_poses_static takes at most " +
                                                   std::to_string(kStaticScanMaxPoses) + " relative poses per chain");
  if (rel_stride < n || cum_stride < n + 1)
    return fail(nullptr, MCMCGPU_ERR_ARGUMENT, "strides shorter than the trajectory");
  if (reinterpret_cast<uintptr_t>(dev_cum_xy) % 16 != 0)
    return fail(nullptr, MCMCGPU_ERR_ARGUMENT, "dev_cum_xy must be 16-byte aligned");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev)
    return fail(nullptr, MCMCGPU_ERR_HIP, "no such HIP device (this library has no CPU path)");
  mcmcgpu_handle h = nullptr;
  HIP_TRY(h, hipSetDevice(device));
  void* tot = nullptr;
  HIP_TRY(h, static_scratch_on(device, &tot));
  const ScanPlan plan = scan_plan(chains, n);
  if (!plan.single_pass && size_t(chains) * plan.nseg * sizeof(Pose) > kStaticScratchBytes)  // cannot happen for chains <= kStaticScanChains
    return fail(nullptr, MCMCGPU_ERR_CAPACITY, "scan plan exceeds the static scratch");
  ScanArgs a{};
  a.relx = dev_rel_x;
  a.rely = dev_rel_y;
  a.relth = dev_rel_th;
  a.pxy = reinterpret_cast<PoseXY*>(dev_cum_xy);
  a.pth = dev_cum_th;
  a.rstride = rel_stride;
  a.pstride = cum_stride;
  a.n = n;
  a.nseg = plan.nseg;
  a.seg_len = plan.seg_len;
  a.tot = static_cast<Pose*>(tot);  // every slot the second pass reads is written by the first: nothing to clear
  HIP_TRY(h, launch_scan(static_cast<hipStream_t>(stream), a, chains, plan.single_pass));
  return 0;
}

int mcmcgpu_get_log_likelihood(mcmcgpu_handle h, double* out) {
  if (!h || !out) return fail(h, MCMCGPU_ERR_ARGUMENT, "null argument");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  HIP_TRY(h, hipMemcpyAsync(out, h->d.ll, sizeof(double) * h->chains, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return check_device_flags(h);
}

int mcmcgpu_full_log_likelihood(mcmcgpu_handle h, double* out) {
  if (!h || !out) return fail(h, MCMCGPU_ERR_ARGUMENT, "null argument");
  if (h->chains == 0) return fail(h, MCMCGPU_ERR_STATE, "no chains");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  double* dout = nullptr;
  // the per-chain results go to the tile-total buffer of the rebuild scan (idle here)
  dout = reinterpret_cast<double*>(h->scan_tot);
  hipLaunchKernelGGL(full_ll_kernel, dim3(h->chains), dim3(h->nthreads), 0, h->stream, h->d, 0u, edges_grown(h), dout,
                     static_cast<double*>(nullptr));
  HIP_TRY(h, hipGetLastError());
  HIP_TRY(h, hipMemcpyAsync(out, dout, sizeof(double) * h->chains, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return 0;
}

int mcmcgpu_get_dims(mcmcgpu_handle h, uint32_t* timesteps, uint32_t* features, uint32_t* chains) {
  if (!h) return MCMCGPU_ERR_ARGUMENT;
  if (timesteps) *timesteps = edges_grown(h);
  if (chains) *chains = h->chains;
  if (features) {
    *features = 0;
    if (h->chains) {
      HIP_TRY(h, hipSetDevice(h->cfg.device));
      HIP_TRY(h, hipMemcpyAsync(features, h->d.nf, sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
      HIP_TRY(h, hipStreamSynchronize(h->stream));
    }
  }
  return 0;
}

int mcmcgpu_get_state(mcmcgpu_handle h, uint32_t chain, double* rel_poses, double* cum_poses,
                      uint32_t* feat_parent, double* feat_estimate, double* feat_world, double* state_weights,
                      double* feat_weights) {
  if (!h) return MCMCGPU_ERR_ARGUMENT;
  if (chain >= h->chains) return fail(h, MCMCGPU_ERR_ARGUMENT, "chain out of range");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  const Dev& d = h->d;
  const uint32_t T = edges_grown(h);
  uint32_t nf = 0;
  HIP_TRY(h, hipMemcpyAsync(&nf, d.nf + chain, sizeof nf, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  auto down = [&](void* dst, const void* src, size_t bytes) {
    return bytes ? hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, h->stream) : hipSuccess;
  };
  if (rel_poses) {
    std::vector<double> x(T), y(T), th(T);
    HIP_TRY(h, down(x.data(), d.relx + size_t(chain) * d.Tcap, sizeof(double) * T));
    HIP_TRY(h, down(y.data(), d.rely + size_t(chain) * d.Tcap, sizeof(double) * T));
    HIP_TRY(h, down(th.data(), d.relth + size_t(chain) * d.Tcap, sizeof(double) * T));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    for (uint32_t t = 0; t < T; ++t) {
      rel_poses[3 * size_t(t)] = x[t];
      rel_poses[3 * size_t(t) + 1] = y[t];
      rel_poses[3 * size_t(t) + 2] = th[t];
    }
  }
  if (cum_poses) {
    std::vector<PoseXY> xy(size_t(T) + 1);
    std::vector<double> th(size_t(T) + 1);
    HIP_TRY(h, down(xy.data(), d.pxy + size_t(chain) * (d.Tcap + 1), sizeof(PoseXY) * (size_t(T) + 1)));
    HIP_TRY(h, down(th.data(), d.pth + size_t(chain) * (d.Tcap + 1), sizeof(double) * (size_t(T) + 1)));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    for (size_t k = 0; k <= T; ++k) {
      cum_poses[3 * k] = xy[k].x;
      cum_poses[3 * k + 1] = xy[k].y;
      cum_poses[3 * k + 2] = th[k];
    }
  }
  if (feat_world || state_weights || feat_weights) {
    double* dworld = h->scratch;                  // [2*Fcap]
    double* dfw = dworld + 2 * size_t(d.Fcap);    // [Fcap]
    double* dsw = dfw + size_t(d.Fcap);           // [Tcap]
    const uint32_t n = std::max(T, nf);
    if (n) {
      hipLaunchKernelGGL(readout_kernel, dim3((n + 255) / 256), dim3(256), 0, h->stream, d, chain, T, dworld, dsw,
                         dfw);
      HIP_TRY(h, hipGetLastError());
    }
    if (feat_world) HIP_TRY(h, down(feat_world, dworld, sizeof(double) * 2 * nf));
    if (feat_weights) HIP_TRY(h, down(feat_weights, dfw, sizeof(double) * nf));
    if (state_weights) HIP_TRY(h, down(state_weights, dsw, sizeof(double) * T));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
  }
  if (feat_parent) {
    HIP_TRY(h, down(feat_parent, d.fpt + size_t(chain) * d.Fcap, sizeof(uint32_t) * nf));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
  }
  if (feat_estimate) {
    std::vector<double> x(nf), y(nf);
    HIP_TRY(h, down(x.data(), d.fex + size_t(chain) * d.Fcap, sizeof(double) * nf));
    HIP_TRY(h, down(y.data(), d.fey + size_t(chain) * d.Fcap, sizeof(double) * nf));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    for (uint32_t f = 0; f < nf; ++f) {
      feat_estimate[2 * size_t(f)] = x[f];
      feat_estimate[2 * size_t(f) + 1] = y[f];
    }
  }
  return 0;
}

int mcmcgpu_get_counters(mcmcgpu_handle h, uint64_t* updates, uint64_t* accepts, uint64_t* units) {
  if (!h) return MCMCGPU_ERR_ARGUMENT;
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  const size_t bytes = sizeof(uint64_t) * h->chains;
  if (updates) HIP_TRY(h, hipMemcpyAsync(updates, h->d.n_updates, bytes, hipMemcpyDeviceToHost, h->stream));
  if (accepts) HIP_TRY(h, hipMemcpyAsync(accepts, h->d.n_accepts, bytes, hipMemcpyDeviceToHost, h->stream));
  if (units) HIP_TRY(h, hipMemcpyAsync(units, h->d.n_units, bytes, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return 0;
}

int mcmcgpu_get_traffic_counters(mcmcgpu_handle h, uint64_t* state_proposals, uint64_t* moved_poses) {
  if (!h) return MCMCGPU_ERR_ARGUMENT;
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  const size_t bytes = sizeof(uint64_t) * h->chains;
  if (state_proposals)
    HIP_TRY(h, hipMemcpyAsync(state_proposals, h->d.n_state_props, bytes, hipMemcpyDeviceToHost, h->stream));
  if (moved_poses) HIP_TRY(h, hipMemcpyAsync(moved_poses, h->d.n_moved, bytes, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return 0;
}

int mcmcgpu_get_trace(mcmcgpu_handle h, uint32_t chain, mcmcgpu_step_record* out, uint32_t capacity,
                      uint32_t* count) {
  if (!h || !count) return fail(h, MCMCGPU_ERR_ARGUMENT, "null argument");
  if (chain >= h->chains) return fail(h, MCMCGPU_ERR_ARGUMENT, "chain out of range");
  if (!h->d.trace) return fail(h, MCMCGPU_ERR_STATE, "handle was created without a trace");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  uint32_t n = 0;
  HIP_TRY(h, hipMemcpyAsync(&n, h->d.trace_n + chain, sizeof n, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  n = std::min(n, h->d.trace_cap);
  n = std::min(n, capacity);
  if (n && out) {
    HIP_TRY(h, hipMemcpyAsync(out, h->d.trace + size_t(chain) * h->d.trace_cap, sizeof(mcmcgpu_step_record) * n,
                              hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
  }
  *count = n;
  return 0;
}

int mcmcgpu_write_summaries(mcmcgpu_handle h, void* dev_out) {
  if (!h || !dev_out) return fail(h, MCMCGPU_ERR_ARGUMENT, "null argument");
  if (h->chains == 0) return fail(h, MCMCGPU_ERR_STATE, "no chains");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  hipLaunchKernelGGL(summaries_kernel, dim3((h->chains + 255) / 256), dim3(256), 0, h->stream, h->d,
                     static_cast<mcmcgpu_chain_summary*>(dev_out));
  HIP_TRY(h, hipGetLastError());
  return 0;
}

int mcmcgpu_export_state(mcmcgpu_handle h, uint32_t chain, void* dev_out) {
  if (!h || !dev_out) return fail(h, MCMCGPU_ERR_ARGUMENT, "null argument");
  if (chain >= h->chains) return fail(h, MCMCGPU_ERR_ARGUMENT, "chain out of range");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  const Dev& d = h->d;
  const size_t T = edges_grown(h), F = d.F;
  double* o = static_cast<double*>(dev_out);
  auto cp = [&](double* dst, const double* src, size_t n) {
    return n ? hipMemcpyAsync(dst, src, n * sizeof(double), hipMemcpyDeviceToDevice, h->stream) : hipSuccess;
  };
  HIP_TRY(h, cp(o, d.relx + size_t(chain) * d.Tcap, T));
  HIP_TRY(h, cp(o + T, d.rely + size_t(chain) * d.Tcap, T));
  HIP_TRY(h, cp(o + 2 * T, d.relth + size_t(chain) * d.Tcap, T));
  HIP_TRY(h, cp(o + 3 * T, d.fex + size_t(chain) * d.Fcap, F));
  HIP_TRY(h, cp(o + 3 * T + F, d.fey + size_t(chain) * d.Fcap, F));
  return 0;
}

namespace {
// Landmarks the chains have met so far (the same for every chain: it depends on the data only).
int chains_features(mcmcgpu_handle h, uint32_t* nf) {
  *nf = 0;
  if (h->chains == 0) return 0;
  HIP_TRY(h, hipMemcpyAsync(nf, h->d.nf, sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return 0;
}
}  // namespace

int mcmcgpu_snapshot_begin(mcmcgpu_handle h, uint32_t chain) {
  if (!h) return MCMCGPU_ERR_ARGUMENT;
  if (chain >= h->chains) return fail(h, MCMCGPU_ERR_ARGUMENT, "chain out of range");
  if (h->snap_pending) return fail(h, MCMCGPU_ERR_STATE, "the previous snapshot has not been collected");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  const uint32_t T = edges_grown(h);
  const uint32_t n = std::max<uint32_t>(std::max(T, h->d.F), 1);
  hipLaunchKernelGGL(snapshot_kernel, dim3((n + 255) / 256), dim3(256), 0, h->stream, h->d, chain, T, h->snap_dev);
  HIP_TRY(h, hipGetLastError());
  const size_t used = 3 * size_t(T) + 2 * size_t(h->d.Fcap) + 2;  // poses, landmarks, log likelihood, count
  HIP_TRY(h, hipMemcpyAsync(h->snap_host, h->snap_dev, used * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipEventRecord(h->snap_event, h->stream));
  h->snap_pending = true;
  h->snap_T = T;
  return 0;
}

int mcmcgpu_snapshot_wait(mcmcgpu_handle h, uint32_t* timesteps, double* rel_poses, uint32_t* features,
                          double* feat_world, double* log_likelihood) {
  if (!h) return MCMCGPU_ERR_ARGUMENT;
  if (!h->snap_pending) return fail(h, MCMCGPU_ERR_STATE, "no snapshot was begun");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  HIP_TRY(h, hipEventSynchronize(h->snap_event));  // only the snapshot: later launches keep running
  h->snap_pending = false;
  const size_t T = h->snap_T;
  const double* s = h->snap_host;
  const uint32_t nf = uint32_t(s[3 * T + 2 * size_t(h->d.Fcap) + 1]);
  if (timesteps) *timesteps = uint32_t(T);
  if (features) *features = nf;
  if (rel_poses) std::memcpy(rel_poses, s, 3 * T * sizeof(double));
  if (feat_world) std::memcpy(feat_world, s + 3 * T, 2 * size_t(nf) * sizeof(double));
  if (log_likelihood) *log_likelihood = s[3 * T + 2 * size_t(nf)];
  return 0;
}

int mcmcgpu_average_base(mcmcgpu_handle h, uint32_t chain, void* dev_base) {
  if (!h) return MCMCGPU_ERR_ARGUMENT;
  if (chain >= h->chains) return fail(h, MCMCGPU_ERR_ARGUMENT, "chain out of range");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  if (!dev_base) {  // the handle's own buffer
    if (int rc = default_exchange_buffers(h)) return rc;
    dev_base = h->avg_base;
  }
  uint32_t nf = 0;
  if (int rc = chains_features(h, &nf)) return rc;
  const uint32_t T = edges_grown(h), n = T + nf;
  if (n == 0) return 0;
  hipLaunchKernelGGL(average_base_kernel, dim3((n + 255) / 256), dim3(256), 0, h->stream, h->d, chain, T, nf,
                     static_cast<double*>(dev_base));
  HIP_TRY(h, hipGetLastError());
  return 0;
}

int mcmcgpu_average_sum(mcmcgpu_handle h, const void* dev_base, int32_t skip_chain, void* dev_sum) {
  if (!h) return MCMCGPU_ERR_ARGUMENT;
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  if (!dev_base || !dev_sum) {  // the handle's own buffers
    if (int rc = default_exchange_buffers(h)) return rc;
    if (!dev_base) dev_base = h->avg_base;
    if (!dev_sum) dev_sum = h->avg_sum;
  }
  uint32_t nf = 0;
  if (int rc = chains_features(h, &nf)) return rc;
  const uint32_t T = edges_grown(h), n = T + nf;
  if (n == 0) return 0;
  hipLaunchKernelGGL(average_sum_kernel, dim3((n + 255) / 256), dim3(256), 0, h->stream, h->d, T, nf,
                     static_cast<const double*>(dev_base), skip_chain, static_cast<double*>(dev_sum));
  HIP_TRY(h, hipGetLastError());
  return 0;
}

int mcmcgpu_average_finish(mcmcgpu_handle h, void* dev_base, const void* dev_sum, uint32_t total_chains) {
  if (!h) return MCMCGPU_ERR_ARGUMENT;
  if (total_chains == 0) return fail(h, MCMCGPU_ERR_ARGUMENT, "total_chains must be positive");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  if (!dev_base || !dev_sum) {  // the handle's own buffers
    if (int rc = default_exchange_buffers(h)) return rc;
    if (!dev_base) dev_base = h->avg_base;
    if (!dev_sum) dev_sum = h->avg_sum;
  }
  uint32_t nf = 0;
  if (int rc = chains_features(h, &nf)) return rc;
  const uint32_t T = edges_grown(h), n = T + nf;
  if (n == 0) return 0;
  hipLaunchKernelGGL(average_finish_kernel, dim3((n + 255) / 256), dim3(256), 0, h->stream, T, nf,
                     static_cast<double*>(dev_base), static_cast<const double*>(dev_sum), double(total_chains));
  HIP_TRY(h, hipGetLastError());
  return 0;
}

int mcmcgpu_get_average(mcmcgpu_handle h, double* cumulative_poses, double* features_world) {
  if (!h) return MCMCGPU_ERR_ARGUMENT;
  if (h->chains == 0) return fail(h, MCMCGPU_ERR_STATE, "no chains initialised");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  uint32_t nf = 0;
  if (int rc = chains_features(h, &nf)) return rc;
  const size_t T = edges_grown(h), n = 3 * T + 2 * size_t(nf);
  if (2 * n > h->scratch_doubles) return fail(h, MCMCGPU_ERR_CAPACITY, "internal staging buffer too small");
  double* base = h->scratch;
  double* sum = h->scratch + n;
  if (int rc = mcmcgpu_average_base(h, 0, base)) return rc;
  if (int rc = mcmcgpu_average_sum(h, base, 0, sum)) return rc;
  if (int rc = mcmcgpu_average_finish(h, base, sum, h->chains)) return rc;
  if (cumulative_poses && T)
    HIP_TRY(h, hipMemcpyAsync(cumulative_poses, base, 3 * T * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if (features_world && nf)
    HIP_TRY(h, hipMemcpyAsync(features_world, base + 3 * T, 2 * size_t(nf) * sizeof(double), hipMemcpyDeviceToHost,
                              h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return 0;
}

int mcmcgpu_comm_unique_id(unsigned char* id) {
  if (!id) return fail(nullptr, MCMCGPU_ERR_ARGUMENT, "null argument");
  if (int rc = need_rccl(nullptr)) return rc;
  ncclUniqueId u;
  mcmcgpu_handle h = nullptr;
  RCCL_TRY(h, rccl().GetUniqueId(&u));
  static_assert(sizeof u == MCMCGPU_UNIQUE_ID_BYTES, "unique id size");
  std::memcpy(id, &u, sizeof u);
  return 0;
}

int mcmcgpu_comm_init(mcmcgpu_handle h, int nranks, int rank, const unsigned char* id, uint32_t slots_per_rank) {
  if (!h || !id) return fail(h, MCMCGPU_ERR_ARGUMENT, "null argument");
  if (nranks < 1 || rank < 0 || rank >= nranks) return fail(h, MCMCGPU_ERR_ARGUMENT, "rank out of range");
  if (slots_per_rank < h->cfg.max_chains) return fail(h, MCMCGPU_ERR_ARGUMENT, "slots_per_rank below the handle's max_chains");
  if (h->comm) return fail(h, MCMCGPU_ERR_STATE, "the handle already has a communicator");
  if (int rc = need_rccl(h)) return rc;
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  if (int rc = ensure_exchange_buffers(h, nranks, slots_per_rank)) return rc;
  ncclUniqueId u;
  std::memcpy(&u, id, sizeof u);
  const ncclResult_t made = rccl().CommInitRank(&h->comm, nranks, u, rank);
  if (made != ncclSuccess) {  // no communicator: back to the buffers of a handle on its own
    h->comm = nullptr;
    ensure_exchange_buffers(h, 1, h->cfg.max_chains);
    return fail(h, MCMCGPU_ERR_HIP, std::string("ncclCommInitRank: ") + rccl().GetErrorString(made));
  }
  h->comm_rank = rank;
  return 0;
}

int mcmcgpu_comm_init_all(mcmcgpu_handle* handles, int n, uint32_t slots_per_rank) {
  if (!handles || n < 1) return fail(nullptr, MCMCGPU_ERR_ARGUMENT, "no handles");
  if (int rc = need_rccl(handles[0])) return rc;
  std::vector<int> devices(n);
  for (int i = 0; i < n; ++i) {
    mcmcgpu_handle h = handles[i];
    if (!h) return fail(nullptr, MCMCGPU_ERR_ARGUMENT, "null handle");
    if (h->comm) return fail(h, MCMCGPU_ERR_STATE, "the handle already has a communicator");
    if (slots_per_rank < h->cfg.max_chains) return fail(h, MCMCGPU_ERR_ARGUMENT, "slots_per_rank below the handle's max_chains");
    devices[i] = h->cfg.device;
    for (int k = 0; k < i; ++k)
      if (devices[k] == devices[i]) return fail(h, MCMCGPU_ERR_ARGUMENT, "RCCL wants every rank on a device of its own");
  }
  for (int i = 0; i < n; ++i) {
    mcmcgpu_handle h = handles[i];
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    if (int rc = ensure_exchange_buffers(h, n, slots_per_rank)) return rc;
  }
  std::vector<ncclComm_t> comms(n, nullptr);
  mcmcgpu_handle h0 = handles[0];
  const ncclResult_t made = rccl().CommInitAll(comms.data(), n, devices.data());
  if (made != ncclSuccess) {  // no communicators: every handle back to the buffers of a handle on its own
    for (int i = 0; i < n; ++i) {
      (void)hipSetDevice(handles[i]->cfg.device);
      ensure_exchange_buffers(handles[i], 1, handles[i]->cfg.max_chains);
    }
    return fail(h0, MCMCGPU_ERR_HIP, std::string("ncclCommInitAll: ") + rccl().GetErrorString(made));
  }
  for (int i = 0; i < n; ++i) {
    handles[i]->comm = comms[i];
    handles[i]->comm_rank = i;
  }
  return 0;
}

int mcmcgpu_comm_info(mcmcgpu_handle h, int* nranks, int* rank, uint32_t* slots_per_rank) {
  if (!h) return MCMCGPU_ERR_ARGUMENT;
  if (nranks) *nranks = h->comm ? h->comm_ranks : 1;
  if (rank) *rank = h->comm ? h->comm_rank : 0;
  if (slots_per_rank) *slots_per_rank = h->gather_recv ? h->comm_slots : h->cfg.max_chains;
  return 0;
}

int mcmcgpu_group_begin(void) {
  if (int rc = need_rccl(nullptr)) return rc;
  mcmcgpu_handle h = nullptr;
  RCCL_TRY(h, rccl().GroupStart());
  return 0;
}

int mcmcgpu_group_end(void) {
  if (int rc = need_rccl(nullptr)) return rc;
  mcmcgpu_handle h = nullptr;
  RCCL_TRY(h, rccl().GroupEnd());
  return 0;
}

int mcmcgpu_allgather_summaries(mcmcgpu_handle h) {
  if (!h) return MCMCGPU_ERR_ARGUMENT;
  if (h->chains == 0) return fail(h, MCMCGPU_ERR_STATE, "no chains");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  // the chains' current summaries go through a row of their own behind the report rows, so the
  // records of the last reported run stay readable (mcmcgpu_get_reports, mcmcgpu_allgather_reports)
  mcmcgpu_chain_summary* const staging = h->reports + size_t(kMaxOwnReports) * h->cfg.max_chains;
  hipLaunchKernelGGL(summaries_kernel, dim3((h->chains + 255) / 256), dim3(256), 0, h->stream, h->d, staging);
  HIP_TRY(h, hipGetLastError());
  return gather_records(h, staging, 1);
}

int mcmcgpu_allgather_reports(mcmcgpu_handle h) {
  if (!h) return MCMCGPU_ERR_ARGUMENT;
  if (h->chains == 0) return fail(h, MCMCGPU_ERR_STATE, "no chains");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  return gather_records(h, h->reports, h->report_intervals);
}

int mcmcgpu_get_gathered(mcmcgpu_handle h, mcmcgpu_chain_summary* out, uint32_t capacity_records, uint32_t* ranks,
                         uint32_t* intervals, uint32_t* slots_per_rank) {
  if (!h) return MCMCGPU_ERR_ARGUMENT;
  if (!h->gather_recv) return fail(h, MCMCGPU_ERR_STATE, "nothing has been gathered");
  const uint32_t R = uint32_t(h->comm_ranks), K = h->gathered_intervals, S = h->comm_slots;
  if (ranks) *ranks = R;
  if (intervals) *intervals = K;
  if (slots_per_rank) *slots_per_rank = S;
  if (!out) return 0;
  const size_t n = size_t(R) * K * S;
  if (capacity_records < n) return fail(h, MCMCGPU_ERR_CAPACITY, "record array too small");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  if (n) HIP_TRY(h, hipMemcpyAsync(out, h->gather_recv, n * sizeof(mcmcgpu_chain_summary), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return check_device_flags(h);
}

int mcmcgpu_broadcast_state(mcmcgpu_handle h, int root, uint32_t chain) {
  if (!h) return MCMCGPU_ERR_ARGUMENT;
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  if (int rc = default_exchange_buffers(h)) return rc;
  const int ranks = h->comm ? h->comm_ranks : 1, rank = h->comm ? h->comm_rank : 0;
  if (root < 0 || root >= ranks) return fail(h, MCMCGPU_ERR_ARGUMENT, "root rank out of range");
  if (rank == root)
    if (int rc = mcmcgpu_export_state(h, chain, h->bcast)) return rc;
  const size_t n = 3 * size_t(edges_grown(h)) + 2 * size_t(h->d.F);
  if (h->comm && n) RCCL_TRY(h, rccl().Broadcast(h->bcast, h->bcast, n, ncclDouble, root, h->comm, h->stream));
  return 0;
}

int mcmcgpu_get_broadcast_state(mcmcgpu_handle h, double* rel_poses, double* feat_estimate) {
  if (!h) return MCMCGPU_ERR_ARGUMENT;
  if (!h->bcast) return fail(h, MCMCGPU_ERR_STATE, "no state has been broadcast");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  const size_t T = edges_grown(h), F = h->d.F;
  std::vector<double> packed(3 * T + 2 * F);
  if (!packed.empty())
    HIP_TRY(h, hipMemcpyAsync(packed.data(), h->bcast, packed.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  for (size_t t = 0; rel_poses && t < T; ++t)
    for (int k = 0; k < 3; ++k) rel_poses[3 * t + k] = packed[size_t(k) * T + t];
  for (size_t f = 0; feat_estimate && f < F; ++f)
    for (int k = 0; k < 2; ++k) feat_estimate[2 * f + k] = packed[3 * T + size_t(k) * F + f];
  return 0;
}

int mcmcgpu_average_collective(mcmcgpu_handle h, int phase, uint32_t total_chains, double* cumulative_poses,
                               double* features_world) {
  if (!h) return MCMCGPU_ERR_ARGUMENT;
  if (h->chains == 0) return fail(h, MCMCGPU_ERR_STATE, "no chains initialised");
  if (phase < 0 || phase > 2) return fail(h, MCMCGPU_ERR_ARGUMENT, "phase must be 0, 1 or 2");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  if (int rc = default_exchange_buffers(h)) return rc;
  uint32_t nf = 0;
  if (int rc = chains_features(h, &nf)) return rc;
  const size_t T = edges_grown(h), n = 3 * T + 2 * size_t(nf);
  const int rank = h->comm ? h->comm_rank : 0;
  if (phase == 0) {  // avg_acc's base vectors: global chain 0, on every rank
    if (rank == 0)
      if (int rc = mcmcgpu_average_base(h, 0, h->avg_base)) return rc;
    if (h->comm && n) RCCL_TRY(h, rccl().Broadcast(h->avg_base, h->avg_base, n, ncclDouble, 0, h->comm, h->stream));
  } else if (phase == 1) {  // the summed differences over all chains of all ranks but the base chain
    if (int rc = mcmcgpu_average_sum(h, h->avg_base, rank == 0 ? 0 : -1, h->avg_sum)) return rc;
    if (h->comm && n) RCCL_TRY(h, rccl().AllReduce(h->avg_sum, h->avg_sum, n, ncclDouble, ncclSum, h->comm, h->stream));
  } else {
    if (int rc = mcmcgpu_average_finish(h, h->avg_base, h->avg_sum, total_chains)) return rc;
    if (cumulative_poses && T)
      HIP_TRY(h, hipMemcpyAsync(cumulative_poses, h->avg_base, 3 * T * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (features_world && nf)
      HIP_TRY(h, hipMemcpyAsync(features_world, h->avg_base + 3 * T, 2 * size_t(nf) * sizeof(double), hipMemcpyDeviceToHost,
                                h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
  }
  return 0;
}

int mcmcgpu_average_buffer(mcmcgpu_handle h, int which, int write, double* host) {
  if (!h || !host) return fail(h, MCMCGPU_ERR_ARGUMENT, "null argument");
  if (which < 0 || which > 1) return fail(h, MCMCGPU_ERR_ARGUMENT, "which must be 0 (base) or 1 (sum)");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  if (int rc = default_exchange_buffers(h)) return rc;
  uint32_t nf = 0;
  if (int rc = chains_features(h, &nf)) return rc;
  const size_t bytes = (3 * size_t(edges_grown(h)) + 2 * size_t(nf)) * sizeof(double);
  double* dev = which == 0 ? h->avg_base : h->avg_sum;
  if (bytes) {
    if (write)
      HIP_TRY(h, hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, h->stream));
    else
      HIP_TRY(h, hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, h->stream));
  }
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return 0;
}

int mcmcgpu_import_state(mcmcgpu_handle h, uint32_t chain, const void* dev_in, double log_likelihood) {
  if (!h || !dev_in) return fail(h, MCMCGPU_ERR_ARGUMENT, "null argument");
  if (chain >= h->chains) return fail(h, MCMCGPU_ERR_ARGUMENT, "chain out of range");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  Dev& d = h->d;
  const size_t T = edges_grown(h), F = d.F;
  const double* in = static_cast<const double*>(dev_in);
  auto cp = [&](double* dst, const double* src, size_t n) {
    return n ? hipMemcpyAsync(dst, src, n * sizeof(double), hipMemcpyDeviceToDevice, h->stream) : hipSuccess;
  };
  HIP_TRY(h, cp(d.relx + size_t(chain) * d.Tcap, in, T));
  HIP_TRY(h, cp(d.rely + size_t(chain) * d.Tcap, in + T, T));
  HIP_TRY(h, cp(d.relth + size_t(chain) * d.Tcap, in + 2 * T, T));
  HIP_TRY(h, cp(d.fex + size_t(chain) * d.Fcap, in + 3 * T, F));
  HIP_TRY(h, cp(d.fey + size_t(chain) * d.Fcap, in + 3 * T + F, F));
  HIP_TRY(h, hipMemcpyAsync(d.ll + chain, &log_likelihood, sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  if (int rc = rebuild_impl(h, chain, 1)) return rc;
  hipLaunchKernelGGL(reweigh_kernel, dim3(1), dim3(256), 0, h->stream, d, chain, uint32_t(T));
  // the control part of the log likelihood (kept apart for tempering) from scratch; the total stays
  // the exporter's figure (the from-scratch total lands in the idle tile-total buffer of the scan)
  hipLaunchKernelGGL(full_ll_kernel, dim3(1), dim3(h->nthreads), 0, h->stream, d, chain, uint32_t(T),
                     reinterpret_cast<double*>(h->scan_tot), d.llc);
  HIP_TRY(h, hipGetLastError());
  return 0;
}

int mcmcgpu_load_state(mcmcgpu_handle h, uint32_t first, uint32_t count, const double* rel_poses,
                       const double* feat_world) {
  if (!h || !rel_poses || !feat_world) return fail(h, MCMCGPU_ERR_ARGUMENT, "null argument");
  if (!h->problem_loaded || h->chains == 0) return fail(h, MCMCGPU_ERR_STATE, "upload a problem and init chains first");
  if (count == 0 || first + count > h->chains) return fail(h, MCMCGPU_ERR_ARGUMENT, "chain range out of bounds");
  if (h->cfg.range_bearing && !h->bearings_loaded)
    return fail(h, MCMCGPU_ERR_STATE, "mcmcgpu_set_bearings must follow mcmcgpu_upload_problem on a range-and-bearing handle");
  Dev& d = h->d;
  const uint32_t T = d.T, F = d.F;
  if (h->next_timestep != T + 1 && !(first == 0 && count == h->chains))
    return fail(h, MCMCGPU_ERR_STATE, "chains that have not seen the whole problem can only be loaded all together");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  // per-landmark bookkeeping for "every reading seen": the parent is the reading
  // growth would end with (strictly smaller |range| wins, in time order)
  std::vector<uint32_t> fmeta(4 * size_t(F));
  for (uint32_t f = 0; f < F; ++f) {
    const uint32_t lo = h->f_start[f], hi = lo + h->f_count[f];
    uint32_t par = lo;
    for (uint32_t o = lo + 1; o < hi; ++o)
      if (std::fabs(h->h_omu[o]) < std::fabs(h->h_omu[par])) par = o;
    fmeta[f] = par - lo;
    fmeta[size_t(F) + f] = hi - lo;
    fmeta[2 * size_t(F) + f] = h->h_ot[par];
    fmeta[3 * size_t(F) + f] = h->h_ot[hi - 1];
  }
  double* drel = h->scratch;                                      // [T][3]
  double* dworld = drel + 3 * size_t(T);                          // [F][2]
  uint32_t* dmeta = reinterpret_cast<uint32_t*>(dworld + 2 * size_t(F));  // [4][F]
  auto up = [&](void* dst, const void* src, size_t bytes) {
    return bytes ? hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, h->stream) : hipSuccess;
  };
  HIP_TRY(h, up(drel, rel_poses, sizeof(double) * 3 * T));
  HIP_TRY(h, up(dworld, feat_world, sizeof(double) * 2 * F));
  HIP_TRY(h, up(dmeta, fmeta.data(), sizeof(uint32_t) * fmeta.size()));
  hipLaunchKernelGGL(load_state_kernel, dim3(count), dim3(256), 0, h->stream, d, first, T, F, drel, dmeta);
  HIP_TRY(h, hipGetLastError());
  h->next_timestep = T + 1;
  if (int rc = rebuild_impl(h, first, count)) return rc;
  if (F) {
    hipLaunchKernelGGL(features_from_world_kernel, dim3((F + 255) / 256, count), dim3(256), 0, h->stream, d, first, F,
                       dworld);
    HIP_TRY(h, hipGetLastError());
  }
  hipLaunchKernelGGL(reweigh_kernel, dim3(count), dim3(256), 0, h->stream, d, first, T);
  hipLaunchKernelGGL(full_ll_kernel, dim3(count), dim3(h->nthreads), 0, h->stream, d, first, T, d.ll, d.llc);
  HIP_TRY(h, hipGetLastError());
  HIP_TRY(h, hipStreamSynchronize(h->stream));  // host staging vector dies here
  return 0;
}

int mcmcgpu_sync(mcmcgpu_handle h) {
  if (!h) return MCMCGPU_ERR_ARGUMENT;
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return check_device_flags(h);
}

int mcmcgpu_kernel_time(mcmcgpu_handle h, double* avg_ms, uint32_t* launches) {
  if (!h) return MCMCGPU_ERR_ARGUMENT;
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  if (int rc = fold_timings(h, true)) return rc;
  const uint32_t n = h->timed_launches[0];
  if (avg_ms) *avg_ms = n ? h->timed_ms[0] / n : 0.0;
  if (launches) *launches = n;
  h->timed_ms[0] = 0.0;
  h->timed_launches[0] = 0;
  return 0;
}

int mcmcgpu_rebuild_time(mcmcgpu_handle h, double* avg_ms, uint32_t* launches) {
  if (!h) return MCMCGPU_ERR_ARGUMENT;
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  if (int rc = fold_timings(h, true)) return rc;
  const uint32_t n = h->timed_launches[1];
  if (avg_ms) *avg_ms = n ? h->timed_ms[1] / n : 0.0;
  if (launches) *launches = n;
  h->timed_ms[1] = 0.0;
  h->timed_launches[1] = 0;
  return 0;
}

void mcmcgpu_seed_sequence(uint32_t seed, uint32_t n, uint32_t* out) {
  RngState r;
  rng_seed(r, seed);
  for (uint32_t i = 0; i < n; ++i) out[i] = rng_next(r);
}

void mcmcgpu_uniform_sequence(uint32_t seed, uint32_t n, double* out) {
  RngState r;
  rng_seed(r, seed);
  for (uint32_t i = 0; i < n; ++i) out[i] = rng_uniform(r);
}

int mcmcgpu_debug_phase_cycles(mcmcgpu_handle h, uint64_t out[16], int reset) {
  if (!h || !out) return MCMCGPU_ERR_ARGUMENT;
#ifdef MG_PHASE_TIMING
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  HIP_TRY(h, hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase_cycles), 16 * sizeof(uint64_t)));
  if (reset) {
    const uint64_t zero[16] = {};
    HIP_TRY(h, hipMemcpyToSymbol(HIP_SYMBOL(g_phase_cycles), zero, sizeof zero));
  }
  return 0;
#else
  (void)reset;
  for (int i = 0; i < 16; ++i) out[i] = 0;
  return fail(h, MCMCGPU_ERR_STATE, "library was built without -DMG_PHASE_TIMING");
#endif
}

int mcmcgpu_selftest_scan(int32_t device, uint32_t n, int32_t p, const int32_t* a, const int32_t* b, int32_t* out_a,
                          int32_t* out_b) {
  if (!a || !b || !out_a || !out_b || p <= 1) return fail(nullptr, MCMCGPU_ERR_ARGUMENT, "bad argument");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev)
    return fail(nullptr, MCMCGPU_ERR_HIP, "no such HIP device");
  mcmcgpu_handle h = nullptr;
  HIP_TRY(h, hipSetDevice(device));
  int32_t* buf = nullptr;
  HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&buf), sizeof(int32_t) * 4 * std::max<uint32_t>(n, 1)));
  int32_t *da = buf, *db = buf + n, *oa = buf + 2 * size_t(n), *ob = buf + 3 * size_t(n);
  hipError_t e = hipMemcpy(da, a, sizeof(int32_t) * n, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(db, b, sizeof(int32_t) * n, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(scan_affine_kernel, dim3(1), dim3(kScanLanes), 0, nullptr, n, p, da, db, oa, ob);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpy(out_a, oa, sizeof(int32_t) * n, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(out_b, ob, sizeof(int32_t) * n, hipMemcpyDeviceToHost);
  (void)hipFree(buf);
  if (e != hipSuccess) return fail(nullptr, MCMCGPU_ERR_HIP, std::string("scan self-test: ") + hipGetErrorString(e));
  return 0;
}

int mcmcgpu_selftest_wave(int32_t device, uint32_t seed, uint32_t n, const uint8_t* kinds, double* out_serial, double* out_wave,
                          uint32_t tree_n, const double* weights, uint32_t n_search, const double* search_u,
                          uint32_t* index_serial, uint32_t* index_wave) {
  if (!kinds || !out_serial || !out_wave || !weights || !search_u || !index_serial || !index_wave || n == 0 || tree_n == 0 ||
      n_search == 0)
    return fail(nullptr, MCMCGPU_ERR_ARGUMENT, "bad argument");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev)
    return fail(nullptr, MCMCGPU_ERR_HIP, "no such HIP device");
  mcmcgpu_handle h = nullptr;
  HIP_TRY(h, hipSetDevice(device));
  // one allocation: kinds | draws (serial, wave) | weights, nodes | search uniforms | indices (serial, wave)
  const size_t nk = (size_t(n) + 7) / 8 * 8;
  const size_t bytes = nk + sizeof(double) * (2 * size_t(n) + 2 * size_t(tree_n) + n_search) + sizeof(uint32_t) * 2 * size_t(n_search);
  unsigned char* buf = nullptr;
  HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&buf), bytes));
  uint8_t* d_kinds = buf;
  double* d_serial = reinterpret_cast<double*>(buf + nk);
  double* d_wave = d_serial + n;
  double* d_weights = d_wave + n;
  double* d_nodes = d_weights + tree_n;
  double* d_u = d_nodes + tree_n;
  uint32_t* d_is = reinterpret_cast<uint32_t*>(d_u + n_search);
  uint32_t* d_iw = d_is + n_search;
  // the tree in the reference's layout (bitree<double>::push_back, src/utility/bitree.hpp:182-184), built here
  std::vector<double> nodes(tree_n);
  for (uint32_t i = 0; i < tree_n; ++i) wt::push_back(wt::Plain{nodes.data()}, i, weights[i]);
  hipError_t e = hipMemcpy(d_kinds, kinds, n, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d_nodes, nodes.data(), sizeof(double) * tree_n, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d_u, search_u, sizeof(double) * n_search, hipMemcpyHostToDevice);
  const char* stage = "copies";
  auto ran = [&](const char* what) {  // every kernel is waited for, so that a failure names it
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess && stage[0] == 'c') stage = what;
  };
  if (e == hipSuccess) {
    hipLaunchKernelGGL(serial_draws_selftest_kernel, dim3(1), dim3(64), 0, nullptr, seed, n, d_kinds, d_serial);
    ran("one-lane draws");
  }
  if (e == hipSuccess) {
    hipLaunchKernelGGL(wave_draws_selftest_kernel, dim3(1), dim3(64), 0, nullptr, seed, n, d_kinds, d_wave);
    ran("64-lane draws");
  }
  if (e == hipSuccess) {
    hipLaunchKernelGGL(search_selftest_kernel, dim3(1), dim3(64), 0, nullptr, tree_n, d_nodes, n_search, d_u, d_is, d_iw);
    ran("descents");
  }
  if (e == hipSuccess) e = hipMemcpy(out_serial, d_serial, sizeof(double) * n, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(out_wave, d_wave, sizeof(double) * n, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(index_serial, d_is, sizeof(uint32_t) * n_search, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(index_wave, d_iw, sizeof(uint32_t) * n_search, hipMemcpyDeviceToHost);
  (void)hipFree(buf);
  if (e != hipSuccess)
    return fail(nullptr, MCMCGPU_ERR_HIP, std::string("wavefront self-test (") + stage + "): " + hipGetErrorString(e));
  return 0;
}

int mcmcgpu_selftest_range(int32_t device, uint32_t n, const double* dx, const double* dy, double* out) {
  if (!dx || !dy || !out || n == 0) return fail(nullptr, MCMCGPU_ERR_ARGUMENT, "bad argument");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev)
    return fail(nullptr, MCMCGPU_ERR_HIP, "no such HIP device");
  mcmcgpu_handle h = nullptr;
  HIP_TRY(h, hipSetDevice(device));
  double* buf = nullptr;
  HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&buf), sizeof(double) * 3 * size_t(n)));
  hipError_t e = hipMemcpy(buf, dx, sizeof(double) * n, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(buf + n, dy, sizeof(double) * n, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(range_selftest_kernel, dim3((n + 255) / 256), dim3(256), 0, nullptr, n, buf, buf + n,
                       buf + 2 * size_t(n));
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpy(out, buf + 2 * size_t(n), sizeof(double) * n, hipMemcpyDeviceToHost);
  (void)hipFree(buf);
  if (e != hipSuccess) return fail(nullptr, MCMCGPU_ERR_HIP, std::string("range self-test: ") + hipGetErrorString(e));
  return 0;
}

}  // extern "C"
