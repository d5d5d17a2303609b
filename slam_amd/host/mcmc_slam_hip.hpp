// GPU-backed drop-in for the reference's sampler classes.
//
// slam::mcmc_slam<ControlModel, ObservationModel> here has the public surface
// of the reference class of the same name (src/slam/mcmc_slam.hpp:37-223):
// constructor (shared_ptr<const slam_data>, seed), non-copyable,
// get_log_likelihood(), set_initialiser(), update(), the timestep_listener and
// slam_result members, and the nested `updater`.  Its private state -- relative
// poses, landmark estimates, both weight trees, the log likelihood and the
// random stream -- lives on the GPU behind an mcmcgpu handle
// (include/mcmcgpu.h); nothing of the sampler runs on the host.
//
// slam::multi_mcmc mirrors src/slam/multi_mcmc.hpp:34-169: C chains (one
// workgroup each) seeded by successive draws of one generator, stepped together,
// best chain = first maximum of the log likelihoods.  The chains live in one
// handle, or sharded over one handle per GPU that exchange through RCCL.
//
// The class only uses these members of its collaborators, so it compiles
// against slam_types.hpp / slam_data.hpp here and against the reference's own
// headers alike: data->current_timestep(), control(t).mean()(i) / stddev()(i),
// observations() / observations_at(t) [positions], observation_info::id() /
// index() / observation().mean()(0) / stddev()(0), slam_result::get_state /
// get_feature, pose::cartesian / x() / y() / bearing(), position::cartesian /
// x() / y(), bitree::push_back / clear.
#ifndef SLAM_AMD_HOST_MCMC_SLAM_HIP_HPP
#define SLAM_AMD_HOST_MCMC_SLAM_HIP_HPP

#include <algorithm>
#include <cstdint>
#include <fstream>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/mcmcgpu.h"
#include "slam_data.hpp"
#include "slam_types.hpp"

namespace slam {

// Capacities of the device buffers a sampler allocates at construction
// (nothing is allocated later).  The defaults hold a million-pose run of one
// chain in well under a gigabyte.
struct gpu_capacity {
  std::uint32_t timesteps = 1u << 20, features = 1u << 12, observations = 1u << 22;
  int device = 0;
  bool feature_uniform_first = true;  // draw order inside a landmark proposal, see mcmcgpu_config
  // A single chain is one workgroup; on a large problem mcmc_slam::update spreads each proposal's
  // reading stream over this many workgroups instead (mcmcgpu_run_split), once the data holds at
  // least split_min_observations readings.  0 = never.
  std::uint32_t split_slices = 64, split_min_observations = 200000;
  // multi_mcmc only: the GPUs its chains are sharded over, one handle each (empty = {device}), and
  // how the handles exchange chain summaries, the best state and the averaging sums.
  //   automatic: one handle needs no exchange; several on different GPUs use RCCL; several that share
  //              a GPU (a rehearsal of the sharded run on one GPU) go through host memory
  //   rccl:      always through a communicator, also for a single handle (a communicator of one rank)
  //   host:      always through host memory
  std::vector<int> devices;
  enum class exchange_mode { automatic, rccl, host } exchange = exchange_mode::automatic;
};

namespace detail {

inline void check(mcmcgpu_handle h, int rc, const char* what) {
  if (rc != 0) throw std::runtime_error(std::string(what) + ": " + mcmcgpu_last_error(h));
}

// The flattened, growing copy of slam_data that is uploaded to the device:
// controls and readings in arrival order, landmarks numbered by first
// appearance (the index into the reference's feature_estimates vector).
template <class ControlModel, class ObservationModel>
class device_problem {
  using data_type = slam_data<ControlModel, ObservationModel>;
  std::vector<double> cmean, cstd, omean, ostd, bmean, bstd;
  std::vector<std::uint32_t> ot, of;
  std::map<std::size_t, std::uint32_t> dense;
  std::vector<std::size_t> ids;  // dense index -> landmark id

 public:
  // what append_to has handed over to one handle so far
  struct cursor {
    bool any = false;
    std::uint32_t T = 0, F = 0;
    std::size_t N = 0;
  };

 private:
  cursor own;  // for users with a single handle

 public:
  // range_bearing_model readings carry a second component (range_bearing_model.hpp:21-40)
  static constexpr bool bearing = ObservationModel::vector_dim == 2;
  static_assert(ObservationModel::vector_dim == 1 || ObservationModel::vector_dim == 2,
                "range-only or range-and-bearing observation models");
  std::uint32_t T() const { return std::uint32_t(cmean.size() / 3); }
  std::uint32_t N() const { return std::uint32_t(ot.size()); }
  std::uint32_t F() const { return std::uint32_t(ids.size()); }
  const std::vector<std::size_t>& feature_ids() const { return ids; }
  const std::map<std::size_t, std::uint32_t>& feature_index() const { return dense; }

  // Takes over everything `data` holds up to timestep `upto` that is not here yet.
  // Returns the dense indices of landmarks met for the first time, with their timestep.
  std::vector<std::pair<std::uint32_t, std::size_t>> absorb(const data_type& data, std::size_t upto) {
    std::vector<std::pair<std::uint32_t, std::size_t>> fresh;
    while (T() < upto) {
      const ControlModel& c = data.control(timestep_type(T()));
      for (int i = 0; i < 3; ++i) cmean.push_back(c.mean()(i));
      for (int i = 0; i < 3; ++i) cstd.push_back(c.stddev()(i));
    }
    const auto& all = data.observations();
    for (std::size_t i = N(); i < all.size() && std::size_t(all[i].first) <= upto; ++i) {
      const auto& info = all[i].second;
      const std::size_t id = info.id();
      auto ins = dense.emplace(id, std::uint32_t(ids.size()));
      if (ins.second) {
        ids.push_back(id);
        fresh.emplace_back(ins.first->second, std::size_t(all[i].first));
      }
      ot.push_back(std::uint32_t(std::size_t(all[i].first)));
      of.push_back(ins.first->second);
      omean.push_back(info.observation().mean()(0));
      ostd.push_back(info.observation().stddev()(0));
      if (bearing) {
        bmean.push_back(info.observation().mean()(ObservationModel::vector_dim - 1));
        bstd.push_back(info.observation().stddev()(ObservationModel::vector_dim - 1));
      }
    }
    return fresh;
  }
  // Online growth: hands the device every timestep absorbed since the last call, one
  // mcmcgpu_append_timestep each (the whole problem is never uploaded again).  irel / ifeat: the
  // initialiser's estimates per state edge (3 doubles) and per landmark (2 doubles, dense order),
  // as far as they reach.
  void append_to(mcmcgpu_handle h, const std::vector<double>* irel, const std::vector<double>* ifeat) {
    append_to(h, own, irel, ifeat);
  }
  void append_to(mcmcgpu_handle h, cursor& sent, const std::vector<double>* irel, const std::vector<double>* ifeat) {
    bool& sent_any = sent.any;
    std::uint32_t &sent_T = sent.T, &sent_F = sent.F;
    std::size_t& sent_N = sent.N;
    while (!sent_any || sent_T < T()) {
      const std::uint32_t tt = sent_any ? sent_T + 1 : 0;
      std::size_t lo = sent_N, hi = lo;
      while (hi < ot.size() && ot[hi] == tt) ++hi;
      std::uint32_t fresh = 0;
      for (std::size_t i = lo; i < hi; ++i)
        if (of[i] >= sent_F + fresh) ++fresh;
      const double* edge_estimate = irel && tt && irel->size() >= 3 * std::size_t(tt) ? irel->data() + 3 * std::size_t(tt - 1) : nullptr;
      const double* landmark_estimates =
          ifeat && fresh && ifeat->size() >= 2 * std::size_t(sent_F + fresh) ? ifeat->data() + 2 * std::size_t(sent_F) : nullptr;
      check(h,
            mcmcgpu_append_timestep(h, tt ? cmean.data() + 3 * std::size_t(tt - 1) : nullptr,
                                    tt ? cstd.data() + 3 * std::size_t(tt - 1) : nullptr, std::uint32_t(hi - lo), of.data() + lo,
                                    omean.data() + lo, ostd.data() + lo, bearing ? bmean.data() + lo : nullptr,
                                    bearing ? bstd.data() + lo : nullptr, edge_estimate, landmark_estimates),
            "mcmcgpu_append_timestep");
      sent_any = true;
      sent_T = tt;
      sent_N = hi;
      sent_F += fresh;
    }
  }
  void upload(mcmcgpu_handle h) const {
    check(h, mcmcgpu_upload_problem(h, T(), cmean.data(), cstd.data(), N(), ot.data(), of.data(), omean.data(),
                                    ostd.data(), F()),
          "mcmcgpu_upload_problem");
    if (bearing) check(h, mcmcgpu_set_bearings(h, N(), bmean.data(), bstd.data()), "mcmcgpu_set_bearings");
  }
};

}  // namespace detail

template <class ControlModel, class ObservationModel>
class mcmc_slam : public slam_result_of<ControlModel, ObservationModel> {
  using slam_result_type = slam_result_of<ControlModel, ObservationModel>;
  using slam_data_type = slam_data<ControlModel, ObservationModel>;
  using state_type = typename slam_result_type::state_type;
  using feature_type = typename slam_result_type::feature_type;
  using trajectory_type = typename slam_result_type::trajectory_type;
  using feature_map_type = typename slam_result_type::feature_map_type;

  const std::shared_ptr<const slam_data_type> data;
  std::shared_ptr<slam_result_type> initialiser;
  mcmcgpu_handle h = nullptr;
  detail::device_problem<ControlModel, ObservationModel> problem;
  std::vector<double> init_rel, init_feat;  // what the initialiser says about new edges
  std::size_t next_timestep = 0;
  std::uint64_t accepts_seen = 0;
  bool async_pending = false;  // update_async has queued updates whose accepts are not in accepts_seen
  std::uint32_t split_slices = 0, split_min_observations = 0;

  // `count` updates, queued: one persistent launch, or kernel sequences with sliced reading streams
  void run_updates(unsigned int count) {
    const bool split = split_slices && problem.N() >= split_min_observations;
    if (split)
      detail::check(h, mcmcgpu_run_split(h, count, split_slices), "mcmcgpu_run_split");
    else
      detail::check(h, mcmcgpu_run(h, count), "mcmcgpu_run");
  }

  // host mirrors for the slam_result read interface, rebuilt on demand
  mutable trajectory_type trajectory_mirror;
  mutable feature_map_type map_mirror;
  mutable bool trajectory_valid = false, map_valid = false;

  void invalidate() { trajectory_valid = map_valid = false; }

 public:
  mcmc_slam(const decltype(data)& data, unsigned int seed, const gpu_capacity& cap = gpu_capacity()) : data(data) {
    mcmcgpu_config cfg{};
    cfg.device = cap.device;
    cfg.max_chains = 1;
    cfg.max_timesteps = cap.timesteps;
    cfg.max_features = cap.features;
    cfg.max_observations = cap.observations;
    cfg.feature_uniform_first = cap.feature_uniform_first ? 1 : 0;
    cfg.range_bearing = decltype(problem)::bearing ? 1 : 0;
    cfg.max_split_slices = cap.split_slices;
    split_slices = cfg.max_split_slices;
    split_min_observations = cap.split_min_observations;
    if (mcmcgpu_create(&cfg, &h) != 0) throw std::runtime_error(std::string("mcmcgpu_create: ") + mcmcgpu_last_error(nullptr));
    const std::uint32_t s = seed;
    detail::check(h, mcmcgpu_init_chains(h, 1, &s, nullptr, 0), "mcmcgpu_init_chains");
  }
  ~mcmc_slam() { mcmcgpu_destroy(h); }
  mcmc_slam(const mcmc_slam&) = delete;
  mcmc_slam& operator=(const mcmc_slam&) = delete;

  double get_log_likelihood() const {
    double ll = 0;
    detail::check(h, mcmcgpu_get_log_likelihood(h, &ll), "mcmcgpu_get_log_likelihood");
    return ll;
  }

  void set_initialiser(const decltype(initialiser)& init) { initialiser = init; }

  // One Metropolis-Hastings step; true if the proposal was accepted.
  bool update() { return update(1) == 1; }

  // `count` steps in one launch; returns how many were accepted.
  unsigned int update(unsigned int count) {
    if (next_timestep == 0 || count == 0) return 0;
    if (async_pending) {  // the accept counter as it stands once the updates queued by update_async have run
      detail::check(h, mcmcgpu_get_counters(h, nullptr, &accepts_seen, nullptr), "mcmcgpu_get_counters");
      async_pending = false;
    }
    run_updates(count);
    std::uint64_t acc = 0;
    detail::check(h, mcmcgpu_get_counters(h, nullptr, &acc, nullptr), "mcmcgpu_get_counters");
    detail::check(h, mcmcgpu_sync(h), "mcmcgpu_sync");
    const unsigned int n = static_cast<unsigned int>(acc - accepts_seen);
    accepts_seen = acc;
    if (n) invalidate();
    return n;
  }

  // timestep_listener: grow the graph up to timestep t with what `data` holds.
  void timestep(timestep_type t) override {
    if (std::size_t(t) + 1 <= next_timestep) return;
    const auto fresh = problem.absorb(*data, std::size_t(t));
    if (initialiser) {
      // -init(t) + init(t+1) per new state edge, -init(t_first) + init(landmark) per new landmark
      for (std::size_t e = init_rel.size() / 3; e < problem.T(); ++e) {
        const state_type d = -initialiser->get_state(timestep_type(e)) + initialiser->get_state(timestep_type(e + 1));
        init_rel.insert(init_rel.end(), {d.x(), d.y(), d.bearing()});
      }
      for (const auto& f : fresh) {
        const feature_type p =
            -initialiser->get_state(timestep_type(f.second)) + initialiser->get_feature(featureid_type(problem.feature_ids()[f.first]));
        init_feat.insert(init_feat.end(), {p.x(), p.y()});
      }
    }
    problem.append_to(h, initialiser ? &init_rel : nullptr, initialiser ? &init_feat : nullptr);
    detail::check(h, mcmcgpu_grow(h, std::uint32_t(std::size_t(t))), "mcmcgpu_grow");
    next_timestep = std::size_t(t) + 1;
    invalidate();
  }

  timestep_type current_timestep() const override { return timestep_type(next_timestep == 0 ? 0 : next_timestep - 1); }

  feature_type get_feature(featureid_type id) const override { return get_feature_map().at(id); }

  // Relative poses as a bitree, like the reference returns them.
  const trajectory_type& get_trajectory() const override {
    if (!trajectory_valid) {
      const std::size_t T = current_timestep();
      std::vector<double> rel(3 * T);
      detail::check(h, mcmcgpu_get_state(h, 0, rel.data(), nullptr, nullptr, nullptr, nullptr, nullptr, nullptr),
                    "mcmcgpu_get_state");
      trajectory_mirror.clear();
      trajectory_mirror.reserve(T);
      for (std::size_t t = 0; t < T; ++t)
        trajectory_mirror.push_back(state_type::cartesian(rel[3 * t], rel[3 * t + 1], rel[3 * t + 2]));
      trajectory_valid = true;
    }
    return trajectory_mirror;
  }

  const feature_map_type& get_feature_map() const override {
    if (!map_valid) {
      std::uint32_t T = 0, F = 0, C = 0;
      detail::check(h, mcmcgpu_get_dims(h, &T, &F, &C), "mcmcgpu_get_dims");
      std::vector<double> world(2 * std::size_t(F));
      detail::check(h, mcmcgpu_get_state(h, 0, nullptr, nullptr, nullptr, nullptr, world.data(), nullptr, nullptr),
                    "mcmcgpu_get_state");
      map_mirror.clear();
      for (std::uint32_t f = 0; f < F; ++f)
        map_mirror[problem.feature_ids()[f]] = feature_type::cartesian(world[2 * f], world[2 * f + 1]);
      map_valid = true;
    }
    return map_mirror;
  }

  // The estimate as a result object of its own, fetched without stalling the sampler: begin the
  // copy, queue more updates behind it, collect it when the CPU consumer is ready for it -- the
  // candidate loop of the reference's main.cpp:331-337 with the clustering back end running
  // beside the next batch of updates instead of after it.
  void snapshot_begin() { detail::check(h, mcmcgpu_snapshot_begin(h, 0), "mcmcgpu_snapshot_begin"); }
  std::unique_ptr<slam_result_impl<state_type, feature_type>> snapshot_collect(double* log_likelihood = nullptr) {
    std::vector<double> rel(3 * problem.T()), world(2 * std::size_t(problem.F()));
    std::uint32_t T = 0, F = 0;
    detail::check(h, mcmcgpu_snapshot_wait(h, &T, rel.data(), &F, world.data(), log_likelihood), "mcmcgpu_snapshot_wait");
    auto result = std::make_unique<slam_result_impl<state_type, feature_type>>();
    result->get_trajectory().reserve(T);
    for (std::uint32_t t = 0; t < T; ++t)
      result->get_trajectory().push_back(state_type::cartesian(rel[3 * t], rel[3 * t + 1], rel[3 * t + 2]));
    for (std::uint32_t f = 0; f < F; ++f)
      result->get_feature_map()[problem.feature_ids()[f]] = feature_type::cartesian(world[2 * f], world[2 * f + 1]);
    return result;
  }
  // updates queued without waiting for them (update(count) waits, to report the accept count);
  // what they accept is not reported by a later update(count): that one counts from its own start
  void update_async(unsigned int count) {
    if (next_timestep == 0 || count == 0) return;
    run_updates(count);
    async_pending = true;
    invalidate();
  }

  mcmcgpu_handle native_handle() const { return h; }

  class updater : public timestep_listener {
    std::shared_ptr<mcmc_slam> instance;
    unsigned int steps, end_steps;

   public:
    updater(const decltype(instance)& instance, unsigned int steps = 0, unsigned int end_steps = 0)
        : instance(instance), steps(steps), end_steps(end_steps) {}
    void timestep(timestep_type t) override {
      instance->timestep(t);
      instance->update_async(steps);  // queued behind the growth; nothing waits per timestep
    }
    void completed() override {
      instance->completed();
      instance->update(end_steps);
    }
  };
};

template <class ControlModel, class ObservationModel>
class multi_mcmc : public slam_result_of<ControlModel, ObservationModel> {
  using slam_result_type = slam_result_of<ControlModel, ObservationModel>;
  using slam_data_type = slam_data<ControlModel, ObservationModel>;
  using state_type = typename slam_result_type::state_type;
  using feature_type = typename slam_result_type::feature_type;
  using trajectory_type = typename slam_result_type::trajectory_type;
  using feature_map_type = typename slam_result_type::feature_map_type;
  using problem_type = detail::device_problem<ControlModel, ObservationModel>;

  const std::shared_ptr<const slam_data_type> data;
  // The chains are sharded over one handle per GPU: handle r holds the global chains
  // [r * slots, r * slots + count[r]), in the order multi_mcmc keeps them in its vector
  // (multi_mcmc.hpp:164-166).  One GPU = one handle = the whole vector.
  std::vector<mcmcgpu_handle> hs;
  std::vector<unsigned int> count;
  std::vector<typename problem_type::cursor> sent;
  unsigned int slots = 0;
  bool use_rccl = false;
  problem_type problem;
  unsigned int chains, mcmc_end_steps;
  std::uint32_t split_slices = 0, split_min_observations = 0;
  std::size_t next_timestep = 0;
  unsigned int max_likelihood = 0;  // index of the best chain
  double best_ll = 0;
  std::vector<double> chain_ll;     // every chain's log likelihood after its latest round of updates

  mcmcgpu_handle owner_of(unsigned int chain, std::uint32_t* local) const {
    *local = chain % slots;
    return hs[chain / slots];
  }
  void group_begin() const {
    if (use_rccl && hs.size() > 1) detail::check(nullptr, mcmcgpu_group_begin(), "mcmcgpu_group_begin");
  }
  void group_end() const {
    if (use_rccl && hs.size() > 1) detail::check(nullptr, mcmcgpu_group_end(), "mcmcgpu_group_end");
  }

  // All-gather of the chains' summaries (reports = false) or of the report records of the launch
  // just queued (reports = true); returns them as [interval][global chain].  Through RCCL every
  // handle ends up with all records and handle 0's copy is read; handles that share a GPU gather
  // their own records only and the host stitches them.
  std::vector<mcmcgpu_chain_summary> gather(bool reports, std::uint32_t* intervals_out) const {
    group_begin();
    for (mcmcgpu_handle h : hs)
      detail::check(h, reports ? mcmcgpu_allgather_reports(h) : mcmcgpu_allgather_summaries(h),
                    reports ? "mcmcgpu_allgather_reports" : "mcmcgpu_allgather_summaries");
    group_end();
    std::vector<mcmcgpu_chain_summary> out, part;
    std::uint32_t K = 0;
    auto place = [&](std::uint32_t first_rank, std::uint32_t R, std::uint32_t S) {  // part: [R][K][S]
      if (out.empty()) out.resize(std::size_t(K) * chains);
      for (std::uint32_t r = 0; r < R; ++r)
        for (std::uint32_t k = 0; k < K; ++k)
          for (std::uint32_t s = 0; s < count[first_rank + r]; ++s)
            out[std::size_t(k) * chains + std::size_t(first_rank + r) * slots + s] = part[(std::size_t(r) * K + k) * S + s];
    };
    const std::size_t readers = use_rccl ? 1 : hs.size();
    for (std::size_t g = 0; g < readers; ++g) {
      std::uint32_t R = 0, S = 0;
      detail::check(hs[g], mcmcgpu_get_gathered(hs[g], nullptr, 0, &R, &K, &S), "mcmcgpu_get_gathered");
      part.resize(std::size_t(R) * K * S);
      detail::check(hs[g], mcmcgpu_get_gathered(hs[g], part.data(), std::uint32_t(part.size()), &R, &K, &S), "mcmcgpu_get_gathered");
      place(std::uint32_t(g), R, S);
    }
    if (use_rccl)  // the other handles' gathers are waited for as well, and their error flags looked at
      for (std::size_t g = 1; g < hs.size(); ++g) detail::check(hs[g], mcmcgpu_sync(hs[g]), "mcmcgpu_sync");
    *intervals_out = K;
    return out;
  }

  // The comparison loop of multi_mcmc::update (multi_mcmc.hpp:109-118) for one round.  The
  // reference steps chain after chain and compares each, right after ITS updates, with the current
  // best chain -- whose own updates of this round have only happened if it comes earlier in the
  // order.  `after` holds every chain's log likelihood at the end of the round, chain_ll the values
  // of the round before.
  void ensure_baseline() {  // the log likelihoods the chains have before their first round
    if (!chain_ll.empty()) return;
    std::uint32_t K = 0;
    const auto rec = gather(false, &K);
    chain_ll.resize(chains);
    for (unsigned int c = 0; c < chains; ++c) chain_ll[c] = rec[c].log_likelihood;
  }
  void compare_round(const double* after) {
    for (unsigned int c = 0; c < chains; ++c) {
      const double best_now = max_likelihood < c ? after[max_likelihood] : chain_ll[max_likelihood];
      if (after[c] > best_now) max_likelihood = c;
    }
    chain_ll.assign(after, after + chains);
    best_ll = chain_ll[max_likelihood];
  }
  std::string report_path;
  mutable trajectory_type trajectory_mirror;
  mutable feature_map_type map_mirror;

 public:
  multi_mcmc(std::shared_ptr<const slam_data_type> data, unsigned int num_chains, unsigned int end_steps,
             unsigned int seed, const gpu_capacity& cap = gpu_capacity(),
             const std::string& report = "multi-mcmc-report.txt")
      : data(data), chains(num_chains), mcmc_end_steps(end_steps), report_path(report) {
    if (num_chains == 0) throw std::runtime_error("multi_mcmc: no chains");
    std::vector<int> devices = cap.devices.empty() ? std::vector<int>{cap.device} : cap.devices;
    slots = (num_chains + unsigned(devices.size()) - 1) / unsigned(devices.size());
    devices.resize((num_chains + slots - 1) / slots);  // fewer chains than GPUs: the last ones stay idle
    bool distinct = true;
    for (std::size_t i = 0; i < devices.size(); ++i)
      for (std::size_t k = 0; k < i; ++k) distinct = distinct && devices[i] != devices[k];
    using mode = gpu_capacity::exchange_mode;
    use_rccl = cap.exchange == mode::rccl || (cap.exchange == mode::automatic && devices.size() > 1 && distinct);
    if (use_rccl && !distinct) throw std::runtime_error("multi_mcmc: RCCL needs every handle on a GPU of its own");
    std::vector<std::uint32_t> seeds(num_chains);
    mcmcgpu_seed_sequence(seed, num_chains, seeds.data());  // successive draws of random_source(seed), chain order
    for (std::size_t g = 0; g < devices.size(); ++g) {
      mcmcgpu_config cfg{};
      cfg.device = devices[g];
      cfg.max_chains = slots;
      cfg.max_timesteps = cap.timesteps;
      cfg.max_features = cap.features;
      cfg.max_observations = cap.observations;
      cfg.feature_uniform_first = cap.feature_uniform_first ? 1 : 0;
      cfg.range_bearing = problem_type::bearing ? 1 : 0;
      if (num_chains <= 64) {  // split mode only pays for few chains (see completed)
        cfg.max_split_slices = cap.split_slices;
        split_slices = cap.split_slices;
        split_min_observations = cap.split_min_observations;
      }
      mcmcgpu_handle h = nullptr;
      if (mcmcgpu_create(&cfg, &h) != 0) {
        const std::string why = mcmcgpu_last_error(nullptr);
        for (mcmcgpu_handle made : hs) mcmcgpu_destroy(made);
        throw std::runtime_error("mcmcgpu_create: " + why);
      }
      hs.push_back(h);
      count.push_back(std::min<unsigned int>(slots, num_chains - unsigned(g) * slots));
      detail::check(h, mcmcgpu_init_chains(h, count[g], seeds.data() + g * slots, nullptr, 0), "mcmcgpu_init_chains");
    }
    sent.resize(hs.size());
    if (use_rccl) detail::check(hs[0], mcmcgpu_comm_init_all(hs.data(), int(hs.size()), slots), "mcmcgpu_comm_init_all");
  }
  ~multi_mcmc() {
    for (mcmcgpu_handle h : hs) mcmcgpu_destroy(h);
  }
  multi_mcmc(const multi_mcmc&) = delete;
  multi_mcmc& operator=(const multi_mcmc&) = delete;

  unsigned int num_chains() const { return chains; }
  unsigned int num_handles() const { return unsigned(hs.size()); }
  bool exchanges_through_rccl() const { return use_rccl; }
  double get_log_likelihood() const { return best_ll; }

  // average_slam_result over the chains (multi_mcmc.hpp:71-73): summed on the devices, only the
  // averaged accumulated poses and landmark positions come back; the tree of relative poses is
  // rebuilt from them as the reference does (average_slam_result.hpp:94-99).  Three phases over
  // all handles: base vectors from global chain 0 to everyone, the sums of differences added up
  // over the handles, the average.
  std::unique_ptr<slam_result_type> get_average() const {
    std::uint32_t T = 0, F = 0, C = 0;
    detail::check(hs[0], mcmcgpu_get_dims(hs[0], &T, &F, &C), "mcmcgpu_get_dims");
    std::vector<double> cum(3 * std::size_t(T)), world(2 * std::size_t(F));
    const std::size_t n = cum.size() + world.size();
    if (use_rccl || hs.size() == 1) {
      for (int phase = 0; phase < 3; ++phase) {
        if (phase < 2) group_begin();
        for (std::size_t g = 0; g < hs.size(); ++g)
          detail::check(hs[g],
                        mcmcgpu_average_collective(hs[g], phase, chains, g == 0 ? cum.data() : nullptr,
                                                   g == 0 ? world.data() : nullptr),
                        "mcmcgpu_average_collective");
        if (phase < 2) group_end();
      }
    } else {  // handles that share a GPU: the host carries base and sums between them
      std::vector<double> base(n), total(n, 0.0), part(n);
      detail::check(hs[0], mcmcgpu_average_collective(hs[0], 0, chains, nullptr, nullptr), "mcmcgpu_average_collective");
      detail::check(hs[0], mcmcgpu_average_buffer(hs[0], 0, 0, base.data()), "mcmcgpu_average_buffer");
      for (std::size_t g = 0; g < hs.size(); ++g) {
        if (g) detail::check(hs[g], mcmcgpu_average_buffer(hs[g], 0, 1, base.data()), "mcmcgpu_average_buffer");
        // handle 0 leaves out the base chain itself; the others own no global chain 0
        detail::check(hs[g], mcmcgpu_average_sum(hs[g], nullptr, g == 0 ? 0 : -1, nullptr), "mcmcgpu_average_sum");
        detail::check(hs[g], mcmcgpu_average_buffer(hs[g], 1, 0, part.data()), "mcmcgpu_average_buffer");
        for (std::size_t i = 0; i < n; ++i) total[i] += part[i];
      }
      detail::check(hs[0], mcmcgpu_average_buffer(hs[0], 1, 1, total.data()), "mcmcgpu_average_buffer");
      detail::check(hs[0], mcmcgpu_average_collective(hs[0], 2, chains, cum.data(), world.data()), "mcmcgpu_average_collective");
    }
    auto result = std::make_unique<slam_result_impl<state_type, feature_type>>();
    result->get_trajectory().reserve(T);
    for (std::uint32_t t = 0; t < T; ++t)
      result->get_trajectory().push_back_accumulated(state_type::cartesian(cum[3 * t], cum[3 * t + 1], cum[3 * t + 2]));
    for (std::uint32_t f = 0; f < F; ++f)
      result->get_feature_map()[problem.feature_ids()[f]] = feature_type::cartesian(world[2 * f], world[2 * f + 1]);
    return result;
  }

  // `count` updates on every chain, then the arg-max (strict >, first maximum wins).
  void update(unsigned int n) {
    ensure_baseline();
    for (mcmcgpu_handle h : hs) detail::check(h, mcmcgpu_run(h, n), "mcmcgpu_run");
    std::uint32_t K = 0;
    const auto rec = gather(false, &K);
    std::vector<double> ll(chains);
    for (unsigned int c = 0; c < chains; ++c) ll[c] = rec[c].log_likelihood;
    compare_round(ll.data());
  }

  // The reference's multi_mcmc::timestep only forwards the timestep to its chains
  // (multi_mcmc.hpp:77-79): the new timestep's data is appended on every handle and the chains grow.
  void timestep(timestep_type t) override {
    if (std::size_t(t) + 1 <= next_timestep) return;
    problem.absorb(*data, std::size_t(t));
    for (std::size_t g = 0; g < hs.size(); ++g) {
      problem.append_to(hs[g], sent[g], nullptr, nullptr);
      detail::check(hs[g], mcmcgpu_grow(hs[g], std::uint32_t(std::size_t(t))), "mcmcgpu_grow");
    }
    next_timestep = std::size_t(t) + 1;
  }

  // End steps with a report line after every 100 updates (multi_mcmc.hpp:121-139).  The chains
  // run up to 256 report intervals per launch and leave one record per chain and interval behind
  // (mcmcgpu_run_reported); the records of all handles are gathered, and the lines, and the best
  // chain after every interval, are derived from them exactly as the reference's round-by-round
  // loop would have.
  void completed() override {
    std::ofstream report(report_path);
    unsigned int remaining = mcmc_end_steps;
    const unsigned int report_every = 100, max_intervals = 256;
    std::vector<double> after(chains);
    ensure_baseline();
    while (remaining > 0) {
      const unsigned int n = std::min(remaining, report_every * max_intervals);
      remaining -= n;
      // few chains on a large data set leave most of the GPU idle with one workgroup each: their
      // reading streams are then cut into slices (measured on 100 000 poses: 1.5x at 16 chains,
      // 1.15x at 64, a loss from about 200 chains on)
      for (mcmcgpu_handle h : hs) {
        if (split_slices && chains <= 64 && problem.N() >= split_min_observations)
          detail::check(h, mcmcgpu_run_split_reported(h, n, split_slices, report_every, nullptr), "mcmcgpu_run_split_reported");
        else
          detail::check(h, mcmcgpu_run_reported(h, n, report_every, nullptr), "mcmcgpu_run_reported");
      }
      std::uint32_t intervals = 0;
      const auto rec = gather(true, &intervals);
      for (std::uint32_t k = 0; k < intervals; ++k) {
        std::uint64_t u = 0, a = 0;
        for (unsigned int c = 0; c < chains; ++c) {
          const mcmcgpu_chain_summary& r = rec[std::size_t(k) * chains + c];
          after[c] = r.log_likelihood;
          u += r.updates;
          a += r.accepts;
        }
        compare_round(after.data());
        report << (double(u) / chains) << '\t' << (double(a) / chains) << '\t' << get_log_likelihood() << '\n';
      }
    }
  }

  timestep_type current_timestep() const override { return timestep_type(next_timestep == 0 ? 0 : next_timestep - 1); }

  // The best chain's trajectory.  Through RCCL its owner broadcasts the packed state to every
  // handle (what a process per GPU would do to let each rank answer); otherwise it is read from
  // the handle that holds it.
  const trajectory_type& get_trajectory() const override {
    const std::size_t T = current_timestep();
    std::vector<double> rel(3 * T);
    std::uint32_t local = 0;
    mcmcgpu_handle owner = owner_of(max_likelihood, &local);
    if (use_rccl) {
      group_begin();
      for (mcmcgpu_handle h : hs)
        detail::check(h, mcmcgpu_broadcast_state(h, int(max_likelihood / slots), local), "mcmcgpu_broadcast_state");
      group_end();
      detail::check(hs[0], mcmcgpu_get_broadcast_state(hs[0], rel.data(), nullptr), "mcmcgpu_get_broadcast_state");
    } else {
      detail::check(owner, mcmcgpu_get_state(owner, local, rel.data(), nullptr, nullptr, nullptr, nullptr, nullptr, nullptr),
                    "mcmcgpu_get_state");
    }
    trajectory_mirror.clear();
    for (std::size_t t = 0; t < T; ++t)
      trajectory_mirror.push_back(state_type::cartesian(rel[3 * t], rel[3 * t + 1], rel[3 * t + 2]));
    return trajectory_mirror;
  }

  const feature_map_type& get_feature_map() const override {
    std::uint32_t local = 0;
    mcmcgpu_handle owner = owner_of(max_likelihood, &local);
    std::uint32_t T = 0, F = 0, C = 0;
    detail::check(owner, mcmcgpu_get_dims(owner, &T, &F, &C), "mcmcgpu_get_dims");
    std::vector<double> world(2 * std::size_t(F));
    detail::check(owner, mcmcgpu_get_state(owner, local, nullptr, nullptr, nullptr, nullptr, world.data(), nullptr, nullptr),
                  "mcmcgpu_get_state");
    map_mirror.clear();
    for (std::uint32_t f = 0; f < F; ++f)
      map_mirror[problem.feature_ids()[f]] = feature_type::cartesian(world[2 * f], world[2 * f + 1]);
    return map_mirror;
  }

  mcmcgpu_handle native_handle(unsigned int g = 0) const { return hs.at(g); }
};

// slam_log_likelihood (src/slam/slam_likelihood.hpp:26-58) of ANY estimate -- the reference calls
// it on the averaged result and on g2o's (main.cpp:318,326) -- evaluated on the device: the
// estimate is loaded into a one-chain handle (mcmcgpu_load_state derives the log likelihood from
// scratch) and read back.  The estimate must cover every timestep and landmark of `data`.
template <class ControlModel, class ObservationModel>
double slam_log_likelihood(const slam_data<ControlModel, ObservationModel>& data,
                           const slam_result_of<ControlModel, ObservationModel>& estimate,
                           const gpu_capacity& cap = gpu_capacity()) {
  detail::device_problem<ControlModel, ObservationModel> problem;
  problem.absorb(data, std::size_t(data.current_timestep()));
  const auto& trajectory = estimate.get_trajectory();
  if (trajectory.size() != problem.T()) throw std::runtime_error("slam_log_likelihood: estimate and data differ in length");
  mcmcgpu_config cfg{};
  cfg.device = cap.device;
  cfg.max_chains = 1;
  cfg.max_timesteps = problem.T();
  cfg.max_features = problem.F();
  cfg.max_observations = problem.N();
  cfg.range_bearing = decltype(problem)::bearing ? 1 : 0;
  mcmcgpu_handle h = nullptr;
  if (mcmcgpu_create(&cfg, &h) != 0) throw std::runtime_error(std::string("mcmcgpu_create: ") + mcmcgpu_last_error(nullptr));
  struct closer {
    mcmcgpu_handle h;
    ~closer() { mcmcgpu_destroy(h); }
  } guard{h};
  problem.upload(h);
  const std::uint32_t seed = 0;  // the generator is never drawn from
  detail::check(h, mcmcgpu_init_chains(h, 1, &seed, nullptr, 0), "mcmcgpu_init_chains");
  std::vector<double> rel, world;
  for (std::size_t t = 0; t < trajectory.size(); ++t) {
    const auto d = trajectory[t];
    rel.insert(rel.end(), {d.x(), d.y(), d.bearing()});
  }
  const auto initial = estimate.get_initial_state();
  for (const std::size_t id : problem.feature_ids()) {
    const auto w = -initial + estimate.get_feature(featureid_type(id));
    world.insert(world.end(), {w.x(), w.y()});
  }
  detail::check(h, mcmcgpu_load_state(h, 0, 1, rel.data(), world.data()), "mcmcgpu_load_state");
  double ll = 0;
  detail::check(h, mcmcgpu_get_log_likelihood(h, &ll), "mcmcgpu_get_log_likelihood");
  return ll;
}

}  // namespace slam

#endif  // SLAM_AMD_HOST_MCMC_SLAM_HIP_HPP
