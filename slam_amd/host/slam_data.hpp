// Host-side data model around the sampler, with the reference's names and call
// sequence: the observer bus (slam_data), the raw dataset, the Plaza range-only
// text loader and the dead-reckoning initialiser.  Stand-alone (std only).
//
// Reference counterparts (paths relative to the reference tree):
//   slam::slam_data        src/slam/slam_data.hpp:28-215
//   slam::dataset_impl     src/slam/dataset_impl.hpp:20-85
//   read_range_only_data   src/dataset.cpp:17-112
//   slam::slam_initialiser src/slam/slam_initialiser.hpp:21-50
#ifndef SLAM_AMD_HOST_SLAM_DATA_HPP
#define SLAM_AMD_HOST_SLAM_DATA_HPP

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <memory>
#include <set>
#include <stdexcept>
#include <string>
#include <utility>

#include "../../include/mcmcgpu.h"
#include "slam_types.hpp"

namespace slam {

// Readings and controls as probability models, in arrival order, plus the
// listeners that are told about every new piece of data.
template <class ControlModel, class ObservationModel>
class slam_data : public data_source {
 public:
  using feature_observations = std::vector<std::pair<timestep_type, ObservationModel>>;  // time-sorted

  class observation_info {
    const slam_data* owner;
    std::size_t fid, i;

   public:
    observation_info(const slam_data* o, std::size_t id, std::size_t index) : owner(o), fid(id), i(index) {}
    featureid_type id() const { return featureid_type(fid); }
    std::size_t index() const { return i; }
    const ObservationModel& observation() const { return owner->m_features.at(fid)[i].second; }
  };

  struct listener : public virtual timestep_listener {
    virtual void control(timestep_type, const ControlModel&) = 0;
    virtual void observation(timestep_type, const observation_info&) = 0;
  };

  using observation_collection = std::vector<std::pair<timestep_type, observation_info>>;

 private:
  std::map<std::size_t, feature_observations> m_features;
  observation_collection m_observations;
  std::vector<ControlModel> m_controls;
  std::vector<std::weak_ptr<timestep_listener>> m_timestep_listeners;
  std::vector<std::weak_ptr<listener>> m_listeners;

 public:
  timestep_type current_timestep() const override { return timestep_type(m_controls.size()); }
  void timestep(timestep_type t) override {
    for (auto& w : m_timestep_listeners)
      if (auto l = w.lock()) l->timestep(t);
  }
  void completed() override {
    for (auto& w : m_timestep_listeners)
      if (auto l = w.lock()) l->completed();
  }
  const ControlModel& control(timestep_type t) const { return m_controls.at(t); }
  const observation_collection& observations() const { return m_observations; }
  // [first, last) positions in observations() of the readings taken at timestep t
  std::pair<std::size_t, std::size_t> observations_at(timestep_type t) const {
    auto lo = std::lower_bound(m_observations.begin(), m_observations.end(), std::size_t(t),
                               [](const auto& o, std::size_t v) { return std::size_t(o.first) < v; });
    auto hi = std::upper_bound(m_observations.begin(), m_observations.end(), std::size_t(t),
                               [](std::size_t v, const auto& o) { return v < std::size_t(o.first); });
    return {std::size_t(lo - m_observations.begin()), std::size_t(hi - m_observations.begin())};
  }
  bool feature_observed(featureid_type f) const { return m_features.count(f) != 0; }
  const feature_observations& get_observations(featureid_type f) const { return m_features.at(f); }

  void add_control(const ControlModel& c) {
    const timestep_type t = current_timestep();
    m_controls.push_back(c);
    for (auto& w : m_listeners)
      if (auto l = w.lock()) l->control(t, m_controls.back());
  }
  void add_observation(featureid_type id, const ObservationModel& obs) {
    const timestep_type t = current_timestep();
    auto& list = m_features[id];
    list.emplace_back(t, obs);
    m_observations.emplace_back(t, observation_info(this, id, list.size() - 1));
    for (auto& w : m_listeners)
      if (auto l = w.lock()) l->observation(t, m_observations.back().second);
  }
  void add_timestep_listener(const std::shared_ptr<timestep_listener>& l) { m_timestep_listeners.push_back(l); }
  void add_listener(const std::shared_ptr<listener>& l) {
    add_timestep_listener(l);
    m_listeners.push_back(l);
  }

  // Replays a dataset in the reference's order: readings of a timestep, then the
  // timestep signal, then the next control (slam_data.hpp:191-215).
  template <class Dataset>
  void add_dataset(const Dataset& data, const typename ControlModel::builder& cb,
                   const typename ObservationModel::builder& ob) {
    auto add_observations = [&](timestep_type t) {
      const auto r = data.observations_at(t);
      for (std::size_t i = r.first; i < r.second; ++i)
        add_observation(data.observations()[i].second.id, ob(data.observations()[i].second.observation));
    };
    add_observations(current_timestep());
    timestep(current_timestep());
    while (current_timestep() < data.current_timestep()) {
      const timestep_type t = current_timestep();
      add_control(cb(data.control(t), data.timedelta(t)));
      add_observations(current_timestep());
      timestep(current_timestep());
    }
    completed();
  }
};

// Raw controls (v, w, slip rate), time deltas and range readings of a run.
template <class ControlModel, class ObservationModel>
class dataset_impl {
 public:
  using control_type = typename ControlModel::vector_type;
  using observation_type = typename ObservationModel::vector_type;
  struct observation_info {
    featureid_type id;
    observation_type observation;
  };
  using observation_collection = std::vector<std::pair<timestep_type, observation_info>>;

 private:
  std::vector<control_type> m_controls;
  utility::bitree<double> m_timedeltas;
  observation_collection m_observations;

 public:
  timestep_type current_timestep() const { return timestep_type(m_controls.size()); }
  const control_type& control(timestep_type t) const { return m_controls.at(t); }
  double timedelta(timestep_type t) const { return m_timedeltas[t]; }
  double timestamp(timestep_type t) const { return m_timedeltas.accumulate(t); }
  timestep_type timestep_at(double timestamp) const { return timestep_type(m_timedeltas.binary_search(timestamp)); }
  const observation_collection& observations() const { return m_observations; }
  std::pair<std::size_t, std::size_t> observations_at(timestep_type t) const {
    auto lo = std::lower_bound(m_observations.begin(), m_observations.end(), std::size_t(t),
                               [](const auto& o, std::size_t v) { return std::size_t(o.first) < v; });
    auto hi = std::upper_bound(m_observations.begin(), m_observations.end(), std::size_t(t),
                               [](std::size_t v, const auto& o) { return v < std::size_t(o.first); });
    return {std::size_t(lo - m_observations.begin()), std::size_t(hi - m_observations.begin())};
  }
  void add_control(double dt, const control_type& c) {
    m_timedeltas.push_back(dt);
    m_controls.push_back(c);
  }
  // for input that is already in arrival order
  void append_observation(timestep_type t, featureid_type id, observation_type obs) {
    m_observations.emplace_back(t, observation_info{id, obs});
  }
  // keeps the collection sorted by timestep; equal timesteps stay in insertion order
  void add_observation(timestep_type t, featureid_type id, observation_type obs) {
    auto at = std::upper_bound(m_observations.begin(), m_observations.end(), std::size_t(t),
                               [](std::size_t v, const auto& o) { return v < std::size_t(o.first); });
    m_observations.insert(at, std::make_pair(t, observation_info{id, obs}));
  }
};

using range_only_dataset = dataset_impl<planar_robot::velocity_slip_model, planar_robot::range_only_model>;

using range_only_ground_truth = slam_result_impl<planar_robot::pose, planar_robot::position>;

// CMU Plaza range-only logs: <name>_GT.txt (time x y heading; the first row gives
// the start time and the initial pose), <name>_DR.txt (time, distance and heading
// increments; control = increment / dt, third component 0), <name>_TD.txt (time
// antenna beacon range; the timestep is found from the time since start, the first
// reading per (timestep, beacon) wins), <name>_TL.txt (beacon id x y).  The sampler
// needs only the dataset; `ground_truth`, when given, receives what the reference's
// loader builds beside it (src/dataset.cpp:30-58): the GPS track relative to its first
// pose and the surveyed beacon positions.
inline std::shared_ptr<range_only_dataset> read_range_only_data(
    const std::string& dir, const std::string& name, std::shared_ptr<range_only_ground_truth>* ground_truth = nullptr) {
  auto open = [&](const char* suffix) {
    const std::string path = dir + "/" + name + "/" + name + suffix;
    std::ifstream f(path);
    if (!f) throw std::runtime_error("cannot open " + path);
    return f;
  };
  auto out = std::make_shared<range_only_dataset>();
  double start;
  {
    auto gt = open("_GT.txt");
    double t, x, y, b;
    if (!(gt >> start >> x >> y >> b)) throw std::runtime_error(name + "_GT.txt is empty");
    if (ground_truth) {
      auto truth = std::make_shared<range_only_ground_truth>();
      const auto first = planar_robot::pose::cartesian(x, y, b);
      truth->set_initial_state(first);
      while (gt >> t >> x >> y >> b)
        truth->get_trajectory().push_back_accumulated(-first + planar_robot::pose::cartesian(x, y, b));
      auto tl = open("_TL.txt");
      double id;
      while (tl >> id >> x >> y)
        truth->get_feature_map().emplace(std::size_t(std::round(id)), planar_robot::position::cartesian(x, y));
      *ground_truth = truth;
    }
  }
  {
    auto dr = open("_DR.txt");
    double stamp, prev = start;
    range_only_dataset::control_type c;
    while (dr >> stamp >> c(0) >> c(1)) {
      const double dt = stamp - prev;
      prev = stamp;
      range_only_dataset::control_type u;
      for (int i = 0; i < 3; ++i) u(i) = c(i) / dt;
      out->add_control(dt, u);
    }
  }
  {
    auto td = open("_TD.txt");
    double stamp, antenna, id;
    range_only_dataset::observation_type r;
    std::set<std::pair<std::size_t, std::size_t>> seen;
    while (td >> stamp >> antenna >> id >> r(0)) {
      const featureid_type fid(std::size_t(std::round(id)));
      const timestep_type t = out->timestep_at(stamp - start);
      if (seen.emplace(std::size_t(t), std::size_t(fid)).second) out->add_observation(t, fid, r);
    }
  }
  return out;
}

// The table the reference's loader leaves next to the logs as controls.txt (src/dataset.cpp:96-109),
// written to a path of the caller's choice instead: per timestep dt, the control that the ground-truth
// pose change amounts to (velocity_slip_model::observe / dt) and the commanded one.
inline void write_controls_txt(const range_only_dataset& data, const range_only_ground_truth& truth,
                               const std::string& path) {
  std::ofstream controls(path);
  if (!controls) throw std::runtime_error("cannot write " + path);
  controls << "# dt, actual v, actual w, actual g, commanded v, commanded w\n";
  for (timestep_type t{0}; t < data.current_timestep(); ++t) {
    const double dt = data.timedelta(t);
    const planar_robot::pose pose_delta = truth.get_trajectory()[t];
    auto actual = planar_robot::velocity_slip_model::observe(pose_delta);
    for (int i = 0; i < 3; ++i) actual(i) /= dt;
    const auto& commanded = data.control(t);
    controls << dt << ' ' << actual(0) << ' ' << actual(1) << ' ' << actual(2) << ' ' << commanded(0) << ' '
             << commanded(1) << '\n';
  }
}

// Binary cache of a range-only dataset: what read_range_only_data produces, as
// flat little-endian arrays that load with five reads instead of a text parse
// (a million-pose log is 30 MB of arrays against hundreds of MB of text).
//   bytes 0..7   "SLAMRO1\0"
//   u32 version (1), u32 control dimension (3), u32 reading dimension (1), u32 0
//   u64 T, u64 N, 24 zero bytes                                  -- 64-byte header
//   f64 dt[T]; f64 control[3][T]  (component-major)
//   u32 timestep[N]; u32 landmark id[N]; each padded to a multiple of 8 bytes
//   f64 range[N]                                                 -- arrival order
namespace dataset_cache_detail {
constexpr char kMagic[8] = {'S', 'L', 'A', 'M', 'R', 'O', '1', '\0'};
struct header {
  char magic[8];
  std::uint32_t version, control_dim, reading_dim, zero;
  std::uint64_t T, N;
  char reserved[24];
};
static_assert(sizeof(header) == 64, "cache header is 64 bytes");
}  // namespace dataset_cache_detail

inline void write_dataset_cache(const range_only_dataset& data, const std::string& path) {
  using namespace dataset_cache_detail;
  const std::size_t T = data.current_timestep(), N = data.observations().size();
  header h{};
  std::memcpy(h.magic, kMagic, 8);
  h.version = 1;
  h.control_dim = 3;
  h.reading_dim = 1;
  h.T = T;
  h.N = N;
  std::vector<double> dt(T), ctrl(3 * T), range(N);
  std::vector<std::uint32_t> ot(N + (N & 1), 0), oid(N + (N & 1), 0);
  for (std::size_t t = 0; t < T; ++t) {
    dt[t] = data.timedelta(timestep_type(t));
    for (int i = 0; i < 3; ++i) ctrl[std::size_t(i) * T + t] = data.control(timestep_type(t))(i);
  }
  for (std::size_t i = 0; i < N; ++i) {
    ot[i] = std::uint32_t(std::size_t(data.observations()[i].first));
    oid[i] = std::uint32_t(std::size_t(data.observations()[i].second.id));
    range[i] = data.observations()[i].second.observation(0);
  }
  std::ofstream f(path, std::ios::binary | std::ios::trunc);
  if (!f) throw std::runtime_error("cannot write " + path);
  f.write(reinterpret_cast<const char*>(&h), sizeof h);
  f.write(reinterpret_cast<const char*>(dt.data()), std::streamsize(dt.size() * sizeof(double)));
  f.write(reinterpret_cast<const char*>(ctrl.data()), std::streamsize(ctrl.size() * sizeof(double)));
  f.write(reinterpret_cast<const char*>(ot.data()), std::streamsize(ot.size() * sizeof(std::uint32_t)));
  f.write(reinterpret_cast<const char*>(oid.data()), std::streamsize(oid.size() * sizeof(std::uint32_t)));
  f.write(reinterpret_cast<const char*>(range.data()), std::streamsize(range.size() * sizeof(double)));
  if (!f) throw std::runtime_error("short write to " + path);
}

inline std::shared_ptr<range_only_dataset> read_dataset_cache(const std::string& path) {
  using namespace dataset_cache_detail;
  std::ifstream f(path, std::ios::binary);
  if (!f) throw std::runtime_error("cannot open " + path);
  header h{};
  f.read(reinterpret_cast<char*>(&h), sizeof h);
  if (!f || std::memcmp(h.magic, kMagic, 8) != 0 || h.version != 1 || h.control_dim != 3 || h.reading_dim != 1)
    throw std::runtime_error(path + " is not a version-1 range-only dataset cache");
  const std::size_t T = h.T, N = h.N;
  std::vector<double> dt(T), ctrl(3 * T), range(N);
  std::vector<std::uint32_t> ot(N + (N & 1)), oid(N + (N & 1));
  f.read(reinterpret_cast<char*>(dt.data()), std::streamsize(dt.size() * sizeof(double)));
  f.read(reinterpret_cast<char*>(ctrl.data()), std::streamsize(ctrl.size() * sizeof(double)));
  f.read(reinterpret_cast<char*>(ot.data()), std::streamsize(ot.size() * sizeof(std::uint32_t)));
  f.read(reinterpret_cast<char*>(oid.data()), std::streamsize(oid.size() * sizeof(std::uint32_t)));
  f.read(reinterpret_cast<char*>(range.data()), std::streamsize(range.size() * sizeof(double)));
  if (!f) throw std::runtime_error(path + " is truncated");
  auto out = std::make_shared<range_only_dataset>();
  for (std::size_t t = 0; t < T; ++t) {
    range_only_dataset::control_type u;
    for (int i = 0; i < 3; ++i) u(i) = ctrl[std::size_t(i) * T + t];
    out->add_control(dt[t], u);
  }
  for (std::size_t i = 0; i < N; ++i) {
    if (ot[i] > T || (i && ot[i] < ot[i - 1])) throw std::runtime_error(path + ": readings out of order");
    range_only_dataset::observation_type r;
    r(0) = range[i];
    out->append_observation(timestep_type(ot[i]), featureid_type(oid[i]), r);
  }
  return out;
}

// Dead reckoning plus a first guess for every new landmark: the estimate a
// sampler with an initialiser starts every new edge from.  Draws one uniform
// per new landmark from its own generator (the library's host generator is
// the reference's random_source: mcmcgpu_uniform_sequence).
template <class ControlModel, class ObservationModel>
class slam_initialiser : public slam_data<ControlModel, ObservationModel>::listener,
                         public slam_result_of<ControlModel, ObservationModel> {
  using data_type = slam_data<ControlModel, ObservationModel>;
  using result_type = slam_result_of<ControlModel, ObservationModel>;
  unsigned int m_seed;
  random_source m_random;
  typename result_type::trajectory_type m_trajectory;
  typename result_type::feature_map_type m_map;


 public:
  explicit slam_initialiser(unsigned int seed) : m_seed(seed), m_random(seed) {}
  unsigned int seed() const { return m_seed; }
  void timestep(timestep_type) override {}
  void control(timestep_type, const ControlModel& c) override {
    // inv_observe of the control mean (velocity_model.hpp:51-63, :150-152)
    m_trajectory.push_back(ControlModel::inv_observe(c.mean()));
  }
  void observation(timestep_type t, const typename data_type::observation_info& o) override {
    if (o.index() == 0) m_map[o.id()] = this->get_state(t) + o.observation().initial_value(m_random);
  }
  timestep_type current_timestep() const override { return timestep_type(m_trajectory.size()); }
  const typename result_type::trajectory_type& get_trajectory() const override { return m_trajectory; }
  const typename result_type::feature_map_type& get_feature_map() const override { return m_map; }
};

}  // namespace slam

#endif  // SLAM_AMD_HOST_SLAM_DATA_HPP
