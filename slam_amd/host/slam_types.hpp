// Host-side value types with the reference's names and member signatures, free
// of Eigen/Boost, so that the GPU-backed sampler in mcmc_slam_hip.hpp can be
// compiled stand-alone (host/main.cpp) or, unchanged, against the reference's
// own headers (it only uses the members declared here).
//
// Reference counterparts (paths relative to the reference tree):
//   planar_robot::pose / position   src/planar_robot/pose.hpp, position.hpp
//   utility::bitree<G>              src/utility/bitree.hpp
//   slam::timestep_type, featureid_type, slam_result, dataset
//                                   src/slam/interfaces.hpp
//   velocity_slip_model / range_only_model (+ builders)
//                                   src/planar_robot/velocity_model.hpp, range_bearing_model.hpp
#ifndef SLAM_AMD_HOST_SLAM_TYPES_HPP
#define SLAM_AMD_HOST_SLAM_TYPES_HPP

#include <array>
#include <cmath>
#include <cstddef>
#include <map>
#include <vector>

namespace planar_robot {

class position {
  double px = 0, py = 0;

 public:
  position() = default;
  static position cartesian(double x, double y) {
    position p;
    p.px = x;
    p.py = y;
    return p;
  }
  static position polar(double dist, double dir) { return cartesian(dist * std::cos(dir), dist * std::sin(dir)); }
  double x() const { return px; }
  double y() const { return py; }
  double distance() const { return std::sqrt(px * px + py * py); }
  double direction() const { return std::atan2(py, px); }
};

class pose {
  double px = 0, py = 0, th = 0;

 public:
  pose() = default;
  static pose cartesian(double x, double y, double bearing) {
    pose p;
    p.px = x;
    p.py = y;
    p.th = bearing;
    return p;
  }
  double x() const { return px; }
  double y() const { return py; }
  double bearing() const { return th; }
  pose& operator+=(const pose& o) {  // compose: t += R(th) o.t, angles add unwrapped
    const double s = std::sin(th), c = std::cos(th);
    px += c * o.px - s * o.py;
    py += s * o.px + c * o.py;
    th += o.th;
    return *this;
  }
  pose operator-() const {
    const double s = std::sin(-th), c = std::cos(-th);
    return cartesian(c * -px - s * -py, s * -px + c * -py, -th);
  }
  friend pose operator+(pose a, const pose& b) { return a += b; }
  friend position operator+(const pose& p, const position& o) {
    const double s = std::sin(p.th), c = std::cos(p.th);
    return position::cartesian(p.px + (c * o.x() - s * o.y()), p.py + (s * o.x() + c * o.y()));
  }
};

// A fixed-size vector with the (i) element access the reference's Eigen vectors have.
template <int N>
struct vec {
  std::array<double, N> v{};
  double& operator()(int i) { return v[i]; }
  double operator()(int i) const { return v[i]; }
};

template <int N, class Assoc>
struct independent_normal {
  using vector_type = vec<N>;
  using associated_type = Assoc;
  static constexpr int vector_dim = N;
  vector_type m_mean, m_stddev;
  const vector_type& mean() const { return m_mean; }
  const vector_type& stddev() const { return m_stddev; }
};

struct velocity_slip_model : independent_normal<3, pose> {
  // builder: mean = dt*u, stddev = dt*sqrt((dt*A) u^2)   (velocity_model.hpp:157-181)
  struct builder {
    double A[3][3];
    builder(double a1, double a2, double a3, double a4, double a5, double a6)
        : A{{a1, a2, 0}, {a3, a4, 0}, {a5, a6, 0}} {}
    static builder defaults() {  // velocity_model.cpp:23-32
      const double d = 3.14159265358979323846 / 180;
      return builder(0.01, 0.0, 1.0 * d * d, 0.01, 0.01 * d * d, 0.001 * d);
    }
    velocity_slip_model operator()(const vector_type& control, double dt) const {
      velocity_slip_model m;
      for (int i = 0; i < 3; ++i) {
        m.m_mean(i) = dt * control(i);
        double acc = (dt * A[i][0]) * (control(0) * control(0));
        acc += (dt * A[i][1]) * (control(1) * control(1));
        acc += (dt * A[i][2]) * (control(2) * control(2));
        m.m_stddev(i) = dt * std::sqrt(acc);
      }
      return m;
    }
  };
};

struct range_only_model : independent_normal<1, position> {
  struct builder {  // range_bearing_model.hpp:180-195
    double stddev;
    explicit builder(double s) : stddev(s) {}
    range_only_model operator()(const vector_type& observation) const {
      range_only_model m;
      m.m_mean(0) = observation(0);
      m.m_stddev(0) = stddev;
      return m;
    }
  };
};

}  // namespace planar_robot

namespace utility {

// Binary indexed tree over a group (operator+, unary -, value-initialised
// identity): node i holds the composition of the values (parent(i), i], with
// parent(i) = i & (i-1).  Range composition and point update in O(log n).
template <class G>
class bitree {
  std::vector<G> node;
  static std::size_t up(std::size_t i) { return i & (i - 1); }
  static std::size_t smear(std::size_t i) {
    for (int s = 1; s < 64; s <<= 1) i |= i >> s;
    return i;
  }
  G tail(std::size_t i, std::size_t anc) const {  // v[anc+1] + ... + v[i]
    G r = G();
    while (i != anc) {
      r = node[i] + r;
      i = up(i);
    }
    return r;
  }
  G as_node(std::size_t i, const G& v) const { return i == 0 ? v : tail(i - 1, up(i)) + v; }

 public:
  std::size_t size() const { return node.size(); }
  bool empty() const { return node.empty(); }
  void clear() { node.clear(); }
  void reserve(std::size_t n) { node.reserve(n); }
  void push_back(const G& v) { node.push_back(as_node(node.size(), v)); }
  // appends the element whose prefix composition accumulate(size()+1) is `total`
  void push_back_accumulated(const G& total) {
    const std::size_t n = node.size();
    if (n == 0)
      node.push_back(total);
    else
      node.push_back(-accumulate(up(n) + 1) + total);
  }
  G get(std::size_t i) const { return i == 0 ? node[0] : -tail(i - 1, up(i)) + node[i]; }
  G operator[](std::size_t i) const { return get(i); }
  G accumulate(std::size_t b, std::size_t e) const {
    if (b == e) return G();
    if (b == 0) return node[0] + tail(e - 1, 0);
    if (e == 0) return -(node[0] + tail(b - 1, 0));
    const std::size_t anc = (b - 1) & ~smear((b - 1) ^ (e - 1));
    return -tail(b - 1, anc) + tail(e - 1, anc);
  }
  G accumulate(std::size_t e) const { return accumulate(0, e); }
  G accumulate() const { return accumulate(node.size()); }
  // index k with accumulate(k) <= x < accumulate(k+1)  (ordered groups)
  std::size_t binary_search(const G& x) const {
    if (node.empty() || x < node[0]) return 0;
    std::size_t lo = 0, hi = node.size();
    G origin = node[0];
    while (lo + 1 < hi) {
      const std::size_t last = hi - 1;
      const std::size_t mid = last & ~(smear(lo ^ last) >> 1);
      if (x < origin + node[mid])
        hi = mid;
      else {
        lo = mid;
        origin = origin + node[lo];
      }
    }
    return hi;
  }
};

}  // namespace utility

namespace slam {

struct timestep_type {
  explicit timestep_type(std::size_t t = 0) : timestep(t) {}
  operator std::size_t() const { return timestep; }
  timestep_type& operator++() {
    ++timestep;
    return *this;
  }

 private:
  std::size_t timestep;
};

struct featureid_type {
  explicit featureid_type(std::size_t f = 0) : featureid(f) {}
  operator std::size_t() const { return featureid; }

 private:
  std::size_t featureid;
};

struct timestep_listener {
  virtual void timestep(timestep_type) = 0;
  virtual void completed() {}
  virtual ~timestep_listener() = default;
};

struct data_source : public virtual timestep_listener {
  virtual timestep_type current_timestep() const = 0;
};

// interfaces.hpp:100-124
template <class State, class Feature>
struct slam_result : public data_source {
  using state_type = State;
  using feature_type = Feature;
  using trajectory_type = utility::bitree<state_type>;
  using feature_map_type = std::map<std::size_t, feature_type>;  // ordered by id, like the reference's flat_map
  virtual state_type get_initial_state() const { return state_type(); }
  virtual state_type get_state(timestep_type t) const { return get_initial_state() + get_trajectory().accumulate(t); }
  virtual feature_type get_feature(featureid_type id) const { return get_feature_map().at(id); }
  virtual const trajectory_type& get_trajectory() const = 0;
  virtual const feature_map_type& get_feature_map() const = 0;
};

// A result that owns its trajectory and map (slam_result_impl.hpp in the reference).
template <class State, class Feature>
class slam_result_impl : public slam_result<State, Feature> {
  using base = slam_result<State, Feature>;
  typename base::trajectory_type trajectory;
  typename base::feature_map_type map;

 public:
  void timestep(timestep_type) override {}
  timestep_type current_timestep() const override { return timestep_type(trajectory.size()); }
  typename base::trajectory_type& get_trajectory() { return trajectory; }
  typename base::feature_map_type& get_feature_map() { return map; }
  const typename base::trajectory_type& get_trajectory() const override { return trajectory; }
  const typename base::feature_map_type& get_feature_map() const override { return map; }
};

template <class ControlModel, class ObsModel>
using slam_result_of = slam_result<typename ControlModel::associated_type, typename ObsModel::associated_type>;

}  // namespace slam

#endif  // SLAM_AMD_HOST_SLAM_TYPES_HPP
