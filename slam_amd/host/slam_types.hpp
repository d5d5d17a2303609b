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
#include <cstdint>
#include <map>
#include <random>
#include <vector>

// src/utility/random.hpp:11-42: a std::mt19937 handed, as the generator, to
// std::uniform_real_distribution<> and std::normal_distribution<>.  The engine is
// fixed by the C++ standard; the two distributions are libstdc++'s, written out
// so that the stream does not depend on the standard library in use:
// generate_canonical<double, 53> takes two engine outputs, (lo + hi 2^32) / 2^64,
// and steps back below 1 if rounding reached it; the normal distribution is the
// Marsaglia polar method, which keeps its second variate for the next call.
class random_source {
  std::mt19937 engine;
  bool saved_available = false;
  double saved = 0.0;

 public:
  using result_type = std::mt19937::result_type;
  explicit random_source(unsigned int seed) : engine(seed) {}
  result_type operator()() { return engine(); }
  double uniform() {
    const double lo = double(engine()), hi = double(engine());
    double r = (lo + hi * 4294967296.0) / 18446744073709551616.0;
    if (r >= 1.0) r = std::nextafter(1.0, 0.0);
    return r;
  }
  double normal() {
    if (saved_available) {
      saved_available = false;
      return saved;
    }
    double x, y, r2;
    do {
      x = 2.0 * uniform() - 1.0;
      y = 2.0 * uniform() - 1.0;
      r2 = x * x + y * y;
    } while (r2 > 1.0 || r2 == 0.0);
    const double mult = std::sqrt(-2 * std::log(r2) / r2);
    saved = x * mult;
    saved_available = true;
    return y * mult;
  }
};

namespace planar_robot {

// Sine and cosine of one angle as ONE sincos call, which is what an optimising g++ makes of the
// pair anyway; glibc's sincos and its sin / cos differ by an ulp now and then, so the pairing is
// written out instead of left to the inliner.
inline void sin_cos(double a, double& s, double& c) { ::sincos(a, &s, &c); }

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
  static position polar(double dist, double dir) {
    double s, c;
    sin_cos(dir, s, c);
    return cartesian(dist * c, dist * s);
  }
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
    double s, c;
    sin_cos(th, s, c);
    px += c * o.px - s * o.py;
    py += s * o.px + c * o.py;
    th += o.th;
    return *this;
  }
  pose operator-() const {
    double s, c;
    sin_cos(-th, s, c);
    return cartesian(c * -px - s * -py, s * -px + c * -py, -th);
  }
  friend pose operator+(pose a, const pose& b) { return a += b; }
  friend position operator+(const pose& p, const position& o) {
    double s, c;
    sin_cos(p.th, s, c);
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
  // a draw: normals in index order, then scaled and shifted (src/utility/random.hpp:94-100)
  vector_type operator()(random_source& random) const {
    vector_type r;
    for (int i = 0; i < N; ++i) r(i) = random.normal();
    for (int i = 0; i < N; ++i) r(i) *= m_stddev(i);
    for (int i = 0; i < N; ++i) r(i) += m_mean(i);
    return r;
  }
};

// src/utility/geometry.hpp:8-16: into [-pi, pi)
inline double wrap_angle(double angle) {
  constexpr double pi = 3.14159265358979323846;
  angle = std::fmod(angle, 2.0 * pi);
  if (angle < -pi)
    angle += 2.0 * pi;
  else if (angle >= pi)
    angle -= 2.0 * pi;
  return angle;
}

struct velocity_slip_model : independent_normal<3, pose> {
  // velocity_model.hpp:37-49: pose change -> (arc length, turn, slip)
  static vector_type observe(const pose& dp) {
    const double r = 0.5 * (dp.x() * dp.x() + dp.y() * dp.y()) / dp.y();
    if (std::isfinite(r)) {
      const double alpha = r >= 0 ? std::atan2(dp.x(), r - dp.y()) : std::atan2(-dp.x(), dp.y() - r);
      return from_steering(r * alpha, alpha, wrap_angle(dp.bearing() - alpha));
    }
    return from_steering(dp.x(), 0.0, dp.bearing());
  }
  // velocity_model.hpp:51-63
  static pose inv_observe(const vector_type& control) {
    const double s = control(0), alpha = control(1), beta = control(2), r = s / alpha;
    if (std::isfinite(r)) {
      double sa, ca;
      sin_cos(alpha, sa, ca);
      return pose::cartesian(r * sa, r - r * ca, alpha + beta);
    }
    return pose::cartesian(s, 0.0, beta);
  }
  static vector_type from_steering(double s, double alpha, double beta = 0.0) {  // :126-129
    vector_type v;
    v(0) = s;
    v(1) = alpha;
    v(2) = beta;
    return v;
  }
  struct proposal_dist {  // :131-153
    const velocity_slip_model& model;
    pose operator()(random_source& random) const { return inv_observe(model(random)); }
  };
  proposal_dist proposal() const { return proposal_dist{*this}; }
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
  static vector_type observe(const position& p) {  // range_bearing_model.hpp:135-139
    vector_type v;
    v(0) = p.distance();
    return v;
  }
  // proposal_dist::initial_value (:171-175): on the measured circle, at a uniformly drawn direction
  position initial_value(random_source& random) const {
    return position::polar(m_mean(0), random.uniform() * 2 * 3.14159265358979323846);
  }
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

// range_bearing_model.hpp:21-117: range and direction of the landmark in the robot frame
struct range_bearing_model : independent_normal<2, position> {
  static vector_type observe(const position& p) {
    vector_type v;
    v(0) = p.distance();
    v(1) = p.direction();
    return v;
  }
  // proposal_dist::initial_value (:93-95): inv_observe of the mean, nothing drawn
  position initial_value(random_source&) const { return position::polar(m_mean(0), m_mean(1)); }
  struct builder {  // :100-116
    double stddev[2];
    builder(double range_stddev, double bearing_stddev) : stddev{range_stddev, bearing_stddev} {}
    range_bearing_model operator()(const vector_type& observation) const {
      range_bearing_model m;
      for (int i = 0; i < 2; ++i) {
        m.m_mean(i) = observation(i);
        m.m_stddev(i) = stddev[i];
      }
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
  State initial_state;
  typename base::trajectory_type trajectory;
  typename base::feature_map_type map;

 public:
  void timestep(timestep_type) override {}
  void set_initial_state(const State& s) { initial_state = s; }
  State get_initial_state() const override { return initial_state; }
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
