// Device-side scalar building blocks of the MCMC-SLAM sampler for gfx950:
// SE(2) poses, the two sensor/motion models, the weighted edge tree and the
// Mersenne-Twister stream.  Everything here is straight-line fp64 code meant
// to run in a single lane (the per-chain serial part of a Metropolis-Hastings
// step); the data-parallel parts live in mcmcgpu.hip.
//
// Reference behaviour reproduced (paths relative to the reference tree):
//   src/planar_robot/pose.hpp:67-87, position.hpp:27-35   poses and points
//   src/planar_robot/velocity_model.hpp:37-63             control <-> pose
//   src/planar_robot/range_bearing_model.hpp:135-176      range-only model
//   src/utility/random.hpp:94-121,250-264                 normal sampling / log pdf
//   src/utility/geometry.hpp:8-16                         wrap_angle
//   src/utility/bitree.hpp:17-48,219-321                  weight tree on doubles
//   libstdc++ 11 <random> (mt19937, generate_canonical, normal_distribution)
#ifndef SLAM_AMD_DEVICE_MATH_HPP
#define SLAM_AMD_DEVICE_MATH_HPP

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mg {

#define MG_DEV __host__ __device__ __forceinline__
#define MG_DEVN __host__ __device__ __noinline__

constexpr double kPi = 3.14159265358979323846264338327950288;
constexpr double kTwoPi = 2.0 * kPi;

struct Vec2 {
  double x, y;
};

struct Pose {
  double x, y, th;
};

MG_DEV Pose pose_identity() { return Pose{0.0, 0.0, 0.0}; }

// a then b:  t = a.t + R(a.th) b.t,  th = a.th + b.th  (angles are never wrapped)
MG_DEV Pose compose(const Pose& a, const Pose& b) {
  double s, c;
  sincos(a.th, &s, &c);
  return Pose{a.x + (c * b.x - s * b.y), a.y + (s * b.x + c * b.y), a.th + b.th};
}

MG_DEV Pose inverse(const Pose& a) {
  double s, c;
  sincos(-a.th, &s, &c);
  const double nx = -a.x, ny = -a.y;
  return Pose{c * nx - s * ny, s * nx + c * ny, -a.th};
}

// p acting on a point:  t + R(th) v
MG_DEV Vec2 act(const Pose& p, const Vec2& v) {
  double s, c;
  sincos(p.th, &s, &c);
  return Vec2{p.x + (c * v.x - s * v.y), p.y + (s * v.x + c * v.y)};
}

// inverse of p acting on a point:  R(-th) (v - t)
MG_DEV Vec2 act_inverse(const Pose& p, const Vec2& v) {
  double s, c;
  sincos(p.th, &s, &c);
  const double dx = v.x - p.x, dy = v.y - p.y;
  return Vec2{c * dx + s * dy, c * dy - s * dx};
}

// A pose together with the cosine and sine of its angle: what act / act_inverse above compute
// first, kept so that transforming many points by one pose takes the pair once.  Same arithmetic.
struct Frame {
  double x, y, c, s;
};
MG_DEV Frame frame_of(const Pose& p) {
  Frame f{p.x, p.y, 0.0, 0.0};
  sincos(p.th, &f.s, &f.c);
  return f;
}
MG_DEV Vec2 act(const Frame& p, const Vec2& v) { return Vec2{p.x + (p.c * v.x - p.s * v.y), p.y + (p.s * v.x + p.c * v.y)}; }
MG_DEV Vec2 act_inverse(const Frame& p, const Vec2& v) {
  const double dx = v.x - p.x, dy = v.y - p.y;
  return Vec2{p.c * dx + p.s * dy, p.c * dy - p.s * dx};
}
// inverse(q) as a frame, from the cosine and sine of -q.th
MG_DEV Frame inverse_frame(const Pose& q, double c, double s) {
  const double nx = -q.x, ny = -q.y;
  return Frame{c * nx - s * ny, s * nx + c * ny, c, s};
}

MG_DEV double wrap_angle(double a) {
  // fmod leaves an argument below the modulus as it is; almost every angle here is one, and the
  // general fmod is a long loop on this hardware
  if (!(fabs(a) < kTwoPi)) a = fmod(a, kTwoPi);
  if (a < -kPi)
    a += kTwoPi;
  else if (a >= kPi)
    a -= kTwoPi;
  return a;
}

// ---- velocity-slip control model -------------------------------------------
// pose increment -> (arc length, turn, slip)
MG_DEV void control_observe(const Pose& dp, double out[3]) {
  const double r = 0.5 * (dp.x * dp.x + dp.y * dp.y) / dp.y;
  if (isfinite(r)) {
    const double alpha = r >= 0 ? atan2(dp.x, r - dp.y) : atan2(-dp.x, dp.y - r);
    out[0] = r * alpha;
    out[1] = alpha;
    out[2] = wrap_angle(dp.th - alpha);
  } else {
    out[0] = dp.x;
    out[1] = 0.0;
    out[2] = dp.th;
  }
}

MG_DEV Pose control_inv_observe(const double v[3]) {
  const double s = v[0], alpha = v[1], beta = v[2], r = s / alpha;
  if (isfinite(r)) {
    double sn, cs;
    sincos(alpha, &sn, &cs);  // one argument reduction for the pair
    return Pose{r * sn, r - r * cs, alpha + beta};
  }
  return Pose{s, 0.0, beta};
}

// log pdf of an independent 3-normal, constants subtracted term by term
// (log_det = log sd[0] + log sd[1] + log sd[2], which the caller keeps per control)
MG_DEV double normal3_log_pdf(const double mu[3], const double sd[3], double log_det, const double x[3]) {
  const double z0 = (x[0] - mu[0]) / sd[0];
  const double z1 = (x[1] - mu[1]) / sd[1];
  const double z2 = (x[2] - mu[2]) / sd[2];
  const double exponent = -0.5 * (z0 * z0 + z1 * z1 + z2 * z2);
  const double log_root_two_pi = 0.5 * log(2 * kPi);
  return exponent - 3 * log_root_two_pi - log_det;
}

// log pdf of the control model at a pose increment
MG_DEV double control_log_pdf(const double mu[3], const double sd[3], double log_det, const Pose& dp) {
  double v[3];
  control_observe(dp, v);
  return normal3_log_pdf(mu, sd, log_det, v);
}
MG_DEV double control_log_det(double sd0, double sd1, double sd2) { return log(sd0) + log(sd1) + log(sd2); }

// ---- range-only observation model ------------------------------------------
// (log_sd = log(sd): a reading's sigma does not change, so its logarithm is taken once and kept)
MG_DEV double range_log_pdf(double mu, double sd, double log_sd, double r) {
  const double z = (r - mu) / sd;
  const double log_root_two_pi = 0.5 * log(2 * kPi);
  return -0.5 * (z * z) - log_root_two_pi - log_sd;
}

MG_DEV double vec_norm(const Vec2& v) { return sqrt(v.x * v.x + v.y * v.y); }

// log pdf of the landmark proposal (range normal x uniform direction)
MG_DEV double range_proposal_log_pdf(double mu, double sd, double log_sd, const Vec2& p) {
  return range_log_pdf(mu, sd, log_sd, vec_norm(p)) - log(2 * kPi);
}

MG_DEV Vec2 polar(double dist, double dir) {
  double sn, cs;
  sincos(dir, &sn, &cs);
  return Vec2{dist * cs, dist * sn};
}

// ---- one landmark reading as a distribution ----------------------------------
// Both observation models of src/planar_robot/range_bearing_model.hpp behind a
// run-time switch: range_only_model (:119-196) and, when `bearing` is set,
// range_bearing_model (:21-117): observe = (|p|, atan2(y, x)), the bearing
// residual is wrapped (:33-36), independent normal in both components.
struct Reading {
  double mu, sd;    // range
  double bmu, bsd;  // bearing
  bool bearing;
  double log_sd, log_bsd;  // log(sd), log(bsd): the same library call on the same bits, taken when the reading is loaded
};

// log_likelihood(observe(p)), src/utility/random.hpp:103-121
MG_DEV double reading_log_pdf(const Reading& m, const Vec2& p) {
  if (!m.bearing) return range_log_pdf(m.mu, m.sd, m.log_sd, vec_norm(p));
  const double log_root_two_pi = 0.5 * log(2 * kPi);
  const double x0 = (vec_norm(p) - m.mu) / m.sd;
  const double x1 = wrap_angle(atan2(p.y, p.x) - m.bmu) / m.bsd;
  return -0.5 * (x0 * x0 + x1 * x1) - 2 * log_root_two_pi - (m.log_sd + m.log_bsd);
}

// proposal_dist::log_likelihood: :166-169 (range only: a uniform direction on top) and :90-92
MG_DEV double reading_proposal_log_pdf(const Reading& m, const Vec2& p) {
  return m.bearing ? reading_log_pdf(m, p) : range_log_pdf(m.mu, m.sd, m.log_sd, vec_norm(p)) - log(2 * kPi);
}

MG_DEV double edge_weight_from_log_pdf(double log_pdf, double dim, double* log_weight) {
  const double lw = log(dim) - log_pdf / dim;
  if (log_weight) *log_weight = lw;
  return exp(lw);
}

// ---- weight tree: Fenwick layout over doubles, bit-reproducible --------------
// node[i] holds the sum of the values (parent(i), i], parent(i) = i & (i-1).
// Only additions and negations are used, in the reference's exact order, so
// totals and searched indices are identical to the CPU for identical weights.
// The serial part runs in one lane, or in all lanes of a wavefront on the same data.  Either way its
// control flow is the same in every active lane, which the compiler cannot see once a value has come
// out of memory: it then guards every branch with a lane mask and keeps indices in vector registers.
// A ballot / first-lane read says it outright, and loop counters, indices and branches become scalar.
#if defined(__HIP_DEVICE_COMPILE__)
MG_DEV bool uniformly(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0; }
MG_DEV uint32_t uniform_u32(uint32_t v) { return uint32_t(__builtin_amdgcn_readfirstlane(int(v))); }
#else
MG_DEV bool uniformly(bool p) { return p; }
MG_DEV uint32_t uniform_u32(uint32_t v) { return v; }
#endif

namespace wt {
MG_DEV uint32_t parent(uint32_t i) { return i & (i - 1); }
MG_DEV uint32_t next_sibling(uint32_t i) { return i + (i & (0u - i)); }
MG_DEV uint32_t suffix_mask(uint32_t i) {
  i |= i >> 1;
  i |= i >> 2;
  i |= i >> 4;
  i |= i >> 8;
  i |= i >> 16;
  return i;
}
MG_DEV uint32_t child_containing(uint32_t p, uint32_t i) {
#if defined(__HIP_DEVICE_COMPILE__)
  return i & ~((0x7FFFFFFFu >> __clz(int(p ^ i))));  // suffix_mask(x) >> 1 of a non-zero x
#else
  return i & ~(suffix_mask(p ^ i) >> 1);
#endif
}

// Where a tree's nodes live.  Plain: an array in global memory.  Cached: the same array, with a copy
// in LDS of every node whose index is a multiple of 2^shift -- the nodes a descent probes first
// and a point update rewrites last -- kept current by writing through.  Both hand out the same
// values, so the arithmetic (and with it every searched index) does not depend on the choice.
struct Plain {
  double* g;
  MG_DEV double ld(uint32_t i) const { return g[i]; }
  MG_DEV void st(uint32_t i, double v) const { g[i] = v; }
  MG_DEV void warm(uint32_t, uint32_t, uint32_t) const {}
};
using LdsDoubles = __attribute__((address_space(3))) double*;
struct Cached {
  double* g;
  uint32_t top;  // LDS byte address of the copy; 0: no copy (LDS address 0 belongs to static data)
  uint32_t shift;
  MG_DEV LdsDoubles lds() const { return (LdsDoubles)(uintptr_t)top; }
  MG_DEV bool held(uint32_t i) const { return top && (i & ((1u << shift) - 1u)) == 0; }
  MG_DEV double ld(uint32_t i) const { return held(i) ? lds()[i >> shift] : g[i]; }
  MG_DEV void st(uint32_t i, double v) const {
    g[i] = v;
    if (held(i)) lds()[i >> shift] = v;
  }
  // A descent has narrowed its range to [lo, hi) and is about to probe `mid`, which is not held in
  // LDS: every node it can still probe lies in the 2^shift nodes from lo on.  Their cache lines
  // (16 nodes each) are requested together, so the dependent probes that follow find them near.
  MG_DEV void warm(uint32_t lo, uint32_t n, uint32_t mid) const {
#if defined(__HIP_DEVICE_COMPILE__)
    if (!top || shift < 5 || mid - lo != (1u << (shift - 1)) || (lo & ((1u << shift) - 1u)) != 0) return;
    double sum = 0.0;
    for (uint32_t j = 16; j < (1u << shift); j += 16)
      if (lo + j < n) sum += g[lo + j];
    asm volatile("" ::"v"(sum));
#endif
  }
};

// node(i) + (node(parent(i)) + ( ... )) up to, not including, `ancestor`.  Any lane for itself.
template <class N>
MG_DEV double sum_to_ancestor(const N& node, uint32_t i, uint32_t ancestor) {
  double result = 0.0;
  for (; i != ancestor; i = parent(i)) result = node.ld(i) + result;
  return result;
}
// The same for the serial program (every active lane on the same arguments): added in the same order,
// but asked for four at a time (which nodes they are is bit arithmetic on i alone), scalar indices.
template <class N>
MG_DEV double sum_to_ancestor_serial(const N& node, uint32_t i_in, uint32_t ancestor_in) {
  uint32_t i = uniform_u32(i_in);
  const uint32_t ancestor = uniform_u32(ancestor_in);
  double result = 0.0;
  while (i != ancestor) {
    const uint32_t j1 = parent(i);
    const bool v1 = j1 != ancestor;
    const uint32_t j2 = v1 ? parent(j1) : ancestor;
    const bool v2 = j2 != ancestor;
    const uint32_t j3 = v2 ? parent(j2) : ancestor;
    const bool v3 = j3 != ancestor;
    const double a0 = node.ld(i);
    const double a1 = v1 ? node.ld(j1) : 0.0;
    const double a2 = v2 ? node.ld(j2) : 0.0;
    const double a3 = v3 ? node.ld(j3) : 0.0;
    result = a0 + result;
    if (v1) result = a1 + result;
    if (v2) result = a2 + result;
    if (v3) result = a3 + result;
    i = v3 ? parent(j3) : ancestor;
  }
  return result;
}
template <class N>
MG_DEV double total(const N& node, uint32_t n) {
  if (n == 0) return 0.0;
  return node.ld(0) + sum_to_ancestor_serial(node, n - 1, 0);
}
template <class N>
MG_DEV double node_value_for(const N& node, uint32_t i, double v) {
  if (i == 0) return v;
  return sum_to_ancestor_serial(node, i - 1, parent(i)) + v;
}
template <class N>
MG_DEV double get(const N& node, uint32_t i) {
  if (i == 0) return node.ld(0);
  return -sum_to_ancestor(node, i - 1, parent(i)) + node.ld(i);
}
template <class N>
MG_DEV void push_back(const N& node, uint32_t n, double v) {
  node.st(n, node_value_for(node, n, v));
}
template <class N>
MG_DEV void set(const N& node, uint32_t n, uint32_t i, double v) {
  double new_value = node_value_for(node, i, v);
  if (i > 0)
    for (uint32_t next; (next = next_sibling(i)) < n; i = next) {
      const double old = node.ld(i);
      double delta = new_value + (-old);
      node.st(i, new_value);
      const uint32_t next_parent = parent(next);
      while ((i = parent(i)) != next_parent) {
        const double a = node.ld(i);
        delta = a + delta + (-a);
      }
      new_value = delta + node.ld(next);
    }
  node.st(i, new_value);
}
// index k with prefix(k) <= value < prefix(k+1); may return n when value >= total.
// The reference's descent (bitree.hpp:302-321), comparison for comparison and addition for addition, two
// levels per round: the node the next comparison asks for and the two that the one after it can ask for
// are requested together, so a round waits for memory once; indices and branches are scalar.
template <class Load, class Warm>
MG_DEV uint32_t search_by(Load ld, Warm warm, uint32_t n_in, double value) {
  const uint32_t n = uniform_u32(n_in);
  if (n == 0) return 0;
  double origin = ld(0u);
  if (uniformly(value < origin)) return 0;
  uint32_t lo = 0, hi = n;
  while (lo + 1 < hi) {
    const uint32_t mid = child_containing(lo, hi - 1);
    const bool open_l = lo + 1 < mid, open_r = mid + 1 < hi;
    const uint32_t mid_l = open_l ? child_containing(lo, mid - 1) : mid;
    const uint32_t mid_r = open_r ? child_containing(mid, hi - 1) : mid;
    warm(lo, n, mid);
    const double at_mid = ld(mid), at_l = ld(mid_l), at_r = ld(mid_r);
#if defined(__HIP_DEVICE_COMPILE__)
    // all three are wanted HERE: left alone, the compiler moves the second level's loads behind the first
    // comparison (each is used on one side of it only) and the round waits for memory twice
    asm volatile("" ::"v"(at_mid), "v"(at_l), "v"(at_r));
#endif
    const double s = origin + at_mid;
    if (uniformly(value < s)) {
      hi = mid;
      if (open_l) {
        const double sl = origin + at_l;
        if (uniformly(value < sl)) {
          hi = mid_l;
        } else {
          lo = mid_l;
          origin = sl;
        }
      }
    } else {
      lo = mid;
      origin = s;
      if (open_r) {
        const double sr = s + at_r;
        if (uniformly(value < sr)) {
          hi = mid_r;
        } else {
          lo = mid_r;
          origin = sr;
        }
      }
    }
  }
  return hi;
}
template <class N>
MG_DEV uint32_t search(const N& node, uint32_t n, double value) {
  return search_by([&](uint32_t i) { return node.ld(i); }, [&](uint32_t lo, uint32_t nn, uint32_t mid) { node.warm(lo, nn, mid); },
                   n, value);
}

#if defined(__HIP_DEVICE_COMPILE__)
// a tree whose every node is in LDS (copy at stride 1): no choice to make per node
MG_DEV uint32_t search(const Cached& node, uint32_t n, double value) {
  if (uniformly(node.top != 0 && node.shift == 0)) {
    const LdsDoubles all = node.lds();
    return search_by([&](uint32_t i) { return all[i]; }, [](uint32_t, uint32_t, uint32_t) {}, n, value);
  }
  return search_by([&](uint32_t i) { return node.ld(i); }, [&](uint32_t lo, uint32_t nn, uint32_t mid) { node.warm(lo, nn, mid); },
                   n, value);
}
#endif

#if defined(__HIP_DEVICE_COMPILE__)
MG_DEV double readlane_f64(double v, int lane) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
// search() for a whole wavefront whose 64 lanes all hold the same arguments (and get the same
// result).  The descent's comparisons are sequential, but the nodes the next six of them can
// probe form a binary tree of 63: lane L loads the node at heap position L of that tree (root 1,
// the node after a "left" at 2L, after a "right" at 2L + 1), all loads go out together, and the
// comparisons then walk that tree with v_readlane -- the serial code's additions and comparisons,
// operation for operation, with one memory latency per six levels.  Once the range [lo, hi) is a
// power of two wide and aligned to its width (after the first "left", and so almost always from
// the second round on) the node under a heap position is arithmetic: lo + (hi - lo) (2p + 1) /
// 2^(d+1) for depth d and path p; before that every lane replays its path with child_containing.
template <class N>
MG_DEV uint32_t search_wave(const N& node, uint32_t n, double value) {
  if (n == 0 || value < node.ld(0)) return 0;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t d = 31u - uint32_t(__clz(int(lane | 1u)));  // depth of this lane's heap position (lane 0: unused)
  const uint32_t path = lane & ((1u << d) - 1u);             // its decisions, the first in the top bit
  constexpr uint32_t kNone = 0xFFFFFFFFu;
  uint32_t lo = 0, hi = n;
  double origin = node.ld(0);
  // (a 32-bit index has at most 32 levels: the bound only keeps corrupted arguments from spinning)
  for (int round = 0; round < 7 && lo + 1 < hi; ++round) {
    uint32_t mine = kNone;
    const uint32_t w = hi - lo;
    if ((w & (w - 1u)) == 0 && (lo & (w - 1u)) == 0) {  // same in all lanes
      const uint32_t step = w >> (d + 1u);
      if (step) mine = lo + step * (2u * path + 1u);
    } else {
      uint32_t l = lo, h = hi;
#pragma unroll
      for (uint32_t k = 0; k < 6; ++k) {
        const bool open = l + 1 < h;
        const uint32_t last = h - 1;
        // child_containing(l, last); "| 1" keeps the count defined when the descent has ended (l == last)
        const uint32_t mid = last & ~(0x7FFFFFFFu >> __clz(int((l ^ last) | 1u)));
        if (k == d) mine = open ? mid : kNone;
        if (k < d) {
          const bool right = (path >> (d - 1u - k)) & 1u;
          l = (open && right) ? mid : l;
          h = (open && !right) ? mid : h;
          if (!open) h = l + 1;  // the path has left the tree: nothing below it
        }
      }
    }
    const double val = node.ld(mine != kNone ? mine : 0u);
    uint32_t at = 1;
    for (int k = 0; k < 6; ++k) {
      const uint32_t mid = uint32_t(__builtin_amdgcn_readlane(int(mine), int(at)));
      if (mid == kNone) break;  // the descent has ended
      const double s = origin + readlane_f64(val, int(at));
      const bool right = __builtin_amdgcn_readfirstlane(int(!(value < s))) != 0;  // the serial code's `else` branch
      if (right) {
        lo = mid;
        origin = s;
      } else {
        hi = mid;
      }
      at = 2u * at + (right ? 1u : 0u);
    }
  }
  return hi;
}
#endif
}  // namespace wt

// ---- random stream -----------------------------------------------------------
// 32-bit Mersenne Twister with libstdc++'s uniform and normal transforms on top;
// the state lives in LDS while a chain kernel runs and is used by one lane.
struct RngState {
  uint32_t mt[624];
  int32_t idx;
  int32_t has_saved;
  double saved;
};

MG_DEV void rng_seed(RngState& r, uint32_t seed) {
  r.mt[0] = seed;
  for (int i = 1; i < 624; ++i) r.mt[i] = 1812433253u * (r.mt[i - 1] ^ (r.mt[i - 1] >> 30)) + uint32_t(i);
  r.idx = 624;
  r.has_saved = 0;
  r.saved = 0.0;
}

MG_DEVN void rng_refill(RngState& r) {
  for (int i = 0; i < 624; ++i) {
    const int i1 = (i + 1 == 624) ? 0 : i + 1;
    const int im = (i + 397 >= 624) ? i + 397 - 624 : i + 397;
    const uint32_t y = (r.mt[i] & 0x80000000u) | (r.mt[i1] & 0x7fffffffu);
    r.mt[i] = r.mt[im] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
  }
  r.idx = 0;
}

MG_DEV uint32_t rng_next(RngState& r) {
  if (r.idx >= 624) rng_refill(r);
  uint32_t y = r.mt[r.idx++];
  y ^= y >> 11;
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= y >> 18;
  return y;
}

// generate_canonical<double,53>: (lo + hi * 2^32) / 2^64, kept below 1
MG_DEV double rng_uniform(RngState& r) {
  const double lo = double(rng_next(r));
  const double hi = double(rng_next(r));
  double ret = (lo + hi * 4294967296.0) / 18446744073709551616.0;
  if (ret >= 1.0) ret = 0x1.fffffffffffffp-1;
  return ret;
}

// Marsaglia polar method; the second variate is kept for the next call
MG_DEV double rng_normal(RngState& r) {
  if (r.has_saved) {
    r.has_saved = 0;
    return r.saved;
  }
  double x, y, r2;
  do {
    x = 2.0 * rng_uniform(r) - 1.0;
    y = 2.0 * rng_uniform(r) - 1.0;
    r2 = x * x + y * y;
  } while (r2 > 1.0 || r2 == 0.0);
  const double mult = sqrt(-2 * log(r2) / r2);
  r.saved = x * mult;
  r.has_saved = 1;
  return y * mult;
}

// The stream as the serial code above draws it, behind the interface the sampler's serial part uses.
struct SerialRng {
  static constexpr bool kWave = false;
  RngState& st;
  MG_DEV double uniform() { return rng_uniform(st); }
  MG_DEV double normal() { return rng_normal(st); }
  MG_DEV void normals(double* out, int count) {  // count <= 4; written out so that `out` stays in registers
    if (count > 0) out[0] = rng_normal(st);
    if (count > 1) out[1] = rng_normal(st);
    if (count > 2) out[2] = rng_normal(st);
    if (count > 3) out[3] = rng_normal(st);
  }
  MG_DEV void flush() {}
};

#if defined(__HIP_DEVICE_COMPILE__)
// The twist of rng_refill by a whole wavefront, 64 words at a time.  Word i is made of the OLD
// words i and i + 1 and of word i + 397 (mod 624), old for i < 227 and new from there on: inside
// a block of 64 every lane reads before any lane writes (one instruction does each for all
// lanes), word i + 397 lies in a block not yet rewritten, and word i - 227 in one already done.
MG_DEV void rng_refill_wave(RngState& r) {
  const int lane = threadIdx.x & 63;
  for (int i0 = 0; i0 < 624; i0 += 64) {
    const int i = i0 + lane;
    uint32_t v = 0;
    if (i < 624) {
      const int i1 = (i + 1 == 624) ? 0 : i + 1;
      const int im = (i + 397 >= 624) ? i + 397 - 624 : i + 397;
      const uint32_t y = (r.mt[i] & 0x80000000u) | (r.mt[i1] & 0x7fffffffu);
      v = r.mt[im] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    // the lanes exchange words through LDS: to the compiler one lane's loads and another lane's
    // stores are unrelated, so the order "all loads of a block, then its stores, then the next
    // block's loads" has to be spelled out
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (i < 624) r.mt[i] = v;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  r.idx = 0;
}

// The same stream for a wavefront whose 64 lanes run the sampler's serial part together (same
// data in every lane).  A window of up to 64 tempered outputs is taken at once, one per lane, and
// handed out with v_readlane; the polar method's trials -- sequential in the reference, the first
// one inside the unit circle wins -- are evaluated side by side, one per lane, on the outputs each
// trial would have drawn, and the first accepted one is taken.  Every value is the serial code's:
// same outputs in the same order, same arithmetic per trial.
struct WaveRng {
  static constexpr bool kWave = true;
  RngState& st;
  uint32_t win = 0;          // lane L: output base + L of the current block of 624 (valid for L < n)
  int base, n = 0, pos = 0;  // the window starts at word `base`, holds n outputs, `pos` of them are used
  int has_saved;
  double saved;
  // Drawing ahead on a premise that may turn out false: mark() notes where the stream stands, restore()
  // goes back there.  Everything a draw changes is in this object except the generator's 624 words,
  // which the twist at the end of a block rewrites in place -- so while `ahead` is set the twist is not
  // done: the stream reports `ran_out` instead and hands out meaningless (but finite) numbers until the
  // caller, who must check the flag, restores the mark.
  bool ahead = false, ran_out = false;
  struct Mark {
    uint32_t win;
    int base, n, pos, has_saved;
    double saved;
  };
  MG_DEV Mark mark() const { return Mark{win, base, n, pos, has_saved, saved}; }
  MG_DEV void restore(const Mark& m) {
    win = m.win;
    base = m.base;
    n = m.n;
    pos = m.pos;
    has_saved = m.has_saved;
    saved = m.saved;
  }
  MG_DEV explicit WaveRng(RngState& s) : st(s), base(s.idx), has_saved(s.has_saved), saved(s.saved) {}
  MG_DEV void flush() {
    st.idx = base + pos;
    st.has_saved = has_saved;
    st.saved = saved;
  }
  MG_DEV void load_window() {
    int idx = base + pos;
    if (idx >= 624) {
      if (ahead) {  // the block's end: see `ahead`
        ran_out = true;
        pos = 0;
        return;
      }
      rng_refill_wave(st);
      idx = 0;
    }
    base = idx;
    pos = 0;
    n = 624 - idx < 64 ? 624 - idx : 64;
    const int lane = threadIdx.x & 63;
    uint32_t y = st.mt[idx + lane < 624 ? idx + lane : 623];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    win = y;
  }
  MG_DEV uint32_t next() {
    if (pos >= n) load_window();
    const uint32_t v = uint32_t(__builtin_amdgcn_readlane(int(win), __builtin_amdgcn_readfirstlane(pos)));
    ++pos;
    return v;
  }
  static MG_DEV double canonical(uint32_t lo32, uint32_t hi32) {  // generate_canonical<double,53>, as rng_uniform
    double ret = (double(lo32) + double(hi32) * 4294967296.0) / 18446744073709551616.0;
    if (ret >= 1.0) ret = 0x1.fffffffffffffp-1;
    return ret;
  }
  MG_DEV double uniform() {
    const uint32_t lo32 = next(), hi32 = next();
    return canonical(lo32, hi32);
  }
  MG_DEV double normal() {
    if (has_saved) {
      has_saved = 0;
      return saved;
    }
    const int lane = threadIdx.x & 63;
    // (a trial fails with probability 1 - pi/4; the bound only keeps corrupted state from spinning)
    for (int guard = 0; guard < 4096; ++guard) {
      if (pos >= n) load_window();
      if (ran_out) return 0.0;
      int trials = (n - pos) >> 2;  // whole trials (four outputs each) left in the window
      if (trials == 0) {  // the next trial straddles the window's end: drawn the serial way
        const double x = 2.0 * uniform() - 1.0;
        const double y = 2.0 * uniform() - 1.0;
        const double r2 = x * x + y * y;
        if (r2 > 1.0 || r2 == 0.0) continue;
        const double mult = sqrt(-2 * log(r2) / r2);
        saved = x * mult;
        has_saved = 1;
        return y * mult;
      }
      if (trials > 16) trials = 16;
      const int at = (pos + 4 * lane) & 63;  // lanes beyond `trials` read something and are ignored
      const uint32_t a0 = uint32_t(__shfl(int(win), at)), a1 = uint32_t(__shfl(int(win), (at + 1) & 63));
      const uint32_t b0 = uint32_t(__shfl(int(win), (at + 2) & 63)), b1 = uint32_t(__shfl(int(win), (at + 3) & 63));
      const double x = 2.0 * canonical(a0, a1) - 1.0;
      const double y = 2.0 * canonical(b0, b1) - 1.0;
      const double r2 = x * x + y * y;
      const bool ok = lane < trials && !(r2 > 1.0 || r2 == 0.0);
      const unsigned long long mask = __ballot(ok);
      if (mask == 0) {  // none of them inside the circle: all consumed
        pos += 4 * trials;
        continue;
      }
      const int first = __ffsll((long long)mask) - 1;
      const double mult = sqrt(-2 * log(r2) / r2);  // every lane for its own trial; one is read
      saved = wt::readlane_f64(x * mult, first);
      has_saved = 1;
      pos += 4 * (first + 1);
      return wt::readlane_f64(y * mult, first);
    }
    return __builtin_nan("");
  }
  // `count` (at most 4) normals in a row, the same numbers as `count` calls of normal().  The trials that
  // one call evaluates side by side are the trials the next call would evaluate: the second accepted
  // trial of the same window serves the call after the next, and so on, so that one evaluation (one
  // logarithm, root and division, in every lane for its own trial) usually serves them all.
  MG_DEV void normals(double* out, int count) {
    int got = 0;
    const int lane = threadIdx.x & 63;
    for (int guard = 0; guard < 4096 && got < count; ++guard) {
      if (has_saved) {  // the second number of the last accepted trial comes first
        has_saved = 0;
        out[got++] = saved;
        continue;
      }
      if (pos >= n) load_window();
      if (ran_out) break;
      int trials = (n - pos) >> 2;
      if (trials == 0) {  // the next trial straddles the window's end: one call the usual way
        out[got++] = normal();
        continue;
      }
      if (trials > 16) trials = 16;
      const int at = (pos + 4 * lane) & 63;
      const uint32_t a0 = uint32_t(__shfl(int(win), at)), a1 = uint32_t(__shfl(int(win), (at + 1) & 63));
      const uint32_t b0 = uint32_t(__shfl(int(win), (at + 2) & 63)), b1 = uint32_t(__shfl(int(win), (at + 3) & 63));
      const double x = 2.0 * canonical(a0, a1) - 1.0;
      const double y = 2.0 * canonical(b0, b1) - 1.0;
      const double r2 = x * x + y * y;
      const bool ok = lane < trials && !(r2 > 1.0 || r2 == 0.0);
      unsigned long long mask = __ballot(ok);
      if (mask == 0) {
        pos += 4 * trials;
        continue;
      }
      const double mult = sqrt(-2 * log(r2) / r2);
      const double from_y = y * mult, from_x = x * mult;
      int used = 0;  // last trial whose numbers were taken
      while (mask != 0 && got < count) {  // accepted trials in their order, as long as numbers are wanted
        used = __ffsll((long long)mask) - 1;
        mask &= mask - 1;
        out[got++] = wt::readlane_f64(from_y, used);
        const double second = wt::readlane_f64(from_x, used);
        if (got < count) {
          out[got++] = second;
        } else {
          saved = second;
          has_saved = 1;
        }
      }
      // enough numbers: the stream stands behind the last trial used; numbers still wanted: the window's
      // accepted trials ran out, and the rejected ones behind the last of them are used up as well
      pos += (got >= count) ? 4 * (used + 1) : 4 * trials;
    }
    for (; got < count; ++got) out[got] = 0.0;  // only after ran_out
  }
};
#endif

}  // namespace mg

#endif  // SLAM_AMD_DEVICE_MATH_HPP
