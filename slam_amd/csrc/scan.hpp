// Inclusive prefix scan over an arbitrary associative, NON-commutative
// operation for gfx950 (64-lane wavefronts).  Used to build the cumulative SE(2)
// poses of a chain from its relative poses.
//
// Shape: one workgroup walks a run of elements tile by tile and carries the
// composition of everything before the tile in registers (chained_scan).  Inside
// a tile every lane owns K consecutive elements: it composes them, the lane
// totals are scanned across the wavefront with lane shuffles (Kogge-Stone), the
// wavefront totals go through LDS, and the lane then composes its K elements
// once more onto its exclusive prefix and stores them.  Every element is read
// once and written once.
//
// The scan replaces the reference's Fenwick tree of poses
// (src/utility/bitree.hpp:155-321): prefix(k) below equals
// bitree::accumulate(k), up to floating-point rounding for poses and exactly
// for integer groups (which is how the structure itself is tested, with the
// affine-map group of the reference's test/bitree.cpp:36-54).
#ifndef SLAM_AMD_SCAN_HPP
#define SLAM_AMD_SCAN_HPP

#include "device_math.hpp"

namespace mg {

constexpr int kScanMaxWaves = 16;

MG_DEV Pose shfl_up_any(const Pose& v, int d) {
  return Pose{__shfl_up(v.x, d), __shfl_up(v.y, d), __shfl_up(v.th, d)};
}

struct PoseOp {
  using value_type = Pose;
  using raw_type = Pose;
  MG_DEV Pose identity() const { return pose_identity(); }
  MG_DEV Pose lift(const Pose& r) const { return r; }
  MG_DEV Pose operator()(const Pose& a, const Pose& b) const { return compose(a, b); }
};

// A pose that carries the cosine and sine of its angle, so that composing two of them is a
// handful of fused multiply-adds instead of a sincos: the scan takes ONE sincos per input pose
// (lift) and none afterwards.  The angle itself stays the plain, unwrapped sum the reference
// keeps (src/planar_robot/pose.hpp:67-71); (c, s) are products of rotations and differ from
// cos/sin of that sum by a few ulp per composition level.
struct PoseCS {
  double x, y, c, s, th;
};
MG_DEV PoseCS shfl_up_any(const PoseCS& v, int d) {
  return PoseCS{__shfl_up(v.x, d), __shfl_up(v.y, d), __shfl_up(v.c, d), __shfl_up(v.s, d), __shfl_up(v.th, d)};
}
// sin and cos of an angle of at most pi/4 in size -- every relative pose of a robot that does not
// spin on the spot -- by the classic minimax polynomials on that interval (Sun's fdlibm kernels, no
// argument reduction needed, below one ulp); anything larger goes to the library's sincos.  The
// library routine spends most of its instructions on the reduction and on selecting the quadrant.
MG_DEV void sincos_of_turn(double x, double* sn, double* cs) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MG_SCAN_LIBM_SINCOS)
  if (fabs(x) <= 0.78539816339744830962) {
    const double z = x * x;
    const double rs = fma(z, fma(z, fma(z, fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08),
                                        2.75573137070700676789e-06), -1.98412698298579493134e-04), 8.33333333332248946124e-03);
    *sn = fma(z * x, fma(z, rs, -1.66666666666666324348e-01), x);
    const double rc = z * fma(z, fma(z, fma(z, fma(z, fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09),
                                                   -2.75573143513906633035e-07), 2.48015872894767294178e-05),
                                     -1.38888888888741095749e-03), 4.16666666666666019037e-02);
    const double hz = 0.5 * z, w = 1.0 - hz;
    *cs = w + (((1.0 - w) - hz) + z * rc);
    return;
  }
#endif
  sincos(x, sn, cs);
}

struct PoseCSOp {
  using value_type = PoseCS;
  using raw_type = Pose;
  MG_DEV PoseCS identity() const { return PoseCS{0.0, 0.0, 1.0, 0.0, 0.0}; }
  MG_DEV PoseCS lift(const Pose& r) const {
    double s, c;
    sincos_of_turn(r.th, &s, &c);
    return PoseCS{r.x, r.y, c, s, r.th};
  }
  // a then b:  t = a.t + R(a) b.t,  R = R(a) R(b),  th = a.th + b.th
  MG_DEV PoseCS operator()(const PoseCS& a, const PoseCS& b) const {
    return PoseCS{fma(a.c, b.x, fma(-a.s, b.y, a.x)), fma(a.s, b.x, fma(a.c, b.y, a.y)), fma(a.c, b.c, -(a.s * b.s)),
                  fma(a.s, b.c, a.c * b.s), a.th + b.th};
  }
};

// x -> a + b*x (mod p); (f.g)(x) = f(g(x))  <=>  (a,b)+(c,d) = (a + b*c, b*d)
struct Affine {
  int32_t a, b;
};
MG_DEV Affine shfl_up_any(const Affine& v, int d) { return Affine{__shfl_up(v.a, d), __shfl_up(v.b, d)}; }
struct AffineOp {
  using value_type = Affine;
  using raw_type = Affine;
  int32_t p;
  MG_DEV Affine identity() const { return Affine{0, 1}; }
  MG_DEV Affine lift(const Affine& r) const { return r; }
  MG_DEV Affine operator()(const Affine& f, const Affine& g) const {
    const long long a = (long long)f.a + (long long)f.b * g.a;
    const long long b = (long long)f.b * g.b;
    return Affine{int32_t(a % p), int32_t(b % p)};
  }
};

// Inclusive scan across the 64 lanes of a wavefront; lane i ends with
// v[0] op v[1] op ... op v[i].
template <class T, class Op>
MG_DEV T wave_scan_inclusive(T v, const Op& op) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const T left = shfl_up_any(v, d);
    if (lane >= d) v = op(left, v);
  }
  return v;
}

// Inclusive scan across a workgroup (blockDim.x a multiple of 64, <= 1024).
// `stage` is LDS room for one T per wavefront.  On return `total` (same in all
// threads) is the composition of the whole tile.
template <class T, class Op>
MG_DEV T block_scan_inclusive(T v, const Op& op, T* stage, T& total) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int nwaves = blockDim.x >> 6;
  v = wave_scan_inclusive(v, op);
  if (lane == 63) stage[wave] = v;
  __syncthreads();
  T before = op.identity();
  T all = op.identity();
  for (int w = 0; w < nwaves; ++w) {
    if (w == wave) before = all;
    all = op(all, stage[w]);
  }
  total = all;
  if (wave > 0) v = op(before, v);
  __syncthreads();
  return v;
}

// Elements [lo, hi) scanned by ONE workgroup, tile by tile (a tile = blockDim.x * K elements),
// `carry` = the composition of everything before `lo`.  load(i) fetches element i in its stored
// form, op.lift turns it into the scanned type, store(i, v) receives carry o e[lo] o ... o e[i].
// Returns carry o e[lo] o ... o e[hi-1] in every thread.  The raw elements of the next TWO tiles are
// in flight while the current one is composed: with a thousand chains being rebuilt at once the
// memory system's queues make a load take several microseconds, more than one tile's arithmetic
// (measured on 1024 chains x 100 000 poses: one tile ahead 4.15 TB/s, two tiles 4.45 TB/s).
// `stage` is LDS room for 2 x kScanMaxWaves values (two tiles alternate, one barrier per tile).
// All threads of the workgroup must call it with the same lo, hi and carry.
template <int K, class Raw, class Load>
MG_DEV auto row_by_elements(Load load) {
  return [load](uint32_t first, Raw* out) {
#pragma unroll
    for (int k = 0; k < K; ++k) out[k] = load(first + k);
  };
}
// load_row(i, out) fetches the K elements i .. i + K - 1 of one lane at once (all of them exist); a
// caller whose elements lie in component arrays turns that into a few wide loads.
template <int K, class Op, class Load, class LoadRow, class Store>
MG_DEV typename Op::value_type chained_scan(const Op& op, uint32_t lo, uint32_t hi, typename Op::value_type carry,
                                            Load load, LoadRow load_row, Store store,
                                            typename Op::value_type (*stage)[kScanMaxWaves]) {
  using T = typename Op::value_type;
  using Raw = typename Op::raw_type;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const uint32_t tile = blockDim.x * K;
  Raw next[K], next2[K];  // the raw elements of the next tile and of the one after it
  auto fetch_into = [&](uint32_t first, Raw* into) {
    if (first + K <= hi) {
      load_row(first, into);
    } else {
#pragma unroll
      for (int k = 0; k < K; ++k)
        if (first + k < hi) into[k] = load(first + k);
    }
  };
#pragma unroll
  for (int k = 0; k < K; ++k) next[k] = Raw{};
  fetch_into(lo + threadIdx.x * K, next);
#pragma unroll
  for (int k = 0; k < K; ++k) next2[k] = Raw{};
  fetch_into(lo + tile + threadIdx.x * K, next2);
  int buf = 0;
  for (uint32_t base = lo; base < hi; base += tile, buf ^= 1) {
    const uint32_t first = base + threadIdx.x * K;
    Raw raw[K];
#pragma unroll
    for (int k = 0; k < K; ++k) raw[k] = next[k];
#pragma unroll
    for (int k = 0; k < K; ++k) next[k] = next2[k];
    fetch_into(first + 2 * tile, next2);
    T v[K];
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = first + k < hi ? op.lift(raw[k]) : op.identity();
    T mine = v[0];
#pragma unroll
    for (int k = 1; k < K; ++k) mine = op(mine, v[k]);
    const T inc = wave_scan_inclusive(mine, op);
    if (lane == 63) stage[buf][wave] = inc;
    T excl = shfl_up_any(inc, 1);
    if (lane == 0) excl = op.identity();
    __syncthreads();
    T before = carry, all = carry;
    for (int w = 0; w < nwaves; ++w) {
      if (w == wave) before = all;
      all = op(all, stage[buf][w]);
    }
    carry = all;
    T acc = op(before, excl);
#pragma unroll
    for (int k = 0; k < K; ++k) {
      acc = op(acc, v[k]);
      if (first + k < hi) store(first + k, acc);
    }
  }
  return carry;
}

}  // namespace mg

#endif  // SLAM_AMD_SCAN_HPP
