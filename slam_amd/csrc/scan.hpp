// Wavefront-level inclusive prefix scan over an arbitrary associative,
// NON-commutative operation (gfx950: 64-lane wavefronts).  Used to build the
// cumulative SE(2) poses of a chain from its relative poses: Kogge-Stone steps
// with lane shuffles inside a wavefront, wavefront totals staged through LDS,
// carries across workgroup-sized tiles applied on the left.
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

MG_DEV Pose shfl_up_any(const Pose& v, int d) {
  return Pose{__shfl_up(v.x, d), __shfl_up(v.y, d), __shfl_up(v.th, d)};
}

struct PoseOp {
  MG_DEV Pose identity() const { return pose_identity(); }
  MG_DEV Pose operator()(const Pose& a, const Pose& b) const { return compose(a, b); }
};

// x -> a + b*x (mod p); (f.g)(x) = f(g(x))  <=>  (a,b)+(c,d) = (a + b*c, b*d)
struct Affine {
  int32_t a, b;
};
MG_DEV Affine shfl_up_any(const Affine& v, int d) { return Affine{__shfl_up(v.a, d), __shfl_up(v.b, d)}; }
struct AffineOp {
  int32_t p;
  MG_DEV Affine identity() const { return Affine{0, 1}; }
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

}  // namespace mg

#endif  // SLAM_AMD_SCAN_HPP
