// Stand-alone probe (not part of the library): how fast can the reading stream run when it is a kernel
// of its own -- no driver code in the same kernel, work split evenly over the grid instead of one
// workgroup per chain?  It evaluates, for CHAINS chains, the sum over prepared landmark segments of
// (ro - rn)(ro + rn - 2 mu) k  with ro = |w_old - t_s|, rn = |w_new - t_s|, exactly the arithmetic of
// RatioStream in slam_amd/csrc/chain_kernels.hpp, on synthetic data of the benchmark's shape (100 000 poses,
// 1 000 landmarks x 1 000 readings, runs of consecutive timesteps), and prints readings per second.
//
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -o tools/stream_probe tools/stream_probe.hip
//   tools/stream_probe [chains=1024] [slices per chain=8] [repeats=5]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

struct alignas(16) PoseXY {
  double x, y;
};
struct alignas(16) ObsRec {
  double mu;
  uint32_t t, f;
};

#define CHECK(e)                                                                      \
  do {                                                                                \
    hipError_t e_ = (e);                                                              \
    if (e_ != hipSuccess) {                                                           \
      std::fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(e_));                    \
      std::exit(1);                                                                   \
    }                                                                                 \
  } while (0)

__device__ __forceinline__ double range_of(double dx, double dy) {
  const double x = fma(dx, dx, fma(dy, dy, 1e-280));
  const double y = __builtin_amdgcn_rsq(x);
  double s = x * y, h = 0.5 * y;
  double r = fma(-h, s, 0.5);
  s = fma(s, r, s);
  h = fma(h, r, h);
  r = fma(-h, s, 0.5);
  return fma(s, r, s);
}

// grid = (slices, chains); a workgroup takes the flat reading indices [lo, hi) of its chain.
// segpre[nseg+1], seglo[nseg] are the same for every chain (the split of a landmark's list at a
// timestep depends on the data only); segw[chain][nseg][4] holds the old and new landmark position.
// WINDOW: 0 = as is; 1 = every pose gather lands in a 16 KB window; 2 = every record load lands in a 64 KB
// window (diagnostic: which of the two streams, if any, the rate depends on)
template <int U, int WINDOW = 0>
__global__ void __launch_bounds__(256) probe_kernel(const ObsRec* __restrict__ orec, const PoseXY* __restrict__ pxy,
                                                    size_t pose_stride, const uint32_t* __restrict__ segpre,
                                                    const uint32_t* __restrict__ seglo, const double* __restrict__ segw,
                                                    uint32_t nseg, double k, double* __restrict__ out) {
  __shared__ double red[4];
  const uint32_t chain = blockIdx.y, slices = gridDim.x;
  const uint32_t total = segpre[nseg];
  const uint32_t per = (total + slices - 1) / slices;
  const uint32_t lo = blockIdx.x * per, hi = min(total, lo + per);
  const PoseXY* __restrict__ poses = pxy + size_t(chain) * pose_stride;
  const double* __restrict__ w4 = segw + size_t(chain) * nseg * 4;
  // segment of the first index of this lane: binary search in the prefix sums
  uint32_t cur = 0;
  {
    uint32_t a = 0, b = nseg;
    const uint32_t i = min(lo + threadIdx.x, total - 1);
    while (a + 1 < b) {
      const uint32_t m = (a + b) >> 1;
      if (segpre[m] <= i)
        a = m;
      else
        b = m;
    }
    cur = a;
  }
  uint32_t seg_end = segpre[cur + 1], seg_base = seglo[cur] - segpre[cur];
  double acc = 0.0;
  constexpr uint32_t stride = 256;
  ObsRec rec[U];
  uint32_t sg[U];
  auto fetch = [&](uint32_t i0) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t i = i0 + u * stride;
      if (i < hi) {
        while (i >= seg_end) {
          ++cur;
          seg_end = segpre[cur + 1];
          seg_base = seglo[cur] - segpre[cur];
        }
        rec[u] = orec[WINDOW == 2 ? ((seg_base + i) & 4095u) : (seg_base + i)];
      } else {
        rec[u] = orec[seglo[0]];
      }
      sg[u] = cur;
    }
  };
  fetch(lo + threadIdx.x);
  for (uint32_t i0 = lo + threadIdx.x; i0 < hi; i0 += U * stride) {
    PoseXY p[U];
    double mu[U];
    uint32_t s[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      p[u] = poses[WINDOW == 1 ? (rec[u].t & 1023u) : rec[u].t];
      mu[u] = rec[u].mu;
      s[u] = sg[u];
    }
    if (i0 + U * stride < hi) fetch(i0 + U * stride);  // the next group's records fly during this group's math
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const double* w = w4 + size_t(s[u]) * 4;
      const double ro = range_of(w[0] - p[u].x, w[1] - p[u].y);
      const double rn = range_of(w[2] - p[u].x, w[3] - p[u].y);
      const double term = (ro - rn) * fma(-2.0, mu[u], ro + rn) * k;
      if (i0 + u * stride < hi) acc += term;
    }
  }
#pragma unroll
  for (int dd = 32; dd > 0; dd >>= 1) acc += __shfl_down(acc, dd);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[size_t(chain) * slices + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

int main(int argc, char** argv) {
  const uint32_t chains = argc > 1 ? std::atoi(argv[1]) : 1024;
  const uint32_t slices = argc > 2 ? std::atoi(argv[2]) : 8;
  const int repeats = argc > 3 ? std::atoi(argv[3]) : 5;
  const uint32_t T = 100000, F = 1000, per_landmark = 1000;
  std::mt19937 gen(7);
  // readings: landmark-major, time-sorted, in runs of consecutive timesteps (visits)
  std::vector<ObsRec> orec;
  std::vector<uint32_t> foff(F + 1, 0);
  for (uint32_t f = 0; f < F; ++f) {
    uint32_t t = gen() % 600, left = per_landmark;
    while (left) {
      uint32_t run = 40 + gen() % 60;
      if (run > left) run = left;
      for (uint32_t j = 0; j < run && t < T; ++j, ++t) orec.push_back(ObsRec{10.0 + (gen() % 1000) * 0.01, t, f});
      left -= run;
      t += 900 + gen() % 400;  // gone until the next lap
      if (t >= T) break;
    }
    foff[f + 1] = uint32_t(orec.size());
  }
  // one proposal: every landmark contributes a prefix or a suffix of its list
  std::vector<uint32_t> segpre(F + 1, 0), seglo(F);
  for (uint32_t f = 0; f < F; ++f) {
    const uint32_t n = foff[f + 1] - foff[f], cut = gen() % (n + 1);
    const bool prefix = gen() & 1;
    seglo[f] = prefix ? foff[f] : foff[f] + cut;
    segpre[f + 1] = segpre[f] + (prefix ? cut : n - cut) * 2 / 5;  // about 20 % of all readings, as in the benchmark
  }
  const uint32_t total = segpre[F];
  std::vector<double> segw(size_t(chains) * F * 4);
  for (auto& v : segw) v = (gen() % 100000) * 0.003;
  std::vector<PoseXY> poses(size_t(T) + 1);
  for (auto& p : poses) p = PoseXY{(gen() % 100000) * 0.003, (gen() % 100000) * 0.003};

  ObsRec* d_orec;
  PoseXY* d_pxy;
  uint32_t *d_segpre, *d_seglo;
  double *d_segw, *d_out;
  CHECK(hipMalloc(&d_orec, orec.size() * sizeof(ObsRec)));
  CHECK(hipMalloc(&d_pxy, size_t(chains) * (T + 1) * sizeof(PoseXY)));
  CHECK(hipMalloc(&d_segpre, segpre.size() * 4));
  CHECK(hipMalloc(&d_seglo, seglo.size() * 4));
  CHECK(hipMalloc(&d_segw, segw.size() * 8));
  CHECK(hipMalloc(&d_out, size_t(chains) * slices * 8));
  CHECK(hipMemcpy(d_orec, orec.data(), orec.size() * sizeof(ObsRec), hipMemcpyHostToDevice));
  for (uint32_t c = 0; c < chains; ++c)  // every chain has a pose table of its own in HBM
    CHECK(hipMemcpy(d_pxy + size_t(c) * (T + 1), poses.data(), poses.size() * sizeof(PoseXY), hipMemcpyHostToDevice));
  CHECK(hipMemcpy(d_segpre, segpre.data(), segpre.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(d_seglo, seglo.data(), seglo.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(d_segw, segw.data(), segw.size() * 8, hipMemcpyHostToDevice));
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  auto run = [&](auto kernel, const char* name) {
    float best = 1e30f;
    for (int r = 0; r < repeats + 1; ++r) {
      CHECK(hipEventRecord(a));
      hipLaunchKernelGGL(kernel, dim3(slices, chains), dim3(256), 0, nullptr, d_orec, d_pxy, size_t(T) + 1, d_segpre,
                         d_seglo, d_segw, F, 0.5, d_out);
      CHECK(hipGetLastError());
      CHECK(hipEventRecord(b));
      CHECK(hipEventSynchronize(b));
      float ms;
      CHECK(hipEventElapsedTime(&ms, a, b));
      if (r > 0 && ms < best) best = ms;
    }
    std::printf("%s: %u chains x %u readings, %u slices: %.3f ms -> %.3e readings/s\n", name, chains, total, slices, best,
                double(chains) * total / (best * 1e-3));
  };
  run(probe_kernel<2>, "U=2");
  run(probe_kernel<4>, "U=4");
  run(probe_kernel<8>, "U=8");
  run(probe_kernel<2, 1>, "U=2, poses from a 16 KB window");
  run(probe_kernel<2, 2>, "U=2, records from a 64 KB window");
  return 0;
}
